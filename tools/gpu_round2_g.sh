python -m pytest tests/test_imageops_gpu.py tests/test_psp_gpu.py -x -q -m gpu 2>&1 | tail -3
run() {
  timeout -k 10 300 python bench.py --registration --frames 256 --no-cpu-baseline --no-reraycast --steps 3 --warmup 1 > gpurun_out/b_reg.log 2>gpurun_out/b_reg.err; echo "rc=$?"
  python - "$1" <<'PY'
import json, sys
d=json.loads(open("gpurun_out/b_reg.log").read().strip().splitlines()[-1])
print("%-10s fps %.0f step %.2f ms iters/frame %.2f" % (sys.argv[1], d["value"], d["ms_per_step"], d.get("ecc_iterations_per_frame", -1)), {n:(round(v["ms_per_step"],3), round(v["avg_launch_ms"]*1e3,1), v.get("achieved_GBps") and round(v["achieved_GBps"])) for n,v in d["kernels"].items() if n in ("ecc_sums_kernel","ecc_solve_kernel","warp_u16_kernel","gauss_pass_kernels")})
print(d["roofline"]["kernel"], round(d["roofline"]["frac"], 3))
PY
}
run list
