# the bench's default (pass A on a second stream beside the ray casting) against --serial, the driver's command, alternating in one call
for i in 1 2 3; do for m in "" "--serial"; do
  timeout -k 10 600 python3 bench.py $m > gpurun_out/ov.json 2> gpurun_out/ov.err || { tail -2 gpurun_out/ov.err; continue; }
  python3 - "$m" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/ov.json").read().strip().splitlines()[-1]); k=d["kernels"]
print("%-10s %7.0f frames/s  step %.3f ms  build %.3f loop %.3f  A %.3f B %.3f primary %.3f retry %.3f  roofline %s %.3f  parity %s" % (sys.argv[1] or "overlapped", d["value"], d["ms_per_step"], d["breakdown_ms"]["projection_build"], d["breakdown_ms"]["frame_loop"], k["scan_compact_kernel"]["ms_per_step"], k["node_rows_kernel"]["ms_per_step"], k["projection_kernel<primary>"]["ms_per_step"], k["projection_kernel<retry>"]["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d.get("parity_checked")))
PY
done; done
