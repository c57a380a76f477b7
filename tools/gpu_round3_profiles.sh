# Round 3: every tracked file under profiles/r03_* comes from these two calls (bash tools/gpu_round3_profiles.sh a|b).
set -o pipefail
part=${1:-a}
mkdir -p gpurun_out/prof_r03 gpurun_out/prof_r03_multi
if [ $part = a ]; then
  bash tools/profile_bench.sh r03 2>&1 | tail -6
  timeout -k 10 400 python3 bench.py --registration > gpurun_out/prof_r03/bench_line_registration.json 2> gpurun_out/prof_r03/bench_line_registration.err; echo "registration rc=$?"
  timeout -k 10 500 python3 bench.py --registration --frames 10000 --steps 2 --warmup 1 > gpurun_out/prof_r03/bench_line_config2_10000_frames.json 2> gpurun_out/prof_r03/bench_line_config2.err; echo "configs[2] rc=$?"
  UPSP_FORCE_COLLECTIVES=1 timeout -k 10 400 python3 bench.py --force-chunked > gpurun_out/prof_r03/bench_line_chunked_rccl.json 2> gpurun_out/prof_r03/chunked.err; echo "chunked (pixel series, rccl) rc=$?"
  UPSP_FORCE_COLLECTIVES=1 timeout -k 10 300 python3 bench.py --force-chunked --wire12 --no-cpu-baseline --no-reraycast > gpurun_out/prof_r03/bench_line_chunked_rccl_12bit.json 2> gpurun_out/prof_r03/chunked12.err; echo "chunked 12 rc=$?"
  UPSP_FORCE_COLLECTIVES=1 timeout -k 10 300 python3 bench.py --force-chunked --row-wire --no-cpu-baseline --no-reraycast > gpurun_out/prof_r03/bench_line_chunked_rccl_rows.json 2> gpurun_out/prof_r03/chunkedrows.err; echo "chunked rows rc=$?"
  bash tools/pmc_script.sh "ecc_cols|gauss_fused|gauss5_quad|ecc_solve|warp_compact|node_rows|hot_scan" tools/prof_ecc.py > gpurun_out/prof_r03/ecc_pmc.txt 2>&1; echo "ecc pmc rc=$?"
else
  PROFILE_TIMEOUT=500 BENCH_TIMEOUT=700 bash tools/profile_bench.sh r03_multi --cameras 4 --model 5m --steps 3 --warmup 1 2>&1 | tail -6
  PMC_TIMEOUT=400 bash tools/pmc_script.sh "node_rows_multi|scan_compact" tools/prof_multi.py > gpurun_out/prof_r03_multi/multi_pmc.txt 2>&1; echo "multi pmc rc=$?"
  bash tools/pmc_script.sh "projection_kernel|witness_kernel|cast_kernel|heavy" tools/prof_proj.py > gpurun_out/prof_r03/proj_pmc.txt 2>&1; echo "proj pmc rc=$?"
  bash tools/pmc_script.sh "projection_kernel|witness_kernel|heavy" tools/prof_proj.py ref > gpurun_out/prof_r03/proj_ref_order_pmc.txt 2>&1; echo "proj ref pmc rc=$?"
fi
