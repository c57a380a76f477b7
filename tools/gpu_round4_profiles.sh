# Round-4 profile files, one gpurun call each part (run from the repository root on the GPU box):
#   bash tools/gpu_round4_profiles.sh a    default bench: kernel stats, FETCH / WRITE summary, the line (with CPU baseline + parity)
#   bash tools/gpu_round4_profiles.sh b    registration: kernel stats, ECC traffic summary, lines (1000 and 10 000 frames), ECC counters
#   bash tools/gpu_round4_profiles.sh c    N > 1 loop on one GPU (one-rank RCCL): default chunks, config3 share; multi-camera line; projection counters
# Files land in gpurun_out/r4p/ ; copy the ones to track into profiles/ as r04_*.
set -o pipefail
part=${1:-a}
o=gpurun_out/r4p; mkdir -p $o
if [ $part = a ]; then
  bash tools/profile_bench.sh r04 || exit 1
  cp gpurun_out/prof_r04/kernel_stats.csv $o/r04_bench_kernel_stats.csv
  cp gpurun_out/prof_r04/summary.json $o/r04_bench_summary.json
  cp gpurun_out/prof_r04/bench_line.json $o/r04_bench_line.json
  python3 bench.py --serial --no-cpu-baseline > $o/r04_bench_line_serial.json 2> $o/serial.err || exit 1
elif [ $part = b ]; then
  BENCH_TIMEOUT=600 bash tools/profile_bench.sh r04_reg --registration || exit 1
  cp gpurun_out/prof_r04_reg/kernel_stats.csv $o/r04_registration_kernel_stats.csv
  cp gpurun_out/prof_r04_reg/summary.json $o/r04_ecc_summary.json
  cp gpurun_out/prof_r04_reg/bench_line.json $o/r04_bench_line_registration.json
  timeout -k 10 500 python3 bench.py --registration --frames 10000 --steps 2 --warmup 1 --no-cpu-baseline > $o/r04_bench_line_config2_10000_frames.json 2> $o/reg10k.err || exit 1
  bash tools/pmc_script.sh "ecc_cols|gauss5_quad|ecc_solve|reblur|hot_repair|warp_compact" tools/prof_ecc.py > $o/r04_ecc_pmc.txt 2>&1 || exit 1
else
  # N > 1 loop on one GPU through a one-rank RCCL communicator: finished inside the step (with the oracle check), deferred
  # (the N > 1 default), and round 3's schedule (pass A per chunk) both ways
  UPSP_FORCE_COLLECTIVES=1 timeout -k 10 500 python3 bench.py --force-chunked > $o/r04_bench_line_chunked_rccl.json 2> $o/ck.err || exit 1
  UPSP_FORCE_COLLECTIVES=1 timeout -k 10 500 python3 bench.py --force-chunked --defer-exchange --no-cpu-baseline > $o/r04_bench_line_chunked_rccl_deferred.json 2>> $o/ck.err || exit 1
  UPSP_FORCE_COLLECTIVES=1 timeout -k 10 500 python3 bench.py --force-chunked --chunk-scan --no-cpu-baseline > $o/r04_bench_line_chunked_rccl_chunk_scan.json 2>> $o/ck.err || exit 1
  UPSP_FORCE_COLLECTIVES=1 timeout -k 10 500 python3 bench.py --force-chunked --chunk-scan --defer-exchange --no-cpu-baseline > $o/r04_bench_line_chunked_rccl_chunk_scan_deferred.json 2>> $o/ck.err || exit 1
  timeout -k 10 500 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 > $o/r04_bench_line_10_steps.json 2>> $o/ck.err || exit 1
  UPSP_FORCE_COLLECTIVES=1 timeout -k 10 600 python3 bench.py --config3-share --steps 3 --warmup 2 --no-cpu-baseline > $o/r04_bench_line_config3_share.json 2>> $o/ck.err || exit 1
  timeout -k 10 600 python3 bench.py --cameras 4 --model 5m --steps 3 --warmup 1 > $o/r04_multi_bench_line.json 2> $o/multi.err || exit 1
  bash tools/pmc_script.sh "projection_kernel|witness_kernel|heavy_kernel" tools/prof_proj.py > $o/r04_proj_pmc.txt 2>&1 || exit 1
fi
ls -la $o
