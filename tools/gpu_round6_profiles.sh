# Round-6 profile files, one gpurun call each part (run from the repository root on the GPU box):
#   bash tools/gpu_round6_profiles.sh a    default bench: kernel stats, FETCH / WRITE summary, the line with the driver's command
#                                          (CPU baseline, parity, configs2 at 10 000 frames), --serial
#   bash tools/gpu_round6_profiles.sh b    registration: kernel stats, ECC traffic summary, the line, ECC counters
#   bash tools/gpu_round6_profiles.sh c    N > 1 loop on one GPU (one-rank RCCL; own block through RCCL = the rehearsal, and in place),
#                                          config3 share, multi-camera line, projection counters, pass-B probe
# Files land in gpurun_out/r6p/ ; copy the ones to track into profiles/ as r06_*.
set -o pipefail
part=${1:-a}
o=gpurun_out/r6p; mkdir -p $o
if [ $part = a ]; then
  bash tools/profile_bench.sh r06 || exit 1
  cp gpurun_out/prof_r06/kernel_stats.csv $o/r06_bench_kernel_stats.csv
  cp gpurun_out/prof_r06/summary.json $o/r06_bench_summary.json
  UPSP_BENCH_TRAFFIC_JSON=gpurun_out/prof_r06/summary.json timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $o/r06_bench_line_driver_command.json 2> $o/driver.err || exit 1
  cp gpurun_out/prof_r06/bench_line.json $o/r06_bench_line.json
  python3 bench.py --serial --no-cpu-baseline > $o/r06_bench_line_serial.json 2> $o/serial.err || exit 1
elif [ $part = b ]; then
  BENCH_TIMEOUT=600 bash tools/profile_bench.sh r06_reg --registration || exit 1
  cp gpurun_out/prof_r06_reg/kernel_stats.csv $o/r06_registration_kernel_stats.csv
  cp gpurun_out/prof_r06_reg/summary.json $o/r06_ecc_summary.json
  cp gpurun_out/prof_r06_reg/bench_line.json $o/r06_bench_line_registration.json
  bash tools/pmc_script.sh "ecc_cols|ecc_blur_ident|gauss5_quad|ecc_solve|reblur|hot_repair|warp_compact" tools/prof_ecc.py > $o/r06_ecc_pmc.txt 2>&1 || exit 1
else
  # N > 1 loop on one GPU through a one-rank RCCL communicator.  "selfrccl": the rank's own block goes through ncclSend / ncclRecv to
  # self (RCCL's kernel on the device beside the frame loop, as between GPUs); without it the block is read in place.
  R="UPSP_FORCE_COLLECTIVES=1 UPSP_EXCHANGE_SELF_RCCL=1"
  env $R timeout -k 10 500 python3 bench.py --force-chunked --defer-exchange --no-cpu-baseline --steps 10 --warmup 3 > $o/r06_bench_line_chunked_rccl_deferred.json 2> $o/ck.err || exit 1
  env $R timeout -k 10 500 python3 bench.py --force-chunked > $o/r06_bench_line_chunked_rccl.json 2>> $o/ck.err || exit 1
  UPSP_FORCE_COLLECTIVES=1 timeout -k 10 500 python3 bench.py --force-chunked --defer-exchange --no-cpu-baseline --steps 10 --warmup 3 > $o/r06_bench_line_chunked_inplace_deferred.json 2>> $o/ck.err || exit 1
  UPSP_FORCE_COLLECTIVES=1 timeout -k 10 500 python3 bench.py --force-chunked --no-cpu-baseline --steps 10 --warmup 3 > $o/r06_bench_line_chunked_inplace.json 2>> $o/ck.err || exit 1
  timeout -k 10 500 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 > $o/r06_bench_line_10_steps.json 2>> $o/ck.err || exit 1
  env $R timeout -k 10 600 python3 bench.py --config3-share --steps 3 --warmup 2 --no-cpu-baseline > $o/r06_bench_line_config3_share.json 2>> $o/ck.err || exit 1
  timeout -k 10 600 python3 bench.py --cameras 4 --model 5m --steps 3 --warmup 1 > $o/r06_multi_bench_line.json 2> $o/multi.err || exit 1
  python3 tools/passb_probe.py > $o/r06_passb_probe.txt 2>&1 || exit 1
  bash tools/pmc_script.sh "projection_kernel|witness_kernel|heavy_kernel" tools/prof_proj.py > $o/r06_proj_pmc.txt 2>&1 || exit 1
fi
ls -la $o
