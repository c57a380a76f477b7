# randomised ECC parity soak (tests/debug/soak_ecc.py), default 300 s: the shipped path (column-walk kernel, general
# iteration with LDS-staged taps), then -- second argument, seconds -- the same cases with direct-load taps
# (UPSP_ECC_DIRECT=1); output -> gpurun_out/soak_ecc*.log
set -o pipefail
mkdir -p gpurun_out
T=${1:-300}
timeout -k 10 $(( T + 200 )) python3 tests/debug/soak_ecc.py $T > gpurun_out/soak_ecc.log 2>&1; echo "soak_ecc rc=$?"
tail -5 gpurun_out/soak_ecc.log
if [ -n "$2" ]; then
  UPSP_ECC_DIRECT=1 timeout -k 10 $(( $2 + 200 )) python3 tests/debug/soak_ecc.py $2 > gpurun_out/soak_ecc_direct.log 2>&1; echo "soak_ecc direct rc=$?"
  tail -5 gpurun_out/soak_ecc_direct.log
fi
