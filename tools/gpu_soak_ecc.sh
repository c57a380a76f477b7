# randomised ECC parity soak (tests/debug/soak_ecc.py), default 300 s per kernel: round 3's column kernel, then round 2's
# all-double kernel (UPSP_ECC_KERNEL=2) on the same cases; output -> gpurun_out/soak_ecc*.log
set -o pipefail
mkdir -p gpurun_out
T=${1:-300}
timeout -k 10 $(( T + 200 )) python3 tests/debug/soak_ecc.py $T > gpurun_out/soak_ecc.log 2>&1; echo "soak_ecc rc=$?"
tail -5 gpurun_out/soak_ecc.log
if [ -n "$2" ]; then
  UPSP_ECC_KERNEL=2 timeout -k 10 $(( $2 + 200 )) python3 tests/debug/soak_ecc.py $2 > gpurun_out/soak_ecc_kernel2.log 2>&1; echo "soak_ecc kernel 2 rc=$?"
  tail -5 gpurun_out/soak_ecc_kernel2.log
fi
