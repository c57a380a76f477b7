set -o pipefail
mkdir -p gpurun_out/r3e
python -m pytest tests/test_imageops_gpu.py tests/test_psp_gpu.py -x -q -m gpu -s > gpurun_out/r3e/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/r3e/tests.log
[ $rc -eq 0 ] || exit 1
bash tools/gpu_ecc_variants.sh "0 4 30" | grep "variant\|ecc_cols\|gauss_fused_kernel<u\|solve"
