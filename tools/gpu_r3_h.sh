set -o pipefail
mkdir -p gpurun_out/r3h
python -m pytest tests/test_imageops_gpu.py tests/test_psp_gpu.py -x -q -m gpu > gpurun_out/r3h/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r3h/tests.log
[ $rc -eq 0 ] || exit 1
bash tools/gpu_ecc_variants.sh "0" | grep "variant"
