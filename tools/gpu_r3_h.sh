set -o pipefail
for fv in "1 0 0" "2 0 0" "0 0 40" "0 0 0"; do
  set -- $fv
  echo "== UPSP_ECC_FUSED=$1 UPSP_GAUSS5_VARIANT=$2 CVARIANT=$3"
  UPSP_ECC_FUSED=$1 UPSP_GAUSS5_VARIANT=$2 bash tools/gpu_ecc_variants.sh "$3" | grep "variant\|gauss\|true"
done
