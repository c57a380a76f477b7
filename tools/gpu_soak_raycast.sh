# Randomised ray-cast soak (tests/debug/soak_raycast.py), three ways: default thresholds; hand-off forced on most rays (wave walk);
# hand-off forced with small stacks (wave walk -> workgroup walk -> one-thread fallback); + the dense soup 10 000 x.
# usage (GPU box, repository root): bash tools/gpu_soak_raycast.sh [seconds per leg] > profiles/rNN_soak.txt
t=${1:-100}
echo "# randomised ray-cast soak (tests/debug/soak_raycast.py), $t s per leg"
echo "# SOAK_SEED=510000, default thresholds:"
SOAK_SEED=510000 python tests/debug/soak_raycast.py $t 2>&1 | tail -1
echo "# UPSP_HEAVY_STEPS=6 UPSP_HEAVY_STEPS_CAST=6 SOAK_SEED=520000 (hand-off forced on most rays: one wave per ray):"
UPSP_HEAVY_STEPS=6 UPSP_HEAVY_STEPS_CAST=6 SOAK_SEED=520000 python tests/debug/soak_raycast.py $t 2>&1 | tail -1
echo "# UPSP_HEAVY_STEPS=20 UPSP_HEAVY_STACK=128 UPSP_HEAVY_STEPS_CAST=20 SOAK_SEED=530000 (small stacks: wave walk -> workgroup walk -> one-thread fallback):"
UPSP_HEAVY_STEPS=20 UPSP_HEAVY_STACK=128 UPSP_HEAVY_STEPS_CAST=20 SOAK_SEED=530000 python tests/debug/soak_raycast.py $t 2>&1 | tail -1
echo "# tests/debug/repeat_heavy.py:"
python tests/debug/repeat_heavy.py 2>&1 | tail -1
