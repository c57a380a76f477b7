python tools/exp_occupancy.py 2>&1 | grep -v amdgpu.ids
for pf in 0 1 0 1; do UPSP_TRAV_PREFETCH=$pf python bench.py --serial --steps 10 --warmup 3 --no-cpu-baseline --no-reraycast 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('prefetch $pf serial step', round(d['ms_per_step'],4), {n:round(k[n]['ms_per_step'],4) for n in k if 'projection_kernel' in n or 'witness' in n})"
done
