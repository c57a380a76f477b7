for cap in 1 2 3 4 6 8; do UPSP_DESC_CAP=$cap UPSP_DESC_CAP2=$cap python bench.py --serial --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('cap $cap: build alone', round(d['breakdown_ms']['projection_build'],4), {n:round(k[n]['ms_per_step'],4) for n in k if 'projection_kernel' in n}, 'pixel rays tunnel', round(d['pixel_rays']['ms'],4))"
done
