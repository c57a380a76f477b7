for k in 4 2 1 8; do
  UPSP_FORCE_COLLECTIVES=1 timeout -k 10 300 python3 bench.py --force-chunked --chunks $k --no-cpu-baseline --no-reraycast > gpurun_out/ck.json 2> gpurun_out/ck.err || { tail -2 gpurun_out/ck.err; continue; }
  python3 - $k <<'PY'
import json,sys
d=json.loads(open("gpurun_out/ck.json").read().strip().splitlines()[-1])
print("chunks", sys.argv[1], round(d["value"]), "frames/s step %.3f ms" % d["ms_per_step"], {a:round(b,3) for a,b in d["breakdown_ms"].items()})
PY
done
