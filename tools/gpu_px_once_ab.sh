# N > 1 loop on one GPU (one-rank RCCL): pass A once beside the build against pass A per chunk (--chunk-scan),
# synchronous and deferred exchange; alternating in one call.  Output: gpurun_out/px_once/*.json + a table.
set -o pipefail
o=gpurun_out/px_once; mkdir -p $o
export UPSP_FORCE_COLLECTIVES=1
run() { # name, args
  timeout -k 10 300 python3 bench.py --force-chunked --no-cpu-baseline --no-reraycast --steps 6 --warmup 2 $2 > $o/$1.json 2>> $o/err.log || { echo "$1 failed"; tail -5 $o/err.log; exit 1; }
  python3 - $o/$1.json $1 <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-28s %8.0f frames/s  step %.3f ms  %s" % (sys.argv[2], d["value"], d["ms_per_step"], {k: round(v, 3) for k, v in d["breakdown_ms"].items()}))
PY
}
for rep in 1 2; do
  run sync_default_$rep "--sync-exchange"
  run sync_chunkscan_$rep "--sync-exchange --chunk-scan"
  run deferred_default_$rep "--defer-exchange"
  run deferred_chunkscan_$rep "--defer-exchange --chunk-scan"
  run deferred_once_k4_$rep "--defer-exchange --chunks 4"
done
