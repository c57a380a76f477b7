# A/B of one environment switch on a bench command, alternating in ONE call:
#   bash tools/gpu_ab_env.sh VAR=value "<bench args>" [repetitions]      -> gpurun_out/ab_env/*.json + a table
set -o pipefail
kv=$1; args=$2; reps=${3:-3}
o=gpurun_out/ab_env; mkdir -p $o
one() { # name, env assignment or ""
  if [ -n "$2" ]; then env_cmd="env $2"; else env_cmd=""; fi
  timeout -k 10 400 $env_cmd python3 bench.py --no-cpu-baseline --no-reraycast $args > $o/$1.json 2>> $o/err.log || { echo "$1 failed"; tail -5 $o/err.log; exit 1; }
  python3 - $o/$1.json $1 <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-16s %8.0f frames/s  step %.4f ms  %s" % (sys.argv[2], d["value"], d["ms_per_step"], {k: round(v, 3) for k, v in d["breakdown_ms"].items()}))
PY
}
for i in $(seq 1 $reps); do
  one base_$i ""
  one with_$i "$kv"
done
