"""The per-rank full sizes of configs[3] / configs[4] on one GPU: footprint and timing facts for DESIGN.md section 3
(the asserts are the tests').   python tools/full_size.py [3|4|both]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fullsize_cases
from oracle import oracle
which = sys.argv[1] if len(sys.argv) > 1 else "both"
if which in ("3", "both"):
    print(json.dumps(fullsize_cases.config3_rank_share(oracle)), flush=True)
if which in ("4", "both"):
    print(json.dumps(fullsize_cases.config4_rank_share()), flush=True)
