"""A/B of the ECC sums launches on bench content: 256 frames of 1024^2 through the registration path, per-label timers.
   python tools/ecc_ab.py [label=ENV=VALUE ...]   e.g.  python tools/ecc_ab.py lds0:UPSP_ECC_LDS=0 lds2:UPSP_ECC_LDS=2"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from upsp_processing_amd import _capi, engine, synthetic as syn
size, F, N = 1024, 256, 100000
frames = torch.empty((F, size, size), dtype=torch.uint16, device="cuda")
syn.synth_frames_torch(F, size, size, out=frames)
pix = torch.randint(0, size * size, (N,), device="cuda", dtype=torch.int32)
variants = [a.split(":", 1) for a in sys.argv[1:]] or [["default", ""]]
res = {}
for rep in range(2):
    for label, envs in variants:
        sets = [e.split("=", 1) for e in envs.split(",") if e]
        for k, v in sets:
            os.environ[k] = v
        pipe = engine.FramePipeline(1, size, size, N, registration=1)
        pipe.set_reference(0, frames[0].to(torch.float32))
        pipe.set_projection(0, pix)
        w = torch.zeros((F, 1, 6), dtype=torch.float32, device="cuda")
        pipe.process(frames.clone(), 0, want_rows=True, warps=w)
        torch.cuda.synchronize()
        _capi.timing_enable(True)
        for _ in range(3):
            pipe.reset()
            pipe.process(frames.clone(), 0, want_rows=True, warps=w)
        torch.cuda.synchronize()
        _capi.timing_enable(False)
        t = _capi.timing_report()
        st = pipe.ecc_stats()
        pipe.close()
        for k, v in sets:
            os.environ.pop(k, None)
        line = {k: [round(v[1] / max(v[0], 1) * 1e3, 1), int(v[0])] for k, v in t.items() if k.startswith("ecc_") or k.startswith("gauss")}
        print(label, rep, "us per launch / launches:", json.dumps(line), "iters/frame %.2f" % (st["frame_iterations"] / max(st["frames"], 1)),
              "warp checksum %.9g" % float(w.double().abs().sum()), flush=True)
