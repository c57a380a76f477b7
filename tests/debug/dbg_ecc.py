import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import oracle as orc
from upsp_processing_amd import engine, synthetic as syn
size = 1024
frames = torch.empty((256, size, size), dtype=torch.uint16, device="cuda")
for f0 in range(0, 256, 50):
    syn.synth_frames_torch(min(50, 256 - f0), size, size, first=f0, out=frames[f0:f0 + 50])
ref = frames[0].to(torch.float32).contiguous()
for f in (201, 206, 202):
    out, M, it = engine.register_pixel(ref, frames[f].contiguous())
    fr = frames[f].cpu().numpy()
    out_o, M_o, it_o = orc.register_pixel(ref.cpu().numpy(), fr)
    print("frame", f, "gpu iters", it, "oracle iters", it_o, "dM", np.abs(M - M_o).max())
    rhos = []
    Mx = None
    for k in range(1, 9):
        Mk, its, rho = orc.find_transform_ecc(ref.cpu().numpy(), fr.astype(np.float32), max_iters=k)
        rhos.append(round(rho, 6))
    print("  oracle rho by iteration:", rhos)
