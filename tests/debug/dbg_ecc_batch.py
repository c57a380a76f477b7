"""One case of soak_ecc.py taken apart: python tests/debug/dbg_ecc_batch.py <seed> <frame>
The frame registered (a) inside its batch, (b) in a batch of its own (reference + the frame), (c) by the single-frame
entry point, (d) by the oracle -- warp matrices and iteration counts side by side."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import oracle
from upsp_processing_amd import engine
from soak_ecc import make_case

seed, f = int(sys.argv[1]), int(sys.argv[2])
H, W, F, interp, frames, ref, pix, ok, truth = make_case(seed)
print("case %d: %d frames of %dx%d, interp %d" % (seed, F, H, W, interp))


def batch(fr):
    pipe = engine.FramePipeline(1, W, H, len(pix), registration=1, interp=interp)
    pipe.set_projection(0, pix)
    pipe.set_reference(0, ref)
    n = fr.shape[0]
    warps = torch.zeros((n, 1, 6), dtype=torch.float32, device="cuda")
    iters = torch.full((n, 1), -1, dtype=torch.int32, device="cuda")
    pipe.process(torch.as_tensor(fr.copy()).cuda(), 0, warps=warps, ecc_iters=iters)
    return warps.cpu().numpy()[:, 0], iters.cpu().numpy()[:, 0]


wa, ia = batch(frames)
wb, ib = batch(frames[[0, f]])
wc, ic = batch(np.concatenate([frames[[0, f]], frames[1:]]))            # same batch, the frame first
img, _ = oracle.fix_hot_pixels(frames[f])
_, Ms, its = engine.register_pixel(torch.as_tensor(ref).cuda(), torch.as_tensor(img.copy()).cuda(), interp=interp)
_, Mo, ito = oracle.register_pixel(ref, img, interp=interp)
np.set_printoptions(precision=7, linewidth=200)
print("in its batch      %2d %s" % (ia[f], wa[f]))
print("batch of its own  %2d %s" % (ib[1], wb[1]))
print("first of the batch%2d %s" % (ic[1], wc[1]))
print("single-frame call %2d %s" % (its, Ms.ravel()))
print("oracle            %2d %s" % (ito, Mo.ravel()))
print("iterations of the batch:", ia.tolist())
