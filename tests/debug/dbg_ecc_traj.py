"""python tests/debug/dbg_ecc_traj.py <seed> <frame>: |M_gpu - M_oracle| of one frame of a soak_ecc.py case after 1, 2, 3, 5, 8
iterations from the identity (stop test off), through the single-frame entry point.  Run it under UPSP_ECC_DIRECT=1 to
take the LDS tile out of the general iteration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import oracle
from upsp_processing_amd import engine
from soak_ecc import make_case

seed, f = int(sys.argv[1]), int(sys.argv[2])
H, W, F, interp, frames, ref, pix, ok, truth = make_case(seed)
img, _ = oracle.fix_hot_pixels(frames[f])
d_ref, d_fr = torch.as_tensor(ref).cuda(), torch.as_tensor(img.copy()).cuda()
out = []
for k in [int(x) for x in os.environ.get('TRAJ', '1,2,3,5,8').split(',')]:
    _, Mg, _ = engine.register_pixel(d_ref, d_fr, max_iters=k, eps=-1.0, interp=interp)
    _, Mo, _ = oracle.register_pixel(ref, img, max_iters=k, eps=-1.0, interp=interp)
    out.append("%d: %.1e %.1e (t = %.4f %.4f)" % (k, np.abs(Mg[:, :2] - Mo[:, :2]).max(), np.abs(Mg[:, 2] - Mo[:, 2]).max(), Mg[0, 2], Mg[1, 2]))
print("%dx%d interp %d" % (H, W, interp)); print(" | ".join(out))
