import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from upsp_processing_amd import engine, synthetic as syn
os.environ["UPSP_REG_BATCH"] = "32"
H, W, F = 96, 160, 70
n = 2500
frames = syn.synth_frames_numpy(F, H, W, seed=H + W, hot=True)
rng = np.random.default_rng(H)
hot_frames = (1, F // 2, F - 1)
for f in hot_frames:
    frames[f, rng.integers(2, H - 2), rng.integers(2, W - 2)] = 4090
frames[hot_frames[1], 0, rng.integers(2, W - 2)] = 4095
frames[hot_frames[2], H - 1, W - 1] = 4095
ref = frames[0].astype(np.float32)
pix = (rng.integers(0, H, n) * W + rng.integers(0, W, n)).astype(np.int32)
out = {}
for mode in ("0", "1"):
    os.environ["UPSP_ECC_FUSED_BLUR"] = mode
    pipe = engine.FramePipeline(1, W, H, n, registration=1)
    pipe.set_projection(0, pix); pipe.set_reference(0, ref)
    rt = torch.full((n, engine.series_ld(F)), -3.0, dtype=torch.float32, device="cuda")
    d = torch.as_tensor(frames.copy()).cuda()
    w = torch.zeros((F, 1, 6), dtype=torch.float32, device="cuda")
    it = torch.full((F, 1), -1, dtype=torch.int32, device="cuda")
    pipe.process(d, 0, rows_t=rt[:, :F], want_rows=False, warps=w, ecc_iters=it)
    torch.cuda.synchronize()
    out[mode] = (w.cpu().numpy()[:, 0], it.cpu().numpy()[:, 0], d.cpu().numpy())
    pipe.close()
a, c = out["0"], out["1"]
dd = np.abs(a[0] - c[0]).max(axis=1)
changed = [(f, int((out["0"][2][f] != frames[f]).sum())) for f in range(F) if (out["0"][2][f] != frames[f]).any()]
print("changed frames", changed)
for f in np.argsort(dd)[::-1][:10]:
    print(f, dd[f], a[1][f], c[1][f], a[0][f], c[0][f])
from oracle import oracle as orc
ea, ec = [], []
for f in range(2, 40):
    if a[1][f] > 6: continue
    img, _ = orc.fix_hot_pixels(frames[f])
    _, M_o, it_o = orc.register_pixel(ref, img)
    Ma, Mc = a[0][f].reshape(2, 3), c[0][f].reshape(2, 3)
    ea.append((np.abs(Ma[:, :2] - M_o[:, :2]).max(), np.abs(Ma[:, 2] - M_o[:, 2]).max()))
    ec.append((np.abs(Mc[:, :2] - M_o[:, :2]).max(), np.abs(Mc[:, 2] - M_o[:, 2]).max()))
    print(f, a[1][f], c[1][f], it_o, "two-kernel vs oracle", ea[-1], "fused vs oracle", ec[-1])
print("two-kernel: max", np.max(ea, axis=0), "mean", np.mean(ea, axis=0))
print("fused:      max", np.max(ec, axis=0), "mean", np.mean(ec, axis=0))
