"""Randomised parity soak of the frame loop (run on a GPU box from the repository root):
    python tests/debug/soak_frames.py [seconds]
Random image sizes, node counts, frame counts (1 .. 1100, calls split at random columns), 1-3 cameras with
weights, hot pixels of every kind (isolated, neighbours, stuck, exactly / more than max_hot, saturated frames),
user skip lists, packed rows (row map), u16 series, small compact budgets (several frame groups per call), both
schedules -- GPU vs the oracle loop: series and repaired frames bit for bit, accumulators exact for one camera
without weights and to 1e-12 otherwise.  Exits non-zero on the first mismatch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import oracle
from upsp_processing_amd import engine

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
t_end = time.time() + budget
seed = int(os.environ.get("SOAK_SEED", "1"))
ncase = 0


def oracle_loop(frames, pix, weight, skipped):
    ncams, F = len(frames), frames[0].shape[0]
    n = pix.shape[1]
    s, ss = np.zeros(n), np.zeros(n)
    rows = np.zeros((F, n), np.float32)
    fixed = [f.copy() for f in frames]
    for f in range(F):
        sol = None
        for c in range(ncams):
            img, _ = oracle.fix_hot_pixels(frames[c][f])
            fixed[c][f] = img
            cs = oracle.project_frame(img, pix[c], None if weight is None else weight[c])
            sol = cs if sol is None else (sol + cs).astype(np.float32)
        sol[skipped] = np.nan
        oracle.accumulate(sol, s, ss)
        rows[f] = sol
    return rows, s, ss, fixed


while time.time() < t_end:
    rng = np.random.default_rng(seed)
    H, W = int(rng.integers(1, 40)) * 2, int(rng.integers(3, 150))
    split = rng.random() < 0.25                     # pass A in two launches (needs whole 128-pixel tiles: H x W a multiple of 128)
    if split:
        H, W = int(rng.choice([2, 4, 8, 16, 32, 64])), int(rng.choice([64, 128, 192]))
        if (H * W) % 128:
            H *= 2
    n = int(rng.integers(1, 4000))
    F = int(rng.choice([1, 2, 63, 64, 65, int(rng.integers(1, 300)), int(rng.integers(300, 1100))]))
    if H * W * F > 6_000_000:
        F = max(1, 6_000_000 // (H * W))
    ncams = int(rng.choice([1, 1, 1, 2, 3]))
    frames = [rng.integers(0, 3500, (F, H, W)).astype(np.uint16) for _ in range(ncams)]
    for fr in frames:
        kind = rng.integers(0, 5)
        if kind >= 1:                               # isolated hot pixels in some frames
            for f in rng.choice(F, min(F, int(rng.integers(1, 8))), replace=False):
                k = int(rng.integers(1, 8))
                fr[f].flat[rng.choice(H * W, min(k, H * W), replace=False)] = rng.integers(4064, 4096, min(k, H * W))
        if kind >= 2:                               # stuck pixels (hot in every frame), one flickering
            for _ in range(int(rng.integers(1, 4))):
                fr[:, rng.integers(0, H), rng.integers(0, W)] = 4095
        if kind >= 3 and F > 2:                     # neighbours + a saturated frame
            y, x = int(rng.integers(0, H)), int(rng.integers(0, W - 1))
            fr[F // 2, y, x] = 4095; fr[F // 2, y, x + 1] = 4080
            fr[F - 1][rng.random((H, W)) < 0.2] = 4090
        if kind == 4:                               # small change: old - new <= min_change keeps the pixel
            fr[0].flat[0] = 4064; fr[0].flat[1:3] = 3900
    pix = rng.integers(-1, H * W, size=(ncams, n)).astype(np.int32)
    pix[:, rng.random(n) < 0.3] = -1
    hot_cols = np.argwhere(frames[0][0] >= 4064)
    if len(hot_cols) and n > 10:                    # many nodes on a hot pixel
        p = int(hot_cols[0][0]) * W + int(hot_cols[0][1])
        pix[0, :min(n, 40)] = p
    weight = rng.random((ncams, n)).astype(np.float32) if (ncams > 1 or rng.random() < 0.2) else None
    sk = oracle.skipped_nodes(pix)
    user_skip = rng.random() < 0.3
    if user_skip:
        sk = sk.copy(); sk[rng.random(n) < 0.1] = True
    rows_o, s_o, ss_o, fixed_o = oracle_loop(frames, pix, weight, sk)
    exact = ncams == 1 and weight is None
    fused = int(rng.choice([0, 1, 2]))
    compact_mb = int(rng.choice([0, 0, 1]))
    pipe = engine.FramePipeline(ncams, W, H, n, fused_scan=fused, compact_mb=compact_mb)
    if split:
        pipe.set_scan_split(True)
    for c in range(ncams):
        pipe.set_projection(c, pix[c], None if weight is None else weight[c])
    if user_skip:
        pipe.set_skipped(torch.as_tensor(sk.astype(np.uint8)).cuda())
    packed = rng.random() < 0.4
    u16 = packed and exact and rng.random() < 0.5
    keep = np.nonzero(~sk)[0] if packed else np.arange(n)
    if packed:
        rowmap = np.full(n, -1, np.int32); rowmap[keep] = np.arange(keep.size)
        pipe.set_row_map(torch.as_tensor(rowmap).cuda())
    ld = F + int(rng.integers(0, 9))
    # row padding (upsp_pipeline_set_row_padding): a pitch of whole 128-byte lines with room behind the frames, declared to the
    # pipeline; the row pass may then write up to the next line boundary, nothing beyond
    pad_on = rng.random() < 0.5
    if pad_on:
        pipe.set_row_padding(True)
        if rng.random() < 0.7:
            ld = (F + 63) // 64 * 64 + 64 * int(rng.integers(0, 2))
    if u16:
        buf = torch.full((max(keep.size, 1), ld), 7, dtype=torch.int32, device="cuda").to(torch.uint16)
    else:
        buf = torch.full((max(keep.size, 1), ld), -7.0, dtype=torch.float32, device="cuda")
    d = [torch.as_tensor(f.copy()).cuda() for f in frames]
    cuts = sorted(set([0, F] + [int(c) for c in rng.integers(0, F + 1, int(rng.integers(0, 3)))]))
    for a, b in zip(cuts[:-1], cuts[1:]):
        if b > a:
            pipe.process([x[a:b].contiguous() for x in d] if (a, b) != (0, F) else d, a, rows_t=buf[:keep.size, :F], col0=a, want_rows=False)
            if (a, b) != (0, F):               # frames are repaired in the slices: copy them back for the comparison
                pass
    tag = "seed %d: %dx%d n=%d F=%d cams=%d fused=%d compact_mb=%d packed=%d u16=%d cuts=%s" % (
        seed, H, W, n, F, ncams, fused, compact_mb, packed, u16, cuts) + (" padded ld=%d" % ld if pad_on else "")
    got = buf[:keep.size, :F].cpu().numpy().astype(np.float32)
    want = rows_o.T[keep]
    ok = np.array_equal(got.view(np.int32), want.view(np.int32)) if not u16 else np.array_equal(got, want)
    if not ok:
        print("SERIES MISMATCH", tag); sys.exit(1)
    per_line = 64 if u16 else 32
    stop = min(ld, (F + per_line - 1) // per_line * per_line) if (pad_on and ld % per_line == 0) else F
    pad = buf[:keep.size, stop:].cpu().numpy()
    if pad.size and not (pad == (7 if u16 else -7.0)).all():
        print("WROTE PAST THE FRAMES", tag, "padding", pad_on, "ld", ld); sys.exit(1)
    if cuts == [0, F]:
        for c in range(ncams):
            if not np.array_equal(d[c].cpu().numpy(), fixed_o[c]):
                print("FRAME REPAIR MISMATCH", tag); sys.exit(1)
    s_g, ss_g = [x.cpu().numpy() for x in pipe.accumulators()]
    live = ~np.isnan(s_o)
    if not np.array_equal(np.isnan(s_g), ~live):
        print("ACCUMULATOR NaN PATTERN", tag); sys.exit(1)
    if exact:
        good = np.array_equal(s_g[live], s_o[live]) and np.array_equal(ss_g[live], ss_o[live])
    else:
        good = np.allclose(s_g[live], s_o[live], rtol=1e-12, atol=1e-9) and np.allclose(ss_g[live], ss_o[live], rtol=1e-12, atol=1e-6)
    if not good:
        print("ACCUMULATOR MISMATCH", tag); sys.exit(1)
    pipe.close()
    ncase += 1
    if ncase % 20 == 0:
        print(tag, "ok", flush=True)
    seed += 1
print("soak: %d cases, all equal" % ncase)
