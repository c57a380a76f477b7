"""Video decode: 12-bit unpack oracle vs the golden produced by the
reference's Python unpacker on the reference's MRAW fixture; MRAW header properties pinned by
cpp/test/test_mraw.cpp:5-13; GPU kernel vs oracle (bit-exact)."""
import hashlib
import json
import os

import numpy as np
import pytest

import refdata


GOLD = refdata.GOLDEN


def golden():
    return json.load(open(os.path.join(refdata.GOLDEN, "mraw_golden.json")))


def test_mraw_header_properties():
    from upsp_processing_amd.video import MrawReader
    r = MrawReader(os.path.join(refdata.GOLDEN, "12bitMRAW.mraw"))
    assert (r.height, r.width, r.bit_depth, r.num_frames) == (1024, 1024, 12, 2)   # test_mraw.cpp
    assert r.frame_rate == 1000 and r.frame_bytes == 1024 * 1024 * 3 // 2
    with pytest.raises(ValueError):
        MrawReader("/nonexistent.mraw")
    with pytest.raises(IndexError):
        r.read_packed(2, 2)


def test_unpack_oracle_matches_reference_golden(oracle):
    g = golden()
    buf = np.fromfile(os.path.join(refdata.GOLDEN, "12bitMRAW.mraw"), dtype=np.uint8)
    pix = oracle.unpack_12bit(buf)
    assert pix.size == 2 * 1024 * 1024 and int(pix.max()) == g["max"] < 4096
    assert hashlib.sha256(pix.astype("<u2").tobytes()).hexdigest() == g["sha256"]
    assert pix[::g["sample_stride"]].astype(int).tolist() == g["sample"]
    # round trip with a packer (python/upsp/video/util.py:39-55 semantics)
    rng = np.random.default_rng(0)
    v = rng.integers(0, 4096, size=4096).astype(np.uint16)
    b = np.zeros(v.size * 3 // 2, np.uint8)
    b[0::3] = v[0::2] >> 4
    b[1::3] = ((v[0::2] & 0x0F) << 4) | (v[1::2] >> 8)
    b[2::3] = v[1::2] & 0xFF
    assert np.array_equal(oracle.unpack_12bit(b), v)


@pytest.mark.gpu
def test_unpack_gpu_bitwise(gpu_lib, oracle):
    import torch
    from upsp_processing_amd import video
    g = golden()
    r = video.MrawReader(os.path.join(refdata.GOLDEN, "12bitMRAW.mraw"))
    frames, cnt = r.read_frames_device(1, 2, hot_thresh=4064)
    out = frames.cpu().numpy()
    assert hashlib.sha256(out.astype("<u2").tobytes()).hexdigest() == g["sha256"]
    assert np.array_equal(cnt.cpu().numpy(), (out.reshape(2, -1) >= 4064).sum(1))
    # ragged sizes / unaligned bases
    rng = np.random.default_rng(1)
    for h, w, f in [(3, 6, 2), (37, 54, 3), (64, 64, 1), (5, 2, 4)]:
        packed = rng.integers(0, 256, size=(f, h * w * 3 // 2)).astype(np.uint8)
        got, c = video.unpack_12bit(torch.as_tensor(packed).cuda(), h, w, hot_thresh=2000)
        exp = np.stack([oracle.unpack_12bit(p) for p in packed]).reshape(f, h, w)
        assert np.array_equal(got.cpu().numpy(), exp)
        assert np.array_equal(c.cpu().numpy(), (exp.reshape(f, -1) >= 2000).sum(1))
    # fed straight into the frame loop
    from upsp_processing_amd import engine
    n = 5000
    pix = torch.as_tensor(rng.integers(-1, 1024 * 1024, size=n).astype(np.int32)).cuda()
    pipe = engine.FramePipeline(1, 1024, 1024, n)
    pipe.set_projection(0, pix)
    rows = pipe.process(frames.clone(), 0).cpu().numpy()
    for fidx in range(2):
        img, _ = oracle.fix_hot_pixels(out[fidx])
        sol = oracle.project_frame(img, pix.cpu().numpy(), None)
        sol[pix.cpu().numpy() < 0] = np.nan
        assert np.array_equal(rows[fidx].view(np.int32), sol.view(np.int32))


# ------------------------------------------------------------------ 10-bit / CINE --------
def _bytes10():
    man = json.load(open(os.path.join(GOLD, "mraw_golden.json")))["unpack10"]
    rng = np.random.default_rng(man["seed"])
    return man, rng.integers(0, 256, man["nbytes"], dtype=np.uint8)


def test_oracle_unpack10_matches_reference_unpacker(oracle):
    """oracle vs the golden produced by python/upsp/video/util.py:unpack_10bpp."""
    import hashlib
    man, b = _bytes10()
    pix = oracle.unpack_10bit(b)
    assert hashlib.sha256(pix.astype("<u2").tobytes()).hexdigest() == man["sha256"]
    assert pix[:16].tolist() == man["head"] and int(pix.astype(np.int64).sum()) == man["sum"]
    lut = (np.arange(1024, dtype=np.uint16)[::-1] * 4).astype(np.uint16)
    assert np.array_equal(oracle.unpack_10bit(b, lut), lut[pix])


def _write_cine(path, frames_bytes, width, height, bpp):
    """Minimal Cine container: CINEFILEHEADER (44 B), BITMAPINFOHEADER (40 B), SETUP (RealBPP @896,
    FrameRate @768), image offsets, then per image {annotation size = 8, image size, pixels}."""
    import struct
    n = len(frames_bytes)
    off_setup = 84
    setup = bytearray(7240)
    setup[140:142] = b"ST"
    struct.pack_into("<H", setup, 142, 7240)
    struct.pack_into("<I", setup, 768, 5000)
    struct.pack_into("<I", setup, 896, bpp)
    off_offsets = off_setup + len(setup)
    first_img = off_offsets + 8 * n
    offs, pos = [], first_img
    for fb in frames_bytes:
        offs.append(pos)
        pos += 8 + len(fb)
    head = bytearray(84)
    head[0:2] = b"CI"
    struct.pack_into("<HHH", head, 2, 44, 0, 1)
    struct.pack_into("<iIiIIII", head, 8, 0, n, 0, n, 44, off_setup, off_offsets)
    struct.pack_into("<IiiHHII", head, 44, 40, width, height, 1, 16, 256, len(frames_bytes[0]))
    with open(path, "wb") as f:
        f.write(head)
        f.write(setup)
        f.write(np.array(offs, "<i8").tobytes())
        for fb in frames_bytes:
            f.write(struct.pack("<II", 8, len(fb)))
            f.write(fb)


def test_cine_reader_header(tmp_path):
    from upsp_processing_amd import video
    W, H = 64, 48
    fb = [bytes(W * H * 12 // 8)] * 3
    p = str(tmp_path / "a.cine")
    _write_cine(p, fb, W, H, 12)
    r = video.CineReader(p)
    assert (r.width, r.height, r.num_frames, r.bit_depth, r.frame_rate) == (W, H, 3, 12, 5000)
    assert r.read_packed(2, 2).shape == (2, W * H * 3 // 2)
    with pytest.raises(IndexError):
        r.read_packed(3, 2)
    _write_cine(p, [bytes(W * H * 10 // 8)], W, H, 10)
    assert video.CineReader(p).bit_depth == 12 and video.CineReader(p).raw_bit_depth == 10


@pytest.mark.gpu
def test_unpack10_gpu_vs_oracle(gpu_lib, oracle):
    import torch
    from upsp_processing_amd import video
    rng = np.random.default_rng(9)
    for (H, W, F) in ((32, 64, 3), (6, 10, 2), (128, 256, 2)):
        b = rng.integers(0, 256, (F, H * W * 5 // 4), dtype=np.uint8)
        lut = rng.integers(0, 4096, 1024).astype(np.uint16)
        for l in (None, lut):
            got = video.unpack_10bit(torch.as_tensor(b).cuda(), H, W, l).cpu().numpy()
            want = np.stack([oracle.unpack_10bit(b[f], l) for f in range(F)]).reshape(F, H, W)
            assert np.array_equal(got, want)
    man, bb = _bytes10()
    got = video.unpack_10bit(torch.as_tensor(bb[None]).cuda(), 64, 256).cpu().numpy().reshape(-1)
    assert got[:16].tolist() == man["head"] and int(got.astype(np.int64).sum()) == man["sum"]


@pytest.mark.gpu
def test_cine_reader_frames(gpu_lib, oracle, tmp_path):
    from upsp_processing_amd import video
    rng = np.random.default_rng(3)
    W, H = 64, 32
    p = str(tmp_path / "a.cine")
    pix = rng.integers(0, 4096, (3, H * W)).astype(np.uint16)
    pk = np.zeros((3, H * W * 3 // 2), np.uint8)
    pk[:, 0::3] = pix[:, 0::2] >> 4
    pk[:, 1::3] = ((pix[:, 0::2] & 0x0F) << 4) | (pix[:, 1::2] >> 8)
    pk[:, 2::3] = pix[:, 1::2] & 0xFF
    _write_cine(p, [pk[i].tobytes() for i in range(3)], W, H, 12)
    fr = video.CineReader(p).read_frames_device(2, 2).cpu().numpy()
    assert np.array_equal(fr.reshape(2, -1), pix[1:])
    b10 = rng.integers(0, 256, (2, H * W * 5 // 4), dtype=np.uint8)
    _write_cine(p, [b10[i].tobytes() for i in range(2)], W, H, 10)
    lut = (np.arange(1024) * 4).astype(np.uint16)
    with pytest.raises(ValueError):
        video.CineReader(p).read_frames_device(1, 1)
    fr = video.CineReader(p, lut=lut).read_frames_device(1, 2).cpu().numpy()
    assert np.array_equal(fr.reshape(2, -1), np.stack([oracle.unpack_10bit(b10[i], lut) for i in range(2)]))
    words = rng.integers(0, 4096, (1, H, W)).astype("<u2")
    _write_cine(p, [words[0].tobytes()], W, H, 8)
    fr = video.CineReader(p).read_frames_device(1, 1).cpu().numpy()
    assert np.array_equal(fr[0], words[0][::-1])


@pytest.mark.gpu
def test_frame_feed_ring(gpu_lib):
    """Pinned staging ring: 7 chunks through 3 slots, uploads on the copy stream, the 12-bit unpack on
    the consumer stream -- every chunk arrives intact although the slots are reused while earlier
    chunks are still being consumed; misuse is refused."""
    import torch
    from upsp_processing_amd import _capi, video
    H, W, n = 64, 96, 5
    fb = H * W * 3 // 2
    feed = video.FrameFeed(n * fb, 3)
    rng = np.random.default_rng(3)
    outs, want = [], []
    for k in range(7):
        raw = rng.integers(0, 256, (n, fb), dtype=np.uint8)

        def fill(dst, raw=raw):
            dst[:raw.size] = raw.reshape(-1)
            return raw.size
        d = feed.upload(fill)
        outs.append(video.unpack_12bit(d.view(n, fb), H, W))
        feed.release()
        b = raw.astype(np.uint16)
        px = np.empty((n, H * W), np.uint16)
        px[:, 0::2] = (b[:, 0::3] << 4) | (b[:, 1::3] >> 4)
        px[:, 1::2] = ((b[:, 1::3] & 0x0F) << 8) | b[:, 2::3]
        want.append(px.reshape(n, H, W))
    torch.cuda.synchronize()
    for o, w in zip(outs, want):
        assert np.array_equal(o.cpu().view(torch.int16).numpy().view(np.uint16), w)
    with pytest.raises(_capi.UpspError):
        feed.upload(lambda dst: feed.slot_bytes + 1)        # more bytes than the slot holds

    def bad_reader(dst):
        raise IndexError("frame out of range")
    with pytest.raises(IndexError):
        feed.upload(bad_reader)                             # the reader fails: the slot goes back (upsp_feed_abort)
    for k in range(4):                                      # ... and the feed is still usable, all the way round the ring
        raw = rng.integers(0, 256, (n, fb), dtype=np.uint8)

        def fill2(dst, raw=raw):
            dst[:raw.size] = raw.reshape(-1)
            return raw.size
        d = feed.upload(fill2)
        got = d.cpu().numpy().copy()
        feed.release()
        assert np.array_equal(got, raw.reshape(-1))
    feed.close()
