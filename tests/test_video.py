"""Video decode: 12-bit unpack oracle vs the golden produced by the
reference's Python unpacker on the reference's MRAW fixture; MRAW header properties pinned by
cpp/test/test_mraw.cpp:5-13; GPU kernel vs oracle (bit-exact)."""
import hashlib
import json
import os

import numpy as np
import pytest

import refdata


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
