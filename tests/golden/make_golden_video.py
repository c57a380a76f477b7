#!/usr/bin/env python3
"""Golden vectors for the 12-bit unpack (build container only; needs /root/reference).

Runs the reference's own Python unpacker (python/upsp/video/util.py:25-36, loaded by file
path so that no other module of the package is imported) on the reference's MRAW fixture
(cpp/test/mraw/12bitMRAW.{cih,mraw}, copied to tests/golden/) and stores SHA-256 of the
unpacked frames, a strided sample and the header properties its C++ test pins
(cpp/test/test_mraw.cpp:5-13: 1024 x 1024, 12 bit, 2 frames)."""
import hashlib
import importlib.util
import json
import os
import shutil

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def main():
    for ext in ("cih", "mraw"):
        dst = os.path.join(HERE, "12bitMRAW." + ext)
        if not os.path.exists(dst):
            shutil.copyfile(os.path.join(REF, "cpp/test/mraw/12bitMRAW." + ext), dst)
            os.chmod(dst, 0o644)
    spec = importlib.util.spec_from_file_location("ref_video_util", os.path.join(REF, "python/upsp/video/util.py"))
    util = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(util)
    buf = open(os.path.join(HERE, "12bitMRAW.mraw"), "rb").read()
    pix = util.unpack_12bpp(buf)
    assert pix.size == 2 * 1024 * 1024
    man = {"width": 1024, "height": 1024, "bit_depth": 12, "num_frames": 2,
           "sha256": hashlib.sha256(pix.astype("<u2").tobytes()).hexdigest(),
           "sample_stride": 65521,
           "sample": pix[::65521].astype(int).tolist(),
           "max": int(pix.max()), "sum": int(pix.astype(np.int64).sum())}
    # 10-bit packing: the reference's unpack_10bpp (util.py:6-22) on a seeded byte string
    rng = np.random.default_rng(20240607)
    b10 = rng.integers(0, 256, 5 * 4096, dtype=np.uint8).tobytes()
    p10 = util.unpack_10bpp(b10)
    man["unpack10"] = {"seed": 20240607, "nbytes": len(b10),
                       "sha256": hashlib.sha256(p10.astype("<u2").tobytes()).hexdigest(),
                       "head": p10[:16].astype(int).tolist(), "sum": int(p10.astype(np.int64).sum())}
    json.dump(man, open(os.path.join(HERE, "mraw_golden.json"), "w"), indent=1)
    print(man["sha256"], man["max"], man["sum"], man["sample"][:8])


if __name__ == "__main__":
    main()
