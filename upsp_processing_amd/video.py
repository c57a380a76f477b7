"""Video decode feeding the frame loop.

Photron MRAW: `.cih` text header + `.mraw` raw frames, 12-bit packed
(cpp/lib/MrawReader.cpp:60-146).  The packed bytes are uploaded as they are on disk and
unpacked in HBM by the library (upsp_unpack_12bit) -- 1.5 instead of 2 bytes per pixel
over PCIe, no host-side bit twiddling."""
import ctypes as C
import os
import re

import numpy as np


class MrawReader:
    """Mirror of upsp::MrawReader (cpp/include/MrawReader.h, cpp/lib/MrawReader.cpp):
    properties from the .cih header, 1-based frame numbers."""

    def __init__(self, mraw_file):
        if not os.path.isfile(mraw_file):
            raise ValueError("Video File is invalid")            # MrawReader.cpp:63-65
        self.mraw_file = mraw_file
        self.cih_file = mraw_file[:mraw_file.rfind(".")] + ".cih"
        tokens = {}
        with open(self.cih_file, "r", errors="replace") as f:
            for line in f:
                toks = re.split(r"\s:\s", line.strip())          # TOKEN_DELIMITER (:79)
                if len(toks) == 2:
                    tokens[toks[0]] = toks[1]
        self.width = int(tokens["Image Width"])
        self.height = int(tokens["Image Height"])
        self.bit_depth = int(tokens["Color Bit"])
        self.frame_rate = int(tokens["Record Rate(fps)"])
        self.num_frames = int(tokens["Total Frame"])
        if self.bit_depth != 12:
            raise NotImplementedError("only 12-bit MRAW is supported (like the reference)")
        self.frame_bytes = self.width * self.height * self.bit_depth // 8

    def read_packed(self, first, count):
        """Packed bytes of frames first..first+count-1 (1-based like read_frame(n))."""
        if first < 1 or first + count - 1 > self.num_frames:
            raise IndexError("frame out of range")
        with open(self.mraw_file, "rb") as f:
            f.seek((first - 1) * self.frame_bytes)
            buf = np.fromfile(f, dtype=np.uint8, count=count * self.frame_bytes)
        return buf.reshape(count, self.frame_bytes)

    def read_frames_device(self, first, count, hot_thresh=None):
        """u16 tensor [count, H, W] on the GPU (+ per-frame hot-pixel counts if requested)."""
        import torch
        packed = torch.as_tensor(self.read_packed(first, count)).cuda()
        return unpack_12bit(packed, self.height, self.width, hot_thresh)


def unpack_12bit(packed, height, width, hot_thresh=None):
    """packed: u8 CUDA tensor [F, H*W*3/2].  Returns u16 [F,H,W] (and u32 [F] hot counts)."""
    import torch
    from . import _capi
    from .engine import _ptr, _stream
    assert packed.is_cuda and packed.dtype == torch.uint8 and packed.is_contiguous()
    f = packed.shape[0] if packed.dim() == 2 else 1
    npix = height * width
    assert packed.numel() == f * npix * 3 // 2
    out = torch.empty((f, height, width), dtype=torch.uint16, device="cuda")
    cnt = torch.zeros(f, dtype=torch.int32, device="cuda") if hot_thresh is not None else None
    _capi.check(_capi.lib().upsp_unpack_12bit(_ptr(packed), f, npix, _ptr(out),
                                              int(hot_thresh or 0), _ptr(cnt), _stream()))
    return out if hot_thresh is None else (out, cnt)
