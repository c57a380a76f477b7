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

    def read_packed_into(self, dst, first, count):
        """The same bytes read straight into `dst` (u8 array, e.g. a pinned FrameFeed slot)."""
        if first < 1 or first + count - 1 > self.num_frames:
            raise IndexError("frame out of range")
        n = count * self.frame_bytes
        with open(self.mraw_file, "rb", buffering=0) as f:
            f.seek((first - 1) * self.frame_bytes)
            got = f.readinto(memoryview(dst.reshape(-1)[:n]))
        if got != n:
            raise IOError("short read from %s" % self.mraw_file)
        return n

    def read_frames_device(self, first, count, hot_thresh=None, feed=None):
        """u16 tensor [count, H, W] on the GPU (+ per-frame hot-pixel counts if requested).
        feed: a FrameFeed -- the bytes go disk -> pinned slot -> device on the feed's copy stream,
        without blocking the host on the transfer."""
        import torch
        if feed is not None:
            packed = feed.upload(lambda dst: self.read_packed_into(dst, first, count))
            out = unpack_12bit(packed.view(count, self.frame_bytes), self.height, self.width, hot_thresh)
            feed.release()
            return out
        packed = torch.as_tensor(self.read_packed(first, count)).cuda()
        return unpack_12bit(packed, self.height, self.width, hot_thresh)


class FrameFeed:
    """Pinned staging ring + copy stream (include/upsp_gpu.h: upsp_feed_*; the reference's read-ahead
    thread, cpp/exec/psp_process.cpp:867-1007).  One upload() / release() pair per chunk of frames:

        packed = feed.upload(fill)       # fill(dst_u8_array) -> number of bytes it wrote
        frames = unpack_12bit(packed.view(n, frame_bytes), H, W)   # current stream waits for the copy only
        feed.release()                   # the slot may be overwritten once the stream gets here

    The host blocks only when all `nslots` slots are in flight."""

    def __init__(self, slot_bytes, nslots=3):
        from . import _capi
        h = C.c_void_p()
        _capi.check(_capi.lib().upsp_feed_create(int(slot_bytes), int(nslots), C.byref(h)))
        self._h, self.slot_bytes, self.nslots = h, int(slot_bytes), int(nslots)
        self._destroy = _capi.lib().upsp_feed_destroy
        self._pending = []

    def close(self):
        if getattr(self, "_h", None):
            self._destroy(self._h)
            self._h = None

    __del__ = close

    def upload(self, fill):
        """Acquire the next slot, let fill(dst) write into its pinned buffer (dst: numpy u8 view of
        the whole slot; returns the byte count), start the upload.  Returns a u8 device tensor view."""
        import torch
        from . import _capi
        from .engine import _stream
        slot, hp = C.c_int(), C.c_void_p()
        _capi.check(_capi.lib().upsp_feed_acquire(self._h, C.byref(slot), C.byref(hp)))
        dst = np.ctypeslib.as_array(C.cast(hp, C.POINTER(C.c_uint8)), shape=(self.slot_bytes,))
        dp = C.c_void_p()
        try:
            n = int(fill(dst))
            _capi.check(_capi.lib().upsp_feed_commit(self._h, slot.value, n, _stream(), C.byref(dp)))
        except BaseException:
            # the reader failed (frame out of range, short read) or the commit was refused: give the slot
            # back unfilled, the feed stays usable (upsp_feed_abort)
            _capi.lib().upsp_feed_abort(self._h, slot.value)
            raise
        self._pending.append(slot.value)
        from .engine import _DevArray
        t = torch.as_tensor(_DevArray(dp.value, max(n, 1), "|u1", self), device="cuda")[:n]
        return t

    def release(self):
        from . import _capi
        from .engine import _stream
        _capi.check(_capi.lib().upsp_feed_release(self._h, self._pending.pop(0), _stream()))


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


def unpack_10bit(packed, height, width, lut=None):
    """packed: u8 CUDA tensor [F, H*W*5/4].  lut: 1024-entry 10 -> 12 bit table or None.
    Returns u16 [F,H,W]."""
    import torch
    from . import _capi
    from .engine import _dev, _ptr, _stream
    assert packed.is_cuda and packed.dtype == torch.uint8 and packed.is_contiguous()
    f = packed.shape[0] if packed.dim() == 2 else 1
    npix = height * width
    assert packed.numel() == f * npix * 5 // 4
    d_lut = None
    if lut is not None:
        d_lut = _dev(np.asarray(lut).astype(np.uint16), torch.uint16)
        assert d_lut.numel() == 1024
    out = torch.empty((f, height, width), dtype=torch.uint16, device="cuda")
    _capi.check(_capi.lib().upsp_unpack_10bit(_ptr(packed), f, npix, _ptr(d_lut), _ptr(out), _stream()))
    return out


class CineReader:
    """Mirror of upsp::CineReader (cpp/include/CineReader.h, cpp/lib/CineReader.cpp) for what
    psp_process needs: properties from CINEFILEHEADER / BITMAPINFOHEADER / SETUP (field offsets of
    the published Vision Research Cine format: SETUP.FrameRate @768, SETUP.RealBPP @896), the
    image-offset table, and frames -- 12-bit packed and 10-bit packed (+ LUT) unpacked on the GPU,
    8-bit-mode files as 16-bit words flipped vertically (read_linear, CineReader.cpp:435-451).

    The 10 -> 12 bit look-up table is camera data published with the format; pass it as `lut`
    (array or path of a raw little-endian u16[1024] file) or through the UPSP_CINE_LUT variable."""

    def __init__(self, cine_file, lut=None):
        import struct
        if not os.path.isfile(cine_file):
            raise ValueError("Video File is invalid")
        self.cine_file = cine_file
        with open(cine_file, "rb") as f:
            head = f.read(44 + 40)
            if len(head) < 84 or head[:2] != b"CI":
                raise ValueError("not a Cine file: %s" % cine_file)
            (self.first_image_no, self.num_frames, off_image_header, off_setup,
             off_image_offsets) = struct.unpack_from("<iIIII", head, 16)
            self.width, self.height = struct.unpack_from("<ii", head, 44 + 4)
            f.seek(off_setup)
            setup = f.read(900)
            self.frame_rate = struct.unpack_from("<I", setup, 768)[0]
            self.raw_bit_depth = struct.unpack_from("<I", setup, 896)[0]     # bits_per_pixel_ (:143)
            f.seek(off_image_offsets)
            self.image_offsets = np.frombuffer(f.read(8 * self.num_frames), "<i8").copy()
        self.bit_depth = 12 if self.raw_bit_depth == 10 else self.raw_bit_depth   # :175
        if self.raw_bit_depth not in (8, 10, 12):
            raise NotImplementedError("Cine bit depth %d not supported" % self.raw_bit_depth)
        self.frame_bytes = (self.width * self.height * 2 if self.raw_bit_depth == 8
                            else self.width * self.height * self.raw_bit_depth // 8)
        if lut is None and os.environ.get("UPSP_CINE_LUT"):
            lut = os.environ["UPSP_CINE_LUT"]
        if isinstance(lut, str):
            lut = np.fromfile(lut, "<u2")
        self.lut = None if lut is None else np.asarray(lut).astype(np.uint16)
        if self.lut is not None and self.lut.size != 1024:
            raise ValueError("the Cine look-up table needs 1024 entries")

    def read_packed(self, first, count):
        """Raw bytes of frames first..first+count-1 (1-based); pixel data start 8 bytes after the
        image offset (annotation size + image size words, CineReader.cpp:456)."""
        if first < 1 or first + count - 1 > self.num_frames:
            raise IndexError("frame out of range")
        out = np.empty((count, self.frame_bytes), np.uint8)
        with open(self.cine_file, "rb") as f:
            for i in range(count):
                f.seek(int(self.image_offsets[first - 1 + i]) + 8)
                out[i] = np.fromfile(f, dtype=np.uint8, count=self.frame_bytes)
        return out

    def read_frames_device(self, first, count, hot_thresh=None, feed=None):
        import torch
        if feed is not None and self.raw_bit_depth == 12:
            def fill(dst):
                dst[:count * self.frame_bytes].reshape(count, self.frame_bytes)[:] = self.read_packed(first, count)
                return count * self.frame_bytes
            packed = feed.upload(fill)
            out = unpack_12bit(packed.view(count, self.frame_bytes), self.height, self.width, hot_thresh)
            feed.release()
            return out
        raw = self.read_packed(first, count)
        if self.raw_bit_depth == 12:
            return unpack_12bit(torch.as_tensor(raw).cuda(), self.height, self.width, hot_thresh)
        if self.raw_bit_depth == 10:
            if self.lut is None:
                raise ValueError("10-bit Cine files need the 10 -> 12 bit look-up table (lut= / UPSP_CINE_LUT)")
            fr = unpack_10bit(torch.as_tensor(raw).cuda(), self.height, self.width, self.lut)
        else:
            words = raw.view("<u2").reshape(count, self.height, self.width)[:, ::-1, :]   # flip(A, B, 0)
            fr = torch.as_tensor(np.ascontiguousarray(words)).cuda()
        if hot_thresh is None:
            return fr
        return fr, (fr.reshape(count, -1) >= hot_thresh).sum(1).to(torch.int32)
