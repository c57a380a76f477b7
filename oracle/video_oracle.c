/*
 * video_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * upsp::unpack_12bit, cpp/lib/PSPVideo.cpp:134-149 : 3 bytes -> 2 pixels, MSBs first
 * (used by MrawReader::read_frame, cpp/lib/MrawReader.cpp:113-146, and by the 12-bit
 * path of CineReader, cpp/lib/CineReader.cpp:428).
 * Pinned by tests/test_video.py against the reference's own Python unpacker
 * (python/upsp/video/util.py:25-36) run on the reference's MRAW fixture
 * (cpp/test/mraw/12bitMRAW.mraw) -- tests/golden/make_golden_video.py.
 */
#include "upsp_oracle.h"

void orc_unpack_12bit(const uint8_t *packed, size_t nbytes, uint16_t *out)
{
    for (size_t i = 0; i + 2 < nbytes; i += 3, out += 2) {
        uint16_t p = packed[i], q = packed[i + 1], r = packed[i + 2];
        out[0] = (uint16_t)((p << 4) | (q >> 4));
        out[1] = (uint16_t)(((q & 0xF) << 8) | r);
    }
}

/* upsp::unpack_10bit, cpp/lib/PSPVideo.cpp:111-132 (+ the LUT of CineReader::read_packed,
 * cpp/lib/CineReader.cpp:409-423, when lut != NULL).  Pinned by tests/test_video.py against the
 * reference's Python unpack_10bpp (python/upsp/video/util.py:6-22), tests/golden/make_golden_video.py. */
void orc_unpack_10bit(const uint8_t *packed, size_t nbytes, const uint16_t *lut, uint16_t *out)
{
    for (size_t i = 0; i + 4 < nbytes; i += 5, out += 4) {
        uint16_t p = packed[i], q = packed[i + 1], r = packed[i + 2], s = packed[i + 3], t = packed[i + 4];
        uint16_t v[4] = {(uint16_t)((p << 2) | (q >> 6)), (uint16_t)(((q & 0x3F) << 4) | (r >> 4)),
                         (uint16_t)(((r & 0x0F) << 6) | (s >> 2)), (uint16_t)(((s & 0x03) << 8) | t)};
        for (int k = 0; k < 4; ++k) out[k] = lut ? lut[v[k]] : v[k];
    }
}
