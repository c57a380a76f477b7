// Video decode on the device: 12-bit packed frames (Photron MRAW, 12-bit Phantom CINE)
// -> u16.  upsp::unpack_12bit, cpp/lib/PSPVideo.cpp:134-149.
//
// Pure byte shuffling, HBM-bound: each lane reads 12 packed bytes (3 dwords, a wave reads
// 768 contiguous bytes) and writes 8 pixels (16 bytes, a wave writes 1 KiB).
#include <hip/hip_runtime.h>

#include "ktimer.h"
#include "upsp_internal.h"

namespace upsp {
namespace {

__device__ __forceinline__ unsigned byte_of(const unsigned w[3], int i)
{
    return (w[i >> 2] >> ((i & 3) * 8)) & 0xFFu;  // little-endian dwords
}

__global__ void __launch_bounds__(256)
    unpack12_kernel(const uint8_t *__restrict__ packed, size_t npix, uint16_t *__restrict__ out,
                    int hot_thresh, unsigned *__restrict__ hot_count)
{
    const size_t f = blockIdx.y;
    const size_t nbytes = npix / 2 * 3;
    const uint8_t *src = packed + f * nbytes;
    uint16_t *dst = out + f * npix;
    const size_t ngroups = npix / 8;  // 12 bytes -> 8 pixels
    const bool aligned = ((reinterpret_cast<size_t>(src) & 3) == 0) &&
                         ((reinterpret_cast<size_t>(dst) & 15) == 0);
    unsigned hot = 0;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups;
         g += (size_t)gridDim.x * blockDim.x) {
        unsigned w[3];
        if (aligned) {
            const unsigned *s32 = reinterpret_cast<const unsigned *>(src) + 3 * g;
            w[0] = s32[0]; w[1] = s32[1]; w[2] = s32[2];
        } else {
            for (int k = 0; k < 3; ++k)
                w[k] = src[12 * g + 4 * k] | (src[12 * g + 4 * k + 1] << 8) |
                       (src[12 * g + 4 * k + 2] << 16) | ((unsigned)src[12 * g + 4 * k + 3] << 24);
        }
        unsigned px[8];
#pragma unroll
        for (int t = 0; t < 4; ++t) {  // 3 bytes p,q,r -> (p<<4)|(q>>4) , ((q&0xF)<<8)|r
            const unsigned p = byte_of(w, 3 * t), q = byte_of(w, 3 * t + 1), r = byte_of(w, 3 * t + 2);
            px[2 * t] = (p << 4) | (q >> 4);
            px[2 * t + 1] = ((q & 0xFu) << 8) | r;
        }
        if (hot_count) {
#pragma unroll
            for (int k = 0; k < 8; ++k) hot += px[k] >= (unsigned)hot_thresh;
        }
        if (aligned) {
            uint4 o;
            o.x = px[0] | (px[1] << 16); o.y = px[2] | (px[3] << 16);
            o.z = px[4] | (px[5] << 16); o.w = px[6] | (px[7] << 16);
            reinterpret_cast<uint4 *>(dst)[g] = o;
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) dst[8 * g + k] = (uint16_t)px[k];
        }
    }
    // tail: remaining pixel pairs (npix not a multiple of 8)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (size_t i = ngroups * 8; i + 1 < npix; i += 2) {
            const size_t b = i / 2 * 3;
            const unsigned p = src[b], q = src[b + 1], r = src[b + 2];
            dst[i] = (uint16_t)((p << 4) | (q >> 4));
            dst[i + 1] = (uint16_t)(((q & 0xFu) << 8) | r);
            if (hot_count) hot += (dst[i] >= hot_thresh) + (dst[i + 1] >= hot_thresh);
        }
    }
    if (hot_count) {  // wave shuffle, then one atomic per workgroup (and only if non-zero)
        __shared__ unsigned wsum[4];
        for (int off = 32; off > 0; off >>= 1) hot += __shfl_down(hot, off);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = hot;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
            if (tot) atomicAdd(&hot_count[f], tot);
        }
    }
}

// 10-bit packed Phantom CINE frames: 5 bytes -> 4 pixels, MSBs first (upsp::unpack_10bit,
// cpp/lib/PSPVideo.cpp:111-132), then the camera's 10 -> 12 bit look-up table
// (CineReader::read_packed, cpp/lib/CineReader.cpp:409-423).  20 bytes in, 16 pixels out per lane.
__global__ void __launch_bounds__(256)
    unpack10_kernel(const uint8_t *__restrict__ packed, size_t npix, const uint16_t *__restrict__ lut,
                    uint16_t *__restrict__ out)
{
    const size_t f = blockIdx.y;
    const uint8_t *src = packed + f * (npix / 4 * 5);
    uint16_t *dst = out + f * npix;
    const size_t ngroups = npix / 16;   // 20 bytes -> 16 pixels
    const bool aligned = ((reinterpret_cast<size_t>(src) & 3) == 0) && ((reinterpret_cast<size_t>(dst) & 15) == 0);
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups;
         g += (size_t)gridDim.x * blockDim.x) {
        unsigned w[5];
        if (aligned) {
            const unsigned *s32 = reinterpret_cast<const unsigned *>(src) + 5 * g;
#pragma unroll
            for (int k = 0; k < 5; ++k) w[k] = s32[k];
        } else {
            for (int k = 0; k < 5; ++k)
                w[k] = src[20 * g + 4 * k] | (src[20 * g + 4 * k + 1] << 8) |
                       (src[20 * g + 4 * k + 2] << 16) | ((unsigned)src[20 * g + 4 * k + 3] << 24);
        }
        auto byte = [&](int i) { return (w[i >> 2] >> ((i & 3) * 8)) & 0xFFu; };
        unsigned px[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned p = byte(5 * q), qq = byte(5 * q + 1), r = byte(5 * q + 2), s_ = byte(5 * q + 3),
                           t = byte(5 * q + 4);
            px[4 * q] = (p << 2) | (qq >> 6);
            px[4 * q + 1] = ((qq & 0x3Fu) << 4) | (r >> 4);
            px[4 * q + 2] = ((r & 0x0Fu) << 6) | (s_ >> 2);
            px[4 * q + 3] = ((s_ & 0x03u) << 8) | t;
        }
        if (lut) {
#pragma unroll
            for (int k = 0; k < 16; ++k) px[k] = lut[px[k]];
        }
        if (aligned) {
            uint4 o0, o1;
            o0.x = px[0] | (px[1] << 16); o0.y = px[2] | (px[3] << 16);
            o0.z = px[4] | (px[5] << 16); o0.w = px[6] | (px[7] << 16);
            o1.x = px[8] | (px[9] << 16); o1.y = px[10] | (px[11] << 16);
            o1.z = px[12] | (px[13] << 16); o1.w = px[14] | (px[15] << 16);
            reinterpret_cast<uint4 *>(dst)[2 * g] = o0;
            reinterpret_cast<uint4 *>(dst)[2 * g + 1] = o1;
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) dst[16 * g + k] = (uint16_t)px[k];
        }
    }
    // tail: remaining groups of 4 pixels (npix a multiple of 4, not of 16)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (size_t i = ngroups * 16; i + 3 < npix; i += 4) {
            const size_t b = i / 4 * 5;
            const unsigned p = src[b], q = src[b + 1], r = src[b + 2], s_ = src[b + 3], t = src[b + 4];
            unsigned v[4] = {(p << 2) | (q >> 6), ((q & 0x3Fu) << 4) | (r >> 4), ((r & 0x0Fu) << 6) | (s_ >> 2),
                             ((s_ & 0x03u) << 8) | t};
            for (int k = 0; k < 4; ++k) dst[i + k] = (uint16_t)(lut ? lut[v[k]] : v[k]);
        }
    }
}

}  // namespace
}  // namespace upsp

using namespace upsp;

extern "C" int upsp_unpack_10bit(const uint8_t *d_packed, int nframes, size_t npix,
                                 const uint16_t *d_lut, uint16_t *d_frames, void *stream)
{
    if (nframes == 0 || npix == 0) return UPSP_OK;
    if (!d_packed || !d_frames || nframes < 0) return fail(UPSP_ERR_INVALID, "bad argument");
    if (npix & 3) return fail(UPSP_ERR_INVALID, "10-bit packing needs a pixel count divisible by 4");
    hipStream_t st = (hipStream_t)stream;
    size_t bx = (npix / 16 + 255) / 256;
    if (bx > 256) bx = 256;
    if (bx < 1) bx = 1;
    KTimed kt("unpack10_kernel", st);
    hipLaunchKernelGGL(unpack10_kernel, dim3((unsigned)bx, (unsigned)nframes), dim3(256), 0, st,
                       d_packed, npix, d_lut, d_frames);
    UPSP_HIP_CHECK(hipGetLastError());
    return UPSP_OK;
}

extern "C" int upsp_unpack_12bit(const uint8_t *d_packed, int nframes, size_t npix,
                                 uint16_t *d_frames, int hot_thresh, uint32_t *d_hot_count,
                                 void *stream)
{
    if (nframes == 0 || npix == 0) return UPSP_OK;
    if (!d_packed || !d_frames || nframes < 0) return fail(UPSP_ERR_INVALID, "bad argument");
    if (npix & 1) return fail(UPSP_ERR_INVALID, "12-bit packing needs an even pixel count");
    hipStream_t st = (hipStream_t)stream;
    size_t bx = (npix / 8 + 255) / 256;
    if (bx > 256) bx = 256;
    if (bx < 1) bx = 1;
    KTimed kt("unpack12_kernel", st);
    hipLaunchKernelGGL(unpack12_kernel, dim3((unsigned)bx, (unsigned)nframes), dim3(256), 0, st,
                       d_packed, npix, d_frames, hot_thresh, d_hot_count);
    UPSP_HIP_CHECK(hipGetLastError());
    return UPSP_OK;
}
