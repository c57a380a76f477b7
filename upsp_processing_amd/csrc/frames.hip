// Per-frame kernels of the psp_process phase-1 loop on MI355X (gfx950):
// hot-pixel repair, nearest-pixel projection (gather) with the double
// accumulators, finals, tiled transpose, multi-camera weights.
//
// All of this is HBM-bound byte/gather work (no MFMA): the kernels are laid out
// for coalesced 16-byte-per-lane streaming of the u16 frames, one surface node
// per lane for the gather (node-contiguous row stores), register-resident
// accumulators across a batch of frames, and an LDS tile for the transpose.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ktimer.h"
#include "pipeline.h"
#include "upsp_internal.h"

namespace upsp {
namespace {

constexpr int kHotCap = 64;  // recorded hot-pixel positions per frame
// ticket counters sit 128 B apart: same-line atomics serialise (~88 per us per line)
constexpr int kTicketStride = 32;

// ---------------------------------------------------------------- hot pixels --
__device__ void fix_frame(uint16_t *__restrict__ img, int rows, int cols, int min_change, int max_hot,
                          unsigned n, unsigned *p, int32_t *status_f);

// Pass 1 of upsp::fix_hot_pixels (cpp/utils/cv_extras.cpp:237-247): find pixels
// >= thresh.  16 bytes (8 pixels) per lane per step; the frame is otherwise only
// streamed through (this is the one compulsory full read of a frame).
__global__ void __launch_bounds__(256)
    hot_scan_kernel(uint16_t *frames, size_t npix, int thresh, unsigned *__restrict__ count,
                    unsigned *__restrict__ pos, unsigned *__restrict__ done, int rows, int cols,
                    int min_change, int max_hot, int32_t *__restrict__ status, const unsigned *__restrict__ only)
{
    const size_t f = blockIdx.y;
    // `only` (may be null): per-frame flags set by a kernel that has already seen every pixel (the registration's
    // pre-blur): frames without a pixel >= thresh have nothing to scan for
    if (only && !only[f]) return;
    const uint4 *__restrict__ src = reinterpret_cast<const uint4 *>(frames + f * npix);
    const size_t nvec = npix / 8;
    const unsigned th = (unsigned)thresh;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    int wrote = 0;
    auto check = [&](const uint4 v, size_t i) {
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
        bool any = false;  // quick reject: any half-word >= thresh ?
#pragma unroll
        for (int k = 0; k < 4; ++k) any |= ((w[k] & 0xFFFFu) >= th) | ((w[k] >> 16) >= th);
        if (any) {
            // a saturated frame holds thousands of such pixels; once the (monotonic) counter is
            // past max_hot the verdict "too many, leave the frame alone" is settled and further
            // same-address atomics would only serialise (~88 per us)
            if (__hip_atomic_load(&count[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > (unsigned)max_hot) return;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned px = (w[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu;
                if (px >= th) {
                    const unsigned slot = atomicAdd(&count[f], 1u);
                    if (slot < (unsigned)kHotCap)
                        wrote |= (int)atomicExch(&pos[f * kHotCap + slot], (unsigned)(i * 8 + k)) | 1;
                }
            }
        }
    };
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    // four independent 16-byte loads in flight per lane
    for (; i + 3 * stride < nvec; i += 4 * stride) {
        const uint4 v0 = src[i], v1 = src[i + stride], v2 = src[i + 2 * stride], v3 = src[i + 3 * stride];
        check(v0, i);
        check(v1, i + stride);
        check(v2, i + 2 * stride);
        check(v3, i + 3 * stride);
    }
    for (; i < nvec; i += stride) check(src[i], i);
    // tail pixels when npix is not a multiple of 8
    if (blockIdx.x == 0 && threadIdx.x < (npix & 7)) {
        const size_t p = nvec * 8 + threadIdx.x;
        if (frames[f * npix + p] >= th) {
            const unsigned slot = atomicAdd(&count[f], 1u);
            if (slot < (unsigned)kHotCap) wrote |= (int)atomicExch(&pos[f * kHotCap + slot], (unsigned)p) | 1;
        }
    }
    // Pass 2 in the same launch: the workgroup that finishes a frame last repairs it (lane 0).
    // Counters and positions only ever move through device-scope atomics (performed at the
    // memory side, coherent across the 8 XCDs) -- an agent-scope fence would cost an L2
    // write-back per workgroup on this part.  `wrote` carries the atomics' return values, so
    // they have completed before the barrier; the ticket follows the barrier.  The last
    // workgroup also zeroes the counters for the next launch (no memset between sub-batches).
    const int any_wrote = __syncthreads_or(wrote);
    if (threadIdx.x == 0) {
        const unsigned ticket = atomicAdd(&done[f * kTicketStride], (unsigned)(any_wrote >= 0));   // always 1
        if (ticket == gridDim.x - 1) {
            const unsigned n = atomicExch(&count[f], 0u);
            atomicExch(&done[f * kTicketStride], 0u);
            unsigned p[kHotCap];
            const unsigned m = n < (unsigned)kHotCap ? n : (unsigned)kHotCap;
            for (unsigned i = 0; i < m; ++i) p[i] = atomicAdd(&pos[f * kHotCap + i], 0u);
            fix_frame(frames + f * npix, rows, cols, min_change, max_hot, n, p, status ? status + f : nullptr);
        }
    }
}

// Pass 2 (cv_extras.cpp:249-274): sequential repair in scan order, one lane per frame.
__device__ void fix_frame(uint16_t *__restrict__ img, int rows, int cols, int min_change, int max_hot,
                          unsigned n, unsigned *p, int32_t *status_f)
{
    if (n > (unsigned)max_hot) {  // "too many pixels look hot": frame untouched
        if (status_f) *status_f = -1;
        return;
    }
    for (unsigned i = 1; i < n; ++i) {  // scan order
        const unsigned v = p[i];
        unsigned j = i;
        while (j > 0 && p[j - 1] > v) {
            p[j] = p[j - 1];
            --j;
        }
        p[j] = v;
    }
    int replaced = 0;
    for (unsigned h = 0; h < n; ++h) {
        const unsigned ph = p[h];
        const int row = (int)(ph / (unsigned)cols), col = (int)(ph % (unsigned)cols);
        uint16_t vals[4];
        unsigned nv = 0;
        if (row > 0) vals[nv++] = img[(size_t)(row - 1) * cols + col];
        if (col > 0) vals[nv++] = img[(size_t)row * cols + col - 1];
        if (row < rows - 1) vals[nv++] = img[(size_t)(row + 1) * cols + col];
        if (col < cols - 1) vals[nv++] = img[(size_t)row * cols + col + 1];
        for (unsigned i = 1; i < nv; ++i) {
            const uint16_t v = vals[i];
            unsigned j = i;
            while (j > 0 && vals[j - 1] > v) {
                vals[j] = vals[j - 1];
                --j;
            }
            vals[j] = v;
        }
        const uint16_t old_val = img[(size_t)row * cols + col];
        const uint16_t new_val = vals[nv / 2];
        if ((int)old_val - (int)new_val > min_change) {
            img[(size_t)row * cols + col] = new_val;
            ++replaced;
        }
    }
    if (status_f) *status_f = replaced;
}

// ------------------------------------------------------------------- gather --
// upsp::project_frame with <= 1 entry per row (cpp/lib/projection.ipp:883-908)
// followed by the frame-loop tail (cpp/exec/psp_process.cpp:1813-1843):
//   sol[n] = sum_c w_c[n] * f32(img_c[pix_c[n]])   (float, camera order)
//   sol[n] = NaN for skipped nodes
//   sumsq[n] += (double)(sol*sol) ; sum[n] += sol          (double)
// One node per lane; pix / weight stay in registers for the whole batch of
// frames, the accumulators too (one 16-byte read-modify-write per node and batch).
struct GatherArgs {
    const void *img[kMaxCams];      // frame 0 of the batch, camera c
    const int32_t *pix[kMaxCams];
    const float *weight[kMaxCams];  // may be null (= 1)
    int is_f32[kMaxCams];
    int ncams;
    size_t npix;
    const int32_t *src;            // overlap source map or null
    const int32_t *rowmap;         // packed rows of rows_t or null
    uint16_t *rows_t16;            // node-major series as u16 (instead of rows_t) or null
};

template <int NCAMS>
__global__ void __launch_bounds__(256)
    gather_kernel(GatherArgs a, const uint8_t *__restrict__ skipped, unsigned nnodes, int nframes,
                  float *__restrict__ rows, double *__restrict__ sum, double *__restrict__ sumsq)
{
    const unsigned n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nnodes) return;
    const int nc = NCAMS > 0 ? NCAMS : a.ncams;
    int32_t px[NCAMS > 0 ? NCAMS : kMaxCams];
    float w[NCAMS > 0 ? NCAMS : kMaxCams];
#pragma unroll
    for (int c = 0; c < nc; ++c) {
        px[c] = a.pix[c][n];
        w[c] = a.weight[c] ? a.weight[c][n] : 1.0f;
    }
    const bool skip = skipped && skipped[n];
    double s = 0.0, ss = 0.0;
    const float qnan = __builtin_nanf("");
#pragma unroll 4
    for (int f = 0; f < nframes; ++f) {
        float sol = 0.0f;
#pragma unroll
        for (int c = 0; c < nc; ++c) {
            float v = 0.0f;
            if (px[c] >= 0) {
                const size_t off = (size_t)f * a.npix + (size_t)px[c];
                const float pxv = a.is_f32[c] ? reinterpret_cast<const float *>(a.img[c])[off]
                                              : (float)reinterpret_cast<const uint16_t *>(a.img[c])[off];
                v = 0.0f + w[c] * pxv;
            }
            sol = (c == 0) ? v : sol + v;
        }
        if (skip) sol = qnan;
        s += (double)sol;
        ss += (double)(sol * sol);
        if (rows) rows[(size_t)f * nnodes + n] = sol;
    }
    sum[n] += s;
    sumsq[n] += ss;
    if (a.src && rows && (unsigned)a.src[n] != n) {   // adjust_solution: stored value = source node's
        const unsigned ns = (unsigned)a.src[n];
        const bool skip2 = skipped && skipped[ns];
        for (int f = 0; f < nframes; ++f) {
            float sol = 0.0f;
            for (int c = 0; c < nc; ++c) {
                const int32_t p2 = a.pix[c][ns];
                float v = 0.0f;
                if (p2 >= 0) {
                    const size_t off = (size_t)f * a.npix + (size_t)p2;
                    const float pxv = a.is_f32[c] ? reinterpret_cast<const float *>(a.img[c])[off]
                                                  : (float)reinterpret_cast<const uint16_t *>(a.img[c])[off];
                    v = 0.0f + (a.weight[c] ? a.weight[c][ns] : 1.0f) * pxv;
                }
                sol = (c == 0) ? v : sol + v;
            }
            rows[(size_t)f * nnodes + n] = skip2 ? qnan : sol;
        }
    }
}

// Fused gather + accumulate + transpose for a sub-batch of <= 64 frames.
// A workgroup owns 64 consecutive nodes.  Phase 1: lane = node, wave g gathers frames
// g, g+4, ... into an LDS tile and keeps its part of the double sums.  Phase 2: the
// tile is written node-major (256-byte rows of the [N x F] time series =
// intensity_transpose layout, psp_process.cpp:2027-2032) and, if requested,
// frame-major (intensity_buf rows) -- both fully coalesced, no second pass over HBM.
template <int NCAMS, bool kStreamStores, bool kHasSrc>
__global__ void __launch_bounds__(256)
    gather_tile_kernel(GatherArgs a, const uint8_t *__restrict__ skipped, unsigned nnodes,
                       int nframes, float *__restrict__ rows, float *__restrict__ rows_t,
                       long long ld_t, double *__restrict__ sum, double *__restrict__ sumsq)
{
    __shared__ float tile[64][65];        // [frame][node]
    __shared__ double part[2][4][64];     // [sum|sumsq][wave][node]
    // the wave index as a scalar: everything that depends only on it (frame < nframes) becomes a
    // uniform branch instead of an exec-mask sequence
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned n0 = blockIdx.x * 64u, n = n0 + lane;
    const int nc = NCAMS > 0 ? NCAMS : a.ncams;
    const bool live = n < nnodes;
    int32_t px[NCAMS > 0 ? NCAMS : kMaxCams];
    float w[NCAMS > 0 ? NCAMS : kMaxCams];
#pragma unroll
    for (int c = 0; c < nc; ++c) {
        px[c] = live ? a.pix[c][n] : -1;
        w[c] = (live && a.weight[c]) ? a.weight[c][n] : 1.0f;
    }
    const bool skip = live && skipped && skipped[n];
    const float qnan = __builtin_nanf("");
    double s = 0.0, ss = 0.0;
    // P3D zone overlaps (adjust_solution, psp_process.cpp:1833-1835): the accumulators take the
    // node's own value, the stored series the value of its source node.  Rare (zone edges).
    const unsigned ns = (kHasSrc && live) ? (unsigned)a.src[n] : n;
    const bool alt = kHasSrc && ns != n;
    // 16 frames per lane (f = wave, wave+4, ...): issue every gather before using any
    float sol[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int f = wave + 4 * i;
        float acc = 0.0f;
        if (f < nframes) {
#pragma unroll
            for (int c = 0; c < nc; ++c) {
                float v = 0.0f;
                if (px[c] >= 0) {
                    const size_t off = (size_t)f * a.npix + (size_t)px[c];
                    const float pxv = a.is_f32[c] ? reinterpret_cast<const float *>(a.img[c])[off]
                                                  : (float)reinterpret_cast<const uint16_t *>(a.img[c])[off];
                    v = 0.0f + w[c] * pxv;
                }
                acc = (c == 0) ? v : acc + v;
            }
        }
        sol[i] = skip ? qnan : acc;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int f = wave + 4 * i;
        if (f < nframes) {
            s += (double)sol[i];
            ss += (double)(sol[i] * sol[i]);
            tile[f][lane] = sol[i];
        }
    }
    if (alt) {   // overwrite the tile column with the source node's solution
        const bool skip2 = skipped && skipped[ns];
        for (int i = 0; i < 16; ++i) {
            const int f = wave + 4 * i;
            if (f >= nframes) break;
            float acc = 0.0f;
            for (int c = 0; c < nc; ++c) {
                const int32_t p2 = a.pix[c][ns];
                float v = 0.0f;
                if (p2 >= 0) {
                    const size_t off = (size_t)f * a.npix + (size_t)p2;
                    const float pxv = a.is_f32[c] ? reinterpret_cast<const float *>(a.img[c])[off]
                                                  : (float)reinterpret_cast<const uint16_t *>(a.img[c])[off];
                    v = 0.0f + (a.weight[c] ? a.weight[c][ns] : 1.0f) * pxv;
                }
                acc = (c == 0) ? v : acc + v;
            }
            tile[f][lane] = skip2 ? qnan : acc;
        }
    }
    part[0][wave][lane] = s;
    part[1][wave][lane] = ss;
    __syncthreads();
    if (wave == 0 && live) {
        // fixed order -> bit-reproducible accumulators
        sum[n] += ((part[0][0][lane] + part[0][1][lane]) + part[0][2][lane]) + part[0][3][lane];
        sumsq[n] += ((part[1][0][lane] + part[1][1][lane]) + part[1][2][lane]) + part[1][3][lane];
    }
    if (rows_t) {
        const bool vec_ok = ((ld_t & 3) == 0) && ((reinterpret_cast<size_t>(rows_t) & 15) == 0);
        if (vec_ok) {
            // 16 lanes x 16 B = one 256-byte row segment, 4 rows per wave instruction
            const int c4 = (lane & 15) * 4, rsub = lane >> 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int j = wave * 16 + i * 4 + rsub;  // node within the tile
                const unsigned nn = n0 + (unsigned)j;
                // packed series (multi-GPU exchange sends only rows some camera sees)
                const long long row = (nn < nnodes) ? (a.rowmap ? (long long)a.rowmap[nn] : (long long)nn) : -1;
                if (row >= 0 && c4 < nframes) {
                    float4 v;
                    v.x = tile[c4][j];
                    v.y = tile[c4 + 1][j];
                    v.z = tile[c4 + 2][j];
                    v.w = tile[c4 + 3][j];
                    float *dst = rows_t + row * ld_t + c4;
                    if (c4 + 3 < nframes) {
                        // streaming store: the series is not read again by the frame loop, keep
                        // L2 / Infinity Cache for the frames the gathers are reading
                        typedef float v4f __attribute__((ext_vector_type(4)));
                        v4f nv = {v.x, v.y, v.z, v.w};
                        if (kStreamStores)
                            __builtin_nontemporal_store(nv, reinterpret_cast<v4f *>(dst));
                        else
                            *reinterpret_cast<v4f *>(dst) = nv;
                    } else {
                        dst[0] = v.x;
                        if (c4 + 1 < nframes) dst[1] = v.y;
                        if (c4 + 2 < nframes) dst[2] = v.z;
                    }
                }
            }
        } else {  // lane = frame, wave g writes nodes g, g+4, ...: 256-byte row segments
            for (int j = wave; j < 64; j += 4) {
                const unsigned nn = n0 + (unsigned)j;
                const long long row = (nn < nnodes) ? (a.rowmap ? (long long)a.rowmap[nn] : (long long)nn) : -1;
                if (row >= 0 && lane < nframes) rows_t[row * ld_t + lane] = tile[lane][j];
            }
        }
    }
    if (a.rows_t16) {
        // the same row segments as u16 (time-series exchange of integer-valued series: half the
        // bytes); NaN (node no camera sees) has no u16 encoding and is stored as 0
        uint16_t *rt = a.rows_t16;
        if (((ld_t & 3) == 0) && ((reinterpret_cast<size_t>(rt) & 7) == 0)) {
            const int c4 = (lane & 15) * 4, rsub = lane >> 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int j = wave * 16 + i * 4 + rsub;
                const unsigned nn = n0 + (unsigned)j;
                const long long row = (nn < nnodes) ? (a.rowmap ? (long long)a.rowmap[nn] : (long long)nn) : -1;
                if (row >= 0 && c4 < nframes) {
                    const unsigned v0 = (unsigned)fmaxf(tile[c4][j], 0.0f), v1 = (unsigned)fmaxf(tile[c4 + 1][j], 0.0f),
                                   v2 = (unsigned)fmaxf(tile[c4 + 2][j], 0.0f), v3 = (unsigned)fmaxf(tile[c4 + 3][j], 0.0f);
                    uint16_t *dst = rt + row * ld_t + c4;
                    if (c4 + 3 < nframes) {
                        typedef unsigned v2u __attribute__((ext_vector_type(2)));
                        const v2u nv = {v0 | (v1 << 16), v2 | (v3 << 16)};
                        if (kStreamStores)
                            __builtin_nontemporal_store(nv, reinterpret_cast<v2u *>(dst));
                        else
                            *reinterpret_cast<v2u *>(dst) = nv;
                    } else {
                        dst[0] = (uint16_t)v0;
                        if (c4 + 1 < nframes) dst[1] = (uint16_t)v1;
                        if (c4 + 2 < nframes) dst[2] = (uint16_t)v2;
                    }
                }
            }
        } else {
            for (int j = wave; j < 64; j += 4) {
                const unsigned nn = n0 + (unsigned)j;
                const long long row = (nn < nnodes) ? (a.rowmap ? (long long)a.rowmap[nn] : (long long)nn) : -1;
                if (row >= 0 && lane < nframes) rt[row * ld_t + lane] = (uint16_t)(unsigned)fmaxf(tile[lane][j], 0.0f);
            }
        }
    }
    if (rows) {    // lane = node, wave g writes frames g, g+4, ...
        for (int f = wave; f < nframes; f += 4)
            if (live) rows[(size_t)f * nnodes + n] = tile[f][lane];
    }
}

// The same tile for the common case -- one camera, no weight vector, u16 frames (raw or registered):
// every value is an exact 16-bit integer, so the LDS tile holds u16 pairs (8.4 KB instead of
// 16.6 KB) and a workgroup is WAVES x 64 lanes with 64 / WAVES gathers in flight per lane.  The
// gather is bound by the latency of its dependent chain (pix -> pixel -> LDS -> store) times the
// workgroups a CU can hold, not by bytes (storing a fifth of the bytes did not change its time):
// the small tile lets a CU hold 8 workgroups instead of 7 and halves the LDS traffic; 64-frame
// launch 46 -> 38 us on MI355X.  Sums of 16-bit integers (and of their float squares) are exact in double, so
// sum / sumsq are bit-identical to gather_tile_kernel's whatever the order.
template <int WAVES, bool kStreamStores, bool kHasSrc>
__global__ void __launch_bounds__(WAVES * 64)
    gather_tile16_kernel(GatherArgs a, const uint8_t *__restrict__ skipped, unsigned nnodes,
                         int nframes, float *__restrict__ rows, float *__restrict__ rows_t,
                         long long ld_t, double *__restrict__ sum, double *__restrict__ sumsq)
{
    constexpr int kPitch = 66;            // 2 rows apart = 4 banks apart: conflict-free both ways
    constexpr int NP = 32 / WAVES;        // frame pairs per lane
    constexpr int RPW = 64 / WAVES;       // rows (nodes) each wave writes
    __shared__ unsigned tile[32][kPitch]; // [frame pair p][node]: frame 2p | frame 2p+1 << 16
    __shared__ double part[2][WAVES][64];
    __shared__ unsigned char skipf[64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned n0 = blockIdx.x * 64u, n = n0 + lane;
    const bool live = n < nnodes;
    const int32_t px = live ? a.pix[0][n] : -1;
    const bool skip = live && skipped && skipped[n];
    const unsigned ns = (kHasSrc && live) ? (unsigned)a.src[n] : n;
    const bool alt = kHasSrc && ns != n;
    const uint16_t *__restrict__ img = reinterpret_cast<const uint16_t *>(a.img[0]);
    const float qnan = __builtin_nanf("");
    unsigned pk[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {        // every gather is issued before any is used
        const int f0 = 2 * (wave + WAVES * i);
        unsigned lo = 0, hi = 0;
        if (px >= 0) {
            if (f0 < nframes) lo = img[(size_t)f0 * a.npix + (size_t)px];
            if (f0 + 1 < nframes) hi = img[(size_t)(f0 + 1) * a.npix + (size_t)px];
        }
        pk[i] = lo | (hi << 16);
    }
    double s = 0.0, ss = 0.0;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int p = wave + WAVES * i;
        if (2 * p < nframes) {
            const float v = skip ? qnan : (float)(pk[i] & 0xFFFFu);
            s += (double)v;
            ss += (double)(v * v);
        }
        if (2 * p + 1 < nframes) {
            const float v = skip ? qnan : (float)(pk[i] >> 16);
            s += (double)v;
            ss += (double)(v * v);
        }
        tile[p][lane] = pk[i];
    }
    bool skip_out = skip;
    if (alt) {   // adjust_solution: the stored series is the source node's
        const int32_t p2 = a.pix[0][ns];
        skip_out = skipped && skipped[ns];
        for (int i = 0; i < NP; ++i) {
            const int f0 = 2 * (wave + WAVES * i);
            if (f0 >= nframes) break;
            unsigned lo = 0, hi = 0;
            if (p2 >= 0) {
                lo = img[(size_t)f0 * a.npix + (size_t)p2];
                if (f0 + 1 < nframes) hi = img[(size_t)(f0 + 1) * a.npix + (size_t)p2];
            }
            tile[f0 >> 1][lane] = lo | (hi << 16);
        }
    }
    if (wave == 0) skipf[lane] = skip_out ? 1 : 0;
    part[0][wave][lane] = s;
    part[1][wave][lane] = ss;
    __syncthreads();
    if (wave == 0 && live) {
        double t0 = part[0][0][lane], t1 = part[1][0][lane];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) {
            t0 += part[0][w][lane];
            t1 += part[1][w][lane];
        }
        sum[n] += t0;
        sumsq[n] += t1;
    }
    uint16_t *__restrict__ rt16 = a.rows_t16;
    if (rows_t || rt16) {
        const bool vec_ok = rows_t ? (((ld_t & 3) == 0) && ((reinterpret_cast<size_t>(rows_t) & 15) == 0))
                                   : (((ld_t & 3) == 0) && ((reinterpret_cast<size_t>(rt16) & 7) == 0));
        if (vec_ok) {
            // 16 lanes x 4 frames = one row segment, 4 rows per wave instruction
            const int c4 = (lane & 15) * 4, rsub = lane >> 4;
#pragma unroll
            for (int i = 0; i < RPW / 4; ++i) {
                const int j = wave * RPW + i * 4 + rsub;
                const unsigned nn = n0 + (unsigned)j;
                const long long row = (nn < nnodes) ? (a.rowmap ? (long long)a.rowmap[nn] : (long long)nn) : -1;
                if (row >= 0 && c4 < nframes) {
                    const unsigned d0 = tile[c4 >> 1][j], d1 = tile[(c4 >> 1) + 1][j];
                    const bool sk = skipf[j] != 0;
                    if (rows_t) {
                        typedef float v4f __attribute__((ext_vector_type(4)));
                        v4f nv;
                        nv.x = sk ? qnan : (float)(d0 & 0xFFFFu);
                        nv.y = sk ? qnan : (float)(d0 >> 16);
                        nv.z = sk ? qnan : (float)(d1 & 0xFFFFu);
                        nv.w = sk ? qnan : (float)(d1 >> 16);
                        float *dst = rows_t + row * ld_t + c4;
                        if (c4 + 3 < nframes) {
                            if (kStreamStores)
                                __builtin_nontemporal_store(nv, reinterpret_cast<v4f *>(dst));
                            else
                                *reinterpret_cast<v4f *>(dst) = nv;
                        } else {
                            dst[0] = nv.x;
                            if (c4 + 1 < nframes) dst[1] = nv.y;
                            if (c4 + 2 < nframes) dst[2] = nv.z;
                        }
                    } else {   // u16 series (exchange wire format): NaN rows are stored as 0
                        typedef unsigned v2u __attribute__((ext_vector_type(2)));
                        const v2u nv = {sk ? 0u : d0, sk ? 0u : d1};
                        uint16_t *dst = rt16 + row * ld_t + c4;
                        if (c4 + 3 < nframes) {
                            if (kStreamStores)
                                __builtin_nontemporal_store(nv, reinterpret_cast<v2u *>(dst));
                            else
                                *reinterpret_cast<v2u *>(dst) = nv;
                        } else {
                            dst[0] = (uint16_t)(nv.x & 0xFFFFu);
                            if (c4 + 1 < nframes) dst[1] = (uint16_t)(nv.x >> 16);
                            if (c4 + 2 < nframes) dst[2] = (uint16_t)(nv.y & 0xFFFFu);
                        }
                    }
                }
            }
        } else {  // lane = frame: 256-byte (128-byte) row segments, any alignment
            for (int j = wave; j < 64; j += WAVES) {
                const unsigned nn = n0 + (unsigned)j;
                const long long row = (nn < nnodes) ? (a.rowmap ? (long long)a.rowmap[nn] : (long long)nn) : -1;
                if (row >= 0 && lane < nframes) {
                    const unsigned v = (tile[lane >> 1][j] >> (16 * (lane & 1))) & 0xFFFFu;
                    const bool sk = skipf[j] != 0;
                    if (rows_t) rows_t[row * ld_t + lane] = sk ? qnan : (float)v;
                    else rt16[row * ld_t + lane] = sk ? (uint16_t)0 : (uint16_t)v;
                }
            }
        }
    }
    if (rows) {    // lane = node, wave g writes frames g, g+WAVES, ...
        const bool sk = skipf[lane] != 0;
        for (int f = wave; f < nframes; f += WAVES)
            if (live) rows[(size_t)f * nnodes + n] = sk ? qnan : (float)((tile[f >> 1][lane] >> (16 * (f & 1))) & 0xFFFFu);
    }
}

// --------------------------------------------- streamed scan + projection (2 passes) --
// One camera, no weights, u16 frames, no image stage, node-major series.  The gather above reads
// every pixel some node sees a second time, lane by lane (2-byte loads, one address per lane, ~1 TA
// cycle per lane, and only fast while the 64-frame sub-batch still sits in the Infinity Cache).
// Here the frame data is turned round on its single way through the chip:
//
//   pass A (scan_compact_kernel, one launch per <= 1024 frames): a workgroup owns a TILE of 128
//     consecutive pixels of one 64-frame group and streams it through registers with coalesced 256-byte
//     loads, counting the hot pixels on the way (that is the scan).  The few pixels of the tile that some
//     node reads ("active" pixels: 66 k of 1 M on the bench model) leave the tile TRANSPOSED through LDS:
//     64 consecutive u16 per active pixel and group into the pixel's series in the compact buffer, which
//     holds all frames of the call.  The work per tile is bounded by its 128 pixels, however many nodes
//     read them.
//   pass B (node_rows_kernel, one launch per <= 1024 frames): a workgroup sweeps whole row pieces --
//     256 / 128 / 64 lanes per row, a lane converts 4 frames (8-byte load from the pixel's series, 16-byte
//     store) -- adds the accumulators and fills the constant rows of nodes without a pixel (NaN when no
//     camera sees them, psp_process.cpp:1821-1825).  Nodes in mesh order, rows in row order: a pure
//     streaming write.  (node_stream_kernel is round 1's form of this pass: 16 lanes per node, 1-KB pieces.)
//
// HBM sees every frame byte once (pass A, reads + 6 % compact stores) and every series byte once (pass B,
// writes + the compact series); nothing depends on the frames staying in a cache between the passes.  (A
// single fused pass that pushes the pixels straight to the nodes' rows was built first: its node work piles
// up on the 10 % of the tiles that cover the model, and reads and writes interleave in HBM -- level with
// scan + gather at 1 Mpix.  DESIGN.md section 4.)
//   * hot pixels: pass A cannot know a frame's count before the whole frame has gone by, so it stores
//     the pixels as they are and counts; between the passes hot_repair_kernel repairs the (rare) frames
//     with 1..max_hot hot pixels in place exactly like fix_frame and lists the replaced pixels, and
//     writes them into the compact series as well: pass B reads repaired values.
//     (Rounds 2-4 repaired behind pass B and corrected the rows and accumulators of the nodes on the
//     replaced pixels; the several-camera schedule still does, launch_hot_fixup_multi.)
constexpr int kFusedPix = 128;     // pixels per tile
constexpr int kFusedPitch = 65;    // dwords per LDS row: 64 (128 px) + 1 pad

// active-pixel map, step 1: flag[pixel] = 1 for pixels some node reads.  The flags need no clearing between two maps: after
// step 2 every flag is 0 or 0x80 | rank, never 1 (the caller zeroes the array once, when it allocates it)
__global__ void __launch_bounds__(256)
    amap_mark_kernel(const int32_t *__restrict__ pix, unsigned nnodes, uint8_t *__restrict__ flag)
{
    const unsigned n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nnodes) return;
    const int32_t p = pix[n];
    if (p >= 0) flag[p] = 1;
}

// step 2: rank of every active pixel inside its 128-pixel tile (flag becomes 0x80 | rank) and the
// number of active pixels per tile.  One wave per 64 pixels, two waves per tile.
__global__ void __launch_bounds__(256)
    amap_rank_kernel(uint8_t *__restrict__ flag, size_t npix, unsigned *__restrict__ tile_cnt)
{
    __shared__ unsigned wave_cnt[4];
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool act = p < npix && flag[p] == 1;                      // (marked in step 1; 0x80 | rank of the previous map: not)
    const unsigned long long m = __ballot(act);
    if (lane == 0) wave_cnt[wave] = (unsigned)__popcll(m);
    __syncthreads();
    const unsigned before = (wave & 1) ? wave_cnt[wave - 1] : 0u;   // first half of the same tile
    if (p < npix) flag[p] = act ? (uint8_t)(0x80u | (before + (unsigned)__popcll(m & ((1ull << lane) - 1ull)))) : (uint8_t)0;
    if (lane == 0 && !(wave & 1)) {
        const size_t tile = (size_t)blockIdx.x * 2 + (wave >> 1);
        if (tile * kFusedPix < npix) tile_cnt[tile] = wave_cnt[wave] + wave_cnt[wave + 1];
    }
}

// exclusive scan of the tile counts (one workgroup): off[0..ntiles], off[ntiles] = total; and of the
// "tile has an active pixel" flags: arank[0..ntiles], arank[ntiles] = number of active tiles.
// NT threads, PT tiles per thread.  Four waves, not sixteen: the launch runs on a side stream beside pass A / pass B, and a
// 1024-thread workgroup waits there until ONE compute unit has sixteen wave slots free at the same moment (kernel trace: 40 us,
// 188 us beside pass B, for a few microseconds of work).
template <int NT, int PT>
__global__ void __launch_bounds__(NT)
    tilemap_scan_kernel(const unsigned *__restrict__ cnt, unsigned *__restrict__ off, unsigned *__restrict__ arank,
                        unsigned ntiles)
{
    // PT tiles per thread; the NT per-thread totals are scanned by wave shuffles, the wave totals by the first wave:
    // three barriers per NT x PT tiles (the Hillis-Steele version through LDS had twenty and took 13 us of every projection)
    constexpr int NW = NT / 64;
    static_assert(NW >= 1 && NW <= 16 && NT % 64 == 0, "whole waves, at most sixteen");
    __shared__ unsigned long long wsum[16];     // low word: pixels, high word: active tiles
    __shared__ unsigned long long carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0ull;
    __syncthreads();
    for (unsigned base = 0; base < ntiles; base += (unsigned)(NT * PT)) {
        const unsigned i0 = base + threadIdx.x * (unsigned)PT;
        unsigned v[PT];
        unsigned long long tot = 0;
#pragma unroll
        for (int k = 0; k < PT; ++k) {
            v[k] = i0 + k < ntiles ? cnt[i0 + k] : 0u;
            tot += (unsigned long long)v[k] + ((unsigned long long)(v[k] != 0u) << 32);
        }
        unsigned long long x = tot;               // inclusive scan inside the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long y = __shfl_up(x, d);
            if (lane >= d) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        if (wave == 0) {
            unsigned long long w = lane < NW ? wsum[lane] : 0ull;
#pragma unroll
            for (int d = 1; d < NW; d <<= 1) {
                const unsigned long long y = __shfl_up(w, d);
                if (lane >= d) w += y;
            }
            if (lane < NW) wsum[lane] = w;        // inclusive totals of the waves
        }
        __syncthreads();
        unsigned long long run = carry + (wave ? wsum[wave - 1] : 0ull) + x - tot;
#pragma unroll
        for (int k = 0; k < PT; ++k)
            if (i0 + k < ntiles) {
                off[i0 + k] = (unsigned)(run & 0xFFFFFFFFull);
                if (arank) arank[i0 + k] = (unsigned)(run >> 32);
                run += (unsigned long long)v[k] + ((unsigned long long)(v[k] != 0u) << 32);
            }
        __syncthreads();
        if (threadIdx.x == 0) carry += wsum[NW - 1];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        off[ntiles] = (unsigned)(carry & 0xFFFFFFFFull);
        if (arank) arank[ntiles] = (unsigned)(carry >> 32);
    }
}

// Visiting order of the tiles in pass A.  The tiles with active pixels are the only ones that write
// (their compact series), and on a real model they are neighbours (the model covers a band of the
// frame): visited in natural order their stores arrive in bursts inside the read stream, which costs
// pass A 0.39 instead of 0.35 ms per 1024 frames (tools/probe/passA_stages.hip).  order[] spreads the
// A active tiles evenly over the n slots (active tile j -> slot floor(j n / A)) and fills the other
// slots with the inactive tiles, both kinds in their natural order (reads of neighbouring workgroups
// stay neighbours, the compact rows of successive active tiles too).
// (the j-th active / inactive tile is found by a binary search in the active-tile ranks: arank[t] = active tiles before
//  tile t, monotone, 32 KB -- thirteen cached loads per thread instead of a kernel that writes the two lists first)
// Order in which pass A visits the tiles.  Default since the second half of round 5: every frame group's tiles that nobody reads
// first -- a pure read stream, no stores -- and the ACTIVE tiles of all groups behind them (scan_compact_kernel, one-dimensional
// grid).  The 135 MB of compact stores cost 40-90 us wherever they sat INSIDE the read stream (spread evenly over the sweep, round 2's
// order: pass A alone 0.36-0.40 ms; in natural order, sixteen bursts: 0.41); behind it the kernel runs at the rate of bare loads --
// 0.313-0.317 ms alone = 6.6 TB/s, 0.373 beside the projection build -- and the series are the last thing written before pass B
// reads them.  UPSP_SCAN_SPREAD=1: the spread order (A/B).
static bool scan_active_last()
{
    static const bool v = [] { const char *e = std::getenv("UPSP_SCAN_SPREAD"); return !(e && *e == '1'); }();
    return v;
}
__device__ __forceinline__ unsigned amap_kth_tile(const unsigned *arank, unsigned ntiles, unsigned k, bool active)
{
    // smallest t with (active ? arank[t + 1] : t + 1 - arank[t + 1]) > k
    unsigned lo = 0, hi = ntiles - 1;
    while (lo < hi) {
        const unsigned mid = (lo + hi) >> 1;
        const unsigned a = arank[mid + 1], c = active ? a : mid + 1 - a;
        if (c > k) hi = mid; else lo = mid + 1;
    }
    return lo;
}
__global__ void __launch_bounds__(256)
    amap_order_kernel(const unsigned *__restrict__ arank, unsigned ntiles, unsigned *__restrict__ order, int active_last)
{
    const unsigned p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= ntiles) return;
    const unsigned long long A = arank[ntiles], n = ntiles;
    if (active_last) {          // the tiles nobody reads first, the active ones behind them (scan_active_last)
        const unsigned nI = ntiles - (unsigned)A;
        order[p] = p < nI ? amap_kth_tile(arank, ntiles, p, false) : amap_kth_tile(arank, ntiles, p - nI, true);
        return;
    }
    const unsigned long long j = ((unsigned long long)p * A + n - 1) / n;    // active slots before slot p
    const bool act = j < A && (j * n) / A == p;
    order[p] = amap_kth_tile(arank, ntiles, act ? (unsigned)j : p - (unsigned)j, act);
}

// step 3: node -> index of its pixel's series in the compact buffer (-1: no pixel)
// (skipped, optional: identify_skipped_nodes of a ONE-camera projection in the same sweep -- a node is skipped iff it has no pixel)
__global__ void __launch_bounds__(256)
    amap_nodes_kernel(const int32_t *__restrict__ pix, unsigned nnodes, const uint8_t *__restrict__ flag,
                      const unsigned *__restrict__ tile_off, int32_t *__restrict__ node_k, uint8_t *__restrict__ skipped)
{
    const unsigned n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nnodes) return;
    const int32_t p = pix[n];
    if (skipped) skipped[n] = p >= 0 ? 0 : 1;
    // -2: the node reads a pixel the map does not hold (only with a map built from a candidate set,
    // upsp_pipeline_set_active_hint): pass B then fetches that node's values from the frames themselves
    int32_t k = -1;
    if (p >= 0) k = flag[p] ? (int32_t)(tile_off[(unsigned)p / kFusedPix] + (flag[p] & 0x7Fu)) : -2;
    node_k[n] = k;
}

// Sum over the 16 lanes of a DPP row (all lanes get the total): four VALU adds fed by DPP moves
// (quad swaps, half-row mirror, row mirror) instead of ds_bpermute round trips.  Exact for the
// integer-valued sums used here.
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double x)
{
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_mov_dpp((int)(b & 0xFFFFFFFFll), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), CTRL, 0xF, 0xF, true);
    return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}
__device__ __forceinline__ double group16_sum(double v)
{
    v += dpp_mov_f64<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov_f64<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov_f64<0x141>(v);   // row_half_mirror
    v += dpp_mov_f64<0x140>(v);   // row_mirror
    return v;
}

// Pass A.  compact: [active pixel][cpitch] u16; blockIdx.y = 64-frame group g of the call: it fills
// columns 64 g .. 64 g + 63 (frames past nframes hold 0).  count / pos: one counter / kHotCap positions
// per frame of the call.  A tile nobody reads (most of them: the model covers part of the frame) is only
// streamed through for the hot-pixel count.
template <bool HOT>
__global__ void __launch_bounds__(256)
    scan_compact_kernel(const uint16_t *__restrict__ frames, size_t npix, int nframes_call,
                        const uint8_t *__restrict__ flag, const unsigned *__restrict__ tile_off,
                        uint16_t *__restrict__ compact, unsigned cpitch, unsigned thresh, unsigned max_hot,
                        unsigned *__restrict__ count, unsigned *__restrict__ pos, const unsigned *__restrict__ order,
                        unsigned ntiles_lin)
{
    __shared__ unsigned tile[64][kFusedPitch];   // [frame][pixel pair]
    __shared__ int act_k[kFusedPix];             // compact index of the tile's pixels, -1 = nobody reads it
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int g = blockIdx.y;
    unsigned tl = blockIdx.x;                                // this workgroup's tile
    if (!ntiles_lin) {
        if (order) tl = order[blockIdx.x];
    } else {
        // one-dimensional grid: every group's inactive tiles first (pure reads), then every group's active tiles
        // (order: inactive tiles ascending, then active tiles ascending; order[2 nt] = arank[nt] = number of active tiles)
        const unsigned nt = ntiles_lin;
        const unsigned A = order[2u * nt], nI = nt - A, ng = gridDim.x / nt, b = blockIdx.x;
        if (b < nI * ng) {
            g = (int)(b / nI);
            tl = order[b - (unsigned)g * nI];
        } else {
            // (group after group; a tile's groups side by side instead -- whole 2-KB rows of the series filling up -- measured the same)
            const unsigned b2 = b - nI * ng;
            g = (int)(b2 / A);
            tl = order[nI + (b2 - (unsigned)g * A)];
        }
    }
    const int nframes = min(64, nframes_call - 64 * g);
    frames += (size_t)g * 64 * npix;
    compact += 64 * g;
    if (HOT) {
        count += 64 * g;
        pos += (size_t)64 * g * kHotCap;
    }
    const size_t p0 = (size_t)tl * kFusedPix + 2u * (unsigned)lane;   // this lane's pixel pair
    const bool in = p0 + 1 < npix;                                            // npix is even
    const unsigned k0 = tile_off[tl];
    const bool any_active = tile_off[tl + 1] != k0;                   // (uniform)
    // wave w streams frames w, w+4, ..: one 256-byte segment per wave load, all 16 issued up front
    unsigned v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int f = wave + 4 * i;
        // streamed: the frames are not read again by this schedule
        v[i] = (in && f < nframes) ? __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(frames + (size_t)f * npix + p0)) : 0u;
    }
    if (any_active && threadIdx.x < kFusedPix) {
        const size_t p = (size_t)tl * kFusedPix + threadIdx.x;
        const unsigned fl = p < npix ? flag[p] : 0u;
        act_k[threadIdx.x] = fl ? (int)(k0 + (fl & 0x7Fu)) : -1;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int f = wave + 4 * i;
        if (any_active) tile[f][lane] = v[i];
        if (HOT && (((v[i] & 0xFFFFu) >= thresh) | ((v[i] >> 16) >= thresh))) {
            // pass 1 of fix_hot_pixels; once the counter is past max_hot the verdict is settled
            if (__hip_atomic_load(&count[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= max_hot) {
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (((v[i] >> (16 * k)) & 0xFFFFu) >= thresh) {
                        const unsigned slot = atomicAdd(&count[f], 1u);
                        if (slot < (unsigned)kHotCap) pos[(size_t)f * kHotCap + slot] = (unsigned)(p0 + k);
                    }
            }
        }
    }
    if (!any_active) return;
    __syncthreads();
    // 8 lanes per active pixel: lane j packs frames 8j .. 8j+7 of the pixel's column into 16 bytes
    const int grp = threadIdx.x >> 3, j8 = threadIdx.x & 7;
#pragma unroll
    for (int r = 0; r < kFusedPix / 32; ++r) {
        const int o = grp + 32 * r;
        const int k = act_k[o];
        if (k < 0) continue;
        const unsigned col = (unsigned)o >> 1, sh = 16u * ((unsigned)o & 1u);
        unsigned w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned a = (tile[8 * j8 + 2 * q][col] >> sh) & 0xFFFFu;
            const unsigned b = (tile[8 * j8 + 2 * q + 1][col] >> sh) & 0xFFFFu;
            w[q] = a | (b << 16);
        }
        *reinterpret_cast<uint4 *>(compact + (size_t)k * cpitch + 8 * j8) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// Pass A in two launches (upsp_pipeline_set_scan_split: for a frame loop that shares the device with other kernels): the tiles nobody
// reads as ONE-WAVE workgroups without LDS and with 16-byte loads (a lane takes 8 pixels of one frame, 16 lanes a tile row, the wave
// four frames per load instruction, 16 loads in flight: the bytes in flight of the one-launch form's four-wave workgroup in a quarter of
// the waves and none of its 17 KB of LDS), then the active tiles on a fixed grid (the body of scan_compact_kernel in a loop).  Pass A
// needs eight of the one-launch workgroups per compute unit (alone, with padded LDS: 8 / 4 / 3 / 2 per CU -> 0.316 / 0.348 / 0.418 /
// 0.571 ms) and does not get them beside a projection build, whose persistent traversal workgroups take waves and LDS first at high
// priority: beside the build 0.362 -> 0.349 ms and the bench step 0.799 -> 0.773 ms (the build gains as well); ALONE the one-launch
// form is faster, 0.314 against 0.336 ms -- a launch boundary and the fixed grid's loop --, hence a switch, off by default.
// Needs npix % 128 == 0 (16-byte loads of whole tiles); otherwise the one-launch form runs.
template <bool HOT>
__device__ __forceinline__ void scan_hot_check(unsigned v, int f, size_t p0, unsigned thresh, unsigned max_hot,
                                               unsigned *__restrict__ count, unsigned *__restrict__ pos)
{
    if (HOT && (((v & 0xFFFFu) >= thresh) | ((v >> 16) >= thresh))) {
        if (__hip_atomic_load(&count[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= max_hot) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (((v >> (16 * k)) & 0xFFFFu) >= thresh) {
                    const unsigned slot = atomicAdd(&count[f], 1u);
                    if (slot < (unsigned)kHotCap) pos[(size_t)f * kHotCap + slot] = (unsigned)(p0 + k);
                }
        }
    }
}
typedef unsigned scan_v4u __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(64)
    scan_inactive_kernel(const uint16_t *__restrict__ frames, size_t npix, int nframes_call, int ngroups, unsigned thresh, unsigned max_hot,
                         unsigned *__restrict__ count, unsigned *__restrict__ pos, const unsigned *__restrict__ order, unsigned nt)
{
    const unsigned A = order[2u * nt], nI = nt - A, b = blockIdx.x;
    if (b >= nI * (unsigned)ngroups) return;
    const int g = (int)(b / nI);
    const unsigned tl = order[b - (unsigned)g * nI];
    const int nframes = min(64, nframes_call - 64 * g);
    const int lane = threadIdx.x, r4 = lane >> 4, seg = lane & 15;
    const size_t p0 = (size_t)tl * kFusedPix + 8u * (unsigned)seg;          // this lane's eight pixels
    const uint16_t *fr = frames + (size_t)g * 64 * npix + p0;
    scan_v4u v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int f = 4 * i + r4;
        const scan_v4u z = {0u, 0u, 0u, 0u};
        v[i] = f < nframes ? __builtin_nontemporal_load(reinterpret_cast<const scan_v4u *>(fr + (size_t)f * npix)) : z;
    }
    unsigned *cnt = count + 64 * g;
    unsigned *ps = pos + (size_t)64 * g * kHotCap;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int f = 4 * i + r4;
        const unsigned m = max(max(v[i].x & 0xFFFFu, v[i].x >> 16), max(v[i].y & 0xFFFFu, v[i].y >> 16));
        const unsigned n = max(max(v[i].z & 0xFFFFu, v[i].z >> 16), max(v[i].w & 0xFFFFu, v[i].w >> 16));
        if (max(m, n) >= thresh) {                          // (rare)
            scan_hot_check<true>(v[i].x, f, p0, thresh, max_hot, cnt, ps);
            scan_hot_check<true>(v[i].y, f, p0 + 2, thresh, max_hot, cnt, ps);
            scan_hot_check<true>(v[i].z, f, p0 + 4, thresh, max_hot, cnt, ps);
            scan_hot_check<true>(v[i].w, f, p0 + 6, thresh, max_hot, cnt, ps);
        }
    }
}
template <bool HOT>
__global__ void __launch_bounds__(256)
    scan_active_kernel(const uint16_t *__restrict__ frames, size_t npix, int nframes_call, int ngroups,
                       const uint8_t *__restrict__ flag, const unsigned *__restrict__ tile_off,
                       uint16_t *__restrict__ compact, unsigned cpitch, unsigned thresh, unsigned max_hot,
                       unsigned *__restrict__ count, unsigned *__restrict__ pos, const unsigned *__restrict__ order, unsigned nt)
{
    __shared__ unsigned tile[64][kFusedPitch];   // [frame][pixel pair]
    __shared__ int act_k[kFusedPix];             // compact index of the tile's pixels, -1 = nobody reads it
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned A = order[2u * nt], nI = nt - A, total = A * (unsigned)ngroups;
    // (Round 6, measured and not kept: the loads of the workgroup's NEXT (tile, group) pair in flight while the current one goes
    //  through LDS -- 26 more registers, pass A 0.349 ms in the step either way: with eight workgroups per compute unit the trips
    //  of the others already cover a trip's wait.)
    for (unsigned b2 = blockIdx.x; b2 < total; b2 += gridDim.x) {          // (uniform)
        const int g = (int)(b2 / A);
        const unsigned tl = order[nI + (b2 - (unsigned)g * A)];
        const int nframes = min(64, nframes_call - 64 * g);
        const uint16_t *fr = frames + (size_t)g * 64 * npix;
        uint16_t *cmp = compact + 64 * g;
        unsigned *cnt = HOT ? count + 64 * g : count;
        unsigned *ps = HOT ? pos + (size_t)64 * g * kHotCap : pos;
        const size_t p0 = (size_t)tl * kFusedPix + 2u * (unsigned)lane;
        const bool in = p0 + 1 < npix;
        const unsigned k0 = tile_off[tl];
        unsigned v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int f = wave + 4 * i;
            v[i] = (in && f < nframes) ? __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(fr + (size_t)f * npix + p0)) : 0u;
        }
        if (threadIdx.x < kFusedPix) {
            const size_t p = (size_t)tl * kFusedPix + threadIdx.x;
            const unsigned fl = p < npix ? flag[p] : 0u;
            act_k[threadIdx.x] = fl ? (int)(k0 + (fl & 0x7Fu)) : -1;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int f = wave + 4 * i;
            tile[f][lane] = v[i];
            scan_hot_check<HOT>(v[i], f, p0, thresh, max_hot, cnt, ps);
        }
        __syncthreads();
        const int grp = threadIdx.x >> 3, j8 = threadIdx.x & 7;
#pragma unroll
        for (int r = 0; r < kFusedPix / 32; ++r) {
            const int o = grp + 32 * r;
            const int k = act_k[o];
            if (k < 0) continue;
            const unsigned col = (unsigned)o >> 1, sh = 16u * ((unsigned)o & 1u);
            unsigned w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned a = (tile[8 * j8 + 2 * q][col] >> sh) & 0xFFFFu;
                const unsigned bb = (tile[8 * j8 + 2 * q + 1][col] >> sh) & 0xFFFFu;
                w[q] = a | (bb << 16);
            }
            *reinterpret_cast<uint4 *>(cmp + (size_t)k * cpitch + 8 * j8) = make_uint4(w[0], w[1], w[2], w[3]);
        }
        __syncthreads();                          // (the tile is read: the next trip may overwrite it)
    }
}

// Pass B, whole rows.  The series of a node over all (<= kGroupFramesMax) frames of a call sits in
// the compact buffer, so its row piece is written once, contiguously: LPR lanes per row (lane l:
// frames 4 l .. 4 l + 3), 256 / LPR rows per sweep, ROWS sweeps per workgroup.  Rows of nodes without a
// pixel are a constant fill with no load at all.  Measured on MI355X (tools/probe/store_shapes.hip,
// 500 958 rows x 4000 B): a workgroup sweeping one whole 4-KB row per store instruction 5.7-5.9 TB/s
// at a 4096-B pitch against 4.3 TB/s for the 1-KB pieces of node_stream_kernel.
// Accumulators: per-lane integer sums (values < 2^16, exact) and double sums of the float squares
// (integers < 2^32, exact in any order), reduced per wave by DPP, per row through LDS.
constexpr int kGroupFramesMax = 1024;
__device__ __forceinline__ unsigned group16_sum_u32(unsigned v)
{
    v += (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);
    v += (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);
    v += (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true);
    v += (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xF, 0xF, true);
    return v;
}
template <int L>
__device__ __forceinline__ double readlane_f64(double x)
{
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xFFFFFFFFll), L);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), L);
    return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}
template <int LPR, int ROWS, bool U16, int AHEAD = 1>
__global__ void __launch_bounds__(256)
    node_rows_kernel(const uint16_t *__restrict__ compact, unsigned cpitch, const int32_t *__restrict__ node_k,
                     const uint8_t *__restrict__ skipped, const int32_t *__restrict__ rowmap, unsigned nnodes,
                     int nframes, float *__restrict__ rows_t, uint16_t *__restrict__ rows_t16, long long ld_t,
                     double *__restrict__ sum, double *__restrict__ sumsq, const uint16_t *__restrict__ frames,
                     size_t npix, const int32_t *__restrict__ pix, bool fresh, int nstore,
                     const uint16_t *__restrict__ compact_b, unsigned cpitch_b, int nframes_a)
{
    // compact_b / cpitch_b / nframes_a: the series of frames [nframes_a, nframes) of the launch come from a second buffer (same
    // rows, its column 0 = frame nframes_a): a launch of the owner's pass B over blocks received from two peers, cut at 128-byte
    // lines of the output rows instead of at the block boundary (launch_node_rows_blocks).  nframes_a = nframes: one buffer.
    // nstore (>= nframes): columns [nframes, nstore) of every stored row are padding the caller gave up (upsp_pipeline_set_row_padding):
    // they are written too (0 / the row's fill value), so that a row ends on a 128-byte line -- a 4000-byte row piece whose last line is
    // a quarter full costs the memory system a partial write per row (tools/probe/store_shapes.hip: 5.2-5.4 TB/s against 6.1-6.2 for
    // whole lines, same store shape)

    // fresh: the accumulators were reset and not touched since -- they are WRITTEN (0 + the sums of this launch; 0 for a node without
    // a pixel) instead of read, added to and written, and upsp_pipeline_reset launches no fill
    constexpr int RPS = 256 / LPR;          // rows per sweep
    constexpr int WPR = LPR / 64;           // waves per row
    constexpr int NR = RPS * ROWS;          // rows (consecutive nodes) per workgroup
    __shared__ int s_k[NR], s_row[NR], s_sk[NR];
    __shared__ unsigned p_s[NR][WPR];
    __shared__ double p_ss[NR][WPR];
    const unsigned n0 = blockIdx.x * (unsigned)NR;
    const int t = threadIdx.x;
    if (t < NR) {
        const unsigned n = n0 + (unsigned)t;
        const bool ok = n < nnodes;
        s_k[t] = ok ? node_k[n] : -1;
        s_sk[t] = (ok && skipped) ? (int)skipped[n] : 0;
        s_row[t] = ok ? (rowmap ? rowmap[n] : (int)n) : -1;     // rows are < 2^31 (node count check at create)
    }
    __syncthreads();
    const int sub = t / LPR, l = t % LPR, wr = l >> 6, lane = t & 63;
    const int f0 = 4 * l;
    const float qnan = __builtin_nanf("");
    const bool vec_ok = U16 ? (((ld_t & 3) == 0) && ((reinterpret_cast<size_t>(rows_t16) & 7) == 0))
                            : (((ld_t & 3) == 0) && ((reinterpret_cast<size_t>(rows_t) & 15) == 0));
    // the series load of sweep j + 1 is issued before the rows of sweep j are converted and stored (measurement switch
    // kRowsAhead: 0 = load at the point of use, round 2's form)
    auto series_load = [&](int j) -> uint2 {
        const int r = j * RPS + sub;
        const int k = s_k[r];
        uint2 w = make_uint2(0u, 0u);
        // (ordinary loads: the compact buffer was written a moment ago and sits in L2 / Infinity Cache)
        if (s_sk[r] == 0 && k >= 0 && f0 < nframes)
            w = f0 < nframes_a ? *reinterpret_cast<const uint2 *>(compact + (size_t)k * cpitch + f0)
                               : *reinterpret_cast<const uint2 *>(compact_b + (size_t)k * cpitch_b + (f0 - nframes_a));
        return w;
    };
    // AHEAD 2: the series of ALL sweeps are requested before the first row is stored -- one memory latency per workgroup instead
    // of one per sweep.  Nothing for series that sit in the Infinity Cache (pass A wrote them a moment ago: 0.40-0.41 ms either
    // way), the difference when they come from HBM (the owner's pass B over blocks that arrived a step ago: tools/passb_probe.py)
    uint2 w_all[AHEAD == 2 ? ROWS : 1];
    if (AHEAD == 2) {
#pragma unroll
        for (int j = 0; j < ROWS; ++j) w_all[j] = series_load(j);
    }
    uint2 w_next = AHEAD == 1 ? series_load(0) : make_uint2(0u, 0u);
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        const int r = j * RPS + sub;                            // (uniform per wave)
        const int k = s_k[r], row = s_row[r];
        const bool sk = s_sk[r] != 0;
        uint2 w = AHEAD == 2 ? w_all[AHEAD == 2 ? j : 0] : (AHEAD == 1 ? w_next : series_load(j));
        if (AHEAD == 1 && j + 1 < ROWS) w_next = series_load(j + 1);
        if (sk || k == -1) {
            // (uniform per wave) a row without data -- no camera sees the node: NaN; no pixel: 0 -- is a constant fill:
            // no load, no sums, no reductions (its partials are not read below).  These are 60 % of the rows of the bench
            // model, and the kernel was bound by its VALU instructions (PMC: 194 M wave-instructions per launch = 0.32 of
            // its 0.40 ms at one per 4 cycles), most of them the per-row reductions.
            if (row >= 0 && f0 < nstore) {
                if (!U16) {
                    float *dst = rows_t + (long long)row * ld_t + f0;
                    typedef float v4f __attribute__((ext_vector_type(4)));
                    const float c = sk ? qnan : 0.0f;
                    const v4f nv = {c, c, c, c};
                    if (vec_ok && f0 + 3 < nstore) {
                        __builtin_nontemporal_store(nv, reinterpret_cast<v4f *>(dst));
                    } else {
                        dst[0] = c;
                        if (f0 + 1 < nstore) dst[1] = c;
                        if (f0 + 2 < nstore) dst[2] = c;
                        if (f0 + 3 < nstore) dst[3] = c;
                    }
                } else {
                    uint16_t *dst = rows_t16 + (long long)row * ld_t + f0;
                    typedef unsigned v2u __attribute__((ext_vector_type(2)));
                    const v2u nv = {0u, 0u};
                    if (vec_ok && f0 + 3 < nstore) {
                        __builtin_nontemporal_store(nv, reinterpret_cast<v2u *>(dst));
                    } else {
                        dst[0] = 0;
                        if (f0 + 1 < nstore) dst[1] = 0;
                        if (f0 + 2 < nstore) dst[2] = 0;
                        if (f0 + 3 < nstore) dst[3] = 0;
                    }
                }
            }
            continue;
        }
        if (k == -2 && f0 < nframes) {                          // (rare) pixel outside the candidate map: from the frames
            const size_t pp = (size_t)pix[n0 + (unsigned)r];
            unsigned v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = f0 + q < nframes ? (unsigned)frames[(size_t)(f0 + q) * npix + pp] : 0u;
            w = make_uint2(v[0] | (v[1] << 16), v[2] | (v[3] << 16));
        }
        const unsigned d[4] = {w.x & 0xFFFFu, w.x >> 16, w.y & 0xFFFFu, w.y >> 16};
        unsigned s = 0u;
        double ss = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (f0 + q < nframes) {
                const float x = (float)d[q];
                s += d[q];
                ss += (double)(x * x);
            }
        s = group16_sum_u32(s);
        ss = group16_sum(ss);
        // the wave's total from its four DPP rows
        const unsigned ws = (unsigned)__builtin_amdgcn_readlane((int)s, 0) + (unsigned)__builtin_amdgcn_readlane((int)s, 16) +
                            (unsigned)__builtin_amdgcn_readlane((int)s, 32) + (unsigned)__builtin_amdgcn_readlane((int)s, 48);
        const double wss = (readlane_f64<0>(ss) + readlane_f64<16>(ss)) + (readlane_f64<32>(ss) + readlane_f64<48>(ss));
        if (lane == 0) {
            p_s[r][wr] = ws;
            p_ss[r][wr] = wss;
        }
        if (row < 0 || f0 >= nstore) continue;                  // row not stored (packed series) / past the end
        // (columns past nframes: padding, written as 0)
        const unsigned e[4] = {d[0], f0 + 1 < nframes ? d[1] : 0u, f0 + 2 < nframes ? d[2] : 0u, f0 + 3 < nframes ? d[3] : 0u};
        if (!U16) {
            float *dst = rows_t + (long long)row * ld_t + f0;
            typedef float v4f __attribute__((ext_vector_type(4)));
            // a node without a pixel: 0 (empty row of the projection matrix); no camera sees it: NaN
            const v4f nv = {sk ? qnan : (float)e[0], sk ? qnan : (float)e[1], sk ? qnan : (float)e[2], sk ? qnan : (float)e[3]};
            if (vec_ok && f0 + 3 < nstore) {
                __builtin_nontemporal_store(nv, reinterpret_cast<v4f *>(dst));
            } else {
                dst[0] = nv.x;
                if (f0 + 1 < nstore) dst[1] = nv.y;
                if (f0 + 2 < nstore) dst[2] = nv.z;
                if (f0 + 3 < nstore) dst[3] = nv.w;
            }
        } else {   // u16 series (exchange wire format; callers store visible rows only)
            uint16_t *dst = rows_t16 + (long long)row * ld_t + f0;
            typedef unsigned v2u __attribute__((ext_vector_type(2)));
            const v2u nv = {sk ? 0u : (e[0] | (e[1] << 16)), sk ? 0u : (e[2] | (e[3] << 16))};
            if (vec_ok && f0 + 3 < nstore) {
                __builtin_nontemporal_store(nv, reinterpret_cast<v2u *>(dst));
            } else {
                dst[0] = (uint16_t)(nv.x & 0xFFFFu);
                if (f0 + 1 < nstore) dst[1] = (uint16_t)(nv.x >> 16);
                if (f0 + 2 < nstore) dst[2] = (uint16_t)(nv.y & 0xFFFFu);
                if (f0 + 3 < nstore) dst[3] = (uint16_t)(nv.y >> 16);
            }
        }
    }
    __syncthreads();
    // accumulators: a node with a pixel gains its sums (NaN when no camera sees it); without a pixel it
    // gains 0, or turns NaN when no camera sees it
    if (t < NR) {
        const unsigned n = n0 + (unsigned)t;
        if (n < nnodes) {
            const bool msk = s_sk[t] != 0;
            if (s_k[t] != -1 || msk) {
                unsigned as = 0u;
                double ass = 0.0;
#pragma unroll
                for (int q = 0; q < WPR; ++q) {
                    as += p_s[t][q];
                    ass += p_ss[t][q];
                }
                sum[n] = msk ? (double)qnan : (fresh ? 0.0 : sum[n]) + (double)as;
                sumsq[n] = msk ? (double)qnan : (fresh ? 0.0 : sumsq[n]) + ass;
            } else if (fresh) {
                sum[n] = 0.0;
                sumsq[n] = 0.0;
            }
        }
    }
}

// The owner's pass B of an exchange in ONE launch: the frames of a run lie in several series buffers (one block per source rank),
// the output rows are cut into windows of <= 1024 columns at 128-byte lines, each window reading the end of one block and the start
// of the next (rows_from_pixel_blocks).  A workgroup takes ROWS consecutive nodes through ALL windows of the launch: a row piece of
// 4 KB per store instruction like in node_rows_kernel, the sums of a lane kept in registers across the windows (values < 2^16, <= 4 per
// lane and window, <= 16 windows: < 2^32; the squares integer-valued doubles -- exact in any order) and reduced once per row instead
// of once per row and window.  A launch per window costs its ramp and tail 8 times per step at 8 ranks: 0.475 ms against
// 0.41 for the same bytes in one launch (tools/passb_probe.py).  f32 rows, identity row map, every window a multiple of 4 columns.
template <int ROWS>
__global__ void __launch_bounds__(256)
    node_rows_windows_kernel(RowWindows win, const int32_t *__restrict__ node_k, const uint8_t *__restrict__ skipped, unsigned nnodes,
                             float *__restrict__ rows_t, long long ld_t, double *__restrict__ sum, double *__restrict__ sumsq)
{
    __shared__ int s_k[ROWS], s_sk[ROWS];
    __shared__ unsigned p_s[ROWS][4];
    __shared__ double p_ss[ROWS][4];
    const unsigned n0 = blockIdx.x * (unsigned)ROWS;
    const int t = threadIdx.x;
    if (t < ROWS) {
        const unsigned n = n0 + (unsigned)t;
        const bool ok = n < nnodes;
        s_k[t] = ok ? node_k[n] : -1;
        s_sk[t] = (ok && skipped) ? (int)skipped[n] : 0;
    }
    __syncthreads();
    const int wr = t >> 6, lane = t & 63, f0 = 4 * t;
    const float qnan = __builtin_nanf("");
    typedef float v4f __attribute__((ext_vector_type(4)));
    unsigned acc_s[ROWS];
    double acc_ss[ROWS];
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        acc_s[j] = 0u;
        acc_ss[j] = 0.0;
    }
    // the series of every row of the workgroup for a window at once, and those of window wi + 1 requested before the rows of
    // window wi are converted and stored (they come from HBM: the latency of a window's loads lies behind the stores of the one before)
    auto series_load = [&](int wi, uint2 *v) {
        const RowWindow W = win.w[wi];                          // (uniform: kernel arguments)
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            const int k = s_k[j];
            v[j] = make_uint2(0u, 0u);
            if (s_sk[j] == 0 && k >= 0 && f0 < W.nframes)
                v[j] = f0 < W.nframes_a ? *reinterpret_cast<const uint2 *>(W.a + (size_t)k * W.pitch_a + f0)
                                        : *reinterpret_cast<const uint2 *>(W.b + (size_t)k * W.pitch_b + (f0 - W.nframes_a));
        }
    };
    uint2 v[ROWS], v_next[ROWS];
    series_load(0, v_next);
    for (int wi = 0; wi < win.n; ++wi) {
        const RowWindow W = win.w[wi];
#pragma unroll
        for (int j = 0; j < ROWS; ++j) v[j] = v_next[j];
        if (wi + 1 < win.n) series_load(wi + 1, v_next);
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            if (n0 + (unsigned)j >= nnodes || f0 >= W.nstore) continue;
            const int k = s_k[j];
            const bool sk = s_sk[j] != 0;
            v4f nv;
            if (sk || k < 0) {                                  // (uniform) no camera sees the node: NaN; no pixel: 0
                const float c = sk ? qnan : 0.0f;
                nv = (v4f){c, c, c, c};
            } else {
                const unsigned d[4] = {v[j].x & 0xFFFFu, v[j].x >> 16, v[j].y & 0xFFFFu, v[j].y >> 16};
                if (f0 < W.nframes) {                           // (nframes is a multiple of 4: the four frames of a lane are valid together)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float x = (float)d[q];
                        acc_s[j] += d[q];
                        acc_ss[j] += (double)(x * x);
                    }
                    nv = (v4f){(float)d[0], (float)d[1], (float)d[2], (float)d[3]};
                } else {
                    nv = (v4f){0.0f, 0.0f, 0.0f, 0.0f};          // padding
                }
            }
            __builtin_nontemporal_store(nv, reinterpret_cast<v4f *>(rows_t + (long long)(n0 + (unsigned)j) * ld_t + W.col + f0));
        }
    }
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        if (s_sk[j] != 0 || s_k[j] < 0) continue;               // (uniform) constant rows: no sums
        const unsigned s = group16_sum_u32(acc_s[j]);
        const double ss = group16_sum(acc_ss[j]);
        const unsigned ws = (unsigned)__builtin_amdgcn_readlane((int)s, 0) + (unsigned)__builtin_amdgcn_readlane((int)s, 16) +
                            (unsigned)__builtin_amdgcn_readlane((int)s, 32) + (unsigned)__builtin_amdgcn_readlane((int)s, 48);
        const double wss = (readlane_f64<0>(ss) + readlane_f64<16>(ss)) + (readlane_f64<32>(ss) + readlane_f64<48>(ss));
        if (lane == 0) {
            p_s[j][wr] = ws;
            p_ss[j][wr] = wss;
        }
    }
    __syncthreads();
    if (t < ROWS) {
        const unsigned n = n0 + (unsigned)t;
        if (n < nnodes) {
            const bool msk = s_sk[t] != 0;
            if (msk) {
                sum[n] = (double)qnan;
                sumsq[n] = (double)qnan;
            } else if (s_k[t] >= 0) {
                unsigned as = 0u;
                double ass = 0.0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    as += p_s[t][q];
                    ass += p_ss[t][q];
                }
                sum[n] = sum[n] + (double)as;
                sumsq[n] = sumsq[n] + ass;
            }
        }
    }
}

// Pass B for several cameras (and weights): sol = sum over the cameras, in camera order, of
// w_c * f32(pixel_c) exactly as gather_tile_kernel forms it (psp_process.cpp:1813-1819), from one
// compact buffer per camera; f32 rows only (node_rows_multi_kernel below).  The accumulators add the frames of a lane first
// and the lanes after (DPP), a different order from the gather's: same values to ~1e-16 relative (the parity bar for the
// accumulators is 1e-12).
struct StreamMultiArgs {
    int ncams;
    const uint16_t *compact[kMaxCams];
    const int32_t *node_k[kMaxCams];
    const float *weight[kMaxCams];
};
// Pass B, whole rows, for several cameras (and weights): sol = sum over the cameras, in camera order, of
// w_c * f32(pixel_c) exactly as gather_tile_kernel forms it (psp_process.cpp:1813-1819), from one compact
// buffer per camera; row layout and lane mapping of node_rows_kernel.  The accumulators add the frames of
// a lane first and the lanes after: a different order from the gather's, same values to ~1e-16 relative
// (the parity bar for the accumulators is 1e-12).
// Round 3: (1) the compact series of EVERY camera and EVERY sweep of the workgroup are requested before any of them
// is used (round 2 walked sweep by sweep, camera by camera: ROWS x cameras dependent load -> convert -> add chains per
// lane); (2) a row no camera sees (NaN in every frame) or that reads no pixel (0) is a constant fill -- no loads, no
// sums, no reductions, like in node_rows_kernel; (3) NC = the camera count at compile time (2 .. 4; 0 = any).
// PIPE: persistent workgroups -- a workgroup takes the node groups blockIdx.x, blockIdx.x + gridDim.x, ... and fetches the staging
// data (series indices, weights, row, accumulators) of its NEXT group into registers before it starts on the rows of the current
// one (two staging areas in LDS): of the two dependent memory latencies per group, staging and series loads, the first is hidden.
// Measured and NOT the default (UPSP_MULTI_PIPE=n workgroups per CU): the loop-carried staging registers take the kernel from 96 to
// 130 VGPRs (3 instead of 5 waves per SIMD): 3.14-3.23 ms against 2.68-2.74; held to 96 registers it spills 33 of them: 4.0 ms.
template <int LPR, int ROWS, int NC, int FPL = 4, int PIPE = 0, int AHEAD = ROWS>
__global__ void __launch_bounds__(256)
    node_rows_multi_kernel(StreamMultiArgs a, unsigned cpitch, const uint8_t *__restrict__ skipped,
                           const int32_t *__restrict__ rowmap, unsigned nnodes, int nframes,
                           float *__restrict__ rows_t, long long ld_t, double *__restrict__ sum,
                           double *__restrict__ sumsq)
{
    // (Rows that end on whole 128-byte lines -- the 24 padding columns behind 1000 frames written too, as node_rows_kernel does --
    //  were measured AGAIN in round 6, with the kernel held at 72 registers / seven waves per SIMD (launch bounds; the padded form
    //  needs two more): 3.10-3.11 ms against 2.61-2.74 for 4 cameras x 2.5 M nodes x 1000 frame sets, two alternations; with every
    //  series load of the four sweeps in front of the first store as well: 3.19-3.22; that order without the padding: 2.78.  The
    //  several-camera pass keeps its 4000-byte rows and its loads one sweep ahead.)
    const int nstore = nframes;
    constexpr int RPS = 256 / LPR, WPR = LPR / 64, NR = RPS * ROWS;
    constexpr int MC = NC ? NC : kMaxCams;
    constexpr int NB = PIPE ? 2 : 1;
    __shared__ int sb_k[NB][NR][MC];
    __shared__ float sb_w[NB][NR][MC];
    __shared__ int sb_row[NB][NR], sb_kind[NB][NR];       // kind: 0 data, 1 no camera sees the node (NaN), 2 reads no pixel (0)
    // per row and wave four partial sums (one per 16-lane DPP row): the last two reduction steps are done by the thread that
    // owns the node's accumulators, from LDS, instead of 16 v_readlane per wave and row
    __shared__ double p_s[NR][WPR * 4], p_ss[NR][WPR * 4];
    const int ncams = NC ? NC : a.ncams;
    const int t = threadIdx.x;
    static_assert(NR * MC <= 256, "one staging element per thread");
    const unsigned ngroups = (nnodes + (unsigned)NR - 1u) / (unsigned)NR;
    // staging of one node group: global loads into registers (stage_load), registers into LDS area b (stage_store)
    struct Stage { int k; float w; int kind, row; double s, ss; };
    auto stage_load = [&](unsigned g) {
        Stage st;
        st.k = -1; st.w = 1.0f; st.kind = 2; st.row = -1; st.s = 0.0; st.ss = 0.0;
        const unsigned nb0 = g * (unsigned)NR;
        if (t < NR * ncams) {
            const int r = t / ncams, c = t % ncams;
            const unsigned n = nb0 + (unsigned)r;
            if (n < nnodes) {
                st.k = a.node_k[c][n];
                st.w = a.weight[c] ? a.weight[c][n] : 1.0f;
            }
        }
        if (t < NR) {
            // the node's accumulators are read HERE, with the staging loads, not after the rows: their read-modify-write was a
            // third dependent memory latency at the end of every group
            const unsigned n = nb0 + (unsigned)t;
            const bool ok = n < nnodes;
            bool any = false;
            for (int c = 0; c < ncams; ++c) any = any || (ok && a.node_k[c][n] >= 0);
            st.kind = (ok && skipped && skipped[n]) ? 1 : (any ? 0 : 2);
            st.row = ok ? (rowmap ? rowmap[n] : (int)n) : -1;
            if (ok) {
                st.s = sum[n];
                st.ss = sumsq[n];
            }
        }
        return st;
    };
    auto stage_store = [&](const Stage &st, int b) {
        if (t < NR * ncams) {
            sb_k[b][t / ncams][t % ncams] = st.k;
            sb_w[b][t / ncams][t % ncams] = st.w;
        }
        if (t < NR) {
            sb_kind[b][t] = st.kind;
            sb_row[b][t] = st.row;
        }
    };
    unsigned g = blockIdx.x;
    if (g >= ngroups) return;
    int buf = 0;
    Stage cur = stage_load(g);
    stage_store(cur, 0);
    __syncthreads();
    for (;;) {
    const unsigned gn = PIPE ? g + gridDim.x : ngroups;
    Stage nxt = cur;
    if (PIPE && gn < ngroups) nxt = stage_load(gn);             // in flight while the rows of group g are written
    const unsigned n0 = g * (unsigned)NR;
    const double acc_s = cur.s, acc_ss = cur.ss;
    int (*s_k)[MC] = sb_k[buf];
    float (*s_w)[MC] = sb_w[buf];
    int *s_row = sb_row[buf], *s_kind = sb_kind[buf];
    const int sub = t / LPR, l = t % LPR, wr = l >> 6, lane = t & 63;
    // FPL frames per lane: 4, or 8 = two groups of 4 that lie 4 * LPR frames apart (every load / store instruction of a wave
    // still covers one contiguous piece of the row; 8 consecutive frames per lane made the stores half-empty: 4.1 ms against
    // 2.99): half as many lanes per row, so the per-row reductions of the double sums are paid half as often
    static_assert(FPL == 4 || FPL == 8, "frames per lane");
    constexpr int GS = 4 * LPR;                                  // distance of a lane's two groups
    const int f0 = 4 * l;
    const float qnan = __builtin_nanf("");
    const bool vec_ok = ((ld_t & 3) == 0) && ((reinterpret_cast<size_t>(rows_t) & 15) == 0);
    typedef float v4f __attribute__((ext_vector_type(4)));
    // the series loads run AHEAD sweeps in front of the rows that use them (AHEAD = ROWS: every load of the workgroup first)
    uint2 tt[ROWS][MC][FPL / 4];
    auto series_load = [&](int j) {
        const int r = j * RPS + sub;                            // (uniform per wave)
        const bool data = s_kind[r] == 0 && f0 < nframes;
#pragma unroll
        for (int c = 0; c < MC; ++c) {
#pragma unroll
            for (int h = 0; h < FPL / 4; ++h) tt[j][c][h] = make_uint2(0u, 0u);
            if (c < ncams) {
                const int k = s_k[r][c];
                if (data && k >= 0) {
                    const uint16_t *src = a.compact[c] + (size_t)k * cpitch + f0;      // (rows padded to 64 frames: in bounds)
                    tt[j][c][0] = *reinterpret_cast<const uint2 *>(src);
                    if (FPL == 8 && f0 + GS < nframes) tt[j][c][FPL / 4 - 1] = *reinterpret_cast<const uint2 *>(src + GS);
                }
            }
        }
    };
#pragma unroll
    for (int j = 0; j < AHEAD && j < ROWS; ++j) series_load(j);
    // ... then the rows
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        if (j + AHEAD < ROWS) series_load(j + AHEAD);
        const int r = j * RPS + sub;
        const int row = s_row[r], kind = s_kind[r];
        const bool stored = row >= 0 && f0 < nstore && n0 + (unsigned)r < nnodes;
        if (kind != 0) {                                        // (uniform per wave) constant fill
            if (stored) {
                float *dst = rows_t + (long long)row * ld_t + f0;
                const float cv = kind == 1 ? qnan : 0.0f;
                const v4f nv = {cv, cv, cv, cv};
#pragma unroll
                for (int h = 0; h < FPL / 4; ++h) {
                    const int fh = f0 + GS * h;
                    if (vec_ok && fh + 3 < nstore) {
                        __builtin_nontemporal_store(nv, reinterpret_cast<v4f *>(dst + GS * h));
                    } else {
                        if (fh < nstore) dst[GS * h] = cv;
                        if (fh + 1 < nstore) dst[GS * h + 1] = cv;
                        if (fh + 2 < nstore) dst[GS * h + 2] = cv;
                        if (fh + 3 < nstore) dst[GS * h + 3] = cv;
                    }
                }
            }
            continue;
        }
        float acc[FPL];
#pragma unroll
        for (int q = 0; q < FPL; ++q) acc[q] = 0.0f;
#pragma unroll
        for (int c = 0; c < MC; ++c) {
            // (uniform per wave) a camera that does not see the node contributes 0.0f + w * 0 = +0 in the gather; every
            // term here is >= +0, so leaving it out changes no bit -- and a node is seen by 1.x cameras on average
            if (c < ncams && s_k[r][c] >= 0) {
                const float w = s_w[r][c];
#pragma unroll
                for (int h = 0; h < FPL / 4; ++h) {
                    const uint2 q = tt[j][c][h];
                    // (the gather forms 0.0f + w * pixel: with acc starting at +0 the extra +0 changes no bit of the sum)
                    const float v[4] = {w * (float)(q.x & 0xFFFFu), w * (float)(q.x >> 16),
                                        w * (float)(q.y & 0xFFFFu), w * (float)(q.y >> 16)};
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) acc[4 * h + qq] = acc[4 * h + qq] + v[qq];
                }
            }
        }
        double s = 0.0, ss = 0.0;
#pragma unroll
        for (int q = 0; q < FPL; ++q)
            if (f0 + (q >> 2) * GS + (q & 3) < nframes) {
                s += (double)acc[q];
                ss += (double)(acc[q] * acc[q]);
            }
        s = group16_sum(s);
        ss = group16_sum(ss);
        if ((lane & 15) == 0) {
            p_s[r][wr * 4 + (lane >> 4)] = s;
            p_ss[r][wr * 4 + (lane >> 4)] = ss;
        }
        if (!stored) continue;
        float *dst = rows_t + (long long)row * ld_t + f0;
#pragma unroll
        for (int h = 0; h < FPL / 4; ++h) {
            const int fh = f0 + GS * h;
            // (columns past the last frame: padding -> 0, whatever the series buffers hold there)
            const v4f nv = {fh < nframes ? acc[4 * h] : 0.0f, fh + 1 < nframes ? acc[4 * h + 1] : 0.0f,
                            fh + 2 < nframes ? acc[4 * h + 2] : 0.0f, fh + 3 < nframes ? acc[4 * h + 3] : 0.0f};
            if (vec_ok && fh + 3 < nstore) {
                __builtin_nontemporal_store(nv, reinterpret_cast<v4f *>(dst + GS * h));
            } else {
                if (fh < nstore) dst[GS * h] = nv.x;
                if (fh + 1 < nstore) dst[GS * h + 1] = nv.y;
                if (fh + 2 < nstore) dst[GS * h + 2] = nv.z;
                if (fh + 3 < nstore) dst[GS * h + 3] = nv.w;
            }
        }
    }
    __syncthreads();
    if (t < NR) {
        const unsigned n = n0 + (unsigned)t;
        if (n < nnodes) {
            const int kind = s_kind[t];
            if (kind == 1) {
                sum[n] = (double)qnan;
                sumsq[n] = (double)qnan;
            } else if (kind == 0) {
                double as = 0.0, ass = 0.0;
#pragma unroll
                for (int q = 0; q < WPR; ++q) {        // (the order the readlane form used: rows of a wave, then the waves)
                    as += (p_s[t][4 * q] + p_s[t][4 * q + 1]) + (p_s[t][4 * q + 2] + p_s[t][4 * q + 3]);
                    ass += (p_ss[t][4 * q] + p_ss[t][4 * q + 1]) + (p_ss[t][4 * q + 2] + p_ss[t][4 * q + 3]);
                }
                sum[n] = acc_s + as;
                sumsq[n] = acc_ss + ass;
            }
        }
    }
    if (!PIPE || gn >= ngroups) break;
    stage_store(nxt, buf ^ 1);        // (area buf ^ 1 was last read two barriers ago)
    __syncthreads();                  // also: p_s / p_ss are free again
    g = gn;
    buf ^= 1;
    cur = nxt;
    }
}

// Frames with 1..max_hot hot pixels: repair in place (same code as the scan kernel's pass 2) and
// list the replaced pixels as (frame, position, old, new) for the patch kernels.  One lane per frame;
// frame f owns slots [f * max_hot, (f + 1) * max_hot) of the list (a frame replaces at most max_hot
// pixels), nch[f] = how many it filled -- no cap, no order that depends on the atomics.
// *ntotal (zero on entry) counts the changes of the call: the list kernels below return at once when
// it stays zero (nearly every call).
__device__ __forceinline__ void
    hot_repair_frame(size_t f, uint16_t *frames, size_t npix, int rows, int cols, int min_change,
                     int max_hot, unsigned *__restrict__ count, const unsigned *__restrict__ pos,
                     unsigned *__restrict__ ntotal, unsigned *__restrict__ nch, uint4 *__restrict__ changes,
                     uint4 *__restrict__ clist = nullptr)
{
    const unsigned n = count[f];
    nch[f] = 0u;
    if (n == 0u) return;
    count[f] = 0u;                             // clean for the next launch
    if (n > (unsigned)max_hot) return;         // "too many pixels look hot": frame untouched
    unsigned p[kHotCap];
    uint16_t before[kHotCap];
    uint16_t *img = frames + f * npix;
    for (unsigned i = 0; i < n; ++i) {
        p[i] = pos[f * kHotCap + i];
        before[i] = img[p[i]];
    }
    fix_frame(img, rows, cols, min_change, max_hot, n, p, nullptr);   // sorts p
    unsigned m = 0;
    for (unsigned i = 0; i < n; ++i) {
        unsigned oldv = 0;
        for (unsigned j = 0; j < n; ++j)
            if (pos[f * kHotCap + j] == p[i]) oldv = before[j];
        const unsigned newv = img[p[i]];
        if (newv != oldv) changes[f * (size_t)max_hot + m++] = make_uint4((unsigned)f, p[i], oldv, newv);
    }
    nch[f] = m;
    if (m) {
        // clist (optional): the same records once more, DENSE, in whatever order the frames get here (a (pixel, frame) pair
        // changes at most once, so the order does not matter)
        const unsigned base = atomicAdd(ntotal, m);
        if (clist)
            for (unsigned j = 0; j < m; ++j) clist[base + j] = changes[f * (size_t)max_hot + j];
    }
}
// ntotal_next (optional): the counter the NEXT fix-up of this buffer will use -- zeroed here, where nothing reads it (the two
// counters of a change buffer are used in turn: no reset launch in front of every fix-up)
__global__ void __launch_bounds__(64)
    hot_repair_kernel(uint16_t *frames, size_t npix, int nframes, int rows, int cols, int min_change,
                      int max_hot, unsigned *__restrict__ count, const unsigned *__restrict__ pos,
                      unsigned *__restrict__ ntotal, unsigned *__restrict__ nch, uint4 *__restrict__ changes,
                      uint4 *__restrict__ clist, unsigned *__restrict__ ntotal_next, const uint8_t *__restrict__ flag = nullptr,
                      const unsigned *__restrict__ tile_off = nullptr, uint16_t *__restrict__ compact = nullptr, unsigned cpitch = 0)
{
    const size_t f = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f == 0 && ntotal_next) *ntotal_next = 0u;
    if (f >= (size_t)nframes) return;
    hot_repair_frame(f, frames, npix, rows, cols, min_change, max_hot, count, pos, ntotal, nch, changes, clist);
    // compact (optional): the replaced pixels go into the compact [active pixel][frame] series as well (pass A stored them as they
    // were; column = the frame's index in this launch) -- by the lane that repaired the frame, no launch of its own
    if (compact) {
        const unsigned m = nch[f];
        for (unsigned j = 0; j < m; ++j) {
            const uint4 ch = changes[f * (size_t)max_hot + j];
            const unsigned fl = flag[ch.y];
            if (fl) compact[(size_t)(tile_off[ch.y / kFusedPix] + (fl & 0x7Fu)) * cpitch + ch.x] = (uint16_t)ch.w;    // (0: nobody reads the pixel)
        }
    }
}


// The same for several cameras.  The value of node n in frame f is sol = sum over the cameras, in order, of
// w_c * f32(frame_c[pix_c[n]]) in float (psp_process.cpp:1813-1819): a replaced pixel of one camera changes it
// non-linearly (float rounding), so the entry is RECOMPUTED from the repaired frames of all cameras --
// the gather's own arithmetic -- and swapped into the row; the accumulators move by (new - old) with old = the
// value the swap returned, which makes a second change on the same (node, frame) a no-op (several cameras,
// or several hot pixels of one camera, can land on one node).  Runs after the frames of ALL cameras are
// repaired.  Accumulators: sums of doubles in another order than the gather's (parity bar 1e-12).
struct HotMultiArgs {
    int ncams;
    uint16_t *frames[kMaxCams];
    const int32_t *pix[kMaxCams];
    const float *weight[kMaxCams];
};
// All cameras in one launch each (blockIdx.y = camera): camera c owns `words` words of d_changes (layout of
// hot_changes_words()), counters / positions at c * nframes, lists at c * npix (head) and c * nnodes (next).
__device__ __forceinline__ unsigned *hot_cam_changes(unsigned *d_changes, size_t words, int c) { return d_changes + (size_t)c * words; }
__device__ __forceinline__ uint4 *hot_cam_list(unsigned *chg, int nframes)
{
    return reinterpret_cast<uint4 *>(chg + 4 + (((size_t)nframes + 3) & ~(size_t)3));
}
__global__ void hot_reset_cams_kernel(unsigned *d_changes, size_t words, int ncams)
{
    if ((int)threadIdx.x < ncams) d_changes[(size_t)threadIdx.x * words] = 0u;
}
__global__ void __launch_bounds__(64)
    hot_repair_cams_kernel(HotMultiArgs a, size_t npix, int nframes, int rows, int cols, int min_change, int max_hot,
                           unsigned *__restrict__ count, const unsigned *__restrict__ pos, unsigned *d_changes, size_t words)
{
    const size_t f = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = (int)blockIdx.y;
    if (f >= (size_t)nframes) return;
    unsigned *chg = hot_cam_changes(d_changes, words, c);
    hot_repair_frame(f, a.frames[c], npix, rows, cols, min_change, max_hot, count + (size_t)c * nframes,
                     pos + (size_t)c * nframes * kHotCap, chg, chg + 4, hot_cam_list(chg, nframes));
}
// Lists only for the pixels that changed: head[] is kHotUnmarked everywhere between two fix-ups; the changed pixels
// are marked (-1 = empty list), the build threads only the nodes on marked pixels (a few hundred atomics instead of
// one per visible node), and after the patch the marks are taken back.
constexpr int32_t kHotUnmarked = -2;
__global__ void __launch_bounds__(256)
    hot_mark_cams_kernel(unsigned *d_changes, size_t words, int nframes, int max_hot, int32_t *__restrict__ head,
                         size_t npix, int32_t value)
{
    const int c = (int)blockIdx.y;
    unsigned *chg = hot_cam_changes(d_changes, words, c);
    if (chg[0] == 0u) return;                  // (uniform per camera)
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (unsigned)nframes * (unsigned)max_hot) return;
    const unsigned f = i / (unsigned)max_hot;
    if (i - f * (unsigned)max_hot >= chg[4 + f]) return;
    head[(size_t)c * npix + hot_cam_list(chg, nframes)[i].y] = value;
}
__global__ void __launch_bounds__(256)
    hot_lists_build_cams_kernel(HotMultiArgs a, unsigned *d_changes, size_t words, unsigned nnodes, size_t npix,
                                int32_t *__restrict__ head, int32_t *__restrict__ next)
{
    const int c = (int)blockIdx.y;
    if (hot_cam_changes(d_changes, words, c)[0] == 0u) return;     // (uniform per camera)
    const unsigned n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nnodes) return;
    const int32_t p = a.pix[c][n];
    if (p < 0) return;
    int32_t *h = head + (size_t)c * npix + p;
    // (a marked entry only ever moves between values >= -1: the plain read cannot mistake it for unmarked)
    if (*h != kHotUnmarked) next[(size_t)c * nnodes + n] = atomicExch(h, (int32_t)n);
}
__global__ void __launch_bounds__(256)
    hot_patch_multi_kernel(HotMultiArgs a, size_t npix, unsigned nnodes, unsigned *d_changes, size_t words, int nframes,
                           int max_hot, const int32_t *__restrict__ head, const int32_t *__restrict__ next,
                           const uint8_t *__restrict__ skipped, float *__restrict__ rows_t, long long ld_t,
                           double *__restrict__ sum, double *__restrict__ sumsq)
{
    const int cc = (int)blockIdx.y;
    unsigned *chg = hot_cam_changes(d_changes, words, cc);
    if (chg[0] == 0u) return;                  // (uniform per camera) nearly every call
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (unsigned)nframes * (unsigned)max_hot) return;
    const unsigned f = i / (unsigned)max_hot;
    if (i - f * (unsigned)max_hot >= chg[4 + f]) return;
    const uint4 ch = hot_cam_list(chg, nframes)[i];
    for (int32_t n = head[(size_t)cc * npix + ch.y]; n >= 0; n = next[(size_t)cc * nnodes + n]) {
        if (skipped && skipped[n]) continue;                  // stays NaN
        float sol = 0.0f;
        for (int c = 0; c < a.ncams; ++c) {
            const int32_t p = a.pix[c][n];
            const float w = a.weight[c] ? a.weight[c][n] : 1.0f;
            const float v = p >= 0 ? 0.0f + w * (float)a.frames[c][(size_t)ch.x * npix + (size_t)p] : 0.0f;
            sol = c == 0 ? v : sol + v;
        }
        const unsigned oldb = atomicExch(reinterpret_cast<unsigned *>(rows_t + (long long)n * ld_t + (long long)ch.x),
                                         __float_as_uint(sol));
        const float old = __uint_as_float(oldb);
        if (old != sol) {
            unsafeAtomicAdd(&sum[n], (double)sol - (double)old);
            unsafeAtomicAdd(&sumsq[n], (double)(sol * sol) - (double)(old * old));
        }
    }
}
__global__ void __launch_bounds__(256) hot_fill_i32_kernel(int32_t *p, size_t n, int32_t v)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ void finals_kernel(const double *__restrict__ sum, const double *__restrict__ sumsq,
                              unsigned nnodes, double nframes, float *__restrict__ avg,
                              float *__restrict__ rms)
{
    const unsigned n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nnodes) return;
    if (avg) avg[n] = (float)(sum[n] / nframes);           // psp_process.cpp:1934
    if (rms) rms[n] = (float)sqrt(sumsq[n] / nframes);     // psp_process.cpp:1935
}

// Row scatter for the time-series exchange: packed block src [nrows][ncols] (f32, or u16 for
// integer-valued series that travelled as u16) -> rows rowidx[r] of dst (f32, row pitch ld, column
// offset already applied).  A wave moves 4 rows, 16 B of output per lane and row.  Bound by the
// strided row-segment writes (190 k x 1 KB segments: 62 us = 3.5 TB/s, one or four rows per wave alike).
template <typename T>
__device__ __forceinline__ void load4(const T *p, float (&o)[4]);
template <>
__device__ __forceinline__ void load4<float>(const float *p, float (&o)[4])
{
    const float4 v = *reinterpret_cast<const float4 *>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
template <>
__device__ __forceinline__ void load4<uint16_t>(const uint16_t *p, float (&o)[4])
{
    const uint2 v = *reinterpret_cast<const uint2 *>(p);
    o[0] = (float)(v.x & 0xFFFFu); o[1] = (float)(v.x >> 16);
    o[2] = (float)(v.y & 0xFFFFu); o[3] = (float)(v.y >> 16);
}

template <typename T>
__global__ void __launch_bounds__(256)
    scatter_rows_kernel(const T *__restrict__ src, long long nrows, int ncols,
                        const long long *__restrict__ rowidx, float *__restrict__ dst, long long ld)
{
    const long long r0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    if (r0 >= nrows) return;
    const int lane = threadIdx.x & 63;
    const int nr = (int)((nrows - r0) < 4 ? (nrows - r0) : 4);
    const bool vec = ((ncols & 3) == 0) && ((ld & 3) == 0) &&
                     ((reinterpret_cast<size_t>(src) & (4 * sizeof(T) - 1)) == 0) &&
                     ((reinterpret_cast<size_t>(dst) & 15) == 0);
    if (vec) {
        typedef float v4f __attribute__((ext_vector_type(4)));   // streamed: the series is not re-read here
        for (int c = lane * 4; c < ncols; c += 256) {
            float v[4][4];
            long long row[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < nr) {
                    row[k] = rowidx[r0 + k];
                    load4<T>(src + (r0 + k) * ncols + c, v[k]);
                }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < nr) {
                    const v4f o = {v[k][0], v[k][1], v[k][2], v[k][3]};
                    __builtin_nontemporal_store(o, reinterpret_cast<v4f *>(dst + row[k] * ld + c));
                }
        }
    } else {
        for (int k = 0; k < nr; ++k) {
            const T *s = src + (r0 + k) * ncols;
            float *d = dst + rowidx[r0 + k] * ld;
            for (int c = lane; c < ncols; c += 64) d[c] = (float)s[c];
        }
    }
}

// ---------------------------------------------------------------- transpose --
// local_transpose (psp_process.cpp:647-689): dst[x][y] = src[y][x].  64x64 f32
// tile through LDS (row stride 65 words: conflict-free column reads), 256-byte
// coalesced rows on both sides.
__global__ void __launch_bounds__(256)
    transpose_kernel(const float *__restrict__ src, long long x_extent, long long y_extent,
                     float *__restrict__ dst, long long ld_dst)
{
    __shared__ float tile[64][65];
    const long long x0 = (long long)blockIdx.x * 64, y0 = (long long)blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
#pragma unroll
    for (int j = 0; j < 64; j += 4) {
        const long long y = y0 + ty + j, x = x0 + tx;
        if (y < y_extent && x < x_extent) tile[ty + j][tx] = src[y * x_extent + x];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 64; j += 4) {
        const long long x = x0 + ty + j, y = y0 + tx;
        if (x < x_extent && y < y_extent) dst[x * ld_dst + y] = tile[tx][ty + j];
    }
}

// ------------------------------------------------------------------ weights --
struct Centers {
    float c[kMaxCams][3];
};

// adjust_projection_for_weights + BestView / AverageViews
// (cpp/lib/projection.ipp:911-1078, 227-268; angle_between cv_extras.ipp:69-73)
__global__ void weights_kernel(int ncams, unsigned nnodes, const int32_t *__restrict__ pix,
                               float *__restrict__ weight, const float *__restrict__ nodes,
                               const float *__restrict__ normals, Centers ctr, int mode)
{
    const unsigned n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nnodes) return;
    float ang[kMaxCams];
    int cnt = 0;
    const float px = nodes[3 * (size_t)n], py = nodes[3 * (size_t)n + 1], pz = nodes[3 * (size_t)n + 2];
    const float nx = normals[3 * (size_t)n], ny = normals[3 * (size_t)n + 1], nz = normals[3 * (size_t)n + 2];
    for (int c = 0; c < ncams; ++c) {
        ang[c] = 0.0f;
        if (pix[(size_t)c * nnodes + n] < 0) continue;
        const float dx = px - ctr.c[c][0], dy = py - ctr.c[c][1], dz = pz - ctr.c[c][2];
        const float dot = dx * nx + dy * ny + dz * nz;
        const double n1 = sqrt((double)dx * dx + (double)dy * dy + (double)dz * dz);
        const double n2 = sqrt((double)nx * nx + (double)ny * ny + (double)nz * nz);
        ang[c] = (float)acos(dot / n1 / n2);
        ++cnt;
    }
    if (cnt < 2) return;
    if (mode == 0) {
        int best = -1;
        for (int c = 0; c < ncams; ++c) {
            if (pix[(size_t)c * nnodes + n] < 0) continue;
            if (best < 0 || ang[c] > ang[best]) best = c;
        }
        for (int c = 0; c < ncams; ++c)
            if (pix[(size_t)c * nnodes + n] >= 0) weight[(size_t)c * nnodes + n] *= (c == best) ? 1.0f : 0.0f;
    } else {
        float s = 0.0f;
        for (int c = 0; c < ncams; ++c)
            if (pix[(size_t)c * nnodes + n] >= 0) s += ang[c];
        for (int c = 0; c < ncams; ++c)
            if (pix[(size_t)c * nnodes + n] >= 0) weight[(size_t)c * nnodes + n] *= ang[c] / s;
    }
}

__global__ void skipped_kernel(int ncams, unsigned nnodes, const int32_t *__restrict__ pix,
                               uint8_t *__restrict__ skipped, unsigned long long *count)
{
    const unsigned n = blockIdx.x * blockDim.x + threadIdx.x;
    bool sk = false;
    if (n < nnodes) {
        bool found = false;
        for (int c = 0; c < ncams; ++c) found |= pix[(size_t)c * nnodes + n] >= 0;
        sk = !found;
        skipped[n] = sk ? 1 : 0;
    }
    const unsigned long long m = __ballot(sk);
    if ((threadIdx.x & 63) == 0 && m && count) atomicAdd(count, (unsigned long long)__popcll(m));
}

struct PixPtrs {
    const int32_t *p[kMaxCams];
};

// identify_skipped_nodes over per-camera arrays (no staging copy, no host sync)
__global__ void skipped_ptrs_kernel(int ncams, unsigned nnodes, PixPtrs pp,
                                    uint8_t *__restrict__ skipped)
{
    const unsigned n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nnodes) return;
    bool found = false;
    for (int c = 0; c < ncams; ++c) found |= pp.p[c][n] >= 0;
    skipped[n] = found ? 0 : 1;
}

__global__ void project_one_kernel(const void *img, int is_f32, const int32_t *__restrict__ pix,
                                   const float *__restrict__ weight, unsigned nnodes,
                                   float *__restrict__ out)
{
    const unsigned n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nnodes) return;
    const int32_t p = pix[n];
    float v = 0.0f;
    if (p >= 0) {
        const float pv = is_f32 ? reinterpret_cast<const float *>(img)[p]
                                : (float)reinterpret_cast<const uint16_t *>(img)[p];
        v = 0.0f + (weight ? weight[n] : 1.0f) * pv;
    }
    out[n] = v;
}

}  // namespace

// ---- launchers used by pipeline.cpp -----------------------------------------
size_t hot_counter_words(int nframes) { return (size_t)nframes * (1 + kTicketStride); }

int launch_hot_fix(uint16_t *d_frames, int nframes, int rows, int cols, int thresh,
                   int min_change, int max_hot, unsigned *d_count, unsigned *d_pos,
                   int32_t *d_status, hipStream_t st, const unsigned *d_only)
{
    if (nframes <= 0) return UPSP_OK;
    if (max_hot < 0 || max_hot >= kHotCap) return fail(UPSP_ERR_INVALID, "max_hot must be in [0,63]");
    const size_t npix = (size_t)rows * cols;
    // d_count holds hot_counter_words(nframes) counters (hot pixels per frame, then one padded
    // ticket counter per frame); zero when allocated, the kernel leaves them zero
    size_t bx = (npix / 8 + 255) / 256;
    if (bx > 128) bx = 128;
    if (bx < 1) bx = 1;
    KTimed kt("hot_scan_kernel", st);
    hipLaunchKernelGGL(hot_scan_kernel, dim3((unsigned)bx, (unsigned)nframes), dim3(256), 0, st,
                       d_frames, npix, thresh, d_count, d_pos, d_count + nframes, rows, cols,
                       min_change, max_hot, d_status, d_only);
    UPSP_HIP_CHECK(hipGetLastError());
    return UPSP_OK;
}

// ---- streamed scan + projection (scan_compact_kernel / node_stream_kernel) ----
size_t tilemap_tiles(size_t npix) { return (npix + kFusedPix - 1) / kFusedPix; }

// Active-pixel map of a projection: d_flag [npix] bytes (0 or 0x80 | rank in its tile), d_off
// [ntiles + 1] (compact index of a tile's first active pixel; d_off[ntiles] = number of active
// pixels), d_node_k [nnodes].  d_cnt [ntiles] is scratch.
int launch_amap_build(const int32_t *d_pix, size_t nnodes, size_t npix, uint8_t *d_flag, unsigned *d_cnt,
                      unsigned *d_off, int32_t *d_node_k, unsigned *d_order, hipStream_t st)
{
    const unsigned ntiles = (unsigned)tilemap_tiles(npix);
    KTimed kt("amap_build", st);
    const dim3 g((unsigned)((nnodes + 255) / 256)), b(256);
    hipLaunchKernelGGL(amap_mark_kernel, g, b, 0, st, d_pix, (unsigned)nnodes, d_flag);
    hipLaunchKernelGGL(amap_rank_kernel, dim3((unsigned)((npix + 255) / 256)), b, 0, st, d_flag, npix, d_cnt);
    // d_order (optional): visiting order [ntiles] + active-tile ranks [ntiles + 1]
    unsigned *arank = d_order ? d_order + ntiles : nullptr;
    // (the visiting order made by the scan's own workgroup from an LDS copy of the ranks -- one launch less -- measured: the map
    //  build 30 -> 45 us; 8192 searches and 64-bit divisions are no work for ONE workgroup)
    hipLaunchKernelGGL((tilemap_scan_kernel<256, 32>), dim3(1), dim3(256), 0, st, (const unsigned *)d_cnt, d_off, arank, ntiles);
    // (d_node_k null: a map built from a candidate set -- the nodes get their rows once the projection is there, launch_amap_nodes)
    if (d_node_k)
        hipLaunchKernelGGL(amap_nodes_kernel, g, b, 0, st, d_pix, (unsigned)nnodes, (const uint8_t *)d_flag,
                           (const unsigned *)d_off, d_node_k, (uint8_t *)nullptr);
    if (d_order)
        hipLaunchKernelGGL(amap_order_kernel, dim3((ntiles + 255) / 256), b, 0, st, (const unsigned *)arank, ntiles, d_order,
                           scan_active_last() ? 1 : 0);
    UPSP_HIP_CHECK(hipGetLastError());
    return UPSP_OK;
}

int group_frames_max() { return kGroupFramesMax; }

// Pass A for `nframes` frames (any number: one workgroup row per 64-frame group) into columns
// [col, col + nframes) of the compact buffer (row pitch cpitch u16 >= col + nframes rounded up to 64).
// d_count: one counter per frame (zero on entry; the repair launches leave them zero), d_pos: 64
// positions per frame.
int launch_scan_compact(uint16_t *d_frames, size_t npix, int nframes, bool hot, int thresh, int max_hot,
                        const uint8_t *d_flag, const unsigned *d_off, const unsigned *d_order, uint16_t *d_compact,
                        unsigned cpitch, int col, unsigned *d_count, unsigned *d_pos, hipStream_t st, bool split)
{
    if (nframes <= 0) return UPSP_OK;
    if (max_hot < 0 || max_hot >= kHotCap) return fail(UPSP_ERR_INVALID, "max_hot must be in [0,63]");
    const int ngroups = (nframes + 63) / 64;
    if ((col & 63) || (unsigned)(col + 64 * ngroups) > cpitch) return fail(UPSP_ERR_INVALID, "scan pass: compact buffer too narrow");
    const unsigned ntiles = (unsigned)tilemap_tiles(npix);
    KTimed kt("scan_compact_kernel", st);
    const bool lin = scan_active_last() && d_order;
    if (lin && split && hot && (npix % kFusedPix) == 0) {
        const unsigned agrid = (unsigned)std::min<size_t>((size_t)ntiles * ngroups, 8u * 256u);
        hipLaunchKernelGGL(scan_inactive_kernel, dim3(ntiles * (unsigned)ngroups), dim3(64), 0, st, d_frames, npix, nframes, ngroups,
                           (unsigned)thresh, (unsigned)max_hot, d_count, d_pos, d_order, ntiles);
        hipLaunchKernelGGL(scan_active_kernel<true>, dim3(agrid), dim3(256), 0, st, d_frames, npix, nframes, ngroups, d_flag, d_off,
                           d_compact + col, cpitch, (unsigned)thresh, (unsigned)max_hot, d_count, d_pos, d_order, ntiles);
        UPSP_HIP_CHECK(hipGetLastError());
        return UPSP_OK;
    }
    const dim3 grid = lin ? dim3(ntiles * (unsigned)ngroups) : dim3(ntiles, (unsigned)ngroups);
    if (hot)
        hipLaunchKernelGGL(scan_compact_kernel<true>, grid, dim3(256), 0, st, d_frames, npix, nframes,
                           d_flag, d_off, d_compact + col, cpitch, (unsigned)thresh, (unsigned)max_hot, d_count, d_pos, d_order, lin ? ntiles : 0u);
    else
        hipLaunchKernelGGL(scan_compact_kernel<false>, grid, dim3(256), 0, st, d_frames, npix, nframes,
                           d_flag, d_off, d_compact + col, cpitch, 0u, 0u, d_count, d_pos, d_order, lin ? ntiles : 0u);
    UPSP_HIP_CHECK(hipGetLastError());
    return UPSP_OK;
}

// Pass B, whole rows, for the g.nframes (<= group_frames_max()) frames parked in the compact buffer.
int launch_amap_nodes(const int32_t *d_pix, size_t nnodes, const uint8_t *d_flag, const unsigned *d_off,
                      int32_t *d_node_k, hipStream_t st, uint8_t *d_skipped_out)
{
    hipLaunchKernelGGL(amap_nodes_kernel, dim3((unsigned)((nnodes + 255) / 256)), dim3(256), 0, st, d_pix, (unsigned)nnodes,
                       d_flag, d_off, d_node_k, d_skipped_out);
    UPSP_HIP_CHECK(hipGetLastError());
    return UPSP_OK;
}

// g.img[0]: the group's first frame (u16) -- only read for nodes whose pixel is missing from the map (node_k == -2)
int launch_node_rows(const PipelineGather &g, const int32_t *d_node_k, const uint16_t *d_compact, unsigned cpitch,
                     hipStream_t st, bool cold_series, bool fresh_acc, const uint16_t *d_compact_b, unsigned cpitch_b,
                     int nframes_a)
{
    if (g.nframes <= 0 || g.nframes > kGroupFramesMax || (unsigned)(d_compact_b ? nframes_a : g.nframes) > cpitch)
        return fail(UPSP_ERR_INVALID, "row pass: too many frames");
    if (d_compact_b && (nframes_a <= 0 || nframes_a >= g.nframes || (nframes_a & 3) || (unsigned)(g.nframes - nframes_a) > cpitch_b))
        return fail(UPSP_ERR_INVALID, "row pass: bad split between the two series buffers");
    const unsigned nn = (unsigned)g.nnodes;
    KTimed kt("node_rows_kernel", st);
    // four sweeps per workgroup (1 / 2 / 4 / 8: 486 / 412 / 370 / 422 us per 1000 frames of the bench model).  Series loads: with
    // rows that end inside a 128-byte line the kernel waits for its stores and the arrangement of the loads does not matter (at
    // use / one sweep ahead / all up front: 0.403-0.413 ms); with padded rows (nstore, 0.32-0.33 ms alone on the device) all loads
    // up front is the default for f32 rows of > 512 frames: in the bench step, where part of the series has left the Infinity
    // Cache behind pass A's frames, 0.370-0.373 ms against 0.381-0.385 at use (tools/passb_probe.py: warm 0.33 either way, cold
    // 0.52 against 0.55; eight sweeps 0.41 cold but 0.375 warm -- the choice for series that are known to be cold)
#define UPSP_NRX(LPR, ROWS, U16, AH)                                                                         \
    hipLaunchKernelGGL((node_rows_kernel<LPR, ROWS, U16, AH>), dim3((nn + (256 / LPR) * ROWS - 1) / ((256 / LPR) * ROWS)), \
                       dim3(256), 0, st, d_compact, cpitch, d_node_k, g.skipped, g.rowmap, nn, g.nframes, g.rows_t,  \
                       g.rows_t16, (long long)g.ld_t, g.sum, g.sumsq, (const uint16_t *)g.img[0], g.npix, g.pix[0], fresh_acc,  \
                       std::min(std::max(g.nstore, g.nframes), 4 * (LPR)), d_compact_b, cpitch_b,                    \
                       d_compact_b ? nframes_a : g.nframes)
#define UPSP_NR(LPR, U16) UPSP_NRX(LPR, 4, U16, 0)
#define UPSP_NR_L(U16)                                                                                       \
    do {                                                                                                     \
        if (g.nframes > 512) UPSP_NR(256, U16); else if (g.nframes > 256) UPSP_NR(128, U16); else UPSP_NR(64, U16); \
    } while (0)
    // (measurement switch UPSP_ROWS_VARIANT = <sweeps><ahead>, f32 rows of > 512 frames: 40, 41, 42 default, 80, 81, 82 for cold series, 162)
    static const int variant = [] { const char *e = getenv("UPSP_ROWS_VARIANT"); return e ? atoi(e) : 0; }();
    const int var = variant ? variant : (cold_series ? 82 : 42);
    if (var && !g.rows_t16 && g.nframes > 512) {
        switch (var) {
        case 41: UPSP_NRX(256, 4, false, 1); break;
        case 42: UPSP_NRX(256, 4, false, 2); break;
        case 80: UPSP_NRX(256, 8, false, 0); break;
        case 81: UPSP_NRX(256, 8, false, 1); break;
        case 82: UPSP_NRX(256, 8, false, 2); break;
        case 162: UPSP_NRX(256, 16, false, 2); break;
        default: UPSP_NRX(256, 4, false, 0); break;
        }
    } else if (g.rows_t16) UPSP_NR_L(true); else UPSP_NR_L(false);
#undef UPSP_NRX
#undef UPSP_NR_L
#undef UPSP_NR
    UPSP_HIP_CHECK(hipGetLastError());
    return UPSP_OK;
}

// The windows of rows_from_pixel_blocks in launches of <= kMaxRowWindows windows each (node_rows_windows_kernel).
int launch_node_rows_windows(const RowWindow *w, int nwin, const int32_t *d_node_k, const uint8_t *d_skipped, size_t nnodes,
                             float *d_rows_t, int64_t ld, double *d_sum, double *d_sumsq, hipStream_t st)
{
    if (nwin < 1 || nnodes == 0 || nnodes >= ((size_t)1 << 31) || (ld & 3) || (reinterpret_cast<size_t>(d_rows_t) & 15))
        return fail(UPSP_ERR_INVALID, "row pass (windows): bad argument");
    for (int i = 0; i < nwin; ++i)
        if (w[i].nframes <= 0 || w[i].nframes > kGroupFramesMax || w[i].nstore < w[i].nframes || w[i].nstore > kGroupFramesMax ||
            ((w[i].nframes | w[i].nstore | w[i].nframes_a) & 3) || (w[i].col & 3) || w[i].nframes_a < 0 || w[i].nframes_a > w[i].nframes ||
            (w[i].nframes_a < w[i].nframes && !w[i].b) || (w[i].nframes_a > 0 && !w[i].a))
            return fail(UPSP_ERR_INVALID, "row pass (windows): bad window");
    // (measurement switch UPSP_WINDOW_ROWS = 4 / 8 / 16 nodes per workgroup)
    static const int rows = [] { const char *e = getenv("UPSP_WINDOW_ROWS"); return e ? atoi(e) : 8; }();
    KTimed kt("node_rows_kernel", st);
    for (int i0 = 0; i0 < nwin; i0 += kMaxRowWindows) {
        RowWindows win;
        win.n = std::min(kMaxRowWindows, nwin - i0);
        for (int i = 0; i < win.n; ++i) win.w[i] = w[i0 + i];
#define UPSP_NRW(ROWS)                                                                                                              \
        hipLaunchKernelGGL(node_rows_windows_kernel<ROWS>, dim3((unsigned)((nnodes + ROWS - 1) / ROWS)), dim3(256), 0, st, win, d_node_k, \
                           d_skipped, (unsigned)nnodes, d_rows_t, (long long)ld, d_sum, d_sumsq)
        if (rows == 4) UPSP_NRW(4); else if (rows == 16) UPSP_NRW(16); else UPSP_NRW(8);
#undef UPSP_NRW
    }
    UPSP_HIP_CHECK(hipGetLastError());
    return UPSP_OK;
}

// Pass B, whole rows, for g.ncams cameras and the g.nframes (<= group_frames_max()) frames parked in the
// per-camera compact buffers (row pitch cpitch).
int launch_node_rows_multi(const PipelineGather &g, const int32_t *const *d_node_k, const uint16_t *const *d_compact,
                           unsigned cpitch, hipStream_t st)
{
    if (g.nframes <= 0 || g.nframes > kGroupFramesMax || (unsigned)g.nframes > cpitch)
        return fail(UPSP_ERR_INVALID, "row pass: too many frames");
    if (g.ncams < 1 || g.ncams > kMaxCams || !g.rows_t) return fail(UPSP_ERR_INVALID, "row pass: bad camera count / no f32 rows");
    StreamMultiArgs a;
    std::memset(&a, 0, sizeof(a));
    a.ncams = g.ncams;
    for (int c = 0; c < g.ncams; ++c) {
        a.compact[c] = d_compact[c];
        a.node_k[c] = d_node_k[c];
        a.weight[c] = g.weight[c];
    }
    const unsigned nn = (unsigned)g.nnodes;
    KTimed kt("node_rows_multi_kernel", st);
#define UPSP_NRM(LPR, ROWS, NC, ...)                                                                          \
    hipLaunchKernelGGL((node_rows_multi_kernel<LPR, ROWS, NC, ##__VA_ARGS__>), dim3((nn + (256 / LPR) * ROWS - 1) / ((256 / LPR) * ROWS)), \
                       dim3(256), 0, st, a, cpitch, g.skipped, g.rowmap, nn, g.nframes, g.rows_t, (long long)g.ld_t, \
                       g.sum, g.sumsq)
#define UPSP_NRM8_NC(LPR, ROWS)                                                                               \
    do {                                                                                                      \
        if (g.ncams == 2) UPSP_NRM(LPR, ROWS, 2, 8, 0, 1);                                                    \
        else if (g.ncams == 3) UPSP_NRM(LPR, ROWS, 3, 8, 0, 1);                                               \
        else UPSP_NRM(LPR, ROWS, 4, 8, 0, 1);                                                                 \
    } while (0)
#define UPSP_NRM_NC(LPR, ROWS)                                                                                \
    do {                                                                                                      \
        if (g.ncams == 2) UPSP_NRM(LPR, ROWS, 2);                                                             \
        else if (g.ncams == 3) UPSP_NRM(LPR, ROWS, 3);                                                        \
        else if (g.ncams == 4) UPSP_NRM(LPR, ROWS, 4);                                                        \
        else UPSP_NRM(LPR, 2, 0);                                                                             \
    } while (0)
    // Measured on 4 cameras, 2.5 M nodes, 1000 frame sets (rounds 2-3).  Sweeps per workgroup: 2 / 4 / 8 -> 4.28 / 3.35 / 3.16 ms
    // (consecutive nodes share pixels, and the per-workgroup staging is paid once per 8 rows; with the leaner sums 2.97).  Eight
    // frames per lane as two groups of four (128 lanes per row of <= 1024 frames): 2.93 against 2.99-3.13 with four (as 8
    // CONSECUTIVE frames per lane, i.e. half-empty store instructions, 4.1).  Series loads ONE sweep in front of the rows that
    // use them (72 VGPRs, 7 waves per SIMD) instead of all four sweeps first (96, 5 waves): 2.71 against 2.86 on one box; two
    // sweeps in front 2.97.  Rejected: row / series indices through v_readfirstlane (3.51), persistent workgroups that prefetch
    // the next group's staging (130 VGPRs: 3.14-3.23).  The kernel is not bound by its instructions (VALU busy 53 %): a
    // workgroup lives ~12 us, most of it the two dependent memory latencies of its staging and its series loads.
    if (g.nframes > 512 && g.ncams >= 2 && g.ncams <= 4) UPSP_NRM8_NC(128, 4);
    else if (g.nframes > 512) UPSP_NRM_NC(256, 8);
    else if (g.nframes > 256) UPSP_NRM_NC(128, 4);
    else UPSP_NRM_NC(64, 4);
#undef UPSP_NRM8_NC
#undef UPSP_NRM_NC
#undef UPSP_NRM
    UPSP_HIP_CHECK(hipGetLastError());
    return UPSP_OK;
}

// Hot-pixel fix-up of all `nframes` frames of a call (any number; g.rows_t / g.rows_t16 point at the
// column of the first frame; d_count / d_pos hold one counter / 64 positions per frame).  Scratch:
// d_changes 4 words (word 0 = number of changes of the call) + nframes words (changes per frame) rounded
// up to a multiple of 4 + 4 * nframes * max_hot words (hot_changes_words()); d_head [npix], d_next [nnodes].
size_t hot_changes_words(int nframes, int max_hot)
{
    // counters (4) + changes per frame (rounded to 4) + the per-frame record slots + the same records once more, dense
    return 4 + (((size_t)nframes + 3) & ~(size_t)3) + 8 * (size_t)nframes * (size_t)std::max(max_hot, 1);
}
// Hot-pixel fix-up for several cameras after a streamed pass over UNREPAIRED frames (pass A counted the hot
// pixels per camera and frame): repair the frames of every camera, then re-project the replaced pixels.
// d_count / d_pos: per camera c at c * nframes (counters) and c * nframes * 64 (positions); d_changes: per camera
// hot_changes_words(nframes, max_hot) words; d_head: ncams * npix, d_next: ncams * nnodes.  g.rows_t points at
// the column of the first frame; identity row map only.
// Repairs the frames with 1 .. max_hot hot pixels (fix_hot_pixels) and writes the replaced pixels into the compact
// series pass A stored (frames of one pass A group: compact column = frame index).
int launch_hot_repair_compact(uint16_t *d_frames, size_t npix, int nframes, int rows, int cols, int min_change, int max_hot,
                              unsigned *d_count, const unsigned *d_pos, unsigned *d_changes, int *parity, const uint8_t *d_flag,
                              const unsigned *d_tile_off, uint16_t *d_compact, unsigned cpitch, hipStream_t st)
{
    if (nframes <= 0) return UPSP_OK;
    KTimed kt("hot_fixup_kernels", st);
    unsigned *nch = d_changes + 4;
    uint4 *list = reinterpret_cast<uint4 *>(d_changes + 4 + (((size_t)nframes + 3) & ~(size_t)3));
    // words 0 / 1 of the buffer: the change counter of this call / of the next one (both zero after the allocation)
    unsigned *ntotal = d_changes + (*parity & 1), *ntotal_next = d_changes + ((*parity & 1) ^ 1);
    *parity ^= 1;
    hipLaunchKernelGGL(hot_repair_kernel, dim3((unsigned)((nframes + 63) / 64)), dim3(64), 0, st, d_frames, npix, nframes,
                       rows, cols, min_change, max_hot, d_count, d_pos, ntotal, nch, list, (uint4 *)nullptr, ntotal_next, d_flag,
                       d_tile_off, max_hot > 0 ? d_compact : (uint16_t *)nullptr, cpitch);
    UPSP_HIP_CHECK(hipGetLastError());
    return UPSP_OK;
}

// Repair alone: d_count / d_pos were filled by a kernel that had the pixels in registers anyway (the registration's
// pre-blur, imageops.hip); afterwards d_changes holds, per frame, how many pixels were replaced (words 4 .. 4 + nframes) and
// the (frame, position, old, new) records behind them (layout of hot_changes_words()).  The counters clean themselves.
int launch_hot_repair_list(uint16_t *d_frames, size_t npix, int nframes, int rows, int cols, int min_change, int max_hot,
                           unsigned *d_count, const unsigned *d_pos, unsigned *d_changes, hipStream_t st)
{
    if (nframes <= 0) return UPSP_OK;
    if (max_hot < 0 || max_hot >= kHotCap) return fail(UPSP_ERR_INVALID, "max_hot must be in [0,63]");
    KTimed kt("hot_fixup_kernels", st);
    unsigned *nch = d_changes + 4;
    uint4 *list = reinterpret_cast<uint4 *>(d_changes + 4 + (((size_t)nframes + 3) & ~(size_t)3));
    // (this caller's consumers read the per-frame counts only: word 2 takes the unused total, the counters of the paths above stay clean)
    hipLaunchKernelGGL(hot_repair_kernel, dim3((unsigned)((nframes + 63) / 64)), dim3(64), 0, st, d_frames, npix, nframes,
                       rows, cols, min_change, max_hot, d_count, d_pos, d_changes + 2, nch, list, (uint4 *)nullptr, (unsigned *)nullptr);
    UPSP_HIP_CHECK(hipGetLastError());
    return UPSP_OK;
}

// `head_clean`: d_head (ncams * npix entries) holds kHotUnmarked everywhere (true after a previous call of this
// function; false after an allocation or when another path used the array): refilled here when false.
int launch_hot_fixup_multi(const PipelineGather &g, uint16_t *const *d_frames, int nframes, int rows, int cols,
                           int min_change, int max_hot, unsigned *d_count, const unsigned *d_pos,
                           unsigned *d_changes, int32_t *d_head, int32_t *d_next, bool head_clean, hipStream_t st)
{
    if (nframes <= 0) return UPSP_OK;
    if (g.rowmap || !g.rows_t) return fail(UPSP_ERR_INVALID, "multi-camera hot-pixel fix-up needs f32 rows without a row map");
    if (max_hot < 0 || max_hot >= kHotCap) return fail(UPSP_ERR_INVALID, "max_hot must be in [0,63]");
    KTimed kt("hot_fixup_kernels", st);
    const size_t words = hot_changes_words(nframes, max_hot);
    const unsigned C = (unsigned)g.ncams;
    HotMultiArgs a;
    std::memset(&a, 0, sizeof(a));
    a.ncams = g.ncams;
    for (int c = 0; c < g.ncams; ++c) {
        a.frames[c] = d_frames[c];
        a.pix[c] = g.pix[c];
        a.weight[c] = g.weight[c];
    }
    if (!head_clean && max_hot > 0) {
        const size_t n = (size_t)g.npix * C;
        hipLaunchKernelGGL(hot_fill_i32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_head, n, kHotUnmarked);
    }
    // every camera's frames are repaired (one launch) before anything is re-projected
    hipLaunchKernelGGL(hot_reset_cams_kernel, dim3(1), dim3(64), 0, st, d_changes, words, g.ncams);
    hipLaunchKernelGGL(hot_repair_cams_kernel, dim3((unsigned)((nframes + 63) / 64), C), dim3(64), 0, st, a, g.npix, nframes,
                       rows, cols, min_change, max_hot, d_count, d_pos, d_changes, words);
    if (max_hot > 0) {
        const dim3 slots((unsigned)(((size_t)nframes * max_hot + 255) / 256), C);
        hipLaunchKernelGGL(hot_mark_cams_kernel, slots, dim3(256), 0, st, d_changes, words, nframes, max_hot, d_head, g.npix,
                           (int32_t)-1);
        hipLaunchKernelGGL(hot_lists_build_cams_kernel, dim3((unsigned)((g.nnodes + 255) / 256), C), dim3(256), 0, st, a,
                           d_changes, words, (unsigned)g.nnodes, g.npix, d_head, d_next);
        hipLaunchKernelGGL(hot_patch_multi_kernel, slots, dim3(256), 0, st, a, g.npix, (unsigned)g.nnodes, d_changes, words,
                           nframes, max_hot, (const int32_t *)d_head, (const int32_t *)d_next, g.skipped, g.rows_t,
                           (long long)g.ld_t, g.sum, g.sumsq);
        hipLaunchKernelGGL(hot_mark_cams_kernel, slots, dim3(256), 0, st, d_changes, words, nframes, max_hot, d_head, g.npix,
                           kHotUnmarked);
    }
    UPSP_HIP_CHECK(hipGetLastError());
    return UPSP_OK;
}

int launch_gather(const PipelineGather &g, hipStream_t st)
{
    if (g.nnodes == 0 || g.nframes <= 0) return UPSP_OK;
    GatherArgs a;
    std::memset(&a, 0, sizeof(a));
    a.ncams = g.ncams;
    a.npix = g.npix;
    a.src = g.src;
    a.rowmap = g.rowmap;
    a.rows_t16 = g.rows_t16;
    for (int c = 0; c < g.ncams; ++c) {
        a.img[c] = g.img[c];
        a.pix[c] = g.pix[c];
        a.weight[c] = g.weight[c];
        a.is_f32[c] = g.is_f32[c];
    }
    if (g.nframes <= 64) {
        KTimed kt("gather_tile_kernel", st);
        const dim3 tgrid((unsigned)((g.nnodes + 63) / 64)), tblock(256);
        if (g.ncams == 1 && !g.weight[0] && !g.is_f32[0]) {
            // exact 16-bit values: u16 LDS tile, more gathers in flight per CU
#define UPSP_T16(W, SS, HS)                                                                    \
    hipLaunchKernelGGL((gather_tile16_kernel<W, SS, HS>), tgrid, dim3(W * 64), 0, st, a,       \
                       g.skipped, (unsigned)g.nnodes, g.nframes, g.rows, g.rows_t,             \
                       (long long)g.ld_t, g.sum, g.sumsq)
            // (non-temporal stores: the series is not read again by the loop; 4 waves 38 us, 2 waves 42, 8 waves 41 per 64 frames)
            if (g.src) UPSP_T16(4, true, true); else UPSP_T16(4, true, false);
#undef UPSP_T16
            UPSP_HIP_CHECK(hipGetLastError());
            return UPSP_OK;
        }
#define UPSP_TILE2(NC, SS, HS)                                                                 \
    hipLaunchKernelGGL((gather_tile_kernel<NC, SS, HS>), tgrid, tblock, 0, st, a, g.skipped,   \
                       (unsigned)g.nnodes, g.nframes, g.rows, g.rows_t, (long long)g.ld_t,     \
                       g.sum, g.sumsq)
#define UPSP_TILE(NC)                                                                          \
    do {                                                                                       \
        if (g.src) UPSP_TILE2(NC, true, true); else UPSP_TILE2(NC, true, false);               \
    } while (0)
        switch (g.ncams) {  // per-camera pix / weight stay in registers for 1..4 cameras
            case 1: UPSP_TILE(1); break;
            case 2: UPSP_TILE(2); break;
            case 3: UPSP_TILE(3); break;
            case 4: UPSP_TILE(4); break;
            default: UPSP_TILE(0); break;
        }
#undef UPSP_TILE
#undef UPSP_TILE2
        UPSP_HIP_CHECK(hipGetLastError());
        return UPSP_OK;
    }
    if (g.rows_t || g.rows_t16) return fail(UPSP_ERR_INVALID, "transposed output needs sub-batches of <= 64 frames");
    const dim3 grid((unsigned)((g.nnodes + 255) / 256)), block(256);
    if (g.ncams == 1)
        hipLaunchKernelGGL((gather_kernel<1>), grid, block, 0, st, a, g.skipped, (unsigned)g.nnodes,
                           g.nframes, g.rows, g.sum, g.sumsq);
    else
        hipLaunchKernelGGL((gather_kernel<0>), grid, block, 0, st, a, g.skipped, (unsigned)g.nnodes,
                           g.nframes, g.rows, g.sum, g.sumsq);
    UPSP_HIP_CHECK(hipGetLastError());
    return UPSP_OK;
}

int launch_skipped(int ncams, size_t nnodes, const int32_t *const *d_pix, uint8_t *d_skipped,
                   hipStream_t st)
{
    PixPtrs pp;
    for (int c = 0; c < kMaxCams; ++c) pp.p[c] = c < ncams ? d_pix[c] : nullptr;
    hipLaunchKernelGGL(skipped_ptrs_kernel, dim3((unsigned)((nnodes + 255) / 256)), dim3(256), 0, st,
                       ncams, (unsigned)nnodes, pp, d_skipped);
    UPSP_HIP_CHECK(hipGetLastError());
    return UPSP_OK;
}

int launch_finals(const double *sum, const double *sumsq, size_t nnodes, uint64_t nframes,
                  float *avg, float *rms, hipStream_t st)
{
    if (nnodes == 0) return UPSP_OK;
    hipLaunchKernelGGL(finals_kernel, dim3((unsigned)((nnodes + 255) / 256)), dim3(256), 0, st, sum,
                       sumsq, (unsigned)nnodes, (double)nframes, avg, rms);
    UPSP_HIP_CHECK(hipGetLastError());
    return UPSP_OK;
}

}  // namespace upsp

using namespace upsp;

// rows rowidx[r] of dst (pitch ld) <- value, columns [0, ncols): the workgroup sweeps one row per store instruction
// (4 KB for a 1000-frame row, like pass B), 8 rows per workgroup, 16-B streaming stores
constexpr int kFillRows = 16;
__global__ void __launch_bounds__(256)
    fill_rows_kernel(long long nrows, int ncols, const long long *__restrict__ rowidx, float *__restrict__ dst,
                     long long ld, float value)
{
    const long long r0 = (long long)blockIdx.x * kFillRows;
    const bool vec = ((ld & 3) == 0) && ((reinterpret_cast<size_t>(dst) & 15) == 0);
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f o = {value, value, value, value};
    long long rows[kFillRows];       // every index first: the stores below do not wait for one load each
#pragma unroll
    for (int k = 0; k < kFillRows; ++k) rows[k] = r0 + k < nrows ? rowidx[r0 + k] : -1;
#pragma unroll
    for (int k = 0; k < kFillRows; ++k) {
        if (rows[k] < 0) break;
        float *d = dst + rows[k] * ld;
        if (vec) {
            int c = (int)threadIdx.x * 4;
            for (; c + 3 < ncols; c += 1024) __builtin_nontemporal_store(o, reinterpret_cast<v4f *>(d + c));
            for (; c < ncols; ++c) d[c] = value;     // (the thread that holds the ragged end)
        } else {
            for (int c = threadIdx.x; c < ncols; c += 256) d[c] = value;
        }
    }
}

extern "C" {

int upsp_fill_rows_f32(float value, size_t nrows, int ncols, const int64_t *d_rowidx, float *d_dst, long long ld,
                       void *stream)
{
    if (nrows == 0 || ncols == 0) return UPSP_OK;
    if (!d_rowidx || !d_dst || ncols < 0 || ld < ncols) return fail(UPSP_ERR_INVALID, "bad argument");
    hipLaunchKernelGGL(fill_rows_kernel, dim3((unsigned)((nrows + kFillRows - 1) / kFillRows)), dim3(256), 0, (hipStream_t)stream,
                       (long long)nrows, ncols, reinterpret_cast<const long long *>(d_rowidx), d_dst, ld, value);
    UPSP_HIP_CHECK(hipGetLastError());
    return UPSP_OK;
}

int upsp_scatter_rows_f32(const float *d_src, size_t nrows, int ncols, const int64_t *d_rowidx,
                          float *d_dst, long long ld, void *stream)
{
    if (nrows == 0 || ncols == 0) return UPSP_OK;
    if (!d_src || !d_rowidx || !d_dst || ncols < 0 || ld < ncols) return fail(UPSP_ERR_INVALID, "bad argument");
    hipLaunchKernelGGL(scatter_rows_kernel<float>, dim3((unsigned)((nrows + 15) / 16)), dim3(256), 0, (hipStream_t)stream,
                       d_src, (long long)nrows, ncols, reinterpret_cast<const long long *>(d_rowidx), d_dst, ld);
    UPSP_HIP_CHECK(hipGetLastError());
    return UPSP_OK;
}

int upsp_scatter_rows_u16(const uint16_t *d_src, size_t nrows, int ncols, const int64_t *d_rowidx,
                          float *d_dst, long long ld, void *stream)
{
    if (nrows == 0 || ncols == 0) return UPSP_OK;
    if (!d_src || !d_rowidx || !d_dst || ncols < 0 || ld < ncols) return fail(UPSP_ERR_INVALID, "bad argument");
    hipLaunchKernelGGL(scatter_rows_kernel<uint16_t>, dim3((unsigned)((nrows + 15) / 16)), dim3(256), 0, (hipStream_t)stream,
                       d_src, (long long)nrows, ncols, reinterpret_cast<const long long *>(d_rowidx), d_dst, ld);
    UPSP_HIP_CHECK(hipGetLastError());
    return UPSP_OK;
}

int upsp_fix_hot_pixels(uint16_t *d_frames, int nframes, int rows, int cols, int thresh,
                        int min_change, int max_hot, int32_t *d_status, void *stream)
{
    if (nframes == 0) return UPSP_OK;
    if (!d_frames || nframes < 0 || rows <= 0 || cols <= 0)
        return fail(UPSP_ERR_INVALID, "bad frame buffer / size");
    hipStream_t st = (hipStream_t)stream;
    unsigned *cnt = nullptr, *pos = nullptr;
    UPSP_HIP_CHECK(hipMalloc(&cnt, sizeof(unsigned) * hot_counter_words(nframes)));
    UPSP_HIP_CHECK(hipMemsetAsync(cnt, 0, sizeof(unsigned) * hot_counter_words(nframes), st));
    hipError_t e = hipMalloc(&pos, sizeof(unsigned) * (size_t)nframes * kHotCap);
    int rc = UPSP_OK;
    if (e != hipSuccess) {
        rc = fail(UPSP_ERR_HIP, hipGetErrorString(e));
    } else {
        rc = launch_hot_fix(d_frames, nframes, rows, cols, thresh, min_change, max_hot, cnt, pos,
                            d_status, st);
        if (rc == UPSP_OK && hipStreamSynchronize(st) != hipSuccess)
            rc = fail(UPSP_ERR_HIP, "hot-pixel kernels failed");
    }
    (void)hipFree(cnt);
    if (pos) (void)hipFree(pos);
    return rc;
}

static int project_one(const void *img, int is_f32, const int32_t *d_pix, const float *d_weight,
                       size_t nnodes, float *d_out, void *stream)
{
    if (nnodes == 0) return UPSP_OK;
    if (!img || !d_pix || !d_out) return fail(UPSP_ERR_INVALID, "null device buffer");
    hipLaunchKernelGGL(project_one_kernel, dim3((unsigned)((nnodes + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, img, is_f32, d_pix, d_weight, (unsigned)nnodes, d_out);
    UPSP_HIP_CHECK(hipGetLastError());
    return UPSP_OK;
}

int upsp_project_frame_u16(const uint16_t *d_img, const int32_t *d_pix, const float *d_weight,
                           size_t nnodes, float *d_out, void *stream)
{
    return project_one(d_img, 0, d_pix, d_weight, nnodes, d_out, stream);
}

int upsp_project_frame_f32(const float *d_img, const int32_t *d_pix, const float *d_weight,
                           size_t nnodes, float *d_out, void *stream)
{
    return project_one(d_img, 1, d_pix, d_weight, nnodes, d_out, stream);
}

int upsp_transpose_f32(const float *d_src, int64_t x_extent, int64_t y_extent, float *d_dst,
                       int64_t ld_dst, void *stream)
{
    if (x_extent == 0 || y_extent == 0) return UPSP_OK;
    if (!d_src || !d_dst || x_extent < 0 || y_extent < 0 || ld_dst < y_extent)
        return fail(UPSP_ERR_INVALID, "bad transpose arguments");
    const long long gx = (x_extent + 63) / 64, gy = (y_extent + 63) / 64;
    if (gy > 65535) return fail(UPSP_ERR_INVALID, "y_extent too large for one launch");
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0,
                       (hipStream_t)stream, d_src, (long long)x_extent, (long long)y_extent, d_dst,
                       (long long)ld_dst);
    UPSP_HIP_CHECK(hipGetLastError());
    return UPSP_OK;
}

/* apportion, cpp/exec/psp_process.cpp:611-624 */
int upsp_apportion(int value, int nbins, int *h_start, int *h_extent)
{
    if (value < 0 || nbins <= 0 || !h_start || !h_extent)
        return fail(UPSP_ERR_INVALID, "bad apportion arguments");
    const unsigned long block = (unsigned long)(value / nbins);
    const unsigned long rem = (unsigned long)value - block * (unsigned long)nbins;
    unsigned long next = 0;
    for (unsigned long b = 0; b < (unsigned long)nbins; ++b) {
        h_start[b] = (int)next;
        h_extent[b] = (int)(block + (b < rem ? 1 : 0));
        next += (unsigned long)h_extent[b];
    }
    return UPSP_OK;
}

int upsp_projection_weights(int ncams, size_t nnodes, const int32_t *d_pix, float *d_weight,
                            const float *d_nodes, const float *d_normals,
                            const double *h_centers, int mode, void *stream)
{
    if (ncams <= 0 || ncams > kMaxCams) return fail(UPSP_ERR_INVALID, "ncams out of range");
    if (nnodes == 0) return UPSP_OK;
    if (!d_pix || !d_weight || !d_nodes || !d_normals || !h_centers)
        return fail(UPSP_ERR_INVALID, "null argument");
    if (mode != 0 && mode != 1) return fail(UPSP_ERR_INVALID, "mode must be 0 or 1");
    Centers ctr;
    std::memset(&ctr, 0, sizeof(ctr));
    for (int c = 0; c < ncams; ++c)
        for (int a = 0; a < 3; ++a) ctr.c[c][a] = (float)h_centers[3 * c + a];
    hipLaunchKernelGGL(weights_kernel, dim3((unsigned)((nnodes + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, ncams, (unsigned)nnodes, d_pix, d_weight, d_nodes,
                       d_normals, ctr, mode);
    UPSP_HIP_CHECK(hipGetLastError());
    return UPSP_OK;
}

int upsp_projection_skipped(int ncams, size_t nnodes, const int32_t *d_pix, uint8_t *d_skipped,
                            uint64_t *h_count, void *stream)
{
    if (ncams <= 0) return fail(UPSP_ERR_INVALID, "ncams out of range");
    if (h_count) *h_count = 0;
    if (nnodes == 0) return UPSP_OK;
    if (!d_pix || !d_skipped) return fail(UPSP_ERR_INVALID, "null argument");
    hipStream_t st = (hipStream_t)stream;
    unsigned long long *cnt = nullptr;
    if (h_count) {
        UPSP_HIP_CHECK(hipMalloc(&cnt, sizeof(*cnt)));
        UPSP_HIP_CHECK(hipMemsetAsync(cnt, 0, sizeof(*cnt), st));
    }
    hipLaunchKernelGGL(skipped_kernel, dim3((unsigned)((nnodes + 255) / 256)), dim3(256), 0, st,
                       ncams, (unsigned)nnodes, d_pix, d_skipped, cnt);
    hipError_t e = hipGetLastError();
    if (h_count && e == hipSuccess) {
        unsigned long long h = 0;
        e = hipMemcpyAsync(&h, cnt, sizeof(h), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        *h_count = h;
    }
    if (cnt) (void)hipFree(cnt);
    if (e != hipSuccess) return fail(UPSP_ERR_HIP, hipGetErrorString(e));
    return UPSP_OK;
}

}  // extern "C"
