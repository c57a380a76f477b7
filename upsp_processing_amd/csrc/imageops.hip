// Image stages of the psp_process frame loop on MI355X (gfx950):
//   register (inverse-map warp; the ECC itself: ecc.hip)   cpp/lib/registration.cpp:32-81
//   patch    (cubic 2-D polynomial over fiducials)          cpp/lib/patches.ipp:98-236
//   filter   (GaussianBlur / blur)                          cpp/exec/psp_process.cpp:1802-1807
// and the driver that runs them on a sub-batch of frames (run_frame_stages).
//
// The arithmetic of these stages lives in OpenCV 4.5.2 / Eigen 3.3.9 in the
// reference (un-vendored, no reference test: PARITY UNPINNED): the kernels follow the published
// algorithms as restated in oracle/image_oracle.c.
#include <hip/hip_runtime.h>
#include <functional>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "imageops.h"
#include "ktimer.h"
#include "pipeline.h"
#include "upsp_internal.h"

namespace upsp {
namespace {

constexpr int kMaxKernel = 63;   // largest odd filter size

__global__ void u16_to_f32_kernel(const uint16_t *__restrict__ src, float *__restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x)
        dst[i] = (float)src[i];
}

// ---------------------------------------------------------------- filters --
struct FilterCoef {
    float k[kMaxKernel];
    int size;
};

// cv::getGaussianKernel(k, sigma <= 0, CV_32F)
int gaussian_coef(int k, FilterCoef &fc)
{
    if (k < 1 || (k & 1) == 0 || k > kMaxKernel) return -1;
    fc.size = k;
    static const float t1[] = {1.f};
    static const float t3[] = {0.25f, 0.5f, 0.25f};
    static const float t5[] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};
    static const float t7[] = {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f};
    const float *fixed = k == 1 ? t1 : k == 3 ? t3 : k == 5 ? t5 : k == 7 ? t7 : nullptr;
    if (fixed) {
        std::memcpy(fc.k, fixed, sizeof(float) * (size_t)k);
        return 0;
    }
    const double sigma = ((k - 1) * 0.5 - 1) * 0.3 + 0.8;
    const double scale2 = -0.5 / (sigma * sigma);
    double sum = 0;
    for (int i = 0; i < k; ++i) {
        const double x = i - (k - 1) * 0.5;
        fc.k[i] = (float)std::exp(scale2 * x * x);
        sum += fc.k[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < k; ++i) fc.k[i] = (float)(fc.k[i] * sum);
    return 0;
}

// Symmetric separable filter, one pass (HORIZ: along x, else along y), float
// accumulation in the order  k[r]*c + sum_j k[r+j]*(a[-j] + a[+j]).
template <typename SRC, bool HORIZ>
__global__ void __launch_bounds__(256)
    gauss_pass_kernel(const SRC *__restrict__ src, float *__restrict__ dst, int rows, int cols,
                      FilterCoef fc)
{
    const size_t npix = (size_t)rows * cols;
    const SRC *s = src + (size_t)blockIdx.y * npix;
    float *d = dst + (size_t)blockIdx.y * npix;
    const int r = fc.size / 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix;
         i += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / (size_t)cols), x = (int)(i % (size_t)cols);
        float acc = fc.k[r] * (float)s[i];
        for (int j = 1; j <= r; ++j) {
            float a, b;
            if (HORIZ) {
                a = (float)s[(size_t)y * cols + reflect101(x - j, cols)];
                b = (float)s[(size_t)y * cols + reflect101(x + j, cols)];
            } else {
                a = (float)s[(size_t)reflect101(y - j, rows) * cols + x];
                b = (float)s[(size_t)reflect101(y + j, rows) * cols + x];
            }
            acc += fc.k[r + j] * (a + b);
        }
        d[i] = acc;
    }
}

// Both passes of the separable Gaussian in one kernel for small kernels (radius R <= 3, the
// sizes 3 / 5 / 7 psp_process is run with): a 64 x 32 output tile, its input with a halo of R
// in LDS (rows / columns reflected at the image border exactly like the two passes do), the
// row pass into a second LDS buffer, the column pass to the output.  Same float operations in
// the same order as gauss_pass_kernel (the row-pass result is rounded to float in LDS as it was
// in the intermediate image), so the result is bit-identical -- with 1/3 of the HBM traffic
// (and none for a separate u16 -> f32 conversion).  Not usable in place.
constexpr int kGaussTW = 64, kGaussTH = 32;
template <typename SRC, int R, int TW = kGaussTW, int TH = kGaussTH>
__global__ void __launch_bounds__(256)
    gauss_fused_kernel(const SRC *__restrict__ src, float *__restrict__ dst, int rows, int cols, FilterCoef fc)
{
    constexpr int IW = TW + 2 * R, IH = TH + 2 * R;
    __shared__ float in[IH][IW + 1];
    __shared__ float hb[IH][TW + 1];
    const size_t npix = (size_t)rows * cols;
    const SRC *s = src + (size_t)blockIdx.z * npix;
    float *d = dst + (size_t)blockIdx.z * npix;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    if (x0 - R >= 0 && x0 + TW + R <= cols && y0 - R >= 0 && y0 + TH + R <= rows) {
        // tile and halo inside the image (all tiles but the outermost ring): no reflection, and (row, column) of the
        // staged element carried along instead of a division per element -- staging was ~40 of the kernel's ~64 VALU
        // instructions per output pixel (PMC: 67 M wave-instructions per 64-frame launch, VALU-bound at 74 %)
        constexpr int dT = 256 / IW, dU = 256 % IW;
        int t = (int)threadIdx.x / IW, u = (int)threadIdx.x % IW;
        const SRC *base = s + (size_t)(y0 - R) * cols + (x0 - R);
        for (int i = threadIdx.x; i < IH * IW; i += 256) {
            in[t][u] = (float)base[(unsigned)(t * cols + u)];
            t += dT;
            u += dU;
            if (u >= IW) {
                u -= IW;
                ++t;
            }
        }
    } else {
        for (int i = threadIdx.x; i < IH * IW; i += 256) {
            const int t = i / IW, u = i % IW;
            // positions past the image (+ halo) are never used: clamp before reflecting
            const int yy = reflect101(min(max(y0 - R + t, -(rows - 1)), 2 * rows - 2), rows);
            const int xx = reflect101(min(max(x0 - R + u, -(cols - 1)), 2 * cols - 2), cols);
            in[t][u] = (float)s[(size_t)yy * cols + xx];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < IH * TW; i += 256) {
        const int t = i / TW, x = i % TW;
        float acc = fc.k[R] * in[t][x + R];
#pragma unroll
        for (int j = 1; j <= R; ++j) acc += fc.k[R + j] * (in[t][x + R - j] + in[t][x + R + j]);
        hb[t][x] = acc;
    }
    __syncthreads();
    // column pass: a thread takes four consecutive rows of one column (4 + 2R reads for four outputs instead of
    // 4 x (2R + 1)); lanes run along x, so the LDS reads stay conflict-free and the stores 256-B row pieces
    static_assert(TH % 4 == 0, "tile height");
    for (int i = threadIdx.x; i < (TH / 4) * TW; i += 256) {
        const int yb = 4 * (i / TW), x = i % TW;
        float h[4 + 2 * R];
#pragma unroll
        for (int j = 0; j < 4 + 2 * R; ++j) h[j] = hb[yb + j][x];
        if (x0 + x < cols) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float acc = fc.k[R] * h[k + R];
#pragma unroll
                for (int j = 1; j <= R; ++j) acc += fc.k[R + j] * (h[k + R - j] + h[k + R + j]);
                if (y0 + yb + k < rows) d[(size_t)(y0 + yb + k) * cols + x0 + x] = acc;
            }
        }
    }
}

// cv::blur: double sums, scale 1/(k*k) (box filter with CV_64F sums for CV_32F input)
template <bool HORIZ>
__global__ void __launch_bounds__(256)
    box_pass_kernel(const void *__restrict__ src_, void *__restrict__ dst_, int rows, int cols, int k)
{
    const size_t npix = (size_t)rows * cols;
    const int r = k / 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix;
         i += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / (size_t)cols), x = (int)(i % (size_t)cols);
        double s = 0;
        if (HORIZ) {
            const float *src = reinterpret_cast<const float *>(src_) + (size_t)blockIdx.y * npix;
            for (int j = -r; j <= r; ++j) s += src[(size_t)y * cols + reflect101(x + j, cols)];
            (reinterpret_cast<double *>(dst_) + (size_t)blockIdx.y * npix)[i] = s;
        } else {
            const double *src = reinterpret_cast<const double *>(src_) + (size_t)blockIdx.y * npix;
            for (int j = -r; j <= r; ++j) s += src[(size_t)reflect101(y + j, rows) * cols + x];
            (reinterpret_cast<float *>(dst_) + (size_t)blockIdx.y * npix)[i] =
                (float)(s * (1.0 / ((double)k * k)));
        }
    }
}

unsigned grid_for_pixels(size_t npix)
{
    size_t g = (npix + 255) / 256;
    return (unsigned)std::min<size_t>(g, 1024);
}

// GaussianBlur(src,dst,Size(k,k),0) for nimg images; tmp: nimg*npix floats
// the four-pixels-per-lane 5 x 5 kernel (defined below); false: not applicable, nothing launched
bool launch_gauss5_quad(const uint16_t *src, float *dst, int nimg, int rows, int cols, const FilterCoef &fc, hipStream_t st,
                        unsigned thresh = 0, unsigned *d_count = nullptr, unsigned *d_pos = nullptr);
inline bool launch_gauss5_quad(const float *, float *, int, int, int, const FilterCoef &, hipStream_t) { return false; }

template <typename SRC>
int launch_gauss(const SRC *src, float *dst, float *tmp, int nimg, int rows, int cols, int k,
                 hipStream_t st)
{
    FilterCoef fc;
    if (gaussian_coef(k, fc) != 0) return fail(UPSP_ERR_INVALID, "filter size must be odd and <= 63");
    const dim3 grid(grid_for_pixels((size_t)rows * cols), (unsigned)nimg), block(256);
    KTimed kt("gauss_pass_kernels", st);
    const int r = k / 2;
    if (r >= 1 && r <= 3 && (const void *)src != (const void *)dst && rows > r && cols > r && nimg <= 65535) {
        const dim3 fgrid((unsigned)((cols + kGaussTW - 1) / kGaussTW), (unsigned)((rows + kGaussTH - 1) / kGaussTH),
                         (unsigned)nimg);
        if (r == 1) hipLaunchKernelGGL((gauss_fused_kernel<SRC, 1>), fgrid, block, 0, st, src, dst, rows, cols, fc);
        else if (r == 2 && launch_gauss5_quad(src, dst, nimg, rows, cols, fc, st)) {}        // (u16 frames, cols % 4 == 0)
        else if (r == 2) hipLaunchKernelGGL((gauss_fused_kernel<SRC, 2>), fgrid, block, 0, st, src, dst, rows, cols, fc);
        else hipLaunchKernelGGL((gauss_fused_kernel<SRC, 3>), fgrid, block, 0, st, src, dst, rows, cols, fc);
        UPSP_HIP_CHECK(hipGetLastError());
        return UPSP_OK;
    }
    hipLaunchKernelGGL((gauss_pass_kernel<SRC, true>), grid, block, 0, st, src, tmp, rows, cols, fc);
    hipLaunchKernelGGL((gauss_pass_kernel<float, false>), grid, block, 0, st, (const float *)tmp, dst,
                       rows, cols, fc);
    UPSP_HIP_CHECK(hipGetLastError());
    return UPSP_OK;
}

// -------------------------------------------------------------- warpAffine --
// cv::warpAffine(u16, M, INTER_LINEAR|NEAREST + WARP_INVERSE_MAP) for every frame.
// list (may be null; [0] = count, then pixel indices): only the listed pixels are produced -- the warped frame of the
// frame loop is a scratch image that nothing but the gather reads when registration is the last image stage, and the
// gather reads the ~6 % of the pixels that carry a node (1000 frames of 1024^2: 2.7 ms for every pixel; 1.2 ms with
// a byte mask tested per pixel and frame; the list is what is left of it).
__global__ void __launch_bounds__(256)
    warp_u16_kernel(const uint16_t *__restrict__ src, uint16_t *__restrict__ dst, int rows, int cols,
                    const EccState *__restrict__ state, int interp, const unsigned *__restrict__ list)
{
    const size_t npix = (size_t)rows * cols;
    const uint16_t *s = src + (size_t)blockIdx.y * npix;
    uint16_t *d = dst + (size_t)blockIdx.y * npix;
    const EccState &es = state[blockIdx.y];
    double M[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) M[i] = es.M[i];
    const bool identity = es.done == 2;
    const size_t nwork = list ? (size_t)list[0] : npix;
    for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwork;
         w += (size_t)gridDim.x * blockDim.x) {
        const size_t i = list ? (size_t)list[1 + w] : w;
        if (identity) {
            d[i] = s[i];
            continue;
        }
        const int y = (int)(i / (size_t)cols), x = (int)(i % (size_t)cols);
        const WarpCoord c = warp_coord(M, x, y, interp);
        uint16_t o;
        if (interp) {
            const float v = bilinear([&](int yy, int xx) { return (float)s[(size_t)yy * cols + xx]; },
                                     rows, cols, c);
            const int iv = (int)rintf(v);  // saturate_cast<ushort>(float)
            o = (uint16_t)max(0, min(65535, iv));
        } else {
            o = ((unsigned)c.sx < (unsigned)cols && (unsigned)c.sy < (unsigned)rows)
                    ? s[(size_t)c.sy * cols + c.sx]
                    : (uint16_t)0;
        }
        d[i] = o;
    }
}

// Registration as the LAST image stage (no patch, no filter) with node-major series wanted: the warped frame is a
// scratch image only the projection reads, and the projection reads the active pixels (~6 % of a frame).  The warp
// then writes those pixels straight into the compact [active pixel][frame] buffer of the streamed schedule -- 64
// pixels x the <= 64 frames of the sub-batch per workgroup, transposed through LDS so that an active pixel's 64 frames
// leave as one 128-byte piece -- and pass B (node_rows_kernel, whole 4-KB rows, 0.63 of the HBM peak) writes the series
// once per <= 1024 frames instead of gather_tile_kernel per 64 (0.29).  pix_of_k: pixel of every compact row.
__global__ void __launch_bounds__(256)
    amap_pixels_kernel(const uint8_t *__restrict__ flag, const unsigned *__restrict__ tile_off, size_t npix,
                       unsigned *__restrict__ pix_of_k)
{
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= npix) return;
    const unsigned fl = flag[p];
    if (fl) pix_of_k[tile_off[p / 128] + (fl & 0x7Fu)] = (unsigned)p;
}

__global__ void __launch_bounds__(256)
    warp_compact_kernel(const uint16_t *__restrict__ src, int rows, int cols, const EccState *__restrict__ state,
                        int nframes, int interp, const unsigned *__restrict__ pix_of_k, const unsigned *__restrict__ nact_ptr,
                        uint16_t *__restrict__ compact, unsigned cpitch, unsigned col0)
{
    __shared__ uint16_t tile[64][66];            // [frame][pixel]
    const unsigned nact = *nact_ptr;
    const unsigned k0 = blockIdx.x * 64u;
    if (k0 >= nact) return;
    {   // blockIdx.y: the 64-frame piece of the launch (a workgroup transposes 64 pixels x <= 64 frames)
        const int h = 64 * (int)blockIdx.y;
        src += (size_t)h * rows * cols;
        state += h;
        nframes = min(64, nframes - h);
        col0 += (unsigned)h;
    }
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned k = k0 + (unsigned)lane;
    const bool valid = k < nact;
    const size_t npix = (size_t)rows * cols;
    const unsigned p = valid ? pix_of_k[k] : 0u;
    const int y = (int)(p / (unsigned)cols), x = (int)(p % (unsigned)cols);
    // A wave takes the frames wave, wave + 4, ... -- four of them per step, their 16 source pixels loaded before any is
    // used (one frame per step was a chain of 16 gather latencies per wave: 31 us per 64-frame sub-batch).  The loads are
    // unconditional on clamped coordinates; a footprint that leaves the image (or nearest-neighbour mode) takes the generic path.
    constexpr int UF = 4;
    for (int fb = wave; fb < nframes; fb += 4 * UF) {            // (uniform per wave)
        WarpCoord c[UF];
        unsigned short t[UF][4];
        bool live[UF], raw[UF], fast[UF];
#pragma unroll
        for (int u = 0; u < UF; ++u) {
            const int f = fb + 4 * u;
            live[u] = f < nframes;
            const EccState &es = state[live[u] ? f : fb];
            raw[u] = es.done == 2;                               // frame 0 of a run: never registered
            double M[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) M[i] = es.M[i];
            c[u] = warp_coord(M, x, y, interp);
            fast[u] = interp && (unsigned)c[u].sx < (unsigned)(cols - 1) && (unsigned)c[u].sy < (unsigned)(rows - 1);
            const uint16_t *s = src + (size_t)(live[u] ? f : fb) * npix;
            const int lx = max(0, min(cols - 2, c[u].sx)), ly = max(0, min(rows - 2, c[u].sy));
            const uint16_t *q = s + (size_t)ly * cols + lx;
            t[u][0] = t[u][1] = t[u][2] = t[u][3] = 0;
            if (rows >= 2 && cols >= 2) {                        // (uniform)
                t[u][0] = q[0];
                t[u][1] = q[1];
                t[u][2] = q[cols];
                t[u][3] = q[cols + 1];
            }
        }
#pragma unroll
        for (int u = 0; u < UF; ++u) {
            if (!live[u]) break;                                 // (uniform)
            const int f = fb + 4 * u;
            uint16_t o;
            if (raw[u]) {
                o = src[(size_t)f * npix + p];
            } else if (fast[u]) {
                // remapBilinear's four-weight form (bilinear() above), same operations in the same order
                const float fx = c[u].ax * (1.f / 32), fy = c[u].ay * (1.f / 32);
                const float w0 = (1.f - fy) * (1.f - fx), w1 = (1.f - fy) * fx, w2 = fy * (1.f - fx), w3 = fy * fx;
                const float v = (float)t[u][0] * w0 + (float)t[u][1] * w1 + (float)t[u][2] * w2 + (float)t[u][3] * w3;
                const int iv = (int)rintf(v);  // saturate_cast<ushort>(float)
                o = (uint16_t)max(0, min(65535, iv));
            } else {
                const uint16_t *s = src + (size_t)f * npix;
                if (interp) {
                    const float v = bilinear([&](int yy, int xx) { return (float)s[(size_t)yy * cols + xx]; }, rows, cols, c[u]);
                    const int iv = (int)rintf(v);
                    o = (uint16_t)max(0, min(65535, iv));
                } else {
                    o = ((unsigned)c[u].sx < (unsigned)cols && (unsigned)c[u].sy < (unsigned)rows) ? s[(size_t)c[u].sy * cols + c[u].sx]
                                                                                                    : (uint16_t)0;
                }
            }
            tile[f][lane] = o;
        }
    }
    __syncthreads();
    // 4 threads per pixel, 16 frames (32 bytes) each: an active pixel's frames of this sub-batch are one 128-byte piece
    const int q = threadIdx.x >> 2, j = threadIdx.x & 3;
    if (k0 + (unsigned)q < nact) {
        unsigned w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int f = 16 * j + 2 * i;
            const unsigned a = f < nframes ? tile[f][q] : 0u, b = f + 1 < nframes ? tile[f + 1][q] : 0u;
            w[i] = a | (b << 16);
        }
        uint16_t *dst = compact + (size_t)(k0 + (unsigned)q) * cpitch + col0 + 16u * (unsigned)j;
        if (16 * j < nframes) *reinterpret_cast<uint4 *>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
        if (16 * j + 8 < nframes) *reinterpret_cast<uint4 *>(dst + 8) = make_uint4(w[4], w[5], w[6], w[7]);
    }
}

// ---- Gaussian 5 x 5 of u16 frames, FOUR pixels per lane (round 3) -------------------------------------------------------
// The tile kernel (gauss_fused_kernel) moves 6 B per pixel at 3.0 TB/s: its 2-byte loads and the LDS round trip keep
// few bytes in flight.  Here a lane owns four consecutive pixels of a row -- one 8-byte load (a wave reads 512
// contiguous bytes), one 16-byte store (a wave writes 1 KB) -- and walks down its rows with the five rows of horizontal
// results rolling in registers; the two pixels it needs from either neighbour come by DPP wave shifts (lanes 0 / 63
// load theirs, 4 bytes each), so the horizontal pass costs 4 shuffles per FOUR pixels.  No LDS, no barrier, waves
// independent.  Same float operations in the same order as gauss_pass_kernel (bit-identical output:
// tests/test_imageops_gpu.py::test_blur_u16_bitwise); reflect-101 by loading reflected rows, and at the left / right
// image edge by taking the reflected columns from the lane's own four pixels.  Needs cols % 4 == 0 and cols >= 8.
template <int U>
__global__ void __launch_bounds__(256)
    gauss5_quad_kernel(const uint16_t *__restrict__ src, float *__restrict__ dst, int rows, int cols, int rpp, float k0, float k1,
                       float k2, unsigned thresh, unsigned *__restrict__ hot_count, unsigned *__restrict__ hot_pos)
{
    // hot_count (may be null): the scan of fix_hot_pixels (cv_extras.cpp:237-247) on the pixels the blur loads anyway -- every
    // pixel >= thresh of the rows this block OWNS is counted per frame and its position recorded (kHotPositions per frame)
    // grid (column groups, row pieces, frames): frames slowest -- 1.27 against 1.35 ms of pre-blur per 1000 frames of 1024^2
    // with the frame as the fast index, and the identity iteration behind it 1-2 % faster
    const int f = blockIdx.z;
    const int bz = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int xw = (bz * 4 + (int)(threadIdx.x >> 6)) * 256;    // first column of this wave
    if (xw >= cols) return;                                                    // (uniform per wave)
    const int x0 = xw + 4 * lane;
    const bool valid = x0 < cols;                                              // all four pixels or none (cols % 4 == 0)
    const int xl = valid ? x0 : cols - 4;                                      // (idle lanes re-read the last quad)
    const int y0 = (int)blockIdx.y * rpp, y1 = min(rows, y0 + rpp);
    if (y1 <= y0) return;
    const size_t npix = (size_t)rows * cols;
    const char *S = reinterpret_cast<const char *>(src + (size_t)f * npix);
    char *B = reinterpret_cast<char *>(dst + (size_t)f * npix);
    const unsigned pitch2 = 2u * (unsigned)cols, cx = 2u * (unsigned)xl;
    const bool left_img = x0 == 0, right_img = x0 + 4 >= cols;                 // the image's own edges: reflected columns
    const bool halo_l = lane == 0 && !left_img, halo_r = lane == 63 && !right_img && valid;
    float h[5][4];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) h[i][j] = 0.f;
    const int niter = (y1 - y0) + 4;                                           // input rows y0 - 2 .. y1 + 1
    for (int g = 0; g < niter; g += U) {
        uint2 q[U];
        unsigned hl[U], hr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int yi = y0 - 2 + g + u;
            const unsigned ro = (unsigned)reflect101(min(max(yi, -(rows - 1)), 2 * rows - 2), rows) * pitch2;
            q[u] = *reinterpret_cast<const uint2 *>(S + (ro + cx));
            hl[u] = hr[u] = 0u;
            if (halo_l) hl[u] = *reinterpret_cast<const unsigned *>(S + (ro + cx - 4u));     // columns x0 - 2, x0 - 1
            if (halo_r) hr[u] = *reinterpret_cast<const unsigned *>(S + (ro + cx + 8u));     // columns x0 + 4, x0 + 5
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = g + u;
            if (k >= niter) break;                                             // (uniform)
            if (hot_count && valid) {
                const int yi = y0 - 2 + k;
                const unsigned v0 = q[u].x & 0xFFFFu, v1 = q[u].x >> 16, v2 = q[u].y & 0xFFFFu, v3 = q[u].y >> 16;
                if (yi >= y0 && yi < y1 && max(max(v0, v1), max(v2, v3)) >= thresh) {       // (rare)
                    const unsigned vv[4] = {v0, v1, v2, v3};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (vv[j] >= thresh) {
                            const unsigned slot = atomicAdd(&hot_count[f], 1u);
                            if (slot < (unsigned)kHotPositions) hot_pos[(size_t)f * kHotPositions + slot] = (unsigned)(yi * cols + x0 + j);
                        }
                }
            }
            const float p0 = (float)(q[u].x & 0xFFFFu), p1 = (float)(q[u].x >> 16), p2 = (float)(q[u].y & 0xFFFFu), p3 = (float)(q[u].y >> 16);
            // the neighbours' pixels: lane i - 1's p2, p3 and lane i + 1's p0, p1 (lanes 0 / 63: the loaded halo)
            float a2 = dpp_shr1((float)(hl[u] & 0xFFFFu), p2), a3 = dpp_shr1((float)(hl[u] >> 16), p3);
            float c0 = dpp_shl1((float)(hr[u] & 0xFFFFu), p0), c1 = dpp_shl1((float)(hr[u] >> 16), p1);
            if (left_img) { a3 = p1; a2 = p2; }                                // columns -1, -2 -> 1, 2
            if (right_img) { c0 = p2; c1 = p1; }                               // columns cols, cols + 1 -> cols - 2, cols - 3
            float n0 = k0 * p0, n1 = k0 * p1, n2 = k0 * p2, n3 = k0 * p3;
            n0 += k1 * (a3 + p1); n1 += k1 * (p0 + p2); n2 += k1 * (p1 + p3); n3 += k1 * (p2 + c0);
            n0 += k2 * (a2 + p2); n1 += k2 * (a3 + p3); n2 += k2 * (p0 + c0); n3 += k2 * (p1 + c1);
#pragma unroll
            for (int j = 0; j < 4; ++j) { h[0][j] = h[1][j]; h[1][j] = h[2][j]; h[2][j] = h[3][j]; h[3][j] = h[4][j]; }
            h[4][0] = n0; h[4][1] = n1; h[4][2] = n2; h[4][3] = n3;
            if (k < 4) continue;
            const int yb = y0 - 2 + k - 2;                                     // blurred row: the middle of the five
            float4 o;
            float *ov = &o.x;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float bn = k0 * h[2][j];
                bn += k1 * (h[1][j] + h[3][j]);
                bn += k2 * (h[0][j] + h[4][j]);
                ov[j] = bn;
            }
            if (valid) *reinterpret_cast<float4 *>(B + ((unsigned)yb * 2u * pitch2 + 4u * (unsigned)x0)) = o;
        }
    }
}

// The blurred pixels a repaired hot pixel reaches: the 5 x 5 neighbourhood of every listed change, recomputed from the
// repaired frame with gauss5_quad_kernel's own operations in its order (bit-identical to blurring the repaired frame).
// One workgroup per frame; frames without a change (nearly all) return at once.
__global__ void __launch_bounds__(256)
    reblur_changes_kernel(const uint16_t *__restrict__ frames, float *__restrict__ dst, int rows, int cols,
                          const unsigned *__restrict__ nch, const uint4 *__restrict__ changes, int max_hot, float k0, float k1, float k2)
{
    const int f = blockIdx.x;
    const unsigned m = nch[f];
    if (m == 0u) return;
    const size_t npix = (size_t)rows * cols;
    const uint16_t *S = frames + (size_t)f * npix;
    float *B = dst + (size_t)f * npix;
    for (unsigned t = threadIdx.x; t < m * 25u; t += 256u) {
        const uint4 ch = changes[(size_t)f * max_hot + t / 25u];
        const int o = (int)(t % 25u);
        const int y = (int)(ch.y / (unsigned)cols) + o / 5 - 2, x = (int)(ch.y % (unsigned)cols) + o % 5 - 2;
        if (y < 0 || y >= rows || x < 0 || x >= cols) continue;
        const int xm1 = reflect101(x - 1, cols), xp1 = reflect101(x + 1, cols), xm2 = reflect101(x - 2, cols), xp2 = reflect101(x + 2, cols);
        float h[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const uint16_t *row = S + (size_t)reflect101(y + i - 2, rows) * cols;
            float hn = k0 * (float)row[x];
            hn += k1 * ((float)row[xm1] + (float)row[xp1]);
            hn += k2 * ((float)row[xm2] + (float)row[xp2]);
            h[i] = hn;
        }
        float bn = k0 * h[2];
        bn += k1 * (h[1] + h[3]);
        bn += k2 * (h[0] + h[4]);
        B[(size_t)y * cols + x] = bn;         // (neighbourhoods of two changes may overlap: the same value from both)
    }
}

bool launch_gauss5_quad(const uint16_t *src, float *dst, int nimg, int rows, int cols, const FilterCoef &fc, hipStream_t st,
                        unsigned thresh, unsigned *d_count, unsigned *d_pos)
{
    if ((cols & 3) || cols < 8 || rows < 3 || (long long)rows * cols >= (1ll << 29) || (reinterpret_cast<uintptr_t>(src) & 7) ||
        (reinterpret_cast<uintptr_t>(dst) & 15) || nimg > 65535 || ((size_t)rows * cols & 3))      // (every frame of the batch 8-byte aligned)
        return false;
    // (1000 frames of 1024^2, ms of pre-blur per step: 2 / 4 / 6 / 8 rows in flight 1.60 / 1.50 / 1.35 / 1.25-1.33 at 64 rows per
    //  piece; 8 rows in flight at 16 / 32 / 48 / 128 rows per piece 1.32 / 1.36 / 1.39 / 1.52; the tile kernel 2.13)
    const int rpp = 64;
    const int pieces = (rows + rpp - 1) / rpp, zb = (cols + 1023) / 1024;
    if (pieces > 65535 || zb > 65535) return false;
    hipLaunchKernelGGL((gauss5_quad_kernel<8>), dim3((unsigned)zb, (unsigned)pieces, (unsigned)nimg), dim3(256), 0, st, src, dst, rows,
                       cols, rpp, fc.k[2], fc.k[3], fc.k[4], thresh, d_count, d_pos);
    return true;
}

// ----------------------------------------------------------------- patches --
// Opt-in (UPSP_PATCH_PINV=1), a better-conditioned answer than the reference's: per cluster coef = P z (P = pseudo-inverse of
// the cubic design matrix in centred, scaled coordinates, built once on the host in double), then evaluated at the
// interior pixels.  One workgroup per (cluster, frame).  The default follows the reference's float QR (patch_qr_kernel).
struct ClusterDesc {
    int b_off, nb, i_off, ni;
    float cx, cy, sx, sy;  // x' = (x - cx) * sx
};

__device__ __forceinline__ void monomials(float x, float y, float *m)
{
    // order of polyfit2D: y^i x^j, i outer, j inner, i + j <= 3 (patches.ipp:185-193)
    const float x2 = x * x, y2 = y * y;
    m[0] = 1.f; m[1] = x; m[2] = x2; m[3] = x2 * x;
    m[4] = y; m[5] = y * x; m[6] = y * x2;
    m[7] = y2; m[8] = y2 * x;
    m[9] = y2 * y;
}

__global__ void __launch_bounds__(256)
    patch_kernel(float *__restrict__ imgs, size_t npix, int cols, const ClusterDesc *__restrict__ cl,
                 int cluster0, const int32_t *__restrict__ b_idx, const float *__restrict__ P,
                 const int32_t *__restrict__ i_idx)
{
    const ClusterDesc d = cl[cluster0 + blockIdx.x];
    float *img = imgs + (size_t)blockIdx.y * npix;
    if (d.nb < 10) return;  // too few boundary points (patches.ipp:103)
    __shared__ double red[4][10];
    __shared__ float coef[10];
    double acc[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[k] = 0.0;
    for (int j = threadIdx.x; j < d.nb; j += blockDim.x) {
        const float z = img[b_idx[d.b_off + j]];
#pragma unroll
        for (int k = 0; k < 10; ++k) acc[k] += (double)P[(size_t)10 * (d.b_off + j) + k] * z;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        double v = acc[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 10)
        coef[threadIdx.x] = (float)(((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) +
                                    red[3][threadIdx.x]);
    __syncthreads();
    for (int j = threadIdx.x; j < d.ni; j += blockDim.x) {
        const int idx = i_idx[d.i_off + j];
        const float x = ((float)(idx % cols) - d.cx) * d.sx, y = ((float)(idx / cols) - d.cy) * d.sy;
        float m[10];
        monomials(x, y, m);
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 10; ++k) v += coef[k] * m[k];
        img[idx] = v;
    }
}

// Host: pseudo-inverse of the m x 10 design matrix by column-pivoted Householder QR
// in double (rank-truncated).  P is returned as [m][10] (row = boundary point).
void pinv_design(const std::vector<double> &A, int m, std::vector<float> &P)
{
    const int nc = 10;
    std::vector<double> Q(A);  // column-major m x nc
    std::vector<double> Rinv;  // unused
    std::vector<int> perm(nc);
    std::vector<double> tau(nc, 0.0);
    for (int k = 0; k < nc; ++k) perm[k] = k;
    auto col = [&](int j) { return &Q[(size_t)j * m]; };
    double maxnorm = 0;
    int rank = nc;
    for (int k = 0; k < nc; ++k) {
        int big = k;
        double bign = -1;
        for (int j = k; j < nc; ++j) {
            double s = 0;
            for (int r = k; r < m; ++r) s += col(j)[r] * col(j)[r];
            if (s > bign) { bign = s; big = j; }
        }
        if (k == 0) maxnorm = std::sqrt(bign);
        if (std::sqrt(bign) <= 1e-10 * maxnorm) { rank = k; break; }
        if (big != k) {
            for (int r = 0; r < m; ++r) std::swap(col(k)[r], col(big)[r]);
            std::swap(perm[k], perm[big]);
        }
        double *c = col(k);
        double tail = 0;
        for (int r = k + 1; r < m; ++r) tail += c[r] * c[r];
        const double c0 = c[k];
        double beta = std::sqrt(c0 * c0 + tail);
        if (c0 >= 0) beta = -beta;
        if (tail == 0) { tau[k] = 0; continue; }
        for (int r = k + 1; r < m; ++r) c[r] /= (c0 - beta);
        tau[k] = (beta - c0) / beta;
        c[k] = beta;
        for (int j = k + 1; j < nc; ++j) {
            double *cj = col(j);
            double t = cj[k];
            for (int r = k + 1; r < m; ++r) t += c[r] * cj[r];
            cj[k] -= tau[k] * t;
            for (int r = k + 1; r < m; ++r) cj[r] -= tau[k] * c[r] * t;
        }
    }
    // P = Pi * [R11^-1 0] * Q^T : apply to each unit vector e_r of R^m
    P.assign((size_t)m * nc, 0.f);
    std::vector<double> v(m);
    for (int r0 = 0; r0 < m; ++r0) {
        std::fill(v.begin(), v.end(), 0.0);
        v[r0] = 1.0;
        for (int k = 0; k < rank; ++k) {
            if (tau[k] == 0) continue;
            const double *c = col(k);
            double t = v[k];
            for (int r = k + 1; r < m; ++r) t += c[r] * v[r];
            v[k] -= tau[k] * t;
            for (int r = k + 1; r < m; ++r) v[r] -= tau[k] * c[r] * t;
        }
        double sol[10] = {0};
        for (int i = rank - 1; i >= 0; --i) {
            double s = v[i];
            for (int j = i + 1; j < rank; ++j) s -= col(j)[i] * sol[j];
            sol[i] = s / col(i)[i];
        }
        for (int k = 0; k < rank; ++k) P[(size_t)r0 * nc + perm[k]] = (float)sol[k];
    }
}

// ---- the reference's arithmetic (default) ------------------------------------------------------------------------------
// polyfit2D (cpp/lib/patches.ipp:172-205) solves  min |A p - z|  with Eigen::ColPivHouseholderQR<MatrixXf> on the RAW pixel
// coordinates: A(r, count) = (float)pow(y, i) * (float)pow(x, j), i outer / j inner, i + j <= 3 -- a float matrix whose columns
// span 1 .. 1e10, so the answer carries the float-QR noise of that conditioning, and "the reference's result" includes it.
// A depends on the cluster's geometry only: it is factored ONCE on the host, in float, operation for operation as Eigen 3.3.9's
// ColPivHouseholderQR::computeInPlace does (column norms with down-dating, pivot transpositions, makeHouseholderInPlace,
// applyHouseholderOnTheLeft, the rank rule of solve()).  Per frame the device does what solve() + polyval2D (:208-236) do: c = Q^T z
// (each reflector: a running float sum over the rows, in row order), back substitution on R, the inverse permutation, and
// the polynomial at the interior pixels, term by term in float.  Lane = FRAME: the 64 frames of a wave share the cluster, so
// the control flow and every matrix element are wave-uniform (scalar loads) and each lane runs the reference's sequential
// arithmetic on its own frame -- same operations, same order, same roundings (-ffp-contract=off, correctly rounded division).
struct QrDesc {
    int b_off, nb, i_off, ni;
    int v_off;        // first float of this cluster's factored matrix in V: nb x 10, column-major (reflector k below the diagonal of
                      // column k, R on and above it)
    int nonzero;      // nonzeroPivots(): columns that take part in the solve
    float tau[10];    // hCoeffs
    int perm[10];     // colsPermutation: solution entry k is coefficient perm[k]
};

constexpr int kPatchLdsRows = 240;      // 240 rows x 64 lanes x 4 B = 60 KB of LDS; longer boundaries keep c in global scratch

__global__ void __launch_bounds__(64)
    patch_qr_kernel(float *__restrict__ imgs, size_t npix, int nimg, int cols, const QrDesc *__restrict__ cl, int cluster0,
                    const int32_t *__restrict__ b_idx, const float *__restrict__ V, const int32_t *__restrict__ i_idx,
                    float *__restrict__ scratch, int scratch_rows)
{
    extern __shared__ float patch_lds[];
    const QrDesc &d = cl[cluster0 + blockIdx.x];
    const int m = d.nb;
    if (m < 10) return;  // too few boundary points (patches.ipp:103)
    const int lane = threadIdx.x;
    const int f = blockIdx.y * 64 + lane;
    const bool live = f < nimg;
    float *img = imgs + (size_t)(live ? f : 0) * npix;
    float *c = (m <= kPatchLdsRows ? patch_lds : scratch + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (size_t)scratch_rows * 64) + lane;
    for (int r = 0; r < m; ++r) c[r * 64] = img[b_idx[d.b_off + r]];
    const float *A = V + d.v_off;
    const int nz = d.nonzero;
    // c = Q^T z : the reflectors in order (HouseholderSequence::applyThisOnTheLeft of the transposed sequence)
    for (int k = 0; k < nz; ++k) {
        const float tau = d.tau[k];
        if (tau == 0.f) continue;
        const float *col = A + (size_t)k * m;
        float tmp = 0.f;
        for (int r = k + 1; r < m; ++r) tmp += col[r] * c[r * 64];
        tmp += c[k * 64];
        c[k * 64] -= tau * tmp;
        for (int r = k + 1; r < m; ++r) c[r * 64] -= tau * col[r] * tmp;
    }
    // R(0:nz, 0:nz) sol = c(0:nz): back substitution, row by row from the bottom
    float sol[10];
#pragma unroll
    for (int i = 9; i >= 0; --i) {
        float sv = 0.f;
        if (i < nz) {
            sv = c[i * 64];
#pragma unroll
            for (int j = i + 1; j < 10; ++j)
                if (j < nz) sv -= A[(size_t)j * m + i] * sol[j];
            sv = sv / A[(size_t)i * m + i];
        }
        sol[i] = sv;
    }
    // poly = colsPermutation * [sol; 0]
    float poly[10];
#pragma unroll
    for (int t = 0; t < 10; ++t) poly[t] = 0.f;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        const int pk = d.perm[k];
#pragma unroll
        for (int t = 0; t < 10; ++t)
            if (k < nz && pk == t) poly[t] = sol[k];
    }
    // polyval2D: z += poly[count] * (T)pow(y, i) * (T)pow(x, j), count running over i outer / j inner
    for (int j = 0; j < d.ni; ++j) {
        const int idx = i_idx[d.i_off + j];
        const double xd = (double)(idx % cols), yd = (double)(idx / cols);      // (integer powers below 2^53: exact, like pow())
        const float px[4] = {1.f, (float)xd, (float)(xd * xd), (float)(xd * xd * xd)};
        const float py[4] = {1.f, (float)yd, (float)(yd * yd), (float)(yd * yd * yd)};
        float acc = 0.f;
        int cnt = 0;
#pragma unroll
        for (int i = 0; i <= 3; ++i)
#pragma unroll
            for (int jj = 0; jj <= 3 - i; ++jj) acc += poly[cnt++] * py[i] * px[jj];
        if (live) img[idx] = acc;
    }
}

// Eigen 3.3.9 ColPivHouseholderQR<MatrixXf>::computeInPlace on the m x 10 column-major matrix A (overwritten with the
// reflectors and R), every intermediate a float.
void colpiv_householder_f32(std::vector<float> &A, int m, QrDesc &d)
{
    constexpr int nc = 10;
    const int size = std::min(m, nc);
    float norm_upd[nc], norm_dir[nc];
    int transp[nc];
    float biggest_norm = 0.f;
    for (int k = 0; k < nc; ++k) {
        float sq = 0.f;
        for (int r = 0; r < m; ++r) sq += A[(size_t)k * m + r] * A[(size_t)k * m + r];
        norm_dir[k] = norm_upd[k] = std::sqrt(sq);
        biggest_norm = std::max(biggest_norm, norm_upd[k]);
    }
    const float th = biggest_norm * FLT_EPSILON / (float)m;
    const float threshold_helper = th * th;
    const float downdate_threshold = std::sqrt(FLT_EPSILON);
    int nonzero = size;
    for (int k = 0; k < size; ++k) {
        int big = k;
        for (int j = k + 1; j < nc; ++j)
            if (norm_upd[j] > norm_upd[big]) big = j;
        const float big_sq = norm_upd[big] * norm_upd[big];
        if (nonzero == size && big_sq < threshold_helper * (float)(m - k)) nonzero = k;
        transp[k] = big;
        if (big != k) {
            for (int r = 0; r < m; ++r) std::swap(A[(size_t)k * m + r], A[(size_t)big * m + r]);
            std::swap(norm_upd[k], norm_upd[big]);
            std::swap(norm_dir[k], norm_dir[big]);
        }
        float *col = &A[(size_t)k * m];
        float tail_sq = 0.f;
        for (int r = k + 1; r < m; ++r) tail_sq += col[r] * col[r];
        const float c0 = col[k];
        float beta, tau;
        if (tail_sq <= FLT_MIN) {
            tau = 0.f;
            beta = c0;
            for (int r = k + 1; r < m; ++r) col[r] = 0.f;
        } else {
            beta = std::sqrt(c0 * c0 + tail_sq);
            if (c0 >= 0.f) beta = -beta;
            for (int r = k + 1; r < m; ++r) col[r] = col[r] / (c0 - beta);
            tau = (beta - c0) / beta;
        }
        d.tau[k] = tau;
        col[k] = beta;
        if (tau != 0.f)
            for (int j = k + 1; j < nc; ++j) {
                float *cj = &A[(size_t)j * m];
                float tmp = 0.f;
                for (int r = k + 1; r < m; ++r) tmp += col[r] * cj[r];
                tmp += cj[k];
                cj[k] -= tau * tmp;
                for (int r = k + 1; r < m; ++r) cj[r] -= tau * col[r] * tmp;
            }
        for (int j = k + 1; j < nc; ++j) {
            if (norm_upd[j] == 0.f) continue;
            float t = std::fabs(A[(size_t)j * m + k]) / norm_upd[j];
            t = (1.f + t) * (1.f - t);
            t = t < 0.f ? 0.f : t;
            const float ratio = norm_upd[j] / norm_dir[j];
            const float t2 = t * ratio * ratio;
            if (t2 <= downdate_threshold) {
                float sq = 0.f;
                for (int r = k + 1; r < m; ++r) sq += A[(size_t)j * m + r] * A[(size_t)j * m + r];
                norm_dir[j] = std::sqrt(sq);
                norm_upd[j] = norm_dir[j];
            } else {
                norm_upd[j] *= std::sqrt(t);
            }
        }
    }
    for (int k = size; k < nc; ++k) d.tau[k] = 0.f;
    d.nonzero = nonzero;
    for (int k = 0; k < nc; ++k) d.perm[k] = k;
    for (int k = 0; k < size; ++k) std::swap(d.perm[k], d.perm[transp[k]]);
}

}  // namespace

// ------------------------------------------------------------ PatchTables --
struct PatchTables {
    int nclusters = 0;
    bool sequential = false;  // a boundary of one cluster overlaps another's interior
    bool pinv = false;        // UPSP_PATCH_PINV=1: the centred double pseudo-inverse instead of the reference's float QR
    ClusterDesc *d_desc = nullptr;
    int32_t *d_bidx = nullptr, *d_iidx = nullptr;
    float *d_P = nullptr;
    QrDesc *d_qr = nullptr;
    float *d_V = nullptr;
    int max_nb = 0;           // longest boundary; beyond kPatchLdsRows the right-hand sides live in d_scratch
    mutable float *d_scratch = nullptr;
    mutable size_t scratch_floats = 0;
};

void patch_tables_free(PatchTables *t)
{
    if (!t) return;
    if (t->d_desc) (void)hipFree(t->d_desc);
    if (t->d_bidx) (void)hipFree(t->d_bidx);
    if (t->d_iidx) (void)hipFree(t->d_iidx);
    if (t->d_P) (void)hipFree(t->d_P);
    if (t->d_qr) (void)hipFree(t->d_qr);
    if (t->d_V) (void)hipFree(t->d_V);
    if (t->d_scratch) (void)hipFree(t->d_scratch);
    delete t;
}

int patch_tables_create(int rows, int cols, int nclusters, const int32_t *b_off, const int32_t *bx,
                        const int32_t *by, const int32_t *i_off, const int32_t *ix,
                        const int32_t *iy, PatchTables **out)
{
    if (!out) return fail(UPSP_ERR_INVALID, "out is null");
    *out = nullptr;
    if (nclusters < 0 || !b_off || !i_off) return fail(UPSP_ERR_INVALID, "bad patch tables");
    const int nbt = b_off[nclusters], nit = i_off[nclusters];
    if ((nbt && (!bx || !by)) || (nit && (!ix || !iy))) return fail(UPSP_ERR_INVALID, "bad patch tables");
    const char *pe = getenv("UPSP_PATCH_PINV");
    const bool pinv = pe && *pe && *pe != '0';
    std::vector<ClusterDesc> desc(nclusters);
    std::vector<QrDesc> qr(nclusters);
    std::vector<float> Vall;
    int max_nb = 0;
    std::vector<int32_t> bidx(std::max(nbt, 1)), iidx(std::max(nit, 1));
    std::vector<float> Pall((size_t)std::max(pinv ? nbt : 0, 1) * 10, 0.f);
    std::vector<int> owner((size_t)rows * cols, -1);
    bool sequential = false;
    for (int c = 0; c < nclusters; ++c) {
        ClusterDesc &d = desc[c];
        d.b_off = b_off[c]; d.nb = b_off[c + 1] - b_off[c];
        d.i_off = i_off[c]; d.ni = i_off[c + 1] - i_off[c];
        if (d.nb < 0 || d.ni < 0) return fail(UPSP_ERR_INVALID, "patch offsets not monotone");
        double mx = 0, my = 0, lox = 1e30, hix = -1e30, loy = 1e30, hiy = -1e30;
        for (int j = 0; j < d.nb; ++j) {
            const int x = bx[d.b_off + j], y = by[d.b_off + j];
            if (x < 0 || y < 0 || x >= cols || y >= rows) return fail(UPSP_ERR_INVALID, "patch pixel outside the frame");
            bidx[d.b_off + j] = y * cols + x;
            mx += x; my += y;
            lox = std::min<double>(lox, x); hix = std::max<double>(hix, x);
            loy = std::min<double>(loy, y); hiy = std::max<double>(hiy, y);
        }
        for (int j = 0; j < d.ni; ++j) {
            const int x = ix[d.i_off + j], y = iy[d.i_off + j];
            if (x < 0 || y < 0 || x >= cols || y >= rows) return fail(UPSP_ERR_INVALID, "patch pixel outside the frame");
            iidx[d.i_off + j] = y * cols + x;
        }
        QrDesc &q = qr[c];
        std::memset(&q, 0, sizeof(q));
        q.b_off = d.b_off; q.nb = d.nb; q.i_off = d.i_off; q.ni = d.ni;
        if (d.nb >= 10 && !pinv) {
            // the reference's design matrix on raw pixel coordinates (patches.ipp:185-193), factored in float
            std::vector<float> A((size_t)d.nb * 10);
            for (int j = 0; j < d.nb; ++j) {
                int cnt = 0;
                for (int i = 0; i <= 3; ++i)
                    for (int jj = 0; jj <= 3; ++jj)
                        if (i + jj <= 3)
                            A[(size_t)cnt++ * d.nb + j] = (float)std::pow((double)by[d.b_off + j], i) * (float)std::pow((double)bx[d.b_off + j], jj);
            }
            colpiv_householder_f32(A, d.nb, q);
            q.v_off = (int)Vall.size();
            Vall.insert(Vall.end(), A.begin(), A.end());
            max_nb = std::max(max_nb, d.nb);
            d.cx = d.cy = 0; d.sx = d.sy = 1;
        } else if (d.nb >= 10) {
            mx /= d.nb; my /= d.nb;
            const double hx = std::max(0.5 * (hix - lox), 1.0), hy = std::max(0.5 * (hiy - loy), 1.0);
            d.cx = (float)mx; d.cy = (float)my;
            d.sx = (float)(1.0 / hx); d.sy = (float)(1.0 / hy);
            std::vector<double> A((size_t)d.nb * 10);
            for (int j = 0; j < d.nb; ++j) {
                // same float coordinate transform the kernel applies to interior pixels
                const float x = ((float)bx[d.b_off + j] - d.cx) * d.sx, y = ((float)by[d.b_off + j] - d.cy) * d.sy;
                const double xd = x, yd = y;
                const double mono[10] = {1, xd, xd * xd, xd * xd * xd, yd, yd * xd, yd * xd * xd,
                                         yd * yd, yd * yd * xd, yd * yd * yd};
                for (int k = 0; k < 10; ++k) A[(size_t)k * d.nb + j] = mono[k];
            }
            std::vector<float> P;
            pinv_design(A, d.nb, P);
            std::memcpy(&Pall[(size_t)d.b_off * 10], P.data(), sizeof(float) * P.size());
        } else {
            d.cx = d.cy = 0; d.sx = d.sy = 1;
        }
    }
    // dependency check: does any cluster read a pixel another cluster writes?
    for (int c = 0; c < nclusters; ++c)
        if (desc[c].nb >= 10)
            for (int j = 0; j < desc[c].ni; ++j) owner[iidx[desc[c].i_off + j]] = c;
    for (int c = 0; c < nclusters && !sequential; ++c)
        for (int j = 0; j < desc[c].nb; ++j) {
            const int o = owner[bidx[desc[c].b_off + j]];
            if (o >= 0 && o != c) { sequential = true; break; }
        }
    PatchTables *t = new PatchTables();
    t->nclusters = nclusters;
    t->sequential = sequential;
    t->pinv = pinv;
    t->max_nb = max_nb;
    hipError_t e = hipMalloc(&t->d_desc, sizeof(ClusterDesc) * std::max(nclusters, 1));
    if (e == hipSuccess) e = hipMalloc(&t->d_qr, sizeof(QrDesc) * std::max(nclusters, 1));
    if (e == hipSuccess) e = hipMalloc(&t->d_V, sizeof(float) * std::max<size_t>(Vall.size(), 1));
    if (e == hipSuccess && nclusters) e = hipMemcpy(t->d_qr, qr.data(), sizeof(QrDesc) * nclusters, hipMemcpyHostToDevice);
    if (e == hipSuccess && !Vall.empty()) e = hipMemcpy(t->d_V, Vall.data(), sizeof(float) * Vall.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&t->d_bidx, sizeof(int32_t) * bidx.size());
    if (e == hipSuccess) e = hipMalloc(&t->d_iidx, sizeof(int32_t) * iidx.size());
    if (e == hipSuccess) e = hipMalloc(&t->d_P, sizeof(float) * Pall.size());
    if (e == hipSuccess && nclusters)
        e = hipMemcpy(t->d_desc, desc.data(), sizeof(ClusterDesc) * nclusters, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(t->d_bidx, bidx.data(), sizeof(int32_t) * bidx.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(t->d_iidx, iidx.data(), sizeof(int32_t) * iidx.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(t->d_P, Pall.data(), sizeof(float) * Pall.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        patch_tables_free(t);
        return fail(UPSP_ERR_HIP, std::string("patch tables: ") + hipGetErrorString(e));
    }
    *out = t;
    return UPSP_OK;
}

static int launch_patch(const PatchTables *t, float *imgs, int nimg, int rows, int cols, hipStream_t st)
{
    if (!t || t->nclusters == 0 || nimg == 0) return UPSP_OK;
    const size_t npix = (size_t)rows * cols;
    if (!t->pinv) {
        // lane = frame: a wave per (cluster, 64 frames)
        const unsigned groups = (unsigned)((nimg + 63) / 64);
        const size_t lds = sizeof(float) * 64 * (size_t)std::min(std::max(t->max_nb, 1), kPatchLdsRows);
        if (t->max_nb > kPatchLdsRows) {
            const size_t want = (size_t)t->nclusters * groups * (size_t)t->max_nb * 64;
            if (want > t->scratch_floats) {
                UPSP_HIP_CHECK(hipStreamSynchronize(st));
                if (t->d_scratch) (void)hipFree(t->d_scratch);
                t->d_scratch = nullptr;
                t->scratch_floats = 0;
                UPSP_HIP_CHECK(hipMalloc(&t->d_scratch, sizeof(float) * want));
                t->scratch_floats = want;
            }
        }
        if (!t->sequential) {
            hipLaunchKernelGGL(patch_qr_kernel, dim3(t->nclusters, groups), dim3(64), lds, st, imgs, npix, nimg, cols, t->d_qr, 0,
                               t->d_bidx, t->d_V, t->d_iidx, t->d_scratch, t->max_nb);
        } else {
            for (int c = 0; c < t->nclusters; ++c)  // cluster order of the reference (patches.ipp:101)
                hipLaunchKernelGGL(patch_qr_kernel, dim3(1, groups), dim3(64), lds, st, imgs, npix, nimg, cols, t->d_qr, c,
                                   t->d_bidx, t->d_V, t->d_iidx, t->d_scratch, t->max_nb);
        }
        UPSP_HIP_CHECK(hipGetLastError());
        return UPSP_OK;
    }
    if (!t->sequential) {
        hipLaunchKernelGGL(patch_kernel, dim3(t->nclusters, nimg), dim3(256), 0, st, imgs, npix, cols,
                           t->d_desc, 0, t->d_bidx, t->d_P, t->d_iidx);
    } else {
        for (int c = 0; c < t->nclusters; ++c)  // cluster order of the reference (patches.ipp:101)
            hipLaunchKernelGGL(patch_kernel, dim3(1, nimg), dim3(256), 0, st, imgs, npix, cols,
                               t->d_desc, c, t->d_bidx, t->d_P, t->d_iidx);
    }
    UPSP_HIP_CHECK(hipGetLastError());
    return UPSP_OK;
}

// ----------------------------------------------------------- FrameScratch --
void frame_scratch_ecc_stats(const FrameScratch *s, unsigned long long *frame_iters, unsigned long long *frames)
{
    *frame_iters = s ? s->ecc_frame_iters : 0;
    *frames = s ? s->ecc_frames : 0;
}

void frame_scratch_free(FrameScratch *s)
{
    if (!s) return;
    for (int c = 0; c < kMaxCams; ++c) {
        if (s->warp[c]) (void)hipFree(s->warp[c]);
        if (s->f32[c]) (void)hipFree(s->f32[c]);
        if (s->f32b[c]) (void)hipFree(s->f32b[c]);
        if (s->tmpl[c]) (void)hipFree(s->tmpl[c]);
    }
    if (s->ecc_img) (void)hipFree(s->ecc_img);
    if (s->ecc_img2) (void)hipFree(s->ecc_img2);
    if (s->center) (void)hipFree(s->center);
    if (s->tsum) (void)hipFree(s->tsum);
    if (s->rtab) (void)hipFree(s->rtab);
    if (s->h_counter) (void)hipHostFree(s->h_counter);
    if (s->ev_counter) (void)hipEventDestroy(s->ev_counter);
    if (s->again_stream) {
        (void)hipStreamSynchronize(s->again_stream);
        (void)hipStreamDestroy(s->again_stream);
    }
    if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
    for (int k = 0; k < 2; ++k)
        if (s->ev_again[k]) (void)hipEventDestroy(s->ev_again[k]);
    if (s->tmp) (void)hipFree(s->tmp);
    if (s->partial) (void)hipFree(s->partial);
    for (int k = 0; k < 2; ++k)
        if (s->partial_id[k]) (void)hipFree(s->partial_id[k]);
    if (s->state) (void)hipFree(s->state);
    if (s->counter) (void)hipFree(s->counter);
    delete s;
}

void frame_scratch_new_reference(FrameScratch *s, int cam)
{
    if (s && cam >= 0 && cam < kMaxCams) s->tmpl_src[cam] = nullptr;
}

int frame_scratch_ensure(FrameScratch **ps, int ncams, int batch, int rows, int cols, bool need_warp,
                         bool need_f32)
{
    FrameScratch *s = *ps;
    if (s && (s->ncams != ncams || s->batch < batch || s->rows != rows || s->cols != cols)) {
        frame_scratch_free(s);
        s = nullptr;
    }
    if (!s) {
        s = new FrameScratch();
        s->ncams = ncams; s->batch = batch; s->rows = rows; s->cols = cols;
        *ps = s;
    }
    const size_t n = (size_t)batch * rows * cols;
    for (int c = 0; c < ncams; ++c) {
        if (need_warp && !s->warp[c]) UPSP_HIP_CHECK(hipMalloc(&s->warp[c], n * sizeof(uint16_t)));
        if (need_f32 && !s->f32[c]) UPSP_HIP_CHECK(hipMalloc(&s->f32[c], n * sizeof(float)));
        if (need_warp && !s->tmpl[c]) UPSP_HIP_CHECK(hipMalloc(&s->tmpl[c], (size_t)rows * cols * sizeof(float)));
    }
    if (need_warp && !s->ecc_img) UPSP_HIP_CHECK(hipMalloc(&s->ecc_img, n * sizeof(float)));
    if (need_warp && !s->h_counter) {
        UPSP_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&s->h_counter), 4 * sizeof(int), hipHostMallocDefault));
        UPSP_HIP_CHECK(hipEventCreateWithFlags(&s->ev_counter, hipEventDisableTiming));
    }
    if (need_warp && !s->rtab) UPSP_HIP_CHECK(hipMalloc(&s->rtab, sizeof(int2) * (size_t)batch * rows));
    if (need_warp && !s->center) {
        UPSP_HIP_CHECK(hipMalloc(&s->center, kMaxCams * sizeof(float)));
        UPSP_HIP_CHECK(hipMemset(s->center, 0, kMaxCams * sizeof(float)));
        UPSP_HIP_CHECK(hipMalloc(&s->tsum, (kMaxCams * 2 + 128) * sizeof(double)));      // (+ the partial sums of ecc_tmpl_sums_kernel)
        UPSP_HIP_CHECK(hipMemset(s->tsum, 0, (kMaxCams * 2 + 128) * sizeof(double)));
    }
    if (!s->tmp) UPSP_HIP_CHECK(hipMalloc(&s->tmp, n * sizeof(double)));
    if (need_warp && !s->partial)
        UPSP_HIP_CHECK(hipMalloc(&s->partial, sizeof(double) * (size_t)batch * kEccStride * kEccSums));
    if (!s->state) UPSP_HIP_CHECK(hipMalloc(&s->state, sizeof(EccState) * (size_t)batch));
    if (!s->counter) UPSP_HIP_CHECK(hipMalloc(&s->counter, 4 * sizeof(int)));
    return UPSP_OK;
}

// ECC registration of nb frames against the blurred template; leaves the warp in state[].
__global__ void __launch_bounds__(256)
    pixel_mask_kernel(const int32_t *__restrict__ pix, unsigned nnodes, unsigned npix, uint8_t *__restrict__ mask)
{
    const unsigned n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nnodes) return;
    const int32_t p = pix[n];
    if (p >= 0 && (unsigned)p < npix) mask[p] = 1;     // (same value from every writer)
}

// list of the set bytes of the mask: one atomic per workgroup of 1024 pixels (order of the workgroups' ranges is free)
__global__ void __launch_bounds__(256)
    pixel_list_kernel(const uint8_t *__restrict__ mask, unsigned npix, unsigned *__restrict__ list)
{
    __shared__ unsigned wave_cnt[4][4], block_base;
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned base = blockIdx.x * 1024u + threadIdx.x;
    unsigned long long m[4];
    bool set[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned i = base + 256u * k;
        set[k] = i < npix && mask[i] != 0;
        m[k] = __ballot(set[k]);
        if (lane == 0) wave_cnt[k][wave] = (unsigned)__popcll(m[k]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned tot = 0;
        for (int k = 0; k < 4; ++k)
            for (int w = 0; w < 4; ++w) {
                const unsigned c = wave_cnt[k][w];
                wave_cnt[k][w] = tot;
                tot += c;
            }
        block_base = tot ? atomicAdd(&list[0], tot) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (set[k])
            list[1 + block_base + wave_cnt[k][wave] + (unsigned)__popcll(m[k] & ((1ull << lane) - 1ull))] = base + 256u * k;
}

int launch_amap_pixels(const uint8_t *d_flag, const unsigned *d_tile_off, size_t npix, unsigned *d_pix_of_k, hipStream_t st)
{
    hipLaunchKernelGGL(amap_pixels_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, d_flag, d_tile_off, npix,
                       d_pix_of_k);
    UPSP_HIP_CHECK(hipGetLastError());
    return UPSP_OK;
}

int launch_pixel_list(const int32_t *d_pix, size_t nnodes, uint8_t *d_mask, unsigned *d_list, size_t npix, hipStream_t st)
{
    UPSP_HIP_CHECK(hipMemsetAsync(d_mask, 0, npix, st));
    UPSP_HIP_CHECK(hipMemsetAsync(d_list, 0, sizeof(unsigned), st));
    if (nnodes)
        hipLaunchKernelGGL(pixel_mask_kernel, dim3((unsigned)((nnodes + 255) / 256)), dim3(256), 0, st, d_pix,
                           (unsigned)nnodes, (unsigned)npix, d_mask);
    hipLaunchKernelGGL(pixel_list_kernel, dim3((unsigned)((npix + 1023) / 1024)), dim3(256), 0, st,
                       (const uint8_t *)d_mask, (unsigned)npix, d_list);
    UPSP_HIP_CHECK(hipGetLastError());
    return UPSP_OK;
}

// GaussianBlur 5 x 5 of nb frames (the ECC's pre-blur, cpp/lib/registration.cpp:57-60) into one of the scratch's two
// blurred-frame buffers: the streamed registration loop enqueues it for sub-batch k + 1 while the host waits for sub-batch
// k's "frames still iterating".  *out = the buffer to hand to run_frame_stages.
int frame_scratch_template(FrameScratch *s, int cam, const float *d_ref, int rows, int cols, hipStream_t st)
{
    if (!s || cam < 0 || cam >= kMaxCams || !s->tmpl[cam]) return fail(UPSP_ERR_INVALID, "ECC template: scratch not set up");
    if (s->tmpl_src[cam] == d_ref) return UPSP_OK;      // blurred template, once per reference image
    int rc = launch_gauss<float>(d_ref, s->tmpl[cam], s->tmp, 1, rows, cols, 5, st);
    if (rc != UPSP_OK) return rc;
    rc = launch_ecc_center(s->tmpl[cam], rows, cols, s->center + cam, st);
    if (rc != UPSP_OK) return rc;
    rc = launch_ecc_tmpl_sums(s->tmpl[cam], rows, cols, s->tsum + 2 * kMaxCams, s->tsum + 2 * cam, st);
    if (rc != UPSP_OK) return rc;
    s->tmpl_src[cam] = d_ref;
    return UPSP_OK;
}

int frame_scratch_preblur(FrameScratch *s, int slot, uint16_t *d_frames, int nb, int rows, int cols, hipStream_t st,
                          const float **out, const HotRepair *hot, int fuse_cam)
{
    if (!s || !s->ecc_img || nb > s->batch) return fail(UPSP_ERR_INVALID, "pre-blur: scratch not set up");
    if (slot && !s->ecc_img2) UPSP_HIP_CHECK(hipMalloc(&s->ecc_img2, (size_t)s->batch * rows * cols * sizeof(float)));
    float *dst = slot ? s->ecc_img2 : s->ecc_img;
    FilterCoef fc;
    if (gaussian_coef(5, fc) != 0) return fail(UPSP_ERR_INVALID, "gaussian 5");
    int rc = UPSP_OK;
    s->ident_for[slot & 1] = nullptr;
    // The blur and the identity iteration of the ECC in one pass (ecc_blur_ident_kernel): the template of camera `fuse_cam` is ready
    // (frame_scratch_template).  The frames the repair changes afterwards -- rare -- take the pass again from their repaired pixels.
    if (fuse_cam >= 0 && fuse_cam < kMaxCams && s->tmpl_src[fuse_cam] && ecc_fused_blur_eligible(rows, cols) &&
        (!hot || hot->max_hot < kHotPositions)) {
        {
            KTimed kt("ecc_blur_ident_kernel", st);
            rc = launch_ecc_blur_ident(s, slot, d_frames, dst, s->tmpl[fuse_cam], s->center + fuse_cam, s->tsum + 2 * fuse_cam, nb, rows, cols, fc.k[2], fc.k[3],
                                       fc.k[4], hot ? (unsigned)hot->thresh : 0u, hot ? hot->d_count : nullptr, hot ? hot->d_pos : nullptr,
                                       nullptr, nullptr, 0, st);
        }
        if (rc != UPSP_OK) return rc;
        if (hot) {
            // The repair (one lane per frame) and the second pass behind it (a handful of workgroups) are ~60 us of latency with the
            // device nearly empty: they run on a stream of their own (FrameScratch::again_stream), forked here and joined in front
            // of this sub-batch's first solve (run_ecc) -- beside whatever the loop's stream does in between (the previous
            // sub-batch's warp and row pass).  UPSP_ECC_AGAIN_STREAM=0: in line (A/B).
            static const bool side = [] { const char *e = std::getenv("UPSP_ECC_AGAIN_STREAM"); return !(e && *e == '0'); }();
            hipStream_t as = st;
            if (side) {
                if (!s->again_stream) {
                    UPSP_HIP_CHECK(hipStreamCreateWithFlags(&s->again_stream, hipStreamNonBlocking));
                    UPSP_HIP_CHECK(hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
                    for (int k = 0; k < 2; ++k) UPSP_HIP_CHECK(hipEventCreateWithFlags(&s->ev_again[k], hipEventDisableTiming));
                }
                as = s->again_stream;
                UPSP_HIP_CHECK(hipEventRecord(s->ev_fork, st));
                UPSP_HIP_CHECK(hipStreamWaitEvent(as, s->ev_fork, 0));
            }
            rc = launch_hot_repair_list(d_frames, (size_t)rows * cols, nb, rows, cols, hot->min_change, hot->max_hot, hot->d_count,
                                        hot->d_pos, hot->d_changes, as);
            if (rc != UPSP_OK) return rc;
            if (hot->max_hot > 0) {
                KTimed kt("hot_fixup_kernels", as);
                rc = launch_ecc_blur_ident(s, slot, d_frames, dst, s->tmpl[fuse_cam], s->center + fuse_cam, s->tsum + 2 * fuse_cam, nb, rows, cols,
                                           fc.k[2], fc.k[3], fc.k[4], 0u, nullptr, nullptr, hot->d_changes + 4,
                                           hot->d_changes + 4 + (((size_t)nb + 3) & ~(size_t)3), hot->max_hot, as);
                if (rc != UPSP_OK) return rc;
            }
            if (side) {
                UPSP_HIP_CHECK(hipEventRecord(s->ev_again[slot & 1], as));
                s->again_pending[slot & 1] = true;
            }
        }
        // UPSP_ECC_FUSED_BLUR=2 (test switch): keep the pass's blurred frames, drop its sums -- the identity iteration runs as its own
        // kernel on them, so the results are bit-identical to the unfused path iff the blurred frames are
        const char *e = std::getenv("UPSP_ECC_FUSED_BLUR");
        if (e && *e == '2') s->ident_for[slot & 1] = nullptr;
        *out = dst;
        return UPSP_OK;
    }
    bool blurred = false;
    if (hot && hot->max_hot < kHotPositions) {
        // The scan of fix_hot_pixels rides on the blur's read of the frames (a pass of its own is 2 MiB per frame = 27 us per
        // 64 frames of 1024^2): the blur counts and lists the pixels >= thresh, one lane per frame repairs the rare frames
        // that hold 1 .. max_hot of them exactly like the reference (in place, scan order), and the blurred values such a
        // pixel reaches -- its 5 x 5 neighbourhood -- are recomputed from the repaired frame: the same bits as repair-then-blur
        // (tests/test_imageops_gpu.py::test_registration_sub_batches_look_ahead, test_frames_gpu.py).
        KTimed kt("gauss_pass_kernels", st);
        blurred = launch_gauss5_quad(d_frames, dst, nb, rows, cols, fc, st, (unsigned)hot->thresh, hot->d_count, hot->d_pos);
    }
    if (blurred) {
        rc = launch_hot_repair_list(d_frames, (size_t)rows * cols, nb, rows, cols, hot->min_change, hot->max_hot, hot->d_count,
                                    hot->d_pos, hot->d_changes, st);
        if (rc != UPSP_OK) return rc;
        const unsigned *nch = hot->d_changes + 4;
        const uint4 *list = reinterpret_cast<const uint4 *>(hot->d_changes + 4 + (((size_t)nb + 3) & ~(size_t)3));
        if (hot->max_hot > 0)
            hipLaunchKernelGGL(reblur_changes_kernel, dim3((unsigned)nb), dim3(256), 0, st, (const uint16_t *)d_frames, dst, rows, cols,
                               nch, list, hot->max_hot, fc.k[2], fc.k[3], fc.k[4]);
        UPSP_HIP_CHECK(hipGetLastError());
    } else {
        if (hot) {
            rc = launch_hot_fix(d_frames, nb, rows, cols, hot->thresh, hot->min_change, hot->max_hot, hot->d_count, hot->d_pos, nullptr, st);
            if (rc != UPSP_OK) return rc;
        }
        rc = launch_gauss<uint16_t>(d_frames, dst, s->tmp, nb, rows, cols, 5, st);      // (tmp: images too small for the fused kernels)
    }
    if (rc == UPSP_OK) *out = dst;
    return rc;
}

int run_frame_stages(FrameScratch *s, int cam, const uint16_t *d_frames, int nb, int64_t first_frame,
                     int rows, int cols, const upsp_pipeline_opts &opts, const float *d_ref,
                     const PatchTables *patches, float *d_warps, int32_t *d_iters, int ncams,
                     const unsigned *d_read_list, const WarpCompact *wc, const void **img_out, int *is_f32_out,
                     hipStream_t st, const float *preblurred, const std::function<int()> *while_waiting)
{
    const size_t npix = (size_t)rows * cols;
    const uint16_t *cur = d_frames;
    const dim3 pgrid(grid_for_pixels(npix), (unsigned)nb), block(256);
    if (opts.registration) {
        int rc = frame_scratch_template(s, cam, d_ref, rows, cols, st);
        if (rc != UPSP_OK) return rc;
        const float *blurred = preblurred;
        if (!blurred) {      // GaussianBlur 5 x 5 of the input frames (findTransformECC's gaussFiltSize)
            s->ident_for[0] = nullptr;      // (this buffer's identity sums, if any, belong to other frames)
            FilterCoef fc;
            if (ecc_fused_blur_eligible(rows, cols) && gaussian_coef(5, fc) == 0) {
                // ... with the identity iteration's sums (the frames are repaired already): the same kernel as the streamed loop's
                // pre-blur, so a frame's warp is the same bits whichever path registers it
                KTimed kt("ecc_blur_ident_kernel", st);
                rc = launch_ecc_blur_ident(s, 0, d_frames, s->ecc_img, s->tmpl[cam], s->center + cam, s->tsum + 2 * cam, nb, rows, cols, fc.k[2], fc.k[3], fc.k[4],
                                           0u, nullptr, nullptr, nullptr, nullptr, 0, st);
                const char *e = std::getenv("UPSP_ECC_FUSED_BLUR");
                if (e && *e == '2') s->ident_for[0] = nullptr;
            } else {
                rc = launch_gauss<uint16_t>(d_frames, s->ecc_img, s->tmp, nb, rows, cols, 5, st);
            }
            if (rc != UPSP_OK) return rc;
            blurred = s->ecc_img;
        }
        rc = run_ecc(s, s->tmpl[cam], s->center + cam, blurred, nb, first_frame, rows, cols, opts.ecc_max_iters, opts.ecc_eps, st,
                         while_waiting);
        if (rc != UPSP_OK) return rc;
        if (wc) {      // registration is the last image stage and node-major series are wanted: straight into the compact buffer
            KTimed kt("warp_u16_kernel", st);
            // one launch for the sub-batch: grid.y = its 64-frame pieces (four launches of 24 us each until round 5)
            hipLaunchKernelGGL(warp_compact_kernel, dim3((unsigned)((wc->max_active + 63) / 64), (unsigned)((nb + 63) / 64)), block, 0, st,
                               d_frames, rows, cols, (const EccState *)s->state, nb, opts.interp, wc->pix_of_k, wc->nact, wc->compact,
                               wc->cpitch, wc->col0);
        } else {
            KTimed kt("warp_u16_kernel", st);
            const bool listed = !opts.patch && !opts.filter && d_read_list;
            // (the list is short -- its length is only known on the device: 64 workgroups per frame stride over it)
            const dim3 wgrid(listed ? 64u : pgrid.x, (unsigned)nb);
            hipLaunchKernelGGL(warp_u16_kernel, wgrid, block, 0, st, d_frames, s->warp[cam], rows, cols,
                               (const EccState *)s->state, opts.interp, listed ? d_read_list : (const unsigned *)nullptr);
        }
        if (d_warps || d_iters) {
            rc = launch_ecc_export(s->state, nb, d_warps ? d_warps + (size_t)cam * 6 : (float *)nullptr, ncams * 6,
                                   d_iters ? d_iters + cam : (int32_t *)nullptr, ncams, st);
            if (rc != UPSP_OK) return rc;
        }
        cur = s->warp[cam];
    }
    if (opts.filter == 1 && !opts.patch) {
        // Gaussian filter straight from the u16 frames (convertTo + GaussianBlur in one pass)
        float *f = s->f32[cam];
        int rc = launch_gauss<uint16_t>(cur, f, s->tmp, nb, rows, cols, opts.filter_size, st);
        if (rc != UPSP_OK) return rc;
        *img_out = f;
        *is_f32_out = 1;
    } else if (opts.patch || opts.filter) {
        float *f = s->f32[cam];
        hipLaunchKernelGGL(u16_to_f32_kernel, dim3(2048), dim3(256), 0, st, cur, f, npix * (size_t)nb);
        if (opts.patch) {
            int rc = launch_patch(patches, f, nb, rows, cols, st);
            if (rc != UPSP_OK) return rc;
        }
        if (opts.filter == 1) {
            // out of place into a second buffer: the fused kernel cannot run in place
            if (!s->f32b[cam])
                UPSP_HIP_CHECK(hipMalloc(&s->f32b[cam], (size_t)s->batch * npix * sizeof(float)));
            int rc = launch_gauss<float>(f, s->f32b[cam], s->tmp, nb, rows, cols, opts.filter_size, st);
            if (rc != UPSP_OK) return rc;
            f = s->f32b[cam];
        } else if (opts.filter == 2) {
            hipLaunchKernelGGL((box_pass_kernel<true>), pgrid, block, 0, st, (const void *)f,
                               (void *)s->tmp, rows, cols, opts.filter_size);
            hipLaunchKernelGGL((box_pass_kernel<false>), pgrid, block, 0, st, (const void *)s->tmp,
                               (void *)f, rows, cols, opts.filter_size);
        }
        *img_out = f;
        *is_f32_out = 1;
    } else {
        *img_out = cur;
        *is_f32_out = 0;
    }
    UPSP_HIP_CHECK(hipGetLastError());
    return UPSP_OK;
}

}  // namespace upsp

using namespace upsp;

extern "C" {

int upsp_register_pixel_u16(const float *d_ref32f, const uint16_t *d_inp, int rows, int cols,
                            int max_iters, double eps, int interp, uint16_t *d_out, float *h_warp6,
                            void *stream)
{
    if (!d_ref32f || !d_inp || !d_out || rows <= 0 || cols <= 0 || max_iters < 1)
        return fail(UPSP_ERR_INVALID, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    FrameScratch *s = nullptr;
    int rc = frame_scratch_ensure(&s, 1, 1, rows, cols, true, false);
    int iters = 0;
    if (rc == UPSP_OK) rc = launch_gauss<float>(d_ref32f, s->tmpl[0], s->tmp, 1, rows, cols, 5, st);
    if (rc == UPSP_OK) rc = launch_ecc_center(s->tmpl[0], rows, cols, s->center, st);
    if (rc == UPSP_OK) rc = launch_ecc_tmpl_sums(s->tmpl[0], rows, cols, s->tsum + 2 * kMaxCams, s->tsum, st);
    FilterCoef fc;
    if (rc == UPSP_OK && ecc_fused_blur_eligible(rows, cols) && gaussian_coef(5, fc) == 0) {      // (the frame loop's kernel: same bits)
        rc = launch_ecc_blur_ident(s, 0, d_inp, s->ecc_img, s->tmpl[0], s->center, s->tsum, 1, rows, cols, fc.k[2], fc.k[3], fc.k[4], 0u, nullptr,
                                   nullptr, nullptr, nullptr, 0, st);
        const char *e = std::getenv("UPSP_ECC_FUSED_BLUR");
        if (e && *e == '2') s->ident_for[0] = nullptr;
    } else if (rc == UPSP_OK) {
        rc = launch_gauss<uint16_t>(d_inp, s->ecc_img, s->tmp, 1, rows, cols, 5, st);
    }
    // first_frame = 1: a stand-alone call always registers (psp_process.cpp:1662-1679)
    if (rc == UPSP_OK) rc = run_ecc(s, s->tmpl[0], s->center, s->ecc_img, 1, 1, rows, cols, max_iters, eps, st);
    if (rc == UPSP_OK) {
        hipLaunchKernelGGL(warp_u16_kernel, dim3(grid_for_pixels((size_t)rows * cols), 1), dim3(256), 0,
                           st, d_inp, d_out, rows, cols, (const EccState *)s->state, interp, (const unsigned *)nullptr);
        EccState h;
        hipError_t e = hipMemcpyAsync(&h, s->state, sizeof(h), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            rc = fail(UPSP_ERR_HIP, hipGetErrorString(e));
        } else {
            if (h_warp6) std::memcpy(h_warp6, h.M, sizeof(h.M));
            iters = h.iters;
        }
    }
    frame_scratch_free(s);
    return rc == UPSP_OK ? iters : rc;
}

int upsp_blur_u16(const uint16_t *d_src, float *d_dst, int nimg, int rows, int cols, int k, void *stream)
{
    if (!d_src || !d_dst || nimg <= 0 || rows <= 0 || cols <= 0) return fail(UPSP_ERR_INVALID, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    float *tmp = nullptr;
    UPSP_HIP_CHECK(hipMalloc(&tmp, sizeof(float) * (size_t)nimg * rows * cols));
    int rc = launch_gauss<uint16_t>(d_src, d_dst, tmp, nimg, rows, cols, k, st);
    hipError_t e = hipStreamSynchronize(st);
    (void)hipFree(tmp);
    if (rc == UPSP_OK && e != hipSuccess) rc = fail(UPSP_ERR_HIP, hipGetErrorString(e));
    return rc;
}

int upsp_blur_f32(const float *d_src, float *d_dst, int rows, int cols, int k, int box, void *stream)
{
    if (!d_src || !d_dst || rows <= 0 || cols <= 0) return fail(UPSP_ERR_INVALID, "bad argument");
    if (k < 1 || (k & 1) == 0 || k > kMaxKernel) return fail(UPSP_ERR_INVALID, "filter size must be odd and <= 63");
    hipStream_t st = (hipStream_t)stream;
    const size_t npix = (size_t)rows * cols;
    void *tmp = nullptr;
    UPSP_HIP_CHECK(hipMalloc(&tmp, npix * sizeof(double)));
    int rc = UPSP_OK;
    if (!box) {
        rc = launch_gauss<float>(d_src, d_dst, (float *)tmp, 1, rows, cols, k, st);
    } else {
        const dim3 grid(grid_for_pixels(npix), 1), block(256);
        hipLaunchKernelGGL((box_pass_kernel<true>), grid, block, 0, st, (const void *)d_src, tmp, rows, cols, k);
        hipLaunchKernelGGL((box_pass_kernel<false>), grid, block, 0, st, (const void *)tmp, (void *)d_dst, rows, cols, k);
    }
    hipError_t e = hipStreamSynchronize(st);
    (void)hipFree(tmp);
    if (rc == UPSP_OK && e != hipSuccess) rc = fail(UPSP_ERR_HIP, hipGetErrorString(e));
    return rc;
}

int upsp_patch_f32(float *d_img, int rows, int cols, int nclusters, const int32_t *h_b_off,
                   const int32_t *h_bx, const int32_t *h_by, const int32_t *h_i_off,
                   const int32_t *h_ix, const int32_t *h_iy, void *stream)
{
    return upsp_patch_frames_f32(d_img, 1, rows, cols, nclusters, h_b_off, h_bx, h_by, h_i_off, h_ix, h_iy, stream);
}

int upsp_patch_frames_f32(float *d_img, int nimg, int rows, int cols, int nclusters, const int32_t *h_b_off,
                          const int32_t *h_bx, const int32_t *h_by, const int32_t *h_i_off,
                          const int32_t *h_ix, const int32_t *h_iy, void *stream)
{
    if (!d_img || nimg < 0 || rows <= 0 || cols <= 0) return fail(UPSP_ERR_INVALID, "bad argument");
    if (nclusters == 0 || nimg == 0) return UPSP_OK;
    PatchTables *t = nullptr;
    int rc = patch_tables_create(rows, cols, nclusters, h_b_off, h_bx, h_by, h_i_off, h_ix, h_iy, &t);
    if (rc != UPSP_OK) return rc;
    rc = launch_patch(t, d_img, nimg, rows, cols, (hipStream_t)stream);
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    patch_tables_free(t);
    if (rc == UPSP_OK && e != hipSuccess) rc = fail(UPSP_ERR_HIP, hipGetErrorString(e));
    return rc;
}

}  // extern "C"
