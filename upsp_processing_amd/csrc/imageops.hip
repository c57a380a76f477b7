// Image stages of the psp_process frame loop on MI355X (gfx950):
//   register (ECC affine + inverse-map warp)   cpp/lib/registration.cpp:32-81
//   patch    (cubic 2-D polynomial over fiducials) cpp/lib/patches.ipp:98-236
//   filter   (GaussianBlur / blur)               cpp/exec/psp_process.cpp:1802-1807
//
// The arithmetic of these stages lives in OpenCV 4.5.2 / Eigen 3.3.9 in the
// reference (un-vendored, no reference test): the kernels follow the published
// algorithms as restated in oracle/image_oracle.c.
//
// ECC on the GPU.  All frames of a sub-batch iterate in lock step.  One iteration
// is ONE pass over the template-sized pixel grid per frame: the warped image, the
// two warped gradients (central differences recomputed from the blurred frame on
// the fly, never stored) and the nearest-neighbour mask are evaluated per pixel and
// folded into 45 double sums; everything OpenCV derives from zero-mean images
// (correlation, Hessian, projections, lambda, the parameter step) follows from
// those sums algebraically, so no second pass and no Jacobian planes exist.  The
// sums are reduced deterministically (fixed block partials, fixed order) and a
// one-lane-per-frame kernel does the 6x6 float LU solve exactly like cv::Mat::inv.
#include <hip/hip_runtime.h>
#include <functional>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ktimer.h"
#include "pipeline.h"
#include "upsp_internal.h"

namespace upsp {
namespace {

constexpr int kEccBlocks = 64;   // partial-sum blocks per frame (full sub-batch)
constexpr int kEccBlocksMax = 512;  // ... when only a few frames are still iterating
constexpr int kEccBorderBlocks = 48; // partial-sum blocks of the band (slots before the interior blocks'); ecc_band_cols_body needs >= 3 x ceil(cols / 256)
constexpr int kEccStride = kEccBlocksMax + kEccBorderBlocks;   // partial sums per (frame, sum)
constexpr int kEccSums = 45;
constexpr int kMaxKernel = 63;   // largest odd filter size

// ------------------------------------------------------------------ utils --
__host__ __device__ inline int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        if (i >= n) i = 2 * n - 2 - i;
    }
    return i;
}

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

int env_int_io(const char *name, int dflt)
{
    const char *v = std::getenv(name);
    return v ? std::atoi(v) : dflt;
}

unsigned grid_for_pixels(size_t npix)
{
    size_t g = (npix + 255) / 256;
    return (unsigned)std::min<size_t>(g, 1024);
}

// GaussianBlur(src,dst,Size(k,k),0) for nimg images; tmp: nimg*npix floats
// the four-pixels-per-lane 5 x 5 kernel (defined with the ECC kernels below); false: not applicable, nothing launched
bool launch_gauss5_quad(const uint16_t *src, float *dst, int nimg, int rows, int cols, const FilterCoef &fc, hipStream_t st,
                        unsigned thresh = 0, unsigned *flag = nullptr, int only_flagged = 0);
inline bool launch_gauss5_quad(const float *, float *, int, int, int, const FilterCoef &, hipStream_t) { return false; }
bool gauss5_quad_applies(const uint16_t *src, const float *dst, int nimg, int rows, int cols);

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
        const int tv = env_int_io("UPSP_GAUSS_TILE", 0);      // (measurement switch: tile shape of the 5 x 5 kernel)
        if (r == 2 && tv) {
#define UPSP_GT(W_, H_)                                                                                       \
    hipLaunchKernelGGL((gauss_fused_kernel<SRC, 2, W_, H_>), dim3((unsigned)((cols + W_ - 1) / W_), (unsigned)((rows + H_ - 1) / H_), \
                                                                  (unsigned)nimg), block, 0, st, src, dst, rows, cols, fc)
            if (tv == 1) UPSP_GT(128, 16); else if (tv == 2) UPSP_GT(128, 32); else if (tv == 3) UPSP_GT(256, 16); else UPSP_GT(256, 8);
#undef UPSP_GT
            UPSP_HIP_CHECK(hipGetLastError());
            return UPSP_OK;
        }
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
struct WarpCoord {
    int sx, sy, ax, ay;
};

// WarpAffineInvoker (OpenCV imgwarp.cpp): AB_BITS 10, INTER_BITS 5, cvRound of doubles
__device__ __forceinline__ WarpCoord warp_coord(const double *M, int x, int y, int interp)
{
    const int AB_SCALE = 1024;
    const int round_delta = interp ? 16 : 512;
    const int adelta = __double2int_rn(M[0] * x * AB_SCALE);
    const int bdelta = __double2int_rn(M[3] * x * AB_SCALE);
    const int X0 = __double2int_rn((M[1] * y + M[2]) * AB_SCALE) + round_delta;
    const int Y0 = __double2int_rn((M[4] * y + M[5]) * AB_SCALE) + round_delta;
    WarpCoord c;
    if (interp) {
        const int X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
        c.sx = X >> 5; c.sy = Y >> 5; c.ax = X & 31; c.ay = Y & 31;
    } else {
        c.sx = (X0 + adelta) >> 10; c.sy = (Y0 + bdelta) >> 10; c.ax = c.ay = 0;
    }
    c.sx = max(-32768, min(32767, c.sx));  // saturate_cast<short>
    c.sy = max(-32768, min(32767, c.sy));
    return c;
}

// remapBilinear<Cast<float,T>,...>, BORDER_CONSTANT 0.  F(y,x) fetches a source pixel.
template <typename F>
__device__ __forceinline__ float bilinear(F fetch, int rows, int cols, WarpCoord c)
{
    const float fx = c.ax * (1.f / 32), fy = c.ay * (1.f / 32);
    const float w0 = (1.f - fy) * (1.f - fx), w1 = (1.f - fy) * fx, w2 = fy * (1.f - fx), w3 = fy * fx;
    const int sx = c.sx, sy = c.sy;
    if ((unsigned)sx < (unsigned)(cols - 1) && (unsigned)sy < (unsigned)(rows - 1))
        return fetch(sy, sx) * w0 + fetch(sy, sx + 1) * w1 + fetch(sy + 1, sx) * w2 + fetch(sy + 1, sx + 1) * w3;
    if (sx >= cols || sx + 1 < 0 || sy >= rows || sy + 1 < 0) return 0.f;
    const bool x0 = sx >= 0 && sx < cols, x1 = sx + 1 >= 0 && sx + 1 < cols;
    const bool y0 = sy >= 0 && sy < rows, y1 = sy + 1 >= 0 && sy + 1 < rows;
    const float v0 = (x0 && y0) ? fetch(sy, sx) : 0.f;
    const float v1 = (x1 && y0) ? fetch(sy, sx + 1) : 0.f;
    const float v2 = (x0 && y1) ? fetch(sy + 1, sx) : 0.f;
    const float v3 = (x1 && y1) ? fetch(sy + 1, sx + 1) : 0.f;
    return v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3;
}

// per-frame ECC state
struct EccState {
    float M[6];
    double rho, last_rho;
    int iters;
    int done;     // 1 converged / iteration cap, 2 identity (frame 0), <0 error
    int band;     // pixels farther than this from every image edge have their whole bilinear footprint (and its
                  // gradient taps) inside the image under M (ecc_band): ecc_interior_kernel takes them, ecc_border_kernel the rest
};

// Width of the border band for the warp M: an affine displacement |M p - p| is largest at a corner of the image; the
// fixed-point source pixel is within 1.02 of M p, the footprint reaches 1 pixel before and 2 behind it.
__device__ __forceinline__ int ecc_band(const float *Mf, int rows, int cols)
{
    double D = 0.0;
    for (int cy = 0; cy < 2; ++cy)
        for (int cx = 0; cx < 2; ++cx) {
            const double x = cx ? cols - 1 : 0, y = cy ? rows - 1 : 0;
            const double dx = fabs((double)Mf[0] * x + (double)Mf[1] * y + (double)Mf[2] - x);
            const double dy = fabs((double)Mf[3] * x + (double)Mf[4] * y + (double)Mf[5] - y);
            D = fmax(D, fmax(dx, dy));
        }
    if (!(D < 1.0e6)) return 1 << 24;      // (also NaN: everything is border)
    return (int)ceil(D) + 3;
}

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

// --------------------------------------------------------------------- ECC --
// Sum slots:
//  0 n   1 Sw   2 Sww   3 St   4 Stt   5 Stw          (masked)
//  6..11  S_all  J_k * w        12..17 S_mask J_k      18..23 S_mask J_k * t
//  24..44 S_all  J_a * J_b  (a <= b, row-major upper triangle)
// IDENT: every active frame still holds the identity warp (first iteration of register_pixel,
// cpp/lib/registration.cpp:52-53): source pixel = target pixel, zero fractions, so the bilinear
// weights are (1,0,0,0) and the general arithmetic reduces EXACTLY to the centre taps
// (x*1 + y*0 + .. = x in float) -- 5 loads and no interpolation instead of 12 loads.
//
// One launch per iteration (ecc_sums2_kernel), the pixels split by WHERE they are:
//   * interior blocks: pixels farther than EccState::band from every edge -- the whole 12-pixel footprint is inside
//     the image and the nearest-neighbour mask is 1 by construction (ecc_band), so there is no test, no list and no
//     fallback in the loop.  The inner rectangle is one index range split evenly over the blocks; a trip loads for
//     several pixels per thread (identity iteration: 2 x 4 consecutive pixels from 16-B loads; general: 4 pixels)
//     before it accumulates any, and lanes past the end of the range add zeros instead of branching;
//   * band blocks: 2 x band rows + 2 x band columns (~1.5 % of a 1024^2 frame), generic bilinear with border
//     handling; first in dispatch order so that their few long trips run beside the interior blocks.
// Same per-pixel arithmetic everywhere, fixed order of the block partials (band slots, then interior slots).
// (Round 2's first form was ONE sweep with a per-wave list of border pixels and an in-place fallback: 146 VGPRs, 3
// waves per SIMD, every trip waiting a memory latency for its own loads -- 460 / 500 us per 64-frame launch for the
// identity / a general iteration where the VALU work is 180 / 350 us; every attempt to put more loads in flight
// INSIDE that loop made the compiler spend 220-256 registers.  Now 215 / 385 us.)
__device__ __forceinline__ void ecc_accumulate(double (&acc)[kEccSums], float w, float gx, float gy, float t, bool m,
                                               int x, int y)
{
    const float X = (float)x, Y = (float)y;
    const float J[6] = {gx * X, gy * X, gx * Y, gy * Y, gx, gy};
    const double mm = m ? 1.0 : 0.0, wd = w, td = t, tm = m ? td : 0.0, wm = m ? wd : 0.0;
    acc[0] += mm;
    acc[1] += wm;
    acc[2] = fma(wm, wd, acc[2]);
    acc[3] += tm;
    acc[4] = fma(tm, td, acc[4]);
    acc[5] = fma(tm, wd, acc[5]);
    double Jd[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) Jd[a] = (double)J[a];
    int h = 24;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        acc[6 + a] = fma(Jd[a], wd, acc[6 + a]);
        acc[12 + a] = fma(Jd[a], mm, acc[12 + a]);
        acc[18 + a] = fma(Jd[a], tm, acc[18 + a]);
#pragma unroll
        for (int b = a; b < 6; ++b) {
            acc[h] = fma(Jd[a], Jd[b], acc[h]);
            ++h;
        }
    }
}

// deterministic block reduction of the 45 sums: wave shuffle tree, then 4 waves through LDS
__device__ __forceinline__ void ecc_block_store(const double (&acc)[kEccSums], double *__restrict__ partial, int f, int slot)
{
    __shared__ double red[4][kEccSums];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < kEccSums; ++k) {
        double v = acc[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < kEccSums) {
        const double v = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
        partial[((size_t)f * kEccSums + threadIdx.x) * kEccStride + slot] = v;
    }
}

struct EccMargins { int top, bottom, left, right; };
__device__ __forceinline__ EccMargins ecc_margins(int band, int rows, int cols)
{
    EccMargins g;
    g.top = min(band, rows / 2);
    g.bottom = min(band, rows - g.top);
    g.left = min(band, cols / 2);
    g.right = min(band, cols - g.left);
    return g;
}

template <bool IDENT, int KP>
__device__ __forceinline__ void ecc_interior_body(const float *__restrict__ img, const float *__restrict__ tmpl, int rows,
                                                  int cols, const EccState *__restrict__ state,
                                                  double *__restrict__ partial, int f, unsigned blk, unsigned nblk)
{
    const EccState &es = state[f];
    const float *I = img + (size_t)f * rows * cols;
    double acc[kEccSums];
#pragma unroll
    for (int k = 0; k < kEccSums; ++k) acc[k] = 0.0;
    const EccMargins g = ecc_margins(IDENT ? 3 : es.band, rows, cols);
    const int x_lo = g.left, x_hi = cols - g.right, y_lo = g.top, y_hi = rows - g.bottom;   // [lo, hi)
    const int W = max(x_hi - x_lo, 0), H = max(y_hi - y_lo, 0);
    // The inner rectangle as ONE index range split evenly over the blocks (whatever the image width: a trip always has
    // all 256 lanes at work except at the end of the block's range); (column, row) carried along, no division per trip.
    if (IDENT) {
        // items = groups of four consecutive pixels of a row (the last group of a row may be short);
        // source pixel = target pixel, zero fractions: w = I, gx / gy = central differences; mask 1
        const unsigned G = ((unsigned)W + 3u) / 4u, total = G * (unsigned)H;
        const unsigned per_block = (total + nblk - 1) / nblk;
        const unsigned j_lo = min(total, blk * per_block), j_hi = min(total, j_lo + per_block);
        unsigned j = j_lo + threadIdx.x;
        unsigned gy = G ? j / G : 0u, gx = G ? j % G : 0u;
        for (unsigned jb = j_lo; jb < j_hi; jb += 256u * KP) {     // (uniform trip count; lanes past the end add zeros)
            float4 c[KP], u[KP], d[KP], t[KP];
            float l[KP], r[KP];
            int xs[KP], ys[KP];
            bool in[KP];
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                in[k] = j < j_hi;
                xs[k] = in[k] ? x_lo + 4 * (int)gx : x_lo;         // (lanes past the end: any address inside the image)
                ys[k] = in[k] ? y_lo + (int)gy : y_lo;
                const float *r1 = I + (size_t)ys[k] * cols + xs[k];
                c[k] = *reinterpret_cast<const float4 *>(r1);      // (4-B aligned 16-B loads)
                u[k] = *reinterpret_cast<const float4 *>(r1 - cols);
                d[k] = *reinterpret_cast<const float4 *>(r1 + cols);
                t[k] = *reinterpret_cast<const float4 *>(tmpl + (size_t)ys[k] * cols + xs[k]);
                l[k] = r1[-1];
                r[k] = r1[4];
                j += 256u;
                gx += 256u;
                while (gx >= G && G) {
                    gx -= G;
                    ++gy;
                }
            }
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                const int x = xs[k], y = ys[k];
                const bool m0 = in[k], m1 = in[k] && x + 1 < x_hi, m2 = in[k] && x + 2 < x_hi, m3 = in[k] && x + 3 < x_hi;
                __builtin_amdgcn_sched_barrier(0);
                ecc_accumulate(acc, m0 ? c[k].x : 0.f, m0 ? -0.5f * l[k] + 0.5f * c[k].y : 0.f,
                               m0 ? -0.5f * u[k].x + 0.5f * d[k].x : 0.f, m0 ? t[k].x : 0.f, m0, x, y);
                __builtin_amdgcn_sched_barrier(0);
                ecc_accumulate(acc, m1 ? c[k].y : 0.f, m1 ? -0.5f * c[k].x + 0.5f * c[k].z : 0.f,
                               m1 ? -0.5f * u[k].y + 0.5f * d[k].y : 0.f, m1 ? t[k].y : 0.f, m1, x + 1, y);
                __builtin_amdgcn_sched_barrier(0);
                ecc_accumulate(acc, m2 ? c[k].z : 0.f, m2 ? -0.5f * c[k].y + 0.5f * c[k].w : 0.f,
                               m2 ? -0.5f * u[k].z + 0.5f * d[k].z : 0.f, m2 ? t[k].z : 0.f, m2, x + 2, y);
                __builtin_amdgcn_sched_barrier(0);
                ecc_accumulate(acc, m3 ? c[k].w : 0.f, m3 ? -0.5f * c[k].z + 0.5f * r[k] : 0.f,
                               m3 ? -0.5f * u[k].w + 0.5f * d[k].w : 0.f, m3 ? t[k].w : 0.f, m3, x + 3, y);
            }
        }
    } else {
        double M[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) M[i] = es.M[i];
        const unsigned total = (unsigned)W * (unsigned)H;
        const unsigned per_block = (total + nblk - 1) / nblk;
        const unsigned q_lo = min(total, blk * per_block), q_hi = min(total, q_lo + per_block);
        // the two terms of the fixed-point coordinate (WarpAffineInvoker's X0 / Y0 per row, adelta / bdelta per column),
        // each rounded on its own, tabulated once per block: same integers as evaluating them per pixel
        constexpr int kTabCols = 2048, kTabRows = 64;
        __shared__ int2 ctab[kTabCols], rtab[kTabRows];
        const unsigned ry_lo = W ? q_lo / (unsigned)W : 0u;
        const unsigned nrows_blk = (q_hi > q_lo && W) ? (q_hi - 1u) / (unsigned)W - ry_lo + 1u : 0u;
        const bool tab = cols <= kTabCols && nrows_blk <= (unsigned)kTabRows;
        if (tab) {
            for (int cx = threadIdx.x; cx < cols; cx += 256)
                ctab[cx] = make_int2(__double2int_rn(M[0] * cx * 1024), __double2int_rn(M[3] * cx * 1024));
            if (threadIdx.x < nrows_blk) {
                const int yy = y_lo + (int)(ry_lo + threadIdx.x);
                rtab[threadIdx.x] = make_int2(__double2int_rn((M[1] * yy + M[2]) * 1024), __double2int_rn((M[4] * yy + M[5]) * 1024));
            }
            __syncthreads();
        }
        unsigned q = q_lo + threadIdx.x;
        unsigned py = W ? q / (unsigned)W : 0u, px = W ? q % (unsigned)W : 0u;     // (inside the inner rectangle)
        for (unsigned qb = q_lo; qb < q_hi; qb += 256u * KP) {    // (uniform trip count; lanes past the end add zeros)
            // KP pixels per thread and trip (q, q + 256, ..): KP x 13 loads in flight
            float v[KP][12], t[KP], fx[KP], fy[KP];
            int xs[KP], ys[KP];
            bool on[KP];
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                on[k] = q < q_hi;
                const unsigned ry = on[k] ? py : ry_lo;
                xs[k] = on[k] ? x_lo + (int)px : x_lo;
                ys[k] = y_lo + (int)ry;
                int2 rt, ct;
                if (tab) {
                    rt = rtab[ry - ry_lo];
                    ct = ctab[xs[k]];
                } else {
                    rt = make_int2(__double2int_rn((M[1] * ys[k] + M[2]) * 1024), __double2int_rn((M[4] * ys[k] + M[5]) * 1024));
                    ct = make_int2(__double2int_rn(M[0] * xs[k] * 1024), __double2int_rn(M[3] * xs[k] * 1024));
                }
                const int Xr = rt.x + ct.x, Yr = rt.y + ct.y;
                const int Xq = (Xr + 16) >> 5, Yq = (Yr + 16) >> 5;
                const int sx = Xq >> 5, sy = Yq >> 5;     // (inside the image by construction: no saturation, no test)
                fx[k] = (Xq & 31) * (1.f / 32);
                fy[k] = (Yq & 31) * (1.f / 32);
                const float *r0 = I + (unsigned)((sy - 1) * cols + sx);
                const float *r1 = r0 + cols, *r2 = r1 + cols, *r3 = r2 + cols;
                v[k][0] = r0[0]; v[k][1] = r0[1];
                v[k][2] = r1[-1]; v[k][3] = r1[0]; v[k][4] = r1[1]; v[k][5] = r1[2];
                v[k][6] = r2[-1]; v[k][7] = r2[0]; v[k][8] = r2[1]; v[k][9] = r2[2];
                v[k][10] = r3[0]; v[k][11] = r3[1];
                t[k] = tmpl[(size_t)ys[k] * cols + xs[k]];
                q += 256u;
                px += 256u;
                while (px >= (unsigned)W && W) {
                    px -= (unsigned)W;
                    ++py;
                }
            }
            float w[KP], gx[KP], gy[KP];
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                // bilinear of I, of [-0.5 0 0.5] along x and along y over the 12-pixel footprint
                const float a0 = v[k][0], a1 = v[k][1];
                const float b_1 = v[k][2], b0 = v[k][3], b1 = v[k][4], b2 = v[k][5];
                const float c_1 = v[k][6], c0 = v[k][7], c1 = v[k][8], c2 = v[k][9];
                const float d0 = v[k][10], d1 = v[k][11];
                const float w0 = (1.f - fy[k]) * (1.f - fx[k]), w1 = (1.f - fy[k]) * fx[k], w2 = fy[k] * (1.f - fx[k]),
                            w3 = fy[k] * fx[k];
                w[k] = b0 * w0 + b1 * w1 + c0 * w2 + c1 * w3;
                gx[k] = (-0.5f * b_1 + 0.5f * b1) * w0 + (-0.5f * b0 + 0.5f * b2) * w1 +
                        (-0.5f * c_1 + 0.5f * c1) * w2 + (-0.5f * c0 + 0.5f * c2) * w3;
                gy[k] = (-0.5f * a0 + 0.5f * c0) * w0 + (-0.5f * a1 + 0.5f * c1) * w1 +
                        (-0.5f * b0 + 0.5f * d0) * w2 + (-0.5f * b1 + 0.5f * d1) * w3;
            }
#pragma unroll
            for (int k = 0; k < KP; ++k)
                ecc_accumulate(acc, on[k] ? w[k] : 0.f, on[k] ? gx[k] : 0.f, on[k] ? gy[k] : 0.f, on[k] ? t[k] : 0.f, on[k],
                               xs[k], ys[k]);
        }
    }
    ecc_block_store(acc, partial, f, (int)(kEccBorderBlocks + blk));
}

// The band: top and bottom strips over the full width, left and right strips between them; generic bilinear
// (border value 0, reflect-101 gradient taps), nearest-neighbour mask evaluated.  Workgroups first_slot .. first_slot +
// kEccBorderBlocks - 1 of the same launch as the interior ones (as a launch of its own the band cost 55 us per 64 frames
// behind 190 / 350 us of interior: a few trips of long dependent chains that nothing overlapped).
__device__ __forceinline__ void ecc_border_body(const float *__restrict__ img, const float *__restrict__ tmpl, int rows,
                                                int cols, const EccState *__restrict__ state, double *__restrict__ partial,
                                                int f, unsigned bidx, int ident)
{
    const EccState &es = state[f];
    const float *I = img + (size_t)f * rows * cols;
    double M[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) M[i] = es.M[i];
    double acc[kEccSums];
#pragma unroll
    for (int k = 0; k < kEccSums; ++k) acc[k] = 0.0;
    auto pix = [&](int yy, int xx) { return I[(size_t)yy * cols + xx]; };
    auto gxf = [&](int yy, int xx) {
        return -0.5f * pix(yy, reflect101(xx - 1, cols)) + 0.5f * pix(yy, reflect101(xx + 1, cols));
    };
    auto gyf = [&](int yy, int xx) {
        return -0.5f * pix(reflect101(yy - 1, rows), xx) + 0.5f * pix(reflect101(yy + 1, rows), xx);
    };
    const EccMargins g = ecc_margins(ident ? 3 : es.band, rows, cols);
    const int H = rows - g.top - g.bottom, side = g.left + g.right;
    // (32-bit indices: run_ecc refuses images of 2^31 pixels)
    const unsigned n_top = (unsigned)g.top * (unsigned)cols, n_bot = (unsigned)g.bottom * (unsigned)cols,
                   n_side = (unsigned)H * (unsigned)side;
    const unsigned total = n_top + n_bot + n_side;
    for (unsigned j = bidx * 256u + threadIdx.x; j < total; j += (unsigned)kEccBorderBlocks * 256u) {
        int x, y;
        if (j < n_top) {
            y = (int)(j / (unsigned)cols);
            x = (int)(j % (unsigned)cols);
        } else if (j < n_top + n_bot) {
            const unsigned q = j - n_top;
            y = rows - g.bottom + (int)(q / (unsigned)cols);
            x = (int)(q % (unsigned)cols);
        } else {
            const unsigned q = j - n_top - n_bot;
            y = g.top + (int)(q / (unsigned)side);
            const int k = (int)(q % (unsigned)side);
            x = k < g.left ? k : cols - side + k;
        }
        const int Xr = __double2int_rn((M[1] * y + M[2]) * 1024) + __double2int_rn(M[0] * x * 1024);
        const int Yr = __double2int_rn((M[4] * y + M[5]) * 1024) + __double2int_rn(M[3] * x * 1024);
        const int Xq = (Xr + 16) >> 5, Yq = (Yr + 16) >> 5;
        WarpCoord c;
        c.sx = max(-32768, min(32767, Xq >> 5));
        c.sy = max(-32768, min(32767, Yq >> 5));
        c.ax = Xq & 31;
        c.ay = Yq & 31;
        const int nx = max(-32768, min(32767, (Xr + 512) >> 10)), ny = max(-32768, min(32767, (Yr + 512) >> 10));
        const bool m = (unsigned)nx < (unsigned)cols && (unsigned)ny < (unsigned)rows;
        float w, gx, gy;
        if (c.sx >= 1 && c.sx + 2 < cols && c.sy >= 1 && c.sy + 2 < rows) {
            // footprint and gradient taps inside the image (all of the band but its outermost ring or two): the 12 pixels
            // directly -- the generic path below evaluates to the same operations on the same values
            const float *r0 = I + (unsigned)((c.sy - 1) * cols + c.sx);
            const float *r1 = r0 + cols, *r2 = r1 + cols, *r3 = r2 + cols;
            const float a0 = r0[0], a1 = r0[1];
            const float b_1 = r1[-1], b0 = r1[0], b1 = r1[1], b2 = r1[2];
            const float c_1 = r2[-1], c0 = r2[0], c1 = r2[1], c2 = r2[2];
            const float d0 = r3[0], d1 = r3[1];
            const float fx = c.ax * (1.f / 32), fy = c.ay * (1.f / 32);
            const float w0 = (1.f - fy) * (1.f - fx), w1 = (1.f - fy) * fx, w2 = fy * (1.f - fx), w3 = fy * fx;
            w = b0 * w0 + b1 * w1 + c0 * w2 + c1 * w3;
            gx = (-0.5f * b_1 + 0.5f * b1) * w0 + (-0.5f * b0 + 0.5f * b2) * w1 +
                 (-0.5f * c_1 + 0.5f * c1) * w2 + (-0.5f * c0 + 0.5f * c2) * w3;
            gy = (-0.5f * a0 + 0.5f * c0) * w0 + (-0.5f * a1 + 0.5f * c1) * w1 +
                 (-0.5f * b0 + 0.5f * d0) * w2 + (-0.5f * b1 + 0.5f * d1) * w3;
        } else {
            w = bilinear(pix, rows, cols, c);
            gx = bilinear(gxf, rows, cols, c);
            gy = bilinear(gyf, rows, cols, c);
        }
        ecc_accumulate(acc, w, gx, gy, tmpl[(size_t)y * cols + x], m, x, y);
    }
    ecc_block_store(acc, partial, f, (int)bidx);
}

// ---- interior of the frame, round 3: one COLUMN per thread ------------------------------------------------------
// The 45 sums are products of three things: the warped gradients {gx, gy}, the pixel coordinates {X, Y, 1} and
// {w, 1, t} (or a second gradient).  A thread that owns ONE column x and walks down its rows has a constant X, so X
// comes out of every sum and is multiplied in once, at the end; Y is the row offset r inside a segment of kEccFlush
// rows (a small exact integer), shifted to the true row when the segment's partial sums are folded into the thread's
// double totals.  What is left per pixel are 21 sums
//     {gx, gy} x {1, w, t} x {1, r}        12        {gx^2, gy^2, gx gy} x {1, r, r^2}     9
// accumulated as packed float pairs (v_pk_fma_f32) over at most kEccFlush rows, + the five scalar sums of w and t
// in double (they decide rho, i.e. the iteration count): ~22 VALU instructions per pixel where the 45 double sums of
// round 2 took ~75, and the fixed-point source coordinate splits into a per-thread column term and a per-row term
// from a small LDS table (~10 instead of ~45).  Float partials: a segment sum of <= 32 terms carries a relative
// error of <~1e-7, random over the 30 000 segments of a frame -- the same order as the float rounding of every
// Jacobian element in cv::findTransformECC itself, and five orders below the 1e-4 parity bar (tests: same iteration
// counts, |dM| ~ 1e-7).  The identity iteration (5 loads, no interpolation) becomes memory-bound.
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int kEccFlush = 32;        // rows per float segment
constexpr int kEccRowTab = 3840;     // rows of one block's piece at most: their per-row coordinate terms are tabulated in LDS

struct EccPart {        // float partial sums of one segment
    v2f G0, G1, Gw0, Gw1, Gt0, Gt1, Q0, Q1, Q2, C01;
    float C2;
};
struct EccTot {         // double totals of the thread's column piece (Y = true row)
    double G0[2], G1[2], Gw0[2], Gw1[2], Gt0[2], Gt1[2], Q0[2], Q1[2], Q2[2], C0, C1, C2;
    double Sw, Sww, St, Stt, Stw, n;
    float cf;           // centre: the float products are taken with (w - cf) and (t - cf), see ecc_part_add
};

__device__ __forceinline__ void ecc_part_zero(EccPart &p)
{
    const v2f z = {0.f, 0.f};
    p.G0 = p.G1 = p.Gw0 = p.Gw1 = p.Gt0 = p.Gt1 = p.Q0 = p.Q1 = p.Q2 = p.C01 = z;
    p.C2 = 0.f;
}

// one pixel: warped value w, warped gradients gx / gy, template t, row offset rf (= r as a float) in its segment.
// MASKED (band pixels): m = the nearest-neighbour mask of cv::findTransformECC; masked sums as in ecc_accumulate
// (n, the scalar sums, sum_mask J, sum_mask J t), the others over all pixels.
// Interior pixels: the products with w and t are taken with (w - c), (t - c), c = T.cf = an INTEGER near the template's mean
// (ecc_center_kernel), and c x (sum of the gradients) is added back in double at the end (ecc_tot_value).  The
// subtraction is exact (12-bit images blurred to floats below 4096 minus an integer below 4096), the identity
// sum g w = sum g (w - c) + c sum g too; what changes is the size of the numbers that get rounded: a product g w with
// w ~ 1800 carries an absolute rounding error of |g| x 1e-4, the same product with |w - c| ~ 100 a tenth of that --
// and what the solve uses is sum J w - mean(w) sum J, a difference that used to cancel the leading 1-2 digits of these
// sums (measured on a 49 x 888 frame against sums taken in exact arithmetic: relative error of sum gy w 3.2e-8 ->
// see DESIGN.md section 8; the reference rounds every one of these sums to FLOAT before its 6 x 6 solve, so a sum that
// lands on another float moves the result by a float ulp amplified by the solve).
template <bool MASKED>
__device__ __forceinline__ void ecc_part_add(EccPart &p, EccTot &T, float w, float gx, float gy, float t, float rf, bool m = true)
{
    const float wc = MASKED ? w : w - T.cf, tc = MASKED ? t : t - T.cf;
    const v2f G = {gx, gy}, R = {rf, rf}, W = {wc, wc}, Tt = {tc, tc};
    const v2f Z = {0.f, 0.f};
    const v2f Gm = (MASKED && !m) ? Z : G;
    const float rf2 = rf * rf;
    const v2f R2 = {rf2, rf2};
    p.G0 += Gm;
    p.G1 = __builtin_elementwise_fma(Gm, R, p.G1);
    const v2f Gw = G * W, Gt = Gm * Tt, Q = G * G;
    p.Gw0 += Gw;
    p.Gw1 = __builtin_elementwise_fma(Gw, R, p.Gw1);
    p.Gt0 += Gt;
    p.Gt1 = __builtin_elementwise_fma(Gt, R, p.Gt1);
    p.Q0 += Q;
    p.Q1 = __builtin_elementwise_fma(Q, R, p.Q1);
    p.Q2 = __builtin_elementwise_fma(Q, R2, p.Q2);
    const float c = gx * gy;
    const v2f Cc = {c, c}, R01 = {1.f, rf};
    p.C01 = __builtin_elementwise_fma(Cc, R01, p.C01);
    p.C2 = __builtin_fmaf(c, rf2, p.C2);
    const double wd = w, td = t;
    const double wm = (MASKED && !m) ? 0.0 : wd, tm = (MASKED && !m) ? 0.0 : td;
    if (MASKED) T.n += m ? 1.0 : 0.0;
    T.Sw += wm;
    T.Sww = fma(wm, wd, T.Sww);
    T.St += tm;
    T.Stt = fma(tm, td, T.Stt);
    T.Stw = fma(tm, wd, T.Stw);
}

// segment -> totals: rows of the segment are yb + r
__device__ __forceinline__ void ecc_part_flush(const EccPart &p, EccTot &T, int yb)
{
    const double Y = (double)yb, Y2 = Y * Y, Yd = 2.0 * Y;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const double g0 = p.G0[k], g1 = p.G1[k], gw0 = p.Gw0[k], gw1 = p.Gw1[k], gt0 = p.Gt0[k], gt1 = p.Gt1[k];
        const double q0 = p.Q0[k], q1 = p.Q1[k], q2 = p.Q2[k];
        T.G0[k] += g0;
        T.G1[k] += fma(Y, g0, g1);
        T.Gw0[k] += gw0;
        T.Gw1[k] += fma(Y, gw0, gw1);
        T.Gt0[k] += gt0;
        T.Gt1[k] += fma(Y, gt0, gt1);
        T.Q0[k] += q0;
        T.Q1[k] += fma(Y, q0, q1);
        T.Q2[k] += fma(Y2, q0, fma(Yd, q1, q2));
    }
    const double c0 = p.C01[0], c1 = p.C01[1], c2 = p.C2;
    T.C0 += c0;
    T.C1 += fma(Y, c0, c1);
    T.C2 += fma(Y2, c0, fma(Yd, c1, c2));
}

// the k-th of the 45 sums (layout of ecc_accumulate) from a thread's totals and its column X
template <int K>
__device__ __forceinline__ double ecc_tot_value(const EccTot &T, double X)
{
    const double X2 = X * X, c = (double)T.cf;      // (sum g w = sum g (w - c) + c sum g: exact, in double)
    switch (K) {
    case 0: return T.n;
    case 1: return T.Sw;
    case 2: return T.Sww;
    case 3: return T.St;
    case 4: return T.Stt;
    case 5: return T.Stw;
    case 6: return X * (T.Gw0[0] + c * T.G0[0]);
    case 7: return X * (T.Gw0[1] + c * T.G0[1]);
    case 8: return T.Gw1[0] + c * T.G1[0];
    case 9: return T.Gw1[1] + c * T.G1[1];
    case 10: return T.Gw0[0] + c * T.G0[0];
    case 11: return T.Gw0[1] + c * T.G0[1];
    case 12: return X * T.G0[0];
    case 13: return X * T.G0[1];
    case 14: return T.G1[0];
    case 15: return T.G1[1];
    case 16: return T.G0[0];
    case 17: return T.G0[1];
    case 18: return X * (T.Gt0[0] + c * T.G0[0]);
    case 19: return X * (T.Gt0[1] + c * T.G0[1]);
    case 20: return T.Gt1[0] + c * T.G1[0];
    case 21: return T.Gt1[1] + c * T.G1[1];
    case 22: return T.Gt0[0] + c * T.G0[0];
    case 23: return T.Gt0[1] + c * T.G0[1];
    // J J^T, upper triangle row-major, J = [gx X, gy X, gx Y, gy Y, gx, gy]
    case 24: return X2 * T.Q0[0];     // (0,0) gx^2 X^2
    case 25: return X2 * T.C0;        // (0,1) gx gy X^2
    case 26: return X * T.Q1[0];      // (0,2) gx^2 X Y
    case 27: return X * T.C1;         // (0,3) gx gy X Y
    case 28: return X * T.Q0[0];      // (0,4) gx^2 X
    case 29: return X * T.C0;         // (0,5) gx gy X
    case 30: return X2 * T.Q0[1];     // (1,1) gy^2 X^2
    case 31: return X * T.C1;         // (1,2) gy gx X Y
    case 32: return X * T.Q1[1];      // (1,3) gy^2 X Y
    case 33: return X * T.C0;         // (1,4) gy gx X
    case 34: return X * T.Q0[1];      // (1,5) gy^2 X
    case 35: return T.Q2[0];          // (2,2) gx^2 Y^2
    case 36: return T.C2;             // (2,3) gx gy Y^2
    case 37: return T.Q1[0];          // (2,4) gx^2 Y
    case 38: return T.C1;             // (2,5) gx gy Y
    case 39: return T.Q2[1];          // (3,3) gy^2 Y^2
    case 40: return T.C1;             // (3,4) gy gx Y
    case 41: return T.Q1[1];          // (3,5) gy^2 Y
    case 42: return T.Q0[0];          // (4,4) gx^2
    case 43: return T.C0;             // (4,5) gx gy
    default: return T.Q0[1];          // (5,5) gy^2
    }
}

// One step of a sum over the lanes of a DPP row with moves only (no ds_bpermute round trips: 45 sums x 6 dependent
// shuffle steps were a ~25 000-cycle latency chain at the end of every block).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add_f64(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xFFFFFFFFll), CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xF, false);
    return v + __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);   // (rows outside the mask add 0.0)
}

// Block reduction of the 45 sums straight from the totals, through LDS in three chunks of 15 (the 45 doubles of a
// thread never exist at once): every thread writes its 15 values, thread (v, p) = (t / 16, t % 16) adds 16 of the 256
// entries of value v (stride 16: conflict-free), the 16 partials of a value sit in one DPP row and are added there.
// ~80 instructions per chunk and thread where the wave-wide DPP reduction of every value took ~30 per VALUE -- 1400
// per thread, a third of the identity kernel's VALU instructions (PMC, profiles/r03_ecc_pmc.txt).  Fixed order: deterministic.
constexpr int kEccChunk = 15;
static_assert(kEccSums == 3 * kEccChunk, "three chunks");
// A block's 45 partial sums leave either as plain stores (the solve is the next launch) or, when the LAST block of the frame
// solves in the same launch (ecc_cols_kernel<..., FUSE>), as device-scope atomic exchanges: performed at the memory side,
// coherent across the 8 XCDs without an L2 write-back (hot_scan_kernel's hand-off).  g_ecc_sink takes the exchanges'
// return values so that they have completed before the block's barrier and ticket.
struct EccOut {
    double *partial;
    bool atomic;
    unsigned long long sink;
};
__device__ __forceinline__ void ecc_out_put(EccOut &o, size_t idx, double v)
{
    if (o.atomic)
        o.sink |= atomicExch(reinterpret_cast<unsigned long long *>(o.partial + idx), (unsigned long long)__double_as_longlong(v));
    else
        o.partial[idx] = v;
}
template <int C, int J>
__device__ __forceinline__ void ecc_tot_put(const EccTot &T, double X, bool on, double (*lds)[256])
{
    lds[J][threadIdx.x] = on ? ecc_tot_value<C * kEccChunk + J>(T, X) : 0.0;
    if constexpr (J + 1 < kEccChunk) ecc_tot_put<C, J + 1>(T, X, on, lds);
}
template <int C>
__device__ __forceinline__ void ecc_tot_store(const EccTot &T, double X, bool on, double (*lds)[256],
                                              EccOut &out, int f, unsigned slot)
{
    ecc_tot_put<C, 0>(T, X, on, lds);
    __syncthreads();
    const int v = (int)threadIdx.x >> 4, p = (int)threadIdx.x & 15;
    if (v < kEccChunk) {                       // (waves 0 .. 3, whole DPP rows)
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) s += lds[v][j * 16 + p];
        s = dpp_add_f64<0xB1, 0xF>(s);
        s = dpp_add_f64<0x4E, 0xF>(s);
        s = dpp_add_f64<0x141, 0xF>(s);
        s = dpp_add_f64<0x140, 0xF>(s);        // every lane of the row holds the block's sum of value v
        if (p == 0) ecc_out_put(out, ((size_t)f * kEccSums + (C * kEccChunk + v)) * kEccStride + slot, s);
    }
    __syncthreads();
    if constexpr (C + 1 < 3) ecc_tot_store<C + 1>(T, X, on, lds, out, f, slot);
}

// a block without any pixel: its partial sums are zero
__device__ __forceinline__ void ecc_store_zeros(EccOut &out, int f, unsigned slot)
{
    if (threadIdx.x < kEccSums) ecc_out_put(out, ((size_t)f * kEccSums + threadIdx.x) * kEccStride + slot, 0.0);
}

// Identity iteration with NEIGHBOUR = 1: the left / right taps come from the neighbouring lanes' centre values by DPP
// wave shifts (lane 0 and lane 63 load theirs: one load instruction per row under a two-lane exec mask), so a trip is
// UR + 2 column loads + UR template loads -- registers and load slots that go into more rows in flight: the kernel is
// bound by the bytes it keeps in flight (measured: 8 rows per trip against 4 ...).  Every lane of the wave must run
// the trip (lanes past the rectangle read valid columns and are left out of the reduction).
// uniform base + 32-bit BYTE offset of the lane (+ a constant): the form the compiler turns into
// `global_load_dword v, v_off, s[base] offset:imm` -- one 32-bit add per ROW of a trip instead of a 64-bit shift-and-add
// per LOAD (26 of the 239 VALU instructions of a general two-row trip, 32 of 166 in the identity trip)
template <int IMM = 0>
__device__ __forceinline__ float ld_f32(const float *base, unsigned byte_off)
{
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + byte_off + IMM);
}

template <int IMM = 0>
__device__ __forceinline__ v2f ld_v2f(const float *base, unsigned byte_off)      // two consecutive pixels, one 8-byte load
{
    return *reinterpret_cast<const v2f *>(reinterpret_cast<const char *>(base) + byte_off + IMM);
}

template <bool IDENT, int UR, int NEIGHBOUR = 0, int GXD = 1, int SHARE = 0>
__device__ __forceinline__ void ecc_cols_trip(const float *__restrict__ I, const float *__restrict__ tmpl, int cols, int x,
                                              int y, int r, int ax, int bx, const int2 *rtab, int rt0, const double *M,
                                              EccPart &P, EccTot &T)
{
    if (IDENT && NEIGHBOUR) {
        // (uniform base + 32-bit lane offsets: one address register per load instead of a 64-bit pair)
        const unsigned pitch = 4u * (unsigned)cols, o0 = 4u * (unsigned)(y * cols + x);
        const int lane = threadIdx.x & 63;
        float cc[UR + 2], tt[UR], hh[UR];
        unsigned ob[UR + 2];
#pragma unroll
        for (int k = 0; k < UR + 2; ++k) ob[k] = o0 + (unsigned)(k - 1) * pitch;
#pragma unroll
        for (int k = 0; k < UR + 2; ++k) cc[k] = ld_f32(I, ob[k]);
#pragma unroll
        for (int k = 0; k < UR; ++k) tt[k] = ld_f32(tmpl, ob[k + 1]);
#pragma unroll
        for (int k = 0; k < UR; ++k) hh[k] = 0.f;
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < UR; ++k) hh[k] = ld_f32<-4>(I, ob[k + 1]);
        } else if (lane == 63) {
#pragma unroll
            for (int k = 0; k < UR; ++k) hh[k] = ld_f32<4>(I, ob[k + 1]);
        }
#pragma unroll
        for (int k = 0; k < UR; ++k) {
            // wave_shr:1 -> lane i reads lane i - 1 (lane 0 keeps `old` = its loaded halo); wave_shl:1 -> lane i + 1
            const float l = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(hh[k]), __float_as_int(cc[k + 1]), 0x138, 0xF, 0xF, false));
            const float rr = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(hh[k]), __float_as_int(cc[k + 1]), 0x130, 0xF, 0xF, false));
            __builtin_amdgcn_sched_barrier(0);      // one row after the other: the rows' temporaries must not all be live at once
            ecc_part_add<false>(P, T, cc[k + 1], 0.5f * (rr - l), 0.5f * (cc[k + 2] - cc[k]), tt[k], (float)(r + k));   // (= -a/2 + b/2 exactly)
        }
    } else if (IDENT) {
        // source pixel = target pixel: w = I, gradients = central differences of I (zero fractions make the bilinear
        // weights (1,0,0,0), so the general arithmetic reduces exactly to these taps)
        const unsigned pitch = 4u * (unsigned)cols, o0 = 4u * (unsigned)(y * cols + x);
        float cc[UR + 2], ll[UR], rr[UR], tt[UR];
        unsigned ob[UR + 2];
#pragma unroll
        for (int k = 0; k < UR + 2; ++k) ob[k] = o0 + (unsigned)(k - 1) * pitch;
#pragma unroll
        for (int k = 0; k < UR + 2; ++k) cc[k] = ld_f32(I, ob[k]);
#pragma unroll
        for (int k = 0; k < UR; ++k) {
            ll[k] = ld_f32<-4>(I, ob[k + 1]);
            rr[k] = ld_f32<4>(I, ob[k + 1]);
            tt[k] = ld_f32(tmpl, ob[k + 1]);
        }
#pragma unroll
        for (int k = 0; k < UR; ++k)
            ecc_part_add<false>(P, T, cc[k + 1], 0.5f * (rr[k] - ll[k]), 0.5f * (cc[k + 2] - cc[k]), tt[k], (float)(r + k));
    } else {
        // footprint of a pixel (rows sy-1 .. sy+2 = a, b, c, d; columns sx-1 .. sx+2 = _1, 0, 1, 2), loaded so that the
        // pairs the arithmetic works on are the pairs the loads deliver: {a0,a1} {b0,b1} {c0,c1} {d0,d1} from 8-byte loads,
        // {b_1,b2} {c_1,c2} from two 4-byte loads each
        v2f A[UR], Bm[UR], Cm[UR], D[UR], Be[UR], Ce[UR];
        float tt[UR], fx[UR], fy[UR];
        int sxk[UR], syk[UR];
        const unsigned pitch = 4u * (unsigned)cols, ot = 4u * (unsigned)(y * cols + x);
#pragma unroll
        for (int k = 0; k < UR; ++k) {
            // WarpAffineInvoker's fixed-point coordinate: per-row term (LDS table of the block's rows) + per-column term,
            // each rounded on its own
            const int2 rt = rtab[y + k - rt0];
            const int Xr = rt.x + ax, Yr = rt.y + bx;
            const int Xq = (Xr + 16) >> 5, Yq = (Yr + 16) >> 5;
            sxk[k] = Xq >> 5;                            // (footprint inside the image by construction: ecc_band)
            syk[k] = Yq >> 5;
            fx[k] = (Xq & 31) * (1.f / 32);
            fy[k] = (Yq & 31) * (1.f / 32);
        }
        // SHARE (two rows per trip): under a warp near the identity the second pixel's footprint is the first one's moved down
        // by exactly one row -- rows a, b, c of the second are rows b, c, d of the first.  When that holds for EVERY lane of
        // the wave (uniform branch), the second pixel loads only what is new: the two outer pixels of its row c and its row d:
        // 11 instead of 16 source loads per trip, the same values in the same registers' roles, so the same bits.
        bool shared = false;
        if (SHARE && UR == 2) shared = __all(sxk[1 % UR] == sxk[0] && syk[1 % UR] == syk[0] + 1) != 0;
#pragma unroll
        for (int k = 0; k < UR; ++k) {
            // (rows and columns < 2^15 here: 24-bit multiply, full rate)
            const unsigned q0 = 4u * (unsigned)(__mul24(syk[k] - 1, cols) + sxk[k]), q1 = q0 + pitch, q2 = q1 + pitch, q3 = q2 + pitch;
            if (SHARE && UR == 2 && k == 1 && shared) {
                Ce[k][0] = ld_f32<-4>(I, q2);
                Ce[k][1] = ld_f32<8>(I, q2);
                D[k] = ld_v2f(I, q3);
            } else {
                A[k] = ld_v2f(I, q0);
                Be[k][0] = ld_f32<-4>(I, q1);
                Bm[k] = ld_v2f(I, q1);
                Be[k][1] = ld_f32<8>(I, q1);
                Ce[k][0] = ld_f32<-4>(I, q2);
                Cm[k] = ld_v2f(I, q2);
                Ce[k][1] = ld_f32<8>(I, q2);
                D[k] = ld_v2f(I, q3);
            }
            tt[k] = ld_f32(tmpl, ot + (unsigned)k * pitch);
        }
        if (SHARE && UR == 2 && shared) {
            A[1 % UR] = Bm[0];
            Be[1 % UR] = Ce[0];
            Bm[1 % UR] = Cm[0];
            Cm[1 % UR] = D[0];
        }
#pragma unroll
        for (int k = 0; k < UR; ++k) {
            // Bilinear interpolation of I and of its central differences along x and y (cv::findTransformECC warps the two
            // gradient IMAGES).  All three are linear in the pixels, so with V_j = the vertical interpolation at column j
            //     w  = V_0 + fx (V_1 - V_0)
            //     gx = 1/2 [ q_0 + fx (q_1 - q_0) ],   q = (b_+ - b_-) + fy ((c_+ - c_-) - (b_+ - b_-))  at columns 0, 1
            //     gy = 1/2 [ p_0 + fx (p_1 - p_0) ],   p = (c - a) + fy ((d - b) - (c - a))  at columns 0, 1
            // -- 19 instructions where the four-weight form of round 2 took ~50.  With zero fractions (identity) these
            // are exactly the taps of the identity iteration: V_j = b_j, w = b_0, gx = (b_1 - b_-1) / 2, gy = (c_0 - a_0) / 2.
            // (gx from the horizontal DIFFERENCES of the pixels, interpolated -- not from differences of the interpolated
            //  V_j: those carry the rounding of values ~2000, 1e-4, into a gradient of a few counts, 1e-5 relative, where the
            //  reference's warped gradient image is good to 1e-7; the ECC iteration amplifies that on small or weakly
            //  textured images -- tests/debug/soak_ecc.py found it, 8e-3 px on a 97 x 258 frame.  Two instructions more.)
            const v2f FY = {fy[k], fy[k]};
            const v2f Vm = __builtin_elementwise_fma(FY, Cm[k] - Bm[k], Bm[k]);     // {V_0, V_1}
            const float w = __builtin_fmaf(fx[k], Vm[1] - Vm[0], Vm[0]);
            float gx;
            if (GXD) {
                // q_j = (b_+ - b_-) + fy ((c_+ - c_-) - (b_+ - b_-)), the bracket regrouped as (c_+ - b_+) - (c_- - b_-) so that the
                // packed differences the loads deliver as pairs (Cm - Bm, Ce - Be) are used as they are
                const v2f Dm = Cm[k] - Bm[k], De = Ce[k] - Be[k];                     // {c0-b0, c1-b1}, {c_1-b_1, c2-b2}
                const float q0 = __builtin_fmaf(fy[k], Dm[1] - De[0], Bm[k][1] - Be[k][0]);
                const float q1 = __builtin_fmaf(fy[k], De[1] - Dm[0], Be[k][1] - Bm[k][0]);
                gx = 0.5f * __builtin_fmaf(fx[k], q1 - q0, q0);
            } else {      // (measurement switch UPSP_ECC_GX=0: the first form, differences of the interpolated columns)
                const v2f Ve = __builtin_elementwise_fma(FY, Ce[k] - Be[k], Be[k]);     // {V_-1, V_2}
                const float dx0 = Vm[1] - Ve[0], dx1 = Ve[1] - Vm[0];
                gx = 0.5f * __builtin_fmaf(fx[k], dx1 - dx0, dx0);
            }
            const v2f E = Cm[k] - A[k], F = D[k] - Bm[k];
            const v2f Pp = __builtin_elementwise_fma(FY, F - E, E);
            const float gy = 0.5f * __builtin_fmaf(fx[k], Pp[1] - Pp[0], Pp[0]);
            ecc_part_add<false>(P, T, w, gx, gy, tt[k], (float)(r + k));
        }
    }
}

// One row of the general iteration split into its loads and its sums, for the software-pipelined walk below: the 12 source pixels
// + the template pixel of row y of this thread's column (same loads, same arithmetic as ecc_cols_trip's general path).
struct EccRow {
    v2f A, Bm, Cm, D, Be, Ce;
    float tt, fx, fy;
};
__device__ __forceinline__ EccRow ecc_row_load(const float *__restrict__ I, const float *__restrict__ tmpl, int cols, int x, int y,
                                               int ax, int bx, const int2 *rtab, int rt0)
{
    EccRow q;
    const unsigned pitch = 4u * (unsigned)cols;
    const int2 rt = rtab[y - rt0];
    const int Xr = rt.x + ax, Yr = rt.y + bx;
    const int Xq = (Xr + 16) >> 5, Yq = (Yr + 16) >> 5;
    const int sx = Xq >> 5, sy = Yq >> 5;
    q.fx = (Xq & 31) * (1.f / 32);
    q.fy = (Yq & 31) * (1.f / 32);
    const unsigned q0 = 4u * (unsigned)(__mul24(sy - 1, cols) + sx), q1 = q0 + pitch, q2 = q1 + pitch, q3 = q2 + pitch;
    q.A = ld_v2f(I, q0);
    q.Be[0] = ld_f32<-4>(I, q1);
    q.Bm = ld_v2f(I, q1);
    q.Be[1] = ld_f32<8>(I, q1);
    q.Ce[0] = ld_f32<-4>(I, q2);
    q.Cm = ld_v2f(I, q2);
    q.Ce[1] = ld_f32<8>(I, q2);
    q.D = ld_v2f(I, q3);
    q.tt = ld_f32(tmpl, 4u * (unsigned)(y * cols + x));
    return q;
}
__device__ __forceinline__ void ecc_row_sum(const EccRow &q, EccPart &P, EccTot &T, float rf)
{
    const v2f FY = {q.fy, q.fy};
    const v2f Vm = __builtin_elementwise_fma(FY, q.Cm - q.Bm, q.Bm);
    const float w = __builtin_fmaf(q.fx, Vm[1] - Vm[0], Vm[0]);
    const v2f Dm = q.Cm - q.Bm, De = q.Ce - q.Be;
    const float g0 = __builtin_fmaf(q.fy, Dm[1] - De[0], q.Bm[1] - q.Be[0]);
    const float g1 = __builtin_fmaf(q.fy, De[1] - Dm[0], q.Be[1] - q.Bm[0]);
    const float gx = 0.5f * __builtin_fmaf(q.fx, g1 - g0, g0);
    const v2f E = q.Cm - q.A, F = q.D - q.Bm;
    const v2f Pp = __builtin_elementwise_fma(FY, F - E, E);
    const float gy = 0.5f * __builtin_fmaf(q.fx, Pp[1] - Pp[0], Pp[0]);
    ecc_part_add<false>(P, T, w, gx, gy, q.tt, rf);
}
// rows [y0, y1) of the column, the loads of rows y + 1 and y + 2 in flight while row y is summed (three row slots rotating;
// float segments of kEccFlush rows from y0 exactly like the trip loop: the same sums in the same order, the same bits)
__device__ __forceinline__ void ecc_rows_pipelined(const float *__restrict__ I, const float *__restrict__ tmpl, int cols, int x,
                                                   int y0, int y1, int ax, int bx, const int2 *rtab, EccTot &T)
{
    EccPart P;
    ecc_part_zero(P);
    EccRow s0, s1, s2;
    s0 = ecc_row_load(I, tmpl, cols, x, y0, ax, bx, rtab, y0);
    s1 = y0 + 1 < y1 ? ecc_row_load(I, tmpl, cols, x, y0 + 1, ax, bx, rtab, y0) : s0;
    int yb = y0;
    auto one = [&](EccRow &cur, EccRow &refill, int y) {
        if (y + 2 < y1) refill = ecc_row_load(I, tmpl, cols, x, y + 2, ax, bx, rtab, y0);
        if (y - yb == kEccFlush) {                                  // (uniform)
            ecc_part_flush(P, T, yb);
            ecc_part_zero(P);
            yb = y;
        }
        ecc_row_sum(cur, P, T, (float)(y - yb));
    };
    int y = y0;
    for (; y + 3 <= y1; y += 3) {
        one(s0, s2, y);
        one(s1, s0, y + 1);
        one(s2, s1, y + 2);
    }
    if (y < y1) one(s0, s2, y);
    if (y + 1 < y1) one(s1, s0, y + 1);
    ecc_part_flush(P, T, yb);
}

// Interior block `blk` of `nblk`: the inner rectangle (farther than the band from every edge) is cut into column tiles
// of 256 and, per tile, into nblk / tiles row pieces; blocks beyond that store zeros.  Needs nblk >= tiles (the host
// checks: cols <= 256 x interior blocks, else the round-2 kernel runs).
template <bool IDENT, int UR, int NEIGHBOUR, int GXD, int SHARE>
__device__ __forceinline__ void ecc_cols_body(const float *__restrict__ img, const float *__restrict__ tmpl, int rows,
                                              int cols, const EccState *__restrict__ state,
                                              EccOut &out, int f, unsigned blk, unsigned nblk,
                                              double (*lds_red)[256], float center)
{
    // one LDS area: the per-row coordinate table while the rows are walked, the reduction chunks afterwards
    static_assert(sizeof(int2) * kEccRowTab <= sizeof(double) * kEccChunk * 256, "row table fits the reduction area");
    int2 *rtab_s = reinterpret_cast<int2 *>(&lds_red[0][0]);
    const EccState &es = state[f];
    const float *I = img + (size_t)f * rows * cols;
    const EccMargins g = ecc_margins(IDENT ? 3 : es.band, rows, cols);
    const int x_lo = g.left, x_hi = cols - g.right, y_lo = g.top, y_hi = rows - g.bottom;   // [lo, hi)
    const int W = max(x_hi - x_lo, 0), H = max(y_hi - y_lo, 0);
    const unsigned tiles = ((unsigned)W + 255u) / 256u;
    const unsigned pieces = tiles ? nblk / tiles : 0u;                       // row pieces per column tile (>= 1)
    const bool work = tiles && blk < tiles * pieces && H > 0;
    const unsigned ct = work ? blk % tiles : 0u, piece = work ? blk / tiles : 0u;
    const int rpp = pieces ? (int)(((unsigned)H + pieces - 1u) / pieces) : 0;  // rows per piece
    const int y0 = work ? min(y_hi, y_lo + (int)piece * rpp) : 0, y1 = work ? min(y_hi, y0 + rpp) : 0;
    const int x_own = x_lo + (int)ct * 256 + (int)threadIdx.x;
    const bool on = work && x_own < x_hi && y1 > y0;
    // NEIGHBOUR: lanes past the rectangle run along on a valid column (the lane after the last one must hold column
    // x_hi, which exists: the band is >= 3 wide) and are left out of the reduction
    const int x = NEIGHBOUR ? min(x_own, cols - 1) : x_own;
    if (!work || y1 <= y0) {            // (uniform) more blocks than pieces: nothing to add
        ecc_store_zeros(out, f, kEccBorderBlocks + blk);
        return;
    }
    double M[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) M[i] = es.M[i];
    if (!IDENT) {      // (the host sizes the grid so that a piece has at most kEccRowTab rows)
        for (int i = threadIdx.x; i < y1 - y0; i += 256) {
            const int yy = y0 + i;
            rtab_s[i] = make_int2(__double2int_rn((M[1] * yy + M[2]) * 1024), __double2int_rn((M[4] * yy + M[5]) * 1024));
        }
        __syncthreads();
    }
    EccTot T;
#pragma unroll
    for (int k = 0; k < 2; ++k)
        T.G0[k] = T.G1[k] = T.Gw0[k] = T.Gw1[k] = T.Gt0[k] = T.Gt1[k] = T.Q0[k] = T.Q1[k] = T.Q2[k] = 0.0;
    T.C0 = T.C1 = T.C2 = T.Sw = T.Sww = T.St = T.Stt = T.Stw = 0.0;
    T.n = on ? (double)(y1 - y0) : 0.0;          // mask = 1 on every interior pixel
    T.cf = center;
    if (on || NEIGHBOUR) {
        const int ax = IDENT ? 0 : __double2int_rn(M[0] * x * 1024), bx = IDENT ? 0 : __double2int_rn(M[3] * x * 1024);
        const int2 *rt = rtab_s;
        if (!IDENT && SHARE == 2) {
            ecc_rows_pipelined(I, tmpl, cols, x, y0, y1, ax, bx, rt, T);
        } else
        for (int yb = y0; yb < y1; yb += kEccFlush) {
            const int ne = min(kEccFlush, y1 - yb);
            EccPart P;
            ecc_part_zero(P);
            int r = 0;
            for (; r + UR <= ne; r += UR) ecc_cols_trip<IDENT, UR, NEIGHBOUR, GXD, SHARE>(I, tmpl, cols, x, yb + r, r, ax, bx, rt, y0, M, P, T);
            for (; r < ne; ++r) ecc_cols_trip<IDENT, 1, NEIGHBOUR, GXD>(I, tmpl, cols, x, yb + r, r, ax, bx, rt, y0, M, P, T);
            ecc_part_flush(P, T, yb);
        }
    }
    __syncthreads();                              // (every read of the row table is done)
    ecc_tot_store<0>(T, (double)x, on, lds_red, out, f, kEccBorderBlocks + blk);
}

// The band of the same launch, also one column per thread (no 45 double accumulators anywhere in this kernel: with
// round 2's band body inside, the kernel needed 146+ VGPRs whatever the interior loop used).  Band blocks:
//   [0, tiles)            top strip    rows [0, top),            one column tile of 256 each
//   [tiles, 2 tiles)      bottom strip rows [rows - bottom, rows)
//   the rest              the left + right strips between them: `side` = left + right columns; a block's 256 threads
//                         are (column, row piece) pairs, so a 6-column band still has 42 threads per block at work
// Generic bilinear (constant-0 border, reflect-101 gradient taps) and the nearest-neighbour mask, as in round 2.
__device__ __forceinline__ void ecc_band_cols_body(const float *__restrict__ img, const float *__restrict__ tmpl, int rows,
                                                   int cols, const EccState *__restrict__ state,
                                                   EccOut &out, int f, unsigned bidx, bool ident,
                                                   double (*lds_red)[256])
{
    const EccState &es = state[f];
    const float *I = img + (size_t)f * rows * cols;
    double M[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) M[i] = es.M[i];
    const EccMargins g = ecc_margins(ident ? 3 : es.band, rows, cols);
    const int tiles = (cols + 255) / 256;
    const int tid = (int)threadIdx.x;
    int x = -1, ya = 0, yb = 0;
    if ((int)bidx < tiles) {
        x = (int)bidx * 256 + tid;
        ya = 0;
        yb = g.top;
    } else if ((int)bidx < 2 * tiles) {
        x = ((int)bidx - tiles) * 256 + tid;
        ya = rows - g.bottom;
        yb = rows;
    } else {
        const int side = g.left + g.right, H = rows - g.top - g.bottom;
        const int sb = (int)bidx - 2 * tiles, nsb = kEccBorderBlocks - 2 * tiles;
        const int tiles_s = (side + 255) / 256;
        int pb = tiles_s ? nsb / tiles_s : 0;                    // blocks per column tile of the strips ...
        if (side > 0 && H > 0 && pb >= 1) {
            // ... of which only as many are used as give every thread ~8 rows (the others store zeros at once)
            const int sp_min = 256 / min(256, side);
            pb = min(pb, max(1, (H + 8 * sp_min - 1) / (8 * sp_min)));
        }
        if (side > 0 && H > 0 && pb >= 1 && sb < tiles_s * pb) {
            const int ts = sb % tiles_s, pblk = sb / tiles_s;
            const int cs = min(256, side - ts * 256), sp = 256 / cs;
            const int c = tid % cs, q = tid / cs;
            if (q < sp) {
                const int pieces = pb * sp, rpp = (H + pieces - 1) / pieces, piece = pblk * sp + q;
                ya = min(rows - g.bottom, g.top + piece * rpp);
                yb = min(rows - g.bottom, ya + rpp);
                const int k = ts * 256 + c;
                x = k < g.left ? k : cols - side + k;
            }
        }
    }
    const bool on = x >= 0 && x < cols && yb > ya;
    if (!__syncthreads_or(on ? 1 : 0)) {     // no thread of the block has a pixel (spare band block, empty strip)
        ecc_store_zeros(out, f, bidx);
        return;
    }
    EccTot T;
#pragma unroll
    for (int k = 0; k < 2; ++k)
        T.G0[k] = T.G1[k] = T.Gw0[k] = T.Gw1[k] = T.Gt0[k] = T.Gt1[k] = T.Q0[k] = T.Q1[k] = T.Q2[k] = 0.0;
    T.C0 = T.C1 = T.C2 = T.Sw = T.Sww = T.St = T.Stt = T.Stw = T.n = 0.0;
    T.cf = 0.f;
    if (on) {
        auto pix = [&](int yy, int xx) { return I[(size_t)yy * cols + xx]; };
        auto gxf = [&](int yy, int xx) {
            return -0.5f * pix(yy, reflect101(xx - 1, cols)) + 0.5f * pix(yy, reflect101(xx + 1, cols));
        };
        auto gyf = [&](int yy, int xx) {
            return -0.5f * pix(reflect101(yy - 1, rows), xx) + 0.5f * pix(reflect101(yy + 1, rows), xx);
        };
        const int ax = __double2int_rn(M[0] * x * 1024), bx = __double2int_rn(M[3] * x * 1024);
        for (int y0 = ya; y0 < yb; y0 += kEccFlush) {
            const int ne = min(kEccFlush, yb - y0);
            EccPart P;
            ecc_part_zero(P);
            for (int r = 0; r < ne; ++r) {
                const int y = y0 + r;
                const int Xr = __double2int_rn((M[1] * y + M[2]) * 1024) + ax;
                const int Yr = __double2int_rn((M[4] * y + M[5]) * 1024) + bx;
                const int Xq = (Xr + 16) >> 5, Yq = (Yr + 16) >> 5;
                WarpCoord c;
                c.sx = max(-32768, min(32767, Xq >> 5));
                c.sy = max(-32768, min(32767, Yq >> 5));
                c.ax = Xq & 31;
                c.ay = Yq & 31;
                const int nx = max(-32768, min(32767, (Xr + 512) >> 10)), ny = max(-32768, min(32767, (Yr + 512) >> 10));
                const bool m = (unsigned)nx < (unsigned)cols && (unsigned)ny < (unsigned)rows;
                float w, gx, gy;
                if (c.sx >= 1 && c.sx + 2 < cols && c.sy >= 1 && c.sy + 2 < rows) {
                    // footprint and gradient taps inside the image (all of the band but its outermost ring or two): the 12
                    // pixels directly -- the generic path below evaluates to the same operations on the same values
                    const float *r0 = I + (unsigned)((c.sy - 1) * cols + c.sx);
                    const float *r1 = r0 + cols, *r2 = r1 + cols, *r3 = r2 + cols;
                    const float a0 = r0[0], a1 = r0[1];
                    const float b_1 = r1[-1], b0 = r1[0], b1 = r1[1], b2 = r1[2];
                    const float c_1 = r2[-1], c0 = r2[0], c1 = r2[1], c2 = r2[2];
                    const float d0 = r3[0], d1 = r3[1];
                    const float fx = c.ax * (1.f / 32), fy = c.ay * (1.f / 32);
                    const float w0 = (1.f - fy) * (1.f - fx), w1 = (1.f - fy) * fx, w2 = fy * (1.f - fx), w3 = fy * fx;
                    w = b0 * w0 + b1 * w1 + c0 * w2 + c1 * w3;
                    gx = (-0.5f * b_1 + 0.5f * b1) * w0 + (-0.5f * b0 + 0.5f * b2) * w1 +
                         (-0.5f * c_1 + 0.5f * c1) * w2 + (-0.5f * c0 + 0.5f * c2) * w3;
                    gy = (-0.5f * a0 + 0.5f * c0) * w0 + (-0.5f * a1 + 0.5f * c1) * w1 +
                         (-0.5f * b0 + 0.5f * d0) * w2 + (-0.5f * b1 + 0.5f * d1) * w3;
                } else {
                    w = bilinear(pix, rows, cols, c);
                    gx = bilinear(gxf, rows, cols, c);
                    gy = bilinear(gyf, rows, cols, c);
                }
                ecc_part_add<true>(P, T, w, gx, gy, tmpl[(size_t)y * cols + x], (float)r, m);
            }
            ecc_part_flush(P, T, y0);
        }
    }
    ecc_tot_store<0>(T, (double)x, on, lds_red, out, f, bidx);
}

// centre of the float products (ecc_part_add): the mean of a 64 x 64 sample grid of the blurred template, rounded to an
// integer and kept inside [0, 4095] (any integer there keeps w - c exact; the nearer to the image's mean, the smaller the
// products).  Once per reference image; fixed order: deterministic.
__global__ void __launch_bounds__(256) ecc_center_kernel(const float *__restrict__ tmpl, int rows, int cols, float *__restrict__ out)
{
    __shared__ double sh[256];
    double a = 0.0;
    for (int k = threadIdx.x; k < 4096; k += 256) {
        const int y = (int)(((long long)(k >> 6) * rows) >> 6), x = (int)(((long long)(k & 63) * cols) >> 6);
        a += (double)tmpl[(size_t)y * cols + x];
    }
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double m = sh[0] / 4096.0;
        out[0] = (m >= 0.0 && m <= 4095.0) ? (float)rint(m) : 0.f;      // (images outside the 12-bit range: no centring)
    }
}

template <bool ATOMIC>
__device__ __forceinline__ void ecc_solve_body(EccState &es, const double *__restrict__ partial, int f, int nblocks, int max_iters, double eps,
                               int rows, int cols, double *Ssh);
constexpr int kEccTicketStride = 32;     // one ticket per frame on its own 128-byte line

// FUSE: the block that finishes a frame LAST (a ticket per frame) reduces the frame's partial sums and solves the iteration in
// the same launch -- no ecc_solve_kernel launch between two sums launches (16 us of a dependent chain + a launch gap, 32 times
// per 1000 frames).  The partials and the ticket move through device-scope atomics only (EccOut).
template <bool IDENT, int UR, int WAVES, int NEIGHBOUR = 0, int GXD = 1, int FUSE = 0, int SHARE = 0>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
    ecc_cols_kernel(const float *__restrict__ img, const float *__restrict__ tmpl, int rows, int cols,
                    EccState *__restrict__ state, double *__restrict__ partial, const float *__restrict__ center,
                    unsigned *__restrict__ tickets, int max_iters, double eps)
{
    __shared__ double lds_red[kEccChunk][256];     // row table / reduction chunks of whichever body runs
    const int f = blockIdx.x;
    if (state[f].done) return;
    const unsigned nint = gridDim.y - (unsigned)kEccBorderBlocks;
    EccOut out = {partial, FUSE != 0, 0ull};
    if (blockIdx.y >= (unsigned)kEccBorderBlocks)
        ecc_cols_body<IDENT, UR, NEIGHBOUR, GXD, SHARE>(img, tmpl, rows, cols, state, out, f, blockIdx.y - (unsigned)kEccBorderBlocks, nint, lds_red, *center);
    else
        ecc_band_cols_body(img, tmpl, rows, cols, state, out, f, blockIdx.y, IDENT, lds_red);
    if (FUSE) {
        // (the exchanges' return values are in out.sink: they have completed before the barrier; the ticket follows it)
        __shared__ int s_last;
        const int any = __syncthreads_or((int)(out.sink == 0x7FF8DEADBEEF0001ull));      // (never true: only the dependency matters)
        if (threadIdx.x == 0) {
            const unsigned ticket = atomicAdd(&tickets[(size_t)f * kEccTicketStride], 1u + (unsigned)any);
            s_last = ticket == gridDim.y - 1u;
            if (s_last) atomicExch(&tickets[(size_t)f * kEccTicketStride], 0u);          // clean for the next launch
        }
        __syncthreads();
        if (s_last) ecc_solve_body<true>(state[f], partial, f, (int)gridDim.y, max_iters, eps, rows, cols, &lds_red[0][0]);
    }
}

// ---- Gaussian 5 x 5 pre-blur and the identity iteration of the ECC in ONE pass ----------------------------------------
// cv::findTransformECC blurs the input (GaussianBlur 5 x 5) and its first iteration runs with the identity warp: the
// warped image IS the blurred image and the warped gradients are its central differences.  Round 2 wrote the blurred
// frame (tile kernel, 6 B per pixel) and read it back for the identity sums (8 B per pixel); here one column-walking
// kernel does both: 2 B (frame) + 4 B (template) in, 4 B (blurred frame, for the later iterations) out.
//
// A WAVE owns 62 columns (lanes 1 .. 62; lanes 0 and 63 carry the two neighbour columns the x-gradient needs) and
// walks down the rows of its piece with everything rolling in registers: the horizontal pass takes the four neighbour
// pixels from the neighbouring lanes by DPP wave shifts (lanes 0 / 63 load theirs), five rows of horizontal results
// give a blurred row, three blurred rows give the gradients -- no LDS, no barrier in the loop, waves independent.
// Same float operations in the same order as gauss_pass_kernel / gauss_fused_kernel (the blurred frame is bit-identical:
// tests/test_imageops_gpu.py compares upsp_blur_u16 with the oracle), reflect-101 by loading reflected rows and
// columns: a lane on a virtual column -1 computes exactly the blurred column 1, so the gradient taps at the image border
// are cv::Sobel's reflected ones without a special case, and the whole frame is "interior" (mask = 1 everywhere under
// the identity warp): no band blocks in this launch.
constexpr int kGcOwn = 62;           // columns a wave owns

__device__ __forceinline__ float dpp_shr1(float old, float v)      // lane i <- lane i - 1, lane 0 <- old
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x138, 0xF, 0xF, false));
}
__device__ __forceinline__ float dpp_shl1(float old, float v)      // lane i <- lane i + 1, lane 63 <- old
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x130, 0xF, 0xF, false));
}

template <bool ECC, int U>
__device__ __forceinline__ void gauss5_cols_body(const uint16_t *__restrict__ src, float *__restrict__ dst,
                                                 const float *__restrict__ tmpl, int rows, int cols, int rpp, float k0,
                                                 float k1, float k2, const EccState *__restrict__ state,
                                                 double *__restrict__ partial, unsigned slot0, double (*lds_red)[256])
{
    const int f = blockIdx.x;
    if (ECC && state[f].done) return;
    const int lane = threadIdx.x & 63;
    const int tile = (int)blockIdx.z * 4 + (int)(threadIdx.x >> 6);        // column tile of this wave
    const int x = tile * kGcOwn - 1 + lane;                                // (virtual for x < 0 or x >= cols)
    const int y0 = (int)blockIdx.y * rpp, y1 = min(rows, y0 + rpp);
    const bool wave_on = tile * kGcOwn < cols && y1 > y0;                   // (uniform per wave)
    const bool own = wave_on && lane >= 1 && lane <= kGcOwn && x >= 0 && x < cols;
    const size_t npix = (size_t)rows * cols;
    const uint16_t *S = src + (size_t)f * npix;
    float *B = dst + (size_t)f * npix;
    EccTot T;
    if (ECC) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
            T.G0[k] = T.G1[k] = T.Gw0[k] = T.Gw1[k] = T.Gt0[k] = T.Gt1[k] = T.Q0[k] = T.Q1[k] = T.Q2[k] = 0.0;
        T.C0 = T.C1 = T.C2 = T.Sw = T.Sww = T.St = T.Stt = T.Stw = 0.0;
        T.n = own ? (double)(y1 - y0) : 0.0;
        T.cf = 0.f;                               // (opt-in measurement kernel: products not centred)
    }
    if (wave_on) {
        // column offsets of the lane's own pixel and of the two pixels only the edge lanes fetch (reflect-101)
        const unsigned cx = 2u * (unsigned)reflect101(x, cols);
        const unsigned ca = 2u * (unsigned)reflect101(lane == 0 ? x - 1 : x + 1, cols);
        const unsigned cb = 2u * (unsigned)reflect101(lane == 0 ? x - 2 : x + 2, cols);
        const bool edge = lane == 0 || lane == 63;
        const unsigned pitch2 = 2u * (unsigned)cols;
        const unsigned ox = 4u * (unsigned)max(0, min(x, cols - 1));        // f32 column offset (template, output)
        float h0 = 0.f, h1 = 0.f, h2 = 0.f, h3 = 0.f, h4 = 0.f;              // h(yi - 4 .. yi)
        float b0 = 0.f, b1 = 0.f, b2 = 0.f;                                  // b(yb - 2 .. yb)
        EccPart P;
        if (ECC) ecc_part_zero(P);
        int yseg = y0;                                                      // first row of the open float segment
        const int niter = (y1 - y0) + 6;
        for (int g = 0; g < niter; g += U) {
            // the loads of U input rows first
            float s[U], ha[U], hb[U], tt[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int yi = y0 - 3 + g + u;
                const unsigned ro = (unsigned)reflect101(min(max(yi, -(rows - 1)), 2 * rows - 2), rows) * pitch2;
                s[u] = (float)*reinterpret_cast<const uint16_t *>(reinterpret_cast<const char *>(S) + (ro + cx));
                ha[u] = hb[u] = 0.f;
                if (edge) {
                    ha[u] = (float)*reinterpret_cast<const uint16_t *>(reinterpret_cast<const char *>(S) + (ro + ca));
                    hb[u] = (float)*reinterpret_cast<const uint16_t *>(reinterpret_cast<const char *>(S) + (ro + cb));
                }
                if (ECC) {
                    const int ye = min(max(yi - 3, 0), rows - 1);
                    tt[u] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(tmpl) + ((unsigned)ye * 2u * pitch2 + ox));
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = g + u;
                if (k >= niter) break;                                     // (uniform)
                const int yi = y0 - 3 + k;
                // horizontal pass: neighbours by wave shifts; lanes 0 / 63 supply the pixels beyond the wave
                const float m1 = dpp_shr1(ha[u], s[u]), p1 = dpp_shl1(ha[u], s[u]);
                const float m2 = dpp_shr1(hb[u], m1), p2 = dpp_shl1(hb[u], p1);
                float hn = k0 * s[u];
                hn += k1 * (m1 + p1);
                hn += k2 * (m2 + p2);
                h0 = h1; h1 = h2; h2 = h3; h3 = h4; h4 = hn;
                if (k < 4) continue;
                // vertical pass: blurred row yb = yi - 2
                float bn = k0 * h2;
                bn += k1 * (h1 + h3);
                bn += k2 * (h0 + h4);
                const int yb = yi - 2;
                if (own && yb >= y0 && yb < y1)
                    __builtin_nontemporal_store(bn, reinterpret_cast<float *>(reinterpret_cast<char *>(B) + ((unsigned)yb * 2u * pitch2 + ox)));
                b0 = b1; b1 = b2; b2 = bn;
                if (!ECC || k < 6) continue;
                // identity iteration at row ye = yi - 3: w = b, gradients = central differences of b
                const int ye = yi - 3;
                const float l = dpp_shr1(0.f, b1), r = dpp_shl1(0.f, b1);
                if (ye - yseg == kEccFlush) {                               // (uniform)
                    ecc_part_flush(P, T, yseg);
                    ecc_part_zero(P);
                    yseg = ye;
                }
                ecc_part_add<false>(P, T, b1, 0.5f * (r - l), 0.5f * (b2 - b0), tt[u], (float)(ye - yseg));
            }
        }
        if (ECC) ecc_part_flush(P, T, yseg);
    }
    if (ECC) {
        const unsigned slot = slot0 + blockIdx.y * gridDim.z + blockIdx.z;
        EccOut out = {partial, false, 0ull};
        ecc_tot_store<0>(T, (double)x, own, lds_red, out, f, slot);
    }
}

// blur + identity iteration: the ECC totals hold the kernel at 4 (3) waves per SIMD
template <int U, int WAVES>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
    gauss5_ecc0_kernel(const uint16_t *__restrict__ src, float *__restrict__ dst, const float *__restrict__ tmpl, int rows,
                       int cols, int rpp, float k0, float k1, float k2, const EccState *__restrict__ state,
                       double *__restrict__ partial, unsigned slot0)
{
    __shared__ double lds_red[kEccChunk][256];
    gauss5_cols_body<true, U>(src, dst, tmpl, rows, cols, rpp, k0, k1, k2, state, partial, slot0, lds_red);
}
// blur alone: ~30 registers, as many waves as the CU takes
template <int U>
__global__ void __launch_bounds__(256)
    gauss5_blur_kernel(const uint16_t *__restrict__ src, float *__restrict__ dst, int rows, int cols, int rpp, float k0, float k1,
                       float k2)
{
    gauss5_cols_body<false, U>(src, dst, nullptr, rows, cols, rpp, k0, k1, k2, nullptr, nullptr, 0u, nullptr);
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
                       float k2, unsigned thresh, unsigned *__restrict__ flag, int only_flagged, int frame_slowest)
{
    // flag (may be null): per-frame words.  only_flagged = 0: set flag[f] when a pixel of the frame is >= thresh (the scan of
    // fix_hot_pixels, cv_extras.cpp:237-247, done on the pixels the blur loads anyway); 1: blur only the frames whose
    // flag is set (their second blur, after the repair)
    // (frame_slowest: grid (column groups, row pieces, frames) instead of (frames, row pieces, column groups) -- measurement switch)
    const int f = frame_slowest ? blockIdx.z : blockIdx.x;
    const int bz = frame_slowest ? blockIdx.x : blockIdx.z;
    if (only_flagged && !flag[f]) return;
    const bool detect = flag && !only_flagged;
    unsigned hot = 0u;
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
            if (detect)
                hot |= (unsigned)((q[u].x & 0xFFFFu) >= thresh) | (unsigned)((q[u].x >> 16) >= thresh) |
                       (unsigned)((q[u].y & 0xFFFFu) >= thresh) | (unsigned)((q[u].y >> 16) >= thresh);
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
    // (every pixel of the frame is loaded by some lane -- halo rows and re-read quads only repeat pixels of the same frame)
    if (detect && __any(hot) && lane == 0) atomicOr(&flag[f], 1u);
}

bool gauss5_quad_applies(const uint16_t *src, const float *dst, int nimg, int rows, int cols)
{
    // UPSP_GAUSS5_QUAD=0: the tile kernel (measurement / test switch)
    return env_int_io("UPSP_GAUSS5_QUAD", 1) && !(cols & 3) && cols >= 8 && rows >= 3 && (long long)rows * cols < (1ll << 29) &&
           !(reinterpret_cast<uintptr_t>(src) & 7) && !(reinterpret_cast<uintptr_t>(dst) & 15) && nimg <= 65535 &&
           !((size_t)rows * cols & 3);          // (every frame of the batch 8-byte aligned)
}

bool launch_gauss5_quad(const uint16_t *src, float *dst, int nimg, int rows, int cols, const FilterCoef &fc, hipStream_t st,
                        unsigned thresh, unsigned *flag, int only_flagged)
{
    if (!gauss5_quad_applies(src, dst, nimg, rows, cols)) return false;
    const int rpp = std::max(8, env_int_io("UPSP_GAUSS5_QUAD_RPP", 64)), uvar = env_int_io("UPSP_GAUSS5_QUAD_U", 8);   // (measurement switches)
    // (1000 frames of 1024^2, ms of pre-blur per step: 2 / 4 / 6 / 8 rows in flight 1.60 / 1.50 / 1.35 / 1.25-1.33 at 64 rows per
    //  piece; 8 rows in flight at 16 / 32 / 48 / 128 rows per piece 1.32 / 1.36 / 1.39 / 1.52; the tile kernel 2.13)
    const int pieces = (rows + rpp - 1) / rpp, zb = (cols + 1023) / 1024;
    if (pieces > 65535 || zb > 65535) return false;
    // frames slowest in the grid: 1.27 against 1.35 ms of pre-blur per 1000 frames, and the identity iteration behind it 1-2 % faster
    const int fs = env_int_io("UPSP_GAUSS5_QUAD_ORDER", 1);
#define UPSP_GQ(UU)                                                                                                 \
    hipLaunchKernelGGL((gauss5_quad_kernel<UU>), fs ? dim3((unsigned)zb, (unsigned)pieces, (unsigned)nimg) : dim3((unsigned)nimg, (unsigned)pieces, (unsigned)zb), dim3(256), 0, st, src, dst, rows, \
                       cols, rpp, fc.k[2], fc.k[3], fc.k[4], thresh, flag, only_flagged, fs)
    if (uvar == 2) UPSP_GQ(2); else if (uvar == 4) UPSP_GQ(4); else if (uvar == 6) UPSP_GQ(6); else UPSP_GQ(8);
#undef UPSP_GQ
    return true;
}

// grid (frames, kEccBorderBlocks + interior blocks): the frame is the FAST index, so the band blocks of all frames
// (slots 0 .. kEccBorderBlocks-1: few trips of long dependent chains) are dispatched first and run beside the interior
// blocks instead of after them (as the last blocks of the launch they were a 50-us tail)
template <bool IDENT, int KP, int WAVES>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
    ecc_sums2_kernel(const float *__restrict__ img, const float *__restrict__ tmpl, int rows, int cols,
                     const EccState *__restrict__ state, double *__restrict__ partial)
{
    const int f = blockIdx.x;
    if (state[f].done) return;
    const unsigned nint = gridDim.y - (unsigned)kEccBorderBlocks;
    if (blockIdx.y >= (unsigned)kEccBorderBlocks)
        ecc_interior_body<IDENT, KP>(img, tmpl, rows, cols, state, partial, f, blockIdx.y - (unsigned)kEccBorderBlocks, nint);
    else
        ecc_border_body(img, tmpl, rows, cols, state, partial, f, blockIdx.y, IDENT ? 1 : 0);
}

// hal::LU32f-based inverse (cv::Mat::inv, DECOMP_LU) of a 6x6 float matrix, same operations in
// the same order.  Everything is unrolled with compile-time indices (the row exchange of the
// partial pivoting is a select over the candidate rows), so both matrices live in registers: the
// one lane that runs this was spending ~25 us per call on dependent scratch / LDS round trips.
__device__ __forceinline__ bool inv6(const float *Ain, float *inv)
{
    float A[6][6], b[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            A[i][j] = Ain[i * 6 + j];
            b[i][j] = i == j ? 1.f : 0.f;
        }
    const float eps = FLT_EPSILON * 10;
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        int k = i;
        float best = fabsf(A[i][i]);
#pragma unroll
        for (int j = i + 1; j < 6; ++j) {
            const float v = fabsf(A[j][i]);
            if (v > best) { best = v; k = j; }
        }
        if (best < eps) ok = false;
#pragma unroll
        for (int r = i + 1; r < 6; ++r) {           // rows i <-> k, k known only at run time
            const bool sw = k == r;
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const float ta = A[i][c], tb = b[i][c];
                A[i][c] = sw ? A[r][c] : ta;
                A[r][c] = sw ? ta : A[r][c];
                b[i][c] = sw ? b[r][c] : tb;
                b[r][c] = sw ? tb : b[r][c];
            }
        }
        const float d = -1 / A[i][i];
#pragma unroll
        for (int j = i + 1; j < 6; ++j) {
            const float alpha = A[j][i] * d;
#pragma unroll
            for (int kk = i + 1; kk < 6; ++kk) A[j][kk] += alpha * A[i][kk];
#pragma unroll
            for (int kk = 0; kk < 6; ++kk) b[j][kk] += alpha * b[i][kk];
        }
    }
    if (!ok) return false;
#pragma unroll
    for (int i = 5; i >= 0; --i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float sacc = b[i][j];
#pragma unroll
            for (int k = i + 1; k < 6; ++k) sacc -= A[i][k] * b[k][j];
            b[i][j] = sacc / A[i][i];
        }
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) inv[i * 6 + j] = b[i][j];
    return true;
}

// One lane per frame: the body of the cv::findTransformECC iteration after the
// image passes (ecc.cpp): meanStdDev, rho, hessian inverse, lambda, deltaP, update.
// One iteration's solve for frame f: wave w reduces sums k = w, w+4, ... over the block partials (lane l takes blocks l, l+64,
// ...; fixed shuffle tree -> deterministic), thread 0 then runs the scalar part.  All 256 threads of the workgroup call it.
// ATOMIC: the partials were written with device-scope atomics by other workgroups of the SAME launch and are read the same way.
template <bool ATOMIC>
__device__ __forceinline__ void ecc_solve_body(EccState &es, const double *__restrict__ partial, int f, int nblocks, int max_iters, double eps,
                               int rows, int cols, double *Ssh)
{
    auto ld = [&](const double *p) -> double {
        if (ATOMIC)
            return __longlong_as_double((long long)atomicAdd(reinterpret_cast<unsigned long long *>(const_cast<double *>(p)), 0ull));
        return *p;
    };
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // every load of the wave's <= 12 sums is issued before the first reduction (one sum after the
    // other made this a chain of 12 global-load latencies: 33 us per launch, 15 % of the ECC path)
    constexpr int kPerWave = (kEccSums + 3) / 4;
    double v[kPerWave];
#pragma unroll
    for (int i = 0; i < kPerWave; ++i) {
        const int k = wave + 4 * i;
        // first block of 64 partials: one independent load per sum
        v[i] = (k < kEccSums && lane < nblocks) ? ld(&partial[((size_t)f * kEccSums + k) * kEccStride + lane]) : 0.0;
    }
    if (nblocks > 64) {   // few active frames -> more, smaller blocks per frame (same order per lane as one loop)
#pragma unroll
        for (int i = 0; i < kPerWave; ++i) {
            const int k = wave + 4 * i;
            if (k < kEccSums) {
                const double *pk = partial + ((size_t)f * kEccSums + k) * kEccStride;
                for (int b = lane + 64; b < nblocks; b += 64) v[i] += ld(&pk[b]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < kPerWave; ++i) {
        double x = v[i];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);   // fixed tree -> deterministic
        const int k = wave + 4 * i;
        if (lane == 0 && k < kEccSums) Ssh[k] = x;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const double *S = Ssh;          // (read from LDS where needed: 90 registers less in the kernels that carry this body)
    const double n = S[0];
    const double mw = n ? S[1] / n : 0, mt = n ? S[3] / n : 0;
    const double vw = n ? S[2] / n - mw * mw : 0, vt = n ? S[4] / n - mt * mt : 0;
    const double sdw = sqrt(vw > 0 ? vw : 0), sdt = sqrt(vt > 0 ? vt : 0);
    const double tmpNorm = sqrt(n * sdt * sdt), imgNorm = sqrt(n * sdw * sdw);
    const double corr = S[5] - n * mt * mw;
    float Hf[36], Hinv[36], ipf[6], tpf[6];
    int h = 24;
    for (int a = 0; a < 6; ++a) {
        ipf[a] = (float)(S[6 + a] - mw * S[12 + a]);
        tpf[a] = (float)(S[18 + a] - mt * S[12 + a]);
        for (int b = a; b < 6; ++b) {
            Hf[a * 6 + b] = Hf[b * 6 + a] = (float)S[h];
            ++h;
        }
    }
    if (!inv6(Hf, Hinv))
        for (int i = 0; i < 36; ++i) Hinv[i] = 0.f;
    es.last_rho = es.rho;
    es.rho = corr / (imgNorm * tmpNorm);
    es.iters += 1;
    if (es.rho != es.rho) {  // "NaN encountered."
        es.done = -1;
        return;
    }
    float iph[6];
    for (int i = 0; i < 6; ++i) {
        double s = 0;
        for (int j = 0; j < 6; ++j) s += (double)Hinv[i * 6 + j] * ipf[j];
        iph[i] = (float)s;
    }
    double d_ip = 0, d_tp = 0;
    for (int i = 0; i < 6; ++i) {
        d_ip += (double)ipf[i] * iph[i];
        d_tp += (double)tpf[i] * iph[i];
    }
    const double lambda_n = imgNorm * imgNorm - d_ip;
    const double lambda_d = corr - d_tp;
    if (lambda_d <= 0.0) {  // "The algorithm stopped before its convergence..."
        es.rho = -1;
        es.done = -2;
        return;
    }
    const float lambda = (float)(lambda_n / lambda_d);
    // errorProjection = J^T (lambda*tz - wz) = lambda*tp - ip
    float epf[6], dp[6];
    for (int i = 0; i < 6; ++i) epf[i] = (float)((double)lambda * tpf[i] - (double)ipf[i]);
    for (int i = 0; i < 6; ++i) {
        double s = 0;
        for (int j = 0; j < 6; ++j) s += (double)Hinv[i * 6 + j] * epf[j];
        dp[i] = (float)s;
    }
    es.M[0] += dp[0]; es.M[3] += dp[1]; es.M[1] += dp[2];
    es.M[4] += dp[3]; es.M[2] += dp[4]; es.M[5] += dp[5];
    es.band = ecc_band(es.M, rows, cols);
    // for (i = 1; i <= N && fabs(rho - last_rho) >= eps; i++)
    if (es.iters >= max_iters || !(fabs(es.rho - es.last_rho) >= eps)) es.done = 1;
}

__global__ void __launch_bounds__(256)
    ecc_solve_kernel(EccState *__restrict__ state, const double *__restrict__ partial,
                     int nframes, int nblocks, int max_iters, double eps, int rows, int cols)
{
    // one workgroup per frame
    const int f = blockIdx.x;
    if (f >= nframes) return;
    EccState &es = state[f];
    if (es.done) return;
    __shared__ double Ssh[kEccSums];
    ecc_solve_body<false>(es, partial, f, nblocks, max_iters, eps, rows, cols, Ssh);
}

__global__ void ecc_init_kernel(EccState *state, int nframes, long long first_frame, double eps)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nframes) return;
    EccState &es = state[f];
    es.M[0] = 1; es.M[1] = 0; es.M[2] = 0; es.M[3] = 0; es.M[4] = 1; es.M[5] = 0;  // eye(2,3)
    es.rho = -1;
    es.last_rho = -eps;
    es.iters = 0;
    es.done = (first_frame + f == 0) ? 2 : 0;  // frame 0 is not registered (psp_process.cpp:1777)
    es.band = 3;                               // identity
}

// one wave: out[0] frames still iterating, [1] frames in error, [2] frame-iterations so far (statistics), [3] iterations of
// the frame that needed most (sizes the next sub-batch's first burst).  Writes all four: nothing to clear beforehand.
__global__ void __launch_bounds__(64) ecc_count_active(const EccState *state, int nframes, int *out)
{
    int active = 0, err = 0, iters = 0, most = 0;
    for (int f = threadIdx.x; f < nframes; f += 64) {
        active += state[f].done == 0;
        err += state[f].done < 0;
        iters += state[f].iters;
        most = max(most, state[f].iters);
    }
    for (int off = 32; off > 0; off >>= 1) {
        active += __shfl_down(active, off);
        err += __shfl_down(err, off);
        iters += __shfl_down(iters, off);
        most = max(most, __shfl_down(most, off));
    }
    if (threadIdx.x == 0) {
        out[0] = active;
        out[1] = err;
        out[2] = iters;
        out[3] = most;
    }
}

__global__ void ecc_export_warps(const EccState *state, int nframes, float *warps, int stride, int32_t *iters, int istride)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nframes) return;
    if (warps)
        for (int i = 0; i < 6; ++i) warps[(size_t)f * stride + i] = state[f].M[i];
    if (iters) iters[(size_t)f * istride] = state[f].iters;
}

// ----------------------------------------------------------------- patches --
// Per cluster: coef = P z (P = pseudo-inverse of the cubic design matrix in centred,
// scaled coordinates, built once on the host in double), then evaluate at the
// interior pixels.  One workgroup per (cluster, frame).
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

}  // namespace

// ------------------------------------------------------------ PatchTables --
struct PatchTables {
    int nclusters = 0;
    bool sequential = false;  // a boundary of one cluster overlaps another's interior
    ClusterDesc *d_desc = nullptr;
    int32_t *d_bidx = nullptr, *d_iidx = nullptr;
    float *d_P = nullptr;
};

void patch_tables_free(PatchTables *t)
{
    if (!t) return;
    if (t->d_desc) (void)hipFree(t->d_desc);
    if (t->d_bidx) (void)hipFree(t->d_bidx);
    if (t->d_iidx) (void)hipFree(t->d_iidx);
    if (t->d_P) (void)hipFree(t->d_P);
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
    std::vector<ClusterDesc> desc(nclusters);
    std::vector<int32_t> bidx(std::max(nbt, 1)), iidx(std::max(nit, 1));
    std::vector<float> Pall((size_t)std::max(nbt, 1) * 10, 0.f);
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
        if (d.nb >= 10) {
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
    hipError_t e = hipMalloc(&t->d_desc, sizeof(ClusterDesc) * std::max(nclusters, 1));
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

// GaussianBlur(u16 -> f32, 5 x 5) of nimg frames by the column-walking kernel; with `state` / `partial` / `tmpl` also the
// identity iteration of the ECC (one partial-sum slot per workgroup: *nslots of them, slots 0 .. *nslots - 1).
static int launch_gauss5_cols(const uint16_t *src, float *dst, int nimg, int rows, int cols, const float *tmpl,
                              const EccState *state, double *partial, int *nslots, hipStream_t st)
{
    FilterCoef fc;
    if (gaussian_coef(5, fc) != 0) return fail(UPSP_ERR_INVALID, "gaussian 5");
    const int ntiles = (cols + kGcOwn - 1) / kGcOwn, zb = (ntiles + 3) / 4;
    int rpp = 64;
    while ((long long)((rows + rpp - 1) / rpp) * zb > kEccStride) rpp *= 2;     // (one slot per workgroup)
    const int pieces = (rows + rpp - 1) / rpp;
    if (nimg > 65535 || pieces > 65535 || zb > 65535) return fail(UPSP_ERR_INVALID, "gaussian: image too large");
    const dim3 grid((unsigned)nimg, (unsigned)pieces, (unsigned)zb);
    const int uvar = env_int_io("UPSP_GAUSS5_VARIANT", 0);       // (measurement switch)
    if (state) {
        KTimed kt("gauss5_ecc0_kernel", st);
#define UPSP_G5E(UU, WV)                                                                                        \
    hipLaunchKernelGGL((gauss5_ecc0_kernel<UU, WV>), grid, dim3(256), 0, st, src, dst, tmpl, rows, cols, rpp, fc.k[2], fc.k[3], \
                       fc.k[4], state, partial, 0u)
        if (uvar == 1) UPSP_G5E(8, 3); else if (uvar == 2) UPSP_G5E(8, 4); else if (uvar == 3) UPSP_G5E(2, 4); else UPSP_G5E(4, 4);
#undef UPSP_G5E
        if (nslots) *nslots = pieces * zb;
    } else {
        KTimed kt("gauss_pass_kernels", st);
#define UPSP_G5B(UU)                                                                                            \
    hipLaunchKernelGGL((gauss5_blur_kernel<UU>), grid, dim3(256), 0, st, src, dst, rows, cols, rpp, fc.k[2], fc.k[3], fc.k[4])
        if (uvar == 1) UPSP_G5B(8); else if (uvar == 2) UPSP_G5B(16); else if (uvar == 3) UPSP_G5B(2); else UPSP_G5B(4);
#undef UPSP_G5B
    }
    UPSP_HIP_CHECK(hipGetLastError());
    return UPSP_OK;
}

// ----------------------------------------------------------- FrameScratch --
struct FrameScratch {
    int ncams = 0, batch = 0, rows = 0, cols = 0;
    uint16_t *warp[kMaxCams] = {nullptr};   // registered u16 frames
    float *f32[kMaxCams] = {nullptr};       // patched / filtered frames
    float *f32b[kMaxCams] = {nullptr};      // Gaussian-filtered frames when the filter input is f32[] itself
    float *ecc_img = nullptr;               // blurred input frames
    float *ecc_img2 = nullptr;              // second buffer: the pre-blur of the NEXT sub-batch runs ahead on another stream (frame_scratch_preblur)
    float *tmp = nullptr;                   // filter intermediate (float), also box sums (double)
    float *tmpl[kMaxCams] = {nullptr};      // blurred ECC templates
    const float *tmpl_src[kMaxCams] = {nullptr};
    float *center = nullptr;                // [kMaxCams] centre of the float products of the ECC sums (ecc_center_kernel)
    unsigned *hot_flag = nullptr;           // [batch] frames in which the pre-blur saw a pixel >= the hot threshold (HotFuse)
    int *h_counter = nullptr;               // pinned: where "frames still iterating" is read back to
    hipEvent_t ev_counter = nullptr;        // ... and the event behind that copy
    unsigned *tickets = nullptr;            // [batch][kEccTicketStride] blocks of a frame that have delivered their sums (fused solve); zero between launches
    double *partial = nullptr;
    EccState *state = nullptr;
    int *counter = nullptr;
    unsigned long long ecc_frame_iters = 0, ecc_frames = 0;   // statistics: ECC iterations summed over frames, frames
    int ecc_first_burst = 3;                                  // iterations issued before the first host check
};

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
    if (s->hot_flag) (void)hipFree(s->hot_flag);
    if (s->tickets) (void)hipFree(s->tickets);
    if (s->h_counter) (void)hipHostFree(s->h_counter);
    if (s->ev_counter) (void)hipEventDestroy(s->ev_counter);
    if (s->tmp) (void)hipFree(s->tmp);
    if (s->partial) (void)hipFree(s->partial);
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
    if (need_warp && !s->hot_flag) UPSP_HIP_CHECK(hipMalloc(&s->hot_flag, sizeof(unsigned) * (size_t)batch));
    if (need_warp && !s->h_counter) {
        UPSP_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&s->h_counter), 4 * sizeof(int), hipHostMallocDefault));
        UPSP_HIP_CHECK(hipEventCreateWithFlags(&s->ev_counter, hipEventDisableTiming));
    }
    if (need_warp && !s->tickets) {
        UPSP_HIP_CHECK(hipMalloc(&s->tickets, sizeof(unsigned) * (size_t)batch * kEccTicketStride));
        UPSP_HIP_CHECK(hipMemset(s->tickets, 0, sizeof(unsigned) * (size_t)batch * kEccTicketStride));
    }
    if (need_warp && !s->center) {
        UPSP_HIP_CHECK(hipMalloc(&s->center, kMaxCams * sizeof(float)));
        UPSP_HIP_CHECK(hipMemset(s->center, 0, kMaxCams * sizeof(float)));
    }
    if (!s->tmp) UPSP_HIP_CHECK(hipMalloc(&s->tmp, n * sizeof(double)));
    if (need_warp && !s->partial)
        UPSP_HIP_CHECK(hipMalloc(&s->partial, sizeof(double) * (size_t)batch * kEccStride * kEccSums));
    if (!s->state) UPSP_HIP_CHECK(hipMalloc(&s->state, sizeof(EccState) * (size_t)batch));
    if (!s->counter) UPSP_HIP_CHECK(hipMalloc(&s->counter, 4 * sizeof(int)));
    return UPSP_OK;
}

// ECC registration of nb frames against the blurred template; leaves the warp in state[].
static int run_ecc(FrameScratch *s, const float *tmpl_blur, const float *d_center, const uint16_t *frames, int nb,
                   int64_t first_frame, int rows, int cols, int max_iters, double eps, hipStream_t st,
                   const float *preblurred = nullptr, const HotFuse *hot = nullptr, uint16_t *frames_rw = nullptr,
                   const std::function<int()> *while_waiting = nullptr)
{
    if ((long long)rows * cols >= (1ll << 31)) return fail(UPSP_ERR_INVALID, "registration: image too large");
    // Pre-blur (GaussianBlur 5 x 5) and the first iteration.  Every frame starts from the identity warp
    // (cpp/lib/registration.cpp:52-53), so the first iteration needs no warp.  Default: the tile kernel blurs, the
    // column kernel's identity variant takes the sums (134 + 130 us per 64 frames of 1024^2).  UPSP_ECC_FUSED=1: ONE
    // column-walking kernel blurs and sums (gauss5_ecc0_kernel) -- built because it moves 10 instead of 14 bytes per
    // pixel, measured at 280-290 us: a wave's 128-byte requests for its 62 u16 pixels of a row keep the L1 waiting on
    // pending misses (PMC: 68 % of the time; 4 / 8 / 16 rows in flight, 3 / 4 / 8 waves per SIMD and streamed loads all
    // within 280-365 us).  Kept as a switch, parity-tested; UPSP_ECC_FUSED=2: column-walking blur alone (184 us) + the
    // identity variant.
    const int fused_env = preblurred ? 0 : env_int_io("UPSP_ECC_FUSED", 0);
    const bool fused = fused_env == 1 && rows >= 5 && cols >= 5;
    const float *blurred = preblurred ? preblurred : s->ecc_img;      // (GaussianBlur 5 x 5 of the frames)
    const dim3 g1((nb + 63) / 64), b1(64);
    hipLaunchKernelGGL(ecc_init_kernel, g1, b1, 0, st, s->state, nb, (long long)first_frame, eps);
    int it = 0;
    if (fused) {
        int nslots = 0;
        int rc = launch_gauss5_cols(frames, s->ecc_img, nb, rows, cols, tmpl_blur, (const EccState *)s->state, s->partial, &nslots, st);
        if (rc != UPSP_OK) return rc;
        KTimed kt2("ecc_solve_kernel", st);
        hipLaunchKernelGGL(ecc_solve_kernel, dim3(nb), dim3(256), 0, st, s->state, (const double *)s->partial, nb, nslots,
                           max_iters, eps, rows, cols);
        it = 1;
    } else if (!preblurred && hot) {
        // fix_hot_pixels inside the pre-blur: blur + flag, repair the flagged frames (normally none), blur those again
        FilterCoef fc;
        if (gaussian_coef(5, fc) != 0) return fail(UPSP_ERR_INVALID, "gaussian 5");
        UPSP_HIP_CHECK(hipMemsetAsync(s->hot_flag, 0, sizeof(unsigned) * (size_t)nb, st));
        {
            KTimed kt("gauss_pass_kernels", st);
            if (!launch_gauss5_quad(frames, s->ecc_img, nb, rows, cols, fc, st, (unsigned)hot->thresh, s->hot_flag, 0))
                return fail(UPSP_ERR_INTERNAL, "pre-blur with the hot-pixel scan: geometry not supported (frame_stages_fuse_hot)");
        }
        int rc = launch_hot_fix(frames_rw, nb, rows, cols, hot->thresh, hot->min_change, hot->max_hot, hot->d_count, hot->d_pos, nullptr,
                                st, s->hot_flag);
        if (rc != UPSP_OK) return rc;
        KTimed kt("gauss_pass_kernels", st);
        (void)launch_gauss5_quad(frames, s->ecc_img, nb, rows, cols, fc, st, 0u, s->hot_flag, 1);
    } else if (!preblurred) {
        int rc = (fused_env == 2 && rows >= 5 && cols >= 5)
                     ? launch_gauss5_cols(frames, s->ecc_img, nb, rows, cols, nullptr, nullptr, nullptr, nullptr, st)
                     : launch_gauss<uint16_t>(frames, s->ecc_img, s->tmp, nb, rows, cols, 5, st);
        if (rc != UPSP_OK) return rc;
    }
    bool first_burst = true, waited = false;
    int active = nb;  // frames still iterating (known to the host after every burst)
    int iters_done = 0, most_iters = 0;
    for (;;) {
        // a few iterations between host checks of the active-frame count; frames that have
        // converged exit at once, so late bursts spread the remaining frames over more blocks
        // (a host check costs a stream round trip of ~40 us; most frames converge within 3-5
        // iterations, the rare oscillating ones run to max_iters, so the bursts grow)
        // (first burst: as many iterations as the previous sub-batch's slowest frame took -- on steady footage every
        //  frame converges with its second iteration, and a third launch pair that finds nothing to do costs 10 us)
        const int burst = first_burst ? std::max(1, s->ecc_first_burst - it) : (it < 7 ? 2 : (it < 15 ? 8 : 16));
        first_burst = false;
        // Interior blocks per frame.  Round 3's kernel: every block ends with a reduction of the 45 sums that costs as
        // much as ~10 rows of its 256 columns, so a block should walk a few hundred rows -- 16 blocks per frame (4 column
        // tiles x 4 row pieces of 256 at 1024^2: 1024 blocks for a full sub-batch = one resident set at 4 per CU), more
        // when few frames are still iterating; at least one per column tile of 256.
        static const int kernel_sel = env_int_io("UPSP_ECC_KERNEL", 3);
        const int tiles = (cols + 255) / 256;
        const bool cols_ok = kernel_sel == 3 && 3 * tiles <= kEccBorderBlocks && tiles <= kEccBlocksMax;
        static const int blocks0 = std::min(std::max(env_int_io("UPSP_ECC_BLOCKS", 64), 1), kEccBlocksMax);     // (measured 16 / 32 / 64: 6.74 / 6.25 / 6.03 ms of sums per 1000 frames)
        int blocks = cols_ok ? blocks0 : kEccBlocks;
        // The column kernel's float segments follow the row pieces, i.e. the block count: it must not depend on how many
        // frames are still iterating, or the last bits of a frame's sums -- and through the reference's float 6 x 6 solve
        // 1e-5 .. 1e-4 px of its warp -- would depend on which frames share its sub-batch (measured: 2.6e-4 px between a
        // frame registered in a batch of 11 and in a batch of 7).  One count per image geometry: the warp of a frame is
        // the same bits in any batch (tests/test_imageops_gpu.py).  UPSP_ECC_BLOCKS_GROW=1: round 2's policy for the
        // column kernel too (more, smaller blocks when few frames are left: faster tails, batch-dependent bits).
        static const int blocks_grow = env_int_io("UPSP_ECC_BLOCKS_GROW", 0);
        const bool grow = !cols_ok || blocks_grow;
        while (blocks < kEccBlocksMax && ((grow && (long long)blocks * active < (cols_ok ? 1024 : 2048)) ||
                                          (cols_ok && (blocks < tiles || rows > (long long)kEccRowTab * (blocks / tiles)))))
            blocks *= 2;
        const int nblocks_total = blocks + kEccBorderBlocks;
        for (int k = 0; k < burst && it < max_iters; ++k, ++it) {
            bool fused_solve = false;
            {
                KTimed kt("ecc_sums_kernel", st);
#define UPSP_ECC_LAUNCH(ID, KPX, WV)                                                                          \
    hipLaunchKernelGGL((ecc_sums2_kernel<ID, KPX, WV>), dim3(nb, blocks + kEccBorderBlocks), dim3(256), 0, st,    \
                       blurred, tmpl_blur, rows, cols, (const EccState *)s->state, s->partial)
#define UPSP_ECC_COLS(ID, URX, WV, ...)                                                                       \
    hipLaunchKernelGGL((ecc_cols_kernel<ID, URX, WV, ##__VA_ARGS__>), dim3(nb, blocks + kEccBorderBlocks), dim3(256), 0, st, \
                       blurred, tmpl_blur, rows, cols, s->state, s->partial, d_center, s->tickets, max_iters, eps)
                // Round 3: one column per thread, factored packed-float sums (ecc_cols_kernel); needs one column tile of 256 per
                // interior block at least.  UPSP_ECC_KERNEL=2 selects round 2's kernel (A/B, and images wider than that).
                static const int cvariant = env_int_io("UPSP_ECC_CVARIANT", 0);
                static const int gx_form = env_int_io("UPSP_ECC_GX", 1);
                // UPSP_ECC_SHARE_ROWS=1 (opt-in, measured and NOT the default): the second row of a general trip re-uses the first one's
                // source rows when its footprint is that one moved down by a row in every lane of the wave (ecc_cols_trip): 11
                // instead of 16 source loads per trip, the same bits (tests/test_imageops_gpu.py) -- and 6.0 instead of 5.4 ms of
                // sums per 1000 frames: the loads then wait for both rows' coordinates and a wave-wide vote.  The kernel is not
                // bound by the number of its L1 requests.  = 2: the rows software-pipelined instead (ecc_rows_pipelined: the loads of
                // rows y + 1, y + 2 in flight while row y is summed; 160 VGPRs, no spill, same bits): 6.03 ms.  Two rows loaded, then
                // two rows summed, three waves per SIMD taking turns, remains the fastest arrangement found.
                const int share_rows = env_int_io("UPSP_ECC_SHARE_ROWS", 0);
                // UPSP_ECC_FUSE_SOLVE=1 (opt-in, measured and NOT the default): the solve in the sums launch, by the block that
                // finishes a frame last.  Same bits (tests/test_imageops_gpu.py), but the sums take 6.13 instead of 5.41 + 0.52 ms
                // per 1000 frames: the 45 partial sums of every block leave as device-scope atomics, and the launch ends with the
                // same one-thread chain the solve kernel is -- all 64 frames finish together, so nothing hides it.
                const int fuse_solve = env_int_io("UPSP_ECC_FUSE_SOLVE", 0);
                fused_solve = false;
                const bool use_cols = cols_ok && blocks >= tiles && rows <= (long long)kEccRowTab * (blocks / tiles) &&
                                      rows < 32768 && cols < 32768;
                // pixels per thread and trip / waves per SIMD of round 2's kernel, measured on 1000 frames of 1024^2
                // (tools/exp_ecc.sh; ms of the sums per step): general iteration 2 px at 3 waves per SIMD (162 VGPRs) 8.7; at 2
                // waves: 1 px 12.2, 2 px 10.1, 3 px 10.5, 4 px 9.6; 1 px at 3 waves 9.9, 3 px at 3 waves (10 spilled registers)
                // 10.0; identity iteration 4 px at 3 waves 9.9 (with the default general form), 8 px 9.6, 8 px at 2 waves 10.4.
                // UPSP_ECC_VARIANT / UPSP_ECC_IVARIANT select the others.
                static const int variant = env_int_io("UPSP_ECC_VARIANT", 4);
                static const int ivariant = env_int_io("UPSP_ECC_IVARIANT", 1);
                if (use_cols) {
                    // UPSP_ECC_CVARIANT = 10 x identity variant + general variant (measurement switch)
                    const int iv = cvariant / 10, gv = cvariant % 10;
                    if (it == 0) {
                        // (measured, us per 64-frame launch on one box: plain 4 rows per trip 172, DPP taps 4 rows 163, 8 rows at
                        //  3 waves per SIMD 191, 8 rows at 4 waves -- 82 spilled registers -- 390)
                        if (iv == 1) UPSP_ECC_COLS(true, 8, 4, 1);
                        else if (iv == 2) UPSP_ECC_COLS(true, 4, 4);
                        else if (iv == 3) UPSP_ECC_COLS(true, 8, 3, 1);
                        else if (fuse_solve) { UPSP_ECC_COLS(true, 4, 4, 1, 1, 1); fused_solve = true; }
                        else UPSP_ECC_COLS(true, 4, 4, 1);
                    } else {
                        // (general iteration, same box: 2 rows per trip at 4 waves per SIMD 316, 4 rows at 3 waves 306, 3 at 3: 313,
                        //  2 at 3: 330 -- flat: neither rows in flight nor occupancy is what bounds it)
                        // with gx taken from pixel differences (GXD, the default since the ECC soak): 2 rows at 4 waves -- 4 spilled
                        // registers -- 6.13 ms of sums per 1000 frames, at 3 waves (146 VGPRs) 5.34, 4 rows at 3 waves 6.38;
                        // the first gx form (UPSP_ECC_GX=0) at 4 waves 5.29
                        if (gv == 1) UPSP_ECC_COLS(false, 4, 3);
                        else if (gv == 3) UPSP_ECC_COLS(false, 3, 3);
                        else if (gv == 5) UPSP_ECC_COLS(false, 2, 4);
                        else if (gv == 6) UPSP_ECC_COLS(false, 1, 4);
                        else if (gv == 7) UPSP_ECC_COLS(false, 1, 5);
                        else if (gx_form == 0) UPSP_ECC_COLS(false, 2, 4, 0, 0);
                        else if (fuse_solve) { UPSP_ECC_COLS(false, 2, 3, 0, 1, 1); fused_solve = true; }
                        else if (share_rows == 2) UPSP_ECC_COLS(false, 2, 3, 0, 1, 0, 2);      // rows software-pipelined, two in flight
                        else if (share_rows) UPSP_ECC_COLS(false, 2, 3, 0, 1, 0, 1);
                        else UPSP_ECC_COLS(false, 2, 3);
                    }
                }
                else if (it == 0) {
                    if (ivariant == 1) UPSP_ECC_LAUNCH(true, 2, 3);
                    else if (ivariant == 2) UPSP_ECC_LAUNCH(true, 2, 2);
                    else UPSP_ECC_LAUNCH(true, 1, 3);
                }
                else if (variant == 1) UPSP_ECC_LAUNCH(false, 3, 2);
                else if (variant == 2) UPSP_ECC_LAUNCH(false, 4, 2);
                else if (variant == 3) UPSP_ECC_LAUNCH(false, 1, 2);
                else if (variant == 4) UPSP_ECC_LAUNCH(false, 2, 3);
                else if (variant == 5) UPSP_ECC_LAUNCH(false, 1, 3);
                else if (variant == 6) UPSP_ECC_LAUNCH(false, 3, 3);
                else UPSP_ECC_LAUNCH(false, 2, 2);
#undef UPSP_ECC_COLS
#undef UPSP_ECC_LAUNCH
            }
            static const int dump = env_int_io("UPSP_ECC_DUMP", -1);      // debug: the 45 sums of frame 0 at iteration `dump`
            if (dump == it) {
                std::vector<double> h((size_t)kEccSums * kEccStride);
                UPSP_HIP_CHECK(hipStreamSynchronize(st));
                UPSP_HIP_CHECK(hipMemcpy(h.data(), s->partial, sizeof(double) * h.size(), hipMemcpyDeviceToHost));
                for (int k = 0; k < kEccSums; ++k) {
                    long double a = 0;
                    for (int b = 0; b < nblocks_total; ++b) a += h[(size_t)k * kEccStride + b];
                    std::fprintf(stderr, "[upsp] ecc sum %2d = %.17Lg\n", k, a);
                }
            }
            if (!fused_solve) {
                KTimed kt2("ecc_solve_kernel", st);
                hipLaunchKernelGGL(ecc_solve_kernel, dim3(nb), dim3(256), 0, st, s->state,
                                   (const double *)s->partial, nb, nblocks_total, max_iters, eps, rows, cols);
            }
        }
        int h[4] = {0, 0, 0, 0};
        hipLaunchKernelGGL(ecc_count_active, dim3(1), dim3(64), 0, st, (const EccState *)s->state, nb,
                           s->counter);
        // the read-back goes to pinned memory behind an event: whatever `while_waiting` enqueues (the next sub-batch's
        // hot-pixel repair and pre-blur) runs on the GPU while the host waits for these four words
        UPSP_HIP_CHECK(hipMemcpyAsync(s->h_counter, s->counter, sizeof(h), hipMemcpyDeviceToHost, st));
        UPSP_HIP_CHECK(hipEventRecord(s->ev_counter, st));
        if (while_waiting && !waited) {
            waited = true;
            const int rcw = (*while_waiting)();
            if (rcw != UPSP_OK) return rcw;
        }
        UPSP_HIP_CHECK(hipEventSynchronize(s->ev_counter));
        for (int i = 0; i < 4; ++i) h[i] = s->h_counter[i];
        if (h[1] > 0)
            return fail(UPSP_ERR_DIVERGED,
                        "ECC registration did not converge (cv::findTransformECC would throw)");
        iters_done = h[2];
        most_iters = h[3];
        if (h[0] == 0 || it >= max_iters) break;
        active = h[0];
    }
    s->ecc_first_burst = std::min(std::max(most_iters, 2), 4);
    s->ecc_frame_iters += (unsigned long long)iters_done;
    s->ecc_frames += (unsigned long long)nb;
    UPSP_HIP_CHECK(hipGetLastError());
    if (std::getenv("UPSP_TRACE_ECC")) {
        std::vector<EccState> h(nb);
        UPSP_HIP_CHECK(hipMemcpy(h.data(), s->state, sizeof(EccState) * nb, hipMemcpyDeviceToHost));
        int tot = 0, mx = 0;
        for (int i = 0; i < nb; ++i) {
            const auto &e = h[i];
            tot += e.iters; mx = std::max(mx, e.iters);
            if (e.iters > 8)
                std::fprintf(stderr, "[upsp]   frame %lld: %d iters rho=%.9f last=%.9f M=[%g %g %g; %g %g %g]\n",
                             (long long)first_frame + i, e.iters, e.rho, e.last_rho, e.M[0], e.M[1], e.M[2], e.M[3], e.M[4], e.M[5]);
        }
        std::fprintf(stderr, "[upsp] ECC sub-batch of %d frames: %d frame-iterations, max %d, %d launches\n",
                     nb, tot, mx, it);
    }
    return UPSP_OK;
}

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

// GaussianBlur 5 x 5 of nb frames (the ECC's pre-blur, cpp/lib/registration.cpp:57-60) into one of the scratch's two blurred-frame
// buffers, on any stream: the streamed registration loop runs it for sub-batch k + 1 while sub-batch k iterates -- a
// memory-bound kernel beside the issue-bound ECC sums, and work for the GPU while the host reads "frames still
// iterating" back.  *out = the buffer to hand to run_frame_stages.
bool frame_stages_fuse_hot(const uint16_t *d_frames, int rows, int cols, const upsp_pipeline_opts &opts)
{
    // (the scratch's blurred-frame buffer comes from hipMalloc: 256-byte aligned)
    // Opt-in (UPSP_HOT_IN_BLUR=1), measured and NOT the default: it saves the 27-us scan per sub-batch and pays it back with
    // what replaces it -- a memset of the flags, a scan launch whose blocks all exit and a second blur launch whose blocks
    // all exit: 9.5-9.6 against 9.38 ms per 1000 frames.
    return opts.registration && !std::getenv("UPSP_ECC_FUSED") && env_int_io("UPSP_HOT_IN_BLUR", 0) &&
           gauss5_quad_applies(d_frames, nullptr, 1, rows, cols);
}

int frame_scratch_preblur(FrameScratch *s, int slot, const uint16_t *d_frames, int nb, int rows, int cols, hipStream_t st,
                          const float **out)
{
    if (!s || !s->ecc_img || nb > s->batch) return fail(UPSP_ERR_INVALID, "pre-blur: scratch not set up");
    if (slot && !s->ecc_img2) UPSP_HIP_CHECK(hipMalloc(&s->ecc_img2, (size_t)s->batch * rows * cols * sizeof(float)));
    float *dst = slot ? s->ecc_img2 : s->ecc_img;
    // (the 5 x 5 tile kernel: no intermediate image, nothing shared with the other stream)
    if (rows <= 2 || cols <= 2) return fail(UPSP_ERR_INVALID, "pre-blur: image too small for the tile kernel");
    int rc = launch_gauss<uint16_t>(d_frames, dst, nullptr, nb, rows, cols, 5, st);
    if (rc == UPSP_OK) *out = dst;
    return rc;
}

int run_frame_stages(FrameScratch *s, int cam, const uint16_t *d_frames, int nb, int64_t first_frame,
                     int rows, int cols, const upsp_pipeline_opts &opts, const float *d_ref,
                     const PatchTables *patches, float *d_warps, int32_t *d_iters, int ncams,
                     const unsigned *d_read_list, const WarpCompact *wc, const void **img_out, int *is_f32_out,
                     hipStream_t st, const float *preblurred, const HotFuse *hot, const std::function<int()> *while_waiting)
{
    const size_t npix = (size_t)rows * cols;
    const uint16_t *cur = d_frames;
    const dim3 pgrid(grid_for_pixels(npix), (unsigned)nb), block(256);
    if (opts.registration) {
        if (s->tmpl_src[cam] != d_ref) {  // blurred template, once per reference image
            int rc = launch_gauss<float>(d_ref, s->tmpl[cam], s->tmp, 1, rows, cols, 5, st);
            if (rc != UPSP_OK) return rc;
            hipLaunchKernelGGL(ecc_center_kernel, dim3(1), dim3(256), 0, st, (const float *)s->tmpl[cam], rows, cols, s->center + cam);
            s->tmpl_src[cam] = d_ref;
        }
        int rc = run_ecc(s, s->tmpl[cam], s->center + cam, d_frames, nb, first_frame, rows, cols, opts.ecc_max_iters,
                         opts.ecc_eps, st, preblurred, hot, const_cast<uint16_t *>(d_frames), while_waiting);
        if (rc != UPSP_OK) return rc;
        if (wc) {      // registration is the last image stage and node-major series are wanted: straight into the compact buffer
            KTimed kt("warp_u16_kernel", st);
            hipLaunchKernelGGL(warp_compact_kernel, dim3((unsigned)((wc->max_active + 63) / 64)), block, 0, st, d_frames, rows,
                               cols, (const EccState *)s->state, nb, opts.interp, wc->pix_of_k, wc->nact, wc->compact,
                               wc->cpitch, wc->col0);
        } else {
            KTimed kt("warp_u16_kernel", st);
            const bool listed = !opts.patch && !opts.filter && d_read_list;
            // (the list is short -- its length is only known on the device: 64 workgroups per frame stride over it)
            const dim3 wgrid(listed ? 64u : pgrid.x, (unsigned)nb);
            hipLaunchKernelGGL(warp_u16_kernel, wgrid, block, 0, st, d_frames, s->warp[cam], rows, cols,
                               (const EccState *)s->state, opts.interp, listed ? d_read_list : (const unsigned *)nullptr);
        }
        if (d_warps || d_iters)
            hipLaunchKernelGGL(ecc_export_warps, dim3((nb + 63) / 64), dim3(64), 0, st,
                               (const EccState *)s->state, nb, d_warps ? d_warps + (size_t)cam * 6 : (float *)nullptr,
                               ncams * 6, d_iters ? d_iters + cam : (int32_t *)nullptr, ncams);
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
    if (rc == UPSP_OK)
        hipLaunchKernelGGL(ecc_center_kernel, dim3(1), dim3(256), 0, st, (const float *)s->tmpl[0], rows, cols, s->center);
    // first_frame = 1: a stand-alone call always registers (psp_process.cpp:1662-1679)
    if (rc == UPSP_OK) rc = run_ecc(s, s->tmpl[0], s->center, d_inp, 1, 1, rows, cols, max_iters, eps, st);
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
    // (UPSP_GAUSS5_COLS=1: the column-walking 5 x 5 kernel instead of the tile kernel -- measurement / test switch)
    if (k == 5 && rows >= 5 && cols >= 5 && (long long)rows * cols < (1ll << 29) && env_int_io("UPSP_GAUSS5_COLS", 0))
        return launch_gauss5_cols(d_src, d_dst, nimg, rows, cols, nullptr, nullptr, nullptr, nullptr, st);
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
    if (!d_img || rows <= 0 || cols <= 0) return fail(UPSP_ERR_INVALID, "bad argument");
    if (nclusters == 0) return UPSP_OK;
    PatchTables *t = nullptr;
    int rc = patch_tables_create(rows, cols, nclusters, h_b_off, h_bx, h_by, h_i_off, h_ix, h_iy, &t);
    if (rc != UPSP_OK) return rc;
    rc = launch_patch(t, d_img, 1, rows, cols, (hipStream_t)stream);
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    patch_tables_free(t);
    if (rc == UPSP_OK && e != hipSuccess) rc = fail(UPSP_ERR_HIP, hipGetErrorString(e));
    return rc;
}

}  // extern "C"
