// ECC affine registration of the psp_process frame loop on MI355X (gfx950): cv::findTransformECC as called by
// upsp::register_pixel, cpp/lib/registration.cpp:32-81 (MOTION_AFFINE, COUNT + EPS, gaussFiltSize 5).
//
// The arithmetic lives in OpenCV 4.5.2 in the reference (un-vendored, no reference test: PARITY UNPINNED); the kernels
// follow the published algorithm as restated in oracle/image_oracle.c.
//
// All frames of a sub-batch iterate in lock step.  One iteration is ONE pass over the template-sized pixel grid per
// frame: the warped image, the two warped gradients (central differences recomputed from the blurred frame on the fly,
// never stored) and the nearest-neighbour mask are evaluated per pixel and folded into 45 sums; everything OpenCV
// derives from zero-mean images (correlation, Hessian, projections, lambda, the parameter step) follows from those
// sums algebraically, so no second pass and no Jacobian planes exist.  The sums are reduced deterministically (fixed
// block partials, fixed order) and one workgroup per frame does the 6 x 6 float LU solve exactly like cv::Mat::inv.
//
// Sum slots:
//  0 n   1 Sw   2 Sww   3 St   4 Stt   5 Stw          (masked)
//  6..11  S_all  J_k * w        12..17 S_mask J_k      18..23 S_mask J_k * t
//  24..44 S_all  J_a * J_b  (a <= b, row-major upper triangle),   J = [gx X, gy X, gx Y, gy Y, gx, gy]
//
// The pixels are split by WHERE they are (one launch, grid = frames x (band blocks + interior blocks), the frame the
// fast index so that the band blocks of all frames are dispatched first and run beside the interior ones):
//   * interior blocks: pixels farther than EccState::band from every edge -- the whole 12-pixel footprint of the
//     bilinear taps and their gradient taps is inside the image and the nearest-neighbour mask is 1 by construction
//     (ecc_band), so the loop has no test and no fallback;
//   * band blocks: 2 x band rows + 2 x band columns (~1.5 % of a 1024^2 frame), generic bilinear with border handling.
//
// Interior, one COLUMN per thread.  The 45 sums are products of three things: the warped gradients {gx, gy}, the pixel
// coordinates {X, Y, 1} and {w, 1, t} (or a second gradient).  A thread that owns ONE column x and walks down its rows
// has a constant X, so X comes out of every sum and is multiplied in once, at the end; Y is the row offset r inside a
// SEGMENT of kEccFlush rows (a small exact integer), shifted to the true row when the segment's partial sums are folded
// into the thread's double totals.  What is left per pixel are 21 sums
//     {gx, gy} x {1, w, t} x {1, r}        12        {gx^2, gy^2, gx gy} x {1, r, r^2}     9
// accumulated as packed float pairs (v_pk_fma_f32) over at most kEccFlush rows, + the five scalar sums of w and t in
// double (they decide rho, i.e. the iteration count).  Float partials: a segment sum of <= 32 terms carries a relative
// error of <~1e-7, random over the 30 000 segments of a frame -- the same order as the float rounding of every Jacobian
// element in cv::findTransformECC itself, and five orders below the 1e-4 parity bar (tests: same iteration counts,
// |dM| ~ 1e-7).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "imageops.h"
#include "ktimer.h"
#include "upsp_internal.h"

namespace upsp {
namespace {

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

typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int kEccFlush = 32;        // rows per float segment (band blocks; interior blocks when UPSP_ECC_ONE_FLUSH=0 or past kEccInteriorMax)
constexpr int kEccFlushLong = 128;   // ... of the interior blocks' ONE-FLUSH form (round 6): the whole row piece of a block is one float
                                     // segment, so the thread's 27 double totals (54 registers) exist only behind the loop

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

__device__ __forceinline__ void ecc_tot_zero(EccTot &T)
{
#pragma unroll
    for (int k = 0; k < 2; ++k)
        T.G0[k] = T.G1[k] = T.Gw0[k] = T.Gw1[k] = T.Gt0[k] = T.Gt1[k] = T.Q0[k] = T.Q1[k] = T.Q2[k] = 0.0;
    T.C0 = T.C1 = T.C2 = T.Sw = T.Sww = T.St = T.Stt = T.Stw = T.n = 0.0;
    T.cf = 0.f;
}

// one pixel: warped value w, warped gradients gx / gy, template t, row offset rf (= r as a float) in its segment.
// MASKED (band pixels): m = the nearest-neighbour mask of cv::findTransformECC; masked sums (n, the scalar sums,
// sum_mask J, sum_mask J t), the others over all pixels.
// Interior pixels: the products with w and t are taken with (w - c), (t - c), c = T.cf = an INTEGER near the template's mean
// (ecc_center_kernel), and c x (sum of the gradients) is added back in double at the end (ecc_tot_value).  The
// subtraction is exact (12-bit images blurred to floats below 4096 minus an integer below 4096), the identity
// sum g w = sum g (w - c) + c sum g too; what changes is the size of the numbers that get rounded: a product g w with
// w ~ 1800 carries an absolute rounding error of |g| x 1e-4, the same product with |w - c| ~ 100 a tenth of that --
// and what the solve uses is sum J w - mean(w) sum J, a difference that used to cancel the leading 1-2 digits of these
// sums (the reference rounds every one of these sums to FLOAT before its 6 x 6 solve, so a sum that lands on another
// float moves the result by a float ulp amplified by the solve).
// TSUMS = false: the caller adds the template's own sums (St, Stt: constants of the template when every pixel counts) itself.
template <bool MASKED, bool TSUMS = true>
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
    if (TSUMS) {
        T.St += tm;
        T.Stt = fma(tm, td, T.Stt);
    }
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

// the k-th of the 45 sums from a thread's totals and its column X
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
// Fixed order: deterministic.
constexpr int kEccChunk = 15;
static_assert(kEccSums == 3 * kEccChunk, "three chunks");
// CH values per round (kEccChunk, or 5 for a kernel that wants its LDS for occupancy: 10 KB instead of 30)
template <int CH, int C, int J>
__device__ __forceinline__ void ecc_tot_put(const EccTot &T, double X, bool on, double (*lds)[256])
{
    lds[J][threadIdx.x] = on ? ecc_tot_value<C * CH + J>(T, X) : 0.0;
    if constexpr (J + 1 < CH) ecc_tot_put<CH, C, J + 1>(T, X, on, lds);
}
template <int C, int CH = kEccChunk>
__device__ __forceinline__ void ecc_tot_store(const EccTot &T, double X, bool on, double (*lds)[256], double *__restrict__ partial,
                                              int f, unsigned slot, unsigned stride = (unsigned)kEccStride)
{
    static_assert(kEccSums % CH == 0 && CH <= 16, "whole rounds");
    ecc_tot_put<CH, C, 0>(T, X, on, lds);
    __syncthreads();
    const int v = (int)threadIdx.x >> 4, p = (int)threadIdx.x & 15;
    if (v < CH) {                              // (whole DPP rows)
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) s += lds[v][j * 16 + p];
        s = dpp_add_f64<0xB1, 0xF>(s);
        s = dpp_add_f64<0x4E, 0xF>(s);
        s = dpp_add_f64<0x141, 0xF>(s);
        s = dpp_add_f64<0x140, 0xF>(s);        // every lane of the row holds the block's sum of value v
        if (p == 0) partial[((size_t)f * kEccSums + (C * CH + v)) * stride + slot] = s;
    }
    __syncthreads();
    if constexpr ((C + 1) * CH < kEccSums) ecc_tot_store<C + 1, CH>(T, X, on, lds, partial, f, slot, stride);
}

// a block without any pixel: its partial sums are zero
__device__ __forceinline__ void ecc_store_zeros(double *__restrict__ partial, int f, unsigned slot)
{
    if (threadIdx.x < kEccSums) partial[((size_t)f * kEccSums + threadIdx.x) * kEccStride + slot] = 0.0;
}

// uniform base + 32-bit BYTE offset of the lane (+ a constant): the form the compiler turns into
// `global_load_dword v, v_off, s[base] offset:imm` -- one 32-bit add per ROW of a trip instead of a 64-bit shift-and-add
// per LOAD
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

// ---- identity iteration (every frame starts from the identity warp, cpp/lib/registration.cpp:52-53) ---------------------
// Source pixel = target pixel, zero fractions: the bilinear weights are (1,0,0,0) and the general arithmetic reduces
// EXACTLY to the centre taps -- w = I, gradients = central differences of I.  UR rows of the thread's column per trip;
// the left / right taps come from the neighbouring lanes' centre values by DPP wave shifts (lane 0 and lane 63 load
// theirs: one load instruction per row under a two-lane exec mask), so a trip is UR + 2 column loads + UR template
// loads.  Every lane of the wave must run the trip (lanes past the rectangle read valid columns and are left out of
// the reduction).
template <int UR>
__device__ __forceinline__ void ecc_ident_trip(const float *__restrict__ I, const float *__restrict__ tmpl, int cols, int x, int y,
                                               int r, EccPart &P, EccTot &T)
{
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
        const float l = dpp_shr1(hh[k], cc[k + 1]), rr = dpp_shl1(hh[k], cc[k + 1]);
        __builtin_amdgcn_sched_barrier(0);      // one row after the other: the rows' temporaries must not all be live at once
        ecc_part_add<false>(P, T, cc[k + 1], 0.5f * (rr - l), 0.5f * (cc[k + 2] - cc[k]), tt[k], (float)(r + k));   // (= -a/2 + b/2 exactly)
    }
}

// ---- general iteration -----------------------------------------------------------------------------------------------------
// Footprint of a pixel (rows sy-1 .. sy+2 = a, b, c, d; columns sx-1 .. sx+2 = _1, 0, 1, 2) held as the pairs the
// arithmetic works on: {a0,a1} {b0,b1} {c0,c1} {d0,d1} and the outer pairs {b_1,b2} {c_1,c2}.
struct EccRow {
    v2f A, Bm, Cm, D, Be, Ce;
    float tt, fx, fy;
};

// Bilinear interpolation of I and of its central differences along x and y (cv::findTransformECC warps the two gradient
// IMAGES).  All three are linear in the pixels, so with V_j = the vertical interpolation at column j
//     w  = V_0 + fx (V_1 - V_0)
//     gx = 1/2 [ q_0 + fx (q_1 - q_0) ],   q = (b_+ - b_-) + fy ((c_+ - c_-) - (b_+ - b_-))  at columns 0, 1
//     gy = 1/2 [ p_0 + fx (p_1 - p_0) ],   p = (c - a) + fy ((d - b) - (c - a))  at columns 0, 1
// With zero fractions (identity) these are exactly the taps of the identity iteration.  gx from the horizontal
// DIFFERENCES of the pixels, interpolated -- not from differences of the interpolated V_j: those carry the rounding of
// values ~2000, 1e-4, into a gradient of a few counts, 1e-5 relative, where the reference's warped gradient image is
// good to 1e-7; the ECC iteration amplifies that on small or weakly textured images (tests/debug/soak_ecc.py found it,
// 8e-3 px on a 97 x 258 frame).
__device__ __forceinline__ void ecc_row_sum(const EccRow &q, EccPart &P, EccTot &T, float rf)
{
    const v2f FY = {q.fy, q.fy};
    const v2f Vm = __builtin_elementwise_fma(FY, q.Cm - q.Bm, q.Bm);     // {V_0, V_1}
    const float w = __builtin_fmaf(q.fx, Vm[1] - Vm[0], Vm[0]);
    // q_j regrouped as (c_+ - b_+) - (c_- - b_-) so that the packed differences the loads deliver as pairs are used as they are
    const v2f Dm = q.Cm - q.Bm, De = q.Ce - q.Be;                         // {c0-b0, c1-b1}, {c_1-b_1, c2-b2}
    const float g0 = __builtin_fmaf(q.fy, Dm[1] - De[0], q.Bm[1] - q.Be[0]);
    const float g1 = __builtin_fmaf(q.fy, De[1] - Dm[0], q.Be[1] - q.Bm[0]);
    const float gx = 0.5f * __builtin_fmaf(q.fx, g1 - g0, g0);
    const v2f E = q.Cm - q.A, F = q.D - q.Bm;
    const v2f Pp = __builtin_elementwise_fma(FY, F - E, E);
    const float gy = 0.5f * __builtin_fmaf(q.fx, Pp[1] - Pp[0], Pp[0]);
    ecc_part_add<false>(P, T, w, gx, gy, q.tt, rf);
}

// WarpAffineInvoker's fixed-point source coordinate of (x, y): per-row term rt (table of the frame's rows, written by the
// solve) + per-column term (ax, bx), each rounded on its own; 1/32-pixel fractions.
__device__ __forceinline__ void ecc_coord(int2 rt, int ax, int bx, int &sx, int &sy, float &fx, float &fy)
{
    const int Xr = rt.x + ax, Yr = rt.y + bx;
    const int Xq = (Xr + 16) >> 5, Yq = (Yr + 16) >> 5;
    sx = Xq >> 5;                            // (footprint inside the image by construction: ecc_band)
    sy = Yq >> 5;
    fx = (Xq & 31) * (1.f / 32);
    fy = (Yq & 31) * (1.f / 32);
}

// the 12 source pixels straight from global memory (segments whose footprint does not fit the LDS tile)
// (q0: byte offset of source pixel (sx, sy - 1) = 4 (cols (sy - 1) + sx); rows, columns < 2^15)
__device__ __forceinline__ void ecc_row_load_direct(const float *__restrict__ I, int cols, unsigned q0, EccRow &q)
{
    const unsigned pitch = 4u * (unsigned)cols;
    const unsigned q1 = q0 + pitch, q2 = q1 + pitch, q3 = q2 + pitch;
    q.A = ld_v2f(I, q0);
    q.Be[0] = ld_f32<-4>(I, q1);
    q.Bm = ld_v2f(I, q1);
    q.Be[1] = ld_f32<8>(I, q1);
    q.Ce[0] = ld_f32<-4>(I, q2);
    q.Cm = ld_v2f(I, q2);
    q.Ce[1] = ld_f32<8>(I, q2);
    q.D = ld_v2f(I, q3);
}

// Source taps from LDS (round 4).  Under a warp near the identity the 12 source pixels of destination pixel (x, y) are a
// 4 x 4 patch at (x, y) + shift: the lanes of a wave read patches one column apart, consecutive rows of a column patches
// one row apart -- every source pixel of a block's tile is wanted 12 times, and round 3's form asked the texture path for
// each of them (8 load instructions per pixel and lane; profiles/r03_ecc_pmc.txt: L1 request floor 82 us of a 221-us
// launch, the waves parked on those loads half of their life).  Here the block stages the source footprint of one float
// SEGMENT (<= 32 destination rows x its 256 columns: rows sy_min - 1 .. sy_max + 2, columns from a multiple of four below
// sx_min - 1 to sx_max + 2) with 16-byte coalesced loads, once, and every lane takes its taps from LDS.  The fixed-point
// source coordinate is separable and monotone -- X(x, y) = a(x) + r(y), both rounded on their own -- so the corners of
// the destination rectangle give the exact bounding box of the footprint (integer arithmetic on four table entries: all
// scalar).  A segment whose box does not fit the tile (strong rotation / scale, a row pitch that is not a multiple of
// four floats) walks the direct path.  The taps are the same floats and the arithmetic is ecc_row_sum either way:
// bit-identical sums (tests/test_imageops_gpu.py::test_ecc_lds_taps_same_bits).
constexpr int kEccTileRows = kEccFlush + 8;      // source rows of a segment: 32 + 3 of the footprint + 5 for scale / rotation
constexpr int kEccTilePitch = 272;               // floats per tile row: 256 + 3 of the footprint + 3 alignment + 10 for scale / shear
constexpr int kEccLdsRows = (kEccTileRows * kEccTilePitch * 4 + 256 * 8 - 1) / (256 * 8);      // the tile in rows of the reduction area
static_assert(kEccLdsRows >= kEccChunk, "the tile area also holds the reduction chunks");

struct EccSeg {          // one float segment of a block (all uniform)
    int r0, c0;          // tile origin in the source image (row, column; c0 a multiple of 4)
    int nr, nc4;         // tile rows, float4 per tile row
    bool fits;
};

__device__ __forceinline__ EccSeg ecc_segment_box(int2 ra, int2 re, int ax_lo, int ax_hi, int bx_lo, int bx_hi, int rows, int cols,
                                                  int tile_rows = kEccTileRows)
{
    const int sx_min = ((min(ra.x, re.x) + ax_lo + 16) >> 5) >> 5, sx_max = ((max(ra.x, re.x) + ax_hi + 16) >> 5) >> 5;
    const int sy_min = ((min(ra.y, re.y) + bx_lo + 16) >> 5) >> 5, sy_max = ((max(ra.y, re.y) + bx_hi + 16) >> 5) >> 5;
    EccSeg g;
    g.r0 = sy_min - 1;
    g.c0 = (sx_min - 1) & ~3;
    g.nr = sy_max + 2 - g.r0 + 1;
    g.nc4 = (sx_max + 2 - g.c0 + 4) >> 2;
    // (interior pixels have their footprint inside the image by construction -- ecc_band; the tests on the image bounds
    //  only keep a tile load from ever leaving the frame)
    g.fits = !(cols & 3) && g.nr <= tile_rows && g.nc4 * 4 <= kEccTilePitch && g.r0 >= 0 && sx_min >= 1 &&
             sy_max + 2 < rows && sx_max + 2 < cols && g.nr > 0 && g.nc4 >= 1;
    return g;
}

// all 256 threads: the segment's source footprint -> LDS (every load issued before the first LDS store).  Wave w takes the
// tile rows w, w + 4, ...: lane l the float4 l of the row (256 floats: one 1-KB request per wave and row); the up to
// four float4 beyond them (columns 256 .. 271 of the tile) are a pass of their own, thread t -> (row t / 4, float4 64 + t % 4).
// Index arithmetic: one add per load (a division per element made the staging a quarter of the kernel's instructions).
template <int TILE_ROWS = kEccTileRows, int PITCH = kEccTilePitch>
__device__ __forceinline__ void ecc_stage_tile(const float *__restrict__ I, int cols, const EccSeg &g, float *tile)
{
    constexpr int kPerWave = (TILE_ROWS + 3) / 4;
    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    const int tr = (int)threadIdx.x >> 2, tc = 64 + ((int)threadIdx.x & 3);       // the tail element of this thread
    const bool main_on = lane < g.nc4 && g.c0 + 4 * lane + 3 < cols;               // (cols % 4 == 0: a float4 is inside the row or outside it)
    const bool tail_on = tr < g.nr && tc < g.nc4 && g.c0 + 4 * tc + 3 < cols;
    const float *src = I + (size_t)(g.r0 + wave) * cols + (g.c0 + 4 * lane);
    const size_t step = 4 * (size_t)cols;
    float4 v[kPerWave], vt = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < kPerWave; ++k) {
        v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (main_on && wave + 4 * k < g.nr) v[k] = *reinterpret_cast<const float4 *>(src + (size_t)k * step);
    }
    if (tail_on) vt = *reinterpret_cast<const float4 *>(I + (size_t)(g.r0 + tr) * cols + (g.c0 + 4 * tc));
    float *dst = tile + wave * PITCH + 4 * lane;
#pragma unroll
    for (int k = 0; k < kPerWave; ++k)
        if (lane < g.nc4 && wave + 4 * k < g.nr) *reinterpret_cast<float4 *>(dst + k * 4 * PITCH) = v[k];
    if (tr < g.nr && tc < g.nc4) *reinterpret_cast<float4 *>(tile + tr * PITCH + 4 * tc) = vt;
}

// rows [yb, yb + ne) of the thread's column: taps from the staged tile (TILE) or straight from global memory; rt = the frame's
// row table at row yb.  UR rows per trip; the template values (one coalesced 4-byte load per pixel) come one trip ahead.
template <int UR, bool TILE>
__device__ __forceinline__ void ecc_walk_segment(const float *tile, const EccSeg &g, const float *__restrict__ I,
                                                 const float *__restrict__ tmpl, int cols, int x, int yb, int ne, int ax, int bx,
                                                 const int2 *__restrict__ rt, EccPart &P, EccTot &T, int rbase = 0, bool pairs = true)
{
    const unsigned pitch = 4u * (unsigned)cols;
    unsigned ot = 4u * (unsigned)(yb * cols + x);
    // Source coordinate of the thread's column, row by row.  The row term rt[r] is the same for every lane, and under a warp
    // near the identity it moves by exactly (0, +1 pixel) from most rows to the next (the x term of a row changes by one 1/1024
    // px every 1 / |M[1]| rows, the y term leaves +1024 every 1 / |M[4] - 1| rows): then every lane's source pixel is the one
    // below the previous row's and its fractions are the same -- the fixed-point arithmetic (a dozen vector instructions per
    // pixel) is skipped on a scalar test, the footprint's address moves down one row.  Same coordinates, same taps, same bits.
    float cfx = 0.f, cfy = 0.f;
    unsigned coff = 0u;                 // TILE: float index of the footprint's corner in the tile; direct: its byte offset in the frame
    int2 prev = make_int2(0, 0);
    bool have = false;
    auto coord = [&](int2 a) {
        if (have && a.x == prev.x && a.y == prev.y + 1024) {       // (uniform)
            coff += TILE ? (unsigned)kEccTilePitch : pitch;
        } else {
            int sx, sy;
            ecc_coord(a, ax, bx, sx, sy, cfx, cfy);
            coff = TILE ? (unsigned)((sy - 1 - g.r0) * kEccTilePitch + (sx - 1 - g.c0)) : 4u * (unsigned)(__mul24(sy - 1, cols) + sx);
        }
        have = true;
        prev = a;
    };
    auto taps = [&](int r, EccRow &q) {
        coord(rt[r]);
        q.fx = cfx;
        q.fy = cfy;
        if (TILE) {
            const float *t0 = tile + coff;
            const float *t1 = t0 + kEccTilePitch, *t2 = t1 + kEccTilePitch, *t3 = t2 + kEccTilePitch;
            q.A[0] = t0[1]; q.A[1] = t0[2];
            q.Be[0] = t1[0]; q.Bm[0] = t1[1]; q.Bm[1] = t1[2]; q.Be[1] = t1[3];
            q.Ce[0] = t2[0]; q.Cm[0] = t2[1]; q.Cm[1] = t2[2]; q.Ce[1] = t2[3];
            q.D[0] = t3[1]; q.D[1] = t3[2];
        } else {
            ecc_row_load_direct(I, cols, coff, q);
        }
    };
    float tn[UR];
#pragma unroll
    for (int k = 0; k < UR; ++k) tn[k] = k < ne ? ld_f32(tmpl, ot + (unsigned)k * pitch) : 0.f;
    int r = 0;
    if constexpr (TILE && UR == 2) {
        // Two rows per trip with every value held as the PAIR {row r, row r + 1} (round 6): the footprint arithmetic of ecc_row_sum is
        // then packed instructions throughout (24 + 8 scalar for two rows instead of 14 packed + 28 scalar), and the taps are single
        // LDS reads with immediate offsets from ONE address per row (the tile pointer in the LDS address space and volatile, so that
        // the compiler neither goes through flat addresses nor merges neighbouring reads into pairs of COLUMNS) -- no address
        // arithmetic per tap.  The same operations on the same values per component: the same bits as ecc_row_sum.
        typedef const volatile __attribute__((address_space(3))) float lds_cvf;
        constexpr int TP = kEccTilePitch;
        // (the row terms come by scalar loads, which share their counter with the LDS reads: fetched one trip ahead, they are
        //  there when the wait for the taps ends instead of being a round trip of their own in front of every trip)
        int2 rn0 = rt[0], rn1 = rt[min(1, ne - 1)];
        // Template values TWO trips ahead (a trip is ~1.4 us of the wave's life at four waves per SIMD: about one L2 / Infinity Cache
        // round trip under load), in two register pairs used in turn -- the loop body is written out twice so that no value has to
        // be moved from one pair to the other (a move would wait for the load it copies).
        float tb0 = pairs && 2 < ne ? ld_f32(tmpl, ot + 2u * pitch) : 0.f, tb1 = pairs && 3 < ne ? ld_f32(tmpl, ot + 3u * pitch) : 0.f;
        auto trip = [&](float &ta0, float &ta1) {            // rows r, r + 1: template values in (ta0, ta1), refilled for rows r + 4, r + 5
            const float t0 = ta0, t1 = ta1;
            ta0 = r + 4 < ne ? ld_f32(tmpl, ot + 4u * pitch) : 0.f;
            ta1 = r + 5 < ne ? ld_f32(tmpl, ot + 5u * pitch) : 0.f;
            const int2 rc0 = rn0, rc1 = rn1;
            rn0 = rt[min(r + 2, ne - 1)];
            rn1 = rt[min(r + 3, ne - 1)];
            coord(rc0);
            lds_cvf *a = (lds_cvf *)tile + coff;
            const float fx0 = cfx, fy0 = cfy;
            coord(rc1);
            lds_cvf *b = (lds_cvf *)tile + coff;
            const v2f FX = {fx0, cfx}, FY = {fy0, cfy};
            const v2f A0 = {a[1], b[1]}, A1 = {a[2], b[2]};
            const v2f B_1 = {a[TP], b[TP]}, B0 = {a[TP + 1], b[TP + 1]}, B1 = {a[TP + 2], b[TP + 2]}, B2 = {a[TP + 3], b[TP + 3]};
            const v2f C_1 = {a[2 * TP], b[2 * TP]}, C0 = {a[2 * TP + 1], b[2 * TP + 1]}, C1 = {a[2 * TP + 2], b[2 * TP + 2]},
                      C2 = {a[2 * TP + 3], b[2 * TP + 3]};
            const v2f D0 = {a[3 * TP + 1], b[3 * TP + 1]}, D1 = {a[3 * TP + 2], b[3 * TP + 2]};
            const v2f Dm0 = C0 - B0, Dm1 = C1 - B1;                                   // {c0-b0}, {c1-b1}
            const v2f V0 = __builtin_elementwise_fma(FY, Dm0, B0), V1 = __builtin_elementwise_fma(FY, Dm1, B1);
            const v2f W = __builtin_elementwise_fma(FX, V1 - V0, V0);
            const v2f De0 = C_1 - B_1, De1 = C2 - B2;                                 // {c_1-b_1}, {c2-b2}
            const v2f G0 = __builtin_elementwise_fma(FY, Dm1 - De0, B1 - B_1);
            const v2f G1 = __builtin_elementwise_fma(FY, De1 - Dm0, B2 - B0);
            const v2f dG = G1 - G0;
            const v2f E0 = C0 - A0, E1 = C1 - A1, F0 = D0 - B0, F1 = D1 - B1;
            const v2f P0 = __builtin_elementwise_fma(FY, F0 - E0, E0), P1 = __builtin_elementwise_fma(FY, F1 - E1, E1);
            const v2f dP = P1 - P0;
            // the last step per row: {gx, gy} is the pair the sums work on
            const float gx0 = 0.5f * __builtin_fmaf(fx0, dG[0], G0[0]), gy0 = 0.5f * __builtin_fmaf(fx0, dP[0], P0[0]);
            const float gx1 = 0.5f * __builtin_fmaf(cfx, dG[1], G0[1]), gy1 = 0.5f * __builtin_fmaf(cfx, dP[1], P0[1]);
            ecc_part_add<false>(P, T, W[0], gx0, gy0, t0, (float)(rbase + r));
            ecc_part_add<false>(P, T, W[1], gx1, gy1, t1, (float)(rbase + r + 1));
            ot += 2u * pitch;
            r += 2;
        };
        while (pairs && r + 4 <= ne) {
            trip(tn[0], tn[1]);
            trip(tb0, tb1);
        }
        if (pairs && r + 2 <= ne) {
            trip(tn[0], tn[1]);
            tn[0] = tb0;          // (what a last single row reads)
            tn[1] = tb1;
        }
    }
    for (; r + UR <= ne; r += UR) {
        EccRow q[UR];
#pragma unroll
        for (int k = 0; k < UR; ++k) {
            q[k].tt = tn[k];
            tn[k] = r + UR + k < ne ? ld_f32(tmpl, ot + (unsigned)(UR + k) * pitch) : 0.f;
        }
#pragma unroll
        for (int k = 0; k < UR; ++k) taps(r + k, q[k]);
#pragma unroll
        for (int k = 0; k < UR; ++k) ecc_row_sum(q[k], P, T, (float)(rbase + r + k));
        ot += (unsigned)UR * pitch;
    }
#pragma unroll
    for (int k = 0; k < UR - 1; ++k)
        if (r + k < ne) {
            EccRow q;
            q.tt = tn[k];
            taps(r + k, q);
            ecc_row_sum(q, P, T, (float)(rbase + r + k));
        }
}

// Interior block `blk` of `nblk`: the inner rectangle (farther than the band from every edge) is cut into column tiles
// of 256 and, per tile, into nblk / tiles row pieces; blocks beyond that store zeros.  Needs nblk >= tiles.  The float
// segments follow the row pieces, i.e. the image geometry alone: the sums of a frame are the same bits in any batch.
// Round 6: the ONE-FLUSH form of the interior blocks.  A block's row piece (<= kEccFlushLong rows; the host picks the form by the image
// height) is ONE float segment: the partial sums run over the whole piece and the double totals are formed once, behind the loop, so
// their 54 registers are not live in it -- the general iteration fits four waves per SIMD (128 registers, 3 spilled; forcing four
// waves on the 32-row form spilled 31, round 5), the identity iteration five.  Four workgroups per compute unit have 40 KB of LDS
// each: the tile of that form has 35 rows -- 3 for the footprint, 4 for shear / scale -- and is staged per 28 rows (the 40-row tile:
// per 32); a segment whose footprint is taller takes the direct loads.  A float segment of 128 rows instead of 32 doubles the
// rounding error of a segment sum (sqrt 4) and quarters their number: the frame's sums move in the ninth digit, iteration counts and
// the 1e-4 / 2e-3 px bars as before (tests, profiles/r06_soak_ecc.txt).  Measured, 512-frame launches: general 1.46 -> 1.375 ms,
// identity 0.83 -> 0.80 ms (one flush at the old occupancy: 1.45 / 0.81; identity <2 rows, 5 waves> 0.91, <3, 5> 0.83; general
// <1 row, 4 waves> 1.47).
constexpr int ecc_tile_rows(int waves) { return waves >= 4 ? 35 : kEccTileRows; }
constexpr int ecc_lds_rows(int waves) { return (ecc_tile_rows(waves) * kEccTilePitch * 4 + 256 * 8 - 1) / (256 * 8); }
template <bool IDENT, int UR, bool ONE = false, int TR = kEccTileRows>
__device__ __forceinline__ void ecc_cols_body(const float *__restrict__ img, const float *__restrict__ tmpl, int rows, int cols,
                                              const EccState *__restrict__ state, const int2 *__restrict__ rtab,
                                              double *__restrict__ partial, int f, unsigned slot0, unsigned blk, unsigned nblk,
                                              double (*lds_red)[256], float center, int force_direct)
{
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
    if (!work || y1 <= y0) {            // (uniform) more blocks than pieces: nothing to add
        ecc_store_zeros(partial, f, slot0 + blk);
        return;
    }
    EccTot T;
    ecc_tot_zero(T);
    T.n = on ? (double)(y1 - y0) : 0.0;          // mask = 1 on every interior pixel
    T.cf = center;
    int x = x_own;
    EccPart P1;                                  // ONE: the partial sums of the whole row piece
    if (ONE) ecc_part_zero(P1);
    if (IDENT) {
        // lanes past the rectangle run along on a valid column (the DPP taps need every lane; the lane after the last one must
        // hold column x_hi, which exists: the band is >= 3 wide) and are left out of the reduction.
        // (Round 4: staging the segment's rows in LDS like the general iteration does was measured and is SLOWER here, 2.03-2.11
        //  against 1.97 ms per 1000 frames: 2.5 coalesced loads per pixel are not what this kernel waits for.)
        x = min(x_own, cols - 1);
        if (ONE) {
            const int ne = y1 - y0;
            int r = 0;
            for (; r + UR <= ne; r += UR) ecc_ident_trip<UR>(I, tmpl, cols, x, y0 + r, r, P1, T);
            for (; r < ne; ++r) ecc_ident_trip<1>(I, tmpl, cols, x, y0 + r, r, P1, T);
        } else {
            for (int yb = y0; yb < y1; yb += kEccFlush) {
                const int ne = min(kEccFlush, y1 - yb);
                EccPart P;
                ecc_part_zero(P);
                int r = 0;
                for (; r + UR <= ne; r += UR) ecc_ident_trip<UR>(I, tmpl, cols, x, yb + r, r, P, T);
                for (; r < ne; ++r) ecc_ident_trip<1>(I, tmpl, cols, x, yb + r, r, P, T);
                ecc_part_flush(P, T, yb);
            }
        }
    } else {
        float *tile = reinterpret_cast<float *>(&lds_red[0][0]);
        const int2 *rt = rtab + (size_t)f * rows;
        const double M0 = es.M[0], M3 = es.M[3];
        // the column terms at the two ends of the block's columns (monotone in x: the extremes of the block)
        const int xa = x_lo + (int)ct * 256, xb = min(xa + 255, x_hi - 1);
        const int axa = __builtin_amdgcn_readfirstlane(__double2int_rn(M0 * xa * 1024)), axb = __builtin_amdgcn_readfirstlane(__double2int_rn(M0 * xb * 1024));
        const int bxa = __builtin_amdgcn_readfirstlane(__double2int_rn(M3 * xa * 1024)), bxb = __builtin_amdgcn_readfirstlane(__double2int_rn(M3 * xb * 1024));
        const int ax = __double2int_rn(M0 * x * 1024), bx = __double2int_rn(M3 * x * 1024);
        bool staged = false;
        // staging segments: as many rows as leave the tile 3 rows for the footprint and >= 4 for shear / scale (40-row tile: 32;
        // the 35-row tile of the four-wave form: 28).  In the one-flush form they are not the float segments.
        constexpr int STG = ONE ? (TR - 7 < kEccFlush ? TR - 7 : kEccFlush) : kEccFlush;
        for (int yb = y0; yb < y1; yb += STG) {
            const int ne = min(STG, y1 - yb);
            EccSeg s = ecc_segment_box(rt[yb], rt[yb + ne - 1], min(axa, axb), max(axa, axb), min(bxa, bxb), max(bxa, bxb), rows, cols, TR);
            if (force_direct & 1) s.fits = false;
            if (s.fits) {                              // (uniform)
                if (staged) __syncthreads();           // every tap of the previous segment has been read
                ecc_stage_tile<TR, kEccTilePitch>(I, cols, s, tile);
                __syncthreads();
                staged = true;
            }
            if (on) {
                if (ONE) {
                    if (s.fits) ecc_walk_segment<UR, true>(tile, s, I, tmpl, cols, x, yb, ne, ax, bx, rt + yb, P1, T, yb - y0, !(force_direct & 2));
                    else ecc_walk_segment<UR, false>(tile, s, I, tmpl, cols, x, yb, ne, ax, bx, rt + yb, P1, T, yb - y0);
                } else {
                    EccPart P;
                    ecc_part_zero(P);
                    if (s.fits) ecc_walk_segment<UR, true>(tile, s, I, tmpl, cols, x, yb, ne, ax, bx, rt + yb, P, T);
                    else ecc_walk_segment<UR, false>(tile, s, I, tmpl, cols, x, yb, ne, ax, bx, rt + yb, P, T);
                    ecc_part_flush(P, T, yb);
                }
            }
        }
    }
    if (ONE) ecc_part_flush(P1, T, y0);
    __syncthreads();                              // (every read of the tile is done: the area becomes the reduction's)
    ecc_tot_store<0>(T, (double)x, on, lds_red, partial, f, slot0 + blk);
}

// The band of the same launch, also one column per thread.  Band blocks (nband of them, >= 3 x tiles):
//   [0, tiles)            top strip    rows [0, top),            one column tile of 256 each
//   [tiles, 2 tiles)      bottom strip rows [rows - bottom, rows)
//   the rest              the left + right strips between them: `side` = left + right columns; a block's 256 threads
//                         are (column, row piece) pairs, so a 6-column band still has 42 threads per block at work
// Generic bilinear (constant-0 border, reflect-101 gradient taps) and the nearest-neighbour mask.
__device__ __forceinline__ void ecc_band_cols_body(const float *__restrict__ img, const float *__restrict__ tmpl, int rows,
                                                   int cols, const EccState *__restrict__ state, const int2 *__restrict__ rtab,
                                                   double *__restrict__ partial, int f, unsigned bidx, unsigned nband, bool ident,
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
        const int sb = (int)bidx - 2 * tiles, nsb = (int)nband - 2 * tiles;
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
        ecc_store_zeros(partial, f, bidx);
        return;
    }
    EccTot T;
    ecc_tot_zero(T);
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
                if (ident) {
                    // identity warp: source pixel = target pixel, zero fractions, mask 1 -- the generic path below evaluates to
                    // exactly these taps (weights (1, 0, 0, 0); x * 0 terms add +-0), without its coordinate arithmetic
                    ecc_part_add<true>(P, T, pix(y, x), gxf(y, x), gyf(y, x), tmpl[(size_t)y * cols + x], (float)r, true);
                    continue;
                }
                const int2 rt = rtab[(size_t)f * rows + y];        // the per-row terms under the frame's M (written by the solve)
                const int Xr = rt.x + ax, Yr = rt.y + bx;
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
    ecc_tot_store<0>(T, (double)x, on, lds_red, partial, f, bidx);
}

// ---- the 5 x 5 pre-blur fused into the identity iteration (round 6) ----------------------------------------------------
// Every frame starts from the identity warp, so its first ECC iteration needs nothing but the blurred frame itself: the blur
// (cpp/lib/registration.cpp:57-60, gaussFiltSize 5) and the identity sums are ONE pass -- the u16 frame is read (2 B / px), the
// blurred f32 frame is written for the later iterations (4 B / px) and its values go into the sums from registers: 10 B / px with
// the template instead of the 14 of gauss5_quad_kernel + ecc_cols_kernel<true>.
//   * A WAVE owns a strip of 58 columns x 16 .. 128 rows (one float segment; the host picks the row count by the image size:
//     launch_ecc_blur_ident).  Lane l holds column
//     58 strip - 3 + l: lanes 3 .. 60 are the strip's own columns, the three lanes on either side its neighbours' -- the horizontal
//     blur taps come by DPP wave shifts (two chained shifts per side), so lanes 2 .. 61 hold a blurred value, and the x-gradient of
//     an own lane is the difference of its neighbours' blurred values, which nobody else has computed yet.  The five rows of
//     horizontal results and three blurred rows roll in registers.  Same float operations in the same order as gauss_pass_kernel:
//     the blurred frame is bit-identical to the unfused path's (tests/test_imageops_gpu.py::test_ecc_fused_blur_*).
//   * Image edges by reflection of the LOADS (BORDER_REFLECT_101 of the blur): the extended signal is symmetric about the edge, so the
//     blurred value "at column -1" is the one at column 1 -- exactly filter2D's reflect-101 tap of the gradient images.  Under the
//     identity warp every pixel is inside the mask and the bilinear weights are (1, 0, 0, 0): no band blocks, all rows x cols pixels
//     are plain ones.
//   * Hot pixels: the scan of fix_hot_pixels rides on the loads as in gauss5_quad_kernel.  In the frames the repair changes
//     afterwards, the workgroups a repaired pixel reaches run again (ecc_blur_ident_again_kernel): blurred pixels and sums from the
//     repaired frame.
//   * Four waves per SIMD (108 registers), the loads of a trip requested one trip ahead: see the loop.
// Block = four wave items (strip-major: neighbouring strips of one row piece), reduced like the interior blocks.
constexpr int kFusedOwn = 58, kFusedHalo = 3;

// wave shifts that leave 0 in the lane without a source (bound_ctrl: no `old` register to initialise, and the compiler may fold the
// shift into the instruction that uses it)
__device__ __forceinline__ float dpp_shr1_z(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_shl1_z(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xF, 0xF, true));
}

// workgroup `blk` of `nblk` of frame f
template <bool HOT, int U>
__device__ __forceinline__ void ecc_blur_ident_block(const uint16_t *__restrict__ src, float *__restrict__ dst, const float *__restrict__ tmpl,
                                                     int rows, int cols, int strips, int pieces, int prows, double *__restrict__ partial,
                                                     const float *__restrict__ center, float k0, float k1, float k2, unsigned thresh,
                                                     unsigned *__restrict__ hot_count, unsigned *__restrict__ hot_pos,
                                                     const double *__restrict__ tsum, int f, int blk, int nblk, double (*lds_red)[256])
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int item = blk * 4 + wave;
    const int piece = item / strips, strip = item - piece * strips;
    const bool work = piece < pieces;
    const int y0 = piece * prows, y1 = min(rows, y0 + prows);
    const int c = strip * kFusedOwn - kFusedHalo + lane;
    const bool own = work && lane >= kFusedHalo && lane < kFusedHalo + kFusedOwn && c < cols;
    EccTot T;
    ecc_tot_zero(T);
    T.cf = *center;
    EccPart P;
    ecc_part_zero(P);
    if (work) {
        // Buffer descriptors (uniform: kernel arguments and the frame index) -- a load or store is `descriptor + the lane's constant
        // 32-bit offset + the row's SCALAR offset`: no vector instruction per access, and a store whose lane offset lies beyond the
        // frame is dropped by the range check (the lanes that own no pixel), so no exec mask either.
        const size_t npix = (size_t)rows * cols;
        const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(src + (size_t)f * npix), 0, (int)(npix * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(dst + (size_t)f * npix, 0, (int)(npix * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rT = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(tmpl), 0, (int)(npix * 4), 0x00020000);
        const unsigned pitch2 = 2u * (unsigned)cols, pitch4 = 4u * (unsigned)cols;
        const unsigned cx2 = 2u * (unsigned)reflect101(min(max(c, -(cols - 1)), 2 * cols - 2), cols);
        const unsigned cown = (unsigned)min(max(c, 0), cols - 1), cown4 = 4u * cown;
        const unsigned cstore4 = own ? cown4 : 0x80000000u;
        // rows -3 .. rows + 2 (rows >= 8): one reflection
        auto rrow = [&](int yy) { return (unsigned)(yy < 0 ? -yy : (yy >= rows ? 2 * rows - 2 - yy : yy)); };
        float h0 = 0.f, h1 = 0.f, h2 = 0.f, h3 = 0.f, h4 = 0.f, Bm = 0.f, B0 = 0.f, Bp = 0.f;
        auto load_row = [&](int yi) -> unsigned {
            return (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rS, (int)cx2, (int)(rrow(yi) * pitch2), 0);
        };
        auto hot_scan = [&](int yi, unsigned pv) {
            if (HOT && own && yi >= y0 && yi < y1 && pv >= thresh) {           // (rare)
                const unsigned slot = atomicAdd(&hot_count[f], 1u);
                if (slot < (unsigned)kHotPositions) hot_pos[(size_t)f * kHotPositions + slot] = (unsigned)yi * (unsigned)cols + cown;
            }
        };
        auto hstep = [&](unsigned pv) {                                        // horizontal pass of one input row
            const float pf = (float)pv;
            const float a1 = dpp_shr1_z(pf), c1 = dpp_shl1_z(pf);
            const float a2 = dpp_shr1_z(a1), c2 = dpp_shl1_z(c1);
            float n = k0 * pf;
            n += k1 * (a1 + c1);
            n += k2 * (a2 + c2);
            h0 = h1; h1 = h2; h2 = h3; h3 = h4; h4 = n;
        };
        auto vstep = [&](int yb, bool store) {                                 // blurred row yb = the middle of the five
            float bn = k0 * h2;
            bn += k1 * (h1 + h3);
            bn += k2 * (h0 + h4);
            Bm = B0; B0 = Bp; Bp = bn;
            if (store) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(bn), rB, (int)cstore4, (int)((unsigned)yb * pitch4), 0);   // (uniform)
        };
        // input rows y0 - 3 .. y0 + 2: the blurred rows y0 - 1 and y0
        {
            unsigned pv[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) pv[k] = load_row(y0 - 3 + k);
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                hot_scan(y0 - 3 + k, pv[k]);
                hstep(pv[k]);
                if (k >= 4) vstep(y0 - 5 + k, k == 5);
            }
        }
        // input row y0 + 3 + r: blurred row y0 + 1 + r, sums of row y0 + r.  STEADY trips: every input row of the trip lies inside
        // the piece (r + 3 < ne for its last row), so there is no reflection, every blurred row is stored, every loaded pixel is the
        // wave's own to scan and no row is past the end -- plain scalar arithmetic per row; the last trip or two take the tests.
        const int ne = y1 - y0;
        // The pixels and template values of a trip are requested ONE TRIP AHEAD (two sets of registers used in turn, the loop written
        // out twice so that nothing is moved behind a load): a frame's pixels come from HBM for the first time -- ~2 us -- and a trip
        // is ~1 000 instruction cycles of the wave.
        auto fetch = [&](int g, unsigned (&qpv)[U], float (&qtv)[U], auto steady_tag) {
            constexpr bool STEADY = decltype(steady_tag)::value;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int yi = y0 + 3 + g + u;
                qpv[u] = STEADY ? (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rS, (int)cx2, (int)((unsigned)yi * pitch2), 0)
                                 : load_row(min(yi, y1 + 2));
                const int yt = STEADY ? y0 + g + u : min(y0 + g + u, rows - 1);
                qtv[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rT, (int)cown4, (int)((unsigned)yt * pitch4), 0));
            }
        };
        auto work = [&](int g, const unsigned (&qpv)[U], const float (&qtv)[U], auto steady_tag) {
            constexpr bool STEADY = decltype(steady_tag)::value;
            // the horizontal pass of two rows at a time as packed pairs (the same operations per component)
            static_assert(U % 2 == 0, "rows in pairs");
            v2f N[U / 2];
#pragma unroll
            for (int u = 0; u < U; u += 2) {
                const v2f Pf = {(float)qpv[u], (float)qpv[u + 1]};
                const v2f A1 = {dpp_shr1_z(Pf[0]), dpp_shr1_z(Pf[1])}, C1 = {dpp_shl1_z(Pf[0]), dpp_shl1_z(Pf[1])};
                const v2f A2 = {dpp_shr1_z(A1[0]), dpp_shr1_z(A1[1])}, C2 = {dpp_shl1_z(C1[0]), dpp_shl1_z(C1[1])};
                const v2f K0 = {k0, k0}, K1 = {k1, k1}, K2 = {k2, k2};
                v2f n = K0 * Pf;
                n += K1 * (A1 + C1);
                n += K2 * (A2 + C2);
                N[u / 2] = n;
            }
            if (HOT && STEADY) {                     // one test per trip: the largest of its pixels
                unsigned mx = qpv[0];
#pragma unroll
                for (int u = 1; u < U; ++u) mx = max(mx, qpv[u]);
                if (own && mx >= thresh) {           // (rare)
#pragma unroll
                    for (int u = 0; u < U; ++u) hot_scan(y0 + 3 + g + u, qpv[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = g + u;
                if (!STEADY && r >= ne) break;                                 // (uniform)
                if (!STEADY) hot_scan(y0 + 3 + r, qpv[u]);
                h0 = h1; h1 = h2; h2 = h3; h3 = h4; h4 = N[u / 2][u & 1];
                vstep(y0 + 1 + r, STEADY || r + 1 < ne);
                const float l = dpp_shr1_z(B0), rr = dpp_shl1_z(B0);
                ecc_part_add<false, false>(P, T, B0, 0.5f * (rr - l), 0.5f * (Bp - Bm), qtv[u], (float)r);
            }
        };
        unsigned pa[U], pb[U];
        float ta[U], tb[U];
        int g = 0;
        fetch(0, pa, ta, std::false_type{});
        while (g + 3 * U + 3 <= ne) {            // trips g, g + U and g + 2 U are steady ones
            fetch(g + U, pb, tb, std::true_type{});
            __builtin_amdgcn_sched_barrier(0);        // (one trip's arithmetic after the other: interleaved they need 122 registers)
            work(g, pa, ta, std::true_type{});
            __builtin_amdgcn_sched_barrier(0);
            fetch(g + 2 * U, pa, ta, std::true_type{});
            __builtin_amdgcn_sched_barrier(0);
            work(g + U, pb, tb, std::true_type{});
            __builtin_amdgcn_sched_barrier(0);
            g += 2 * U;
        }
        while (g < ne) {                          // the last trips (pa / ta hold trip g)
            if (g + U < ne) fetch(g + U, pb, tb, std::false_type{});
            work(g, pa, ta, std::false_type{});
#pragma unroll
            for (int u = 0; u < U; ++u) {
                pa[u] = pb[u];
                ta[u] = tb[u];
            }
            g += U;
        }
    }
    ecc_part_flush(P, T, y0);
    T.n = own ? (double)(y1 - y0) : 0.0;          // mask = 1 on every pixel
    if (blk == 0 && (int)threadIdx.x == kFusedHalo) {      // (pixel (0, 0): the thread that owns it carries the template's sums)
        T.St = tsum[0];
        T.Stt = tsum[1];
    }
    ecc_tot_store<0, 5>(T, (double)c, own, lds_red, partial, f, (unsigned)blk, (unsigned)nblk);
}

// first pass: grid (frames, workgroups per frame)
template <bool HOT, int U, int W>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W, W)))
    ecc_blur_ident_kernel(const uint16_t *__restrict__ src, float *__restrict__ dst, const float *__restrict__ tmpl, int rows, int cols,
                          int strips, int pieces, int prows, double *__restrict__ partial, const float *__restrict__ center, float k0,
                          float k1, float k2, unsigned thresh, unsigned *__restrict__ hot_count, unsigned *__restrict__ hot_pos,
                          const double *__restrict__ tsum)
{
    __shared__ double lds_red[5][256];                                          // (10 KB: the workgroups of a compute unit are bounded by registers)
    ecc_blur_ident_block<HOT, U>(src, dst, tmpl, rows, cols, strips, pieces, prows, partial, center, k0, k1, k2, thresh, hot_count, hot_pos, tsum,
                                 (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y, lds_red);
}

// Second pass, over the frames the hot-pixel repair changed: the first pass's grid.  Workgroup (f, blk) looks at its frame's change
// records and runs again if a repaired pixel reaches it -- a repair moves the blurred values within 2 pixels and, through their
// gradients, the sums of the pixels within 3; a workgroup whose four wave items own none of those keeps its blurred pixels and its
// sums.  Measured per 512-frame launch with 1 % of the frames changed: ~40 us, the time of ONE workgroup that does run (134 rows with
// nothing beside it on its compute unit) -- the 18 000 that return at once are dispatched meanwhile.  One workgroup per 8 frames or
// per frame, running what is reached one after the other: 55 / 140 us.
constexpr int kAgainFrames = 1;
template <int U>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
    ecc_blur_ident_again_kernel(const uint16_t *__restrict__ src, float *__restrict__ dst, const float *__restrict__ tmpl, int rows, int cols,
                                int strips, int pieces, int prows, int nframes, double *__restrict__ partial, const float *__restrict__ center, float k0,
                                float k1, float k2, const unsigned *__restrict__ nchanged, const uint4 *__restrict__ changes, int max_hot,
                                const double *__restrict__ tsum)
{
    __shared__ double lds_red[5][256];
    const int blk = blockIdx.y;
    for (int j = 0; j < kAgainFrames; ++j) {
        const int f = (int)blockIdx.x * kAgainFrames + j;
        if (f >= nframes) break;
        const unsigned m = nchanged[f];
        bool hit = false;
        for (unsigned i = 0; i < m; ++i) {                                     // (uniform: m <= max_hot)
            const unsigned pos = changes[(size_t)f * max_hot + i].y;
            const int py = (int)(pos / (unsigned)cols), px = (int)(pos % (unsigned)cols);
            for (int w = 0; w < 4; ++w) {
                const int it = blk * 4 + w, pc = it / strips, sp = it - pc * strips;
                hit |= pc < pieces && px >= sp * kFusedOwn - 3 && px < (sp + 1) * kFusedOwn + 3 && py >= pc * prows - 3 &&
                       py < (pc + 1) * prows + 3;
            }
        }
        if (!hit) continue;                                                    // (uniform)
        ecc_blur_ident_block<false, U>(src, dst, tmpl, rows, cols, strips, pieces, prows, partial, center, k0, k1, k2, 0u, nullptr, nullptr, tsum, f,
                                       blk, (int)gridDim.y, lds_red);
    }
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

// sum t and sum t^2 over the whole blurred template, in double (64 workgroups + one wave, fixed order): what every pixel's share of
// St and Stt adds up to when the mask is 1 everywhere -- the identity iteration (ecc_blur_ident_kernel) takes them from here
constexpr int kTmplSumBlocks = 64;
__global__ void __launch_bounds__(256) ecc_tmpl_sums_kernel(const float *__restrict__ tmpl, size_t npix, double *__restrict__ part)
{
    __shared__ double sh[2][256];
    double a = 0.0, b = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (size_t)kTmplSumBlocks * 256) {
        const double t = (double)tmpl[i];
        a += t;
        b = fma(t, t, b);
    }
    sh[0][threadIdx.x] = a;
    sh[1][threadIdx.x] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = sh[0][0];
        part[2 * blockIdx.x + 1] = sh[1][0];
    }
}
__global__ void __launch_bounds__(64) ecc_tmpl_sums_final(const double *__restrict__ part, double *__restrict__ out)
{
    double a = part[2 * threadIdx.x], b = part[2 * threadIdx.x + 1];
    static_assert(kTmplSumBlocks == 64, "one wave");
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_down(a, off);
        b += __shfl_down(b, off);
    }
    if (threadIdx.x == 0) {
        out[0] = a;
        out[1] = b;
    }
}

// grid (frames, nband + interior blocks)
template <bool IDENT, int UR, int WAVES, bool ONE = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
    ecc_cols_kernel(const float *__restrict__ img, const float *__restrict__ tmpl, int rows, int cols,
                    const EccState *__restrict__ state, const int2 *__restrict__ rtab, double *__restrict__ partial,
                    const float *__restrict__ center, unsigned nband, int force_direct)
{
    __shared__ double lds_red[IDENT ? kEccChunk : ecc_lds_rows(WAVES)][256];     // source tile / reduction chunks of whichever body runs
    static_assert(ecc_lds_rows(WAVES) >= kEccChunk, "the tile area also holds the reduction chunks");
    static_assert(IDENT || WAVES <= 3 || sizeof(lds_red) * WAVES <= 160 * 1024, "the tiles of the workgroups of a compute unit fit its LDS");
    const int f = blockIdx.x;
    if (state[f].done) return;
    // (Round 6, measured and removed: a start delay of 0 / 1 / 2 units by a hash of the frame index for the interior workgroups of
    //  the general iteration, to put the three workgroups of a compute unit out of phase so that they hide each other's tile staging.
    //  Units of 3 / 6 / 12 x 0.85 us: 1.464 -> 1.507 / 1.551 / 1.645 ms per 512-frame launch, two alternations -- the delay is
    //  simply added; the workgroups are not waiting for each other's phase.)
    if (blockIdx.y >= nband)
        ecc_cols_body<IDENT, UR, ONE, ecc_tile_rows(WAVES)>(img, tmpl, rows, cols, state, rtab, partial, f, nband, blockIdx.y - nband,
                                                            gridDim.y - nband, lds_red, *center, force_direct);
    else
        ecc_band_cols_body(img, tmpl, rows, cols, state, rtab, partial, f, blockIdx.y, nband, IDENT, lds_red);
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

// The body of the cv::findTransformECC iteration after the image passes (ecc.cpp): meanStdDev, rho, hessian inverse,
// lambda, deltaP, update -- one lane, from the 45 sums in S.
__device__ __forceinline__ void ecc_solve_scalar(EccState &es, const double *S, int max_iters, double eps, int rows, int cols)
{
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

// One iteration's solve, one workgroup per frame: wave w reduces sums k = w, w+4, ... over the block partials (lane l
// takes blocks l, l+64, ...; fixed shuffle tree -> deterministic), thread 0 runs the scalar part, then -- for a frame that
// goes on iterating -- all threads write the per-row terms of the fixed-point source coordinate under the NEW matrix
// (WarpAffineInvoker: X0 = round((M01 y + M02) 1024), Y0 likewise; the sums kernel adds the per-column terms).
__global__ void __launch_bounds__(256)
    ecc_solve_kernel(EccState *__restrict__ state, const double *__restrict__ partial, int2 *__restrict__ rtab,
                     int nframes, int nblocks, int max_iters, double eps, int rows, int cols, int stride)
{
    const int f = blockIdx.x;
    if (f >= nframes) return;
    EccState &es = state[f];
    if (es.done) return;
    __shared__ double Ssh[kEccSums];
    __shared__ float Msh[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // every load of the wave's <= 12 sums is issued before the first reduction (one sum after the
    // other made this a chain of 12 global-load latencies)
    constexpr int kPerWave = (kEccSums + 3) / 4;
    double v[kPerWave];
#pragma unroll
    for (int i = 0; i < kPerWave; ++i) {
        const int k = wave + 4 * i;
        v[i] = (k < kEccSums && lane < nblocks) ? partial[((size_t)f * kEccSums + k) * stride + lane] : 0.0;
    }
    if (nblocks > 64) {
#pragma unroll
        for (int i = 0; i < kPerWave; ++i) {
            const int k = wave + 4 * i;
            if (k < kEccSums) {
                const double *pk = partial + ((size_t)f * kEccSums + k) * stride;
                for (int b = lane + 64; b < nblocks; b += 64) v[i] += pk[b];
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
    if (threadIdx.x == 0) {
        ecc_solve_scalar(es, Ssh, max_iters, eps, rows, cols);
#pragma unroll
        for (int i = 0; i < 6; ++i) Msh[i] = es.M[i];
        Msh[6] = es.done == 0 ? 1.f : 0.f;
    }
    __syncthreads();
    if (Msh[6] != 0.f) {
        const double M1 = Msh[1], M2 = Msh[2], M4 = Msh[4], M5 = Msh[5];
        int2 *rt = rtab + (size_t)f * rows;
        for (int y = threadIdx.x; y < rows; y += 256)
            rt[y] = make_int2(__double2int_rn((M1 * y + M2) * 1024), __double2int_rn((M4 * y + M5) * 1024));
    }
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

}  // namespace

int launch_ecc_center(const float *tmpl_blur, int rows, int cols, float *d_center, hipStream_t st)
{
    hipLaunchKernelGGL(ecc_center_kernel, dim3(1), dim3(256), 0, st, tmpl_blur, rows, cols, d_center);
    UPSP_HIP_CHECK(hipGetLastError());
    return UPSP_OK;
}

int launch_ecc_export(const EccState *state, int nb, float *d_warps, int wstride, int32_t *d_iters, int istride, hipStream_t st)
{
    hipLaunchKernelGGL(ecc_export_warps, dim3((nb + 63) / 64), dim3(64), 0, st, state, nb, d_warps, wstride, d_iters, istride);
    UPSP_HIP_CHECK(hipGetLastError());
    return UPSP_OK;
}

bool ecc_fused_blur_eligible(int rows, int cols)
{
    const char *e = std::getenv("UPSP_ECC_FUSED_BLUR");       // =0: pre-blur and identity iteration as two kernels (A/B, tests)
    return !(e && *e == '0') && rows >= 8 && cols >= 8 && rows < 32768 && cols < 32768 && (long long)rows * cols < (1ll << 29);
}

int launch_ecc_tmpl_sums(const float *tmpl_blur, int rows, int cols, double *d_part, double *d_out, hipStream_t st)
{
    hipLaunchKernelGGL(ecc_tmpl_sums_kernel, dim3(kTmplSumBlocks), dim3(256), 0, st, tmpl_blur, (size_t)rows * cols, d_part);
    hipLaunchKernelGGL(ecc_tmpl_sums_final, dim3(1), dim3(64), 0, st, (const double *)d_part, d_out);
    UPSP_HIP_CHECK(hipGetLastError());
    return UPSP_OK;
}

int launch_ecc_blur_ident(FrameScratch *s, int slot, const uint16_t *d_frames, float *dst, const float *tmpl_blur, const float *d_center,
                          const double *d_tsum, int nb, int rows, int cols, float k0, float k1, float k2, unsigned thresh,
                          unsigned *hot_count, unsigned *hot_pos, const unsigned *only_changed, const void *changes, int max_hot,
                          hipStream_t st)
{
    if (!s || slot < 0 || slot > 1 || nb > s->batch || !ecc_fused_blur_eligible(rows, cols) || (only_changed && (!changes || max_hot < 1)))
        return fail(UPSP_ERR_INVALID, "fused pre-blur: not set up for this geometry");
    // Rows per piece (= per float segment): 128, fewer on small images -- about as many wave items as a 1024^2 frame has (128), but not
    // under 16 rows (every piece blurs 6 rows more than it owns).  A property of the image geometry alone.  (The float error of a
    // frame's sums grows with the segment length: with 128-row segments on every image the soak's statistics, mostly images of a
    // few hundred pixels, were 1.7 x those of the two-kernel path, whose pieces on such images are 4 .. 32 rows.)
    const int strips = (cols + kFusedOwn - 1) / kFusedOwn;
    const int pieces_want = (128 + strips - 1) / strips;
    const int prows = std::min(kEccFlushLong, std::max(16, (rows + pieces_want - 1) / pieces_want));
    const int pieces = (rows + prows - 1) / prows;
    const int blocks = (strips * pieces + 3) / 4;
    const size_t words = (size_t)s->batch * kEccSums * blocks;
    if (s->partial_id_words < words) {
        UPSP_HIP_CHECK(hipStreamSynchronize(st));
        for (int k = 0; k < 2; ++k) {
            if (s->partial_id[k]) UPSP_HIP_CHECK(hipFree(s->partial_id[k]));
            s->partial_id[k] = nullptr;
            UPSP_HIP_CHECK(hipMalloc(&s->partial_id[k], words * sizeof(double)));
        }
        s->partial_id_words = words;
    }
    const dim3 grid((unsigned)nb, (unsigned)blocks);
    if (only_changed)
        hipLaunchKernelGGL((ecc_blur_ident_again_kernel<4>), dim3((unsigned)((nb + kAgainFrames - 1) / kAgainFrames), (unsigned)blocks),
                           dim3(256), 0, st, d_frames, dst, tmpl_blur, rows, cols, strips, pieces, prows, nb, s->partial_id[slot], d_center, k0, k1,
                           k2, only_changed, (const uint4 *)changes, max_hot, d_tsum);
    else if (hot_count)
        hipLaunchKernelGGL((ecc_blur_ident_kernel<true, 4, 4>), grid, dim3(256), 0, st, d_frames, dst, tmpl_blur, rows, cols, strips, pieces,
                           prows, s->partial_id[slot], d_center, k0, k1, k2, thresh, hot_count, hot_pos, d_tsum);
    else
        hipLaunchKernelGGL((ecc_blur_ident_kernel<false, 4, 4>), grid, dim3(256), 0, st, d_frames, dst, tmpl_blur, rows, cols, strips, pieces,
                           prows, s->partial_id[slot], d_center, k0, k1, k2, 0u, (unsigned *)nullptr, (unsigned *)nullptr, d_tsum);
    UPSP_HIP_CHECK(hipGetLastError());
    s->ident_for[slot] = dst;
    s->ident_blocks = blocks;
    return UPSP_OK;
}

int run_ecc(FrameScratch *s, const float *tmpl_blur, const float *d_center, const float *blurred, int nb, int64_t first_frame,
            int rows, int cols, int max_iters, double eps, hipStream_t st, const std::function<int()> *while_waiting)
{
    // cv::warpAffine saturates source coordinates to short: images of 32768 rows or columns are outside what the reference
    // itself registers; the kernels rely on it for 24-bit multiplies and the band-block count
    if (rows >= 32768 || cols >= 32768) return fail(UPSP_ERR_INVALID, "registration: image dimension >= 32768");
    if (!s || !s->state || !s->partial || !s->rtab || nb > s->batch) return fail(UPSP_ERR_INVALID, "registration: scratch not set up");
    // Every frame starts from the identity warp (cpp/lib/registration.cpp:52-53), so the first iteration needs no warp
    // and no interpolation (ecc_cols_kernel<true>); the later ones take their source taps from an LDS-staged tile.
    hipLaunchKernelGGL(ecc_init_kernel, dim3((nb + 63) / 64), dim3(64), 0, st, s->state, nb, (long long)first_frame, eps);
    // Workgroups per frame: one count per image geometry.  The float segments follow the row pieces, i.e. the block count:
    // it must not depend on how many frames are still iterating, or the last bits of a frame's sums -- and through the
    // reference's float 6 x 6 solve 1e-5 .. 1e-4 px of its warp -- would depend on which frames share its sub-batch.
    // Every block ends with a reduction of the 45 sums that costs as much as ~10 rows of its 256 columns.  32 interior blocks
    // per frame (4 column tiles x 8 row pieces of 128 rows at 1024^2) + 16 band blocks: with 256-frame sub-batches a launch
    // has 12 288 workgroups either way, and the fewer epilogues win -- ms per 1000 frames of configs[2] at (interior, band)
    // = (64, 48) / (32, 24) / (32, 16) / (16, 12) / (8, 12): 7.47 / 7.29 / 7.23 / 7.26 / 7.35.  (Round 3, 64-frame launches:
    // 16 / 32 / 64 interior blocks gave 6.74 / 6.25 / 6.03 ms of sums -- there the chip needed the blocks.)  At least one
    // interior block per column tile.
    // Images taller than 8 x kEccFlushLong rows get more row pieces, so that a piece stays within kEccFlushLong rows and the
    // interior blocks keep their one-flush form (2048^2: 8 tiles x 16 pieces = 128 blocks; up to kEccInteriorMax).
    const int tiles = (cols + 255) / 256;
    const int pieces_tall = (std::max(rows - 6, 1) + kEccFlushLong - 1) / kEccFlushLong;
    const int blocks = std::max(std::max(kEccInteriorBlocks, tiles), std::min(tiles * pieces_tall, kEccInteriorMax / tiles * tiles));
    const int nband = std::max(kEccBandBlocks, 3 * tiles);
    static_assert(kEccInteriorMax >= 128 && kEccBandMax >= 3 * 128, "column tiles of an image narrower than 32768");
    const int nblocks_total = blocks + nband;
    // UPSP_ECC_DIRECT=1 (test switch): every segment of the general iteration takes the direct loads instead of the LDS
    // tile -- the same bits (tests/test_imageops_gpu.py::test_ecc_lds_taps_same_bits)
    const char *direct_env = std::getenv("UPSP_ECC_DIRECT");
    const char *pairs_env = std::getenv("UPSP_ECC_PAIRS");       // =0 (A/B): the taps of the tile path row by row, as in rounds 4-5
    const int force_direct = ((direct_env && std::atoi(direct_env) != 0) ? 1 : 0) | ((pairs_env && *pairs_env == '0') ? 2 : 0);
    // the one-flush form of the interior blocks (ecc_cols_body): when a block's row piece has <= kEccFlushLong rows -- a property of
    // the image geometry alone, like the block count.  UPSP_ECC_ONE_FLUSH=0: the 32-row float segments of rounds 3-5 (A/B).
    const int pieces_min = std::max(blocks / tiles, 1);
    const char *one_env = std::getenv("UPSP_ECC_ONE_FLUSH");
    const bool one_flush_on = !(one_env && *one_env == '0');
    const bool one_flush = one_flush_on && (rows - 6 + pieces_min - 1) / pieces_min <= kEccFlushLong;
    // the second pass of the fused pre-blur over this buffer may still be running on its own stream (frame_scratch_preblur)
    for (int k = 0; k < 2; ++k)
        if (s->again_pending[k] && blurred == (k ? s->ecc_img2 : s->ecc_img)) {
            UPSP_HIP_CHECK(hipStreamWaitEvent(st, s->ev_again[k], 0));
            s->again_pending[k] = false;
        }
    // the identity iteration's sums came with the pre-blur (launch_ecc_blur_ident): one use per blurred buffer
    const double *ident_partial = nullptr;
    for (int k = 0; k < 2; ++k)
        if (s->ident_for[k] && s->ident_for[k] == blurred) {
            ident_partial = s->partial_id[k];
            s->ident_for[k] = nullptr;
        }
    bool first_burst = true, waited = false;
    int it = 0, iters_done = 0, most_iters = 0;
    for (;;) {
        // a few iterations between host checks of the active-frame count; frames that have converged exit at once
        // (a host check costs a stream round trip of ~40 us; most frames converge within 3-5 iterations, the rare
        // oscillating ones run to max_iters, so the bursts grow).  First burst: as many iterations as the previous
        // sub-batch's slowest frame took -- on steady footage every frame converges with its second iteration.
        const int burst = first_burst ? std::max(1, s->ecc_first_burst - it) : (it < 7 ? 2 : (it < 15 ? 8 : 16));
        first_burst = false;
        for (int k = 0; k < burst && it < max_iters; ++k, ++it) {
            const bool summed = it == 0 && ident_partial;
            if (!summed) {
                KTimed kt(it == 0 ? "ecc_sums_identity" : "ecc_sums_general", st);
                const dim3 grid((unsigned)nb, (unsigned)nblocks_total);
#define UPSP_ECC_LAUNCH(IDENT, UR, WAVES, ONE, FD)                                                                             \
    hipLaunchKernelGGL((ecc_cols_kernel<IDENT, UR, WAVES, ONE>), grid, dim3(256), 0, st, blurred, tmpl_blur, rows, cols,         \
                       (const EccState *)s->state, (const int2 *)s->rtab, s->partial, d_center, (unsigned)nband, FD)
                if (it == 0) {
                    if (one_flush) UPSP_ECC_LAUNCH(true, 4, 5, true, 0);
                    else UPSP_ECC_LAUNCH(true, 4, 4, false, 0);
                } else {
                    if (one_flush) UPSP_ECC_LAUNCH(false, 2, 4, true, force_direct);
                    else UPSP_ECC_LAUNCH(false, 2, 3, false, force_direct);
                }
#undef UPSP_ECC_LAUNCH
            }
            KTimed kt2("ecc_solve_kernel", st);
            hipLaunchKernelGGL(ecc_solve_kernel, dim3(nb), dim3(256), 0, st, s->state, summed ? ident_partial : (const double *)s->partial,
                               s->rtab, nb, summed ? s->ident_blocks : nblocks_total, max_iters, eps, rows, cols,
                               summed ? s->ident_blocks : kEccStride);
        }
        hipLaunchKernelGGL(ecc_count_active, dim3(1), dim3(64), 0, st, (const EccState *)s->state, nb, s->counter);
        // the read-back goes to pinned memory behind an event: whatever `while_waiting` enqueues (the next sub-batch's
        // hot-pixel repair and pre-blur) runs on the GPU while the host waits for these four words
        UPSP_HIP_CHECK(hipMemcpyAsync(s->h_counter, s->counter, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
        UPSP_HIP_CHECK(hipEventRecord(s->ev_counter, st));
        if (while_waiting && !waited) {
            waited = true;
            const int rcw = (*while_waiting)();
            if (rcw != UPSP_OK) return rcw;
        }
        UPSP_HIP_CHECK(hipEventSynchronize(s->ev_counter));
        if (s->h_counter[1] > 0)
            return fail(UPSP_ERR_DIVERGED, "ECC registration did not converge (cv::findTransformECC would throw)");
        iters_done = s->h_counter[2];
        most_iters = s->h_counter[3];
        if (s->h_counter[0] == 0 || it >= max_iters) break;
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
        std::fprintf(stderr, "[upsp] ECC sub-batch of %d frames: %d frame-iterations, max %d, %d launches\n", nb, tot, mx, it);
    }
    return UPSP_OK;
}

}  // namespace upsp
