// psp_process phase 2 on the device: node-major time series -> delta-Cp.
//
//   per node i (psp_process.cpp:2452-2507):
//     y[f]   = Iref[i] / I[i][f]
//     fit    = least-squares polynomial of degree d through y over f/F   (TransPolyFitter)
//     cp[f]  = (y[f] - fit[f]) * gain[i] * 144 / qbar
//     avg, rms over f (double partials)
//
// Pure streaming over the [nodes x frames] slice: one wave per node row, 256-byte coalesced
// reads and writes, 4 B read + 4 B written per sample.  The reference solves the same
// (degree+1)-column least-squares problem per node with a float column-pivoted QR of the
// Vandermonde matrix in x = f/F (cpp/lib/filtering.ipp:48-79).  The fit is the orthogonal
// projection of y onto the polynomials, so it is computed here from the normal equations in
// the centred variable t = 2 f/F - 1 (monomials in t on [-1,1): Gram matrix condition ~1e5,
// harmless in double): moments m_k = sum_f t^k y_f in double, c = G^-1 m with G^-1 built once
// on the host in long double, fit = Horner(c, t).  No per-node factorisation; agrees with the
// exact least-squares fit to ~1e-7 relative (float rounding of the result), closer than the
// float QR it replaces (whose error grows with F, ~1e-5 at F = 3000).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>

#include "ktimer.h"
#include "upsp_internal.h"

namespace upsp {
namespace {

constexpr int kMaxCoef = 8;     // degree <= 7 (the reference uses 6, psp_process.cpp:1099)
constexpr int kRowCache = 16;   // rows of <= 64*16 frames stay in registers between the passes

struct PolyBasis {
    int nc;                           // coefficients requested (degree + 1)
    double scale;                     // t = f * scale - 1,  scale = 2 / nframes
    double ginv[kMaxCoef][kMaxCoef];  // inverse Gram matrix of 1, t, .., t^(nc-1) over the samples
    double tx[kMaxCoef][kMaxCoef];    // monomials in t -> monomials in x = f/nframes
};

// Inverse of the Gram matrix in long double (Gauss-Jordan, partial pivoting).  With fewer
// frames than coefficients only the leading F x F block is used: the fit then interpolates.
void build_basis(int nframes, int nc, PolyBasis &b)
{
    std::memset(&b, 0, sizeof(b));
    b.nc = nc;
    b.scale = 2.0 / (double)nframes;
    const int ne = nc < nframes ? nc : nframes;
    long double S[2 * kMaxCoef] = {0};
    for (int f = 0; f < nframes; ++f) {
        const long double t = (long double)f * b.scale - 1.0L;
        long double p = 1.0L;
        for (int k = 0; k < 2 * ne - 1; ++k) {
            S[k] += p;
            p *= t;
        }
    }
    long double G[kMaxCoef][2 * kMaxCoef];
    for (int i = 0; i < ne; ++i)
        for (int j = 0; j < ne; ++j) {
            G[i][j] = S[i + j];
            G[i][ne + j] = i == j ? 1.0L : 0.0L;
        }
    for (int c = 0; c < ne; ++c) {
        int piv = c;
        for (int r = c + 1; r < ne; ++r)
            if (fabsl(G[r][c]) > fabsl(G[piv][c])) piv = r;
        for (int j = 0; j < 2 * ne; ++j) std::swap(G[c][j], G[piv][j]);
        const long double d = G[c][c];
        for (int j = 0; j < 2 * ne; ++j) G[c][j] /= d;
        for (int r = 0; r < ne; ++r)
            if (r != c) {
                const long double m = G[r][c];
                for (int j = 0; j < 2 * ne; ++j) G[r][j] -= m * G[c][j];
            }
    }
    for (int i = 0; i < ne; ++i)
        for (int j = 0; j < ne; ++j) b.ginv[i][j] = (double)G[i][ne + j];
    // (2x - 1)^k = sum_c tx[k][c] x^c
    long double row[kMaxCoef] = {1.0L};
    for (int k = 0; k < nc; ++k) {
        for (int c = 0; c < kMaxCoef; ++c) b.tx[k][c] = (double)row[c];
        long double nxt[kMaxCoef];
        for (int c = 0; c < kMaxCoef; ++c) nxt[c] = -row[c] + (c > 0 ? 2.0L * row[c - 1] : 0.0L);
        for (int c = 0; c < kMaxCoef; ++c) row[c] = nxt[c];
    }
}

struct P2Args {
    const float *in;
    long long ld_in;
    float *out;
    long long ld_out;
    unsigned nnodes;
    int nframes;
    const float *iref, *coverage, *steady, *model_temp;
    float temp_scalar;
    float cal[6];
    float qbar, ps;
    double *sum, *sumsq;
    float *avg, *rms, *gain;
    float *poly;   // [nnodes][nc] monomial coefficients (optional)
    PolyBasis b;
};

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);   // same bits in every lane
    return v;
}

// moments of one sample: m[k] += t^k * y
template <int NC>
__device__ __forceinline__ void add_moments(double t, double y, double m[NC])
{
    double p = y;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        m[k] += p;
        p *= t;
    }
}

template <int NC>
__device__ __forceinline__ double horner(const double c[NC], double t)
{
    double acc = c[NC - 1];
#pragma unroll
    for (int k = NC - 2; k >= 0; --k) acc = fma(acc, t, c[k]);
    return acc;
}

// MODE 0: out = fit(in)             (TransPolyFitter::eval_fit on raw rows)
// MODE 1: out = delta-Cp of Iref/in (phase-2 node loop)
template <int NC, int MODE, bool CACHED>
__global__ void __launch_bounds__(256) phase2_kernel(const P2Args a)
{
    const int lane = threadIdx.x & 63;
    const unsigned node = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (node >= a.nnodes) return;   // whole wave leaves together; no workgroup barriers below
    const int F = a.nframes;
    const float *row = a.in + (long long)node * a.ld_in;
    float *orow = a.out + (long long)node * a.ld_out;
    const int niter = (F + 63) >> 6;

    if (MODE == 1 && a.coverage[node] == 0.0f) {          // psp_process.cpp:2466-2472
        const float qnan = __builtin_nanf("");
        for (int j = 0; j < niter; ++j) {
            const int f = j * 64 + lane;
            if (f < F) orow[f] = qnan;
        }
        if (lane == 0) {
            if (a.sum) a.sum[node] = (double)qnan;
            if (a.sumsq) a.sumsq[node] = (double)qnan;
            if (a.avg) a.avg[node] = qnan;
            if (a.rms) a.rms[node] = qnan;
            if (a.gain) a.gain[node] = qnan;
            if (a.poly)
                for (int c = 0; c < NC; ++c) a.poly[(size_t)node * NC + c] = 0.0f;  // skip_fit
        }
        return;
    }
    const float iref = MODE == 1 ? a.iref[node] : 0.0f;
    double m[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) m[k] = 0.0;
    float yc[CACHED ? kRowCache : 1];

    // pass 1: moments of y against 1, t, .., t^(NC-1)
    if (CACHED) {
        // every load of the row is issued before the first use (clamped index, no branches);
        // lanes past the end contribute y = 0
        float v[kRowCache];
#pragma unroll
        for (int j = 0; j < kRowCache; ++j) {
            const int f = j * 64 + lane;
            v[j] = row[f < F ? f : F - 1];
        }
#pragma unroll
        for (int j = 0; j < kRowCache; ++j) {
            const int f = j * 64 + lane;
            const float q = MODE == 1 ? iref / v[j] : v[j];   // :2479-2481
            const float y = f < F ? q : 0.0f;
            add_moments<NC>(fma((double)f, a.b.scale, -1.0), (double)y, m);
            yc[j] = y;
        }
    } else {
        for (int j = 0; j < niter; ++j) {
            const int f = j * 64 + lane;
            if (f < F) {
                const float v = row[f];
                const float y = MODE == 1 ? iref / v : v;
                add_moments<NC>(fma((double)f, a.b.scale, -1.0), (double)y, m);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) m[k] = wave_sum(m[k]);
    double coef[NC];   // polynomial in t
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        double c = 0.0;
#pragma unroll
        for (int j = 0; j < NC; ++j) c = fma(a.b.ginv[k][j], m[j], c);
        coef[k] = c;
    }

    double gain = 0.0;
    if (MODE == 1) {
        const float steady = a.steady ? a.steady[node] : 0.0f;
        const float T = a.model_temp ? a.model_temp[node] : a.temp_scalar;
        const float Pss = a.qbar * steady + a.ps;          // :2475
        // PaintCalibration::get_gain, non_cv_upsp.cpp:66-68 (float, same association)
        const float g = a.cal[0] + a.cal[1] * T + a.cal[2] * T * T +
                        (a.cal[3] + a.cal[4] * T + a.cal[5] * T * T) * Pss;
        gain = (double)g;
    }

    // pass 2: evaluate, detrend, scale, reduce
    double s = 0.0, ss = 0.0;
    // (p * 12 * 12 is exact in double for a float p; the division by qbar becomes a multiply
    // by its reciprocal: same float result unless the double quotient sits within one double
    // ulp of a float rounding boundary)
    const double cp_scale = 144.0 / (double)a.qbar;
    auto emit = [&](int f, float y) {
        const float fit = (float)horner<NC>(coef, fma((double)f, a.b.scale, -1.0));
        if (MODE == 0) {
            orow[f] = fit;
        } else {
            const float pressure = (float)((double)(y - fit) * gain);                    // :2488
            const float cp = (float)((double)pressure * cp_scale);                       // :2491
            orow[f] = cp;
            const double cpd = (double)cp;
            ss = fma(cpd, cpd, ss);    // :2495 (exact square; the reference rounds it to float first)
            s += cpd;
        }
    };
    if (CACHED) {
#pragma unroll
        for (int j = 0; j < kRowCache; ++j) {
            const int f = j * 64 + lane;
            if (f < F) emit(f, yc[j]);
        }
    } else {
        for (int j = 0; j < niter; ++j) {
            const int f = j * 64 + lane;
            if (f < F) {
                const float v = row[f];
                emit(f, MODE == 1 ? iref / v : v);
            }
        }
    }
    if (MODE == 1) {
        s = wave_sum(s);
        ss = wave_sum(ss);
    }
    if (lane == 0) {
        if (MODE == 1) {
            if (a.sum) a.sum[node] = s;
            if (a.sumsq) a.sumsq[node] = ss;
            if (a.avg) a.avg[node] = (float)(s / (double)F);            // :2540
            if (a.rms) a.rms[node] = (float)sqrt(ss / (double)F);       // :2541
            if (a.gain) a.gain[node] = (float)gain;                     // :2542
        }
        if (a.poly) {
            for (int c = 0; c < NC; ++c) {
                double v = 0.0;
                for (int k = 0; k < NC; ++k) v += coef[k] * a.b.tx[k][c];
                a.poly[(size_t)node * NC + c] = (float)v;
            }
        }
    }
}

template <int MODE>
int launch(const P2Args &a, hipStream_t st)
{
    const dim3 grid((a.nnodes + 3u) / 4u), block(256);
    const bool cached = a.nframes <= 64 * kRowCache;
    KTimed kt(MODE ? "phase2_kernel" : "transpoly_kernel", st);
#define UPSP_P2(NC)                                                                          \
    case NC:                                                                                 \
        if (cached)                                                                          \
            hipLaunchKernelGGL((phase2_kernel<NC, MODE, true>), grid, block, 0, st, a);      \
        else                                                                                 \
            hipLaunchKernelGGL((phase2_kernel<NC, MODE, false>), grid, block, 0, st, a);     \
        break;
    switch (a.b.nc) {
        UPSP_P2(1) UPSP_P2(2) UPSP_P2(3) UPSP_P2(4) UPSP_P2(5) UPSP_P2(6) UPSP_P2(7) UPSP_P2(8)
        default: return fail(UPSP_ERR_INVALID, "polynomial degree must be 0..7");
    }
#undef UPSP_P2
    UPSP_HIP_CHECK(hipGetLastError());
    return UPSP_OK;
}

int check_common(const void *in, const void *out, long long ld_in, long long ld_out, size_t nnodes,
                 int nframes, int degree)
{
    if (!in || !out) return fail(UPSP_ERR_INVALID, "null device buffer");
    if (nframes <= 0) return fail(UPSP_ERR_INVALID, "nframes must be positive");
    if (ld_in < nframes || ld_out < nframes) return fail(UPSP_ERR_INVALID, "row stride < nframes");
    if (degree < 0 || degree + 1 > kMaxCoef) return fail(UPSP_ERR_INVALID, "polynomial degree must be 0..7");
    if (nnodes > 0xFFFFFFF0ull) return fail(UPSP_ERR_INVALID, "too many nodes for one launch");
    return UPSP_OK;
}

}  // namespace
}  // namespace upsp

using namespace upsp;

extern "C" {

int upsp_transpoly_fit(const float *d_data_t, long long ld_in, size_t npts, int nframes, int degree,
                       float *d_fit_t, long long ld_out, float *d_poly, void *stream)
{
    if (npts == 0) return UPSP_OK;
    int rc = check_common(d_data_t, d_fit_t, ld_in, ld_out, npts, nframes, degree);
    if (rc != UPSP_OK) return rc;
    P2Args a;
    std::memset(&a, 0, sizeof(a));
    a.in = d_data_t; a.ld_in = ld_in; a.out = d_fit_t; a.ld_out = ld_out;
    a.nnodes = (unsigned)npts; a.nframes = nframes; a.poly = d_poly;
    build_basis(nframes, degree + 1, a.b);
    return launch<0>(a, (hipStream_t)stream);
}

int upsp_phase2_pressure(const float *d_intensity_t, long long ld_in, size_t nnodes, int nframes,
                         const float *d_iref, const float *d_coverage, const float *d_steady,
                         const float *d_model_temp, float model_temp, const float paint_cal[6],
                         float qbar, float ps, int degree, float *d_pressure_t, long long ld_out,
                         double *d_sum, double *d_sumsq, float *d_avg, float *d_rms, float *d_gain,
                         void *stream)
{
    if (nnodes == 0) return UPSP_OK;
    int rc = check_common(d_intensity_t, d_pressure_t, ld_in, ld_out, nnodes, nframes, degree);
    if (rc != UPSP_OK) return rc;
    if (!d_iref || !d_coverage || !paint_cal) return fail(UPSP_ERR_INVALID, "null Iref / coverage / paint calibration");
    P2Args a;
    std::memset(&a, 0, sizeof(a));
    a.in = d_intensity_t; a.ld_in = ld_in; a.out = d_pressure_t; a.ld_out = ld_out;
    a.nnodes = (unsigned)nnodes; a.nframes = nframes;
    a.iref = d_iref; a.coverage = d_coverage; a.steady = d_steady; a.model_temp = d_model_temp;
    a.temp_scalar = model_temp;
    for (int i = 0; i < 6; ++i) a.cal[i] = paint_cal[i];
    a.qbar = qbar; a.ps = ps;
    a.sum = d_sum; a.sumsq = d_sumsq; a.avg = d_avg; a.rms = d_rms; a.gain = d_gain;
    build_basis(nframes, degree + 1, a.b);
    return launch<1>(a, (hipStream_t)stream);
}

}  // extern "C"
