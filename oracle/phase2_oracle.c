/*
 * phase2_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * psp_process phase 2 restated: per node Iref/I, degree-d polynomial detrend over time
 * (upsp::TransPolyFitter<float>, cpp/lib/filtering.ipp:12-79), gain, delta-Cp, rms / avg
 * (cpp/exec/psp_process.cpp:2452-2507, finals :2537-2545).
 *
 * PARITY: the fit is Eigen 3.3.9 ColPivHouseholderQR<MatrixXf> (un-vendored) -> restated
 * in qr_f32.h; pinned by the reference's own known-answer test
 * cpp/test/test_filtering.cpp:19-113 (degree 6, 25 frames, 13 points, |fit - y| < 1e-4),
 * replayed in tests/test_phase2_oracle.py.
 */
#include "upsp_oracle.h"
#include "qr_f32.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* TransPolyFitter ctor, filtering.ipp:13-26: A(f,c) = pow((float)f / n_frames, c), stored
 * as float; std::pow(float, unsigned) evaluates in double. Column-major [ncoef][nframes]. */
void orc_transpoly_design(int nframes, int degree, float *A)
{
    const int nc = degree + 1;
    for (int f = 0; f < nframes; ++f) {
        const float x = (float)f / (float)(unsigned)nframes;
        for (int c = 0; c < nc; ++c) A[(size_t)c * nframes + f] = (float)pow((double)x, (double)c);
    }
}

/* TransPolyFitter::eval_fit for one point (n_pts = 1), filtering.ipp:48-79:
 * poly = colPivHouseholderQr(A).solve(y); fit = A * poly (float). */
int orc_transpoly_fit(const float *A, int nframes, int ncoef, const float *y, float *poly,
                      float *fit)
{
    float *W = (float *)malloc(sizeof(float) * (size_t)nframes * (size_t)(ncoef + 1));
    float *c = W + (size_t)nframes * ncoef;
    memcpy(W, A, sizeof(float) * (size_t)nframes * ncoef);
    memcpy(c, y, sizeof(float) * (size_t)nframes);
    const int rank = orc_colpiv_qr_solve_f32(W, c, nframes, ncoef, poly);
    free(W);
    if (fit)
        for (int f = 0; f < nframes; ++f) {
            float acc = 0.0f;
            for (int k = 0; k < ncoef; ++k) acc += A[(size_t)k * nframes + f] * poly[k];
            fit[f] = acc;
        }
    return rank;
}

/* PaintCalibration::get_gain, cpp/lib/non_cv_upsp.cpp:66-68 */
float orc_paint_gain(const float cal[6], float T, float Pss)
{
    return cal[0] + cal[1] * T + cal[2] * T * T + (cal[3] + cal[4] * T + cal[5] * T * T) * Pss;
}

/* phase-2 node loop, psp_process.cpp:2452-2507.  intensity_t / pressure_t: [nnodes][nframes]
 * node-major slices of this rank; iref = sol_avg_final, coverage, steady, model_temp indexed
 * like the slice.  sum / sumsq: double partials (local_avg / local_rms), gain_out f64.
 * Nodes without coverage: pressure row untouched, NaN partials. */
void orc_phase2(const float *intensity_t, size_t nnodes, int nframes, const float *iref,
                const float *coverage, const float *steady, const float *model_temp,
                const float cal[6], float qbar, float ps, int degree, float *pressure_t,
                double *sum, double *sumsq, double *gain_out, int threads)
{
    const int nc = degree + 1;
    float *A = (float *)malloc(sizeof(float) * (size_t)nframes * nc);
    orc_transpoly_design(nframes, degree, A);
#ifdef _OPENMP
#pragma omp parallel num_threads(threads > 0 ? threads : omp_get_max_threads())
#endif
    {
        float *y = (float *)malloc(sizeof(float) * (size_t)nframes * 2);
        float *fit = y + nframes;
        float poly[32];
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 64)
#endif
        for (long long i = 0; i < (long long)nnodes; ++i) {
            if (coverage[i] == 0) {                                   /* :2466-2472 */
                sum[i] = sumsq[i] = gain_out[i] = (double)NAN;
                continue;
            }
            const float Pss = qbar * steady[i] + ps;                  /* :2475 */
            const double gain = (double)orc_paint_gain(cal, model_temp[i], Pss);
            gain_out[i] = gain;
            const float *row = intensity_t + (size_t)i * nframes;
            for (int f = 0; f < nframes; ++f) y[f] = iref[i] / row[f]; /* :2479-2481 */
            orc_transpoly_fit(A, nframes, nc, y, poly, fit);          /* :2484 */
            double s = 0.0, ss = 0.0;
            float *out = pressure_t + (size_t)i * nframes;
            for (int f = 0; f < nframes; ++f) {
                const float pressure = (float)((double)(y[f] - fit[f]) * gain);    /* :2488 */
                const float cp = (float)((double)pressure * 12.0 * 12.0 / (double)qbar); /* :2491 */
                out[f] = cp;
                ss += (double)(cp * cp);                               /* :2495 */
                s += (double)cp;
            }
            sum[i] = s;
            sumsq[i] = ss;
        }
        free(y);
    }
    free(A);
}
