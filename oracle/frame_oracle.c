/*
 * frame_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Restatement of the integer / gather / accumulator work of the psp_process
 * phase-1 frame loop (cpp/exec/psp_process.cpp:1743-1851, 1926-1979),
 * upsp::fix_hot_pixels (cpp/utils/cv_extras.cpp:230-275), upsp::project_frame
 * (cpp/lib/projection.ipp:883-908), apportion / local_transpose
 * (cpp/exec/psp_process.cpp:611-624, 647-689).
 */
#include "upsp_oracle.h"

#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

/* upsp::fix_hot_pixels, cpp/utils/cv_extras.cpp:230-275.
 * Returns number of pixels replaced, -1 if more than max_hot pixels look hot
 * (the reference returns without touching the frame). */
int orc_fix_hot_pixels(uint16_t *img, int rows, int cols, int thresh, int min_change, int max_hot)
{
    int n_pix = rows * cols;
    int *locs = (int *)malloc(sizeof(int) * (size_t)(max_hot > 0 ? max_hot : 1));
    int n_hot = 0;
    for (int p = 0; p < n_pix; ++p) {
        if (img[p] >= thresh) {
            if (n_hot >= max_hot) {
                free(locs);
                return -1;
            }
            locs[n_hot++] = p;
        }
    }
    int replaced = 0;
    for (int h = 0; h < n_hot; ++h) {
        uint16_t vals[4];
        unsigned n_vals = 0;
        int row = locs[h] / cols, col = locs[h] % cols;
        if (row > 0) vals[n_vals++] = img[(row - 1) * cols + col];
        if (col > 0) vals[n_vals++] = img[row * cols + col - 1];
        if (row < rows - 1) vals[n_vals++] = img[(row + 1) * cols + col];
        if (col < cols - 1) vals[n_vals++] = img[row * cols + col + 1];
        for (unsigned i = 1; i < n_vals; ++i) { /* std::sort */
            uint16_t v = vals[i];
            unsigned j = i;
            while (j > 0 && vals[j - 1] > v) {
                vals[j] = vals[j - 1];
                --j;
            }
            vals[j] = v;
        }
        uint16_t old_val = img[row * cols + col];
        uint16_t new_val = vals[n_vals / 2];
        if ((int)old_val - (int)new_val > min_change) {
            img[row * cols + col] = new_val;
            ++replaced;
        }
    }
    free(locs);
    return replaced;
}

/* upsp::project_frame for a <=1-nnz-per-row matrix, cpp/lib/projection.ipp:883-908 :
 * Eigen row-major sparse * dense = sum over the row's entries of value*in[col],
 * starting from 0 -> for one entry: 0 + w*img (float). */
void orc_project_frame_f32(const float *img, const int32_t *pix, const float *weight,
                           size_t nnodes, float *out)
{
    for (size_t n = 0; n < nnodes; ++n) {
        float w = weight ? weight[n] : 1.0f;
        out[n] = (pix[n] >= 0) ? 0.0f + w * img[pix[n]] : 0.0f;
    }
}
void orc_project_frame_u16(const uint16_t *img, const int32_t *pix, const float *weight,
                           size_t nnodes, float *out)
{
    for (size_t n = 0; n < nnodes; ++n) {
        float w = weight ? weight[n] : 1.0f;
        out[n] = (pix[n] >= 0) ? 0.0f + w * (float)img[pix[n]] : 0.0f; /* convertTo CV_32F */
    }
}

/* psp_process.cpp:1827-1831 */
void orc_accumulate(const float *sol, size_t nnodes, double *sum, double *sumsq)
{
    for (size_t i = 0; i < nnodes; ++i) {
        sumsq[i] += (sol[i] * sol[i]); /* float product promoted to double */
        sum[i] += sol[i];
    }
}

/* The plain frame loop (one camera, no registration / patch / filter) the way the reference runs it,
 * psp_process.cpp:1742-1851: `#pragma omp parallel` with thread-private double accumulators,
 * `omp for schedule(dynamic, 1)` over the frames, partials merged under `omp critical`.  Per frame:
 * fix_hot_pixels in place (:1772), project_frame (:1810), NaN for the skipped nodes (:1822-1825),
 * accumulators (:1828-1831), row stored (:1837-1839; rows may be NULL -- the timed baseline keeps
 * only the accumulators).  The summation order over frames depends on the thread schedule, exactly
 * as in the reference (parity bar for avg / rms is relative). */
void orc_frame_loop_u16(uint16_t *frames, int nframes, int rows, int cols, const int32_t *pix,
                        const float *weight, const int32_t *skipped, size_t nskipped, size_t nnodes,
                        int thresh, int min_change, int max_hot, float *out_rows, double *sum,
                        double *sumsq, int threads, double *seconds)
{
    /* seconds (may be NULL): [0] = set-up of the thread-private accumulators (allocated and zeroed =
     * first touch of 16 B x N per thread), [1] = the frame loop proper, [2] = merge.  The baseline
     * quotes [1] per frame and [0] + [2] per run: a production run amortises the latter over
     * thousands of frames per rank. */
    const size_t npix = (size_t)rows * (size_t)cols;
    double t0 = omp_get_wtime(), t1 = t0, t2 = t0;
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
    {
        double *l_rms = (double *)calloc(nnodes, sizeof(double));
        double *l_avg = (double *)calloc(nnodes, sizeof(double));
        float *sol_own = out_rows ? NULL : (float *)malloc(sizeof(float) * nnodes);
        if (seconds) { /* timing runs only: touch the pages now, so that [1] is the steady state */
            memset(l_rms, 0, sizeof(double) * nnodes);
            memset(l_avg, 0, sizeof(double) * nnodes);
            if (sol_own) memset(sol_own, 0, sizeof(float) * nnodes);
#pragma omp barrier
#pragma omp master
            t1 = omp_get_wtime();
        }
#pragma omp for schedule(dynamic, 1) nowait
        for (int f = 0; f < nframes; ++f) {
            uint16_t *img = frames + (size_t)f * npix;
            float *sol = out_rows ? out_rows + (size_t)f * nnodes : sol_own;
            orc_fix_hot_pixels(img, rows, cols, thresh, min_change, max_hot);
            orc_project_frame_u16(img, pix, weight, nnodes, sol);
            for (size_t i = 0; i < nskipped; ++i) sol[skipped[i]] = NAN;
            for (size_t i = 0; i < nnodes; ++i) {
                l_rms[i] += (sol[i] * sol[i]);
                l_avg[i] += sol[i];
            }
        }
        if (seconds) {
#pragma omp barrier
#pragma omp master
            t2 = omp_get_wtime();
        }
#pragma omp critical
        for (size_t i = 0; i < nnodes; ++i) {
            sumsq[i] += l_rms[i];
            sum[i] += l_avg[i];
        }
        free(l_rms);
        free(l_avg);
        free(sol_own);
    }
    if (seconds) {
        seconds[0] = t1 - t0;
        seconds[1] = t2 - t1;
        seconds[2] = omp_get_wtime() - t2;
    }
}

/* psp_process.cpp:1933-1936 */
void orc_finals(const double *sum, const double *sumsq, size_t nnodes, uint64_t nframes,
                float *avg, float *rms)
{
    for (size_t i = 0; i < nnodes; ++i) {
        avg[i] = (float)(sum[i] / (double)nframes);
        rms[i] = (float)sqrt(sumsq[i] / (double)nframes);
    }
}

/* apportion, psp_process.cpp:611-624 */
void orc_apportion(int value, int nbins, int *start, int *extent)
{
    unsigned long block = (unsigned long)(value / nbins);
    unsigned long rem = (unsigned long)value - block * (unsigned long)nbins;
    unsigned long next = 0;
    for (unsigned long b = 0; b < (unsigned long)nbins; ++b) {
        start[b] = (int)next;
        extent[b] = (int)(block + (b < rem));
        next += (unsigned long)extent[b];
    }
}

/* local_transpose, psp_process.cpp:647-689 */
void orc_transpose(const float *src, int x_extent, int y_extent, float *dst)
{
    for (int y = 0; y < y_extent; ++y)
        for (int x = 0; x < x_extent; ++x)
            dst[(size_t)x * (size_t)y_extent + (size_t)y] = src[(size_t)y * (size_t)x_extent + (size_t)x];
}
