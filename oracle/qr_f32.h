/*
 * qr_f32.h -- CPU ORACLE helper (test infrastructure, NOT product code).
 *
 * Float least squares  min |A x - c|  by column-pivoted Householder QR, restating the
 * published algorithm of Eigen 3.3.9 ColPivHouseholderQR<MatrixXf>::compute() + solve()
 * (Eigen/src/QR/ColPivHouseholderQR.h, Eigen/src/Householder/Householder.h; un-vendored,
 * PARITY UNPINNED) with every intermediate kept in float like the template instantiation
 * the reference uses (cpp/lib/patches.ipp:204, cpp/lib/filtering.ipp:65-66).
 *
 * A is m x nc column-major and is destroyed, c (length m) is destroyed; x gets nc values.
 * Returns the numerical rank Eigen would use for the solve.
 */
#ifndef UPSP_ORACLE_QR_F32_H
#define UPSP_ORACLE_QR_F32_H

#include <float.h>
#include <math.h>
#include <stdlib.h>

static int orc_colpiv_qr_solve_f32(float *A, float *c, int m, int nc, float *x)
{
    const int size = m < nc ? m : nc;
    float *hcoef = (float *)malloc(sizeof(float) * (size_t)(3 * nc + size + 1));
    float *normU = hcoef + size + 1, *normD = normU + nc, *sol = normD + nc;
    int *trans = (int *)malloc(sizeof(int) * (size_t)(2 * nc));
    int *perm = trans + nc;
    float maxnorm = 0;
    for (int k = 0; k < nc; ++k) {
        float s = 0;
        for (int r = 0; r < m; ++r) s += A[(size_t)k * m + r] * A[(size_t)k * m + r];
        normD[k] = normU[k] = sqrtf(s);
        if (normU[k] > maxnorm) maxnorm = normU[k];
    }
    float th = maxnorm * FLT_EPSILON / (float)m;
    const float threshold_helper = th * th;
    const float downdate = sqrtf(FLT_EPSILON);
    int nonzero = size;
    for (int k = 0; k < size; ++k) {
        int big = k;
        for (int j = k + 1; j < nc; ++j)
            if (normU[j] > normU[big]) big = j;
        float bigsq = normU[big] * normU[big];
        if (nonzero == size && bigsq < threshold_helper * (float)(m - k)) nonzero = k;
        trans[k] = big;
        if (big != k) {
            for (int r = 0; r < m; ++r) {
                float t = A[(size_t)k * m + r];
                A[(size_t)k * m + r] = A[(size_t)big * m + r];
                A[(size_t)big * m + r] = t;
            }
            float t = normU[k]; normU[k] = normU[big]; normU[big] = t;
            t = normD[k]; normD[k] = normD[big]; normD[big] = t;
        }
        /* makeHouseholderInPlace on A(k:m, k) */
        float *col = &A[(size_t)k * m];
        float tail = 0;
        for (int r = k + 1; r < m; ++r) tail += col[r] * col[r];
        float c0 = col[k], beta, tau;
        if (tail <= FLT_MIN) {
            tau = 0;
            beta = c0;
            for (int r = k + 1; r < m; ++r) col[r] = 0;
        } else {
            beta = sqrtf(c0 * c0 + tail);
            if (c0 >= 0) beta = -beta;
            for (int r = k + 1; r < m; ++r) col[r] = col[r] / (c0 - beta);
            tau = (beta - c0) / beta;
        }
        hcoef[k] = tau;
        col[k] = beta;
        /* applyHouseholderOnTheLeft to the remaining columns */
        if (tau != 0)
            for (int j = k + 1; j < nc; ++j) {
                float *cj = &A[(size_t)j * m];
                float tmp = 0;
                for (int r = k + 1; r < m; ++r) tmp += col[r] * cj[r];
                tmp += cj[k];
                cj[k] -= tau * tmp;
                for (int r = k + 1; r < m; ++r) cj[r] -= tau * col[r] * tmp;
            }
        for (int j = k + 1; j < nc; ++j) {
            if (normU[j] != 0) {
                float temp = fabsf(A[(size_t)j * m + k]) / normU[j];
                temp = (1 + temp) * (1 - temp);
                temp = temp < 0 ? 0 : temp;
                float r2 = normU[j] / normD[j];
                float temp2 = temp * r2 * r2;
                if (temp2 <= downdate) {
                    float s = 0;
                    for (int r = k + 1; r < m; ++r) s += A[(size_t)j * m + r] * A[(size_t)j * m + r];
                    normD[j] = sqrtf(s);
                    normU[j] = normD[j];
                } else {
                    normU[j] *= sqrtf(temp);
                }
            }
        }
    }
    /* c = Q^T c */
    for (int k = 0; k < nonzero; ++k) {
        float *col = &A[(size_t)k * m];
        float tau = hcoef[k];
        if (tau == 0) continue;
        float tmp = 0;
        for (int r = k + 1; r < m; ++r) tmp += col[r] * c[r];
        tmp += c[k];
        c[k] -= tau * tmp;
        for (int r = k + 1; r < m; ++r) c[r] -= tau * col[r] * tmp;
    }
    /* back substitution on R(0:nonzero, 0:nonzero) */
    for (int i = 0; i < nc; ++i) sol[i] = 0;
    for (int i = nonzero - 1; i >= 0; --i) {
        float s = c[i];
        for (int j = i + 1; j < nonzero; ++j) s -= A[(size_t)j * m + i] * sol[j];
        sol[i] = s / A[(size_t)i * m + i];
    }
    /* undo the column permutation (sequence of transpositions) */
    for (int k = 0; k < nc; ++k) perm[k] = k;
    for (int k = 0; k < size; ++k) {
        int t = perm[k];
        perm[k] = perm[trans[k]];
        perm[trans[k]] = t;
    }
    for (int k = 0; k < nc; ++k) x[k] = 0;
    for (int k = 0; k < nonzero; ++k) x[perm[k]] = sol[k];
    free(hcoef);
    free(trans);
    return nonzero;
}

#endif
