/*
 * image_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Restatement of the image operators the psp_process frame loop calls in
 * un-vendored third-party libraries:
 *
 *   OpenCV 4.5.2 (docs/sphinx/dependencies.rst:35; opencv-python-headless==4.5.2.54,
 *   pyproject.toml:12):
 *     cv::GaussianBlur / cv::blur      call site cpp/exec/psp_process.cpp:1802-1807
 *     cv::findTransformECC             call site cpp/lib/registration.cpp:64
 *     cv::warpAffine                   call site cpp/lib/registration.cpp:69-73
 *   Eigen 3.3.9 (docs/sphinx/dependencies.rst:26):
 *     colPivHouseholderQr().solve()    call site cpp/lib/patches.ipp:204
 *
 * PARITY UNPINNED: none of these sources is under the reference tree and the
 * reference has no test for register_pixel, the filters or PatchClusters.  The
 * functions below restate the published algorithms (modules/video/src/ecc.cpp,
 * modules/imgproc/src/imgwarp.cpp, smooth.dispatch.cpp, box_filter;
 * Eigen/src/QR/ColPivHouseholderQR.h) with the precision of every intermediate
 * the libraries use (float images, double accumulators, float 6x6 LU ...).
 * Reduction orders inside the libraries' SIMD loops are not reproducible.
 */
#include "upsp_oracle.h"
#include "qr_f32.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------- helpers -- */

static int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        if (i >= n) i = 2 * n - 2 - i;
    }
    return i;
}

static int cv_round_d(double v) { return (int)lrint(v); }

/* ------------------------------------------------------------- filters -- */

/* cv::getGaussianKernel(k, sigma<=0, CV_32F) */
int orc_gaussian_kernel(int k, float *coef)
{
    static const float tab1[] = {1.f};
    static const float tab3[] = {0.25f, 0.5f, 0.25f};
    static const float tab5[] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};
    static const float tab7[] = {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f};
    if (k < 1 || (k & 1) == 0) return -1;
    const float *fixed = k == 1 ? tab1 : k == 3 ? tab3 : k == 5 ? tab5 : k == 7 ? tab7 : NULL;
    if (fixed) {
        memcpy(coef, fixed, sizeof(float) * (size_t)k);
        return 0;
    }
    double sigma = ((k - 1) * 0.5 - 1) * 0.3 + 0.8;
    double scale2 = -0.5 / (sigma * sigma);
    double sum = 0;
    for (int i = 0; i < k; ++i) {
        double x = i - (k - 1) * 0.5;
        double t = exp(scale2 * x * x);
        coef[i] = (float)t;
        sum += coef[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < k; ++i) coef[i] = (float)(coef[i] * sum);
    return 0;
}

/* GaussianBlur(src,dst,Size(k,k),0) (symmetric separable filter, float) or
 * blur(src,dst,Size(k,k)) (box: double running sums, scale 1/(k*k)); BORDER_REFLECT_101 */
void orc_blur_f32(const float *src, float *dst, int rows, int cols, int k, int box)
{
    const int r = k / 2;
    size_t np = (size_t)rows * cols;
    if (box) {
        double *tmp = (double *)malloc(sizeof(double) * np);
        for (int y = 0; y < rows; ++y)
            for (int x = 0; x < cols; ++x) {
                double s = 0;
                for (int j = -r; j <= r; ++j) s += src[(size_t)y * cols + reflect101(x + j, cols)];
                tmp[(size_t)y * cols + x] = s;
            }
        const double scale = 1.0 / ((double)k * k);
        for (int y = 0; y < rows; ++y)
            for (int x = 0; x < cols; ++x) {
                double s = 0;
                for (int j = -r; j <= r; ++j) s += tmp[(size_t)reflect101(y + j, rows) * cols + x];
                dst[(size_t)y * cols + x] = (float)(s * scale);
            }
        free(tmp);
        return;
    }
    float kc[64];
    float *kk = k <= 64 ? kc : (float *)malloc(sizeof(float) * (size_t)k);
    orc_gaussian_kernel(k, kk);
    float *tmp = (float *)malloc(sizeof(float) * np);
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            const float *row = src + (size_t)y * cols;
            float s = kk[r] * row[x];
            for (int j = 1; j <= r; ++j)
                s += kk[r + j] * (row[reflect101(x - j, cols)] + row[reflect101(x + j, cols)]);
            tmp[(size_t)y * cols + x] = s;
        }
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            float s = kk[r] * tmp[(size_t)y * cols + x];
            for (int j = 1; j <= r; ++j)
                s += kk[r + j] * (tmp[(size_t)reflect101(y - j, rows) * cols + x] +
                                  tmp[(size_t)reflect101(y + j, rows) * cols + x]);
            dst[(size_t)y * cols + x] = s;
        }
    free(tmp);
    if (kk != kc) free(kk);
}

/* ---------------------------------------------------------- warpAffine -- */

typedef struct {
    int sx, sy, ax, ay;
} warp_coord;

/* Fixed-point source coordinate of dst pixel (x,y): WarpAffineInvoker (imgwarp.cpp).
 * AB_BITS = 10, INTER_BITS = 5.  M is the float matrix widened to double. */
static warp_coord warp_coord_at(const double M[6], int x, int y, int interp)
{
    const int AB_BITS = 10, AB_SCALE = 1 << 10, INTER_BITS = 5, TAB = 32;
    const int round_delta = interp ? AB_SCALE / TAB / 2 : AB_SCALE / 2;
    int adelta = cv_round_d(M[0] * x * AB_SCALE);
    int bdelta = cv_round_d(M[3] * x * AB_SCALE);
    int X0 = cv_round_d((M[1] * y + M[2]) * AB_SCALE) + round_delta;
    int Y0 = cv_round_d((M[4] * y + M[5]) * AB_SCALE) + round_delta;
    warp_coord c;
    if (interp) {
        int X = (X0 + adelta) >> (AB_BITS - INTER_BITS);
        int Y = (Y0 + bdelta) >> (AB_BITS - INTER_BITS);
        c.sx = X >> INTER_BITS;
        c.sy = Y >> INTER_BITS;
        c.ax = X & (TAB - 1);
        c.ay = Y & (TAB - 1);
    } else {
        c.sx = (X0 + adelta) >> AB_BITS;
        c.sy = (Y0 + bdelta) >> AB_BITS;
        c.ax = c.ay = 0;
    }
    if (c.sx < -32768) c.sx = -32768; /* saturate_cast<short> */
    if (c.sx > 32767) c.sx = 32767;
    if (c.sy < -32768) c.sy = -32768;
    if (c.sy > 32767) c.sy = 32767;
    return c;
}

/* remapBilinear<Cast<float,T>, RemapNoVec, float>, BORDER_CONSTANT value 0 */
static float bilinear_at(const float *f32, const uint16_t *u16, int rows, int cols, warp_coord c)
{
#define PIX(yy, xx) (f32 ? f32[(size_t)(yy) * cols + (xx)] : (float)u16[(size_t)(yy) * cols + (xx)])
    const float fx = c.ax * (1.f / 32), fy = c.ay * (1.f / 32);
    const float w0 = (1.f - fy) * (1.f - fx), w1 = (1.f - fy) * fx, w2 = fy * (1.f - fx), w3 = fy * fx;
    const int sx = c.sx, sy = c.sy;
    if ((unsigned)sx < (unsigned)(cols - 1) && (unsigned)sy < (unsigned)(rows - 1))
        return PIX(sy, sx) * w0 + PIX(sy, sx + 1) * w1 + PIX(sy + 1, sx) * w2 + PIX(sy + 1, sx + 1) * w3;
    if (sx >= cols || sx + 1 < 0 || sy >= rows || sy + 1 < 0) return 0.f;
    float v0 = (sx >= 0 && sy >= 0 && sx < cols && sy < rows) ? PIX(sy, sx) : 0.f;
    float v1 = (sx + 1 >= 0 && sy >= 0 && sx + 1 < cols && sy < rows) ? PIX(sy, sx + 1) : 0.f;
    float v2 = (sx >= 0 && sy + 1 >= 0 && sx < cols && sy + 1 < rows) ? PIX(sy + 1, sx) : 0.f;
    float v3 = (sx + 1 >= 0 && sy + 1 >= 0 && sx + 1 < cols && sy + 1 < rows) ? PIX(sy + 1, sx + 1) : 0.f;
    return v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3;
#undef PIX
}

static void widen6(const float M[6], double D[6])
{
    for (int i = 0; i < 6; ++i) D[i] = M[i];
}

void orc_warp_affine_f32(const float *src, float *dst, int rows, int cols, const float M[6],
                         int interp)
{
    double D[6];
    widen6(M, D);
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            warp_coord c = warp_coord_at(D, x, y, interp);
            float v;
            if (interp)
                v = bilinear_at(src, NULL, rows, cols, c);
            else
                v = ((unsigned)c.sx < (unsigned)cols && (unsigned)c.sy < (unsigned)rows)
                        ? src[(size_t)c.sy * cols + c.sx]
                        : 0.f;
            dst[(size_t)y * cols + x] = v;
        }
}

void orc_warp_affine_u16(const uint16_t *src, uint16_t *dst, int rows, int cols, const float M[6],
                         int interp)
{
    double D[6];
    widen6(M, D);
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            warp_coord c = warp_coord_at(D, x, y, interp);
            uint16_t o;
            if (interp) {
                float v = bilinear_at(NULL, src, rows, cols, c);
                long iv = lrintf(v); /* saturate_cast<ushort>(float) */
                o = (uint16_t)(iv < 0 ? 0 : iv > 65535 ? 65535 : iv);
            } else {
                o = ((unsigned)c.sx < (unsigned)cols && (unsigned)c.sy < (unsigned)rows)
                        ? src[(size_t)c.sy * cols + c.sx]
                        : 0;
            }
            dst[(size_t)y * cols + x] = o;
        }
}

/* ------------------------------------------------------------------ ECC -- */

/* hal::LU32f based inverse of a 6x6 float matrix (Mat::inv, DECOMP_LU). 0 = singular */
static int inv6_f32(const float *Ain, float *inv)
{
    enum { n = 6 };
    float A[n][n], b[n][n];
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            A[i][j] = Ain[i * n + j];
            b[i][j] = i == j ? 1.f : 0.f;
        }
    const float eps = FLT_EPSILON * 10;
    for (int i = 0; i < n; ++i) {
        int k = i;
        for (int j = i + 1; j < n; ++j)
            if (fabsf(A[j][i]) > fabsf(A[k][i])) k = j;
        if (fabsf(A[k][i]) < eps) return 0;
        if (k != i) {
            for (int j = i; j < n; ++j) {
                float t = A[i][j];
                A[i][j] = A[k][j];
                A[k][j] = t;
            }
            for (int j = 0; j < n; ++j) {
                float t = b[i][j];
                b[i][j] = b[k][j];
                b[k][j] = t;
            }
        }
        float d = -1 / A[i][i];
        for (int j = i + 1; j < n; ++j) {
            float alpha = A[j][i] * d;
            for (int kk = i + 1; kk < n; ++kk) A[j][kk] += alpha * A[i][kk];
            for (int kk = 0; kk < n; ++kk) b[j][kk] += alpha * b[i][kk];
        }
    }
    for (int i = n - 1; i >= 0; --i)
        for (int j = 0; j < n; ++j) {
            float s = b[i][j];
            for (int k = i + 1; k < n; ++k) s -= A[i][k] * b[k][j];
            b[i][j] = s / A[i][i];
        }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) inv[i * n + j] = b[i][j];
    return 1;
}

/* y = A(6x6 float) * x(6 float): cv::gemm accumulates in double for CV_32F */
static void mat6_vec(const float *A, const float *x, float *y)
{
    for (int i = 0; i < 6; ++i) {
        double s = 0;
        for (int j = 0; j < 6; ++j) s += (double)A[i * 6 + j] * x[j];
        y[i] = (float)s;
    }
}

static double dot6(const float *a, const float *b)
{
    double s = 0;
    for (int i = 0; i < 6; ++i) s += (double)a[i] * b[i];
    return s;
}

/* cv::findTransformECC, MOTION_AFFINE, no input mask, gaussFiltSize = 5 */
int orc_find_transform_ecc(const float *ref, const float *inp, int rows, int cols, float M[6],
                           int max_iters, double eps, double *rho_out)
{
    const size_t np = (size_t)rows * cols;
    float *tmpl = (float *)malloc(sizeof(float) * np);   /* templateFloat */
    float *img = (float *)malloc(sizeof(float) * np);    /* imageFloat */
    float *gx = (float *)malloc(sizeof(float) * np);
    float *gy = (float *)malloc(sizeof(float) * np);
    float *w = (float *)malloc(sizeof(float) * np);      /* imageWarped */
    float *gxw = (float *)malloc(sizeof(float) * np);
    float *gyw = (float *)malloc(sizeof(float) * np);
    float *tz = (float *)malloc(sizeof(float) * np);     /* templateZM */
    uint8_t *mask = (uint8_t *)malloc(np);
    orc_blur_f32(ref, tmpl, rows, cols, 5, 0);
    orc_blur_f32(inp, img, rows, cols, 5, 0);
    /* preMask = ones blurred * (0.5/0.95) rounded = 1 everywhere -> gradients unchanged.
     * filter2D with [-0.5 0 0.5], BORDER_REFLECT_101 */
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            float a = img[(size_t)y * cols + reflect101(x - 1, cols)];
            float c = img[(size_t)y * cols + reflect101(x + 1, cols)];
            gx[(size_t)y * cols + x] = -0.5f * a + 0.5f * c;
            a = img[(size_t)reflect101(y - 1, rows) * cols + x];
            c = img[(size_t)reflect101(y + 1, rows) * cols + x];
            gy[(size_t)y * cols + x] = -0.5f * a + 0.5f * c;
        }

    double rho = -1, last_rho = -eps;
    int it, status = 0;
    for (it = 1; it <= max_iters && fabs(rho - last_rho) >= eps; ++it) {
        orc_warp_affine_f32(img, w, rows, cols, M, 1);
        orc_warp_affine_f32(gx, gxw, rows, cols, M, 1);
        orc_warp_affine_f32(gy, gyw, rows, cols, M, 1);
        double D[6];
        widen6(M, D);
        double n = 0, sw = 0, sww = 0, st = 0, stt = 0;
        for (int y = 0; y < rows; ++y)
            for (int x = 0; x < cols; ++x) {
                warp_coord c = warp_coord_at(D, x, y, 0);
                size_t i = (size_t)y * cols + x;
                mask[i] = ((unsigned)c.sx < (unsigned)cols && (unsigned)c.sy < (unsigned)rows) ? 1 : 0;
                if (mask[i]) {
                    n += 1;
                    sw += w[i];
                    sww += (double)w[i] * w[i];
                    st += tmpl[i];
                    stt += (double)tmpl[i] * tmpl[i];
                }
            }
        /* meanStdDev */
        double mw = n ? sw / n : 0, mt = n ? st / n : 0;
        double vw = n ? sww / n - mw * mw : 0, vt = n ? stt / n - mt * mt : 0;
        double sdw = sqrt(vw > 0 ? vw : 0), sdt = sqrt(vt > 0 ? vt : 0);
        const float mwf = (float)mw, mtf = (float)mt;
        for (size_t i = 0; i < np; ++i) {
            if (mask[i]) {
                w[i] = w[i] - mwf;
                tz[i] = tmpl[i] - mtf;
            } else {
                tz[i] = 0.f;
            }
        }
        const double tmpNorm = sqrt(n * sdt * sdt), imgNorm = sqrt(n * sdw * sdw);

        /* jacobian columns: gx*X, gy*X, gx*Y, gy*Y, gx, gy ; hessian, projections */
        double H[6][6] = {{0}}, ip[6] = {0}, tp[6] = {0}, corr = 0;
        for (int y = 0; y < rows; ++y)
            for (int x = 0; x < cols; ++x) {
                size_t i = (size_t)y * cols + x;
                const float X = (float)x, Y = (float)y;
                const float J[6] = {gxw[i] * X, gyw[i] * X, gxw[i] * Y, gyw[i] * Y, gxw[i], gyw[i]};
                for (int a = 0; a < 6; ++a) {
                    for (int b = a; b < 6; ++b) H[a][b] += (double)J[a] * J[b];
                    ip[a] += (double)J[a] * w[i];
                    tp[a] += (double)J[a] * tz[i];
                }
                corr += (double)tz[i] * w[i];
            }
        float Hf[36], Hinv[36], ipf[6], tpf[6], iph[6];
        for (int a = 0; a < 6; ++a)
            for (int b = 0; b < 6; ++b) Hf[a * 6 + b] = (float)(a <= b ? H[a][b] : H[b][a]);
        for (int a = 0; a < 6; ++a) {
            ipf[a] = (float)ip[a];
            tpf[a] = (float)tp[a];
        }
        if (!inv6_f32(Hf, Hinv)) memset(Hinv, 0, sizeof(Hinv)); /* Mat::inv returns zeros */
        last_rho = rho;
        rho = corr / (imgNorm * tmpNorm);
        if (rho != rho) {
            status = -1; /* "NaN encountered." */
            break;
        }
        mat6_vec(Hinv, ipf, iph);
        const double lambda_n = imgNorm * imgNorm - dot6(ipf, iph);
        const double lambda_d = corr - dot6(tpf, iph);
        if (lambda_d <= 0.0) {
            rho = -1;
            status = -2; /* "The algorithm stopped before its convergence..." */
            break;
        }
        const float lambda = (float)(lambda_n / lambda_d);
        /* error = lambda*templateZM - imageWarped ; errorProjection = J^T error */
        double ep[6] = {0};
        for (int y = 0; y < rows; ++y)
            for (int x = 0; x < cols; ++x) {
                size_t i = (size_t)y * cols + x;
                const float X = (float)x, Y = (float)y;
                const float J[6] = {gxw[i] * X, gyw[i] * X, gxw[i] * Y, gyw[i] * Y, gxw[i], gyw[i]};
                const float e = lambda * tz[i] - w[i];
                for (int a = 0; a < 6; ++a) ep[a] += (double)J[a] * e;
            }
        float epf[6], dp[6];
        for (int a = 0; a < 6; ++a) epf[a] = (float)ep[a];
        mat6_vec(Hinv, epf, dp);
        /* update_warping_matrix_ECC, MOTION_AFFINE */
        M[0] += dp[0];
        M[3] += dp[1];
        M[1] += dp[2];
        M[4] += dp[3];
        M[2] += dp[4];
        M[5] += dp[5];
    }
    if (rho_out) *rho_out = rho;
    free(tmpl); free(img); free(gx); free(gy); free(w); free(gxw); free(gyw); free(tz); free(mask);
    return status < 0 ? status : it - 1;
}

/* upsp::register_pixel, cpp/lib/registration.cpp:32-81 */
int orc_register_pixel_u16(const float *ref, const uint16_t *inp, int rows, int cols, float M[6],
                           int max_iters, double eps, int interp, uint16_t *out)
{
    const size_t np = (size_t)rows * cols;
    float *f = (float *)malloc(sizeof(float) * np);
    for (size_t i = 0; i < np; ++i) f[i] = (float)inp[i]; /* convertTo CV_32F */
    M[0] = 1; M[1] = 0; M[2] = 0; M[3] = 0; M[4] = 1; M[5] = 0; /* eye(2,3) */
    int it = orc_find_transform_ecc(ref, f, rows, cols, M, max_iters, eps, NULL);
    free(f);
    if (it < 0) return it;
    orc_warp_affine_u16(inp, out, rows, cols, M, interp);
    return it;
}

/* -------------------------------------------------------------- patches -- */

/* polyfit2D, cpp/lib/patches.ipp:172-205: A(ind, count) = pow(y,i)*pow(x,j), i+j<=3,
 * i outer / j inner; solve by Eigen::ColPivHouseholderQR<MatrixXf>. */
int orc_polyfit2d(const int32_t *x, const int32_t *y, const float *z, int m, float poly[10])
{
    enum { nc = 10 };
    if (m < nc) return -1;
    float *A = (float *)malloc(sizeof(float) * (size_t)m * nc); /* column-major like Eigen */
    float *c = (float *)malloc(sizeof(float) * (size_t)m);
    for (int r = 0; r < m; ++r) {
        int cnt = 0;
        for (int i = 0; i <= 3; ++i)
            for (int j = 0; j <= 3; ++j)
                if (i + j <= 3)
                    A[(size_t)cnt++ * m + r] = (float)pow((double)y[r], i) * (float)pow((double)x[r], j);
        c[r] = z[r];
    }
    const int nonzero = orc_colpiv_qr_solve_f32(A, c, m, nc, poly);
    free(A);
    free(c);
    return nonzero;
}

/* polyval2D, cpp/lib/patches.ipp:208-236 */
void orc_polyval2d(const int32_t *x, const int32_t *y, int n, const float poly[10], float *z)
{
    for (int r = 0; r < n; ++r) {
        float acc = 0;
        int cnt = 0;
        for (int i = 0; i <= 3; ++i)
            for (int j = 0; j <= 3; ++j)
                if (i + j <= 3) acc += poly[cnt++] * (float)pow((double)y[r], i) * (float)pow((double)x[r], j);
        z[r] = acc;
    }
}

/* PatchClusters<float>::operator(), cpp/lib/patches.ipp:98-165 */
void orc_patch_clusters(float *img, int cols, int nclusters, const int32_t *b_off,
                        const int32_t *bx, const int32_t *by, const int32_t *i_off,
                        const int32_t *ix, const int32_t *iy)
{
    for (int c = 0; c < nclusters; ++c) {
        int nb = b_off[c + 1] - b_off[c], ni = i_off[c + 1] - i_off[c];
        if (nb < 10) continue; /* too few points (:103) */
        float *z = (float *)malloc(sizeof(float) * (size_t)nb);
        for (int j = 0; j < nb; ++j)
            z[j] = img[(size_t)by[b_off[c] + j] * cols + bx[b_off[c] + j]];
        float poly[10];
        orc_polyfit2d(bx + b_off[c], by + b_off[c], z, nb, poly);
        float *zi = (float *)malloc(sizeof(float) * (size_t)(ni > 0 ? ni : 1));
        orc_polyval2d(ix + i_off[c], iy + i_off[c], ni, poly, zi);
        for (int j = 0; j < ni; ++j)
            img[(size_t)iy[i_off[c] + j] * cols + ix[i_off[c] + j]] = zi[j];
        free(z);
        free(zi);
    }
}
