/*
 * patchsetup_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Phase-0 patch set-up of psp_process restated in C (InitializeImagePatches,
 * cpp/exec/psp_process.cpp:2088-2182): target visibility (getTargets :55-112), image
 * diameters (get_target_diameters :114-165), clustering (cluster_points,
 * cpp/lib/patches.ipp:239-276), pixel lists (get_target_boundary :279-327,
 * get_cluster_boundary :330-485, PatchClusters ctor :14-54, threshold_bounds :57-94) and the
 * histogram threshold (intensity_histc cpp/lib/image_processing.ipp:10-50, find_peaks /
 * first_min_threshold cpp/utils/clustering.ipp:9-96).
 *
 * PARITY: the reference ships no test for any of these (cpp/test/test_projection.h is a TODO
 * list); integer / pixel-list work is restated exactly, the float / double mix of every
 * expression follows the C++ types; projectPoints is OpenCV (un-vendored, restated in
 * proj_oracle.c).  PARITY UNPINNED except through the ray caster and kd-tree it calls.
 */
#include "upsp_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_PI 3.141592653589793 /* cpp/include/utils/general_utils.h:17-18 */

/* cvRound(float) = cvtss2si: NaN / beyond the int range -> 0x80000000 (out of frame), no wrap-around */
static int cv_round_f(float v)
{
    if (!(v >= -2147483648.0f && v < 2147483648.0f)) return (int)0x80000000u;
    return (int)lrint((double)v);
}

static int in_frame_i(int w, int h, float u, float v)
{ /* upsp::contains(cv::Size, cv::Point2i(Point2f)) cpp/lib/projection.cpp:10-13 */
    const int x = cv_round_f(u), y = cv_round_f(v);
    return x >= 0 && x < w && y >= 0 && y < h;
}

/* getTargets, psp_process.cpp:55-112.  keep[i] = 1 for targets that stay. */
void orc_get_targets(const orc_bvh *bvh, const orc_kdtree *kd, const orc_camera *cam,
                     const float *normals3, const float *xyz3, size_t n, float oblique_thresh,
                     uint8_t *keep)
{
    double cc[3];
    orc_cam_center(cam, cc);
    const float orig[3] = {(float)cc[0], (float)cc[1], (float)cc[2]};
    for (size_t i = 0; i < n; ++i) {
        keep[i] = 0;
        const float *p = &xyz3[3 * i];
        float uv[2];
        orc_project_point(cam, p, uv);
        if (uv[0] < 0 || uv[1] < 0 || uv[0] >= cam->width || uv[1] >= cam->height) continue;
        float dir[3] = {p[0] - orig[0], p[1] - orig[1], p[2] - orig[2]};
        const float len = sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
        if (len != 0) { dir[0] /= len; dir[1] /= len; dir[2] /= len; }
        orc_ray ray;
        orc_hit hit;
        orc_ray_init(&ray, orig, dir);
        orc_hit_init(&hit);
        if (!orc_bvh_intersect(bvh, &ray, &hit, NULL, NULL)) continue;
        if ((double)hit.t < (double)len - 1e-3) continue;       /* occluded, :90 */
        const double hp[3] = {hit.pos[0], hit.pos[1], hit.pos[2]};
        const int32_t near = orc_kd_nearest(kd, hp, NULL);
        const float *nr = &normals3[3 * (size_t)near];
        const float cos_theta = nr[0] * dir[0] + nr[1] * dir[1] + nr[2] * dir[2];
        const float ang = (float)acos((double)cos_theta);
        if (ang > oblique_thresh) keep[i] = 1;
    }
}

/* upsp::get_perpendicular, cpp/utils/cv_extras.ipp:29-66 */
static void perpendicular(const float vin[3], float out[3])
{
    out[0] = out[1] = out[2] = 0;
    const float norm = (float)sqrt((double)vin[0] * vin[0] + (double)vin[1] * vin[1] + (double)vin[2] * vin[2]);
    if (norm == 0) return;
    const float v[3] = {vin[0] / norm, vin[1] / norm, vin[2] / norm};
    const float a0 = fabsf(v[0]), a1 = fabsf(v[1]), a2 = fabsf(v[2]);
    const int m = a0 > a1 ? (a0 > a2 ? 0 : 2) : (a1 > a2 ? 1 : 2);
    if (m == 0) { out[1] = 1; out[0] = -(out[1] * v[1] + out[2] * v[2]) / v[0]; }
    else if (m == 1) { out[0] = 1; out[1] = -(out[0] * v[0] + out[2] * v[2]) / v[1]; }
    else { out[0] = 1; out[2] = -(out[0] * v[0] + out[1] * v[1]) / v[2]; }
    const double n2 = sqrt((double)out[0] * out[0] + (double)out[1] * out[1] + (double)out[2] * out[2]);
    for (int k = 0; k < 3; ++k) out[k] = (float)((double)out[k] / n2);
}

/* get_target_diameters, psp_process.cpp:114-165.  uv2 = projected target centres. */
void orc_target_diameters(const orc_kdtree *kd, const orc_camera *cam, const float *normals3,
                          const float *xyz3, const float *uv2, const float *diam_in, size_t n,
                          float *diam_out)
{
    for (size_t i = 0; i < n; ++i) {
        diam_out[i] = 0;
        if (diam_in[i] == 0.0f || !in_frame_i(cam->width, cam->height, uv2[2 * i], uv2[2 * i + 1])) continue;
        const float *p = &xyz3[3 * i];
        const double pos[3] = {p[0], p[1], p[2]};
        const float *nr = &normals3[3 * (size_t)orc_kd_nearest(kd, pos, NULL)];
        float a[3], b[3];
        perpendicular(nr, a);
        b[0] = a[1] * nr[2] - a[2] * nr[1];
        b[1] = a[2] * nr[0] - a[0] * nr[2];
        b[2] = a[0] * nr[1] - a[1] * nr[0];
        float theta = 0.0f, acc = 0.0f;
        for (int j = 0; j < 4; ++j) {
            const double ca = 0.5 * diam_in[i] * cosf(theta), sb = 0.5 * diam_in[i] * sinf(theta);
            float est[3];
            for (int k = 0; k < 3; ++k) {
                const float pa = (float)((double)a[k] * ca), pb = (float)((double)b[k] * sb);
                est[k] = (p[k] + pa) + pb;
            }
            float q[2];
            orc_project_point(cam, est, q);
            const float dx = q[0] - uv2[2 * i], dy = q[1] - uv2[2 * i + 1];
            acc = (float)((double)acc + 2.0 * sqrt((double)dx * dx + (double)dy * dy));
            theta = (float)((double)theta + 2 * ORC_PI / 4);
        }
        diam_out[i] = (float)((double)acc / 4.0);
    }
}

/* cluster_points, patches.ipp:239-276.  order[] lists the targets cluster after cluster in
 * discovery order; cl_off[c]..cl_off[c+1] delimit cluster c.  Returns the cluster count. */
int orc_cluster_points(const float *uv2, const float *diam, int n, int bound_pts, int32_t *order,
                       int32_t *cl_off)
{
    int *pts = (int *)malloc(sizeof(int) * (size_t)(n ? n : 1));
    int npts = n, nord = 0, ncl = 0;
    for (int i = 0; i < n; ++i) pts[i] = i;
    cl_off[0] = 0;
    while (npts) {
        const int start = nord;
        order[nord++] = pts[0];
        memmove(pts, pts + 1, sizeof(int) * (size_t)(npts - 1));
        --npts;
        for (int head = start; head < nord; ++head) {
            const int r = order[head];
            int w = 0;
            for (int k = 0; k < npts; ++k) {
                const int t = pts[k];
                const float dx = uv2[2 * r] - uv2[2 * t], dy = uv2[2 * r + 1] - uv2[2 * t + 1];
                const double dist = sqrt((double)dx * dx + (double)dy * dy);
                const double lim = (double)(float)bound_pts + 0.5 * (double)(diam[r] + diam[t]);
                if (dist <= lim) order[nord++] = t;
                else pts[w++] = t;
            }
            npts = w;
        }
        cl_off[++ncl] = nord;
    }
    free(pts);
    return ncl;
}

static void target_box(float u, float v, float d, int box[4])
{ /* get_target_boundary(targ, t_min, t_max), patches.ipp:279-285 */
    box[0] = (int)floor((double)u - 0.5 * d);
    box[1] = (int)floor((double)v - 0.5 * d);
    box[2] = (int)ceil((double)u + 0.5 * d);
    box[3] = (int)ceil((double)v + 0.5 * d);
}

typedef struct { int32_t *x, *y; size_t n, cap; } ptlist;
static void pl_push(ptlist *l, int x, int y)
{
    if (l->n == l->cap) {
        l->cap = l->cap ? 2 * l->cap : 256;
        l->x = (int32_t *)realloc(l->x, sizeof(int32_t) * l->cap);
        l->y = (int32_t *)realloc(l->y, sizeof(int32_t) * l->cap);
    }
    l->x[l->n] = x;
    l->y[l->n++] = y;
}

static void single_boundary(const int box[4], int bp, int buf, ptlist *in, ptlist *bd)
{ /* patches.ipp:288-327 */
    for (int x = box[0]; x <= box[2]; ++x)
        for (int y = box[1]; y <= box[3]; ++y) pl_push(in, x, y);
    for (int x = box[0] - bp - buf; x <= box[2] + bp + buf; ++x)
        for (int y = box[1] - bp - buf; y <= box[3] + bp + buf; ++y)
            if (x < box[0] - buf || x > box[2] + buf || y < box[1] - buf || y > box[3] + buf) pl_push(bd, x, y);
}

static int block_has2(const int *cl, int dy, int x0, int y0, int lx, int ly)
{
    for (int x = x0; x < x0 + lx; ++x)
        for (int y = y0; y < y0 + ly; ++y)
            if (cl[x * dy + y] == 2) return 1;
    return 0;
}

static void cluster_boundary(const int (*boxes)[4], int nt, unsigned bp, unsigned buf, ptlist *in, ptlist *bd)
{ /* patches.ipp:330-485 (unsigned window arithmetic kept) */
    int tmin[2] = {INT_MAX, INT_MAX}, tmax[2] = {0, 0};
    for (int i = 0; i < nt; ++i) {
        if (boxes[i][0] < tmin[0]) tmin[0] = boxes[i][0];
        if (boxes[i][1] < tmin[1]) tmin[1] = boxes[i][1];
        if (boxes[i][2] > tmax[0]) tmax[0] = boxes[i][2];
        if (boxes[i][3] > tmax[1]) tmax[1] = boxes[i][3];
    }
    tmin[0] -= (int)(bp + buf); tmin[1] -= (int)(bp + buf);
    tmax[0] += (int)(bp + buf); tmax[1] += (int)(bp + buf);
    const unsigned dx = (unsigned)(tmax[0] - tmin[0] + 1), dy = (unsigned)(tmax[1] - tmin[1] + 1);
    int *cl = (int *)calloc((size_t)dx * dy, sizeof(int));
    for (int i = 0; i < nt; ++i)
        for (int x = boxes[i][0] - tmin[0]; x <= boxes[i][2] - tmin[0]; ++x)
            for (int y = boxes[i][1] - tmin[1]; y <= boxes[i][3] - tmin[1]; ++y) cl[(unsigned)x * dy + (unsigned)y] = 2;
    for (unsigned x = 0; x < dx; ++x) {
        unsigned lo = dy, hi = dy;
        for (unsigned y = 0; y < dy; ++y) if (cl[x * dy + y] == 2) { lo = y; break; }
        if (lo == dy) continue;
        for (unsigned y = dy - 1;; --y) { if (cl[x * dy + y] == 2) { hi = y; break; } }
        for (unsigned y = lo; y <= hi; ++y) cl[x * dy + y] = 2;
    }
    for (unsigned y = 0; y < dy; ++y) {
        unsigned lo = dx, hi = dx;
        for (unsigned x = 0; x < dx; ++x) if (cl[x * dy + y] == 2) { lo = x; break; }
        if (lo == dx) continue;
        for (unsigned x = dx - 1;; --x) { if (cl[x * dy + y] == 2) { hi = x; break; } }
        for (unsigned x = lo; x <= hi; ++x) cl[x * dy + y] = 2;
    }
    for (unsigned x = 0; x < dx; ++x) {
        const unsigned min_x = x <= bp + buf ? 0 : x - bp - buf;
        const unsigned len_x = (x + bp + buf < dx - 1 ? x + bp + buf : dx - 1) - min_x + 1;
        const unsigned bmin_x = x <= buf ? 0 : x - buf;
        const unsigned blen_x = (x + buf < dx - 1 ? x + buf : dx - 1) - bmin_x + 1;
        for (unsigned y = 0; y < dy; ++y) {
            const unsigned min_y = y <= bp + buf ? 0 : y - bp - buf;
            const unsigned len_y = (y + bp + buf < dy - 1 ? y + bp + buf : dy - 1) - min_y + 1;
            if (cl[x * dy + y] == 2) { pl_push(in, (int)x + tmin[0], (int)y + tmin[1]); continue; }
            if (bp > 0 && buf > 0) {
                const unsigned bmin_y = y <= buf ? 0 : y - buf;
                const unsigned blen_y = (y + buf < dy - 1 ? y + buf : dy - 1) - bmin_y + 1;
                if (!block_has2(cl, (int)dy, (int)bmin_x, (int)bmin_y, (int)blen_x, (int)blen_y) &&
                    block_has2(cl, (int)dy, (int)min_x, (int)min_y, (int)len_x, (int)len_y)) {
                    pl_push(bd, (int)x + tmin[0], (int)y + tmin[1]);
                    cl[x * dy + y] = 1;
                }
                continue;
            }
            if (bp > 0 && block_has2(cl, (int)dy, (int)min_x, (int)min_y, (int)len_x, (int)len_y)) {
                pl_push(bd, (int)x + tmin[0], (int)y + tmin[1]);
                cl[x * dy + y] = 1;
            }
        }
    }
    free(cl);
}

/* intensity_histc, image_processing.ipp:10-50 (u16 image); edges[bins+1], counts[bins] */
void orc_intensity_histc(const uint16_t *img, size_t npix, unsigned depth, int bins, int32_t *edges,
                         int32_t *counts)
{
    if (depth > 16) depth = 16;
    const unsigned max_value = 1u << depth;
    if (bins == -1) bins = (int)max_value;
    const uint16_t bin_sz = (uint16_t)ceil((double)(max_value / (unsigned)bins));
    memset(counts, 0, sizeof(int32_t) * (size_t)bins);
    for (size_t i = 0; i < npix; ++i)
        if (img[i] < max_value) ++counts[img[i] / bin_sz];
    for (int i = 0; i <= bins; ++i) edges[i] = i * bin_sz;
}

/* find_peaks, clustering.ipp:9-60 (including the `break` that ends the scan) */
int orc_find_peaks(const double *data, int n, unsigned separation, uint32_t *peaks)
{
    int np = 0;
    if (n < 3) return 0;
    int plateau = 0;
    unsigned plateau_begin = 0;
    for (unsigned i = 1; i < (unsigned)n - 1; ++i) {
        if (isinf(data[i]) || (data[i] > data[i - 1] && data[i] > data[i + 1])) {
            if (np > 0 && (i - peaks[np - 1]) < separation) {
                if (data[peaks[np - 1]] < data[i]) peaks[np - 1] = i;
                break;
            }
            peaks[np++] = i;
        } else if (data[i] > data[i - 1] && data[i] == data[i + 1]) {
            plateau = 1;
            plateau_begin = i;
        } else if (plateau) {
            if (data[i] < data[i + 1]) {
                plateau = 0;
            } else if (data[i] > data[i + 1]) {
                plateau = 0;
                const unsigned pi = (i + plateau_begin) / 2;
                if (np > 0 && (pi - peaks[np - 1]) < separation) {
                    if (data[peaks[np - 1]] < data[pi]) peaks[np - 1] = pi;
                    break;
                }
                peaks[np++] = pi;
            }
        }
    }
    return np;
}

/* first_min_threshold, clustering.ipp:62-96 (int counts) */
unsigned orc_first_min_threshold(const int32_t *counts, int n, unsigned separation)
{
    double *d = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
    uint32_t *mx = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(n ? n : 1));
    uint32_t *mn = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(n ? n : 1));
    unsigned res = 0;
    for (int i = 0; i < n; ++i) d[i] = counts[i];
    const int nmax = orc_find_peaks(d, n, separation, mx);
    if (nmax) {
        for (int i = 0; i < n; ++i) d[i] = 1.0 / counts[i];
        const int nmin = orc_find_peaks(d, n, separation, mn);
        for (int i = 0; i < nmin; ++i)
            if (mn[i] > mx[0]) { res = mn[i]; break; }
    }
    free(d); free(mx); free(mn);
    return res;
}

/* PatchClusters ctor + threshold_bounds (patches.ipp:14-94) for clusters given by
 * orc_cluster_points.  ref: first frame u16 [rows][cols] or NULL (no thresholding).
 * Output arrays are malloc'ed (caller frees with orc_free): offsets have ncl+1 entries. */
void orc_patch_tables(const float *uv2, const float *diam, const int32_t *order,
                      const int32_t *cl_off, int ncl, int cols, int rows, unsigned bound_pts,
                      unsigned buffer, const uint16_t *ref, unsigned thresh, unsigned offset,
                      int32_t **b_off, int32_t **bx, int32_t **by, int32_t **i_off, int32_t **ix,
                      int32_t **iy)
{
    ptlist IN = {0}, BD = {0};
    *b_off = (int32_t *)calloc((size_t)ncl + 1, sizeof(int32_t));
    *i_off = (int32_t *)calloc((size_t)ncl + 1, sizeof(int32_t));
    for (int c = 0; c < ncl; ++c) {
        ptlist in = {0}, bd = {0};
        const int nt = cl_off[c + 1] - cl_off[c];
        int(*boxes)[4] = (int(*)[4])malloc(sizeof(int[4]) * (size_t)nt);
        for (int k = 0; k < nt; ++k) {
            const int t = order[cl_off[c] + k];
            target_box(uv2[2 * t], uv2[2 * t + 1], diam[t], boxes[k]);
        }
        if (nt > 1) cluster_boundary((const int(*)[4])boxes, nt, bound_pts, buffer, &in, &bd);
        else single_boundary(boxes[0], (int)bound_pts, (int)buffer, &in, &bd);
        free(boxes);
        for (size_t k = 0; k < in.n; ++k)
            if (in.x[k] >= 0 && in.x[k] < cols && in.y[k] >= 0 && in.y[k] < rows) pl_push(&IN, in.x[k], in.y[k]);
        for (size_t k = 0; k < bd.n; ++k) {
            const int x = bd.x[k], y = bd.y[k];
            if (!(x >= 0 && x < cols && y >= 0 && y < rows)) continue;
            if (ref) { /* threshold_bounds */
                const int y0 = y - (int)offset > 0 ? y - (int)offset : 0, x0 = x - (int)offset > 0 ? x - (int)offset : 0;
                const int x1 = x + (int)offset < cols - 1 ? x + (int)offset : cols - 1;
                const int y1 = y + (int)offset < rows - 1 ? y + (int)offset : rows - 1;
                double mn = 1e300;
                for (int yy = y0; yy <= y1; ++yy)
                    for (int xx = x0; xx <= x1; ++xx)
                        if (ref[(size_t)yy * cols + xx] < mn) mn = ref[(size_t)yy * cols + xx];
                if (mn < (double)thresh) continue;
            }
            pl_push(&BD, x, y);
        }
        free(in.x); free(in.y); free(bd.x); free(bd.y);
        (*i_off)[c + 1] = (int32_t)IN.n;
        (*b_off)[c + 1] = (int32_t)BD.n;
    }
    *ix = IN.x; *iy = IN.y; *bx = BD.x; *by = BD.y;
}

void orc_free(void *p) { free(p); }
