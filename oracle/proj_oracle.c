/*
 * proj_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Restatement of the projection-matrix construction of psp_process
 * (cpp/exec/psp_process.cpp:167-355), camera model used by it
 * (cpp/lib/CameraCal.ipp:218-231, cpp/lib/CameraCal.cpp:192-203 -> OpenCV 4.5.2
 * cv::projectPoints, un-vendored, restated from its published formula), the
 * multi-camera weights (cpp/lib/projection.ipp:227-268, 911-1078) and
 * identify_skipped_nodes (cpp/lib/projection.ipp:857-880).
 */
#include "upsp_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* cv::projectPoints (calib3d cvProjectPoints2Internal) for one point: all double,
 * result stored to Point2f. */
void orc_project_point(const orc_camera *cam, const float xyz[3], float uv[2])
{
    const double *R = cam->R, *t = cam->t, *k = cam->dist;
    double X = xyz[0], Y = xyz[1], Z = xyz[2];
    double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
    double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
    double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
    z = z ? 1. / z : 1;
    x *= z;
    y *= z;
    double r2 = x * x + y * y;
    double r4 = r2 * r2;
    double r6 = r4 * r2;
    double a1 = 2 * x * y;
    double a2 = r2 + 2 * x * x;
    double a3 = r2 + 2 * y * y;
    double cdist = 1 + k[0] * r2 + k[1] * r4 + k[4] * r6;
    double xd = x * cdist + k[2] * a1 + k[3] * a2;
    double yd = y * cdist + k[2] * a3 + k[3] * a1;
    double fx = cam->K[0], fy = cam->K[4], cx = cam->K[2], cy = cam->K[5];
    uv[0] = (float)(xd * fx + cx);
    uv[1] = (float)(yd * fy + cy);
}

/* CameraCal::get_cam_center, cpp/lib/CameraCal.cpp:192-203 */
void orc_cam_center(const orc_camera *cam, double c[3])
{
    const double *R = cam->R, *t = cam->t;
    for (int i = 0; i < 3; ++i) c[i] = -(R[0 + i] * t[0] + R[3 + i] * t[1] + R[6 + i] * t[2]);
}

/* Imath normalize helpers (same semantics as rt_oracle.c) */
static float len3(const float v[3])
{
    float l2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    if (l2 < 2.0f * 1.17549435e-38f) {
        float ax = fabsf(v[0]), ay = fabsf(v[1]), az = fabsf(v[2]);
        float m = ax;
        if (m < ay) m = ay;
        if (m < az) m = az;
        if (m == 0.0f) return 0.0f;
        ax /= m;
        ay /= m;
        az /= m;
        return m * sqrtf(ax * ax + ay * ay + az * az);
    }
    return sqrtf(l2);
}
static void norm3(float v[3])
{
    float l = len3(v);
    if (l != 0.0f) {
        v[0] /= l;
        v[1] /= l;
        v[2] /= l;
    }
}

/* cv::Point2f -> cv::Point2i conversion = saturate_cast<int>(float) = cvRound(float): round half to
 * even through cvtss2si, which answers 0x80000000 ("integer indefinite") for NaN and for values
 * outside the int range -- a strongly distorted far-off node (|pt| ~ 1e10) is out of frame, it must
 * not wrap around into it. */
static int cv_round(float v)
{
    if (!(v >= -2147483648.0f && v < 2147483648.0f)) return INT_MIN;
    return (int)lrintf(v);
}

static int tri_has_node(const int32_t *tri_nodes3, int32_t prim, int32_t nidx)
{
    return tri_nodes3[prim * 3 + 0] == nidx || tri_nodes3[prim * 3 + 1] == nidx ||
           tri_nodes3[prim * 3 + 2] == nidx;
}

/* create_projection_mat, cpp/exec/psp_process.cpp:167-355 */
int64_t orc_create_projection(const orc_bvh *bvh, const orc_camera *cam, const float *nodes3,
                              const float *normals3, const uint8_t *datanode,
                              const int32_t *tri_nodes3, size_t nnodes, float oblique_thresh,
                              int32_t *pix, float *uv, uint8_t *nodecount, uint64_t *nrays,
                              int threads)
{
    const int W = cam->width, H = cam->height;
    double cc[3];
    orc_cam_center(cam, cc);
    const float orig[3] = {(float)cc[0], (float)cc[1], (float)cc[2]}; /* :193-194 */
    int64_t accepted = 0;
    uint64_t rays = 0;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    (void)threads;
#endif

#pragma omp parallel for schedule(dynamic, 500) num_threads(threads) reduction(+ : accepted, rays)
    for (long long n = 0; n < (long long)nnodes; ++n) {
        pix[n] = -1;
        uv[2 * n] = 0.f;
        uv[2 * n + 1] = 0.f;
        if (datanode && !datanode[n]) continue; /* :241 */

        const float *ipos = &nodes3[3 * n];
        float pt[2];
        orc_project_point(cam, ipos, pt); /* :248 */
        /* upsp::contains(Size, Point2i(pt)) :252 -> cvRound */
        int rx = cv_round(pt[0]), ry = cv_round(pt[1]);
        if (!(rx >= 0 && ry >= 0 && rx < W && ry < H)) continue;

        float dir[3] = {ipos[0] - orig[0], ipos[1] - orig[1], ipos[2] - orig[2]};
        norm3(dir); /* :256 */
        orc_ray ray;
        orc_ray_init(&ray, orig, dir);
        orc_hit hit;
        orc_hit_init(&hit);
        ++rays;
        if (!orc_bvh_intersect(bvh, &ray, &hit, NULL, NULL)) continue; /* :260-261 */

        const int32_t nidx = (int32_t)n;
        int visible = tri_has_node(tri_nodes3, hit.primID, nidx); /* :263-267 */
        if (!visible) {
            const float L = 1e-4f; /* :270-276 */
            static const float sp[6][3] = {{-1, 0, 0}, {1, 0, 0},  {0, -1, 0},
                                           {0, 1, 0},  {0, 0, -1}, {0, 0, 1}};
            for (int tidx = 0; !visible && tidx < 6; ++tidx) {
                float pos2[3] = {ipos[0] + sp[tidx][0] * L, ipos[1] + sp[tidx][1] * L,
                                 ipos[2] + sp[tidx][2] * L};
                /* un-normalised direction, :280-282 */
                float dir2[3] = {pos2[0] - orig[0], pos2[1] - orig[1], pos2[2] - orig[2]};
                orc_ray ray2;
                orc_ray_init(&ray2, orig, dir2);
                orc_hit hit2;
                orc_hit_init(&hit2);
                ++rays;
                if (!orc_bvh_intersect(bvh, &ray2, &hit2, NULL, NULL)) continue;
                visible = tri_has_node(tri_nodes3, hit2.primID, nidx);
            }
        }
        if (!visible) continue;

        /* oblique test :298-306.  `float theta = acos(cos_theta)` with a float argument (:304-305) in a translation unit
         * that includes <math.h> besides <cmath> (upsp.h:38 -> PSPHDF5.h:13; utils/pspError.h:5 -> pspOstr.h:13): with
         * libstdc++ that header brings std::acos's overloads into the global namespace, so the call resolves to
         * acos(float) = acosf -- checked with g++ 11 on a three-line TU with the same two includes.  acosf is
         * this platform's libm (glibc 2.35: fdlibm's float kernel, < 1 ulp; glibc >= 2.41: correctly rounded), i.e. the
         * last bit of theta is the libm's, not the reference's: orc_oblique_ambiguous() counts the nodes where that bit
         * decides (acosf and the correctly rounded double acos on opposite sides of the threshold). */
        const float *nn = &normals3[3 * n];
        float cos_theta = nn[0] * dir[0] + nn[1] * dir[1] + nn[2] * dir[2];
        float theta = acosf(cos_theta);
        if (!(theta > oblique_thresh)) continue;

        uv[2 * n] = pt[0] / W; /* :311-314 */
        uv[2 * n + 1] = pt[1] / H;
        int px = (int)roundf(pt[0]), py = (int)roundf(pt[1]); /* :319 std::round */
        int idx = py * W + px;                                /* :186 */
        /* the reference would index column -1 / row H when pt lands exactly on
         * x.5 ties where cvRound and round disagree; such nodes are rejected here */
        if (idx < 0 || idx >= W * H) continue;
        pix[n] = idx;
        ++accepted;
    }

    if (nodecount) { /* :335-347 */
        memset(nodecount, 0, (size_t)W * H);
        for (size_t n = 0; n < nnodes; ++n)
            if (pix[n] >= 0 && nodecount[pix[n]] < 255) nodecount[pix[n]]++;
    }
    if (nrays) *nrays = rays;
    return accepted;
}

/* In-frame data nodes whose oblique verdict (psp_process.cpp:304-306) depends on the last bit of acos: libm's acosf
 * (what the reference's call resolves to, see orc_create_projection) and the double-precision acos narrowed to float
 * (what the GPU engine evaluates) disagree on `theta > oblique_thresh`.  Such a node is outside any bit-exact claim
 * against "the reference" -- its entry depends on the libm the reference was linked with.  Counted regardless of
 * what the rays say (a superset of the nodes whose matrix entry could differ). */
int64_t orc_oblique_ambiguous(const orc_camera *cam, const float *nodes3, const float *normals3, const uint8_t *datanode,
                              size_t nnodes, float oblique_thresh)
{
    const int W = cam->width, H = cam->height;
    double cc[3];
    orc_cam_center(cam, cc);
    const float orig[3] = {(float)cc[0], (float)cc[1], (float)cc[2]};
    int64_t count = 0;
    for (size_t n = 0; n < nnodes; ++n) {
        if (datanode && !datanode[n]) continue;
        const float *ipos = &nodes3[3 * n];
        float pt[2];
        orc_project_point(cam, ipos, pt);
        int rx = cv_round(pt[0]), ry = cv_round(pt[1]);
        if (!(rx >= 0 && ry >= 0 && rx < W && ry < H)) continue;
        float dir[3] = {ipos[0] - orig[0], ipos[1] - orig[1], ipos[2] - orig[2]};
        norm3(dir);
        const float *nn = &normals3[3 * n];
        float cos_theta = nn[0] * dir[0] + nn[1] * dir[1] + nn[2] * dir[2];
        const int a = acosf(cos_theta) > oblique_thresh;
        const int b = (float)acos((double)cos_theta) > oblique_thresh;
        count += a != b;
    }
    return count;
}

/* angle_between<float>, cpp/utils/cv_extras.ipp:69-73 (dot in float, norms in double) */
static double angle_between_f(const float v1[3], const float v2[3])
{
    float dot = v1[0] * v2[0] + v1[1] * v2[1] + v1[2] * v2[2];
    double n1 = sqrt((double)v1[0] * v1[0] + (double)v1[1] * v1[1] + (double)v1[2] * v1[2]);
    double n2 = sqrt((double)v2[0] * v2[0] + (double)v2[1] * v2[1] + (double)v2[2] * v2[2]);
    double ang = dot / n1 / n2;
    return acos(ang);
}

/* adjust_projection_for_weights + BestView / AverageViews,
 * cpp/lib/projection.ipp:911-1078, 227-268.  Cameras that see a node are
 * visited in camera order (the reference's heap order among equal rows is
 * implementation-defined; it matters only for exactly equal angles). */
void orc_adjust_weights(int ncams, size_t nnodes, const int32_t *pix, float *weight,
                        const float *nodes3, const float *normals3, const double *centers3,
                        int mode)
{
    float *ang = (float *)malloc(sizeof(float) * (size_t)ncams);
    int *who = (int *)malloc(sizeof(int) * (size_t)ncams);
    for (size_t n = 0; n < nnodes; ++n) {
        int cnt = 0;
        for (int c = 0; c < ncams; ++c) {
            if (pix[(size_t)c * nnodes + n] < 0) continue;
            float center[3] = {(float)centers3[3 * c], (float)centers3[3 * c + 1],
                               (float)centers3[3 * c + 2]};
            float dir[3] = {nodes3[3 * n] - center[0], nodes3[3 * n + 1] - center[1],
                            nodes3[3 * n + 2] - center[2]};
            ang[cnt] = (float)angle_between_f(dir, &normals3[3 * n]);
            who[cnt++] = c;
        }
        if (cnt < 2) continue;
        if (mode == 0) { /* BestView: first maximum */
            int best = 0;
            for (int i = 1; i < cnt; ++i)
                if (ang[i] > ang[best]) best = i;
            for (int i = 0; i < cnt; ++i)
                weight[(size_t)who[i] * nnodes + n] *= (i == best) ? 1.0f : 0.0f;
        } else { /* AverageViews */
            float sum = 0.0f;
            for (int i = 0; i < cnt; ++i) sum += ang[i];
            for (int i = 0; i < cnt; ++i) weight[(size_t)who[i] * nnodes + n] *= ang[i] / sum;
        }
    }
    free(ang);
    free(who);
}

/* identify_skipped_nodes, cpp/lib/projection.ipp:857-880 */
size_t orc_skipped_nodes(int ncams, size_t nnodes, const int32_t *pix, uint8_t *skipped)
{
    size_t cnt = 0;
    for (size_t n = 0; n < nnodes; ++n) {
        int found = 0;
        for (int c = 0; c < ncams && !found; ++c) found = pix[(size_t)c * nnodes + n] >= 0;
        skipped[n] = (uint8_t)!found;
        cnt += !found;
    }
    return cnt;
}
