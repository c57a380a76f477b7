/*
 * rt_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference ray caster cpp/raycast/pspRT.cpp
 * (types in cpp/include/utils/pspRT.h) including the un-vendored Imath 3.1.x
 * pieces it calls (V3f::length/normalize, Box3f, Line3f, intersects(Box,Line)).
 * Build with -ffp-contract=off: the double-precision fallback of the triangle
 * test triggers on U/V/W == 0.0f exactly, so no FMA contraction is allowed.
 *
 * Pinned by tests/test_oracle_kat.py against the reference's own known-answer
 * tests (test/python/test_visibility.py).
 */
#include "upsp_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------- Imath subset --- */

/* Imath::Vec3<float>::length(): sqrt(dot) unless dot < 2*FLT_MIN (lengthTiny) */
static float v3_length(const float v[3])
{
    float l2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    if (l2 < 2.0f * FLT_MIN) {
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

/* Imath::Vec3<float>::normalize(): divide by length unless it is zero */
static void v3_normalize(float v[3])
{
    float l = v3_length(v);
    if (l != 0.0f) {
        v[0] /= l;
        v[1] /= l;
        v[2] /= l;
    }
}

/* Imath::Line3f(p0,p1): pos=p0, dir=(p1-p0).normalize(); the reference builds
 * Line3f(ray.o, ray.o + ray.d) at every node visit (pspRT.cpp:382). */
void orc_line_dir(const float o[3], const float d[3], float ldir[3])
{
    for (int a = 0; a < 3; ++a) {
        float p1 = o[a] + d[a];
        ldir[a] = p1 - o[a];
    }
    v3_normalize(ldir);
}

/* Imath::intersects(const Box3f&, const Line3f&, V3f& ip), ImathBoxAlgo.h
 * (ip is unused by the reference and not computed here). */
int orc_box_hit(const float bmin[3], const float bmax[3], const float pos[3], const float dir[3])
{
    /* b.isEmpty() */
    if (bmax[0] < bmin[0] || bmax[1] < bmin[1] || bmax[2] < bmin[2]) return 0;
    /* b.intersects(r.pos): origin inside (inclusive) */
    if (pos[0] >= bmin[0] && pos[0] <= bmax[0] && pos[1] >= bmin[1] && pos[1] <= bmax[1] &&
        pos[2] >= bmin[2] && pos[2] <= bmax[2])
        return 1;

    const float TMAX = FLT_MAX;
    float tFrontMax = -1.0f;
    float tBackMin = TMAX;

    for (int a = 0; a < 3; ++a) {
        if (dir[a] > 0.0f) {
            if (pos[a] > bmax[a]) return 0;
            float d = bmax[a] - pos[a];
            if (dir[a] > 1.0f || d < TMAX * dir[a]) {
                float t = d / dir[a];
                if (tBackMin > t) tBackMin = t;
            }
            if (pos[a] <= bmin[a]) {
                float d2 = bmin[a] - pos[a];
                float t = (dir[a] > 1.0f || d2 < TMAX * dir[a]) ? d2 / dir[a] : TMAX;
                if (tFrontMax < t) tFrontMax = t;
            }
        } else if (dir[a] < 0.0f) {
            if (pos[a] < bmin[a]) return 0;
            float d = bmin[a] - pos[a];
            if (dir[a] < -1.0f || d > TMAX * dir[a]) {
                float t = d / dir[a];
                if (tBackMin > t) tBackMin = t;
            }
            if (pos[a] >= bmax[a]) {
                float d2 = bmax[a] - pos[a];
                float t = (dir[a] < -1.0f || d2 > TMAX * dir[a]) ? d2 / dir[a] : TMAX;
                if (tFrontMax < t) tFrontMax = t;
            }
        } else {
            if (pos[a] < bmin[a] || pos[a] > bmax[a]) return 0;
        }
    }
    return tFrontMax <= tBackMin;
}

/* ------------------------------------------------------------ Ray/Hit --- */

/* rt::Hit::Hit(), pspRT.cpp:21-22 */
void orc_hit_init(orc_hit *h)
{
    memset(h, 0, sizeof(*h));
    h->t = FLT_MAX;
    h->geomID = -1;
    h->primID = -1;
}

static int max_dim(float ax, float ay, float az)
{
    /* MAX_DIM macro, pspRT.cpp:41 */
    return (ax > ay) ? (ax > az ? 0 : 2) : (ay > az ? 1 : 2);
}

/* rt::Ray::Ray(o,d), pspRT.cpp:45-69 */
void orc_ray_init(orc_ray *r, const float o[3], const float d[3])
{
    for (int a = 0; a < 3; ++a) {
        r->o[a] = o[a];
        r->d[a] = d[a];
    }
    r->kz = max_dim(fabsf(d[0]), fabsf(d[1]), fabsf(d[2]));
    r->kx = r->kz + 1;
    if (r->kx == 3) r->kx = 0;
    r->ky = r->kx + 1;
    if (r->ky == 3) r->ky = 0;
    if (d[r->kz] < 0.0f) {
        int tmp = r->kx;
        r->kx = r->ky;
        r->ky = tmp;
    }
    r->Sx = d[r->kx] / d[r->kz];
    r->Sy = d[r->ky] / d[r->kz];
    r->Sz = 1.0f / d[r->kz];
    r->inv[0] = 1.0f / d[0];
    r->inv[1] = 1.0f / d[1];
    r->inv[2] = 1.0f / d[2];
}

/* ----------------------------------------------------------- Triangle --- */

/* rt::Triangle::intersect, pspRT.cpp:109-193 */
int orc_tri_intersect(const orc_ray *ray, const float *oA, const float *oB, const float *oC,
                      int32_t primID, orc_hit *hit)
{
    float A[3], B[3], C[3];
    for (int a = 0; a < 3; ++a) {
        A[a] = oA[a] - ray->o[a];
        B[a] = oB[a] - ray->o[a];
        C[a] = oC[a] - ray->o[a];
    }
    const int kx = ray->kx, ky = ray->ky, kz = ray->kz;
    const float Ax = A[kx] - ray->Sx * A[kz];
    const float Ay = A[ky] - ray->Sy * A[kz];
    const float Bx = B[kx] - ray->Sx * B[kz];
    const float By = B[ky] - ray->Sy * B[kz];
    const float Cx = C[kx] - ray->Sx * C[kz];
    const float Cy = C[ky] - ray->Sy * C[kz];

    float U = Cx * By - Cy * Bx;
    float V = Ax * Cy - Ay * Cx;
    float W = Bx * Ay - By * Ax;

    if (U == 0.0f || V == 0.0f || W == 0.0f) {
        double CxBy = (double)Cx * (double)By;
        double CyBx = (double)Cy * (double)Bx;
        U = (float)(CxBy - CyBx);
        double AxCy = (double)Ax * (double)Cy;
        double AyCx = (double)Ay * (double)Cx;
        V = (float)(AxCy - AyCx);
        double BxAy = (double)Bx * (double)Ay;
        double ByAx = (double)By * (double)Ax;
        W = (float)(BxAy - ByAx);
    }

    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return 0;

    float det = U + V + W;
    if (det == 0.0f) return 0;

    const float Az = ray->Sz * A[kz];
    const float Bz = ray->Sz * B[kz];
    const float Cz = ray->Sz * C[kz];
    const float T = U * Az + V * Bz + W * Cz;

    float xorf_T = fabsf(T);
    if (signbit(T) != signbit(det)) xorf_T = -xorf_T;

    const float ray_near = 0.0f;
    const float hit_t = INFINITY;
    float abs_det = fabsf(det);
    if (xorf_T < ray_near * abs_det || hit_t * abs_det < xorf_T) return 0;

    const float rcpDet = 1.0f / det;
    hit->u = U * rcpDet;
    hit->v = V * rcpDet;
    hit->w = W * rcpDet;
    hit->t = T * rcpDet;
    for (int a = 0; a < 3; ++a) hit->pos[a] = ray->o[a] + hit->t * ray->d[a];
    hit->primID = primID;

    float e1[3], e2[3];
    for (int a = 0; a < 3; ++a) {
        e1[a] = oB[a] - oA[a];
        e2[a] = oC[a] - oA[a];
    }
    const float EPS = 1e-3f;
    if (v3_length(e1) < EPS || v3_length(e2) < EPS) {
        hit->nrm[0] = hit->nrm[1] = hit->nrm[2] = 0.0f;
    } else {
        /* Imath cross: (y*v.z - z*v.y, z*v.x - x*v.z, x*v.y - y*v.x) */
        float n0 = e1[1] * e2[2] - e1[2] * e2[1];
        float n1 = e1[2] * e2[0] - e1[0] * e2[2];
        float n2 = e1[0] * e2[1] - e1[1] * e2[0];
        float dot = n0 * ray->d[0] + n1 * ray->d[1] + n2 * ray->d[2];
        if (dot > 0.0f) {
            n0 = -n0;
            n1 = -n1;
            n2 = -n2;
        }
        hit->nrm[0] = n0;
        hit->nrm[1] = n1;
        hit->nrm[2] = n2;
    }
    return 1;
}

/* ---------------------------------------------------------- BVH build --- */

typedef struct {
    float mn[3], mx[3];
} box3;

static void box_empty(box3 *b)
{
    /* Imath::Box::makeEmpty(): min = +max(), max = lowest() */
    for (int a = 0; a < 3; ++a) {
        b->mn[a] = FLT_MAX;
        b->mx[a] = -FLT_MAX;
    }
}
static void box_extend_pt(box3 *b, const float *p)
{
    for (int a = 0; a < 3; ++a) {
        if (p[a] < b->mn[a]) b->mn[a] = p[a];
        if (p[a] > b->mx[a]) b->mx[a] = p[a];
    }
}
static void box_extend_box(box3 *b, const box3 *o)
{
    for (int a = 0; a < 3; ++a) {
        if (o->mn[a] < b->mn[a]) b->mn[a] = o->mn[a];
        if (o->mx[a] > b->mx[a]) b->mx[a] = o->mx[a];
    }
}
static int box_is_empty(const box3 *b)
{
    return b->mx[0] < b->mn[0] || b->mx[1] < b->mn[1] || b->mx[2] < b->mn[2];
}
static void box_size(const box3 *b, float s[3])
{
    if (box_is_empty(b)) {
        s[0] = s[1] = s[2] = 0.0f;
        return;
    }
    for (int a = 0; a < 3; ++a) s[a] = b->mx[a] - b->mn[a];
}
static int box_major_axis(const box3 *b)
{
    float s[3];
    box_size(b, s);
    int major = 0;
    for (int a = 1; a < 3; ++a)
        if (s[a] > s[major]) major = a;
    return major;
}
/* rt::SurfaceArea, pspRT.cpp:266-270 */
static float surface_area(const box3 *b)
{
    float d[3];
    box_size(b, d);
    return 2 * (d[0] * d[1] + d[0] * d[2] + d[1] * d[2]);
}
/* rt::Offset(...)[dim], pspRT.cpp:257-264 */
static float offset_dim(const box3 *b, const float *p, int dim)
{
    float o = p[dim] - b->mn[dim];
    if (b->mx[dim] > b->mn[dim]) o /= b->mx[dim] - b->mn[dim];
    return o;
}

typedef struct {
    size_t prim;
    box3 bounds;
    float centroid[3];
} prim_info;

typedef struct build_node {
    box3 bounds;
    struct build_node *child[2];
    int split_axis, first_prim, nprims;
} build_node;

typedef struct {
    prim_info *info;
    int32_t *ordered;
    int n_ordered;
    int total_nodes;
} build_ctx;

#define NBUCKETS 12
#define PRIMS_PER_LEAF 4

static int bucket_of(const box3 *cb, const prim_info *pi, int dim)
{
    int b = (int)(NBUCKETS * offset_dim(cb, pi->centroid, dim));
    if (b == NBUCKETS) b = NBUCKETS - 1;
    return b;
}

static build_node *make_leaf(build_ctx *cx, build_node *node, int bgn, int end, const box3 *bounds)
{
    node->first_prim = cx->n_ordered;
    for (int i = bgn; i < end; ++i) cx->ordered[cx->n_ordered++] = (int32_t)cx->info[i].prim;
    node->nprims = end - bgn;
    node->bounds = *bounds;
    node->child[0] = node->child[1] = NULL;
    return node;
}

/* rt::BVH::recursiveBuild, pspRT.cpp:456-572 */
static build_node *recursive_build(build_ctx *cx, int bgn, int end)
{
    build_node *node = (build_node *)calloc(1, sizeof(build_node));
    cx->total_nodes++;

    box3 bounds;
    box_empty(&bounds);
    for (int i = bgn; i < end; ++i) box_extend_box(&bounds, &cx->info[i].bounds);

    int nprims = end - bgn;
    if (nprims <= PRIMS_PER_LEAF) return make_leaf(cx, node, bgn, end, &bounds);

    box3 cb;
    box_empty(&cb);
    for (int i = bgn; i < end; ++i) box_extend_pt(&cb, cx->info[i].centroid);
    int dim = box_major_axis(&cb);

    if (cb.mx[dim] == cb.mn[dim]) return make_leaf(cx, node, bgn, end, &bounds);

    int count[NBUCKETS];
    box3 bb[NBUCKETS];
    for (int i = 0; i < NBUCKETS; ++i) {
        count[i] = 0;
        box_empty(&bb[i]);
    }
    for (int i = bgn; i < end; ++i) {
        int b = bucket_of(&cb, &cx->info[i], dim);
        count[b]++;
        box_extend_box(&bb[b], &cx->info[i].bounds);
    }

    float cost[NBUCKETS - 1];
    for (int i = 0; i < NBUCKETS - 1; ++i) {
        box3 b0, b1;
        box_empty(&b0);
        box_empty(&b1);
        int c0 = 0, c1 = 0;
        for (int j = 0; j <= i; ++j) {
            box_extend_box(&b0, &bb[j]);
            c0 += count[j];
        }
        for (int j = i + 1; j < NBUCKETS; ++j) {
            box_extend_box(&b1, &bb[j]);
            c1 += count[j];
        }
        cost[i] = 1.f + ((float)c0 * surface_area(&b0) + (float)c1 * surface_area(&b1)) /
                            surface_area(&bounds);
    }
    float min_cost = cost[0];
    int min_bucket = 0;
    for (int i = 1; i < NBUCKETS - 1; ++i) {
        if (cost[i] < min_cost) {
            min_cost = cost[i];
            min_bucket = i;
        }
    }
    /* shouldSplit is always true here (nprims > PRIMS_PER_LEAF), pspRT.cpp:540 */

    /* std::partition as implemented by libstdc++ for bidirectional iterators
     * (the order of elements inside a leaf, hence tie-breaking between equal-t
     * hits of one leaf, follows it). */
    prim_info *first = &cx->info[bgn], *last = &cx->info[end];
    for (;;) {
        for (;;) {
            if (first == last) goto done;
            if (bucket_of(&cb, first, dim) <= min_bucket)
                ++first;
            else
                break;
        }
        --last;
        for (;;) {
            if (first == last) goto done;
            if (!(bucket_of(&cb, last, dim) <= min_bucket))
                --last;
            else
                break;
        }
        prim_info tmp = *first;
        *first = *last;
        *last = tmp;
        ++first;
    }
done:;
    int mid = (int)(first - cx->info);
    if (mid == bgn || mid == end) {
        /* cannot happen for finite inputs (buckets 0 and 11 are non-empty); NaN
         * vertices would recurse forever in the reference -- stop with a leaf */
        return make_leaf(cx, node, bgn, end, &bounds);
    }

    node->split_axis = dim;
    node->nprims = 0;
    node->child[0] = recursive_build(cx, bgn, mid);
    node->child[1] = recursive_build(cx, mid, end);
    node->bounds = node->child[0]->bounds;
    box_extend_box(&node->bounds, &node->child[1]->bounds);
    return node;
}

/* rt::BVH::flattenTree, pspRT.cpp:433-454 */
static int flatten(orc_node *nodes, build_node *bn, int *offset, int depth, int *maxdepth)
{
    orc_node *ln = &nodes[*offset];
    for (int a = 0; a < 3; ++a) {
        ln->bmin[a] = bn->bounds.mn[a];
        ln->bmax[a] = bn->bounds.mx[a];
    }
    int my = (*offset)++;
    if (depth > *maxdepth) *maxdepth = depth;
    if (bn->nprims > 0) {
        ln->offset = bn->first_prim;
        ln->nprims = (uint16_t)bn->nprims;
        ln->axis = 0;
    } else {
        ln->axis = (uint8_t)bn->split_axis;
        ln->nprims = 0;
        flatten(nodes, bn->child[0], offset, depth + 1, maxdepth);
        ln->offset = flatten(nodes, bn->child[1], offset, depth + 1, maxdepth);
    }
    ln->pad = 0;
    free(bn);
    return my;
}

/* rt::CreateTriangleMesh + rt::BVH::BVH, pspRT.cpp:206-222, 313-344 */
orc_bvh *orc_bvh_create(const float *tris9, size_t ntris)
{
    if (ntris == 0) return NULL;
    orc_bvh *b = (orc_bvh *)calloc(1, sizeof(orc_bvh));
    b->ntris = ntris;
    b->verts = (float *)malloc(sizeof(float) * 9 * ntris);
    memcpy(b->verts, tris9, sizeof(float) * 9 * ntris);
    b->prim_ids = (int32_t *)malloc(sizeof(int32_t) * ntris);

    build_ctx cx;
    cx.info = (prim_info *)malloc(sizeof(prim_info) * ntris);
    cx.ordered = b->prim_ids;
    cx.n_ordered = 0;
    cx.total_nodes = 0;
    for (size_t i = 0; i < ntris; ++i) {
        prim_info *pi = &cx.info[i];
        pi->prim = i;
        box_empty(&pi->bounds);
        /* Triangle::bounds(), pspRT.cpp:96-107 */
        box_extend_pt(&pi->bounds, &tris9[9 * i + 0]);
        box_extend_pt(&pi->bounds, &tris9[9 * i + 3]);
        box_extend_pt(&pi->bounds, &tris9[9 * i + 6]);
        /* PrimitiveInfo ctor, pspRT.cpp:247-251 */
        for (int a = 0; a < 3; ++a)
            pi->centroid[a] = .5f * pi->bounds.mn[a] + .5f * pi->bounds.mx[a];
    }
    build_node *root = recursive_build(&cx, 0, (int)ntris);
    b->nnodes = cx.total_nodes;
    b->nodes = (orc_node *)malloc(sizeof(orc_node) * (size_t)b->nnodes);
    int offset = 0, maxdepth = 0;
    flatten(b->nodes, root, &offset, 0, &maxdepth);
    b->depth = maxdepth;
    free(cx.info);
    return b;
}

void orc_bvh_destroy(orc_bvh *b)
{
    if (!b) return;
    free(b->verts);
    free(b->prim_ids);
    free(b->nodes);
    free(b);
}

/* ------------------------------------------------------ BVH intersect --- */

/* rt::BVH::intersect, pspRT.cpp:359-431 */
int orc_bvh_intersect(const orc_bvh *b, const orc_ray *ray, orc_hit *isect,
                      uint32_t *nodes_visited, uint32_t *tris_tested)
{
    int anyhit = 0;
    int dirIsNeg[3] = {ray->inv[0] < 0, ray->inv[1] < 0, ray->inv[2] < 0};
    int toVisit = 0, cur = 0;
    int stack[64];
    float ldir[3];
    orc_line_dir(ray->o, ray->d, ldir);
    uint32_t nv = 0, nt = 0;

    for (;;) {
        const orc_node *node = &b->nodes[cur];
        ++nv;
        if (orc_box_hit(node->bmin, node->bmax, ray->o, ldir)) {
            if (node->nprims > 0) {
                orc_hit hitrec;
                orc_hit_init(&hitrec);
                for (int i = 0; i < node->nprims; ++i) {
                    int32_t prim = b->prim_ids[node->offset + i];
                    const float *v = &b->verts[9 * (size_t)prim];
                    ++nt;
                    if (orc_tri_intersect(ray, v, v + 3, v + 6, prim, &hitrec)) {
                        anyhit = 1;
                        if (hitrec.t < isect->t) *isect = hitrec;
                    }
                }
                if (toVisit == 0) break;
                cur = stack[--toVisit];
            } else {
                if (dirIsNeg[node->axis]) {
                    stack[toVisit++] = cur + 1;
                    cur = node->offset;
                } else {
                    stack[toVisit++] = node->offset;
                    cur = cur + 1;
                }
            }
        } else {
            if (toVisit == 0) break;
            cur = stack[--toVisit];
        }
    }
    if (nodes_visited) *nodes_visited = nv;
    if (tris_tested) *tris_tested = nt;
    return anyhit;
}

void orc_bvh_intersect_batch(const orc_bvh *b, const float *org3, const float *dir3, size_t n,
                             int org_stride, uint8_t *hit, float *t, int32_t *prim, float *uvw3,
                             float *pos3, float *nrm3, int threads, uint64_t *nodes_visited,
                             uint64_t *tris_tested)
{
    uint64_t tot_nv = 0, tot_nt = 0;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    (void)threads;
#endif
#pragma omp parallel for schedule(dynamic, 1024) num_threads(threads) reduction(+ : tot_nv, tot_nt)
    for (long long i = 0; i < (long long)n; ++i) {
        orc_ray r;
        orc_hit h;
        uint32_t nv, nt;
        orc_ray_init(&r, org3 + (size_t)org_stride * (size_t)i, dir3 + 3 * (size_t)i);
        orc_hit_init(&h);
        int any = orc_bvh_intersect(b, &r, &h, &nv, &nt);
        tot_nv += nv;
        tot_nt += nt;
        if (hit) hit[i] = (uint8_t)any;
        if (t) t[i] = h.t;
        if (prim) prim[i] = any ? h.primID : -1;
        if (uvw3) {
            uvw3[3 * i + 0] = h.u;
            uvw3[3 * i + 1] = h.v;
            uvw3[3 * i + 2] = h.w;
        }
        if (pos3) memcpy(pos3 + 3 * i, h.pos, sizeof(float) * 3);
        if (nrm3) memcpy(nrm3 + 3 * i, h.nrm, sizeof(float) * 3);
    }
    if (nodes_visited) *nodes_visited = tot_nv;
    if (tris_tested) *tris_tested = tot_nt;
}
