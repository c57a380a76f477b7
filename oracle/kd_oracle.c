/*
 * kd_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Restates the vendored kd-tree of the reference (cpp/raycast/pspKdtree.c) for the one query
 * the hot path's set-up uses: kd_nearest over all model nodes (insert_rec :131-157,
 * kd_insert :159-171, kd_nearest_i :260-321, kd_nearest :323-372, hyperrect_dist_sq :634-648).
 * Same tree (sequential insertion, split dimension cycling, `<` goes left), same traversal
 * order, same strict `<` on double squared distances -> same node also on exact ties.
 *
 * PARITY: pinned against the real thing.  pspKdtree.c compiles from its own source
 * (oracle/Makefile target `ref` -> oracle/_ref/libpspkdtree.so, only where /root/reference
 * exists); tests/test_nearest_oracle.py compares both on the reference's grid and on random
 * clouds with duplicated points.
 */
#include "upsp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

struct orc_kdtree {
    size_t n;
    double *pos;      /* 3 per node, insertion order = node index */
    int32_t *left, *right;
    uint8_t *dir;
    double rmin[3], rmax[3];
};

orc_kdtree *orc_kd_build(const float *nodes3, size_t n)
{
    orc_kdtree *t = (orc_kdtree *)calloc(1, sizeof(*t));
    t->n = n;
    t->pos = (double *)malloc(sizeof(double) * 3 * (n ? n : 1));
    t->left = (int32_t *)malloc(sizeof(int32_t) * (n ? n : 1));
    t->right = (int32_t *)malloc(sizeof(int32_t) * (n ? n : 1));
    t->dir = (uint8_t *)malloc(n ? n : 1);
    for (size_t i = 0; i < n; ++i) {
        double *p = &t->pos[3 * i];
        for (int a = 0; a < 3; ++a) p[a] = (double)nodes3[3 * i + a];   /* kd_insert3(float->double) */
        t->left[i] = t->right[i] = -1;
        if (i == 0) {
            t->dir[0] = 0;
            for (int a = 0; a < 3; ++a) t->rmin[a] = t->rmax[a] = p[a];
            continue;
        }
        int32_t cur = 0;                                /* insert_rec, iteratively */
        for (;;) {
            const int d = t->dir[cur];
            int32_t *next = (p[d] < t->pos[3 * (size_t)cur + d]) ? &t->left[cur] : &t->right[cur];
            if (*next < 0) {
                *next = (int32_t)i;
                t->dir[i] = (uint8_t)((d + 1) % 3);
                break;
            }
            cur = *next;
        }
        for (int a = 0; a < 3; ++a) {                   /* hyperrect_extend */
            if (p[a] < t->rmin[a]) t->rmin[a] = p[a];
            if (p[a] > t->rmax[a]) t->rmax[a] = p[a];
        }
    }
    return t;
}

void orc_kd_free(orc_kdtree *t)
{
    if (!t) return;
    free(t->pos); free(t->left); free(t->right); free(t->dir); free(t);
}

static double sq(double v) { return v * v; }

static double rect_dist_sq(const double *mn, const double *mx, const double *pos)
{
    double r = 0;
    for (int i = 0; i < 3; ++i) {
        if (pos[i] < mn[i]) r += sq(mn[i] - pos[i]);
        else if (pos[i] > mx[i]) r += sq(mx[i] - pos[i]);
    }
    return r;
}

typedef struct { int32_t node; int8_t phase, side; double saved; } kd_frame;

int32_t orc_kd_nearest(const orc_kdtree *t, const double pos[3], double *dist2_out)
{
    if (!t || t->n == 0) return -1;
    double mn[3], mx[3];
    memcpy(mn, t->rmin, sizeof(mn));
    memcpy(mx, t->rmax, sizeof(mx));
    int32_t best = 0;                                   /* first guess: the root (:348-352) */
    double best_d = 0;
    for (int i = 0; i < 3; ++i) best_d += sq(t->pos[i] - pos[i]);
    size_t cap = 256, top = 0;
    kd_frame *st = (kd_frame *)malloc(sizeof(kd_frame) * cap);
    st[top++] = (kd_frame){0, 0, 0, 0.0};
    while (top) {
        kd_frame *f = &st[top - 1];
        const int32_t nd = f->node;
        const int dir = t->dir[nd];
        const double split = t->pos[3 * (size_t)nd + dir];
        if (f->phase == 0) {
            f->side = (pos[dir] - split <= 0) ? 0 : 1;  /* 0: nearer = left */
            f->phase = 1;
            const int32_t nearer = f->side == 0 ? t->left[nd] : t->right[nd];
            if (nearer >= 0) {
                double *coord = f->side == 0 ? &mx[dir] : &mn[dir];
                f->saved = *coord;
                *coord = split;
                if (top == cap) { cap *= 2; st = (kd_frame *)realloc(st, sizeof(kd_frame) * cap); f = &st[top - 1]; }
                st[top++] = (kd_frame){nearer, 0, 0, 0.0};
                f->phase = 2;                           /* restore on return */
            }
            continue;
        }
        if (f->phase == 2) {                            /* back from the nearer subtree */
            double *coord = f->side == 0 ? &mx[dir] : &mn[dir];
            *coord = f->saved;
            f->phase = 1;
        }
        if (f->phase == 1) {
            double d = 0;
            for (int i = 0; i < 3; ++i) d += sq(t->pos[3 * (size_t)nd + i] - pos[i]);
            if (d < best_d) { best = nd; best_d = d; }
            const int32_t farther = f->side == 0 ? t->right[nd] : t->left[nd];
            if (farther >= 0) {
                double *coord = f->side == 0 ? &mn[dir] : &mx[dir];
                f->saved = *coord;
                *coord = split;
                if (rect_dist_sq(mn, mx, pos) < best_d) {
                    f->phase = 3;
                    if (top == cap) { cap *= 2; st = (kd_frame *)realloc(st, sizeof(kd_frame) * cap); }
                    st[top++] = (kd_frame){farther, 0, 0, 0.0};
                    continue;
                }
                *coord = f->saved;
            }
            --top;
            continue;
        }
        /* phase 3: back from the farther subtree */
        {
            double *coord = f->side == 0 ? &mn[dir] : &mx[dir];
            *coord = f->saved;
            --top;
        }
    }
    free(st);
    if (dist2_out) *dist2_out = best_d;
    return best;
}

void orc_kd_nearest_batch(const orc_kdtree *t, const double *query3, size_t nq, int32_t *index,
                          double *dist2)
{
    for (size_t q = 0; q < nq; ++q) index[q] = orc_kd_nearest(t, &query3[3 * q], dist2 ? &dist2[q] : NULL);
}

/* upsp::interpolate, cpp/lib/interpolation.ipp:16-70 (+ nearest_k_neighbors, cpp/lib/models.ipp:503-571),
 * exhaustive: k nearest by (distance, index) ascending, inverse-distance weights, float sums.
 * PARITY UNPINNED: the reference's test for it is disabled (cpp/test/run_tests.cpp:15). */
void orc_interpolate_idw(const float *src3, const float *data, size_t nsrc, const float *q3, size_t nq,
                         int k, float p, float *out, int32_t *nbr)
{
    double *bd = (double *)malloc(sizeof(double) * (size_t)k);
    int32_t *bi = (int32_t *)malloc(sizeof(int32_t) * (size_t)k);
    for (size_t q = 0; q < nq; ++q) {
        int have = 0;
        for (size_t n = 0; n < nsrc; ++n) {
            const float dx = q3[3 * q] - src3[3 * n], dy = q3[3 * q + 1] - src3[3 * n + 1], dz = q3[3 * q + 2] - src3[3 * n + 2];
            const double d = sqrt((double)dx * dx + (double)dy * dy + (double)dz * dz);
            if (have < k || d < bd[have - 1]) {
                int j = have < k ? have : k - 1;
                while (j > 0 && bd[j - 1] > d) { bd[j] = bd[j - 1]; bi[j] = bi[j - 1]; --j; }
                bd[j] = d; bi[j] = (int32_t)n;
                if (have < k) ++have;
            }
        }
        float acc = 0.0f, total = 0.0f;
        for (int j = 0; j < have; ++j) {
            const float dist = (float)bd[j];
            if (dist == 0.0f) { total = 1.0f; acc = data[bi[j]]; break; }
            const float pw = p == 2.0f ? dist * dist : powf(dist, p);   /* = correctly rounded pow(dist, 2) */
            const float w = (float)(1.0 / (double)pw);
            acc += data[bi[j]] * w;
            total += w;
        }
        out[q] = acc / total;
        if (nbr) for (int j = 0; j < k; ++j) nbr[q * (size_t)k + j] = j < have ? bi[j] : -1;
    }
    free(bd); free(bi);
}
