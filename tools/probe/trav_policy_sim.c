/* Wave-scheduling study of the one-ray-per-lane traversal (tools/probe; CPU only, links oracle/libupsp_oracle.so).
 *
 * Every ray's sequence of steps on the wide records (N = one wide step: four grandchild boxes; T = one triangle test; the first
 * test of a leaf is marked) does not depend on how the 64 rays of a wave are interleaved -- only the number of wave ROUNDS does.
 * This program records the sequences for a ray file (scene soup + origin + directions, written by trav_policy_sim.py), packs 64
 * consecutive rays per wave as the dense lists of raycast.hip do, and counts rounds and (rounds x instructions per round) for
 * several interleaving policies.  The pruning of the device kernel is approximated by the slab entry distance (step counts
 * within a few per cent of the kernel's statistics counters).
 *
 *   gcc -O2 -o /tmp/trav_policy_sim tools/probe/trav_policy_sim.c -Ioracle -Loracle -lupsp_oracle -lm -Wl,-rpath,$PWD/oracle
 */
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "upsp_oracle.h"

typedef struct { unsigned char *s; int n, cap; } seq_t;
static void seq_push(seq_t *q, unsigned char c)
{
    if (q->n == q->cap) { q->cap = q->cap ? 2 * q->cap : 64; q->s = realloc(q->s, q->cap); }
    q->s[q->n++] = c;
}

static float slab_near(const orc_node *nd, const orc_ray *r)
{
    float tn = -FLT_MAX;
    for (int a = 0; a < 3; ++a) {
        float t0 = (nd->bmin[a] - r->o[a]) * r->inv[a], t1 = (nd->bmax[a] - r->o[a]) * r->inv[a];
        float lo = fminf(t0, t1);
        if (lo > tn) tn = lo;
    }
    return tn;
}

/* the wide walk of one ray; returns t of the closest hit */
static float walk(const orc_bvh *b, const orc_ray *ray, const float *ldir, seq_t *q)
{
    int neg[3] = {ray->inv[0] < 0, ray->inv[1] < 0, ray->inv[2] < 0};
    int stack[256], sp = 0;
    orc_hit best;
    orc_hit_init(&best);
    const orc_node *root = &b->nodes[0];
    if (!orc_box_hit(root->bmin, root->bmax, ray->o, ldir)) return best.t;
    int cur = 0;
    for (;;) {
        const orc_node *nd = &b->nodes[cur];
        if (nd->nprims > 0) {
            for (int i = 0; i < nd->nprims; ++i) {
                int prim = b->prim_ids[nd->offset + i];
                const float *v = &b->verts[9 * (size_t)prim];
                orc_hit h;
                orc_hit_init(&h);
                seq_push(q, i == 0 ? 'L' : 't');
                if (orc_tri_intersect(ray, v, v + 3, v + 6, prim, &h) && h.t < best.t) best = h;
            }
        } else {
            seq_push(q, 'N');
            /* slots in visiting order: near child's group first, near grandchild first inside a group */
            int kids[2] = {cur + 1, nd->offset};
            if (neg[nd->axis]) { kids[0] = nd->offset; kids[1] = cur + 1; }
            int slots[4], ns = 0;
            for (int k = 0; k < 2; ++k) {
                const orc_node *c = &b->nodes[kids[k]];
                if (c->nprims > 0) slots[ns++] = kids[k];
                else if (neg[c->axis]) { slots[ns++] = c->offset; slots[ns++] = kids[k] + 1; }
                else { slots[ns++] = kids[k] + 1; slots[ns++] = c->offset; }
            }
            int acc[4], na = 0;
            for (int k = 0; k < ns; ++k) {
                const orc_node *c = &b->nodes[slots[k]];
                if (!orc_box_hit(c->bmin, c->bmax, ray->o, ldir)) continue;
                if (slab_near(c, ray) > best.t) continue;
                acc[na++] = slots[k];
            }
            for (int k = na - 1; k >= 1; --k) stack[sp++] = acc[k];
            if (na) { cur = acc[0]; continue; }
        }
        if (!sp) break;
        cur = stack[--sp];
    }
    return best.t;
}

typedef struct { double rounds_n, rounds_t, lanes_n, lanes_t, instr; } cost_t;
static double CN = 165, CT = 120, CO = 12;     /* VALU per node round / triangle round / outer iteration (argv 3, 4) */

/* lane state over its sequence */
typedef struct { const unsigned char *s; int n, p; } lane_t;
static int at_node(const lane_t *l) { return l->p < l->n && l->s[l->p] == 'N'; }
static int at_leaf(const lane_t *l) { return l->p < l->n && l->s[l->p] != 'N'; }

/* P0: raycast.hip's trav_run -- up to cap node rounds, then every lane that holds a leaf tests the WHOLE leaf */
static void policy_current(lane_t *L, int nl, int cap, cost_t *c)
{
    for (;;) {
        int live = 0;
        for (int i = 0; i < nl; ++i) live += L[i].p < L[i].n;
        if (!live) break;
        c->instr += CO;
        for (int d = 0; d < cap; ++d) {
            int k = 0;
            for (int i = 0; i < nl; ++i) k += at_node(&L[i]);
            if (!k) break;
            for (int i = 0; i < nl; ++i) if (at_node(&L[i])) ++L[i].p;
            c->rounds_n += 1; c->lanes_n += k; c->instr += CN;
        }
        /* leaf phase: lanes at a leaf run that leaf to its end */
        int on[64], any = 0;
        for (int i = 0; i < nl; ++i) { on[i] = at_leaf(&L[i]); any |= on[i]; }
        int first = 1;
        while (any) {
            int k = 0;
            for (int i = 0; i < nl; ++i) if (on[i]) { ++L[i].p; ++k; }
            c->rounds_t += 1; c->lanes_t += k; c->instr += CT;
            any = 0;
            for (int i = 0; i < nl; ++i) { on[i] = on[i] && L[i].p < L[i].n && L[i].s[L[i].p] == 't'; any |= on[i]; }
            first = 0;
        }
        (void)first;
    }
}

/* P1: one step per round, the kind chosen per round: node round when at least thr_n lanes want one (or nobody wants a triangle),
   triangle round when at least thr_t want one (or nobody wants a node); both kinds may run in one round */
static void policy_mixed(lane_t *L, int nl, int thr_n, int thr_t, cost_t *c)
{
    for (;;) {
        int kn = 0, kt = 0;
        for (int i = 0; i < nl; ++i) { kn += at_node(&L[i]); kt += at_leaf(&L[i]); }
        if (!kn && !kt) break;
        c->instr += CO;
        int run_n = kn && (kn >= thr_n || !kt), run_t = kt && (kt >= thr_t || !kn);
        if (!run_n && !run_t) { if (kn * CT >= kt * CN) run_n = 1; else run_t = 1; }
        unsigned char was_leaf[64];
        for (int i = 0; i < nl; ++i) was_leaf[i] = (unsigned char)at_leaf(&L[i]);
        if (run_n) {
            for (int i = 0; i < nl; ++i) if (at_node(&L[i]) && !was_leaf[i]) ++L[i].p;
            c->rounds_n += 1; c->lanes_n += kn; c->instr += CN;
        }
        if (run_t) {
            for (int i = 0; i < nl; ++i) if (was_leaf[i]) ++L[i].p;
            c->rounds_t += 1; c->lanes_t += kt; c->instr += CT;
        }
    }
}

/* P2: the majority kind only (weighted by its cost) */
static void policy_majority(lane_t *L, int nl, cost_t *c)
{
    for (;;) {
        int kn = 0, kt = 0;
        for (int i = 0; i < nl; ++i) { kn += at_node(&L[i]); kt += at_leaf(&L[i]); }
        if (!kn && !kt) break;
        c->instr += CO;
        if (kn >= kt) {
            for (int i = 0; i < nl; ++i) if (at_node(&L[i])) ++L[i].p;
            c->rounds_n += 1; c->lanes_n += kn; c->instr += CN;
        } else {
            for (int i = 0; i < nl; ++i) if (at_leaf(&L[i])) ++L[i].p;
            c->rounds_t += 1; c->lanes_t += kt; c->instr += CT;
        }
    }
}

static void report(const char *name, const cost_t *c, long nwaves, long nrays)
{
    printf("%-34s rounds/wave N %6.1f T %6.1f  lanes/round N %5.1f T %5.1f  instr/wave %8.0f  instr/ray %7.1f\n", name,
           c->rounds_n / nwaves, c->rounds_t / nwaves, c->rounds_n ? c->lanes_n / c->rounds_n : 0.0,
           c->rounds_t ? c->lanes_t / c->rounds_t : 0.0, c->instr / nwaves, c->instr / nrays);
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s rays.bin [lanes per wave]\n", argv[0]); return 2; }
    const int nl = argc > 2 ? atoi(argv[2]) : 64;
    if (argc > 4) { CN = atof(argv[3]); CT = atof(argv[4]); }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    long long ntris, nrays;
    float org[3];
    if (fread(&ntris, 8, 1, f) != 1 || fread(&nrays, 8, 1, f) != 1 || fread(org, 4, 3, f) != 3) return 1;
    float *soup = malloc(sizeof(float) * 9 * (size_t)ntris), *dirs = malloc(sizeof(float) * 3 * (size_t)nrays);
    if (fread(soup, 4, 9 * (size_t)ntris, f) != 9 * (size_t)ntris || fread(dirs, 4, 3 * (size_t)nrays, f) != 3 * (size_t)nrays) return 1;
    fclose(f);
    orc_bvh *b = orc_bvh_create(soup, (size_t)ntris);
    fprintf(stderr, "tree: %d nodes, %lld triangles, %lld rays\n", b->nnodes, ntris, nrays);

    seq_t *Q = calloc((size_t)nrays, sizeof(seq_t));
    long kept = 0;
    double sn = 0, st = 0;
    int longest = 0;
    for (long long i = 0; i < nrays; ++i) {
        orc_ray r;
        float ldir[3];
        orc_ray_init(&r, org, dirs + 3 * i);
        orc_line_dir(org, dirs + 3 * i, ldir);
        seq_t q = {0, 0, 0};
        walk(b, &r, ldir, &q);
        if (!q.n) { free(q.s); continue; }        /* misses the root box: never on the dense list */
        for (int k = 0; k < q.n; ++k) { if (q.s[k] == 'N') sn += 1; else st += 1; }
        if (q.n > longest) longest = q.n;
        Q[kept++] = q;
    }
    printf("rays on the list %ld: wide steps %.2f + triangle tests %.2f per ray, longest %d steps\n", kept, sn / kept, st / kept, longest);
    /* argv[5]: ray -> wave assignment.  0 = list order (default); B > 0 = stable counting sort of the list into B bins by the ray's own
       step count (what a REPEATED build knows from the build before it: length-homogeneous waves), bin edges = equal-population
       quantiles; -1 = full sort by step count (the floor of any binning) */
    const int bins = argc > 5 ? atoi(argv[5]) : 0;
    if (bins != 0 && kept > 0) {
        int *hist = calloc((size_t)longest + 2, sizeof(int));
        for (long i = 0; i < kept; ++i) ++hist[Q[i].n];
        int *bin_of = calloc((size_t)longest + 2, sizeof(int));
        const int B = bins < 0 ? longest + 1 : bins;
        if (bins < 0) { for (int n = 0; n <= longest; ++n) bin_of[n] = n; }
        else {
            long acc = 0; int b = 0;
            for (int n = 0; n <= longest; ++n) {
                bin_of[n] = b;
                acc += hist[n];
                while (b + 1 < B && acc >= (long)((double)kept * (b + 1) / B)) ++b;
            }
            printf("bins by step count (upper edges):");
            for (int n = 0; n < longest; ++n) if (bin_of[n] != bin_of[n + 1]) printf(" %d", n);
            printf(" %d\n", longest);
        }
        long *start = calloc((size_t)B + 1, sizeof(long));
        for (long i = 0; i < kept; ++i) ++start[bin_of[Q[i].n] + 1];
        for (int b = 0; b < B; ++b) start[b + 1] += start[b];
        seq_t *S = malloc(sizeof(seq_t) * (size_t)kept);
        for (long i = 0; i < kept; ++i) S[start[bin_of[Q[i].n]]++] = Q[i];
        memcpy(Q, S, sizeof(seq_t) * (size_t)kept);
        free(S); free(start); free(bin_of); free(hist);
    }
    const long nwaves = (kept + nl - 1) / nl;
    lane_t L[64];
#define RUN(name, call)                                                                     \
    do {                                                                                    \
        cost_t c = {0, 0, 0, 0, 0};                                                         \
        for (long w = 0; w < nwaves; ++w) {                                                 \
            int m = 0;                                                                      \
            for (long i = w * nl; i < kept && i < (w + 1) * nl; ++i, ++m) { L[m].s = Q[i].s; L[m].n = Q[i].n; L[m].p = 0; } \
            call;                                                                           \
        }                                                                                   \
        report(name, &c, nwaves, kept);                                                     \
    } while (0)
    {
        /* the floor: every wave as long as its longest ray, each round at the cheaper kind */
        double fl = 0;
        for (long w = 0; w < nwaves; ++w) {
            int mx = 0;
            for (long i = w * nl; i < kept && i < (w + 1) * nl; ++i) if (Q[i].n > mx) mx = Q[i].n;
            fl += mx;
        }
        printf("longest ray of a wave: %.1f steps on average (x %d lanes = %.1f lane slots per ray step)\n", fl / nwaves, nl,
               fl * nl / (sn + st));
    }
    RUN("current, cap 1", policy_current(L, m, 1, &c));
    RUN("current, cap 2", policy_current(L, m, 2, &c));
    RUN("current, cap 4", policy_current(L, m, 4, &c));
    RUN("current, cap 8", policy_current(L, m, 8, &c));
    RUN("both kinds every round", policy_mixed(L, m, 1, 1, &c));
    RUN("mixed, thresholds 8 / 8", policy_mixed(L, m, 8, 8, &c));
    RUN("mixed, thresholds 16 / 16", policy_mixed(L, m, 16, 16, &c));
    RUN("mixed, thresholds 24 / 24", policy_mixed(L, m, 24, 24, &c));
    RUN("mixed, thresholds 16 / 8", policy_mixed(L, m, 16, 8, &c));
    RUN("mixed, thresholds 8 / 16", policy_mixed(L, m, 8, 16, &c));
    RUN("mixed, thresholds 32 / 32", policy_mixed(L, m, 32, 32, &c));
    RUN("majority kind only", policy_majority(L, m, &c));
    return 0;
}
