// MI355X (gfx950) BVH ray caster: closest-hit / any-hit batch queries and the
// projection-build kernel (create_projection_mat of psp_process).
//
// Execution model
//   * one ray per lane, 64-lane wavefronts, 256-thread workgroups;
//   * persistent waves: a wave pulls blocks of work items from a global queue and
//     re-fills lanes whose ray has finished as soon as fewer than kRefill lanes
//     are still traversing (ballot + prefix popcount assigns queue slots to idle
//     lanes), so divergence in ray length does not leave the SIMD half empty;
//   * the traversal stack lives in LDS, laid out [entry][thread] so that every
//     push/pop of a wave is one conflict-free ds_read/ds_write_b32;
//   * interior nodes are 64-byte records with BOTH child boxes (4 x dwordx4 per
//     lane), triangles are 48-byte records (3 x dwordx4) in leaf order.
//
// Numerical contract (see DESIGN.md "parity"): compiled with -ffp-contract=off.
//   * ray set-up, the watertight triangle test with its double fallback and the
//     hit record follow cpp/raycast/pspRT.cpp:45-69 and :109-193 operation by
//     operation (IEEE + - * / only);
//   * a box is entered iff Imath::intersects(Box3f, Line3f(o,o+d)) accepts it
//     (call site pspRT.cpp:382-385).  The decision is made with reciprocal
//     multiplies and an error bound; only when the margin is inside the bound is
//     the exact division form evaluated, so the accepted set is identical;
//   * on top of that a subtree is skipped when its near plane along the ray's
//     major axis lies beyond the current closest hit.  Every t the triangle test
//     can produce for a triangle inside the box is >= that plane distance (up to
//     7 ulp, a 4e-6 guard is used), so skipping never changes t, primID or the
//     tie-break (strict `<`, near-child-first order, pspRT.cpp:395,410-419).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "ktimer.h"
#include "upsp_internal.h"

namespace upsp {

thread_local std::string g_error;
void set_error(const std::string &msg) { g_error = msg; }
int fail(int status, const std::string &msg)
{
    g_error = msg;
    return status;
}

namespace {

constexpr int kBlock = 256;       // threads per workgroup (4 waves)
constexpr unsigned kPackWavesPerSimd = 3;   // dense ray lists are spread over this many waves per SIMD, all resident at once
constexpr int kChunkMax = 1024;    // largest block of work items taken from the global queue at once
constexpr int kRefillDefault = 40;  // leave the traversal loop to re-fill below this many live lanes
constexpr int kDone = INT32_MIN;  // "no current node"

struct Ray {
    float ox, oy, oz;     // rt::Ray::o
    float dx, dy, dz;     // rt::Ray::d
    float Sx, Sy, Sz;     // watertight shear (pspRT.cpp:63-65)
    int kx, ky, kz;
    unsigned neg;         // bit a = ray.i[a] < 0 (dirIsNeg, pspRT.cpp:369)
    // Box test (Imath::intersects(Box3f, Line3f(o,o+d), ip)) in per-axis MIRRORED form:
    // an axis whose line direction l is negative is reflected (o -> -o, [lo,hi] ->
    // [-hi,-lo], l -> -l), which maps the library's `dir < 0` branch onto its `dir > 0`
    // branch with bit-identical differences, quotients and comparisons.
    float mox, moy, moz;  // mirrored origin
    float mlx, mly, mlz;  // |l| (0 for an axis the line does not move along)
    float six, siy, siz;  // ~1/l, SIGNED (filters only; the mirrored forms use |.|)
    float cx, cy, cz;     // o * (1/l): the slab filter evaluates t = fma(plane, 1/l, -c) (ray_classify)
    float slabE;          // absolute error bound of those t against the library's quotients, for every box inside the scene
    float limx, limy, limz;  // FLT_MAX * |l| (the library's overflow guard)
    unsigned flip;        // bit a: axis mirrored (l < 0)
    unsigned zero;        // bit a: l == 0
    unsigned big;         // bit a: |l| > 1
    float absSz;          // |Sz|
    bool prune_ok;        // major axis of d and of l agree in sign -> depth pruning valid
    bool exact_only;      // some |l| is tiny / not finite: skip the reciprocal filter
    bool simple;          // no special case of the box test can occur inside this scene (ray_classify)
};

__device__ __forceinline__ float pick(float x, float y, float z, int k)
{
    return k == 0 ? x : (k == 1 ? y : z);
}

// Imath::Vec3<float>::length(): sqrt(dot), scaled form below 2*FLT_MIN
__device__ __forceinline__ float imath_length(float x, float y, float z)
{
    float l2 = x * x + y * y + z * z;
    if (l2 < 2.0f * FLT_MIN) {
        float ax = fabsf(x), ay = fabsf(y), az = fabsf(z);
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

// rt::Ray::Ray(o, d), pspRT.cpp:45-69, plus the per-ray constants of the box test.
__device__ __forceinline__ void ray_setup(Ray &r, float ox, float oy, float oz, float dx,
                                          float dy, float dz)
{
    r.ox = ox; r.oy = oy; r.oz = oz;
    r.dx = dx; r.dy = dy; r.dz = dz;
    float ax = fabsf(dx), ay = fabsf(dy), az = fabsf(dz);
    int kz = (ax > ay) ? (ax > az ? 0 : 2) : (ay > az ? 1 : 2);  // MAX_DIM, pspRT.cpp:41
    int kx = kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    float dkz = pick(dx, dy, dz, kz);
    if (dkz < 0.0f) { int t = kx; kx = ky; ky = t; }
    r.kx = kx; r.ky = ky; r.kz = kz;
    r.Sx = pick(dx, dy, dz, kx) / dkz;
    r.Sy = pick(dx, dy, dz, ky) / dkz;
    r.Sz = 1.0f / dkz;
    r.absSz = fabsf(r.Sz);
    r.neg = ((1.0f / dx) < 0.0f ? 1u : 0u) | ((1.0f / dy) < 0.0f ? 2u : 0u) |
            ((1.0f / dz) < 0.0f ? 4u : 0u);
    // Imath::Line3f(o, o + d): dir = ((o+d) - o).normalize()
    float lx = (ox + dx) - ox, ly = (oy + dy) - oy, lz = (oz + dz) - oz;
    float len = imath_length(lx, ly, lz);
    if (len != 0.0f) { lx /= len; ly /= len; lz /= len; }
    r.flip = (lx < 0.0f ? 1u : 0u) | (ly < 0.0f ? 2u : 0u) | (lz < 0.0f ? 4u : 0u);
    r.zero = (lx == 0.0f ? 1u : 0u) | (ly == 0.0f ? 2u : 0u) | (lz == 0.0f ? 4u : 0u);
    r.mox = lx < 0.0f ? -ox : ox;
    r.moy = ly < 0.0f ? -oy : oy;
    r.moz = lz < 0.0f ? -oz : oz;
    r.mlx = fabsf(lx); r.mly = fabsf(ly); r.mlz = fabsf(lz);
    r.big = (r.mlx > 1.0f ? 1u : 0u) | (r.mly > 1.0f ? 2u : 0u) | (r.mlz > 1.0f ? 4u : 0u);
    r.six = copysignf(__builtin_amdgcn_rcpf(r.mlx), lx);
    r.siy = copysignf(__builtin_amdgcn_rcpf(r.mly), ly);
    r.siz = copysignf(__builtin_amdgcn_rcpf(r.mlz), lz);
    r.cx = r.cy = r.cz = 0.0f;
    r.slabE = __builtin_inff();
    r.limx = FLT_MAX * r.mlx; r.limy = FLT_MAX * r.mly; r.limz = FLT_MAX * r.mlz;
    const float tiny = 1e-18f;
    r.exact_only = (lx != 0.0f && r.mlx < tiny) || (ly != 0.0f && r.mly < tiny) ||
                   (lz != 0.0f && r.mlz < tiny) || !(len == len) || len == 0.0f;
    const float lkz = pick(lx, ly, lz, kz);
    r.prune_ok = lkz != 0.0f && ((lkz < 0.0f) == (dkz < 0.0f));
    r.simple = false;
}

// One axis of the library test in mirrored form (line direction >= 0).  All
// predicates are computed unconditionally (no early return): the kernel's hot loop
// stays branch-free.  d?: differences to the far / near plane.
struct AxisEval {
    float dB, dF;
    bool out, inside, okB, okF, front;
};

__device__ __forceinline__ AxisEval axis_eval(float lo, float hi, float mo, float lim, bool flip,
                                              bool zero, bool big)
{
    const float mlo = flip ? -hi : lo, mhi = flip ? -lo : hi;
    AxisEval e;
    e.dB = mhi - mo;            // >= 0 unless the origin is beyond the far plane
    e.dF = mlo - mo;            // >= 0 iff the origin is at / before the near plane
    // bitwise (not short-circuit) logic: keeps the evaluation free of branches
    const bool beyond = e.dB < 0.0f, before = e.dF > 0.0f;
    e.out = beyond | (zero & before);
    e.inside = !(beyond | before);
    e.okB = !zero & (big | (e.dB < lim));
    e.okF = big | (e.dF < lim);
    e.front = !zero & (e.dF >= 0.0f);
    return e;
}

struct BoxEval {
    bool accept;     // decided: entered
    bool undecided;  // filter margin too small -> needs the exact divisions
};

// Filtered decision with reciprocal multiplies (relative error <= 2 ulp per quotient).
__device__ __forceinline__ BoxEval box_filter(const Ray &r, float lox, float loy, float loz,
                                              float hix, float hiy, float hiz, float &dFk)
{
    const AxisEval ex = axis_eval(lox, hix, r.mox, r.limx, r.flip & 1u, r.zero & 1u, r.big & 1u);
    const AxisEval ey = axis_eval(loy, hiy, r.moy, r.limy, r.flip & 2u, r.zero & 2u, r.big & 2u);
    const AxisEval ez = axis_eval(loz, hiz, r.moz, r.limz, r.flip & 4u, r.zero & 4u, r.big & 4u);
    dFk = pick(ex.dF, ey.dF, ez.dF, r.kz);
    const bool inside = ex.inside & ey.inside & ez.inside;
    const bool out = ex.out | ey.out | ez.out;
    float tBack = FLT_MAX, tFront = -1.0f;
    const float ilx = fabsf(r.six), ily = fabsf(r.siy), ilz = fabsf(r.siz);
    const float bx = ex.dB * ilx, by = ey.dB * ily, bz = ez.dB * ilz;
    tBack = (ex.okB & (tBack > bx)) ? bx : tBack;
    tBack = (ey.okB & (tBack > by)) ? by : tBack;
    tBack = (ez.okB & (tBack > bz)) ? bz : tBack;
    const float fx = ex.okF ? ex.dF * ilx : FLT_MAX;
    const float fy = ey.okF ? ey.dF * ily : FLT_MAX;
    const float fz = ez.okF ? ez.dF * ilz : FLT_MAX;
    tFront = (ex.front & (tFront < fx)) ? fx : tFront;
    tFront = (ey.front & (tFront < fy)) ? fy : tFront;
    tFront = (ez.front & (tFront < fz)) ? fz : tFront;
    const float e = 6e-7f;  // > 2 * (rcp 1 ulp + mul 0.5 ulp)
    const bool sure_acc = tFront + fabsf(tFront) * e <= tBack - tBack * e;
    const bool sure_rej = tFront - fabsf(tFront) * e > tBack + tBack * e;
    BoxEval b;
    b.accept = inside | (!out & sure_acc);
    b.undecided = !inside & !out & ((!sure_acc & !sure_rej) | r.exact_only);
    return b;
}

// The same filtered decision for rays of the "simple" class (ray_classify): the line moves
// along all three axes and the library's overflow guards (d < FLT_MAX * |l|) hold for every
// box inside the scene bounds, so okB = okF = true, front = (dF >= 0), out = (some dB < 0),
// and the sequential min / max updates collapse into min3 / max3.  Same values, fewer VALU ops.
__device__ __forceinline__ BoxEval box_filter_simple(const Ray &r, float lox, float loy, float loz,
                                                     float hix, float hiy, float hiz, float &dFk)
{
    const bool fx_ = r.flip & 1u, fy_ = r.flip & 2u, fz_ = r.flip & 4u;
    const float dBx = (fx_ ? -lox : hix) - r.mox, dFx = (fx_ ? -hix : lox) - r.mox;
    const float dBy = (fy_ ? -loy : hiy) - r.moy, dFy = (fy_ ? -hiy : loy) - r.moy;
    const float dBz = (fz_ ? -loz : hiz) - r.moz, dFz = (fz_ ? -hiz : loz) - r.moz;
    dFk = pick(dFx, dFy, dFz, r.kz);
    const bool out = fminf(fminf(dBx, dBy), dBz) < 0.0f;
    const bool inside = !out & !(fmaxf(fmaxf(dFx, dFy), dFz) > 0.0f);
    const float ilx = fabsf(r.six), ily = fabsf(r.siy), ilz = fabsf(r.siz);
    const float tBack = fminf(FLT_MAX, fminf(fminf(dBx * ilx, dBy * ily), dBz * ilz));
    const float fx = dFx >= 0.0f ? dFx * ilx : -1.0f;
    const float fy = dFy >= 0.0f ? dFy * ily : -1.0f;
    const float fz = dFz >= 0.0f ? dFz * ilz : -1.0f;
    const float tFront = fmaxf(fmaxf(fx, fy), fmaxf(fz, -1.0f));
    const float e = 6e-7f;
    const bool sure_acc = tFront + fabsf(tFront) * e <= tBack - tBack * e;
    const bool sure_rej = tFront - fabsf(tFront) * e > tBack + tBack * e;
    BoxEval b;
    b.accept = inside | (!out & sure_acc);
    b.undecided = !inside & !out & !sure_acc & !sure_rej;
    return b;
}

// Round 6: the first filter of the "simple" class -- a plain slab test, a third of the instructions of box_filter_simple.
// For a simple ray (all three line components non-zero, every guard of the library's test satisfied for boxes inside the scene) the
// library's verdict is  min_i dB_i >= 0  and  max_i qF_i <= min_i qB_i  with  q = fl(fl(plane - o) / |l|)  per axis (an axis whose
// near quotient is negative takes no part in the library's maximum, but then it is below every far quotient of a box the origin is
// not beyond, so including it changes nothing; "origin inside" is the case where all of them are negative).  Here every quotient is
// replaced by  t = fma(plane, 1/l, -o/l)  with the SIGNED reciprocal (near / far = min / max of the two planes' t, no mirroring, no
// selects): |t - q| <= 6.1 u (|plane| + |o|) / |l|  (u = 2^-24: the rounded reciprocal 2 u, the rounded product o/l one u on the
// o/l term, the fused multiply-add one u, the library's two roundings 2 u), which for every box inside the scene bounds is below
// the per-ray constant slabE = 6e-7 x max_i (max |bound_i| + |o_i|) / |l_i| (ray_classify).  A box is decided here when the two
// sides are more than 2 slabE apart and the far side is more than slabE from zero; everything else -- grazing contact, the origin
// on a face -- goes to box_filter_simple and, inside its own margin, to the library's divisions.  Same verdicts, bit for bit.
// dFk: the mirrored near-plane difference on the ray's major axis exactly as box_filter_simple forms it (the pruning test uses it).
__device__ __forceinline__ BoxEval box_filter_slab(const Ray &r, float lox, float loy, float loz, float hix, float hiy, float hiz,
                                                   float &dFk)
{
    const float t0x = __builtin_fmaf(lox, r.six, -r.cx), t1x = __builtin_fmaf(hix, r.six, -r.cx);
    const float t0y = __builtin_fmaf(loy, r.siy, -r.cy), t1y = __builtin_fmaf(hiy, r.siy, -r.cy);
    const float t0z = __builtin_fmaf(loz, r.siz, -r.cz), t1z = __builtin_fmaf(hiz, r.siz, -r.cz);
    const float tn = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z));
    const float tf = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z));
    const float lo_k = pick(lox, loy, loz, r.kz), hi_k = pick(hix, hiy, hiz, r.kz), o_k = pick(r.ox, r.oy, r.oz, r.kz);
    dFk = ((r.flip >> r.kz) & 1u) ? o_k - hi_k : lo_k - o_k;      // = (flip ? -hi : lo) - (flip ? -o : o), the same float
    const float gap = tn - tf, e2 = 2.0f * r.slabE;
    BoxEval b;
    b.accept = (gap <= -e2) & (tf >= r.slabE);
    b.undecided = !b.accept & !((gap > e2) | (tf < -r.slabE));   // (NaN anywhere: undecided)
    return b;
}

// The library's arithmetic (true divisions); only evaluated for undecided boxes.
__device__ __forceinline__ bool box_exact(const Ray &r, float lox, float loy, float loz, float hix,
                                       float hiy, float hiz)
{
    const AxisEval ex = axis_eval(lox, hix, r.mox, r.limx, r.flip & 1u, r.zero & 1u, r.big & 1u);
    const AxisEval ey = axis_eval(loy, hiy, r.moy, r.limy, r.flip & 2u, r.zero & 2u, r.big & 2u);
    const AxisEval ez = axis_eval(loz, hiz, r.moz, r.limz, r.flip & 4u, r.zero & 4u, r.big & 4u);
    if (ex.inside && ey.inside && ez.inside) return true;
    if (ex.out || ey.out || ez.out) return false;
    float tBack = FLT_MAX, tFront = -1.0f;
    if (ex.okB) { const float t = ex.dB / r.mlx; if (tBack > t) tBack = t; }
    if (ey.okB) { const float t = ey.dB / r.mly; if (tBack > t) tBack = t; }
    if (ez.okB) { const float t = ez.dB / r.mlz; if (tBack > t) tBack = t; }
    if (ex.front) { const float t = ex.okF ? ex.dF / r.mlx : FLT_MAX; if (tFront < t) tFront = t; }
    if (ey.front) { const float t = ey.okF ? ey.dF / r.mly : FLT_MAX; if (tFront < t) tFront = t; }
    if (ez.front) { const float t = ez.okF ? ez.dF / r.mlz : FLT_MAX; if (tFront < t) tFront = t; }
    return tFront <= tBack;
}

// Imath::intersects(box, Line3f(o,o+d), ip): same accept/reject as the library.
// (Boxes of the tree are never empty, so isEmpty() is not re-tested per visit.)
__device__ __forceinline__ bool box_hit(const Ray &r, float lox, float loy, float loz, float hix,
                                        float hiy, float hiz)
{
    float dFk;
    const BoxEval b = box_filter(r, lox, loy, loz, hix, hiy, hiz, dFk);
    bool acc = b.accept;
    if (b.undecided) acc = box_exact(r, lox, loy, loz, hix, hiy, hiz);
    return acc;
}

struct TriHit {
    float t, u, v, w;
};

// rt::Triangle::intersect, pspRT.cpp:109-173 (hit distance and barycentrics).
__device__ __forceinline__ bool tri_test(const Ray &r, float ax, float ay, float az, float bx,
                                         float by, float bz, float cx, float cy, float cz,
                                         TriHit &h)
{
    const float Ax_ = ax - r.ox, Ay_ = ay - r.oy, Az_ = az - r.oz;
    const float Bx_ = bx - r.ox, By_ = by - r.oy, Bz_ = bz - r.oz;
    const float Cx_ = cx - r.ox, Cy_ = cy - r.oy, Cz_ = cz - r.oz;
    const float Akx = pick(Ax_, Ay_, Az_, r.kx), Aky = pick(Ax_, Ay_, Az_, r.ky),
                Akz = pick(Ax_, Ay_, Az_, r.kz);
    const float Bkx = pick(Bx_, By_, Bz_, r.kx), Bky = pick(Bx_, By_, Bz_, r.ky),
                Bkz = pick(Bx_, By_, Bz_, r.kz);
    const float Ckx = pick(Cx_, Cy_, Cz_, r.kx), Cky = pick(Cx_, Cy_, Cz_, r.ky),
                Ckz = pick(Cx_, Cy_, Cz_, r.kz);
    const float Ax = Akx - r.Sx * Akz;
    const float Ay = Aky - r.Sy * Akz;
    const float Bx = Bkx - r.Sx * Bkz;
    const float By = Bky - r.Sy * Bkz;
    const float Cx = Ckx - r.Sx * Ckz;
    const float Cy = Cky - r.Sy * Ckz;

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
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return false;
    const float det = U + V + W;
    if (det == 0.0f) return false;
    const float Az = r.Sz * Akz;
    const float Bz = r.Sz * Bkz;
    const float Cz = r.Sz * Ckz;
    const float T = U * Az + V * Bz + W * Cz;
    float xorf_T = fabsf(T);
    if (signbit(T) != signbit(det)) xorf_T = -xorf_T;
    const float abs_det = fabsf(det);
    const float inf = __builtin_inff();
    if (xorf_T < 0.0f * abs_det || inf * abs_det < xorf_T) return false;
    const float rcpDet = 1.0f / det;
    h.u = U * rcpDet;
    h.v = V * rcpDet;
    h.w = W * rcpDet;
    h.t = T * rcpDet;
    return true;
}

struct Scene {
    const float4 *nodes;  // 4 per interior node (binary records: cooperative walk, witness chains)
    const float4 *wide;   // 8 per wide node (GpuWide: the one-ray-per-lane traversal)
    const float4 *tris;   // 3 per triangle slot
    int root_ref;         // the traversal's root: a wide node (>= 0) or a leaf
    int root_ref2;        // the binary tree's root
    int refill;           // re-fill threshold (live lanes)
    int desc_cap;         // interior-node steps before the lanes that already hold a leaf get to test it (0 = no cap)
    int xcd;              // 1: XCD-aware static assignment of the work items (queue_init)
    unsigned heavy_steps; // a ray that needs more node visits + triangle tests than this is handed to heavy_kernel (0 = never)
    unsigned *heavy_items;   // list of the work items handed over (kHeavyCap entries), count in work[kWorkHeavyCount]
    unsigned heavy_stack;    // entries of heavy_kernel's stack that may be used (<= kHeavyStack; tests shrink it)
    unsigned char *heavy_scratch;   // the cooperative walk's stacks, kHeavyScratchBytes per workgroup (global memory, see heavy_kernel)
    unsigned pack_waves;     // projection passes: waves a ray list is spread over when the grid can take it at once (0: queue)
    unsigned chunk;       // work items per queue grab (multiple of 64)
    float rlo[3], rhi[3];
    const unsigned *adj_off, *adj_slot;   // node -> adjacent triangle slots (may be null)
    const uint2 *slot_path;               // triangle slot -> (offset, length) of its box chain (may be null)
    const unsigned *path_ref;             // chain entries: (interior node << 1) | side
    int *witness;                         // per node: slot hit by the primary ray (retry nodes)
    unsigned *hist;                       // debug (UPSP_DEBUG_HIST with statistics on): [0..47] steps per residual ray, [48..55] witness verdicts
    // No walk may outlive the tree: a ray visits every interior node and every leaf at most once, so a traversal call
    // (or a cooperative walk) that needs more rounds than `round_cap` = 2 x nodes + slack is running on a broken tree
    // (cyclic child references, overwritten records).  It sets a bit of *err and ends; the entry points return
    // UPSP_ERR_INTERNAL at their next synchronisation (the reference DIEs on a broken BVH, pspRT.cpp:362-365 -- an
    // error, never a wedged device).
    unsigned round_cap;
    unsigned *err;                        // bit 0 one-lane traversal, bit 1 cooperative walk, bit 2 its one-thread fallback
    // primary pass of a repeated projection build: rays listed in kRayBins bins by their previous step count (see RayBins)
    int slab;                             // 1: box_filter_slab in front of the mirrored filter (UPSP_SLAB_FILTER=0: off)
    float slab_scale;                     // >= 1: widens the slab filter's error bound (tests)
    unsigned short *steps_out;            // per node: steps of its primary ray (may be null)
    unsigned nbins;                       // 0: one dense list in node order
    unsigned bin_stride;                  // list of bin b = todo_rays + b * bin_stride
};

struct Trav {
    int cur;        // current ref, kDone when finished
    int sp;         // stack entries
    float best_t;   // closest t so far (FLT_MAX = none)
    float limit;    // best_t with the pruning guard applied
    int best_slot;  // triangle slot of the closest hit
    bool any;       // rt::BVH::intersect return value
    float own_min;  // visibility rays: stop as soon as the closest hit is nearer than this
    unsigned n_nodes, n_tris;
    unsigned ray_nodes, ray_tris;  // STATS: steps of the current ray
    unsigned w_node_rounds, w_tri_rounds;   // STATS: rounds of the WAVE (counted by its first executing lane)
    unsigned n_boxes, n_band;               // STATS: boxes the slab filter saw / left undecided
    unsigned steps;                // node visits + triangle tests of the current ray
};

__device__ __forceinline__ void trav_begin(Trav &s, const Ray &r, const Scene &sc)
{
    s.sp = 0;
    s.ray_nodes = s.ray_tris = 0;
    s.steps = 0;
    s.best_t = FLT_MAX;
    s.limit = __builtin_inff();
    s.best_slot = -1;
    s.any = false;
    s.own_min = -__builtin_inff();
    s.cur = box_hit(r, sc.rlo[0], sc.rlo[1], sc.rlo[2], sc.rhi[0], sc.rhi[1], sc.rhi[2])
                ? sc.root_ref
                : kDone;
}

// Marks the ray "simple" when no special case of the library's box test can occur for any box
// inside the scene bounds: all three line components non-zero, reciprocal filter usable, and
// every overflow guard `d < FLT_MAX * |l|` satisfied because |d| <= D = the largest distance
// along an axis between the origin and the scene bounds.
__device__ __forceinline__ void ray_classify(Ray &r, const Scene &sc)
{
    const float D = fmaxf(fmaxf(fmaxf(fabsf(sc.rlo[0] - r.ox), fabsf(sc.rhi[0] - r.ox)),
                                fmaxf(fabsf(sc.rlo[1] - r.oy), fabsf(sc.rhi[1] - r.oy))),
                          fmaxf(fabsf(sc.rlo[2] - r.oz), fabsf(sc.rhi[2] - r.oz)));
    const float lim = fminf(fminf(r.limx, r.limy), r.limz);
    r.simple = (r.zero == 0u) & !r.exact_only & (D < lim) & (D == D) & (D < 1e30f);
    // constants of the slab filter (box_filter_slab); a ray that is not simple never reaches it
    r.cx = r.ox * r.six;
    r.cy = r.oy * r.siy;
    r.cz = r.oz * r.siz;
    const float ex = (fmaxf(fabsf(sc.rlo[0]), fabsf(sc.rhi[0])) + fabsf(r.ox)) * fabsf(r.six);
    const float ey = (fmaxf(fabsf(sc.rlo[1]), fabsf(sc.rhi[1])) + fabsf(r.oy)) * fabsf(r.siy);
    const float ez = (fmaxf(fabsf(sc.rlo[2]), fabsf(sc.rhi[2])) + fabsf(r.oz)) * fabsf(r.siz);
    r.slabE = 6e-7f * sc.slab_scale * fmaxf(fmaxf(ex, ey), ez);
}

// (the LDS stack is laid out [entry][thread]: STRIDE = threads per workgroup)
template <int STRIDE = kBlock>
__device__ __forceinline__ void trav_pop(Trav &s, const int *stack)
{
    if (s.sp == 0) {
        s.cur = kDone;
    } else {
        --s.sp;
        s.cur = stack[s.sp * STRIDE];
    }
}

// One step of the traversal on the wide records (GpuWide, upsp_internal.h): one 128-byte fetch decides TWO levels of the
// reference's descent.  The record lists, for each child of a binary node the reference would be standing on, the child
// itself (a leaf) or the child's two children, so the boxes tested here are boxes the reference tests too -- minus the
// interior child's own box, which it enters before its children's: that test is implied (a box inside another is hit only
// by rays that hit the outer one; Imath's slab test is monotone in the box planes: both quotients of an axis move the
// right way when the box grows, the guards only ever turn a quotient into +-infinity on the growing side, and the
// "origin inside" / "origin beyond the far plane" shortcuts are monotone too).  Every slot is entered iff the library
// accepts its box (filter, exact divisions when undecided) and the depth-first near-first order is the reference's
// applied twice: first the group of the near CHILD (sign of the ray along the node's axis, pspRT.cpp:410-419), inside a
// group the near grandchild (the child's axis).  So the leaves reached and their order -- hence t, primID and every tie --
// are the binary walk's.  The accepted slots are pushed in reverse visiting order and the first one popped back.
template <bool ANYHIT, bool STATS, int STRIDE = kBlock>
__device__ __forceinline__ void wide_step(Trav &s, const Ray &r, const Scene &sc, int *stack)
{
    const float4 *np = sc.wide + 8 * (size_t)s.cur;
    const float4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3], q4 = np[4], q5 = np[5], q6 = np[6], q7 = np[7];
    ++s.steps;
    if (STATS) { ++s.n_nodes; ++s.ray_nodes; }
    const int ref0 = __float_as_int(q6.x), ref1 = __float_as_int(q6.y), ref2 = __float_as_int(q6.z), ref3 = __float_as_int(q6.w);
    const unsigned meta = __float_as_uint(q7.x);
    float dF0, dF1, dF2, dF3;
    BoxEval b0, b1, b2, b3;
    if (__ballot(!r.simple) == 0ull) {   // wave-uniform: every lane holds a simple-class ray
        if (sc.slab) {
            // the slab filter decides all but the grazing boxes; those go through the mirrored filter (and, inside ITS margin,
            // through the library's divisions below)
            b0 = box_filter_slab(r, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, dF0);
            b1 = box_filter_slab(r, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, dF1);
            b2 = box_filter_slab(r, q3.x, q3.y, q3.z, q3.w, q4.x, q4.y, dF2);
            b3 = box_filter_slab(r, q4.z, q4.w, q5.x, q5.y, q5.z, q5.w, dF3);
            if (STATS) {
                s.n_boxes += 4u;
                s.n_band += (b0.undecided ? 1u : 0u) + (b1.undecided ? 1u : 0u) + (b2.undecided ? 1u : 0u) + (b3.undecided ? 1u : 0u);
            }
            if (__ballot(b0.undecided | b1.undecided | b2.undecided | b3.undecided) != 0ull) {
                float d;
                if (b0.undecided) b0 = box_filter_simple(r, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, d);
                if (b1.undecided) b1 = box_filter_simple(r, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, d);
                if (b2.undecided) b2 = box_filter_simple(r, q3.x, q3.y, q3.z, q3.w, q4.x, q4.y, d);
                if (b3.undecided) b3 = box_filter_simple(r, q4.z, q4.w, q5.x, q5.y, q5.z, q5.w, d);
            }
        } else {
            b0 = box_filter_simple(r, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, dF0);
            b1 = box_filter_simple(r, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, dF1);
            b2 = box_filter_simple(r, q3.x, q3.y, q3.z, q3.w, q4.x, q4.y, dF2);
            b3 = box_filter_simple(r, q4.z, q4.w, q5.x, q5.y, q5.z, q5.w, dF3);
        }
    } else {
        b0 = box_filter(r, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, dF0);
        b1 = box_filter(r, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, dF1);
        b2 = box_filter(r, q3.x, q3.y, q3.z, q3.w, q4.x, q4.y, dF2);
        b3 = box_filter(r, q4.z, q4.w, q5.x, q5.y, q5.z, q5.w, dF3);
    }
    bool h0 = b0.accept, h1 = b1.accept, h2 = b2.accept, h3 = b3.accept;
    if (__ballot(b0.undecided | b1.undecided | b2.undecided | b3.undecided) != 0ull) {  // rare: grazing contact
        if (b0.undecided) h0 = box_exact(r, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y);
        if (b1.undecided) h1 = box_exact(r, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w);
        if (b2.undecided) h2 = box_exact(r, q3.x, q3.y, q3.z, q3.w, q4.x, q4.y);
        if (b3.undecided) h3 = box_exact(r, q4.z, q4.w, q5.x, q5.y, q5.z, q5.w);
    }
    // (a slot that does not exist carries an inverted box -- rejected on every axis -- and the empty reference)
    h0 &= ref0 != kWideEmpty; h1 &= ref1 != kWideEmpty; h2 &= ref2 != kWideEmpty; h3 &= ref3 != kWideEmpty;
    if (!ANYHIT) {
        // near plane of the box along the ray's major axis, as a ray parameter: beyond the closest hit -> nothing in it can win
        h0 = h0 & !(r.prune_ok & (dF0 * r.absSz > s.limit));
        h1 = h1 & !(r.prune_ok & (dF1 * r.absSz > s.limit));
        h2 = h2 & !(r.prune_ok & (dF2 * r.absSz > s.limit));
        h3 = h3 & !(r.prune_ok & (dF3 * r.absSz > s.limit));
    }
    // visiting order: near child's group first, the near grandchild first inside a group (ordered nodes: always left first)
    auto far_first = [&](unsigned m3) { return !(m3 & kMetaOrdered) && ((r.neg >> (m3 & 3u)) & 1u); };
    const bool swapN = far_first(meta & 7u), swapL = far_first((meta >> 3) & 7u), swapR = far_first((meta >> 6) & 7u);
    // left group in its order, right group in its order
    const int la = swapL ? ref1 : ref0, lb = swapL ? ref0 : ref1;
    const bool hla = swapL ? h1 : h0, hlb = swapL ? h0 : h1;
    const int ra = swapR ? ref3 : ref2, rb = swapR ? ref2 : ref3;
    const bool hra = swapR ? h3 : h2, hrb = swapR ? h2 : h3;
    const int o0 = swapN ? ra : la, o1 = swapN ? rb : lb, o2 = swapN ? la : ra, o3 = swapN ? lb : rb;
    const bool g0 = swapN ? hra : hla, g1 = swapN ? hrb : hlb, g2 = swapN ? hla : hra, g3 = swapN ? hlb : hrb;
    // push the accepted slots last-to-first, then take the top: the first accepted becomes the current node
    if (g3) { stack[s.sp * STRIDE] = o3; ++s.sp; }
    if (g2) { stack[s.sp * STRIDE] = o2; ++s.sp; }
    if (g1) { stack[s.sp * STRIDE] = o1; ++s.sp; }
    if (g0) {
        s.cur = o0;
    } else {
        trav_pop<STRIDE>(s, stack);
    }
}

// Tests the triangles of one leaf in leaf order; returns true when an any-hit query is done.
template <bool ANYHIT, bool STATS>
__device__ __forceinline__ bool leaf_step(Trav &s, const Ray &r, const Scene &sc, int leaf)
{
    const unsigned code = (unsigned)(~leaf);
    const unsigned first = code >> kLeafBits, count = (code & (kMaxLeaf - 1)) + 1;
    for (unsigned i = 0; i < count; ++i) {
        const float4 *tp = sc.tris + 3 * (size_t)(first + i);
        const float4 a = tp[0], b = tp[1], c = tp[2];
        ++s.steps;
        if (STATS) { ++s.n_tris; ++s.ray_tris; }
        if (STATS && (threadIdx.x & 63u) == (unsigned)__ffsll((long long)__ballot(true)) - 1u) ++s.w_tri_rounds;
        TriHit h;
        if (tri_test(r, a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z, h)) {
            s.any = true;
            if (ANYHIT) return true;
            if (h.t < s.best_t) {  // strict <, pspRT.cpp:395
                s.best_t = h.t;
                s.best_slot = (int)(first + i);
                s.limit = h.t + fabsf(h.t) * 4e-6f;
                // visibility ray: every triangle of the target node is farther than this hit,
                // so the closest hit cannot be one of them any more -- outcome decided
                if (h.t < s.own_min) return true;
            }
        }
    }
    return false;
}

// Runs the lane's traversal until it finishes (returns) -- or, when `more` work is
// queued, until fewer than `refill` lanes of the wave are still busy.  While-while with a CAPPED
// descent: at most `desc_cap` interior-node rounds, then the lanes that hold a leaf test it while the
// others wait (they continue their descent in the next round).  The uncapped form -- every live lane
// descends until ALL of them hold a leaf -- leaves the early lanes idle for as long as the deepest
// descent of the wave; with the cap both kernels gain (primary 0.288 -> 0.265 ms at 6 rounds,
// residual retries 0.179 -> 0.145 ms at 4; 1 round: 0.354 / 0.190, 2: 0.295 / 0.157, 8: 0.269 / 0.150).
// The sequence of operations of each LANE is unchanged (node steps down to a leaf, the leaf, pop), only
// the interleaving of the lanes differs, so results are bit-identical.
// (A speculative variant that parks one leaf and keeps descending was measured 15 % slower:
// the delayed pruning limit costs more node visits than the better lane occupancy saves.)
template <bool ANYHIT, bool STATS, int STRIDE = kBlock>
__device__ __forceinline__ void trav_run(Trav &s, const Ray &r, const Scene &sc, int *stack,
                                         bool more, int refill)
{
    const int cap = sc.desc_cap;
    unsigned rounds = 0;       // (the same in every lane that is still in the loop)
    while (s.cur != kDone) {
        if (++rounds > sc.round_cap) {      // never on a well-formed tree (Scene::round_cap)
            atomicOr(sc.err, 1u);
            s.cur = kDone;
            s.sp = 0;
            break;
        }
        if (cap == 0) {
            unsigned inner = 0;
            while (s.cur >= 0 && ++inner <= sc.round_cap) wide_step<ANYHIT, STATS, STRIDE>(s, r, sc, stack);
            if (s.cur >= 0) {                   // a descent longer than the tree: the same broken-tree verdict, at once
                atomicOr(sc.err, 1u);           // (not round_cap re-entries of round_cap steps each)
                s.cur = kDone;
                s.sp = 0;
                break;
            }
        } else {
            for (int d = 0; d < cap && __ballot(s.cur >= 0) != 0ull; ++d) {
                if (STATS && (threadIdx.x & 63u) == (unsigned)__ffsll((long long)__ballot(true)) - 1u) ++s.w_node_rounds;
                if (s.cur >= 0) wide_step<ANYHIT, STATS, STRIDE>(s, r, sc, stack);
            }
        }
        if (s.cur != kDone && s.cur < 0) {
            if (leaf_step<ANYHIT, STATS>(s, r, sc, s.cur)) {
                s.cur = kDone;
                s.sp = 0;
            } else {
                trav_pop<STRIDE>(s, stack);
            }
        }
        if (more && __popcll(__ballot(s.cur != kDone)) < refill) break;
        if (sc.heavy_steps && __ballot(s.cur != kDone && s.steps > sc.heavy_steps) != 0ull) break;
    }
}

__device__ __forceinline__ unsigned lane_id() { return threadIdx.x & 63u; }

// Wave-level work distribution.  `cur`/`end` are wave-uniform.
struct WaveQueue {
    unsigned *head;
    unsigned total;
    unsigned cur, end;
    unsigned chunk, base;
    bool exhausted;
};

// Every wave owns chunk #wave statically (no atomic storm at launch); further chunks
// come from the shared counter, offset by the statically assigned range.
__device__ __forceinline__ void queue_init(WaveQueue &q, unsigned *head, unsigned total,
                                           unsigned chunk, bool xcd_aware = true)
{
    q.head = head;
    q.total = total;
    q.chunk = chunk;
    const unsigned waves_per_block = blockDim.x >> 6;
    // XCD-aware static assignment: workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one), each
    // with its own 4-MiB L2.  Work items are in mesh order, i.e. spatially coherent, so the workgroups of one XCD take
    // one contiguous eighth of the item range and that XCD's L2 holds one region's subtrees instead of all of them.
    // (Placement is not guaranteed by HIP: this is for speed only, any mapping is a bijection.)
    const unsigned nb = gridDim.x, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned vb = xcd_aware ? xcd * (nb >> 3) + min(xcd, nb & 7u) + slot : blockIdx.x;
    const unsigned wave = vb * waves_per_block + (threadIdx.x >> 6);
    q.base = gridDim.x * waves_per_block * chunk;
    const unsigned long long first = (unsigned long long)wave * chunk;
    q.cur = first < total ? (unsigned)first : total;
    q.end = min(q.cur + chunk, total);
    // the statically assigned chunks already cover every item: nobody has to ask the shared counter
    // (one same-address atomic per wave, 6144 of them, just to learn that it is empty: ~25 us of the
    // residual retry traversal)
    q.exhausted = (unsigned long long)q.base >= total;
}

// Gives idle lanes (want == true) a work item; returns true and the index for lanes
// that received one.  Must be called by all lanes of the wave.
__device__ __forceinline__ bool queue_take(WaveQueue &q, bool want, unsigned &item)
{
    const unsigned long long idle = __ballot(want);
    if (idle == 0ull) return false;
    if (q.cur >= q.end) {
        if (q.exhausted) return false;
        unsigned c = 0;
        if (lane_id() == 0) c = atomicAdd(q.head, q.chunk);
        c = __builtin_amdgcn_readfirstlane(c);
        if (c >= q.total - min(q.total, q.base)) {
            q.exhausted = true;
            return false;
        }
        c += q.base;
        q.cur = c;
        q.end = min(c + q.chunk, q.total);
    }
    const unsigned n_idle = __popcll(idle);
    const unsigned take = min(n_idle, q.end - q.cur);
    const unsigned rank = __popcll(idle & ((1ull << lane_id()) - 1ull));
    const bool got = want && rank < take;
    item = q.cur + rank;
    q.cur += take;
    return got;
}

__device__ __forceinline__ bool queue_has_more(const WaveQueue &q)
{
    return q.cur < q.end || !q.exhausted;
}

// Hit record of the winning triangle: Hit.{u,v,w,t,pos,nrm,primID} (pspRT.cpp:173-190).
__device__ __forceinline__ void write_hit(const Ray &r, const Scene &sc, const Trav &s,
                                          size_t i, const upsp_hits &out)
{
    float t = FLT_MAX, u = 0.f, v = 0.f, w = 0.f;
    float px = 0.f, py = 0.f, pz = 0.f, nx = 0.f, ny = 0.f, nz = 0.f;
    int prim = -1;
    if (s.best_slot >= 0) {
        const float4 *tp = sc.tris + 3 * (size_t)s.best_slot;
        const float4 a = tp[0], b = tp[1], c = tp[2];
        TriHit h;
        tri_test(r, a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z, h);
        t = h.t; u = h.u; v = h.v; w = h.w;
        prim = __float_as_int(a.w);
        px = r.ox + t * r.dx;
        py = r.oy + t * r.dy;
        pz = r.oz + t * r.dz;
        const float e1x = b.x - a.x, e1y = b.y - a.y, e1z = b.z - a.z;
        const float e2x = c.x - a.x, e2y = c.y - a.y, e2z = c.z - a.z;
        const float EPS = 1e-3f;
        if (!(imath_length(e1x, e1y, e1z) < EPS || imath_length(e2x, e2y, e2z) < EPS)) {
            nx = e1y * e2z - e1z * e2y;
            ny = e1z * e2x - e1x * e2z;
            nz = e1x * e2y - e1y * e2x;
            if (nx * r.dx + ny * r.dy + nz * r.dz > 0.0f) { nx = -nx; ny = -ny; nz = -nz; }
        }
    }
    if (out.hit) out.hit[i] = s.any ? 1 : 0;
    if (out.t) out.t[i] = t;
    if (out.prim) out.prim[i] = prim;
    if (out.uvw) { out.uvw[3 * i] = u; out.uvw[3 * i + 1] = v; out.uvw[3 * i + 2] = w; }
    if (out.pos) { out.pos[3 * i] = px; out.pos[3 * i + 1] = py; out.pos[3 * i + 2] = pz; }
    if (out.nrm) { out.nrm[3 * i] = nx; out.nrm[3 * i + 1] = ny; out.nrm[3 * i + 2] = nz; }
}

// End-of-kernel statistics: wave shuffle reduction, block reduction through LDS (the
// traversal stack is dead by then), at most one atomic per counter and WORKGROUP --
// thousands of same-address atomics from every wave cost more than the traversal.
// Must be called by all threads of the block.  extra: optional 4th counter -> work[11].
__device__ __forceinline__ void flush_stats(unsigned *work, int *lds, unsigned n_nodes,
                                            unsigned n_tris, unsigned n_rays, unsigned extra = 0)
{
    for (int off = 32; off > 0; off >>= 1) {
        n_nodes += __shfl_down(n_nodes, off);
        n_tris += __shfl_down(n_tris, off);
        n_rays += __shfl_down(n_rays, off);
        extra += __shfl_down(extra, off);
    }
    __syncthreads();  // every wave has left the traversal loop: the stack area is free
    const unsigned wave = threadIdx.x >> 6;
    if (lane_id() == 0) {
        lds[wave * 4 + 0] = (int)n_nodes;
        lds[wave * 4 + 1] = (int)n_tris;
        lds[wave * 4 + 2] = (int)n_rays;
        lds[wave * 4 + 3] = (int)extra;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        unsigned v = 0;
        for (unsigned w = 0; w < (blockDim.x >> 6); ++w) v += (unsigned)lds[w * 4 + threadIdx.x];
        if (v) {
            if (threadIdx.x < 3)
                atomicAdd(reinterpret_cast<unsigned long long *>(work + 2) + threadIdx.x,
                          (unsigned long long)v);
            else
                atomicAdd(&work[11], v);
        }
    }
}

// ------------------------------------------------------------------------
//  Batch query kernel: n independent rays (closest hit or occlusion).
// ------------------------------------------------------------------------
constexpr int kWorkHeavyCast = 16;    // rays cast_kernel handed to heavy_cast_kernel (the projection's [16], [17] alike)
constexpr unsigned kHeavyCapCast = 65536;
constexpr int kWorkPrimaryCount = 18; // length of the dense ray list of the pass that runs (cast_entry_kernel /
                                      // retry_list_kernel<kPixInFrame>)

// A ray list spread over `pack_waves` waves (3 per SIMD: enough to overlap the node fetches of one wave with the box
// tests of another; measured with 193 k primary rays of a projection build: 64 per wave 149 us, 48: 156, 32: 180,
// 16: 194), at least 1 ray per wave (a few thousand rays run one or two to a wave, each for as long as its own chain
// of steps and no longer), at most 64.  Static assignment, nobody touches the shared counter; the waves that get any
// are spread evenly over the XCDs (workgroup b runs on XCD b & 7), one contiguous eighth of the list per XCD.
// Returns false when the grid cannot take the list at once (the caller falls back to the queue).
__device__ __forceinline__ bool queue_init_spread(WaveQueue &q, unsigned *work, unsigned total, unsigned pack_waves)
{
    if (!pack_waves) return false;
    const unsigned pack = min(max((total + pack_waves - 1u) / pack_waves, 1u), 64u);
    const unsigned wpb = blockDim.x >> 6;
    const unsigned nchunks = (total + pack - 1u) / pack;
    const unsigned per_xcd = ((nchunks + wpb - 1u) / wpb + 7u) >> 3;
    if (per_xcd * 8u > gridDim.x) return false;
    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned long long first = ((unsigned long long)(xcd * per_xcd + slot) * wpb + (threadIdx.x >> 6)) * pack;
    q.head = work;
    q.total = total;
    q.chunk = pack;
    q.base = 0;
    q.cur = (slot < per_xcd && first < total) ? (unsigned)first : total;
    q.end = min(q.cur + pack, total);
    q.exhausted = true;
    return true;
}

// The binned dense list of a repeated projection build (kRayBins lists, see kRayBins): this wave's view of every bin and its
// static chunk.  XCD x takes [cnt * x / 8, cnt * (x + 1) / 8) of EVERY bin -- node order inside a bin, so still one region of the
// model per XCD -- and its waves take `pack` consecutive items of the concatenation of those pieces (short bins first).  Item j of
// the local index space is looked up with bin_item().  Returns false (blo / blen = the whole bins, index space = their
// concatenation, caller falls back to the queue) when the grid cannot take the list at once.
constexpr int kWorkBinCountFwd = 22;
constexpr int kWorkWordsFwd = 32;
__device__ __forceinline__ bool queue_init_binned(WaveQueue &q, unsigned *work, unsigned total, unsigned pack_waves, unsigned nbins)
{
    const unsigned pack = pack_waves ? min(max((total + pack_waves - 1u) / pack_waves, 1u), 64u) : 0u;
    const unsigned wpb = blockDim.x >> 6;
    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    unsigned lmax = 0, L = 0;
#pragma unroll
    for (unsigned b = 0; b < 8u; ++b) {
        const unsigned cnt = b < nbins ? (unsigned)__builtin_amdgcn_readfirstlane((int)work[kWorkBinCountFwd + b]) : 0u;
        lmax += cnt / 8u + 1u;
        L += (unsigned)(((unsigned long long)cnt * (xcd + 1u)) >> 3) - (unsigned)(((unsigned long long)cnt * xcd) >> 3);
    }
    if (!pack || (unsigned long long)slots * wpb * pack < lmax) return false;
    const unsigned long long first = ((unsigned long long)slot * wpb + (threadIdx.x >> 6)) * pack;
    q.head = work;
    q.total = L;
    q.chunk = pack;
    q.base = 0;
    q.cur = (slot < slots && first < L) ? (unsigned)first : L;
    q.end = min(q.cur + pack, L);
    q.exhausted = true;
    return true;
}
// item j of that index space -> position in the bins' lists (per_xcd: the index space queue_init_binned set up; else the
// concatenation of the whole bins).  The bin counts are read again (six cached words per RAY, not per step): keeping the
// sub-ranges of every bin in registers across the traversal loop cost the kernel 40 more spilled scalar registers.
__device__ __forceinline__ unsigned bin_item(unsigned j, const unsigned *work, bool per_xcd, unsigned nbins, unsigned stride)
{
    const unsigned xcd = blockIdx.x & 7u;
    unsigned idx = 0;
    bool found = false;
#pragma unroll
    for (unsigned b = 0; b < 8u; ++b) {
        const unsigned cnt = b < nbins ? work[kWorkBinCountFwd + b] : 0u;
        const unsigned lo = per_xcd ? (unsigned)(((unsigned long long)cnt * xcd) >> 3) : 0u;
        const unsigned hi = per_xcd ? (unsigned)(((unsigned long long)cnt * (xcd + 1u)) >> 3) : cnt;
        const bool here = !found && j < hi - lo;
        idx = here ? b * stride + lo + j : idx;
        found |= here;
        j -= found ? 0u : hi - lo;
    }
    return idx;
}

// First pass of a large batch: the rays that miss the root box (trav_begin's test; on the bench's pixel rays 4 of 5)
// get their "no hit" record here, the others go on a dense list (one atomic per 2048 rays) that cast_kernel<LISTED>
// works off with queue_init_spread.  Same records as the one-pass form.
constexpr int kEntryItems = 8;
template <bool ANYHIT>
__global__ void __launch_bounds__(256)
    cast_entry_kernel(Scene sc, const float *__restrict__ org, int org_stride, const float *__restrict__ dir,
                      unsigned n, upsp_hits out, unsigned *__restrict__ list, unsigned *work)
{
    __shared__ unsigned wave_cnt[kEntryItems][4], block_base;
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned base = blockIdx.x * (256u * kEntryItems) + threadIdx.x;
    unsigned long long m[kEntryItems];
    bool need[kEntryItems];
#pragma unroll
    for (int k = 0; k < kEntryItems; ++k) {
        const unsigned i = base + 256u * k;
        need[k] = false;
        if (i < n) {
            const float *o = org + (size_t)org_stride * i;
            const float *d = dir + 3 * (size_t)i;
            Ray r;
            ray_setup(r, o[0], o[1], o[2], d[0], d[1], d[2]);
            ray_classify(r, sc);
            need[k] = box_hit(r, sc.rlo[0], sc.rlo[1], sc.rlo[2], sc.rhi[0], sc.rhi[1], sc.rhi[2]);
            if (!need[k]) {
                if (ANYHIT) {
                    if (out.hit) out.hit[i] = 0;
                } else {
                    Trav s;
                    s.any = false;
                    s.best_slot = -1;
                    write_hit(r, sc, s, i, out);
                }
            }
        }
        m[k] = __ballot(need[k]);
        if (lane == 0) wave_cnt[k][wave] = (unsigned)__popcll(m[k]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned tot = 0;
        for (int k = 0; k < kEntryItems; ++k)
            for (int w = 0; w < 4; ++w) {   // exclusive prefix in place
                const unsigned c = wave_cnt[k][w];
                wave_cnt[k][w] = tot;
                tot += c;
            }
        block_base = tot ? atomicAdd(&work[kWorkPrimaryCount], tot) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kEntryItems; ++k)
        if (need[k])
            list[block_base + wave_cnt[k][wave] + (unsigned)__popcll(m[k] & ((1ull << lane) - 1ull))] = base + 256u * k;
}

template <bool ANYHIT, bool STATS, bool LISTED = false>
__global__ void __launch_bounds__(kBlock)
    cast_kernel(Scene sc, const float *__restrict__ org, int org_stride,
                const float *__restrict__ dir, unsigned n, upsp_hits out, unsigned *work,
                const unsigned *__restrict__ list = nullptr)
{
    extern __shared__ int lds_stack[];
    int *stack = lds_stack + threadIdx.x;
    WaveQueue q;
    const unsigned total = LISTED ? work[kWorkPrimaryCount] : n;
    if (!(LISTED && queue_init_spread(q, work, total, sc.pack_waves))) queue_init(q, work, total, sc.chunk, sc.xcd != 0);
    Ray r;
    r.simple = true;   // idle lanes must not veto the wave-uniform fast path
    Trav s;
    s.cur = kDone;
    s.n_nodes = s.n_tris = 0;
    s.n_boxes = s.n_band = 0;
    s.w_node_rounds = s.w_tri_rounds = 0;
    unsigned item = 0, my_rays = 0;
    bool busy = false;

    for (;;) {
        // ---- re-fill idle lanes ----
        for (;;) {
            unsigned it;
            const bool got = queue_take(q, !busy, it);
            if (got) {
                if (LISTED) it = list[it];
                item = it;
                const float *o = org + (size_t)org_stride * it;
                const float *d = dir + 3 * (size_t)it;
                ray_setup(r, o[0], o[1], o[2], d[0], d[1], d[2]);
                ray_classify(r, sc);
                trav_begin(s, r, sc);
                busy = true;
                ++my_rays;
            }
            if (__ballot(!busy) == 0ull || !queue_has_more(q)) break;
        }
        if (__ballot(busy) == 0ull) break;
        // ---- traverse ----
        if (busy) {
            trav_run<ANYHIT, STATS>(s, r, sc, stack, queue_has_more(q), sc.refill);
            if (!STATS && sc.heavy_steps && s.cur != kDone && s.steps > sc.heavy_steps) {
                // a ray that keeps going (through a vertex shared by ~1000 triangles, along a stack of slivers): handed
                // to heavy_cast_kernel, where a whole wave walks it (see heavy_kernel)
                const unsigned slot = atomicAdd(&work[kWorkHeavyCast], 1u);
                if (slot < kHeavyCapCast) {
                    sc.heavy_items[slot] = item;
                    s.cur = kDone;
                    s.sp = 0;
                    busy = false;
                } else {
                    s.steps = 0;     // list full: carry on here
                }
            }
            if (busy && s.cur == kDone) {
                if (STATS) {
                    atomicMax(&work[8], s.ray_nodes);
                    atomicMax(&work[9], s.ray_tris);
                }
                if (ANYHIT) {
                    if (out.hit) out.hit[item] = s.any ? 1 : 0;
                } else {
                    write_hit(r, sc, s, item, out);
                }
                busy = false;
            }
        }
    }
    if (STATS) {
        if (s.w_node_rounds) atomicAdd(&work[14], s.w_node_rounds);
        if (s.w_tri_rounds) atomicAdd(&work[15], s.w_tri_rounds);
        unsigned nb = s.n_boxes, nu = s.n_band;       // slab filter: boxes seen / left to the mirrored filter (one atomic per wave)
        for (int off = 32; off > 0; off >>= 1) {
            nb += __shfl_down(nb, off);
            nu += __shfl_down(nu, off);
        }
        if (lane_id() == 0 && nb) {
            atomicAdd(&work[28], nb >> 2);            // (wide steps: 4 boxes each; a u32 holds 4 G steps)
            atomicAdd(&work[29], nu);
        }
    }
    if (STATS) flush_stats(work, lds_stack, s.n_nodes, s.n_tris, my_rays);
}

// ------------------------------------------------------------------------
//  Projection build: create_projection_mat (psp_process.cpp:167-355)
// ------------------------------------------------------------------------
struct Cam {
    double K[9], dist[5], R[9], t[3];
    float ox, oy, oz;  // camera centre narrowed to f32 (psp_process.cpp:193-194)
    int W, H;
};

// cv::projectPoints for one point (double arithmetic), result as Point2f.
__host__ __device__ inline void project_point(const double *K, const double *k, const double *R,
                                              const double *t, float X_, float Y_, float Z_,
                                              float &u, float &v)
{
    const double X = X_, Y = Y_, Z = Z_;
    double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
    double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
    double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
    z = z ? 1. / z : 1;
    x *= z;
    y *= z;
    const double r2 = x * x + y * y;
    const double r4 = r2 * r2;
    const double r6 = r4 * r2;
    const double a1 = 2 * x * y;
    const double a2 = r2 + 2 * x * x;
    const double a3 = r2 + 2 * y * y;
    const double cdist = 1 + k[0] * r2 + k[1] * r4 + k[4] * r6;
    const double xd = x * cdist + k[2] * a1 + k[3] * a2;
    const double yd = y * cdist + k[2] * a3 + k[3] * a1;
    u = (float)(xd * K[0] + K[2]);
    v = (float)(yd * K[4] + K[5]);
}

__device__ __forceinline__ bool tri_has_node(const int32_t *tri_nodes, int prim, int nidx)
{
    const int32_t *t = tri_nodes + 3 * (size_t)prim;
    return (t[0] == nidx) | (t[1] == nidx) | (t[2] == nidx);
}

// Node states kept in pix[] while the projection build is in flight
constexpr int32_t kPixNone = -1;      // no entry in the sparse matrix (final)
constexpr int32_t kPixVisible = -2;   // seen by a ray, oblique test pending
constexpr int32_t kPixInFrame = -3;   // projects into the frame, primary ray pending
constexpr int32_t kPixRetry = -4;     // primary ray hit a foreign triangle: retries pending

// Work words of the projection build (upsp_bvh::d_work):
//   [0] queue head   [2..7] three 64-bit statistics   [8],[9] longest ray (stats)
//   [10] nodes whose primary ray hit a triangle that does not contain them
//   [11] primary rays cast
constexpr int kWorkRetryCount = 10;
constexpr int kWorkTodoCount = 12;   // rays witness_kernel could not decide
constexpr int kWorkHeavyCount = 16;  // work items handed to heavy_kernel: [16] by the primary pass, [17] by the retry pass
constexpr int kWorkWords = kWorkWordsFwd;
constexpr unsigned kHeavyCap = 65536;
// Length-homogeneous waves.  The 64 rays of a wave run in lock step and the wave lasts as long as its longest ray: on the bench model
// a wave's longest ray has 36 steps for 25 of the mean (tools/probe/trav_policy_sim.c).  A build that is REPEATED on the same model
// (model motion: bench.py rebuilds the projection every step) knows every ray's length from the build before it: the dense list of
// the primary pass is cut into kRayBins bins by that step count (bin edges = sampled quantiles, step_edges_kernel), node order kept
// inside a bin, and a wave takes 64 consecutive rays of ONE bin -- simulated: longest ray of a wave 36.3 -> 27 steps, rounds per
// wave 54 -> 44.  Every XCD takes an eighth of EVERY bin (the long rays would otherwise all land on one XCD).  The first build of a
// handle (or one with another node count) uses the plain list.  Ordering only: every ray's own sequence of steps is unchanged.
// MEASURED (round 6, bench model, three alternations): the primary pass alone 0.143 -> 0.160 ms, the build alone 0.295 -> 0.35 ms,
// the default step 0.823-0.831 -> 0.878-0.883 ms.  The rounds do drop, but a wave's 64 rays now come from a six times wider stretch
// of the node list: their paths through the tree share fewer records, and the pass is bound by the latency of those fetches, not by
// the lanes that idle in a round (the instruction count is not it either: the slab filter below removes a quarter of a wide step's
// box-test instructions for 3 % of the build).  OFF by default (UPSP_RAY_BINS=1 turns it on); kept for models whose ray lengths
// spread more than their paths diverge.
constexpr unsigned kRayBins = 6;          // (the lists of bins 0..5 share d_todo_rays: 6 x nnodes entries)
constexpr int kWorkBinCount = kWorkBinCountFwd;   // [22..27]: rays per bin

// Step 1 (elementwise, fp64): cal.map_point_to_image + in-frame test
// (psp_process.cpp:241-252).  The image point is parked in uv[].
// The oblique test of psp_process.cpp:298-306: angle between the node's normal and the PRIMARY ray direction
// (normalised camera -> node vector, Imath arithmetic), acos in double, compared as float.
__device__ __forceinline__ bool oblique_forward(const Cam &cam, const float *__restrict__ nodes,
                                                const float *__restrict__ normals, unsigned n, float oblique_thresh)
{
    float dx = nodes[3 * (size_t)n] - cam.ox, dy = nodes[3 * (size_t)n + 1] - cam.oy,
          dz = nodes[3 * (size_t)n + 2] - cam.oz;
    const float len = imath_length(dx, dy, dz);
    if (len != 0.0f) { dx /= len; dy /= len; dz /= len; }
    const float *nn = normals + 3 * (size_t)n;
    const float cos_theta = nn[0] * dx + nn[1] * dy + nn[2] * dz;
    const float theta = (float)acos((double)cos_theta);
    return theta > oblique_thresh;
}

// cull: the reference casts the rays of a node first and applies the oblique test to the nodes they see
// (psp_process.cpp:254-306); the test itself uses nothing the rays produce (node normal, primary direction), and a
// node that fails it gets no entry whatever the rays said.  With `cull` it is therefore applied HERE and the nodes
// that fail never cast a ray: same matrix entries, uv and node-count image -- on a closed body two thirds of the
// in-frame nodes (every back-facing one, which are also the ones that go through the six retries) drop out.
// Without it the rays are cast as the reference casts them (needed for its ray count, upsp_projection_build h_nrays).
__global__ void __launch_bounds__(256)
    project_nodes_kernel(Cam cam, const float *__restrict__ nodes,
                         const uint8_t *__restrict__ datanode, unsigned nnodes,
                         const float *__restrict__ normals, float oblique_thresh, int cull,
                         int32_t *__restrict__ pix, float *__restrict__ uv, unsigned *__restrict__ work)
{
    // the build's work words start from zero: cleared here, by the first launch of the chain (a hipMemsetAsync is a launch of its
    // own -- 6 us on the build's critical path beside the frame loop's passes)
    if (blockIdx.x == 0 && threadIdx.x < (unsigned)kWorkWordsFwd) work[threadIdx.x] = 0u;
    const unsigned n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nnodes) return;
    int32_t state = kPixNone;
    float u = 0.f, v = 0.f;
    if (!datanode || datanode[n]) {  // :241
        project_point(cam.K, cam.dist, cam.R, cam.t, nodes[3 * (size_t)n], nodes[3 * (size_t)n + 1],
                      nodes[3 * (size_t)n + 2], u, v);  // :248
        // upsp::contains(Size, Point2i(pt)) :252 ; Point2f->Point2i = cvRound (cvtss2si: NaN and
        // values beyond the int range give INT_MIN, i.e. out of frame)
        const bool finite_int = (fabsf(u) < 2147483648.0f) & (fabsf(v) < 2147483648.0f);   // false for NaN
        const int rx = finite_int ? (int)rintf(u) : -1, ry = finite_int ? (int)rintf(v) : -1;
        if ((rx >= 0) & (ry >= 0) & (rx < cam.W) & (ry < cam.H))
            state = kPixInFrame;
        if (cull && state == kPixInFrame && !oblique_forward(cam, nodes, normals, n, oblique_thresh))
            state = kPixNone;
    }
    pix[n] = state;
    uv[2 * (size_t)n] = state == kPixInFrame ? u : 0.f;      // default uv = (0,0) (:179-182)
    uv[2 * (size_t)n + 1] = state == kPixInFrame ? v : 0.f;
}

// Step 1 + the pixel of :319 for every in-frame node, whatever the rays will say (upsp_projection_candidate_pixels)
// normals != nullptr: ... and only for the nodes that pass the oblique test of :298-306 (a node that fails it has no entry
// whatever its rays say): the candidates are then exactly the nodes that cast a primary ray in upsp_projection_build
__global__ void __launch_bounds__(256)
    candidate_pixels_kernel(Cam cam, const float *__restrict__ nodes, const uint8_t *__restrict__ datanode,
                            unsigned nnodes, const float *__restrict__ normals, float oblique_thresh,
                            int32_t *__restrict__ pix)
{
    const unsigned n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nnodes) return;
    int32_t out = -1;
    if (!datanode || datanode[n]) {
        float u, v;
        project_point(cam.K, cam.dist, cam.R, cam.t, nodes[3 * (size_t)n], nodes[3 * (size_t)n + 1],
                      nodes[3 * (size_t)n + 2], u, v);
        const bool finite_int = (fabsf(u) < 2147483648.0f) & (fabsf(v) < 2147483648.0f);
        const int rx = finite_int ? (int)rintf(u) : -1, ry = finite_int ? (int)rintf(v) : -1;
        if ((rx >= 0) & (ry >= 0) & (rx < cam.W) & (ry < cam.H) &&
            (!normals || oblique_forward(cam, nodes, normals, n, oblique_thresh))) {
            const int px_ = (int)roundf(u), py_ = (int)roundf(v);
            const long long idx = (long long)py_ * cam.W + px_;
            if (idx >= 0 && idx < (long long)cam.W * cam.H) out = (int32_t)idx;
        }
    }
    pix[n] = out;
}

// Step 2 -- one PRIMARY ray per in-frame node (psp_process.cpp:254-267).  Nodes hit on
// a foreign triangle go to the retry list.
// Step 3 -- the <= 6 jittered retries (psp_process.cpp:269-296) of every listed node
// as INDEPENDENT rays, six adjacent work items per node.  The reference stops at the
// first retry that sees the node; the outcome ("any retry sees it") is the same, so
// all six are cast and the reference's ray count is reconstructed from the bit mask.
// Visibility rays (camera -> a model node) only ask whether the CLOSEST hit lies on a triangle
// of that node (psp_process.cpp:263-267, 289-294).  With the node -> triangle adjacency the ray
// is first tested against the node's own triangles (same tri_test arithmetic as the traversal):
//   * their largest t bounds the search: a triangle farther than that is never the answer, so
//     the traversal starts with that pruning limit instead of +inf (nothing behind the node is
//     visited);
//   * their smallest t is an exit: once the running closest hit is nearer than every own
//     triangle the outcome is "foreign triangle" whatever else is found.
// The traversal itself is unchanged (same boxes accepted, same order, strict <), so the verdict
// is the reference's: an own triangle the reference cannot reach through its box tests is not
// reached here either, it only tightens nothing.  kMaxOwn caps the direct tests (polar fans).
constexpr unsigned kMaxOwn = 16;
struct OwnBound {
    float tmin, tmax;
    bool known, hit;   // known: adjacency available and small enough; hit: some own triangle is hit
};
__device__ __forceinline__ OwnBound own_bound(const Ray &r, const Scene &sc, unsigned node)
{
    OwnBound o;
    o.tmin = __builtin_inff();
    o.tmax = -__builtin_inff();
    o.known = o.hit = false;
    if (!sc.adj_off) return o;
    const unsigned b = sc.adj_off[node], e = sc.adj_off[node + 1];
    if (e - b > kMaxOwn) return o;
    o.known = true;
    for (unsigned k = b; k < e; ++k) {
        const float4 *tp = sc.tris + 3 * (size_t)sc.adj_slot[k];
        const float4 a = tp[0], bb = tp[1], c = tp[2];
        TriHit h;
        if (tri_test(r, a.x, a.y, a.z, bb.x, bb.y, bb.z, c.x, c.y, c.z, h)) {
            o.hit = true;
            o.tmin = fminf(o.tmin, h.t);
            o.tmax = fmaxf(o.tmax, h.t);
        }
    }
    return o;
}

// own_bound with the loads batched (witness_kernel, which has registers to spare): all adjacent
// slots first, then the triangle records four at a time, tests in the same order -> same result.
__device__ __forceinline__ OwnBound own_bound_batched(const Ray &r, const Scene &sc, unsigned node)
{
    OwnBound o;
    o.tmin = __builtin_inff();
    o.tmax = -__builtin_inff();
    o.known = o.hit = false;
    if (!sc.adj_off) return o;
    const unsigned b = sc.adj_off[node], e = sc.adj_off[node + 1];
    const unsigned cnt = e - b;
    if (cnt > kMaxOwn) return o;
    o.known = true;
    for (unsigned k0 = 0; k0 < cnt; k0 += 4u) {
        unsigned slot[4];
#pragma unroll
        for (unsigned i = 0; i < 4u; ++i) slot[i] = k0 + i < cnt ? sc.adj_slot[b + k0 + i] : 0u;
        float4 ta[4], tb[4], tc[4];
#pragma unroll
        for (unsigned i = 0; i < 4u; ++i) {
            const float4 *tp = sc.tris + 3 * (size_t)slot[i];
            ta[i] = tp[0]; tb[i] = tp[1]; tc[i] = tp[2];
        }
#pragma unroll
        for (unsigned i = 0; i < 4u; ++i) {
            TriHit h;
            if (k0 + i < cnt && tri_test(r, ta[i].x, ta[i].y, ta[i].z, tb[i].x, tb[i].y, tb[i].z, tc[i].x, tc[i].y, tc[i].z, h)) {
                o.hit = true;
                o.tmin = fminf(o.tmin, h.t);
                o.tmax = fmaxf(o.tmax, h.t);
            }
        }
    }
    return o;
}

// Applies the bound to a freshly started traversal.  Returns false when the ray needs no
// traversal at all (retry rays that miss every own triangle cannot see the node).
template <int PHASE>
__device__ __forceinline__ bool apply_own_bound(Trav &s, const OwnBound &o)
{
    if (!o.known) return true;
    if (!o.hit) return PHASE == 0;   // primary: still has to learn whether anything is hit (:261)
    s.limit = o.tmax + fabsf(o.tmax) * 4e-6f;
    s.own_min = o.tmin;
    return true;
}

// Occluder witness.  A node is listed for retries because its primary ray hit a foreign triangle W
// first.  The six retry rays differ from the primary ray by 1e-4 model units at the node, so they
// almost always hit W as well.  A retry ray is decided "not visible" without any traversal when
//   * W is hit at t_w < own_min (nearer than every triangle of the node, tested directly), and
//   * the reference's traversal reaches W: the root box (trav_begin) and every child box on the
//     chain root -> leaf(W) accept the ray (the reference descends on the box test alone, no
//     pruning by distance -- cpp/raycast/pspRT.cpp:380-423), evaluated with the traversal's own box test.
// Then the closest hit the reference finds is at t <= t_w < own_min, i.e. on a foreign triangle.
// Anything else (W missed, not nearer, a box on the chain rejected) falls back to the traversal.
__device__ __forceinline__ bool box_accepts(const Ray &r, bool all_simple, const float2 a, const float2 b,
                                            const float2 c)
{
    float dF;
    const BoxEval e = all_simple ? box_filter_simple(r, a.x, a.y, b.x, b.y, c.x, c.y, dF)
                                 : box_filter(r, a.x, a.y, b.x, b.y, c.x, c.y, dF);
    return e.undecided ? box_exact(r, a.x, a.y, b.x, b.y, c.x, c.y) : e.accept;
}

// Retry k of a listed node (psp_process.cpp:269-281): target = node position +- 1e-4 on one axis,
// UN-normalised direction.
__device__ __forceinline__ void retry_ray(Ray &r, const Cam &cam, const float *__restrict__ nodes,
                                          unsigned node, int k)
{
    const float L = 1e-4f;
    const float sgn = (k & 1) ? L : -L;
    const float qx = nodes[3 * (size_t)node] + ((k >> 1) == 0 ? sgn : 0.0f);
    const float qy = nodes[3 * (size_t)node + 1] + ((k >> 1) == 1 ? sgn : 0.0f);
    const float qz = nodes[3 * (size_t)node + 2] + ((k >> 1) == 2 ? sgn : 0.0f);
    ray_setup(r, cam.ox, cam.oy, cam.oz, qx - cam.ox, qy - cam.oy, qz - cam.oz);
}

// Step 3a: decides the retry rays that need no traversal.  One wave = kWitNodes listed nodes x 6
// rays (lane = 6 * node + retry).  Per ray: root box, the node's own triangles (own_bound), the
// witness triangle; then the box chains of the wave's witness leaves are staged through LDS by
// all lanes at once (every load independent: the per-ray walk inside the traversal kernel was a
// chain of ~25 dependent loads and cost more than the traversals it saved) and each ray tests
// its chain from LDS.  Output: todo_mask[listed node] = retries that still need the traversal.
constexpr int kWitNodes = 10;
constexpr int kWitSeg = 16;   // chain boxes staged per node and pass
__global__ void __launch_bounds__(64)
    witness_kernel(Scene sc, Cam cam, const float *__restrict__ nodes,
                   const unsigned *__restrict__ retry_nodes, unsigned *__restrict__ todo_mask,
                   const unsigned *__restrict__ work)
{
    __shared__ float boxes[kWitNodes][kWitSeg][6];
    __shared__ unsigned chain_off[kWitNodes], chain_len[kWitNodes];
    const unsigned count = work[kWorkRetryCount];
    const unsigned lane = threadIdx.x;
    const unsigned j = lane / 6u, k = lane % 6u;
    const unsigned li = blockIdx.x * kWitNodes + j;
    if (blockIdx.x * kWitNodes >= count) return;          // whole wave past the list (uniform)
    const bool active = lane < 6u * kWitNodes && li < count;
    if (lane < kWitNodes) chain_len[lane] = 0u;
    __syncthreads();
    Ray r;
    r.simple = true;
    bool undecided = false, chain = false;
    unsigned len = 0;
    if (active) {
        const unsigned node = retry_nodes[li];
        retry_ray(r, cam, nodes, node, (int)k);
        ray_classify(r, sc);
        int why = 48;   // debug verdict code (sc.hist)
        if (box_hit(r, sc.rlo[0], sc.rlo[1], sc.rlo[2], sc.rhi[0], sc.rhi[1], sc.rhi[2])) {   // trav_begin
            const OwnBound ob = own_bound_batched(r, sc, node);
            why = 50;
            if (!ob.known) {
                undecided = true;
                why = 49;
            } else if (ob.hit) {              // (no own triangle hit: cannot see the node, decided)
                const int slot = sc.witness[node];
                bool near_hit = false;
                why = 51;
                if (slot >= 0) {
                    const float4 *tp = sc.tris + 3 * (size_t)slot;
                    const float4 a = tp[0], b = tp[1], c = tp[2];
                    TriHit h;
                    const bool wh = tri_test(r, a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z, h);
                    near_hit = wh && h.t < ob.tmin;
                    why = !wh ? 52 : near_hit ? 55 : 53;
                }
                if (near_hit) {
                    const uint2 pl = sc.slot_path[slot];
                    chain = true;
                    len = pl.y;
                    chain_off[j] = pl.x;      // the six rays of a node write the same values
                    chain_len[j] = pl.y;
                } else {
                    undecided = true;
                }
            }
        }
        if (sc.hist) atomicAdd(&sc.hist[why], 1u);
    }
    __syncthreads();
    if (__ballot(chain) != 0ull) {
        unsigned maxlen = 0;
        for (int n = 0; n < kWitNodes; ++n) maxlen = max(maxlen, chain_len[n]);
        const bool all_simple = __ballot(!r.simple) == 0ull;
        const float2 *nodes2 = reinterpret_cast<const float2 *>(sc.nodes);
        bool ok = true;
        for (unsigned base = 0; base < maxlen; base += kWitSeg) {
            // stage kWitNodes x kWitSeg boxes: record = 16 floats, left box = floats 0..5, right 6..11
            for (unsigned b = lane; b < kWitNodes * kWitSeg; b += 64u) {
                const unsigned jj = b / kWitSeg, e = base + b % kWitSeg;
                if (e < chain_len[jj]) {
                    const unsigned ref = sc.path_ref[chain_off[jj] + e];
                    const float2 *p = nodes2 + 8 * (size_t)(ref >> 1) + 3 * (ref & 1u);
                    const float2 a = p[0], bb = p[1], c = p[2];
                    float *d = boxes[jj][b % kWitSeg];
                    d[0] = a.x; d[1] = a.y; d[2] = bb.x; d[3] = bb.y; d[4] = c.x; d[5] = c.y;
                }
            }
            __syncthreads();
            const unsigned n = (chain && len > base) ? min((unsigned)kWitSeg, len - base) : 0u;
            for (unsigned e = 0; e < (unsigned)kWitSeg; ++e) {
                if (__ballot(e < n && ok) == 0ull) break;
                if (e < n && ok) {
                    const float *d = boxes[j][e];
                    ok = box_accepts(r, all_simple, make_float2(d[0], d[1]), make_float2(d[2], d[3]),
                                     make_float2(d[4], d[5]));
                }
            }
            __syncthreads();
        }
        if (chain && !ok) {   // a box on the chain rejects the ray: ask the traversal
            undecided = true;
            if (sc.hist) atomicAdd(&sc.hist[54], 1u);
        }
    }
    const unsigned long long m = __ballot(undecided);
    if (active && k == 0u) todo_mask[li] = (unsigned)((m >> (6u * j)) & 63ull);
}

// Step 3b: compact the undecided retries into a ray list (item = 6 * listed node + retry); one
// queue atomic per workgroup of 1024 listed nodes.
__global__ void __launch_bounds__(256)
    todo_list_kernel(const unsigned *__restrict__ todo_mask, unsigned *__restrict__ todo_rays,
                     unsigned *work)
{
    constexpr int kItems = 4;
    __shared__ unsigned wave_cnt[kItems][4], block_base;
    const unsigned count = work[kWorkRetryCount];
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned base = blockIdx.x * (256u * kItems) + threadIdx.x;
    if (blockIdx.x * (256u * kItems) >= count) return;
    unsigned mask[kItems], pre[kItems];
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
        const unsigned li = base + 256u * i;
        mask[i] = li < count ? todo_mask[li] : 0u;
        unsigned c = (unsigned)__popc(mask[i]), incl = c;   // inclusive wave scan of the counts
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned v = __shfl_up(incl, off);
            if ((int)lane >= off) incl += v;
        }
        pre[i] = incl - c;
        if (lane == 63) wave_cnt[i][wave] = incl;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned tot = 0;
        for (int i = 0; i < kItems; ++i)
            for (int w = 0; w < 4; ++w) {
                const unsigned c = wave_cnt[i][w];
                wave_cnt[i][w] = tot;
                tot += c;
            }
        block_base = tot ? atomicAdd(&work[kWorkTodoCount], tot) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
        unsigned slot = block_base + wave_cnt[i][wave] + pre[i];
        const unsigned li = base + 256u * i;
        for (unsigned mm = mask[i]; mm; mm &= mm - 1u)
            todo_rays[slot++] = 6u * li + (unsigned)(__ffs((int)mm) - 1);
    }
}

// BS = threads per workgroup.  The passes that run beside other work (the residual retries: the frame loop's pass A holds the
// LDS of every CU in 16-KB pieces by then) use one-wave workgroups: a 10-KB stack fits a hole a 40-KB one waits for
// (measured beside pass A: 0.21 against 0.07 ms alone for the 4-wave form).
template <bool STATS, int PHASE, int BS>
__global__ void __launch_bounds__(BS) __attribute__((amdgpu_waves_per_eu(4, 8)))
    projection_kernel(Scene sc, Cam cam, const float *__restrict__ nodes,
                      const int32_t *__restrict__ tri_nodes, unsigned nnodes,
                      int32_t *__restrict__ pix, unsigned *__restrict__ retry_nodes,
                      unsigned *__restrict__ retry_mask, const unsigned *__restrict__ todo_rays,
                      unsigned *work)
{
    extern __shared__ int lds_stack[];
    int *stack = lds_stack + threadIdx.x;
    const unsigned total = PHASE == 0 ? work[kWorkPrimaryCount] : PHASE == 1 ? work[kWorkRetryCount] * 6u : work[kWorkTodoCount];
    // PHASE 0 works off the dense list of nodes that cast a ray (retry_list_kernel<kPixInFrame>; `todo_rays` holds it
    // during this pass).  When the resident waves can take the whole list at once the assignment is static and nobody
    // touches the shared counter: after the early oblique test a camera has ~200 k rays for ~6000 resident waves, and
    // draining a sparse item range through the queue (thousands of same-address atomics, chunks without a single ray)
    // cost more than the traversals -- the pass took 160 us with 1400 rays and 250 us with 193 k.
    WaveQueue q;
    const bool binned = PHASE == 0 && sc.nbins != 0u;      // a repeated build: the list in bins by the rays' previous lengths
    bool bins_per_xcd = false;
    if (binned) {
        bins_per_xcd = queue_init_binned(q, work, total, sc.pack_waves, sc.nbins);
        if (!bins_per_xcd) queue_init(q, work, total, sc.chunk, false);
    } else {
        const bool packed = queue_init_spread(q, work, total, sc.pack_waves);
        if (!packed) queue_init(q, work, total, sc.chunk, sc.xcd != 0 && PHASE == 0);   // (retry lists are not in mesh order)
    }
    Ray r;
    r.simple = true;   // idle lanes must not veto the wave-uniform fast path
    Trav s;
    s.cur = kDone;
    s.n_nodes = s.n_tris = 0;
    s.n_boxes = s.n_band = 0;
    s.w_node_rounds = s.w_tri_rounds = 0;
    unsigned item = 0, my_rays = 0;
    bool busy = false, bounded = false;

    for (;;) {
        for (;;) {
            unsigned it;
            const bool got = queue_take(q, !busy, it);
            if (got && PHASE == 0) {
                it = todo_rays[binned ? bin_item(it, work, bins_per_xcd, sc.nbins, sc.bin_stride) : it];
                {
                    float dx = nodes[3 * (size_t)it] - cam.ox, dy = nodes[3 * (size_t)it + 1] - cam.oy,
                          dz = nodes[3 * (size_t)it + 2] - cam.oz;
                    const float len = imath_length(dx, dy, dz);  // .normalize() :256
                    if (len != 0.0f) { dx /= len; dy /= len; dz /= len; }
                    ray_setup(r, cam.ox, cam.oy, cam.oz, dx, dy, dz);
                    ray_classify(r, sc);
                    trav_begin(s, r, sc);
                    item = it;
                    busy = true;
                    ++my_rays;
                }
            }
            if (got && PHASE >= 1) {
                // PHASE 1: every retry of every listed node; PHASE 2: the rays witness_kernel left
                const unsigned ray = PHASE == 2 ? todo_rays[it] : it;
                const unsigned node = retry_nodes[ray / 6u];
                retry_ray(r, cam, nodes, node, (int)(ray % 6u));
                ray_classify(r, sc);
                trav_begin(s, r, sc);
                item = ray;
                ++my_rays;
                busy = apply_own_bound<1>(s, own_bound(r, sc, node));   // false: cannot see it
            }
            if (__ballot(!busy) == 0ull || !queue_has_more(q)) break;
        }
        if (__ballot(busy) == 0ull) break;

        if (busy) {
            trav_run<false, STATS, BS>(s, r, sc, stack, queue_has_more(q), sc.refill);
            if (sc.heavy_steps && s.cur != kDone && s.steps > sc.heavy_steps) {
                // a ray that keeps going (e.g. through a vertex shared by ~1000 triangles: every box of the fan is
                // pierced, every triangle hit) would hold its wave for as long as ONE lane needs for thousands of
                // dependent steps: hand it to heavy_kernel, where a whole wave walks it
                const unsigned slot = atomicAdd(&work[kWorkHeavyCount + (PHASE ? 1 : 0)], 1u);
                if (slot < kHeavyCap) {
                    sc.heavy_items[slot] = item;
                    if (PHASE == 0 && sc.steps_out) sc.steps_out[item] = (unsigned short)0xFFFFu;      // (the last bin)
                    s.cur = kDone;
                    s.sp = 0;
                    busy = false;
                } else {
                    s.steps = 0;     // list full: carry on here (and do not ask again at once)
                }
            }
            if (busy && PHASE == 0 && s.cur == kDone && bounded && !s.any) {
                // An own triangle is hit when tested directly but was not reached through the
                // boxes, and nothing nearer exists: whether the ray hits ANYTHING (retries or
                // no entry, :261) is decided beyond the bound -> classic unbounded traversal.
                bounded = false;
                trav_begin(s, r, sc);
            }
            if (busy && s.cur == kDone) {
                busy = false;
                if (STATS && PHASE == 2 && sc.hist)
                    atomicAdd(&sc.hist[min(47u, (s.ray_nodes + s.ray_tris) >> 4)], 1u);
                const unsigned node = PHASE == 0 ? item : retry_nodes[item / 6u];
                bool visible = false;
                if (s.any && s.best_slot >= 0) {
                    const int prim = __float_as_int(sc.tris[3 * (size_t)s.best_slot].w);
                    visible = tri_has_node(tri_nodes, prim, (int)node);  // :263-267 / :289-294
                }
                if (PHASE == 0) {
                    // primary ray missed everything -> no entry (:261); hit on a foreign
                    // triangle -> jittered retries (listed by retry_list_kernel)
                    if (sc.steps_out) sc.steps_out[node] = (unsigned short)min(max(s.steps, 1u), 0xFFFEu);
                    pix[node] = visible ? kPixVisible : (s.any ? kPixRetry : kPixNone);
                    if (sc.witness && !visible && s.any) sc.witness[node] = s.best_slot;
                } else if (visible) {
                    atomicOr(&retry_mask[item / 6u], 1u << (item % 6u));
                }
            }
        }
    }
    if (STATS) {
        if (s.w_node_rounds) atomicAdd(&work[14], s.w_node_rounds);
        if (s.w_tri_rounds) atomicAdd(&work[15], s.w_tri_rounds);
        unsigned nb = s.n_boxes, nu = s.n_band;       // slab filter: boxes seen / left to the mirrored filter (one atomic per wave)
        for (int off = 32; off > 0; off >>= 1) {
            nb += __shfl_down(nb, off);
            nu += __shfl_down(nu, off);
        }
        if (lane_id() == 0 && nb) {
            atomicAdd(&work[28], nb >> 2);            // (wide steps: 4 boxes each; a u32 holds 4 G steps)
            atomicAdd(&work[29], nu);
        }
    }
    flush_stats(work, lds_stack, STATS ? s.n_nodes : 0u, STATS ? s.n_tris : 0u,
                PHASE == 0 ? my_rays : 0u, PHASE == 0 ? my_rays : 0u);
}

// One ray per WAVE.  The work items projection_kernel handed over are walked with a wave-wide stack in LDS:
// every lane pops one entry -- an interior node (both child boxes tested with the traversal's own filtered /
// exact box test, accepted children pushed) or a leaf (its triangles tested) -- so a ray that pierces thousands
// of boxes advances 64 of them per step instead of one.  No pruning: exactly the boxes the reference enters
// (pspRT.cpp:380-423).  The reference keeps the FIRST minimum in its near-child-first depth-first order
// (strict <, :395); here every entry carries its position in that order as a key -- one bit per level (0 =
// the child visited first), the triangle's position in its leaf in the low bits -- and the winner is the
// smallest t, then the smallest key.  Same closest hit, same verdict as the one-lane traversal.
constexpr unsigned kHeavyStack = 4096;
// The stack of a cooperative walk lives in GLOBAL memory (one region per workgroup: keys, refs, depths), not in LDS: with 53 KB of
// static LDS per workgroup the kernel -- which on most models only finds an empty list and returns -- could not start beside
// the frame loop's pass A, whose workgroups hold the LDS of every CU in 16-KB pieces (measured: 0.14 ms from launch to end
// per empty launch, two launches on the critical path of every projection build; 0.007 ms alone).  The region stays in L2;
// a walk's round costs two more cache round trips, rays that need the walk are rare by design.
constexpr size_t kHeavyScratchBytes = (size_t)kHeavyStack * (8 + 4 + 1);
constexpr unsigned kHeavyGridMax = 512;
struct HeavyBest {
    float t;
    unsigned long long key;
    int slot;
};
// The workgroup-wide walk of one ray (see heavy_kernel): all kHeavyThreads threads call it with the same ray; q_* are
// the workgroup's LDS stack arrays (kHeavyStack entries).  Out: whether any triangle is hit and the slot of the
// reference's closest hit.  STOP_AT_ANY (occlusion queries): the walk ends with the first round that finds a hit.
// (Four waves: with one, the ray through a 1000-triangle fan -- a frontier of ~1000 open boxes -- took 16 rounds per
// level of the fan where it now takes 4: heavy_kernel 100 -> see DESIGN.md.)
constexpr unsigned kHeavyThreads = 256;
template <bool STOP_AT_ANY>
__device__ __forceinline__ void heavy_walk(const Ray &r, const Scene &sc, int *q_ref, unsigned long long *q_key,
                                           unsigned char *q_depth, bool &any_w, int &best_slot)
{
    constexpr unsigned NW = kHeavyThreads / 64;
    __shared__ unsigned s_cnt[NW], s_flag[2];            // per-wave push counts; [0] overflow, [1] any hit
    __shared__ float s_bt[NW];
    __shared__ unsigned long long s_bk[NW];
    __shared__ int s_bs[NW], s_any[NW];
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    unsigned top = 0;                         // (uniform) entries on the stack
    if (box_hit(r, sc.rlo[0], sc.rlo[1], sc.rlo[2], sc.rhi[0], sc.rhi[1], sc.rhi[2])) {   // trav_begin
        if (tid == 0) {
            q_ref[0] = sc.root_ref2;
            q_key[0] = 0ull;
            q_depth[0] = 0;
        }
        top = 1;
    }
    if (tid < 2) s_flag[tid] = 0u;
    __syncthreads();
    HeavyBest best;
    best.t = FLT_MAX;
    best.key = ~0ull;
    best.slot = -1;
    bool any = false, overflow = false;
    unsigned rounds = 0;
    while (top > 0) {
        if (++rounds > sc.round_cap) {       // (uniform) every round pops at least one entry, a tree has < round_cap of them
            if (tid == 0) atomicOr(sc.err, 2u);
            break;
        }
        const unsigned n = min(top, kHeavyThreads);
        const bool have = tid < n;
        int ref = 0;
        unsigned long long key = 0ull;
        unsigned depth = 0;
        if (have) {
            ref = q_ref[top - n + tid];
            key = q_key[top - n + tid];
            depth = q_depth[top - n + tid];
        }
        top -= n;
        bool pF = false, pS = false;
        int first = 0, second = 0;
        if (have && ref >= 0) {
            const float4 *np = sc.nodes + 4 * (size_t)ref;
            const float4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
            const int left = __float_as_int(q3.x), right = __float_as_int(q3.y);
            const unsigned meta = __float_as_uint(q3.z);
            const bool hL = box_hit(r, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y);
            const bool hR = box_hit(r, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w);
            const bool swap = !(meta & kMetaOrdered) && ((r.neg >> (meta & 3u)) & 1u);
            first = swap ? right : left;
            second = swap ? left : right;
            pF = swap ? hR : hL;
            pS = swap ? hL : hR;
        } else if (have) {
            const unsigned code = (unsigned)(~ref);
            const unsigned f0 = code >> kLeafBits, cnt = (code & (kMaxLeaf - 1)) + 1;
            for (unsigned i = 0; i < cnt; ++i) {
                const float4 *tp = sc.tris + 3 * (size_t)(f0 + i);
                const float4 a = tp[0], b = tp[1], c = tp[2];
                TriHit th;
                if (tri_test(r, a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z, th)) {
                    any = true;
                    const unsigned long long k = key | (unsigned long long)i;
                    if (th.t < best.t || (th.t == best.t && k < best.key)) {
                        best.t = th.t;
                        best.key = k;
                        best.slot = (int)(f0 + i);
                    }
                }
            }
        }
        // push the accepted children (prefix counts per wave, then over the waves); the level's bit sits below the bits of
        // the levels above.  The stack is popped in any order: the keys decide, not the order of the stack.
        const unsigned long long mF = __ballot(pF), mS = __ballot(pS);
        const unsigned long long lt = (1ull << lane) - 1ull;
        if (lane == 0) s_cnt[wave] = (unsigned)__popcll(mF) + (unsigned)__popcll(mS);
        if (have && ref >= 0 && depth >= 56u) s_flag[0] = 1u;      // deeper than the keys hold
        if (STOP_AT_ANY && any) s_flag[1] = 1u;
        __syncthreads();                                            // (also: every read of this round's entries is done)
        unsigned base = top, add = 0;
#pragma unroll
        for (unsigned w = 0; w < NW; ++w) {
            const unsigned c = s_cnt[w];
            if (w < wave) base += c;
            add += c;
        }
        if (top + add > sc.heavy_stack || s_flag[0]) {
            overflow = true;   // more open boxes than the stack holds (or deeper than the keys): one thread walks it alone
            break;
        }
        if (STOP_AT_ANY && s_flag[1]) break;
        const unsigned pos = base + (unsigned)__popcll(mF & lt) + (unsigned)__popcll(mS & lt);
        const unsigned long long bit = 1ull << (62u - depth);
        if (pF) {
            q_ref[pos] = first;
            q_key[pos] = key;
            q_depth[pos] = (unsigned char)(depth + 1u);
        }
        if (pS) {
            const unsigned p2 = pos + (pF ? 1u : 0u);
            q_ref[p2] = second;
            q_key[p2] = key | bit;
            q_depth[p2] = (unsigned char)(depth + 1u);
        }
        top += add;
        __syncthreads();
    }
    __syncthreads();
    if (overflow) {
        // the reference's loop as it stands (pspRT.cpp:380-423), one thread, its stack in q_ref
        best.t = FLT_MAX;
        best.key = ~0ull;
        best.slot = -1;
        any = false;
        if (tid == 0) {
            int sp = 0;
            int cur = sc.root_ref2;
            for (unsigned visits = 0;; ++visits) {
                if (visits > sc.round_cap) {
                    atomicOr(sc.err, 4u);
                    break;
                }
                if (cur >= 0) {
                    const float4 *np = sc.nodes + 4 * (size_t)cur;
                    const float4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
                    const int left = __float_as_int(q3.x), right = __float_as_int(q3.y);
                    const unsigned meta = __float_as_uint(q3.z);
                    const bool hL = box_hit(r, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y);
                    const bool hR = box_hit(r, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w);
                    const bool swap = !(meta & kMetaOrdered) && ((r.neg >> (meta & 3u)) & 1u);
                    const int first = swap ? right : left, second = swap ? left : right;
                    const bool hF = swap ? hR : hL, hS = swap ? hL : hR;
                    if (hF) {
                        cur = first;
                        if (hS && sp < (int)kHeavyStack) q_ref[sp++] = second;
                        continue;
                    }
                    if (hS) {
                        cur = second;
                        continue;
                    }
                } else {
                    const unsigned code = (unsigned)(~cur);
                    const unsigned f0 = code >> kLeafBits, cnt = (code & (kMaxLeaf - 1)) + 1;
                    for (unsigned i = 0; i < cnt; ++i) {
                        const float4 *tp = sc.tris + 3 * (size_t)(f0 + i);
                        const float4 a = tp[0], b = tp[1], c = tp[2];
                        TriHit th;
                        if (tri_test(r, a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z, th)) {
                            any = true;
                            if (th.t < best.t) {
                                best.t = th.t;
                                best.key = 0ull;
                                best.slot = (int)(f0 + i);
                            }
                        }
                    }
                }
                if (sp == 0) break;
                cur = q_ref[--sp];
            }
        }
    }
    // winner of each wave -- smallest t, then smallest key -- then of the workgroup
    float bt = best.t;
    for (int off = 32; off > 0; off >>= 1) bt = fminf(bt, __shfl_xor(bt, off));
    unsigned long long bk = best.t == bt ? best.key : ~0ull;
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(bk, off);
        bk = o < bk ? o : bk;
    }
    const bool any_wave = __ballot(any) != 0ull;
    const unsigned long long win = __ballot(any && best.t == bt && best.key == bk);
    const int wl = win ? __ffsll((long long)win) - 1 : 0;
    const int slot_wave = __shfl(best.slot, wl);
    if (lane == 0) {
        s_bt[wave] = any_wave ? bt : FLT_MAX;
        s_bk[wave] = any_wave ? bk : ~0ull;
        s_bs[wave] = any_wave ? slot_wave : -1;
        s_any[wave] = any_wave ? 1 : 0;
    }
    __syncthreads();
    any_w = false;
    best_slot = -1;
    float wt = FLT_MAX;
    unsigned long long wk = ~0ull;
#pragma unroll
    for (unsigned w = 0; w < NW; ++w) {
        if (!s_any[w]) continue;
        if (!any_w || s_bt[w] < wt || (s_bt[w] == wt && s_bk[w] < wk)) {
            wt = s_bt[w];
            wk = s_bk[w];
            best_slot = s_bs[w];
        }
        any_w = true;
    }
    __syncthreads();       // (the per-wave slots are free for the next ray)
}

template <int PHASE>
__global__ void __launch_bounds__(kHeavyThreads)
    heavy_kernel(Scene sc, Cam cam, const float *__restrict__ nodes, const int32_t *__restrict__ tri_nodes,
                 int32_t *__restrict__ pix, const unsigned *__restrict__ retry_nodes,
                 unsigned *__restrict__ retry_mask, const unsigned *__restrict__ work, int count_word,
                 const unsigned *__restrict__ items)
{
    unsigned char *scratch = sc.heavy_scratch + (size_t)blockIdx.x * kHeavyScratchBytes;
    unsigned long long *q_key = reinterpret_cast<unsigned long long *>(scratch);
    int *q_ref = reinterpret_cast<int *>(scratch + (size_t)kHeavyStack * 8);
    unsigned char *q_depth = scratch + (size_t)kHeavyStack * 12;
    const unsigned lane = threadIdx.x;
    const unsigned count = min(work[count_word], kHeavyCap);
    for (unsigned h = blockIdx.x; h < count; h += gridDim.x) {
        const unsigned item = items[h];
        const unsigned node = PHASE == 0 ? item : retry_nodes[item / 6u];
        Ray r;
        if (PHASE == 0) {
            float dx = nodes[3 * (size_t)item] - cam.ox, dy = nodes[3 * (size_t)item + 1] - cam.oy,
                  dz = nodes[3 * (size_t)item + 2] - cam.oz;
            const float len = imath_length(dx, dy, dz);
            if (len != 0.0f) { dx /= len; dy /= len; dz /= len; }
            ray_setup(r, cam.ox, cam.oy, cam.oz, dx, dy, dz);
        } else {
            retry_ray(r, cam, nodes, node, (int)(item % 6u));
        }
        ray_classify(r, sc);
        bool any_w;
        int best_slot;
        heavy_walk<false>(r, sc, q_ref, q_key, q_depth, any_w, best_slot);
        if (lane == 0) {
            {
                bool visible = false;
                if (any_w && best_slot >= 0) {
                    const int prim = __float_as_int(sc.tris[3 * (size_t)best_slot].w);
                    visible = tri_has_node(tri_nodes, prim, (int)node);
                }
                if (PHASE == 0) {
                    pix[node] = visible ? kPixVisible : (any_w ? kPixRetry : kPixNone);
                    if (sc.witness && !visible && any_w) sc.witness[node] = best_slot;
                } else if (visible) {
                    atomicOr(&retry_mask[item / 6u], 1u << (item % 6u));
                }
            }
        }
        __syncthreads();
    }
}

// The same walk for the rays a batch query gave up (cast_kernel): full hit record, or the occlusion flag.
template <bool ANYHIT>
__global__ void __launch_bounds__(kHeavyThreads)
    heavy_cast_kernel(Scene sc, const float *__restrict__ org, int org_stride, const float *__restrict__ dir,
                      upsp_hits out, const unsigned *__restrict__ work, int count_word, const unsigned *__restrict__ items)
{
    unsigned char *scratch = sc.heavy_scratch + (size_t)blockIdx.x * kHeavyScratchBytes;
    unsigned long long *q_key = reinterpret_cast<unsigned long long *>(scratch);
    int *q_ref = reinterpret_cast<int *>(scratch + (size_t)kHeavyStack * 8);
    unsigned char *q_depth = scratch + (size_t)kHeavyStack * 12;
    const unsigned count = min(work[count_word], kHeavyCap);
    for (unsigned h = blockIdx.x; h < count; h += gridDim.x) {
        const unsigned item = items[h];
        const float *o = org + (size_t)org_stride * item;
        const float *d = dir + 3 * (size_t)item;
        Ray r;
        ray_setup(r, o[0], o[1], o[2], d[0], d[1], d[2]);
        ray_classify(r, sc);
        bool any_w;
        int best_slot;
        heavy_walk<ANYHIT>(r, sc, q_ref, q_key, q_depth, any_w, best_slot);
        if (threadIdx.x == 0) {
            if (ANYHIT) {
                if (out.hit) out.hit[item] = any_w ? 1 : 0;
            } else {
                Trav s;
                s.any = any_w;
                s.best_slot = any_w ? best_slot : -1;
                write_hit(r, sc, s, item, out);
            }
        }
        __syncthreads();
    }
}

// ---- one ray per WAVE ----------------------------------------------------------------------------------------------------
// A batch ends when its longest rays end: on the frame-filling sphere 1 % of the pixel rays need more than 64 node visits +
// triangle tests, a few thousand more than 128, the longest ~300 -- each a chain of dependent fetches in ONE lane of a wave whose
// other lanes have long finished.  Such a ray leaves the one-lane traversal at `heavy_steps` and a whole wave walks it here: the
// frontier of open boxes lives on a stack in LDS (512 entries per wave), every lane pops one entry per round -- an interior node
// of the BINARY tree: both child boxes tested, the accepted ones pushed with their depth-first keys; a leaf: its triangles tested --
// so a ray of 300 sequential steps is ~25 rounds.  The keys make the result independent of the order of the stack (heavy_walk):
// closest t, then the smallest key = the first triangle the reference's near-first depth-first walk would have met.  No
// __syncthreads: the four waves of a workgroup walk four different rays.  A frontier that outgrows the stack (or a tree deeper
// than the keys) goes on to heavy_cast_kernel's list (one workgroup per ray, 4096 entries in global memory, one-thread fallback).
constexpr unsigned kWaveStack = 512;
template <bool STOP_AT_ANY>
__device__ __forceinline__ bool wave_walk(const Ray &r, const Scene &sc, int *q_ref, unsigned long long *q_key, unsigned char *q_depth,
                                          bool &any_w, int &best_slot)
{
    const unsigned lane = threadIdx.x & 63u;
    unsigned top = 0;                         // (uniform)
    if (box_hit(r, sc.rlo[0], sc.rlo[1], sc.rlo[2], sc.rhi[0], sc.rhi[1], sc.rhi[2])) {
        if (lane == 0) {
            q_ref[0] = sc.root_ref2;
            q_key[0] = 0ull;
            q_depth[0] = 0;
        }
        top = 1;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    HeavyBest best;
    best.t = FLT_MAX;
    best.key = ~0ull;
    best.slot = -1;
    bool any = false, overflow = false;
    unsigned rounds = 0;
    while (top > 0) {
        if (++rounds > sc.round_cap) {
            if (lane == 0) atomicOr(sc.err, 2u);
            break;
        }
        const unsigned n = min(top, 64u);
        const bool have = lane < n;
        int ref = 0;
        unsigned long long key = 0ull;
        unsigned depth = 0;
        if (have) {
            ref = q_ref[top - n + lane];
            key = q_key[top - n + lane];
            depth = q_depth[top - n + lane];
        }
        top -= n;
        bool pF = false, pS = false;
        int first = 0, second = 0;
        if (have && ref >= 0) {
            const float4 *np = sc.nodes + 4 * (size_t)ref;
            const float4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
            const int left = __float_as_int(q3.x), right = __float_as_int(q3.y);
            const unsigned meta = __float_as_uint(q3.z);
            const bool hL = box_hit(r, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y);
            const bool hR = box_hit(r, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w);
            const bool swap = !(meta & kMetaOrdered) && ((r.neg >> (meta & 3u)) & 1u);
            first = swap ? right : left;
            second = swap ? left : right;
            pF = swap ? hR : hL;
            pS = swap ? hL : hR;
        } else if (have) {
            const unsigned code = (unsigned)(~ref);
            const unsigned f0 = code >> kLeafBits, cnt = (code & (kMaxLeaf - 1)) + 1;
            for (unsigned i = 0; i < cnt; ++i) {
                const float4 *tp = sc.tris + 3 * (size_t)(f0 + i);
                const float4 a = tp[0], b = tp[1], c = tp[2];
                TriHit th;
                if (tri_test(r, a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z, th)) {
                    any = true;
                    const unsigned long long k = key | (unsigned long long)i;
                    if (th.t < best.t || (th.t == best.t && k < best.key)) {
                        best.t = th.t;
                        best.key = k;
                        best.slot = (int)(f0 + i);
                    }
                }
            }
        }
        const unsigned long long mF = __ballot(pF), mS = __ballot(pS);
        const unsigned add = (unsigned)__popcll(mF) + (unsigned)__popcll(mS);
        // (sc.heavy_stack: the tests shrink it to push rays on to the workgroup walk and its one-thread fallback)
        if (top + add > min(kWaveStack, sc.heavy_stack) || __ballot(have && ref >= 0 && depth >= 56u) != 0ull) {
            overflow = true;
            break;
        }
        if (STOP_AT_ANY && __ballot(any) != 0ull) break;
        // (every read of this round's entries is done: the loads above were consumed by the ballots' operands)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const unsigned long long lt = (1ull << lane) - 1ull;
        const unsigned pos = top + (unsigned)__popcll(mF & lt) + (unsigned)__popcll(mS & lt);
        const unsigned long long bit = 1ull << (62u - depth);
        if (pF) {
            q_ref[pos] = first;
            q_key[pos] = key;
            q_depth[pos] = (unsigned char)(depth + 1u);
        }
        if (pS) {
            const unsigned p2 = pos + (pF ? 1u : 0u);
            q_ref[p2] = second;
            q_key[p2] = key | bit;
            q_depth[p2] = (unsigned char)(depth + 1u);
        }
        top += add;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (overflow) return false;
    // the wave's winner: smallest t, then smallest key
    float bt = best.t;
    for (int off = 32; off > 0; off >>= 1) bt = fminf(bt, __shfl_xor(bt, off));
    unsigned long long bk = best.t == bt ? best.key : ~0ull;
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(bk, off);
        bk = o < bk ? o : bk;
    }
    any_w = __ballot(any) != 0ull;
    const unsigned long long win = __ballot(any && best.t == bt && best.key == bk);
    const int wl = win ? __ffsll((long long)win) - 1 : 0;
    best_slot = any_w ? __shfl(best.slot, wl) : -1;
    return true;
}

template <bool ANYHIT>
__global__ void __launch_bounds__(256)
    wave_cast_kernel(Scene sc, const float *__restrict__ org, int org_stride, const float *__restrict__ dir, upsp_hits out,
                     unsigned *__restrict__ work, const unsigned *__restrict__ items, unsigned *__restrict__ items_over)
{
    __shared__ unsigned long long s_key[4][kWaveStack];
    __shared__ int s_ref[4][kWaveStack];
    __shared__ unsigned char s_depth[4][kWaveStack];
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const unsigned count = min(work[kWorkHeavyCast], kHeavyCapCast);
    for (unsigned h = blockIdx.x * 4u + wave; h < count; h += gridDim.x * 4u) {
        const unsigned item = items[h];
        const float *o = org + (size_t)org_stride * item;
        const float *d = dir + 3 * (size_t)item;
        Ray r;
        ray_setup(r, o[0], o[1], o[2], d[0], d[1], d[2]);
        ray_classify(r, sc);
        bool any_w = false;
        int best_slot = -1;
        const bool done = wave_walk<ANYHIT>(r, sc, s_ref[wave], s_key[wave], s_depth[wave], any_w, best_slot);
        if (lane == 0) {
            if (!done) {
                const unsigned slot = atomicAdd(&work[kWorkHeavyCast + 1], 1u);
                items_over[slot] = item;               // (at most `count` <= kHeavyCapCast of them)
            } else if (ANYHIT) {
                if (out.hit) out.hit[item] = any_w ? 1 : 0;
            } else {
                Trav s;
                s.any = any_w;
                s.best_slot = any_w ? best_slot : -1;
                write_hit(r, sc, s, item, out);
            }
        }
    }
}

// The same for the projection build's visibility rays (work items of the primary pass, PHASE 0, or of the retry passes): the
// verdicts heavy_kernel writes, one ray per wave; a frontier that outgrows the LDS stack goes on to heavy_kernel's list.
constexpr int kWorkWaveOver = 19;     // [19] / [20]: rays wave_proj_kernel passed on (primary / retry phase)
// WPB waves per workgroup.  The projection build launches it with ONE (6.6 KB of LDS): the build runs on a stream of its own
// beside pass A / pass B, its hand-off lists are empty on well-shaped models, and 1280 four-wave workgroups with 26 KB of LDS each
// had to find their place among the passes' workgroups just to read a zero (kernel trace: 146 us beside pass A, 9 us in a gap).
template <int PHASE, int WPB>
__global__ void __launch_bounds__(64 * WPB)
    wave_proj_kernel(Scene sc, Cam cam, const float *__restrict__ nodes, const int32_t *__restrict__ tri_nodes,
                     int32_t *__restrict__ pix, const unsigned *__restrict__ retry_nodes, unsigned *__restrict__ retry_mask,
                     unsigned *__restrict__ work, const unsigned *__restrict__ items, unsigned *__restrict__ items_over)
{
    __shared__ unsigned long long s_key[WPB][kWaveStack];
    __shared__ int s_ref[WPB][kWaveStack];
    __shared__ unsigned char s_depth[WPB][kWaveStack];
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const unsigned count = min(work[kWorkHeavyCount + (PHASE ? 1 : 0)], kHeavyCap);
    for (unsigned h = blockIdx.x * (unsigned)WPB + wave; h < count; h += gridDim.x * (unsigned)WPB) {
        const unsigned item = items[h];
        const unsigned node = PHASE == 0 ? item : retry_nodes[item / 6u];
        Ray r;
        if (PHASE == 0) {
            float dx = nodes[3 * (size_t)item] - cam.ox, dy = nodes[3 * (size_t)item + 1] - cam.oy,
                  dz = nodes[3 * (size_t)item + 2] - cam.oz;
            const float len = imath_length(dx, dy, dz);
            if (len != 0.0f) { dx /= len; dy /= len; dz /= len; }
            ray_setup(r, cam.ox, cam.oy, cam.oz, dx, dy, dz);
        } else {
            retry_ray(r, cam, nodes, node, (int)(item % 6u));
        }
        ray_classify(r, sc);
        bool any_w = false;
        int best_slot = -1;
        const bool done = wave_walk<false>(r, sc, s_ref[wave], s_key[wave], s_depth[wave], any_w, best_slot);
        if (lane == 0) {
            if (!done) {
                const unsigned slot = atomicAdd(&work[kWorkWaveOver + (PHASE ? 1 : 0)], 1u);
                items_over[slot] = item;
            } else {
                bool visible = false;
                if (any_w && best_slot >= 0) {
                    const int prim = __float_as_int(sc.tris[3 * (size_t)best_slot].w);
                    visible = tri_has_node(tri_nodes, prim, (int)node);
                }
                if (PHASE == 0) {
                    pix[node] = visible ? kPixVisible : (any_w ? kPixRetry : kPixNone);
                    if (sc.witness && !visible && any_w) sc.witness[node] = best_slot;
                } else if (visible) {
                    atomicOr(&retry_mask[item / 6u], 1u << (item % 6u));
                }
            }
        }
    }
}

// Step 2b (elementwise): compact the nodes that need retries into a list.  One queue atomic per
// workgroup of kRetryListItems x 256 nodes: atomics on one address retire at ~88 per us on this
// part, so one per 256 nodes (1957 of them) made this a 24-us kernel.
constexpr int kRetryListItems = 8;
// (STATE = kPixRetry: the retry list + its cleared bit masks, count in work[kWorkRetryCount];
//  STATE = kPixInFrame: the nodes that cast a primary ray, count in work[kWorkPrimaryCount], no masks)
template <int32_t STATE>
__global__ void __launch_bounds__(256)
    retry_list_kernel(const int32_t *__restrict__ pix, unsigned nnodes,
                      unsigned *__restrict__ retry_nodes, unsigned *__restrict__ retry_mask,
                      unsigned *work)
{
    __shared__ unsigned wave_cnt[kRetryListItems][4], block_base;
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned base = blockIdx.x * (256u * kRetryListItems) + threadIdx.x;
    // the queue head for the retry passes (the primary pass used it; no launch of its own for one word)
    if (STATE == kPixRetry && blockIdx.x == 0 && threadIdx.x == 0) work[0] = 0u;
    unsigned long long m[kRetryListItems];
    bool need[kRetryListItems];
#pragma unroll
    for (int k = 0; k < kRetryListItems; ++k) {
        const unsigned n = base + 256u * k;
        need[k] = n < nnodes && pix[n] == STATE;
        m[k] = __ballot(need[k]);
        if (lane == 0) wave_cnt[k][wave] = (unsigned)__popcll(m[k]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned tot = 0;
        for (int k = 0; k < kRetryListItems; ++k)
            for (int w = 0; w < 4; ++w) {   // exclusive prefix in place
                const unsigned c = wave_cnt[k][w];
                wave_cnt[k][w] = tot;
                tot += c;
            }
        block_base = tot ? atomicAdd(&work[STATE == kPixRetry ? kWorkRetryCount : kWorkPrimaryCount], tot) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kRetryListItems; ++k)
        if (need[k]) {
            const unsigned slot = block_base + wave_cnt[k][wave] + (unsigned)__popcll(m[k] & ((1ull << lane) - 1ull));
            retry_nodes[slot] = base + 256u * k;
            if (STATE == kPixRetry) retry_mask[slot] = 0u;
        }
}

// The dense list of the primary pass in kRayBins bins by the step count of the build before (steps_prev, 0 = unknown -> bin 0);
// edges[j] = largest step count of bin j (j < kRayBins - 1).  One atomic per workgroup and non-empty bin.
__global__ void __launch_bounds__(256)
    primary_list_binned_kernel(const int32_t *__restrict__ pix, unsigned nnodes, const unsigned short *__restrict__ steps_prev,
                               const unsigned *__restrict__ edges, unsigned *__restrict__ lists, unsigned stride, unsigned *work)
{
    __shared__ unsigned wave_cnt[kRayBins][kRetryListItems][4], bin_base[kRayBins];
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned base = blockIdx.x * (256u * kRetryListItems) + threadIdx.x;
    unsigned e[kRayBins - 1];
#pragma unroll
    for (unsigned j = 0; j + 1 < kRayBins; ++j) e[j] = edges[j];
    bool need[kRetryListItems];
    unsigned bin[kRetryListItems], rank[kRetryListItems];
#pragma unroll
    for (int k = 0; k < kRetryListItems; ++k) {
        const unsigned n = base + 256u * k;
        need[k] = n < nnodes && pix[n] == kPixInFrame;
        const unsigned st = need[k] ? (unsigned)steps_prev[n] : 0u;
        unsigned b = 0;
#pragma unroll
        for (unsigned j = 0; j + 1 < kRayBins; ++j) b += st > e[j] ? 1u : 0u;
        bin[k] = b;
        rank[k] = 0;
#pragma unroll
        for (unsigned bb = 0; bb < kRayBins; ++bb) {
            const unsigned long long m = __ballot(need[k] && b == bb);
            if (need[k] && b == bb) rank[k] = (unsigned)__popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) wave_cnt[bb][k][wave] = (unsigned)__popcll(m);
        }
    }
    __syncthreads();
    if (threadIdx.x < kRayBins) {
        const unsigned bb = threadIdx.x;
        unsigned tot = 0;
        for (int k = 0; k < kRetryListItems; ++k)
            for (int w = 0; w < 4; ++w) {   // exclusive prefix in place
                const unsigned c = wave_cnt[bb][k][w];
                wave_cnt[bb][k][w] = tot;
                tot += c;
            }
        bin_base[bb] = tot ? atomicAdd(&work[kWorkBinCount + bb], tot) : 0u;
        if (tot) atomicAdd(&work[kWorkPrimaryCount], tot);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kRetryListItems; ++k)
        if (need[k]) lists[(size_t)bin[k] * stride + bin_base[bin[k]] + wave_cnt[bin[k]][k][wave] + rank[k]] = base + 256u * k;
}

// Bin edges for the NEXT build from a sample of this build's step counts (<= 16 384 nodes, every stride-th): quantiles k / kRayBins
// of the nodes that cast a ray.  One workgroup; edges[kRayBins - 1] = 1 marks them valid.
__global__ void __launch_bounds__(256)
    step_edges_kernel(const unsigned short *__restrict__ steps, unsigned nnodes, unsigned *__restrict__ edges)
{
    constexpr unsigned kBuckets = 256;
    __shared__ unsigned hist[kBuckets];
    hist[threadIdx.x] = 0u;
    __syncthreads();
    const unsigned stride = max(1u, nnodes / 16384u);
    for (unsigned n = threadIdx.x * stride; n < nnodes; n += 256u * stride) {
        const unsigned st = steps[n];
        if (st) atomicAdd(&hist[min(st, kBuckets - 1u)], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned total = 0;
        for (unsigned i = 0; i < kBuckets; ++i) total += hist[i];
        unsigned acc = 0, j = 0;
        for (unsigned i = 0; i < kBuckets && j + 1 < kRayBins; ++i) {
            acc += hist[i];
            while (j + 1 < kRayBins && (unsigned long long)acc * kRayBins >= (unsigned long long)total * (j + 1)) edges[j++] = i;
        }
        while (j + 1 < kRayBins) edges[j++] = kBuckets;
        edges[kRayBins - 1] = 1u;
    }
}

// Step 4: retry outcome per listed node + the reference's ray count
// (1 + index of the first successful retry, or 6).
__global__ void projection_retry_outcome_kernel(int32_t *__restrict__ pix,
                                                const unsigned *__restrict__ retry_nodes,
                                                const unsigned *__restrict__ retry_mask,
                                                unsigned *work)
{
    const unsigned count = work[kWorkRetryCount];
    unsigned rays = 0;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        const unsigned mask = retry_mask[i];
        rays += mask ? (unsigned)__ffs((int)mask) : 6u;
        pix[retry_nodes[i]] = mask ? kPixVisible : kPixNone;
    }
    __shared__ int red[16];
    flush_stats(work, red, 0u, 0u, rays);
}

// Step 5 (elementwise): oblique test with the PRIMARY direction, then the matrix entry
// (psp_process.cpp:298-322) for every node a ray has seen.
__global__ void __launch_bounds__(256)
    projection_finish_kernel(Cam cam, const float *__restrict__ nodes,
                             const float *__restrict__ normals, unsigned nnodes,
                             float oblique_thresh, int32_t *__restrict__ pix,
                             float *__restrict__ uv)
{
    const unsigned n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nnodes) return;
    const int32_t state = pix[n];
    int32_t out = kPixNone;
    float ou = 0.f, ov = 0.f;
    if (state == kPixVisible) {
        const float u = uv[2 * (size_t)n], v = uv[2 * (size_t)n + 1];
        if (oblique_forward(cam, nodes, normals, n, oblique_thresh)) {
            const int px_ = (int)roundf(u), py_ = (int)roundf(v);  // :319 std::round
            const long long idx = (long long)py_ * cam.W + px_;
            if (idx >= 0 && idx < (long long)cam.W * cam.H) {
                out = (int32_t)idx;
                ou = u / (float)cam.W;  // :311-314
                ov = v / (float)cam.H;
            }
        }
    }
    pix[n] = out;
    uv[2 * (size_t)n] = ou;
    uv[2 * (size_t)n + 1] = ov;
}

// Streams the BVH (nodes + triangle records) through the cache hierarchy once so
// that the pointer-chasing traversal that follows finds it in the 256 MiB Infinity
// Cache instead of paying an HBM miss per step (the frame loop evicts it between
// projection builds).  16 bytes per lane, fully coalesced: ~15 us for 70 MB.
__global__ void __launch_bounds__(256)
    bvh_prefetch_kernel(const uint4 *__restrict__ a, size_t na, const uint4 *__restrict__ b,
                        size_t nb, unsigned *sink)
{
    unsigned acc = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < na; i += stride) {
        const uint4 v = a[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += stride) {
        const uint4 v = b[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x9e3779b9u) sink[15] = acc;  // keeps the loads alive, practically never taken
}

__global__ void nodecount_kernel(const int32_t *__restrict__ pix, unsigned nnodes,
                                 unsigned *__restrict__ counts)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nnodes && pix[i] >= 0) atomicAdd(&counts[pix[i]], 1u);
}
__global__ void saturate_kernel(const unsigned *__restrict__ counts, unsigned n,
                                uint8_t *__restrict__ out)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (uint8_t)min(counts[i], 255u);  // psp_process.cpp:335-347
}

// ------------------------------------------------------------------------
//  host side
// ------------------------------------------------------------------------
int env_int(const char *name, int dflt)
{
    const char *v = std::getenv(name);
    return v ? std::atoi(v) : dflt;
}

int stack_entries(const upsp_bvh *b)
{
    // LDS stack depth per thread: a wide step leaves at most three entries behind and descends one wide level (two of the
    // binary tree's), a leaf none: 3 x wide levels + 1, rounded up to a multiple of 8, >= 8
    int d = 3 * (int)b->wide_depth + 1;
    d = ((d + 7) / 8) * 8;
    return d < 8 ? 8 : d;
}

size_t lds_bytes(const upsp_bvh *b)
{
    return (size_t)stack_entries(b) * kBlock * sizeof(int);
}

Scene make_scene(const upsp_bvh *b, size_t items, int grid)
{
    Scene sc;
    // chunk: ~1/2 of a wave's fair share, multiple of 64, in [64, kChunkMax]
    size_t share = items / ((size_t)grid * (kBlock / 64) * 2 + 1);
    share = (share / 64) * 64;
    sc.chunk = (unsigned)std::min<size_t>(std::max<size_t>(share, 64), kChunkMax);
    sc.nodes = reinterpret_cast<const float4 *>(b->d_nodes);
    sc.wide = reinterpret_cast<const float4 *>(b->d_wide);
    sc.tris = reinterpret_cast<const float4 *>(b->d_tris);
    sc.root_ref = b->wide_root;
    sc.root_ref2 = b->root_ref;
    sc.refill = kRefillDefault;
    sc.desc_cap = 6;        // (measured 1 / 2 / 6 / 8 rounds: primary 0.354 / 0.295 / 0.262 / 0.269 ms in round 2; on the wide records of
                            //  round 4: 1 / 2 / 3 / 4 / 6 / 8 -> 0.161 / 0.142 / 0.141 / 0.134 / 0.142 / 0.150 ms, within the run-to-run spread)
    // (measured: primary traversal 267-271 -> 263 us, batch queries alike; residual retries +2 %: their list is not
    //  in mesh order, so they keep the plain mapping)
    sc.xcd = 1;
    sc.pack_waves = 0;
    sc.heavy_steps = 0;
    sc.heavy_items = nullptr;
    sc.heavy_scratch = nullptr;
    sc.heavy_stack = 0;
    sc.adj_off = sc.adj_slot = nullptr;
    sc.slot_path = nullptr;
    sc.path_ref = nullptr;
    sc.witness = nullptr;
    sc.hist = nullptr;
    // (read per call: the tests switch the slab filter off and widen its band in one process)
    sc.slab = env_int("UPSP_SLAB_FILTER", 1) != 0;
    sc.slab_scale = (float)std::max(1, env_int("UPSP_SLAB_SCALE", 1));      // test switch: a wider band = more boxes through the fallback
    sc.steps_out = nullptr;
    sc.nbins = 0;
    sc.bin_stride = 0;
    // (UPSP_ROUND_CAP: the tests set it low to see the error come back)
    const int cap_env = env_int("UPSP_ROUND_CAP", 0);
    sc.round_cap = cap_env > 0 ? (unsigned)cap_env : 2u * (unsigned)b->info.n_gpu_nodes + 130u;
    sc.err = b->d_err;
    for (int a = 0; a < 3; ++a) {
        sc.rlo[a] = b->root_min[a];
        sc.rhi[a] = b->root_max[a];
    }
    return sc;
}

struct DeviceProps {
    int cus = 0;
    bool ok = false;
};
DeviceProps &props()
{
    static DeviceProps p;
    if (!p.ok) {
        hipDeviceProp_t dp;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&dp, dev) == hipSuccess) {
            p.cus = dp.multiProcessorCount;
            p.ok = true;
        }
    }
    return p;
}

// persistent grid: enough workgroups to fill the chip at the LDS-limited occupancy
int grid_for(size_t items, size_t lds_bytes)
{
    const int cus = props().cus > 0 ? props().cus : 256;
    int per_cu = (int)((160u * 1024u) / (lds_bytes ? lds_bytes : 1));
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    size_t want = (items + kBlock - 1) / kBlock;
    size_t cap = (size_t)cus * (size_t)per_cu;
    size_t g = want < cap ? want : cap;
    return (int)(g ? g : 1);
}

void prefetch_bvh(upsp_bvh *b, hipStream_t st)
{
    KTimed kt("bvh_prefetch_kernel", st);
    const size_t na = (size_t)b->n_wide * 8, nb = (size_t)b->info.ntris * 3;
    hipLaunchKernelGGL(bvh_prefetch_kernel, dim3(2048), dim3(256), 0, st,
                       reinterpret_cast<const uint4 *>(b->d_wide), na,
                       reinterpret_cast<const uint4 *>(b->d_tris), nb, b->d_work);
}

// The walks' error word (Scene::err), read at a point where the host synchronises with `st` anyway.
int check_walk_error(upsp_bvh *b, hipStream_t st)
{
    unsigned e = 0;
    UPSP_HIP_CHECK(hipMemcpyAsync(&e, b->d_err, sizeof(e), hipMemcpyDeviceToHost, st));
    UPSP_HIP_CHECK(hipStreamSynchronize(st));
    if (!e) return UPSP_OK;
    UPSP_HIP_CHECK(hipMemsetAsync(b->d_err, 0, sizeof(unsigned), st));
    char msg[160];
    std::snprintf(msg, sizeof(msg), "BVH walk exceeded its round cap (flags 0x%x: 1 traversal, 2 cooperative walk, 4 its "
                  "fallback): broken tree -- results of that call are invalid", e);
    return fail(UPSP_ERR_INTERNAL, msg);
}

int read_stats(upsp_bvh *b, hipStream_t st)
{
    unsigned long long h[3];
    unsigned mx[2];
    UPSP_HIP_CHECK(hipMemcpyAsync(h, b->d_work + 2, sizeof(h), hipMemcpyDeviceToHost, st));
    UPSP_HIP_CHECK(hipMemcpyAsync(mx, b->d_work + 8, sizeof(mx), hipMemcpyDeviceToHost, st));
    {
        const int rc = check_walk_error(b, st);     // (synchronises st)
        if (rc != UPSP_OK) return rc;
    }
    if (std::getenv("UPSP_TRACE_STATS")) {
        unsigned wr[2] = {0, 0};
        UPSP_HIP_CHECK(hipMemcpy(wr, b->d_work + 14, sizeof(wr), hipMemcpyDeviceToHost));
        std::fprintf(stderr, "[upsp] longest ray: %u node steps, %u triangle tests; lanes active per wave round: node steps "
                     "%.1f of 64 (%llu lane steps / %u rounds), triangle tests %.1f of 64 (%llu / %u)\n", mx[0], mx[1],
                     wr[0] ? (double)h[0] / wr[0] : 0.0, h[0], wr[0], wr[1] ? (double)h[1] / wr[1] : 0.0, h[1], wr[1]);
    }
    b->last_stats[0] = h[0];
    b->last_stats[1] = h[1];
    b->last_stats[2] = h[2];
    {
        unsigned sl[2] = {0, 0};
        UPSP_HIP_CHECK(hipMemcpy(sl, b->d_work + 28, sizeof(sl), hipMemcpyDeviceToHost));
        b->last_slab[0] = 4ull * sl[0];
        b->last_slab[1] = sl[1];
    }
    return UPSP_OK;
}

template <bool ANYHIT>
int launch_cast(const upsp_bvh *cb, const float *d_org, int org_stride, const float *d_dir,
                size_t n, const upsp_hits &out, hipStream_t st)
{
    upsp_bvh *b = const_cast<upsp_bvh *>(cb);
    if (!b) return fail(UPSP_ERR_INVALID, "null BVH");
    if (n == 0) return UPSP_OK;
    if (!d_org || !d_dir) return fail(UPSP_ERR_INVALID, "null ray buffers");
    if (org_stride != 0 && org_stride != 3) return fail(UPSP_ERR_INVALID, "org_stride must be 0 or 3");
    if (n > 0xF0000000ull) return fail(UPSP_ERR_INVALID, "too many rays in one call");
    const int entries = stack_entries(b);
    if (entries > 64) return fail(UPSP_ERR_DEPTH, "BVH deeper than 64 levels");
    const size_t lds = lds_bytes(b);
    UPSP_HIP_CHECK(hipMemsetAsync(b->d_work, 0, kWorkWords * sizeof(unsigned), st));
    const int grid = grid_for(n, lds);
    Scene sc = make_scene(b, n, grid);
    // rays that need more than this many node visits + triangle tests leave the one-lane traversal (heavy_cast_kernel).
    // Measured on the bench's 1 Mi pixel rays (tools/exp_heavy_cast.sh; us per call):
    //   UV-sphere tunnel model (1000-triangle polar fans, slivers): off 1307, 640: 940, 384: 703, 256: 519, 160: 397, 128: 352
    //   frame-filling cube sphere (42 % hits, many long grazing rays): off 564, 384: 582, 256: 583, 160: 662, 128: 679
    //   cube-sphere tunnel model: off 166, any threshold 174-180 (the empty launch)
    // -- above the projection's 160: a batch may hold any ray, and thousands that are merely long are cheaper where they are
    // Round 5: the rays that leave go to a WAVE each first (wave_cast_kernel: LDS stack, four rays per workgroup, thousands at
    // once) and only a frontier that outgrows that stack to the workgroup walk -- so the threshold can sit where the tail of the
    // batch starts: frame-filling sphere, 1 Mi rays: see LAB_NOTES.md section 12.
    const int heavy_steps = env_int("UPSP_HEAVY_STEPS_CAST", 96);   // (read per call: the tests move it)
    const bool heavy_on = heavy_steps > 0 && !b->stats_on && b->info.depth <= 56;
    if (heavy_on) {
        if (!b->d_heavy) UPSP_HIP_CHECK(hipMalloc(&b->d_heavy, sizeof(unsigned) * 2 * kHeavyCap));      // (two lists)
    if (!b->d_heavy_scratch) UPSP_HIP_CHECK(hipMalloc(&b->d_heavy_scratch, kHeavyScratchBytes * kHeavyGridMax));
        sc.heavy_steps = (unsigned)heavy_steps;
        sc.heavy_items = b->d_heavy;
        sc.heavy_scratch = reinterpret_cast<unsigned char *>(b->d_heavy_scratch);
        sc.heavy_stack = kHeavyStack;
    }
    auto launch_heavy = [&]() {
        if (!heavy_on) return;
        {
            KTimed ktw("wave_cast_kernel", st);
            const int cus = props().cus > 0 ? props().cus : 256;
            hipLaunchKernelGGL((wave_cast_kernel<ANYHIT>), dim3((unsigned)cus * 5u), dim3(256), 0, st, sc, d_org, org_stride, d_dir, out,
                               b->d_work, (const unsigned *)b->d_heavy, b->d_heavy + kHeavyCap);
        }
        KTimed kth("heavy_cast_kernel", st);
        hipLaunchKernelGGL((heavy_cast_kernel<ANYHIT>), dim3(512), dim3(kHeavyThreads), 0, st, sc, d_org, org_stride, d_dir, out,
                           (const unsigned *)b->d_work, kWorkHeavyCast + 1, (const unsigned *)(b->d_heavy + kHeavyCap));
    };
    // a sweep of the tree through the caches in front of the FIRST large batch of a handle; in front of every batch it cost 13 us
    // + two turnarounds of a 0.65-ms call and bought nothing (back-to-back 1 Mi-ray batches: the traversal 0.57-0.60 ms with it,
    // 0.53-0.58 without)
    if (n >= 65536 && !b->batch_warmed) {
        prefetch_bvh(b, st);
        b->batch_warmed = true;
    }
    if (n >= 65536 && !b->stats_on) {
        if (b->cast_list_capacity < n) {
            if (b->d_cast_list) (void)hipFree(b->d_cast_list);
            b->d_cast_list = nullptr;
            b->cast_list_capacity = 0;
            UPSP_HIP_CHECK(hipMalloc(&b->d_cast_list, sizeof(unsigned) * n));
            b->cast_list_capacity = n;
        }
        sc.pack_waves = kPackWavesPerSimd * 4u * (unsigned)(props().cus > 0 ? props().cus : 256);
        {
            KTimed kte("cast_entry_kernel", st);
            const dim3 egrid((unsigned)((n + 256 * kEntryItems - 1) / (256 * kEntryItems)));
            hipLaunchKernelGGL((cast_entry_kernel<ANYHIT>), egrid, dim3(256), 0, st, sc, d_org, org_stride, d_dir,
                               (unsigned)n, out, b->d_cast_list, b->d_work);
        }
        {
            KTimed kt(ANYHIT ? "cast_kernel<anyhit>" : "cast_kernel<closest>", st);
            hipLaunchKernelGGL((cast_kernel<ANYHIT, false, true>), dim3(grid), dim3(kBlock), lds, st, sc,
                               d_org, org_stride, d_dir, (unsigned)n, out, b->d_work, (const unsigned *)b->d_cast_list);
        }
        launch_heavy();
        UPSP_HIP_CHECK(hipGetLastError());
        if (heavy_on && std::getenv("UPSP_TRACE_STATS")) {       // diagnostics: how many rays left the one-lane traversal
            unsigned nh = 0;
            UPSP_HIP_CHECK(hipMemcpyAsync(&nh, b->d_work + kWorkHeavyCast, sizeof(nh), hipMemcpyDeviceToHost, st));
            UPSP_HIP_CHECK(hipStreamSynchronize(st));
            std::fprintf(stderr, "[upsp] %u of %zu rays needed more than %d steps (cooperative walk)\n", nh, n, heavy_steps);
        }
        return UPSP_OK;
    }
    {
        KTimed kt(ANYHIT ? "cast_kernel<anyhit>" : "cast_kernel<closest>", st);
        if (b->stats_on)
            hipLaunchKernelGGL((cast_kernel<ANYHIT, true>), dim3(grid), dim3(kBlock), lds, st, sc,
                               d_org, org_stride, d_dir, (unsigned)n, out, b->d_work);
        else
            hipLaunchKernelGGL((cast_kernel<ANYHIT, false>), dim3(grid), dim3(kBlock), lds, st, sc,
                               d_org, org_stride, d_dir, (unsigned)n, out, b->d_work);
    }
    launch_heavy();
    UPSP_HIP_CHECK(hipGetLastError());
    if (b->stats_on) return read_stats(b, st);
    return UPSP_OK;
}

}  // namespace
}  // namespace upsp

namespace upsp {
namespace {
struct DevBuf {
    void *p = nullptr;
    ~DevBuf()
    {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
};

// Small host batches (the per-ray calls of the pybind `BVH.intersect`): one pinned
// host block + one device block cached in the BVH, one H2D copy, one launch, one D2H
// copy on a private stream -- no allocation, no device-wide synchronisation.
constexpr size_t kStageRays = 4096;
constexpr size_t kStageBytesPerRay = 24 + 1 + 4 + 4 + 12 + 12 + 12 + 3;  // in + outputs (+pad)

template <bool ANYHIT>
int cast_host_small(upsp_bvh *b, const float *h_org, int org_stride, const float *h_dir, size_t n,
                    const upsp_hits &h_out)
{
    const size_t cap = kStageRays * 72;
    if (!b->d_stage) {
        UPSP_HIP_CHECK(hipMalloc(&b->d_stage, cap));
        UPSP_HIP_CHECK(hipHostMalloc(&b->h_stage, cap, hipHostMallocDefault));
        UPSP_HIP_CHECK(hipStreamCreateWithFlags(&b->stage_stream, hipStreamNonBlocking));
    }
    hipStream_t st = b->stage_stream;
    // layout (both sides): org[3n] dir[3n] | t[n] prim[n] uvw[3n] pos[3n] nrm[3n] hit[n]
    char *h = static_cast<char *>(b->h_stage), *d = static_cast<char *>(b->d_stage);
    const size_t in_bytes = 24 * n, o_t = in_bytes, o_prim = o_t + 4 * n, o_uvw = o_prim + 4 * n,
                 o_pos = o_uvw + 12 * n, o_nrm = o_pos + 12 * n, o_hit = o_nrm + 12 * n,
                 total = o_hit + n;
    float *ho = reinterpret_cast<float *>(h);
    for (size_t i = 0; i < n; ++i) {
        const float *o = h_org + (org_stride ? 3 * i : 0);
        ho[3 * i] = o[0]; ho[3 * i + 1] = o[1]; ho[3 * i + 2] = o[2];
    }
    std::memcpy(h + 12 * n, h_dir, 12 * n);
    UPSP_HIP_CHECK(hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, st));
    upsp_hits dv;
    std::memset(&dv, 0, sizeof(dv));
    dv.hit = reinterpret_cast<uint8_t *>(d + o_hit);
    if (!ANYHIT) {
        dv.t = reinterpret_cast<float *>(d + o_t);
        dv.prim = reinterpret_cast<int32_t *>(d + o_prim);
        dv.uvw = reinterpret_cast<float *>(d + o_uvw);
        dv.pos = reinterpret_cast<float *>(d + o_pos);
        dv.nrm = reinterpret_cast<float *>(d + o_nrm);
    }
    int rc = launch_cast<ANYHIT>(b, reinterpret_cast<const float *>(d), 3,
                                 reinterpret_cast<const float *>(d + 12 * n), n, dv, st);
    if (rc != UPSP_OK) return rc;
    UPSP_HIP_CHECK(hipMemcpyAsync(h + in_bytes, d + in_bytes, total - in_bytes, hipMemcpyDeviceToHost, st));
    rc = check_walk_error(b, st);                   // (synchronises st)
    if (rc != UPSP_OK) return rc;
    if (h_out.hit) std::memcpy(h_out.hit, h + o_hit, n);
    if (!ANYHIT) {
        if (h_out.t) std::memcpy(h_out.t, h + o_t, 4 * n);
        if (h_out.prim) std::memcpy(h_out.prim, h + o_prim, 4 * n);
        if (h_out.uvw) std::memcpy(h_out.uvw, h + o_uvw, 12 * n);
        if (h_out.pos) std::memcpy(h_out.pos, h + o_pos, 12 * n);
        if (h_out.nrm) std::memcpy(h_out.nrm, h + o_nrm, 12 * n);
    }
    return UPSP_OK;
}

template <bool ANYHIT>
int cast_host(const upsp_bvh *bvh, const float *h_org, int org_stride, const float *h_dir,
              size_t n, const upsp_hits &h_out)
{
    if (!bvh) return fail(UPSP_ERR_INVALID, "null BVH");
    if (n == 0) return UPSP_OK;
    if (!h_org || !h_dir) return fail(UPSP_ERR_INVALID, "null ray buffers");
    if (org_stride != 0 && org_stride != 3) return fail(UPSP_ERR_INVALID, "org_stride must be 0 or 3");
    if (n <= kStageRays && !bvh->stats_on)
        return cast_host_small<ANYHIT>(const_cast<upsp_bvh *>(bvh), h_org, org_stride, h_dir, n, h_out);
    DevBuf org, dir, hit, t, prim, uvw, pos, nrm;
    const size_t no = org_stride ? 3 * n : 3;
    UPSP_HIP_CHECK(org.alloc(no * 4));
    UPSP_HIP_CHECK(dir.alloc(3 * n * 4));
    UPSP_HIP_CHECK(hipMemcpy(org.p, h_org, no * 4, hipMemcpyHostToDevice));
    UPSP_HIP_CHECK(hipMemcpy(dir.p, h_dir, 3 * n * 4, hipMemcpyHostToDevice));
    upsp_hits d;
    std::memset(&d, 0, sizeof(d));
    if (h_out.hit) { UPSP_HIP_CHECK(hit.alloc(n)); d.hit = (uint8_t *)hit.p; }
    if (!ANYHIT) {
        if (h_out.t) { UPSP_HIP_CHECK(t.alloc(n * 4)); d.t = (float *)t.p; }
        if (h_out.prim) { UPSP_HIP_CHECK(prim.alloc(n * 4)); d.prim = (int32_t *)prim.p; }
        if (h_out.uvw) { UPSP_HIP_CHECK(uvw.alloc(n * 12)); d.uvw = (float *)uvw.p; }
        if (h_out.pos) { UPSP_HIP_CHECK(pos.alloc(n * 12)); d.pos = (float *)pos.p; }
        if (h_out.nrm) { UPSP_HIP_CHECK(nrm.alloc(n * 12)); d.nrm = (float *)nrm.p; }
    }
    int rc = launch_cast<ANYHIT>(bvh, (const float *)org.p, org_stride, (const float *)dir.p, n,
                                 d, nullptr);
    if (rc != UPSP_OK) return rc;
    rc = check_walk_error(const_cast<upsp_bvh *>(bvh), nullptr);   // (synchronises the stream)
    if (rc != UPSP_OK) return rc;
    if (d.hit) UPSP_HIP_CHECK(hipMemcpy(h_out.hit, d.hit, n, hipMemcpyDeviceToHost));
    if (d.t) UPSP_HIP_CHECK(hipMemcpy(h_out.t, d.t, n * 4, hipMemcpyDeviceToHost));
    if (d.prim) UPSP_HIP_CHECK(hipMemcpy(h_out.prim, d.prim, n * 4, hipMemcpyDeviceToHost));
    if (d.uvw) UPSP_HIP_CHECK(hipMemcpy(h_out.uvw, d.uvw, n * 12, hipMemcpyDeviceToHost));
    if (d.pos) UPSP_HIP_CHECK(hipMemcpy(h_out.pos, d.pos, n * 12, hipMemcpyDeviceToHost));
    if (d.nrm) UPSP_HIP_CHECK(hipMemcpy(h_out.nrm, d.nrm, n * 12, hipMemcpyDeviceToHost));
    return UPSP_OK;
}
}  // namespace
}  // namespace upsp

using namespace upsp;

extern "C" {

const char *upsp_last_error(void) { return g_error.c_str(); }

int upsp_version(void) { return 100; }

int upsp_device_info(int *n_devices, char *arch, int *n_cus)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    if (n_devices) *n_devices = n;
    if (arch) arch[0] = 0;
    if (n_cus) *n_cus = 0;
    if (n == 0) return fail(UPSP_ERR_NO_DEVICE, "no HIP device visible");
    hipDeviceProp_t dp;
    int dev = 0;
    UPSP_HIP_CHECK(hipGetDevice(&dev));
    UPSP_HIP_CHECK(hipGetDeviceProperties(&dp, dev));
    if (arch) {
        std::strncpy(arch, dp.gcnArchName, 31);
        arch[31] = 0;
    }
    if (n_cus) *n_cus = dp.multiProcessorCount;
    return UPSP_OK;
}

int upsp_bvh_create(const float *h_tris9, size_t ntris, upsp_bvh **out)
{
    if (!out) return fail(UPSP_ERR_INVALID, "out is null");
    *out = nullptr;
    if (ntris == 0 || !h_tris9) return fail(UPSP_ERR_EMPTY, "BVH::BVH() : no primitives!");
    if (ntris >= (1u << 25)) return fail(UPSP_ERR_INVALID, "more than 2^25 triangles");
    auto t0 = std::chrono::steady_clock::now();
    HostBvh hb;
    build_bvh(h_tris9, ntris, hb);
    if (hb.depth > 64) return fail(UPSP_ERR_DEPTH, "BVH deeper than 64 levels");

    upsp_bvh *b = new upsp_bvh();
    UPSP_HIP_CHECK(hipGetDevice(&b->device));
    const size_t nb = std::max<size_t>(hb.nodes.size(), 1) * sizeof(GpuNode);
    const size_t tb = hb.tris.size() * sizeof(GpuTri);
    hipError_t e = hipMalloc(&b->d_nodes, nb);
    if (e == hipSuccess) e = hipMalloc(&b->d_tris, tb);
    if (e == hipSuccess) e = hipMalloc(&b->d_work, (kWorkWords + 4) * sizeof(unsigned));   // + the walks' error word
    if (e == hipSuccess && !hb.nodes.empty())
        e = hipMemcpy(b->d_nodes, hb.nodes.data(), hb.nodes.size() * sizeof(GpuNode),
                      hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(b->d_tris, hb.tris.data(), tb, hipMemcpyHostToDevice);
    const size_t wb = std::max<size_t>(hb.wide.size(), 1) * sizeof(GpuWide);
    if (e == hipSuccess) e = hipMalloc(&b->d_wide, wb);
    if (e == hipSuccess && !hb.wide.empty()) e = hipMemcpy(b->d_wide, hb.wide.data(), hb.wide.size() * sizeof(GpuWide), hipMemcpyHostToDevice);
    b->wide_root = hb.wide_root;
    b->n_wide = (uint32_t)hb.wide.size();
    b->wide_depth = hb.wide_depth;
    if (e == hipSuccess) e = hipMemset(b->d_work, 0, (kWorkWords + 4) * sizeof(unsigned));
    b->d_err = b->d_work + kWorkWords;      // (outlives the per-call clearing of the work words)
    if (e != hipSuccess) {
        upsp_bvh_destroy(b);
        return fail(UPSP_ERR_HIP, std::string("BVH upload: ") + hipGetErrorString(e));
    }
    {
        // box chain of every leaf: the child boxes a ray must pass, root downwards, to reach it
        // (depth-first with an explicit stack; `cur` is the chain of the frame being expanded)
        std::vector<uint32_t> slot_path(2 * hb.tris.size(), 0u), path_ref;
        struct Frame { int32_t ref; uint32_t entry; uint32_t depth; };
        std::vector<Frame> st2;
        std::vector<uint32_t> cur(65, 0u);
        st2.push_back({hb.root_ref, 0u, 0u});
        while (!st2.empty()) {
            const Frame f = st2.back();
            st2.pop_back();
            if (f.depth > 0) cur[f.depth - 1] = f.entry;
            if (f.ref < 0) {
                const uint32_t code = (uint32_t)(~f.ref);
                const uint32_t first = code >> kLeafBits, count = (code & (kMaxLeaf - 1)) + 1;
                const uint32_t off = (uint32_t)path_ref.size();
                path_ref.insert(path_ref.end(), cur.begin(), cur.begin() + f.depth);
                for (uint32_t k = 0; k < count; ++k) {
                    slot_path[2 * (size_t)(first + k)] = off;
                    slot_path[2 * (size_t)(first + k) + 1] = f.depth;
                }
                continue;
            }
            int32_t left, right;
            std::memcpy(&left, &hb.nodes[(size_t)f.ref].q[12], 4);
            std::memcpy(&right, &hb.nodes[(size_t)f.ref].q[13], 4);
            st2.push_back({right, ((uint32_t)f.ref << 1) | 1u, f.depth + 1});
            st2.push_back({left, ((uint32_t)f.ref << 1) | 0u, f.depth + 1});
        }
        for (int k = 0; k < 4; ++k) path_ref.push_back(path_ref.empty() ? 0u : path_ref.back());  // read-ahead pad
        e = hipMalloc(&b->d_slot_path, slot_path.size() * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc(&b->d_path_ref, path_ref.size() * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemcpy(b->d_slot_path, slot_path.data(), slot_path.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(b->d_path_ref, path_ref.data(), path_ref.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            upsp_bvh_destroy(b);
            return fail(UPSP_ERR_HIP, std::string("BVH upload: ") + hipGetErrorString(e));
        }
    }
    b->root_ref = hb.root_ref;
    b->top_nodes = hb.top_nodes;
    b->prim_slot.assign(ntris, 0u);
    for (size_t sl = 0; sl < hb.tris.size(); ++sl) b->prim_slot[(size_t)hb.tris[sl].prim] = (uint32_t)sl;
    std::memset(&b->info, 0, sizeof(b->info));
    for (int a = 0; a < 3; ++a) {
        b->root_min[a] = b->info.bounds_min[a] = hb.root_min[a];
        b->root_max[a] = b->info.bounds_max[a] = hb.root_max[a];
    }
    b->info.ntris = ntris;
    b->info.n_ref_nodes = hb.n_ref_nodes;
    b->info.n_gpu_nodes = (uint32_t)hb.nodes.size();
    b->info.depth = hb.depth;
    b->info.max_leaf = hb.max_leaf;
    b->info.device_bytes = nb + tb + wb;
    b->info.build_seconds =
        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    *out = b;
    return UPSP_OK;
}

// A second handle on the same tree: the immutable arrays (nodes, wide records, triangles, leaf paths, the node -> triangle
// adjacency as it is now) are shared, the scratch of the queries (work words, retry / witness / hand-off lists, stacks) is its
// own -- so queries on the two handles may run at the same time on different streams (the projection builds of several cameras
// of one model, cpp/exec/psp_process.cpp:1586-1660, are independent of each other).
int upsp_bvh_share(const upsp_bvh *src, upsp_bvh **out)
{
    if (!src || !out) return fail(UPSP_ERR_INVALID, "null argument");
    *out = nullptr;
    upsp_bvh *b = new upsp_bvh();
    b->shared_from = src->shared_from ? src->shared_from : src;
    b->device = src->device;
    b->d_nodes = src->d_nodes;
    b->d_wide = src->d_wide;
    b->wide_root = src->wide_root;
    b->n_wide = src->n_wide;
    b->wide_depth = src->wide_depth;
    b->d_tris = src->d_tris;
    b->d_slot_path = src->d_slot_path;
    b->d_path_ref = src->d_path_ref;
    b->d_adj_off = src->d_adj_off;
    b->d_adj_slot = src->d_adj_slot;
    b->adj_src = src->adj_src;
    b->adj_nnodes = src->adj_nnodes;
    b->prim_slot = src->prim_slot;
    b->root_ref = src->root_ref;
    b->top_nodes = src->top_nodes;
    for (int a = 0; a < 3; ++a) {
        b->root_min[a] = src->root_min[a];
        b->root_max[a] = src->root_max[a];
    }
    b->info = src->info;
    hipError_t e = hipMalloc(&b->d_work, (kWorkWords + 4) * sizeof(unsigned));
    if (e == hipSuccess) e = hipMemset(b->d_work, 0, (kWorkWords + 4) * sizeof(unsigned));
    if (e != hipSuccess) {
        upsp_bvh_destroy(b);
        return fail(UPSP_ERR_HIP, std::string("BVH share: ") + hipGetErrorString(e));
    }
    b->d_err = b->d_work + kWorkWords;
    *out = b;
    return UPSP_OK;
}

void upsp_bvh_destroy(upsp_bvh *b)
{
    if (!b) return;
    if (!b->shared_from) {
        if (b->d_nodes) (void)hipFree(b->d_nodes);
        if (b->d_wide) (void)hipFree(b->d_wide);
        if (b->d_tris) (void)hipFree(b->d_tris);
        if (b->d_adj_off) (void)hipFree(b->d_adj_off);
        if (b->d_adj_slot) (void)hipFree(b->d_adj_slot);
        if (b->d_slot_path) (void)hipFree(b->d_slot_path);
        if (b->d_path_ref) (void)hipFree(b->d_path_ref);
    }
    if (b->d_work) (void)hipFree(b->d_work);
    if (b->d_retry_nodes) (void)hipFree(b->d_retry_nodes);
    if (b->d_retry_mask) (void)hipFree(b->d_retry_mask);
    if (b->d_witness) (void)hipFree(b->d_witness);
    if (b->d_todo_mask) (void)hipFree(b->d_todo_mask);
    if (b->d_todo_rays) (void)hipFree(b->d_todo_rays);
    if (b->d_heavy) (void)hipFree(b->d_heavy);
    if (b->d_heavy_scratch) (void)hipFree(b->d_heavy_scratch);
    if (b->d_cast_list) (void)hipFree(b->d_cast_list);
    if (b->d_steps) (void)hipFree(b->d_steps);
    if (b->d_step_edges) (void)hipFree(b->d_step_edges);
    if (b->h_handoff) (void)hipHostFree(b->h_handoff);
    if (b->ev_handoff) (void)hipEventDestroy(b->ev_handoff);
    if (b->d_stage) (void)hipFree(b->d_stage);
    if (b->h_stage) (void)hipHostFree(b->h_stage);
    if (b->stage_stream) (void)hipStreamDestroy(b->stage_stream);
    delete b;
}

int upsp_bvh_get_info(const upsp_bvh *b, upsp_bvh_info *info)
{
    if (!b || !info) return fail(UPSP_ERR_INVALID, "null argument");
    *info = b->info;
    return UPSP_OK;
}

int upsp_bvh_set_tri_nodes(upsp_bvh *b, const int32_t *d_tri_nodes, size_t nnodes, void *stream)
{
    if (!b) return fail(UPSP_ERR_INVALID, "null BVH");
    if (b->shared_from) return fail(UPSP_ERR_INVALID, "upsp_bvh_set_tri_nodes: set the adjacency on the owner of the tree, then share");
    if (b->d_adj_off) (void)hipFree(b->d_adj_off);
    if (b->d_adj_slot) (void)hipFree(b->d_adj_slot);
    b->d_adj_off = b->d_adj_slot = nullptr;
    b->adj_src = nullptr;
    b->adj_nnodes = 0;
    if (!d_tri_nodes || nnodes == 0) return UPSP_OK;          // cleared
    const size_t ntris = b->info.ntris;
    if (nnodes >= 0xFFFFFFF0ull) return fail(UPSP_ERR_INVALID, "too many nodes");
    hipStream_t st = (hipStream_t)stream;
    std::vector<int32_t> tn(3 * ntris);
    UPSP_HIP_CHECK(hipMemcpyAsync(tn.data(), d_tri_nodes, tn.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    UPSP_HIP_CHECK(hipStreamSynchronize(st));
    std::vector<uint32_t> off(nnodes + 1, 0u), slot(3 * ntris);
    for (size_t i = 0; i < tn.size(); ++i) {
        if (tn[i] < 0 || (size_t)tn[i] >= nnodes) return fail(UPSP_ERR_INVALID, "triangle node index out of range");
        ++off[(size_t)tn[i] + 1];
    }
    for (size_t n = 0; n < nnodes; ++n) off[n + 1] += off[n];
    std::vector<uint32_t> fill(off.begin(), off.end() - 1);
    for (size_t t = 0; t < ntris; ++t)
        for (int k = 0; k < 3; ++k) {
            const size_t n = (size_t)tn[3 * t + k];
            // a degenerate triangle listing the node twice is entered twice: harmless (min / max)
            slot[fill[n]++] = b->prim_slot[t];
        }
    UPSP_HIP_CHECK(hipMalloc(&b->d_adj_off, off.size() * sizeof(uint32_t)));
    UPSP_HIP_CHECK(hipMalloc(&b->d_adj_slot, std::max<size_t>(slot.size(), 1) * sizeof(uint32_t)));
    UPSP_HIP_CHECK(hipMemcpy(b->d_adj_off, off.data(), off.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    UPSP_HIP_CHECK(hipMemcpy(b->d_adj_slot, slot.data(), slot.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    b->adj_src = d_tri_nodes;
    b->adj_nnodes = nnodes;
    return UPSP_OK;
}

int upsp_bvh_check(upsp_bvh *b, void *stream)
{
    if (!b) return fail(UPSP_ERR_INVALID, "null BVH");
    return check_walk_error(b, (hipStream_t)stream);
}

int upsp_bvh_enable_stats(upsp_bvh *b, int on)
{
    if (!b) return fail(UPSP_ERR_INVALID, "null BVH");
    b->stats_on = on ? 1 : 0;
    return UPSP_OK;
}

int upsp_bvh_last_stats(const upsp_bvh *b, uint64_t *nodes, uint64_t *tris, uint64_t *rays)
{
    if (!b) return fail(UPSP_ERR_INVALID, "null BVH");
    if (nodes) *nodes = b->last_stats[0];
    if (tris) *tris = b->last_stats[1];
    if (rays) *rays = b->last_stats[2];
    return UPSP_OK;
}

int upsp_bvh_last_filter_stats(const upsp_bvh *b, uint64_t *boxes, uint64_t *undecided)
{
    if (!b) return fail(UPSP_ERR_INVALID, "null BVH");
    if (boxes) *boxes = b->last_slab[0];
    if (undecided) *undecided = b->last_slab[1];
    return UPSP_OK;
}

int upsp_bvh_intersect(const upsp_bvh *bvh, const float *d_org, int org_stride,
                       const float *d_dir, size_t n, const upsp_hits *d_out, void *stream)
{
    if (!d_out) return fail(UPSP_ERR_INVALID, "null output");
    return launch_cast<false>(bvh, d_org, org_stride, d_dir, n, *d_out, (hipStream_t)stream);
}

int upsp_bvh_occluded(const upsp_bvh *bvh, const float *d_org, int org_stride,
                      const float *d_dir, size_t n, uint8_t *d_hit, void *stream)
{
    if (!d_hit) return fail(UPSP_ERR_INVALID, "null output");
    upsp_hits out;
    std::memset(&out, 0, sizeof(out));
    out.hit = d_hit;
    return launch_cast<true>(bvh, d_org, org_stride, d_dir, n, out, (hipStream_t)stream);
}

int upsp_bvh_intersect_host(const upsp_bvh *bvh, const float *h_org, int org_stride,
                            const float *h_dir, size_t n, const upsp_hits *h_out)
{
    if (!h_out) return fail(UPSP_ERR_INVALID, "null output");
    return cast_host<false>(bvh, h_org, org_stride, h_dir, n, *h_out);
}

int upsp_bvh_occluded_host(const upsp_bvh *bvh, const float *h_org, int org_stride,
                           const float *h_dir, size_t n, uint8_t *h_hit)
{
    if (!h_hit) return fail(UPSP_ERR_INVALID, "null output");
    upsp_hits out;
    std::memset(&out, 0, sizeof(out));
    out.hit = h_hit;
    return cast_host<true>(bvh, h_org, org_stride, h_dir, n, out);
}

int upsp_projection_fetch_counts(upsp_bvh *b, uint64_t *nrays, uint64_t *primary_rays,
                                 uint64_t *retry_nodes, void *stream)
{
    if (!b) return fail(UPSP_ERR_INVALID, "null BVH");
    hipStream_t st = (hipStream_t)stream;
    int rc = read_stats(b, st);
    if (rc != UPSP_OK) return rc;
    unsigned cnt[2] = {0, 0};
    UPSP_HIP_CHECK(hipMemcpy(cnt, b->d_work + kWorkRetryCount, sizeof(cnt), hipMemcpyDeviceToHost));
    b->last_retry_nodes = cnt[0];
    b->last_primary = cnt[1];
    if (nrays) *nrays = b->last_stats[2];
    if (primary_rays) *primary_rays = b->last_primary;
    if (retry_nodes) *retry_nodes = b->last_retry_nodes;
    return UPSP_OK;
}

int upsp_projection_last_counts(const upsp_bvh *b, uint64_t *primary_rays, uint64_t *retry_nodes)
{
    if (!b) return fail(UPSP_ERR_INVALID, "null BVH");
    if (primary_rays) *primary_rays = b->last_primary;
    if (retry_nodes) *retry_nodes = b->last_retry_nodes;
    return UPSP_OK;
}

int upsp_camera_center(const upsp_camera *cam, double c[3])
{
    if (!cam || !c) return fail(UPSP_ERR_INVALID, "null argument");
    const double *R = cam->R, *t = cam->t;
    for (int i = 0; i < 3; ++i) c[i] = -(R[0 + i] * t[0] + R[3 + i] * t[1] + R[6 + i] * t[2]);
    return UPSP_OK;
}

int upsp_project_points_host(const upsp_camera *cam, const float *h_xyz, size_t n, float *h_uv)
{
    if (!cam || (!h_xyz && n) || (!h_uv && n)) return fail(UPSP_ERR_INVALID, "null argument");
    for (size_t i = 0; i < n; ++i)
        project_point(cam->K, cam->dist, cam->R, cam->t, h_xyz[3 * i], h_xyz[3 * i + 1],
                      h_xyz[3 * i + 2], h_uv[2 * i], h_uv[2 * i + 1]);
    return UPSP_OK;
}

int upsp_projection_build(upsp_bvh *b, const upsp_camera *cam, const float *d_nodes,
                          const float *d_normals, const uint8_t *d_datanode,
                          const int32_t *d_tri_nodes, size_t nnodes, float oblique_thresh,
                          int32_t *d_pix, float *d_uv, uint8_t *d_nodecount, uint64_t *h_nrays,
                          void *stream)
{
    if (!b || !cam) return fail(UPSP_ERR_INVALID, "null BVH / camera");
    if (nnodes == 0) {
        if (h_nrays) *h_nrays = 0;
        return UPSP_OK;
    }
    if (!d_nodes || !d_normals || !d_tri_nodes || !d_pix || !d_uv)
        return fail(UPSP_ERR_INVALID, "null device buffer");
    if (cam->width <= 0 || cam->height <= 0) return fail(UPSP_ERR_INVALID, "bad image size");
    if (nnodes > 0xF0000000ull) return fail(UPSP_ERR_INVALID, "too many nodes");
    hipStream_t st = (hipStream_t)stream;
    const int entries = stack_entries(b);
    if (entries > 64) return fail(UPSP_ERR_DEPTH, "BVH deeper than 64 levels");
    const size_t lds = lds_bytes(b);

    Cam c;
    std::memcpy(c.K, cam->K, sizeof(c.K));
    std::memcpy(c.dist, cam->dist, sizeof(c.dist));
    std::memcpy(c.R, cam->R, sizeof(c.R));
    std::memcpy(c.t, cam->t, sizeof(c.t));
    double cc[3];
    upsp_camera_center(cam, cc);
    c.ox = (float)cc[0];
    c.oy = (float)cc[1];
    c.oz = (float)cc[2];
    c.W = cam->width;
    c.H = cam->height;

    // retry list: worst case every node
    if (b->retry_capacity < nnodes) {
        if (b->d_retry_nodes) (void)hipFree(b->d_retry_nodes);
        if (b->d_retry_mask) (void)hipFree(b->d_retry_mask);
        if (b->d_witness) (void)hipFree(b->d_witness);
        if (b->d_todo_mask) (void)hipFree(b->d_todo_mask);
        if (b->d_todo_rays) (void)hipFree(b->d_todo_rays);
        b->d_retry_nodes = b->d_retry_mask = b->d_todo_mask = b->d_todo_rays = nullptr;
        b->d_witness = nullptr;
        b->retry_capacity = 0;
        UPSP_HIP_CHECK(hipMalloc(&b->d_retry_nodes, sizeof(unsigned) * nnodes));
        UPSP_HIP_CHECK(hipMalloc(&b->d_retry_mask, sizeof(unsigned) * nnodes));
        UPSP_HIP_CHECK(hipMalloc(&b->d_witness, sizeof(int32_t) * nnodes));
        UPSP_HIP_CHECK(hipMalloc(&b->d_todo_mask, sizeof(unsigned) * nnodes));
        UPSP_HIP_CHECK(hipMalloc(&b->d_todo_rays, sizeof(unsigned) * 6 * nnodes));
        b->retry_capacity = nnodes;
    }
    // length-homogeneous waves (kRayBins): OFF by default -- measured slower (see kRayBins); UPSP_RAY_BINS=1 turns them on
    const bool bins_on = env_int("UPSP_RAY_BINS", 0) != 0 && nnodes < 0x7FFFFFFFull / kRayBins;
    if (!bins_on) b->steps_valid = false;
    if (bins_on && b->steps_nnodes != nnodes) {
        if (b->d_steps) (void)hipFree(b->d_steps);
        b->d_steps = nullptr;
        b->steps_valid = false;
        b->steps_nnodes = 0;
        UPSP_HIP_CHECK(hipMalloc(&b->d_steps, sizeof(uint16_t) * nnodes));
        UPSP_HIP_CHECK(hipMemsetAsync(b->d_steps, 0, sizeof(uint16_t) * nnodes, st));
        if (!b->d_step_edges) UPSP_HIP_CHECK(hipMalloc(&b->d_step_edges, sizeof(uint32_t) * kRayBins));
        b->steps_nnodes = nnodes;
    }
    if (!b->d_heavy) UPSP_HIP_CHECK(hipMalloc(&b->d_heavy, sizeof(unsigned) * 2 * kHeavyCap));      // (two lists)
    if (!b->d_heavy_scratch) UPSP_HIP_CHECK(hipMalloc(&b->d_heavy_scratch, kHeavyScratchBytes * kHeavyGridMax));
    // (the work words are cleared by project_nodes_kernel, the chain's first launch)
    const int grid = grid_for(nnodes, lds);
    Scene sc = make_scene(b, nnodes, grid);
    // step 3 runs over 6 x (listed nodes), a count only the device knows: full
    // persistent grid, chunk sized for the typical case (half of the nodes listed)
    const int grid1 = grid_for(6 * nnodes, lds);
    Scene sc1 = make_scene(b, 3 * nnodes, grid1);
    // rays that need more than this many node visits + triangle tests leave the one-lane traversal (heavy_kernel);
    // the keys of heavy_kernel hold 56 levels.  Measured on MI355X (1M-triangle models, 1024^2 camera), build time:
    //   UV-sphere model (1000-triangle polar fans): off 2.89 ms, 320: 1.26, 256: 1.14, 160: 1.05, 128: 1.02, 96: 1.10, 64: 1.74
    //   cube-sphere model (valence <= 6):           off 0.59 ms, 256: 0.59, 160: 0.59, 128: 0.63 (ONE ray), 96: 0.66, 64: 1.25
    // a ray costs heavy_kernel 40-100 us however few steps it had left (one round per tree level, ~1.3 us each), so the
    // threshold sits above the longest ordinary ray of the well-shaped model
    // (read per build, not once: the tests move both to drive many rays through heavy_kernel and its fallback)
    const int heavy_steps = env_int("UPSP_HEAVY_STEPS", 96);
    const int heavy_stack = env_int("UPSP_HEAVY_STACK", (int)kHeavyStack);
    bool heavy_on = heavy_steps > 0 && b->info.depth <= 56 && b->d_heavy;
    // The hand-off machinery only when the last build of the SAME view needed it: its hand-off counts (work[16..20]) were copied to
    // pinned memory behind an event; when that copy has arrived, belongs to this camera / node array / node count and says that no
    // ray was handed off, this build runs without the threshold -- every ray finishes in its lane, same results -- and without the
    // four walk launches (6 us each of finding their lists empty, on the critical path of a build that runs beside the frame loop).
    // UPSP_HEAVY_ADAPT=0: always launch them (the tests of the walks).
    const bool adapt = heavy_on && env_int("UPSP_HEAVY_ADAPT", 1) != 0 && !h_nrays && !b->stats_on;
    if (adapt && b->handoff_pending && b->handoff_was_on && b->handoff_nodes == (const void *)d_nodes && b->handoff_nnodes == nnodes &&
        std::memcmp(&b->handoff_cam, cam, sizeof(upsp_camera)) == 0 && hipEventQuery(b->ev_handoff) == hipSuccess) {
        const uint32_t *h = b->h_handoff;
        if (h[0] == 0u && h[1] == 0u && h[3] == 0u && h[4] == 0u) heavy_on = false;
    } else if (adapt && b->handoff_pending && !b->handoff_was_on && b->handoff_nodes == (const void *)d_nodes && b->handoff_nnodes == nnodes &&
               std::memcmp(&b->handoff_cam, cam, sizeof(upsp_camera)) == 0) {
        // the view that needed no hand-off stays without it -- but every 16th build looks again (the nodes may have moved in place)
        if (++b->handoff_skips < 16u) heavy_on = false;
        else b->handoff_skips = 0;
    }
    const bool heavy_launch = heavy_on;
    if (heavy_on) {
        sc.heavy_steps = sc1.heavy_steps = (unsigned)heavy_steps;
        sc.heavy_items = sc1.heavy_items = b->d_heavy;
        sc.heavy_scratch = sc1.heavy_scratch = reinterpret_cast<unsigned char *>(b->d_heavy_scratch);
        sc.heavy_stack = sc1.heavy_stack = (unsigned)std::min<int>(std::max(heavy_stack, 128), (int)kHeavyStack);
    }
    sc.pack_waves = sc1.pack_waves = kPackWavesPerSimd * 4u * (unsigned)(props().cus > 0 ? props().cus : 256);
    const int heavy_grid = (int)kHeavyGridMax;
    const int wave_grid = 5 * (props().cus > 0 ? props().cus : 256);
#define UPSP_LAUNCH_HEAVY(PHASE, SC)                                                                             \
    if (heavy_launch) {                                                                                          \
        KTimed kth("heavy_kernel", st);                                                                          \
        hipLaunchKernelGGL((wave_proj_kernel<PHASE, 1>), dim3(4 * wave_grid), dim3(64), 0, st, SC, c, d_nodes, d_tri_nodes, d_pix, \
                           (const unsigned *)b->d_retry_nodes, b->d_retry_mask, b->d_work, (const unsigned *)b->d_heavy,  \
                           b->d_heavy + kHeavyCap);                                                              \
        hipLaunchKernelGGL((heavy_kernel<PHASE>), dim3(heavy_grid), dim3(kHeavyThreads), 0, st, SC, c, d_nodes, d_tri_nodes, \
                           d_pix, (const unsigned *)b->d_retry_nodes, b->d_retry_mask, (const unsigned *)b->d_work,     \
                           kWorkWaveOver + (PHASE ? 1 : 0), (const unsigned *)(b->d_heavy + kHeavyCap));           \
    }
    bool use_witness = false;
    if (b->d_adj_off && b->adj_src == (const void *)d_tri_nodes && b->adj_nnodes == nnodes) {
        // bounded visibility rays (own_bound) for the retry pass only: measured on MI355X the
        // primary rays gain nothing (the near-first traversal finds the front surface at once
        // and prunes behind it; 0.44 -> 0.47 ms with the extra own-triangle tests), the retries
        // lose the rays that miss every own triangle and exit early: 0.70 -> 0.59 ms
        sc1.adj_off = b->d_adj_off;
        sc1.adj_slot = b->d_adj_slot;
        // occluder witness (witness_kernel): the primary pass records the triangle it hit, the
        // retries test that triangle and its box chain first and only the rest is traversed
        if (b->d_slot_path && b->d_path_ref && b->d_witness && b->d_todo_mask && b->d_todo_rays) {
            use_witness = true;
            sc.witness = b->d_witness;
            sc1.witness = b->d_witness;
            sc1.slot_path = reinterpret_cast<const uint2 *>(b->d_slot_path);
            sc1.path_ref = b->d_path_ref;
        }
    }
    const dim3 egrid((unsigned)((nnodes + 255) / 256)), eblock(256);
    // (a cache-warming sweep over the BVH before the traversals was measured neutral in round 1 and 7-12 us slower per build in
    //  round 2: not done here; the batch queries keep theirs)
    {
        KTimed kt("project_nodes_kernel", st);
        // (UPSP_OBLIQUE_CULL=0: always cast the reference's rays; 2: never, even when they are counted -- for profiling)
        static const int cull_env = env_int("UPSP_OBLIQUE_CULL", 1);
        const int cull = (cull_env == 2 || (cull_env == 1 && !h_nrays && !b->stats_on)) ? 1 : 0;
        hipLaunchKernelGGL(project_nodes_kernel, egrid, eblock, 0, st, c, d_nodes, d_datanode,
                           (unsigned)nnodes, d_normals, oblique_thresh, cull, d_pix, d_uv, b->d_work);
    }
    const bool binned = bins_on && b->steps_valid;
    if (bins_on) sc.steps_out = b->d_steps;
    {
        KTimed kt("primary_list_kernel", st);
        const dim3 lgrid((unsigned)((nnodes + 256 * kRetryListItems - 1) / (256 * kRetryListItems)));
        if (binned) {
            sc.nbins = kRayBins;
            sc.bin_stride = (unsigned)nnodes;
            hipLaunchKernelGGL(primary_list_binned_kernel, lgrid, eblock, 0, st, (const int32_t *)d_pix, (unsigned)nnodes,
                               (const unsigned short *)b->d_steps, (const unsigned *)b->d_step_edges, b->d_todo_rays, (unsigned)nnodes,
                               b->d_work);
        } else {
            hipLaunchKernelGGL(retry_list_kernel<kPixInFrame>, lgrid, eblock, 0, st, (const int32_t *)d_pix,
                               (unsigned)nnodes, b->d_todo_rays, (unsigned *)nullptr, b->d_work);
        }
    }
#define UPSP_LAUNCH_PROJ(STATS, PHASE, G, SC)                                                    \
    do {                                                                                         \
        if (PHASE == 0)     /* (one-wave workgroups for the primary pass too: the pass 0.28 -> 0.21 ms beside pass A, which then takes 0.47 instead of 0.40: step 0.87 -> 0.93 ms) */ \
            hipLaunchKernelGGL((projection_kernel<STATS, PHASE, kBlock>), dim3(G), dim3(kBlock), lds, st, SC, c, \
                               d_nodes, d_tri_nodes, (unsigned)nnodes, d_pix, b->d_retry_nodes,  \
                               b->d_retry_mask, (const unsigned *)b->d_todo_rays, b->d_work);    \
        else                                                                                     \
            hipLaunchKernelGGL((projection_kernel<STATS, PHASE, 64>), dim3((G) * (kBlock / 64)), dim3(64), lds / (kBlock / 64), st, SC, c, \
                               d_nodes, d_tri_nodes, (unsigned)nnodes, d_pix, b->d_retry_nodes,  \
                               b->d_retry_mask, (const unsigned *)b->d_todo_rays, b->d_work);    \
    } while (0)
    {
        KTimed kt("projection_kernel<primary>", st);
        if (b->stats_on) UPSP_LAUNCH_PROJ(true, 0, grid, sc); else UPSP_LAUNCH_PROJ(false, 0, grid, sc);
    }
    UPSP_LAUNCH_HEAVY(0, sc)
    if (bins_on) {      // bin edges for the next build of this handle, from the step counts the primary pass just wrote
        hipLaunchKernelGGL(step_edges_kernel, dim3(1), dim3(256), 0, st, (const unsigned short *)b->d_steps, (unsigned)nnodes, b->d_step_edges);
        b->steps_valid = true;
    }
    // (the queue head, work[0], is reset by retry_list_kernel<kPixRetry> -- nothing else touches it in that launch)
    {
        KTimed kt("retry_list_kernel", st);
        const dim3 lgrid((unsigned)((nnodes + 256 * kRetryListItems - 1) / (256 * kRetryListItems)));
        hipLaunchKernelGGL(retry_list_kernel<kPixRetry>, lgrid, eblock, 0, st, (const int32_t *)d_pix,
                           (unsigned)nnodes, b->d_retry_nodes, b->d_retry_mask, b->d_work);
    }
    unsigned *d_hist = nullptr;
    if (use_witness && b->stats_on && std::getenv("UPSP_DEBUG_HIST")) {
        UPSP_HIP_CHECK(hipMalloc(&d_hist, 64 * sizeof(unsigned)));
        UPSP_HIP_CHECK(hipMemsetAsync(d_hist, 0, 64 * sizeof(unsigned), st));
        sc1.hist = d_hist;
    }
    if (use_witness) {
        {
            KTimed kt("witness_kernels", st);
            hipLaunchKernelGGL(witness_kernel, dim3((unsigned)((nnodes + kWitNodes - 1) / kWitNodes)), dim3(64), 0,
                               st, sc1, c, d_nodes, (const unsigned *)b->d_retry_nodes, b->d_todo_mask,
                               (const unsigned *)b->d_work);
            hipLaunchKernelGGL(todo_list_kernel, dim3((unsigned)((nnodes + 1023) / 1024)), dim3(256), 0, st,
                               (const unsigned *)b->d_todo_mask, b->d_todo_rays, b->d_work);
        }
        Scene sc2 = sc1;
        sc2.desc_cap = 4;       // (residual retries: 0.179 -> 0.145 ms at 4 rounds in round 2)
        sc2.chunk = 64;   // few rays are left (~3.5 %; 16 lanes per wave and 16-ray chunks: 0.37 instead of 0.21 ms)
        // the few thousand residual rays over ONE wave per SIMD, not three: alone the pass takes 8 % longer (4 rays to a wave instead
        // of 1-2, each wave as long as its longest chain), but in the frame loop's schedule it runs beside pass A, which needs its
        // eight workgroups per compute unit -- pass A 0.376 -> 0.361 ms, the step 0.810 -> 0.798 (256 or 96 waves in all: the same step)
        sc2.pack_waves = 4u * (unsigned)(props().cus > 0 ? props().cus : 256);
        {
            KTimed kt("projection_kernel<retry>", st);
            if (b->stats_on) UPSP_LAUNCH_PROJ(true, 2, grid1, sc2); else UPSP_LAUNCH_PROJ(false, 2, grid1, sc2);
        }
        UPSP_LAUNCH_HEAVY(2, sc2)
        if (d_hist) {   // debug: (node visits + triangle tests) per residual ray, 16 per bin; witness verdicts per retry ray
            unsigned h[64];
            UPSP_HIP_CHECK(hipMemcpy(h, d_hist, sizeof(h), hipMemcpyDeviceToHost));
            (void)hipFree(d_hist);
            std::fprintf(stderr, "upsp residual-ray steps histogram (bin = 16 steps):");
            for (int i = 0; i < 48; ++i) std::fprintf(stderr, " %u", h[i]);
            std::fprintf(stderr, "\nupsp witness verdicts: root box missed %u, adjacency unknown %u, no own triangle hit %u, "
                         "no witness slot %u, witness missed %u, witness behind the node %u, chain rejected %u, decided %u\n",
                         h[48], h[49], h[50], h[51], h[52], h[53], h[54], h[55]);
        }
    } else {
        {
            KTimed kt("projection_kernel<retry>", st);
            if (b->stats_on) UPSP_LAUNCH_PROJ(true, 1, grid1, sc1); else UPSP_LAUNCH_PROJ(false, 1, grid1, sc1);
        }
        UPSP_LAUNCH_HEAVY(1, sc1)
    }
#undef UPSP_LAUNCH_PROJ
#undef UPSP_LAUNCH_HEAVY
    {
        KTimed ktf("projection_finish_kernels", st);
        hipLaunchKernelGGL(projection_retry_outcome_kernel, dim3(256), dim3(256), 0, st, d_pix,
                           (const unsigned *)b->d_retry_nodes, (const unsigned *)b->d_retry_mask,
                           b->d_work);
        hipLaunchKernelGGL(projection_finish_kernel, egrid, eblock, 0, st, c, d_nodes, d_normals,
                           (unsigned)nnodes, oblique_thresh, d_pix, d_uv);
    }
    UPSP_HIP_CHECK(hipGetLastError());

    if (d_nodecount) {
        const unsigned npix = (unsigned)c.W * (unsigned)c.H;
        unsigned *counts = nullptr;
        UPSP_HIP_CHECK(hipMalloc(&counts, (size_t)npix * 4));
        hipError_t e = hipMemsetAsync(counts, 0, (size_t)npix * 4, st);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(nodecount_kernel, dim3(((unsigned)nnodes + 255) / 256), dim3(256),
                               0, st, d_pix, (unsigned)nnodes, counts);
            hipLaunchKernelGGL(saturate_kernel, dim3((npix + 255) / 256), dim3(256), 0, st, counts,
                               npix, d_nodecount);
            e = hipStreamSynchronize(st);
        }
        (void)hipFree(counts);
        if (e != hipSuccess) return fail(UPSP_ERR_HIP, hipGetErrorString(e));
    }
    if (adapt) {
        if (!b->h_handoff) {
            UPSP_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&b->h_handoff), 8 * sizeof(uint32_t), hipHostMallocDefault));
            UPSP_HIP_CHECK(hipEventCreateWithFlags(&b->ev_handoff, hipEventDisableTiming));
        }
        if (!b->handoff_pending || hipEventQuery(b->ev_handoff) == hipSuccess) {      // (never two copies into the words at once)
            UPSP_HIP_CHECK(hipMemcpyAsync(b->h_handoff, b->d_work + kWorkHeavyCount, 5 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            UPSP_HIP_CHECK(hipEventRecord(b->ev_handoff, st));
            b->handoff_pending = true;
            b->handoff_cam = *cam;
            b->handoff_nodes = d_nodes;
            b->handoff_nnodes = nnodes;
            b->handoff_was_on = heavy_launch;
        }
    } else {
        b->handoff_pending = false;
    }
    if (h_nrays || b->stats_on) {
        int rc = read_stats(b, st);
        if (rc != UPSP_OK) return rc;
        if (h_nrays) *h_nrays = b->last_stats[2];
        unsigned cnt[2] = {0, 0};
        UPSP_HIP_CHECK(hipMemcpy(cnt, b->d_work + kWorkRetryCount, sizeof(cnt), hipMemcpyDeviceToHost));
        b->last_retry_nodes = cnt[0];
        b->last_primary = cnt[1];
    }
    return UPSP_OK;
}

static int candidate_pixels_impl(const upsp_camera *cam, const float *d_nodes, const float *d_normals, const uint8_t *d_datanode,
                                 size_t nnodes, float oblique_thresh, int32_t *d_pix, void *stream);

int upsp_projection_candidate_pixels(const upsp_camera *cam, const float *d_nodes, const uint8_t *d_datanode,
                                     size_t nnodes, int32_t *d_pix, void *stream)
{
    return candidate_pixels_impl(cam, d_nodes, nullptr, d_datanode, nnodes, 0.0f, d_pix, stream);
}

int upsp_projection_candidate_pixels_oblique(const upsp_camera *cam, const float *d_nodes, const float *d_normals,
                                             const uint8_t *d_datanode, size_t nnodes, float oblique_thresh, int32_t *d_pix,
                                             void *stream)
{
    if (!d_normals) return fail(UPSP_ERR_INVALID, "null normals");
    return candidate_pixels_impl(cam, d_nodes, d_normals, d_datanode, nnodes, oblique_thresh, d_pix, stream);
}

static int candidate_pixels_impl(const upsp_camera *cam, const float *d_nodes, const float *d_normals, const uint8_t *d_datanode,
                                 size_t nnodes, float oblique_thresh, int32_t *d_pix, void *stream)
{
    if (!cam || !d_nodes || !d_pix) return fail(UPSP_ERR_INVALID, "null argument");
    if (cam->width <= 0 || cam->height <= 0) return fail(UPSP_ERR_INVALID, "bad image size");
    if (nnodes == 0) return UPSP_OK;
    if (nnodes > 0xF0000000ull) return fail(UPSP_ERR_INVALID, "too many nodes");
    Cam c;
    std::memcpy(c.K, cam->K, sizeof(c.K));
    std::memcpy(c.dist, cam->dist, sizeof(c.dist));
    std::memcpy(c.R, cam->R, sizeof(c.R));
    std::memcpy(c.t, cam->t, sizeof(c.t));
    double cc[3];
    upsp_camera_center(cam, cc);        // (the oblique test takes the camera -> node direction, like the build)
    c.ox = (float)cc[0];
    c.oy = (float)cc[1];
    c.oz = (float)cc[2];
    c.W = cam->width;
    c.H = cam->height;
    hipLaunchKernelGGL(candidate_pixels_kernel, dim3((unsigned)((nnodes + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       c, d_nodes, d_datanode, (unsigned)nnodes, d_normals, oblique_thresh, d_pix);
    UPSP_HIP_CHECK(hipGetLastError());
    return UPSP_OK;
}

}  // extern "C"
