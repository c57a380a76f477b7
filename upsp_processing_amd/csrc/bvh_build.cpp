// Host-side BVH construction for the MI355X ray caster.
//
// The tree is the reference's tree: top-down SAH with 12 centroid buckets on the
// widest centroid axis, leaves of <= 4 triangles (or any size when the centroid
// extent is zero) -- rt::BVH::recursiveBuild, cpp/raycast/pspRT.cpp:456-572 --
// because closest-hit ties are broken by traversal order (strict `<`,
// pspRT.cpp:395) and the engine must return the same primID as the reference.
// Only the memory layout is new: interior nodes become 64-byte records holding
// both child boxes (upsp_internal.h), triangles become 48-byte records stored in
// leaf order, so that one wave-wide fetch serves both box tests of a step.
//
// Works on structure-of-arrays primitive data and an index permutation; the
// permutation is split in place with the same element order libstdc++'s
// std::partition produces (pspRT.cpp:545-553), which fixes the order of the
// triangles inside a leaf.
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "upsp_internal.h"

namespace upsp {
namespace {

struct Box {
    float lo[3], hi[3];
    void clear()
    {
        for (int a = 0; a < 3; ++a) {
            lo[a] = FLT_MAX;
            hi[a] = -FLT_MAX;
        }
    }
    void grow(const float *p)
    {
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], p[a]);
            hi[a] = std::max(hi[a], p[a]);
        }
    }
    void grow(const Box &b)
    {
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], b.lo[a]);
            hi[a] = std::max(hi[a], b.hi[a]);
        }
    }
    bool empty() const { return hi[0] < lo[0] || hi[1] < lo[1] || hi[2] < lo[2]; }
    // Imath::Box::size(): zero vector for an empty box
    float extent(int a) const { return empty() ? 0.0f : hi[a] - lo[a]; }
    int widest() const
    {
        int m = 0;
        for (int a = 1; a < 3; ++a)
            if (extent(a) > extent(m)) m = a;
        return m;
    }
    // rt::SurfaceArea (pspRT.cpp:266-270)
    float area() const
    {
        float dx = extent(0), dy = extent(1), dz = extent(2);
        return 2 * (dx * dy + dx * dz + dy * dz);
    }
};

// Per-triangle data shared by every builder (read-only) + the permutation they split in place
// (each builder works on its own disjoint range of it).
struct Prims {
    const float *soup;
    std::vector<Box> pbox;       // per input triangle
    std::vector<float> cen;      // 3 per input triangle
    std::vector<uint32_t> perm;  // working permutation of triangle ids

    Prims(const float *s, size_t n) : soup(s), pbox(n), cen(3 * n), perm(n)
    {
        for (size_t i = 0; i < n; ++i) {
            Box &b = pbox[i];
            b.clear();
            b.grow(s + 9 * i);
            b.grow(s + 9 * i + 3);
            b.grow(s + 9 * i + 6);
            for (int a = 0; a < 3; ++a) cen[3 * i + a] = .5f * b.lo[a] + .5f * b.hi[a];
            perm[i] = (uint32_t)i;
        }
    }
};

// A subtree whose construction is deferred to a worker thread: the skeleton builder only needs
// its bounds (for the parent's record) and leaves a placeholder child reference behind.
struct Job {
    uint32_t lo, hi, depth;
    uint32_t parent;   // node of the skeleton holding the placeholder
    int slot;          // 0 = left, 1 = right, -1 = the whole tree is this one job
};

struct Builder {
    const float *soup;
    const std::vector<Box> &pbox;
    const std::vector<float> &cen;
    std::vector<uint32_t> &perm;
    HostBvh &out;
    std::vector<Job> *jobs = nullptr;   // non-null: defer subtrees of <= defer_below triangles
    uint32_t defer_below = 0;
    static constexpr int kBuckets = 12;
    static constexpr int kLeafPrims = 4;

    Builder(Prims &p, HostBvh &o) : soup(p.soup), pbox(p.pbox), cen(p.cen), perm(p.perm), out(o) {}

    int bucket(const Box &cb, uint32_t id, int dim) const
    {
        // rt::Offset(centroidBounds, centroid)[dim] (pspRT.cpp:257-264)
        float o = cen[3 * (size_t)id + dim] - cb.lo[dim];
        if (cb.hi[dim] > cb.lo[dim]) o /= cb.hi[dim] - cb.lo[dim];
        int b = (int)(kBuckets * o);
        return b == kBuckets ? kBuckets - 1 : b;
    }

    // Emits triangles [lo,hi) of perm as one or more device leaves; returns child ref.
    int32_t emit_leaf(uint32_t lo, uint32_t hi, const Box &box, uint32_t depth)
    {
        uint32_t n = hi - lo;
        out.max_leaf = std::max(out.max_leaf, n);
        out.depth = std::max(out.depth, depth);
        if (n <= (uint32_t)kMaxLeaf) {
            uint32_t first = (uint32_t)out.tris.size();
            for (uint32_t i = lo; i < hi; ++i) {
                uint32_t id = perm[i];
                GpuTri t;
                std::memset(&t, 0, sizeof(t));
                std::memcpy(t.a, soup + 9 * (size_t)id, 12);
                std::memcpy(t.b, soup + 9 * (size_t)id + 3, 12);
                std::memcpy(t.c, soup + 9 * (size_t)id + 6, 12);
                t.prim = (int32_t)id;
                out.tris.push_back(t);
            }
            return ~(int32_t)((first << kLeafBits) | (n - 1));
        }
        // oversized leaf (zero centroid extent, pspRT.cpp:485-494): ordered chain of
        // pseudo-nodes whose child boxes are the leaf box, preserving the order.
        uint32_t me = (uint32_t)out.nodes.size();
        out.nodes.emplace_back();
        int32_t l = emit_leaf(lo, lo + kMaxLeaf, box, depth + 1);
        int32_t r = emit_leaf(lo + kMaxLeaf, hi, box, depth + 1);
        write_node(me, box, box, l, r, kMetaOrdered);
        return (int32_t)me;
    }

    void write_node(uint32_t idx, const Box &L, const Box &R, int32_t l, int32_t r, uint32_t meta)
    {
        float *q = out.nodes[idx].q;
        q[0] = L.lo[0]; q[1] = L.lo[1]; q[2] = L.lo[2]; q[3] = L.hi[0];
        q[4] = L.hi[1]; q[5] = L.hi[2]; q[6] = R.lo[0]; q[7] = R.lo[1];
        q[8] = R.lo[2]; q[9] = R.hi[0]; q[10] = R.hi[1]; q[11] = R.hi[2];
        std::memcpy(&q[12], &l, 4);
        std::memcpy(&q[13], &r, 4);
        std::memcpy(&q[14], &meta, 4);
        q[15] = 0.0f;
    }

    // Returns the child ref of the subtree over perm[lo,hi) and its bounds.
    // child reference that stands for deferred job j until the subtrees are stitched in
    static int32_t placeholder(size_t j) { return (int32_t)(0x40000000u | (uint32_t)j); }

    int32_t build(uint32_t lo, uint32_t hi, uint32_t depth, Box &bounds)
    {
        bounds.clear();
        for (uint32_t i = lo; i < hi; ++i) bounds.grow(pbox[perm[i]]);
        uint32_t n = hi - lo;
        if (jobs && n <= defer_below && n > (uint32_t)kLeafPrims) {
            jobs->push_back(Job{lo, hi, depth, 0u, 0});
            return placeholder(jobs->size() - 1);
        }
        out.n_ref_nodes++;
        if (n <= (uint32_t)kLeafPrims) return emit_leaf(lo, hi, bounds, depth);

        Box cb;
        cb.clear();
        for (uint32_t i = lo; i < hi; ++i) cb.grow(&cen[3 * (size_t)perm[i]]);
        int dim = cb.widest();
        if (cb.hi[dim] == cb.lo[dim]) return emit_leaf(lo, hi, bounds, depth);

        int cnt[kBuckets] = {0};
        Box bb[kBuckets];
        for (auto &b : bb) b.clear();
        for (uint32_t i = lo; i < hi; ++i) {
            int b = bucket(cb, perm[i], dim);
            cnt[b]++;
            bb[b].grow(pbox[perm[i]]);
        }
        // SAH cost of splitting after each bucket (pspRT.cpp:513-536)
        float best_cost = 0.0f;
        int best = 0;
        const float parent_area = bounds.area();
        for (int s = 0; s < kBuckets - 1; ++s) {
            Box b0, b1;
            b0.clear();
            b1.clear();
            int c0 = 0, c1 = 0;
            for (int j = 0; j <= s; ++j) {
                b0.grow(bb[j]);
                c0 += cnt[j];
            }
            for (int j = s + 1; j < kBuckets; ++j) {
                b1.grow(bb[j]);
                c1 += cnt[j];
            }
            float cost = 1.f + ((float)c0 * b0.area() + (float)c1 * b1.area()) / parent_area;
            if (s == 0 || cost < best_cost) {
                best_cost = cost;
                best = s;
            }
        }

        // in-place split, element order of libstdc++ std::partition (bidirectional)
        uint32_t f = lo, l = hi;
        while (true) {
            while (f != l && bucket(cb, perm[f], dim) <= best) ++f;
            if (f == l) break;
            --l;
            while (f != l && !(bucket(cb, perm[l], dim) <= best)) --l;
            if (f == l) break;
            std::swap(perm[f], perm[l]);
            ++f;
        }
        uint32_t mid = f;
        if (mid == lo || mid == hi) return emit_leaf(lo, hi, bounds, depth);  // non-finite input

        uint32_t me = (uint32_t)out.nodes.size();
        out.nodes.emplace_back();
        Box L, R;
        const size_t j0 = jobs ? jobs->size() : 0;
        int32_t lref = build(lo, mid, depth + 1, L);
        const size_t j1 = jobs ? jobs->size() : 0;
        int32_t rref = build(mid, hi, depth + 1, R);
        if (jobs) {   // remember where the placeholders of directly deferred children live
            if (j1 == j0 + 1 && lref == placeholder(j0)) { (*jobs)[j0].parent = me; (*jobs)[j0].slot = 0; }
            if (jobs->size() == j1 + 1 && rref == placeholder(j1)) { (*jobs)[j1].parent = me; (*jobs)[j1].slot = 1; }
        }
        write_node(me, L, R, lref, rref, (uint32_t)dim);
        return (int32_t)me;
    }
};

}  // namespace

// The first `top` interior nodes are renumbered in breadth-first order (the rest keep
// their depth-first order behind them), so the upper levels of the tree -- the nodes every
// ray visits -- form one contiguous block of cache lines.
// Only indices change; topology, child order and leaf order are untouched.
static void relabel_top_breadth_first(HostBvh &out, uint32_t top)
{
    const uint32_t n = (uint32_t)out.nodes.size();
    if (n == 0 || out.root_ref < 0) return;
    top = std::min(top, n);
    std::vector<uint32_t> order;   // new index -> old index
    std::vector<int32_t> newid(n, -1);
    order.reserve(n);
    auto child = [&](uint32_t node, int k) {
        int32_t r;
        std::memcpy(&r, &out.nodes[node].q[12 + k], 4);
        return r;
    };
    order.push_back((uint32_t)out.root_ref);
    newid[out.root_ref] = 0;
    for (size_t head = 0; head < order.size() && order.size() < top; ++head)
        for (int k = 0; k < 2; ++k) {
            const int32_t c = child(order[head], k);
            if (c >= 0 && newid[c] < 0 && order.size() < top) {
                newid[c] = (int32_t)order.size();
                order.push_back((uint32_t)c);
            }
        }
    out.top_nodes = (uint32_t)order.size();
    for (uint32_t i = 0; i < n; ++i)
        if (newid[i] < 0) {
            newid[i] = (int32_t)order.size();
            order.push_back(i);
        }
    std::vector<GpuNode> renum(n);
    for (uint32_t ni = 0; ni < n; ++ni) {
        GpuNode g = out.nodes[order[ni]];
        for (int k = 0; k < 2; ++k) {
            int32_t r;
            std::memcpy(&r, &g.q[12 + k], 4);
            if (r >= 0) {
                r = newid[r];
                std::memcpy(&g.q[12 + k], &r, 4);
            }
        }
        renum[ni] = g;
    }
    out.nodes.swap(renum);
    out.root_ref = newid[out.root_ref];
}

// Two levels of the binary tree per record (GpuWide): a wide node stands for a binary interior node the traversal enters
// -- the root and every interior grandchild, great-great-grandchild, ... -- and lists, for each of that node's two
// children, the child itself (a leaf) or the child's two children.  Topology, boxes, leaf order and the per-node split
// axes are copied, nothing is recomputed: the set of leaves a ray reaches and the order it reaches them in are the
// binary tree's.
static void collapse_wide(HostBvh &out)
{
    out.wide.clear();
    out.wide_depth = 0;
    if (out.root_ref < 0 || out.nodes.empty()) {
        out.wide_root = out.root_ref;
        return;
    }
    out.wide.reserve(out.nodes.size() / 2 + 16);
    struct Item { int32_t node; uint32_t self, depth; };
    std::vector<Item> todo;
    auto child = [&](int32_t n, int k) { int32_t r; std::memcpy(&r, &out.nodes[n].q[12 + k], 4); return r; };
    auto meta_of = [&](int32_t n) { uint32_t m; std::memcpy(&m, &out.nodes[n].q[14], 4); return m & 7u; };
    auto new_wide = [&](int32_t n, uint32_t depth) {
        const uint32_t idx = (uint32_t)out.wide.size();
        out.wide.emplace_back();
        todo.push_back({n, idx, depth});
        return idx;
    };
    out.wide_root = (int32_t)new_wide(out.root_ref, 1);
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        out.wide_depth = std::max(out.wide_depth, it.depth);
        float q[32];
        std::memset(q, 0, sizeof(q));
        int32_t refs[4] = {kWideEmpty, kWideEmpty, kWideEmpty, kWideEmpty};
        int32_t pending[4] = {-1, -1, -1, -1};       // binary interior node behind a slot (gets its own wide node)
        uint32_t meta = meta_of(it.node);
        // a slot that does not exist: an inverted box (the box test rejects it on every axis) and the empty reference
        for (int s = 0; s < 4; ++s) {
            float *b = q + 12 * (s / 2) + 6 * (s % 2);
            b[0] = b[1] = b[2] = FLT_MAX;
            b[3] = b[4] = b[5] = -FLT_MAX;
        }
        for (int k = 0; k < 2; ++k) {
            const int32_t c = child(it.node, k);
            const float *cbox = &out.nodes[it.node].q[6 * k];     // (lo.xyz, hi.xyz) of child k
            if (c < 0) {                                           // leaf: the child itself, in the group's first slot
                std::memcpy(q + 12 * k, cbox, 6 * sizeof(float));
                refs[2 * k] = c;
            } else {                                               // interior: its two children
                meta |= meta_of(c) << (3 + 3 * k);
                std::memcpy(q + 12 * k, &out.nodes[c].q[0], 12 * sizeof(float));
                for (int j = 0; j < 2; ++j) {
                    const int32_t g = child(c, j);
                    if (g < 0) refs[2 * k + j] = g;
                    else pending[2 * k + j] = g;
                }
            }
        }
        // (the up to four wide children of a node get consecutive indices: one 512-byte stretch of memory)
        for (int s = 0; s < 4; ++s)
            if (pending[s] >= 0) refs[s] = (int32_t)new_wide(pending[s], it.depth + 1);
        std::memcpy(q + 24, refs, sizeof(refs));
        std::memcpy(q + 28, &meta, 4);
        std::memcpy(out.wide[it.self].q, q, sizeof(q));
    }
}

// The tree is the one the sequential recursion produces (same splits, same leaf order); only the
// numbering of nodes / triangle slots differs: the skeleton (upper levels, built first) comes
// first, the deferred subtrees follow in discovery order.  Nothing depends on the numbering.
void build_bvh(const float *tris9, size_t ntris, HostBvh &out)
{
    out = HostBvh();
    out.nodes.reserve(ntris / 2 + 16);
    out.tris.reserve(ntris);
    Prims prims(tris9, ntris);
    unsigned nthreads = std::thread::hardware_concurrency();
    if (const char *e = std::getenv("UPSP_BUILD_THREADS")) nthreads = (unsigned)std::max(1, std::atoi(e));
    nthreads = std::min(std::max(nthreads, 1u), 32u);
    Box root;
    if (nthreads <= 1 || ntris < 65536) {
        Builder b(prims, out);
        out.root_ref = b.build(0, (uint32_t)ntris, 0, root);
    } else {
        // skeleton: upper levels here, subtrees of <= ntris / (8 * threads) triangles deferred
        std::vector<Job> jobs;
        Builder top(prims, out);
        top.jobs = &jobs;
        top.defer_below = (uint32_t)std::max<size_t>(ntris / (8 * (size_t)nthreads), 1024);
        out.root_ref = top.build(0, (uint32_t)ntris, 0, root);
        std::vector<HostBvh> sub(jobs.size());
        std::vector<int32_t> sub_ref(jobs.size());
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            for (size_t j = next++; j < jobs.size(); j = next++) {
                Builder b(prims, sub[j]);
                Box bb;
                sub_ref[j] = b.build(jobs[j].lo, jobs[j].hi, jobs[j].depth, bb);
            }
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < nthreads; ++t) pool.emplace_back(worker);
        worker();
        for (auto &t : pool) t.join();
        // stitch: append every subtree, relocating its interior indices and leaf slots
        for (size_t j = 0; j < jobs.size(); ++j) {
            const uint32_t node_off = (uint32_t)out.nodes.size(), tri_off = (uint32_t)out.tris.size();
            auto reloc = [&](int32_t r) -> int32_t {
                if (r >= 0) return r + (int32_t)node_off;
                const uint32_t code = (uint32_t)(~r);
                return ~(int32_t)((((code >> kLeafBits) + tri_off) << kLeafBits) | (code & (kMaxLeaf - 1)));
            };
            for (GpuNode g : sub[j].nodes) {
                for (int k = 0; k < 2; ++k) {
                    int32_t r;
                    std::memcpy(&r, &g.q[12 + k], 4);
                    r = reloc(r);
                    std::memcpy(&g.q[12 + k], &r, 4);
                }
                out.nodes.push_back(g);
            }
            out.tris.insert(out.tris.end(), sub[j].tris.begin(), sub[j].tris.end());
            const int32_t ref = reloc(sub_ref[j]);
            if (jobs[j].slot < 0 || out.root_ref == Builder::placeholder(j)) {
                out.root_ref = ref;
            } else {
                std::memcpy(&out.nodes[jobs[j].parent].q[12 + jobs[j].slot], &ref, 4);
            }
            out.n_ref_nodes += sub[j].n_ref_nodes;
            out.depth = std::max(out.depth, sub[j].depth);
            out.max_leaf = std::max(out.max_leaf, sub[j].max_leaf);
            sub[j] = HostBvh();
        }
    }
    for (int a = 0; a < 3; ++a) {
        out.root_min[a] = root.lo[a];
        out.root_max[a] = root.hi[a];
    }
    relabel_top_breadth_first(out, kTopNodesMax);
    collapse_wide(out);
}

}  // namespace upsp
