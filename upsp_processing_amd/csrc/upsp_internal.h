// Internal declarations shared by the host/HIP translation units of libupsp_gpu.so.
#ifndef UPSP_INTERNAL_H
#define UPSP_INTERNAL_H

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "upsp_gpu.h"

namespace upsp {

// ---- error plumbing --------------------------------------------------------
void set_error(const std::string &msg);
int fail(int status, const std::string &msg);

#define UPSP_HIP_CHECK(expr)                                                              \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess)                                                             \
            return ::upsp::fail(UPSP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

// ---- device BVH layout -----------------------------------------------------
// One 64-byte record per INTERIOR node of the reference's binary SAH tree; it
// carries the boxes of both children so one fetch decides both descents.
//   q0 = (lmin.x lmin.y lmin.z lmax.x)   q1 = (lmax.y lmax.z rmin.x rmin.y)
//   q2 = (rmin.z rmax.x rmax.y rmax.z)   q3 = (left, right, meta, 0) as int bits
// child ref >= 0 : interior node index; < 0 : leaf, ~ref = (first_slot << 6) | (count-1)
// meta bits 0-1 : split axis (LinearNode::axis, pspRT.h:79); bit 2 : ordered
// (always visit left first; used when an oversized leaf is split into a chain).
struct alignas(64) GpuNode {
    float q[16];
};
static_assert(sizeof(GpuNode) == 64, "GpuNode must be 64 bytes");

// The same tree with every other level folded away: one 128-byte record (one cache line) per node that a traversal
// ENTERS, holding the boxes and references of up to four descendants -- for each child of the binary node, the child
// itself when it is a leaf, its two children when it is interior.  A traversal step decides two levels of the
// reference's descent with one dependent fetch instead of two.
//   q[0..11]  boxes of slots 0, 1 (the left child's group), packed like GpuNode's pair;  q[12..23] slots 2, 3 (right group)
//   q[24..27] slot references as int bits (>= 0: wide node, < 0: leaf code as above, kWideEmpty: no such slot)
//   q[28]     meta as int bits: bits 0-2 the binary node's (axis | ordered << 2), bits 3-5 its left child's, 6-8 its right child's
// The reference visits the children of a node near-first by the sign of the ray direction along the node's split axis
// (pspRT.cpp:410-419); applied twice that fixes the order of the four slots (raycast.hip: wide_step).
struct alignas(128) GpuWide {
    float q[32];
};
static_assert(sizeof(GpuWide) == 128, "GpuWide must be 128 bytes");
constexpr int32_t kWideEmpty = (int32_t)0x7FFFFFFF;

// 48-byte triangle record in leaf order: (A, primID) (B, -) (C, -)
struct alignas(16) GpuTri {
    float a[3];
    int32_t prim;
    float b[3];
    int32_t pad0;
    float c[3];
    int32_t pad1;
};
static_assert(sizeof(GpuTri) == 48, "GpuTri must be 48 bytes");

constexpr int kLeafBits = 6;
constexpr int kMaxLeaf = 1 << kLeafBits;  // triangles addressable by one leaf ref
constexpr uint32_t kMetaOrdered = 4u;
constexpr uint32_t kTopNodesMax = 255;  // upper-tree nodes renumbered breadth-first (contiguous in memory)

struct HostBvh {
    std::vector<GpuNode> nodes;
    std::vector<GpuTri> tris;
    int32_t root_ref = 0;
    float root_min[3] = {0, 0, 0}, root_max[3] = {0, 0, 0};
    uint32_t n_ref_nodes = 0;
    uint32_t depth = 0;
    uint32_t max_leaf = 0;
    uint32_t top_nodes = 0;  // nodes [0, top_nodes) are the breadth-first upper tree
    std::vector<GpuWide> wide;   // the two-levels-per-step form of the same tree (collapse_wide)
    int32_t wide_root = 0;       // >= 0: index into wide; < 0: the root is a leaf (its code)
    uint32_t wide_depth = 0;     // levels of wide nodes on the longest path
};

// SAH build with the reference's topology and leaf order (pspRT.cpp:456-572).
void build_bvh(const float *tris9, size_t ntris, HostBvh &out);

}  // namespace upsp

struct upsp_bvh {
    upsp::GpuNode *d_nodes = nullptr;
    upsp::GpuWide *d_wide = nullptr;   // the same tree, two levels per record (the traversal kernels' layout)
    int32_t wide_root = 0;
    uint32_t n_wide = 0, wide_depth = 0;
    upsp::GpuTri *d_tris = nullptr;
    uint32_t *d_work = nullptr;   // [0] work-queue head, [2..7] 64-bit stats, [8..10] see raycast.hip
    uint32_t *d_err = nullptr;    // error word of the walks (round cap exceeded), behind the work words
    uint32_t *d_retry_nodes = nullptr, *d_retry_mask = nullptr;  // projection-build retry list
    std::vector<uint32_t> prim_slot;   // host: triangle slot (leaf order) of every input triangle
    // node -> adjacent triangle slots (CSR), set by upsp_bvh_set_tri_nodes (bounded visibility rays)
    uint32_t *d_adj_off = nullptr, *d_adj_slot = nullptr;
    // root-to-leaf box chain of every triangle slot (occluder witness test of the retry rays):
    // d_slot_path[slot] = (offset, length) into d_path_ref, entries (interior node << 1) | side
    uint32_t *d_slot_path = nullptr, *d_path_ref = nullptr;
    int32_t *d_witness = nullptr;      // per node: triangle slot the primary ray hit (retry nodes)
    uint32_t *d_todo_mask = nullptr, *d_todo_rays = nullptr;   // retries the witness test left undecided
    uint32_t *d_heavy = nullptr;                               // work items handed to heavy_kernel
    void *d_heavy_scratch = nullptr;                           // ... and the stacks of its cooperative walks (global memory)
    uint32_t *d_cast_list = nullptr;                           // batch queries: rays that enter the root box
    size_t cast_list_capacity = 0;
    const void *adj_src = nullptr;     // the d_tri_nodes buffer the adjacency was built from
    size_t adj_nnodes = 0;
    size_t retry_capacity = 0;
    void *d_stage = nullptr, *h_stage = nullptr;  // small host batches (pybind per-ray calls)
    struct ihipStream_t *stage_stream = nullptr;
    int32_t root_ref = 0;
    uint32_t top_nodes = 0;
    float root_min[3], root_max[3];
    upsp_bvh_info info;
    int device = 0;
    int stats_on = 0;
    bool batch_warmed = false;     // the first large batch query swept the tree through the caches (prefetch_bvh)
    uint64_t last_stats[3] = {0, 0, 0};
    uint64_t last_slab[2] = {0, 0};      // boxes the slab filter saw / left undecided (statistics on)
    uint64_t last_primary = 0, last_retry_nodes = 0;
    const upsp_bvh *shared_from = nullptr;   // upsp_bvh_share: the tree / adjacency arrays belong to that handle
    // Length-homogeneous waves of a REPEATED projection build (round 6): the step count of every node's primary ray as the
    // build before this one measured it (0 = no ray then), and the bin edges derived from a sample of them.  The next build's
    // dense ray list is binned by it, so that the 64 rays of a wave end together.  Ordering only -- results do not depend on it.
    // Hand-off machinery only when the last build of the SAME view needed it (round 6): the counts of rays handed to the one-ray-
    // per-wave / per-workgroup walks are read back without a wait (pinned words behind an event); when the previous build of this
    // handle with the same camera, nodes and node count handed off none, the next one runs without the hand-off threshold and
    // without the four (then empty) walk launches -- every ray finishes in its lane, same results.
    uint32_t *h_handoff = nullptr;           // pinned: work[16..20] of the last build
    struct ihipEvent_t *ev_handoff = nullptr;
    bool handoff_pending = false;
    upsp_camera handoff_cam;
    const void *handoff_nodes = nullptr;
    size_t handoff_nnodes = 0;
    bool handoff_was_on = false;
    unsigned handoff_skips = 0;
    uint16_t *d_steps = nullptr;
    uint32_t *d_step_edges = nullptr;
    size_t steps_nnodes = 0;
    bool steps_valid = false;
};

#endif
