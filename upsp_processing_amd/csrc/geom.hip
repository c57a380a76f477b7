// Nearest model node to a set of query points.
//
// The reference answers this with a kd-tree over all nodes (kd_nearest,
// cpp/raycast/pspKdtree.c:284-372; built by TriModel_::generate_kd_tree,
// cpp/lib/TriModel.ipp:915-937) and uses it only on a handful of points per camera:
// the hit position of every visible target (getTargets, psp_process.cpp:95-100) and the
// target centres (get_target_diameters, :136-141).  With tens of queries against ~1e6 nodes
// an exhaustive scan is one short HBM/L2-bound pass (12 B per node per query, the node array
// is read once from HBM and then served from L2 / Infinity Cache), needs no build step and no
// recursion: each workgroup scans a slice of the nodes for one query, a second tiny kernel
// merges the slices.
//
// Distances are the kd-tree's: sum over x,y,z of (double(node) - query)^2 in that order, in
// double (-ffp-contract=off), compared with strict '<'.  The minimum DISTANCE is therefore
// identical; the INDEX is identical unless two nodes are exactly equidistant, where the
// kd-tree returns whichever its traversal meets first and this scan returns the lowest index.
#include <hip/hip_runtime.h>

#include "ktimer.h"
#include "upsp_internal.h"

namespace upsp {
namespace {

constexpr int kSlices = 64;   // workgroups per query

struct Best {
    double d2;
    int idx;
};

__device__ __forceinline__ bool better(double d2, int idx, double bd2, int bidx)
{
    return d2 < bd2 || (d2 == bd2 && idx < bidx);
}

__global__ void __launch_bounds__(256)
    nearest_scan_kernel(const float *__restrict__ nodes, unsigned nnodes,
                        const double *__restrict__ query, Best *__restrict__ partial)
{
    const unsigned q = blockIdx.y;
    const double qx = query[3 * q], qy = query[3 * q + 1], qz = query[3 * q + 2];
    const unsigned per = (nnodes + kSlices - 1) / kSlices;
    const unsigned lo = blockIdx.x * per, hi = min(nnodes, lo + per);
    double bd2 = __builtin_inf();
    int bidx = 0x7fffffff;
    for (unsigned n = lo + threadIdx.x; n < hi; n += 256) {
        const double dx = (double)nodes[3 * (size_t)n] - qx;
        const double dy = (double)nodes[3 * (size_t)n + 1] - qy;
        const double dz = (double)nodes[3 * (size_t)n + 2] - qz;
        const double d2 = (0.0 + dx * dx + dy * dy) + dz * dz;   // kd_nearest_i order
        if (d2 < bd2) {   // ascending n per thread: strict '<' keeps the lowest index
            bd2 = d2;
            bidx = (int)n;
        }
    }
    // NaN distances never win (all comparisons false); an all-NaN slice reports idx = INT_MAX
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double od2 = __shfl_xor(bd2, off);
        const int oidx = __shfl_xor(bidx, off);
        if (better(od2, oidx, bd2, bidx)) {
            bd2 = od2;
            bidx = oidx;
        }
    }
    __shared__ Best wbest[4];
    if ((threadIdx.x & 63) == 0) wbest[threadIdx.x >> 6] = Best{bd2, bidx};
    __syncthreads();
    if (threadIdx.x == 0) {
        Best b = wbest[0];
        for (int w = 1; w < 4; ++w)
            if (better(wbest[w].d2, wbest[w].idx, b.d2, b.idx)) b = wbest[w];
        partial[(size_t)q * kSlices + blockIdx.x] = b;
    }
}

__global__ void __launch_bounds__(64)
    nearest_merge_kernel(const Best *__restrict__ partial, unsigned nq, int32_t *__restrict__ index,
                         double *__restrict__ dist2)
{
    const unsigned q = blockIdx.x;
    if (q >= nq) return;
    Best b = partial[(size_t)q * kSlices + threadIdx.x];   // kSlices == 64 == one wave
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double od2 = __shfl_xor(b.d2, off);
        const int oidx = __shfl_xor(b.idx, off);
        if (better(od2, oidx, b.d2, b.idx)) {
            b.d2 = od2;
            b.idx = oidx;
        }
    }
    if (threadIdx.x == 0) {
        index[q] = b.idx == 0x7fffffff ? -1 : b.idx;
        if (dist2) dist2[q] = b.d2;
    }
}

}  // namespace
}  // namespace upsp

using namespace upsp;

extern "C" int upsp_nearest_nodes(const float *d_nodes3, size_t nnodes, const double *d_query3,
                                  size_t nqueries, int32_t *d_index, double *d_dist2, void *stream)
{
    if (nqueries == 0) return UPSP_OK;
    if (!d_nodes3 || !d_query3 || !d_index) return fail(UPSP_ERR_INVALID, "null device buffer");
    if (nnodes == 0) return fail(UPSP_ERR_INVALID, "nearest-node query on an empty model");
    if (nnodes > 0x7ffffff0ull || nqueries > 65535) return fail(UPSP_ERR_INVALID, "too many nodes / queries");
    hipStream_t st = (hipStream_t)stream;
    Best *partial = nullptr;
    UPSP_HIP_CHECK(hipMallocAsync((void **)&partial, sizeof(Best) * nqueries * kSlices, st));
    {
        KTimed kt("nearest_nodes", st);
        hipLaunchKernelGGL(nearest_scan_kernel, dim3(kSlices, (unsigned)nqueries), dim3(256), 0, st,
                           d_nodes3, (unsigned)nnodes, d_query3, partial);
        hipLaunchKernelGGL(nearest_merge_kernel, dim3((unsigned)nqueries), dim3(64), 0, st, partial,
                           (unsigned)nqueries, d_index, d_dist2);
    }
    hipError_t e = hipGetLastError();
    (void)hipFreeAsync(partial, st);
    if (e != hipSuccess) return fail(UPSP_ERR_HIP, hipGetErrorString(e));
    return UPSP_OK;
}
