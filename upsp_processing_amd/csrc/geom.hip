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

#include <algorithm>
#include <cmath>
#include <vector>

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

// ---------------------------------------------------------------------------------------------
// upsp::interpolate (cpp/lib/interpolation.ipp:16-70): values on the nodes of one grid carried to
// the nodes of another by inverse-distance weighting over the k nearest source nodes
// (nearest_k_neighbors, cpp/lib/models.ipp:503-571: best-first octree search, neighbours come out
// in ascending distance).  psp_process uses it to put a structured steady-state Cp / temperature
// solution onto an unstructured model grid (k = 10, p = 2; psp_process.cpp:2341-2344, 2374-2377).
//
// Source nodes are binned into a uniform grid on the host (counting sort, once); one query per
// lane walks cubic shells of cells outwards and keeps the k best candidates sorted in registers;
// it stops when the k-th distance cannot be beaten by any unvisited shell.
namespace upsp {
namespace {

constexpr int kMaxK = 16;

struct CellGrid {
    float lo[3], inv_cell;
    int dim[3];
    const unsigned *cell_start;   // [ncells + 1]
    const unsigned *cell_pts;     // source node ids, grouped by cell
};

__global__ void __launch_bounds__(256)
    idw_kernel(CellGrid g, const float *__restrict__ src_nodes, const float *__restrict__ src_data,
               const float *__restrict__ q_nodes, unsigned nq, int k, float p, float *__restrict__ out,
               int32_t *__restrict__ nbr_out)
{
    const unsigned q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const float px = q_nodes[3 * (size_t)q], py = q_nodes[3 * (size_t)q + 1], pz = q_nodes[3 * (size_t)q + 2];
    double bd[kMaxK];
    int bi[kMaxK];
    int have = 0;
    const float cell = 1.0f / g.inv_cell;
    int c[3];
    c[0] = min(max((int)floorf((px - g.lo[0]) * g.inv_cell), 0), g.dim[0] - 1);
    c[1] = min(max((int)floorf((py - g.lo[1]) * g.inv_cell), 0), g.dim[1] - 1);
    c[2] = min(max((int)floorf((pz - g.lo[2]) * g.inv_cell), 0), g.dim[2] - 1);
    // distance from the query to the border of its (clamped) cell block of radius r is at least
    // r * cell - (offset inside / outside the home cell); use the conservative bound below
    const float off = fmaxf(fmaxf(fabsf(px - (g.lo[0] + (c[0] + 0.5f) * cell)), fabsf(py - (g.lo[1] + (c[1] + 0.5f) * cell))),
                            fabsf(pz - (g.lo[2] + (c[2] + 0.5f) * cell)));
    const int rmax = max(max(g.dim[0], g.dim[1]), g.dim[2]);
    for (int r = 0; r <= rmax; ++r) {
        for (int dz = -r; dz <= r; ++dz) {
            const int z = c[2] + dz;
            if (z < 0 || z >= g.dim[2]) continue;
            for (int dy = -r; dy <= r; ++dy) {
                const int y = c[1] + dy;
                if (y < 0 || y >= g.dim[1]) continue;
                const bool face = (abs(dz) == r) || (abs(dy) == r);
                for (int dx = -r; dx <= r; dx += (face ? 1 : 2 * r > 0 ? 2 * r : 1)) {   // shell cells only
                    const int x = c[0] + dx;
                    if (x < 0 || x >= g.dim[0]) continue;
                    const size_t ci = ((size_t)z * g.dim[1] + y) * g.dim[0] + x;
                    for (unsigned e = g.cell_start[ci]; e < g.cell_start[ci + 1]; ++e) {
                        const int n = (int)g.cell_pts[e];
                        // cv::norm(pos - node): double sqrt of the float differences' squares
                        const float ddx = px - src_nodes[3 * (size_t)n], ddy = py - src_nodes[3 * (size_t)n + 1],
                                    ddz = pz - src_nodes[3 * (size_t)n + 2];
                        const double d = sqrt((double)ddx * ddx + (double)ddy * ddy + (double)ddz * ddz);
                        if (have < k || d < bd[have - 1] || (d == bd[have - 1] && n < bi[have - 1])) {
                            int j = have < k ? have : k - 1;      // insertion, ascending (distance, index)
                            while (j > 0 && (bd[j - 1] > d || (bd[j - 1] == d && bi[j - 1] > n))) {
                                bd[j] = bd[j - 1];
                                bi[j] = bi[j - 1];
                                --j;
                            }
                            bd[j] = d;
                            bi[j] = n;
                            if (have < k) ++have;
                        }
                    }
                }
            }
        }
        // every unvisited point is farther than this from the query
        const float reach = (r + 0.5f) * cell - off;
        if (have == k && bd[k - 1] <= (double)reach) break;
    }
    // inverse-distance weighting in float, neighbours in ascending distance (interpolation.ipp:46-64)
    float acc = 0.0f, total = 0.0f;
    for (int j = 0; j < have; ++j) {
        const float dist = (float)bd[j];
        if (dist == 0.0f) {
            total = 1.0f;
            acc = src_data[bi[j]];
            break;
        }
        // pow(dist, 2) is dist * dist rounded once -- what a correctly rounded pow returns
        const float pw = p == 2.0f ? dist * dist : powf(dist, p);
        const float weight = (float)(1.0 / (double)pw);
        acc += src_data[bi[j]] * weight;
        total += weight;
    }
    out[q] = acc / total;
    if (nbr_out)
        for (int j = 0; j < k; ++j) nbr_out[(size_t)q * k + j] = j < have ? bi[j] : -1;
}

}  // namespace
}  // namespace upsp

extern "C" int upsp_interpolate_idw(const float *h_src_nodes3, const float *h_src_data, size_t nsrc,
                                    const float *d_query_nodes3, size_t nquery, int k, float p,
                                    float *d_out, int32_t *d_neighbors, void *stream)
{
    if (nquery == 0) return UPSP_OK;
    if (!h_src_nodes3 || !h_src_data || !d_query_nodes3 || !d_out) return fail(UPSP_ERR_INVALID, "null buffer");
    if (nsrc == 0) return fail(UPSP_ERR_INVALID, "no source nodes");
    if (k < 1 || k > kMaxK) return fail(UPSP_ERR_INVALID, "k must be in [1,16]");
    if (nsrc > 0x7ffffff0ull || nquery > 0xfffffff0ull) return fail(UPSP_ERR_INVALID, "too many nodes");
    hipStream_t st = (hipStream_t)stream;
    // uniform grid with ~4 source nodes per cell on average (cubic cells)
    float lo[3] = {h_src_nodes3[0], h_src_nodes3[1], h_src_nodes3[2]}, hi[3] = {lo[0], lo[1], lo[2]};
    for (size_t i = 0; i < nsrc; ++i)
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], h_src_nodes3[3 * i + a]);
            hi[a] = std::max(hi[a], h_src_nodes3[3 * i + a]);
        }
    double ext[3];
    for (int a = 0; a < 3; ++a) ext[a] = std::max((double)hi[a] - lo[a], 1e-6);
    // surface grids are ~2-D: size the cells from the two largest extents
    double e_sorted[3] = {ext[0], ext[1], ext[2]};
    std::sort(e_sorted, e_sorted + 3);
    const double area = e_sorted[2] * e_sorted[1];
    float cell = (float)std::sqrt(area * 4.0 / (double)nsrc);
    if (!(cell > 0)) cell = 1.0f;
    CellGrid g;
    size_t ncells = 1;
    for (;;) {
        ncells = 1;
        for (int a = 0; a < 3; ++a) {
            g.dim[a] = std::max(1, (int)std::floor(ext[a] / cell) + 1);
            ncells *= (size_t)g.dim[a];
        }
        if (ncells <= (size_t)64 * 1024 * 1024) break;
        cell *= 1.26f;
    }
    for (int a = 0; a < 3; ++a) g.lo[a] = lo[a];
    g.inv_cell = 1.0f / cell;
    std::vector<unsigned> start(ncells + 1, 0u), pts(nsrc), cell_of(nsrc);
    for (size_t i = 0; i < nsrc; ++i) {
        size_t ci = 0, mul = 1;
        for (int a = 0; a < 3; ++a) {
            const int c = std::min(std::max((int)std::floor((h_src_nodes3[3 * i + a] - lo[a]) * g.inv_cell), 0), g.dim[a] - 1);
            ci += (size_t)c * mul;
            mul *= (size_t)g.dim[a];
        }
        cell_of[i] = (unsigned)ci;
        ++start[ci + 1];
    }
    for (size_t c = 0; c < ncells; ++c) start[c + 1] += start[c];
    {
        std::vector<unsigned> fill(start.begin(), start.end() - 1);
        for (size_t i = 0; i < nsrc; ++i) pts[fill[cell_of[i]]++] = (unsigned)i;   // ascending ids inside a cell
    }
    unsigned *d_start = nullptr, *d_pts = nullptr;
    float *d_src = nullptr, *d_data = nullptr;
    hipError_t e = hipMalloc(&d_start, start.size() * sizeof(unsigned));
    if (e == hipSuccess) e = hipMalloc(&d_pts, pts.size() * sizeof(unsigned));
    if (e == hipSuccess) e = hipMalloc(&d_src, nsrc * 3 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&d_data, nsrc * sizeof(float));
    if (e == hipSuccess) e = hipMemcpyAsync(d_start, start.data(), start.size() * sizeof(unsigned), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_pts, pts.data(), pts.size() * sizeof(unsigned), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_src, h_src_nodes3, nsrc * 3 * sizeof(float), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_data, h_src_data, nsrc * sizeof(float), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        g.cell_start = d_start;
        g.cell_pts = d_pts;
        KTimed kt("idw_kernel", st);
        hipLaunchKernelGGL(idw_kernel, dim3((unsigned)((nquery + 255) / 256)), dim3(256), 0, st, g,
                           (const float *)d_src, (const float *)d_data, d_query_nodes3, (unsigned)nquery, k, p,
                           d_out, d_neighbors);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);    // host vectors and temporaries go away
    if (d_start) (void)hipFree(d_start);
    if (d_pts) (void)hipFree(d_pts);
    if (d_src) (void)hipFree(d_src);
    if (d_data) (void)hipFree(d_data);
    if (e != hipSuccess) return fail(UPSP_ERR_HIP, hipGetErrorString(e));
    return UPSP_OK;
}
