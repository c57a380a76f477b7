// End-of-run exchanges of psp_process phase 1 between the ranks of one job, behind the C ABI:
//
//   * sum of the per-rank double accumulators      MPI_Reduce + MPI_Bcast, cpp/exec/psp_process.cpp:1866-1872, 2019-2023
//   * time-series exchange [frames_r x N] -> [nodes_r x F]   global_transpose, cpp/exec/psp_process.cpp:707-771
//     with apportion() (:611-624) deciding who owns which frames and nodes
//
// One process per GPU.  The transport is RCCL (the ROCm NCCL) over xGMI: every GPU pair has its own link, so the
// exchange of a chunk is ONE group of point-to-point sends / receives in which all links of a GPU carry exactly one
// block at the same time -- no ring, no staging through a root.  The rank's frames are produced in K chunks and the
// exchange of chunk k is in flight (on the exchange's own stream) while chunk k + 1 is processed.  Rows of nodes no
// camera sees are NaN in every frame on every rank (psp_process.cpp:1821-1825): they do not travel; the receiver
// fills them.  With one camera, no weights and no float image stage the travelling rows are exact 16-bit integers
// and go as u16 -- or, for 12-bit cameras, packed to 12 bits (3 bytes per 2 frames).
//
// librccl is NOT a link-time dependency: its entry points are taken from the RCCL the running process has already loaded
// (PyTorch brings its own copy -- into a local scope, so it is found by walking the loaded objects, not by a global symbol
// lookup -- and two RCCLs in one process are one too many), from librccl.so.1 otherwise, or from the build UPSP_RCCL_LIBRARY names.
//
// A second transport, "local", runs all ranks of a group inside ONE process on one GPU (device-to-device copies in
// place of the links): the pool this was built on has one GPU per box and RCCL refuses two ranks on one device, so
// this is how the bookkeeping of W > 1 ranks -- block offsets, ragged slices, chunk placement -- is executed on real
// device buffers by the tests (tests/cpp/exchange_test.cpp).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <link.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "ktimer.h"
#include "pipeline.h"
#include "upsp_gpu.h"
#include "upsp_internal.h"

using namespace upsp;

namespace {

// ---- RCCL entry points, resolved at first use ---------------------------------------------------------------
struct Rccl {
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    std::string why;
};

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        void *h = nullptr;
        // UPSP_RCCL_LIBRARY: the RCCL build to bind (a path for dlopen), used instead of whatever the process holds -- a site's own
        // build of librccl, or the tests' stand-in that lets several rank PROCESSES share one GPU (tests/shim/rccl_shim.cpp).
        // Loaded RTLD_LOCAL: its symbols serve this library only and do not interpose another RCCL in the process (PyTorch's).
        const char *forced = getenv("UPSP_RCCL_LIBRARY");
        if (forced && *forced) {
            h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
            if (!h) {
                const char *e = dlerror();
                r.why = std::string("UPSP_RCCL_LIBRARY=") + forced + ": " + (e ? e : "dlopen failed");
            }
        }
        const bool only_forced = forced && *forced;
        std::string loaded;                               // path of a librccl this process holds already, if any
        if (!only_forced)
            dl_iterate_phdr(
                [](struct dl_phdr_info *info, size_t, void *data) -> int {
                    if (info->dlpi_name && std::strstr(info->dlpi_name, "librccl.so")) {
                        *static_cast<std::string *>(data) = info->dlpi_name;
                        return 1;
                    }
                    return 0;
                },
                &loaded);
        auto sym = [&](const char *name) -> void * {
            if (only_forced) {
                void *p = h ? dlsym(h, name) : nullptr;
                if (!p && r.why.empty()) r.why = std::string("UPSP_RCCL_LIBRARY=") + forced + " has no symbol " + name;
                return p;
            }
            void *p = dlsym(RTLD_DEFAULT, name);          // in the global scope of the process?
            if (!p) {
                // An RCCL the process has ALREADY loaded (PyTorch loads its own copy into a local scope, where the lookup above
                // does not see it): the same instance serves both -- one set of proxy threads, one build.  Otherwise the
                // system's; RTLD_LOCAL either way (a second copy's symbols must not interpose the first's late bindings).
                if (!h && !loaded.empty()) h = dlopen(loaded.c_str(), RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);
                if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
                if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
                if (h) p = dlsym(h, name);
            }
            if (!p && r.why.empty()) r.why = std::string("RCCL symbol not found: ") + name;
            return p;
        };
#define UPSP_SYM(field, name) r.field = reinterpret_cast<decltype(r.field)>(sym(name))
        UPSP_SYM(GetUniqueId, "ncclGetUniqueId");
        UPSP_SYM(CommInitRank, "ncclCommInitRank");
        UPSP_SYM(CommDestroy, "ncclCommDestroy");
        UPSP_SYM(CommCount, "ncclCommCount");
        UPSP_SYM(CommUserRank, "ncclCommUserRank");
        UPSP_SYM(GroupStart, "ncclGroupStart");
        UPSP_SYM(GroupEnd, "ncclGroupEnd");
        UPSP_SYM(Send, "ncclSend");
        UPSP_SYM(Recv, "ncclRecv");
        UPSP_SYM(AllReduce, "ncclAllReduce");
        UPSP_SYM(GetErrorString, "ncclGetErrorString");
#undef UPSP_SYM
        r.ok = r.why.empty();
    });
    return r;
}

int nccl_fail(ncclResult_t e, const char *what)
{
    Rccl &r = rccl();
    return fail(UPSP_ERR_HIP, std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(e) : "RCCL error"));
}
#define UPSP_NCCL_CHECK(call, what)                              \
    do {                                                         \
        const ncclResult_t e_ = (call);                          \
        if (e_ != ncclSuccess) return nccl_fail(e_, what);       \
    } while (0)
// between ncclGroupStart and ncclGroupEnd: a failed call closes the group before it returns (an open group would swallow
// every later RCCL call of the process), and the caller's state is left as it was before the call
#define UPSP_NCCL_GROUP_CHECK(call, what)                        \
    do {                                                         \
        const ncclResult_t e_ = (call);                          \
        if (e_ != ncclSuccess) {                                 \
            rccl().GroupEnd();                                   \
            return nccl_fail(e_, what);                          \
        }                                                        \
    } while (0)

// ---- the local transport: every rank of the group lives in this process ------------------------------------------
struct LocalPost {
    const void *ptr;
    size_t bytes;
    hipEvent_t ready;
};
struct LocalGroup {
    int world = 0;
    std::mutex mu;
    std::map<std::tuple<int, int, int>, LocalPost> sends;          // (src, dst, tag) -> posted block
    std::vector<std::pair<double *, double *>> reduce_posts;         // accumulators posted for the current all-reduce
    size_t reduce_n = 0;
};

}  // namespace

struct XferStream {
    hipStream_t st = nullptr;
    ~XferStream()
    {
        if (st) {
            (void)hipStreamSynchronize(st);
            (void)hipStreamDestroy(st);
        }
    }
};
struct upsp_comm {
    int kind = 0;                 // 0 RCCL, 1 local
    int rank = 0, world = 1;
    ncclComm_t nccl = nullptr;
    bool own = false;             // the communicator was created here (upsp_comm_create) and is destroyed here
    std::shared_ptr<LocalGroup> local;
    // ONE transfer stream per communicator, shared by every exchange on it: RCCL orders the launches of a communicator anyway,
    // and launches that alternate between two user streams (two exchanges used in turn) pay an extra cross-stream hand-over
    // inside RCCL per launch (0.1-0.16 ms before every second step's pass B, measured)
    std::shared_ptr<XferStream> xfer;      // (exchanges keep a reference: the stream outlives whichever is destroyed first)
};

namespace {

// apportion (psp_process.cpp:611-624): contiguous near-equal ranges
void apportion(int64_t value, int nbins, std::vector<int64_t> &start, std::vector<int64_t> &extent)
{
    const int64_t block = value / nbins, rem = value % nbins;
    start.assign(nbins, 0);
    extent.assign(nbins, 0);
    int64_t next = 0;
    for (int b = 0; b < nbins; ++b) {
        start[b] = next;
        extent[b] = block + (b < rem ? 1 : 0);
        next += extent[b];
    }
}

// nframes cut into nchunks contiguous pieces whose boundaries are multiples of `align` (as evenly as that allows;
// trailing pieces may be empty) -- every rank cuts every rank's frame range the same way, so all block shapes are
// known everywhere without communication
void aligned_chunks(int64_t nframes, int nchunks, int64_t align, std::vector<int64_t> &start, std::vector<int64_t> &extent)
{
    std::vector<int64_t> b(nchunks + 1);
    for (int k = 0; k < nchunks; ++k) {
        // (round half to even like Python's round(): the Python host code and this must agree on every boundary)
        const double q = (double)k * (double)nframes / (double)nchunks / (double)align;
        b[k] = std::min<int64_t>(nframes, (int64_t)std::nearbyint(q) * align);
    }
    b[nchunks] = nframes;
    for (int k = 1; k <= nchunks; ++k) b[k] = std::max(b[k], b[k - 1]);
    start.assign(b.begin(), b.end() - 1);
    extent.resize(nchunks);
    for (int k = 0; k < nchunks; ++k) extent[k] = b[k + 1] - b[k];
}

// ---- 12-bit wire format -----------------------------------------------------------------------------------------------
// rows [R][fc] u16 (pitch fc) -> [R][3 * ceil(fc / 2)] bytes: two values in three bytes (v0 >> 4, (v0 & 15) << 4 | v1 >> 8, v1 & 255);
// a value above 4095 sets *err
__global__ void __launch_bounds__(256)
    pack12_rows_kernel(const uint16_t *__restrict__ src, long long nrows, int fc, uint8_t *__restrict__ dst, unsigned *err)
{
    const int pairs = (fc + 1) / 2;
    const long long total = nrows * pairs;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / pairs;
        const int p = (int)(i % pairs);
        const unsigned v0 = src[r * fc + 2 * p], v1 = (2 * p + 1 < fc) ? src[r * fc + 2 * p + 1] : 0u;
        if ((v0 | v1) > 4095u) atomicOr(err, 1u);
        uint8_t *d = dst + (r * pairs + p) * 3;
        d[0] = (uint8_t)(v0 >> 4);
        d[1] = (uint8_t)(((v0 & 15u) << 4) | (v1 >> 8));
        d[2] = (uint8_t)(v1 & 255u);
    }
}

// 12-bit block [nrows][3 * ceil(fc / 2) bytes] from the wire -> rows rowidx[r] of dst (f32, pitch ld), columns [0, fc): a wave
// per row, a lane widens four values (6 bytes) per trip and stores them as one 16-byte piece when the row allows it
__global__ void __launch_bounds__(256)
    place12_rows_kernel(const uint8_t *__restrict__ src, long long nrows, int fc, const long long *__restrict__ rowidx,
                        float *__restrict__ dst, long long ld)
{
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= nrows) return;
    const int lane = threadIdx.x & 63;
    const int pairs = (fc + 1) / 2;
    const uint8_t *s = src + r * pairs * 3;
    float *d = dst + rowidx[r] * ld;
    const bool vec = ((ld & 3) == 0) && ((reinterpret_cast<size_t>(dst) & 15) == 0);
    typedef float v4f __attribute__((ext_vector_type(4)));
    for (int c = 4 * lane; c < fc; c += 256) {           // values c .. c + 3 = pairs c / 2, c / 2 + 1
        const uint8_t *q = s + (c / 2) * 3;
        const unsigned b0 = q[0], b1 = q[1], b2 = q[2];
        const bool two = c + 2 < fc;
        const unsigned b3 = two ? q[3] : 0u, b4 = two ? q[4] : 0u, b5 = two ? q[5] : 0u;
        const v4f o = {(float)((b0 << 4) | (b1 >> 4)), (float)(((b1 & 15u) << 8) | b2),
                       (float)((b3 << 4) | (b4 >> 4)), (float)(((b4 & 15u) << 8) | b5)};
        if (vec && c + 3 < fc) {
            __builtin_nontemporal_store(o, reinterpret_cast<v4f *>(d + c));
        } else {
            d[c] = o.x;
            if (c + 1 < fc) d[c + 1] = o.y;
            if (c + 2 < fc) d[c + 2] = o.z;
            if (c + 3 < fc) d[c + 3] = o.w;
        }
    }
}

__global__ void __launch_bounds__(256)
    keep_differs_kernel(const uint8_t *__restrict__ skipped, const uint8_t *__restrict__ keep, size_t n, unsigned *flag)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && ((skipped[i] == 0) != (keep[i] != 0))) atomicOr(flag, 1u);
}

// send[i][0 .. fc) = compact[send_k[i]][0 .. fc): the pixel rows the destinations read, out of the sender's compact buffer
// (a wave per row; 8 bytes per lane and trip)
__global__ void __launch_bounds__(256)
    gather_pixel_rows_kernel(const uint16_t *__restrict__ compact, unsigned cpitch, const uint32_t *__restrict__ send_k,
                             long long nrows, int fc, uint16_t *__restrict__ send)
{
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= nrows) return;
    const int lane = threadIdx.x & 63;
    const uint16_t *s = compact + (size_t)send_k[r] * cpitch;
    uint16_t *d = send + r * fc;
    if ((fc & 3) == 0 && (cpitch & 3u) == 0u) {
        for (int c = 4 * lane; c < fc; c += 256) *reinterpret_cast<uint2 *>(d + c) = *reinterpret_cast<const uint2 *>(s + c);
    } else {
        for (int c = lane; c < fc; c += 64) d[c] = s[c];
    }
}

// received block [nrows][fc] (u16, or packed 12 bit) -> columns [col, col + fc) of dst [nrows][pitch] u16
template <int WIRE>
__global__ void __launch_bounds__(256)
    place_pixel_rows_kernel(const uint8_t *__restrict__ src, long long nrows, int fc, uint16_t *__restrict__ dst, long long pitch)
{
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= nrows) return;
    const int lane = threadIdx.x & 63;
    uint16_t *d = dst + r * pitch;
    if (WIRE == 2) {
        const uint16_t *s = reinterpret_cast<const uint16_t *>(src) + r * fc;
        for (int c = lane; c < fc; c += 64) d[c] = s[c];
    } else {
        const int pairs = (fc + 1) / 2;
        const uint8_t *s = src + r * pairs * 3;
        for (int p = lane; p < pairs; p += 64) {
            const unsigned b0 = s[3 * p], b1 = s[3 * p + 1], b2 = s[3 * p + 2];
            d[2 * p] = (uint16_t)((b0 << 4) | (b1 >> 4));
            if (2 * p + 1 < fc) d[2 * p + 1] = (uint16_t)(((b1 & 15u) << 8) | b2);
        }
    }
}

__global__ void __launch_bounds__(256)
    nodek_differs_kernel(const int32_t *__restrict__ a, const int32_t *__restrict__ b, const uint8_t *__restrict__ sk,
                         const uint8_t *__restrict__ keep, size_t n, unsigned *flag)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool k0 = sk ? sk[i] == 0 : true;
    if (a[i] != b[i] || (k0 != (keep[i] != 0))) atomicOr(flag, 1u);
}

__global__ void sum_posts_kernel(double *const *bufs, int nb, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int b = 0; b < nb; ++b) s += bufs[b][i];       // rank order: the same sum on every rank
    for (int b = 0; b < nb; ++b) bufs[b][i] = s;
}

}  // namespace

struct upsp_exchange {
    upsp_comm *c = nullptr;
    int K = 1;
    int64_t F = 0, N = 0;
    std::vector<int64_t> frame_start, frame_count, node_start, node_count;
    std::vector<std::vector<int64_t>> chunk_start, chunk_count;       // [rank][chunk]
    // travelling rows (all nodes when no skip list was given)
    bool have_rows = false;
    int64_t nvis = 0;
    std::vector<int64_t> cut;                  // cut[d] .. cut[d + 1]: packed rows that go to rank d
    int32_t *d_rowmap = nullptr;
    int64_t *d_vis_mine = nullptr, *d_nan_mine = nullptr;
    int64_t n_vis_mine = 0, n_nan_mine = 0;
    uint8_t *d_keep = nullptr;
    unsigned *d_flags = nullptr;               // [0] travelling set changed under assume_same, [1] 12-bit overflow
    // chunks in flight
    int k = 0;
    int wire = 0;                              // bytes per element of the chunks submitted so far: 4, 2; 12 = packed 12 bit
    std::vector<std::vector<void *>> stage;    // [chunk][source rank] received block
    std::vector<std::vector<size_t>> stage_bytes;
    std::vector<const void *> self_block;      // [chunk] this rank's own block where it lies in the send buffer (not copied, not sent)
    std::vector<void *> packed;                // [chunk] 12-bit send buffer
    std::vector<size_t> packed_bytes;
    std::shared_ptr<XferStream> xfer;           // the communicator's transfer stream (shared by its exchanges)
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_ready = nullptr, ev_done = nullptr;
    // pixel-series mode (upsp_exchange_set_pixels): what travels are the series of the ACTIVE PIXELS each destination's
    // nodes read; `cut` / `nvis` then count pixel rows
    int mode = 0;                               // 0 node rows, 1 pixel series
    uint32_t *d_send_k = nullptr;               // compact row (sender side) of every travelling pixel row, by destination
    int32_t *d_node_local = nullptr;            // my nodes -> row of d_compact_me (-1: no pixel)
    uint8_t *d_skipped_me = nullptr;            // my nodes: no camera sees them
    int32_t *d_nodek_prev = nullptr;            // node_k the lists were derived from (assume_same check)
    uint16_t *d_compact_me = nullptr;           // [rows_in][fpad]: the pixel series my nodes read, all frames of the run
    size_t compact_me_bytes = 0;
    int64_t fpad = 0;
    bool row_padding = false;                   // upsp_exchange_set_row_padding
    std::vector<void *> gathered;               // [chunk] send buffer: the rows of d_send_k out of the sender's compact buffer
    std::vector<size_t> gathered_bytes;
    uint64_t bytes_sent = 0, bytes_received = 0;   // of the pass in flight (to other ranks: what crosses a link)
    uint64_t last_sent = 0, last_received = 0;     // of the last finished pass
};

namespace {

size_t wire_row_bytes(int wire, int64_t fc)
{
    return wire == 12 ? (size_t)((fc + 1) / 2) * 3 : (size_t)fc * (size_t)wire;
}

// The block of chunk k that came from rank s: its staging buffer -- or, for this rank's own block over RCCL, the place in the
// send buffer where it lies (a block for oneself needs no transfer and no copy: the send buffer stays untouched until the pass
// is finished anyway).  UPSP_EXCHANGE_SELF_RCCL=1 sends it through ncclSend / ncclRecv to self like every other block (the
// one-GPU rehearsals of the N > 1 loop, where RCCL's kernel is then on the device beside pass B as it is between GPUs).
const void *block_of(const upsp_exchange *x, int k, int s)
{
    return (s == x->c->rank && x->self_block[k]) ? x->self_block[k] : x->stage[k][s];
}
bool self_through_rccl()
{
    static const bool v = [] {
        const char *e = getenv("UPSP_EXCHANGE_SELF_RCCL");
        return e && *e && *e != '0';
    }();
    return v;
}

void free_rows(upsp_exchange *x)
{
    for (void *p : {(void *)x->d_rowmap, (void *)x->d_vis_mine, (void *)x->d_nan_mine, (void *)x->d_keep, (void *)x->d_send_k,
                    (void *)x->d_node_local, (void *)x->d_skipped_me, (void *)x->d_nodek_prev})
        if (p) (void)hipFree(p);
    x->d_rowmap = nullptr;
    x->d_vis_mine = x->d_nan_mine = nullptr;
    x->d_keep = nullptr;
    x->d_send_k = nullptr;
    x->d_node_local = nullptr;
    x->d_skipped_me = nullptr;
    x->d_nodek_prev = nullptr;
    x->have_rows = false;
}

int ensure_buffer(void *&p, size_t &have, size_t want)
{
    if (want <= have) return UPSP_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    have = 0;
    UPSP_HIP_CHECK(hipMalloc(&p, std::max<size_t>(want, 1)));
    have = want;
    return UPSP_OK;
}

}  // namespace

extern "C" {

int upsp_comm_unique_id(uint8_t id[128])
{
    if (!id) return fail(UPSP_ERR_INVALID, "null id");
    Rccl &r = rccl();
    if (!r.ok) return fail(UPSP_ERR_HIP, "RCCL is not available: " + r.why);
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId u;
    UPSP_NCCL_CHECK(r.GetUniqueId(&u), "ncclGetUniqueId");
    std::memcpy(id, &u, 128);
    return UPSP_OK;
}

int upsp_comm_library(char *buf, size_t cap)
{
    if (!buf || cap == 0) return fail(UPSP_ERR_INVALID, "bad argument");
    Rccl &r = rccl();
    if (!r.ok) return fail(UPSP_ERR_HIP, "RCCL is not available: " + r.why);
    Dl_info info;
    const char *path = (dladdr(reinterpret_cast<void *>(r.GetUniqueId), &info) && info.dli_fname) ? info.dli_fname : "";
    std::snprintf(buf, cap, "%s", path);
    return UPSP_OK;
}

int upsp_comm_create(const uint8_t id[128], int rank, int world, upsp_comm **out)
{
    if (!out) return fail(UPSP_ERR_INVALID, "out is null");
    *out = nullptr;
    if (!id || world < 1 || rank < 0 || rank >= world) return fail(UPSP_ERR_INVALID, "bad rank / world");
    Rccl &r = rccl();
    if (!r.ok) return fail(UPSP_ERR_HIP, "RCCL is not available: " + r.why);
    ncclUniqueId u;
    std::memcpy(&u, id, 128);
    ncclComm_t comm = nullptr;
    UPSP_NCCL_CHECK(r.CommInitRank(&comm, world, u, rank), "ncclCommInitRank");
    upsp_comm *c = new upsp_comm();
    c->kind = 0;
    c->rank = rank;
    c->world = world;
    c->nccl = comm;
    c->own = true;
    *out = c;
    return UPSP_OK;
}

int upsp_comm_from_nccl(void *nccl_comm, upsp_comm **out)
{
    if (!out) return fail(UPSP_ERR_INVALID, "out is null");
    *out = nullptr;
    if (!nccl_comm) return fail(UPSP_ERR_INVALID, "null communicator");
    Rccl &r = rccl();
    if (!r.ok) return fail(UPSP_ERR_HIP, "RCCL is not available: " + r.why);
    int rank = 0, world = 0;
    UPSP_NCCL_CHECK(r.CommCount((ncclComm_t)nccl_comm, &world), "ncclCommCount");
    UPSP_NCCL_CHECK(r.CommUserRank((ncclComm_t)nccl_comm, &rank), "ncclCommUserRank");
    upsp_comm *c = new upsp_comm();
    c->kind = 0;
    c->rank = rank;
    c->world = world;
    c->nccl = (ncclComm_t)nccl_comm;
    c->own = false;
    *out = c;
    return UPSP_OK;
}

int upsp_comm_create_local(int world, upsp_comm **out_ranks)
{
    if (!out_ranks || world < 1 || world > 64) return fail(UPSP_ERR_INVALID, "bad argument");
    auto g = std::make_shared<LocalGroup>();
    g->world = world;
    for (int r = 0; r < world; ++r) {
        upsp_comm *c = new upsp_comm();
        c->kind = 1;
        c->rank = r;
        c->world = world;
        c->local = g;
        out_ranks[r] = c;
    }
    return UPSP_OK;
}

void upsp_comm_destroy(upsp_comm *c)
{
    if (!c) return;
    if (c->xfer && c->xfer->st) (void)hipStreamSynchronize(c->xfer->st);     // nothing of ours in flight on the communicator
    if (c->kind == 0 && c->own && c->nccl && rccl().CommDestroy) (void)rccl().CommDestroy(c->nccl);
    delete c;
}

int upsp_comm_rank(const upsp_comm *c, int *rank, int *world)
{
    if (!c) return fail(UPSP_ERR_INVALID, "null communicator");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (c->kind == 0 && c->nccl) {
        // what RCCL itself says about the communicator (ncclCommCount / ncclCommUserRank), not what it was created with
        Rccl &r = rccl();
        int n = 0, me = 0;
        if (r.CommCount && world) {
            UPSP_NCCL_CHECK(r.CommCount(c->nccl, &n), "ncclCommCount");
            *world = n;
        }
        if (r.CommUserRank && rank) {
            UPSP_NCCL_CHECK(r.CommUserRank(c->nccl, &me), "ncclCommUserRank");
            *rank = me;
        }
    }
    return UPSP_OK;
}

int upsp_allreduce_sums(upsp_comm *c, double *d_sum, double *d_sumsq, size_t n, void *stream)
{
    if (!c || !d_sum || !d_sumsq) return fail(UPSP_ERR_INVALID, "bad argument");
    if (n == 0) return UPSP_OK;
    hipStream_t st = (hipStream_t)stream;
    if (c->kind == 0) {
        Rccl &r = rccl();
        KTimed kt("allreduce_sums", st);
        UPSP_NCCL_CHECK(r.GroupStart(), "ncclGroupStart");
        UPSP_NCCL_GROUP_CHECK(r.AllReduce(d_sum, d_sum, n, ncclDouble, ncclSum, c->nccl, st), "ncclAllReduce");
        UPSP_NCCL_GROUP_CHECK(r.AllReduce(d_sumsq, d_sumsq, n, ncclDouble, ncclSum, c->nccl, st), "ncclAllReduce");
        UPSP_NCCL_CHECK(r.GroupEnd(), "ncclGroupEnd");
        return UPSP_OK;
    }
    // local transport: the call of the LAST rank of the group does the sums for everybody (the ranks are driven one
    // after the other from one thread; every buffer must be complete on its stream by then: synchronised here)
    LocalGroup &g = *c->local;
    std::lock_guard<std::mutex> lk(g.mu);
    UPSP_HIP_CHECK(hipStreamSynchronize(st));
    if (g.reduce_posts.empty()) g.reduce_n = n;
    if (g.reduce_n != n) return fail(UPSP_ERR_INVALID, "local all-reduce: sizes differ between the ranks");
    g.reduce_posts.emplace_back(d_sum, d_sumsq);
    if ((int)g.reduce_posts.size() < g.world) return UPSP_OK;
    for (int which = 0; which < 2; ++which) {
        std::vector<double *> h(g.world);
        for (int b = 0; b < g.world; ++b) h[b] = which ? g.reduce_posts[b].second : g.reduce_posts[b].first;
        double **d = nullptr;
        UPSP_HIP_CHECK(hipMalloc(&d, sizeof(double *) * g.world));
        UPSP_HIP_CHECK(hipMemcpy(d, h.data(), sizeof(double *) * g.world, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(sum_posts_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (double *const *)d, g.world, n);
        UPSP_HIP_CHECK(hipStreamSynchronize(st));
        (void)hipFree(d);
    }
    g.reduce_posts.clear();
    return UPSP_OK;
}

int upsp_exchange_create(upsp_comm *c, int64_t nframes_total, int64_t nnodes, int nchunks, upsp_exchange **out)
{
    if (!out) return fail(UPSP_ERR_INVALID, "out is null");
    *out = nullptr;
    if (!c || nframes_total < 0 || nnodes <= 0 || nchunks < 1 || nchunks > 1024) return fail(UPSP_ERR_INVALID, "bad argument");
    upsp_exchange *x = new upsp_exchange();
    x->c = c;
    x->K = nchunks;
    x->F = nframes_total;
    x->N = nnodes;
    apportion(nframes_total, c->world, x->frame_start, x->frame_count);
    apportion(nnodes, c->world, x->node_start, x->node_count);
    x->chunk_start.resize(c->world);
    x->chunk_count.resize(c->world);
    for (int s = 0; s < c->world; ++s) aligned_chunks(x->frame_count[s], nchunks, 64, x->chunk_start[s], x->chunk_count[s]);
    x->stage.assign(nchunks, std::vector<void *>(c->world, nullptr));
    x->stage_bytes.assign(nchunks, std::vector<size_t>(c->world, 0));
    x->self_block.assign(nchunks, nullptr);
    x->packed.assign(nchunks, nullptr);
    x->packed_bytes.assign(nchunks, 0);
    x->gathered.assign(nchunks, nullptr);
    x->gathered_bytes.assign(nchunks, 0);
    hipError_t e = hipSuccess;
    if (!c->xfer) {
        c->xfer = std::make_shared<XferStream>();
        // LOW priority: a block has until the pass is finished to arrive (with two exchanges in turn: a whole step), and RCCL's
        // workgroups should not hold compute units the frame loop's kernels are waiting for -- one GPU through one-rank RCCL, deferred
        // exchange, one call: 1.19 ms per step against 1.29 at normal priority (plain loop 0.95-1.00; tools/gpu_n1_ab.sh).
        // That was measured with ONE rank only (this pool has one GPU per box).  Between GPUs a starved low-priority RCCL kernel
        // also stalls the PEER's send / receive kernels, which spin on its compute units: until that has been measured on a
        // node with >= 2 GPUs a group of more than one rank keeps NORMAL priority.  UPSP_XFER_PRIORITY=low|normal|high overrides.
        const char *pe = getenv("UPSP_XFER_PRIORITY");
        int least = 0, greatest = 0;
        const bool have_range = hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess;
        const bool want_low = pe ? pe[0] == 'l' : c->world == 1;
        const bool want_high = pe && pe[0] == 'h';
        if (have_range && (want_low || want_high))
            e = hipStreamCreateWithPriority(&c->xfer->st, hipStreamNonBlocking, want_high ? greatest : least);
        else
            e = hipStreamCreateWithFlags(&c->xfer->st, hipStreamNonBlocking);
    }
    x->xfer = c->xfer;
    x->comm_stream = c->xfer->st;
    if (e == hipSuccess) e = hipEventCreateWithFlags(&x->ev_ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&x->ev_done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc(&x->d_flags, 2 * sizeof(unsigned));
    if (e == hipSuccess) e = hipMemset(x->d_flags, 0, 2 * sizeof(unsigned));
    if (e != hipSuccess) {
        upsp_exchange_destroy(x);
        return fail(UPSP_ERR_HIP, std::string("exchange: ") + hipGetErrorString(e));
    }
    *out = x;
    return UPSP_OK;
}

void upsp_exchange_destroy(upsp_exchange *x)
{
    if (!x) return;
    if (x->comm_stream) (void)hipStreamSynchronize(x->comm_stream);
    free_rows(x);
    for (auto &v : x->stage)
        for (void *p : v)
            if (p) (void)hipFree(p);
    for (void *p : x->packed)
        if (p) (void)hipFree(p);
    for (void *p : x->gathered)
        if (p) (void)hipFree(p);
    if (x->d_compact_me) (void)hipFree(x->d_compact_me);
    if (x->d_flags) (void)hipFree(x->d_flags);
    if (x->ev_ready) (void)hipEventDestroy(x->ev_ready);
    if (x->ev_done) (void)hipEventDestroy(x->ev_done);
    delete x;
}

int upsp_exchange_layout(const upsp_exchange *x, int64_t *frame_start, int64_t *frame_count, int64_t *node_start,
                         int64_t *node_count)
{
    if (!x) return fail(UPSP_ERR_INVALID, "null exchange");
    const int r = x->c->rank;
    if (frame_start) *frame_start = x->frame_start[r];
    if (frame_count) *frame_count = x->frame_count[r];
    if (node_start) *node_start = x->node_start[r];
    if (node_count) *node_count = x->node_count[r];
    return UPSP_OK;
}

int upsp_exchange_chunk(const upsp_exchange *x, int k, int64_t *first_frame, int64_t *nframes)
{
    if (!x || k < 0 || k >= x->K) return fail(UPSP_ERR_INVALID, "bad chunk");
    const int r = x->c->rank;
    if (first_frame) *first_frame = x->chunk_start[r][k];
    if (nframes) *nframes = x->chunk_count[r][k];
    return UPSP_OK;
}

int upsp_exchange_set_skipped(upsp_exchange *x, const uint8_t *d_skipped, int assume_same, void *stream)
{
    if (!x) return fail(UPSP_ERR_INVALID, "null exchange");
    hipStream_t st = (hipStream_t)stream;
    const int W = x->c->world, me = x->c->rank;
    const size_t N = (size_t)x->N;
    if (assume_same && x->have_rows && x->mode == 0 && d_skipped && x->d_keep) {
        // the caller states that the set is the one of the previous call (the projection did not change): nothing is
        // read back, the claim is checked on the device and upsp_exchange_finish reports a broken one
        hipLaunchKernelGGL(keep_differs_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, d_skipped,
                           (const uint8_t *)x->d_keep, N, x->d_flags);
        UPSP_HIP_CHECK(hipGetLastError());
        return UPSP_OK;
    }
    // the travelling set: one device -> host read of the flags (once per projection), lists built on the host
    std::vector<uint8_t> sk(N, 0);
    if (d_skipped) {
        UPSP_HIP_CHECK(hipMemcpyAsync(sk.data(), d_skipped, N, hipMemcpyDeviceToHost, st));
        UPSP_HIP_CHECK(hipStreamSynchronize(st));
    }
    std::vector<int32_t> rowmap(N, -1);
    std::vector<uint8_t> keep(N);
    std::vector<int64_t> vis_mine, nan_mine;
    x->cut.assign(W + 1, 0);
    int64_t row = 0;
    for (int d = 0; d < W; ++d) {
        x->cut[d] = row;
        for (int64_t n = x->node_start[d]; n < x->node_start[d] + x->node_count[d]; ++n) {
            keep[n] = sk[n] == 0;
            if (keep[n]) {
                rowmap[n] = (int32_t)row++;
                if (d == me) vis_mine.push_back(n - x->node_start[me]);
            } else if (d == me) {
                nan_mine.push_back(n - x->node_start[me]);
            }
        }
    }
    x->cut[W] = row;
    x->nvis = row;
    UPSP_HIP_CHECK(hipStreamSynchronize(st));        // launches queued earlier may still read the old lists
    free_rows(x);
    x->n_vis_mine = (int64_t)vis_mine.size();
    x->n_nan_mine = (int64_t)nan_mine.size();
    UPSP_HIP_CHECK(hipMalloc(&x->d_rowmap, sizeof(int32_t) * N));
    UPSP_HIP_CHECK(hipMalloc(&x->d_keep, N));
    UPSP_HIP_CHECK(hipMalloc(&x->d_vis_mine, sizeof(int64_t) * std::max<size_t>(vis_mine.size(), 1)));
    UPSP_HIP_CHECK(hipMalloc(&x->d_nan_mine, sizeof(int64_t) * std::max<size_t>(nan_mine.size(), 1)));
    UPSP_HIP_CHECK(hipMemcpy(x->d_rowmap, rowmap.data(), sizeof(int32_t) * N, hipMemcpyHostToDevice));
    UPSP_HIP_CHECK(hipMemcpy(x->d_keep, keep.data(), N, hipMemcpyHostToDevice));
    if (!vis_mine.empty())
        UPSP_HIP_CHECK(hipMemcpy(x->d_vis_mine, vis_mine.data(), sizeof(int64_t) * vis_mine.size(), hipMemcpyHostToDevice));
    if (!nan_mine.empty())
        UPSP_HIP_CHECK(hipMemcpy(x->d_nan_mine, nan_mine.data(), sizeof(int64_t) * nan_mine.size(), hipMemcpyHostToDevice));
    x->have_rows = true;
    x->mode = 0;
    return UPSP_OK;
}

int upsp_exchange_rows(const upsp_exchange *x, const int32_t **d_rowmap, int64_t *packed_rows)
{
    if (!x || !x->have_rows || x->mode != 0) return fail(UPSP_ERR_INVALID, "exchange: upsp_exchange_set_skipped first");
    if (d_rowmap) *d_rowmap = x->d_rowmap;
    if (packed_rows) *packed_rows = x->nvis;
    return UPSP_OK;
}

// chunk k of this rank on the wire: rows [cut[p], cut[p + 1]) x fc go to rank p, my rows x chunk k of rank p come back
static int submit_core(upsp_exchange *x, const void *d_chunk, int wire, hipStream_t st)
{
    if (wire != 4 && wire != 2 && wire != 12) return fail(UPSP_ERR_INVALID, "exchange: wire format is 4 (f32), 2 (u16) or 12 (u16 packed to 12 bit)");
    if (x->k >= x->K) return fail(UPSP_ERR_INVALID, "exchange: every chunk was already submitted (finish first)");
    if (x->k > 0 && wire != x->wire) return fail(UPSP_ERR_INVALID, "exchange: the chunks of one pass share a wire format");
    const int W = x->c->world, me = x->c->rank, k = x->k;
    const int64_t fc = x->chunk_count[me][k];
    if (fc > 0 && x->nvis > 0 && !d_chunk) return fail(UPSP_ERR_INVALID, "exchange: null chunk");
    x->wire = wire;
    if (k == 0) x->bytes_sent = x->bytes_received = 0;
    const size_t rb = wire_row_bytes(wire, fc);
    const uint8_t *send = static_cast<const uint8_t *>(d_chunk);
    if (wire == 12 && fc > 0 && x->nvis > 0) {
        int rc = ensure_buffer(x->packed[k], x->packed_bytes[k], rb * (size_t)x->nvis);
        if (rc != UPSP_OK) return rc;
        KTimed kt("pack12_rows_kernel", st);
        hipLaunchKernelGGL(pack12_rows_kernel, dim3(2048), dim3(256), 0, st, static_cast<const uint16_t *>(d_chunk),
                           (long long)x->nvis, (int)fc, static_cast<uint8_t *>(x->packed[k]), x->d_flags + 1);
        send = static_cast<const uint8_t *>(x->packed[k]);
    }
    // staging for what arrives: from rank s, my rows x its chunk k (my own block is read where it lies in the send buffer)
    const int64_t rows_in = x->cut[me + 1] - x->cut[me];
    const bool self_alias = x->c->kind == 0 && !self_through_rccl();
    x->self_block[k] = (self_alias && rows_in > 0 && fc > 0) ? send + rb * (size_t)x->cut[me] : nullptr;
    for (int s = 0; s < W; ++s) {
        if (s == me && self_alias) continue;
        const size_t want = wire_row_bytes(wire, x->chunk_count[s][k]) * (size_t)rows_in;
        int rc = ensure_buffer(x->stage[k][s], x->stage_bytes[k][s], want);
        if (rc != UPSP_OK) return rc;
    }
    UPSP_HIP_CHECK(hipEventRecord(x->ev_ready, st));
    if (x->c->kind == 0) {
        Rccl &r = rccl();
        UPSP_HIP_CHECK(hipStreamWaitEvent(x->comm_stream, x->ev_ready, 0));
        // (x->k and the byte counters move only once the whole group is in: a failed call leaves the exchange where it was)
        uint64_t sent = 0, received = 0;
        const bool any_peer = W > 1 || !self_alias;         // (one rank, own block in place: nothing for RCCL to do)
        if (any_peer) UPSP_NCCL_CHECK(r.GroupStart(), "ncclGroupStart");
        for (int p = 0; p < W && any_peer; ++p) {
            if (p == me && self_alias) continue;
            const size_t out_b = rb * (size_t)(x->cut[p + 1] - x->cut[p]);
            const size_t in_b = wire_row_bytes(wire, x->chunk_count[p][k]) * (size_t)rows_in;
            if (out_b) {
                UPSP_NCCL_GROUP_CHECK(r.Send(send + rb * (size_t)x->cut[p], out_b, ncclUint8, p, x->c->nccl, x->comm_stream), "ncclSend");
                if (p != me) sent += out_b;
            }
            if (in_b) {
                UPSP_NCCL_GROUP_CHECK(r.Recv(x->stage[k][p], in_b, ncclUint8, p, x->c->nccl, x->comm_stream), "ncclRecv");
                if (p != me) received += in_b;
            }
        }
        if (any_peer) UPSP_NCCL_CHECK(r.GroupEnd(), "ncclGroupEnd");
        // "everything submitted so far has arrived" is marked HERE, behind this chunk: the transfer stream is shared by the
        // exchanges of the communicator, and a mark set only when the pass is finished would sit behind whatever another
        // exchange has submitted in the meantime (two exchanges in turn: the next step's blocks)
        UPSP_HIP_CHECK(hipEventRecord(x->ev_done, x->comm_stream));
        x->bytes_sent += sent;
        x->bytes_received += received;
        x->k += 1;
        return UPSP_OK;
    }
    x->k += 1;
    // local transport: post my blocks; the receivers copy them when they finish
    LocalGroup &g = *x->c->local;
    std::lock_guard<std::mutex> lk(g.mu);
    for (int p = 0; p < W; ++p) {
        const size_t out_b = rb * (size_t)(x->cut[p + 1] - x->cut[p]);
        hipEvent_t ev = nullptr;
        UPSP_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        UPSP_HIP_CHECK(hipEventRecord(ev, st));
        g.sends[std::make_tuple(me, p, k)] = LocalPost{send + rb * (size_t)x->cut[p], out_b, ev};
        if (p != me) x->bytes_sent += out_b;
    }
    return UPSP_OK;
}

// every block of the pass is in its staging buffer once `st` gets there
static int receive_core(upsp_exchange *x, hipStream_t st)
{
    if (x->k != x->K) return fail(UPSP_ERR_INVALID, "exchange: finish before every chunk was submitted");
    const int W = x->c->world, me = x->c->rank;
    const int64_t rows_in = x->cut[me + 1] - x->cut[me];
    if (x->c->kind == 0) {
        UPSP_HIP_CHECK(hipStreamWaitEvent(st, x->ev_done, 0));      // (recorded behind the last chunk's group, submit_core)
        return UPSP_OK;
    }
    LocalGroup &g = *x->c->local;
    std::lock_guard<std::mutex> lk(g.mu);
    for (int k = 0; k < x->K; ++k)
        for (int s = 0; s < W; ++s) {
            const size_t in_b = wire_row_bytes(x->wire, x->chunk_count[s][k]) * (size_t)rows_in;
            auto it = g.sends.find(std::make_tuple(s, me, k));
            if (it == g.sends.end()) return fail(UPSP_ERR_INVALID, "local exchange: a rank has not submitted yet");
            if (it->second.bytes != in_b) return fail(UPSP_ERR_INVALID, "local exchange: block sizes disagree between the ranks");
            UPSP_HIP_CHECK(hipStreamWaitEvent(st, it->second.ready, 0));
            if (in_b) UPSP_HIP_CHECK(hipMemcpyAsync(x->stage[k][s], it->second.ptr, in_b, hipMemcpyDeviceToDevice, st));
            if (s != me) x->bytes_received += in_b;
            (void)hipEventDestroy(it->second.ready);
            g.sends.erase(it);
        }
    return UPSP_OK;
}

int upsp_exchange_submit(upsp_exchange *x, const void *d_chunk, int wire, void *stream)
{
    if (!x || !x->have_rows || x->mode != 0) return fail(UPSP_ERR_INVALID, "exchange: upsp_exchange_set_skipped first");
    return submit_core(x, d_chunk, wire, (hipStream_t)stream);
}

int upsp_exchange_finish(upsp_exchange *x, float *d_series, int64_t ld, void *stream)
{
    if (!x || !x->have_rows || x->mode != 0 || !d_series) return fail(UPSP_ERR_INVALID, "bad argument");
    if (ld < x->F) return fail(UPSP_ERR_INVALID, "exchange: ld smaller than the frame count");
    hipStream_t st = (hipStream_t)stream;
    const int W = x->c->world, me = x->c->rank;
    const int64_t rows_in = x->cut[me + 1] - x->cut[me];
    int rc = receive_core(x, st);
    if (rc != UPSP_OK) return rc;
    {
        KTimed kt("exchange_place_kernels", st);
        for (int k = 0; k < x->K; ++k)
            for (int s = 0; s < W; ++s) {
                const int64_t fs = x->chunk_count[s][k];
                if (!fs || !rows_in) continue;
                float *dst = d_series + x->frame_start[s] + x->chunk_start[s][k];
                if (x->wire == 4)
                    rc = upsp_scatter_rows_f32(static_cast<const float *>(block_of(x, k, s)), (size_t)rows_in, (int)fs, x->d_vis_mine, dst, ld, stream);
                else if (x->wire == 2)
                    rc = upsp_scatter_rows_u16(static_cast<const uint16_t *>(block_of(x, k, s)), (size_t)rows_in, (int)fs, x->d_vis_mine, dst, ld, stream);
                else
                    hipLaunchKernelGGL(place12_rows_kernel, dim3((unsigned)((rows_in + 3) / 4)), dim3(256), 0, st,
                                       static_cast<const uint8_t *>(block_of(x, k, s)), (long long)rows_in, (int)fs,
                                       (const long long *)x->d_vis_mine, dst, (long long)ld);
                if (rc != UPSP_OK) return rc;
            }
        UPSP_HIP_CHECK(hipGetLastError());
    }
    // the rows that do not travel: NaN in every frame (psp_process.cpp:1821-1825), written by every exchange
    if (x->n_nan_mine) {
        rc = upsp_fill_rows_f32(__builtin_nanf(""), (size_t)x->n_nan_mine, (int)x->F, x->d_nan_mine, d_series, ld, stream);
        if (rc != UPSP_OK) return rc;
    }
    x->k = 0;
    x->last_sent = x->bytes_sent;
    x->last_received = x->bytes_received;
    return UPSP_OK;
}

// ---- pixel-series mode --------------------------------------------------------------------------------------------------
// What a node's series is made of is the series of the PIXEL it reads, and on a model finer than the pixel grid several
// nodes read the same pixel (the bench model: 190 k travelling nodes on 66 k active pixels).  Here the ranks exchange
// the active pixels' u16 series -- every destination gets the pixels its node slice reads, each once -- and the OWNER of
// a node runs pass B over all frames of the run: the series come out the same, the accumulators complete (no partial
// sums to reduce: other ranks' slices stay zero, so upsp_allreduce_sums still delivers the full vectors), and a third of
// the bytes cross the links.
int upsp_exchange_set_pixels(upsp_exchange *x, const int32_t *d_node_k, const uint8_t *d_skipped, int assume_same, void *stream)
{
    if (!x || !d_node_k) return fail(UPSP_ERR_INVALID, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int W = x->c->world, me = x->c->rank;
    const size_t N = (size_t)x->N;
    if (assume_same && x->have_rows && x->mode == 1 && x->d_nodek_prev && x->d_keep) {
        hipLaunchKernelGGL(nodek_differs_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, d_node_k,
                           (const int32_t *)x->d_nodek_prev, d_skipped, (const uint8_t *)x->d_keep, N, x->d_flags);
        UPSP_HIP_CHECK(hipGetLastError());
        return UPSP_OK;
    }
    std::vector<int32_t> nk(N);
    std::vector<uint8_t> sk(N, 0);
    UPSP_HIP_CHECK(hipMemcpyAsync(nk.data(), d_node_k, sizeof(int32_t) * N, hipMemcpyDeviceToHost, st));
    if (d_skipped) UPSP_HIP_CHECK(hipMemcpyAsync(sk.data(), d_skipped, N, hipMemcpyDeviceToHost, st));
    UPSP_HIP_CHECK(hipStreamSynchronize(st));
    std::vector<uint32_t> send_k;
    std::vector<int32_t> local(std::max<int64_t>(x->node_count[me], 1), -1);
    std::vector<uint8_t> keep(N), sk_me(std::max<int64_t>(x->node_count[me], 1), 0);
    x->cut.assign(W + 1, 0);
    for (int d = 0; d < W; ++d) {
        x->cut[d] = (int64_t)send_k.size();
        const int64_t n0 = x->node_start[d], n1 = n0 + x->node_count[d];
        std::vector<uint32_t> ks;
        for (int64_t n = n0; n < n1; ++n) {
            keep[n] = sk[n] == 0;
            // -1: the node has no pixel (series of zeros).  -2: its pixel lies outside the candidate map the table was made
            // with (upsp_pipeline_set_active_hint) -- it HAS a series, which this exchange would silently replace by zeros
            if (keep[n] && nk[n] < -1)
                return fail(UPSP_ERR_INVALID, "exchange: node table made with a candidate-pixel map that misses a node's pixel "
                                              "(take it from upsp_pipeline_pixel_series)");
            if (keep[n] && nk[n] >= 0) ks.push_back((uint32_t)nk[n]);
        }
        std::sort(ks.begin(), ks.end());
        ks.erase(std::unique(ks.begin(), ks.end()), ks.end());
        if (d == me) {
            for (int64_t n = n0; n < n1; ++n) {
                sk_me[n - n0] = sk[n];
                if (keep[n] && nk[n] >= 0)
                    local[n - n0] = (int32_t)(std::lower_bound(ks.begin(), ks.end(), (uint32_t)nk[n]) - ks.begin());
            }
        }
        send_k.insert(send_k.end(), ks.begin(), ks.end());
    }
    x->cut[W] = (int64_t)send_k.size();
    free_rows(x);
    x->nvis = (int64_t)send_k.size();
    const size_t nn = (size_t)std::max<int64_t>(x->node_count[me], 1);
    UPSP_HIP_CHECK(hipMalloc(&x->d_send_k, sizeof(uint32_t) * std::max<size_t>(send_k.size(), 1)));
    UPSP_HIP_CHECK(hipMalloc(&x->d_node_local, sizeof(int32_t) * nn));
    UPSP_HIP_CHECK(hipMalloc(&x->d_skipped_me, nn));
    UPSP_HIP_CHECK(hipMalloc(&x->d_nodek_prev, sizeof(int32_t) * N));
    UPSP_HIP_CHECK(hipMalloc(&x->d_keep, N));
    if (!send_k.empty()) UPSP_HIP_CHECK(hipMemcpy(x->d_send_k, send_k.data(), sizeof(uint32_t) * send_k.size(), hipMemcpyHostToDevice));
    UPSP_HIP_CHECK(hipMemcpy(x->d_node_local, local.data(), sizeof(int32_t) * nn, hipMemcpyHostToDevice));
    UPSP_HIP_CHECK(hipMemcpy(x->d_skipped_me, sk_me.data(), nn, hipMemcpyHostToDevice));
    UPSP_HIP_CHECK(hipMemcpy(x->d_nodek_prev, nk.data(), sizeof(int32_t) * N, hipMemcpyHostToDevice));
    UPSP_HIP_CHECK(hipMemcpy(x->d_keep, keep.data(), N, hipMemcpyHostToDevice));
    // the pixel series my nodes read, all frames of the run: [rows_in][fpad] u16
    const int64_t rows_in = x->cut[me + 1] - x->cut[me];
    x->fpad = (x->F + 63) / 64 * 64;
    const size_t want = sizeof(uint16_t) * (size_t)std::max<int64_t>(rows_in, 1) * (size_t)std::max<int64_t>(x->fpad, 64);
    int rc = ensure_buffer(reinterpret_cast<void *&>(x->d_compact_me), x->compact_me_bytes, want);
    if (rc != UPSP_OK) return rc;
    x->have_rows = true;
    x->mode = 1;
    return UPSP_OK;
}

int upsp_exchange_pixel_rows(const upsp_exchange *x, int64_t *rows_out, int64_t *rows_in)
{
    if (!x || !x->have_rows || x->mode != 1) return fail(UPSP_ERR_INVALID, "exchange: upsp_exchange_set_pixels first");
    if (rows_out) *rows_out = x->nvis;
    if (rows_in) *rows_in = x->cut[x->c->rank + 1] - x->cut[x->c->rank];
    return UPSP_OK;
}

int upsp_exchange_submit_pixels(upsp_exchange *x, const uint16_t *d_compact, uint32_t cpitch, int wire, void *stream)
{
    if (!x || !x->have_rows || x->mode != 1) return fail(UPSP_ERR_INVALID, "exchange: upsp_exchange_set_pixels first");
    if (wire != 2 && wire != 12) return fail(UPSP_ERR_INVALID, "exchange: pixel series travel as u16 (2) or packed to 12 bits (12)");
    if (x->k >= x->K) return fail(UPSP_ERR_INVALID, "exchange: every chunk was already submitted (finish first)");
    hipStream_t st = (hipStream_t)stream;
    const int k = x->k;
    const int64_t fc = x->chunk_count[x->c->rank][k];
    if (fc > 0 && x->nvis > 0) {
        if (!d_compact || (int64_t)cpitch < fc) return fail(UPSP_ERR_INVALID, "exchange: compact buffer too narrow for the chunk");
        int rc = ensure_buffer(x->gathered[k], x->gathered_bytes[k], sizeof(uint16_t) * (size_t)x->nvis * (size_t)fc);
        if (rc != UPSP_OK) return rc;
        KTimed kt("gather_pixel_rows_kernel", st);
        hipLaunchKernelGGL(gather_pixel_rows_kernel, dim3((unsigned)((x->nvis + 3) / 4)), dim3(256), 0, st, d_compact, cpitch,
                           (const uint32_t *)x->d_send_k, (long long)x->nvis, (int)fc, static_cast<uint16_t *>(x->gathered[k]));
        UPSP_HIP_CHECK(hipGetLastError());
    }
    return submit_core(x, x->gathered[k], wire, st);
}

int upsp_exchange_set_row_padding(upsp_exchange *x, int on)
{
    if (!x) return fail(UPSP_ERR_INVALID, "null exchange");
    x->row_padding = on != 0;
    return UPSP_OK;
}

int upsp_exchange_finish_pixels(upsp_exchange *x, float *d_series, int64_t ld, double *d_sum_mine, double *d_sumsq_mine, void *stream)
{
    if (!x || !x->have_rows || x->mode != 1 || !d_series || !d_sum_mine || !d_sumsq_mine) return fail(UPSP_ERR_INVALID, "bad argument");
    if (ld < x->F) return fail(UPSP_ERR_INVALID, "exchange: ld smaller than the frame count");
    hipStream_t st = (hipStream_t)stream;
    const int W = x->c->world, me = x->c->rank;
    const int64_t rows_in = x->cut[me + 1] - x->cut[me];
    int rc = receive_core(x, st);
    if (rc != UPSP_OK) return rc;
    // One chunk per rank, u16 on the wire, every block's frame count a multiple of four: what arrived from source s IS a
    // [pixel row][frames of s] series buffer pass B can read as it lies (8-byte loads of four frames; row pitch = the block's
    // frame count), so the owner's pass B runs once per source block into that block's columns -- no copy into one long
    // series buffer first.  (With several chunks per rank the pieces would be short row segments: placed, below.)
    bool direct = x->wire == 2 && x->K == 1 && rows_in > 0;
    for (int s = 0; s < W && direct; ++s) direct = (x->chunk_count[s][0] % 4) == 0;
    // columns the last launch of pass B may write: up to the next 128-byte line of the rows when the caller gave up its padding
    const int64_t pad_to = x->row_padding && (ld % 32) == 0 ? std::min<int64_t>(ld, (x->F + 31) / 32 * 32) : x->F;
    if (direct) {
        if (x->node_count[me] > 0 && x->F > 0) {
            std::vector<upsp::SeriesBlock> blocks;                 // (frame_start[] ascends with the rank: the blocks in time order)
            for (int s = 0; s < W; ++s)
                blocks.push_back({static_cast<const uint16_t *>(block_of(x, 0, s)), (unsigned)x->chunk_count[s][0], x->chunk_count[s][0]});
            rc = upsp::rows_from_pixel_blocks(blocks.data(), W, x->d_node_local, x->d_skipped_me, (size_t)x->node_count[me], d_series, ld,
                                              pad_to, d_sum_mine, d_sumsq_mine, st);
            if (rc != UPSP_OK) return rc;
        }
        x->k = 0;
        x->last_sent = x->bytes_sent;
        x->last_received = x->bytes_received;
        return UPSP_OK;
    }
    {
        KTimed kt("exchange_place_kernels", st);
        for (int k = 0; k < x->K; ++k)
            for (int s = 0; s < W; ++s) {
                const int64_t fs = x->chunk_count[s][k];
                if (!fs || !rows_in) continue;
                uint16_t *dst = x->d_compact_me + x->frame_start[s] + x->chunk_start[s][k];
                const dim3 grid((unsigned)((rows_in + 3) / 4)), block(256);
                if (x->wire == 2)
                    hipLaunchKernelGGL(place_pixel_rows_kernel<2>, grid, block, 0, st, static_cast<const uint8_t *>(block_of(x, k, s)),
                                       (long long)rows_in, (int)fs, dst, (long long)x->fpad);
                else
                    hipLaunchKernelGGL(place_pixel_rows_kernel<12>, grid, block, 0, st, static_cast<const uint8_t *>(block_of(x, k, s)),
                                       (long long)rows_in, (int)fs, dst, (long long)x->fpad);
            }
        UPSP_HIP_CHECK(hipGetLastError());
    }
    // pass B by the owner of the nodes, over every frame of the run: series, NaN rows, complete accumulators
    if (x->node_count[me] > 0 && x->F > 0) {
        const upsp::SeriesBlock all = {x->d_compact_me, (unsigned)x->fpad, x->F};
        rc = upsp::rows_from_pixel_blocks(&all, 1, x->d_node_local, x->d_skipped_me, (size_t)x->node_count[me], d_series, ld, pad_to,
                                          d_sum_mine, d_sumsq_mine, st);
        if (rc != UPSP_OK) return rc;
    }
    x->k = 0;
    x->last_sent = x->bytes_sent;
    x->last_received = x->bytes_received;
    return UPSP_OK;
}

int upsp_exchange_verify(upsp_exchange *x, void *stream)
{
    if (!x) return fail(UPSP_ERR_INVALID, "null exchange");
    unsigned h[2] = {0, 0};
    UPSP_HIP_CHECK(hipMemcpyAsync(h, x->d_flags, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream));
    UPSP_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    if (h[0] || h[1]) UPSP_HIP_CHECK(hipMemset(x->d_flags, 0, sizeof(h)));
    if (h[0]) return fail(UPSP_ERR_INVALID, "exchange: the skipped-node set changed although assume_same was given");
    if (h[1]) return fail(UPSP_ERR_INVALID, "exchange: a series value above 4095 was sent in the 12-bit wire format");
    return UPSP_OK;
}

int upsp_exchange_bytes(const upsp_exchange *x, uint64_t *sent, uint64_t *received)
{
    if (!x) return fail(UPSP_ERR_INVALID, "null exchange");
    if (sent) *sent = x->last_sent;
    if (received) *received = x->last_received;
    return UPSP_OK;
}

}  // extern "C"
