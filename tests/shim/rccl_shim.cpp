// TEST INFRASTRUCTURE ONLY -- a stand-in for librccl that lets SEVERAL PROCESSES ON ONE GPU form a communicator.
//
// Why: the pool this is built on hands out one-GPU boxes and RCCL refuses two ranks on one device, so the RCCL branch of
// csrc/exchange.hip (grouped ncclSend / ncclRecv per peer, ncclAllReduce; global_transpose and MPI_Reduce/Bcast of the
// reference, cpp/exec/psp_process.cpp:707-771, 1866-1872) would otherwise never run with more than one rank before an
// 8-GPU node sees it.  libupsp_gpu.so resolves its RCCL entry points from the library UPSP_RCCL_LIBRARY names; the tests
// point that at this file's .so and start W rank processes on cuda:0.
//
// What it is: the eleven entry points exchange.hip binds, with NCCL's semantics as far as that code relies on them --
//   * point-to-point messages match in order per (source, destination) pair; a send never blocks (every message is a
//     POSIX shared-memory object of its own: device -> host copy by the sender, host -> device by the receiver);
//   * grouped calls run at the outermost ncclGroupEnd: every send first, then the receives, then the all-reduces;
//   * stream order: the source buffer is read after everything queued on `stream` before the call, the destination is
//     complete before the call returns (stricter in time than RCCL, never looser in order);
//   * ncclAllReduce (sum / min / max of f64, f32, i32, i64): every rank sends its vector to every rank, sums in rank order.
// Every wait has a limit (UPSP_SHIM_TIMEOUT_S, default 120 s) and ends in ncclSystemError instead of a hang.
// Nothing in upsp_processing_amd/ knows about this file; no product path loads it by itself.
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct ShimComm;

namespace {

constexpr int kMaxRanks = 16;
constexpr uint64_t kMagic = 0x4d49485350535055ull;      // "UPSPSHIM"

struct Slot {                                   // one direction of one pair
    std::atomic<uint64_t> posted;               // messages the source has completed
    std::atomic<uint64_t> consumed;             // messages the destination has taken (the source stays < 64 ahead: bytes[] is a ring)
    std::atomic<uint64_t> bytes[64];            // size of message (seq % 64): the receiver checks it against its own count
};
struct Ctrl {
    std::atomic<uint32_t> arrived;
    std::atomic<uint32_t> left;
    Slot slot[kMaxRanks][kMaxRanks];            // [src][dst]
};

struct Op {
    int kind;                                   // 0 send, 1 recv, 2 all-reduce
    const void *src;
    void *dst;
    size_t count;
    ncclDataType_t dt;
    ncclRedOp_t op;
    int peer;
    hipStream_t st;
};

double now_s()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
double limit_s()
{
    const char *e = getenv("UPSP_SHIM_TIMEOUT_S");
    const double v = e ? atof(e) : 0.0;
    return v > 0.0 ? v : 120.0;
}
void nap()
{
    timespec ts = {0, 50000};
    nanosleep(&ts, nullptr);
}
size_t type_size(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

thread_local int g_depth = 0;
thread_local std::vector<std::pair<ShimComm *, Op>> g_queue;

}  // namespace

struct ShimComm {
    int rank = 0, world = 1;
    std::string name;
    Ctrl *ctrl = nullptr;
    uint64_t sent[kMaxRanks] = {};              // messages sent to / taken from each peer so far
    uint64_t taken[kMaxRanks] = {};
};

namespace {

std::string msg_name(const ShimComm *c, int src, int dst, uint64_t seq)
{
    char b[160];
    std::snprintf(b, sizeof(b), "%s_%d_%d_%llu", c->name.c_str(), src, dst, (unsigned long long)seq);
    return b;
}

ncclResult_t do_send(ShimComm *c, const void *d_src, size_t bytes, int peer, hipStream_t st)
{
    if (peer < 0 || peer >= c->world) return ncclInvalidArgument;
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;       // the producer of the buffer is done
    const uint64_t seq = c->sent[peer];
    const std::string nm = msg_name(c, c->rank, peer, seq);
    Slot &s = c->ctrl->slot[c->rank][peer];
    const double t0 = now_s(), lim = limit_s();
    while (seq - s.consumed.load(std::memory_order_acquire) >= 64) {       // (never in the exchanges' own call pattern)
        if (now_s() - t0 > lim) return ncclSystemError;
        nap();
    }
    if (bytes) {
        const int fd = shm_open(nm.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0) return ncclSystemError;
        if (ftruncate(fd, (off_t)bytes) != 0) { close(fd); shm_unlink(nm.c_str()); return ncclSystemError; }
        void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) { shm_unlink(nm.c_str()); return ncclSystemError; }
        const hipError_t e = hipMemcpy(p, d_src, bytes, hipMemcpyDeviceToHost);
        munmap(p, bytes);
        if (e != hipSuccess) { shm_unlink(nm.c_str()); return ncclUnhandledCudaError; }
    }
    s.bytes[seq % 64].store(bytes, std::memory_order_relaxed);
    s.posted.store(seq + 1, std::memory_order_release);
    c->sent[peer] = seq + 1;
    return ncclSuccess;
}

// the next message of `peer` for this rank -> host vector (dst == nullptr) or device buffer
ncclResult_t do_recv(ShimComm *c, void *d_dst, std::vector<uint8_t> *h_dst, size_t bytes, int peer, hipStream_t st)
{
    if (peer < 0 || peer >= c->world) return ncclInvalidArgument;
    const uint64_t seq = c->taken[peer];
    Slot &s = c->ctrl->slot[peer][c->rank];
    const double t0 = now_s(), lim = limit_s();
    while (s.posted.load(std::memory_order_acquire) <= seq) {
        if (now_s() - t0 > lim) {
            std::fprintf(stderr, "rccl shim: rank %d waited %.0f s for message %llu of rank %d\n", c->rank, lim, (unsigned long long)seq, peer);
            return ncclSystemError;
        }
        nap();
    }
    c->taken[peer] = seq + 1;
    const size_t have = (size_t)s.bytes[seq % 64].load(std::memory_order_relaxed);
    s.consumed.store(seq + 1, std::memory_order_release);
    const std::string nm = msg_name(c, peer, c->rank, seq);
    if (have != bytes) {
        std::fprintf(stderr, "rccl shim: rank %d expects %zu bytes from rank %d, message %llu has %zu\n", c->rank, bytes, peer,
                     (unsigned long long)seq, have);
        if (have) shm_unlink(nm.c_str());
        return ncclInvalidArgument;
    }
    if (!bytes) return ncclSuccess;
    const int fd = shm_open(nm.c_str(), O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    void *p = mmap(nullptr, bytes, PROT_READ, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { shm_unlink(nm.c_str()); return ncclSystemError; }
    ncclResult_t rc = ncclSuccess;
    if (h_dst) {
        h_dst->assign(static_cast<const uint8_t *>(p), static_cast<const uint8_t *>(p) + bytes);
    } else {
        if (hipMemcpyAsync(d_dst, p, bytes, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            rc = ncclUnhandledCudaError;
    }
    munmap(p, bytes);
    shm_unlink(nm.c_str());
    return rc;
}

template <typename T>
void reduce_into(std::vector<uint8_t> &acc, const std::vector<uint8_t> &v, ncclRedOp_t op)
{
    T *a = reinterpret_cast<T *>(acc.data());
    const T *b = reinterpret_cast<const T *>(v.data());
    const size_t n = acc.size() / sizeof(T);
    for (size_t i = 0; i < n; ++i) a[i] = op == ncclSum ? (T)(a[i] + b[i]) : op == ncclMin ? (b[i] < a[i] ? b[i] : a[i]) : (b[i] > a[i] ? b[i] : a[i]);
}

ncclResult_t do_allreduce_post(ShimComm *c, const Op &o)
{
    const size_t bytes = o.count * type_size(o.dt);
    for (int p = 0; p < c->world; ++p) {
        const ncclResult_t rc = do_send(c, o.src, bytes, p, o.st);
        if (rc != ncclSuccess) return rc;
    }
    return ncclSuccess;
}
ncclResult_t do_allreduce_take(ShimComm *c, const Op &o)
{
    if (o.op != ncclSum && o.op != ncclMin && o.op != ncclMax) return ncclInvalidArgument;
    const size_t bytes = o.count * type_size(o.dt);
    std::vector<uint8_t> acc, v;
    for (int p = 0; p < c->world; ++p) {                 // rank order: the same sum on every rank
        const ncclResult_t rc = do_recv(c, nullptr, p == 0 ? &acc : &v, bytes, p, o.st);
        if (rc != ncclSuccess) return rc;
        if (p == 0) continue;
        switch (o.dt) {
        case ncclFloat64: reduce_into<double>(acc, v, o.op); break;
        case ncclFloat32: reduce_into<float>(acc, v, o.op); break;
        case ncclInt32: reduce_into<int32_t>(acc, v, o.op); break;
        case ncclInt64: reduce_into<int64_t>(acc, v, o.op); break;
        default: return ncclInvalidArgument;
        }
    }
    if (bytes && (hipMemcpyAsync(o.dst, acc.data(), bytes, hipMemcpyHostToDevice, o.st) != hipSuccess || hipStreamSynchronize(o.st) != hipSuccess))
        return ncclUnhandledCudaError;
    return ncclSuccess;
}

ncclResult_t run(std::vector<std::pair<ShimComm *, Op>> &q)
{
    // sends (and the posting half of the all-reduces) first: none of them waits for another rank
    for (auto &e : q) {
        ncclResult_t rc = ncclSuccess;
        if (e.second.kind == 0) rc = do_send(e.first, e.second.src, e.second.count * type_size(e.second.dt), e.second.peer, e.second.st);
        if (rc != ncclSuccess) return rc;
    }
    for (auto &e : q)
        if (e.second.kind == 1) {
            const ncclResult_t rc = do_recv(e.first, e.second.dst, nullptr, e.second.count * type_size(e.second.dt), e.second.peer, e.second.st);
            if (rc != ncclSuccess) return rc;
        }
    // all-reduces after the point-to-point messages of the group, one at a time and in call order (per pair the messages of
    // an all-reduce follow the group's sends on every rank alike)
    for (auto &e : q)
        if (e.second.kind == 2) {
            ncclResult_t rc = do_allreduce_post(e.first, e.second);
            if (rc == ncclSuccess) rc = do_allreduce_take(e.first, e.second);
            if (rc != ncclSuccess) return rc;
        }
    return ncclSuccess;
}

ncclResult_t enqueue(ShimComm *c, const Op &o)
{
    if (!c || type_size(o.dt) == 0) return ncclInvalidArgument;
    g_queue.emplace_back(c, o);
    if (g_depth > 0) return ncclSuccess;
    const ncclResult_t rc = run(g_queue);
    g_queue.clear();
    return rc;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    if (!id) return ncclInvalidArgument;
    std::memset(id, 0, sizeof(*id));
    uint64_t rnd[2] = {0, 0};
    FILE *f = std::fopen("/dev/urandom", "rb");
    if (!f || std::fread(rnd, sizeof(rnd), 1, f) != 1) {
        if (f) std::fclose(f);
        return ncclSystemError;
    }
    std::fclose(f);
    std::memcpy(id->internal, &kMagic, 8);
    std::snprintf(id->internal + 8, 100, "/upsp_rccl_shim_%016llx%016llx", (unsigned long long)rnd[0], (unsigned long long)rnd[1]);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    uint64_t magic = 0;
    std::memcpy(&magic, id.internal, 8);
    if (magic != kMagic) return ncclInvalidArgument;      // an id some other RCCL made
    ShimComm *c = new ShimComm();
    c->rank = rank;
    c->world = nranks;
    c->name = std::string(id.internal + 8);
    const int fd = shm_open(c->name.c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)sizeof(Ctrl)) != 0) {     // (fresh objects are zero-filled: every counter starts at 0)
        if (fd >= 0) close(fd);
        delete c;
        return ncclSystemError;
    }
    void *p = mmap(nullptr, sizeof(Ctrl), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) {
        delete c;
        return ncclSystemError;
    }
    c->ctrl = static_cast<Ctrl *>(p);
    c->ctrl->arrived.fetch_add(1, std::memory_order_acq_rel);
    const double t0 = now_s(), lim = limit_s();
    while (c->ctrl->arrived.load(std::memory_order_acquire) < (uint32_t)nranks) {
        if (now_s() - t0 > lim) {
            std::fprintf(stderr, "rccl shim: rank %d of %d: only %u ranks arrived within %.0f s\n", rank, nranks,
                         c->ctrl->arrived.load(), lim);
            munmap(p, sizeof(Ctrl));
            shm_unlink(c->name.c_str());
            delete c;
            return ncclSystemError;
        }
        nap();
    }
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    ShimComm *c = reinterpret_cast<ShimComm *>(comm);
    if (!c) return ncclInvalidArgument;
    // the last rank to leave removes the control block (every mapping stays valid until it is unmapped)
    if (c->ctrl->left.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->world) shm_unlink(c->name.c_str());
    munmap(c->ctrl, sizeof(Ctrl));
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int *count)
{
    if (!comm || !count) return ncclInvalidArgument;
    *count = reinterpret_cast<const ShimComm *>(comm)->world;
    return ncclSuccess;
}

ncclResult_t ncclCommUserRank(const ncclComm_t comm, int *rank)
{
    if (!comm || !rank) return ncclInvalidArgument;
    *rank = reinterpret_cast<const ShimComm *>(comm)->rank;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart()
{
    ++g_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    const ncclResult_t rc = run(g_queue);
    g_queue.clear();
    return rc;
}

ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    return enqueue(reinterpret_cast<ShimComm *>(comm), Op{0, sendbuff, nullptr, count, datatype, ncclSum, peer, stream});
}

ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    return enqueue(reinterpret_cast<ShimComm *>(comm), Op{1, nullptr, recvbuff, count, datatype, ncclSum, peer, stream});
}

ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream)
{
    return enqueue(reinterpret_cast<ShimComm *>(comm), Op{2, sendbuff, recvbuff, count, datatype, op, -1, stream});
}

const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "rccl shim: HIP call failed";
    case ncclSystemError: return "rccl shim: system error (shared memory, or a rank that never showed up)";
    case ncclInvalidArgument: return "rccl shim: invalid argument (or message sizes that disagree between two ranks)";
    case ncclInvalidUsage: return "rccl shim: invalid usage";
    default: return "rccl shim: error";
    }
}

}  // extern "C"
