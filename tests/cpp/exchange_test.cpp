// The C-ABI exchange (include/upsp_gpu.h section 3b) driven the way a C++ psp_process would drive it
// (cpp/exec/psp_process.cpp:707-771, 1866-1872): communicator -> exchange -> set_skipped -> chunks -> finish,
// plus the all-reduce of the accumulators.
//   exchange_test local W         W ranks in this process on one GPU (device-to-device copies in place of the links)
//   exchange_test pixels W        the pixel-series mode (the owners of the nodes run pass B) with W local ranks
//   exchange_test rccl1           a one-rank RCCL communicator (RCCL refuses two ranks on one GPU)
//   exchange_test rccl RANK WORLD IDFILE   one rank of a real multi-GPU job on device RANK (rank 0 writes the id file)
//   exchange_test ranks RANK WORLD IDFILE DEVICE   the same on device DEVICE: with UPSP_RCCL_LIBRARY naming tests/shim's
//                                 stand-in, WORLD rank PROCESSES share one GPU and drive the RCCL branch of csrc/exchange.hip
//                                 (every wire of the node rows, the pixel-series mode placed and in place, two exchanges in turn)
// Truth: series(n, f) = (31 n + 7 f) mod 4096 for the nodes that travel, NaN for every 5th node (no camera sees it).
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "upsp_gpu.h"

#define CHECK(call)                                                                                  \
    do {                                                                                             \
        const int rc_ = (call);                                                                      \
        if (rc_ != 0) {                                                                              \
            std::fprintf(stderr, "%s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #call, rc_, upsp_last_error()); \
            std::exit(1);                                                                            \
        }                                                                                            \
    } while (0)
#define HIPCHECK(call)                                                                   \
    do {                                                                                 \
        const hipError_t e_ = (call);                                                    \
        if (e_ != hipSuccess) {                                                          \
            std::fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            std::exit(1);                                                                \
        }                                                                                \
    } while (0)

static const int64_t N = 1003, F = 517;     // ragged on purpose: neither divides by 2, 3 or 64
static const int K = 3;

static bool skipped(int64_t n) { return n % 5 == 2; }
static unsigned truth(int64_t n, int64_t f) { return (unsigned)((31 * n + 7 * f) % 4096); }

struct RankState {
    upsp_comm *comm = nullptr;
    upsp_exchange *x = nullptr;
    std::vector<void *> chunks;
    double *d_sum = nullptr, *d_sumsq = nullptr;
    float *d_series = nullptr;
    int64_t f0 = 0, nf = 0, n0 = 0, nn = 0;
};

static void submit_all(RankState &r, const uint8_t *d_skipped, int wire)
{
    CHECK(upsp_exchange_set_skipped(r.x, d_skipped, 0, nullptr));
    const int32_t *d_rowmap = nullptr;
    int64_t packed = 0;
    CHECK(upsp_exchange_rows(r.x, &d_rowmap, &packed));
    std::vector<int32_t> rowmap(N);
    HIPCHECK(hipMemcpy(rowmap.data(), d_rowmap, sizeof(int32_t) * N, hipMemcpyDeviceToHost));
    for (int k = 0; k < K; ++k) {
        int64_t c0 = 0, fc = 0;
        CHECK(upsp_exchange_chunk(r.x, k, &c0, &fc));
        const size_t esz = wire == 4 ? 4 : 2;
        std::vector<uint8_t> h((size_t)packed * fc * esz + 1);
        for (int64_t n = 0; n < N; ++n) {
            if (rowmap[n] < 0) continue;
            for (int64_t f = 0; f < fc; ++f) {
                const unsigned v = truth(n, r.f0 + c0 + f);
                if (wire == 4) reinterpret_cast<float *>(h.data())[(size_t)rowmap[n] * fc + f] = (float)v;
                else reinterpret_cast<uint16_t *>(h.data())[(size_t)rowmap[n] * fc + f] = (uint16_t)v;
            }
        }
        void *d = nullptr;
        HIPCHECK(hipMalloc(&d, h.size()));
        HIPCHECK(hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice));
        r.chunks.push_back(d);
        CHECK(upsp_exchange_submit(r.x, d, wire, nullptr));
    }
}

static int check_rank(RankState &r, int world)
{
    std::vector<float> got((size_t)r.nn * F);
    HIPCHECK(hipMemcpy(got.data(), r.d_series, sizeof(float) * got.size(), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int64_t i = 0; i < r.nn; ++i)
        for (int64_t f = 0; f < F; ++f) {
            const float g = got[(size_t)i * F + f];
            const int64_t n = r.n0 + i;
            const bool ok = skipped(n) ? std::isnan(g) : g == (float)truth(n, f);
            if (!ok && bad++ < 5) std::fprintf(stderr, "node %lld frame %lld: got %g\n", (long long)n, (long long)f, g);
        }
    std::vector<double> s(N), ss(N);
    HIPCHECK(hipMemcpy(s.data(), r.d_sum, sizeof(double) * N, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(ss.data(), r.d_sumsq, sizeof(double) * N, hipMemcpyDeviceToHost));
    for (int64_t n = 0; n < N; ++n) {
        const double want = (double)world * (world - 1) / 2.0 + (double)world * n, want2 = 2.0 * world * n;
        if (s[n] != want || ss[n] != want2) {
            if (bad++ < 5) std::fprintf(stderr, "all-reduce node %lld: %g %g (want %g %g)\n", (long long)n, s[n], ss[n], want, want2);
        }
    }
    return bad;
}

static void prepare(RankState &r, int rank)
{
    CHECK(upsp_exchange_create(r.comm, F, N, K, &r.x));
    CHECK(upsp_exchange_layout(r.x, &r.f0, &r.nf, &r.n0, &r.nn));
    std::vector<double> s(N), ss(N);
    for (int64_t n = 0; n < N; ++n) {
        s[n] = rank + (double)n;          // sum over ranks: W (W - 1) / 2 + W n
        ss[n] = 2.0 * n;
    }
    HIPCHECK(hipMalloc(&r.d_sum, sizeof(double) * N));
    HIPCHECK(hipMalloc(&r.d_sumsq, sizeof(double) * N));
    HIPCHECK(hipMemcpy(r.d_sum, s.data(), sizeof(double) * N, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(r.d_sumsq, ss.data(), sizeof(double) * N, hipMemcpyHostToDevice));
    HIPCHECK(hipMalloc(&r.d_series, sizeof(float) * (size_t)(r.nn > 0 ? r.nn : 1) * F));
    HIPCHECK(hipMemset(r.d_series, 0x7b, sizeof(float) * (size_t)(r.nn > 0 ? r.nn : 1) * F));
}

// ---- pixel-series mode: A active pixels, node n reads pixel (7 n) mod A (every 9th node: none), pixel series
// px(k, f) = (13 k + 5 f) mod 4096
static const int64_t A = 211;
static int32_t node_pixel(int64_t n) { return n % 9 == 4 ? -1 : (int32_t)((7 * n) % A); }
static unsigned px(int64_t k, int64_t f) { return (unsigned)((13 * k + 5 * f) % 4096); }

// one rank's chunks of the pixel-series mode: its own compact buffer per chunk (row pitch wider than the chunk, 0xFFFF beyond it)
static void submit_pixel_chunks(RankState &r, int K, int wire, std::vector<void *> &bufs)
{
    for (int k = 0; k < K; ++k) {
        int64_t c0 = 0, fc = 0;
        CHECK(upsp_exchange_chunk(r.x, k, &c0, &fc));
        const unsigned cp = (unsigned)((fc + 63) / 64 * 64 + 64);
        std::vector<uint16_t> h((size_t)A * cp, 0xFFFF);
        for (int64_t a = 0; a < A; ++a)
            for (int64_t f = 0; f < fc; ++f) h[(size_t)a * cp + f] = (uint16_t)px(a, r.f0 + c0 + f);
        void *d = nullptr;
        HIPCHECK(hipMalloc(&d, h.size() * 2));
        HIPCHECK(hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        bufs.push_back(d);
        CHECK(upsp_exchange_submit_pixels(r.x, static_cast<const uint16_t *>(d), cp, wire, nullptr));
    }
}

// series of this rank's nodes over all F frames, NaN rows, and the COMPLETE accumulators (after the all-reduce) against the closed form
static int check_pixel_rank(RankState &r, const int64_t F, const std::vector<int32_t> &nk)
{
    int bad = 0;
    std::vector<float> got((size_t)r.nn * F);
    HIPCHECK(hipMemcpy(got.data(), r.d_series, sizeof(float) * got.size(), hipMemcpyDeviceToHost));
    std::vector<double> s(N), ss(N);
    HIPCHECK(hipMemcpy(s.data(), r.d_sum, sizeof(double) * N, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(ss.data(), r.d_sumsq, sizeof(double) * N, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < r.nn; ++i) {
        const int64_t n = r.n0 + i;
        for (int64_t f = 0; f < F; ++f) {
            const float g = got[(size_t)i * F + f];
            const float want = nk[n] < 0 ? 0.f : (float)px(nk[n], f);
            if (skipped(n) ? !std::isnan(g) : g != want) {
                if (bad++ < 5) std::fprintf(stderr, "pixels: node %lld frame %lld: got %g want %g\n", (long long)n, (long long)f, g, want);
            }
        }
    }
    for (int64_t n = 0; n < N; ++n) {
        double ws = 0, wss = 0;
        for (int64_t f = 0; f < F; ++f) {
            const double v = nk[n] < 0 ? 0.0 : (double)px(nk[n], f);
            ws += v;
            wss += v * v;
        }
        const bool ok = skipped(n) ? (std::isnan(s[n]) && std::isnan(ss[n])) : (s[n] == ws && ss[n] == wss);
        if (!ok && bad++ < 5) std::fprintf(stderr, "pixels: accumulators of node %lld: %g %g (want %g %g)\n", (long long)n, s[n], ss[n], ws, wss);
    }
    return bad;
}

// F, K: frames of the run and chunks per rank.  (240, 1): every rank's share is a multiple of four frames and goes out as ONE
// block per peer -- the owner's pass B then reads the received blocks as they lie (no placing pass); (517, 3): ragged, placed.
static int bad_count_global = 0;
static int run_pixels(int W, const int64_t F, const int K)
{
    int bad = 0;
    std::vector<uint8_t> sk(N);
    std::vector<int32_t> nk(N);
    for (int64_t n = 0; n < N; ++n) {
        sk[n] = skipped(n) ? 1 : 0;
        nk[n] = node_pixel(n);
    }
    uint8_t *d_sk = nullptr;
    int32_t *d_nk = nullptr;
    HIPCHECK(hipMalloc(&d_sk, N));
    HIPCHECK(hipMalloc(&d_nk, sizeof(int32_t) * N));
    HIPCHECK(hipMemcpy(d_sk, sk.data(), N, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(d_nk, nk.data(), sizeof(int32_t) * N, hipMemcpyHostToDevice));
    {
        // a node table that points outside the series buffer (-2: a pixel the candidate map does not hold) is refused
        std::vector<int32_t> bad(nk);
        bad[N / 2] = -2;
        int32_t *d_bad = nullptr;
        HIPCHECK(hipMalloc(&d_bad, sizeof(int32_t) * N));
        HIPCHECK(hipMemcpy(d_bad, bad.data(), sizeof(int32_t) * N, hipMemcpyHostToDevice));
        std::vector<upsp_comm *> c1(1);
        CHECK(upsp_comm_create_local(1, c1.data()));
        upsp_exchange *x1 = nullptr;
        CHECK(upsp_exchange_create(c1[0], F, N, K, &x1));
        if (upsp_exchange_set_pixels(x1, d_bad, d_sk, 0, nullptr) == UPSP_OK) {
            std::fprintf(stderr, "pixels: a node row of -2 was accepted\n");
            ++bad_count_global;
        }
        upsp_exchange_destroy(x1);
        upsp_comm_destroy(c1[0]);
        HIPCHECK(hipFree(d_bad));
    }
    for (int wire : {2, 12}) {
        std::vector<upsp_comm *> comms(W);
        CHECK(upsp_comm_create_local(W, comms.data()));
        std::vector<RankState> ranks(W);
        std::vector<std::vector<void *>> bufs(W);
        for (int r = 0; r < W; ++r) {
            ranks[r].comm = comms[r];
            CHECK(upsp_exchange_create(comms[r], F, N, K, &ranks[r].x));
            CHECK(upsp_exchange_layout(ranks[r].x, &ranks[r].f0, &ranks[r].nf, &ranks[r].n0, &ranks[r].nn));
            HIPCHECK(hipMalloc(&ranks[r].d_sum, sizeof(double) * N));
            HIPCHECK(hipMalloc(&ranks[r].d_sumsq, sizeof(double) * N));
            HIPCHECK(hipMemset(ranks[r].d_sum, 0, sizeof(double) * N));
            HIPCHECK(hipMemset(ranks[r].d_sumsq, 0, sizeof(double) * N));
            HIPCHECK(hipMalloc(&ranks[r].d_series, sizeof(float) * (size_t)(ranks[r].nn > 0 ? ranks[r].nn : 1) * F));
            CHECK(upsp_exchange_set_pixels(ranks[r].x, d_nk, d_sk, 0, nullptr));
        }
        for (int r = 0; r < W; ++r) submit_pixel_chunks(ranks[r], K, wire, bufs[r]);    // every rank submits its chunks
        for (int r = 0; r < W; ++r)
            CHECK(upsp_exchange_finish_pixels(ranks[r].x, ranks[r].d_series, F, ranks[r].d_sum + ranks[r].n0, ranks[r].d_sumsq + ranks[r].n0, nullptr));
        for (int r = 0; r < W; ++r) CHECK(upsp_allreduce_sums(comms[r], ranks[r].d_sum, ranks[r].d_sumsq, N, nullptr));
        HIPCHECK(hipDeviceSynchronize());
        uint64_t tot_s = 0, tot_rows = 0;
        for (int r = 0; r < W; ++r) {
            CHECK(upsp_exchange_verify(ranks[r].x, nullptr));
            bad += check_pixel_rank(ranks[r], F, nk);
            uint64_t sent = 0, recv = 0;
            int64_t rows_out = 0, rows_in = 0;
            CHECK(upsp_exchange_bytes(ranks[r].x, &sent, &recv));
            CHECK(upsp_exchange_pixel_rows(ranks[r].x, &rows_out, &rows_in));
            tot_s += sent;
            tot_rows += (uint64_t)rows_in;
        }
        std::printf("pixels W=%d wire=%d: %s, %llu bytes between ranks, %llu pixel rows for %lld travelling nodes\n", W, wire,
                    bad ? "FAILED" : "ok", (unsigned long long)tot_s, (unsigned long long)tot_rows, (long long)(N - N / 5));
        for (int r = 0; r < W; ++r) {
            upsp_exchange_destroy(ranks[r].x);
            upsp_comm_destroy(comms[r]);
        }
    }
    return bad + bad_count_global;
}

// ---- one rank of a multi-process job (real RCCL between GPUs, or the tests' stand-in between processes on one GPU) ----------
static int read_ids(const char *path, int rank, std::vector<std::vector<uint8_t>> &ids)
{
    const size_t n = ids.size();
    if (rank == 0) {
        for (auto &id : ids) {
            id.resize(128);
            CHECK(upsp_comm_unique_id(id.data()));
        }
        std::string tmp = std::string(path) + ".tmp";
        FILE *f = std::fopen(tmp.c_str(), "wb");
        for (auto &id : ids) std::fwrite(id.data(), 1, 128, f);
        std::fclose(f);
        std::rename(tmp.c_str(), path);
        return 0;
    }
    FILE *f = nullptr;
    for (int i = 0; i < 600 && !(f = std::fopen(path, "rb")); ++i) {
        struct timespec ts = {0, 100000000};
        nanosleep(&ts, nullptr);
    }
    if (!f) { std::fprintf(stderr, "no id file\n"); return 1; }
    for (size_t i = 0; i < n; ++i) {
        ids[i].resize(128);
        if (std::fread(ids[i].data(), 1, 128, f) != 128) { std::fprintf(stderr, "short id file\n"); return 1; }
    }
    std::fclose(f);
    return 0;
}

static void pixel_rank_setup(RankState &r, upsp_comm *comm, int64_t F, int K, const int32_t *d_nk, const uint8_t *d_sk)
{
    r.comm = comm;
    CHECK(upsp_exchange_create(comm, F, N, K, &r.x));
    CHECK(upsp_exchange_layout(r.x, &r.f0, &r.nf, &r.n0, &r.nn));
    HIPCHECK(hipMalloc(&r.d_sum, sizeof(double) * N));
    HIPCHECK(hipMalloc(&r.d_sumsq, sizeof(double) * N));
    HIPCHECK(hipMemset(r.d_sum, 0, sizeof(double) * N));
    HIPCHECK(hipMemset(r.d_sumsq, 0, sizeof(double) * N));
    HIPCHECK(hipMalloc(&r.d_series, sizeof(float) * (size_t)(r.nn > 0 ? r.nn : 1) * F));
    HIPCHECK(hipMemset(r.d_series, 0x7b, sizeof(float) * (size_t)(r.nn > 0 ? r.nn : 1) * F));
    CHECK(upsp_exchange_set_pixels(r.x, d_nk, d_sk, 0, nullptr));
}

static int run_process_rank(int rank, int world, const char *idfile, const uint8_t *d_sk)
{
    std::vector<std::vector<uint8_t>> ids(8);      // one id per communicator: 3 + 4 + 1 below
    if (read_ids(idfile, rank, ids)) return 1;
    int bad = 0, next_id = 0;
    int nr = -1, nw = -1;
    // node rows, every wire (K = 3 ragged chunks of F = 517 frames)
    for (int wire : {4, 2, 12}) {
        RankState r;
        CHECK(upsp_comm_create(ids[next_id++].data(), rank, world, &r.comm));
        CHECK(upsp_comm_rank(r.comm, &nr, &nw));
        if (nr != rank || nw != world) { std::fprintf(stderr, "communicator says rank %d of %d\n", nr, nw); ++bad; }
        prepare(r, rank);
        submit_all(r, d_sk, wire);
        CHECK(upsp_allreduce_sums(r.comm, r.d_sum, r.d_sumsq, N, nullptr));
        CHECK(upsp_exchange_finish(r.x, r.d_series, F, nullptr));
        HIPCHECK(hipDeviceSynchronize());
        CHECK(upsp_exchange_verify(r.x, nullptr));
        bad += check_rank(r, world);
        uint64_t sent = 0, recv = 0;
        CHECK(upsp_exchange_bytes(r.x, &sent, &recv));
        std::printf("ranks %d of %d rows wire=%d: %s, %llu bytes to other ranks\n", rank, world, wire, bad ? "FAILED" : "ok", (unsigned long long)sent);
        upsp_exchange_destroy(r.x);
        upsp_comm_destroy(r.comm);
    }
    // pixel-series mode
    std::vector<int32_t> nk(N);
    for (int64_t n = 0; n < N; ++n) nk[n] = node_pixel(n);
    int32_t *d_nk = nullptr;
    HIPCHECK(hipMalloc(&d_nk, sizeof(int32_t) * N));
    HIPCHECK(hipMemcpy(d_nk, nk.data(), sizeof(int32_t) * N, hipMemcpyHostToDevice));
    struct Shape { int64_t F; int K; int wire; };
    for (const Shape sh : {Shape{517, 3, 2}, Shape{517, 3, 12}, Shape{240, 1, 2}, Shape{240, 1, 12}}) {
        upsp_comm *comm = nullptr;
        CHECK(upsp_comm_create(ids[next_id++].data(), rank, world, &comm));
        RankState r;
        std::vector<void *> bufs;
        pixel_rank_setup(r, comm, sh.F, sh.K, d_nk, d_sk);
        submit_pixel_chunks(r, sh.K, sh.wire, bufs);
        CHECK(upsp_exchange_finish_pixels(r.x, r.d_series, sh.F, r.d_sum + r.n0, r.d_sumsq + r.n0, nullptr));
        CHECK(upsp_allreduce_sums(comm, r.d_sum, r.d_sumsq, N, nullptr));
        HIPCHECK(hipDeviceSynchronize());
        CHECK(upsp_exchange_verify(r.x, nullptr));
        bad += check_pixel_rank(r, sh.F, nk);
        int64_t rows_out = 0, rows_in = 0;
        CHECK(upsp_exchange_pixel_rows(r.x, &rows_out, &rows_in));
        std::printf("ranks %d of %d pixels F=%lld K=%d wire=%d: %s, %lld pixel rows in\n", rank, world, (long long)sh.F, sh.K, sh.wire,
                    bad ? "FAILED" : "ok", (long long)rows_in);
        upsp_exchange_destroy(r.x);
        upsp_comm_destroy(comm);
    }
    // two exchanges in turn on ONE communicator (the N > 1 schedule of bench.py): both submitted before the first is finished,
    // three rounds; the communicator's transfer stream is shared and every finish waits for its own arrival mark only
    {
        upsp_comm *comm = nullptr;
        CHECK(upsp_comm_create(ids[next_id++].data(), rank, world, &comm));
        RankState r[2];
        std::vector<void *> bufs;
        for (int i = 0; i < 2; ++i) pixel_rank_setup(r[i], comm, 240, 1, d_nk, d_sk);
        submit_pixel_chunks(r[0], 1, 2, bufs);
        for (int round = 1; round <= 3; ++round) {
            RankState &cur = r[round & 1], &prev = r[(round - 1) & 1];
            submit_pixel_chunks(cur, 1, 2, bufs);                               // this step's blocks go out ...
            HIPCHECK(hipMemset(prev.d_sum, 0, sizeof(double) * N));
            HIPCHECK(hipMemset(prev.d_sumsq, 0, sizeof(double) * N));
            CHECK(upsp_exchange_finish_pixels(prev.x, prev.d_series, 240, prev.d_sum + prev.n0, prev.d_sumsq + prev.n0, nullptr));   // ... then the previous step's pass B
            CHECK(upsp_allreduce_sums(comm, prev.d_sum, prev.d_sumsq, N, nullptr));
            HIPCHECK(hipDeviceSynchronize());
            bad += check_pixel_rank(prev, 240, nk);
        }
        RankState &last = r[3 & 1];
        HIPCHECK(hipMemset(last.d_sum, 0, sizeof(double) * N));
        HIPCHECK(hipMemset(last.d_sumsq, 0, sizeof(double) * N));
        CHECK(upsp_exchange_finish_pixels(last.x, last.d_series, 240, last.d_sum + last.n0, last.d_sumsq + last.n0, nullptr));
        CHECK(upsp_allreduce_sums(comm, last.d_sum, last.d_sumsq, N, nullptr));
        HIPCHECK(hipDeviceSynchronize());
        bad += check_pixel_rank(last, 240, nk);
        std::printf("ranks %d of %d two exchanges in turn: %s\n", rank, world, bad ? "FAILED" : "ok");
        for (int i = 0; i < 2; ++i) upsp_exchange_destroy(r[i].x);
        upsp_comm_destroy(comm);
    }
    return bad;
}

int main(int argc, char **argv)
{
    const std::string mode = argc > 1 ? argv[1] : "local";
    std::vector<uint8_t> sk(N);
    for (int64_t n = 0; n < N; ++n) sk[n] = skipped(n) ? 1 : 0;
    uint8_t *d_sk = nullptr;
    if (mode == "rccl") {       // select the device before anything else
        HIPCHECK(hipSetDevice(std::atoi(argv[2])));
    }
    if (mode == "ranks") {
        if (argc < 6) { std::fprintf(stderr, "usage: exchange_test ranks RANK WORLD IDFILE DEVICE\n"); return 2; }
        HIPCHECK(hipSetDevice(std::atoi(argv[5])));
    }
    HIPCHECK(hipMalloc(&d_sk, N));
    HIPCHECK(hipMemcpy(d_sk, sk.data(), N, hipMemcpyHostToDevice));
    int bad = 0;
    if (mode == "ranks") return run_process_rank(std::atoi(argv[2]), std::atoi(argv[3]), argv[4], d_sk) ? 1 : 0;
    if (mode == "pixels") {
        const int W = argc > 2 ? std::atoi(argv[2]) : 2;
        return (run_pixels(W, 517, 3) + run_pixels(W, 240, 1)) ? 1 : 0;
    }
    if (mode == "local") {
        const int W = argc > 2 ? std::atoi(argv[2]) : 2;
        for (int wire : {4, 2, 12}) {
            std::vector<upsp_comm *> comms(W);
            CHECK(upsp_comm_create_local(W, comms.data()));
            std::vector<RankState> ranks(W);
            for (int r = 0; r < W; ++r) {
                ranks[r].comm = comms[r];
                prepare(ranks[r], r);
            }
            for (int r = 0; r < W; ++r) submit_all(ranks[r], d_sk, wire);                     // every rank submits ...
            for (int r = 0; r < W; ++r) CHECK(upsp_allreduce_sums(comms[r], ranks[r].d_sum, ranks[r].d_sumsq, N, nullptr));
            for (int r = 0; r < W; ++r) CHECK(upsp_exchange_finish(ranks[r].x, ranks[r].d_series, F, nullptr));   // ... then every rank finishes
            HIPCHECK(hipDeviceSynchronize());
            uint64_t sent = 0, recv = 0, tot_s = 0, tot_r = 0;
            for (int r = 0; r < W; ++r) {
                CHECK(upsp_exchange_verify(ranks[r].x, nullptr));
                bad += check_rank(ranks[r], W);
                CHECK(upsp_exchange_bytes(ranks[r].x, &sent, &recv));
                tot_s += sent;
                tot_r += recv;
            }
            if (tot_s != tot_r) {
                std::fprintf(stderr, "bytes sent %llu != received %llu\n", (unsigned long long)tot_s, (unsigned long long)tot_r);
                ++bad;
            }
            std::printf("local W=%d wire=%d: %s, %llu bytes between ranks\n", W, wire, bad ? "FAILED" : "ok", (unsigned long long)tot_s);
            for (int r = 0; r < W; ++r) {
                upsp_exchange_destroy(ranks[r].x);
                upsp_comm_destroy(comms[r]);
            }
        }
    } else {
        int rank = 0, world = 1;
        uint8_t id[128];
        if (mode == "rccl") {
            rank = std::atoi(argv[2]);
            world = std::atoi(argv[3]);
            const char *path = argv[4];
            if (rank == 0) {
                CHECK(upsp_comm_unique_id(id));
                std::string tmp = std::string(path) + ".tmp";
                FILE *f = std::fopen(tmp.c_str(), "wb");
                std::fwrite(id, 1, 128, f);
                std::fclose(f);
                std::rename(tmp.c_str(), path);
            } else {
                FILE *f = nullptr;
                for (int i = 0; i < 600 && !(f = std::fopen(path, "rb")); ++i) {
                    struct timespec ts = {0, 100000000};
                    nanosleep(&ts, nullptr);
                }
                if (!f || std::fread(id, 1, 128, f) != 128) { std::fprintf(stderr, "no id file\n"); return 1; }
                std::fclose(f);
            }
        } else {
            CHECK(upsp_comm_unique_id(id));
        }
        for (int wire : {4, 2, 12}) {
            RankState r;
            if (mode != "rccl" && wire != 4) CHECK(upsp_comm_unique_id(id));      // (an id makes one communicator)
            CHECK(upsp_comm_create(id, rank, world, &r.comm));
            prepare(r, rank);
            submit_all(r, d_sk, wire);
            CHECK(upsp_allreduce_sums(r.comm, r.d_sum, r.d_sumsq, N, nullptr));
            CHECK(upsp_exchange_finish(r.x, r.d_series, F, nullptr));
            HIPCHECK(hipDeviceSynchronize());
            CHECK(upsp_exchange_verify(r.x, nullptr));
            bad += check_rank(r, world);
            uint64_t sent = 0, recv = 0;
            CHECK(upsp_exchange_bytes(r.x, &sent, &recv));
            std::printf("rccl rank %d of %d wire=%d: %s, %llu bytes to other ranks\n", rank, world, wire, bad ? "FAILED" : "ok",
                        (unsigned long long)sent);
            upsp_exchange_destroy(r.x);
            upsp_comm_destroy(r.comm);
            if (mode == "rccl") break;      // (one communicator per id)
        }
    }
    return bad ? 1 : 0;
}
