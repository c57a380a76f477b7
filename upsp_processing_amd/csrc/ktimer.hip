// Per-kernel timing registry: hipEvent pairs recorded on the stream the kernel is
// launched on; resolved at report time (after the caller has synchronised).
#include "ktimer.h"

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "upsp_internal.h"

namespace upsp {
namespace {
struct Span {
    hipEvent_t a, b;
};
struct Entry {
    std::vector<Span> spans;
};
std::mutex g_mu;
bool g_on = false;
std::map<std::string, Entry> g_entries;
std::vector<std::string> g_order;
std::string g_current;
hipEvent_t g_start;
// launch order of the spans (UPSP_TRACE_TIMELINE: upsp_timing_report prints where every span sat on the device's time line)
struct Seq {
    std::string name;
    Span span;
};
std::vector<Seq> g_seq;
}  // namespace

// roctx ranges (rocprofv3 --marker-trace shows them as a time line of phases): libroctx64 is looked up at first use,
// not linked; without it the labels are wall-clock lines only.
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
};
Roctx &roctx()
{
    static Roctx r;
    static std::once_flag once;
    std::call_once(once, [] {
        void *p = dlsym(RTLD_DEFAULT, "roctxRangePushA");
        void *h = nullptr;
        if (!p) {
            h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
            if (h) p = dlsym(h, "roctxRangePushA");
        }
        void *q = dlsym(RTLD_DEFAULT, "roctxRangePop");
        if (!q && h) q = dlsym(h, "roctxRangePop");
        if (p && q) {
            r.push = reinterpret_cast<int (*)(const char *)>(p);
            r.pop = reinterpret_cast<int (*)()>(q);
        }
    });
    return r;
}
bool roctx_wanted()
{
    static const bool on = std::getenv("UPSP_ROCTX") != nullptr;
    return on;
}

struct Phase {
    std::string label;
    std::chrono::steady_clock::time_point t0;
};
std::vector<Phase> g_phases;
std::chrono::steady_clock::time_point g_base;
bool g_have_base = false;

bool ktimer_on() { return g_on; }

void ktimer_begin(const char *name, hipStream_t st)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (roctx_wanted() && roctx().push) roctx().push(name);
    g_current = name;
    if (hipEventCreate(&g_start) != hipSuccess) return;
    (void)hipEventRecord(g_start, st);
}

void ktimer_end(hipStream_t st)
{
    std::lock_guard<std::mutex> lk(g_mu);
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, st);
    if (!g_entries.count(g_current)) g_order.push_back(g_current);
    g_entries[g_current].spans.push_back({g_start, e});
    g_seq.push_back({g_current, {g_start, e}});
    if (roctx_wanted() && roctx().pop) roctx().pop();
}
}  // namespace upsp

using namespace upsp;

extern "C" {

int upsp_timing_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_mu);
    g_on = on != 0;
    if (!g_on) return UPSP_OK;
    for (auto &kv : g_entries)
        for (auto &s : kv.second.spans) {
            (void)hipEventDestroy(s.a);
            (void)hipEventDestroy(s.b);
        }
    g_entries.clear();
    g_order.clear();
    g_seq.clear();
    return UPSP_OK;
}

int upsp_phase_begin(const char *label)
{
    if (!label) return fail(UPSP_ERR_INVALID, "null label");
    std::lock_guard<std::mutex> lk(g_mu);
    if (roctx().push) roctx().push(label);
    const auto now = std::chrono::steady_clock::now();
    if (!g_have_base) {
        g_base = now;
        g_have_base = true;
    }
    g_phases.push_back({label, now});
    return UPSP_OK;
}

int upsp_phase_end(double *seconds)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_phases.empty()) return fail(UPSP_ERR_INVALID, "upsp_phase_end without upsp_phase_begin");
    if (roctx().pop) roctx().pop();
    const auto now = std::chrono::steady_clock::now();
    const Phase ph = g_phases.back();
    g_phases.pop_back();
    const double dt = std::chrono::duration<double>(now - ph.t0).count();
    if (seconds) *seconds = dt;
    // the reference's timedBarrierPoint line (cpp/exec/psp_process.cpp:585-606), without the barrier
    static const bool print = std::getenv("UPSP_PHASE_TIMES") != nullptr;
    if (print)
        std::fprintf(stderr, "+++ %s   [total elapsed:%g,  this phase:%g]\n", ph.label.c_str(),
                     std::chrono::duration<double>(now - g_base).count(), dt);
    return UPSP_OK;
}

int upsp_timing_report(char *buf, size_t cap)
{
    if (!buf || cap == 0) return fail(UPSP_ERR_INVALID, "null buffer");
    std::lock_guard<std::mutex> lk(g_mu);
    std::string out;
    for (const auto &name : g_order) {
        const Entry &e = g_entries[name];
        double total = 0;
        int n = 0;
        std::vector<float> each;
        for (const auto &s : e.spans) {
            float ms = 0;
            if (hipEventSynchronize(s.b) == hipSuccess && hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
                total += ms;
                each.push_back(ms);
                ++n;
            }
        }
        std::sort(each.begin(), each.end());
        // name, spans, total ms, then the spread of the spans: min, median, max
        char line[320];
        std::snprintf(line, sizeof(line), "%s %d %.6f %.6f %.6f %.6f\n", name.c_str(), n, total,
                      n ? each.front() : 0.f, n ? each[n / 2] : 0.f, n ? each.back() : 0.f);
        out += line;
    }
    if (std::getenv("UPSP_TRACE_TIMELINE") && !g_seq.empty()) {
        // start and duration of every span relative to the first one: what an unprofiled run's device time line looked like
        for (const auto &q : g_seq) {
            float t0 = 0, dt = 0;
            if (hipEventSynchronize(q.span.b) != hipSuccess) continue;
            if (hipEventElapsedTime(&t0, g_seq.front().span.a, q.span.a) != hipSuccess) continue;
            if (hipEventElapsedTime(&dt, q.span.a, q.span.b) != hipSuccess) continue;
            std::fprintf(stderr, "timeline %10.1f us  +%8.1f us  %s\n", t0 * 1e3, dt * 1e3, q.name.c_str());
        }
    }
    if (out.size() + 1 > cap) return fail(UPSP_ERR_INVALID, "timing report buffer too small");
    std::memcpy(buf, out.c_str(), out.size() + 1);
    return UPSP_OK;
}
}


// ---- device copy / fill probe: the measured HBM rate the roofline fractions are also quoted against -----------------
// "denominator: measured peak of a rocprof'd device copy kernel on gfx950".  A streaming float4 copy, 16 B per
// lane and trip, and its store half alone; the launch shape is not guessed: non-temporal and plain accesses at 8, 16 and 32
// workgroups per CU are each timed and the FASTEST is what is reported.
namespace upsp {
namespace {
typedef float v4f_probe __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ void __launch_bounds__(256) copy_probe_kernel(const v4f_probe *__restrict__ src, v4f_probe *__restrict__ dst, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
        else dst[i] = src[i];
    }
}
template <bool NT>
__global__ void __launch_bounds__(256) fill_probe_kernel(v4f_probe *__restrict__ dst, size_t n16)
{
    const v4f_probe v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        if (NT) __builtin_nontemporal_store(v, dst + i);
        else dst[i] = v;
    }
}
}  // namespace
}  // namespace upsp

extern "C" int upsp_copy_probe(const void *d_src, void *d_dst, size_t bytes, int reps, float *ms_per_rep, void *stream)
{
    using namespace upsp;
    if (!d_dst || bytes < 16 || reps < 1 || !ms_per_rep) return fail(UPSP_ERR_INVALID, "bad argument");
    if (((reinterpret_cast<size_t>(d_src) | reinterpret_cast<size_t>(d_dst)) & 15) != 0) return fail(UPSP_ERR_INVALID, "copy probe: 16-byte aligned buffers");
    hipStream_t st = (hipStream_t)stream;
    int dev = 0, cus = 256;
    UPSP_HIP_CHECK(hipGetDevice(&dev));
    UPSP_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const size_t n16 = bytes / 16;
    hipEvent_t a = nullptr, b = nullptr;
    UPSP_HIP_CHECK(hipEventCreate(&a));
    UPSP_HIP_CHECK(hipEventCreate(&b));
    float best = 0.f;
    hipError_t e = hipSuccess;
    for (int shape = 0; shape < 6 && e == hipSuccess; ++shape) {
        const bool nt = shape < 3;
        const unsigned grid = (unsigned)std::min<size_t>((n16 + 255) / 256, (size_t)cus * (8u << (shape % 3)));
        auto launch = [&] {
            const v4f_probe *s4 = static_cast<const v4f_probe *>(d_src);
            v4f_probe *d4 = static_cast<v4f_probe *>(d_dst);
            if (d_src && nt) hipLaunchKernelGGL(copy_probe_kernel<true>, dim3(grid), dim3(256), 0, st, s4, d4, n16);
            else if (d_src) hipLaunchKernelGGL(copy_probe_kernel<false>, dim3(grid), dim3(256), 0, st, s4, d4, n16);
            else if (nt) hipLaunchKernelGGL(fill_probe_kernel<true>, dim3(grid), dim3(256), 0, st, d4, n16);
            else hipLaunchKernelGGL(fill_probe_kernel<false>, dim3(grid), dim3(256), 0, st, d4, n16);
        };
        launch();                               // (untimed: first touch, code load)
        e = hipEventRecord(a, st);
        for (int i = 0; i < reps && e == hipSuccess; ++i) launch();
        if (e == hipSuccess) e = hipEventRecord(b, st);
        if (e == hipSuccess) e = hipEventSynchronize(b);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
        if (e == hipSuccess && ms > 0.f && (best == 0.f || ms < best)) best = ms;
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    if (e != hipSuccess) return fail(UPSP_ERR_HIP, std::string("copy probe: ") + hipGetErrorString(e));
    *ms_per_rep = best / (float)reps;
    return UPSP_OK;
}


// ---- read-only / write-only probes in the product kernels' own access shapes (round 6) -----------------------------------------------
// The copy probe above mixes a read and a write stream and is SLOWER than the frame loop's passes (5.3 TB/s against 6.0-6.6 for
// pass A, 5.5-5.6 for pass B: a fraction above 1).  Pass A is a pure read stream (non-temporal 16-byte loads, several in flight per
// lane before the first use, one-wave or four-wave workgroups), pass B a pure write stream (a workgroup sweeping whole 4-KB row
// pieces with 16-byte non-temporal stores): each is divided by the probe of ITS shape.  The fastest of a few launch shapes is
// reported, like above.
namespace upsp {
namespace {
template <int UNROLL, int THREADS>
__global__ void __launch_bounds__(THREADS) read_probe_kernel(const v4f_probe *__restrict__ src, size_t n16, unsigned *sink)
{
    // a workgroup takes UNROLL consecutive pieces of THREADS x 16 bytes: all loads issued before the first use
    const size_t base = (size_t)blockIdx.x * (UNROLL * THREADS) + threadIdx.x;
    v4f_probe v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const size_t i = base + (size_t)u * THREADS;
        v[u] = i < n16 ? __builtin_nontemporal_load(src + i) : v4f_probe{0.f, 0.f, 0.f, 0.f};
    }
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    if (acc == 1.2345e38f) sink[0] = 1u;        // (keeps the loads alive; practically never taken)
}
template <int ROWS>
__global__ void __launch_bounds__(256) write_probe_kernel(v4f_probe *__restrict__ dst, size_t n16)
{
    // pass B's shape: a workgroup writes ROWS consecutive 4-KB pieces, one 16-byte non-temporal store per lane and piece
    const size_t base = (size_t)blockIdx.x * (ROWS * 256) + threadIdx.x;
    const v4f_probe v = {1.f, 2.f, 3.f, (float)blockIdx.x};
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const size_t i = base + (size_t)r * 256;
        if (i < n16) __builtin_nontemporal_store(v, dst + i);
    }
}
}  // namespace
}  // namespace upsp

extern "C" int upsp_bandwidth_probe(int kind, void *d_buf, size_t bytes, int reps, float *ms_per_rep, void *stream)
{
    using namespace upsp;
    if (!d_buf || bytes < 16 || reps < 1 || !ms_per_rep || (kind != 0 && kind != 1)) return fail(UPSP_ERR_INVALID, "bad argument");
    if ((reinterpret_cast<size_t>(d_buf) & 15) != 0) return fail(UPSP_ERR_INVALID, "bandwidth probe: 16-byte aligned buffer");
    hipStream_t st = (hipStream_t)stream;
    const size_t n16 = bytes / 16;
    unsigned *sink = nullptr;
    UPSP_HIP_CHECK(hipMalloc(&sink, sizeof(unsigned)));
    hipEvent_t a = nullptr, b = nullptr;
    hipError_t e = hipEventCreate(&a);
    if (e == hipSuccess) e = hipEventCreate(&b);
    float best = 0.f;
    v4f_probe *p4 = static_cast<v4f_probe *>(d_buf);
    for (int shape = 0; shape < 4 && e == hipSuccess; ++shape) {
        auto grid = [&](size_t per_block) { return dim3((unsigned)((n16 + per_block - 1) / per_block)); };
        auto launch = [&] {
            if (kind == 0) {
                if (shape == 0) hipLaunchKernelGGL((read_probe_kernel<4, 64>), grid(4 * 64), dim3(64), 0, st, (const v4f_probe *)p4, n16, sink);
                else if (shape == 1) hipLaunchKernelGGL((read_probe_kernel<8, 64>), grid(8 * 64), dim3(64), 0, st, (const v4f_probe *)p4, n16, sink);
                else if (shape == 2) hipLaunchKernelGGL((read_probe_kernel<4, 256>), grid(4 * 256), dim3(256), 0, st, (const v4f_probe *)p4, n16, sink);
                else hipLaunchKernelGGL((read_probe_kernel<8, 256>), grid(8 * 256), dim3(256), 0, st, (const v4f_probe *)p4, n16, sink);
            } else {
                if (shape == 0) hipLaunchKernelGGL((write_probe_kernel<1>), grid(256), dim3(256), 0, st, p4, n16);
                else if (shape == 1) hipLaunchKernelGGL((write_probe_kernel<4>), grid(4 * 256), dim3(256), 0, st, p4, n16);
                else if (shape == 2) hipLaunchKernelGGL((write_probe_kernel<8>), grid(8 * 256), dim3(256), 0, st, p4, n16);
                else hipLaunchKernelGGL((write_probe_kernel<16>), grid(16 * 256), dim3(256), 0, st, p4, n16);
            }
        };
        launch();                               // (untimed: first touch, code load)
        e = hipEventRecord(a, st);
        for (int i = 0; i < reps && e == hipSuccess; ++i) launch();
        if (e == hipSuccess) e = hipEventRecord(b, st);
        if (e == hipSuccess) e = hipEventSynchronize(b);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
        if (e == hipSuccess && ms > 0.f && (best == 0.f || ms < best)) best = ms;
    }
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
    (void)hipFree(sink);
    if (e != hipSuccess) return fail(UPSP_ERR_HIP, std::string("bandwidth probe: ") + hipGetErrorString(e));
    *ms_per_rep = best / (float)reps;
    return UPSP_OK;
}
