// Store-shape probe, second pass: whole-row writers for the node-major series (MI355X).
// Each variant is timed in 3 interleaved rounds (5 launches each); min and median reported.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ void st4(float *p, v4f v)
{
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(p));
    else *reinterpret_cast<v4f *>(p) = v;
}
// torch-like fill: one-shot, a workgroup writes a contiguous 16-KB chunk (4 stores per lane, 4 KB apart)
template <bool NT>
__global__ void __launch_bounds__(256) chunk_fill_kernel(float *out, size_t n)
{
    const size_t base = (size_t)blockIdx.x * 4096 + 4 * threadIdx.x;
    const v4f v = {0.f, 1.f, 2.f, 3.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (base + 1024 * i + 3 < n) st4<NT>(out + base + 1024 * i, v);
}
// wave per row, R rows per wave (consecutive), one-shot
template <bool NT, int R>
__global__ void __launch_bounds__(256) rows_wave_kernel(float *out, unsigned nrows, int rowf, long long ld)
{
    const unsigned row0 = (blockIdx.x * 4u + (threadIdx.x >> 6)) * R;
    const int lane = threadIdx.x & 63;
    const v4f v = {(float)row0, 1.f, 2.f, 3.f};
#pragma unroll 1
    for (int r = 0; r < R; ++r) {
        const unsigned row = row0 + r;
        if (row >= nrows) return;
        float *dst = out + (long long)row * ld;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = 4 * lane + 256 * i;
            if (f + 3 < rowf) st4<NT>(dst + f, v);
        }
    }
}
// workgroup per R consecutive rows: every lane one 16-B store per row (4 KB sweep per instruction)
template <bool NT, int R>
__global__ void __launch_bounds__(256) rows_wg_kernel(float *out, unsigned nrows, int rowf, long long ld)
{
    const unsigned row0 = blockIdx.x * R;
    const v4f v = {(float)row0, 1.f, 2.f, 3.f};
    const int f = 4 * threadIdx.x;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned row = row0 + r;
        if (row < nrows && f + 3 < rowf) st4<NT>(out + (long long)row * ld + f, v);
    }
}
// mixed: 60 % of the rows constant (no loads), 40 % read a 2-byte-per-sample source row (cache resident)
template <bool NT>
__global__ void __launch_bounds__(256) rows_mixed_kernel(float *out, const uint2 *src, unsigned nsrc, unsigned nrows, int rowf, long long ld)
{
    const unsigned row = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const int lane = threadIdx.x & 63;
    float *dst = out + (long long)row * ld;
    const bool vis = (row % 5u) < 2u;
    if (!vis) {
        const float q = __builtin_nanf("");
        const v4f v = {q, q, q, q};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = 4 * lane + 256 * i;
            if (f + 3 < rowf) st4<NT>(dst + f, v);
        }
        return;
    }
    const uint2 *s = src + (size_t)((row * 2654435761u) % nsrc) * 256;   // 1024 u16 per source row
    uint2 w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = s[lane + 64 * i];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = 4 * lane + 256 * i;
        const v4f v = {(float)(w[i].x & 0xFFFFu), (float)(w[i].x >> 16), (float)(w[i].y & 0xFFFFu), (float)(w[i].y >> 16)};
        if (f + 3 < rowf) st4<NT>(dst + f, v);
    }
}

struct Var { std::string name; double bytes; std::function<void()> fn; std::vector<float> ms; };

int main()
{
    const unsigned N = 500958; const int F = 1000;
    float *out; CK(hipMalloc(&out, (size_t)N * 1152 * 4 + (1 << 20)));
    const unsigned nsrc = 66000;
    uint2 *src; CK(hipMalloc(&src, (size_t)nsrc * 2048)); CK(hipMemset(src, 1, (size_t)nsrc * 2048));
    const double rb = (double)N * F * 4;
    std::vector<Var> vars;
    auto add = [&](std::string n, double b, std::function<void()> f) { vars.push_back({n, b, f, {}}); };
    add("chunk fill 16 KB/wg nt", rb, [&] { chunk_fill_kernel<true><<<(unsigned)(((size_t)N * F + 4095) / 4096), 256>>>(out, (size_t)N * F); });
    add("chunk fill 16 KB/wg plain", rb, [&] { chunk_fill_kernel<false><<<(unsigned)(((size_t)N * F + 4095) / 4096), 256>>>(out, (size_t)N * F); });
    for (long long ld : {1000ll, 1024ll, 1088ll}) {
        std::string s = ", ld " + std::to_string(ld);
        add("wave/row nt" + s, rb, [=] { rows_wave_kernel<true, 1><<<(N + 3) / 4, 256>>>(out, N, F, ld); });
        add("wave/row plain" + s, rb, [=] { rows_wave_kernel<false, 1><<<(N + 3) / 4, 256>>>(out, N, F, ld); });
        add("wave/4 rows nt" + s, rb, [=] { rows_wave_kernel<true, 4><<<(N + 15) / 16, 256>>>(out, N, F, ld); });
        add("wave/16 rows nt" + s, rb, [=] { rows_wave_kernel<true, 16><<<(N + 63) / 64, 256>>>(out, N, F, ld); });
        add("wg/row nt" + s, rb, [=] { rows_wg_kernel<true, 1><<<N, 256>>>(out, N, F, ld); });
        add("wg/row plain" + s, rb, [=] { rows_wg_kernel<false, 1><<<N, 256>>>(out, N, F, ld); });
        add("wg/4 rows nt" + s, rb, [=] { rows_wg_kernel<true, 4><<<(N + 3) / 4, 256>>>(out, N, F, ld); });
        add("wg/4 rows plain" + s, rb, [=] { rows_wg_kernel<false, 4><<<(N + 3) / 4, 256>>>(out, N, F, ld); });
        add("mixed wave/row nt" + s, rb, [=] { rows_mixed_kernel<true><<<(N + 3) / 4, 256>>>(out, src, nsrc, N, F, ld); });
        add("mixed wave/row plain" + s, rb, [=] { rows_mixed_kernel<false><<<(N + 3) / 4, 256>>>(out, src, nsrc, N, F, ld); });
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto &v : vars) { v.fn(); }
    CK(hipDeviceSynchronize());
    for (int round = 0; round < 3; ++round)
        for (auto &v : vars) {
            v.fn();
            for (int r = 0; r < 5; ++r) {
                CK(hipEventRecord(e0, 0)); v.fn(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); v.ms.push_back(ms);
            }
        }
    CK(hipGetLastError());
    for (auto &v : vars) {
        std::sort(v.ms.begin(), v.ms.end());
        const float mn = v.ms.front(), med = v.ms[v.ms.size() / 2];
        printf("%-36s min %7.3f ms (%5.2f TB/s)  median %7.3f ms (%5.2f TB/s)\n", v.name.c_str(), mn, v.bytes / mn / 1e9, med, v.bytes / med / 1e9);
    }
    return 0;
}
