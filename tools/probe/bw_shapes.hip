// Micro-benchmark of the store / load shapes the frame loop could use (MI355X).
//   hipcc -O3 --offload-arch=gfx950 tools/probe/bw_shapes.hip -o gpurun_out/bw_shapes && gpurun_out/bw_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

// ---- stores -------------------------------------------------------------------------------
// wave per row: row of `rowf` floats at pitch `ld` floats; lane l writes 16 B at 4 l + 256 i
template <bool NT>
__global__ void __launch_bounds__(256) rows_wave_kernel(float *out, unsigned nrows, int rowf, long long ld)
{
    const unsigned row = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const int lane = threadIdx.x & 63;
    float *dst = out + (long long)row * ld;
    const v4f v = {(float)row, 1.f, 2.f, 3.f};
    for (int f = 4 * lane; f + 3 < rowf; f += 256) {
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(dst + f));
        else *reinterpret_cast<v4f *>(dst + f) = v;
    }
}
// persistent variant: grid-stride over rows
template <bool NT>
__global__ void __launch_bounds__(256) rows_wave_persist_kernel(float *out, unsigned nrows, int rowf, long long ld)
{
    const int lane = threadIdx.x & 63;
    for (unsigned row = blockIdx.x * 4u + (threadIdx.x >> 6); row < nrows; row += gridDim.x * 4u) {
        float *dst = out + (long long)row * ld;
        const v4f v = {(float)row, 1.f, 2.f, 3.f};
        for (int f = 4 * lane; f + 3 < rowf; f += 256) {
            if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(dst + f));
            else *reinterpret_cast<v4f *>(dst + f) = v;
        }
    }
}
// workgroup per row-block: a workgroup of 256 lanes writes RB consecutive rows, one 4-KB (1024 floats) sweep per instruction
template <bool NT, int RB>
__global__ void __launch_bounds__(256) rows_wg_kernel(float *out, unsigned nrows, int rowf, long long ld)
{
    const unsigned row0 = blockIdx.x * RB;
    const v4f v = {(float)row0, 1.f, 2.f, 3.f};
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const unsigned row = row0 + r;
        if (row >= nrows) return;
        float *dst = out + (long long)row * ld;
        const int f = 4 * threadIdx.x;
        if (f + 3 < rowf) {
            if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(dst + f));
            else *reinterpret_cast<v4f *>(dst + f) = v;
        }
    }
}
// current shape: 16 lanes per node write a 1-KB piece (4 x 256 B, 64 frames apart ... contiguous 1 KB) of rows
template <bool NT>
__global__ void __launch_bounds__(256) pieces_kernel(float *out, unsigned nrows, int col0, long long ld)
{
    const int lane = threadIdx.x & 63, grp = lane >> 4, gl = lane & 15;
    const unsigned wbase = (blockIdx.x * 4u + (threadIdx.x >> 6)) * 16u;
    const v4f v = {(float)wbase, 1.f, 2.f, 3.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned row = wbase + 4 * i + grp;
        if (row >= nrows) continue;
        float *dst = out + (long long)row * ld + col0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(dst + 64 * c + 4 * gl));
            else *reinterpret_cast<v4f *>(dst + 64 * c + 4 * gl) = v;
        }
    }
}
template <bool NT>
__global__ void __launch_bounds__(256) fill_kernel(float *out, size_t n4)
{
    const v4f v = {0.f, 1.f, 2.f, 3.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(out) + i);
        else reinterpret_cast<v4f *>(out)[i] = v;
    }
}
// ---- loads --------------------------------------------------------------------------------
// pass-A shape: a workgroup owns a tile of TILE bytes of every frame of a 64-frame group; wave w
// reads frames w, w+4, ...; W = bytes per lane (4, 8, 16)
template <int W, bool NT>
__global__ void __launch_bounds__(256) tile_read_kernel(const unsigned char *frames, size_t fbytes, int nframes, unsigned *sink)
{
    constexpr int TILE = 64 * W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t off = (size_t)blockIdx.x * TILE + (size_t)lane * W;
    const unsigned char *base = frames + (size_t)blockIdx.y * 64 * fbytes + off;
    unsigned acc = 0;
    if (W == 4) {
        unsigned v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const unsigned *p = reinterpret_cast<const unsigned *>(base + (size_t)(wave + 4 * i) * fbytes);
            v[i] = NT ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc |= v[i];
    } else if (W == 8) {
        v2u v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const v2u *p = reinterpret_cast<const v2u *>(base + (size_t)(wave + 4 * i) * fbytes);
            v[i] = NT ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc |= v[i].x | v[i].y;
    } else {
        v4u v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const v4u *p = reinterpret_cast<const v4u *>(base + (size_t)(wave + 4 * i) * fbytes);
            v[i] = NT ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc |= v[i].x | v[i].y | v[i].z | v[i].w;
    }
    if (acc == 0xFFFFFFFFu) sink[0] = acc;   // never (data is < 4096 per u16)
}
template <bool NT>
__global__ void __launch_bounds__(256) linear_read_kernel(const v4u *in, size_t n4, unsigned *sink)
{
    unsigned acc = 0;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        v4u a = NT ? __builtin_nontemporal_load(in + i) : in[i];
        v4u b = NT ? __builtin_nontemporal_load(in + i + stride) : in[i + stride];
        v4u c = NT ? __builtin_nontemporal_load(in + i + 2 * stride) : in[i + 2 * stride];
        v4u d = NT ? __builtin_nontemporal_load(in + i + 3 * stride) : in[i + 3 * stride];
        acc |= a.x | a.y | a.z | a.w | b.x | b.y | b.z | b.w | c.x | c.y | c.z | c.w | d.x | d.y | d.z | d.w;
    }
    for (; i < n4; i += stride) { v4u a = in[i]; acc |= a.x | a.y | a.z | a.w; }
    if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

template <typename F>
static void timeit(const char *name, double bytes, F launch)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); launch();
    CK(hipDeviceSynchronize());
    float best = 1e30f, tot = 0;
    const int reps = 7;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, 0));
        launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best; tot += ms;
    }
    CK(hipGetLastError());
    printf("%-52s avg %8.3f ms  best %8.3f ms  %6.2f TB/s (best)\n", name, tot / reps, best, bytes / best / 1e9);
    fflush(stdout);
}

int main()
{
    const unsigned N = 500958; const int F = 1000;
    const size_t maxld = 1152;
    float *out; CK(hipMalloc(&out, (size_t)N * maxld * 4 + 4096));
    unsigned *sink; CK(hipMalloc(&sink, 64));
    const double rowbytes = (double)N * F * 4;
    printf("== stores: %u rows x %d floats (%.2f GB)\n", N, F, rowbytes / 1e9);
    timeit("fill contiguous nt, grid 4096", rowbytes, [&] { fill_kernel<true><<<4096, 256>>>(out, (size_t)N * F / 4); });
    timeit("fill contiguous plain, grid 4096", rowbytes, [&] { fill_kernel<false><<<4096, 256>>>(out, (size_t)N * F / 4); });
    timeit("fill contiguous nt, grid 1024", rowbytes, [&] { fill_kernel<true><<<1024, 256>>>(out, (size_t)N * F / 4); });
    timeit("fill contiguous nt, grid 16384", rowbytes, [&] { fill_kernel<true><<<16384, 256>>>(out, (size_t)N * F / 4); });
    for (long long ld : {1000ll, 1024ll, 1056ll, 1088ll}) {
        char nm[128];
        snprintf(nm, sizeof nm, "wave/row nt, ld %lld", ld);
        timeit(nm, rowbytes, [&] { rows_wave_kernel<true><<<(N + 3) / 4, 256>>>(out, N, F, ld); });
        snprintf(nm, sizeof nm, "wave/row plain, ld %lld", ld);
        timeit(nm, rowbytes, [&] { rows_wave_kernel<false><<<(N + 3) / 4, 256>>>(out, N, F, ld); });
        snprintf(nm, sizeof nm, "wave/row persistent(2048 wg) nt, ld %lld", ld);
        timeit(nm, rowbytes, [&] { rows_wave_persist_kernel<true><<<2048, 256>>>(out, N, F, ld); });
        snprintf(nm, sizeof nm, "wg/row nt, ld %lld", ld);
        timeit(nm, rowbytes, [&] { rows_wg_kernel<true, 1><<<N, 256>>>(out, N, F, ld); });
        snprintf(nm, sizeof nm, "wg/4 rows nt, ld %lld", ld);
        timeit(nm, rowbytes, [&] { rows_wg_kernel<true, 4><<<(N + 3) / 4, 256>>>(out, N, F, ld); });
        snprintf(nm, sizeof nm, "1-KB pieces nt (x4 launches of 256 cols), ld %lld", ld);
        timeit(nm, (double)N * 1024 * 4, [&] { for (int c = 0; c < 4; ++c) pieces_kernel<true><<<(N + 63) / 64, 256>>>(out, N, 256 * c > F - 256 ? F - 256 - (F % 4) : 256 * c, ld); });
    }
    CK(hipFree(out));
    // loads: 1024 frames of 1 Mpix u16 = 2 GiB
    const size_t fbytes = 2u << 20; const int NF = 1024;
    unsigned char *frames; CK(hipMalloc(&frames, fbytes * NF));
    CK(hipMemset(frames, 1, fbytes * NF));
    const double rb = (double)fbytes * NF;
    printf("== loads: %d frames x 2 MiB (%.2f GB)\n", NF, rb / 1e9);
    timeit("linear read 16 B/lane plain, grid 4096", rb, [&] { linear_read_kernel<false><<<4096, 256>>>((const v4u *)frames, fbytes * NF / 16, sink); });
    timeit("linear read 16 B/lane nt, grid 4096", rb, [&] { linear_read_kernel<true><<<4096, 256>>>((const v4u *)frames, fbytes * NF / 16, sink); });
    timeit("linear read 16 B/lane nt, grid 2048", rb, [&] { linear_read_kernel<true><<<2048, 256>>>((const v4u *)frames, fbytes * NF / 16, sink); });
    timeit("tile read 4 B/lane nt, one launch", rb, [&] { tile_read_kernel<4, true><<<dim3(fbytes / 256, NF / 64), 256>>>(frames, fbytes, 64, sink); });
    timeit("tile read 4 B/lane plain, one launch", rb, [&] { tile_read_kernel<4, false><<<dim3(fbytes / 256, NF / 64), 256>>>(frames, fbytes, 64, sink); });
    timeit("tile read 4 B/lane nt, 16 launches", rb, [&] { for (int g = 0; g < NF / 64; ++g) tile_read_kernel<4, true><<<dim3(fbytes / 256, 1), 256>>>(frames + (size_t)g * 64 * fbytes, fbytes, 64, sink); });
    timeit("tile read 8 B/lane nt, one launch", rb, [&] { tile_read_kernel<8, true><<<dim3(fbytes / 512, NF / 64), 256>>>(frames, fbytes, 64, sink); });
    timeit("tile read 16 B/lane nt, one launch", rb, [&] { tile_read_kernel<16, true><<<dim3(fbytes / 1024, NF / 64), 256>>>(frames, fbytes, 64, sink); });
    timeit("tile read 16 B/lane plain, one launch", rb, [&] { tile_read_kernel<16, false><<<dim3(fbytes / 1024, NF / 64), 256>>>(frames, fbytes, 64, sink); });
    return 0;
}
