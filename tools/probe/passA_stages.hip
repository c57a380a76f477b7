// Where does pass A (scan_compact_kernel) lose against the bare tile read?  Stages of the kernel on
// bench-like data: 1024 frames of 1 Mpix, random 12-bit pixels, 10 % of the tiles active (77 of 128 px).
//   STAGE 0: loads only   1: + hot compare   2: + LDS tile writes (active tiles)   3: + compact stores (full kernel)
//   VAR: 0 = 4 B/lane, 64 frames x 128 px per workgroup;  1 = 8 B/lane, 32 frames x 256 px;  2 = 16 B/lane, 16 frames x 512 px
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int kPix = 128, kPitch = 65, kHotCap = 64;

template <int STAGE, bool NT, int RUN = 0>
__global__ void __launch_bounds__(256)
    passA(const uint16_t *__restrict__ frames, size_t npix, int nframes_call, const uint8_t *__restrict__ flag,
          const unsigned *__restrict__ tile_off, uint16_t *__restrict__ compact, size_t cpitch, size_t gstride, unsigned thresh,
          unsigned max_hot, unsigned *__restrict__ count, unsigned *__restrict__ pos, unsigned *sink)
{
    __shared__ unsigned tile[64][kPitch];
    __shared__ int act_k[kPix];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = blockIdx.y;
    // RUN > 0: runs of RUN consecutive tiles stay together, the runs are visited in a strided order
    unsigned bx = blockIdx.x;
    if (RUN > 0) {
        const unsigned nruns = gridDim.x / RUN;          // (grid is a multiple of RUN here)
        const unsigned run = bx / RUN, in_run = bx % RUN;
        bx = (unsigned)(((unsigned long long)run * 2654435761ull) % nruns) * RUN + in_run;   // odd multiplier, nruns power of 2
    }
#define blockIdx_x bx
    const int nframes = min(64, nframes_call - 64 * g);
    frames += (size_t)g * 64 * npix;
    compact += gstride * g;
    count += 64 * g;
    pos += (size_t)64 * g * kHotCap;
    const size_t p0 = (size_t)blockIdx_x * kPix + 2u * (unsigned)lane;
    const bool in = p0 + 1 < npix;
    const unsigned k0 = tile_off[blockIdx_x];
    const bool any_active = STAGE >= 2 && tile_off[blockIdx_x + 1] != k0;
    unsigned v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int f = wave + 4 * i;
        const unsigned *p = reinterpret_cast<const unsigned *>(frames + (size_t)f * npix + p0);
        v[i] = (in && f < nframes) ? (NT ? __builtin_nontemporal_load(p) : *p) : 0u;
    }
    if (any_active && threadIdx.x < kPix) {
        const size_t p = (size_t)blockIdx_x * kPix + threadIdx.x;
        const unsigned fl = p < npix ? flag[p] : 0u;
        act_k[threadIdx.x] = fl ? (int)(k0 + (fl & 0x7Fu)) : -1;
    }
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int f = wave + 4 * i;
        if (any_active) tile[f][lane] = v[i];
        if (STAGE == 0) acc |= v[i];
        if (STAGE >= 1 && (((v[i] & 0xFFFFu) >= thresh) | ((v[i] >> 16) >= thresh))) {
            if (__hip_atomic_load(&count[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= max_hot) {
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (((v[i] >> (16 * k)) & 0xFFFFu) >= thresh) {
                        const unsigned slot = atomicAdd(&count[f], 1u);
                        if (slot < (unsigned)kHotCap) pos[(size_t)f * kHotCap + slot] = (unsigned)(p0 + k);
                    }
            }
        }
    }
    if (STAGE == 0 && acc == 0xFFFFFFFFu) sink[0] = acc;
    if (!any_active) return;
    __syncthreads();
    if (STAGE < 3) return;
    const int grp = threadIdx.x >> 3, j8 = threadIdx.x & 7;
#pragma unroll
    for (int r = 0; r < kPix / 32; ++r) {
        const int o = grp + 32 * r;
        const int k = act_k[o];
        if (k < 0) continue;
        const unsigned col = (unsigned)o >> 1, sh = 16u * ((unsigned)o & 1u);
        unsigned w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned a = (tile[8 * j8 + 2 * q][col] >> sh) & 0xFFFFu;
            const unsigned b = (tile[8 * j8 + 2 * q + 1][col] >> sh) & 0xFFFFu;
            w[q] = a | (b << 16);
        }
        if (STAGE == 4)        // same stores into a small cache-resident region (no HBM write traffic)
            *reinterpret_cast<uint4 *>(compact + (size_t)(k & 127) * cpitch + 8 * j8) = make_uint4(w[0], w[1], w[2], w[3]);
        else if (STAGE == 5) {
            typedef unsigned v4u __attribute__((ext_vector_type(4)));
            const v4u t = {w[0], w[1], w[2], w[3]};
            __builtin_nontemporal_store(t, reinterpret_cast<v4u *>(compact + (size_t)k * cpitch + 8 * j8));
        } else if (STAGE == 6) {   // LDS reads + packing but no store at all unless impossible value
            if ((w[0] & w[1] & w[2] & w[3]) == 0xFFFFFFFFu) *reinterpret_cast<uint4 *>(compact + (size_t)k * cpitch + 8 * j8) = make_uint4(w[0], w[1], w[2], w[3]);
        } else
            *reinterpret_cast<uint4 *>(compact + (size_t)k * cpitch + 8 * j8) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

#undef blockIdx_x
// persistent variant of the full kernel shape without LDS: grid-stride over (tile, group) items, loads only + hot compare
template <bool NT>
__global__ void __launch_bounds__(256)
    passA_persist(const uint16_t *__restrict__ frames, size_t npix, int ngroups, unsigned ntiles, unsigned thresh, unsigned *sink)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned hot = 0;
    for (unsigned item = blockIdx.x; item < ntiles * (unsigned)ngroups; item += gridDim.x) {
        const unsigned tile = item % ntiles, g = item / ntiles;
        const uint16_t *fr = frames + (size_t)g * 64 * npix + (size_t)tile * kPix + 2u * (unsigned)lane;
        unsigned v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const unsigned *p = reinterpret_cast<const unsigned *>(fr + (size_t)(wave + 4 * i) * npix);
            v[i] = NT ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) hot += (((v[i] & 0xFFFFu) >= thresh) | ((v[i] >> 16) >= thresh)) ? 1u : 0u;
    }
    if (hot == 0xFFFFFFFFu) sink[0] = hot;
}

__global__ void fill_random(uint16_t *f, size_t n, unsigned seed)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        f[i] = (uint16_t)(x % 3800u);
    }
}

struct Var { std::string name; double bytes; std::function<void()> fn; std::vector<float> ms; };

int main(int argc, char **argv)
{
    const bool clustered = argc > 1;
    const size_t npix = 1u << 20; const int NF = 1024; const unsigned ntiles = npix / kPix;
    uint16_t *frames; CK(hipMalloc(&frames, npix * 2 * NF));
    fill_random<<<4096, 256>>>(frames, npix * NF, 12345u);
    std::vector<uint8_t> hflag(npix, 0); std::vector<unsigned> hoff(ntiles + 1, 0);
    unsigned k = 0;
    for (unsigned t = 0; t < ntiles; ++t) {
        hoff[t] = k;
        const bool act = clustered ? (t >= 3686 && t < 3686 + 820) : (t % 10 == 3);
        if (act) for (int i = 0; i < 77; ++i) { hflag[(size_t)t * kPix + i] = 0x80 | i; ++k; }
    }
    hoff[ntiles] = k;
    uint8_t *flag; unsigned *off; uint16_t *compact; unsigned *count, *pos, *sink;
    CK(hipMalloc(&flag, npix)); CK(hipMemcpy(flag, hflag.data(), npix, hipMemcpyHostToDevice));
    CK(hipMalloc(&off, 4 * (ntiles + 1))); CK(hipMemcpy(off, hoff.data(), 4 * (ntiles + 1), hipMemcpyHostToDevice));
    CK(hipMalloc(&compact, (size_t)k * 1024 * 2));
    CK(hipMalloc(&count, 4 * NF)); CK(hipMemset(count, 0, 4 * NF));
    CK(hipMalloc(&pos, 4 * NF * kHotCap)); CK(hipMalloc(&sink, 64));
    printf("active pixels %u (%.1f MB compact)\n", k, k * 2048.0 / 1e6);
    const double rb = (double)npix * 2 * NF;
    std::vector<Var> vars;
    auto add = [&](std::string n, std::function<void()> f) { vars.push_back({n, rb, f, {}}); };
    const dim3 grid(ntiles, NF / 64);
#define ADD(S, NT) add(std::string("stage " #S) + (NT ? " nt" : " plain"), [=] { passA<S, NT><<<grid, 256>>>(frames, npix, NF, flag, off, compact, (size_t)1024, (size_t)64, 4064u, 5u, count, pos, sink); })
    ADD(0, true); ADD(2, true); ADD(3, true); ADD(3, false);
#define ADDR(R) add("stage 3 nt, runs of " #R " tiles permuted", [=] { passA<3, true, R><<<grid, 256>>>(frames, npix, NF, flag, off, compact, (size_t)1024, (size_t)64, 4064u, 5u, count, pos, sink); })
    ADD(4, true); ADD(5, true); ADD(6, true);
    add("stage 3 nt, group-major compact", [=] { passA<3, true><<<grid, 256>>>(frames, npix, NF, flag, off, compact, (size_t)64, (size_t)k * 64, 4064u, 5u, count, pos, sink); });
    add("stage 3 plain loads, group-major compact", [=] { passA<3, false><<<grid, 256>>>(frames, npix, NF, flag, off, compact, (size_t)64, (size_t)k * 64, 4064u, 5u, count, pos, sink); });
    add("persistent 2048 wg, loads + hot compare, nt", [=] { passA_persist<true><<<2048, 256>>>(frames, npix, NF / 64, ntiles, 4064u, sink); });
    add("persistent 1024 wg, loads + hot compare, nt", [=] { passA_persist<true><<<1024, 256>>>(frames, npix, NF / 64, ntiles, 4064u, sink); });
    add("persistent 4096 wg, loads + hot compare, nt", [=] { passA_persist<true><<<4096, 256>>>(frames, npix, NF / 64, ntiles, 4064u, sink); });
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto &v : vars) v.fn();
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
        printf("%-48s min %7.3f ms (%5.2f TB/s)  median %7.3f ms (%5.2f TB/s)\n", v.name.c_str(), mn, v.bytes / mn / 1e9, med, v.bytes / med / 1e9);
    }
    return 0;
}
