// What bounds ecc_sums_kernel?  The IDENT form (5 neighbour loads + template) with parts switched off:
//   ACC   0: 6 masked moments only, 1: all 45 double sums
//   COORD 0: no fixed-point source coordinate / mask (f64 mul, rndne, cvt), 1: as in the kernel
//   F32   1: accumulate in float instead of double (NOT the reference arithmetic: cost probe only)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int ACC, int COORD, int F32>
__global__ void __launch_bounds__(256)
    probe(const float *__restrict__ img, const float *__restrict__ tmpl, int rows, int cols, const float *Mf, double *partial)
{
    typedef typename std::conditional<F32 == 1, float, double>::type T;
    const int f = blockIdx.y;
    const size_t npix = (size_t)rows * cols;
    const float *I = img + (size_t)f * npix;
    double M[6];
    for (int i = 0; i < 6; ++i) M[i] = Mf[i];
    T acc[45];
#pragma unroll
    for (int k = 0; k < 45; ++k) acc[k] = 0;
    const unsigned npx = (unsigned)npix;
    const unsigned per_block = (npx + gridDim.x - 1) / gridDim.x;
    const unsigned lo = blockIdx.x * per_block, hi = min(npx, lo + per_block);
    unsigned i = lo + threadIdx.x;
    int y = (int)(i / (unsigned)cols), x = (int)(i % (unsigned)cols);
    for (; i < hi; i += 256u) {
        bool m = true;
        if (COORD) {
            const int Xr = __double2int_rn((M[1] * y + M[2]) * 1024) + __double2int_rn(M[0] * x * 1024);
            const int Yr = __double2int_rn((M[4] * y + M[5]) * 1024) + __double2int_rn(M[3] * x * 1024);
            const int nx = max(-32768, min(32767, (Xr + 512) >> 10)), ny = max(-32768, min(32767, (Yr + 512) >> 10));
            m = (unsigned)nx < (unsigned)cols && (unsigned)ny < (unsigned)rows;
        }
        float w = 0, gx = 0, gy = 0;
        if (x >= 1 && x + 2 < cols && y >= 1 && y + 2 < rows) {
            const float *r1 = I + i;
            w = r1[0];
            gx = -0.5f * r1[-1] + 0.5f * r1[1];
            gy = -0.5f * r1[-cols] + 0.5f * r1[cols];
        }
        const float X = (float)x, Y = (float)y;
        const float J[6] = {gx * X, gy * X, gx * Y, gy * Y, gx, gy};
        const float t = tmpl[i];
        const T mm = m ? 1 : 0, wd = w, td = t, tm = m ? td : (T)0, wm = m ? wd : (T)0;
        acc[0] += mm; acc[1] += wm; acc[2] += wm * wd; acc[3] += tm; acc[4] += tm * td; acc[5] += tm * wd;
        if (ACC) {
            T Jd[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) Jd[a] = (T)J[a];
            int h = 24;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                acc[6 + a] += Jd[a] * wd; acc[12 + a] += Jd[a] * mm; acc[18 + a] += Jd[a] * tm;
#pragma unroll
                for (int b = a; b < 6; ++b) { acc[h] += Jd[a] * Jd[b]; ++h; }
            }
        } else {
            acc[6] += (T)(J[0] + J[1] + J[2] + J[3] + J[4] + J[5]);
        }
        x += 256;
        while (x >= cols) { x -= cols; ++y; }
    }
    T tot = 0;
#pragma unroll
    for (int k = 0; k < 45; ++k) tot += acc[k];
    if (tot == (T)123456789) partial[0] = (double)tot;
}

struct Var { std::string name; std::function<void()> fn; std::vector<float> ms; };
int main()
{
    const int rows = 1024, cols = 1024, NF = 64;
    float *img, *tmpl, *M; double *partial;
    CK(hipMalloc(&img, sizeof(float) * rows * cols * NF)); CK(hipMemset(img, 0, sizeof(float) * rows * cols * NF));
    CK(hipMalloc(&tmpl, sizeof(float) * rows * cols)); CK(hipMemset(tmpl, 0, sizeof(float) * rows * cols));
    CK(hipMalloc(&M, 24)); CK(hipMalloc(&partial, 64));
    const float hM[6] = {1, 0, 0, 0, 1, 0};
    CK(hipMemcpy(M, hM, 24, hipMemcpyHostToDevice));
    std::vector<Var> vars;
    const dim3 grid(64, NF);
#define ADD(A, C, F) vars.push_back({"ACC " #A " COORD " #C " F32 " #F, [=] { probe<A, C, F><<<grid, 256>>>(img, tmpl, rows, cols, M, partial); }, {}})
    ADD(1, 1, 0); ADD(1, 0, 0); ADD(0, 1, 0); ADD(0, 0, 0); ADD(1, 1, 1); ADD(1, 0, 1); ADD(0, 0, 1);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto &v : vars) v.fn();
    CK(hipDeviceSynchronize());
    for (int round = 0; round < 3; ++round)
        for (auto &v : vars) {
            v.fn();
            for (int r = 0; r < 3; ++r) {
                CK(hipEventRecord(e0, 0)); v.fn(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); v.ms.push_back(ms);
            }
        }
    CK(hipGetLastError());
    for (auto &v : vars) {
        std::sort(v.ms.begin(), v.ms.end());
        printf("%-28s min %7.3f ms = %6.2f us per frame-iteration (median %7.3f)\n", v.name.c_str(), v.ms.front(), v.ms.front() * 1e3 / NF, v.ms[v.ms.size() / 2]);
    }
    return 0;
}
