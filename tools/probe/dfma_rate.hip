// Issue rate of the double-precision instructions ecc_sums_kernel uses, on gfx950: NACC independent chains per lane,
// 1 .. 8 waves per SIMD.  Prints cycles per wave-instruction (wall clock x clock rate / instructions per SIMD).
//   fma / add / mul: v_fma_f64, v_add_f64, v_mul_f64;  rnd: v_rndne_f64;  ldexp: v_ldexp_f64;
//   cvt_i: v_cvt_i32_f64 + v_cvt_f64_i32 (a pair per step);  cvt_f: v_cvt_f32_f64 + v_cvt_f64_f32 (a pair per step)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NACC, int MODE>
__global__ void __launch_bounds__(256) probe(double *out, int iters, double a0, double b0)
{
    double acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = (double)(threadIdx.x + k) + 0.25;
    const double a = a0 + (double)threadIdx.x * 1e-9, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) {
            if (MODE == 0) acc[k] = __builtin_fma(a, b, acc[k]);
            else if (MODE == 1) acc[k] = acc[k] + a;
            else if (MODE == 2) acc[k] = acc[k] * a;
            else if (MODE == 3) acc[k] = __builtin_rint(acc[k]);
            else if (MODE == 4) acc[k] = __builtin_ldexp(acc[k], 1);
            else if (MODE == 5) acc[k] = (double)__double2int_rn(acc[k]);
            else acc[k] = (double)(float)acc[k];
        }
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < NACC; ++k) s += acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int MODE>
void run(const char *name, int blocks_per_cu, int per_step)
{
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount, blocks = cus * blocks_per_cu, iters = 2048;
    double *out; CK(hipMalloc(&out, sizeof(double) * blocks * 256));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((probe<NACC, MODE>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001, 0.9999999);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<NACC, MODE>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001, 0.9999999);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_simd = (double)blocks_per_cu * iters * NACC * per_step;
    const double clk = p.clockRate * 1e3;
    printf("%-10s %d waves/SIMD: %.2f cycles per wave-instruction\n", name, blocks_per_cu, ms * 1e-3 * clk / instr_per_simd);
    CK(hipFree(out));
}

int main()
{
    for (int w : {4, 8}) {
        run<32, 0>("fma", w, 1);
        run<32, 1>("add", w, 1);
        run<32, 2>("mul", w, 1);
        run<32, 3>("rndne", w, 1);
        run<32, 4>("ldexp", w, 1);
        run<32, 5>("cvt_i32", w, 2);
        run<32, 6>("cvt_f32", w, 2);
    }
    return 0;
}
