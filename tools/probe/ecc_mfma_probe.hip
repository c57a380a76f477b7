// Round 6, review item 1(c): the 45 sums of an ECC iteration as the upper triangle of A^T A, A = [J1..J6, w, t, 1] (9 columns) -- on MFMA?
//   hipcc -O3 --offload-arch=gfx950 tools/probe/ecc_mfma_probe.hip -o gpurun_out/ecc_mfma_probe && gpurun_out/ecc_mfma_probe
// Two kernels do the ACCUMULATION ONLY for the same number of pixels per wave, operands already in registers in the layout each
// form needs (i.e. the MFMA form is given its transposition for free):
//   packed : the product kernel's factored form -- a lane owns a column, 13 packed-f32 + 5 scalar f32 + 5 f64 operations per pixel
//            (ecc_part_add), 64 pixels per wave and step
//   mfma   : v_mfma_f32_16x16x4_f32, C[16x16] += A[16x4] B[4x16] with A = B^T = the 9 (of 16) per-pixel factors of FOUR pixels:
//            one instruction per 4 pixels, 16 instructions per 64 pixels, 256 products per pixel for the 45 that are wanted
// Prints cycles per 64 pixels and wave for both.  (f64 MFMA: v_mfma_f64_16x16x4_f64 has the same shape at half the rate.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) packed_kernel(const float *in, float *out, int steps)
{
    float gx = in[threadIdx.x], gy = in[threadIdx.x + 256], w = in[threadIdx.x + 512], t = in[threadIdx.x + 768];
    v2f G0 = {0, 0}, G1 = G0, Gw0 = G0, Gw1 = G0, Gt0 = G0, Gt1 = G0, Q0 = G0, Q1 = G0, Q2 = G0, C01 = G0;
    float C2 = 0.f;
    double Sw = 0, Sww = 0, St = 0, Stt = 0, Stw = 0;
    for (int s = 0; s < steps; ++s) {
        const float rf = (float)(s & 31), rf2 = rf * rf;
        const v2f G = {gx, gy}, R = {rf, rf}, W = {w, w}, T = {t, t}, R2 = {rf2, rf2};
        G0 += G;
        G1 = __builtin_elementwise_fma(G, R, G1);
        const v2f Gw = G * W, Gt = G * T, Q = G * G;
        Gw0 += Gw; Gw1 = __builtin_elementwise_fma(Gw, R, Gw1);
        Gt0 += Gt; Gt1 = __builtin_elementwise_fma(Gt, R, Gt1);
        Q0 += Q; Q1 = __builtin_elementwise_fma(Q, R, Q1); Q2 = __builtin_elementwise_fma(Q, R2, Q2);
        const float c = gx * gy;
        const v2f Cc = {c, c}, R01 = {1.f, rf};
        C01 = __builtin_elementwise_fma(Cc, R01, C01);
        C2 = __builtin_fmaf(c, rf2, C2);
        const double wd = w, td = t;
        Sw += wd; Sww = fma(wd, wd, Sww); St += td; Stt = fma(td, td, Stt); Stw = fma(td, wd, Stw);
        gx += 1e-3f; gy -= 1e-3f; w += 0.5f; t += 0.25f;      // (keeps the loop from being folded)
    }
    const v2f S = G0 + G1 + Gw0 + Gw1 + Gt0 + Gt1 + Q0 + Q1 + Q2 + C01;
    out[blockIdx.x * 256 + threadIdx.x] = S.x + S.y + C2 + (float)(Sw + Sww + St + Stt + Stw);
}

__global__ void __launch_bounds__(256) mfma_kernel(const float *in, float *out, int steps)
{
    // lane (i, k) = (lane % 16, lane / 16) holds factor i of pixel k of the current group of four
    float a = in[threadIdx.x];
    v4f acc = {0, 0, 0, 0};
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {        // 16 groups of 4 pixels = the 64 pixels a wave of the packed form takes per step
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a, acc, 0, 0, 0);
            a += 1e-3f;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

int main()
{
    float *in, *out;
    CK(hipMalloc(&in, 4096 * 4));
    CK(hipMemset(in, 0, 4096 * 4));
    CK(hipMalloc(&out, 4 * 256 * 4096));
    const int steps = 4096, blocks = 2048;      // 8 workgroups of 4 waves per CU: every SIMD holds 8 waves
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int which = 0; which < 2; ++which) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(a));
            if (which == 0) hipLaunchKernelGGL(packed_kernel, dim3(blocks), dim3(256), 0, 0, in, out, steps);
            else hipLaunchKernelGGL(mfma_kernel, dim3(blocks), dim3(256), 0, 0, in, out, steps);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, a, b));
            if (rep == 1) {
                // wave-steps per SIMD: blocks * 4 waves / (256 CUs * 4 SIMDs) waves per SIMD, each `steps` steps of 64 pixels
                const double per_simd = (double)blocks * 4 / 1024 * steps;
                printf("%-7s %8.3f ms  = %6.1f SIMD cycles per 64 pixels (2.4 GHz)  -> a 1 Mpx frame costs the chip %.2f us\n",
                       which == 0 ? "packed" : "mfma", ms, ms * 1e-3 * 2.4e9 / per_simd, ms * 1e3 / ((double)blocks * 4 * steps) * 16384);
            }
        }
    }
    return 0;
}
