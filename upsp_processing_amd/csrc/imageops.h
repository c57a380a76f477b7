// Declarations shared by the image-stage translation units (imageops.hip: filters, warp, patch, stage driver;
// ecc.hip: the ECC registration, cpp/lib/registration.cpp:32-81).
#ifndef UPSP_IMAGEOPS_H
#define UPSP_IMAGEOPS_H

#include <hip/hip_runtime.h>

#include <cstdint>
#include <functional>

#include "pipeline.h"

namespace upsp {

// partial sums of one ECC iteration: per frame and sum one slot per workgroup -- band blocks first, then interior blocks
constexpr int kEccSums = 45;
constexpr int kEccInteriorBlocks = 32;    // interior workgroups per frame (one count per image geometry: see run_ecc)
constexpr int kEccInteriorMax = 512;      // ... at most (images wider than 64 column tiles of 256)
constexpr int kEccBandBlocks = 16;        // band workgroups per frame, 3 x column tiles of 256 at least
constexpr int kEccBandMax = 3 * 128;      // ... at most (columns < 32768)
constexpr int kEccStride = kEccInteriorMax + kEccBandMax;   // slots per (frame, sum)

__host__ __device__ inline int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        if (i >= n) i = 2 * n - 2 - i;
    }
    return i;
}

// -------------------------------------------------------------- warpAffine --
struct WarpCoord {
    int sx, sy, ax, ay;
};

// WarpAffineInvoker (OpenCV imgwarp.cpp): AB_BITS 10, INTER_BITS 5, cvRound of doubles
__device__ __forceinline__ WarpCoord warp_coord(const double *M, int x, int y, int interp)
{
    const int AB_SCALE = 1024;
    const int round_delta = interp ? 16 : 512;
    const int adelta = __double2int_rn(M[0] * x * AB_SCALE);
    const int bdelta = __double2int_rn(M[3] * x * AB_SCALE);
    const int X0 = __double2int_rn((M[1] * y + M[2]) * AB_SCALE) + round_delta;
    const int Y0 = __double2int_rn((M[4] * y + M[5]) * AB_SCALE) + round_delta;
    WarpCoord c;
    if (interp) {
        const int X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
        c.sx = X >> 5; c.sy = Y >> 5; c.ax = X & 31; c.ay = Y & 31;
    } else {
        c.sx = (X0 + adelta) >> 10; c.sy = (Y0 + bdelta) >> 10; c.ax = c.ay = 0;
    }
    c.sx = max(-32768, min(32767, c.sx));  // saturate_cast<short>
    c.sy = max(-32768, min(32767, c.sy));
    return c;
}

// remapBilinear<Cast<float,T>,...>, BORDER_CONSTANT 0.  F(y,x) fetches a source pixel.
template <typename F>
__device__ __forceinline__ float bilinear(F fetch, int rows, int cols, WarpCoord c)
{
    const float fx = c.ax * (1.f / 32), fy = c.ay * (1.f / 32);
    const float w0 = (1.f - fy) * (1.f - fx), w1 = (1.f - fy) * fx, w2 = fy * (1.f - fx), w3 = fy * fx;
    const int sx = c.sx, sy = c.sy;
    if ((unsigned)sx < (unsigned)(cols - 1) && (unsigned)sy < (unsigned)(rows - 1))
        return fetch(sy, sx) * w0 + fetch(sy, sx + 1) * w1 + fetch(sy + 1, sx) * w2 + fetch(sy + 1, sx + 1) * w3;
    if (sx >= cols || sx + 1 < 0 || sy >= rows || sy + 1 < 0) return 0.f;
    const bool x0 = sx >= 0 && sx < cols, x1 = sx + 1 >= 0 && sx + 1 < cols;
    const bool y0 = sy >= 0 && sy < rows, y1 = sy + 1 >= 0 && sy + 1 < rows;
    const float v0 = (x0 && y0) ? fetch(sy, sx) : 0.f;
    const float v1 = (x1 && y0) ? fetch(sy, sx + 1) : 0.f;
    const float v2 = (x0 && y1) ? fetch(sy + 1, sx) : 0.f;
    const float v3 = (x1 && y1) ? fetch(sy + 1, sx + 1) : 0.f;
    return v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3;
}

// per-frame ECC state
struct EccState {
    float M[6];
    double rho, last_rho;
    int iters;
    int done;     // 1 converged / iteration cap, 2 identity (frame 0), <0 error
    int band;     // pixels farther than this from every image edge have their whole bilinear footprint (and its
                  // gradient taps) inside the image under M (ecc_band): interior blocks take them, band blocks the rest
};

// lane i <- lane i - 1 (lane 0 <- old) / lane i <- lane i + 1 (lane 63 <- old): DPP wave shifts
__device__ __forceinline__ float dpp_shr1(float old, float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x138, 0xF, 0xF, false));
}
__device__ __forceinline__ float dpp_shl1(float old, float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x130, 0xF, 0xF, false));
}

// ----------------------------------------------------------- FrameScratch --
struct FrameScratch {
    int ncams = 0, batch = 0, rows = 0, cols = 0;
    uint16_t *warp[kMaxCams] = {nullptr};   // registered u16 frames
    float *f32[kMaxCams] = {nullptr};       // patched / filtered frames
    float *f32b[kMaxCams] = {nullptr};      // Gaussian-filtered frames when the filter input is f32[] itself
    float *ecc_img = nullptr;               // blurred input frames
    float *ecc_img2 = nullptr;              // second buffer: the pre-blur of the NEXT sub-batch is enqueued while this one's
                                            // "frames still iterating" is read back (frame_scratch_preblur)
    float *tmp = nullptr;                   // filter intermediate (float), also box sums (double)
    float *tmpl[kMaxCams] = {nullptr};      // blurred ECC templates
    const float *tmpl_src[kMaxCams] = {nullptr};
    float *center = nullptr;                // [kMaxCams] centre of the float products of the ECC sums (ecc_center_kernel)
    double *tsum = nullptr;                 // [kMaxCams][2] sum t, sum t^2 of the blurred templates (ecc_tmpl_sums_kernel)
    int *h_counter = nullptr;               // pinned: where "frames still iterating" is read back to
    hipEvent_t ev_counter = nullptr;        // ... and the event behind that copy
    double *partial = nullptr;              // [batch][kEccSums][kEccStride] block partial sums of one iteration
    double *partial_id[2] = {nullptr, nullptr};   // [batch][kEccSums][ident_blocks]: the identity iteration's sums of the blurred-frame
                                                  // buffer `slot`, written with the pre-blur (ecc_blur_ident_kernel)
    size_t partial_id_words = 0;                  // ... doubles allocated per slot
    const float *ident_for[2] = {nullptr, nullptr};   // the blurred-frame buffer whose identity sums partial_id[slot] holds (one use)
    int ident_blocks = 0;                         // ... workgroups per frame of that launch
    // The second pass of the fused pre-blur (the workgroups a repaired hot pixel reaches: a handful, ~40 us of one workgroup's
    // latency) runs on a stream of its own beside whatever the frame loop's stream does between the pre-blur and the first solve of
    // that sub-batch; run_ecc waits for ev_again[slot] when again_pending[slot].
    hipStream_t again_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_again[2] = {nullptr, nullptr};
    bool again_pending[2] = {false, false};
    int2 *rtab = nullptr;                   // [batch][rows] per-row terms of the fixed-point source coordinate under the frame's M
    EccState *state = nullptr;
    int *counter = nullptr;
    unsigned long long ecc_frame_iters = 0, ecc_frames = 0;   // statistics: ECC iterations summed over frames, frames
    int ecc_first_burst = 3;                                  // iterations issued before the first host check
};

// ecc.hip
// findTransformECC of nb frames (already blurred 5 x 5: `blurred` [nb][rows][cols] f32) against the blurred template;
// leaves the warp matrices in s->state.  while_waiting (may be null): called once while the host waits for the first
// read-back of "frames still iterating" (it may enqueue work on `st`).
int run_ecc(FrameScratch *s, const float *tmpl_blur, const float *d_center, const float *blurred, int nb, int64_t first_frame,
            int rows, int cols, int max_iters, double eps, hipStream_t st, const std::function<int()> *while_waiting = nullptr);
int launch_ecc_center(const float *tmpl_blur, int rows, int cols, float *d_center, hipStream_t st);
// The 5 x 5 pre-blur of nb u16 frames fused with the identity iteration's sums (ecc_blur_ident_kernel): blurred frames -> dst,
// sums -> s->partial_id[slot]; s->ident_for[slot] = dst tells run_ecc that its first iteration is already summed.  hot_count /
// hot_pos (may be null): the scan of fix_hot_pixels on the way; only_changed (may be null): the second pass, over the frames the repair
// changed -- per-frame change counts + their records (uint4 [frame][max_hot], .y = pixel position): only the workgroups a changed
// pixel reaches run again.
bool ecc_fused_blur_eligible(int rows, int cols);
int launch_ecc_blur_ident(FrameScratch *s, int slot, const uint16_t *d_frames, float *dst, const float *tmpl_blur, const float *d_center,
                          const double *d_tsum, int nb, int rows, int cols, float k0, float k1, float k2, unsigned thresh,
                          unsigned *hot_count, unsigned *hot_pos, const unsigned *only_changed, const void *changes, int max_hot,
                          hipStream_t st);
// d_out[0 .. 1] = sum t, sum t^2 of the blurred template (the identity iteration's St, Stt); d_part: 128 doubles of scratch
int launch_ecc_tmpl_sums(const float *tmpl_blur, int rows, int cols, double *d_part, double *d_out, hipStream_t st);
int launch_ecc_export(const EccState *state, int nb, float *d_warps, int wstride, int32_t *d_iters, int istride, hipStream_t st);

}  // namespace upsp
#endif
