// A torch-free C++ caller of the per-frame pipeline: what a C++ psp_process does between createBVH and the reductions
// (cpp/exec/psp_process.cpp:44-53 createBVH, :167-355 create_projection_mat, :1771-1843 the frame loop,
// cpp/include/projection.h:351-353 project_frame) through the C ABI of libupsp_gpu.so and the HIP runtime alone --
// hipMalloc / hipMemcpy, upsp_bvh_create -> upsp_bvh_set_tri_nodes -> upsp_projection_build ->
// upsp_pipeline_create / set_projection / process / accumulators / finalize.
//
//   frame_loop_test tests/golden/frame_loop_sphere.bin
//
// The golden file (tests/golden/make_golden_frame_loop.py) holds the mesh, the camera and what the ORACLE computed for
// them: the projection's pixel per node, the 8 series rows (NaN rows for the nodes no camera sees), the double
// accumulators.  Everything is compared bit for bit; prints "frame loop ok ..." and exits 0.
// Then the RE-RAYCAST loop (model motion: the projection rebuilt for every batch of frames) through upsp_pipeline_step -- one call
// per step, the frames re-uploaded by the frames hook -- must give the same series, accumulators and finals (three steps).
//
//   frame_loop_test --perf model.bin nframes steps
//
// The same one-call step on a model file written by tests/test_frame_loop_cpp_gpu.py (the bench's 1 M-triangle tunnel model, a
// 1024 x 1024 camera), `nframes` resident frames, `steps` timed steps behind 5 warm-up steps: prints "step perf: <ms> ms per step".
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

template <typename T>
static std::vector<T> read_vec(std::FILE *f, size_t n)
{
    std::vector<T> v(n);
    if (std::fread(v.data(), sizeof(T), n, f) != n) {
        std::fprintf(stderr, "golden file too short\n");
        std::exit(2);
    }
    return v;
}

template <typename T>
static T *to_device(const std::vector<T> &h)
{
    T *d = nullptr;
    HIPCHECK(hipMalloc(reinterpret_cast<void **>(&d), sizeof(T) * h.size()));
    HIPCHECK(hipMemcpy(d, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
    return d;
}

// make_frames() of tests/golden/make_golden_frame_loop.py
static uint16_t frame_value(int f, int y, int x, int W)
{
    const uint32_t i = (uint32_t)(y * W + x);
    uint32_t h = i * 2654435761u + (uint32_t)(f * 40503 + 12345);
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    return (uint16_t)(((300u + 5u * (uint32_t)x + 3u * (uint32_t)y + 11u * (uint32_t)f + (h >> 21)) & 0xFFFu) % 3000u);
}

// --perf: the one-call re-raycast step at the bench's size, timed from C++
static int perf_main(const char *path, int nframes, int steps)
{
    std::FILE *fh = std::fopen(path, "rb");
    if (!fh) {
        std::perror(path);
        return 2;
    }
    const std::vector<uint32_t> hdr = read_vec<uint32_t>(fh, 8);
    if (hdr[0] != 0x5550534Du) {
        std::fprintf(stderr, "bad magic\n");
        return 2;
    }
    const size_t T = hdr[1], N = hdr[2];
    const int W = (int)hdr[3], H = (int)hdr[4];
    upsp_camera cam;
    std::memset(&cam, 0, sizeof(cam));
    {
        const std::vector<double> K = read_vec<double>(fh, 9), dist = read_vec<double>(fh, 5), R = read_vec<double>(fh, 9), t = read_vec<double>(fh, 3);
        std::memcpy(cam.K, K.data(), sizeof(cam.K));
        std::memcpy(cam.dist, dist.data(), sizeof(cam.dist));
        std::memcpy(cam.R, R.data(), sizeof(cam.R));
        std::memcpy(cam.t, t.data(), sizeof(cam.t));
        cam.width = W;
        cam.height = H;
    }
    const float thresh = read_vec<float>(fh, 1)[0];
    const std::vector<float> verts = read_vec<float>(fh, 3 * N), normals = read_vec<float>(fh, 3 * N);
    const std::vector<int32_t> tris = read_vec<int32_t>(fh, 3 * T);
    std::fclose(fh);
    std::vector<float> soup(9 * T);
    for (size_t k = 0; k < 3 * T; ++k) std::memcpy(&soup[3 * k], &verts[3 * (size_t)tris[k]], 3 * sizeof(float));
    upsp_bvh *bvh = nullptr;
    CHECK(upsp_bvh_create(soup.data(), T, &bvh));
    float *d_nodes = to_device(verts), *d_normals = to_device(normals);
    int32_t *d_tri_nodes = to_device(tris);
    CHECK(upsp_bvh_set_tri_nodes(bvh, d_tri_nodes, N, nullptr));
    // frames: one generated frame (values below the hot-pixel threshold), copied nframes times on the device
    std::vector<uint16_t> one((size_t)H * W);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) one[(size_t)y * W + x] = (uint16_t)(600 + ((x * 7 + y * 13) & 1023));
    uint16_t *d_frames = nullptr;
    HIPCHECK(hipMalloc(reinterpret_cast<void **>(&d_frames), sizeof(uint16_t) * one.size() * (size_t)nframes));
    HIPCHECK(hipMemcpy(d_frames, one.data(), sizeof(uint16_t) * one.size(), hipMemcpyHostToDevice));
    for (int f = 1; f < nframes; ++f)
        HIPCHECK(hipMemcpyAsync(d_frames + (size_t)f * one.size(), d_frames, sizeof(uint16_t) * one.size(), hipMemcpyDeviceToDevice, nullptr));
    HIPCHECK(hipDeviceSynchronize());
    upsp_pipeline_opts opts;
    upsp_pipeline_default_opts(&opts);
    upsp_pipeline *pipe = nullptr;
    CHECK(upsp_pipeline_create(1, W, H, N, &opts, &pipe));
    CHECK(upsp_pipeline_set_row_padding(pipe, 1));
    CHECK(upsp_pipeline_set_scan_split(pipe, 1));
    const int64_t ld = ((int64_t)nframes + 63) / 64 * 64;
    float *d_rows_t = nullptr, *d_avg = nullptr, *d_rms = nullptr;
    HIPCHECK(hipMalloc(reinterpret_cast<void **>(&d_rows_t), sizeof(float) * (size_t)ld * N));
    HIPCHECK(hipMalloc(reinterpret_cast<void **>(&d_avg), sizeof(float) * N));
    HIPCHECK(hipMalloc(reinterpret_cast<void **>(&d_rms), sizeof(float) * N));
    hipStream_t st = nullptr;
    HIPCHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    upsp_step_args sa;
    std::memset(&sa, 0, sizeof(sa));
    sa.bvh = bvh; sa.cam = &cam; sa.d_nodes = d_nodes; sa.d_normals = d_normals; sa.d_tri_nodes = d_tri_nodes; sa.oblique_thresh = thresh;
    sa.nframes = nframes; sa.d_frames = d_frames; sa.d_rows_t = d_rows_t; sa.ld_t = ld; sa.d_avg = d_avg; sa.d_rms = d_rms;
    sa.nframes_total = (uint64_t)nframes;
    for (int k = 0; k < 5; ++k) CHECK(upsp_pipeline_step(pipe, &sa, st));
    HIPCHECK(hipStreamSynchronize(st));
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < steps; ++k) CHECK(upsp_pipeline_step(pipe, &sa, st));
    CHECK(upsp_pipeline_step_finish(pipe, st));
    HIPCHECK(hipStreamSynchronize(st));
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / steps;
    CHECK(upsp_bvh_check(bvh, st));
    // sanity: some node is seen and its average is a pixel value
    std::vector<float> avg(N);
    HIPCHECK(hipMemcpy(avg.data(), d_avg, sizeof(float) * N, hipMemcpyDeviceToHost));
    size_t seen = 0;
    for (size_t n = 0; n < N; ++n) seen += !std::isnan(avg[n]) && avg[n] >= 600.f && avg[n] < 1624.f;
    std::printf("step perf: %.4f ms per step of %d frames (%zu triangles, %zu nodes, %zu nodes with a series), %.0f frames/s\n", ms, nframes, T, N,
                seen, nframes / (ms * 1e-3));
    upsp_pipeline_destroy(pipe);
    upsp_bvh_destroy(bvh);
    return seen ? 0 : 1;
}

int main(int argc, char **argv)
{
    if (argc >= 5 && std::strcmp(argv[1], "--perf") == 0) return perf_main(argv[2], std::atoi(argv[3]), std::atoi(argv[4]));
    if (argc < 2) {
        std::fprintf(stderr, "usage: frame_loop_test golden.bin\n");
        return 2;
    }
    std::FILE *fh = std::fopen(argv[1], "rb");
    if (!fh) {
        std::perror(argv[1]);
        return 2;
    }
    const std::vector<uint32_t> hdr = read_vec<uint32_t>(fh, 8);
    if (hdr[0] != 0x55505350u) {
        std::fprintf(stderr, "bad magic\n");
        return 2;
    }
    const size_t T = hdr[1], N = hdr[2];
    const int W = (int)hdr[3], H = (int)hdr[4], F = (int)hdr[5];
    upsp_camera cam;
    std::memset(&cam, 0, sizeof(cam));
    {
        const std::vector<double> K = read_vec<double>(fh, 9), dist = read_vec<double>(fh, 5), R = read_vec<double>(fh, 9),
                                  t = read_vec<double>(fh, 3);
        std::memcpy(cam.K, K.data(), sizeof(cam.K));
        std::memcpy(cam.dist, dist.data(), sizeof(cam.dist));
        std::memcpy(cam.R, R.data(), sizeof(cam.R));
        std::memcpy(cam.t, t.data(), sizeof(cam.t));
        cam.width = W;
        cam.height = H;
    }
    const float thresh = read_vec<float>(fh, 1)[0];
    const std::vector<float> verts = read_vec<float>(fh, 3 * N), normals = read_vec<float>(fh, 3 * N);
    const std::vector<int32_t> tris = read_vec<int32_t>(fh, 3 * T);
    const std::vector<int32_t> want_pix = read_vec<int32_t>(fh, N);
    const std::vector<float> want_rows = read_vec<float>(fh, (size_t)F * N);
    const std::vector<double> want_sum = read_vec<double>(fh, N), want_sumsq = read_vec<double>(fh, N);
    std::fclose(fh);

    // model.extract_tris (cpp/lib/TriModel.ipp:261-299): 9 floats per triangle + the triangles' node ids
    std::vector<float> soup(9 * T);
    for (size_t k = 0; k < 3 * T; ++k) std::memcpy(&soup[3 * k], &verts[3 * (size_t)tris[k]], 3 * sizeof(float));

    upsp_bvh *bvh = nullptr;
    CHECK(upsp_bvh_create(soup.data(), T, &bvh));                                  // createBVH, :44-53
    float *d_nodes = to_device(verts), *d_normals = to_device(normals);
    int32_t *d_tri_nodes = to_device(tris);
    CHECK(upsp_bvh_set_tri_nodes(bvh, d_tri_nodes, N, nullptr));
    int32_t *d_pix = nullptr;
    float *d_uv = nullptr;
    HIPCHECK(hipMalloc(reinterpret_cast<void **>(&d_pix), sizeof(int32_t) * N));
    HIPCHECK(hipMalloc(reinterpret_cast<void **>(&d_uv), sizeof(float) * 2 * N));
    uint64_t nrays = 0;
    CHECK(upsp_projection_build(bvh, &cam, d_nodes, d_normals, nullptr, d_tri_nodes, N, thresh, d_pix, d_uv, nullptr, &nrays,
                                nullptr));                                         // create_projection_mat, :167-355
    std::vector<int32_t> pix(N);
    HIPCHECK(hipMemcpy(pix.data(), d_pix, sizeof(int32_t) * N, hipMemcpyDeviceToHost));
    size_t bad = 0, visible = 0;
    for (size_t n = 0; n < N; ++n) {
        bad += pix[n] != want_pix[n];
        visible += pix[n] >= 0;
    }
    if (bad || visible != hdr[6]) {
        std::fprintf(stderr, "projection: %zu of %zu entries differ from the oracle's (%zu visible, expected %u)\n", bad, N, visible, hdr[6]);
        return 1;
    }

    // frames: generated here (integer pattern, hot pixels placed like the generator does)
    std::vector<uint16_t> frames((size_t)F * H * W);
    for (int f = 0; f < F; ++f)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) frames[((size_t)f * H + y) * W + x] = frame_value(f, y, x, W);
    auto hot = [&](int f, int y, int x) { frames[((size_t)f * H + y) * W + x] = (uint16_t)(4095 - (y % 16)); };
    hot(1, 40, 50); hot(1, 41, 50);
    for (int k = 1; k <= 7; ++k) hot(3, 10 * k, 10 * k);
    hot(5, 0, 7); hot(5, 100, 100); hot(5, 100, 101); hot(5, 200, 13); hot(5, 255, 255);
    uint16_t *d_frames = to_device(frames);

    upsp_pipeline_opts opts;
    upsp_pipeline_default_opts(&opts);
    upsp_pipeline *pipe = nullptr;
    CHECK(upsp_pipeline_create(1, W, H, N, &opts, &pipe));
    CHECK(upsp_pipeline_set_projection(pipe, 0, d_pix, nullptr));
    float *d_rows = nullptr, *d_rows_t = nullptr;
    const int64_t ld = 64;                                                         // node-major pitch >= F
    HIPCHECK(hipMalloc(reinterpret_cast<void **>(&d_rows), sizeof(float) * (size_t)F * N));
    HIPCHECK(hipMalloc(reinterpret_cast<void **>(&d_rows_t), sizeof(float) * (size_t)ld * N));
    uint16_t *const frame_ptrs[1] = {d_frames};
    // the frame loop, :1771-1843: two calls (5 + 3 frames) like a caller that reads its video in chunks
    CHECK(upsp_pipeline_process(pipe, frame_ptrs, 5, 0, d_rows, d_rows_t, ld, 0, nullptr, nullptr));
    uint16_t *const frame_ptrs2[1] = {d_frames + (size_t)5 * H * W};
    CHECK(upsp_pipeline_process(pipe, frame_ptrs2, F - 5, 5, d_rows + (size_t)5 * N, d_rows_t, ld, 5, nullptr, nullptr));
    HIPCHECK(hipDeviceSynchronize());

    std::vector<float> rows((size_t)F * N), rows_t((size_t)ld * N);
    HIPCHECK(hipMemcpy(rows.data(), d_rows, sizeof(float) * rows.size(), hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(rows_t.data(), d_rows_t, sizeof(float) * rows_t.size(), hipMemcpyDeviceToHost));
    size_t bad_rows = 0, bad_t = 0, nan_rows = 0;
    for (int f = 0; f < F; ++f)
        for (size_t n = 0; n < N; ++n) {
            uint32_t a, b, c;
            std::memcpy(&a, &rows[(size_t)f * N + n], 4);
            std::memcpy(&b, &want_rows[(size_t)f * N + n], 4);
            std::memcpy(&c, &rows_t[n * (size_t)ld + f], 4);
            const bool both_nan = std::isnan(rows[(size_t)f * N + n]) && std::isnan(want_rows[(size_t)f * N + n]);
            bad_rows += !(a == b || both_nan);
            bad_t += !(c == a || (std::isnan(rows_t[n * (size_t)ld + f]) && both_nan));
            nan_rows += f == 0 && both_nan;
        }
    // the frames were repaired in place like fix_hot_pixels does (cv_extras.cpp:230-275): checksum of all pixels
    HIPCHECK(hipMemcpy(frames.data(), d_frames, sizeof(uint16_t) * frames.size(), hipMemcpyDeviceToHost));
    uint64_t sum_px = 0;
    for (uint16_t v : frames) sum_px += v;
    double *d_sum = nullptr, *d_sumsq = nullptr;
    CHECK(upsp_pipeline_accumulators(pipe, &d_sum, &d_sumsq));
    std::vector<double> sum(N), sumsq(N);
    HIPCHECK(hipMemcpy(sum.data(), d_sum, sizeof(double) * N, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(sumsq.data(), d_sumsq, sizeof(double) * N, hipMemcpyDeviceToHost));
    size_t bad_acc = 0;
    for (size_t n = 0; n < N; ++n) {
        const bool nan = std::isnan(want_sum[n]);
        // every term is an integer-valued float (u16 pixels, weight 1): the double sums are exact in any order
        bad_acc += nan ? !(std::isnan(sum[n]) && std::isnan(sumsq[n])) : !(sum[n] == want_sum[n] && sumsq[n] == want_sumsq[n]);
    }
    // finals, :1930-1936
    float *d_avg = nullptr, *d_rms = nullptr;
    HIPCHECK(hipMalloc(reinterpret_cast<void **>(&d_avg), sizeof(float) * N));
    HIPCHECK(hipMalloc(reinterpret_cast<void **>(&d_rms), sizeof(float) * N));
    CHECK(upsp_pipeline_finalize(pipe, (uint64_t)F, d_avg, d_rms, nullptr));
    HIPCHECK(hipDeviceSynchronize());
    std::vector<float> avg(N), rms(N);
    HIPCHECK(hipMemcpy(avg.data(), d_avg, sizeof(float) * N, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(rms.data(), d_rms, sizeof(float) * N, hipMemcpyDeviceToHost));
    size_t bad_fin = 0;
    for (size_t n = 0; n < N; ++n) {
        if (std::isnan(want_sum[n])) {
            bad_fin += !(std::isnan(avg[n]) && std::isnan(rms[n]));
        } else {
            const float a = (float)(want_sum[n] / (double)F), r = (float)std::sqrt(want_sumsq[n] / (double)F);
            bad_fin += !(avg[n] == a && rms[n] == r);
        }
    }
    // the same frames once more, node-major series only: the streamed two-pass schedule (pass A + pass B) instead of scan + gather
    CHECK(upsp_pipeline_reset(pipe));
    float *d_rows_t2 = nullptr;
    HIPCHECK(hipMalloc(reinterpret_cast<void **>(&d_rows_t2), sizeof(float) * (size_t)ld * N));
    CHECK(upsp_pipeline_process(pipe, frame_ptrs, F, 0, nullptr, d_rows_t2, ld, 0, nullptr, nullptr));
    HIPCHECK(hipDeviceSynchronize());
    std::vector<float> rows_t2((size_t)ld * N);
    HIPCHECK(hipMemcpy(rows_t2.data(), d_rows_t2, sizeof(float) * rows_t2.size(), hipMemcpyDeviceToHost));
    for (size_t n = 0; n < N; ++n)
        for (int f = 0; f < F; ++f) bad_t += std::memcmp(&rows_t2[n * (size_t)ld + f], &rows_t[n * (size_t)ld + f], 4) != 0;
    // ---- the re-raycast loop: upsp_pipeline_step, three steps on the same frames (uploaded again by the hook every step) ----
    size_t bad_step = 0;
    {
        struct Upload { const std::vector<uint16_t> *h; uint16_t *d; } up = {&frames, d_frames};
        // (the host copy of the frames as they were generated: `frames` holds the repaired ones by now)
        std::vector<uint16_t> fresh((size_t)F * H * W);
        for (int f = 0; f < F; ++f)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) fresh[((size_t)f * H + y) * W + x] = frame_value(f, y, x, W);
        auto hot2 = [&](int f, int y, int x) { fresh[((size_t)f * H + y) * W + x] = (uint16_t)(4095 - (y % 16)); };
        hot2(1, 40, 50); hot2(1, 41, 50);
        for (int k = 1; k <= 7; ++k) hot2(3, 10 * k, 10 * k);
        hot2(5, 0, 7); hot2(5, 100, 100); hot2(5, 100, 101); hot2(5, 200, 13); hot2(5, 255, 255);
        up.h = &fresh;
        hipStream_t st = nullptr;
        HIPCHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        float *d_rt[3], *d_a[3], *d_r[3];
        for (int k = 0; k < 3; ++k) {
            HIPCHECK(hipMalloc(reinterpret_cast<void **>(&d_rt[k]), sizeof(float) * (size_t)ld * N));
            HIPCHECK(hipMalloc(reinterpret_cast<void **>(&d_a[k]), sizeof(float) * N));
            HIPCHECK(hipMalloc(reinterpret_cast<void **>(&d_r[k]), sizeof(float) * N));
        }
        upsp_step_args sa;
        std::memset(&sa, 0, sizeof(sa));
        sa.bvh = bvh; sa.cam = &cam; sa.d_nodes = d_nodes; sa.d_normals = d_normals; sa.d_tri_nodes = d_tri_nodes;
        sa.oblique_thresh = thresh; sa.nframes = F; sa.d_frames = d_frames; sa.first_frame = 0; sa.ld_t = ld; sa.nframes_total = (uint64_t)F;
        sa.frames_hook = [](void *user, void *stream) {
            const Upload *u = static_cast<const Upload *>(user);
            (void)hipMemcpyAsync(u->d, u->h->data(), sizeof(uint16_t) * u->h->size(), hipMemcpyHostToDevice, (hipStream_t)stream);
        };
        sa.frames_user = &up;
        for (int k = 0; k < 3; ++k) {
            sa.d_rows_t = d_rt[k]; sa.d_avg = d_a[k]; sa.d_rms = d_r[k];
            CHECK(upsp_pipeline_step(pipe, &sa, st));
        }
        CHECK(upsp_pipeline_step_finish(pipe, st));
        HIPCHECK(hipStreamSynchronize(st));
        for (int k = 0; k < 3; ++k) {
            std::vector<float> rt((size_t)ld * N), a2(N), r2(N);
            HIPCHECK(hipMemcpy(rt.data(), d_rt[k], sizeof(float) * rt.size(), hipMemcpyDeviceToHost));
            HIPCHECK(hipMemcpy(a2.data(), d_a[k], sizeof(float) * N, hipMemcpyDeviceToHost));
            HIPCHECK(hipMemcpy(r2.data(), d_r[k], sizeof(float) * N, hipMemcpyDeviceToHost));
            for (size_t n = 0; n < N; ++n) {
                for (int f = 0; f < F; ++f) bad_step += std::memcmp(&rt[n * (size_t)ld + f], &rows_t[n * (size_t)ld + f], 4) != 0;
                bad_step += std::memcmp(&a2[n], &avg[n], 4) != 0 || std::memcmp(&r2[n], &rms[n], 4) != 0;
            }
        }
        const int32_t *d_pix_now = nullptr;
        CHECK(upsp_pipeline_projection(pipe, 0, &d_pix_now));
        std::vector<int32_t> pix_now(N);
        HIPCHECK(hipMemcpy(pix_now.data(), d_pix_now, sizeof(int32_t) * N, hipMemcpyDeviceToHost));
        for (size_t n = 0; n < N; ++n) bad_step += pix_now[n] != want_pix[n];
        HIPCHECK(hipStreamDestroy(st));
    }
    upsp_pipeline_destroy(pipe);
    upsp_bvh_destroy(bvh);
    if (bad_step) {
        std::fprintf(stderr, "upsp_pipeline_step: %zu values differ from the plain frame loop\n", bad_step);
        return 1;
    }
    if (bad_rows || bad_t || bad_acc || bad_fin || (uint32_t)(sum_px & 0xFFFFFFFFu) != hdr[7]) {
        std::fprintf(stderr, "frame loop: %zu row values, %zu transposed values, %zu accumulators, %zu finals differ; pixel checksum %u (expected %u)\n",
                     bad_rows, bad_t, bad_acc, bad_fin, (uint32_t)(sum_px & 0xFFFFFFFFu), hdr[7]);
        return 1;
    }
    std::printf("frame loop ok: %zu triangles, %zu nodes, %zu visible, %llu rays, %d frames, %zu NaN rows; re-raycast loop (upsp_pipeline_step) ok\n",
                T, N, visible, (unsigned long long)nrays, F, nan_rows);
    return 0;
}
