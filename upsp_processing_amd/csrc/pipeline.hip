// Host orchestration of the per-frame pipeline: the body of the psp_process
// phase-1 frame loop (cpp/exec/psp_process.cpp:1743-1851) as a sequence of batched
// kernel launches on one HIP stream.
//
// Frames are processed in sub-batches sized so that a sub-batch of u16 frames
// (and its f32 working copies when patching / filtering is on) stays inside the
// 256 MiB Infinity Cache between the streaming hot-pixel scan and the gather:
// HBM then sees each frame byte once.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#include "pipeline.h"
#include "upsp_internal.h"

using namespace upsp;

struct upsp_pipeline {
    int ncams = 0, width = 0, height = 0;
    size_t nnodes = 0;
    upsp_pipeline_opts opts;
    int32_t *d_pix[kMaxCams] = {nullptr};
    int32_t *d_pix_next[kMaxCams] = {nullptr};      // upsp_pipeline_projection_target: where the caller builds the NEXT projection
    float *d_weight[kMaxCams] = {nullptr};
    bool has_proj[kMaxCams] = {false};
    unsigned *d_read_list[kMaxCams] = {nullptr};  // count + pixels with a node (registration as the last image stage)
    uint8_t *d_read_mask = nullptr;               // scratch of the list build
    bool read_list_valid[kMaxCams] = {false};
    uint8_t *d_skipped = nullptr;
    int32_t *d_src = nullptr;        // overlap source map (P3D adjust_solution), optional
    int32_t *d_rowmap = nullptr;     // packed time-series rows (node -> row, < 0 = not stored), optional
    bool skipped_user = false, skipped_valid = false;
    double *d_sum = nullptr, *d_sumsq = nullptr;
    size_t acc_stride = 0;          // d_sumsq = d_sum + acc_stride
    // hot-pixel scratch (per frame of a sub-batch)
    unsigned *d_hot_count = nullptr, *d_hot_pos = nullptr;
    int hot_capacity = 0;
    // streamed scan + projection: active-pixel map (built on first use after a projection change),
    // compact pixel-series buffer of a sub-batch, change list of the hot-pixel fix-up
    uint8_t *d_aflag = nullptr;
    unsigned *d_tile_off = nullptr, *d_tile_cnt = nullptr, *d_tile_order = nullptr;
    int32_t *d_node_k = nullptr;
    // a second set of the five map arrays: upsp_pipeline_set_active_hint builds the NEXT map in the set the launches already
    // queued do not read, so the caller may issue it on another stream beside the frame loop that still uses the current one
    uint8_t *alt_aflag = nullptr;
    unsigned *alt_tile_off = nullptr, *alt_tile_cnt = nullptr, *alt_tile_order = nullptr;
    int32_t *alt_node_k = nullptr;
    uint16_t *d_compact = nullptr;
    size_t compact_bytes = 0;        // allocated size of d_compact
    uint16_t *d_compact_alt = nullptr;   // upsp_pipeline_step: the other compact buffer (pass A of step s + 1 beside pass B of step s)
    size_t compact_bytes_alt = 0;
    unsigned *d_changes = nullptr;   // hot-pixel change list of a call (frames.hip: hot_changes_words)
    bool row_padding = false;        // upsp_pipeline_set_row_padding
    int changes_parity = 0;          // which of its two change counters the next one-camera fix-up uses
    bool acc_unset = false;          // upsp_pipeline_reset was called and the accumulators were not zeroed yet (done by whoever touches them first)
    size_t changes_words = 0;
    int32_t *d_head = nullptr, *d_next = nullptr;   // pixel -> nodes lists of the hot-pixel re-projection
    bool head_clean = false;        // d_head holds 'unmarked' everywhere (multi-camera fix-up leaves it so)
    size_t head_elems = 0, next_elems = 0;
    bool tilemap_valid = false;
    // candidate map (upsp_pipeline_set_active_hint): the map outlives projection changes, only node_k is redone
    bool hint_active = false, node_k_valid = false;
    bool scan_split = false;         // pass A in two launches (upsp_pipeline_set_scan_split)
    // pass A already run by upsp_pipeline_prescan for these frames (consumed by the next matching process call)
    const uint16_t *prescan_frames = nullptr;
    int prescan_n = 0;
    // generation of the active-pixel map (bumped by every rebuild / invalidation): a prescan is only consumed by a
    // process call that still sees the map its compact buffer was laid out with
    uint64_t map_gen = 0, prescan_gen = 0;
    unsigned *d_pix_of_k = nullptr;     // pixel of every compact row (registration straight into the compact buffer)
    uint64_t pixk_gen = ~0ull;
    // the same per camera for the multi-camera streamed schedule
    uint8_t *m_aflag[kMaxCams] = {nullptr};
    unsigned *m_tile_off[kMaxCams] = {nullptr};
    unsigned *m_tile_order[kMaxCams] = {nullptr};    // pass A's visiting order per camera (launch_amap_build)
    int32_t *m_node_k[kMaxCams] = {nullptr};
    uint16_t *m_compact[kMaxCams] = {nullptr};
    size_t m_compact_bytes[kMaxCams] = {0};
    bool m_valid[kMaxCams] = {false};
    // the same scratch for upsp_pipeline_fix_hot_pixels (may run on another stream than process)
    unsigned *d_pre_count = nullptr, *d_pre_pos = nullptr;
    int pre_capacity = 0;
    // registration / patch / filter state
    float *d_ref[kMaxCams] = {nullptr};
    int32_t *d_ecc_iters = nullptr;   // caller's buffer (upsp_pipeline_set_ecc_iterations_out), optional
    upsp::PatchTables *patches[kMaxCams] = {nullptr};
    upsp::FrameScratch *scratch = nullptr;
    int batch = 32;
    // upsp_pipeline_step: the side stream of the projection build (high priority: it needs few wave slots but needs them early), the
    // events that order it against the caller's stream, the candidate pixels / uv scratch of the build, the finals the step before
    // left to do
    struct Step {
        hipStream_t side = nullptr;
        hipStream_t side2 = nullptr;    // the builds of odd steps (two builds in flight: see upsp_pipeline_step)
        upsp_bvh *bvh2 = nullptr;       // ... on a handle of their own over the caller's tree (upsp_bvh_share: own query scratch)
        const upsp_bvh *bvh2_of = nullptr;
        int32_t *d_cand2 = nullptr;
        float *d_uv2 = nullptr;
        hipStream_t scan = nullptr;     // pass A + repair of a step, beside the previous step's pass B (normal priority)
        hipEvent_t ev_map[2] = {nullptr, nullptr}, ev_side[2] = {nullptr, nullptr}, ev_repaired[2] = {nullptr, nullptr};
        hipEvent_t ev_end[3] = {nullptr, nullptr, nullptr};
        int32_t *d_cand = nullptr;
        float *d_uv = nullptr;
        uint64_t count = 0;
        bool finals_due = false;
        float *d_avg = nullptr, *d_rms = nullptr;
        uint64_t ntotal = 0;
    } step;
};

namespace {

void free_dev(void *p)
{
    if (p) (void)hipFree(p);
}

int ensure_hot_scratch(unsigned *&count, unsigned *&pos, int &capacity, int nframes)
{
    if (nframes <= capacity) return UPSP_OK;
    free_dev(count);
    free_dev(pos);
    count = pos = nullptr;
    capacity = 0;
    UPSP_HIP_CHECK(hipMalloc(&count, sizeof(unsigned) * upsp::hot_counter_words(nframes)));   // counts + tickets
    UPSP_HIP_CHECK(hipMemset(count, 0, sizeof(unsigned) * upsp::hot_counter_words(nframes)));
    UPSP_HIP_CHECK(hipStreamSynchronize(nullptr));      // (allocation time only: done before a launch on any other stream can see the buffer)
    UPSP_HIP_CHECK(hipDeviceSynchronize());   // rare (allocation): zeroed before any stream uses it
    UPSP_HIP_CHECK(hipMalloc(&pos, sizeof(unsigned) * (size_t)nframes * 64));
    capacity = nframes;
    return UPSP_OK;
}

// The active-pixel map no longer describes what the next frame loop reads: a pass A that already ran
// (upsp_pipeline_prescan) laid its compact buffer out with the old map and must not be consumed.
void invalidate_map(upsp_pipeline *p)
{
    p->tilemap_valid = false;
    p->prescan_frames = nullptr;
    ++p->map_gen;
}

int ensure_hot(upsp_pipeline *p, int nframes)
{
    return ensure_hot_scratch(p->d_hot_count, p->d_hot_pos, p->hot_capacity, nframes);
}

}  // namespace

extern "C" {

void upsp_pipeline_default_opts(upsp_pipeline_opts *o)
{
    if (!o) return;
    std::memset(o, 0, sizeof(*o));
    o->hot_enable = 1;
    o->hot_thresh = 4064;     // cpp/include/utils/cv_extras.h:154-155
    o->hot_min_change = 512;
    o->hot_max = 5;
    o->registration = 0;      // RegistrationType::None (cpp/lib/upsp_inputs.cpp:29-33)
    o->ecc_max_iters = 50;
    o->ecc_eps = 1e-3;
    o->interp = 1;
    o->filter = 0;
    o->filter_size = 1;
    o->patch = 0;
    o->fused_scan = 0;
}

int upsp_pipeline_create(int ncams, int width, int height, size_t nnodes,
                         const upsp_pipeline_opts *opts, upsp_pipeline **out)
{
    if (!out) return fail(UPSP_ERR_INVALID, "out is null");
    *out = nullptr;
    if (ncams <= 0 || ncams > kMaxCams) return fail(UPSP_ERR_INVALID, "ncams out of range");
    if (width <= 0 || height <= 0) return fail(UPSP_ERR_INVALID, "bad image size");
    if (nnodes == 0 || nnodes > 0xF0000000ull) return fail(UPSP_ERR_INVALID, "bad node count");
    upsp_pipeline *p = new upsp_pipeline();
    p->ncams = ncams;
    p->width = width;
    p->height = height;
    p->nnodes = nnodes;
    if (opts)
        p->opts = *opts;
    else
        upsp_pipeline_default_opts(&p->opts);
    if (p->opts.filter != 0 && (p->opts.filter_size < 1 || (p->opts.filter_size % 2) == 0)) {
        delete p;
        return fail(UPSP_ERR_INVALID, "filter_size must be odd");  // psp_process.cpp:1296
    }
    // the two accumulators in ONE allocation (sumsq behind sum, 256-byte aligned): upsp_pipeline_reset is one fill
    p->acc_stride = (nnodes + 31) & ~(size_t)31;
    hipError_t e = hipMalloc(&p->d_sum, sizeof(double) * 2 * p->acc_stride);
    if (e == hipSuccess) p->d_sumsq = p->d_sum + p->acc_stride;
    if (e == hipSuccess) e = hipMalloc(&p->d_skipped, nnodes);
    if (e == hipSuccess) e = hipMemset(p->d_sum, 0, sizeof(double) * 2 * p->acc_stride);
    if (e == hipSuccess) e = hipMemset(p->d_skipped, 0, nnodes);
    if (e != hipSuccess) {
        upsp_pipeline_destroy(p);
        return fail(UPSP_ERR_HIP, std::string("pipeline alloc: ") + hipGetErrorString(e));
    }
    // sub-batch: one gather tile = 64 frames; without image stages that is also what keeps
    // the u16 frames of a sub-batch (128 MiB at 1 Mpix) inside the Infinity Cache between
    // the hot-pixel scan and the gather.  With registration the ECC iterations run in lock
    // step over the sub-batch and late iterations have few active frames: 64 frames per
    // launch keep the chip busy.  Patch / filter only: bounded by the f32 working copies.
    const size_t per_frame = (size_t)width * height * (size_t)ncams *
                             (2 + ((p->opts.patch || p->opts.filter) ? 4 : 0));
    size_t b = (p->opts.registration ? (512u << 20) : (128u << 20)) / std::max<size_t>(per_frame, 1);
    // plain projection path: always a full 64-frame tile.  With several cameras the frames of a
    // sub-batch no longer fit the Infinity Cache, but short tiles cost more than the misses
    // (4 cameras, 5 M triangles: 0.84 / 0.65 / 0.57 ms per 64 frame sets at 16 / 32 / 64 frames)
    if (!(p->opts.registration || p->opts.patch || p->opts.filter)) b = 64;
    // patch / filter: the f32 working copies of a sub-batch should stay cache-resident between
    // the two blur passes and the gather; 32 frames measured best at 1 Mpix (21: -20 %, 64: -20 %)
    else if (!p->opts.registration) b = std::max<size_t>(b, std::min<size_t>(32, (256u << 20) / std::max<size_t>(per_frame, 1)));
    p->batch = (int)std::min<size_t>(std::max<size_t>(b, 1), 64);
    *out = p;
    return UPSP_OK;
}

void upsp_pipeline_destroy(upsp_pipeline *p)
{
    if (!p) return;
    for (int c = 0; c < kMaxCams; ++c) {
        free_dev(p->d_pix[c]);
        free_dev(p->d_pix_next[c]);
        free_dev(p->d_weight[c]);
        free_dev(p->d_read_list[c]);
        free_dev(p->d_ref[c]);
        upsp::patch_tables_free(p->patches[c]);
    }
    upsp::frame_scratch_free(p->scratch);
    free_dev(p->d_read_mask);
    free_dev(p->d_skipped);
    free_dev(p->d_src);
    free_dev(p->d_rowmap);
    free_dev(p->d_sum);             // (d_sumsq lives in the same allocation)
    free_dev(p->d_hot_count);
    free_dev(p->d_hot_pos);
    free_dev(p->d_pre_count);
    free_dev(p->d_pre_pos);
    free_dev(p->d_aflag);
    free_dev(p->d_tile_off);
    free_dev(p->d_tile_cnt);
    free_dev(p->d_tile_order);
    free_dev(p->d_node_k);
    free_dev(p->alt_aflag);
    free_dev(p->alt_tile_off);
    free_dev(p->alt_tile_cnt);
    free_dev(p->alt_tile_order);
    free_dev(p->alt_node_k);
    free_dev(p->d_compact);
    free_dev(p->d_compact_alt);
    free_dev(p->d_pix_of_k);
    free_dev(p->d_changes);
    free_dev(p->d_head);
    free_dev(p->d_next);
    if (p->step.side) {
        (void)hipStreamSynchronize(p->step.side);
        (void)hipStreamDestroy(p->step.side);
    }
    if (p->step.scan) {
        (void)hipStreamSynchronize(p->step.scan);
        (void)hipStreamDestroy(p->step.scan);
    }
    if (p->step.side2) {
        (void)hipStreamSynchronize(p->step.side2);
        (void)hipStreamDestroy(p->step.side2);
    }
    if (p->step.bvh2) upsp_bvh_destroy(p->step.bvh2);
    free_dev(p->step.d_cand2);
    free_dev(p->step.d_uv2);
    for (hipEvent_t e : {p->step.ev_map[0], p->step.ev_map[1], p->step.ev_side[0], p->step.ev_side[1], p->step.ev_repaired[0],
                         p->step.ev_repaired[1], p->step.ev_end[0], p->step.ev_end[1], p->step.ev_end[2]})
        if (e) (void)hipEventDestroy(e);
    free_dev(p->step.d_cand);
    free_dev(p->step.d_uv);
    for (int c = 0; c < kMaxCams; ++c) {
        free_dev(p->m_aflag[c]);
        free_dev(p->m_tile_off[c]);
        free_dev(p->m_tile_order[c]);
        free_dev(p->m_node_k[c]);
        free_dev(p->m_compact[c]);
    }
    delete p;
}

int upsp_pipeline_set_projection(upsp_pipeline *p, int cam, const int32_t *d_pix,
                                 const float *d_weight)
{
    if (!p || cam < 0 || cam >= p->ncams || !d_pix) return fail(UPSP_ERR_INVALID, "bad argument");
    if (d_pix == p->d_pix_next[cam]) {
        std::swap(p->d_pix[cam], p->d_pix_next[cam]);      // built in place (upsp_pipeline_projection_target): no copy
    } else if (d_pix != p->d_pix[cam]) {                   // (the current buffer itself: rewritten in place by the caller)
        if (!p->d_pix[cam]) UPSP_HIP_CHECK(hipMalloc(&p->d_pix[cam], sizeof(int32_t) * p->nnodes));
        UPSP_HIP_CHECK(hipMemcpy(p->d_pix[cam], d_pix, sizeof(int32_t) * p->nnodes,
                                 hipMemcpyDeviceToDevice));
    }
    if (d_weight) {
        if (!p->d_weight[cam]) UPSP_HIP_CHECK(hipMalloc(&p->d_weight[cam], sizeof(float) * p->nnodes));
        UPSP_HIP_CHECK(hipMemcpy(p->d_weight[cam], d_weight, sizeof(float) * p->nnodes,
                                 hipMemcpyDeviceToDevice));
    } else {
        free_dev(p->d_weight[cam]);
        p->d_weight[cam] = nullptr;
    }
    p->has_proj[cam] = true;
    p->read_list_valid[cam] = false;
    p->m_valid[cam] = false;
    if (!p->hint_active) invalidate_map(p);
    p->node_k_valid = false;
    if (!p->skipped_user) p->skipped_valid = false;
    return UPSP_OK;
}

int upsp_pipeline_set_projection_async(upsp_pipeline *p, int cam, const int32_t *d_pix,
                                       const float *d_weight, void *stream)
{
    if (!p || cam < 0 || cam >= p->ncams || !d_pix) return fail(UPSP_ERR_INVALID, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (d_pix == p->d_pix_next[cam]) {
        std::swap(p->d_pix[cam], p->d_pix_next[cam]);      // built in place (upsp_pipeline_projection_target): no copy
    } else if (d_pix != p->d_pix[cam]) {
        if (!p->d_pix[cam]) UPSP_HIP_CHECK(hipMalloc(&p->d_pix[cam], sizeof(int32_t) * p->nnodes));
        UPSP_HIP_CHECK(hipMemcpyAsync(p->d_pix[cam], d_pix, sizeof(int32_t) * p->nnodes,
                                      hipMemcpyDeviceToDevice, st));
    }
    if (d_weight) {
        if (!p->d_weight[cam]) UPSP_HIP_CHECK(hipMalloc(&p->d_weight[cam], sizeof(float) * p->nnodes));
        UPSP_HIP_CHECK(hipMemcpyAsync(p->d_weight[cam], d_weight, sizeof(float) * p->nnodes,
                                      hipMemcpyDeviceToDevice, st));
    } else if (p->d_weight[cam]) {
        // a weight vector may still be read by launches queued earlier: release it stream-ordered
        UPSP_HIP_CHECK(hipStreamSynchronize(st));
        free_dev(p->d_weight[cam]);
        p->d_weight[cam] = nullptr;
    }
    p->has_proj[cam] = true;
    p->read_list_valid[cam] = false;
    p->m_valid[cam] = false;
    if (!p->hint_active) invalidate_map(p);
    p->node_k_valid = false;
    if (!p->skipped_user) p->skipped_valid = false;
    return UPSP_OK;
}

int upsp_pipeline_projection_target(upsp_pipeline *p, int cam, int32_t **d_pix)
{
    if (!p || cam < 0 || cam >= p->ncams || !d_pix) return fail(UPSP_ERR_INVALID, "bad argument");
    if (!p->d_pix_next[cam]) UPSP_HIP_CHECK(hipMalloc(&p->d_pix_next[cam], sizeof(int32_t) * p->nnodes));
    *d_pix = p->d_pix_next[cam];
    return UPSP_OK;
}

int upsp_pipeline_projection(upsp_pipeline *p, int cam, const int32_t **d_pix)
{
    if (!p || cam < 0 || cam >= p->ncams || !d_pix) return fail(UPSP_ERR_INVALID, "bad argument");
    if (!p->has_proj[cam]) return fail(UPSP_ERR_INVALID, "projection not set");
    *d_pix = p->d_pix[cam];
    return UPSP_OK;
}

int upsp_pipeline_set_hot_enable(upsp_pipeline *p, int enable)
{
    if (!p) return fail(UPSP_ERR_INVALID, "null pipeline");
    p->opts.hot_enable = enable ? 1 : 0;
    return UPSP_OK;
}

int upsp_pipeline_fix_hot_pixels(upsp_pipeline *p, uint16_t *d_frames, int nframes, void *stream)
{
    if (!p || nframes < 0 || (nframes > 0 && !d_frames)) return fail(UPSP_ERR_INVALID, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    const size_t npix = (size_t)p->width * p->height;
    // same 64-frame launches as the frame loop (the counters clean themselves between launches)
    const int B = 64;
    int rc = ensure_hot_scratch(p->d_pre_count, p->d_pre_pos, p->pre_capacity, std::min(B, std::max(nframes, 1)));
    for (int f0 = 0; f0 < nframes && rc == UPSP_OK; f0 += B)
        rc = launch_hot_fix(d_frames + (size_t)f0 * npix, std::min(B, nframes - f0), p->height, p->width,
                            p->opts.hot_thresh, p->opts.hot_min_change, p->opts.hot_max,
                            p->d_pre_count, p->d_pre_pos, nullptr, st);
    return rc;
}

int upsp_pipeline_set_skipped(upsp_pipeline *p, const uint8_t *d_skipped)
{
    if (!p) return fail(UPSP_ERR_INVALID, "null pipeline");

    if (d_skipped) {
        UPSP_HIP_CHECK(hipMemcpy(p->d_skipped, d_skipped, p->nnodes, hipMemcpyDeviceToDevice));
        p->skipped_user = true;
        p->skipped_valid = true;
    } else {
        p->skipped_user = false;
        p->skipped_valid = false;
    }
    return UPSP_OK;
}

int upsp_pipeline_set_overlap_source(upsp_pipeline *p, const int32_t *d_src)
{
    if (!p) return fail(UPSP_ERR_INVALID, "null pipeline");
    if (!d_src) {
        if (p->d_src) (void)hipFree(p->d_src);
        p->d_src = nullptr;
        return UPSP_OK;
    }
    if (!p->d_src) UPSP_HIP_CHECK(hipMalloc(&p->d_src, sizeof(int32_t) * std::max<size_t>(p->nnodes, 1)));
    UPSP_HIP_CHECK(hipMemcpy(p->d_src, d_src, sizeof(int32_t) * p->nnodes, hipMemcpyDeviceToDevice));
    return UPSP_OK;
}

static int set_row_map_impl(upsp_pipeline *p, const int32_t *d_rowmap, bool async, hipStream_t st)
{
    if (!p) return fail(UPSP_ERR_INVALID, "null pipeline");
    if (!d_rowmap) {
        if (p->d_rowmap) {
            if (async) UPSP_HIP_CHECK(hipStreamSynchronize(st));   // launches queued earlier may still read it
            (void)hipFree(p->d_rowmap);
        }
        p->d_rowmap = nullptr;
        return UPSP_OK;
    }
    if (!p->d_rowmap) UPSP_HIP_CHECK(hipMalloc(&p->d_rowmap, sizeof(int32_t) * std::max<size_t>(p->nnodes, 1)));
    if (async)
        UPSP_HIP_CHECK(hipMemcpyAsync(p->d_rowmap, d_rowmap, sizeof(int32_t) * p->nnodes, hipMemcpyDeviceToDevice, st));
    else
        UPSP_HIP_CHECK(hipMemcpy(p->d_rowmap, d_rowmap, sizeof(int32_t) * p->nnodes, hipMemcpyDeviceToDevice));
    return UPSP_OK;
}

int upsp_pipeline_set_row_map(upsp_pipeline *p, const int32_t *d_rowmap)
{
    return set_row_map_impl(p, d_rowmap, false, nullptr);
}

int upsp_pipeline_set_row_map_async(upsp_pipeline *p, const int32_t *d_rowmap, void *stream)
{
    return set_row_map_impl(p, d_rowmap, true, (hipStream_t)stream);
}

int upsp_pipeline_set_scan_split(upsp_pipeline *p, int on)
{
    if (!p) return fail(UPSP_ERR_INVALID, "null pipeline");
    p->scan_split = on != 0;
    return UPSP_OK;
}

int upsp_pipeline_set_row_padding(upsp_pipeline *p, int on)
{
    if (!p) return fail(UPSP_ERR_INVALID, "null pipeline");
    p->row_padding = on != 0;
    return UPSP_OK;
}

// columns a whole-row pass B may store for `ns` frames that start at column `c0` of rows of pitch ld: up to the next 128-byte line
// (32 floats / 64 u16) when the caller declared the pitch padding writable AND the call ends within 512 bytes of the pitch, i.e. at
// the end of the row (engine.series_ld rounds a row up to 256 bytes and may add 256 more).  A call that fills a window elsewhere in a wider
// matrix (chunks in any order, live data to the right) stores its own columns and nothing else.
static int padded_store(const upsp_pipeline *p, int64_t c0, int ns, int64_t ld, int per_line)
{
    if (!p->row_padding || (ld % per_line) != 0) return ns;
    const int64_t end = c0 + ns;
    if (ld - end >= 4 * per_line) return ns;
    const int64_t stop = std::min<int64_t>(ld, (end + per_line - 1) / per_line * per_line);
    return ns + (int)(stop - end);
}

int upsp_pipeline_set_reference(upsp_pipeline *p, int cam, const float *d_ref32f)
{
    if (!p || cam < 0 || cam >= p->ncams || !d_ref32f) return fail(UPSP_ERR_INVALID, "bad argument");
    const size_t bytes = sizeof(float) * (size_t)p->width * p->height;
    if (!p->d_ref[cam]) UPSP_HIP_CHECK(hipMalloc(&p->d_ref[cam], bytes));
    UPSP_HIP_CHECK(hipMemcpy(p->d_ref[cam], d_ref32f, bytes, hipMemcpyDeviceToDevice));
    upsp::frame_scratch_new_reference(p->scratch, cam);
    return UPSP_OK;
}

int upsp_pipeline_set_patches(upsp_pipeline *p, int cam, int nclusters, const int32_t *h_b_off,
                              const int32_t *h_bx, const int32_t *h_by, const int32_t *h_i_off,
                              const int32_t *h_ix, const int32_t *h_iy)
{
    if (!p || cam < 0 || cam >= p->ncams) return fail(UPSP_ERR_INVALID, "bad argument");
    upsp::patch_tables_free(p->patches[cam]);
    p->patches[cam] = nullptr;
    if (nclusters == 0) return UPSP_OK;
    return upsp::patch_tables_create(p->height, p->width, nclusters, h_b_off, h_bx, h_by, h_i_off,
                                     h_ix, h_iy, &p->patches[cam]);
}

static int acc_zero_now(upsp_pipeline *p, hipStream_t st, bool on_stream);

int upsp_pipeline_accumulators(upsp_pipeline *p, double **d_sum, double **d_sumsq)
{
    if (!p) return fail(UPSP_ERR_INVALID, "null pipeline");
    int rc = acc_zero_now(p, nullptr, false);      // (a deferred reset that nothing has acted on yet: the caller is about to read or add)
    if (rc != UPSP_OK) return rc;
    if (d_sum) *d_sum = p->d_sum;
    if (d_sumsq) *d_sumsq = p->d_sumsq;
    return UPSP_OK;
}

int upsp_pipeline_accumulators_async(upsp_pipeline *p, double **d_sum, double **d_sumsq, void *stream)
{
    if (!p) return fail(UPSP_ERR_INVALID, "null pipeline");
    int rc = acc_zero_now(p, (hipStream_t)stream, true);      // (a pending deferred reset: cleared on the stream that will use them)
    if (rc != UPSP_OK) return rc;
    if (d_sum) *d_sum = p->d_sum;
    if (d_sumsq) *d_sumsq = p->d_sumsq;
    return UPSP_OK;
}

int upsp_pipeline_ecc_stats(upsp_pipeline *p, uint64_t *frame_iterations, uint64_t *frames)
{
    if (!p || !frame_iterations || !frames) return fail(UPSP_ERR_INVALID, "bad argument");
    unsigned long long a = 0, b = 0;
    upsp::frame_scratch_ecc_stats(p->scratch, &a, &b);
    *frame_iterations = a;
    *frames = b;
    return UPSP_OK;
}

int upsp_pipeline_set_ecc_iterations_out(upsp_pipeline *p, int32_t *d_iters)
{
    if (!p) return fail(UPSP_ERR_INVALID, "null pipeline");
    p->d_ecc_iters = d_iters;
    return UPSP_OK;
}

// After upsp_pipeline_reset_deferred the accumulators are zeroed by whoever touches them first: the streamed one-camera loop WRITES
// them in its first pass B (no fill launch, no read of 8 B x N), everything else clears them first -- on the stream that is about
// to use them, or (no stream at hand: upsp_pipeline_accumulators) after waiting for the device, because a null-stream memset is
// not ordered against work queued on non-blocking streams.
static int acc_zero_now(upsp_pipeline *p, hipStream_t st, bool on_stream)
{
    if (!p->acc_unset) return UPSP_OK;
    if (on_stream) {
        UPSP_HIP_CHECK(hipMemsetAsync(p->d_sum, 0, sizeof(double) * 2 * p->acc_stride, st));
    } else {
        UPSP_HIP_CHECK(hipDeviceSynchronize());
        UPSP_HIP_CHECK(hipMemset(p->d_sum, 0, sizeof(double) * 2 * p->acc_stride));
        UPSP_HIP_CHECK(hipDeviceSynchronize());
    }
    p->acc_unset = false;
    return UPSP_OK;
}

int upsp_pipeline_reset(upsp_pipeline *p)
{
    if (!p) return fail(UPSP_ERR_INVALID, "null pipeline");
    p->acc_unset = true;
    return acc_zero_now(p, nullptr, false);       // eager: zero when the call returns
}

int upsp_pipeline_reset_deferred(upsp_pipeline *p)
{
    if (!p) return fail(UPSP_ERR_INVALID, "null pipeline");
    p->acc_unset = true;
    return UPSP_OK;
}

int upsp_pipeline_finalize(upsp_pipeline *p, uint64_t nframes_total, float *d_avg, float *d_rms,
                           void *stream)
{
    if (!p || nframes_total == 0) return fail(UPSP_ERR_INVALID, "bad argument");
    int rc = acc_zero_now(p, (hipStream_t)stream, true);
    if (rc != UPSP_OK) return rc;
    return launch_finals(p->d_sum, p->d_sumsq, p->nnodes, nframes_total, d_avg, d_rms,
                         (hipStream_t)stream);
}

// ---- streamed schedule, one camera: shared by upsp_pipeline_prescan and the frame loop ----
// Builds the active-pixel map from `d_pix_src` (the projection, or a candidate set) if there is none.
static int streamed_map(upsp_pipeline *p, const int32_t *d_pix_src, size_t npix, hipStream_t st, bool want_nodes = true)
{
    if (p->tilemap_valid) return UPSP_OK;
    const size_t ntiles = tilemap_tiles(npix);
    if (!p->d_aflag) {
        UPSP_HIP_CHECK(hipMalloc(&p->d_aflag, npix));
        // (once: launch_amap_build leaves the flags in a state it can start from.  On `st`: a memset on the null stream is not
        //  ordered with a non-blocking stream, and the map may be built on one)
        UPSP_HIP_CHECK(hipMemsetAsync(p->d_aflag, 0, npix, st));
    }
    if (!p->d_tile_off) UPSP_HIP_CHECK(hipMalloc(&p->d_tile_off, sizeof(unsigned) * (ntiles + 1)));
    if (!p->d_tile_cnt) UPSP_HIP_CHECK(hipMalloc(&p->d_tile_cnt, sizeof(unsigned) * ntiles));
    if (!p->d_node_k) UPSP_HIP_CHECK(hipMalloc(&p->d_node_k, sizeof(int32_t) * p->nnodes));
    if (!p->d_tile_order) UPSP_HIP_CHECK(hipMalloc(&p->d_tile_order, sizeof(unsigned) * 4 * (ntiles + 1)));
    int rc = launch_amap_build(d_pix_src, p->nnodes, npix, p->d_aflag, p->d_tile_cnt, p->d_tile_off,
                               want_nodes ? p->d_node_k : nullptr, p->d_tile_order, st);
    if (rc != UPSP_OK) return rc;
    p->tilemap_valid = true;
    p->prescan_frames = nullptr;     // a rebuilt map: whatever pass A wrote before belongs to the old one
    ++p->map_gen;
    return UPSP_OK;
}

// Frames per group (one pass A launch + one pass B launch): as many as the compact buffer may hold, at most
// group_frames_max().  One series of `cp` u16 per active pixel.  The number of active pixels is known on the
// device only; reading it back costs a stream round trip per projection, after which the host has to issue the
// whole frame loop with the GPU idle (measured with the projection rebuilt every step: chunked N > 1 loop
// 0.98 -> 3.0 ms per 1000 frames).  So the buffer is sized for the bound min(nodes, pixels), within the budget
// opts.compact_mb.  Also sizes the hot-pixel scratch of the call.
// frames one pass A / pass B group holds: the compact buffer [min(nodes, pixels)][frames] u16 within opts.compact_mb
static int streamed_group_frames(const upsp_pipeline *p, size_t npix)
{
    const size_t budget = (size_t)(p->opts.compact_mb > 0 ? p->opts.compact_mb : 2048) << 20;
    const size_t nact = std::max<size_t>(std::min(p->nnodes, npix), 1);
    const int S = (int)std::min<size_t>((size_t)group_frames_max(), (budget / (2 * nact)) / 64 * 64);
    return std::max(S, 64);
}

int upsp_pipeline_series_frames_max(const upsp_pipeline *p)
{
    if (!p) return fail(UPSP_ERR_INVALID, "bad argument");
    return streamed_group_frames(p, (size_t)p->width * p->height);
}

static int streamed_buffers(upsp_pipeline *p, size_t npix, int nframes, hipStream_t st, int *S_out, unsigned *cp_out)
{
    const size_t nact = std::max<size_t>(std::min(p->nnodes, npix), 1);
    const int S = streamed_group_frames(p, npix);
    const unsigned cp = (unsigned)((std::min(nframes, S) + 63) / 64 * 64);
    if (nact * cp * 2 > p->compact_bytes) {
        if (p->d_compact) {
            UPSP_HIP_CHECK(hipStreamSynchronize(st));    // launches queued earlier may still use it
            free_dev(p->d_compact);
            p->d_compact = nullptr;
            p->compact_bytes = 0;
        }
        UPSP_HIP_CHECK(hipMalloc(&p->d_compact, nact * cp * 2));
        p->compact_bytes = nact * cp * 2;
    }
    if (p->opts.hot_enable) {
        int rc = ensure_hot(p, nframes);   // one counter per frame of the CALL
        if (rc != UPSP_OK) return rc;
        const size_t words = hot_changes_words(nframes, p->opts.hot_max);
        if (words > p->changes_words) {
            if (p->d_changes) UPSP_HIP_CHECK(hipStreamSynchronize(st));
            free_dev(p->d_changes);
            p->d_changes = nullptr;
            p->changes_words = 0;
            UPSP_HIP_CHECK(hipMalloc(&p->d_changes, sizeof(unsigned) * words));
                UPSP_HIP_CHECK(hipMemsetAsync(p->d_changes, 0, sizeof(unsigned) * 4, st));      // (the change counters)
                p->changes_parity = 0;
            p->changes_words = words;
        }
    }
    *S_out = S;
    *cp_out = cp;
    return UPSP_OK;
}

static int streamed_pass_a(upsp_pipeline *p, uint16_t *fr, size_t npix, int s0, int ns, unsigned cp, hipStream_t st)
{
    const bool hot = p->opts.hot_enable != 0;
    return launch_scan_compact(fr + (size_t)s0 * npix, npix, ns, hot, p->opts.hot_thresh, p->opts.hot_max, p->d_aflag,
                               p->d_tile_off, p->d_tile_order, p->d_compact, cp, 0,
                               hot ? p->d_hot_count + s0 : nullptr, hot ? p->d_hot_pos + (size_t)s0 * 64 : nullptr, st, p->scan_split);
}

// fix_hot_pixels for the frames of a pass A group, between pass A (which counted and listed the hot pixels) and pass B: the frames
// are repaired in place and the replaced pixels written into the compact series, so pass B reads repaired values and nothing is
// left to correct behind it.  (Until round 5 the plain loop repaired AFTER pass B and patched the rows and accumulators of the
// nodes on the replaced pixels -- a sweep over all nodes per call.  Same step time either way, 0.874-0.881 against 0.879-0.889 ms:
// 36 us between the passes against 26 us behind them, pass B 0.36 against 0.38 ms.  This order needs no node sweep, and a prescan
// that is dropped leaves no counts behind.)
static int streamed_repair(upsp_pipeline *p, uint16_t *fr, size_t npix, int s0, int ns, unsigned cp, hipStream_t st)
{
    if (!p->opts.hot_enable) return UPSP_OK;
    return launch_hot_repair_compact(fr + (size_t)s0 * npix, npix, ns, p->height, p->width, p->opts.hot_min_change, p->opts.hot_max,
                                     p->d_hot_count + s0, p->d_hot_pos + (size_t)s0 * 64, p->d_changes, &p->changes_parity, p->d_aflag,
                                     p->d_tile_off, p->d_compact, cp, st);
}

int upsp_pipeline_set_active_hint(upsp_pipeline *p, const int32_t *d_pix_candidates, void *stream)
{
    if (!p) return fail(UPSP_ERR_INVALID, "null pipeline");
    if (!d_pix_candidates) {
        p->hint_active = false;
        invalidate_map(p);
        p->node_k_valid = false;
        return UPSP_OK;
    }
    const size_t npix = (size_t)p->width * p->height;
    if (p->ncams != 1 || (npix % 2) != 0 || p->nnodes >= ((size_t)1 << 31))
        return fail(UPSP_ERR_INVALID, "active hint: one camera, even pixel count");
    invalidate_map(p);
    // the other set of map arrays: whatever is queued (pass B, the hot-pixel fix-up, an exchange's node table of the step before)
    // keeps reading the current one; the set taken here was last used two maps ago (the caller orders THAT, if its streams can
    // run that far apart)
    std::swap(p->d_aflag, p->alt_aflag);
    std::swap(p->d_tile_off, p->alt_tile_off);
    std::swap(p->d_tile_cnt, p->alt_tile_cnt);
    std::swap(p->d_tile_order, p->alt_tile_order);
    std::swap(p->d_node_k, p->alt_node_k);
    int rc = streamed_map(p, d_pix_candidates, npix, (hipStream_t)stream, /*want_nodes=*/false);   // (node rows: with the projection)
    if (rc != UPSP_OK) return rc;
    p->hint_active = true;
    p->node_k_valid = false;
    p->prescan_frames = nullptr;
    return UPSP_OK;
}

int upsp_pipeline_prescan(upsp_pipeline *p, uint16_t *d_frames, int nframes, void *stream)
{
    if (!p || !d_frames || nframes <= 0) return fail(UPSP_ERR_INVALID, "bad argument");
    const size_t npix = (size_t)p->width * p->height;
    if (p->ncams != 1 || p->d_weight[0] || p->opts.registration || p->opts.patch || p->opts.filter || p->d_src ||
        (npix % 2) != 0 || p->batch != 64 || p->opts.fused_scan == 2)
        return fail(UPSP_ERR_INVALID, "prescan: plain one-camera path with the streamed schedule only");
    if (!p->tilemap_valid) {
        if (!p->has_proj[0]) return fail(UPSP_ERR_INVALID, "prescan: neither an active hint nor a projection is set");
        int rc = streamed_map(p, p->d_pix[0], npix, (hipStream_t)stream);
        if (rc != UPSP_OK) return rc;
        p->node_k_valid = true;
    }
    int S = 0;
    unsigned cp = 0;
    int rc = streamed_buffers(p, npix, nframes, (hipStream_t)stream, &S, &cp);
    if (rc != UPSP_OK) return rc;
    if (nframes > S) return fail(UPSP_ERR_INVALID, "prescan: more frames than one pass A / pass B group holds");
    rc = streamed_pass_a(p, d_frames, npix, 0, nframes, cp, (hipStream_t)stream);
    if (rc == UPSP_OK) rc = streamed_repair(p, d_frames, npix, 0, nframes, cp, (hipStream_t)stream);
    if (rc != UPSP_OK) return rc;
    p->prescan_frames = d_frames;
    p->prescan_n = nframes;
    p->prescan_gen = p->map_gen;
    return UPSP_OK;
}

// The node -> series-row table (and the skipped flags) of the current projection, on `stream`: the block the streamed frame
// loop would run at the head of its next call (process_impl), for callers that have a stream on which the projection is ready
// earlier than on the frame loop's.
int upsp_pipeline_prepare_rows(upsp_pipeline *p, void *stream)
{
    if (!p) return fail(UPSP_ERR_INVALID, "null pipeline");
    const size_t npix = (size_t)p->width * p->height;
    if (p->ncams != 1 || !p->has_proj[0] || p->d_weight[0] || p->opts.registration || p->opts.patch || p->opts.filter || p->d_src ||
        (npix % 2) != 0 || p->batch != 64 || p->opts.fused_scan == 2 || p->nnodes >= ((size_t)1 << 31))
        return UPSP_OK;                       // (not the streamed plain path: the process call prepares what it needs)
    hipStream_t st = (hipStream_t)stream;
    if (!p->tilemap_valid) {
        int rc = streamed_map(p, p->d_pix[0], npix, st);
        if (rc != UPSP_OK) return rc;
        p->node_k_valid = true;
        p->hint_active = false;
    }
    if (!p->node_k_valid) {
        const bool with_flags = !p->skipped_valid;
        int rc = launch_amap_nodes(p->d_pix[0], p->nnodes, p->d_aflag, p->d_tile_off, p->d_node_k, st, with_flags ? p->d_skipped : nullptr);
        if (rc != UPSP_OK) return rc;
        p->node_k_valid = true;
        if (with_flags) p->skipped_valid = true;
    }
    return UPSP_OK;
}

int upsp_pipeline_row_tables(upsp_pipeline *p, const int32_t **d_node_k, const uint8_t **d_skipped)
{
    if (!p) return fail(UPSP_ERR_INVALID, "null pipeline");
    if (!p->node_k_valid || !p->skipped_valid || !p->d_node_k || !p->d_skipped)
        return fail(UPSP_ERR_INVALID, "row tables: not derived for the current projection (upsp_pipeline_prepare_rows)");
    if (d_node_k) *d_node_k = p->d_node_k;
    if (d_skipped) *d_skipped = p->d_skipped;
    return UPSP_OK;
}

// Pass A as an operator (pixel-series exchange: the sender runs pass A only, the owner of a node runs pass B over all
// frames of the run).  Leaves the REPAIRED series of every active pixel over these frames in the pipeline's compact
// buffer and repairs the frames in place.
int upsp_pipeline_pixel_series(upsp_pipeline *p, uint16_t *d_frames, int nframes, void *stream, const uint16_t **d_compact,
                               uint32_t *cpitch, const int32_t **d_node_k, const uint32_t **d_nactive)
{
    // (nframes == 0: only the active-pixel map and the node -> row table of the current projection are made ready)
    if (!p || nframes < 0 || (nframes > 0 && !d_frames)) return fail(UPSP_ERR_INVALID, "bad argument");
    const size_t npix = (size_t)p->width * p->height;
    if (p->ncams != 1 || p->d_weight[0] || p->opts.registration || p->opts.patch || p->opts.filter || p->d_src ||
        (npix % 2) != 0 || p->nnodes >= ((size_t)1 << 31))
        return fail(UPSP_ERR_INVALID, "pixel series: plain one-camera path only (integer-valued series)");
    if (!p->has_proj[0]) return fail(UPSP_ERR_INVALID, "pixel series: projection not set");
    hipStream_t st = (hipStream_t)stream;
    int rc = UPSP_OK;
    // Pass A of exactly these frames may already have run on a candidate-pixel map (upsp_pipeline_set_active_hint +
    // upsp_pipeline_prescan, e.g. beside the projection build; the caller orders the streams): then the map stays, the nodes
    // get their rows in THAT buffer, and only the hot-pixel repair is left.  The candidate set must hold every pixel a node
    // reads: a node outside it gets node_k == -2, which upsp_exchange_set_pixels refuses (its owner would have no series).
    const bool prescanned = p->hint_active && p->tilemap_valid && nframes > 0 && p->prescan_frames == d_frames &&
                            p->prescan_n == nframes && p->prescan_gen == p->map_gen;
    if (p->hint_active && !prescanned) {
        // otherwise what is exchanged follows from the projection alone: a candidate map is a superset chosen by the caller,
        // its series would carry pixels no node reads
        p->hint_active = false;
        invalidate_map(p);
        p->node_k_valid = false;
    }
    if (!p->tilemap_valid) {
        rc = streamed_map(p, p->d_pix[0], npix, st);
        if (rc != UPSP_OK) return rc;
        p->node_k_valid = true;
        p->hint_active = false;
    }
    if (!p->node_k_valid) {
        rc = launch_amap_nodes(p->d_pix[0], p->nnodes, p->d_aflag, p->d_tile_off, p->d_node_k, st);
        if (rc != UPSP_OK) return rc;
        p->node_k_valid = true;
    }
    int S = 0;
    unsigned cp = 0;
    rc = streamed_buffers(p, npix, std::max(nframes, 1), st, &S, &cp);
    if (rc != UPSP_OK) return rc;
    if (nframes > S) return fail(UPSP_ERR_INVALID, "pixel series: more frames than one pass A group holds (<= 1024 per call)");
    p->prescan_frames = nullptr;
    if (nframes > 0 && !prescanned) rc = streamed_pass_a(p, d_frames, npix, 0, nframes, cp, st);
    if (rc != UPSP_OK) return rc;
    if (nframes > 0 && !prescanned) rc = streamed_repair(p, d_frames, npix, 0, nframes, cp, st);       // (a prescan repaired them)
    if (d_compact) *d_compact = p->d_compact;
    if (cpitch) *cpitch = cp;
    if (d_node_k) *d_node_k = p->d_node_k;
    if (d_nactive) *d_nactive = p->d_tile_off + tilemap_tiles(npix);
    return rc;
}

// Pass B as an operator: node-major series (and accumulators) of `nnodes` nodes over the frames of `nblocks` series buffers that
// follow each other in time (pipeline.h).  d_node_k [nnodes]: row of the buffers per node (< 0: no pixel -> 0); d_skipped: NaN rows.
}  // extern "C"
int upsp::rows_from_pixel_blocks(const SeriesBlock *blocks, int nblocks, const int32_t *d_node_k, const uint8_t *d_skipped,
                                 size_t nnodes, float *d_rows_t, int64_t ld, int64_t pad_to, double *d_sum, double *d_sumsq,
                                 hipStream_t st)
{
    if (!blocks || nblocks < 1 || !d_node_k || !d_rows_t || !d_sum || !d_sumsq || nnodes == 0 || nnodes >= ((size_t)1 << 31))
        return fail(UPSP_ERR_INVALID, "bad argument");
    std::vector<SeriesBlock> bl;
    std::vector<int64_t> start;
    int64_t total = 0;
    bool mult4 = true;
    for (int b = 0; b < nblocks; ++b) {
        if (blocks[b].nframes < 0 || (int64_t)blocks[b].cpitch < blocks[b].nframes || (blocks[b].nframes && !blocks[b].compact))
            return fail(UPSP_ERR_INVALID, "rows from pixel series: pitch smaller than the frame count");
        if (!blocks[b].nframes) continue;
        // (the row pass reads four frames per 8-byte load)
        if ((blocks[b].cpitch % 4) != 0 || (reinterpret_cast<size_t>(blocks[b].compact) & 7) != 0)
            return fail(UPSP_ERR_INVALID, "rows from pixel series: series rows must start on 8-byte boundaries (pitch a multiple of 4)");
        bl.push_back(blocks[b]);
        start.push_back(total);
        total += blocks[b].nframes;
        mult4 = mult4 && (blocks[b].nframes % 4) == 0;
    }
    start.push_back(total);
    if (ld < total || pad_to < total || pad_to > ld) return fail(UPSP_ERR_INVALID, "rows from pixel series: row pitch smaller than the frame count");
    PipelineGather g;
    g.ncams = 1;
    g.nnodes = nnodes;
    g.skipped = d_skipped;
    g.sum = d_sum;
    g.sumsq = d_sumsq;
    g.ld_t = ld;
    const int64_t G = group_frames_max();
    const int nb = (int)bl.size();
    // (measurement switches: UPSP_ROWS_LINE_CUT=0 one launch per block, round 4's form; UPSP_ROWS_WINDOWS=0 one launch per window)
    static const bool line_cut = [] { const char *e = getenv("UPSP_ROWS_LINE_CUT"); return !(e && *e == '0'); }();
    static const bool one_launch = [] { const char *e = getenv("UPSP_ROWS_WINDOWS"); return !(e && *e == '0'); }();
    std::vector<RowWindow> wins;
    int a = 0;
    for (int64_t c = 0; c < total;) {
        while (start[a + 1] <= c) ++a;
        const int64_t end_a = start[a + 1];
        int64_t hi = std::min(c + G, end_a);
        if (mult4 && line_cut) {
            // a launch ends on a 128-byte line of the output rows (32 columns) and takes the columns up to there from the next
            // block -- a 4000-byte row piece per source otherwise starts and ends inside a line, which the memory system writes
            // as partial lines (tools/probe/store_shapes.hip: 5.2-5.4 against 6.1-6.2 TB/s)
            const int64_t lim = std::min(c + G, a + 1 < nb ? start[a + 2] : end_a);
            const int64_t cut = lim == total ? total : lim / 32 * 32;
            if (cut > c) hi = cut;
        }
        RowWindow w;
        w.nframes = (int)(hi - c);
        w.nframes_a = (int)(std::min(hi, end_a) - c);
        w.nstore = hi == total ? (int)std::min<int64_t>(pad_to - c, G) : w.nframes;
        w.col = c;
        w.a = bl[a].compact + (c - start[a]);
        w.pitch_a = bl[a].cpitch;
        w.b = w.nframes_a < w.nframes ? bl[a + 1].compact : nullptr;
        w.pitch_b = w.b ? bl[a + 1].cpitch : 0;
        wins.push_back(w);
        c = hi;
    }
    // several windows: one launch takes a workgroup's rows through all of them (frames.hip: node_rows_windows_kernel)
    bool together = one_launch && mult4 && wins.size() > 1 && (ld % 4) == 0 && (reinterpret_cast<size_t>(d_rows_t) & 15) == 0;
    for (const RowWindow &w : wins) together = together && (w.nstore % 4) == 0;
    if (together) return launch_node_rows_windows(wins.data(), (int)wins.size(), d_node_k, d_skipped, nnodes, d_rows_t, ld, d_sum, d_sumsq, st);
    int rc = UPSP_OK;
    for (size_t i = 0; i < wins.size() && rc == UPSP_OK; ++i) {
        const RowWindow &w = wins[i];
        g.nframes = w.nframes;
        g.nstore = w.nstore;
        g.rows_t = d_rows_t + w.col;
        // (the caller's series: received from a peer, or kept from an earlier call -- not the pipeline's own pass A of a moment ago)
        if (w.b)
            rc = launch_node_rows(g, d_node_k, w.a, w.pitch_a, st, true, false, w.b, w.pitch_b, w.nframes_a);
        else
            rc = launch_node_rows(g, d_node_k, w.a, w.pitch_a, st, true);
    }
    return rc;
}

extern "C" {
int upsp_rows_from_pixel_series(const uint16_t *d_compact, uint32_t cpitch, const int32_t *d_node_k, const uint8_t *d_skipped,
                                size_t nnodes, int64_t nframes, float *d_rows_t, int64_t ld, double *d_sum, double *d_sumsq,
                                void *stream)
{
    if (!d_compact || !d_node_k || !d_rows_t || !d_sum || !d_sumsq || nnodes == 0 || nnodes >= ((size_t)1 << 31) || nframes < 0)
        return fail(UPSP_ERR_INVALID, "bad argument");
    if (ld < nframes || (int64_t)cpitch < nframes) return fail(UPSP_ERR_INVALID, "rows from pixel series: pitch smaller than the frame count");
    if (nframes == 0) return UPSP_OK;
    const upsp::SeriesBlock b = {d_compact, cpitch, nframes};
    return upsp::rows_from_pixel_blocks(&b, 1, d_node_k, d_skipped, nnodes, d_rows_t, ld, nframes, d_sum, d_sumsq, (hipStream_t)stream);
}

int upsp_rows_from_pixel_blocks(const uint16_t *const *d_compact, const uint32_t *cpitch, const int64_t *nframes, int nblocks,
                                const int32_t *d_node_k, const uint8_t *d_skipped, size_t nnodes, float *d_rows_t, int64_t ld,
                                int64_t pad_to, double *d_sum, double *d_sumsq, void *stream)
{
    if (!d_compact || !cpitch || !nframes || nblocks < 1) return fail(UPSP_ERR_INVALID, "bad argument");
    std::vector<upsp::SeriesBlock> b((size_t)nblocks);
    for (int i = 0; i < nblocks; ++i) b[(size_t)i] = {d_compact[i], cpitch[i], nframes[i]};
    return upsp::rows_from_pixel_blocks(b.data(), nblocks, d_node_k, d_skipped, nnodes, d_rows_t, ld, pad_to, d_sum, d_sumsq,
                                        (hipStream_t)stream);
}

static int process_impl(upsp_pipeline *p, uint16_t *const *d_frames, int nframes,
                        int64_t first_frame, float *d_rows, float *d_rows_t, uint16_t *d_rows_t16,
                        int64_t ld_t, int64_t col0, float *d_warps, void *stream)
{
    if (!p || !d_frames || nframes < 0) return fail(UPSP_ERR_INVALID, "bad argument");
    if (nframes == 0) return UPSP_OK;
    if (d_rows_t16) {
        // the stored values must be exact 16-bit integers: one camera, weight 1, the gather reading
        // u16 pixels (raw or warped frames; the patch and filter stages produce floats)
        if (p->ncams != 1 || p->d_weight[0] || p->opts.patch || p->opts.filter)
            return fail(UPSP_ERR_INVALID, "u16 series needs one camera, no weights, no patch / filter stage");
        // NaN has no u16 encoding: rows of nodes no camera sees must not be stored (the caller keeps them
        // out with a row map; rows of skipped nodes that the map does store are written as 0)
        if (!p->d_rowmap) return fail(UPSP_ERR_INVALID, "u16 series needs a row map (upsp_pipeline_set_row_map) that leaves out the skipped nodes");
        if (ld_t < col0 + nframes) return fail(UPSP_ERR_INVALID, "ld_t too small");
    }
    hipStream_t st = (hipStream_t)stream;
    // a prescan is consumed -- or dropped -- by the very next process call, whichever path that call takes
    const uint16_t *prescan_frames = p->prescan_frames;
    p->prescan_frames = nullptr;
    for (int c = 0; c < p->ncams; ++c) {
        if (!d_frames[c]) return fail(UPSP_ERR_INVALID, "null frame pointer");
        if (!p->has_proj[c]) return fail(UPSP_ERR_INVALID, "projection not set for a camera");
        if (p->opts.registration && !p->d_ref[c])
            return fail(UPSP_ERR_INVALID, "registration enabled but no reference frame set");
    }
    if (d_rows_t && ld_t < col0 + nframes) return fail(UPSP_ERR_INVALID, "ld_t too small");
    const bool need_f32 = p->opts.patch || p->opts.filter;
    const bool need_stage = need_f32 || p->opts.registration;
    const size_t npix = (size_t)p->width * p->height;
    const int B = p->batch;
    // identify_skipped_nodes over all cameras (projection.ipp:857-880), stream-ordered.  One camera on the streamed schedule with
    // a candidate-pixel map kept across the projection change: the sweep that gives the nodes their series rows writes the
    // flags as well (a node of a one-camera projection is skipped iff it has no pixel) -- one launch instead of two.
    const bool skipped_with_nodes = !p->skipped_valid && p->ncams == 1 && p->tilemap_valid && !p->node_k_valid && !p->d_weight[0] &&
                                    !need_stage && !d_rows && (d_rows_t || d_rows_t16) && !p->d_src && (npix % 2) == 0 && B == 64 &&
                                    p->nnodes < ((size_t)1 << 31) && p->opts.fused_scan != 2;
    if (!p->skipped_valid && !skipped_with_nodes) {
        int rc = upsp::launch_skipped(p->ncams, p->nnodes, p->d_pix, p->d_skipped, st);
        if (rc != UPSP_OK) return rc;
        p->skipped_valid = true;
    }
    if (need_stage) {
        int rc = upsp::frame_scratch_ensure(&p->scratch, p->ncams, B, p->height, p->width,
                                            p->opts.registration != 0, need_f32);
        if (rc != UPSP_OK) return rc;
    }
    int rc = UPSP_OK;
    // (Round 1 also had a two-stream schedule -- scan + repair on the caller's stream, the gathers on an internal one.  Measured on
    //  MI355X the two kernels compete for HBM and evict each other's working set from the Infinity Cache: 1.93 ms per 1000
    //  frames overlapped against 1.72 back to back on one stream.  Removed.)
    // Streamed two-pass schedule (frames.hip: scan_compact_kernel + node_rows_kernel) for the plain
    // path: one camera, no weights, u16 frames straight from the caller, node-major series only.
    // Measured on MI355X per 64-frame sub-batch of 1-Mpix frames: 25.8 + 32.4 us against 25.4 + 38.7 us
    // for scan kernel + gather kernel (f32 rows of all nodes), 25.5 + 21.6 against 25.4 + 32.2 us with
    // packed u16 rows (multi-GPU exchange); 4-Mpix frames (sub-batch larger than the Infinity Cache):
    // 1.3x.  It needs the active-pixel map once per projection (25 us).  Default when eligible.
    const int fused_mode = p->opts.fused_scan;
    const bool fused_ok = p->ncams == 1 && !p->d_weight[0] && !need_stage && !d_rows &&
                          (d_rows_t || d_rows_t16) && !p->d_src && (npix % 2) == 0 && B == 64 &&
                          p->nnodes < ((size_t)1 << 31);
    const bool fused = fused_ok && fused_mode != 2;
    if (fused) {
        if (!p->tilemap_valid) {
            rc = streamed_map(p, p->d_pix[0], npix, st);
            if (rc != UPSP_OK) return rc;
            p->node_k_valid = true;
            p->hint_active = false;
        }
        if (!p->node_k_valid) {     // candidate map kept across a projection change: only node -> series index is redone
            rc = launch_amap_nodes(p->d_pix[0], p->nnodes, p->d_aflag, p->d_tile_off, p->d_node_k, st,
                                   skipped_with_nodes ? p->d_skipped : nullptr);
            if (rc != UPSP_OK) return rc;
            p->node_k_valid = true;
            if (skipped_with_nodes) p->skipped_valid = true;
        }
        int S = 0;
        unsigned cp = 0;
        rc = streamed_buffers(p, npix, nframes, st, &S, &cp);
        if (rc != UPSP_OK) return rc;
        // accumulators untouched since the last reset: the first pass B writes them
        // (the flag is cleared once that pass B has been queued: a call that fails before leaves the reset pending)
        bool fresh = p->acc_unset;
        PipelineGather g;
        g.ncams = 1;
        g.npix = npix;
        g.nnodes = p->nnodes;
        g.skipped = p->d_skipped;
        g.rowmap = p->d_rowmap;
        g.sum = p->d_sum;
        g.sumsq = p->d_sumsq;
        g.pix[0] = p->d_pix[0];
        g.ld_t = ld_t;
        uint16_t *fr = d_frames[0];
        // pass A of exactly these frames may already have run (upsp_pipeline_prescan, e.g. on another stream
        // while the projection was being built; the caller orders the streams)
        const bool prescanned = prescan_frames == fr && p->prescan_n == nframes && nframes <= S &&
                                p->prescan_gen == p->map_gen;
        for (int s0 = 0; s0 < nframes && rc == UPSP_OK; s0 += S) {
            const int ns = std::min(S, nframes - s0);
            if (!prescanned) {
                rc = streamed_pass_a(p, fr, npix, s0, ns, cp, st);
                if (rc == UPSP_OK) rc = streamed_repair(p, fr, npix, s0, ns, cp, st);
            }
            g.nframes = ns;
            g.img[0] = fr + (size_t)s0 * npix;
            g.rows_t = d_rows_t ? d_rows_t + col0 + s0 : nullptr;
            g.rows_t16 = d_rows_t16 ? d_rows_t16 + col0 + s0 : nullptr;
            g.nstore = padded_store(p, col0 + s0, ns, ld_t, d_rows_t ? 32 : 64);
            if (rc == UPSP_OK) rc = launch_node_rows(g, p->d_node_k, p->d_compact, cp, st, false, fresh);
            if (rc == UPSP_OK && fresh) {
                p->acc_unset = false;
                fresh = false;
            }
        }
        return rc;
    }
    {
        int rcz = acc_zero_now(p, st, true);     // every other schedule adds to the accumulators
        if (rcz != UPSP_OK) return rcz;
    }
    // Registration as the last image stage, one camera, node-major series: per 64-frame sub-batch hot-pixel repair ->
    // ECC -> warp of the active pixels straight into the compact buffer; per <= 1024 frames ONE pass B (whole rows)
    // instead of a gather per sub-batch.
    const bool reg_streamed = p->ncams == 1 && !p->d_weight[0] && p->opts.registration && !need_f32 &&
                              !d_rows && (d_rows_t || d_rows_t16) && !p->d_src && (npix % 2) == 0 && npix < 0xFFFFFFFFull &&
                              p->nnodes < ((size_t)1 << 31) && fused_mode != 2;
    if (reg_streamed) {
        if (!p->tilemap_valid) {
            rc = streamed_map(p, p->d_pix[0], npix, st);
            if (rc != UPSP_OK) return rc;
            p->node_k_valid = true;
            p->hint_active = false;
        }
        if (!p->node_k_valid) {
            rc = launch_amap_nodes(p->d_pix[0], p->nnodes, p->d_aflag, p->d_tile_off, p->d_node_k, st);
            if (rc != UPSP_OK) return rc;
            p->node_k_valid = true;
        }
        const size_t nact_max = std::max<size_t>(std::min(p->nnodes, npix), 1);
        if (!p->d_pix_of_k) UPSP_HIP_CHECK(hipMalloc(&p->d_pix_of_k, sizeof(unsigned) * nact_max));
        if (p->pixk_gen != p->map_gen) {
            rc = launch_amap_pixels(p->d_aflag, p->d_tile_off, npix, p->d_pix_of_k, st);
            if (rc != UPSP_OK) return rc;
            p->pixk_gen = p->map_gen;
        }
        int S = 0;
        unsigned cp = 0;
        rc = streamed_buffers(p, npix, nframes, st, &S, &cp);
        if (rc != UPSP_OK) return rc;
        // Sub-batches of this path: 512 frames where the blurred f32 copies allow it (two buffers of RB frames: 4 GiB at
        // 1 Mpix, the cap; 256 frames until the end of round 5: configs[2] at 10 000 frames 156.1 / 158.5 / 160.4 k frames/s
        // at 256 / 384 / 512).  Every launch of the lock-step ECC loop -- sums, the one-lane-per-frame solve, pre-blur, repair,
        // warp -- is paid per sub-batch, the solve (16 us of latency, nothing beside it) and the launch tails most of all;
        // configs[2], ms per 1000 frames at 64 / 128 / 192 / 256 / 512 frames: 7.96 / 7.71 / 7.62 / 7.54 / 7.45.  The sums of a
        // frame do not depend on its neighbours (block count per frame fixed by the image), so the bits are the ones of
        // 64-frame sub-batches.  UPSP_REG_BATCH=n: another size (tests: 64).
        int RB = B;
        if (B == 64) {
            const char *e = std::getenv("UPSP_REG_BATCH");
            int want = e ? std::max(64, std::atoi(e) / 64 * 64) : 512;
            while (want > 64 && npix * sizeof(float) * 2 * (size_t)want > ((size_t)4 << 30)) want -= 64;
            RB = std::min(want, 1024);
        }
        rc = upsp::frame_scratch_ensure(&p->scratch, 1, RB, p->height, p->width, true, false);
        if (rc != UPSP_OK) return rc;
        PipelineGather g;
        g.ncams = 1;
        g.npix = npix;
        g.nnodes = p->nnodes;
        g.skipped = p->d_skipped;
        g.rowmap = p->d_rowmap;
        g.sum = p->d_sum;
        g.sumsq = p->d_sumsq;
        g.pix[0] = p->d_pix[0];
        g.ld_t = ld_t;
        WarpCompact wc;
        wc.pix_of_k = p->d_pix_of_k;
        wc.nact = p->d_tile_off + tilemap_tiles(npix);
        wc.max_active = nact_max;
        wc.compact = p->d_compact;
        wc.cpitch = cp;
        uint16_t *fr = d_frames[0];
        // sub-batches of the whole call, in order
        struct Sub { int f0, nb, s0; };
        std::vector<Sub> subs;
        for (int s0 = 0; s0 < nframes; s0 += S)
            for (int f0 = s0; f0 < std::min(s0 + S, nframes); f0 += RB) subs.push_back({f0, std::min(RB, std::min(s0 + S, nframes) - f0), s0});
        // One stream; sub-batch k + 1's hot-pixel repair and pre-blur (upsp::frame_scratch_preblur: the scan of fix_hot_pixels
        // rides on the blur, psp_process.cpp:1772) are enqueued while the host waits for sub-batch k's "frames still
        // iterating" (two blurred-frame buffers): the GPU has work during the read-back and nothing runs beside anything.
        // (Measured and rejected in round 3: the same kernels on a stream of their own beside the ECC sums -- same bits, 10.3
        // instead of 9.87 ms per 1000 frames: the second 256 MB of blurred frames pushes the first out of the Infinity Cache
        // between the identity and the general iteration.)
        upsp::HotRepair hot;
        if (p->opts.hot_enable) {
            rc = ensure_hot(p, RB);
            if (rc != UPSP_OK) return rc;
            hot.thresh = p->opts.hot_thresh;
            hot.min_change = p->opts.hot_min_change;
            hot.max_hot = p->opts.hot_max;
            hot.d_count = p->d_hot_count;
            hot.d_pos = p->d_hot_pos;
            hot.d_changes = p->d_changes;         // (streamed_buffers: sized for the frames of the call)
        }
        const float *blurred[2] = {nullptr, nullptr};
        std::vector<char> blur_done(subs.size(), 0);
        // the blurred template first: the pre-blur of the frames takes the identity iteration's sums with it
        rc = upsp::frame_scratch_template(p->scratch, 0, p->d_ref[0], p->height, p->width, st);
        if (rc != UPSP_OK) return rc;
        auto preblur = [&](size_t i) -> int {                  // repair + pre-blur of sub-batch i on the caller's stream
            if (blur_done[i]) return UPSP_OK;
            blur_done[i] = 1;
            return upsp::frame_scratch_preblur(p->scratch, (int)(i & 1), fr + (size_t)subs[i].f0 * npix, subs[i].nb, p->height, p->width,
                                               st, &blurred[i & 1], p->opts.hot_enable ? &hot : nullptr, /*fuse_cam=*/0);
        };
        for (size_t i = 0; i < subs.size() && rc == UPSP_OK; ++i) {
            const int f0 = subs[i].f0, nb = subs[i].nb, s0 = subs[i].s0;
            uint16_t *frames = fr + (size_t)f0 * npix;
            std::function<int()> next_blur;
            const std::function<int()> *while_waiting = nullptr;
            rc = preblur(i);                                // (already done while sub-batch i - 1 waited, normally)
            if (rc != UPSP_OK) break;
            const float *pre = blurred[i & 1];
            if (i + 1 < subs.size()) {
                next_blur = [&, i]() { return preblur(i + 1); };
                while_waiting = &next_blur;
            }
            wc.col0 = (unsigned)(f0 - s0);
            const void *img = nullptr;
            int is_f32 = 0;
            rc = upsp::run_frame_stages(p->scratch, 0, frames, nb, first_frame + f0, p->height, p->width, p->opts, p->d_ref[0],
                                        nullptr, d_warps ? d_warps + (size_t)f0 * 6 : nullptr,
                                        p->d_ecc_iters ? p->d_ecc_iters + f0 : nullptr, 1, nullptr, &wc, &img, &is_f32, st, pre, while_waiting);
            const bool last_of_group = i + 1 == subs.size() || subs[i + 1].s0 != s0;
            if (rc == UPSP_OK && last_of_group) {
                const int ns = std::min(S, nframes - s0);
                g.nframes = ns;
                g.img[0] = fr + (size_t)s0 * npix;
                g.rows_t = d_rows_t ? d_rows_t + col0 + s0 : nullptr;
                g.rows_t16 = d_rows_t16 ? d_rows_t16 + col0 + s0 : nullptr;
                g.nstore = padded_store(p, col0 + s0, ns, ld_t, d_rows_t ? 32 : 64);
                // (the group's series were written sub-batch by sub-batch over the whole registration of <= 1024 frames: most of them
                //  have left the Infinity Cache -- pass B 0.60 ms per 1000 frames here against 0.41 in the plain loop, r04)
                rc = launch_node_rows(g, p->d_node_k, p->d_compact, cp, st, /*cold_series=*/true);
            }
        }
        return rc;
    }
    // Several cameras (weights allowed): the same two passes with one active-pixel map and one whole-call compact
    // buffer per camera; pass B (node_rows_multi_kernel) sums the cameras in order with their weights.
    const bool multi_ok = p->ncams > 1 && !need_stage && !d_rows && d_rows_t && !d_rows_t16 && !p->d_src &&
                          (npix % 2) == 0 && B == 64 && p->nnodes < ((size_t)1 << 31);
    // default (fused_scan = 0): streamed from 192 frames per call on -- measured on the 5 M-triangle / 4-camera
    // shape (tools/scale_5m.py): 256 frame sets 1.68 ms against 2.20 ms for scan + gather, 64 frame sets 0.84
    // against 0.57 ms (short calls write 256-B row pieces either way and pay pass A on top)
    if (multi_ok && (fused_mode == 1 || (fused_mode == 0 && nframes >= 192))) {
        const size_t ntiles = tilemap_tiles(npix);
        // frames per group (pass A per camera, then ONE whole-row pass B): the per-camera compact buffers
        // (2 B x min(nodes, pixels) x frames) share the budget opts.compact_mb (default 2048 MiB per camera)
        const size_t budget = (size_t)(p->opts.compact_mb > 0 ? p->opts.compact_mb : 2048 * p->ncams) << 20;
        const size_t nact = std::max<size_t>(std::min(p->nnodes, npix), 1);
        int S = (int)std::min<size_t>((size_t)group_frames_max(), (budget / ((size_t)p->ncams * 2 * nact)) / 64 * 64);
        S = std::max(S, 64);
        const unsigned cp = (unsigned)((std::min(nframes, S) + 63) / 64 * 64);
        for (int c = 0; c < p->ncams; ++c) {
            if (nact * cp * 2 > p->m_compact_bytes[c]) {
                if (p->m_compact[c]) {
                    UPSP_HIP_CHECK(hipStreamSynchronize(st));
                    free_dev(p->m_compact[c]);
                    p->m_compact[c] = nullptr;
                    p->m_compact_bytes[c] = 0;
                }
                UPSP_HIP_CHECK(hipMalloc(&p->m_compact[c], nact * cp * 2));
                p->m_compact_bytes[c] = nact * cp * 2;
            }
            if (p->m_valid[c]) continue;
            if (!p->m_aflag[c]) {
                UPSP_HIP_CHECK(hipMalloc(&p->m_aflag[c], npix));
                UPSP_HIP_CHECK(hipMemset(p->m_aflag[c], 0, npix));
                UPSP_HIP_CHECK(hipStreamSynchronize(nullptr));      // (allocation time only)
            }
            if (!p->m_tile_off[c]) UPSP_HIP_CHECK(hipMalloc(&p->m_tile_off[c], sizeof(unsigned) * (ntiles + 1)));
            if (!p->m_tile_order[c]) UPSP_HIP_CHECK(hipMalloc(&p->m_tile_order[c], sizeof(unsigned) * 4 * (ntiles + 1)));
            if (!p->d_tile_cnt) UPSP_HIP_CHECK(hipMalloc(&p->d_tile_cnt, sizeof(unsigned) * ntiles));
            if (!p->m_node_k[c]) UPSP_HIP_CHECK(hipMalloc(&p->m_node_k[c], sizeof(int32_t) * p->nnodes));
            rc = launch_amap_build(p->d_pix[c], p->nnodes, npix, p->m_aflag[c], p->d_tile_cnt, p->m_tile_off[c],
                                   p->m_node_k[c], p->m_tile_order[c], st);
            if (rc != UPSP_OK) return rc;
            p->m_valid[c] = true;
        }
        PipelineGather g;
        g.ncams = p->ncams;
        g.npix = npix;
        g.nnodes = p->nnodes;
        g.skipped = p->d_skipped;
        g.rowmap = p->d_rowmap;
        g.sum = p->d_sum;
        g.sumsq = p->d_sumsq;
        g.ld_t = ld_t;
        for (int c = 0; c < p->ncams; ++c) {
            g.weight[c] = p->d_weight[c];
            g.pix[c] = p->d_pix[c];
        }
        // Hot pixels.  Without a row map the count rides on pass A (one read of every frame instead of two) and the
        // rare replaced pixels are re-projected through all cameras afterwards (launch_hot_fixup_multi); with a row
        // map the scan + repair stays a pass of its own per 64 frames, pass A following while they are in cache.
        const bool hot = p->opts.hot_enable != 0;
        const bool hot_fused = hot && !p->d_rowmap;
        if (hot_fused) {
            rc = ensure_hot(p, nframes * p->ncams);
            if (rc != UPSP_OK) return rc;
            const size_t words = hot_changes_words(nframes, p->opts.hot_max) * (size_t)p->ncams;
            if (words > p->changes_words) {
                if (p->d_changes) UPSP_HIP_CHECK(hipStreamSynchronize(st));
                free_dev(p->d_changes);
                p->d_changes = nullptr;
                p->changes_words = 0;
                UPSP_HIP_CHECK(hipMalloc(&p->d_changes, sizeof(unsigned) * words));
                UPSP_HIP_CHECK(hipMemsetAsync(p->d_changes, 0, sizeof(unsigned) * 4, st));      // (the change counters)
                p->changes_parity = 0;
                p->changes_words = words;
            }
            if (p->head_elems < npix * p->ncams) {
                if (p->d_head) UPSP_HIP_CHECK(hipStreamSynchronize(st));
                free_dev(p->d_head);
                p->d_head = nullptr;
                UPSP_HIP_CHECK(hipMalloc(&p->d_head, sizeof(int32_t) * npix * p->ncams));
                p->head_elems = npix * p->ncams;
                p->head_clean = false;
            }
            if (p->next_elems < p->nnodes * p->ncams) {
                if (p->d_next) UPSP_HIP_CHECK(hipStreamSynchronize(st));
                free_dev(p->d_next);
                p->d_next = nullptr;
                UPSP_HIP_CHECK(hipMalloc(&p->d_next, sizeof(int32_t) * p->nnodes * p->ncams));
                p->next_elems = p->nnodes * p->ncams;
            }
        }
        for (int s0 = 0; s0 < nframes && rc == UPSP_OK; s0 += S) {
            const int ns = std::min(S, nframes - s0);
            for (int c = 0; c < p->ncams && rc == UPSP_OK; ++c) {
                uint16_t *fr = d_frames[c];
                if (hot_fused) {
                    rc = launch_scan_compact(fr + (size_t)s0 * npix, npix, ns, true, p->opts.hot_thresh, p->opts.hot_max,
                                             p->m_aflag[c], p->m_tile_off[c], p->m_tile_order[c], p->m_compact[c], cp, 0,
                                             p->d_hot_count + (size_t)c * nframes + s0,
                                             p->d_hot_pos + ((size_t)c * nframes + s0) * 64, st);
                    continue;
                }
                for (int f0 = s0; f0 < s0 + ns && rc == UPSP_OK; f0 += B) {
                    const int nb = std::min(B, s0 + ns - f0);
                    if (hot) {
                        rc = ensure_hot(p, nb);
                        if (rc == UPSP_OK)
                            rc = launch_hot_fix(fr + (size_t)f0 * npix, nb, p->height, p->width, p->opts.hot_thresh,
                                                p->opts.hot_min_change, p->opts.hot_max, p->d_hot_count, p->d_hot_pos,
                                                nullptr, st);
                    }
                    if (rc == UPSP_OK)
                        rc = launch_scan_compact(fr + (size_t)f0 * npix, npix, nb, false, 0, 0, p->m_aflag[c],
                                                 p->m_tile_off[c], p->m_tile_order[c], p->m_compact[c], cp, f0 - s0,
                                                 nullptr, nullptr, st);
                }
            }
            g.nframes = ns;
            g.rows_t = d_rows_t + col0 + s0;
            // (no row padding here: measured on 4 cameras, 2.5 M nodes, 1000 frame sets the padded rows are SLOWER,
            //  2.91-2.97 ms against 2.65-2.70, both at 72 VGPRs -- the one-camera kernel gains 10 % from them)
            if (rc == UPSP_OK) rc = launch_node_rows_multi(g, p->m_node_k, p->m_compact, cp, st);
        }
        if (rc == UPSP_OK && hot_fused) {
            g.rows_t = d_rows_t + col0;
            rc = launch_hot_fixup_multi(g, d_frames, nframes, p->height, p->width, p->opts.hot_min_change, p->opts.hot_max,
                                        p->d_hot_count, p->d_hot_pos, p->d_changes, p->d_head, p->d_next, p->head_clean, st);
            p->head_clean = rc == UPSP_OK;
        }
        return rc;
    }
    int kbatch = 0;
    for (int f0 = 0; f0 < nframes && rc == UPSP_OK; f0 += B, ++kbatch) {
        const int nb = std::min(B, nframes - f0);
        PipelineGather g;
        g.ncams = p->ncams;
        g.npix = npix;
        g.nnodes = p->nnodes;
        g.nframes = nb;
        g.skipped = p->d_skipped;
        g.src = p->d_src;
        g.rowmap = p->d_rowmap;
        g.sum = p->d_sum;
        g.sumsq = p->d_sumsq;
        g.rows = d_rows ? d_rows + (size_t)f0 * p->nnodes : nullptr;
        g.rows_t = d_rows_t ? d_rows_t + col0 + f0 : nullptr;
        g.rows_t16 = d_rows_t16 ? d_rows_t16 + col0 + f0 : nullptr;
        g.ld_t = ld_t;
        for (int c = 0; c < p->ncams && rc == UPSP_OK; ++c) {
            uint16_t *frames = d_frames[c] + (size_t)f0 * npix;
            if (p->opts.hot_enable) {  // psp_process.cpp:1772
                rc = ensure_hot(p, nb);
                if (rc == UPSP_OK)
                    rc = launch_hot_fix(frames, nb, p->height, p->width, p->opts.hot_thresh,
                                        p->opts.hot_min_change, p->opts.hot_max, p->d_hot_count,
                                        p->d_hot_pos, nullptr, st);
                if (rc != UPSP_OK) break;
            }
            const void *img = frames;
            int is_f32 = 0;
            if (need_stage) {
                const unsigned *read_list = nullptr;
                if (p->opts.registration && !p->opts.patch && !p->opts.filter && npix < 0xFFFFFFFFull) {
                    if (!p->d_read_list[c]) UPSP_HIP_CHECK(hipMalloc(&p->d_read_list[c], sizeof(unsigned) * (npix + 1)));
                    if (!p->d_read_mask) UPSP_HIP_CHECK(hipMalloc(&p->d_read_mask, npix));
                    if (!p->read_list_valid[c]) {
                        rc = upsp::launch_pixel_list(p->d_pix[c], p->nnodes, p->d_read_mask, p->d_read_list[c], npix, st);
                        if (rc != UPSP_OK) break;
                        p->read_list_valid[c] = true;
                    }
                    read_list = p->d_read_list[c];
                }
                rc = upsp::run_frame_stages(p->scratch, c, frames, nb, first_frame + f0,
                                            p->height, p->width, p->opts, p->d_ref[c],
                                            p->patches[c],
                                            d_warps ? d_warps + ((size_t)f0 * p->ncams) * 6 : nullptr,
                                            p->d_ecc_iters ? p->d_ecc_iters + (size_t)f0 * p->ncams : nullptr,
                                            p->ncams, read_list, nullptr, &img, &is_f32, st);
                if (rc != UPSP_OK) break;
            }
            g.img[c] = img;
            g.is_f32[c] = is_f32;
            g.pix[c] = p->d_pix[c];
            g.weight[c] = p->d_weight[c];
        }
        if (rc == UPSP_OK) rc = launch_gather(g, st);
    }
    return rc;
}

int upsp_pipeline_process(upsp_pipeline *p, uint16_t *const *d_frames, int nframes,
                          int64_t first_frame, float *d_rows, float *d_rows_t, int64_t ld_t,
                          int64_t col0, float *d_warps, void *stream)
{
    return process_impl(p, d_frames, nframes, first_frame, d_rows, d_rows_t, nullptr, ld_t, col0,
                        d_warps, stream);
}

int upsp_pipeline_process_u16(upsp_pipeline *p, uint16_t *const *d_frames, int nframes,
                              int64_t first_frame, uint16_t *d_series_u16, int64_t ld_t,
                              int64_t col0, float *d_warps, void *stream)
{
    if (!d_series_u16) return fail(UPSP_ERR_INVALID, "null series buffer");
    return process_impl(p, d_frames, nframes, first_frame, nullptr, nullptr, d_series_u16, ld_t,
                        col0, d_warps, stream);
}


// ---- one step of a frame loop whose projection is rebuilt per batch (model motion) --------------------------------------------
// The whole schedule of such a step behind one call (the reference's frame loop is one function, cpp/exec/psp_process.cpp:1743-1851,
// and create_projection_mat a call in front of it, :1591-1640).  Two streams, the caller's (`stream`) and a high-priority side stream
// the pipeline owns:
//
//   side    [end of step s-2]  candidate pixels -> active-pixel map (second set of map arrays)
//           [repair of step s-1]  frames_hook (the frames may be rewritten here)            -> ev_map
//           projection build straight into the pipeline's buffer
//           [end of step s-1]  finals of step s-1, projection hand-over, node -> row sweep + skipped flags, tail_hook -> ev_side
//   stream  [ev_map]  pass A (+ hot-pixel repair) on the candidate map, beside the build     -> ev_repaired
//           [ev_side] pass B                                                                -> end of step s
//
// so the caller's stream carries pass A, the repair and pass B and nothing else, and the host runs a step ahead of the device.
// Same results as the plain sequence upsp_projection_build -> upsp_pipeline_set_projection -> upsp_pipeline_reset ->
// upsp_pipeline_process -> upsp_pipeline_finalize on one stream (tests/test_frames_gpu.py::test_pipeline_step_*).
static int step_setup(upsp_pipeline *p)
{
    upsp_pipeline::Step &s = p->step;
    if (s.side) return UPSP_OK;
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess)
        UPSP_HIP_CHECK(hipStreamCreateWithPriority(&s.side, hipStreamNonBlocking, greatest));
    else
        UPSP_HIP_CHECK(hipStreamCreateWithFlags(&s.side, hipStreamNonBlocking));
    for (hipEvent_t *e : {&s.ev_map[0], &s.ev_map[1], &s.ev_side[0], &s.ev_side[1], &s.ev_repaired[0], &s.ev_repaired[1], &s.ev_end[0],
                          &s.ev_end[1], &s.ev_end[2]})
        UPSP_HIP_CHECK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    // (the two build streams FIRST and nothing that is not used: the runtime deals its few hardware queues out in the order streams are
    //  created -- with two more normal-priority streams created in front of the second build stream the two builds shared a queue
    //  and the step went from 0.77 to 1.13 ms, round 6.  The pass-A stream of the measurement switch is created when first used.)
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess)
        UPSP_HIP_CHECK(hipStreamCreateWithPriority(&s.side2, hipStreamNonBlocking, greatest));
    else
        UPSP_HIP_CHECK(hipStreamCreateWithFlags(&s.side2, hipStreamNonBlocking));
    UPSP_HIP_CHECK(hipMalloc(&s.d_cand, sizeof(int32_t) * p->nnodes));
    UPSP_HIP_CHECK(hipMalloc(&s.d_uv, sizeof(float) * 2 * p->nnodes));
    UPSP_HIP_CHECK(hipMalloc(&s.d_cand2, sizeof(int32_t) * p->nnodes));
    UPSP_HIP_CHECK(hipMalloc(&s.d_uv2, sizeof(float) * 2 * p->nnodes));
    return UPSP_OK;
}

int upsp_pipeline_step(upsp_pipeline *p, const upsp_step_args *a, void *stream)
{
    if (!p || !a || !a->bvh || !a->cam || !a->d_nodes || !a->d_normals || !a->d_tri_nodes || !a->d_frames || a->nframes <= 0)
        return fail(UPSP_ERR_INVALID, "step: bad argument");
    const size_t npix = (size_t)p->width * p->height;
    if (p->ncams != 1 || p->d_weight[0] || p->opts.registration || p->opts.patch || p->opts.filter || p->d_src || (npix % 2) != 0 ||
        p->batch != 64 || p->opts.fused_scan == 2 || p->nnodes >= ((size_t)1 << 31) || a->nframes > streamed_group_frames(p, npix))
        return fail(UPSP_ERR_INVALID, "step: plain one-camera path with the streamed schedule, <= 1024 frames per step");
    int rc = step_setup(p);
    if (rc != UPSP_OK) return rc;
    upsp_pipeline::Step &s = p->step;
    const uint64_t n = s.count;
    // TWO BUILDS IN FLIGHT (round 6).  A step's side block is one long chain -- candidate pixels, map, the primary pass (0.14 ms
    // alone, 0.41 ms beside the passes), the retry lists, witness test, residual traversal, hand-over: 0.82 ms in the kernel trace
    // of a 0.83-ms step, i.e. with ONE side stream the chain of step s + 1 starts when the chain of step s ends and pass B waits
    // ~90 us for it every step.  Odd steps therefore build on a second high-priority stream and a second handle over the same
    // tree (upsp_bvh_share: the tree is shared, the query scratch is not): the chain of step s + 1 starts as soon as the host
    // has issued it and runs beside the chain of step s.  Every cross-step dependency is already an event wait below (map arrays
    // of step s - 2, repair of step s - 1 in front of the frames hook, end of step s - 1 in front of the finals and the skipped
    // flags); the two projection buffers and candidate / uv scratch exist per parity.  UPSP_STEP_ONE_SIDE=1: one stream (A/B).
    static const bool two_sides = [] { const char *e = std::getenv("UPSP_STEP_ONE_SIDE"); return !(e && *e == '1'); }();
    const bool odd = two_sides && (n & 1u);
    hipStream_t main = (hipStream_t)stream, side = odd ? s.side2 : s.side;
    upsp_bvh *bvh = a->bvh;
    if (odd) {
        // (a handle over another tree, or over this one before its adjacency was rebuilt -- upsp_bvh_set_tri_nodes frees and
        //  reallocates the arrays a shared handle only points at)
        if (s.bvh2 && (s.bvh2_of != a->bvh || s.bvh2->d_adj_off != a->bvh->d_adj_off || s.bvh2->d_slot_path != a->bvh->d_slot_path ||
                       s.bvh2->d_wide != a->bvh->d_wide || s.bvh2->adj_src != a->bvh->adj_src)) {
            UPSP_HIP_CHECK(hipStreamSynchronize(s.side2));
            upsp_bvh_destroy(s.bvh2);
            s.bvh2 = nullptr;
        }
        if (!s.bvh2) {
            rc = upsp_bvh_share(a->bvh, &s.bvh2);
            if (rc != UPSP_OK) return rc;
            s.bvh2_of = a->bvh;
        }
        bvh = s.bvh2;
    }
    int32_t *d_cand = odd ? s.d_cand2 : s.d_cand;
    float *d_uv = odd ? s.d_uv2 : s.d_uv;
    // ---- side stream ----
    if (n < 2) {                                         // (a side stream's first block starts behind whatever the caller has queued)
        UPSP_HIP_CHECK(hipEventRecord(s.ev_end[2], main));
        UPSP_HIP_CHECK(hipStreamWaitEvent(side, s.ev_end[2], 0));
    }
    // the set of map arrays the new map is built in was last read by the launches of the step before the previous one
    if (n >= 2) UPSP_HIP_CHECK(hipStreamWaitEvent(side, s.ev_end[(n - 2) % 3], 0));
    rc = upsp_projection_candidate_pixels_oblique(a->cam, a->d_nodes, a->d_normals, a->d_datanode, p->nnodes, a->oblique_thresh, d_cand, side);
    if (rc != UPSP_OK) return rc;
    rc = upsp_pipeline_set_active_hint(p, d_cand, side);
    if (rc != UPSP_OK) return rc;
    if (a->frames_hook) {
        // behind the map (which runs beside the previous step's pass A): in front of it the whole side block -- and with it this
        // step's pass B -- would wait for that repair.  The wait also times the build: its ray casting starts with the previous
        // step's pass B, a write stream it disturbs little (0.343 ms with or without company), not beside the whole of pass A.
        // (Round 6, measured and not kept: the hook on a stream of its own, so that the build starts as soon as the map is built --
        //  the primary pass 0.35 -> 0.25 ms, but pass A beside it 0.349 -> 0.399 ms and the step 0.758 -> 0.81-0.83 ms; with the
        //  ray casting gated on the repair again through an event: 0.79-0.83.)
        if (n >= 1) UPSP_HIP_CHECK(hipStreamWaitEvent(side, s.ev_repaired[(n - 1) % 2], 0));
        a->frames_hook(a->frames_user, side);
    }
    UPSP_HIP_CHECK(hipEventRecord(s.ev_map[n % 2], side));
    int32_t *target = nullptr;
    rc = upsp_pipeline_projection_target(p, 0, &target);
    if (rc != UPSP_OK) return rc;
    rc = upsp_projection_build(bvh, a->cam, a->d_nodes, a->d_normals, a->d_datanode, a->d_tri_nodes, p->nnodes, a->oblique_thresh,
                               target, d_uv, nullptr, nullptr, side);
    if (rc != UPSP_OK) return rc;
    // the previous step's pass B: its sums (finals), and it read the skipped flags the sweep below rewrites
    if (n >= 1) UPSP_HIP_CHECK(hipStreamWaitEvent(side, s.ev_end[(n - 1) % 3], 0));
    if (s.finals_due) {
        rc = upsp_pipeline_finalize(p, s.ntotal, s.d_avg, s.d_rms, side);
        if (rc != UPSP_OK) return rc;
        s.finals_due = false;
    }
    rc = upsp_pipeline_set_projection_async(p, 0, target, nullptr, side);
    if (rc == UPSP_OK) rc = upsp_pipeline_prepare_rows(p, side);
    if (rc != UPSP_OK) return rc;
    if (a->tail_hook) a->tail_hook(a->tail_user, side);
    UPSP_HIP_CHECK(hipEventRecord(s.ev_side[n % 2], side));
    // ---- pass A + repair ----
    // On a stream of their own, into the OTHER of two compact buffers: pass A of this step then runs beside pass B of the step
    // before (a read stream beside a write stream) instead of behind it -- the ~30 us between the end of one and the start of the
    // other, and both kernels' ramps, disappear from the step.  What it reads -- the frames (the hook's, ordered by ev_map), this
    // step's map -- nobody writes; the buffer it writes was last read by pass B two steps ago (the side stream waited for that).
    // MEASURED AND NOT THE DEFAULT (round 6, three alternations in one call): 1.09-1.10 ms per step against 0.82-0.83 with pass A
    // behind pass B on the caller's stream -- a read stream and a write stream side by side each take twice as long (pass A 0.35 ->
    // 0.71 ms, pass B 0.35 -> 0.81 ms, the primary pass beside them 0.36 -> 0.70 ms): the memory system moves a mix of reads and
    // writes no faster than the copy probe does (5.3 TB/s) and the two kernels also take each other's wave slots.
    // UPSP_STEP_SCAN_STREAM=1 (measurement switch) turns it on.
    static const bool scan_beside = [] { const char *e = std::getenv("UPSP_STEP_SCAN_STREAM"); return e && *e == '1'; }();
    if (scan_beside && !s.scan) UPSP_HIP_CHECK(hipStreamCreateWithFlags(&s.scan, hipStreamNonBlocking));
    hipStream_t scan = scan_beside ? s.scan : main;
    if (scan_beside) {
        std::swap(p->d_compact, p->d_compact_alt);
        std::swap(p->compact_bytes, p->compact_bytes_alt);
    }
    UPSP_HIP_CHECK(hipStreamWaitEvent(scan, s.ev_map[n % 2], 0));
    rc = upsp_pipeline_prescan(p, a->d_frames, a->nframes, scan);      // (in two launches when upsp_pipeline_set_scan_split says so)
    if (rc != UPSP_OK) return rc;
    UPSP_HIP_CHECK(hipEventRecord(s.ev_repaired[n % 2], scan));
    // ---- the caller's stream: pass B ----
    if (scan_beside) UPSP_HIP_CHECK(hipStreamWaitEvent(main, s.ev_repaired[n % 2], 0));
    UPSP_HIP_CHECK(hipStreamWaitEvent(main, s.ev_side[n % 2], 0));
    if (a->d_rows_t) {
        rc = upsp_pipeline_reset_deferred(p);            // (this step's sums start from zero: pass B writes them)
        if (rc != UPSP_OK) return rc;
        uint16_t *frames[1] = {a->d_frames};
        rc = upsp_pipeline_process(p, frames, a->nframes, a->first_frame, nullptr, a->d_rows_t, a->ld_t, a->col0, nullptr, main);
        if (rc != UPSP_OK) return rc;
        s.finals_due = a->d_avg || a->d_rms;
        s.d_avg = a->d_avg;
        s.d_rms = a->d_rms;
        s.ntotal = a->nframes_total ? a->nframes_total : (uint64_t)a->nframes;
    }
    UPSP_HIP_CHECK(hipEventRecord(s.ev_end[n % 3], main));
    ++s.count;
    return UPSP_OK;
}

int upsp_pipeline_step_mark_end(upsp_pipeline *p, void *stream)
{
    if (!p || !p->step.side || p->step.count == 0) return fail(UPSP_ERR_INVALID, "step: no step issued");
    UPSP_HIP_CHECK(hipEventRecord(p->step.ev_end[(p->step.count - 1) % 3], (hipStream_t)stream));
    return UPSP_OK;
}

int upsp_pipeline_step_finish(upsp_pipeline *p, void *stream)
{
    if (!p) return fail(UPSP_ERR_INVALID, "null pipeline");
    upsp_pipeline::Step &s = p->step;
    if (!s.side) return UPSP_OK;
    hipStream_t main = (hipStream_t)stream;
    if (s.finals_due) {          // the last step's finals: behind its pass B on the caller's stream
        int rc = upsp_pipeline_finalize(p, s.ntotal, s.d_avg, s.d_rms, main);
        if (rc != UPSP_OK) return rc;
        s.finals_due = false;
    }
    // whatever the pipeline's own streams still hold (nothing a finished step needs) is ordered in front of the caller's next launch
    UPSP_HIP_CHECK(hipEventRecord(s.ev_map[0], s.side));
    UPSP_HIP_CHECK(hipStreamWaitEvent(main, s.ev_map[0], 0));
    if (s.scan) {
        UPSP_HIP_CHECK(hipEventRecord(s.ev_map[1], s.scan));
        UPSP_HIP_CHECK(hipStreamWaitEvent(main, s.ev_map[1], 0));
    }
    UPSP_HIP_CHECK(hipEventRecord(s.ev_side[0], s.side2));
    UPSP_HIP_CHECK(hipStreamWaitEvent(main, s.ev_side[0], 0));
    return UPSP_OK;
}

}  // extern "C"
