// Shared declarations of the per-frame pipeline (frames.hip, registration.hip,
// imageops.hip, pipeline.hip).
#ifndef UPSP_PIPELINE_H
#define UPSP_PIPELINE_H

#include <functional>
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace upsp {

constexpr int kMaxCams = 16;

struct PipelineGather {
    int ncams = 0;
    size_t npix = 0;      // pixels per frame
    size_t nnodes = 0;
    int nframes = 0;
    int nstore = 0;       // pass B (whole rows): columns [nframes, nstore) of a stored row are padding that may be written (0 = none)
    const void *img[kMaxCams] = {nullptr};  // first frame of the batch (u16 or f32)
    int is_f32[kMaxCams] = {0};
    const int32_t *pix[kMaxCams] = {nullptr};
    const float *weight[kMaxCams] = {nullptr};
    const uint8_t *skipped = nullptr;
    const int32_t *rowmap = nullptr;  // node -> row of rows_t (packed series), < 0 = row not stored
    const int32_t *src = nullptr;  // overlap source map (adjust_solution): stored value of node n = sol[src[n]]
    float *rows = nullptr;  // [nframes][nnodes], may be null
    float *rows_t = nullptr;  // node-major: rows_t[n*ld_t + f], may be null
    uint16_t *rows_t16 = nullptr;  // the node-major series as u16 instead (pitch ld_t elements), may be null
    int64_t ld_t = 0;
    double *sum = nullptr, *sumsq = nullptr;
};

size_t hot_counter_words(int nframes);   // size of the counter buffer launch_hot_fix needs (zeroed)
int launch_hot_fix(uint16_t *d_frames, int nframes, int rows, int cols, int thresh,
                   int min_change, int max_hot, unsigned *d_count, unsigned *d_pos,
                   int32_t *d_status, hipStream_t st, const unsigned *d_only = nullptr);
int launch_gather(const PipelineGather &g, hipStream_t st);
// streamed scan + projection (one camera, no weights, u16 frames, no image stage): frames.hip
size_t tilemap_tiles(size_t npix);
// d_order: optional, 4 * (tiles + 1) words; its first `tiles` words = the order pass A visits the tiles in
int launch_amap_build(const int32_t *d_pix, size_t nnodes, size_t npix, uint8_t *d_flag, unsigned *d_cnt,
                      unsigned *d_off, int32_t *d_node_k, unsigned *d_order, hipStream_t st);
// d_skipped_out (optional): identify_skipped_nodes of a one-camera projection in the same sweep
int launch_amap_nodes(const int32_t *d_pix, size_t nnodes, const uint8_t *d_flag, const unsigned *d_off,
                      int32_t *d_node_k, hipStream_t st, uint8_t *d_skipped_out = nullptr);
int group_frames_max();    // frames per pass B (whole rows) of the one-camera streamed schedule
int launch_scan_compact(uint16_t *d_frames, size_t npix, int nframes, bool hot, int thresh, int max_hot,
                        const uint8_t *d_flag, const unsigned *d_off, const unsigned *d_order, uint16_t *d_compact,
                        unsigned cpitch, int col, unsigned *d_count, unsigned *d_pos, hipStream_t st, bool split = false);
// cold_series: the compact series were not written a moment ago (they come from HBM, not from the Infinity Cache): every
// sweep's series are requested up front
// fresh_acc: the accumulators hold nothing yet (reset, untouched since): they are written, not added to
// d_compact_b / cpitch_b / nframes_a: frames [nframes_a, g.nframes) of the launch read a second series buffer (same rows; its column
// 0 = frame nframes_a; nframes_a a multiple of 4)
int launch_node_rows(const PipelineGather &g, const int32_t *d_node_k, const uint16_t *d_compact, unsigned cpitch,
                     hipStream_t st, bool cold_series = false, bool fresh_acc = false, const uint16_t *d_compact_b = nullptr,
                     unsigned cpitch_b = 0, int nframes_a = 0);
// Pass B over the series of the same pixel rows held in `nblocks` buffers, one per run of consecutive frames (blocks received from
// the peers of an exchange: [pixel row][frames of the source], pitch = cpitch[b] u16): node-major rows d_rows_t [nnodes][ld] over
// all frames, accumulators.  The launches are cut at 128-byte lines of the output rows (32 columns), not at the block boundaries,
// when every block holds a multiple of 4 frames; pad_to (>= the frame total, <= ld): columns the last launch may write (padding).
struct SeriesBlock {
    const uint16_t *compact;
    unsigned cpitch;
    int64_t nframes;
};
// one window of the output rows (columns col .. col + nframes): frames [0, nframes_a) from series buffer a, the rest from b
struct RowWindow {
    const uint16_t *a, *b;
    unsigned pitch_a, pitch_b;
    int nframes, nframes_a, nstore;
    long long col;
};
constexpr int kMaxRowWindows = 16;
struct RowWindows {
    int n;
    RowWindow w[kMaxRowWindows];
};
int launch_node_rows_windows(const RowWindow *w, int nwin, const int32_t *d_node_k, const uint8_t *d_skipped, size_t nnodes,
                             float *d_rows_t, int64_t ld, double *d_sum, double *d_sumsq, hipStream_t st);
int rows_from_pixel_blocks(const SeriesBlock *blocks, int nblocks, const int32_t *d_node_k, const uint8_t *d_skipped, size_t nnodes,
                           float *d_rows_t, int64_t ld, int64_t pad_to, double *d_sum, double *d_sumsq, hipStream_t st);
int launch_node_rows_multi(const PipelineGather &g, const int32_t *const *d_node_k, const uint16_t *const *d_compact,
                           unsigned cpitch, hipStream_t st);
size_t hot_changes_words(int nframes, int max_hot);   // size of d_changes for the repair launches
// parity: which of the buffer's two change counters this call uses (flipped by the call; the other one is zeroed for the next call;
// both zero after the allocation)
int launch_hot_repair_compact(uint16_t *d_frames, size_t npix, int nframes, int rows, int cols, int min_change, int max_hot,
                              unsigned *d_count, const unsigned *d_pos, unsigned *d_changes, int *parity, const uint8_t *d_flag,
                              const unsigned *d_tile_off, uint16_t *d_compact, unsigned cpitch, hipStream_t st);
int launch_hot_repair_list(uint16_t *d_frames, size_t npix, int nframes, int rows, int cols, int min_change, int max_hot,
                           unsigned *d_count, const unsigned *d_pos, unsigned *d_changes, hipStream_t st);
constexpr int kHotPositions = 64;   // hot-pixel positions recorded per frame (d_pos: kHotPositions words per frame)
int launch_hot_fixup_multi(const PipelineGather &g, uint16_t *const *d_frames, int nframes, int rows, int cols,
                           int min_change, int max_hot, unsigned *d_count, const unsigned *d_pos,
                           unsigned *d_changes, int32_t *d_head, int32_t *d_next, bool head_clean, hipStream_t st);
int launch_skipped(int ncams, size_t nnodes, const int32_t *const *d_pix, uint8_t *d_skipped,
                   hipStream_t st);
int launch_finals(const double *sum, const double *sumsq, size_t nnodes, uint64_t nframes,
                  float *avg, float *rms, hipStream_t st);

// ---- image stages (imageops.hip / registration.hip) --------------------------
struct PatchTables;   // device copies of PatchClusters' boundary / interior lists
struct FrameScratch;  // per-sub-batch working images (warped u16, f32)

int patch_tables_create(int rows, int cols, int nclusters, const int32_t *h_b_off,
                        const int32_t *h_bx, const int32_t *h_by, const int32_t *h_i_off,
                        const int32_t *h_ix, const int32_t *h_iy, PatchTables **out);
void patch_tables_free(PatchTables *t);
int frame_scratch_ensure(FrameScratch **s, int ncams, int batch, int rows, int cols,
                         bool need_warp, bool need_f32);
void frame_scratch_free(FrameScratch *s);
void frame_scratch_new_reference(FrameScratch *s, int cam);  // ECC template changed
void frame_scratch_ecc_stats(const FrameScratch *s, unsigned long long *frame_iters, unsigned long long *frames);

}  // namespace upsp

struct upsp_pipeline_opts;
namespace upsp {
// target of the warp when registration is the last image stage of the streamed schedule: the compact
// [active pixel][frame] buffer (column col0 on, pitch cpitch); pix_of_k = pixel of every compact row,
// *nact = rows in use (device), max_active = bound of it (grid size)
struct WarpCompact {
    const unsigned *pix_of_k = nullptr, *nact = nullptr;
    size_t max_active = 0;
    uint16_t *compact = nullptr;
    unsigned cpitch = 0, col0 = 0;
};
int launch_amap_pixels(const uint8_t *d_flag, const unsigned *d_tile_off, size_t npix, unsigned *d_pix_of_k, hipStream_t st);
// register -> patch -> filter for `nb` frames of camera `cam` (psp_process.cpp:1776-1807).
// *img_out / *is_f32_out: image the gather reads.  d_warps: [nb][ncams][6] or null; d_iters: [nb][ncams] or null.
// d_read_list (may be null): [1 + rows*cols] words, count and then the pixels the gather will read.  When registration
// is the last image stage (no patch, no filter) only those pixels of the warped frames are produced.
int run_frame_stages(FrameScratch *s, int cam, const uint16_t *d_frames, int nb,
                     int64_t first_frame, int rows, int cols, const upsp_pipeline_opts &opts,
                     const float *d_ref, const PatchTables *patches, float *d_warps, int32_t *d_iters, int ncams,
                     const unsigned *d_read_list, const WarpCompact *wc, const void **img_out, int *is_f32_out,
                     hipStream_t st, const float *preblurred = nullptr, const std::function<int()> *while_waiting = nullptr);
// fix_hot_pixels (cpp/utils/cv_extras.cpp:230-275) scratch of a sub-batch: one counter + kHotPositions positions per frame,
// the change list of hot_changes_words(nb, max_hot) words
struct HotRepair {
    int thresh = 0, min_change = 0, max_hot = 0;
    unsigned *d_count = nullptr, *d_pos = nullptr, *d_changes = nullptr;
};
// fix_hot_pixels + the ECC's 5 x 5 pre-blur of nb frames into blurred-frame buffer `slot` (0 / 1) of the scratch; the frames are
// repaired in place (hot == null: no repair)
// fuse_cam >= 0: the blurred template of that camera is ready (frame_scratch_template) -- the blur also takes the sums of the ECC's
// identity iteration (ecc.hip: ecc_blur_ident_kernel), which run_frame_stages then skips
int frame_scratch_preblur(FrameScratch *s, int slot, uint16_t *d_frames, int nb, int rows, int cols, hipStream_t st,
                          const float **out, const HotRepair *hot, int fuse_cam = -1);
// the blurred ECC template of camera `cam` for the reference image d_ref (once per reference image)
int frame_scratch_template(FrameScratch *s, int cam, const float *d_ref, int rows, int cols, hipStream_t st);
// d_list[0] = number of distinct pixels with a node, d_list[1..] = those pixels (any order); d_mask: npix bytes of scratch
int launch_pixel_list(const int32_t *d_pix, size_t nnodes, uint8_t *d_mask, unsigned *d_list, size_t npix, hipStream_t st);
}  // namespace upsp
#endif
