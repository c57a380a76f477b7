/*
 * upsp_gpu.h -- C ABI of libupsp_gpu.so, the MI355X (gfx950) engine for the
 * ray-cast + per-frame projection hot path of nasa/upsp-processing.
 *
 * This is the drop-in boundary: every entry point names the reference interface
 * (file:line under the upstream repository) it replaces.  Plain pointers and
 * sizes only; no torch / OpenCV / Eigen / Imath types.  Pointers named d_* are
 * DEVICE pointers (HBM of the current HIP device), h_* are host pointers.
 * `stream` is a hipStream_t passed as void* (NULL = default stream).
 *
 * All functions return UPSP_OK (0) or a negative upsp_status; none of them
 * exits the process (the reference's DIE()/ASSERT() macros call exit(),
 * cpp/include/utils/pspError.h:47-98).  upsp_last_error() returns a message
 * for the calling thread.
 */
#ifndef UPSP_GPU_H
#define UPSP_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum upsp_status {
    UPSP_OK = 0,
    UPSP_ERR_INVALID = -1,   /* bad argument (null pointer, size mismatch ...) */
    UPSP_ERR_EMPTY = -2,     /* BVH without primitives (pspRT.cpp:320-323, 362-365) */
    UPSP_ERR_DEPTH = -3,     /* BVH deeper than the 64-entry traversal stack (pspRT.cpp:374) */
    UPSP_ERR_HIP = -4,       /* HIP runtime error, see upsp_last_error() */
    UPSP_ERR_NO_DEVICE = -5, /* no gfx950 device visible */
    UPSP_ERR_DIVERGED = -6,  /* ECC registration failed (cv::findTransformECC would throw) */
    UPSP_ERR_INTERNAL = -7   /* a BVH walk ran past its round cap (2 x nodes: broken tree); the reference DIEs on a
                                broken BVH (pspRT.cpp:362-365) -- here an error code, never a wedged device */
} upsp_status;

const char *upsp_last_error(void);
/* Library / device identification: writes e.g. "gfx950" into arch (>= 32 bytes). */
int upsp_device_info(int *n_devices, char *arch, int *n_cus);
int upsp_version(void);

/* ======================================================================== *
 *  1.  BVH ray-triangle intersector   (reference: cpp/raycast/pspRT.cpp)
 * ======================================================================== */

typedef struct upsp_bvh upsp_bvh;

/* Replaces rt::CreateBVH(raw, stride) (cpp/include/utils/pspRT.h:131-135),
 * rt::CreateTriangleMesh + rt::BVH::BVH (cpp/raycast/pspRT.cpp:206-222,313-344)
 * and createBVH() of psp_process (cpp/exec/psp_process.cpp:44-53).
 * h_tris9: triangle soup, 9 floats per triangle (x,y,z of the 3 vertices).
 * The SAH tree has the reference's topology and leaf order; it is laid out for
 * the GPU (two child boxes per 64-byte node, 48-byte triangle records) and
 * uploaded to the current device. */
int upsp_bvh_create(const float *h_tris9, size_t ntris, upsp_bvh **out);
void upsp_bvh_destroy(upsp_bvh *bvh);
/* A second handle on the same tree (and on the adjacency upsp_bvh_set_tri_nodes has set by then) with query scratch of its own:
 * queries on the two handles may run at the same time on different streams -- create_projection_mat of the cameras of one model
 * (cpp/exec/psp_process.cpp:1586-1660) does not depend on the other cameras.  The owner must outlive its shares. */
int upsp_bvh_share(const upsp_bvh *src, upsp_bvh **out);

typedef struct upsp_bvh_info {
    uint64_t ntris;
    uint32_t n_ref_nodes;   /* LinearNode count of the reference layout (leaves + interior) */
    uint32_t n_gpu_nodes;   /* 64-byte interior nodes on the device */
    uint32_t depth;         /* tree height = worst-case traversal stack depth */
    uint32_t max_leaf;      /* largest leaf (triangles) */
    float bounds_min[3];    /* rt::BVH::bounds() (pspRT.cpp:353-357) */
    float bounds_max[3];
    uint64_t device_bytes;  /* HBM footprint of nodes + triangles */
    double build_seconds;
} upsp_bvh_info;
int upsp_bvh_get_info(const upsp_bvh *bvh, upsp_bvh_info *info);

/* createBVH(model, triNodes) also returns the triangle -> node ids (cpp/exec/psp_process.cpp:44-53,
 * TriModel_::extract_tris cpp/lib/TriModel.ipp:261-299).  Handing them to the BVH (device array,
 * 3 ints per input triangle, node ids in [0, nnodes)) lets upsp_projection_build bound every
 * camera -> node ray by the node's own triangles: same visibility verdicts, far fewer node
 * visits.  Optional; used only by projection builds that pass the same d_tri_nodes pointer and
 * nnodes.  Call again when the array's contents change; (NULL, 0) clears. */
int upsp_bvh_set_tri_nodes(upsp_bvh *bvh, const int32_t *d_tri_nodes, size_t nnodes, void *stream);

/* Per-ray outputs, structure of arrays; any pointer may be NULL (not written). */
typedef struct upsp_hits {
    uint8_t *hit;   /* [n]   return value of rt::BVH::intersect: any triangle hit with t>=0 */
    float *t;       /* [n]   Hit.t   (FLT_MAX when no hit, pspRT.cpp:21) */
    int32_t *prim;  /* [n]   Hit.primID = index into the input soup (-1 when no hit) */
    float *uvw;     /* [n*3] Hit.u, v, w */
    float *pos;     /* [n*3] Hit.pos = o + t*d (pspRT.cpp:179) */
    float *nrm;     /* [n*3] Hit.nrm (un-normalised, pspRT.cpp:182-190) */
} upsp_hits;

/* Replaces n calls of  rt::Ray(o,d) + rt::Hit() + rt::BVH::intersect(ray,&hit)
 * (cpp/raycast/pspRT.cpp:45-69, 359-431; pybind: cpp/pybind11/raycast.cpp:16-35).
 * Closest hit (min t, first found in the reference's traversal order on ties).
 * org_stride = 3: one origin per ray; org_stride = 0: a single shared origin. */
int upsp_bvh_intersect(const upsp_bvh *bvh, const float *d_org, int org_stride,
                       const float *d_dir, size_t n, const upsp_hits *d_out, void *stream);
/* Same with host buffers (staged through HBM); convenience for the bindings. */
int upsp_bvh_intersect_host(const upsp_bvh *bvh, const float *h_org, int org_stride,
                            const float *h_dir, size_t n, const upsp_hits *h_out);

/* Occlusion query: only the boolean return value of rt::BVH::intersect, i.e.
 * what VisibilityChecker.does_intersect consumes
 * (python/upsp/cam_cal_utils/visibility.py:392-420).  Stops at the first hit. */
int upsp_bvh_occluded(const upsp_bvh *bvh, const float *d_org, int org_stride,
                      const float *d_dir, size_t n, uint8_t *d_hit, void *stream);
int upsp_bvh_occluded_host(const upsp_bvh *bvh, const float *h_org, int org_stride,
                           const float *h_dir, size_t n, uint8_t *h_hit);

/* Traversal statistics of the last intersect/occluded/projection launch on this
 * BVH (summed over rays): interior nodes fetched, triangles tested, rays cast.
 * Collected only after upsp_bvh_enable_stats(bvh, 1). */
int upsp_bvh_enable_stats(upsp_bvh *bvh, int on);

/* Waits for `stream` and reports whether any walk launched on this BVH since the last check ran past its
 * round cap (UPSP_ERR_INTERNAL; a ray visits every node of a well-formed tree at most once, so a walk that needs
 * more than 2 x nodes rounds is running on a broken tree and ends with an error flag instead of looping:
 * the reference DIEs on a broken BVH, cpp/raycast/pspRT.cpp:362-365).  The host-buffer queries, the statistics and
 * upsp_projection_fetch_counts / upsp_projection_build(h_nrays != NULL) check by themselves; callers that queue
 * device-side launches without synchronising call this at their own synchronisation point. */
int upsp_bvh_check(upsp_bvh *bvh, void *stream);
int upsp_bvh_last_stats(const upsp_bvh *bvh, uint64_t *nodes, uint64_t *tris, uint64_t *rays);
/* The same launch's box tests of the one-lane traversal: boxes the first filter (a plain slab test with a per-ray error bound)
 * saw, and how many of them it left undecided -- those go through the mirrored reciprocal filter and, inside its margin, through
 * Imath's divisions (rt::BVH::intersect's box test, cpp/raycast/pspRT.cpp:376-383).  Statistics on. */
int upsp_bvh_last_filter_stats(const upsp_bvh *bvh, uint64_t *boxes, uint64_t *undecided);

/* ======================================================================== *
 *  2.  Projection build   (reference: create_projection_mat,
 *                           cpp/exec/psp_process.cpp:167-355)
 * ======================================================================== */

/* Inputs of cv::projectPoints as used by CameraCal::map_point_to_image
 * (cpp/lib/CameraCal.ipp:218-231); get_cam_center (cpp/lib/CameraCal.cpp:192-203)
 * is derived from R, t. */
typedef struct upsp_camera {
    double K[9];    /* cameraMatrix, row-major */
    double dist[5]; /* k1,k2,p1,p2,k3 */
    double R[9];    /* model->camera rotation, row-major */
    double t[3];
    int32_t width, height;
} upsp_camera;

/* One camera.  For every node: project to the image, in-frame test, primary ray
 * camera->node, visibility by the hit triangle's node ids, <=6 jittered retries,
 * oblique-angle test; result is the <=1-nnz-per-row projection matrix as two
 * dense arrays:
 *   d_pix[n]    = round(v)*W + round(u) of the nearest pixel, -1 when the node
 *                 has no entry in the reference's sparse matrix
 *   d_uv[2n..]  = (u/W, v/H) (camNN-uv file, psp_process.cpp:1615-1620), 0 otherwise
 *   d_nodecount = [H*W] u8 saturating nodes-per-pixel image (may be NULL)
 * d_datanode (may be NULL) = Model::is_datanode mask; d_tri_nodes = extract_tris()
 * triNodes [3*ntris] (cpp/lib/TriModel.ipp:261-299).
 * oblique_thresh = deg2rad(180 - oblique_angle) as float (psp_process.cpp:1602).
 * h_nrays (may be NULL): the number of rays the reference casts for this camera (1 primary ray per in-frame node
 * + its sequential retries); the call then waits for the device.
 * Order of the tests: the reference applies the oblique test last (psp_process.cpp:298-306), to the nodes its rays
 * have seen; it depends on the node normal and the primary direction only, and a node that fails it has no entry
 * whatever the rays said.  With h_nrays == NULL the test is applied BEFORE any ray is cast and the nodes that fail
 * cast none (identical d_pix / d_uv / d_nodecount; on a closed body two thirds of the in-frame nodes drop out,
 * the back-facing ones that would go through all six retries first).  With h_nrays != NULL (or statistics
 * enabled) the rays are cast as the reference casts them, because counting them means casting them.
 * UPSP_OBLIQUE_CULL=0 in the environment keeps the reference's order in every call. */
int upsp_projection_build(upsp_bvh *bvh, const upsp_camera *cam, const float *d_nodes,
                          const float *d_normals, const uint8_t *d_datanode,
                          const int32_t *d_tri_nodes, size_t nnodes, float oblique_thresh,
                          int32_t *d_pix, float *d_uv, uint8_t *d_nodecount,
                          uint64_t *h_nrays, void *stream);

/* Work actually done by the last upsp_projection_build on this BVH: primary rays cast
 * and nodes that went through the jittered retries (all six retry rays of such a node
 * are cast as independent rays; *h_nrays above reports the REFERENCE's sequential
 * count, 1 + index of the first successful retry). */
int upsp_projection_last_counts(const upsp_bvh *bvh, uint64_t *primary_rays,
                                uint64_t *retry_nodes);

/* upsp_projection_build with h_nrays == NULL does not wait for the device.  The counters of the
 * most recent build stay in the BVH's work buffer; this call waits for `stream` and reads them
 * (ray count, primary rays, nodes that went through the retries -- of the nodes that passed the oblique test,
 * see upsp_projection_build: NOT the reference's count). */
int upsp_projection_fetch_counts(upsp_bvh *bvh, uint64_t *nrays, uint64_t *primary_rays,
                                 uint64_t *retry_nodes, void *stream);

/* Step 1 of create_projection_mat alone (psp_process.cpp:241-252 + :319): d_pix[n] = the pixel node n is stored
 * at IF a ray sees it (same arithmetic as upsp_projection_build: cv::projectPoints in double, in-frame test on the
 * cvRound-ed point, std::round for the stored pixel), -1 for nodes outside the frame / the data-node mask.  Every
 * projection of this camera is a subset: input of upsp_pipeline_set_active_hint. */
int upsp_projection_candidate_pixels(const upsp_camera *cam, const float *d_nodes, const uint8_t *d_datanode,
                                     size_t nnodes, int32_t *d_pix, void *stream);
/* The same restricted to the nodes that pass the oblique test of psp_process.cpp:298-306 (node normal against the camera ->
 * node direction; `oblique_thresh` as for upsp_projection_build): a node that fails it has no entry whatever its rays say, so
 * this is still a superset of the projection built with the same threshold -- on a closed body a third of the pixels of the
 * plain candidate set (the back-facing nodes drop out), i.e. a pass A that stores a third of the series. */
int upsp_projection_candidate_pixels_oblique(const upsp_camera *cam, const float *d_nodes, const float *d_normals,
                                             const uint8_t *d_datanode, size_t nnodes, float oblique_thresh, int32_t *d_pix,
                                             void *stream);

/* adjust_projection_for_weights with BestView (mode 0) / AverageViews (mode 1)
 * (cpp/lib/projection.ipp:911-1078, 227-268).  d_pix, d_weight: [ncams*nnodes];
 * d_weight is scaled in place (caller initialises it to 1).  h_centers: ncams*3
 * doubles (CameraCal::get_cam_center). */
int upsp_projection_weights(int ncams, size_t nnodes, const int32_t *d_pix, float *d_weight,
                            const float *d_nodes, const float *d_normals,
                            const double *h_centers, int mode, void *stream);

/* identify_skipped_nodes (cpp/lib/projection.ipp:857-880): d_skipped[n] = 1 iff
 * no camera has an entry for node n.  *h_count (may be NULL) = number skipped. */
int upsp_projection_skipped(int ncams, size_t nnodes, const int32_t *d_pix,
                            uint8_t *d_skipped, uint64_t *h_count, void *stream);

/* camera centre C = -R^T t  (cpp/lib/CameraCal.cpp:192-203) */
int upsp_camera_center(const upsp_camera *cam, double h_center[3]);
/* cv::projectPoints on n points (host convenience, same arithmetic as the kernel) */
int upsp_project_points_host(const upsp_camera *cam, const float *h_xyz, size_t n, float *h_uv);

/* ======================================================================== *
 *  3.  Per-frame pipeline   (reference: frame loop of psp_process phase 1,
 *                            cpp/exec/psp_process.cpp:1743-1851)
 * ======================================================================== */

typedef struct upsp_pipeline upsp_pipeline;

typedef struct upsp_pipeline_opts {
    /* fix_hot_pixels defaults, cpp/include/utils/cv_extras.h:154-155 */
    int32_t hot_enable, hot_thresh, hot_min_change, hot_max;
    /* registration: 0 none, 1 pixel (ECC affine) ; upsp_inputs.h:22-38 */
    int32_t registration;
    int32_t ecc_max_iters;  /* 50   (psp_process.cpp:1781) */
    double ecc_eps;         /* 1e-3 (psp_process.cpp:1782) */
    int32_t interp;         /* 1 linear, 0 nearest (PixelInterpolationType) */
    /* filter: 0 none, 1 gaussian, 2 box ; filter_size odd (psp_process.cpp:1296) */
    int32_t filter, filter_size;
    /* polynomial target patcher on/off (TargetPatchType) */
    int32_t patch;
    /* frame-loop schedule of the plain path (one camera, no weights, no image stage, node-major series):
     * 0 / 1 = streamed two-pass schedule (pass A: scan + compact pixel series of up to 1024 frames per
     * launch, pass B: every node's row piece written once, whole), 2 = scan kernel + gather kernel per
     * 64 frames (relies on the sub-batch staying in the Infinity Cache).  Same results.  Several cameras
     * (weights allowed): 1 = streamed (one compact buffer per camera, pass B sums the cameras in order),
     * 2 = scan + gather, 0 = streamed for calls of >= 192 frames (rows bit-identical, accumulators to 1e-12). */
    int32_t fused_scan;
    /* streamed schedule: budget in MiB for the compact pixel-series buffer (2 B x min(nodes, pixels) x
     * frames of a group, allocated on first use); 0 = default (2048).  It bounds the frames per pass A /
     * pass B pair (64 .. 1024). */
    int32_t compact_mb;
    int32_t reserved[3];
} upsp_pipeline_opts;

void upsp_pipeline_default_opts(upsp_pipeline_opts *o);

int upsp_pipeline_create(int ncams, int width, int height, size_t nnodes,
                         const upsp_pipeline_opts *opts, upsp_pipeline **out);
void upsp_pipeline_destroy(upsp_pipeline *p);

/* Projection of camera `cam` (copied): d_pix [nnodes] int32, d_weight [nnodes] f32
 * (NULL = all ones).  Equivalent of elems.projs[c] (psp_process.cpp:1591-1640). */
int upsp_pipeline_set_projection(upsp_pipeline *p, int cam, const int32_t *d_pix,
                                 const float *d_weight);
/* The same with the copies ordered on `stream` instead of blocking the host: the projection build
 * (upsp_projection_build) and the frame loop can then be queued back to back. */
int upsp_pipeline_set_projection_async(upsp_pipeline *p, int cam, const int32_t *d_pix,
                                       const float *d_weight, void *stream);
/* A projection that is rebuilt while the frame loop runs (model motion, docs/sphinx/known-issues.rst:18-30; bench.py rebuilds it
 * every step) can be built straight into the pipeline: *d_pix = a buffer of nnodes int32 owned by the pipeline that none of its
 * queued launches reads; pass it as upsp_projection_build's d_pix and then to upsp_pipeline_set_projection[_async], which takes
 * it over without the device copy (a 2-MB blit that waits 0.2 ms for its turn beside pass A) and hands out the previous
 * projection's buffer at the next call -- so the caller orders a build behind the launches of the process call BEFORE the
 * last one (they may read that buffer).  Replaces nothing in the reference: create_projection_mat returns its matrix by value
 * (cpp/exec/psp_process.cpp:167-355). */
int upsp_pipeline_projection_target(upsp_pipeline *p, int cam, int32_t **d_pix);
/* The projection of camera `cam` the pipeline currently holds (elems.projs[c]): *d_pix = its nnodes int32, pipeline-owned, valid
 * until the next upsp_pipeline_set_projection* / upsp_pipeline_step of that camera is two calls old (the buffers are used in turn).
 * After upsp_pipeline_step: the projection that step built (stream-ordered on the step's stream). */
int upsp_pipeline_projection(upsp_pipeline *p, int cam, const int32_t **d_pix);
/* fix_hot_pixels (cpp/utils/cv_extras.cpp:230-275, called at cpp/exec/psp_process.cpp:1772) does not
 * depend on the projection: upsp_pipeline_fix_hot_pixels queues the scan + repair of `nframes`
 * resident frames (in place, pipeline's thresholds) on `stream` -- e.g. a second stream while the
 * projection is still being built -- and upsp_pipeline_set_hot_enable(p, 0) makes the following
 * upsp_pipeline_process calls skip their own scan of those frames (1 switches it back on).  The
 * caller orders the streams (event between the scan and the first process call). */
int upsp_pipeline_fix_hot_pixels(upsp_pipeline *p, uint16_t *d_frames, int nframes, void *stream);
int upsp_pipeline_set_hot_enable(upsp_pipeline *p, int enable);
/* Pass A of the streamed schedule (hot-pixel count + transposed series of the pixels nodes read) depends on
 * the frames and on WHICH pixels are read, not on the visibility verdicts: with a candidate set -- the pixel every
 * in-frame node would be stored at, upsp_projection_candidate_pixels, a superset of any projection of that camera --
 * it can run while create_projection_mat is still casting rays (fix_hot_pixels, psp_process.cpp:1772, and the ray
 * cast, :167-355, are independent in the reference too):
 *   upsp_pipeline_set_active_hint(p, d_candidates, s2)   map of the candidate pixels (kept across projection changes
 *                                                         until cleared with NULL)
 *   upsp_pipeline_prescan(p, d_frames, n, s2)             pass A of n <= 1024 frames on stream s2 and their hot-pixel
 *                                                         repair (in place, like the process call does it)
 *   upsp_pipeline_set_projection_async(...) ; upsp_pipeline_process(p, &d_frames, n, ...)   -- after s2's work
 *                                                         (caller's event): pass B only
 * A node whose final pixel is missing from the candidate set is served from the frames directly (correct, slow).
 * The map arrays exist twice and every call with a candidate set builds into the pair the launches already queued do not
 * read: the call for the NEXT frame batch may be issued (on another stream) while pass B / the fix-up of the current one are
 * still queued or running; the pair it takes was last read by the launches of the call before the previous one -- a caller
 * whose streams can drift that far apart orders that with an event.
 * One camera, plain path. */
int upsp_pipeline_set_active_hint(upsp_pipeline *p, const int32_t *d_pix_candidates, void *stream);
int upsp_pipeline_prescan(upsp_pipeline *p, uint16_t *d_frames, int nframes, void *stream);
/* What the streamed frame loop derives from a NEW projection before its pass B -- every node's row in the compact pixel series
 * (from the active-pixel map) and, unless upsp_pipeline_set_skipped gave them, the flags of identify_skipped_nodes
 * (cpp/lib/projection.ipp:857-880) -- queued on `stream` now instead of inside the next upsp_pipeline_process call: on the
 * stream that built the projection, behind upsp_pipeline_set_projection_async, it runs beside pass A instead of between
 * pass A and pass B.  The caller orders `stream` behind the launches of the previous process call (they read the skipped
 * flags) and the next process call behind `stream`.  One camera, plain path; elsewhere the call does nothing. */
int upsp_pipeline_prepare_rows(upsp_pipeline *p, void *stream);
/* The two tables upsp_pipeline_prepare_rows (or the last process / pixel-series call) derived, for a caller that hands them on on the
 * same stream -- upsp_exchange_set_pixels behind the projection build instead of in the frame loop's stream: *d_node_k [nnodes] = row of
 * the compact pixel series per node (-1: no pixel, -2: pixel outside a candidate map), *d_skipped [nnodes] = identify_skipped_nodes'
 * flags.  Pipeline-owned, valid until the next projection / map change; UPSP_ERR_INVALID when they have not been derived. */
int upsp_pipeline_row_tables(upsp_pipeline *p, const int32_t **d_node_k, const uint8_t **d_skipped);
/* Nodes set to NaN in every row (psp_process.cpp:1822-1825); NULL = derive from
 * the projections with identify_skipped_nodes. */
int upsp_pipeline_set_skipped(upsp_pipeline *p, const uint8_t *d_skipped);

/* P3D zone overlaps: model.adjust_solution(sol) (cpp/lib/P3DModel.ipp:143-157) runs on every frame
 * AFTER the accumulators took the node's own value and BEFORE the row is stored
 * (cpp/exec/psp_process.cpp:1827-1839).  d_src [nnodes] int32: src[n] = node whose value node n
 * holds after the copy loop (n itself for ordinary nodes).  NULL switches it off. */
int upsp_pipeline_set_overlap_source(upsp_pipeline *pipe, const int32_t *d_src);

/* Packed time series: d_rowmap [nnodes] int32 gives the row of d_rows_t a node's series goes to;
 * < 0 = the row is not stored (used by the multi-GPU exchange, which does not send the all-NaN
 * rows of nodes no camera sees; the reference's global_transpose, cpp/exec/psp_process.cpp:707-771,
 * moves them).  Accumulators and frame-major rows are unaffected.  NULL = identity. */
int upsp_pipeline_set_row_map(upsp_pipeline *pipe, const int32_t *d_rowmap);
/* The same with the copy ordered on `stream` (no host block between the projection build and the frame loop). */
int upsp_pipeline_set_row_map_async(upsp_pipeline *pipe, const int32_t *d_rowmap, void *stream);

/* Row padding.  A node-major series buffer whose pitch ld_t is a multiple of 128 bytes usually ends every row with columns nobody
 * reads (engine.series_ld rounds 1000 frames up to 1024).  on != 0 declares: the columns between the LAST frame of a row and its
 * pitch hold no data of the caller's, and a whole-row pass B whose frames end within 512 bytes of the pitch (the end of the row: a
 * pitch is a row length rounded up by less than that) may write up to the next 128-byte boundary (0, or NaN in the row of a node
 * no camera sees).  A 4000-byte row piece then ends with a whole 128-byte line instead of a
 * quarter of one: 0.42 -> 0.37 ms per 1000 frames of the bench model.  Only the end of a row is ever padded -- a call that
 * fills a column window elsewhere in a wider matrix (chunks in any order, on any stream, live data to its right) stores its own
 * columns only.  Used by the one-camera whole-row passes (plain frames, registration as the last image stage, f32 or u16
 * series); the several-camera row pass ignores it (measured slower with it).  Off by default: intensity_transpose
 * (cpp/exec/psp_process.cpp:2027-2032) has no padding. */
int upsp_pipeline_set_row_padding(upsp_pipeline *p, int on);
/* Pass A of the one-camera streamed loop in two launches -- the tiles nobody reads as one-wave workgroups without LDS (16-byte loads),
 * then the active tiles -- for a frame loop that shares the device with other kernels (a projection build on a stream of its own:
 * bench.py's step 0.799 -> 0.773 ms).  Pass A alone on the device is faster in one launch (0.314 against 0.336 ms per 1000 frames of
 * 1 Mpix): off by default.  Same results either way.  Takes effect when the hot-pixel scan is on and the frame has a multiple of 128
 * pixels; ignored otherwise.  (A scheduling hint of this engine; fix_hot_pixels / project_frame have no counterpart.) */
int upsp_pipeline_set_scan_split(upsp_pipeline *p, int on);

/* ---- one step of a frame loop whose projection is rebuilt per batch of frames --------------------------------------------------
 * Model motion (docs/sphinx/known-issues.rst:18-30): the ray cast of create_projection_mat (cpp/exec/psp_process.cpp:167-355,
 * called once per run at :1591-1640) has to be repeated while the frame loop (:1743-1851) runs.  upsp_pipeline_step queues ONE such
 * step -- candidate pixels, active-pixel map, projection build, hand-over, pass A + hot-pixel repair, pass B, the previous step's
 * finals -- with the build on a high-priority stream the pipeline owns and only pass A, the repair and pass B on the caller's
 * `stream`; every ordering event between the two lives inside the library.  The host returns at once and may issue the next step
 * (the device then works on two steps at a time: build s + 1 beside pass B of step s).  One camera, plain path (no weights, no
 * image stage), nframes <= upsp_pipeline_series_frames_max.
 *   d_rows_t == NULL   no pass B and no reset: the step ends with the repaired pixel series in the pipeline's compact buffer
 *                      (upsp_pipeline_pixel_series on the same frames then returns it without another pass A): the multi-GPU
 *                      frame loop, whose pass B runs on the owner of a node; that caller marks the end of ITS step with
 *                      upsp_pipeline_step_mark_end (the point after which the side stream may rewrite the skipped flags).
 *   d_avg / d_rms      finals of THIS step's sums over nframes_total frames (0: nframes), written while the NEXT step's build
 *                      runs, or by upsp_pipeline_step_finish; NULL: none.
 *   frames_hook        called (on the host, inside the call) with a stream of the pipeline at the point where the previous step no
 *                      longer reads or repairs d_frames and this step has not yet scanned them: the caller queues whatever refills
 *                      the frames there (unpack of the next batch; bench.py puts the hot pixels back).  Keep it short: the step's
 *                      projection build is queued behind it on the same stream (which is also what starts the ray casting with
 *                      the previous step's pass B rather than beside its pass A).
 *   tail_hook          called with the side stream behind the node -> row sweep of the NEW projection and behind the end of the
 *                      previous step: a caller's own per-step work that needs upsp_pipeline_row_tables or the complete sums of
 *                      the step before (the multi-GPU loop: all-reduce + finals, the exchange's pixel table).
 * upsp_pipeline_step_finish(p, stream): the finals the last step left, and `stream` ordered behind the side stream. */
typedef void (*upsp_step_hook)(void *user, void *stream);
typedef struct upsp_step_args {
    upsp_bvh *bvh;
    const upsp_camera *cam;
    const float *d_nodes, *d_normals;     /* [3 nnodes] each */
    const uint8_t *d_datanode;            /* may be NULL */
    const int32_t *d_tri_nodes;           /* [3 ntris], the buffer given to upsp_bvh_set_tri_nodes */
    float oblique_thresh;                 /* as for upsp_projection_build */
    int32_t nframes;
    uint16_t *d_frames;                   /* [nframes][H][W], repaired in place */
    int64_t first_frame;
    float *d_rows_t;                      /* node-major series, may be NULL (see above) */
    int64_t ld_t, col0;
    float *d_avg, *d_rms;                 /* may be NULL */
    uint64_t nframes_total;
    upsp_step_hook frames_hook;
    void *frames_user;
    upsp_step_hook tail_hook;
    void *tail_user;
} upsp_step_args;
int upsp_pipeline_step(upsp_pipeline *p, const upsp_step_args *args, void *stream);
int upsp_pipeline_step_mark_end(upsp_pipeline *p, void *stream);
int upsp_pipeline_step_finish(upsp_pipeline *p, void *stream);

/* Receiving side of the packed exchange: block d_src [nrows][ncols] f32 (contiguous) is copied
 * to rows d_rowidx[r] (int64) of d_dst (row pitch ld floats; add the column offset to d_dst). */
int upsp_scatter_rows_f32(const float *d_src, size_t nrows, int ncols, const int64_t *d_rowidx,
                          float *d_dst, long long ld, void *stream);
/* The same for a block that travelled as u16 (upsp_pipeline_process_u16): values are widened to f32. */
int upsp_scatter_rows_u16(const uint16_t *d_src, size_t nrows, int ncols, const int64_t *d_rowidx,
                          float *d_dst, long long ld, void *stream);
/* The rows that do NOT travel (nodes no camera sees, psp_process.cpp:1821-1825: NaN in every frame): columns [0, ncols) of
 * rows d_rowidx[r] of d_dst <- value.  Part of every exchange, like the scatter of the rows that do travel. */
int upsp_fill_rows_f32(float value, size_t nrows, int ncols, const int64_t *d_rowidx, float *d_dst, long long ld,
                       void *stream);
/* ECC template of camera `cam` = first frame as f32 (elems.first_frames[c],
 * psp_process.cpp:2057-2058). */
int upsp_pipeline_set_reference(upsp_pipeline *p, int cam, const float *d_ref32f);
/* Patch tables of camera `cam` (PatchClusters members bounds_x/y, internal_x/y,
 * cpp/include/patches.h:76-110): CSR-style offsets [nclusters+1] into the
 * boundary / interior pixel lists (host pointers, copied). */
int upsp_pipeline_set_patches(upsp_pipeline *p, int cam, int nclusters,
                              const int32_t *h_b_off, const int32_t *h_bx, const int32_t *h_by,
                              const int32_t *h_i_off, const int32_t *h_ix, const int32_t *h_iy);

/* Process `nframes` consecutive frames of every camera.
 *   d_frames[c]      u16 frames of camera c, [nframes][H][W] contiguous.  Hot-pixel
 *                    correction is applied IN PLACE like the reference (:1772).
 *   first_frame      global index f of the first frame (frame 0 is never
 *                    registered in the loop, psp_process.cpp:1777)
 *   d_rows           [nframes][nnodes] f32 = intensity_buf rows (may be NULL)
 *   d_rows_t, ld_t   optional transposed output: d_rows_t[n*ld_t + col0 + f]
 *                    (= intensity_transpose layout, psp_process.cpp:2027-2032)
 *   d_warps          optional [nframes][ncams][6] f32 ECC warp matrices
 * The double accumulators sum / sumsq (psp_process.cpp:1827-1831) held by the
 * pipeline are updated. */
int upsp_pipeline_process(upsp_pipeline *p, uint16_t *const *d_frames, int nframes,
                          int64_t first_frame, float *d_rows, float *d_rows_t, int64_t ld_t,
                          int64_t col0, float *d_warps, void *stream);

/* upsp_pipeline_process with the node-major time series stored as u16 (d_series_u16[n*ld_t + col0
 * + f], pitch in elements) and no other output: the wire format of the multi-GPU time-series
 * exchange (global_transpose, cpp/exec/psp_process.cpp:707-771), half the bytes of f32.  Lossless
 * and accepted only when every stored value is an exact 16-bit integer: one camera, no weight
 * vector, no patch / filter stage (raw or registered u16 frames) -- UPSP_ERR_INVALID otherwise.
 * NaN has no u16 encoding, so a row map that leaves out the nodes no camera sees is REQUIRED
 * (upsp_pipeline_set_row_map; UPSP_ERR_INVALID without one -- a row of a skipped node that the map
 * does store is written as 0); upsp_scatter_rows_u16 widens the received blocks.  The frames are
 * repaired in place like in upsp_pipeline_process. */
int upsp_pipeline_process_u16(upsp_pipeline *p, uint16_t *const *d_frames, int nframes,
                              int64_t first_frame, uint16_t *d_series_u16, int64_t ld_t,
                              int64_t col0, float *d_warps, void *stream);

/* Statistics of the registration stage since the pipeline was created: ECC iterations summed over
 * the registered frames (cv::findTransformECC's loop count, cpp/lib/registration.cpp:64) and the
 * number of frames that went through it. */
int upsp_pipeline_ecc_stats(upsp_pipeline *p, uint64_t *frame_iterations, uint64_t *frames);
/* Per-frame iteration counts of the registration stage (the value cv::findTransformECC's loop ends with,
 * cpp/lib/registration.cpp:64): d_iters int32 [nframes][ncams] of every following upsp_pipeline_process
 * call is filled stream-ordered (frame 0 of a run, never registered: 0); NULL switches it off. */
int upsp_pipeline_set_ecc_iterations_out(upsp_pipeline *p, int32_t *d_iters);

/* Accumulator access (device pointers to nnodes doubles each), used for the
 * cross-GPU sum that replaces MPI_Reduce (psp_process.cpp:1866-1872). */
int upsp_pipeline_accumulators(upsp_pipeline *p, double **d_sum, double **d_sumsq);
/* The same for a caller that will use them on `stream`: a pending upsp_pipeline_reset_deferred is carried out there
 * (hipMemsetAsync) instead of after a wait for the whole device. */
int upsp_pipeline_accumulators_async(upsp_pipeline *p, double **d_sum, double **d_sumsq, void *stream);
/* Zeroes the two accumulators; they are zero when the call returns (it waits for the device: launches queued on any stream that
 * add to them are finished first).  Pointers from upsp_pipeline_accumulators stay valid. */
int upsp_pipeline_reset(upsp_pipeline *p);
/* The same without touching the device: the accumulators are zeroed by whichever call of this pipeline uses them next, on that
 * call's stream -- the one-camera streamed frame loop WRITES them in its first pass B instead (no fill launch, no read of
 * 8 B x N: a step that re-raycasts saves a launch on the frame loop's stream); upsp_pipeline_finalize clears them on its
 * stream, upsp_pipeline_accumulators waits for the device and clears them.  CONTRACT: between this call and that next call
 * the contents behind pointers obtained EARLIER from upsp_pipeline_accumulators are undefined (stale sums) and must not be read,
 * added to, or handed to upsp_allreduce_sums / upsp_exchange_finish_pixels -- fetch the pointers again with
 * upsp_pipeline_accumulators (which performs the pending zeroing) first.  A process call that fails before its first row pass is
 * queued leaves the reset pending. */
int upsp_pipeline_reset_deferred(upsp_pipeline *p);
/* avg = sum/N, rms = sqrt(sumsq/N) narrowed to f32 (psp_process.cpp:1933-1936) */
int upsp_pipeline_finalize(upsp_pipeline *p, uint64_t nframes_total, float *d_avg,
                           float *d_rms, void *stream);

/* ---- frame feed from host memory ------------------------------------------------------------
 * Replaces the reference's read-ahead thread + per-frame wait (cpp/exec/psp_process.cpp:867-1007,
 * 1756-1764): a ring of `nslots` pinned host buffers of `slot_bytes`, each with a device twin, and a
 * copy stream of its own.  Per chunk of frames:
 *   upsp_feed_acquire  -> next slot and its pinned host pointer (blocks only while that slot's
 *                         previous upload / consumer are still in flight); fill it (read the file
 *                         straight into it)
 *   upsp_feed_commit   -> hipMemcpyAsync host -> device twin on the copy stream; `consumer_stream`
 *                         is made to wait for it; returns the device pointer
 *   ... launches on consumer_stream that read the device pointer (upsp_unpack_12bit, upsp_pipeline_process)
 *   upsp_feed_release  -> marks the point on consumer_stream after which the device twin may be
 *                         overwritten
 * Slots are used round-robin and must be released in the order they were committed. */
typedef struct upsp_feed upsp_feed;
int upsp_feed_create(size_t slot_bytes, int nslots, upsp_feed **out);
void upsp_feed_destroy(upsp_feed *f);
int upsp_feed_acquire(upsp_feed *f, int *slot, void **h_ptr);
int upsp_feed_commit(upsp_feed *f, int slot, size_t nbytes, void *consumer_stream, void **d_ptr);
/* gives an acquired slot back unfilled (the reader failed: short read, frame out of range); it is the
 * next slot upsp_feed_acquire hands out */
int upsp_feed_abort(upsp_feed *f, int slot);
int upsp_feed_release(upsp_feed *f, int slot, void *consumer_stream);

/* ---- stand-alone per-frame operators (same kernels the pipeline uses) ------ */

/* upsp::fix_hot_pixels (cpp/utils/cv_extras.cpp:230-275) on nframes frames in place;
 * d_status[f] (may be NULL) = pixels replaced, -1 if more than max_hot looked hot. */
int upsp_fix_hot_pixels(uint16_t *d_frames, int nframes, int rows, int cols, int thresh,
                        int min_change, int max_hot, int32_t *d_status, void *stream);
/* upsp::project_frame (cpp/lib/projection.ipp:883-908) for one u16 / f32 image */
int upsp_project_frame_u16(const uint16_t *d_img, const int32_t *d_pix, const float *d_weight,
                           size_t nnodes, float *d_out, void *stream);
int upsp_project_frame_f32(const float *d_img, const int32_t *d_pix, const float *d_weight,
                           size_t nnodes, float *d_out, void *stream);
/* local_transpose (cpp/exec/psp_process.cpp:647-689): dst[x][y] = src[y][x];
 * src is [y_extent][x_extent], dst rows have leading dimension ld_dst >= y_extent. */
int upsp_transpose_f32(const float *d_src, int64_t x_extent, int64_t y_extent, float *d_dst,
                       int64_t ld_dst, void *stream);
/* apportion (cpp/exec/psp_process.cpp:611-624) */
int upsp_apportion(int value, int nbins, int *h_start, int *h_extent);

/* upsp::register_pixel (cpp/lib/registration.cpp:32-81) on one frame:
 * ECC affine (cv::findTransformECC semantics) then inverse-map warp of the u16 frame.
 * h_warp6: resulting 2x3 matrix; returns iterations (>=1) or a negative status. */
int upsp_register_pixel_u16(const float *d_ref32f, const uint16_t *d_inp, int rows, int cols,
                            int max_iters, double eps, int interp, uint16_t *d_out,
                            float *h_warp6, void *stream);
/* cv::GaussianBlur(img,img,Size(k,k),0) / cv::blur(img,img,Size(k,k)) as used at
 * psp_process.cpp:1802-1807 ; box = 0 gaussian, 1 box */
int upsp_blur_f32(const float *d_src, float *d_dst, int rows, int cols, int k, int box,
                  void *stream);
/* convertTo(CV_32F) + cv::GaussianBlur(Size(k,k), 0) of nimg u16 frames [nimg][rows][cols] in one pass (the pre-blur of
 * cv::findTransformECC, cpp/lib/registration.cpp:57-60, and the filter stage on raw frames, psp_process.cpp:1802-1804);
 * sizes 3 / 5 / 7 in one fused tile kernel (2 B in, 4 B out per pixel). */
int upsp_blur_u16(const uint16_t *d_src, float *d_dst, int nimg, int rows, int cols, int k, void *stream);
/* PatchClusters<float>::operator() (cpp/lib/patches.ipp:98-165) on one f32 image: per cluster polyfit2D (:172-205, float
 * column-pivoted Householder QR on raw pixel coordinates, the reference's arithmetic operation for operation) over the
 * boundary pixels, polyval2D (:208-236) at the interior ones; clusters of fewer than 10 boundary pixels are skipped (:103). */
int upsp_patch_f32(float *d_img, int rows, int cols, int nclusters, const int32_t *h_b_off,
                   const int32_t *h_bx, const int32_t *h_by, const int32_t *h_i_off,
                   const int32_t *h_ix, const int32_t *h_iy, void *stream);
/* the same on nimg f32 images [nimg][rows][cols] (the frames of a sub-batch of the loop, psp_process.cpp:1797-1800): the tables
 * are built once, a wave takes one cluster on 64 frames */
int upsp_patch_frames_f32(float *d_imgs, int nimg, int rows, int cols, int nclusters, const int32_t *h_b_off,
                          const int32_t *h_bx, const int32_t *h_by, const int32_t *h_i_off,
                          const int32_t *h_ix, const int32_t *h_iy, void *stream);

/* ======================================================================== *
 *  3b. Video decode -> device   (reference:
 *      upsp::unpack_12bit cpp/lib/PSPVideo.cpp:134-149, MrawReader::read_frame
 *      cpp/lib/MrawReader.cpp:113-146, 12-bit CineReader path cpp/lib/CineReader.cpp:428)
 * ======================================================================== */

/* Unpacks nframes frames of 12-bit packed pixels (3 bytes -> 2 pixels, MSBs first,
 * npix*3/2 bytes per frame, npix even) into u16 [nframes][npix] on the device.  Uploading the
 * packed bytes and unpacking in HBM moves 1.5 instead of 2 bytes per pixel over PCIe.
 * If d_hot_count != NULL (nframes counters, zeroed by the caller) the kernel also counts
 * the pixels >= hot_thresh per frame while they are in registers (pass 1 of fix_hot_pixels). */
int upsp_unpack_12bit(const uint8_t *d_packed, int nframes, size_t npix, uint16_t *d_frames,
                      int hot_thresh, uint32_t *d_hot_count, void *stream);

/* 10-bit packed Phantom CINE frames (5 bytes -> 4 pixels, MSBs first; upsp::unpack_10bit,
 * cpp/lib/PSPVideo.cpp:111-132; npix divisible by 4, npix*5/4 bytes per frame) -> u16, passed
 * through the 1024-entry 10 -> 12 bit look-up table d_lut when it is not NULL
 * (CineReader::read_packed, cpp/lib/CineReader.cpp:409-423; the table is camera data published
 * with the Cine file format and is supplied by the caller). */
int upsp_unpack_10bit(const uint8_t *d_packed, int nframes, size_t npix, const uint16_t *d_lut,
                      uint16_t *d_frames, void *stream);

/* ======================================================================== *
 *  3c. Phase 2: node-major time series -> delta-Cp   (reference:
 *      phase-2 node loop cpp/exec/psp_process.cpp:2452-2507, finals :2537-2545,
 *      upsp::TransPolyFitter<float> cpp/lib/filtering.ipp:12-79 / cpp/include/filtering.h:24-87,
 *      upsp::PaintCalibration::get_gain cpp/lib/non_cv_upsp.cpp:66-68)
 * ======================================================================== */

/* TransPolyFitter::eval_fit for npts rows: d_data_t [npts][ld_in >= nframes] (node-major, the
 * layout the reference maps as column-major Eigen data) -> d_fit_t [npts][ld_out] = value of
 * the degree-`degree` least-squares polynomial through each row over x = (float)f/nframes.
 * degree 0..7.  d_poly (optional) [npts][degree+1] = monomial coefficients (poly_ of the
 * reference, filtering.ipp:65-66).  May run in place (d_fit_t == d_data_t). */
int upsp_transpoly_fit(const float *d_data_t, long long ld_in, size_t npts, int nframes, int degree,
                       float *d_fit_t, long long ld_out, float *d_poly, void *stream);

/* Phase-2 node loop over this rank's node slice.  d_intensity_t [nnodes][ld_in] f32 (the
 * intensity_transpose block); per-node inputs d_iref (sol_avg_final), d_coverage, d_steady
 * (NULL = wind-off: zeros, :2354-2356), d_model_temp (NULL = the scalar `model_temp`, :2319)
 * indexed like the slice; paint_cal = a,b,c,d,e,f (host array).  Writes d_pressure_t
 * [nnodes][ld_out] (delta-Cp; NaN rows where coverage == 0 -- the reference leaves those rows
 * of its malloc'ed buffer unwritten), and, each optional: d_sum / d_sumsq (double partials,
 * local_avg / local_rms), d_avg = sum/F, d_rms = sqrt(sumsq/F), d_gain (NaN where coverage==0).
 * May run in place (d_pressure_t == d_intensity_t). */
int upsp_phase2_pressure(const float *d_intensity_t, long long ld_in, size_t nnodes, int nframes,
                         const float *d_iref, const float *d_coverage, const float *d_steady,
                         const float *d_model_temp, float model_temp, const float paint_cal[6],
                         float qbar, float ps, int degree, float *d_pressure_t, long long ld_out,
                         double *d_sum, double *d_sumsq, float *d_avg, float *d_rms, float *d_gain,
                         void *stream);

/* ======================================================================== *
 *  3d. Nearest model node   (reference: kd_nearest
 *      cpp/raycast/pspKdtree.c:284-372 over TriModel_::generate_kd_tree
 *      cpp/lib/TriModel.ipp:915-937; callers cpp/exec/psp_process.cpp:95-100,136-141)
 * ======================================================================== */

/* For each query point (double xyz, like the kd-tree's pos argument) the index of the nearest
 * node of d_nodes3 [nnodes][3] f32 and, optionally, the squared distance (double, accumulated
 * x,y,z like kd_nearest_i).  Exhaustive scan: same minimum distance as the kd-tree; the same
 * index unless two nodes are exactly equidistant (then the lowest index).  nqueries <= 65535. */
int upsp_nearest_nodes(const float *d_nodes3, size_t nnodes, const double *d_query3, size_t nqueries,
                       int32_t *d_index, double *d_dist2, void *stream);

/* upsp::interpolate (cpp/lib/interpolation.ipp:16-70): inverse-distance weighting over the k
 * nearest source nodes (ascending distance, float accumulation, exact hit -> that value), used to
 * carry a structured steady-state solution onto an unstructured model grid
 * (cpp/exec/psp_process.cpp:2341-2344, 2374-2377: k = 10, p = 2).  Source nodes / data are HOST
 * arrays (binned into a uniform grid here, once), queries and result live on the device.
 * d_neighbors (optional) [nquery][k] receives the neighbour ids (-1 padded).  k <= 16.
 * Waits for `stream` before returning. */
int upsp_interpolate_idw(const float *h_src_nodes3, const float *h_src_data, size_t nsrc,
                         const float *d_query_nodes3, size_t nquery, int k, float p, float *d_out,
                         int32_t *d_neighbors, void *stream);

/* ======================================================================== *
 *  3b.  Between the ranks of a job: one process per GPU, RCCL over xGMI
 *       (reference: MPI in psp_process -- apportion cpp/exec/psp_process.cpp:611-624,
 *        MPI_Reduce + MPI_Bcast of the accumulators :1866-1872, 2019-2023,
 *        global_transpose of the time series :707-771)
 * ======================================================================== */

/* A communicator of `world` ranks.  librccl is looked up in the running process first (PyTorch
 * brings its own copy) and in librccl.so.1 otherwise; it is not a link-time dependency.
 *   upsp_comm_unique_id   rank 0 makes the id (ncclGetUniqueId) and hands the 128 bytes to the others
 *                         (MPI_Bcast in a C++ host, any byte broadcast otherwise)
 *   upsp_comm_create      ncclCommInitRank on the CURRENT device (select it first)
 *   upsp_comm_from_nccl   wraps an ncclComm_t the host program already owns (not destroyed here)
 *   upsp_comm_create_local  `world` ranks in THIS process on the current device, device-to-device copies in
 *                         place of the links (tests: RCCL refuses two ranks on one GPU); the ranks are driven one
 *                         after the other from one thread: every rank submits before any rank finishes, and the
 *                         all-reduce completes with the call of the last rank */
typedef struct upsp_comm upsp_comm;
int upsp_comm_unique_id(uint8_t id[128]);
/* path of the RCCL build the library bound (the one the process had loaded, librccl.so.1, or UPSP_RCCL_LIBRARY) */
int upsp_comm_library(char *buf, size_t cap);
int upsp_comm_create(const uint8_t id[128], int rank, int world, upsp_comm **out);
int upsp_comm_from_nccl(void *nccl_comm, upsp_comm **out);
int upsp_comm_create_local(int world, upsp_comm **out_ranks);
void upsp_comm_destroy(upsp_comm *c);
int upsp_comm_rank(const upsp_comm *c, int *rank, int *world);

/* Sum of the double accumulators over the ranks, in place, on `stream` (MPI_Reduce + MPI_Bcast,
 * cpp/exec/psp_process.cpp:1866-1872, 2019-2023): one grouped pair of all-reduces of n doubles. */
int upsp_allreduce_sums(upsp_comm *c, double *d_sum, double *d_sumsq, size_t n, void *stream);

/* Time-series exchange (global_transpose, cpp/exec/psp_process.cpp:707-771): every rank holds the series of ALL
 * nodes over ITS frames and ends with the series of ITS nodes over ALL frames.  Frames and nodes are apportioned
 * like cpp/exec/psp_process.cpp:611-624, 1519-1529 (upsp_exchange_layout).  The rank's frames are produced in
 * `nchunks` chunks on 64-frame boundaries (upsp_exchange_chunk); chunk k is sent -- one grouped send / receive per
 * peer on the exchange's own stream, every xGMI link carrying one block -- while chunk k + 1 is processed.
 *   upsp_exchange_set_skipped  which rows travel: nodes no camera sees (d_skipped, identical on every rank; NULL:
 *                         none) are NaN in every frame everywhere and are filled by the receiver.  One host read;
 *                         assume_same != 0: the caller states the set did not change since the last call -- checked
 *                         on the device, reported by upsp_exchange_verify.
 *   upsp_exchange_rows    the row map for upsp_pipeline_set_row_map (packed row of every travelling node, ordered
 *                         by destination) and the number of packed rows: the frame loop writes chunk buffers
 *                         [packed_rows][chunk frames] (f32, or u16 with upsp_pipeline_process_u16) itself
 *   upsp_exchange_submit  chunk k (k = 0, 1, ... in order) as produced on `stream`; wire = 4 (f32 buffer), 2 (u16
 *                         buffer) or 12 (u16 buffer, packed to 12 bits for the wire: 12-bit cameras; a value above
 *                         4095 is an error at verify).  The buffer must stay untouched until finish.
 *   upsp_exchange_finish  waits for the transfers on `stream`, places every received block into d_series
 *                         [nodes of this rank][ld >= F] (f32; u16 / 12-bit blocks are widened) and writes the NaN rows
 *   upsp_exchange_bytes   bytes this rank sent to / received from OTHER ranks in the last finished pass */
typedef struct upsp_exchange upsp_exchange;
int upsp_exchange_create(upsp_comm *c, int64_t nframes_total, int64_t nnodes, int nchunks, upsp_exchange **out);
void upsp_exchange_destroy(upsp_exchange *x);
int upsp_exchange_layout(const upsp_exchange *x, int64_t *frame_start, int64_t *frame_count, int64_t *node_start,
                         int64_t *node_count);
int upsp_exchange_chunk(const upsp_exchange *x, int k, int64_t *first_frame, int64_t *nframes);
int upsp_exchange_set_skipped(upsp_exchange *x, const uint8_t *d_skipped, int assume_same, void *stream);
int upsp_exchange_rows(const upsp_exchange *x, const int32_t **d_rowmap, int64_t *packed_rows);
int upsp_exchange_submit(upsp_exchange *x, const void *d_chunk, int wire, void *stream);
int upsp_exchange_finish(upsp_exchange *x, float *d_series, int64_t ld, void *stream);
int upsp_exchange_verify(upsp_exchange *x, void *stream);
int upsp_exchange_bytes(const upsp_exchange *x, uint64_t *sent, uint64_t *received);

/* The same exchange with the series of the ACTIVE PIXELS on the wire instead of the node rows (plain one-camera path:
 * a node's series is its pixel's, and a model finer than the pixel grid has several nodes per pixel -- the bench model
 * 190 k travelling nodes on 66 k pixels).  Every destination receives the pixels its node slice reads, each once, and
 * runs pass B (the series + accumulators of ITS nodes over ALL frames) itself:
 *   upsp_pipeline_pixel_series  pass A alone: the REPAIRED series of every active pixel over `nframes` (<= 1024) frames
 *                         in the pipeline's compact buffer [active pixel][*cpitch] u16 (frames repaired in place like
 *                         fix_hot_pixels); *d_node_k [nnodes]: compact row of every node (< 0: none).  If pass A of exactly
 *                         these frames already ran on a candidate-pixel map (upsp_pipeline_set_active_hint +
 *                         upsp_pipeline_prescan, e.g. beside the projection build) it is not repeated: the nodes get their
 *                         rows in that buffer (the candidates must hold every pixel a node reads: a node outside them gets
 *                         -2, which upsp_exchange_set_pixels refuses), only the hot-pixel repair is left
 *   upsp_exchange_set_pixels    which pixel rows go where, from d_node_k (identical on every rank) and the skipped nodes
 *                         (one host read per projection; assume_same as above)
 *   upsp_exchange_submit_pixels chunk k out of the sender's compact buffer (wire 2: u16, 12: packed to 12 bits); d_compact
 *                         points at the chunk's first frame (a buffer holding all frames of the rank: d_compact + c0)
 *   upsp_exchange_finish_pixels places what arrived and runs pass B: d_series [nodes of this rank][ld >= F] f32 with NaN
 *                         rows for the skipped nodes; d_sum_mine / d_sumsq_mine [nodes of this rank] gain the sums over all
 *                         frames (pass the slice of the full-length accumulators: the other ranks' slices stay zero, and
 *                         upsp_allreduce_sums then delivers the complete vectors like in the row mode)
 *   upsp_rows_from_pixel_series pass B alone (used by the above): series and accumulators of nnodes nodes from pixel series */
int upsp_pipeline_pixel_series(upsp_pipeline *pipe, uint16_t *d_frames, int nframes, void *stream, const uint16_t **d_compact,
                               uint32_t *cpitch, const int32_t **d_node_k, const uint32_t **d_nactive);
/* Largest `nframes` one upsp_pipeline_pixel_series / upsp_pipeline_prescan call takes for this pipeline (a multiple of 64,
 * 64 .. 1024: the compact buffer [min(nodes, pixels)][frames] u16 must fit opts.compact_mb) -- a caller that cuts its frames
 * into chunks (psp_process.cpp:1519-1529 + the chunked exchange) keeps every chunk within it. */
int upsp_pipeline_series_frames_max(const upsp_pipeline *pipe);
int upsp_rows_from_pixel_series(const uint16_t *d_compact, uint32_t cpitch, const int32_t *d_node_k, const uint8_t *d_skipped,
                                size_t nnodes, int64_t nframes, float *d_rows_t, int64_t ld, double *d_sum, double *d_sumsq,
                                void *stream);
/* Pass B over the series of the same pixel rows held in `nblocks` buffers that follow each other in time (block b: [row][cpitch[b]]
 * u16 holding nframes[b] frames -- what arrived from peer b of the exchange): rows d_rows_t [nnodes][ld] over all the frames.
 * When every block holds a multiple of 4 frames the launches are cut at 128-byte lines of the output rows (every 32 columns), not
 * at the block boundaries: a launch reads the end of one block and the start of the next.  pad_to (frame total <= pad_to <= ld):
 * columns the last launch may write -- the frame total, or up to the next 128-byte line where the caller has padding there. */
int upsp_rows_from_pixel_blocks(const uint16_t *const *d_compact, const uint32_t *cpitch, const int64_t *nframes, int nblocks,
                                const int32_t *d_node_k, const uint8_t *d_skipped, size_t nnodes, float *d_rows_t, int64_t ld,
                                int64_t pad_to, double *d_sum, double *d_sumsq, void *stream);
int upsp_exchange_set_pixels(upsp_exchange *x, const int32_t *d_node_k, const uint8_t *d_skipped, int assume_same, void *stream);
int upsp_exchange_pixel_rows(const upsp_exchange *x, int64_t *rows_out, int64_t *rows_in);
int upsp_exchange_submit_pixels(upsp_exchange *x, const uint16_t *d_compact, uint32_t cpitch, int wire, void *stream);
/* upsp_pipeline_set_row_padding for the owner's pass B: on != 0 declares columns [F, min(ld, F rounded up to 32)) of the d_series
 * buffer handed to upsp_exchange_finish_pixels padding that may be written. */
int upsp_exchange_set_row_padding(upsp_exchange *x, int on);
int upsp_exchange_finish_pixels(upsp_exchange *x, float *d_series, int64_t ld, double *d_sum_mine, double *d_sumsq_mine,
                                void *stream);

/* ======================================================================== *
 *  4.  Measurement support (no reference counterpart; the reference only has
 *      psp::BlockTimer / timedBarrierPoint wall-clock prints, pspTimer.h:10-41)
 * ======================================================================== */

/* Per-kernel timing with HIP events recorded on the launch stream.  enable(1) clears
 * earlier records.  report(): one line per kernel "name calls total_ms min_ms median_ms max_ms" (the last
 * three: spread of the single timed spans); the caller must have synchronised the stream(s). */
int upsp_timing_enable(int on);
int upsp_timing_report(char *buf, size_t cap);

/* Device copy probe: a streaming 16-byte-per-lane copy kernel d_src -> d_dst of `bytes` (d_src NULL: the store half alone),
 * `reps` launches timed with HIP events on `stream`, the fastest of six launch shapes (non-temporal / plain accesses at 8, 16,
 * 32 workgroups per CU).  The measured HBM rate SURVEY.md 8(d) names as the roofline's
 * denominator: GB/s = (2 x bytes, or bytes for the fill) / ms_per_rep / 1e6. */
int upsp_copy_probe(const void *d_src, void *d_dst, size_t bytes, int reps, float *ms_per_rep, void *stream);
/* Read-only (kind 0) and write-only (kind 1) stream over `bytes` of d_buf in the access shapes of the frame loop's two passes --
 * pass A: non-temporal 16-byte loads, several in flight per lane, one-wave or four-wave workgroups; pass B: a workgroup sweeping
 * whole 4-KB row pieces with 16-byte non-temporal stores -- the fastest of four launch shapes each, *ms_per_rep by HIP events over
 * `reps` launches: the measured denominators of bench.py's roofline (a read-bound kernel is divided by the read rate, a
 * write-bound one by the write rate; the copy probe mixes both streams and is slower than either pass). */
int upsp_bandwidth_probe(int kind, void *d_buf, size_t bytes, int reps, float *ms_per_rep, void *stream);

/* Phase labels (reference: timedBarrierPoint / psp::BlockTimer, cpp/exec/psp_process.cpp:585-606): begin / end
 * nest; every phase is a roctx range (rocprofv3 --marker-trace: libroctx64 is looked up at run time, not linked) and,
 * with UPSP_PHASE_TIMES set, a "+++ label [total elapsed, this phase]" line on stderr like the reference's.  The
 * caller synchronises its stream(s) before upsp_phase_end when the phase's GPU work is to be inside the wall-clock figure.
 * UPSP_ROCTX=1 additionally wraps every timed kernel group of upsp_timing_* in a range of its own. */
int upsp_phase_begin(const char *label);
int upsp_phase_end(double *seconds);

#ifdef __cplusplus
}
#endif
#endif /* UPSP_GPU_H */
