/*
 * upsp_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference algorithms on the hot path of
 * nasa/upsp-processing (BVH ray-triangle intersector + psp_process phase-1
 * frame loop).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product path (libupsp_gpu.so)
 * never links, imports or calls it.
 *
 * Every function cites the reference file:line (relative to the upstream
 * repository root) whose arithmetic it follows.
 *
 * PARITY STATUS
 *   rt_* (ray cast)      pinned: reference KATs test/python/test_visibility.py
 *                        (5 toy cases + 148 608-node regression) and
 *                        test/python/test_photogrammetry.py hit position.
 *                        The ray/box test lives in Imath 3.1.x (vcpkg tag
 *                        2023.04.15, un-vendored): restated from its published
 *                        algorithm (ImathBoxAlgo.h intersects(Box,Line3,V3)).
 *   proj_* / frame_*     pinned only through the ray-cast KATs above; the
 *                        reference ships no test for create_projection_mat,
 *                        project_frame, fix_hot_pixels (cpp/test/test_projection.h
 *                        is a TODO list).  Integer / gather work is restated
 *                        exactly; cv::projectPoints (OpenCV 4.5.2, un-vendored)
 *                        is restated from its published formula.
 *   ecc_* / warp / blur  PARITY UNPINNED: arithmetic lives in OpenCV 4.5.2
 *                        (findTransformECC, warpAffine, GaussianBlur), not in
 *                        the reference tree and covered by no reference test.
 *   patch_*              PARITY UNPINNED: Eigen 3.3.9 colPivHouseholderQr
 *                        (un-vendored), no reference test for PatchClusters.
 *   patch set-up         PARITY UNPINNED (no reference test): getTargets, diameters,
 *                        clustering, pixel lists, histogram threshold (patchsetup_oracle.c).
 *   kd_*                 pinned: the vendored kd-tree itself (cpp/raycast/pspKdtree.c) is
 *                        compiled into oracle/_ref/ and compared node-for-node.
 *   transpoly / phase2   pinned: cpp/test/test_filtering.cpp:19-113 (TransPolyfitter
 *                        known-answer test); the QR itself is Eigen (un-vendored),
 *                        restated in qr_f32.h.
 */
#ifndef UPSP_ORACLE_H
#define UPSP_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- rt ---- */

/* rt::Ray  (cpp/include/utils/pspRT.h:37-50, ctor cpp/raycast/pspRT.cpp:45-69) */
typedef struct orc_ray {
    float o[3], d[3], inv[3];
    int kx, ky, kz;
    float Sx, Sy, Sz;
} orc_ray;

/* rt::Hit  (cpp/include/utils/pspRT.h:26-36, ctor cpp/raycast/pspRT.cpp:21-36) */
typedef struct orc_hit {
    float pos[3], nrm[3];
    float t, u, v, w;
    int32_t geomID, primID; /* reference leaves these uninitialised; oracle uses -1 */
} orc_hit;

/* BVH::LinearNode (cpp/include/utils/pspRT.h:71-82), 32 bytes */
typedef struct orc_node {
    float bmin[3], bmax[3];
    int32_t offset;   /* leaf: primitivesOffset, interior: secondChildOffset */
    uint16_t nprims;  /* 0 -> interior */
    uint8_t axis;
    uint8_t pad;
} orc_node;

typedef struct orc_bvh {
    size_t ntris;
    float *verts;      /* 9*ntris, copy of the input soup (mesh->p) */
    int32_t *prim_ids; /* ordered primitive list: leaf slot -> original triangle index */
    orc_node *nodes;
    int32_t nnodes;
    int32_t depth;     /* max stack depth a near-first DFS can reach (tree height) */
} orc_bvh;

void orc_ray_init(orc_ray *r, const float o[3], const float d[3]);
void orc_hit_init(orc_hit *h);

/* rt::Triangle::intersect, cpp/raycast/pspRT.cpp:109-193 */
int orc_tri_intersect(const orc_ray *ray, const float *A, const float *B, const float *C,
                      int32_t primID, orc_hit *hit);

/* Imath::intersects(Box3f, Line3f(o, o+d), V3f&) as called at pspRT.cpp:382-385.
 * ldir = normalised (o+d)-o  (computed by orc_line_dir). */
void orc_line_dir(const float o[3], const float d[3], float ldir[3]);
int orc_box_hit(const float bmin[3], const float bmax[3], const float pos[3], const float ldir[3]);

/* rt::CreateTriangleMesh + rt::BVH::BVH  (pspRT.cpp:206-222, 313-344, 433-572).
 * Returns NULL for ntris == 0 (reference prints a message and leaves nodes==nullptr). */
orc_bvh *orc_bvh_create(const float *tris9, size_t ntris);
void orc_bvh_destroy(orc_bvh *b);

/* rt::BVH::intersect, cpp/raycast/pspRT.cpp:359-431.  *hit must be initialised
 * (orc_hit_init) by the caller, exactly like the reference.  Optional counters. */
int orc_bvh_intersect(const orc_bvh *b, const orc_ray *ray, orc_hit *hit,
                      uint32_t *nodes_visited, uint32_t *tris_tested);

/* Batch driver (OpenMP over rays; threads<=0 -> all cores).  Outputs may be NULL. */
void orc_bvh_intersect_batch(const orc_bvh *b, const float *org3, const float *dir3, size_t n,
                             int org_stride, /* 0: one shared origin, 3: per-ray */
                             uint8_t *hit, float *t, int32_t *prim, float *uvw3,
                             float *pos3, float *nrm3, int threads,
                             uint64_t *nodes_visited, uint64_t *tris_tested);

/* --------------------------------------------------------- projection --- */

/* Pin-hole + Brown distortion camera, the inputs of cv::projectPoints as used by
 * upsp::CameraCal::map_point_to_image (cpp/lib/CameraCal.ipp:218-231). */
typedef struct orc_camera {
    double K[9];     /* cameraMatrix row-major */
    double dist[5];  /* k1,k2,p1,p2,k3 (psp_process reads only the first 4: CameraCal.cpp:27,46) */
    double R[9];     /* model->camera rotation, row-major */
    double t[3];
    int32_t width, height;
} orc_camera;

/* cv::projectPoints for one point (double arithmetic, result narrowed to float). */
void orc_project_point(const orc_camera *cam, const float xyz[3], float uv[2]);
/* CameraCal::get_cam_center, cpp/lib/CameraCal.cpp:192-203 : C = -R^T t (double) */
void orc_cam_center(const orc_camera *cam, double c[3]);

/* create_projection_mat, cpp/exec/psp_process.cpp:167-355.
 * pix[n]  = round(v)*W+round(u) for accepted nodes, -1 otherwise (CSR with <=1 nnz/row)
 * uv[2n]  = (u/W, v/H) for accepted nodes, 0 otherwise
 * nodecount (W*H u8, may be NULL) saturating nodes-per-pixel
 * returns number of accepted nodes; *nrays = rays cast. */
int64_t orc_create_projection(const orc_bvh *bvh, const orc_camera *cam,
                              const float *nodes3, const float *normals3,
                              const uint8_t *datanode, /* may be NULL = all data nodes */
                              const int32_t *tri_nodes3, size_t nnodes,
                              float oblique_thresh, int32_t *pix, float *uv,
                              uint8_t *nodecount, uint64_t *nrays, int threads);

/* Nodes whose oblique-test verdict depends on the last bit of libm's acosf (see proj_oracle.c). */
int64_t orc_oblique_ambiguous(const orc_camera *cam, const float *nodes3, const float *normals3,
                              const uint8_t *datanode, size_t nnodes, float oblique_thresh);

/* adjust_projection_for_weights + BestView/AverageViews
 * (cpp/lib/projection.ipp:911-1078, 227-268; angle_between cv_extras.ipp:69-73).
 * pix[c*nnodes+n] <0 = not seen; weight[c*nnodes+n] in/out (1.0 where seen).
 * mode 0 = BestView, 1 = AverageViews. */
void orc_adjust_weights(int ncams, size_t nnodes, const int32_t *pix, float *weight,
                        const float *nodes3, const float *normals3,
                        const double *centers3, int mode);

/* identify_skipped_nodes, cpp/lib/projection.ipp:857-880: skipped[n]=1 iff no camera sees n */
size_t orc_skipped_nodes(int ncams, size_t nnodes, const int32_t *pix, uint8_t *skipped);

/* ------------------------------------------------------------- frames --- */

/* upsp::fix_hot_pixels, cpp/utils/cv_extras.cpp:230-275 (defaults cv_extras.h:154-155) */
int orc_fix_hot_pixels(uint16_t *img, int rows, int cols, int thresh, int min_change, int max_hot);

/* upsp::project_frame with <=1 nnz per row, cpp/lib/projection.ipp:883-908 :
 * out[n] = weight[n]*img[pix[n]] (0 where pix<0).  img is f32 (after convertTo). */
void orc_project_frame_f32(const float *img, const int32_t *pix, const float *weight,
                           size_t nnodes, float *out);
void orc_project_frame_u16(const uint16_t *img, const int32_t *pix, const float *weight,
                           size_t nnodes, float *out);

/* frame-loop tail, cpp/exec/psp_process.cpp:1813-1843 : sum cameras (float), NaN
 * for skipped nodes, double accumulators. */
void orc_frame_loop_u16(uint16_t *frames, int nframes, int rows, int cols, const int32_t *pix,
                        const float *weight, const int32_t *skipped, size_t nskipped, size_t nnodes,
                        int thresh, int min_change, int max_hot, float *out_rows, double *sum,
                        double *sumsq, int threads, double *seconds);
void orc_accumulate(const float *sol, size_t nnodes, double *sum, double *sumsq);
/* finals, psp_process.cpp:1930-1979 */
void orc_finals(const double *sum, const double *sumsq, size_t nnodes, uint64_t nframes,
                float *avg, float *rms);

/* apportion, psp_process.cpp:611-624 */
void orc_apportion(int value, int nbins, int *start, int *extent);
/* local_transpose, psp_process.cpp:647-689 : dst[x][y] = src[y][x] */
void orc_transpose(const float *src, int x_extent, int y_extent, float *dst);

/* ------------------------------------------------------ image ops ------- */

/* cv::GaussianBlur(img, img, Size(k,k), 0) / cv::blur, BORDER_REFLECT_101
 * (psp_process.cpp:1802-1807).  PARITY UNPINNED (OpenCV). */
int orc_gaussian_kernel(int k, float *coef /* k */);
void orc_blur_f32(const float *src, float *dst, int rows, int cols, int k, int box);

/* cv::warpAffine(src_u16|f32, dst, M, size, INTER_LINEAR|WARP_INVERSE_MAP,
 * BORDER_CONSTANT 0) as used by register_pixel (cpp/lib/registration.cpp:69-73).
 * PARITY UNPINNED (OpenCV). interp: 1 = INTER_LINEAR, 0 = INTER_NEAREST */
void orc_warp_affine_u16(const uint16_t *src, uint16_t *dst, int rows, int cols,
                         const float M[6], int interp);
void orc_warp_affine_f32(const float *src, float *dst, int rows, int cols,
                         const float M[6], int interp);

/* cv::findTransformECC(ref, inp, M, MOTION_AFFINE, {COUNT+EPS, max_iters, eps})
 * as called by register_pixel (cpp/lib/registration.cpp:32-66); gaussFiltSize=5.
 * M in/out (row-major 2x3).  Returns iterations used, <0 on divergence
 * (OpenCV throws).  rho_out may be NULL.  PARITY UNPINNED (OpenCV). */
int orc_find_transform_ecc(const float *ref, const float *inp, int rows, int cols,
                           float M[6], int max_iters, double eps, double *rho_out);

/* upsp::register_pixel, cpp/lib/registration.cpp:32-81 (u16 input frame) */
int orc_register_pixel_u16(const float *ref, const uint16_t *inp, int rows, int cols,
                           float M[6], int max_iters, double eps, int interp, uint16_t *out);

/* PatchClusters<float>::operator(), polyfit2D, polyval2D
 * (cpp/lib/patches.ipp:98-236).  One cluster: boundary (bx,by)[nb], interior
 * (ix,iy)[ni]; img modified in place.  Float column-pivoted Householder QR on raw
 * pixel coordinates like Eigen's colPivHouseholderQr().solve().  PARITY UNPINNED. */
int orc_polyfit2d(const int32_t *x, const int32_t *y, const float *z, int n, float poly[10]);
void orc_polyval2d(const int32_t *x, const int32_t *y, int n, const float poly[10], float *z);
void orc_patch_clusters(float *img, int cols, int nclusters, const int32_t *b_off,
                        const int32_t *bx, const int32_t *by, const int32_t *i_off,
                        const int32_t *ix, const int32_t *iy);

/* ------------------------------------------------------------ kd-tree --- */

/* kd_nearest over all model nodes, cpp/raycast/pspKdtree.c:131-171,260-372 (vendored in the
 * reference; oracle/_ref/libpspkdtree.so is the real thing where /root/reference exists). */
typedef struct orc_kdtree orc_kdtree;
orc_kdtree *orc_kd_build(const float *nodes3, size_t n);
void orc_kd_free(orc_kdtree *t);
int32_t orc_kd_nearest(const orc_kdtree *t, const double pos[3], double *dist2_out);
/* upsp::interpolate, cpp/lib/interpolation.ipp:16-70 (exhaustive k-nearest + IDW). PARITY UNPINNED. */
void orc_interpolate_idw(const float *src3, const float *data, size_t nsrc, const float *q3, size_t nq,
                         int k, float p, float *out, int32_t *nbr);
void orc_kd_nearest_batch(const orc_kdtree *t, const double *query3, size_t nq, int32_t *index,
                          double *dist2);

/* ------------------------------------------------- phase-0 patch set-up --- */

/* getTargets, cpp/exec/psp_process.cpp:55-112 */
void orc_get_targets(const orc_bvh *bvh, const orc_kdtree *kd, const orc_camera *cam,
                     const float *normals3, const float *xyz3, size_t n, float oblique_thresh,
                     uint8_t *keep);
/* get_target_diameters, psp_process.cpp:114-165 */
void orc_target_diameters(const orc_kdtree *kd, const orc_camera *cam, const float *normals3,
                          const float *xyz3, const float *uv2, const float *diam_in, size_t n,
                          float *diam_out);
/* cluster_points, cpp/lib/patches.ipp:239-276 */
int orc_cluster_points(const float *uv2, const float *diam, int n, int bound_pts, int32_t *order,
                       int32_t *cl_off);
/* intensity_histc cpp/lib/image_processing.ipp:10-50; find_peaks / first_min_threshold
 * cpp/utils/clustering.ipp:9-96 */
void orc_intensity_histc(const uint16_t *img, size_t npix, unsigned depth, int bins, int32_t *edges,
                         int32_t *counts);
int orc_find_peaks(const double *data, int n, unsigned separation, uint32_t *peaks);
unsigned orc_first_min_threshold(const int32_t *counts, int n, unsigned separation);
/* PatchClusters ctor + threshold_bounds, patches.ipp:14-94,279-485 */
void orc_patch_tables(const float *uv2, const float *diam, const int32_t *order,
                      const int32_t *cl_off, int ncl, int cols, int rows, unsigned bound_pts,
                      unsigned buffer, const uint16_t *ref, unsigned thresh, unsigned offset,
                      int32_t **b_off, int32_t **bx, int32_t **by, int32_t **i_off, int32_t **ix,
                      int32_t **iy);
void orc_free(void *p);

/* ------------------------------------------------------------ phase 2 --- */

/* upsp::TransPolyFitter<float> (cpp/lib/filtering.ipp:12-79): design matrix
 * A[c*nframes+f] = pow((float)f/nframes, c) and the per-point fit + evaluation.
 * Eigen colPivHouseholderQr -> qr_f32.h; pinned by cpp/test/test_filtering.cpp:19-113. */
void orc_transpoly_design(int nframes, int degree, float *A);
int orc_transpoly_fit(const float *A, int nframes, int ncoef, const float *y, float *poly,
                      float *fit);
/* PaintCalibration::get_gain, cpp/lib/non_cv_upsp.cpp:66-68 ; cal = a,b,c,d,e,f */
float orc_paint_gain(const float cal[6], float T, float Pss);
/* phase-2 node loop, cpp/exec/psp_process.cpp:2452-2507 */
void orc_phase2(const float *intensity_t, size_t nnodes, int nframes, const float *iref,
                const float *coverage, const float *steady, const float *model_temp,
                const float cal[6], float qbar, float ps, int degree, float *pressure_t,
                double *sum, double *sumsq, double *gain_out, int threads);

/* ------------------------------------------------------------- video ---- */

/* upsp::unpack_12bit, cpp/lib/PSPVideo.cpp:134-149 (MRAW / 12-bit CINE frames) */
void orc_unpack_12bit(const uint8_t *packed, size_t nbytes, uint16_t *out);
/* upsp::unpack_10bit, cpp/lib/PSPVideo.cpp:111-132, + optional 10->12 bit LUT (CineReader.cpp:409-423) */
void orc_unpack_10bit(const uint8_t *packed, size_t nbytes, const uint16_t *lut, uint16_t *out);

#ifdef __cplusplus
}
#endif
#endif
