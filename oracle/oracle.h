/*
 * CPU oracle for the ORB front-end hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This library restates, on the CPU and in scalar C++, what the reference does on the
 * path  ORBExtractor::detect/compute -> ORBMatcher::match -> Initializer F/H RANSAC.
 * The reference forwards the extractor and the brute-force matcher to OpenCV 3.4.9
 * (include/orb_slam/geometry/orb_extractor.h:57-80,100-119; src/geometry/orb_matcher.cpp:663-677),
 * whose source is not part of the reference tree: those parts follow the published
 * OpenCV 3.4 algorithms (features2d orb.cpp / fast.cpp / fast_score.cpp / keypoint.cpp,
 * imgproc resize / smooth / color, core lapack JacobiSVD) and are pinned by the
 * reference's own known answers (4583 / 4652 keypoints, the 79-pair F/H goldens,
 * the three-maxima case, the undistorted image bounds, and - for the descriptor stage,
 * statistically - the 750..800 filtered matches of src/tests/geometry/orb_matcher_tests.cpp:41,
 * met with 776 under Hamming distances).  Individual keypoint coordinates, descriptor bits,
 * match indices, inlier sets and scores are NOT recorded by any reference test: for those,
 * parity is unpinned.
 *
 * Nothing in the shipped product may include, link or call this file; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 */
#ifndef ORB_ORACLE_H
#define ORB_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same 28-byte layout as cv::KeyPoint. */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orc_keypoint;

/* Same 16-byte layout as cv::DMatch. */
typedef struct {
    int32_t query, train, img;
    float distance;
} orc_match;

typedef struct {
    int32_t nfeatures;      /* n_key_points */
    float   scale_factor;   /* 1.2f */
    int32_t nlevels;        /* level_pyramid */
    int32_t edge_threshold; /* 31 */
    int32_t first_level;    /* 0 (only 0 supported) */
    int32_t wta_k;          /* 2 (only 2 supported) */
    int32_t patch_size;     /* 31 (only 31 supported) */
    int32_t fast_threshold; /* score_threshold */
    /* behaviour switches for the parts of OpenCV that differ between builds */
    int32_t gray_variant;   /* 0: 14-bit (1868,9617,4899)  1: 15-bit (3735,19235,9798) */
    int32_t resize_variant; /* 0: INTER_LINEAR_EXACT       1: legacy INTER_LINEAR */
    int32_t retain_variant; /* 0: nth_element at n-1       1: nth_element at n (older OpenCV) */
    /* descriptor-stage switches; 0 everywhere is the behaviour the HIP path reproduces, the others exist to measure
       how far each open question about the reference's OpenCV build could move the descriptors */
    int32_t blur_variant;   /* 0: 8-bit fixed-point separable kernel (generic filter engine)
                               1: float separable kernel, one rounding at the end (what an IPP-backed build computes,
                                  up to its own rounding)   2: variant 1 on level 0 only (IPP is only picked for
                                  images large enough to be split over threads) */
    int32_t round_variant;  /* 0: cvRound = nearest-even (SSE2 build)  1: floor(v + 0.5) (the portable fallback) */
    float   blur_sigma;     /* 0 = 2.0 (orb.cpp) */
} orc_orb_params;

/* ---- extraction ---- */
/* Atlas geometry: rects = 4 ints per level (x, y, w, h of the un-padded level inside the atlas). */
void orc_orb_layout(const orc_orb_params* p, int w, int h, int nlevels,
                    int* atlas_w, int* atlas_h, int* rects, float* scales);
void orc_gray(const orc_orb_params* p, const uint8_t* img, int w, int h, int stride, int channels, uint8_t* gray);
/* Builds the padded pyramid atlas (atlas_w*atlas_h bytes, zero elsewhere). */
void orc_orb_pyramid(const orc_orb_params* p, const uint8_t* img, int w, int h, int stride, int channels,
                     int nlevels, uint8_t* atlas);
/* Blurs every level of an atlas in place exactly as cv::ORB does before describing. */
void orc_orb_blur(const orc_orb_params* p, int w, int h, int nlevels, uint8_t* atlas);
/* FAST-9/16 corner score map of one image (0 = no corner), before non-max suppression. */
void orc_fast_scores(const uint8_t* img, int w, int h, int stride, int threshold, uint8_t* scores);
/* FAST keypoints (raster order) after 3x3 non-max suppression. Returns the count (may exceed cap; only cap written). */
int  orc_fast_detect(const uint8_t* img, int w, int h, int stride, int threshold, orc_keypoint* out, int cap);
/* cv::ORB::detect. Returns the number of keypoints (may exceed cap; only cap written). */
int  orc_orb_detect(const orc_orb_params* p, const uint8_t* img, int w, int h, int stride, int channels,
                    orc_keypoint* out, int cap);
/* Per-stage detect: candidates after the FAST-score cull, in the order cv::ORB holds them, with Harris responses. */
/* Opt-in distribution mode, SELF-ORACLE (the reference has no counterpart; distribute_oracle.cpp states the rule):
   per-cell adaptive FAST threshold (low_threshold where a cell has no corner at p->fast_threshold) + quad-tree
   distribution with cv::ORB's per-level quotas.  orc_quadtree_select works on one level's candidate list. */
int  orc_quadtree_select(const int32_t* xs, const int32_t* ys, const int32_t* scores, int n, int w, int h, int lo,
                         int quota, int high_threshold, int cell_size, int32_t* kept);
int  orc_orb_detect_distributed(const orc_orb_params* p, const uint8_t* img, int w, int h, int stride, int channels,
                                int low_threshold, int cell_size, orc_keypoint* out, int cap);
int  orc_orb_detect_stage1(const orc_orb_params* p, const uint8_t* img, int w, int h, int stride, int channels,
                           orc_keypoint* out, int cap, int* per_level_counts);
/* cv::ORB::compute. kps is filtered/reordered in place; returns the new count; desc gets count*32 bytes. */
int  orc_orb_compute(const orc_orb_params* p, const uint8_t* img, int w, int h, int stride, int channels,
                     orc_keypoint* kps, int n, uint8_t* desc);
/* The reference's own scale tables (orb_extractor.h:39-54). out = 4 arrays of nlevels floats, log factor returned. */
float orc_scale_tables(float scale_factor, int nlevels, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2);
float orc_fast_atan2(float y, float x);
int  orc_retain_best(const float* keys, int n, int n_points, int32_t* perm_out);

/* ---- camera (adjacent) ---- */
typedef struct { double fx, fy, cx, cy, k1, k2, p1, p2, k3; int32_t width, height; } orc_camera;
void orc_undistort_points(const orc_camera* c, const float* xy_in, int n, float* xy_out);
/* out = min_x, max_x, min_y, max_y (camera.cpp:129-162) */
void orc_image_bounds(const orc_camera* c, float* out4);

/* ---- matching ---- */
int  orc_descriptor_distance(const uint8_t* a, const uint8_t* b);
/* cv::BFMatcher(NORM_HAMMING).match: one (train idx, distance) per query; lowest train index wins ties. */
void orc_match_hamming(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, int32_t* dist);
/* ORBMatcher::filterMatches (orb_matcher.cpp:890-924). In place; returns the new count. n_desc_rows = descriptors.rows. */
int  orc_filter_matches(orc_match* m, int n, int n_desc_rows);
/* BruteForceWithRadiusMatcher::match (orb_matcher.cpp:25-74). Returns the match count. */
int  orc_match_radius(const orc_keypoint* kp, int n, const orc_keypoint* ref_kp, int n_ref,
                      const uint8_t* desc, const uint8_t* ref_desc, double max_pixel_dist, orc_match* out);
void orc_three_maxima(const int* hist_sizes, int L, int* ind1, int* ind2, int* ind3);

/* Frame grid + windowed matchers (frame.cpp:89-218,304-367; orb_matcher.cpp:115-352). */
typedef struct {
    float min_x, min_y, max_x, max_y; /* undistorted image bounds */
    int32_t grid_cols, grid_rows;
    int32_t grid_size_x, grid_size_y; /* integer cell sizes (frame.cpp:309-310) */
} orc_grid_params;
/* Frame::assignFeaturesToGrid: filters kps in place (returns new count); writes CSR cells in grid_[col][row] order:
   cell_start has cols*rows+1 entries, cell_items has (new count) entries. */
int  orc_assign_grid(const orc_grid_params* g, orc_keypoint* kps, int n, int32_t* cell_start, int32_t* cell_items);
/* Frame-to-frame projection matcher, orb_matcher.cpp:115-255.
   ref side: n_ref keypoints, has_mp flags, projected pixel (already world->frame, z sign in proj_z), best descriptor.
   level_mode: 0 = frame roughly level (octave-1..octave+1), 1 = frame in front (4-arg overload), 2 = frame behind. */
int  orc_match_projection_frame(const orc_grid_params* g, const int32_t* cell_start, const int32_t* cell_items,
                                const orc_keypoint* kps, int n, const uint8_t* desc,
                                const orc_keypoint* ref_kps, int n_ref, const uint8_t* ref_has_mp,
                                const float* proj_xy, const float* proj_z, const uint8_t* ref_desc,
                                const float* scale_factors, int level_mode, float radius, int check_orientation,
                                orc_match* out);
/* Map-point projection matcher, orb_matcher.cpp:257-352 (track properties precomputed by the caller). */
int  orc_match_projection_points(const orc_grid_params* g, const int32_t* cell_start, const int32_t* cell_items,
                                 const orc_keypoint* kps, int n, const uint8_t* desc,
                                 int n_pts, const uint8_t* in_view, const float* proj_xy, const int32_t* pred_level,
                                 const float* view_cos, const uint8_t* pt_desc,
                                 const float* scale_factors, float radius, float nn_ratio, orc_match* out);

/* BoW matcher, orb_matcher.cpp:386-497: FeatureVector maps as CSR (node ids ascending). */
int  orc_match_bow(const orc_keypoint* kps, int n, const uint8_t* desc, const orc_keypoint* ref_kps, int n_ref,
                   const uint8_t* ref_desc, const uint8_t* ref_has_mp, const uint32_t* node_ids,
                   const int32_t* node_start, const int32_t* node_items, int n_nodes, const uint32_t* ref_node_ids,
                   const int32_t* ref_node_start, const int32_t* ref_node_items, int n_ref_nodes, int low_threshold,
                   float nn_ratio, int check_orientation, orc_match* out);

/* Epipolar matcher, orb_matcher.cpp:520-661 (quirks kept, see the definition).  Returns -1 when a node index is
   out of range for the arrays the reference indexes with it. */
int  orc_match_epipolar(const orc_keypoint* kps, int n, const uint8_t* desc, const orc_keypoint* ref_kps, int n_ref,
                        const uint8_t* ref_desc, const uint8_t* has_mp, const uint8_t* ref_has_mp, const float* F9,
                        const float* scale_sigmas, const uint32_t* node_ids, const int32_t* node_start,
                        const int32_t* node_items, int n_nodes, const uint32_t* ref_node_ids,
                        const int32_t* ref_node_start, const int32_t* ref_node_items, int n_ref_nodes,
                        int low_threshold, int check_orientation, orc_match* out);

/* Candidate search of ORBMatcher::fuse, orb_matcher.cpp:806-867. */
void orc_fuse_search(const orc_grid_params* g, const int32_t* cell_start, const int32_t* cell_items,
                     const orc_keypoint* kps, const uint8_t* desc, int n_pts, const uint8_t* in_view,
                     const float* proj_xy, const int32_t* pred_level, const uint8_t* pt_desc,
                     const float* scale_factors, const float* inv_scale_sigmas, float radius_multiplier,
                     int32_t* best_idx, int32_t* best_dist);

/* MapPoint::computeBestDescriptor (map_point.cpp:109-166) over descriptor sets laid out back to back. */
void orc_best_descriptor(const uint8_t* desc, const int32_t* point_start, int n_points, int32_t* best);

/* ---- two-view geometry ---- */
void orc_normalize_points(const float* xy, int n, float* xy_norm, float* T9);
/* cv::SVDecomp(A, w, u, vt, MODIFY_A|FULL_UV) on CV_32F, stock (non-LAPACK) OpenCV.  u may be NULL. vt is n x n. */
void orc_svd_f32(const float* A, int m, int n, float* w, float* u, float* vt);
void orc_fundamental_8pt(const float* pts_norm, const float* ref_norm, int n, float* F9);
void orc_homography_dlt(const float* pts_norm, const float* ref_norm, int n, float* H9);
double orc_score_fundamental(const float* F9, const float* pts, const float* ref, int n, double sigma, uint8_t* inliers);
double orc_score_homography(const float* H9, const float* Hinv9, const float* pts, const float* ref, int n,
                            double sigma, uint8_t* inliers);
/* std::shuffle-based set generation of Initializer.cpp:261-277 with a fixed mt19937 seed per iteration (seed0+it). */
void orc_ransac_sets(int n, int iterations, uint32_t seed0, uint32_t* sets8);
/* Initializer::findFundamentalAndHomography with injected sets (Initializer.cpp:254-585).
   best_it[0/1] = winning iteration of F / H (-1 when no iteration scored > 0, the matrix is then left untouched). */
void orc_ransac_fh(const float* pts, const float* ref, int n, const uint32_t* sets8, int iterations, double sigma,
                   float* F9, float* H9, uint8_t* inl_f, uint8_t* inl_h, double* score_f, double* score_h,
                   int* best_it);
/* All per-iteration hypotheses and scores (for parity tests of the solver/scorer). */
void orc_ransac_fh_all(const float* pts, const float* ref, int n, const uint32_t* sets8, int iterations, double sigma,
                       float* F_all, float* H_all, float* Hinv_all, float* score_f, float* score_h);

/* ---- pose recovery after RANSAC (Initializer.cpp:165-252); OpenCV parts restated, parity unpinned ---- */
/* cv::SVD::compute on a square CV_64F matrix (flags 0): w descending, u (n x n), vt (n x n). u / vt may be NULL. */
void orc_svd_square_f64(const double* A, int n, double* w, double* u, double* vt);
/* cv::triangulatePoints: P1, P2 3x4 row-major, x1 / x2 n points (x, y) -> n homogeneous points (x, y, z, w). */
void orc_triangulate_points(const double* P1, const double* P2, const double* x1, const double* x2, int n,
                            double* out4);
/* Initializer::findRtWithFundamental: returns cv::recoverPose's count. */
int  orc_recover_pose_fundamental(const float* F9, const float* K9, const double* inlier_points,
                                  const double* inlier_ref_points, int n, double* R9, double* t3);
/* Initializer::triangulatePoints. */
void orc_triangulate_initial(const float* K9, const double* R9, const double* t3, const double* inlier_points,
                             const double* inlier_ref_points, int n, double* points3);
/* src/Initializer.cpp:112-163 (findRtWithHomography): cv::decomposeHomographyMat and
   cv::filterHomographyDecompByVisibleRefpoints of OpenCV 3.4.9; the last one is the whole function as it would run
   if the filter accepted the reference's double points (it asserts on them, so the reference returns false). */
int  orc_decompose_homography(const float* H9, const float* K9, double* R, double* t, double* nrm);
int  orc_filter_homography_visible(const double* R, const double* nrm, int nsol, const float* before_points,
                                   const float* after_points, int npoints, int32_t* solutions);
int  orc_recover_pose_homography(const float* H9, const float* K9, const double* inlier_points,
                                 const double* inlier_ref_points, int n, double* R9, double* t3);

#ifdef __cplusplus
}
#endif
#endif
