/*
 * orbx — C ABI of the MI355X (gfx950) ORB front-end: extraction, descriptor matching and
 * two-view F/H RANSAC.  Every entry point replaces one interface of the reference
 * (saifullah3396/orb_slam); the citation after each declaration names it.
 *
 * Conventions
 *  - plain pointers and sizes only; nothing is thrown across this boundary; every call returns an orbx_status.
 *  - outputs are caller-allocated with an explicit capacity; overflow is reported (ORBX_CAPACITY), never truncated.
 *  - handles own device buffers and one HIP stream and serve one call at a time: a call that arrives from a second
 *    thread while one is running returns ORBX_BUSY and touches nothing (give every thread its own handle; calls on
 *    different handles run side by side).
 *  - pointers are HOST pointers unless the parameter name starts with d_ (device pointer on the handle's GPU).
 *  - orbx_keypoint / orbx_match are layout-identical to cv::KeyPoint / cv::DMatch, so std::vector storage
 *    of the reference's callers can be passed without copies.
 */
#ifndef ORBX_H
#define ORBX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    ORBX_OK = 0,
    ORBX_BAD_ARG = 1,
    ORBX_CAPACITY = 2,
    ORBX_DEVICE = 3,
    ORBX_UNSUPPORTED = 4,
    ORBX_BUSY = 5 /* the handle is serving a call of another thread */
} orbx_status;

typedef struct { float x, y, size, angle, response; int32_t octave, class_id; } orbx_keypoint; /* == cv::KeyPoint */
typedef struct { int32_t query, train, img; float distance; } orbx_match;                      /* == cv::DMatch */

/* Parameters of ORBExtractor (include/orb_slam/geometry/orb_extractor.h:28-37, cfg/orb_extractor.yaml). */
typedef struct {
    int32_t n_key_points;    /* default 8000 */
    float   scale_factor;    /* default 1.2f */
    int32_t level_pyramid;   /* default 4   */
    int32_t edge_threshold;  /* 31 */
    int32_t first_level;     /* 0  (only 0 is supported) */
    int32_t wta_k;           /* 2  (only 2 is supported) */
    int32_t patch_size;      /* 31 (only 31 is supported) */
    int32_t score_threshold; /* FAST threshold, default 20 */
} orbx_extractor_params;

typedef struct orbx_extractor orbx_extractor;
typedef struct orbx_matcher orbx_matcher;
typedef struct orbx_ransac orbx_ransac;

const char* orbx_status_string(orbx_status s);
/* Last HIP error text seen by the calling thread (empty string when none). */
const char* orbx_last_error(void);
/* Number of visible GPUs; 0 when the HIP runtime finds none. */
int orbx_device_count(void);

/* ------------------------------------------------------------------ extraction ------------------------------ */

/* Replaces ORBExtractor::ORBExtractor (orb_extractor.h:28-81): validates the parameters and reserves device
   buffers for images of max_width x max_height on GPU `device`; batches hold at most max_batch frames.
   max_width / max_height are a reservation, not a limit: a larger image (up to 16384 x 16384) makes the buffers
   grow at that call, as cv::ORB takes whatever size it is handed. */
orbx_status orbx_extractor_create(const orbx_extractor_params* params, int max_width, int max_height,
                                  int max_batch, int device, orbx_extractor** out);
void orbx_extractor_destroy(orbx_extractor* h);

/* Replaces the getters scaleFactors()/scaleSigmas()/invScaleSigmas()/logScaleFactor() (orb_extractor.h:39-54,
   86-92): the reference's own float tables (repeated multiplication), `level_pyramid` entries each. */
orbx_status orbx_extractor_scale_tables(const orbx_extractor* h, float* scale_factors, float* inv_scale_factors,
                                        float* scale_sigma_sqrd, float* inv_scale_sigma_sqrd,
                                        float* log_scale_factor);

/* Replaces ORBExtractor::detect (orb_extractor.h:100-104 -> cv::ORB::detect).  image: 8-bit, channels 1 (gray)
   or 3 (BGR), `stride` bytes per row.  Writes at most `capacity` keypoints; *count receives the number found
   (ORBX_CAPACITY when it exceeds capacity, nothing is written then). */
orbx_status orbx_detect(orbx_extractor* h, const uint8_t* image, int width, int height, int stride, int channels,
                        orbx_keypoint* keypoints, int capacity, int* count);

/* Replaces ORBExtractor::compute (orb_extractor.h:113-119 -> cv::ORB::compute).  keypoints is filtered and,
   when not sorted by octave, regrouped in place exactly as cv::ORB does; *count is updated; descriptors
   receives *count rows of 32 bytes.
   MonoFrame::extractFeatures calls detect and then compute on the same image (src/mono_frame.cpp:75,102), where
   cv::ORB builds the pyramid twice.  Here the handle remembers the image of its last detect / compute (address,
   geometry and a hash of every pixel) and a compute on that image finds the upload and the pyramid still on the
   device; any change of a pixel between the two calls makes the compute start from the new pixels.  Results are
   identical with the reuse switched off (orbx_extractor_set_image_cache). */
orbx_status orbx_compute(orbx_extractor* h, const uint8_t* image, int width, int height, int stride, int channels,
                         orbx_keypoint* keypoints, int* count, uint8_t* descriptors);

/* Batched detect + compute on frames already resident in HBM (the call sequence of MonoFrame::extractFeatures,
   src/mono_frame.cpp:72-110, with the host steps between detect and compute left out).
   d_images: n_frames images, `frame_stride` bytes apart.  Results stay on the device inside the handle
   (see orbx_extractor_device_results) and are copied to the host arrays when those are non-NULL:
   keypoints[n_frames * capacity], counts[n_frames], descriptors[n_frames * capacity * 32]. */
orbx_status orbx_extract_batch(orbx_extractor* h, const uint8_t* d_images, int n_frames, size_t frame_stride,
                               int width, int height, int stride, int channels,
                               orbx_keypoint* keypoints, int capacity, int32_t* counts, uint8_t* descriptors);

/* With keypoints, counts and descriptors all NULL, orbx_extract_batch returns as soon as its kernels are queued on
   the handle's stream: the device results (orbx_extractor_device_results / _device_grid) may be consumed by work
   queued on the same stream right away, and the host never waits inside the call.  orbx_extractor_finish then waits
   for that batch and reports what the synchronous call would have returned: ORBX_CAPACITY when a frame has more
   keypoints than `capacity`; counts[n_frames] when non-NULL.  *redone = 1 (when non-NULL) says that frames were
   left over by the queued selection (more frames beyond the LDS form than the global form has workspace slices, or a
   candidate list that overflowed) and the batch was finished inside this call - by further device passes, or by the
   host selection for an overflowed list: the device results are now the final ones, and whatever consumed them
   before has to run again.  A batch
   that was never finished is waited for and dropped by the next call on the handle; a finish that arrives after
   that returns ORBX_BAD_ARG (orbx_last_error() says why) and writes nothing.  No counterpart in the reference (its
   extraction is synchronous). */
orbx_status orbx_extractor_finish(orbx_extractor* h, int32_t* counts, int* redone);

/* Switches the reuse of the uploaded image and its pyramid between orbx_detect and orbx_compute (default on). */
orbx_status orbx_extractor_set_image_cache(orbx_extractor* h, int enabled);

/* No counterpart in the reference: which kernels build the pyramid (cv::ORB's cvtColor + resize chain).  kind 0
   (default) and 1: level by level - gray conversion, then one launch per level; 2: the one-pass kernel whenever the
   geometry suits it - one workgroup per frame reads the image once and writes every level, the rows in flight kept
   in LDS rings, one launch per batch (it moves less than half the bytes but measures slower on 792 VGA frames, 0.98
   against 0.77 ms, which is why it is not the default).  rows_per_step > 0 sets how many image rows a step of the
   one-pass kernel converts (default 8).  Results are identical. */
orbx_status orbx_extractor_set_pyramid_kernel(orbx_extractor* h, int kind, int rows_per_step);

/* No counterpart in the reference (its extractor is cv::ORB: one FAST threshold, retainBest by Harris response,
   include/orb_slam/geometry/orb_extractor.h:57-80,100-104): an opt-in keypoint distribution, OFF by default so that
   results equal cv::ORB's.  mode 1:
     - FAST-9/16 runs at low_threshold; a corner's score is the largest threshold at which it still is a corner;
     - per-cell adaptive threshold: every level's keypoint area is cut into cells of about cell_size pixels; a cell
       that holds a corner at the configured score_threshold keeps only such corners, a cell without one keeps every
       corner at low_threshold;
     - quad-tree distribution with cv::ORB's per-level quotas: the area starts as round(width / height) boxes (1..8);
       round by round every box with more than one corner is cut into up to four (at the ceilings of its half width
       and height; empty quadrants dropped, the others taking the box's place in order); once cutting all of them
       could pass the quota, boxes are cut largest population first (earlier box first among equals) until the quota
       is reached; every box keeps its strongest corner (first in raster order among equals), in box order;
     - keypoint response = the corner score; size, angle, octave, undistortion, grid and descriptors as otherwise.
   A level keeps at most quota + 2 keypoints, so give orbx_extract_batch a capacity of n_key_points + 3 * levels.
   The mode lives in the device-resident batch path; orbx_detect runs a batch of one.  A frame with more corners at
   the low threshold than the candidate list holds (a sixteenth of the pyramid's pixels) is refused with
   ORBX_UNSUPPORTED.  mode 0 restores cv::ORB's selection. */
orbx_status orbx_extractor_set_distribution(orbx_extractor* h, int mode, int low_threshold, int cell_size);

/* No counterpart in the reference: the small levels of the pyramid in one launch.  Levels 1 .. base_level are resized
   level by level (each spread over the whole chip); the levels above base_level, which are too small to fill the
   chip one at a time, are then produced by ONE launch in which a workgroup per frame reads level base_level's rows
   once and walks the remaining levels through LDS rings.  base_level 0 turns this off.  rows_per_step: rows of
   level base_level a step takes (0 keeps the current value, default 16).  A geometry the kernel does not suit is
   built level by level throughout.  Results are identical either way. */
orbx_status orbx_extractor_set_pyramid_tail(orbx_extractor* h, int base_level, int rows_per_step);

/* Camera model of Camera::undistortPoints (src/geometry/camera.cpp:97-126 -> cv::undistortPoints with P = K).
   Intrinsics and distortion as the reference holds them (cv::Mat_<float> widened to double). */
typedef struct { double fx, fy, cx, cy, k1, k2, p1, p2, k3; } orbx_camera;

/* Frame bucket grid (Frame::setupGrid / assignFeaturesToGrid / pointInGrid, src/frame.cpp:304-367). */
typedef struct {
    float min_x, min_y, max_x, max_y; /* Camera::minX() ... (src/geometry/camera.cpp:149-158) */
    int32_t grid_cols, grid_rows;     /* /orb_slam/tracker/grid_cols, grid_rows */
    int32_t grid_size_x, grid_size_y; /* undistWidth()/grid_cols, undistHeight()/grid_rows as int (frame.cpp:309-310) */
} orbx_grid;

/* Makes orbx_extract_batch follow MonoFrame::extractFeatures (src/mono_frame.cpp:72-110) step by step: after
   detect the keypoints are undistorted (camera != NULL and k1 != 0, camera.cpp:101), bucketed and filtered by
   Frame::assignFeaturesToGrid (grid != NULL), and described at the undistorted coordinates.  NULL clears. */
orbx_status orbx_extractor_set_frame_model(orbx_extractor* h, const orbx_camera* camera, const orbx_grid* grid);

/* orbx_extract_batch has two equivalent forms of the steps between the detect kernels and the describe kernels
   (keypoint selection, undistortion, grid, border filter): on the device (default, nothing crosses PCIe) or on
   the host with the reference's own std:: calls (enabled = 0).  Results are identical; the switch exists for
   tests and for frames with more candidates than the device form holds on chip. */
orbx_status orbx_extractor_set_device_select(orbx_extractor* h, int enabled);

/* No counterpart in the reference: scheduling of orbx_extract_batch.  Enabled (the default), the Gaussian blur of a
   batch runs on a second stream of the handle beside the latency-bound kernels between FAST and the descriptors
   (ordered emission, Harris/angle, selection): best when ONE batch is in flight on the GPU.  A caller that keeps
   several batches in flight on handles of their own gets more from the chip with it disabled (the other batches fill
   those gaps, and the blur then never competes with them).  Results are identical either way. */
orbx_status orbx_extractor_set_blur_fork(orbx_extractor* h, int enabled);

/* Camera::undistortPoints on n (x,y) float pairs; host arithmetic in double as cv::undistortPoints. */
orbx_status orbx_undistort_points(const orbx_camera* camera, const float* xy_in, int n, float* xy_out);

/* Device views of the per-frame bucket grids of the last orbx_extract_batch (set_frame_model with a grid):
   cell_start int32[frames][cols*rows + 1], cell_items int32[frames][kp_capacity]. */
orbx_status orbx_extractor_device_grid(orbx_extractor* h, const int32_t** d_cell_start, const int32_t** d_cell_items);

/* Device views of the last orbx_extract_batch: keypoints (orbx_keypoint[frames][kp_capacity]),
   descriptors (uint8[frames][kp_capacity][32]), counts (int32[frames]). */
orbx_status orbx_extractor_device_results(orbx_extractor* h, const orbx_keypoint** d_keypoints,
                                          const uint8_t** d_descriptors, const int32_t** d_counts,
                                          int* kp_capacity);

/* Stage probes used by the parity tests (no counterpart in the reference: they expose the intermediate images
   cv::ORB keeps private).  what: 0 = pyramid atlas, 1 = FAST score map, 2 = blurred atlas.  The atlas has
   the geometry reported by orbx_extractor_layout. */
orbx_status orbx_extractor_layout(const orbx_extractor* h, int width, int height, int* atlas_w, int* atlas_h,
                                  int32_t* rects_xywh, float* layer_scales);
orbx_status orbx_extractor_read_plane(orbx_extractor* h, int frame, int what, uint8_t* out, size_t out_bytes);
/* Per-kernel launch durations (ms) of the last call, measured with HIP events on the handle's stream.
   names: gray, resize, fast, nms, emit, harris, blur, brief, select, angle.  Returns the number written (<= cap). */
int orbx_extractor_last_timings(const orbx_extractor* h, float* ms, int cap);

/* ------------------------------------------------------------------ matching -------------------------------- */

orbx_status orbx_matcher_create(int device, orbx_matcher** out);
void orbx_matcher_destroy(orbx_matcher* h);

/* Replaces CVORBMatcher::match (src/geometry/orb_matcher.cpp:671-677 -> cv::BFMatcher(NORM_HAMMING)::match):
   for every query row the train row of minimum Hamming distance, lowest train index on ties.
   train_idx[i] = -1 when n_train == 0. */
orbx_status orbx_match_hamming(orbx_matcher* h, const uint8_t* query, int n_query, const uint8_t* train,
                               int n_train, int32_t* train_idx, int32_t* distance);

/* Batched form on device-resident descriptors: pair p matches d_query + p*query_stride (n_query[p] rows) against
   d_train + p*train_stride (n_train[p] rows); results d_train_idx/d_distance[p*out_stride + i] (device).
   Descriptor blocks and strides must be multiples of 16 bytes; at most 65 535 pairs per call. */
orbx_status orbx_match_hamming_batch(orbx_matcher* h, const uint8_t* d_query, const int32_t* d_n_query,
                                     size_t query_stride, const uint8_t* d_train, const int32_t* d_n_train,
                                     size_t train_stride, int n_pairs, int max_query, int max_train,
                                     int32_t* d_train_idx, int32_t* d_distance, size_t out_stride);

/* The same match (orb_matcher.cpp:671-677) for EVERY query set against EVERY train set in one call - the many-to-many
   matching of frames against a key-frame database: pair p = q * n_train_sets + t matches query set q
   (d_query + q*query_stride, n_query[q] rows) against train set t (d_train + t*train_stride, n_train[t] rows); results
   at [p*out_stride + i].  n_query_sets * n_train_sets <= 65535. */
orbx_status orbx_match_hamming_cross(orbx_matcher* h, const uint8_t* d_query, const int32_t* d_n_query,
                                     size_t query_stride, int n_query_sets, const uint8_t* d_train,
                                     const int32_t* d_n_train, size_t train_stride, int n_train_sets, int max_query,
                                     int max_train, int32_t* d_train_idx, int32_t* d_distance, size_t out_stride);

/* Replaces ORBMatcher::filterMatches (orb_matcher.cpp:890-924).  Host arithmetic (sort + unique with the same
   std:: calls).  In place; *count updated.  n_descriptor_rows = descriptors.rows of the reference call. */
orbx_status orbx_filter_matches(orbx_match* matches, int* count, int n_descriptor_rows);

/* Replaces BruteForceWithRadiusMatcher::match (orb_matcher.cpp:25-74): per keypoint the reference keypoint
   within `max_pixel_dist` pixels with the smallest mean absolute descriptor-byte difference. */
orbx_status orbx_match_radius(orbx_matcher* h, const orbx_keypoint* keypoints, int n, const orbx_keypoint* ref_keypoints,
                              int n_ref, const uint8_t* descriptors, const uint8_t* ref_descriptors,
                              double max_pixel_dist, orbx_match* matches, int* count);

/* Replaces Frame::assignFeaturesToGrid: drops out-of-grid keypoints in place (*count updated) and writes the
   cells in grid_[col][row] order as CSR: cell_start[cols*rows + 1], cell_items[*count]. */
orbx_status orbx_assign_grid(const orbx_grid* g, orbx_keypoint* keypoints, int* count, int32_t* cell_start,
                             int32_t* cell_items);

/* Replaces BruteForceWithProjectionMatcher::match(frame, ref_frame, ref_T_f, ref_map_points, matches)
   (orb_matcher.cpp:115-255) together with Frame::getFeaturesAroundPoint (frame.cpp:151-218),
   applyRotationConstraint / createMatches (orb_matcher.h:57-93) and computeThreeMaxima (utils.cpp:202-236).
   The caller has projected every reference map point into the frame (Frame::worldToCamera / cameraToFrame):
   proj_xy = pixel, proj_z = depth in the camera frame, ref_has_point[i] = 0 when feature i has no map point,
   ref_descriptors row i = MapPoint::bestDescriptor().  level_mode: 0 neither, 1 frame in front, 2 frame behind
   (the z test of orb_matcher.cpp:141-144). */
orbx_status orbx_match_projection_frame(orbx_matcher* h, const orbx_grid* g, const int32_t* cell_start,
                                        const int32_t* cell_items, const orbx_keypoint* keypoints, int n,
                                        const uint8_t* descriptors, const orbx_keypoint* ref_keypoints, int n_ref,
                                        const uint8_t* ref_has_point, const float* proj_xy, const float* proj_z,
                                        const uint8_t* ref_descriptors, const float* scale_factors, int n_levels,
                                        int level_mode, float radius, int check_orientation,
                                        orbx_match* matches, int* count);

/* Batched form of orbx_match_projection_frame on the device results of an extractor (orbx_extract_batch with a
   frame model): pair p matches frame d_frame_index[p] against reference frame d_ref_index[p].  The reference
   features are the keypoints of the reference frame, their best descriptors its descriptors.  d_proj_xy
   [pairs][kp_capacity][2] = projection of every reference feature into the frame; d_ref_has_point / d_proj_z
   [pairs][kp_capacity] may be NULL (every feature has a point / depth positive).  Outputs (device):
   d_matched [pairs][kp_capacity] = reference index matched to each frame feature (<= 0: none, as the reference's
   `matched` vector), d_distance likewise, d_n_matches [pairs] = number of entries createMatches would emit. */
orbx_status orbx_match_projection_frame_batch(orbx_matcher* h, const orbx_grid* g, int n_pairs,
                                              const int32_t* d_frame_index, const int32_t* d_ref_index,
                                              const orbx_keypoint* d_keypoints, const uint8_t* d_descriptors,
                                              const int32_t* d_counts, int kp_capacity,
                                              const int32_t* d_cell_start, const int32_t* d_cell_items,
                                              const uint8_t* d_ref_has_point, const float* d_proj_xy,
                                              const float* d_proj_z, const float* scale_factors, int n_levels,
                                              int level_mode, float radius, int check_orientation,
                                              int32_t* d_matched, int32_t* d_distance, int32_t* d_n_matches);

/* Replaces BowOrbMatcher::match (orb_matcher.cpp:386-497).  The two DBoW2::FeatureVector maps (node id -> feature
   indices) are passed as CSR arrays with ascending node ids: node_ids[n_nodes], node_start[n_nodes + 1],
   node_items[node_start[n_nodes]].  ref_has_point[i] = reference feature i has a map point that is not bad.
   low_threshold / nn_ratio / check_orientation are the matcher's members (orb_matcher.h:197-201: 50, 0.6, true). */
orbx_status orbx_match_bow(orbx_matcher* h, const orbx_keypoint* keypoints, int n, const uint8_t* descriptors,
                           const orbx_keypoint* ref_keypoints, int n_ref, const uint8_t* ref_descriptors,
                           const uint8_t* ref_has_point, const uint32_t* node_ids, const int32_t* node_start,
                           const int32_t* node_items, int n_nodes, const uint32_t* ref_node_ids,
                           const int32_t* ref_node_start, const int32_t* ref_node_items, int n_ref_nodes,
                           int low_threshold, float nn_ratio, int check_orientation, orbx_match* matches,
                           int* count);

/* Replaces EpipolarConstraintWithBowMatcher::match (orb_matcher.cpp:520-661) with the fundamental matrix
   (KeyFrame::computeFundamentalMat, row-major 3x3) supplied by the caller; feature vectors as in orbx_match_bow.
   has_point[i] / ref_has_point[i] = feature i of the (reference) key frame already has a map point.  scale_sigmas =
   ORBExtractor::scaleSigmas() of the reference frame.  The reference's arithmetic is kept as written, including
   its reading of row idx of BOTH descriptor matrices (:573) and of matched[ref_idx] (:565); where those reads
   would leave the arrays (a feature index of a common node >= n_ref, a reference index >= n) ORBX_BAD_ARG is
   returned.  low_threshold / check_orientation: orb_matcher.h (50, true). */
orbx_status orbx_match_epipolar(orbx_matcher* h, const orbx_keypoint* keypoints, int n, const uint8_t* descriptors,
                                const orbx_keypoint* ref_keypoints, int n_ref, const uint8_t* ref_descriptors,
                                const uint8_t* has_point, const uint8_t* ref_has_point, const float* fundamental,
                                const float* scale_sigmas, int n_levels, const uint32_t* node_ids,
                                const int32_t* node_start, const int32_t* node_items, int n_nodes,
                                const uint32_t* ref_node_ids, const int32_t* ref_node_start,
                                const int32_t* ref_node_items, int n_ref_nodes, int low_threshold,
                                int check_orientation, orbx_match* matches, int* count);

/* Many frame pairs per call for the two matchers over vocabulary nodes: one workgroup per pair, so the key frames a
   local-mapping step compares (src/local_mapping/local_mapper.cpp:277,462) are matched side by side.  A pair holds
   exactly the arguments of the single-pair entry points.  matches: pair p writes from matches + (sum of n of the
   earlier pairs), at most n of them; counts[p] = how many. */
typedef struct orbx_bow_pair {
    const orbx_keypoint* keypoints; int n; const uint8_t* descriptors;
    const orbx_keypoint* ref_keypoints; int n_ref; const uint8_t* ref_descriptors;
    const uint8_t* ref_has_point;
    const uint32_t* node_ids; const int32_t* node_start; const int32_t* node_items; int n_nodes;
    const uint32_t* ref_node_ids; const int32_t* ref_node_start; const int32_t* ref_node_items; int n_ref_nodes;
} orbx_bow_pair;
typedef struct orbx_epipolar_pair {
    const orbx_keypoint* keypoints; int n; const uint8_t* descriptors;
    const orbx_keypoint* ref_keypoints; int n_ref; const uint8_t* ref_descriptors;
    const uint8_t* has_point; const uint8_t* ref_has_point;
    const float* fundamental;
    const uint32_t* node_ids; const int32_t* node_start; const int32_t* node_items; int n_nodes;
    const uint32_t* ref_node_ids; const int32_t* ref_node_start; const int32_t* ref_node_items; int n_ref_nodes;
} orbx_epipolar_pair;
orbx_status orbx_match_bow_batch(orbx_matcher* h, const orbx_bow_pair* pairs, int n_pairs, int low_threshold,
                                 float nn_ratio, int check_orientation, orbx_match* matches, int* counts);
orbx_status orbx_match_epipolar_batch(orbx_matcher* h, const orbx_epipolar_pair* pairs, int n_pairs,
                                      const float* scale_sigmas, int n_levels, int low_threshold,
                                      int check_orientation, orbx_match* matches, int* counts);

/* Replaces the candidate search of ORBMatcher::fuse (orb_matcher.cpp:806-867) for the points that passed the
   caller's own tests (:808-814: not bad, not yet observed in the key frame, KeyFrame::isInCameraViewLocal, whose
   projection and predicted level arrive in proj_xy / pred_level).  best_index[p] = the key frame's feature with the
   smallest descriptor distance in the window (-1: none), best_distance[p] that distance (256: none); the caller
   applies best_distance <= low_threshold and the map bookkeeping (:869-884). */
orbx_status orbx_fuse_search(orbx_matcher* h, const orbx_grid* g, const int32_t* cell_start, const int32_t* cell_items,
                             const orbx_keypoint* keypoints, int n, const uint8_t* descriptors, int n_points,
                             const uint8_t* in_view, const float* proj_xy, const int32_t* pred_level,
                             const uint8_t* point_descriptors, const float* scale_factors,
                             const float* inv_scale_sigmas, int n_levels, float radius_multiplier,
                             int32_t* best_index, int32_t* best_distance);

/* Replaces the arithmetic of MapPoint::computeBestDescriptor (src/map_point.cpp:109-166) for a batch of map points:
   the observed descriptors of point p are rows point_start[p] .. point_start[p + 1] of `descriptors` (32 bytes
   each, at most 1024 per point); best_index[p] = the row inside the set with the least median Hamming distance to
   the set (first such row), -1 for an empty set. */
orbx_status orbx_best_descriptor_batch(orbx_matcher* h, const uint8_t* descriptors, const int32_t* point_start,
                                       int n_points, int32_t* best_index);

/* Replaces BruteForceWithProjectionMatcher::match(frame, points_3d, matches) (orb_matcher.cpp:257-352) with the
   track properties (Frame::isInCameraView, frame.cpp:220-268) supplied by the caller. */
orbx_status orbx_match_projection_points(orbx_matcher* h, const orbx_grid* g, const int32_t* cell_start,
                                         const int32_t* cell_items, const orbx_keypoint* keypoints, int n,
                                         const uint8_t* descriptors, int n_points, const uint8_t* in_view,
                                         const float* proj_xy, const int32_t* pred_level, const float* view_cos,
                                         const uint8_t* point_descriptors, const float* scale_factors, int n_levels,
                                         float radius, float nn_ratio, orbx_match* matches, int* count);

/* Batched form of orbx_match_projection_points on the device results of an extractor: pair p matches the map points
   of its own list (d_n_points[p] of them; per-pair arrays [pairs][points_capacity]: in_view, proj_xy (x 2),
   pred_level, view_cos, point descriptors (x 32)) against frame d_frame_index[p] - what Tracker::projectLocalPoints
   (src/tracker.cpp:606-607 -> orb_matcher.cpp:257-352) does for one frame, for many frames per call.  Outputs as in
   orbx_match_projection_frame_batch: d_matched / d_distance [pairs][kp_capacity] per frame feature, d_n_matches. */
orbx_status orbx_match_projection_points_batch(orbx_matcher* h, const orbx_grid* g, int n_pairs,
                                               const int32_t* d_frame_index, const orbx_keypoint* d_keypoints,
                                               const uint8_t* d_descriptors, const int32_t* d_counts,
                                               int kp_capacity, const int32_t* d_cell_start,
                                               const int32_t* d_cell_items, int points_capacity,
                                               const int32_t* d_n_points, const uint8_t* d_in_view,
                                               const float* d_proj_xy, const int32_t* d_pred_level,
                                               const float* d_view_cos, const uint8_t* d_point_descriptors,
                                               const float* scale_factors, int n_levels, float radius,
                                               float nn_ratio, int32_t* d_matched, int32_t* d_distance,
                                               int32_t* d_n_matches);

/* ------------------------------------------------------------------ two-view RANSAC ------------------------- */

orbx_status orbx_ransac_create(int device, orbx_ransac** out);
void orbx_ransac_destroy(orbx_ransac* h);

/* Replaces Initializer::findFundamentalAndHomography (src/Initializer.cpp:254-304) with everything it calls:
   geometry::normalizePoints (src/geometry/utils.cpp:18-52), computeFundamentalMat (:61-103),
   computeHomographyMat (:112-154), findFundamentalMat / checkFundamentalScore (Initializer.cpp:306-444),
   findHomographyMat / checkHomographyScore (:446-585).
   points / ref_points: n (x,y) float pairs (points_ / ref_points_).  sets8: iterations x 8 indices into them
   (ransac_sets_).  Outputs: F, H row-major 3x3; inliers_* n bytes (0/1); scores as the reference stores them;
   best_iteration[0/1] = winning iteration of F / H, -1 when no hypothesis scored above 0 (the matrix is then
   left untouched, as fundamental_mat_ / homography_mat_ stay empty in the reference). */
orbx_status orbx_ransac_fh(orbx_ransac* h, const float* points, const float* ref_points, int n,
                           const uint32_t* sets8, int iterations, double sigma,
                           float* F, float* H, uint8_t* inliers_f, uint8_t* inliers_h,
                           double* score_f, double* score_h, int32_t* best_iteration);

/* Batched form on device-resident data: pair p uses points d_points + p*n*2 etc.; every pair has n points and
   `iterations` sets.  Outputs are device arrays: d_F/d_H [pairs][9], d_inliers_* [pairs][n], d_scores [pairs][2]
   (float-rounded scores, F then H), d_best_iteration [pairs][2].  The sets are device data, so the call cannot
   inspect them without waiting for the GPU: a set index >= n is never used as an address, and the pair it belongs
   to comes back with d_best_iteration -2, score 0 and no inliers (its d_F / d_H rows are left untouched). */
orbx_status orbx_ransac_fh_batch(orbx_ransac* h, const float* d_points, const float* d_ref_points, int n,
                                 const uint32_t* d_sets8, int iterations, int n_pairs, double sigma,
                                 float* d_F, float* d_H, uint8_t* d_inliers_f, uint8_t* d_inliers_h,
                                 float* d_scores, int32_t* d_best_iteration);

/* orbx_ransac_fh_batch for callers that cannot allocate device memory while they run (tight serving loops,
   harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments, same results bit for bit; the
   intermediate arrays (normalised points, every hypothesis and its score) live in a fixed device array of about
   10.1 MB that each GPU the library runs on holds until the process ends, and the call makes no hipMalloc / hipFree
   (nor any other allocating call: its events exist since orbx_ransac_create).  Limits per call: n_pairs <= 1024,
   n_pairs * iterations <= 32768, n_pairs * n <= 131072; beyond them ORBX_CAPACITY is returned (orbx_last_error()
   names the limits) and nothing runs.  Every call on a GPU shares that GPU's array, whichever handle it goes
   through: the library orders such calls on the device (each waits for an event the previous one recorded), so two
   handles or two streams never overlap inside the array - they take turns.  The array keeps the last call's
   contents; the call clears the words that must start at zero itself.  Not exercised under HIP stream capture (the
   call records timing events and forks onto a second stream of the handle). */
orbx_status orbx_ransac_fh_batch_static(orbx_ransac* h, const float* d_points, const float* d_ref_points, int n,
                                        const uint32_t* d_sets8, int iterations, int n_pairs, double sigma,
                                        float* d_F, float* d_H, uint8_t* d_inliers_f, uint8_t* d_inliers_h,
                                        float* d_scores, int32_t* d_best_iteration);

/* Per-hypothesis probe for the parity tests: all hypotheses and scores of one pair (host arrays, iterations
   rows): F_all/H_all/Hinv_all [it][9], score_f/score_h [it]. */
orbx_status orbx_ransac_fh_all(orbx_ransac* h, const float* points, const float* ref_points, int n,
                               const uint32_t* sets8, int iterations, double sigma,
                               float* F_all, float* H_all, float* Hinv_all, float* score_f, float* score_h);

/* The three public solvers of the reference's geometry/utils.h, for ANY number of correspondences (the RANSAC entry
   points above run them on the 8-point sets of their hypotheses).  Host pointers in and out.

   orbx_normalize_points replaces geometry::normalizePoints (include/orb_slam/geometry/utils.h:24-27,
   src/geometry/utils.cpp:18-52): normalized receives n (x, y) pairs, T9 the row-major 3x3 transform.
   orbx_fundamental_npt replaces geometry::computeFundamentalMat (utils.h:36-39, utils.cpp:61-103): the n x 9
   system, cv::SVDecomp, vt.row(8), rank-2 enforcement by a second SVD.
   orbx_homography_dlt replaces geometry::computeHomographyMat (utils.h:48-51, utils.cpp:112-154): the 2n x 9 system,
   vt.row(8).
   Both solvers take n_systems independent systems of n correspondences each, back to back (system q owns points
   q*n .. q*n + n - 1 and row q of the 9-float output); the reference's functions are the n_systems = 1 case. */
orbx_status orbx_normalize_points(orbx_ransac* h, const float* points, int n, float* normalized, float* T9);
orbx_status orbx_fundamental_npt(orbx_ransac* h, const float* points, const float* ref_points, int n, int n_systems,
                                 float* F9);
orbx_status orbx_homography_dlt(orbx_ransac* h, const float* points, const float* ref_points, int n, int n_systems,
                                float* H9);

/* Replaces Initializer::findRtWithFundamental (src/Initializer.cpp:165-199) for a batch of frame pairs: E = K^T F K
   as the reference forms it (CV_32F), then cv::recoverPose(E, inlier_ref_points, inlier_points, R, t, focal, pp) of
   OpenCV 3.4 in double (decomposeEssentialMat, triangulation of every correspondence against the four candidate
   poses, cheirality count with the distance limit 50).  fundamental: n_pairs x 9 (row-major, float);
   intrinsics: the camera's 3x3 intrinsicMatrix() (float); points / ref_points: the inlier correspondences of all
   pairs back to back as (x, y) doubles, pair p owning rows pair_start[p] .. pair_start[p + 1].
   Out: rotation n_pairs x 9, translation n_pairs x 3, n_good = recoverPose's return value. */
orbx_status orbx_recover_pose_batch(orbx_ransac* h, int n_pairs, const float* fundamental, const float* intrinsics,
                                    const double* points, const double* ref_points, const int32_t* pair_start,
                                    double* rotation, double* translation, int32_t* n_good);

/* Initializer::findRtWithHomography (src/Initializer.cpp:112-163) for a batch of frame pairs, as it is written:
   cv::decomposeHomographyMat(H, K) of OpenCV 3.4.9 in double (Hnorm = K^-1 H K over its middle singular value, the
   analytic decomposition into [Ra, ta, na], [Ra, -ta, -na], [Rb, tb, nb], [Rb, -tb, -nb], or the single motion
   [Hnorm, 0, 0] of a pure rotation), then cv::filterHomographyDecompByVisibleRefpoints(rotations, normals,
   inlier_points, inlier_ref_points) and the first solution it keeps.  In the reference that second call never
   completes: the function's first statement is CV_Assert(both point arrays are CV_32FC2), the reference hands it
   std::vector<cv::Point2d> (CV_64FC2), the cv::Exception is caught (Initializer.cpp:150-155) and the function
   returns false - which is what the Initializer facade does unless told otherwise.  This entry point computes what
   the call sequence yields when the same coordinates arrive as floats (they are narrowed here, as Point2f would
   hold them).  homography: n_pairs x 9 (row-major, float); points / ref_points / pair_start as for
   orbx_recover_pose_batch.  Out: solution[p] = index of the chosen motion or -1 when none is visible (rotation and
   translation rows are then zero; the reference would leave R and t empty and fail in triangulatePoints);
   all_solutions (may be NULL): n_pairs x 4 x 15 doubles (R 9, t 3, n 3), n_solutions (may be NULL): 1 or 4. */
orbx_status orbx_recover_pose_homography_batch(orbx_ransac* h, int n_pairs, const float* homography,
                                               const float* intrinsics, const double* points, const double* ref_points,
                                               const int32_t* pair_start, double* rotation, double* translation,
                                               int32_t* solution, double* all_solutions, int32_t* n_solutions);

/* Replaces Initializer::triangulatePoints (src/Initializer.cpp:201-252): cv::triangulatePoints with
   projection = K [R | t] and projection_ref = K [I | 0], each point divided by its fourth coordinate.
   points_3d: one (x, y, z) per correspondence, in the order of points. */
orbx_status orbx_triangulate_batch(orbx_ransac* h, int n_pairs, const float* intrinsics, const double* rotation,
                                   const double* translation, const double* points, const double* ref_points,
                                   const int32_t* pair_start, double* points_3d);

/* The reference's set generator (Initializer.cpp:261-277: cumulative std::shuffle, first 8) with a caller-chosen
   seed per iteration (seed0 + it) in place of std::random_device.  Host arithmetic. */
orbx_status orbx_ransac_make_sets(int n, int iterations, uint32_t seed0, uint32_t* sets8);

/* No counterpart in the reference: scheduling of the RANSAC entry points.  Enabled (the default), the homography
   chain (solve, score) runs on a second stream of the handle beside the fundamental chain, as the reference's two
   threads do (src/Initializer.cpp:291-303); disabled, the kernels run one after the other on the handle's stream
   (each then has the chip to itself: what a profile of single kernels wants).  Results are identical. */
orbx_status orbx_ransac_set_fork(orbx_ransac* h, int enabled);

/* Per-kernel launch durations (ms) of the last batch call: normalize, solve_f, score_f, score_h, select, solve_h
   (the homography kernels run on a side stream beside the fundamental ones, so the durations overlap). */
int orbx_ransac_last_timings(const orbx_ransac* h, float* ms, int cap);

/* Test hook (no counterpart in the reference): runs the library's own replay of cv::KeyPointsFilter::retainBest
   (std::nth_element + std::partition, the order-defining step of cv::ORB::detect) on the host.  perm_out receives
   the original indices of the kept responses in their final order. */
orbx_status orbx_selftest_retain_best(const float* keys, int n, int n_points, int32_t* perm_out, int* new_n);
/* The same on the device, by the wave-parallel form the batch extractor uses (n <= 8192). */
orbx_status orbx_selftest_retain_best_device(int device, const float* keys, int n, int n_points, int32_t* perm_out,
                                             int* new_n);

/* Test hook (no counterpart in the reference), host only: plans the one-pass pyramid kernel for an image size
   (which rows every level emits in which step, the depth of every LDS ring), checks the plan against its rings, and
   - with image and atlas non-NULL - executes the same tables on the CPU.  info[4] = steps, LDS bytes, atlas width,
   atlas height.  ORBX_UNSUPPORTED when the geometry does not suit that kernel (the extractor then builds the pyramid
   level by level).  base_level > 0 plans the tail form (orbx_extractor_set_pyramid_tail): the atlas passed in must
   already hold pyramid level base_level, the levels above it are computed from there (image may then be NULL). */
orbx_status orbx_selftest_stream_pyramid(const orbx_extractor_params* params, const uint8_t* image, int width,
                                         int height, int stride, int channels, int rows_per_step, size_t max_lds_bytes,
                                         int base_level, uint8_t* atlas, size_t atlas_capacity, int32_t* info);

/* Measurement hook (no counterpart in the reference): the vector issue rate this GPU sustains right now, in
   wave-instructions per second over the whole chip, for one instruction mix: kind 0 = v_add_u32, 1 = v_xor_b32 +
   v_bcnt_u32_b32 (the Hamming step), 2 = v_mul_f64 + v_add_f64 (un-fused double, the F scorer), 3 = v_mul_f32 +
   v_add_f32 (the H scorer).  Multiply by 64 for lane-operations per second.  Runs on the NULL stream, about 1 ms. */
orbx_status orbx_probe_issue_rate(int device, int kind, double* wave_instructions_per_second);

/* Measurement hook (no counterpart in the reference): which kernel the brute-force Hamming entry points
   (orbx_match_hamming, _batch, _cross) launch.  0 (default): the int8 matrix-core form, distance = |query| + the
   dot product of the train bits as {0, 1} with the query bits as {+1, -1}; 1: the vector form (xor + popcount).
   Same results bit for bit. */
orbx_status orbx_matcher_set_hamming_kernel(orbx_matcher* h, int kind);

/* Sets the HIP stream (hipStream_t as void*) a handle launches on; NULL restores the handle's own stream. */
orbx_status orbx_extractor_set_stream(orbx_extractor* h, void* hip_stream);
orbx_status orbx_matcher_set_stream(orbx_matcher* h, void* hip_stream);
orbx_status orbx_ransac_set_stream(orbx_ransac* h, void* hip_stream);

/* ---- sequence input (callers' side of the extraction path) ---------------------------------------------------- */

typedef struct orbx_sequence orbx_sequence;
typedef struct orbx_feeder orbx_feeder;

/* Replaces TUMRGBDCamera::setupCameraStream (src/geometry/rgbd_camera.cpp:225-252): reads dataset_dir + "assoc.txt",
   one `time rgb_file time depth_file` per non-empty line; the first time stamp of a line is kept.  An empty list is
   an error (the reference asserts). */
orbx_status orbx_sequence_open(const char* dataset_dir, orbx_sequence** out);
void orbx_sequence_close(orbx_sequence* seq);
int orbx_sequence_length(const orbx_sequence* seq);
/* The strings stay valid until orbx_sequence_close.  Any of the three outputs may be NULL. */
orbx_status orbx_sequence_entry(const orbx_sequence* seq, int index, double* time_stamp, const char** rgb_file,
                                const char** depth_file);

/* What cv::imread returns for a PNG under the two flag values the reference uses
   (include/orb_slam/geometry/rgbd_camera.h:211-213: CV_LOAD_IMAGE_UNCHANGED for the colour image; :224: the default
   IMREAD_COLOR for the depth image).  UNCHANGED: channels in BGR(A) order, 8- or 16-bit samples as stored (host byte
   order), gray+alpha expanded to four channels.  COLOR: always 3 x 8 bit, BGR; 16-bit samples keep their high byte,
   alpha is dropped, gray is replicated.  Non-interlaced PNGs of bit depth 8 or 16; tRNS chunks are ignored.
   With out == NULL only the geometry is returned. */
enum { ORBX_IMREAD_UNCHANGED = -1, ORBX_IMREAD_COLOR = 1 };
orbx_status orbx_png_decode(const uint8_t* file, size_t size, int mode, uint8_t* out, size_t out_capacity, int* width,
                            int* height, int* channels, int* bytes_per_sample);
orbx_status orbx_png_read(const char* path, int mode, uint8_t* out, size_t out_capacity, int* width, int* height,
                          int* channels, int* bytes_per_sample);
/* The depth scaling of TUMRGBDCamera::imageDepth (rgbd_camera.h:224-226): image.convertTo(CV_32F, depth_scale) on
   the 8-bit image cv::imread returned; depth_scale is 255 * 0.0002 for TUM (rgbd_camera.cpp:212). */
orbx_status orbx_depth_to_float(const uint8_t* bgr8, size_t n_values, float depth_scale, float* out);

/* Keeps batches of frames flowing into HBM while the caller processes the previous batch.  No counterpart in the
   reference (it reads one image per tracker step); this is what feeds orbx_extract_batch.  n_slots batches of
   `batch` frames (width x height x channels bytes each) exist on the device; with 3 slots one is being processed,
   one is on the bus, one is being decoded.  n_threads decode PNGs. */
orbx_status orbx_feeder_create(int device, int width, int height, int channels, int batch, int n_slots, int n_threads,
                               orbx_feeder** out);
void orbx_feeder_destroy(orbx_feeder* f);
/* Frames first .. first + count - 1 of the sequence's colour images (decoded as IMREAD_UNCHANGED; every image must
   have the feeder's geometry).  Starting a job ends the previous one. */
orbx_status orbx_feeder_start_sequence(orbx_feeder* f, const orbx_sequence* seq, int first, int count);
/* `count` decoded frames already in host memory, frame i at frames + i * frame_stride.  The range is page-locked
   for the duration of the job unless already_pinned says the caller has done so; uploads go straight from it. */
orbx_status orbx_feeder_start_memory(orbx_feeder* f, const uint8_t* frames, size_t frame_stride, int count,
                                     int already_pinned);
/* Hands out the next batch in order: *d_frames (device memory, n_frames frames back to back), n_frames == 0 at the
   end of the job.  The batch handed out by the previous call is recycled.  With a consumer stream the call does
   not block on the GPU: the stream is made to wait for the upload, and the recycled batch is not overwritten
   before the work queued on that stream so far has finished.  With consumer_stream == NULL the call waits for the
   upload on the host, and the caller must have finished with the previous batch. */
orbx_status orbx_feeder_next(orbx_feeder* f, void* consumer_stream, const uint8_t** d_frames, int* n_frames);

#ifdef __cplusplus
}
#endif
#endif /* ORBX_H */
