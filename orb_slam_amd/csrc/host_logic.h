// Host-side pieces of the path whose result ORDER is defined by libstdc++ algorithms in the reference
// (std::nth_element / std::partition inside cv::KeyPointsFilter::retainBest, std::sort in
// ORBMatcher::filterMatches, std::shuffle in Initializer): they run on the host with the same std:: calls.
#pragma once

#include <vector>

#include "extract_types.h"
#include "orbx.h"

namespace orbx {
namespace host {

// cv::KeyPointsFilter::retainBest (features2d keypoint.cpp).
void retain_best(std::vector<orbx_keypoint>& k, int n_points);

// cv::KeyPointsFilter::runByImageBorder; the rectangle test is on the rounded coordinates.
void run_by_image_border(std::vector<orbx_keypoint>& k, int w, int h, int border);

// cv::ORB detectAndCompute(useProvidedKeypoints), regrouping when the octaves are not ascending.
void regroup_by_level(std::vector<orbx_keypoint>& k, int nlevels);

// Host half of cv::ORB computeKeyPoints() for one frame: cands = level-major raster-ordered corners with FAST
// score, Harris response and angle; applies retainBest(2*quota) on the FAST score, retainBest(quota) on the
// Harris response, and scales the coordinates to level 0.
void select_keypoints(const Cand* cands, int n, const int* level_counts, int nlevels, const int* quotas,
                      const float* scales, int patch_size, std::vector<orbx_keypoint>& out);

// ORBExtractor's own tables (reference include/orb_slam/geometry/orb_extractor.h:39-54).
void scale_tables(float scale_factor, int nlevels, float* sf, float* isf, float* ss, float* iss, float* log_sf);

// cv::undistortPoints with P = K as Camera::undistortPoints calls it (reference src/geometry/camera.cpp:97-126):
// 5 fixed-point iterations of the inverse Brown model in double, re-projected with the same intrinsics.
void undistort_points(const orbx_camera& c, const float* xy_in, int n, float* xy_out);

}  // namespace host
}  // namespace orbx
