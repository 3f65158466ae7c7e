/*
 * CPU oracle, matching half (TEST INFRASTRUCTURE ONLY; see oracle.h).
 * Follows src/geometry/orb_matcher.cpp, include/orb_slam/geometry/orb_matcher.h:45-96,
 * src/geometry/utils.cpp:184-236 and src/frame.cpp:89-218,304-367 of the reference.
 */
#include "oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

extern "C" {

/* utils.cpp:184-200 (SWAR popcount over 8 x int32) */
int orc_descriptor_distance(const uint8_t* a, const uint8_t* b) {
    int32_t pa[8], pb[8];
    std::memcpy(pa, a, 32);
    std::memcpy(pb, b, 32);
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        unsigned int v = pa[i] ^ pb[i];
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

/* orb_matcher.cpp:671-677 -> cv::BFMatcher(NORM_HAMMING, crossCheck=false)::match (k = 1 batchDistance):
   train rows scanned in index order, candidate kept only when strictly smaller. */
void orc_match_hamming(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, int32_t* dist) {
    for (int i = 0; i < nq; i++) {
        int best = 0x7fffffff, bi = -1;
        for (int j = 0; j < nt; j++) {
            int d = orc_descriptor_distance(q + (size_t)i * 32, t + (size_t)j * 32);
            if (d < best) { best = d; bi = j; }
        }
        idx[i] = bi;
        dist[i] = bi < 0 ? -1 : best;
    }
}

/* orb_matcher.cpp:890-924, same std:: calls (std::sort order among equal trainIdx is libstdc++'s). */
int orc_filter_matches(orc_match* m, int n, int n_desc_rows) {
    if (n <= 0) return 0;
    std::vector<orc_match> matches(m, m + n);
    auto min_max = std::minmax_element(matches.begin(), matches.end(),
                                       [](const orc_match& a, const orc_match& b) { return a.distance < b.distance; });
    double min_dist = min_max.first->distance;
    std::vector<orc_match> good;
    int rows = std::min(n_desc_rows, n); /* the reference indexes matches[i] for i < descriptors.rows */
    for (int i = 0; i < rows; i++)
        if (matches[i].distance <= std::max(2 * min_dist, 30.0)) good.push_back(matches[i]);
    std::sort(good.begin(), good.end(), [](const orc_match& a, const orc_match& b) { return a.train < b.train; });
    std::vector<orc_match> out;
    if (!good.empty()) out.push_back(matches[0]);
    for (size_t i = 1; i < good.size(); i++)
        if (good[i].train != good[i - 1].train) out.push_back(good[i]);
    std::copy(out.begin(), out.end(), m);
    return (int)out.size();
}

/* orb_matcher.cpp:25-74: distance is mean |a-b| over the 32 bytes, candidates within a pixel radius. */
int orc_match_radius(const orc_keypoint* kp, int n, const orc_keypoint* ref_kp, int n_ref, const uint8_t* desc,
                     const uint8_t* ref_desc, double max_pixel_dist, orc_match* out) {
    double max_sq = max_pixel_dist * max_pixel_dist;
    int count = 0;
    for (int i = 0; i < n; i++) {
        bool matched = false;
        double min_feature_dist = 99999999.0;
        int matched_id = 0;
        for (int j = 0; j < n_ref; j++) {
            float dx = ref_kp[j].x - kp[i].x;
            float dy = ref_kp[j].y - kp[i].y;
            float sqrd = dx * dx + dy * dy;
            if (sqrd <= max_sq) {
                int s = 0;
                for (int k = 0; k < 32; k++) s += std::abs((int)desc[(size_t)i * 32 + k] - (int)ref_desc[(size_t)j * 32 + k]);
                double feature_dist = (double)s / 32;
                if (feature_dist < min_feature_dist) {
                    min_feature_dist = feature_dist;
                    matched_id = j;
                    matched = true;
                }
            }
        }
        if (matched) out[count++] = orc_match{i, matched_id, -1, (float)min_feature_dist};
    }
    return count;
}

/* utils.cpp:202-236 */
void orc_three_maxima(const int* hist_sizes, int L, int* ind1, int* ind2, int* ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = hist_sizes[i];
        if (s > max1) {
            max3 = max2; max2 = max1; max1 = s;
            *ind3 = *ind2; *ind2 = *ind1; *ind1 = i;
        } else if (s > max2) {
            max3 = max2; max2 = s;
            *ind3 = *ind2; *ind2 = i;
        } else if (s > max3) {
            max3 = s;
            *ind3 = i;
        }
    }
    if (max2 < 0.1f * (float)max1) { *ind2 = -1; *ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { *ind3 = -1; }
}

/* frame.cpp:326-367 */
int orc_assign_grid(const orc_grid_params* g, orc_keypoint* kps, int n, int32_t* cell_start, int32_t* cell_items) {
    int ncell = g->grid_cols * g->grid_rows;
    std::vector<std::vector<int>> grid(ncell);
    std::vector<orc_keypoint> filt;
    int count = 0;
    for (int i = 0; i < n; i++) {
        int pos_y = (kps[i].y - g->min_y) / g->grid_size_y;
        int pos_x = (kps[i].x - g->min_x) / g->grid_size_x;
        if (pos_y < 0 || pos_y >= g->grid_rows || pos_x < 0 || pos_x >= g->grid_cols) continue;
        filt.push_back(kps[i]);
        grid[pos_x * g->grid_rows + pos_y].push_back(count++);
    }
    int k = 0;
    for (int c = 0; c < ncell; c++) {
        cell_start[c] = k;
        for (int v : grid[c]) cell_items[k++] = v;
    }
    cell_start[ncell] = k;
    std::copy(filt.begin(), filt.end(), kps);
    return count;
}

} // extern "C"

namespace {

/* frame.cpp:89-118 (returns true at the end; the reference falls off the end of a non-void function) */
bool box_around(const orc_grid_params* g, float px, float py, float r, int& left, int& right, int& up, int& down) {
    left = std::max(0, (int)floor(px - g->min_x - r) / g->grid_size_x);
    if (left >= g->grid_cols) return false;
    right = std::min(g->grid_cols - 1, (int)ceil(px - g->min_x + r) / g->grid_size_x);
    if (right < 0) return false;
    up = std::max(0, (int)floor(py - g->min_y - r) / g->grid_size_y);
    if (up >= g->grid_rows) return false;
    down = std::min(g->grid_rows - 1, (int)ceil(py - g->min_y + r) / g->grid_size_y);
    if (down < 0) return false;
    return true;
}

/* frame.cpp:151-183 (4-argument overload: compares keypoint coordinates with the CELL index) and
   frame.cpp:185-218 (5-argument overload). min-only mode selects the former. */
/* A grid built before cv::ORB::compute dropped border keypoints (mono_frame.cpp:97-103) names slots past the end of
   the shortened vector; the reference reads beyond its vector there.  Defined here as: such an entry names no
   feature (n = number of keypoints). */
bool features_around(const orc_grid_params* g, const int32_t* cell_start, const int32_t* cell_items,
                     const orc_keypoint* kps, int n, float px, float py, float radius, int min_level, int max_level,
                     bool min_only, std::vector<int>& out) {
    int left, right, up, down;
    if (!box_around(g, px, py, radius, left, right, up, down)) return false;
    for (int x = left; x <= right; ++x)
        for (int y = up; y <= down; ++y) {
            int c = x * g->grid_rows + y;
            for (int i = cell_start[c]; i < cell_start[c + 1]; i++) {
                if (cell_items[i] >= n) continue;
                const orc_keypoint& k = kps[cell_items[i]];
                if (min_only) {
                    if (k.octave < min_level) continue;
                    float dx = fabsf(k.x - x), dy = fabsf(k.y - y);
                    if (dx < radius && dy < radius) out.push_back(cell_items[i]);
                } else {
                    if (k.octave < min_level || k.octave > max_level) continue;
                    float dx = fabsf(k.x - px), dy = fabsf(k.y - py);
                    if (dx < radius && dy < radius) out.push_back(cell_items[i]);
                }
            }
        }
    return true;
}

bool within_bounds(const orc_grid_params* g, float x, float y) {
    return x > g->min_x && x < g->max_x && y > g->min_y && y < g->max_y; /* camera.cpp:164-172 */
}

int create_matches(const std::vector<int>& matched, const std::vector<int>& dists, orc_match* out) {
    int c = 0; /* orb_matcher.h:81-93 */
    for (size_t i = 0; i < matched.size(); ++i)
        if (matched[i] > 0) out[c++] = orc_match{(int)i, matched[i], -1, (float)dists[i]};
    return c;
}

} // namespace

extern "C" {

/* orb_matcher.cpp:115-255 */
int orc_match_projection_frame(const orc_grid_params* g, const int32_t* cell_start, const int32_t* cell_items,
                               const orc_keypoint* kps, int n, const uint8_t* desc, const orc_keypoint* ref_kps,
                               int n_ref, const uint8_t* ref_has_mp, const float* proj_xy, const float* proj_z,
                               const uint8_t* ref_desc, const float* scale_factors, int level_mode, float radius,
                               int check_orientation, orc_match* out) {
    const int hist_length = 30, high_threshold = 100;
    std::vector<int> matched(n, -1), feature_dists(n, -1);
    std::vector<std::vector<int>> rot_hist(hist_length);
    const float factor = 1.0f / hist_length;
    for (int ref_idx = 0; ref_idx < n_ref; ++ref_idx) {
        if (!ref_has_mp[ref_idx]) continue;
        if (proj_z[ref_idx] < 0) continue;
        float px = proj_xy[2 * ref_idx], py = proj_xy[2 * ref_idx + 1];
        if (!within_bounds(g, px, py)) continue;
        int level = ref_kps[ref_idx].octave;
        float radius_scaled = radius * scale_factors[level];
        std::vector<int> close;
        bool ok;
        if (level_mode == 1) ok = features_around(g, cell_start, cell_items, kps, n, px, py, radius_scaled, level, 0, true, close);
        else if (level_mode == 2) ok = features_around(g, cell_start, cell_items, kps, n, px, py, radius_scaled, 0, level, false, close);
        else ok = features_around(g, cell_start, cell_items, kps, n, px, py, radius_scaled, level - 1, level + 1, false, close);
        if (!ok) continue;
        int min_dist = 256, matched_id = -1;
        for (int idx : close) {
            if (matched[idx] > 0) continue;
            int d = orc_descriptor_distance(ref_desc + (size_t)ref_idx * 32, desc + (size_t)idx * 32);
            if (d <= min_dist) { min_dist = d; matched_id = idx; }
        }
        if (min_dist <= high_threshold && matched_id >= 0) {
            matched[matched_id] = ref_idx;
            feature_dists[matched_id] = min_dist;
            if (check_orientation) {
                float rot_diff = ref_kps[ref_idx].angle - kps[matched_id].angle;
                if (rot_diff < 0.0) rot_diff += 360.0f;
                int bin = round(rot_diff * factor);
                if (bin == hist_length) bin = 0;
                rot_hist[bin].push_back(matched_id);
            }
        }
    }
    if (check_orientation) {
        int sizes[30], i1 = -1, i2 = -1, i3 = -1;
        for (int i = 0; i < hist_length; i++) sizes[i] = (int)rot_hist[i].size();
        orc_three_maxima(sizes, hist_length, &i1, &i2, &i3);
        for (int i = 0; i < hist_length; i++)
            if (i != i1 && i != i2 && i != i3)
                for (int v : rot_hist[i]) matched[v] = 0; /* orb_matcher.h:72-77 writes `false` */
    }
    return create_matches(matched, feature_dists, out);
}

/* orb_matcher.cpp:257-352.  The reference reads key_points[matched_id_2] with matched_id_2 == -1 while only
   one candidate has been seen (out-of-bounds); here that state counts as "octaves differ", i.e. assign. */
int orc_match_projection_points(const orc_grid_params* g, const int32_t* cell_start, const int32_t* cell_items,
                                const orc_keypoint* kps, int n, const uint8_t* desc, int n_pts,
                                const uint8_t* in_view, const float* proj_xy, const int32_t* pred_level,
                                const float* view_cos, const uint8_t* pt_desc, const float* scale_factors,
                                float radius, float nn_ratio, orc_match* out) {
    const int high_threshold = 100;
    std::vector<int> matched(n, -1), feature_dists(n, -1);
    for (int ref_idx = 0; ref_idx < n_pts; ++ref_idx) {
        if (!in_view[ref_idx]) continue;
        int level = pred_level[ref_idx];
        float by_cos = view_cos[ref_idx] > 0.998 ? 2.5f : 4.0f; /* orb_matcher.h:50-55 */
        float radius_scaled = radius * by_cos * scale_factors[level];
        std::vector<int> close;
        if (!features_around(g, cell_start, cell_items, kps, n, proj_xy[2 * ref_idx], proj_xy[2 * ref_idx + 1],
                             radius_scaled, level - 1, level, false, close))
            continue;
        int min_dist = 256, id1 = -1, min2 = 256, id2 = -1;
        for (int idx : close) {
            if (matched[idx] > 0) continue;
            int d = orc_descriptor_distance(pt_desc + (size_t)ref_idx * 32, desc + (size_t)idx * 32);
            if (d < min_dist) { min2 = min_dist; id2 = id1; min_dist = d; id1 = idx; }
            else if (d < min2) { min2 = d; id2 = idx; }
            if (min_dist <= high_threshold) {
                if (id2 >= 0 && kps[id1].octave == kps[id2].octave && min_dist > nn_ratio * min2) continue;
                matched[id1] = ref_idx;
                feature_dists[id1] = min_dist;
            }
        }
    }
    return create_matches(matched, feature_dists, out);
}

/* BowOrbMatcher::match, orb_matcher.cpp:386-497.  The two DBoW2::FeatureVector maps (node id -> feature indices,
   ascending node id) arrive as CSR arrays; the lower_bound jumps of the reference visit exactly the common nodes in
   ascending order.  ref_has_mp[i] = "reference feature i has a map point that is not bad". */
int orc_match_bow(const orc_keypoint* kps, int n, const uint8_t* desc, const orc_keypoint* ref_kps, int n_ref,
                  const uint8_t* ref_desc, const uint8_t* ref_has_mp, const uint32_t* node_ids,
                  const int32_t* node_start, const int32_t* node_items, int n_nodes, const uint32_t* ref_node_ids,
                  const int32_t* ref_node_start, const int32_t* ref_node_items, int n_ref_nodes, int low_threshold,
                  float nn_ratio, int check_orientation, orc_match* out) {
    (void)n_ref;
    const int hist_length = 30;
    std::vector<int> matched(n, -1), feature_dists(n, -1);
    std::vector<std::vector<int>> rot_hist(hist_length);
    const float factor = 1.0f / hist_length;
    int f = 0, rf = 0;
    while (rf < n_ref_nodes && f < n_nodes) {
        if (ref_node_ids[rf] == node_ids[f]) {
            for (int a = ref_node_start[rf]; a < ref_node_start[rf + 1]; ++a) {
                const int ref_idx = ref_node_items[a];
                if (!ref_has_mp[ref_idx]) continue;
                int min_dist = 256, min_dist_2 = 256, matched_idx = -1;
                for (int b = node_start[f]; b < node_start[f + 1]; ++b) {
                    const int idx = node_items[b];
                    if (matched[idx] > 0) continue;
                    const int d = orc_descriptor_distance(ref_desc + (size_t)ref_idx * 32, desc + (size_t)idx * 32);
                    if (d <= min_dist) { min_dist = d; matched_idx = idx; }
                    else if (d < min_dist_2) min_dist_2 = d;
                }
                if (min_dist <= low_threshold && (float)min_dist < nn_ratio * (float)min_dist_2) {
                    matched[matched_idx] = ref_idx;
                    feature_dists[matched_idx] = min_dist;
                    if (check_orientation) {
                        float rot_diff = ref_kps[ref_idx].angle - kps[matched_idx].angle;
                        if (rot_diff < 0.0) rot_diff += 360.0f;
                        int bin = round(rot_diff * factor);
                        if (bin == hist_length) bin = 0;
                        rot_hist[bin].push_back(matched_idx);
                    }
                }
            }
            f++;
            rf++;
        } else if (ref_node_ids[rf] < node_ids[f]) {
            while (rf < n_ref_nodes && ref_node_ids[rf] < node_ids[f]) rf++; /* lower_bound */
        } else {
            while (f < n_nodes && node_ids[f] < ref_node_ids[rf]) f++;
        }
    }
    if (check_orientation) {
        int sizes[30], i1 = -1, i2 = -1, i3 = -1;
        for (int i = 0; i < hist_length; i++) sizes[i] = (int)rot_hist[i].size();
        orc_three_maxima(sizes, hist_length, &i1, &i2, &i3);
        for (int i = 0; i < hist_length; i++)
            if (i != i1 && i != i2 && i != i3)
                for (int v : rot_hist[i]) matched[v] = 0;
    }
    return create_matches(matched, feature_dists, out);
}

/* EpipolarConstraintWithBowMatcher::match + check_epipolar_dist, orb_matcher.cpp:520-661, as written:
   - the descriptor distance of the inner loop is taken between row `idx` of BOTH descriptor matrices (:573), so it
     does not depend on the candidate, and with `dist > min_feature_dist` (:576) the LAST candidate of the node that
     passes the epipolar test is the one kept;
   - the "already matched" test reads matched[ref_idx] (:565), an array indexed by the features of `frame`.
   Both reads need idx < n_ref and ref_idx < n for every index of a common node; -1 is returned otherwise.
   has_mp[i] / ref_has_mp[i] = the (ref) key frame's feature i already has a map point (any, bad or not). */
int orc_match_epipolar(const orc_keypoint* kps, int n, const uint8_t* desc, const orc_keypoint* ref_kps, int n_ref,
                       const uint8_t* ref_desc, const uint8_t* has_mp, const uint8_t* ref_has_mp, const float* F9,
                       const float* scale_sigmas, const uint32_t* node_ids, const int32_t* node_start,
                       const int32_t* node_items, int n_nodes, const uint32_t* ref_node_ids,
                       const int32_t* ref_node_start, const int32_t* ref_node_items, int n_ref_nodes,
                       int low_threshold, int check_orientation, orc_match* out) {
    const int hist_length = 30;
    std::vector<int> matched(n, -1), feature_dists(n, -1);
    std::vector<std::vector<int>> rot_hist(hist_length);
    const float factor = 1.0f / hist_length;
    auto epipolar_ok = [&](const orc_keypoint& kp1, const orc_keypoint& kp2) { /* :628-661 */
        const float a = kp1.x * F9[0] + kp1.y * F9[3] + F9[6];
        const float b = kp1.x * F9[1] + kp1.y * F9[4] + F9[7];
        const float c = kp1.x * F9[2] + kp1.y * F9[5] + F9[8];
        const float num = a * kp2.x + b * kp2.y + c;
        const float den = a * a + b * b;
        if (den == 0) return false;
        const float dsqr = num * num / den;
        return dsqr < 3.84 * scale_sigmas[kp2.octave];
    };
    int f = 0, rf = 0;
    while (rf < n_ref_nodes && f < n_nodes) {
        if (ref_node_ids[rf] == node_ids[f]) {
            for (int b = node_start[f]; b < node_start[f + 1]; ++b) {
                const int idx = node_items[b];
                if (has_mp[idx]) continue;
                if (idx >= n_ref) return -1;
                int min_feature_dist = low_threshold, matched_id = -1;
                for (int a = ref_node_start[rf]; a < ref_node_start[rf + 1]; ++a) {
                    const int ref_idx = ref_node_items[a];
                    if (ref_idx >= n) return -1;
                    if (matched[ref_idx] > 0 || ref_has_mp[ref_idx]) continue;
                    const int dist = orc_descriptor_distance(ref_desc + (size_t)idx * 32, desc + (size_t)idx * 32);
                    if (dist > low_threshold || dist > min_feature_dist) continue;
                    if (epipolar_ok(kps[idx], ref_kps[ref_idx])) {
                        matched_id = ref_idx;
                        min_feature_dist = dist;
                    }
                }
                if (matched_id >= 0) {
                    matched[idx] = matched_id;
                    feature_dists[idx] = min_feature_dist;
                    if (check_orientation) {
                        float rot_diff = kps[idx].angle - ref_kps[matched_id].angle;
                        if (rot_diff < 0.0) rot_diff += 360.0f;
                        int bin = round(rot_diff * factor);
                        if (bin == hist_length) bin = 0;
                        rot_hist[bin].push_back(idx);
                    }
                }
            }
            f++;
            rf++;
        } else if (ref_node_ids[rf] < node_ids[f]) {
            while (rf < n_ref_nodes && ref_node_ids[rf] < node_ids[f]) rf++; /* lower_bound */
        } else {
            while (f < n_nodes && node_ids[f] < ref_node_ids[rf]) f++;
        }
    }
    if (check_orientation) {
        int sizes[30], i1 = -1, i2 = -1, i3 = -1;
        for (int i = 0; i < hist_length; i++) sizes[i] = (int)rot_hist[i].size();
        orc_three_maxima(sizes, hist_length, &i1, &i2, &i3);
        for (int i = 0; i < hist_length; i++)
            if (i != i1 && i != i2 && i != i3)
                for (int v : rot_hist[i]) matched[v] = 0;
    }
    return create_matches(matched, feature_dists, out);
}

/* The search of ORBMatcher::fuse, orb_matcher.cpp:806-867, for points whose frustum test (:808-814) the caller has
   done: per point the feature of the key frame with the smallest descriptor distance (`<`, first in scan order)
   inside the window of radius radius_multiplier * scale_factor[level] on levels [level-1, level] whose squared
   reprojection error passes e2 * inv_sigma2[octave] <= 5.99.  `matched` of the reference is never written inside
   fuse, so the points are independent.  best_idx = -1 / best_dist = 256 when nothing qualifies; the caller applies
   `best_dist <= low_threshold` and the map bookkeeping (:869-884). */
void orc_fuse_search(const orc_grid_params* g, const int32_t* cell_start, const int32_t* cell_items,
                     const orc_keypoint* kps, const uint8_t* desc, int n_pts, const uint8_t* in_view,
                     const float* proj_xy, const int32_t* pred_level, const uint8_t* pt_desc,
                     const float* scale_factors, const float* inv_scale_sigmas, float radius_multiplier,
                     int32_t* best_idx, int32_t* best_dist) {
    const int n = 0x7fffffff; /* the grid of a key frame is built from the very keypoints passed here */
    for (int p = 0; p < n_pts; ++p) {
        best_idx[p] = -1;
        best_dist[p] = 256;
        if (!in_view[p]) continue;
        const int level = pred_level[p];
        const float px = proj_xy[2 * p], py = proj_xy[2 * p + 1];
        const float radius_scaled = radius_multiplier * scale_factors[level];
        std::vector<int> close;
        if (!features_around(g, cell_start, cell_items, kps, n, px, py, radius_scaled, level - 1, level, false, close))
            continue;
        int min_feature_dist = 256, matched_id = -1;
        for (int idx : close) {
            const orc_keypoint& k = kps[idx];
            if (k.octave < level - 1 || k.octave > level) continue;
            const float ex = px - k.x, ey = py - k.y;
            const float e_sqrd = ex * ex + ey * ey;
            if (e_sqrd * inv_scale_sigmas[k.octave] > 5.99) continue;
            const int dist = orc_descriptor_distance(pt_desc + (size_t)p * 32, desc + (size_t)idx * 32);
            if (dist < min_feature_dist) { min_feature_dist = dist; matched_id = idx; }
        }
        best_idx[p] = matched_id;
        best_dist[p] = min_feature_dist;
    }
}

/* MapPoint::computeBestDescriptor, map_point.cpp:109-166, for n_points descriptor sets laid out back to back
   (point p owns rows point_start[p] .. point_start[p+1]).  best[p] = row inside the set, -1 for an empty set. */
void orc_best_descriptor(const uint8_t* desc, const int32_t* point_start, int n_points, int32_t* best) {
    for (int p = 0; p < n_points; p++) {
        const uint8_t* d = desc + (size_t)point_start[p] * 32;
        const int n = point_start[p + 1] - point_start[p];
        if (n <= 0) { best[p] = -1; continue; }
        std::vector<std::vector<float>> distances(n, std::vector<float>(n, 0.f));
        for (int i = 0; i < n; i++)
            for (int j = i + 1; j < n; j++) {
                int dist = orc_descriptor_distance(d + (size_t)i * 32, d + (size_t)j * 32);
                distances[i][j] = (float)dist;
                distances[j][i] = (float)dist;
            }
        int best_median = 2147483647, best_idx = 0;
        for (int i = 0; i < n; i++) {
            std::vector<int> dists(distances[i].begin(), distances[i].end());
            std::sort(dists.begin(), dists.end());
            int median = dists[(size_t)(0.5 * (n - 1))];
            if (median < best_median) { best_median = median; best_idx = i; }
        }
        best[p] = best_idx;
    }
}

} // extern "C"
