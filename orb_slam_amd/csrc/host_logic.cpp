// Host-side logic of the orbx library (see host_logic.h).  Plain C++; no device code.
#include "host_logic.h"
#include "pyramid_stream.h"
#include "stdsel.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <random>

namespace orbx {
namespace host {

void retain_best(std::vector<orbx_keypoint>& k, int n_points) {
    if (n_points >= 0 && k.size() > (size_t)n_points) {
        if (n_points == 0) {
            k.clear();
            return;
        }
        std::nth_element(k.begin(), k.begin() + n_points - 1, k.end(),
                         [](const orbx_keypoint& a, const orbx_keypoint& b) { return a.response > b.response; });
        const float ambiguous = k[n_points - 1].response;
        auto new_end = std::partition(k.begin() + n_points, k.end(),
                                      [ambiguous](const orbx_keypoint& p) { return p.response >= ambiguous; });
        k.resize(new_end - k.begin());
    }
}

void run_by_image_border(std::vector<orbx_keypoint>& k, int w, int h, int border) {
    if (border <= 0) return;
    if (h <= border * 2 || w <= border * 2) {
        k.clear();
        return;
    }
    const int x0 = border, y0 = border, x1 = w - border, y1 = h - border;
    k.erase(std::remove_if(k.begin(), k.end(),
                           [=](const orbx_keypoint& p) {
                               // cv::Point2f -> cv::Point conversion rounds to nearest (ties to even)
                               const int px = (int)lrintf(p.x), py = (int)lrintf(p.y);
                               return !(x0 <= px && px < x1 && y0 <= py && py < y1);
                           }),
            k.end());
}

void regroup_by_level(std::vector<orbx_keypoint>& k, int nlevels) {
    std::vector<std::vector<orbx_keypoint>> by_level(nlevels);
    for (const auto& q : k) by_level[q.octave].push_back(q);
    k.clear();
    for (const auto& v : by_level) k.insert(k.end(), v.begin(), v.end());
}

void select_keypoints(const Cand* cands, int n, const int* level_counts, int nlevels, const int* quotas,
                      const float* scales, int patch_size, std::vector<orbx_keypoint>& out) {
    out.clear();
    std::vector<orbx_keypoint> kps;
    int offset = 0;
    for (int level = 0; level < nlevels; level++) {
        const int cnt = level_counts[level];
        kps.resize(cnt);
        for (int i = 0; i < cnt; i++) {
            const Cand& c = cands[offset + i];
            // cv::FAST emits KeyPoint(x, y, 7.f, -1, score); class_id carries the candidate index meanwhile
            kps[i] = orbx_keypoint{(float)c.x, (float)c.y, 7.f, -1.f, (float)c.score, 0, offset + i};
        }
        offset += cnt;
        retain_best(kps, 2 * quotas[level]);
        const float sf = scales[level];
        for (auto& k : kps) {
            k.octave = level;
            k.size = patch_size * sf;
            k.response = cands[k.class_id].harris;
        }
        retain_best(kps, quotas[level]);
        for (auto& k : kps) {
            k.angle = cands[k.class_id].angle;
            k.class_id = -1;
            k.x *= sf;
            k.y *= sf;
        }
        out.insert(out.end(), kps.begin(), kps.end());
    }
    (void)n;
}

void scale_tables(float scale_factor, int nlevels, float* sf, float* isf, float* ss, float* iss, float* log_sf) {
    std::vector<float> a(nlevels), b(nlevels), c(nlevels), d(nlevels);
    a[0] = 1.0f;
    c[0] = 1.0f;
    b[0] = 1.0f / a[0];
    d[0] = 1.0f / c[0];
    for (int i = 1; i < nlevels; ++i) {
        a[i] = a[i - 1] * scale_factor;
        b[i] = 1.0f / a[i];
        c[i] = a[i] * a[i];
        d[i] = 1.0f / c[i];
    }
    if (sf) std::memcpy(sf, a.data(), sizeof(float) * nlevels);
    if (isf) std::memcpy(isf, b.data(), sizeof(float) * nlevels);
    if (ss) std::memcpy(ss, c.data(), sizeof(float) * nlevels);
    if (iss) std::memcpy(iss, d.data(), sizeof(float) * nlevels);
    if (log_sf) *log_sf = (float)std::log((double)scale_factor);
}

void undistort_points(const orbx_camera& c, const float* xy_in, int n, float* xy_out) {
    const double ifx = 1. / c.fx, ify = 1. / c.fy;
    for (int i = 0; i < n; i++) {
        double x = xy_in[2 * i], y = xy_in[2 * i + 1];
        x = (x - c.cx) * ifx;
        y = (y - c.cy) * ify;
        const double x0 = x, y0 = y;
        for (int j = 0; j < 5; j++) {
            const double r2 = x * x + y * y;
            const double icdist = 1. / (1 + ((c.k3 * r2 + c.k2) * r2 + c.k1) * r2);
            const double dx = 2 * c.p1 * x * y + c.p2 * (r2 + 2 * x * x);
            const double dy = c.p1 * (r2 + 2 * y * y) + 2 * c.p2 * x * y;
            x = (x0 - dx) * icdist;
            y = (y0 - dy) * icdist;
        }
        xy_out[2 * i] = (float)(x * c.fx + c.cx);
        xy_out[2 * i + 1] = (float)(y * c.fy + c.cy);
    }
}

}  // namespace host
}  // namespace orbx

extern "C" {

// ORBMatcher::filterMatches (reference src/geometry/orb_matcher.cpp:890-924).
orbx_status orbx_filter_matches(orbx_match* matches, int* count, int n_descriptor_rows) {
    if (!count || *count < 0 || (*count > 0 && !matches)) return ORBX_BAD_ARG;
    const int n = *count;
    if (n == 0) return ORBX_OK;
    std::vector<orbx_match> m(matches, matches + n);
    auto mm = std::minmax_element(m.begin(), m.end(),
                                  [](const orbx_match& a, const orbx_match& b) { return a.distance < b.distance; });
    const double min_dist = mm.first->distance;
    std::vector<orbx_match> good;
    const int rows = std::min(n_descriptor_rows, n);  // the reference reads matches[i] for i < descriptors.rows
    for (int i = 0; i < rows; i++)
        if (m[i].distance <= std::max(2 * min_dist, 30.0)) good.push_back(m[i]);
    std::sort(good.begin(), good.end(), [](const orbx_match& a, const orbx_match& b) { return a.train < b.train; });
    std::vector<orbx_match> out;
    if (!good.empty()) out.push_back(m[0]);
    for (size_t i = 1; i < good.size(); i++)
        if (good[i].train != good[i - 1].train) out.push_back(good[i]);
    std::copy(out.begin(), out.end(), matches);
    *count = (int)out.size();
    return ORBX_OK;
}

// Host build of the device-side retainBest replay (stdsel.h), exposed so that the CPU tests can compare it with
// the real std::nth_element / std::partition.
orbx_status orbx_selftest_retain_best(const float* keys, int n, int n_points, int32_t* perm_out, int* new_n) {
    if (n < 0 || !new_n || (n > 0 && (!keys || !perm_out))) return ORBX_BAD_ARG;
    std::vector<orbx::stdsel::Elem> a(n);
    for (int i = 0; i < n; i++) a[i] = orbx::stdsel::Elem{keys[i], i};
    const int m = orbx::stdsel::retain_best(a.data(), n, n_points);
    for (int i = 0; i < m; i++) perm_out[i] = a[i].idx;
    *new_n = m;
    return ORBX_OK;
}

orbx_status orbx_selftest_stream_pyramid(const orbx_extractor_params* params, const uint8_t* image, int width,
                                         int height, int stride, int channels, int rows_per_step, size_t max_lds_bytes,
                                         int base_level, uint8_t* atlas, size_t atlas_capacity, int32_t* info) {
    if (!params || !info || width <= 0 || height <= 0 || (channels != 1 && channels != 3)) return ORBX_BAD_ARG;
    orbx::HostGeometry hg;
    orbx_status st = orbx::compute_geometry(*params, width, height, params->level_pyramid, hg);
    if (st != ORBX_OK) return st;
    if (base_level < 0 || base_level >= hg.g.nlevels) return ORBX_BAD_ARG;
    orbx::StreamPlan sp;
    info[2] = hg.g.atlas_w;
    info[3] = hg.g.atlas_h;
    const orbx::LevelInfo& base = hg.g.L[base_level];
    if (!orbx::build_stream_plan(hg, base.w, base.h, rows_per_step, max_lds_bytes, sp, base_level)) return ORBX_UNSUPPORTED;
    info[0] = sp.prm.n_steps;
    info[1] = (int32_t)sp.lds_bytes;
    if (!atlas || (base_level == 0 && !image)) return ORBX_OK;
    if (atlas_capacity < hg.atlas_bytes || (base_level == 0 && stride < width * channels)) return ORBX_CAPACITY;
    // an access outside the atlas or the rings would be a fault on the device: report it as such
    const bool inside = base_level == 0
        ? orbx::emulate_stream_pyramid(sp, image, stride, channels, atlas, hg.g.atlas_h)
        : orbx::emulate_stream_pyramid(sp, atlas + (size_t)base.y * hg.g.atlas_w + base.x, hg.g.atlas_w, 1, atlas,
                                       hg.g.atlas_h, true);
    return inside ? ORBX_OK : ORBX_DEVICE;
}

orbx_status orbx_undistort_points(const orbx_camera* camera, const float* xy_in, int n, float* xy_out) {
    if (!camera || n < 0 || (n > 0 && (!xy_in || !xy_out))) return ORBX_BAD_ARG;
    orbx::host::undistort_points(*camera, xy_in, n, xy_out);
    return ORBX_OK;
}

// Frame::assignFeaturesToGrid / pointInGrid (reference src/frame.cpp:326-367).
orbx_status orbx_assign_grid(const orbx_grid* g, orbx_keypoint* keypoints, int* count, int32_t* cell_start,
                             int32_t* cell_items) {
    if (!g || !count || *count < 0 || !cell_start || (*count > 0 && (!keypoints || !cell_items))) return ORBX_BAD_ARG;
    if (g->grid_cols <= 0 || g->grid_rows <= 0 || g->grid_size_x <= 0 || g->grid_size_y <= 0) return ORBX_BAD_ARG;
    const int n = *count, ncell = g->grid_cols * g->grid_rows;
    std::vector<int> cell_of(n, -1);
    std::vector<int> fill(ncell + 1, 0);
    int kept = 0;
    for (int i = 0; i < n; i++) {
        const int pos_y = (keypoints[i].y - g->min_y) / g->grid_size_y;
        const int pos_x = (keypoints[i].x - g->min_x) / g->grid_size_x;
        if (pos_y < 0 || pos_y >= g->grid_rows || pos_x < 0 || pos_x >= g->grid_cols) continue;
        cell_of[i] = pos_x * g->grid_rows + pos_y;
        fill[cell_of[i] + 1]++;
        kept++;
    }
    for (int c = 0; c < ncell; c++) fill[c + 1] += fill[c];
    for (int c = 0; c <= ncell; c++) cell_start[c] = fill[c];
    int k = 0;
    for (int i = 0; i < n; i++) {
        if (cell_of[i] < 0) continue;
        cell_items[fill[cell_of[i]]++] = k;
        keypoints[k++] = keypoints[i];
    }
    *count = kept;
    return ORBX_OK;
}

// Initializer::findFundamentalAndHomography, set generation (reference src/Initializer.cpp:261-277).
orbx_status orbx_ransac_make_sets(int n, int iterations, uint32_t seed0, uint32_t* sets8) {
    if (n < 8 || iterations < 0 || (iterations > 0 && !sets8)) return ORBX_BAD_ARG;
    std::vector<size_t> all(n);
    for (int i = 0; i < n; ++i) all[i] = i;
    for (int it = 0; it < iterations; it++) {
        std::shuffle(all.begin(), all.end(), std::mt19937{seed0 + (uint32_t)it});
        for (int j = 0; j < 8; j++) sets8[it * 8 + j] = (uint32_t)all[j];
    }
    return ORBX_OK;
}

}  // extern "C"
