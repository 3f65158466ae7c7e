/*
 * CPU oracle, opt-in keypoint distribution (TEST INFRASTRUCTURE ONLY; see oracle.h).
 *
 * SELF-ORACLE: the reference has no counterpart.  Its extractor is cv::ORB (one FAST threshold, retainBest by Harris
 * response: include/orb_slam/geometry/orb_extractor.h:57-80,100-104); per-cell adaptive FAST thresholds and a quad-tree
 * distribution of the keypoints are features of other ORB-SLAM code bases.  This file DEFINES the opt-in mode of
 * orbx_extractor_set_distribution in plain sequential code, and the HIP path must reproduce it bit for bit:
 *
 *   candidates of a level = the FAST-9/16 corners at the LOW threshold after 3x3 non-max suppression and the border
 *   rule, in raster order, each with its corner score (the largest threshold at which it still is a corner);
 *   A. adaptive threshold per cell: the level's keypoint area [lo, w-lo) x [lo, h-lo) is cut into cells of about
 *      cell_size pixels; in a cell that holds at least one corner at the HIGH threshold (score >= high) only those
 *      count, in a cell without one every corner at the low threshold counts;
 *   B. quad-tree: the area starts as round(width / height) root boxes side by side (at least 1, at most 8); round by
 *      round every box with more than one corner (and more than one pixel) is cut at the ceiling of its half width and
 *      half height into up to four boxes that replace it in place, empty ones dropped; once cutting every such box
 *      could pass the level's quota N the boxes are cut largest population first (earlier box first among equals)
 *      until N boxes exist; the strongest corner of every box is kept (first in raster order among equals), in the
 *      order of the boxes.
 */
#include "oracle.h"

#include <algorithm>
#include <vector>

namespace {

struct Box {
    int x0, y0, x1, y1;
    std::vector<int> members;   /* candidate indices, ascending */
};

int ceil_half(int v) { return (v + 1) / 2; }

} // namespace

extern "C" int orc_quadtree_select(const int32_t* xs, const int32_t* ys, const int32_t* scores, int n, int w, int h,
                                   int lo, int quota, int high_threshold, int cell_size, int32_t* kept) {
    const int aw = w - 2 * lo, ah = h - 2 * lo;
    if (n <= 0 || aw <= 0 || ah <= 0 || quota <= 0) return 0;
    /* A. per-cell adaptive threshold */
    const int ncol = std::max(aw / std::max(cell_size, 1), 1), nrow = std::max(ah / std::max(cell_size, 1), 1);
    const int wc = (aw + ncol - 1) / ncol, hc = (ah + nrow - 1) / nrow;
    auto cell_of = [&](int i) {
        const int cx = std::min((xs[i] - lo) / wc, ncol - 1), cy = std::min((ys[i] - lo) / hc, nrow - 1);
        return cy * ncol + cx;
    };
    std::vector<char> strong((size_t)ncol * nrow, 0);
    for (int i = 0; i < n; i++)
        if (scores[i] >= high_threshold) strong[cell_of(i)] = 1;
    std::vector<int> active;
    for (int i = 0; i < n; i++)
        if (scores[i] >= high_threshold || !strong[cell_of(i)]) active.push_back(i);
    /* B. quad-tree */
    int n_roots = (2 * aw + ah) / (2 * ah);
    n_roots = std::min(std::max(n_roots, 1), 8);
    std::vector<Box> boxes;
    for (int r = 0; r < n_roots; r++) {
        Box b{(int)((long long)r * aw / n_roots), 0, (int)((long long)(r + 1) * aw / n_roots), ah, {}};
        for (int i : active)
            if (xs[i] - lo >= b.x0 && xs[i] - lo < b.x1) b.members.push_back(i);
        if (!b.members.empty()) boxes.push_back(b);
    }
    auto can_cut = [](const Box& b) { return b.members.size() > 1 && (b.x1 - b.x0 > 1 || b.y1 - b.y0 > 1); };
    auto cut = [&](const Box& b, std::vector<Box>& out) {
        const int xm = b.x0 + ceil_half(b.x1 - b.x0), ym = b.y0 + ceil_half(b.y1 - b.y0);
        Box q[4] = {{b.x0, b.y0, xm, ym, {}}, {xm, b.y0, b.x1, ym, {}}, {b.x0, ym, xm, b.y1, {}}, {xm, ym, b.x1, b.y1, {}}};
        for (int i : b.members) q[(xs[i] - lo >= xm ? 1 : 0) + (ys[i] - lo >= ym ? 2 : 0)].members.push_back(i);
        for (auto& c : q)
            if (!c.members.empty()) out.push_back(c);
    };
    for (int round = 0; round < 32; round++) {
        std::vector<int> cuttable;
        for (size_t j = 0; j < boxes.size(); j++)
            if (can_cut(boxes[j])) cuttable.push_back((int)j);
        if ((int)boxes.size() >= quota || cuttable.empty()) break;
        std::vector<char> do_cut(boxes.size(), 0);
        if ((int)boxes.size() + 3 * (int)cuttable.size() <= quota) {
            for (int j : cuttable) do_cut[j] = 1;
        } else {
            std::stable_sort(cuttable.begin(), cuttable.end(),
                             [&](int a, int b) { return boxes[a].members.size() > boxes[b].members.size(); });
            int total = (int)boxes.size();
            for (int j : cuttable) {
                std::vector<Box> tmp;
                cut(boxes[j], tmp);
                do_cut[j] = 1;
                total += (int)tmp.size() - 1;
                if (total >= quota) break;
            }
        }
        std::vector<Box> next;
        for (size_t j = 0; j < boxes.size(); j++) {
            if (do_cut[j]) cut(boxes[j], next);
            else next.push_back(boxes[j]);
        }
        boxes.swap(next);
    }
    int m = 0;
    for (const Box& b : boxes) {
        int best = b.members[0];
        for (int i : b.members)
            if (scores[i] > scores[best]) best = i;
        kept[m++] = best;
    }
    return m;
}
