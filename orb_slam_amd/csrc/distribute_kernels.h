// Opt-in keypoint distribution (gfx950): per-cell adaptive FAST threshold + quad-tree selection, one workgroup per
// (frame, pyramid level).  The reference has no counterpart (its extractor is cv::ORB: one threshold, retainBest by
// Harris response, include/orb_slam/geometry/orb_extractor.h:57-80,100-104); the rule is this library's own, stated in
// include/orbx.h at orbx_extractor_set_distribution, and every test compares this kernel with a sequential
// restatement of that rule.
//
// Input: the level's candidates in raster order (FAST corners at the LOW threshold after non-max suppression and the
// border rule, with their corner scores) - what the emission kernels leave in the candidate list.
//   A. a cell of the level's keypoint area that holds a corner at the HIGH threshold keeps only those, a cell without
//      one keeps every corner at the low threshold (one bit per cell in LDS, set by the strong corners);
//   B. the boxes of the quad-tree live in LDS as (x range, y range, population); a round counts every corner into
//      the quadrant of its box (LDS atomics), decides which boxes are cut (all of them while that cannot pass the
//      quota; otherwise largest population first until the quota is reached: ranks by comparison counting, the cut-off
//      by a serial walk over the ranked gains), numbers the new boxes by a prefix sum in list order - empty quadrants
//      get no number, which is the compaction - and re-labels every corner with its new box;
//   C. the strongest corner per box (a packed atomic maximum: score, then the lower index) goes to the level's
//      output list in box order.
// A corner's box label lives in a global work array, 16 bits per candidate; a thread always visits the same
// candidates (index = thread + 256 k), so a label is only ever read by the thread that wrote it.
#pragma once

#include "extract_types.h"

namespace orbx {

struct DistributeParams {
    int high_threshold;   // a corner with score >= this is a corner at the high threshold
    int cell_size;        // side of an adaptive-threshold cell, about
    int max_nodes;        // LDS slots per box array: largest level quota + 4
    int cell_words;       // LDS words of the cell bit set: enough for the level with the most cells
    int out_stride;       // ints per (frame, level) in the output list
};

constexpr int kDistributeThreads = 256;
constexpr int kDistributeMaxRoots = 8;

__global__ __launch_bounds__(kDistributeThreads) void k_distribute_level(
    const Cand* __restrict__ cand, int cand_cap, const int* __restrict__ level_counts, const int* __restrict__ totals,
    Geom g, DistributeParams dp, uint16_t* __restrict__ label_ws, int* __restrict__ out_idx, int* __restrict__ out_n) {
    extern __shared__ int s_dyn[];
    __shared__ int s_m, s_cuttable, s_new_m;
    const int lvl = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int NM = dp.max_nodes;
    // box arrays, two generations: x range (x0 | x1 << 16), y range, population
    int* s_bx = s_dyn;                 // [2][NM]
    int* s_by = s_bx + 2 * NM;         // [2][NM]
    int* s_cnt = s_by + 2 * NM;        // [2][NM]
    int* s_cc = s_cnt + 2 * NM;        // [4 * NM] population of the four quadrants of every box
    int* s_pos = s_cc + 4 * NM;        // [NM] number of a box's first successor in the next generation
    int* s_cut = s_pos + NM;           // [NM] 1: the box is cut in this round
    int* s_ord = s_cut + NM;           // [NM] boxes that can be cut, largest population first
    int* s_strong = s_ord + NM;        // [cell_words] bit per cell: holds a corner at the high threshold
    unsigned* s_best = (unsigned*)s_cc;  // [NM] at the end: (score << 24) | (0xffffff - index) of the best corner

    int* my_n = out_n + (size_t)b * kMaxLevels + lvl;
    int* my_out = out_idx + ((size_t)b * kMaxLevels + lvl) * dp.out_stride;
    const LevelInfo L = g.L[lvl];
    const int total = totals[b];
    int n = level_counts[b * kMaxLevels + lvl], off = 0;
    for (int j = 0; j < lvl; j++) off += level_counts[b * kMaxLevels + j];
    const int lo = max(3, g.edge);
    const int aw = L.w - 2 * lo, ah = L.h - 2 * lo;
    const int quota = min(L.quota, NM - 4);
    // a candidate list that overflowed (the selection kernel reports it), an empty level, or nothing wanted
    if (total > cand_cap || off + n > cand_cap || n <= 0 || aw <= 0 || ah <= 0 || quota <= 0 || n > 0xffffff) {
        if (tid == 0) *my_n = 0;
        return;
    }
    const Cand* cd = cand + (size_t)b * cand_cap + off;
    uint16_t* label = label_ws + (size_t)b * cand_cap + off;
    const int csz = max(dp.cell_size, 1);
    const int ncol = max(aw / csz, 1), nrow = max(ah / csz, 1);
    const int wc = (aw + ncol - 1) / ncol, hc = (ah + nrow - 1) / nrow;
    const bool cells_fit = (ncol * nrow + 31) / 32 <= dp.cell_words;   // host-sized; checked anyway
    auto cell_of = [&](int x, int y) { return min(y / hc, nrow - 1) * ncol + min(x / wc, ncol - 1); };
    // coordinates relative to the keypoint area, clamped into it (the border rule already holds for every candidate)
    auto rel = [&](const Cand& c, int& x, int& y, int& sc) {
        x = min(max((int)c.x - lo, 0), aw - 1);
        y = min(max((int)c.y - lo, 0), ah - 1);
        sc = (int)c.score;
    };

    // ---- A. cells with a strong corner
    for (int i = tid; i < dp.cell_words; i += kDistributeThreads) s_strong[i] = 0;
    __syncthreads();
    if (cells_fit)
        for (int i = tid; i < n; i += kDistributeThreads) {
            int x, y, sc;
            rel(cd[i], x, y, sc);
            if (sc >= dp.high_threshold) {
                const int c = cell_of(x, y);
                atomicOr(&s_strong[c >> 5], 1 << (c & 31));
            }
        }
    // ---- B. roots
    int n_roots = (2 * aw + ah) / (2 * ah);
    n_roots = min(min(max(n_roots, 1), kDistributeMaxRoots), NM);
    if (tid < kDistributeMaxRoots) s_cc[tid] = 0;
    __syncthreads();
    auto root_of = [&](int x) {
        int r = 0;
        for (int k = 1; k < n_roots; k++) r += x >= (int)((long long)k * aw / n_roots);
        return r;
    };
    for (int i = tid; i < n; i += kDistributeThreads) {
        int x, y, sc;
        rel(cd[i], x, y, sc);
        bool active = sc >= dp.high_threshold;
        if (!active && cells_fit) {
            const int c = cell_of(x, y);
            active = ((s_strong[c >> 5] >> (c & 31)) & 1) == 0;
        }
        int lab = 0xffff;
        if (active) {
            lab = root_of(x);
            atomicAdd(&s_cc[lab], 1);
        }
        label[i] = (uint16_t)lab;
    }
    __syncthreads();
    if (tid == 0) {
        int m = 0;
        for (int r = 0; r < n_roots; r++) {
            s_pos[r] = m;
            if (s_cc[r] > 0) {
                const int x0 = (int)((long long)r * aw / n_roots), x1 = (int)((long long)(r + 1) * aw / n_roots);
                s_bx[m] = x0 | (x1 << 16);
                s_by[m] = 0 | (ah << 16);
                s_cnt[m] = s_cc[r];
                m++;
            }
        }
        s_m = m;
    }
    __syncthreads();
    for (int i = tid; i < n; i += kDistributeThreads) {
        const int lab = label[i];
        if (lab != 0xffff) label[i] = (uint16_t)s_pos[min(lab, kDistributeMaxRoots - 1)];
    }
    __syncthreads();
    int gen = 0;   // which generation of the box arrays is current
    for (int round = 0; round < 32; round++) {
        const int m = s_m;
        int* bx = s_bx + gen * NM;
        int* by = s_by + gen * NM;
        int* cnt = s_cnt + gen * NM;
        auto can_cut = [&](int j) {
            const int w = (bx[j] >> 16) - (bx[j] & 0xffff), hh = (by[j] >> 16) - (by[j] & 0xffff);
            return cnt[j] > 1 && (w > 1 || hh > 1);
        };
        if (tid == 0) s_cuttable = 0;
        __syncthreads();
        {
            int mine = 0;
            for (int j = tid; j < m; j += kDistributeThreads) mine += can_cut(j);
            if (mine) atomicAdd(&s_cuttable, mine);
        }
        for (int j = tid; j < 4 * m; j += kDistributeThreads) s_cc[j] = 0;
        __syncthreads();
        const int n_cuttable = s_cuttable;
        if (m >= quota || n_cuttable == 0 || m <= 0) break;   // uniform: every thread reads the same LDS words
        // quadrant populations of the boxes that can be cut
        for (int i = tid; i < n; i += kDistributeThreads) {
            const int lab = label[i];
            if (lab >= m) continue;   // 0xffff: not active
            if (!can_cut(lab)) continue;
            int x, y, sc;
            rel(cd[i], x, y, sc);
            const int x0 = bx[lab] & 0xffff, x1 = bx[lab] >> 16, y0 = by[lab] & 0xffff, y1 = by[lab] >> 16;
            const int xm = x0 + (x1 - x0 + 1) / 2, ym = y0 + (y1 - y0 + 1) / 2;
            atomicAdd(&s_cc[4 * lab + (x >= xm ? 1 : 0) + (y >= ym ? 2 : 0)], 1);
        }
        __syncthreads();
        // which boxes are cut
        const bool all = m + 3 * n_cuttable <= quota;
        for (int j = tid; j < m; j += kDistributeThreads) s_cut[j] = all && can_cut(j);
        if (!all) {
            // rank among the boxes that can be cut: larger population first, earlier box first among equals
            for (int j = tid; j < m; j += kDistributeThreads) {
                if (!can_cut(j)) continue;
                int r = 0;
                for (int k = 0; k < m; k++)
                    if (can_cut(k)) r += cnt[k] > cnt[j] || (cnt[k] == cnt[j] && k < j);
                s_ord[min(r, NM - 1)] = j;
            }
            __syncthreads();
            if (tid == 0) {
                int have = m;
                for (int r = 0; r < n_cuttable && have < quota; r++) {
                    const int j = min(max(s_ord[r], 0), m - 1);
                    int parts = 0;
                    for (int q = 0; q < 4; q++) parts += s_cc[4 * j + q] > 0;
                    s_cut[j] = 1;
                    have += parts - 1;
                }
            }
        }
        __syncthreads();
        // number the next generation in list order: a cut box is followed by its non-empty quadrants, any other by itself
        if (tid < 64) {
            int carry = 0;
            for (int base = 0; base < m; base += 64) {
                const int j = base + tid;
                int parts = 0;
                if (j < m) {
                    parts = 1;
                    if (s_cut[j]) {
                        parts = 0;
                        for (int q = 0; q < 4; q++) parts += s_cc[4 * j + q] > 0;
                    }
                }
                const int incl = wave_scan_incl(parts);
                if (j < m) s_pos[j] = carry + incl - parts;
                carry += __builtin_amdgcn_readlane(incl, 63);
            }
            if (tid == 0) s_new_m = carry;
        }
        __syncthreads();
        const int new_m = s_new_m;
        if (new_m > NM || new_m < m) break;   // cannot happen: at most quota + 3 boxes; kept as a guard for the arrays
        int* nbx = s_bx + (gen ^ 1) * NM;
        int* nby = s_by + (gen ^ 1) * NM;
        int* ncnt = s_cnt + (gen ^ 1) * NM;
        for (int j = tid; j < m; j += kDistributeThreads) {
            const int p = s_pos[j];
            if (!s_cut[j]) {
                nbx[p] = bx[j];
                nby[p] = by[j];
                ncnt[p] = cnt[j];
                continue;
            }
            const int x0 = bx[j] & 0xffff, x1 = bx[j] >> 16, y0 = by[j] & 0xffff, y1 = by[j] >> 16;
            const int xm = x0 + (x1 - x0 + 1) / 2, ym = y0 + (y1 - y0 + 1) / 2;
            int k = p;
            for (int q = 0; q < 4; q++) {
                const int c = s_cc[4 * j + q];
                if (c <= 0) continue;
                if (k < NM) {
                    nbx[k] = ((q & 1) ? xm : x0) | (((q & 1) ? x1 : xm) << 16);
                    nby[k] = ((q & 2) ? ym : y0) | (((q & 2) ? y1 : ym) << 16);
                    ncnt[k] = c;
                }
                k++;
            }
        }
        // every corner moves to its new box
        for (int i = tid; i < n; i += kDistributeThreads) {
            const int lab = label[i];
            if (lab >= m) continue;
            int nl = s_pos[lab];
            if (s_cut[lab]) {
                int x, y, sc;
                rel(cd[i], x, y, sc);
                const int x0 = bx[lab] & 0xffff, x1 = bx[lab] >> 16, y0 = by[lab] & 0xffff, y1 = by[lab] >> 16;
                const int xm = x0 + (x1 - x0 + 1) / 2, ym = y0 + (y1 - y0 + 1) / 2;
                const int q = (x >= xm ? 1 : 0) + (y >= ym ? 2 : 0);
                for (int k = 0; k < q; k++) nl += s_cc[4 * lab + k] > 0;
            }
            label[i] = (uint16_t)nl;
        }
        __syncthreads();
        if (tid == 0) s_m = new_m;
        gen ^= 1;
        __syncthreads();
    }
    // ---- C. the strongest corner of every box
    __syncthreads();
    const int m = min(s_m, NM);
    for (int j = tid; j < m; j += kDistributeThreads) s_best[j] = 0u;
    __syncthreads();
    for (int i = tid; i < n; i += kDistributeThreads) {
        const int lab = label[i];
        if (lab >= m) continue;
        atomicMax(&s_best[lab], ((unsigned)cd[i].score << 24) | (0xffffffu - (unsigned)i));
    }
    __syncthreads();
    const int m_out = min(m, dp.out_stride);
    for (int j = tid; j < m_out; j += kDistributeThreads) my_out[j] = off + (int)(0xffffffu - (s_best[j] & 0xffffffu));
    if (tid == 0) *my_n = m_out;
}

}  // namespace orbx
