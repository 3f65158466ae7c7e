// Host-side geometry of the extractor for one image size: cv::ORB's level sizes and atlas placement, per-level
// quotas, FAST tiles, blur strips and the resize coefficient tables.  Plain C++ (no device calls), so that the
// planning code built on it can be checked on a machine without a GPU.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "extract_types.h"
#include "orbx.h"

namespace orbx {

struct ResizePlan { int cx_off, cy_off, ax_begin, n_dwords; bool narrow; };

struct HostGeometry {
    Geom g;
    std::vector<Tile> tiles;
    std::vector<Strip> strips;
    std::vector<Coef> coefs;
    std::vector<CoefX> coefx;
    std::vector<CoefY> coefy;
    ResizePlan rplan[kMaxLevels];
    size_t atlas_bytes = 0;
};

namespace geomdetail {
inline int cv_round(double v) { return (int)lrint(v); }
inline int cv_floor(double v) { int i = (int)v; return i - (i > v); }
inline int cv_ceil(double v) { int i = (int)v; return i + (i < v); }
inline int div_up(int a, int b) { return (a + b - 1) / b; }
constexpr int kResizeRowsHost = 8;      // must equal kResizeRows / kResizeSrcRows of the strip kernels
constexpr int kResizeSrcRowsHost = 12;
}  // namespace geomdetail

// cv::ORB detectAndCompute(): border, level sizes and atlas placement; computeKeyPoints(): per-level quotas.
inline orbx_status compute_geometry(const orbx_extractor_params& prm, int w, int hgt, int nlevels, HostGeometry& out) {
    constexpr int kResizeRows = geomdetail::kResizeRowsHost, kResizeSrcRows = geomdetail::kResizeSrcRowsHost;
    if (nlevels < 1 || nlevels > kMaxLevels) return ORBX_UNSUPPORTED;
    const orbx_extractor_params& p = prm;
    Geom& g = out.g;
    std::memset(&g, 0, sizeof(g));
    const int HARRIS_BLOCK_SIZE = 9;
    int half_patch = p.patch_size / 2;
    int desc_patch = geomdetail::cv_ceil(half_patch * std::sqrt(2.0));
    int border = std::max(p.edge_threshold, std::max(desc_patch, HARRIS_BLOCK_SIZE / 2)) + 1;
    g.border = border;
    g.nlevels = nlevels;
    g.edge = p.edge_threshold;
    auto layer_scale = [&](int level) {
        return (float)std::pow((double)p.scale_factor, (double)(level - p.first_level));
    };
    int level_dy = hgt + border * 2;
    int ofs_x = 0, ofs_y = 0;
    int buf_w = (geomdetail::cv_round(w / layer_scale(0)) + border * 2 + 15) & -16;
    for (int level = 0; level < nlevels; level++) {
        float scale = layer_scale(level);
        int sw = geomdetail::cv_round(w / scale), sh = geomdetail::cv_round(hgt / scale);
        if (sw < 1 || sh < 1) return ORBX_UNSUPPORTED;
        int whole_w = sw + border * 2, whole_h = sh + border * 2;
        if (ofs_x + whole_w > buf_w) {
            ofs_x = 0;
            ofs_y += level_dy;
            level_dy = whole_h;
        }
        LevelInfo& L = g.L[level];
        L.x = ofs_x + border;
        L.y = ofs_y + border;
        L.w = sw;
        L.h = sh;
        L.scale = scale;
        L.inv_scale = 1.f / scale;
        ofs_x += whole_w;
    }
    g.atlas_w = buf_w;
    g.atlas_h = ofs_y + level_dy;

    // quotas (computeKeyPoints): levels beyond the configured count (provided keypoints only) get none
    {
        int nl = p.level_pyramid;
        float factor = (float)(1.0 / (double)p.scale_factor);
        float ndesired = p.n_key_points * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nl));
        int sum = 0;
        for (int level = 0; level < nlevels; level++) {
            if (level < nl - 1) {
                g.L[level].quota = (int)lrintf(ndesired);
                sum += g.L[level].quota;
                ndesired *= factor;
            } else if (level == nl - 1) {
                g.L[level].quota = std::max(p.n_key_points - sum, 0);
            } else {
                g.L[level].quota = 0;
            }
        }
    }

    // mask words, tiles, resize coefficients
    out.tiles.clear();
    out.strips.clear();
    out.coefs.clear();
    int words = 0;
    for (int level = 0; level < nlevels; level++) {
        LevelInfo& L = g.L[level];
        L.word_base = words;
        L.words_per_row = geomdetail::div_up(L.w, 64);
        words += L.words_per_row * L.h;
        for (int y0 = 0; y0 < L.h; y0 += kTileH)
            for (int x0 = 0; x0 < L.w; x0 += kTileW) out.tiles.push_back(Tile{(int16_t)level, 0, (int16_t)x0, (int16_t)y0});
        for (int y0 = 0; y0 < L.h; y0 += 2 * kStripRows)
            for (int ax0 = L.x & ~3; ax0 < L.x + L.w; ax0 += 128)
                out.strips.push_back(Strip{(int16_t)level, 0, (int16_t)ax0, (int16_t)y0});
        if (level > 0) {
            // cv::resize INTER_LINEAR_EXACT, interpolationLinear<uchar>::getCoeffs: Q8 weights from IEEE double
            const LevelInfo& S = g.L[level - 1];
            auto fill = [&](int src, int dst) {
                double inv_scale = (double)dst / src;
                double scale = 1.0 / inv_scale;
                for (int v = 0; v < dst; v++) {
                    double f = scale * ((double)v + 0.5) - 0.5;
                    int i = geomdetail::cv_floor(f);
                    Coef c;
                    if (i >= 0 && src > 1) {
                        if (i < src - 1) {
                            c.ofs = (int16_t)i;
                            c.w1 = (int16_t)geomdetail::cv_round((f - (double)i) * 256.0);
                        } else {
                            c.ofs = (int16_t)(src - 1);
                            c.w1 = 0;
                        }
                    } else {
                        c.ofs = 0;
                        c.w1 = 0;
                    }
                    out.coefs.push_back(c);
                }
            };
            L.coef_x = (int)out.coefs.size();
            fill(S.w, L.w);
            L.coef_y = (int)out.coefs.size();
            fill(S.h, L.h);
        }
    }
    // atlas-indexed coefficient groups of the dword-wise resize (reflect-101 border folded in)
    out.coefx.clear();
    out.coefy.clear();
    for (int level = 1; level < nlevels; level++) {
        const LevelInfo& L = g.L[level];
        const LevelInfo& S = g.L[level - 1];
        ResizePlan& rp = out.rplan[level];
        auto refl = [](int p, int len) {
            if (len == 1) return 0;
            while ((unsigned)p >= (unsigned)len) p = p < 0 ? -p : 2 * (len - 1) - p;
            return p;
        };
        const int x_first = L.x - border, x_last = L.x + L.w + border;  // padded rect [x_first, x_last)
        rp.ax_begin = x_first & ~3;
        rp.n_dwords = (x_last - rp.ax_begin + 3) / 4;
        rp.cx_off = (int)out.coefx.size();
        rp.narrow = true;  // every group's left samples lie within 8 bytes of the group's aligned base
        for (int i = 0; i < rp.n_dwords * 4; i++) {
            const int ax = rp.ax_begin + i;
            CoefX c{0, 0};
            if (ax >= x_first && ax < x_last) {
                const Coef& k = out.coefs[L.coef_x + refl(ax - L.x, L.w)];
                const int o1 = std::min(k.ofs + 1, S.w - 1);
                c.src_ax = (int16_t)(S.x + k.ofs);
                c.w1f = (uint16_t)((k.w1 & 0x1ff) | ((o1 != k.ofs) ? 0x4000 : 0) | 0x8000);
            }
            out.coefx.push_back(c);
        }
        for (int q = 0; q < rp.n_dwords; q++) {
            int lo = 0x7fffffff, hi = -1;
            for (int j = 0; j < 4; j++) {
                const CoefX& c = out.coefx[rp.cx_off + 4 * q + j];
                if (c.w1f & 0x8000) { lo = std::min(lo, (int)c.src_ax); hi = std::max(hi, (int)c.src_ax); }
            }
            if (hi >= 0 && hi - (lo & ~3) >= 8) rp.narrow = false;
        }
        rp.cy_off = (int)out.coefy.size();
        for (int ty = 0; ty < L.h + 2 * border; ty++) {
            const Coef& k = out.coefs[L.coef_y + refl(ty - border, L.h)];
            out.coefy.push_back(CoefY{S.y + k.ofs, S.y + std::min(k.ofs + 1, S.h - 1), k.w1});
        }
        // the strip kernel parks at most kResizeSrcRows filtered source rows per block of kResizeRows output rows
        for (int t0 = 0; t0 < L.h + 2 * border; t0 += kResizeRows) {
            int lo = 0x7fffffff, hi = -1;
            for (int ty = t0; ty < std::min(t0 + kResizeRows, L.h + 2 * border); ty++) {
                const CoefY& c = out.coefy[rp.cy_off + ty];
                lo = std::min(lo, std::min(c.row0, c.row1));
                hi = std::max(hi, std::max(c.row0, c.row1));
            }
            if (hi - lo + 1 > kResizeSrcRows) rp.narrow = false;
        }
    }
    g.total_words = words;
    g.total_tiles = (int)out.tiles.size();
    out.atlas_bytes = (size_t)g.atlas_w * g.atlas_h;
    return ORBX_OK;
}

}  // namespace orbx
