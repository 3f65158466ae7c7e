/*
 * CPU oracle, extraction half (TEST INFRASTRUCTURE ONLY; see oracle.h).
 *
 * Restates cv::ORB::detect / cv::ORB::compute of OpenCV 3.4.x as called by
 * ORBExtractor (include/orb_slam/geometry/orb_extractor.h:57-80,100-119) with
 * WTA_K = 2, HARRIS_SCORE, patch 31, firstLevel 0.  OpenCV is a third-party
 * dependency pinned at 3.4.9 (CMakeLists.txt:26) that is absent from the reference
 * tree; every function names the OpenCV routine whose published algorithm it follows.
 * Compile with -ffp-contract=off: none of the reference arithmetic is fused.
 */
#include "oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

inline int cv_round(double v) { return (int)lrint(v); }      /* cvRound: nearest, ties to even */
inline int cv_roundf(float v) { return (int)lrintf(v); }
inline int cv_floor(double v) { int i = (int)v; return i - (i > v); }
inline int cv_ceil(double v) { int i = (int)v; return i + (i < v); }

const int kBitPattern31[256 * 4] = {
#include "bit_pattern_31.inc"
};

struct Rect { int x, y, w, h; };

/* orb.cpp getScale(): (float)pow(scaleFactor_as_double, level - firstLevel) */
inline float layer_scale(const orc_orb_params* p, int level) {
    return (float)std::pow((double)p->scale_factor, (double)(level - p->first_level));
}

/* orb.cpp detectAndCompute(): border and atlas placement of every level. */
struct Layout {
    int border;
    int atlas_w, atlas_h;
    std::vector<Rect> rect;
    std::vector<float> scale;
};

Layout make_layout(const orc_orb_params* p, int w, int h, int nlevels) {
    Layout L;
    const int HARRIS_BLOCK_SIZE = 9;
    int half_patch = p->patch_size / 2;
    int desc_patch = cv_ceil(half_patch * std::sqrt(2.0));
    L.border = std::max(p->edge_threshold, std::max(desc_patch, HARRIS_BLOCK_SIZE / 2)) + 1;
    int border = L.border;
    L.rect.resize(nlevels);
    L.scale.resize(nlevels);
    int level_dy = h + border * 2;
    int ofs_x = 0, ofs_y = 0;
    int buf_w = (cv_round(w / layer_scale(p, 0)) + border * 2 + 15) & -16;
    for (int level = 0; level < nlevels; level++) {
        float scale = layer_scale(p, level);
        L.scale[level] = scale;
        int sw = cv_round(w / scale), sh = cv_round(h / scale);
        int whole_w = sw + border * 2, whole_h = sh + border * 2;
        if (ofs_x + whole_w > buf_w) {
            ofs_x = 0;
            ofs_y += level_dy;
            level_dy = whole_h;
        }
        L.rect[level] = Rect{ofs_x + border, ofs_y + border, sw, sh};
        ofs_x += whole_w;
    }
    L.atlas_w = buf_w;
    L.atlas_h = ofs_y + level_dy;
    return L;
}

/* imgproc color.cpp RGB2Gray<uchar> for BGR input. */
void bgr_to_gray(const orc_orb_params* p, const uint8_t* img, int w, int h, int stride, uint8_t* gray) {
    for (int y = 0; y < h; y++) {
        const uint8_t* s = img + (size_t)y * stride;
        uint8_t* d = gray + (size_t)y * w;
        for (int x = 0; x < w; x++, s += 3) {
            int b = s[0], g = s[1], r = s[2];
            if (p->gray_variant == 0)
                d[x] = (uint8_t)((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14);
            else
                d[x] = (uint8_t)((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15);
        }
    }
}

/* core copy.cpp borderInterpolate(BORDER_REFLECT_101) */
inline int reflect101(int p, int len) {
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

/* Fill the `border` pixels around rect r of the atlas by reflect-101 of the level itself. */
void make_border(uint8_t* atlas, int aw, const Rect& r, int border) {
    for (int y = -border; y < r.h + border; y++) {
        int sy = reflect101(y, r.h);
        uint8_t* drow = atlas + (size_t)(r.y + y) * aw + r.x;
        const uint8_t* srow = atlas + (size_t)(r.y + sy) * aw + r.x;
        for (int x = -border; x < r.w + border; x++) {
            if (y >= 0 && y < r.h && x >= 0 && x < r.w) { x = r.w - 1; continue; }
            drow[x] = srow[reflect101(x, r.w)];
        }
    }
}

/* imgproc resize.cpp, INTER_LINEAR_EXACT for 8-bit: interpolationLinear<uchar>::getCoeffs with
   ufixedpoint16 (Q8.8) weights; hlineResize keeps Q8.8, vlineResize makes Q16.16 and rounds. */
struct LinCoef { int ofs; int w0, w1; };

void exact_coeffs(int src, int dst, std::vector<LinCoef>& c) {
    c.resize(dst);
    double inv_scale = (double)dst / src;
    double scale = 1.0 / inv_scale;
    for (int v = 0; v < dst; v++) {
        double f = scale * ((double)v + 0.5) - 0.5;
        int i = cv_floor(f);
        if (i >= 0 && src > 1) {
            if (i < src - 1) {
                int w1 = cv_round((f - (double)i) * 256.0);
                c[v] = LinCoef{i, 256 - w1, w1};
            } else {
                c[v] = LinCoef{src - 1, 256, 0};
            }
        } else {
            c[v] = LinCoef{0, 256, 0};
        }
    }
}

void resize_linear_exact(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride) {
    std::vector<LinCoef> cx, cy;
    exact_coeffs(sw, dw, cx);
    exact_coeffs(sh, dh, cy);
    for (int y = 0; y < dh; y++) {
        const uint8_t* r0 = src + (size_t)cy[y].ofs * sstride;
        const uint8_t* r1 = src + (size_t)std::min(cy[y].ofs + 1, sh - 1) * sstride;
        for (int x = 0; x < dw; x++) {
            int o = cx[x].ofs, o1 = std::min(o + 1, sw - 1);
            uint32_t h0 = (uint32_t)(cx[x].w0 * r0[o] + cx[x].w1 * r0[o1]);
            uint32_t h1 = (uint32_t)(cx[x].w0 * r1[o] + cx[x].w1 * r1[o1]);
            uint32_t v = (uint32_t)cy[y].w0 * h0 + (uint32_t)cy[y].w1 * h1;
            dst[(size_t)y * dstride + x] = (uint8_t)((v + (1u << 15)) >> 16);
        }
    }
}

/* imgproc resize.cpp, legacy INTER_LINEAR for 8-bit (kept only as an alternative hypothesis for builds
   older than the INTER_LINEAR_EXACT switch): 11-bit weights, HResizeLinear + VResizeLinear<uchar,int,short>. */
void resize_linear_legacy(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride) {
    const int COEF = 2048;
    std::vector<int> xo(dw), yo(dh);
    std::vector<short> xa(dw * 2), ya(dh * 2);
    double sx = (double)sw / dw, sy = (double)sh / dh; /* scale_x = 1/inv_scale_x */
    sx = 1.0 / ((double)dw / sw);
    sy = 1.0 / ((double)dh / sh);
    for (int x = 0; x < dw; x++) {
        float fx = (float)((x + 0.5) * sx - 0.5);
        int ix = cv_floor(fx);
        fx -= ix;
        if (ix < 0) { fx = 0; ix = 0; }
        if (ix >= sw - 1) { fx = 0; ix = sw - 1; }
        xo[x] = ix;
        xa[2 * x] = (short)cv_round((1.f - fx) * COEF);   /* saturate_cast<short>(float) */
        xa[2 * x + 1] = (short)cv_round(fx * COEF);
    }
    for (int y = 0; y < dh; y++) {
        float fy = (float)((y + 0.5) * sy - 0.5);
        int iy = cv_floor(fy);
        fy -= iy;
        yo[y] = iy;
        ya[2 * y] = (short)cv_round((1.f - fy) * COEF);
        ya[2 * y + 1] = (short)cv_round(fy * COEF);
    }
    std::vector<int> row0(dw), row1(dw);
    for (int y = 0; y < dh; y++) {
        int y0 = std::min(std::max(yo[y], 0), sh - 1), y1 = std::min(std::max(yo[y] + 1, 0), sh - 1);
        const uint8_t* r0 = src + (size_t)y0 * sstride;
        const uint8_t* r1 = src + (size_t)y1 * sstride;
        for (int x = 0; x < dw; x++) {
            int o = xo[x], o1 = std::min(o + 1, sw - 1);
            row0[x] = r0[o] * xa[2 * x] + r0[o1] * xa[2 * x + 1];
            row1[x] = r1[o] * xa[2 * x] + r1[o1] * xa[2 * x + 1];
        }
        int b0 = ya[2 * y], b1 = ya[2 * y + 1];
        for (int x = 0; x < dw; x++) {
            int v = (((b0 * (row0[x] >> 4)) >> 16) + ((b1 * (row1[x] >> 4)) >> 16) + 2) >> 2;
            dst[(size_t)y * dstride + x] = (uint8_t)std::min(std::max(v, 0), 255);
        }
    }
}

void build_pyramid(const orc_orb_params* p, const Layout& L, const uint8_t* gray, int w, int h, uint8_t* atlas) {
    int aw = L.atlas_w;
    std::memset(atlas, 0, (size_t)aw * L.atlas_h);
    int nlevels = (int)L.rect.size();
    for (int level = 0; level < nlevels; level++) {
        const Rect& r = L.rect[level];
        uint8_t* cur = atlas + (size_t)r.y * aw + r.x;
        if (level == 0) {
            for (int y = 0; y < h; y++) std::memcpy(cur + (size_t)y * aw, gray + (size_t)y * w, w);
        } else {
            const Rect& q = L.rect[level - 1];
            const uint8_t* prev = atlas + (size_t)q.y * aw + q.x;
            if (p->resize_variant == 0) resize_linear_exact(prev, q.w, q.h, aw, cur, r.w, r.h, aw);
            else resize_linear_legacy(prev, q.w, q.h, aw, cur, r.w, r.h, aw);
        }
        make_border(atlas, aw, r, L.border);
    }
}

/* features2d fast.cpp makeOffsets(patternSize 16) */
const int kCircle[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                            {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

/* features2d fast_score.cpp cornerScore<16> */
int corner_score16(const uint8_t* ptr, const int* pixel, int threshold) {
    const int K = 8, N = K * 3 + 1;
    int k, v = ptr[0];
    short d[N];
    for (k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
    int a0 = threshold;
    for (k = 0; k < 16; k += 2) {
        int a = std::min((int)d[k + 1], (int)d[k + 2]);
        a = std::min(a, (int)d[k + 3]);
        if (a <= a0) continue;
        a = std::min(a, (int)d[k + 4]);
        a = std::min(a, (int)d[k + 5]);
        a = std::min(a, (int)d[k + 6]);
        a = std::min(a, (int)d[k + 7]);
        a = std::min(a, (int)d[k + 8]);
        a0 = std::max(a0, std::min(a, (int)d[k]));
        a0 = std::max(a0, std::min(a, (int)d[k + 9]));
    }
    int b0 = -a0;
    for (k = 0; k < 16; k += 2) {
        int b = std::max((int)d[k + 1], (int)d[k + 2]);
        b = std::max(b, (int)d[k + 3]);
        b = std::max(b, (int)d[k + 4]);
        b = std::max(b, (int)d[k + 5]);
        if (b >= b0) continue;
        b = std::max(b, (int)d[k + 6]);
        b = std::max(b, (int)d[k + 7]);
        b = std::max(b, (int)d[k + 8]);
        b0 = std::min(b0, std::max(b, (int)d[k]));
        b0 = std::min(b0, std::max(b, (int)d[k + 9]));
    }
    return -b0 - 1;
}

/* features2d fast.cpp FAST_t<16>: score of every pixel (0 when not a corner). */
void fast_score_map(const uint8_t* img, int w, int h, int stride, int threshold, uint8_t* scores) {
    const int K = 8, N = 25;
    int pixel[25];
    for (int k = 0; k < 16; k++) pixel[k] = kCircle[k][0] + kCircle[k][1] * stride;
    for (int k = 16; k < 25; k++) pixel[k] = pixel[k - 16];
    threshold = std::min(std::max(threshold, 0), 255);
    std::memset(scores, 0, (size_t)w * h);
    for (int i = 3; i < h - 3; i++) {
        const uint8_t* ptr = img + (size_t)i * stride + 3;
        for (int j = 3; j < w - 3; j++, ptr++) {
            int v = ptr[0];
            bool corner = false;
            {
                int vt = v - threshold, count = 0;
                for (int k = 0; k < N; k++) {
                    int x = ptr[pixel[k]];
                    if (x < vt) { if (++count > K) { corner = true; break; } }
                    else count = 0;
                }
            }
            if (!corner) {
                int vt = v + threshold, count = 0;
                for (int k = 0; k < N; k++) {
                    int x = ptr[pixel[k]];
                    if (x > vt) { if (++count > K) { corner = true; break; } }
                    else count = 0;
                }
            }
            if (corner) scores[(size_t)i * w + j] = (uint8_t)corner_score16(ptr, pixel, threshold);
        }
    }
}

/* features2d fast.cpp FAST_t<16>, non-max suppression part: strict maximum over the 8 neighbours. */
void fast_detect(const uint8_t* img, int w, int h, int stride, int threshold, std::vector<orc_keypoint>& out) {
    out.clear();
    if (w < 7 || h < 7) return;
    std::vector<uint8_t> sc((size_t)w * h);
    fast_score_map(img, w, h, stride, threshold, sc.data());
    for (int y = 3; y < h - 3; y++) {
        const uint8_t* pp = &sc[(size_t)(y - 1) * w];
        const uint8_t* pc = &sc[(size_t)y * w];
        const uint8_t* pn = &sc[(size_t)(y + 1) * w];
        for (int x = 3; x < w - 3; x++) {
            int s = pc[x];
            if (!s) continue;
            if (s > pc[x + 1] && s > pc[x - 1] && s > pp[x - 1] && s > pp[x] && s > pp[x + 1] &&
                s > pn[x - 1] && s > pn[x] && s > pn[x + 1]) {
                orc_keypoint k{(float)x, (float)y, 7.f, -1.f, (float)s, 0, -1};
                out.push_back(k);
            }
        }
    }
}

/* features2d keypoint.cpp KeyPointsFilter::runByImageBorder; the Rect test is on the ROUNDED point
   (Point2f -> Point2i conversion is saturate_cast = cvRound). */
void run_by_image_border(std::vector<orc_keypoint>& k, int w, int h, int border) {
    if (border <= 0) return;
    if (h <= border * 2 || w <= border * 2) { k.clear(); return; }
    int x0 = border, y0 = border, x1 = w - border, y1 = h - border;
    k.erase(std::remove_if(k.begin(), k.end(),
                           [=](const orc_keypoint& p) {
                               int px = cv_roundf(p.x), py = cv_roundf(p.y);
                               return !(x0 <= px && px < x1 && y0 <= py && py < y1);
                           }),
            k.end());
}

/* features2d keypoint.cpp KeyPointsFilter::retainBest: same std:: calls, so the same libstdc++ order. */
void retain_best(std::vector<orc_keypoint>& k, int n_points, int variant) {
    if (n_points >= 0 && k.size() > (size_t)n_points) {
        if (n_points == 0) { k.clear(); return; }
        auto greater = [](const orc_keypoint& a, const orc_keypoint& b) { return a.response > b.response; };
        if (variant == 0) std::nth_element(k.begin(), k.begin() + n_points - 1, k.end(), greater);
        else std::nth_element(k.begin(), k.begin() + n_points, k.end(), greater);
        float ambiguous = k[n_points - 1].response;
        auto new_end = std::partition(k.begin() + n_points, k.end(),
                                      [=](const orc_keypoint& p) { return p.response >= ambiguous; });
        k.resize(new_end - k.begin());
    }
}

/* orb.cpp HarrisResponses(blockSize 7, k 0.04f) */
void harris_responses(const uint8_t* atlas, int step, const Layout& L, std::vector<orc_keypoint>& pts) {
    const int blockSize = 7;
    const float harris_k = 0.04f;
    int r = blockSize / 2;
    float scale = 1.f / ((1 << 2) * blockSize * 255.f);
    float scale_sq_sq = scale * scale * scale * scale;
    for (auto& kp : pts) {
        int x0 = cv_roundf(kp.x), y0 = cv_roundf(kp.y), z = kp.octave;
        const uint8_t* ptr0 = atlas + (size_t)(y0 - r + L.rect[z].y) * step + x0 - r + L.rect[z].x;
        int a = 0, b = 0, c = 0;
        for (int i = 0; i < blockSize; i++)
            for (int j = 0; j < blockSize; j++) {
                const uint8_t* ptr = ptr0 + i * step + j;
                int Ix = (ptr[1] - ptr[-1]) * 2 + (ptr[-step + 1] - ptr[-step - 1]) + (ptr[step + 1] - ptr[step - 1]);
                int Iy = (ptr[step] - ptr[-step]) * 2 + (ptr[step - 1] - ptr[-step - 1]) + (ptr[step + 1] - ptr[-step + 1]);
                a += Ix * Ix;
                b += Iy * Iy;
                c += Ix * Iy;
            }
        kp.response = ((float)a * b - (float)c * c - harris_k * ((float)a + b) * ((float)a + b)) * scale_sq_sq;
    }
}

/* orb.cpp computeKeyPoints(): end of each row of the circular patch. */
void make_umax(int half_patch, std::vector<int>& umax) {
    umax.assign(half_patch + 2, 0);
    int v, v0, vmax = cv_floor(half_patch * std::sqrt(2.f) / 2 + 1);
    int vmin = cv_ceil(half_patch * std::sqrt(2.f) / 2);
    for (v = 0; v <= vmax; ++v) umax[v] = cv_round(std::sqrt((double)half_patch * half_patch - v * v));
    for (v = half_patch, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

/* core mathfuncs.cpp fastAtan2 (scalar atan_f32) */
const float atan2_p1 = 0.9997878412794807f * (float)(180 / M_PI);
const float atan2_p3 = -0.3258083974640975f * (float)(180 / M_PI);
const float atan2_p5 = 0.1555786518463281f * (float)(180 / M_PI);
const float atan2_p7 = -0.04432655554792128f * (float)(180 / M_PI);

float fast_atan2(float y, float x) {
    float ax = std::abs(x), ay = std::abs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((atan2_p7 * c2 + atan2_p5) * c2 + atan2_p3) * c2 + atan2_p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((atan2_p7 * c2 + atan2_p5) * c2 + atan2_p3) * c2 + atan2_p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* orb.cpp ICAngles */
void ic_angles(const uint8_t* atlas, int step, const Layout& L, std::vector<orc_keypoint>& pts,
               const std::vector<int>& u_max, int half_k) {
    for (auto& kp : pts) {
        const Rect& layer = L.rect[kp.octave];
        const uint8_t* center = atlas + (size_t)(cv_roundf(kp.y) + layer.y) * step + cv_roundf(kp.x) + layer.x;
        int m_01 = 0, m_10 = 0;
        for (int u = -half_k; u <= half_k; ++u) m_10 += u * center[u];
        for (int v = 1; v <= half_k; ++v) {
            int v_sum = 0;
            int d = u_max[v];
            for (int u = -d; u <= d; ++u) {
                int val_plus = center[u + v * step], val_minus = center[u - v * step];
                v_sum += (val_plus - val_minus);
                m_10 += u * (val_plus + val_minus);
            }
            m_01 += v * v_sum;
        }
        kp.angle = fast_atan2((float)m_01, (float)m_10);
    }
}

void per_level_quota(const orc_orb_params* p, int nlevels, std::vector<int>& quota) {
    quota.resize(nlevels);
    float factor = (float)(1.0 / (double)p->scale_factor);
    float ndesired = p->nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int level = 0; level < nlevels - 1; level++) {
        quota[level] = cv_roundf(ndesired);
        sum += quota[level];
        ndesired *= factor;
    }
    quota[nlevels - 1] = std::max(p->nfeatures - sum, 0);
}

/* orb.cpp computeKeyPoints(), first half: FAST + border + cull by FAST score, then Harris for all. */
void detect_stage1(const orc_orb_params* p, const Layout& L, const uint8_t* atlas,
                   std::vector<orc_keypoint>& all, std::vector<int>& counters, std::vector<int>& quota) {
    int nlevels = (int)L.rect.size();
    per_level_quota(p, nlevels, quota);
    all.clear();
    counters.assign(nlevels, 0);
    std::vector<orc_keypoint> kps;
    for (int level = 0; level < nlevels; level++) {
        const Rect& r = L.rect[level];
        fast_detect(atlas + (size_t)r.y * L.atlas_w + r.x, r.w, r.h, L.atlas_w, p->fast_threshold, kps);
        run_by_image_border(kps, r.w, r.h, p->edge_threshold);
        retain_best(kps, 2 * quota[level], p->retain_variant);
        counters[level] = (int)kps.size();
        float sf = L.scale[level];
        for (auto& k : kps) {
            k.octave = level;
            k.size = p->patch_size * sf;
        }
        all.insert(all.end(), kps.begin(), kps.end());
    }
    harris_responses(atlas, L.atlas_w, L, all);
}

void detect_impl(const orc_orb_params* p, const uint8_t* img, int w, int h, int stride, int channels,
                 std::vector<orc_keypoint>& out) {
    out.clear();
    if (w <= 0 || h <= 0) return;
    int nlevels = p->nlevels;
    Layout L = make_layout(p, w, h, nlevels);
    std::vector<uint8_t> gray((size_t)w * h), atlas((size_t)L.atlas_w * L.atlas_h);
    orc_gray(p, img, w, h, stride, channels, gray.data());
    build_pyramid(p, L, gray.data(), w, h, atlas.data());

    std::vector<orc_keypoint> all;
    std::vector<int> counters, quota;
    detect_stage1(p, L, atlas.data(), all, counters, quota);
    if (all.empty()) return;

    std::vector<orc_keypoint> kps;
    int offset = 0;
    for (int level = 0; level < nlevels; level++) {
        kps.assign(all.begin() + offset, all.begin() + offset + counters[level]);
        offset += counters[level];
        retain_best(kps, quota[level], p->retain_variant);
        out.insert(out.end(), kps.begin(), kps.end());
    }
    std::vector<int> umax;
    int half_patch = p->patch_size / 2;
    make_umax(half_patch, umax);
    ic_angles(atlas.data(), L.atlas_w, L, out, umax, half_patch);
    for (auto& k : out) {
        float scale = L.scale[k.octave];
        k.x *= scale;
        k.y *= scale;
    }
}

/* imgproc smooth.cpp GaussianBlur(7x7, sigma 2) on an 8-bit sub-matrix without BORDER_ISOLATED:
   generic separable path with the symmetric 8-bit fixed-point kernel (getGaussianKernel -> x256 rounded). */
void gaussian_kernel_f32(float cf[7], double sigma) {
    const int n = 7;
    double scale2x = -0.5 / (sigma * sigma);
    double sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        double t = std::exp(scale2x * x * x);
        cf[i] = (float)t;
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) cf[i] = (float)(cf[i] * sum);
}

void gaussian_kernel_q8(int k[7], double sigma) {
    const int n = 7;
    double scale2x = -0.5 / (sigma * sigma);
    float cf[7];
    double sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        double t = std::exp(scale2x * x * x);
        cf[i] = (float)t;
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) {
        cf[i] = (float)(cf[i] * sum);
        k[i] = cv_roundf(cf[i] * 256.f);
    }
}

/* Investigation variant only (blur_variant 1/2): the same kernel applied in float, rounded once. */
void blur_level_float(const Rect& r, int aw, uint8_t* atlas, double sigma) {
    float kf[7];
    gaussian_kernel_f32(kf, sigma);
    std::vector<float> tmp((size_t)(r.h + 6) * r.w);
    for (int y = -3; y < r.h + 3; y++) {
        const uint8_t* s = atlas + (size_t)(r.y + y) * aw + r.x;
        float* t = &tmp[(size_t)(y + 3) * r.w];
        for (int x = 0; x < r.w; x++) {
            float acc = 0;
            for (int i = 0; i < 7; i++) acc += kf[i] * s[x + i - 3];
            t[x] = acc;
        }
    }
    for (int y = 0; y < r.h; y++) {
        uint8_t* d = atlas + (size_t)(r.y + y) * aw + r.x;
        for (int x = 0; x < r.w; x++) {
            float acc = 0;
            for (int i = 0; i < 7; i++) acc += kf[i] * tmp[(size_t)(y + i) * r.w + x];
            int v = cv_roundf(acc);
            d[x] = (uint8_t)std::min(std::max(v, 0), 255);
        }
    }
}

void blur_levels(const orc_orb_params* p, const Layout& L, uint8_t* atlas) {
    const double sigma = p->blur_sigma > 0 ? (double)p->blur_sigma : 2.0;
    int kq[7];
    gaussian_kernel_q8(kq, sigma);
    int aw = L.atlas_w;
    for (size_t level = 0; level < L.rect.size(); level++) {
        const Rect& r = L.rect[level];
        if (p->blur_variant == 1 || (p->blur_variant == 2 && level == 0)) {
            blur_level_float(r, aw, atlas, sigma);
            continue;
        }
        /* row pass over rows y-3..y+3 of the un-blurred atlas (its border already holds reflect-101 data) */
        std::vector<int> tmp((size_t)(r.h + 6) * r.w);
        for (int y = -3; y < r.h + 3; y++) {
            const uint8_t* s = atlas + (size_t)(r.y + y) * aw + r.x;
            int* t = &tmp[(size_t)(y + 3) * r.w];
            for (int x = 0; x < r.w; x++) {
                int acc = 0;
                for (int i = 0; i < 7; i++) acc += kq[i] * s[x + i - 3];
                t[x] = acc;
            }
        }
        for (int y = 0; y < r.h; y++) {
            uint8_t* d = atlas + (size_t)(r.y + y) * aw + r.x;
            for (int x = 0; x < r.w; x++) {
                int acc = 0;
                for (int i = 0; i < 7; i++) acc += kq[i] * tmp[(size_t)(y + i) * r.w + x];
                int v = (acc + (1 << 15)) >> 16;
                d[x] = (uint8_t)std::min(std::max(v, 0), 255);
            }
        }
    }
}

/* orb.cpp computeOrbDescriptors(), WTA_K == 2 */
inline int round_as(float v, int variant) { return variant == 0 ? cv_roundf(v) : (int)std::floor((double)v + 0.5); }

void orb_descriptors(const uint8_t* atlas, int step, const Layout& L, const std::vector<orc_keypoint>& kps,
                     uint8_t* desc, int rv) {
    for (size_t j = 0; j < kps.size(); j++) {
        const orc_keypoint& kpt = kps[j];
        const Rect& layer = L.rect[kpt.octave];
        float scale = 1.f / L.scale[kpt.octave];
        float angle = kpt.angle;
        angle *= (float)(M_PI / 180.f);
        float a = (float)cos((double)angle), b = (float)sin((double)angle);
        const uint8_t* center = atlas + (size_t)(round_as(kpt.y * scale, rv) + layer.y) * step +
                                round_as(kpt.x * scale, rv) + layer.x;
        const int* pat = kBitPattern31;
        uint8_t* d = desc + j * 32;
        for (int i = 0; i < 32; i++, pat += 32) {
            int val = 0;
            for (int bit = 0; bit < 8; bit++) {
                const int* q = pat + bit * 4;
                float x0 = q[0] * a - q[1] * b, y0 = q[0] * b + q[1] * a;
                float x1 = q[2] * a - q[3] * b, y1 = q[2] * b + q[3] * a;
                int t0 = center[round_as(y0, rv) * step + round_as(x0, rv)];
                int t1 = center[round_as(y1, rv) * step + round_as(x1, rv)];
                val |= (t0 < t1) << bit;
            }
            d[i] = (uint8_t)val;
        }
    }
}

} // namespace

extern "C" {

void orc_orb_layout(const orc_orb_params* p, int w, int h, int nlevels, int* atlas_w, int* atlas_h, int* rects,
                    float* scales) {
    Layout L = make_layout(p, w, h, nlevels);
    *atlas_w = L.atlas_w;
    *atlas_h = L.atlas_h;
    for (int i = 0; i < nlevels; i++) {
        rects[4 * i] = L.rect[i].x;
        rects[4 * i + 1] = L.rect[i].y;
        rects[4 * i + 2] = L.rect[i].w;
        rects[4 * i + 3] = L.rect[i].h;
        if (scales) scales[i] = L.scale[i];
    }
}

void orc_gray(const orc_orb_params* p, const uint8_t* img, int w, int h, int stride, int channels, uint8_t* gray) {
    if (channels == 1) {
        for (int y = 0; y < h; y++) std::memcpy(gray + (size_t)y * w, img + (size_t)y * stride, w);
    } else {
        bgr_to_gray(p, img, w, h, stride, gray);
    }
}

void orc_orb_pyramid(const orc_orb_params* p, const uint8_t* img, int w, int h, int stride, int channels, int nlevels,
                     uint8_t* atlas) {
    Layout L = make_layout(p, w, h, nlevels);
    std::vector<uint8_t> gray((size_t)w * h);
    orc_gray(p, img, w, h, stride, channels, gray.data());
    build_pyramid(p, L, gray.data(), w, h, atlas);
}

void orc_orb_blur(const orc_orb_params* p, int w, int h, int nlevels, uint8_t* atlas) {
    Layout L = make_layout(p, w, h, nlevels);
    blur_levels(p, L, atlas);
}

void orc_fast_scores(const uint8_t* img, int w, int h, int stride, int threshold, uint8_t* scores) {
    fast_score_map(img, w, h, stride, threshold, scores);
}

int orc_fast_detect(const uint8_t* img, int w, int h, int stride, int threshold, orc_keypoint* out, int cap) {
    std::vector<orc_keypoint> k;
    fast_detect(img, w, h, stride, threshold, k);
    for (size_t i = 0; i < k.size() && (int)i < cap; i++) out[i] = k[i];
    return (int)k.size();
}

int orc_orb_detect(const orc_orb_params* p, const uint8_t* img, int w, int h, int stride, int channels,
                   orc_keypoint* out, int cap) {
    std::vector<orc_keypoint> k;
    detect_impl(p, img, w, h, stride, channels, k);
    for (size_t i = 0; i < k.size() && (int)i < cap; i++) out[i] = k[i];
    return (int)k.size();
}

/* The opt-in distribution mode (no counterpart in the reference; defined by distribute_oracle.cpp): FAST at the low
   threshold per level, border rule, per-cell adaptive threshold + quad-tree selection with the level's cv::ORB quota,
   response = FAST corner score, then orientation and level-0 coordinates as in detect_impl. */
int orc_orb_detect_distributed(const orc_orb_params* p, const uint8_t* img, int w, int h, int stride, int channels,
                               int low_threshold, int cell_size, orc_keypoint* out, int cap) {
    if (w <= 0 || h <= 0) return 0;
    const int nlevels = p->nlevels;
    Layout L = make_layout(p, w, h, nlevels);
    std::vector<uint8_t> gray((size_t)w * h), atlas((size_t)L.atlas_w * L.atlas_h);
    orc_gray(p, img, w, h, stride, channels, gray.data());
    build_pyramid(p, L, gray.data(), w, h, atlas.data());
    std::vector<int> quota;
    per_level_quota(p, nlevels, quota);
    std::vector<orc_keypoint> kept_all, kps;
    const int lo = std::max(3, p->edge_threshold);
    for (int level = 0; level < nlevels; level++) {
        const Rect& r = L.rect[level];
        fast_detect(atlas.data() + (size_t)r.y * L.atlas_w + r.x, r.w, r.h, L.atlas_w, low_threshold, kps);
        run_by_image_border(kps, r.w, r.h, p->edge_threshold);
        std::vector<int32_t> xs(kps.size()), ys(kps.size()), sc(kps.size()), kept((size_t)quota[level] + 8);
        for (size_t i = 0; i < kps.size(); i++) {
            xs[i] = (int)kps[i].x;
            ys[i] = (int)kps[i].y;
            sc[i] = (int)kps[i].response;
        }
        const int m = orc_quadtree_select(xs.data(), ys.data(), sc.data(), (int)kps.size(), r.w, r.h, lo, quota[level],
                                          p->fast_threshold, cell_size, kept.data());
        for (int j = 0; j < m; j++) {
            orc_keypoint k = kps[kept[j]];
            k.octave = level;
            k.size = p->patch_size * L.scale[level];
            kept_all.push_back(k);
        }
    }
    std::vector<int> umax;
    const int half_patch = p->patch_size / 2;
    make_umax(half_patch, umax);
    ic_angles(atlas.data(), L.atlas_w, L, kept_all, umax, half_patch);
    for (auto& k : kept_all) {
        const float scale = L.scale[k.octave];
        k.x *= scale;
        k.y *= scale;
    }
    for (size_t i = 0; i < kept_all.size() && (int)i < cap; i++) out[i] = kept_all[i];
    return (int)kept_all.size();
}

int orc_orb_detect_stage1(const orc_orb_params* p, const uint8_t* img, int w, int h, int stride, int channels,
                          orc_keypoint* out, int cap, int* per_level_counts) {
    Layout L = make_layout(p, w, h, p->nlevels);
    std::vector<uint8_t> gray((size_t)w * h), atlas((size_t)L.atlas_w * L.atlas_h);
    orc_gray(p, img, w, h, stride, channels, gray.data());
    build_pyramid(p, L, gray.data(), w, h, atlas.data());
    std::vector<orc_keypoint> all;
    std::vector<int> counters, quota;
    detect_stage1(p, L, atlas.data(), all, counters, quota);
    for (int i = 0; i < p->nlevels; i++) per_level_counts[i] = counters[i];
    for (size_t i = 0; i < all.size() && (int)i < cap; i++) out[i] = all[i];
    return (int)all.size();
}

int orc_orb_compute(const orc_orb_params* p, const uint8_t* img, int w, int h, int stride, int channels,
                    orc_keypoint* kps, int n, uint8_t* desc) {
    if (w <= 0 || h <= 0) return n;
    /* orb.cpp detectAndCompute(useProvidedKeypoints): levels = max octave + 1 */
    int nlevels = 0;
    bool sorted = true;
    for (int i = 0; i < n; i++) {
        int level = kps[i].octave;
        if (i > 0 && level < kps[i - 1].octave) sorted = false;
        nlevels = std::max(nlevels, level);
    }
    nlevels++;
    Layout L = make_layout(p, w, h, nlevels);
    std::vector<uint8_t> gray((size_t)w * h), atlas((size_t)L.atlas_w * L.atlas_h);
    orc_gray(p, img, w, h, stride, channels, gray.data());
    build_pyramid(p, L, gray.data(), w, h, atlas.data());

    std::vector<orc_keypoint> k(kps, kps + n);
    run_by_image_border(k, w, h, p->edge_threshold);
    if (!sorted) {
        std::vector<std::vector<orc_keypoint>> by_level(nlevels);
        for (auto& q : k) by_level[q.octave].push_back(q);
        k.clear();
        for (auto& v : by_level) k.insert(k.end(), v.begin(), v.end());
    }
    if (k.empty()) return 0;
    blur_levels(p, L, atlas.data());
    orb_descriptors(atlas.data(), L.atlas_w, L, k, desc, p->round_variant);
    std::copy(k.begin(), k.end(), kps);
    return (int)k.size();
}

float orc_scale_tables(float scale_factor, int nlevels, float* scale, float* inv_scale, float* sigma2,
                       float* inv_sigma2) {
    /* orb_extractor.h:39-54 */
    float log_sf = log(scale_factor);
    scale[0] = 1.0f;
    sigma2[0] = 1.0f;
    inv_scale[0] = 1.0f / scale[0];
    inv_sigma2[0] = 1.0f / sigma2[0];
    for (int i = 1; i < nlevels; ++i) {
        scale[i] = scale[i - 1] * scale_factor;
        inv_scale[i] = 1.0f / scale[i];
        sigma2[i] = scale[i] * scale[i];
        inv_sigma2[i] = 1.0f / sigma2[i];
    }
    return log_sf;
}

float orc_fast_atan2(float y, float x) { return fast_atan2(y, x); }

/* cv::KeyPointsFilter::retainBest on bare responses (the real std::nth_element / std::partition): perm_out receives
   the original indices in the order the kept keypoints end up in; returns the new length. */
int orc_retain_best(const float* keys, int n, int n_points, int32_t* perm_out) {
    std::vector<orc_keypoint> k(n);
    for (int i = 0; i < n; i++) k[i] = orc_keypoint{0.f, 0.f, 0.f, 0.f, keys[i], 0, i};
    retain_best(k, n_points, 0);
    for (size_t i = 0; i < k.size(); i++) perm_out[i] = k[i].class_id;
    return (int)k.size();
}

/* calib3d undistort.cpp cvUndistortPoints with P = K, 5 fixed-point iterations (camera.cpp:97-126). */
void orc_undistort_points(const orc_camera* c, const float* xy_in, int n, float* xy_out) {
    double ifx = 1. / c->fx, ify = 1. / c->fy;
    for (int i = 0; i < n; i++) {
        double x = xy_in[2 * i], y = xy_in[2 * i + 1];
        double u = x, v = y;
        (void)u; (void)v;
        x = (x - c->cx) * ifx;
        y = (y - c->cy) * ify;
        double x0 = x, y0 = y;
        for (int j = 0; j < 5; j++) {
            double r2 = x * x + y * y;
            double icdist = 1. / (1 + ((c->k3 * r2 + c->k2) * r2 + c->k1) * r2);
            double dx = 2 * c->p1 * x * y + c->p2 * (r2 + 2 * x * x);
            double dy = c->p1 * (r2 + 2 * y * y) + 2 * c->p2 * x * y;
            x = (x0 - dx) * icdist;
            y = (y0 - dy) * icdist;
        }
        double xx = x * c->fx + c->cx;
        double yy = y * c->fy + c->cy;
        xy_out[2 * i] = (float)xx;
        xy_out[2 * i + 1] = (float)yy;
    }
}

void orc_image_bounds(const orc_camera* c, float* out4) {
    float b[8] = {0.f, 0.f, (float)c->width, 0.f, 0.f, (float)c->height, (float)c->width, (float)c->height};
    if (c->k1 != 0.0) orc_undistort_points(c, b, 4, b);
    out4[0] = std::max(std::min(b[0], b[4]), 0.f);
    out4[1] = std::min(std::max(b[2], b[6]), (float)c->width);
    out4[2] = std::max(std::min(b[1], b[3]), 0.f);
    out4[3] = std::min(std::max(b[5], b[7]), (float)c->height);
}

} // extern "C"
