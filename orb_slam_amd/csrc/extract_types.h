// Plain data shared by the extractor kernels and the host logic (no device code).
#pragma once

#include <stdint.h>

namespace orbx {

constexpr int kMaxLevels = 16;
constexpr int kEmitMaxChunks = 64;  // runs the mask words of a frame are cut into for the ordered emission
constexpr int kTileW = 64;   // FAST tile: 64 x 64 level pixels per 256-thread workgroup (32 rows: more halo and barriers per pixel; 128: too few workgroups per CU)
constexpr int kTileH = 64;

struct LevelInfo {
    int x, y, w, h;        // un-padded level rectangle inside the atlas
    int word_base;         // first 64-pixel mask word of this level
    int words_per_row;     // ceil(w / 64)
    int quota;             // keypoints wanted on this level (cv::ORB nfeaturesPerLevel)
    int coef_x, coef_y;    // offsets of this level's resize coefficient rows in the coefficient table
    float scale;           // cv::ORB layer scale  (float)pow(scaleFactor, level)
    float inv_scale;       // 1.f / scale
};

struct Geom {
    int nlevels;
    int atlas_w, atlas_h;
    int border;
    int total_words;
    int total_tiles;
    int edge;              // edge_threshold
    LevelInfo L[kMaxLevels];
};

// One candidate corner: level coordinates, FAST score, then Harris response and IC angle once computed.
struct Cand {
    uint16_t x, y;
    uint8_t score, level;
    uint16_t pad;
    float harris;
    float angle;
};
static_assert(sizeof(Cand) == 16, "Cand layout");

// Resize coefficient of one destination column/row: source offset and the Q8 weight of the second sample.
struct Coef {
    int16_t ofs;
    int16_t w1;
};

// A strip of the dword-wise kernels: 128 atlas pixels (32 lanes x 4 bytes, 4-byte aligned in the atlas) by
// 2 x kStripRows level rows (the two 32-lane halves of a wavefront take one half each).
constexpr int kStripRows = 32;
struct Strip {
    int16_t level, pad;
    int16_t ax0;   // atlas x of the strip's first pixel (multiple of 4)
    int16_t y0;    // first level row
};

// Resize coefficients indexed by ATLAS position (reflect-101 border already folded in), so that a lane's four
// adjacent output pixels read one aligned 16-byte entry group.
struct CoefX {
    int16_t src_ax;    // atlas x of the left source sample
    uint16_t w1f;      // bits 0-8: Q8 weight of the right sample (0..256); bit 14: right sample = left + 1 (else
                       // the same sample); bit 15: pixel lies inside the padded level rect
};
struct CoefY {
    int32_t row0, row1;  // atlas rows of the two source samples
    int32_t w1;          // Q8 weight of the lower sample
};

struct Tile {
    int16_t level, pad;
    int16_t x0, y0;
};

// One 64-pixel run of a level row after non-max suppression: a bit per keypoint and the FAST scores of the first
// eight of them (lowest x first); a ninth or later keypoint of the run has its score in the score plane.
struct MaskWord {
    unsigned long long bits;
    unsigned long long scores;
};

}  // namespace orbx
