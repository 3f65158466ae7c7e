// Device kernels of the ORB extractor (gfx950).  All arithmetic that decides a result is integer, or float
// evaluated without contraction (the translation unit is compiled with -ffp-contract=off), so results are
// bit-identical to the scalar CPU evaluation order of cv::ORB.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "extract_types.h"
#include "pyramid_stream.h"
#include "orbx.h"
#include "stdsel.h"

namespace orbx {

__device__ __forceinline__ int reflect101(int p, int len) {
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

// Four bytes from any address; the compiler picks the widest load the target allows for byte alignment.
__device__ __forceinline__ unsigned load_u32_any(const uint8_t* p) {
    unsigned v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

__device__ __forceinline__ unsigned gray_of(unsigned b, unsigned g, unsigned r) {
    return (b * 1868u + g * 9617u + r * 4899u + (1u << 13)) >> 14;
}

// Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  For kernels that gather from one frame's
// planes, a 1-D grid of 8 * chunks * ceil(frames / 8) workgroups decoded here keeps all chunks of a frame on one XCD,
// so a cache line of the frame is fetched into one L2 only.  Returns false for the padding workgroups.
__device__ __forceinline__ bool frame_chunk_of_block(int chunks, int frames, int& frame, int& chunk) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    frame = (j / chunks) * 8 + xcd;
    chunk = j - (j / chunks) * chunks;
    return frame < frames;
}

// K1: BGR/gray -> level 0 of the atlas including its reflect-101 border, sixteen atlas pixels per lane (48 source
// bytes in three 16-byte loads, one 16-byte store).
// cv::cvtColor(BGR2GRAY) 8-bit: (B*1868 + G*9617 + R*4899 + 8192) >> 14.
__device__ __forceinline__ uint4 load_u128_any(const uint8_t* p) {
    uint4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}

__device__ __forceinline__ unsigned gray4(unsigned w0, unsigned w1, unsigned w2) {  // 12 BGR bytes -> 4 gray bytes
    return gray_of(w0 & 255u, (w0 >> 8) & 255u, (w0 >> 16) & 255u) |
           gray_of(w0 >> 24, w1 & 255u, (w1 >> 8) & 255u) << 8 |
           gray_of((w1 >> 16) & 255u, w1 >> 24, w2 & 255u) << 16 |
           gray_of((w2 >> 8) & 255u, (w2 >> 16) & 255u, w2 >> 24) << 24;
}

// blockIdx.y = 0: the 16-pixel groups that lie wholly inside the image (one 16-byte store per lane);
// blockIdx.y = 1: the groups that touch the reflected border columns, one pixel per lane.
__global__ __launch_bounds__(256) void k_gray_level0(const uint8_t* __restrict__ img, size_t frame_stride, int w, int h,
                                                     int stride, int ch, uint8_t* __restrict__ atlas,
                                                     size_t atlas_bytes, int aw, int rx, int ry, int border) {
    const int b = blockIdx.z;
    const int ax_begin = (rx - border) & ~15;
    const int x_first = rx - border, x_last = rx + w + border;
    const int nq = (x_last - ax_begin + 15) >> 4;            // 16-pixel groups per row
    const int q_in0 = (rx - ax_begin + 15) >> 4;             // first group with all pixels at sx >= 0
    const int q_in1 = (rx + w - ax_begin) >> 4;              // one past the last group with all pixels at sx < w
    const int n_in = max(q_in1 - q_in0, 0);
    const int rows = h + 2 * border;
    const int item = blockIdx.x * 256 + threadIdx.x;
    const uint8_t* frame = img + (size_t)b * frame_stride;
    uint8_t* plane = atlas + (size_t)b * atlas_bytes;
    if (blockIdx.y == 0) {
        if (n_in == 0) return;
        const int ty = item / n_in, q = q_in0 + item - ty * n_in;
        if (ty >= rows) return;
        const int ax = ax_begin + 16 * q, sx = ax - rx;
        const uint8_t* row = frame + (size_t)reflect101(ty - border, h) * stride;
        uint8_t* d = plane + (size_t)(ry - border + ty) * aw + ax;
        uint4 out;
        if (ch == 1) out = load_u128_any(row + sx);
        else {
            const uint8_t* s = row + (size_t)sx * 3;
            const uint4 v0 = load_u128_any(s), v1 = load_u128_any(s + 16), v2 = load_u128_any(s + 32);
            out.x = gray4(v0.x, v0.y, v0.z);
            out.y = gray4(v0.w, v1.x, v1.y);
            out.z = gray4(v1.z, v1.w, v2.x);
            out.w = gray4(v2.y, v2.z, v2.w);
        }
        if (((aw | ax) & 15) == 0) *(uint4*)d = out;
        else __builtin_memcpy(d, &out, 16);
        return;
    }
    const int n_edge = (nq - n_in) * 16;                     // edge pixels per row
    if (n_edge == 0) return;
    const int ty = item / n_edge, e = item - ty * n_edge;
    if (ty >= rows) return;
    const int left = q_in0 * 16;                             // edge pixels left of the interior groups
    const int ax = ax_begin + (e < left ? e : e - left + q_in1 * 16);
    if (ax < x_first || ax >= x_last) return;
    const uint8_t* s = frame + (size_t)reflect101(ty - border, h) * stride + (size_t)reflect101(ax - rx, w) * ch;
    plane[(size_t)(ry - border + ty) * aw + ax] = (uint8_t)(ch == 1 ? s[0] : gray_of(s[0], s[1], s[2]));
}

// K2 (dword-wise): level l from level l-1 by cv::resize(INTER_LINEAR_EXACT), border included.  A lane produces four
// adjacent atlas pixels: one 16-byte coefficient group (atlas-indexed, reflect-101 folded in on the host), three
// aligned dwords per source row covering the (at most 7 pixel wide) source span, one dword store.
__device__ __forceinline__ unsigned pick_byte(unsigned d0, unsigned d1, unsigned d2, int k) {
    const unsigned d = k < 4 ? d0 : (k < 8 ? d1 : d2);
    return (d >> ((k & 3) * 8)) & 0xffu;
}

__global__ __launch_bounds__(256) void k_resize_level4(uint8_t* __restrict__ atlas, size_t atlas_bytes, int aw,
                                                       LevelInfo dst, int border, const CoefX* __restrict__ cxs,
                                                       const CoefY* __restrict__ cys, int ax_begin, int n_dwords) {
    const int dc = blockIdx.x * 64 + (threadIdx.x & 63);
    const int ty = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dc >= n_dwords || ty >= dst.h + 2 * border) return;
    const CoefY cy = cys[ty];
    const uint4 raw = *(const uint4*)(cxs + dc * 4);
    const unsigned cw[4] = {raw.x, raw.y, raw.z, raw.w};
    int sx[4], w1[4], step[4];
    bool ok[4];
    int lo = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        sx[j] = (int)(short)(cw[j] & 0xffffu);
        const unsigned f = cw[j] >> 16;
        w1[j] = (int)(f & 0x1ffu);
        step[j] = (int)((f >> 14) & 1u);
        ok[j] = (f >> 15) != 0;
        if (ok[j]) lo = min(lo, sx[j]);
    }
    if (lo == 0x7fffffff) return;
    const int base = lo & ~3;
    uint8_t* fb = atlas + (size_t)blockIdx.z * atlas_bytes;
    const unsigned* r0 = (const unsigned*)(fb + (size_t)cy.row0 * aw + base);
    const unsigned* r1 = (const unsigned*)(fb + (size_t)cy.row1 * aw + base);
    const unsigned a0 = r0[0], a1 = r0[1], a2 = r0[2];
    const unsigned b0 = r1[0], b1 = r1[1], b2 = r1[2];
    const unsigned wy1 = (unsigned)cy.w1, wy0 = 256u - wy1;
    unsigned out = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int k = ok[j] ? sx[j] - base : 0;
        const unsigned wx1 = (unsigned)w1[j], wx0 = 256u - wx1;
        const unsigned h0 = wx0 * pick_byte(a0, a1, a2, k) + wx1 * pick_byte(a0, a1, a2, k + step[j]);
        const unsigned h1 = wx0 * pick_byte(b0, b1, b2, k) + wx1 * pick_byte(b0, b1, b2, k + step[j]);
        const unsigned v = wy0 * h0 + wy1 * h1;
        out |= ((v + (1u << 15)) >> 16) << (8 * j);
    }
    uint8_t* d = fb + (size_t)(dst.y - border + ty) * aw + ax_begin + dc * 4;
    if (ok[0] && ok[1] && ok[2] && ok[3]) *(unsigned*)d = out;
    else {
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (ok[j]) d[j] = (uint8_t)(out >> (8 * j));
    }
}

// a.lo * b.lo + a.hi * b.hi + c on the 16-bit halves (v_dot2_u32_u16).
__device__ __forceinline__ unsigned udot2_u16(unsigned a, unsigned b, unsigned c) {
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_udot2(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b), c, false);
}

// Vertical blend of two parked rows of horizontal sums (four 16-bit sums in a uint2 each) with the Q8 weight w1 of
// the lower row: ((256 - w1) * a + w1 * b + 2^15) >> 16 per pixel, four pixels packed into a dword.  The sums of a
// pixel are paired into one register (v_perm), one dot product per pixel does the blend and the rounding, and two
// more v_perm collect byte 2 of the four results.
__device__ __forceinline__ unsigned blend_rows_q8(uint2 a, uint2 b, unsigned w1) {
    const unsigned wy = (256u - w1) | (w1 << 16);
    const unsigned r0 = udot2_u16(__builtin_amdgcn_perm(b.x, a.x, 0x05040100u), wy, 1u << 15);
    const unsigned r1 = udot2_u16(__builtin_amdgcn_perm(b.x, a.x, 0x07060302u), wy, 1u << 15);
    const unsigned r2 = udot2_u16(__builtin_amdgcn_perm(b.y, a.y, 0x05040100u), wy, 1u << 15);
    const unsigned r3 = udot2_u16(__builtin_amdgcn_perm(b.y, a.y, 0x07060302u), wy, 1u << 15);
    return __builtin_amdgcn_perm(r1, r0, 0x0c0c0602u) | __builtin_amdgcn_perm(r3, r2, 0x06020c0cu);
}

// K2 (strip form, for scale factors whose four-pixel source span fits two dwords and whose kResizeRows output rows
// draw on at most kResizeSrcRows source rows): a wavefront produces kResizeRows output rows of a 256-pixel-wide
// column strip.  Everything that depends only on x - byte selectors, weights, clamp mask - is set up once per lane.
// Pass A filters every source row of the block horizontally, all row loads in flight together: three dwords per
// lane, the four left and four right samples lined up with v_perm, four Q8 sums packed into two dwords and parked
// in the wavefront's LDS slice.  Pass B blends two parked rows per output row.  A source row is filtered once per
// block, not once per output row that uses it.
constexpr int kResizeRows = 8;
constexpr int kResizeSrcRows = 12;

__global__ __launch_bounds__(256) void k_resize_strip(uint8_t* __restrict__ atlas, size_t atlas_bytes, int aw,
                                                      LevelInfo dst, int border, const CoefX* __restrict__ cxs,
                                                      const CoefY* __restrict__ cys, int ax_begin, int n_dwords) {
    __shared__ uint2 s_h[4][kResizeSrcRows][64];
    const int lane = threadIdx.x & 63;
    const int dc = blockIdx.x * 64 + lane;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rows = dst.h + 2 * border;
    const int ty0 = (blockIdx.y * 4 + wv) * kResizeRows;
    if (ty0 >= rows) return;
    const int ty1 = min(ty0 + kResizeRows, rows);
    // source rows of the block (uniform): [r_lo, r_hi]
    int r_lo = 0x7fffffff, r_hi = -1;
    for (int ty = ty0; ty < ty1; ty++) {
        const CoefY cy = cys[ty];
        r_lo = min(r_lo, min(cy.row0, cy.row1));
        r_hi = max(r_hi, max(cy.row0, cy.row1));
    }
    const int n_src = r_hi - r_lo + 1;
    const bool live = dc < n_dwords;
    const uint4 raw = *(const uint4*)(cxs + (live ? dc : 0) * 4);
    const unsigned cw[4] = {raw.x, raw.y, raw.z, raw.w};
    int sx[4];
    bool ok[4];
    unsigned wgt[4], stepmask = 0;
    int lo = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        sx[j] = (int)(short)(cw[j] & 0xffffu);
        const unsigned f = cw[j] >> 16;
        const unsigned w1 = f & 0x1ffu;
        wgt[j] = (256u - w1) | (w1 << 16);
        if ((f >> 14) & 1u) stepmask |= 0xffu << (8 * j);
        ok[j] = live && (f >> 15) != 0;
        if (ok[j]) lo = min(lo, sx[j]);
    }
    const bool any = lo != 0x7fffffff;
    const int base = any ? lo & ~3 : 0;
    unsigned sel = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) sel |= (unsigned)(ok[j] ? sx[j] - base : 0) << (8 * j);
    uint8_t* fb = atlas + (size_t)blockIdx.z * atlas_bytes;
    // pass A
    struct Row3 { unsigned d0, d1, d2; };
    Row3 src[kResizeSrcRows];
#pragma unroll
    for (int k = 0; k < kResizeSrcRows; k++)
        __builtin_memcpy(&src[k], fb + (size_t)min(r_lo + k, r_hi) * aw + base, 12);
#pragma unroll
    for (int k = 0; k < kResizeSrcRows; k++) {
        if (k < n_src) {
            const unsigned d0 = src[k].d0, d1 = src[k].d1, d2 = src[k].d2;
            const unsigned e0 = __builtin_amdgcn_alignbyte(d1, d0, 1), e1 = __builtin_amdgcn_alignbyte(d2, d1, 1);
            const unsigned A4 = __builtin_amdgcn_perm(d1, d0, sel);
            unsigned B4 = __builtin_amdgcn_perm(e1, e0, sel);
            B4 = (B4 & stepmask) | (A4 & ~stepmask);
            unsigned hh[4];
            // (a_j | b_j << 16) . (wx0_j | wx1_j << 16)
            const unsigned ab[4] = {__builtin_amdgcn_perm(B4, A4, 0x0c040c00u), __builtin_amdgcn_perm(B4, A4, 0x0c050c01u),
                                    __builtin_amdgcn_perm(B4, A4, 0x0c060c02u), __builtin_amdgcn_perm(B4, A4, 0x0c070c03u)};
#pragma unroll
            for (int j = 0; j < 4; j++) hh[j] = udot2_u16(ab[j], wgt[j], 0u);
            s_h[wv][k][lane] = make_uint2(hh[0] | (hh[1] << 16), hh[2] | (hh[3] << 16));  // each sum <= 255 * 256
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (!any) return;
    // pass B
    const bool whole = ok[0] && ok[1] && ok[2] && ok[3];
    for (int ty = ty0; ty < ty1; ty++) {
        const CoefY cy = cys[ty];
        const uint2 a = s_h[wv][cy.row0 - r_lo][lane], b = s_h[wv][cy.row1 - r_lo][lane];
        const unsigned out = blend_rows_q8(a, b, (unsigned)cy.w1);
        uint8_t* d = fb + (size_t)(dst.y - border + ty) * aw + ax_begin + dc * 4;
        if (whole) *(unsigned*)d = out;
        else {
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (ok[j]) d[j] = (uint8_t)(out >> (8 * j));
        }
    }
}

// K2 (strip form, narrow levels): the same work with G = 32 or 16 lanes across and 64 / G row blocks per wavefront, for
// levels whose width in dwords would leave most of a second 64-lane strip idle.  Everything that is per wavefront in
// k_resize_strip (row block, source row range, row coefficients) is per lane group here.
template <int G>
__global__ __launch_bounds__(256) void k_resize_strip_sub(uint8_t* __restrict__ atlas, size_t atlas_bytes, int aw,
                                                          LevelInfo dst, int border, const CoefX* __restrict__ cxs,
                                                          const CoefY* __restrict__ cys, int ax_begin, int n_dwords) {
    __shared__ uint2 s_h[4][kResizeSrcRows][64];
    constexpr int kBlocks = 64 / G;  // row blocks per wavefront
    const int lane = threadIdx.x & 63;
    const int gl = lane % G, sub = lane / G;
    const int dc = blockIdx.x * G + gl;
    const int wv = threadIdx.x >> 6;
    const int rows = dst.h + 2 * border;
    const int ty0 = ((blockIdx.y * 4 + wv) * kBlocks + sub) * kResizeRows;
    const bool rows_ok = ty0 < rows;
    if (__ballot(rows_ok) == 0ull) return;
    const int ty1 = min(ty0 + kResizeRows, rows);
    CoefY cyv[kResizeRows];
    int r_lo = 0x7fffffff, r_hi = -1;
#pragma unroll
    for (int t = 0; t < kResizeRows; t++) {
        cyv[t] = cys[min(ty0 + t, rows - 1)];  // rows past the end repeat the last one: the range is unchanged
        r_lo = min(r_lo, min(cyv[t].row0, cyv[t].row1));
        r_hi = max(r_hi, max(cyv[t].row0, cyv[t].row1));
    }
    const int n_src = r_hi - r_lo + 1;
    const bool live = rows_ok && dc < n_dwords;
    const uint4 raw = *(const uint4*)(cxs + (live ? dc : 0) * 4);
    const unsigned cw[4] = {raw.x, raw.y, raw.z, raw.w};
    int sx[4];
    bool ok[4];
    unsigned wgt[4], stepmask = 0;
    int lo = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        sx[j] = (int)(short)(cw[j] & 0xffffu);
        const unsigned f = cw[j] >> 16;
        const unsigned w1 = f & 0x1ffu;
        wgt[j] = (256u - w1) | (w1 << 16);
        if ((f >> 14) & 1u) stepmask |= 0xffu << (8 * j);
        ok[j] = live && (f >> 15) != 0;
        if (ok[j]) lo = min(lo, sx[j]);
    }
    const bool any = lo != 0x7fffffff;
    const int base = any ? lo & ~3 : 0;
    unsigned sel = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) sel |= (unsigned)(ok[j] ? sx[j] - base : 0) << (8 * j);
    uint8_t* fb = atlas + (size_t)blockIdx.z * atlas_bytes;
    struct Row3 { unsigned d0, d1, d2; };
    Row3 src[kResizeSrcRows];
#pragma unroll
    for (int k = 0; k < kResizeSrcRows; k++)
        __builtin_memcpy(&src[k], fb + ((unsigned)__umul24((unsigned)min(r_lo + k, r_hi), (unsigned)aw) + (unsigned)base), 12);
#pragma unroll
    for (int k = 0; k < kResizeSrcRows; k++) {
        if (k < n_src) {
            const unsigned d0 = src[k].d0, d1 = src[k].d1, d2 = src[k].d2;
            const unsigned e0 = __builtin_amdgcn_alignbyte(d1, d0, 1), e1 = __builtin_amdgcn_alignbyte(d2, d1, 1);
            const unsigned A4 = __builtin_amdgcn_perm(d1, d0, sel);
            unsigned B4 = __builtin_amdgcn_perm(e1, e0, sel);
            B4 = (B4 & stepmask) | (A4 & ~stepmask);
            unsigned hh[4];
            const unsigned ab[4] = {__builtin_amdgcn_perm(B4, A4, 0x0c040c00u), __builtin_amdgcn_perm(B4, A4, 0x0c050c01u),
                                    __builtin_amdgcn_perm(B4, A4, 0x0c060c02u), __builtin_amdgcn_perm(B4, A4, 0x0c070c03u)};
#pragma unroll
            for (int j = 0; j < 4; j++) hh[j] = udot2_u16(ab[j], wgt[j], 0u);
            s_h[wv][k][lane] = make_uint2(hh[0] | (hh[1] << 16), hh[2] | (hh[3] << 16));
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (!any) return;
    const bool whole = ok[0] && ok[1] && ok[2] && ok[3];
#pragma unroll
    for (int t = 0; t < kResizeRows; t++) {
        const int ty = ty0 + t;
        if (ty >= ty1) break;
        const CoefY cy = cyv[t];
        const uint2 a = s_h[wv][cy.row0 - r_lo][lane], b = s_h[wv][cy.row1 - r_lo][lane];
        const unsigned out = blend_rows_q8(a, b, (unsigned)cy.w1);
        uint8_t* d = fb + (size_t)(dst.y - border + ty) * aw + ax_begin + dc * 4;
        if (whole) *(unsigned*)d = out;
        else {
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (ok[j]) d[j] = (uint8_t)(out >> (8 * j));
        }
    }
}

// K1+K2 in one pass (see pyramid_stream.h): one workgroup per frame walks the image top to bottom in steps; in a step
// level 0 converts its next rows to gray and every level l >= 1 emits the rows whose two source rows of level l-1
// were completed by earlier steps, reading them from level l-1's LDS ring.  Each emitted row goes to the atlas (with
// its reflect-101 copy in the border where it has one) and into the level's own ring for the level above.  One
// barrier per step: a step only reads ring rows that earlier steps wrote, and the host has checked that no row
// written in a step shares a slot with a row the same step reads.
// The resize arithmetic is that of k_resize_strip: the four left and four right samples of a lane's four pixels
// lined up with v_perm from three source dwords, Q8 horizontal sums by v_dot2_u32_u16, vertical blend and rounding
// by blend_rows_q8 - here with both source rows filtered per output row (the filtered rows are not kept: a ring of
// 16-bit sums would double the LDS).
__device__ __forceinline__ uint2 hfilter_q8(unsigned d0, unsigned d1, unsigned d2, unsigned sel, unsigned stepmask,
                                            const unsigned (&wgt)[4]) {
    const unsigned e0 = __builtin_amdgcn_alignbyte(d1, d0, 1), e1 = __builtin_amdgcn_alignbyte(d2, d1, 1);
    const unsigned A4 = __builtin_amdgcn_perm(d1, d0, sel);
    unsigned B4 = __builtin_amdgcn_perm(e1, e0, sel);
    B4 = (B4 & stepmask) | (A4 & ~stepmask);
    const unsigned h0 = udot2_u16(__builtin_amdgcn_perm(B4, A4, 0x0c040c00u), wgt[0], 0u);
    const unsigned h1 = udot2_u16(__builtin_amdgcn_perm(B4, A4, 0x0c050c01u), wgt[1], 0u);
    const unsigned h2 = udot2_u16(__builtin_amdgcn_perm(B4, A4, 0x0c060c02u), wgt[2], 0u);
    const unsigned h3 = udot2_u16(__builtin_amdgcn_perm(B4, A4, 0x0c070c03u), wgt[3], 0u);
    return make_uint2(h0 | (h1 << 16), h2 | (h3 << 16));  // each sum <= 255 * 256
}

// Work is bound to threads once, so that nothing a lane needs in a step comes from global memory at the moment it
// is needed: every thread OWNS up to kStreamColsPerThread columns (aligned dwords) of the levels >= 1 - the list of
// all levels' columns dealt round-robin, so each thread has a wide and a narrow level - and keeps their coefficient
// records in registers for the whole frame; the row records of the rows a step emits are fetched one step ahead by a
// few threads and handed over through a small LDS cache; the image bytes level 0 converts in step s + 1 are loaded
// into registers at the top of step s.  What is left on the critical path of a step is LDS and arithmetic.
// (four wavefronts per SIMD = four frames per CU: with 792 frames on 256 CUs every frame must be resident at once)
// TAIL: plan level 0 is a pyramid level an earlier kernel has already written: `img` then points at that level's
// interior in the atlas (one channel, stride = atlas width, frame_stride = atlas bytes); its rows only enter the ring.
template <bool TAIL>
__global__ __launch_bounds__(kStreamThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_pyramid_stream(const uint8_t* __restrict__ img, size_t frame_stride,
                                                                   int stride, int ch, uint8_t* __restrict__ atlas,
                                                                   size_t atlas_bytes, StreamParams P,
                                                                   const StreamCol* __restrict__ cols,
                                                                   const StreamRow* __restrict__ rows,
                                                                   const short* __restrict__ sched) {
    extern __shared__ __align__(16) unsigned s_ring[];
    const int tid = threadIdx.x;
    const uint8_t* frame = img + (size_t)blockIdx.x * frame_stride;
    uint8_t* plane = atlas + (size_t)blockIdx.x * atlas_bytes;
    const int aw = P.aw, nl = P.nlevels, rc_rows = P.rc_rows;
    // behind the rings: [2] x { row records [nl][rc_rows], row ranges [kMaxLevels] as (first, count) }
    StreamRow* s_rows = (StreamRow*)(s_ring + P.ring_dwords);
    int2* s_range = (int2*)(s_rows + 2 * nl * rc_rows);

    // ---- the columns this thread owns
    StreamCol col[kStreamColsPerThread];
    int c_level[kStreamColsPerThread], c_src[kStreamColsPerThread], c_src_pitch[kStreamColsPerThread],
        c_own[kStreamColsPerThread], c_own_pitch[kStreamColsPerThread], c_gx[kStreamColsPerThread],
        c_gy[kStreamColsPerThread];
#pragma unroll
    for (int k = 0; k < kStreamColsPerThread; k++) {
        const int f = tid + kStreamThreads * k;
        c_level[k] = 0;   // level 0 owns no column here: "nothing to do"
        if (f < P.total_cols) {
            int l = 1;
            while (l + 1 < nl && f >= P.L[l + 1].col_off) l++;
            col[k] = cols[f];
            if (col[k].ok != 0u) {
                const int j = f - P.L[l].col_off;
                c_level[k] = l;
                c_src[k] = P.L[l - 1].lds_off + col[k].j0;
                c_src_pitch[k] = P.L[l - 1].pitch;
                c_own[k] = P.L[l].depth > 0 ? P.L[l].lds_off + j : -1;
                c_own_pitch[k] = P.L[l].pitch;
                c_gx[k] = P.L[l].ax_begin + 4 * j;
                c_gy[k] = P.L[l].ay;
            }
        }
    }

    // ---- level 0 items of a step: 16-pixel groups inside the image, single pixels in the reflected border
    const StreamLevel& L0 = P.L[0];
    const int rx = L0.ax, x_first = rx - P.border, x_last = rx + L0.w + P.border;
    static_assert(kStreamGrayItems == 2, "two named register sets below");
    // the image bytes of one of this thread's two 16-pixel groups (a, b, c: 48 bytes of BGR, or a: 16 gray bytes) and
    // of one of its two border pixels (B | G << 8 | R << 16, or the gray value)
    struct GrayRegs { uint4 a, b, c; unsigned e; };
    GrayRegs g0 = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0), 0u}, g1 = g0;
    auto gray_fetch_item = [&](int s, int m, GrayRegs gr) __attribute__((always_inline)) {
        const int y0 = sched[(size_t)s * nl * 2], ny = sched[(size_t)s * nl * 2 + 1] - y0;
        const int i = tid + kStreamThreads * m;
        if (i < ny * P.n_in) {
            const int ry = (int)(__umul24((unsigned)i, P.magic_in) >> 20);
            const int q = P.q_in0 + i - ry * P.n_in;
            const int sx = L0.ax_begin + 16 * q - rx;
            const uint8_t* row = frame + (size_t)(y0 + ry) * stride;
            if (ch == 1) gr.a = load_u128_any(row + sx);
            else {
                const uint8_t* sp = row + (size_t)sx * 3;
                gr.a = load_u128_any(sp);
                gr.b = load_u128_any(sp + 16);
                gr.c = load_u128_any(sp + 32);
            }
        }
        if (i < ny * P.n_edge) {
            const int ry = (int)(__umul24((unsigned)i, P.magic_edge) >> 20);
            const int e = i - ry * P.n_edge;
            const int ax = L0.ax_begin + (e < P.left_edge ? e : e - P.left_edge + P.q_in1 * 16);
            gr.e = 0u;
            if (ax >= x_first && ax < x_last) {
                const uint8_t* sp = frame + (size_t)(y0 + ry) * stride + (size_t)reflect101(ax - rx, L0.w) * ch;
                gr.e = ch == 1 ? (unsigned)sp[0] : ((unsigned)sp[0] | ((unsigned)sp[1] << 8) | ((unsigned)sp[2] << 16));
            }
        }
        return gr;
    };
    auto gray_fetch = [&](int s) __attribute__((always_inline)) {   // the image bytes of step s into registers
        g0 = gray_fetch_item(s, 0, g0);
        g1 = gray_fetch_item(s, 1, g1);
    };
    // the row records and row ranges of step s, fetched by the first nl * rc_rows (and nl) threads
    struct RowRegs { StreamRow row; int2 range; };
    RowRegs nx = {StreamRow{0u, 0, -1, 0}, make_int2(0, 0)};
    auto rows_fetch = [&](int s, RowRegs r) __attribute__((always_inline)) {
        if (tid < nl * rc_rows) {
            const int l = tid / rc_rows, i = tid - l * rc_rows;
            const int y0 = sched[(size_t)(s * nl + l) * 2], y1 = sched[(size_t)(s * nl + l) * 2 + 1];
            if (y0 + i < y1) r.row = rows[P.L[l].row_off + y0 + i];
        }
        if (tid < nl) {
            const int y0 = sched[(size_t)(s * nl + tid) * 2], y1 = sched[(size_t)(s * nl + tid) * 2 + 1];
            r.range = make_int2(y0, y1 - y0);
        }
        return r;
    };
    auto rows_publish = [&](int s, const RowRegs r) __attribute__((always_inline)) {
        if (tid < nl * rc_rows) s_rows[(s & 1) * nl * rc_rows + tid] = r.row;
        if (tid < nl) s_range[(s & 1) * kMaxLevels + tid] = r.range;
    };
    gray_fetch(0);
    nx = rows_fetch(0, nx);
    rows_publish(0, nx);
    __syncthreads();

    for (int s = 0; s < P.n_steps; s++) {
        const StreamRow* rcache = s_rows + (s & 1) * nl * rc_rows;
        const int2* range = s_range + (s & 1) * kMaxLevels;
        const bool more = s + 1 < P.n_steps;
        if (more) nx = rows_fetch(s + 1, nx);   // in flight while this step works
        // ---- level 0: the bytes fetched during the previous step
        auto gray_emit_item = [&](int m, const GrayRegs& gr) __attribute__((always_inline)) {
            const int y0 = range[0].x, ny = range[0].y;
            const int i = tid + kStreamThreads * m;
            if (i < ny * P.n_in) {
                const int ry = (int)(__umul24((unsigned)i, P.magic_in) >> 20);
                const int q = P.q_in0 + i - ry * P.n_in, y = y0 + ry;
                const StreamRow r = rcache[ry];
                const int ax = L0.ax_begin + 16 * q;
                uint4 out;
                if (ch == 1) out = gr.a;
                else {
                    const uint4 v0 = gr.a, v1 = gr.b, v2 = gr.c;
                    out.x = gray4(v0.x, v0.y, v0.z);
                    out.y = gray4(v0.w, v1.x, v1.y);
                    out.z = gray4(v1.z, v1.w, v2.x);
                    out.w = gray4(v2.y, v2.z, v2.w);
                }
                if (!TAIL) {
                    __builtin_memcpy(plane + (size_t)(L0.ay + y) * aw + ax, &out, 16);
                    if (r.mirror >= 0) __builtin_memcpy(plane + (size_t)r.mirror * aw + ax, &out, 16);
                }
                if (L0.depth > 0)
                    *(uint4*)(s_ring + L0.lds_off + __umul24((r.slots >> 16) & 255u, (unsigned)L0.pitch) + 4 * q) = out;
            }
            if (i < ny * P.n_edge) {
                const int ry = (int)(__umul24((unsigned)i, P.magic_edge) >> 20);
                const int e = i - ry * P.n_edge, y = y0 + ry;
                const int ax = L0.ax_begin + (e < P.left_edge ? e : e - P.left_edge + P.q_in1 * 16);
                if (ax >= x_first && ax < x_last) {
                    const StreamRow r = rcache[ry];
                    const uint8_t v = (uint8_t)(ch == 1 ? gr.e : gray_of(gr.e & 255u, (gr.e >> 8) & 255u, (gr.e >> 16) & 255u));
                    if (!TAIL) {
                        plane[(size_t)(L0.ay + y) * aw + ax] = v;
                        if (r.mirror >= 0) plane[(size_t)r.mirror * aw + ax] = v;
                    }
                    if (L0.depth > 0)
                        ((uint8_t*)(s_ring + L0.lds_off + __umul24((r.slots >> 16) & 255u, (unsigned)L0.pitch)))[ax - L0.ax_begin] = v;
                }
            }
        };
        gray_emit_item(0, g0);
        gray_emit_item(1, g1);
        if (more) gray_fetch(s + 1);   // the loads return while the levels above are worked on
        // ---- levels >= 1: this thread's columns, every row their level emits in this step
        auto emit_column = [&](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value;   // a constant: the column records stay in registers
            const int l = c_level[k];
            if (l == 0) return;
            const int y0 = range[l].x, ny = range[l].y;
            const StreamRow* rl = rcache + l * rc_rows;
            for (int ry = 0; ry < ny; ry++) {
                const StreamRow r = rl[ry];
                const unsigned* ra = s_ring + c_src[k] + __umul24(r.slots & 255u, (unsigned)c_src_pitch[k]);
                const unsigned* rb = s_ring + c_src[k] + __umul24((r.slots >> 8) & 255u, (unsigned)c_src_pitch[k]);
                const uint2 ha = hfilter_q8(ra[0], ra[1], ra[2], col[k].sel, col[k].stepmask, col[k].wgt);
                const uint2 hb = hfilter_q8(rb[0], rb[1], rb[2], col[k].sel, col[k].stepmask, col[k].wgt);
                const unsigned out = blend_rows_q8(ha, hb, (unsigned)r.w1);
                uint8_t* d = plane + (size_t)(c_gy[k] + y0 + ry) * aw + c_gx[k];
                uint8_t* m = r.mirror >= 0 ? plane + (size_t)r.mirror * aw + c_gx[k] : nullptr;
                if (col[k].ok == 15u) {
                    *(unsigned*)d = out;
                    if (m) *(unsigned*)m = out;
                } else {
#pragma unroll
                    for (int t = 0; t < 4; t++)
                        if (col[k].ok & (1u << t)) {
                            d[t] = (uint8_t)(out >> (8 * t));
                            if (m) m[t] = (uint8_t)(out >> (8 * t));
                        }
                }
                if (c_own[k] >= 0) s_ring[c_own[k] + __umul24((r.slots >> 16) & 255u, (unsigned)c_own_pitch[k])] = out;
            }
        };
        static_assert(kStreamColsPerThread == 3, "one call per owned column");
        emit_column(std::integral_constant<int, 0>{});
        emit_column(std::integral_constant<int, 1>{});
        emit_column(std::integral_constant<int, 2>{});
        if (more) rows_publish(s + 1, nx);   // into the other buffer: nobody reads it before the barrier
        __syncthreads();
    }
}

// K3+K4a fused: FAST-9/16 score and 3x3 strict non-max suppression of one 64 x kTileH tile, never leaving LDS.
//   phase 1  every pixel of the tile (plus a one-pixel ring for the suppression) takes the four-point compass test,
//            four pixels per lane from whole staged dwords, evaluated on 64-bit lane masks with scalar logic; the
//            pixels that pass (under a tenth) are queued.  Every wavefront has a queue of its own: its fill count
//            is a scalar register, appending needs no atomic, and phase 2 can start without a workgroup barrier;
//   phase 2  the queued pixels get their cornerScore with full lanes; the score itself says whether the
//            nine-contiguous test holds;
//   phase 3  suppression + border rule for the queued corners -> one mask word per tile row; the score byte is
//            stored only where a keypoint survives (all the emission reads).  DENSE also stores every score of
//            the tile (tests).
constexpr int kFPitch = 80;              // LDS bytes per staged row: 64 + 2*4 halo + 3 alignment, rounded up
constexpr int kFRows = kTileH + 8;
constexpr int kSW = kTileW + 2, kSH = kTileH + 2;
constexpr int kFGroups = 18;             // staged dwords per row that can hold a scored pixel (66 columns + alignment)
constexpr int kQCols = 4 * kFGroups;     // columns a queue entry can name
constexpr int kFPassRows = 256 / kFGroups;                    // rows of dwords the 256 threads cover per pass (14)
constexpr int kFPasses = (kSH + kFPassRows - 1) / kFPassRows; // passes over the ring-grown tile (5)
constexpr int kQSeg = kFPasses * 64 * 4;                      // queue entries one wavefront can produce
// the last wavefront owns the highest rows of every pass, which the final pass does not reach
constexpr int kQTotal = 3 * kQSeg + (kFPasses - 1) * 64 * 4;
static_assert(kFPassRows * kFGroups <= 256 && (kFPasses - 1) * kFPassRows + (192 / kFGroups) >= kSH,
              "the last wavefront must have no row in the final pass");

// i / D for 0 <= i < 16384 and a small divisor (exact while i * (m * D - 2^20) < 2^20, which the assert bounds): one 24-bit
// multiply and a shift instead of the quarter-rate 32-bit multiply-high of a general division.
template <unsigned D>
__device__ __forceinline__ int div_small(int i) {
    constexpr unsigned m = ((1u << 20) + D - 1) / D;
    static_assert(m * D - (1u << 20) < 64, "multiplier error too large for i < 16384");
    return (int)(__umul24((unsigned)i, m) >> 20);
}

typedef short short2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ short2v as_short2v(unsigned w) { return __builtin_bit_cast(short2v, w); }
typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ushort2v as_ushort2v(unsigned w) { return __builtin_bit_cast(ushort2v, w); }

template <bool DENSE>
__global__ __launch_bounds__(256) void k_fast_nms(const uint8_t* __restrict__ atlas, uint8_t* __restrict__ score,
                                                  MaskWord* __restrict__ mask, size_t atlas_bytes, Geom g,
                                                  const Tile* __restrict__ tiles, int threshold, int frame0,
                                                  int* __restrict__ chunk_counts, int words_per_chunk) {
    __shared__ __align__(16) unsigned t32[kFRows * kFPitch / 4];
    __shared__ unsigned s32[(kSW * kSH + 3) / 4];
    __shared__ unsigned short queue[kQTotal];
    // a patch of equal pixels for the lanes of a pass that have no row: every read of the compass test (the centre
    // dword, its two neighbours, three rows up and down) finds zeros there, and equal pixels never pass
    __shared__ unsigned s_flat[2 * 3 * (kFPitch / 4) + 4];
    // the row masks and their packed scores are first needed in phase 3, when the staged pixels are dead: they share
    // the staging buffer, which brings a tile's LDS under 20 KB - eight tiles per CU instead of seven
    unsigned* s_mask = t32;
    unsigned* s_sc8 = t32 + 2 * kTileH;
    uint8_t* t = (uint8_t*)t32;
    uint8_t* st = (uint8_t*)s32;
    // workgroups go round-robin over the 8 XCDs (each with its own L2): give every XCD one contiguous run of tiles,
    // so the halo a tile shares with its neighbours is fetched into one L2 only
    const int per_xcd = (g.total_tiles + 7) >> 3;
    const int tile_id = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (tile_id >= g.total_tiles) return;
    const Tile tl = tiles[tile_id];
    const LevelInfo L = g.L[tl.level];
    const int b = blockIdx.y + frame0;
    const int aw = g.atlas_w;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lo = max(3, g.edge);
    const bool level_ok = g.edge <= 0 || (L.w > 2 * g.edge && L.h > 2 * g.edge);
    // pixels whose score can matter: the keypoint area [lo, w-lo) grown by one, inside the FAST domain [3, w-3)
    const int in = DENSE ? 3 : max(lo - 1, 3);
    const int xs = max(tl.x0 - 1, in), xe = min(tl.x0 + kTileW + 1, L.w - in);
    const int ys = max(tl.y0 - 1, in), ye = min(tl.y0 + kTileH + 1, L.h - in);
    MaskWord* mk = mask + (size_t)b * g.total_words + L.word_base + (tl.x0 >> 6);
    uint8_t* sc = score + (size_t)b * atlas_bytes + (size_t)(L.y + tl.y0) * aw + L.x + tl.x0;
    const bool empty = xs >= xe || ys >= ye || (!DENSE && !level_ok);
    if (empty && !DENSE) {
        if (tid < kTileH && tl.y0 + tid < L.h) mk[(size_t)(tl.y0 + tid) * L.words_per_row] = MaskWord{0ull, 0ull};
        return;
    }
    // stage the tile with a 4-pixel halo, aligned dwords (levels sit at least 32 pixels inside the atlas)
    const int ox = L.x + tl.x0 - 4, oy = L.y + tl.y0 - 4;
    const int abase = ox & ~3, off = ox & 3;
    const uint8_t* fb = atlas + (size_t)b * atlas_bytes;
    {
        // 16-byte segments, all loads in flight before the LDS stores.  A segment may run past the end of an atlas
        // row into the next one (or into the buffer's slack after the last row): those bytes belong to pixels
        // outside the level's padded rectangle, which no valid pixel reads.
        constexpr int kSegs = kFRows * (kFPitch / 16), kSteps = (kSegs + 255) / 256;
        uint4 tmp[kSteps];
#pragma unroll
        for (int k = 0; k < kSteps; k++) {
            const int i = min(tid + 256 * k, kSegs - 1);
            const int r = div_small<kFPitch / 16>(i), c = i - r * (kFPitch / 16);
            const int ay = min(oy + r, g.atlas_h - 1);
            tmp[k] = load_u128_any(fb + (size_t)ay * aw + abase + 16 * c);
        }
#pragma unroll
        for (int k = 0; k < kSteps; k++)
            if (tid + 256 * k < kSegs) ((uint4*)t32)[tid + 256 * k] = tmp[k];
    }
    for (int i = tid; i < (kSW * kSH + 3) / 4; i += 256) s32[i] = 0u;
    if (tid < 2 * 3 * (kFPitch / 4) + 4) s_flat[tid] = 0u;
    __syncthreads();
    constexpr int dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
    constexpr int dy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
    // phase 1: a lane takes the four pixels of one staged dword.  Thread -> (row r0 of a pass, dword gq) once; a pass
    // covers kFPassRows rows, so everything that depends on the row advances by a constant per pass.
    unsigned short* wq = queue + wv * kQSeg;   // this wavefront's queue
    int nq = 0;                                // its fill count (uniform)
    {
        const int nrows = empty ? 0 : ye - ys;
        const int r0 = div_small<kFGroups>(tid), gq = tid - __mul24(r0, kFGroups);
        // rows this lane still has when a pass starts at row rb: rows_left > rb (none for the threads beyond the
        // last whole row of dwords)
        const int rows_left = r0 < kFPassRows ? nrows - r0 : 0;
        // all five reads of a lane sit at non-negative offsets from one address: three rows up and one dword left
        constexpr int kUp = 3 * (kFPitch / 4) + 1;
        const unsigned* tr = t32 + __mul24(ys - tl.y0 + 4 + r0, kFPitch / 4) + gq - kUp;
        int code = __mul24(r0, kQCols) + 4 * gq;  // (row of the ring-grown tile, staged column)
        for (int rb = 0; rb < nrows; rb += kFPassRows, tr += kFPassRows * (kFPitch / 4), code += kFPassRows * kQCols) {
            // lanes without a row read the patch of equal pixels
            const unsigned* trc = rb < rows_left ? tr : s_flat;
            const unsigned C = trc[kUp], Lw = trc[kUp - 1], Rw = trc[kUp + 1];
            const unsigned P0 = trc[kUp + 3 * (kFPitch / 4)], P8 = trc[kUp - 3 * (kFPitch / 4)];
            const unsigned P4 = __builtin_amdgcn_alignbyte(Rw, C, 3), P12 = __builtin_amdgcn_alignbyte(C, Lw, 1);
            // any arc of nine covers two neighbouring compass points (ring positions 0, 4, 8, 12), one of {0, 8} and
            // one of {4, 12}: a corner needs min(max(p0, p8), max(p4, p12)) above the band or
            // max(min(p0, p8), min(p4, p12)) below it.  Both sides fold into ONE compare per pixel, so the scalar unit
            // (one per CU, shared by the four SIMDs) only counts bits.  Columns outside the scored range are dropped
            // when the queue is read (phase 2).
            // Two pixels per instruction, without unpacking.  Even bytes: one AND leaves them as two 16-bit values;
            // unsigned packed min / max and a saturating subtract (max(difference, 0) is all a compare against a
            // threshold >= 0 needs) do the rest.  Odd bytes: the 16-bit halves are used AS THEY ARE.  The upper byte
            // decides every unsigned 16-bit minimum and maximum, so the upper byte of each result is exact; the lower
            // byte (the neighbouring pixel) only perturbs a difference by less than one unit of the upper byte.
            // A pixel whose margin exceeds the threshold still exceeds threshold * 256 (margin >= t + 1 gives at least
            // (t + 1) * 256 - 255); one whose margin equals the threshold may pass too, which only costs its score.
            unsigned long long pass[4];
            bool mine[4];
            int total = 0;
#pragma unroll
            for (int hsel = 0; hsel < 2; hsel++) {
                const unsigned keep = hsel ? 0xffffffffu : 0x00ff00ffu;
                const ushort2v v = as_ushort2v(C & keep);
                const ushort2v p0 = as_ushort2v(P0 & keep), p4 = as_ushort2v(P4 & keep);
                const ushort2v p8 = as_ushort2v(P8 & keep), p12 = as_ushort2v(P12 & keep);
                const ushort2v up = __builtin_elementwise_sub_sat(
                    __builtin_elementwise_min(__builtin_elementwise_max(p0, p8), __builtin_elementwise_max(p4, p12)), v);
                const ushort2v dn = __builtin_elementwise_sub_sat(
                    v, __builtin_elementwise_max(__builtin_elementwise_min(p0, p8), __builtin_elementwise_min(p4, p12)));
                const ushort2v m = __builtin_elementwise_max(up, dn);
                const unsigned thr16 = (unsigned)(hsel ? threshold << 8 : threshold);
                mine[hsel] = (unsigned)m.x > thr16;
                mine[hsel + 2] = (unsigned)m.y > thr16;
                pass[hsel] = __ballot(mine[hsel]);
                pass[hsel + 2] = __ballot(mine[hsel + 2]);
                total += __popcll(pass[hsel]) + __popcll(pass[hsel + 2]);
            }
            if (total == 0) continue;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                // rank among the passing lanes below this one, counted by the mask-bit-count pair
                if (mine[j])
                    wq[nq + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(pass[j] >> 32),
                                                           __builtin_amdgcn_mbcnt_lo((unsigned)pass[j], 0u))] =
                        (unsigned short)(code + j);
                nq += __popcll(pass[j]);
            }
        }
    }
    // the queue was written and is read by this wavefront only: order its LDS traffic, no workgroup barrier
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // phase 2: cornerScore of the queued pixels: max(threshold, best arc-of-9 minimum of |v - p|) - 1
    for (int q = lane; q < nq; q += 64) {
        const int code = wq[q];
        const int rr = div_small<kQCols>(code);
        const int x = tl.x0 - 4 - off + (code - __mul24(rr, kQCols));  // level x of the queued pixel
        if (x < xs || x >= xe) {
            wq[q] = 0xffffu;
            continue;
        }
        const int r = ys + rr - tl.y0 + 1, c = x - tl.x0 + 1;   // position in the ring-grown score tile
        const int pos = __mul24(r, kSW) + c;
        wq[q] = (unsigned short)pos;
        const uint8_t* ctr = t + __mul24(r + 3, kFPitch) + c + 3 + off;
        const int v = ctr[0];
        // Ring pixels as half-precision numbers, two per register: 0x6400 | p is 1024 + p exactly, so floating-point
        // minima and maxima order them like the bytes, and gfx950 has a packed three-operand form of both
        // (v_pk_minimum3_f16 / v_pk_maximum3_f16: four comparisons per instruction).  Register j holds ring positions
        // (j, j + 8); position j + 8 .. j + 15 is the same register with its halves swapped, which the instruction's
        // operand selectors do for free.  m3[j]: minima over 3 consecutive positions from j (and from j + 8);
        // m9 = min3(m3[j], m3[j + 3], m3[j + 6]): the arc of nine from j (and from j + 8).
        // bright margin = max over arcs of (min p) - v, dark margin = v - min over arcs of (max p).
        half2v Q[16], lo3[16], hi3[16];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const unsigned ra = ctr[dy[k] * kFPitch + dx[k]], rb = ctr[dy[k + 8] * kFPitch + dx[k + 8]];
            Q[k] = __builtin_bit_cast(half2v, __builtin_amdgcn_perm(rb, ra, 0x0c040c00u) | 0x64006400u);
            Q[k + 8] = Q[k].yx;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            lo3[k] = __builtin_elementwise_minimum(__builtin_elementwise_minimum(Q[k], Q[k + 1]), Q[k + 2]);
            hi3[k] = __builtin_elementwise_maximum(__builtin_elementwise_maximum(Q[k], Q[k + 1]), Q[k + 2]);
            lo3[k + 8] = lo3[k].yx;
            hi3[k + 8] = hi3[k].yx;
        }
        half2v lo9[8], hi9[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            lo9[k] = __builtin_elementwise_minimum(__builtin_elementwise_minimum(lo3[k], lo3[k + 3]), lo3[k + 6]);
            hi9[k] = __builtin_elementwise_maximum(__builtin_elementwise_maximum(hi3[k], hi3[k + 3]), hi3[k + 6]);
        }
        half2v up = __builtin_elementwise_maximum(lo9[6], lo9[7]), dn = __builtin_elementwise_minimum(hi9[6], hi9[7]);
#pragma unroll
        for (int k = 0; k < 6; k += 3) {
            up = __builtin_elementwise_maximum(__builtin_elementwise_maximum(lo9[k], lo9[k + 1]),
                                               __builtin_elementwise_maximum(lo9[k + 2], up));
            dn = __builtin_elementwise_minimum(__builtin_elementwise_minimum(hi9[k], hi9[k + 1]),
                                               __builtin_elementwise_minimum(hi9[k + 2], dn));
        }
        up = __builtin_elementwise_maximum(up, up.yx);
        dn = __builtin_elementwise_minimum(dn, dn.yx);
        // the low byte of 0x6400 | p is p again
        const int brightest_min = (int)(__builtin_bit_cast(unsigned, up) & 0xffu);
        const int darkest_max = (int)(__builtin_bit_cast(unsigned, dn) & 0xffu);
        const int best = max(threshold, max(brightest_min - v, v - darkest_max));
        if (best > threshold) st[pos] = (uint8_t)(best - 1);  // an arc of nine clears the band <=> a corner
    }
    __syncthreads();
    if (tid < 2 * kTileH) {
        s_mask[tid] = 0u;
        s_sc8[tid] = 0u;
    }
    __syncthreads();
    // phase 3: suppression and border rule for the queued corners of the tile proper; bits gathered per tile row
    for (int q = lane; q < nq; q += 64) {
        const int pos = wq[q];
        if (pos == 0xffff) continue;
        const int c = st[pos];
        const int r = div_small<kSW>(pos) - 1, cc = pos - __mul24(r + 1, kSW) - 1;
        bool kp = false;
        if (c != 0 && r >= 0 && r < kTileH && cc >= 0 && cc < kTileW) {
            const int y = tl.y0 + r, x = tl.x0 + cc;
            if (level_ok && x >= lo && x < L.w - lo && y >= lo && y < L.h - lo) {
                const uint8_t* s = st + pos;
                // all eight neighbours read at once (no short circuit: one LDS round trip instead of up to eight)
                const int n0 = s[1], n1 = s[-1], n2 = s[-kSW - 1], n3 = s[-kSW], n4 = s[-kSW + 1], n5 = s[kSW - 1],
                          n6 = s[kSW], n7 = s[kSW + 1];
                kp = c > max(max(max(n0, n1), max(n2, n3)), max(max(n4, n5), max(n6, n7)));
            }
        }
        if (kp) atomicOr(&s_mask[2 * r + (cc >> 5)], 1u << (cc & 31));
        else wq[q] = 0xffffu;
    }
    __syncthreads();
    // the scores of the first eight keypoints of a row word travel with the word; later ones (rare) go to the plane
    for (int q = lane; q < nq; q += 64) {
        const int pos = wq[q];
        if (pos == 0xffff) continue;
        const int r = div_small<kSW>(pos) - 1, cc = pos - __mul24(r + 1, kSW) - 1;
        const unsigned long long row = (unsigned long long)s_mask[2 * r] | ((unsigned long long)s_mask[2 * r + 1] << 32);
        const int rank = __popcll(row & ((1ull << cc) - 1ull));
        const unsigned c = st[pos];
        if (rank < 8) atomicOr(&s_sc8[2 * r + (rank >> 2)], c << (8 * (rank & 3)));
        else if (!DENSE) sc[(size_t)r * aw + cc] = (uint8_t)c;
    }
    __syncthreads();
    if (tid < 64) {  // kTileH <= 64: one wavefront writes the tile's rows and totals their keypoints per emission run
        const bool row = tid < kTileH && tl.y0 + tid < L.h;
        int cnt = 0, run = -1;
        if (row) {
            MaskWord w;
            w.bits = (unsigned long long)s_mask[2 * tid] | ((unsigned long long)s_mask[2 * tid + 1] << 32);
            w.scores = (unsigned long long)s_sc8[2 * tid] | ((unsigned long long)s_sc8[2 * tid + 1] << 32);
            mk[(size_t)(tl.y0 + tid) * L.words_per_row] = w;
            cnt = __popcll(w.bits);
            run = (L.word_base + (tl.y0 + tid) * L.words_per_row + (tl.x0 >> 6)) / words_per_chunk;
        }
        if (chunk_counts) {
            // rows of one tile fall into very few runs: one atomic per run present
            unsigned long long todo = __ballot(cnt > 0);
            while (todo) {
                const int leader = __ffsll((long long)todo) - 1;
                const int r0 = __builtin_amdgcn_readlane(run, leader);
                const bool mine = cnt > 0 && run == r0;
                int sum = mine ? cnt : 0;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
                if (tid == leader) atomicAdd(&chunk_counts[b * kEmitMaxChunks + r0], sum);
                todo &= ~__ballot(mine);
            }
        }
    }
    if (DENSE) {
        for (int r = wv; r < kTileH; r += 4) {
            const int y = tl.y0 + r, x = tl.x0 + lane;
            if (y < L.h && x < L.w) sc[(size_t)r * aw + lane] = st[(r + 1) * kSW + lane + 1];
        }
    }
}

// Inclusive prefix sum over the 64 lanes (DPP row shifts, then the two cross-row broadcasts).
__device__ __forceinline__ int wave_scan_incl(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast31 -> rows 2, 3
    return v;
}

// K4b: ordered emission, in three small kernels so that a frame is not tied to one workgroup.  The mask words of a
// frame (level-major raster order) are cut into `chunks` equal runs:
//   (k_fast_nms)   totals the keypoints of its tile's rows into the per-run counters;
//   k_emit_write   a workgroup per run: its base rank is the sum of the earlier runs' counts, its four wavefronts own
//                  a quarter of the run each and append their corners at their exact rank, so the candidate list has
//                  the order cv::FAST emits; per-level counts and score histograms are accumulated on the side;
//   k_emit_finish  totals, and from the histograms the score of the (2*quota)-th best corner of every level: the
//                  value KeyPointsFilter::retainBest(2*quota) keeps everything at or above.

__global__ __launch_bounds__(256) void k_emit_write(const MaskWord* __restrict__ mask, const uint8_t* __restrict__ score,
                                                    size_t atlas_bytes, Geom g, int chunks, int per,
                                                    const int* __restrict__ chunk_counts, Cand* __restrict__ cand,
                                                    int cap, int* __restrict__ level_counts, int* __restrict__ hist_g) {
    __shared__ int hist[kMaxLevels][256];
    __shared__ int lcount[kMaxLevels];
    __shared__ int wave_tot[4];
    const int b = blockIdx.y, c = blockIdx.x;
    const int tid = threadIdx.x;
    for (int i = tid; i < kMaxLevels * 256; i += 256) (&hist[0][0])[i] = 0;
    if (tid < kMaxLevels) lcount[tid] = 0;
    const MaskWord* mk = mask + (size_t)b * g.total_words;
    const uint8_t* sc = score + (size_t)b * atlas_bytes;
    Cand* out = cand + (size_t)b * cap;
    const int lane = tid & 63, wv = tid >> 6;
    const int c0 = min(c * per, g.total_words);
    const int quarter = per >> 2;  // per is a multiple of 256
    const int w0 = min(c0 + wv * quarter, g.total_words), w1 = min(min(w0 + quarter, c0 + per), g.total_words);
    int mine = 0;
    for (int base = w0; base < w1; base += 64) {
        const int word = base + lane;
        mine += word < w1 ? __popcll(mk[word].bits) : 0;
    }
    mine = wave_scan_incl(mine);
    if (lane == 63) wave_tot[wv] = mine;
    int before = 0;  // corners of the earlier runs of this frame
    for (int k = lane; k < c; k += 64) before += chunk_counts[b * kEmitMaxChunks + k];
    before = __builtin_amdgcn_readlane(wave_scan_incl(before), 63);
    __syncthreads();
    int running = before;
    for (int k = 0; k < wv; k++) running += wave_tot[k];
    for (int base = w0; base < w1; base += 64) {
        const int word = base + lane;
        const MaskWord mw = word < w1 ? mk[word] : MaskWord{0ull, 0ull};
        unsigned long long m = mw.bits;
        const int cnt = __popcll(m);
        const int incl = wave_scan_incl(cnt);
        int pos = running + incl - cnt;
        running += __builtin_amdgcn_readlane(incl, 63);
        if (m) {
            int l = 0;
            while (l + 1 < g.nlevels && word >= g.L[l + 1].word_base) l++;
            const LevelInfo L = g.L[l];
            const int r = word - L.word_base;
            const int y = r / L.words_per_row;
            const int x0 = (r - y * L.words_per_row) * 64;
            atomicAdd(&lcount[l], cnt);
            int rank = 0;
            while (m) {
                const int bit = __ffsll((long long)m) - 1;
                m &= m - 1;
                const int x = x0 + bit;
                const int s = rank < 8 ? (int)((mw.scores >> (8 * rank)) & 255ull)
                                       : (int)sc[(size_t)(L.y + y) * g.atlas_w + L.x + x];
                rank++;
                if (pos < cap) {
                    Cand cd;
                    cd.x = (uint16_t)x;
                    cd.y = (uint16_t)y;
                    cd.score = (uint8_t)s;
                    cd.level = (uint8_t)l;
                    cd.pad = 0;
                    cd.harris = 0.f;
                    cd.angle = -1.f;
                    out[pos] = cd;
                }
                atomicAdd(&hist[l][s], 1);
                pos++;
            }
        }
    }
    __syncthreads();
    int* hg = hist_g + (size_t)b * kMaxLevels * 256;
    for (int i = tid; i < g.nlevels * 256; i += 256) {
        const int v = (&hist[0][0])[i];
        if (v) atomicAdd(&hg[i], v);
    }
    if (tid < g.nlevels && lcount[tid]) atomicAdd(&level_counts[b * kMaxLevels + tid], lcount[tid]);
}

__global__ __launch_bounds__(256) void k_emit_finish(Geom g, int chunks, const int* __restrict__ chunk_counts,
                                                     const int* __restrict__ level_counts,
                                                     const int* __restrict__ hist_g, int* __restrict__ thresholds,
                                                     int* __restrict__ totals) {
    const int b = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // a wavefront per level: lane l holds bins 255-4l .. 252-4l; the threshold is the highest score s with
    // (number of corners scoring >= s) >= 2 * quota
    for (int l = wv; l < g.nlevels; l += 4) {
        const int4 v = *(const int4*)(hist_g + ((size_t)b * kMaxLevels + l) * 256 + 252 - 4 * lane);
        const int n = level_counts[b * kMaxLevels + l];
        const int want = 2 * g.L[l].quota;
        const int c0 = v.w, c1 = c0 + v.z, c2 = c1 + v.y, c3 = c2 + v.x;
        const int incl = wave_scan_incl(c3);
        const int excl = incl - c3;
        int t = 0;
        if (n > want) {
            if (want == 0) t = 256;
            else {
                const unsigned long long reach = __ballot(incl >= want);
                const int src = __ffsll((long long)reach) - 1;  // n > want guarantees a lane reaches it
                const int k = excl + c0 >= want ? 0 : excl + c1 >= want ? 1 : excl + c2 >= want ? 2 : 3;
                t = __builtin_amdgcn_readlane(255 - 4 * lane - k, src);
            }
        }
        if (lane == 0) thresholds[b * kMaxLevels + l] = t;
    }
    if (wv == 0) {
        int all = lane < chunks ? chunk_counts[b * kEmitMaxChunks + lane] : 0;
        all = wave_scan_incl(all);
        if (lane == 63) totals[b] = all;
    }
}

// cv::fastAtan2 (scalar atan_f32), degrees in [0, 360).
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    const float eps = (float)2.2204460492503131e-16;
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

struct UMax { int v[16]; };

// Sum over the wavefront, valid in lane 63 only (row reduction, then the two cross-row broadcasts): no scalar
// round trip, for results that one lane consumes.
__device__ __forceinline__ int wave_sum_lane63(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);  // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);  // row_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast31 -> rows 2, 3
    return v;
}

// K5/K6: Harris response (cv::ORB HarrisResponses, 7x7 block, k = 0.04) and intensity-centroid angle
// (cv::ORB ICAngles, radius 15) of one candidate per wavefront.  Candidates whose FAST score is below the level's
// retainBest(2*quota) value are skipped: cv::ORB has dropped them before it computes either quantity.
// Loads: lane = half a patch row, ONE 16-byte load per lane and candidate.  The kernel runs at the rate of the
// texture addresser, which takes the lanes of a gather one by one, so nothing is loaded twice: the 9 x 9 pixels the
// Harris block and its Sobel ring need are the middle of the patch rows -4 .. 4, and the lanes that hold them park
// them in LDS.  Angle: byte sums and |u|-weighted byte sums by dot products (all integer, so the summation order does
// not matter).  Harris: when kHarrisPark candidates are parked, eight of them are reduced at a time - a lane takes one
// block row of one candidate, seven pixels in a row, from three parked patch rows.
// cand_ro is the candidate array again, read-only: position, score and level of a record never change here (only its
// response and angle are written), so the records can be fetched through the scalar path.
constexpr int kHarrisPark = 16;
// candidates of a wavefront whose pixels are in flight or being reduced (3, 4 and 6 measured the same as 2: the kernel
// runs at the rate its gathers are addressed, not at their latency)
constexpr int kHarrisAhead = 2;

__global__ __launch_bounds__(256) void k_harris_angle(const uint8_t* __restrict__ atlas, size_t atlas_bytes, Geom g,
                                                      Cand* __restrict__ cand, const uint4* __restrict__ cand_ro,
                                                      int cap, const int* __restrict__ totals,
                                                      const int* __restrict__ thresholds, UMax umax,
                                                      float scale_sq_sq, int chunks, int frames) {
    // per wavefront: the five sums and the index of the parked candidates, and their middle patch rows: row v + 4 of
    // a block holds the pixels x - 7 .. x + 4 of patch row v (12 bytes of 16)
    __shared__ int s_res[4][kHarrisPark][6];
    __shared__ uint4 s_blk[4][kHarrisPark][9];
    const int lane = threadIdx.x & 63;
    int b, chunk;
    if (!frame_chunk_of_block(chunks, frames, b, chunk)) return;
    const int total = min(totals[b], cap);
    const int aw = g.atlas_w;
    unsigned long long umax4 = 0ull;  // sixteen 4-bit row half-widths
#pragma unroll
    for (int k = 0; k < 16; k++) umax4 |= (unsigned long long)(umax.v[k] & 15) << (4 * k);
    // per-lane constants of the angle part
    const int v = (lane >> 1) - 15;
    const bool right = lane & 1;
    const int d = lane < 62 ? (int)((umax4 >> (4 * (v < 0 ? -v : v))) & 15ull) : -1;
    const int row_ofs = v * aw + (right ? 1 : -15);
    unsigned msk[4], wgt[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (right) {  // byte i of the 16 holds u = i + 1; keep u <= d
            const int keep = min(max(d - 4 * j, 0), 4);
            msk[j] = keep == 0 ? 0u : 0xffffffffu >> (8 * (4 - keep));
            wgt[j] = 0x04030201u + 0x04040404u * j;
        } else {      // byte i holds u = i - 15; keep -u <= d
            const int skip = min(max(15 - d - 4 * j, 0), 4);
            msk[j] = skip == 4 ? 0u : 0xffffffffu << (8 * skip);
            wgt[j] = 0x0c0d0e0fu - 0x04040404u * j;
        }
        if (d < 0) msk[j] = 0u;
    }
    // The frame's candidates are dealt to chunks * 4 wavefronts.  A wavefront's work per candidate is a chain of
    // dependent loads (candidate -> pixels), so it runs as a software pipeline: the pixels of the next
    // candidate and the record of the one after it are in flight while the current candidate is reduced.
    const int stride = chunks * 4;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int idx0 = chunk * 4 + wv;
    const int thr_l = lane < kMaxLevels ? thresholds[b * kMaxLevels + lane] : 0;
    Cand* cbase = cand + (size_t)b * cap;
    const uint4* cro = cand_ro + (size_t)b * cap;
    const uint8_t* fb = atlas + (size_t)b * atlas_bytes;
    // a candidate record is the same for all lanes: its index is wavefront-uniform, so it is read once per wavefront
    auto record = [&](int i) { return cro[min(i, max(total - 1, 0))]; };
    struct Px { uint4 q; bool live; };
    auto fetch_px = [&](const uint4& rec, bool valid) {
        Px p;
        p.q = make_uint4(0u, 0u, 0u, 0u);
        const unsigned xy = __builtin_amdgcn_readfirstlane(rec.x), sl = __builtin_amdgcn_readfirstlane(rec.y);
        const int lvl = (int)((sl >> 8) & (kMaxLevels - 1));
        p.live = valid && (int)(sl & 0xffu) >= __builtin_amdgcn_readlane(thr_l, lvl);
        if (p.live) {
            const LevelInfo L = g.L[lvl];
            // per-lane offset from the candidate's centre pixel (an atlas is far smaller than 4 GB: the frame base
            // stays a scalar pair and the lane offset a single 32-bit register)
            const int cofs = (L.y + (int)(xy >> 16)) * aw + L.x + (int)(xy & 0xffffu);
            p.q = load_u128_any(fb + (unsigned)(cofs + row_ofs));
        }
        return p;
    };
    int (*res)[6] = s_res[wv];
    uint4 (*blk)[9] = s_blk[wv];
    int n_res = 0;
    // the angle sums of a candidate are ADDED into its parked record (see the reduction below): they start at zero
    if (lane < kHarrisPark) {
#pragma unroll
        for (int k = 0; k < 6; k++) res[lane][k] = 0;
    }
    // parking place of this lane's part of a block (lanes 22 .. 39 hold patch rows -4 .. 4)
    const bool parks = v >= -4 && v <= 4 && lane < 62;
    const int park_row = min(max(v + 4, 0), 8);
    auto finish = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // Harris sums of the parked candidates, eight at a time: lane = (candidate, block row); byte 3 + k of a
        // parked row is the pixel x - 4 + k
        const int hc_ = lane >> 3, hr_ = lane & 7;
        for (int p0 = 0; p0 < n_res; p0 += 8) {
            const int c = p0 + hc_;
            const int r = min(hr_, 6);
            const uint4 A = blk[c][r], B = blk[c][r + 1], C = blk[c][r + 2];
            int S[9], D[9];
#pragma unroll
            for (int k = 0; k < 9; k++) {
                const int a_ = k == 0 ? (int)(A.x >> 24) : k < 5 ? (int)((A.y >> (8 * (k - 1))) & 255u) : (int)((A.z >> (8 * (k - 5))) & 255u);
                const int b_ = k == 0 ? (int)(B.x >> 24) : k < 5 ? (int)((B.y >> (8 * (k - 1))) & 255u) : (int)((B.z >> (8 * (k - 5))) & 255u);
                const int c_ = k == 0 ? (int)(C.x >> 24) : k < 5 ? (int)((C.y >> (8 * (k - 1))) & 255u) : (int)((C.z >> (8 * (k - 5))) & 255u);
                S[k] = a_ + c_ + 2 * b_;   // the Sobel column sum: Ix = S[k + 1] - S[k - 1]
                D[k] = c_ - a_;            // Iy = D[k - 1] + 2 D[k] + D[k + 1]
            }
            int ha = 0, hb = 0, hc = 0;
#pragma unroll
            for (int k = 1; k < 8; k++) {
                const int Ix = S[k + 1] - S[k - 1];
                const int Iy = D[k - 1] + D[k + 1] + 2 * D[k];
                ha += __mul24(Ix, Ix);
                hb += __mul24(Iy, Iy);
                hc += __mul24(Ix, Iy);
            }
            if (hr_ == 7) ha = hb = hc = 0;
            // the seven block rows of a candidate sit in eight consecutive lanes: a running sum over 1, 2 and 4 lanes
            // below leaves the candidate's totals in its last lane
#define ORBX_ROW_SHR_ADD(ctrl)                                           \
    ha += __builtin_amdgcn_update_dpp(0, ha, ctrl, 0xF, 0xF, true);      \
    hb += __builtin_amdgcn_update_dpp(0, hb, ctrl, 0xF, 0xF, true);      \
    hc += __builtin_amdgcn_update_dpp(0, hc, ctrl, 0xF, 0xF, true);
            ORBX_ROW_SHR_ADD(0x111)  // row_shr:1 (zeros shifted in)
            ORBX_ROW_SHR_ADD(0x112)  // row_shr:2
            ORBX_ROW_SHR_ADD(0x114)  // row_shr:4
#undef ORBX_ROW_SHR_ADD
            if (hr_ == 7 && c < n_res) {
                res[c][0] = ha;
                res[c][1] = hb;
                res[c][2] = hc;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // The response and the angle of a candidate are scalar work (a division, a polynomial): every lane finishes
        // one parked candidate
        if (lane < n_res) {
            const float fa = (float)res[lane][0], fb2 = (float)res[lane][1], fc = (float)res[lane][2];
            const float resp = (fa * fb2 - fc * fc - 0.04f * (fa + fb2) * (fa + fb2)) * scale_sq_sq;
            Cand* cp = cbase + res[lane][5];
            cp->harris = resp;
            cp->angle = fast_atan2_deg((float)res[lane][4], (float)res[lane][3]);
            res[lane][3] = 0;
            res[lane][4] = 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        n_res = 0;
    };
    const bool odd = lane & 1;
    auto process = [&](const Px& px, int idx) {
        if (!px.live) return;
        const unsigned q[4] = {px.q.x, px.q.y, px.q.z, px.q.w};
        // the middle of patch rows -4 .. 4, for the Harris block: left halves bring x - 7 .. x, right halves x + 1 .. x + 4
        if (parks) {
            unsigned* row = (unsigned*)&blk[n_res][park_row];
            if (right) row[2] = q[0];
            else *(uint2*)row = make_uint2(q[2], q[3]);
        }
        unsigned ssum = 0, wsum = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned m = q[j] & msk[j];
            ssum = __builtin_amdgcn_udot4(m, 0x01010101u, ssum, false);
            wsum = __builtin_amdgcn_udot4(m, wgt[j], wsum, false);
        }
        // Two sums over the wavefront at the price of one: the lanes of a pair exchange what the other one keeps
        // (even lanes: m10, odd lanes: m01), the quad and two row shifts by 4 and 8 leave every row's totals in its
        // lanes 12 (m10) and 13 (m01), which add them into the parked record in LDS.
        const int w10 = right ? (int)wsum : -(int)wsum, w01 = v * (int)ssum;
        int z = (odd ? w01 : w10) + __builtin_amdgcn_update_dpp(0, odd ? w10 : w01, 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
        z += __builtin_amdgcn_update_dpp(0, z, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
        z += __builtin_amdgcn_update_dpp(0, z, 0x114, 0xF, 0xF, true);   // row_shr:4, zeros shifted in
        z += __builtin_amdgcn_update_dpp(0, z, 0x118, 0xF, 0xF, true);   // row_shr:8
        if ((lane & 14) == 12) atomicAdd(&res[n_res][3 + (lane & 1)], z);
        if (lane == 63) res[n_res][5] = idx;
        n_res = __builtin_amdgcn_readfirstlane(n_res + 1);
        if (n_res == kHarrisPark) finish();
    };
    // kHarrisAhead candidates per trip, in as many sets of registers that take turns (no copies from "next" to
    // "current"): slot j holds the candidates j, j + kHarrisAhead, ... of this wavefront.  While a candidate is reduced
    // the pixels of the next kHarrisAhead - 1 are in flight, and each slot already holds the record of its next one.
    uint4 rec[kHarrisAhead];
    Px px[kHarrisAhead];
#pragma unroll
    for (int k = 0; k < kHarrisAhead; k++) rec[k] = record(idx0 + k * stride);
#pragma unroll
    for (int k = 0; k < kHarrisAhead - 1; k++) {
        px[k] = fetch_px(rec[k], idx0 + k * stride < total);
        rec[k] = record(idx0 + (k + kHarrisAhead) * stride);
    }
    for (int idx = idx0; idx < total; idx += kHarrisAhead * stride) {
#pragma unroll
        for (int k = 0; k < kHarrisAhead; k++) {
            constexpr int kLast = kHarrisAhead - 1;
            const int j = (k + kLast) % kHarrisAhead;
            px[j] = fetch_px(rec[j], idx + (k + kLast) * stride < total);
            rec[j] = record(idx + (k + kLast + kHarrisAhead) * stride);
            process(px[k], idx + k * stride);
        }
    }
    if (n_res > 0) finish();
}

// K5 by itself, for the batch path: the Harris response of the candidates above the retainBest(2 * quota) score and
// nothing else - the orientation is only needed for the keypoints the selection keeps (k_angle_kept).  A response
// needs nine rows of nine pixels, so ONE gather serves seven candidates: lane = (candidate, row), 16 bytes per lane
// (the pixels x - 7 .. x + 8 of that row; bytes 3 .. 11 are used).  A wavefront takes 64 consecutive candidates,
// lists the live ones (position in the atlas, index), and works the list off 28 at a time: four gathers in flight,
// the rows parked in LDS, then lane = (candidate, block row) as in k_harris_angle, eight candidates per step.
constexpr int kHarrisGroup = 7;                 // candidates per gather: 7 x 9 rows = 63 lanes
constexpr int kHarrisRound = 4 * kHarrisGroup;  // candidates parked per round

__global__ __launch_bounds__(256) void k_harris(const uint8_t* __restrict__ atlas, size_t atlas_bytes, Geom g,
                                                Cand* __restrict__ cand, int cap, const int* __restrict__ totals,
                                                const int* __restrict__ thresholds, float scale_sq_sq, int chunks,
                                                int frames) {
    __shared__ int2 s_list[4][64];
    __shared__ uint4 s_blk[4][kHarrisRound + 4][9];   // the reduction reads whole steps of eight candidates
    __shared__ int s_org[kMaxLevels];
    const int lane = threadIdx.x & 63;
    int b, chunk;
    if (!frame_chunk_of_block(chunks, frames, b, chunk)) return;
    const int total = min(totals[b], cap);
    const int aw = g.atlas_w;
    // atlas offset of every level's pixel (0, 0): the level differs from lane to lane here
#pragma unroll
    for (int l = 0; l < kMaxLevels; l++)
        if (threadIdx.x == l) s_org[l] = g.L[l].y * aw + g.L[l].x;
    __syncthreads();
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    Cand* cbase = cand + (size_t)b * cap;
    const uint8_t* fb = atlas + (size_t)b * atlas_bytes;
    int2* list = s_list[wv];
    uint4 (*blk)[9] = s_blk[wv];
    const int c7 = div_small<9>(lane), row = lane - 9 * c7;   // lane 63: no candidate
    const int row_ofs = (row - 4) * aw - 7;
    const int hc_ = lane >> 3, hr_ = lane & 7;
    auto fence = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    const int stride = chunks * 4 * 64;
    for (int first = (chunk * 4 + wv) * 64; first < total; first += stride) {
        const int i = first + lane;
        bool live = false;
        int cofs = 0;
        if (i < total) {
            const uint2 r = *(const uint2*)(cbase + i);   // x | y << 16, score | level << 8
            const int lvl = (int)((r.y >> 8) & (kMaxLevels - 1));
            live = (int)(r.y & 0xffu) >= thresholds[b * kMaxLevels + lvl];
            cofs = s_org[lvl] + (int)(r.x >> 16) * aw + (int)(r.x & 0xffffu);
        }
        const unsigned long long m = __ballot(live);
        const int nlive = __popcll(m);
        if (live)
            list[__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] =
                make_int2(cofs, i);
        fence();
        for (int g0 = 0; g0 < nlive; g0 += kHarrisRound) {
            const int n_round = min(kHarrisRound, nlive - g0);
            uint4 q[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int ci = kHarrisGroup * u + c7;
                ok[u] = c7 < kHarrisGroup && ci < n_round;
                q[u] = make_uint4(0u, 0u, 0u, 0u);
                if (ok[u]) q[u] = load_u128_any(fb + (unsigned)(list[g0 + ci].x + row_ofs));
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (ok[u]) blk[kHarrisGroup * u + c7][row] = q[u];
            fence();
            // byte 3 + k of a parked row is the pixel x - 4 + k
            for (int p0 = 0; p0 < n_round; p0 += 8) {
                const int c = p0 + hc_;
                const int r = min(hr_, 6);
                const uint4 A = blk[c][r], B = blk[c][r + 1], C = blk[c][r + 2];
                int S[9], D[9];
#pragma unroll
                for (int k = 0; k < 9; k++) {
                    const int a_ = k == 0 ? (int)(A.x >> 24) : k < 5 ? (int)((A.y >> (8 * (k - 1))) & 255u) : (int)((A.z >> (8 * (k - 5))) & 255u);
                    const int b_ = k == 0 ? (int)(B.x >> 24) : k < 5 ? (int)((B.y >> (8 * (k - 1))) & 255u) : (int)((B.z >> (8 * (k - 5))) & 255u);
                    const int c_ = k == 0 ? (int)(C.x >> 24) : k < 5 ? (int)((C.y >> (8 * (k - 1))) & 255u) : (int)((C.z >> (8 * (k - 5))) & 255u);
                    S[k] = a_ + c_ + 2 * b_;   // the Sobel column sum: Ix = S[k + 1] - S[k - 1]
                    D[k] = c_ - a_;            // Iy = D[k - 1] + 2 D[k] + D[k + 1]
                }
                int ha = 0, hb = 0, hc = 0;
#pragma unroll
                for (int k = 1; k < 8; k++) {
                    const int Ix = S[k + 1] - S[k - 1];
                    const int Iy = D[k - 1] + D[k + 1] + 2 * D[k];
                    ha += __mul24(Ix, Ix);
                    hb += __mul24(Iy, Iy);
                    hc += __mul24(Ix, Iy);
                }
                if (hr_ == 7) ha = hb = hc = 0;
                // a running sum over 1, 2 and 4 lanes leaves a candidate's totals in the last of its eight lanes
#define ORBX_ROW_SHR_ADD(ctrl)                                           \
    ha += __builtin_amdgcn_update_dpp(0, ha, ctrl, 0xF, 0xF, true);      \
    hb += __builtin_amdgcn_update_dpp(0, hb, ctrl, 0xF, 0xF, true);      \
    hc += __builtin_amdgcn_update_dpp(0, hc, ctrl, 0xF, 0xF, true);
                ORBX_ROW_SHR_ADD(0x111)  // row_shr:1 (zeros shifted in)
                ORBX_ROW_SHR_ADD(0x112)  // row_shr:2
                ORBX_ROW_SHR_ADD(0x114)  // row_shr:4
#undef ORBX_ROW_SHR_ADD
                if (hr_ == 7 && c < n_round) {
                    const float fa = (float)ha, fb2 = (float)hb, fc = (float)hc;
                    cbase[list[g0 + c].y].harris = (fa * fb2 - fc * fc - 0.04f * (fa + fb2) * (fa + fb2)) * scale_sq_sq;
                }
            }
            fence();   // the next round parks over these rows, the next block lists over these entries
        }
    }
}

// K6 for the keypoints the selection kept (batch path): the intensity-centroid angle of k_harris_angle, one keypoint
// per wavefront and step.  src[frame][slot] holds where the keypoint was detected (x | y << 14 | level << 28, written
// by k_select_frame: the keypoint's own coordinates may have been undistorted since); frames whose selection has not
// finished (status != 0) have no keypoints yet.
__global__ __launch_bounds__(256) void k_angle_kept(const uint8_t* __restrict__ atlas, size_t atlas_bytes, Geom g,
                                                    orbx_keypoint* __restrict__ kps, const int* __restrict__ src,
                                                    int kp_cap, const int* __restrict__ kp_counts,
                                                    const int* __restrict__ status, UMax umax, int chunks,
                                                    int frames) {
    __shared__ int s_res[4][kHarrisPark][3];
    const int lane = threadIdx.x & 63;
    int b, chunk;
    if (!frame_chunk_of_block(chunks, frames, b, chunk)) return;
    const int total = status[b] == 0 ? min(kp_counts[b], kp_cap) : 0;
    const int aw = g.atlas_w;
    unsigned long long umax4 = 0ull;  // sixteen 4-bit row half-widths
#pragma unroll
    for (int k = 0; k < 16; k++) umax4 |= (unsigned long long)(umax.v[k] & 15) << (4 * k);
    const int v = (lane >> 1) - 15;
    const bool right = lane & 1;
    const int d = lane < 62 ? (int)((umax4 >> (4 * (v < 0 ? -v : v))) & 15ull) : -1;
    const int row_ofs = v * aw + (right ? 1 : -15);
    unsigned msk[4], wgt[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (right) {  // byte i of the 16 holds u = i + 1; keep u <= d
            const int keep = min(max(d - 4 * j, 0), 4);
            msk[j] = keep == 0 ? 0u : 0xffffffffu >> (8 * (4 - keep));
            wgt[j] = 0x04030201u + 0x04040404u * j;
        } else {      // byte i holds u = i - 15; keep -u <= d
            const int skip = min(max(15 - d - 4 * j, 0), 4);
            msk[j] = skip == 4 ? 0u : 0xffffffffu << (8 * skip);
            wgt[j] = 0x0c0d0e0fu - 0x04040404u * j;
        }
        if (d < 0) msk[j] = 0u;
    }
    const int stride = chunks * 4;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int idx0 = chunk * 4 + wv;
    orbx_keypoint* kbase = kps + (size_t)b * kp_cap;
    const int* sbase = src + (size_t)b * kp_cap;
    const uint8_t* fb = atlas + (size_t)b * atlas_bytes;
    // the slot is wavefront-uniform: its source word is read once per wavefront
    auto source = [&](int i) { return sbase[min(i, max(total - 1, 0))]; };
    auto fetch_px = [&](int word, bool valid) {
        uint4 q = make_uint4(0u, 0u, 0u, 0u);
        const unsigned w = __builtin_amdgcn_readfirstlane((unsigned)word);
        if (valid) {
            const LevelInfo L = g.L[w >> 28];
            const int cofs = (L.y + (int)((w >> 14) & 0x3fffu)) * aw + L.x + (int)(w & 0x3fffu);
            q = load_u128_any(fb + (unsigned)(cofs + row_ofs));
        }
        return q;
    };
    int (*res)[3] = s_res[wv];
    int n_res = 0;
    if (lane < kHarrisPark) res[lane][0] = res[lane][1] = 0;   // the sums are ADDED into the parked record
    auto fence = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto finish = [&]() {
        fence();
        if (lane < n_res) {   // the polynomial and the division are one lane's work per keypoint
            kbase[res[lane][2]].angle = fast_atan2_deg((float)res[lane][1], (float)res[lane][0]);
            res[lane][0] = 0;
            res[lane][1] = 0;
        }
        fence();
        n_res = 0;
    };
    const bool odd = lane & 1;
    auto process = [&](const uint4& px, int idx) {
        if (idx >= total) return;
        const unsigned q[4] = {px.x, px.y, px.z, px.w};
        unsigned ssum = 0, wsum = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned m = q[j] & msk[j];
            ssum = __builtin_amdgcn_udot4(m, 0x01010101u, ssum, false);
            wsum = __builtin_amdgcn_udot4(m, wgt[j], wsum, false);
        }
        // m10 and m01 in one reduction: the lanes of a pair exchange what the other one keeps (even: m10, odd: m01)
        const int w10 = right ? (int)wsum : -(int)wsum, w01 = v * (int)ssum;
        int z = (odd ? w01 : w10) + __builtin_amdgcn_update_dpp(0, odd ? w10 : w01, 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
        z += __builtin_amdgcn_update_dpp(0, z, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
        z += __builtin_amdgcn_update_dpp(0, z, 0x114, 0xF, 0xF, true);   // row_shr:4, zeros shifted in
        z += __builtin_amdgcn_update_dpp(0, z, 0x118, 0xF, 0xF, true);   // row_shr:8
        if ((lane & 14) == 12) atomicAdd(&res[n_res][lane & 1], z);
        if (lane == 63) res[n_res][2] = idx;
        n_res = __builtin_amdgcn_readfirstlane(n_res + 1);
        if (n_res == kHarrisPark) finish();
    };
    // pixels of the next keypoint and the source word of the one after it in flight while this one is reduced
    int src_a = source(idx0), src_b = source(idx0 + stride);
    uint4 px_a = fetch_px(src_a, idx0 < total);
    for (int idx = idx0; idx < total; idx += 2 * stride) {
        src_a = source(idx + 2 * stride);
        const uint4 px_b = fetch_px(src_b, idx + stride < total);
        process(px_a, idx);
        src_b = source(idx + 3 * stride);
        px_a = fetch_px(src_a, idx + 2 * stride < total);
        process(px_b, idx + stride);
    }
    if (n_res > 0) finish();
}

// Frame model of MonoFrame::extractFeatures for the device-resident batch path.
struct FrameModel {
    int has_camera, has_grid;
    orbx_camera cam;
    orbx_grid grid;
    int width, height, edge, patch_size;
};

constexpr int kSelectMaxCands = 8192;  // candidates of one frame the selection kernel holds in LDS

__device__ __forceinline__ bool border_ok(const orbx_keypoint& k, int w, int h, int border) {
    // cv::KeyPointsFilter::runByImageBorder: Rect(border, border, w - 2 border, h - 2 border).contains(round(pt))
    if (border <= 0) return true;
    if (h <= border * 2 || w <= border * 2) return false;
    const int px = __float2int_rn(k.x), py = __float2int_rn(k.y);
    return border <= px && px < w - border && border <= py && py < h - border;
}

// Ordered in-place compaction of s_kp[0..n) by one wavefront; keep() decides per keypoint.  Returns the new count
// (valid on every lane).
template <typename Keep>
__device__ __forceinline__ int compact_ordered(orbx_keypoint* s_kp, int n, int lane, Keep keep) {
    int running = 0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        orbx_keypoint k;
        bool f = false;
        if (i < n) {
            k = s_kp[i];
            f = keep(k);
        }
        const unsigned long long bal = __ballot(f);
        if (f) s_kp[running + __popcll(bal & ((1ull << lane) - 1ull))] = k;
        running += __popcll(bal);
    }
    return running;
}

// Device half of cv::ORB::detect after the kernels above, plus the host steps MonoFrame::extractFeatures performs
// between detect and compute - one workgroup per frame, one wavefront per pyramid level:
//   retainBest(2*quota) on the FAST score, retainBest(quota) on the Harris response (the level's wave replays
//   libstdc++'s nth_element + partition, stdsel.h), keypoint assembly (size, angle, level-0 coordinates; with kp_src
//   the angle is left to k_angle_kept and kp_src[frame][slot] receives the place of detection),
//   Camera::undistortPoints, Frame::assignFeaturesToGrid (ordered filter + CSR cells), and the border filter of
//   cv::ORB::compute.  status[frame]: 0 ok, 1 more keypoints than kp_cap, 2 more candidates than the LDS holds
//   (3 between the two launches: left to the full-size one).
// GLOBAL = false: the working arrays live in the workgroup's LDS (max_cands candidates, kp_cap keypoint slots must
// fit it).  GLOBAL = true: the same code on a per-frame slice of a global workspace, for frames beyond that (4K
// frames, thousands of keypoints): slower per frame, but nothing crosses PCIe and nothing has a fixed limit; pass 2
// of that form takes only the frames an LDS launch marked with status 2.
template <bool GLOBAL>
__global__ __launch_bounds__(1024) void k_select_frame(const Cand* __restrict__ cand, int cand_cap,
                                                        const int* __restrict__ level_counts,
                                                        const int* __restrict__ totals, Geom g, FrameModel fm,
                                                        orbx_keypoint* __restrict__ kps_out, int kp_cap,
                                                        int* __restrict__ kp_counts, int* __restrict__ cell_start,
                                                        int* __restrict__ cell_items, int* __restrict__ status,
                                                        int max_cands, int pass, unsigned char* __restrict__ ws,
                                                        size_t ws_stride, int ws_slices,
                                                        int* __restrict__ ws_counter, int* __restrict__ kp_src,
                                                        const int* __restrict__ picked_idx = nullptr,
                                                        const int* __restrict__ picked_n = nullptr,
                                                        int picked_stride = 0) {
    extern __shared__ __align__(16) unsigned char s_lds[];
    __shared__ int s_n[kMaxLevels], s_off[kMaxLevels], s_cnt, s_slice;
    typedef typename std::conditional<GLOBAL, uint32_t, uint16_t>::type St;
    const int b = blockIdx.x;
    if (pass == 1 && status[b] != 3) return;
    if (pass == 2 && status[b] != 2) return;
    int slice = b;
    if (GLOBAL && pass == 2) {
        // frames arrive here rarely: they share a few workspace slices, handed out in arrival order
        if (totals[b] > cand_cap) return;  // the candidate list itself overflowed: the host path enlarges it
        if (threadIdx.x == 0) s_slice = atomicAdd(ws_counter, 1);
        __syncthreads();
        slice = s_slice;
        if (slice >= ws_slices) return;    // status stays 2
    }
    unsigned char* s_raw = GLOBAL ? ws + (size_t)slice * ws_stride : s_lds;
    stdsel::Elem* s_el = (stdsel::Elem*)s_raw;                                  // [max_cands]
    orbx_keypoint* s_kp = (orbx_keypoint*)(s_raw + sizeof(stdsel::Elem) * max_cands);  // [kp_cap]
    int* s_cell = (int*)(s_kp + kp_cap);                                         // [kp_cap]
    int* s_cells = (int*)s_raw;                                                  // [ncell + 1] + [kp_cap] item lists, alias s_el later
    // stopper lists of the wave-parallel partitions; they alias s_kp / s_cell, which are idle during the selection
    St* s_lp = (St*)s_kp;                                                        // [max_cands]
    St* s_rp = s_lp + max_cands;                                                 // [max_cands]
    const int tid = threadIdx.x, lane = tid & 63, lvl = tid >> 6;
    const Cand* cd = cand + (size_t)b * cand_cap;
    // picked_idx: the opt-in distribution mode (k_distribute_level) has already chosen every level's keypoints; both
    // culls are skipped and only the chosen candidates enter the working arrays
    const bool picked = picked_idx != nullptr;
    int total = totals[b];
    const bool overflowed = picked && total > cand_cap;   // the candidate list itself: k_distribute_level chose nothing
    if (picked) {
        total = 0;
        for (int j = 0; j < g.nlevels; j++) total += picked_n[b * kMaxLevels + j];
    }
    // pass 0: the launch with the small LDS leaves frames that do not fit it to pass 1 (status 3); pass 1 takes only
    // those; pass -1 is a single launch for every frame; pass 2 (global form) takes the frames an LDS launch left
    // A frame this launch cannot finish leaves no keypoints and an EMPTY grid: whatever is queued behind the batch
    // on the stream (a matcher walking the cells) then finds nothing instead of what an earlier batch left there.
    auto leave_empty = [&](int st) {
        if (fm.has_grid) {
            const int ncell = fm.grid.grid_cols * fm.grid.grid_rows;
            int* cs = cell_start + (size_t)b * (ncell + 1);
            for (int i = tid; i <= ncell; i += blockDim.x) cs[i] = 0;
        }
        if (tid == 0) { status[b] = st; kp_counts[b] = 0; }
    };
    if (overflowed) {
        leave_empty(2);
        return;
    }
    if (total > max_cands || total > cand_cap) {
        leave_empty((pass == 0 && total <= cand_cap) ? 3 : 2);
        return;
    }
    int n = 0, off = 0;
    const int* per_level = picked ? picked_n : level_counts;
    if (lvl < g.nlevels) {
        n = per_level[b * kMaxLevels + lvl];
        for (int j = 0; j < lvl; j++) off += per_level[b * kMaxLevels + j];
    }
    if (picked && lvl < g.nlevels) {
        const int* mine = picked_idx + ((size_t)b * kMaxLevels + lvl) * picked_stride;
        for (int i = lane; i < n; i += 64) s_el[off + i] = stdsel::Elem{0.f, min(max(mine[i], 0), cand_cap - 1)};
    }
    // the levels' candidates are one contiguous list: all wavefronts load it together, four loads in flight per lane
    for (int base = 0; !picked && base < total; base += 4 * (int)blockDim.x) {
        int sc[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = base + u * (int)blockDim.x + tid;
            sc[u] = i < total ? (int)cd[i].score : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = base + u * (int)blockDim.x + tid;
            if (i < total) s_el[i] = stdsel::Elem{(float)sc[u], i};
        }
    }
    __syncthreads();
    if (!picked && lvl < g.nlevels)
        n = stdsel::wave_retain_best(s_el + off, n, 2 * g.L[lvl].quota, s_lp + off, s_rp + off, lane);
    __syncthreads();
    if (!picked && lvl < g.nlevels)
        for (int base = 0; base < n; base += 256) {   // four gathers in flight per lane
            int ix[4];
            float hv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = base + 64 * u + lane;
                ix[u] = i < n ? s_el[off + i].idx : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) hv[u] = ix[u] >= 0 ? cd[ix[u]].harris : 0.f;
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (ix[u] >= 0) s_el[off + base + 64 * u + lane].key = hv[u];
        }
    __syncthreads();
    if (lvl < g.nlevels) {
        if (!picked) n = stdsel::wave_retain_best(s_el + off, n, g.L[lvl].quota, s_lp + off, s_rp + off, lane);
        if (lane == 0) s_n[lvl] = n;
    }
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int l = 0; l < g.nlevels; l++) { s_off[l] = acc; acc += s_n[l]; }
        s_cnt = acc;
    }
    __syncthreads();
    int count = s_cnt;
    if (count > kp_cap) {
        leave_empty(1);
        return;
    }
    if (lvl < g.nlevels) {
        const float sf = g.L[lvl].scale;
        const int oo = s_off[lvl];
        for (int i = lane; i < n; i += 64) {
            const Cand c = cd[s_el[off + i].idx];
            orbx_keypoint k;
            k.x = (float)c.x * sf;
            k.y = (float)c.y * sf;
            k.size = fm.patch_size * sf;
            k.angle = c.angle;
            k.response = picked ? (float)c.score : c.harris;   // the distribution mode ranks by corner score
            k.octave = lvl;
            // with kp_src the orientation is computed after this kernel, for the kept keypoints only: where the
            // keypoint was detected travels in class_id until the final store
            k.class_id = kp_src ? (int)((unsigned)c.x | (unsigned)c.y << 14 | (unsigned)lvl << 28) : -1;
            s_kp[oo + i] = k;
        }
    }
    __syncthreads();
    if (fm.has_camera && fm.cam.k1 != 0.0) {
        // cv::undistortPoints with P = K: 5 fixed-point iterations in double (camera.cpp:107-113)
        const orbx_camera c = fm.cam;
        const double ifx = 1. / c.fx, ify = 1. / c.fy;
        for (int i = tid; i < count; i += blockDim.x) {
            double x = s_kp[i].x, y = s_kp[i].y;
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
            s_kp[i].x = (float)(x * c.fx + c.cx);
            s_kp[i].y = (float)(y * c.fy + c.cy);
        }
        __syncthreads();
    }
    if (fm.has_grid) {
        // Frame::assignFeaturesToGrid / pointInGrid (frame.cpp:326-367)
        const orbx_grid gr = fm.grid;
        const int ncell = gr.grid_cols * gr.grid_rows;
        auto cell_of = [&](const orbx_keypoint& k) -> int {
            const int pos_y = (int)((k.y - gr.min_y) / gr.grid_size_y);
            const int pos_x = (int)((k.x - gr.min_x) / gr.grid_size_x);
            if (pos_y < 0 || pos_y >= gr.grid_rows || pos_x < 0 || pos_x >= gr.grid_cols) return -1;
            return pos_x * gr.grid_rows + pos_y;
        };
        // keypoints outside the grid are dropped (ordered compaction by one wavefront) - rarely any: look first
        int* s_items = s_cells + ncell + 1;   // [kp_cap] the cells' item lists in the order the atomics fill them
        bool outside = false;
        for (int i = tid; i < count; i += blockDim.x) outside |= cell_of(s_kp[i]) < 0;
        if (__syncthreads_or(outside)) {
            if (lvl == 0) {
                count = compact_ordered(s_kp, count, lane, [&](const orbx_keypoint& k) { return cell_of(k) >= 0; });
                if (lane == 0) s_cnt = count;
            }
            __syncthreads();
            count = s_cnt;
        }
        for (int i = tid; i <= ncell; i += blockDim.x) s_cells[i] = 0;
        __syncthreads();
        for (int i = tid; i < count; i += blockDim.x) {
            const int c = cell_of(s_kp[i]);
            s_cell[i] = c;
            atomicAdd(&s_cells[c + 1], 1);
        }
        __syncthreads();
        {
            // inclusive scan of the per-cell counts -> CSR offsets: every wavefront scans one segment of the cells,
            // then adds the totals of the segments before its own
            const int nw = blockDim.x >> 6;
            const int seg = ((ncell + nw - 1) / nw + 63) & ~63;
            const int s0 = 1 + lvl * seg, s1 = min(s0 + seg, ncell + 1);
            int carry = 0;
            for (int base = s0; base < s1; base += 64) {
                const int i = base + lane;
                const int v = wave_scan_incl(i < s1 ? s_cells[i] : 0);
                if (i < s1) s_cells[i] = v + carry;
                carry += __builtin_amdgcn_readlane(v, 63);
            }
            if (lane == 0) s_n[lvl] = carry;   // s_n is free again: the level counts went into s_off / s_cnt
            __syncthreads();
            int before = 0;
            for (int w = 0; w < lvl; w++) before += s_n[w];
            if (before)
                for (int i = s0 + lane; i < s1; i += 64) s_cells[i] += before;
        }
        __syncthreads();
        int* cs = cell_start + (size_t)b * (ncell + 1);
        int* ci = cell_items + (size_t)b * kp_cap;
        for (int i = tid; i <= ncell; i += blockDim.x) cs[i] = s_cells[i];
        __syncthreads();
        // fill: slots by atomics (s_cells[c] walks from the start of cell c to the start of cell c + 1); then every
        // keypoint goes to the place its index has among its cell's items (the reference appends features in index
        // order): a rank within a list of a few entries, no sorting pass over the cells
        for (int i = tid; i < count; i += blockDim.x) s_items[atomicAdd(&s_cells[s_cell[i]], 1)] = i;
        __syncthreads();
        for (int i = tid; i < count; i += blockDim.x) {
            const int c = s_cell[i];
            const int i0 = c ? s_cells[c - 1] : 0, i1 = s_cells[c];
            int rank = 0;
            for (int j = i0; j < i1; j++) rank += s_items[j] < i;
            ci[i0 + rank] = i;
        }
    }
    // cv::ORB::compute -> runByImageBorder on the moved keypoints; the grid above is left as built, exactly as the
    // reference leaves it: its entries now name the slots the remaining keypoints moved into, and the last
    // (pre - count) entries name slots past the end.  The reference reads past the end of its vector there; here
    // those slots receive a feature no window accepts (octave -128), so the result does not depend on what an earlier
    // batch left in the buffer.
    const int pre = count;
    bool drop = false;
    if (fm.has_camera)
        for (int i = tid; i < count; i += blockDim.x) drop |= !border_ok(s_kp[i], fm.width, fm.height, fm.edge);
    if (__syncthreads_or(drop)) {   // rarely any: the ordered compaction is one wavefront's work
        if (lvl == 0) {
            count = compact_ordered(s_kp, count, lane,
                                    [&](const orbx_keypoint& k) { return border_ok(k, fm.width, fm.height, fm.edge); });
            if (lane == 0) s_cnt = count;
        }
        __syncthreads();
        count = s_cnt;
    }
    orbx_keypoint* out = kps_out + (size_t)b * kp_cap;
    for (int i = tid; i < count; i += blockDim.x) {
        orbx_keypoint k = s_kp[i];
        if (kp_src) {
            kp_src[(size_t)b * kp_cap + i] = k.class_id;
            k.class_id = -1;
        }
        out[i] = k;
    }
    for (int i = count + tid; i < pre; i += blockDim.x) out[i] = orbx_keypoint{0.f, 0.f, 0.f, -1.f, 0.f, -128, -1};
    if (tid == 0) { kp_counts[b] = count; status[b] = 0; }
}

// Test hook: retainBest of an arbitrary key array by the wave-parallel replay (one wavefront).
__global__ __launch_bounds__(64) void k_selftest_retain_best(const float* __restrict__ keys, int n, int n_points,
                                                             int* __restrict__ perm_out, int* __restrict__ n_out) {
    extern __shared__ __align__(16) unsigned char s_raw[];
    stdsel::Elem* s_el = (stdsel::Elem*)s_raw;
    uint16_t* s_lp = (uint16_t*)(s_el + n);
    uint16_t* s_rp = s_lp + n;
    const int lane = threadIdx.x;
    for (int i = lane; i < n; i += 64) s_el[i] = stdsel::Elem{keys[i], i};
    __syncthreads();
    const int m = stdsel::wave_retain_best(s_el, n, n_points, s_lp, s_rp, lane);
    __syncthreads();
    for (int i = lane; i < m; i += 64) perm_out[i] = s_el[i].idx;
    if (lane == 0) *n_out = m;
}

// Packs the per-frame candidate lists back to back (offsets = exclusive prefix of totals) for one D2H copy.
__global__ void k_scan_totals(const int* __restrict__ totals, int n, int cap, int* __restrict__ offsets) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int acc = 0;
        for (int i = 0; i < n; i++) {
            offsets[i] = acc;
            acc += min(totals[i], cap);
        }
        offsets[n] = acc;
    }
}

__global__ void k_pack_cands(const Cand* __restrict__ cand, int cap, const int* __restrict__ totals,
                             const int* __restrict__ offsets, Cand* __restrict__ packed) {
    const int b = blockIdx.y;
    const int n = min(totals[b], cap);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) packed[offsets[b] + i] = cand[(size_t)b * cap + i];
}

// K7: cv::GaussianBlur(7x7, sigma 2) of every level as cv::ORB applies it (8-bit fixed-point separable filter,
// taps round(g*256), rows then columns, (sum + 2^15) >> 16).  Dword-wise: a lane owns 4 adjacent atlas pixels of a
// 32-row column strip.  Per input row it loads the three aligned dwords around them, forms the four horizontal sums
// with v_alignbyte + v_dot4_u32_u8 (they fit 16 bits: the taps sum to 257), keeps the last seven rows of sums in
// registers and emits one packed output dword per row.  No LDS; the halo rows/columns come from the atlas border.
// Reads the un-blurred atlas, writes level interiors (plus at most 3 bytes of never-read border per row end).
struct BlurTaps { unsigned wa, wb; int k[4]; };

// a * b + c on the low 24 bits of a (wavefront-uniform) and b, in ONE instruction
__device__ __forceinline__ unsigned mad24(unsigned a, unsigned b, unsigned c) {
    unsigned d;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "s"(a), "v"(b), "v"(c));
    return d;
}

// two signed 16-bit halves -> two bytes saturated to 0 .. 255 in the low half of the result
__device__ __forceinline__ unsigned sat_pk_u8(unsigned two_i16) {
    unsigned d;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(d) : "v"(two_i16));
    return d;
}

__global__ __launch_bounds__(256) void k_blur(const uint8_t* __restrict__ atlas, uint8_t* __restrict__ blurred,
                                              size_t atlas_bytes, Geom g, const Strip* __restrict__ strips,
                                              int n_strips, BlurTaps bt) {
    const int sidx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (sidx >= n_strips) return;
    const Strip st = strips[sidx];
    const LevelInfo L = g.L[st.level];
    const int lane = threadIdx.x & 63;
    const int ax = st.ax0 + (lane & 31) * 4;                 // atlas x of this lane's 4 pixels
    const int y0 = st.y0 + (lane >> 5) * kStripRows;         // first level row of this half
    if (ax + 3 < L.x || ax >= L.x + L.w || y0 >= L.h) return;
    const int rows = min(kStripRows, L.h - y0);
    const int aw = g.atlas_w;
    const size_t fo = (size_t)blockIdx.y * atlas_bytes;
    // uniform frame base + 32-bit lane offset (an atlas is far smaller than 4 GB): no 64-bit multiplies per row
    const uint8_t* src = atlas + fo;
    uint8_t* dst = blurred + fo;
    const unsigned so = (unsigned)(L.y + y0 - 3) * (unsigned)aw + (unsigned)ax - 4u;
    const unsigned d_o = (unsigned)(L.y + y0) * (unsigned)aw + (unsigned)ax;
    // Straight-line code: every row load is unconditional (row index clamped into the strip's own rows, whose
    // results are then simply not stored), so the loads of later rows are in flight while earlier rows are filtered.
    struct Row3 { unsigned lft, c, rgt; };
    auto load_row = [&](int r) {
        Row3 v;
        __builtin_memcpy(&v, src + (so + __umul24((unsigned)min(r, rows + 5), (unsigned)aw)), 12);
        return v;
    };
    // a multiply-add takes at most one scalar operand: the tap; the rounding constant lives in a vector register
    // (written by an opaque instruction, or the compiler would re-create it in front of every use)
    const unsigned kv[4] = {(unsigned)bt.k[0], (unsigned)bt.k[1], (unsigned)bt.k[2], (unsigned)bt.k[3]};
    unsigned round_half;
    asm volatile("v_mov_b32 %0, 0x8000" : "=v"(round_half));
    constexpr int kBlurAhead = 8;
    Row3 pf[kBlurAhead];
#pragma unroll
    for (int r = 0; r < kBlurAhead; r++) pf[r] = load_row(r);
    unsigned h[7][4];
#pragma unroll
    for (int r = 0; r < kStripRows + 6; r++) {
        const Row3 cur = pf[r % kBlurAhead];
        if (r + kBlurAhead < kStripRows + 6) pf[r % kBlurAhead] = load_row(r + kBlurAhead);
        const unsigned lft = cur.lft, c = cur.c, rgt = cur.rgt;
        unsigned* hr = h[r % 7];
        hr[0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(c, lft, 1), bt.wa,
                                       __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(rgt, c, 1), bt.wb, 0u, false), false);
        hr[1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(c, lft, 2), bt.wa,
                                       __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(rgt, c, 2), bt.wb, 0u, false), false);
        hr[2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(c, lft, 3), bt.wa,
                                       __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(rgt, c, 3), bt.wb, 0u, false), false);
        hr[3] = __builtin_amdgcn_udot4(c, bt.wa, __builtin_amdgcn_udot4(rgt, bt.wb, 0u, false), false);
        if (r >= 6) {
            // rows r-6 .. r hold taps 0 .. 6 (symmetric): k3*h3 + k2*(h2+h4) + k1*(h1+h5) + k0*(h0+h6) + 2^15, as three
            // full-rate adds and a chain of four 24-bit multiply-adds (taps <= 256, sums < 2^17; left to itself the
            // compiler multiplies first and sums with three-operand adds, two more half-rate instructions per pixel)
            unsigned acc[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                acc[j] = mad24(kv[0], h[(r - 6) % 7][j] + h[r % 7][j], round_half);
                acc[j] = mad24(kv[1], h[(r - 5) % 7][j] + h[(r - 1) % 7][j], acc[j]);
                acc[j] = mad24(kv[2], h[(r - 4) % 7][j] + h[(r - 2) % 7][j], acc[j]);
                acc[j] = mad24(kv[3], h[(r - 3) % 7][j], acc[j]);
            }
            // >> 16 and the clamp to 255 (the taps sum to 257) for two pixels at a time: the upper halves of two sums
            // side by side are two 16-bit integers, which one instruction narrows to two saturated bytes
            const unsigned lo = sat_pk_u8(__builtin_amdgcn_perm(acc[1], acc[0], 0x07060302u));
            const unsigned hi = sat_pk_u8(__builtin_amdgcn_perm(acc[3], acc[2], 0x07060302u));
            const unsigned out = lo | (hi << 16);
            if (r - 6 < rows) *(unsigned*)(dst + (d_o + __umul24((unsigned)(r - 6), (unsigned)aw))) = out;
        }
    }
}

struct BriefPattern { int8_t p[1024]; };

// Steering of one keypoint for rBRIEF: cos / sin of the angle (evaluated in double, as cv::ORB does), the rounded
// centre in its level, the level's place and size in the atlas.
struct Steer {
    float a, b;
    int cx, cy;
    int org;   // atlas offset of the level's pixel (0, 0)
    int w, h;
    int pad;
};

__global__ __launch_bounds__(256) void k_steer(Geom g, const orbx_keypoint* __restrict__ kps,
                                               const int* __restrict__ counts, int kp_cap, Steer* __restrict__ steer) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= counts[b] || i >= kp_cap) return;
    const orbx_keypoint kp = kps[(size_t)b * kp_cap + i];
    // (the slots of a frame whose selection failed hold whatever was there: an octave or a centre out of range must
    // not become an address - keypoints of a finished selection are never changed by the clamps)
    const LevelInfo L = g.L[min((unsigned)kp.octave, (unsigned)(kMaxLevels - 1))];
    float angle = kp.angle;
    angle *= (float)(3.14159265358979323846 / 180.f);
    Steer s;
    s.a = (float)cos((double)angle);
    s.b = (float)sin((double)angle);
    s.cx = min(max(__float2int_rn(kp.x * L.inv_scale), -13), L.w + 12);   // samples reach 18 further: inside the 32-pixel border
    s.cy = min(max(__float2int_rn(kp.y * L.inv_scale), -13), L.h + 12);
    s.org = L.y * g.atlas_w + L.x;
    s.w = L.w;
    s.h = L.h;
    s.pad = 0;
    steer[(size_t)b * kp_cap + i] = s;
}

// K8: rotated BRIEF (cv::ORB computeOrbDescriptors, WTA_K = 2).  32 lanes per keypoint, one descriptor byte per
// lane.  Samples inside the level come from the blurred plane; samples in the level's border come from the
// un-blurred atlas, which is what cv::ORB reads there (it blurs level interiors in place).
// The pattern's points lie within 18.4 pixels of the centre, so the rotated, rounded samples stay within +-18: when
// that square is inside the level (always, for keypoints cv::ORB::detect returns with the default edge threshold)
// the 39 x 39 patch is staged in LDS with 4-byte row loads and the 512 samples are LDS reads; otherwise every
// sample is fetched from memory.
constexpr int kBriefR = 19;                      // staged patch radius
constexpr int kBriefPitch = 40;                  // bytes per staged row (10 dwords)
constexpr int kBriefRows = 2 * kBriefR + 1;

__global__ __launch_bounds__(256) void k_brief(const uint8_t* __restrict__ atlas, const uint8_t* __restrict__ blurred,
                                               size_t atlas_bytes, int aw, const Steer* __restrict__ steer,
                                               const int* __restrict__ counts, int kp_cap,
                                               uint8_t* __restrict__ desc, const BriefPattern* __restrict__ pat,
                                               int chunks, int frames) {
    __shared__ unsigned s_patch[8][kBriefRows * kBriefPitch / 4];
    int b, chunk;
    if (!frame_chunk_of_block(chunks, frames, b, chunk)) return;
    const int slot = threadIdx.x >> 5;
    const int idx = chunk * 8 + slot;
    const int lane = threadIdx.x & 31;
    if (idx >= counts[b] || idx >= kp_cap) return;
    const uint4* sp = (const uint4*)(steer + (size_t)b * kp_cap + idx);
    const uint4 s0 = sp[0], s1 = sp[1];
    const float a = __uint_as_float(s0.x), bb = __uint_as_float(s0.y);
    const int cx = (int)s0.z, cy = (int)s0.w, lw = (int)s1.y, lh = (int)s1.z;
    const size_t org = (size_t)b * atlas_bytes + (size_t)(int)s1.x;
    // four signed bytes per test (x0, y0, x1, y1), this lane's eight tests in two 16-byte loads
    const uint4* qp = (const uint4*)(pat->p + lane * 32);
    const uint4 qa = qp[0], qb = qp[1];
    const int q[8] = {(int)qa.x, (int)qa.y, (int)qa.z, (int)qa.w, (int)qb.x, (int)qb.y, (int)qb.z, (int)qb.w};
    const bool staged = cx - kBriefR >= 0 && cx + kBriefR + 1 < lw && cy - kBriefR >= 0 && cy + kBriefR < lh;
    if (staged) {
        // uniform frame base + one 32-bit offset per lane (an atlas is far smaller than 4 GB)
        const uint8_t* fbb = blurred + (size_t)b * atlas_bytes;
        const unsigned corner = (unsigned)((int)s1.x + (cy - kBriefR) * aw + (cx - kBriefR));
        constexpr int kDwords = kBriefRows * (kBriefPitch / 4), kSteps = (kDwords + 31) / 32;
        unsigned tmp[kSteps];
#pragma unroll
        for (int k = 0; k < kSteps; k++) {  // all row loads in flight, then the LDS stores
            const int i = min(lane + 32 * k, kDwords - 1);
            const int r = div_small<kBriefPitch / 4>(i), c = i - r * (kBriefPitch / 4);
            tmp[k] = load_u32_any(fbb + (corner + __umul24((unsigned)r, (unsigned)aw) + 4u * (unsigned)c));
        }
#pragma unroll
        for (int k = 0; k < kSteps; k++)
            if (lane + 32 * k < kDwords) s_patch[slot][lane + 32 * k] = tmp[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint8_t* sp8 = (const uint8_t*)s_patch[slot] + kBriefR * kBriefPitch + kBriefR;  // the centre pixel
    constexpr float kMagic = 12582912.f;           // 1.5 * 2^23
    constexpr unsigned kMagicBits = 0x4B400000u;   // its bit pattern
    int val = 0;
#pragma unroll
    for (int bit = 0; bit < 8; bit++) {
        const int pq = q[bit];
        const float px0 = (float)(int)(signed char)(pq & 0xff), py0 = (float)(int)(signed char)((pq >> 8) & 0xff);
        const float px1 = (float)(int)(signed char)((pq >> 16) & 0xff), py1 = (float)(int)(signed char)(pq >> 24);
        const float x0 = px0 * a - py0 * bb, y0 = px0 * bb + py0 * a;
        const float x1 = px1 * a - py1 * bb, y1 = px1 * bb + py1 * a;
        // cvRound without v_rndne + v_cvt (two half-rate instructions per coordinate): adding 1.5 * 2^23 rounds to
        // the nearest even integer in the adder (|x| < 2^22), and the sum's bit pattern is kMagicBits + that integer
        const unsigned bx0 = __float_as_uint(x0 + kMagic), by0 = __float_as_uint(y0 + kMagic);
        const unsigned bx1 = __float_as_uint(x1 + kMagic), by1 = __float_as_uint(y1 + kMagic);
        int t0, t1;
        if (staged) {
            // dy * pitch + dx straight from the bit patterns: the low 24 bits of by are 2^22 + dy, and the constants
            // fold into the patch's base address
            constexpr unsigned kFold = 0x400000u * (unsigned)kBriefPitch + kMagicBits;
            t0 = sp8[(int)(__umul24(by0, (unsigned)kBriefPitch) + bx0 - kFold)];
            t1 = sp8[(int)(__umul24(by1, (unsigned)kBriefPitch) + bx1 - kFold)];
        } else {
            const int dx0 = (int)(bx0 - kMagicBits), dy0 = (int)(by0 - kMagicBits);
            const int dx1 = (int)(bx1 - kMagicBits), dy1 = (int)(by1 - kMagicBits);
            const int sx0 = cx + dx0, sy0 = cy + dy0, sx1 = cx + dx1, sy1 = cy + dy1;
            const bool in0 = (unsigned)sx0 < (unsigned)lw && (unsigned)sy0 < (unsigned)lh;
            const bool in1 = (unsigned)sx1 < (unsigned)lw && (unsigned)sy1 < (unsigned)lh;
            const uint8_t* src0 = in0 ? blurred : atlas;
            const uint8_t* src1 = in1 ? blurred : atlas;
            t0 = src0[org + (ptrdiff_t)sy0 * aw + sx0];
            t1 = src1[org + (ptrdiff_t)sy1 * aw + sx1];
        }
        val |= (int)(t0 < t1) << bit;
    }
    desc[((size_t)b * kp_cap + idx) * 32 + lane] = (uint8_t)val;
}

}  // namespace orbx
