// Descriptor matching on gfx950: brute-force Hamming (cv::BFMatcher replacement), the reference's
// radius + mean-absolute-difference matcher, and the two projection ("windowed") matchers.
// Reference: src/geometry/orb_matcher.cpp, include/orb_slam/geometry/orb_matcher.h:45-96,
// src/frame.cpp:89-218, src/geometry/utils.cpp:184-236.
#include <algorithm>
#include <cmath>

#include "common.h"

using namespace orbx;

namespace {

constexpr unsigned long long kNoMatch = ~0ull;
constexpr int kListCap = 2048;   // candidates of one search window
constexpr int kMaxFeatures = 8192;  // features per frame the windowed matcher keeps flags for in LDS

// K9: brute-force Hamming.  A lane holds kHamQ queries (8 dwords each) in registers.  A train row is the same for
// every lane of a wavefront, so it is fetched with SCALAR loads (two rows per 64-byte load) and enters the xor as a
// scalar operand: no LDS staging, no workgroup barrier, and the vector pipes execute nothing but xor, popcount and
// the running minimum (17.5 instructions per descriptor pair).  Within a train chunk the key is
// distance << 16 | row, between chunks distance << 32 | train index: either way the strict-less,
// lowest-index-wins rule of cv::BFMatcher is a plain minimum, so chunks are merged with atomicMin.
// popcount(x) + acc in ONE instruction (v_bcnt_u32_b32 adds its second operand).  Left to itself the compiler counts
// every word into a fresh register and sums them with v_add3_u32: three more half-rate instructions per descriptor pair.
__device__ __forceinline__ unsigned popc_acc(unsigned x, unsigned acc) {
    unsigned d;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(acc));
    return d;
}

__device__ __forceinline__ unsigned hamming256(const uint4& qa, const uint4& qb, const uint4& a, const uint4& b) {
    // one chain of eight: callers keep several independent distances in flight, which hides the chain's latency
    unsigned d = popc_acc(qa.x ^ a.x, 0u);
    d = popc_acc(qb.x ^ b.x, d);
    d = popc_acc(qa.y ^ a.y, d);
    d = popc_acc(qb.y ^ b.y, d);
    d = popc_acc(qa.z ^ a.z, d);
    d = popc_acc(qb.z ^ b.z, d);
    d = popc_acc(qa.w ^ a.w, d);
    d = popc_acc(qb.w ^ b.w, d);
    return d;
}

constexpr int kHamQ = 2;               // queries per lane
constexpr int kHamBlockQ = 256 * kHamQ;  // queries per workgroup

__global__ __launch_bounds__(256) void k_bf_hamming(const uint8_t* __restrict__ q, const int* __restrict__ nq_arr,
                                                    int nq_const, size_t q_stride, const uint8_t* __restrict__ t,
                                                    const int* __restrict__ nt_arr, int nt_const, size_t t_stride,
                                                    int chunk, unsigned long long* __restrict__ best,
                                                    size_t best_stride, int t_sets) {
    // pair p compares query set qs with train set ts: qs = ts = p, or (t_sets > 0) every query set with every
    // train set, p = qs * t_sets + ts
    const int p = blockIdx.z;
    const int qs = t_sets > 0 ? p / t_sets : p, ts = t_sets > 0 ? p - qs * t_sets : p;
    // the per-set counts come from device memory another kernel wrote: never beyond the capacity the buffers were
    // sized for (nq_const / nt_const), never negative
    const int nq = nq_arr ? min(max(nq_arr[qs], 0), nq_const) : nq_const;
    const int nt = nt_arr ? min(max(nt_arr[ts], 0), nt_const) : nt_const;
    const int t0 = blockIdx.y * chunk;
    const int t1 = min(nt, t0 + chunk);
    if (blockIdx.x * kHamBlockQ >= nq || t0 >= t1) return;
    // query j of this lane: blockIdx.x * kHamBlockQ + j * 256 + threadIdx.x (adjacent lanes read adjacent rows)
    uint4 qa[kHamQ], qb[kHamQ];
    unsigned best_k[kHamQ];
#pragma unroll
    for (int j = 0; j < kHamQ; j++) {
        const int qi = blockIdx.x * kHamBlockQ + j * 256 + threadIdx.x;
        qa[j] = qb[j] = make_uint4(0, 0, 0, 0);
        if (qi < nq) {
            const uint4* qp = (const uint4*)(q + (size_t)qs * q_stride + (size_t)qi * 32);
            qa[j] = qp[0];
            qb[j] = qp[1];
        }
        best_k[j] = 0xffffffffu;
    }
    // The train row is the same for every lane: its address is uniform, so the compiler fetches it with scalar
    // loads into scalar registers and the vector pipes see nothing but xor, popcount and the running minimum.
    const uint4* tp = (const uint4*)(t + (size_t)ts * t_stride);
    struct Rows2 { uint4 a0, b0, a1, b1; };   // two consecutive train rows: one 64-byte scalar load
    auto step2 = [&](const Rows2& r, int k) {
        const unsigned rel = (unsigned)(k - t0);
#pragma unroll
        for (int u = 0; u < kHamQ; u++) {
            // key = distance << 16 | row within the chunk: strict-less with the lowest row winning is a plain minimum
            // (one three-way minimum takes both rows)
            const unsigned d0 = hamming256(qa[u], qb[u], r.a0, r.b0), d1 = hamming256(qa[u], qb[u], r.a1, r.b1);
            best_k[u] = min(min(best_k[u], (d0 << 16) + rel), (d1 << 16) + rel + 1u);
        }
    };
    auto load2 = [&](int k) { return *(const Rows2*)(tp + 2 * (size_t)k); };
    // Scalar loads return out of order, so a wait for one of them waits for all that are outstanding.  The loop
    // therefore keeps exactly one in flight: wait for the rows about to be compared (arrived() makes the compiler
    // place that wait), THEN issue the load of the rows after them, then compare - the load has the whole
    // comparison of two rows to arrive.  Past the end the last two rows are taken again (same keys: no change).
    auto arrived = [](const Rows2& r) { asm volatile("" ::"s"(r.a0.x), "s"(r.b1.w)); };
    if (t1 - t0 >= 2) {
        const int last = t1 - 2;
        int ka = t0;
        Rows2 A = load2(ka);
        for (int k = t0; k < t1; k += 4) {
            const int kb = min(k + 2, last);
            arrived(A);
            __builtin_amdgcn_sched_barrier(0);
            const Rows2 B = load2(kb);
            __builtin_amdgcn_sched_barrier(0);
            step2(A, ka);
            const int kc = min(k + 4, last);
            arrived(B);
            __builtin_amdgcn_sched_barrier(0);
            A = load2(kc);
            __builtin_amdgcn_sched_barrier(0);
            step2(B, kb);
            ka = kc;
        }
    } else {
        Rows2 one;
        one.a0 = one.a1 = tp[2 * (size_t)t0];
        one.b0 = one.b1 = tp[2 * (size_t)t0 + 1];
        step2(one, t0);   // the row twice: its second key is the larger one
    }
#pragma unroll
    for (int j = 0; j < kHamQ; j++) {
        const int qi = blockIdx.x * kHamBlockQ + j * 256 + threadIdx.x;
        if (qi < nq)
            atomicMin(&best[(size_t)p * best_stride + qi],
                      ((unsigned long long)(best_k[j] >> 16) << 32) | (unsigned)(t0 + (int)(best_k[j] & 0xffffu)));
    }
}

// K9 on the matrix cores.  The Hamming distance of two descriptors is a dot product: with the train bits a_k as
// {0, 1} and the query bits b_k as b'_k = 1 - 2 b_k in {+1, -1},
//     sum_k a_k b'_k = #(a=1, b=0) - #(a=1, b=1) = |a| - 2 #(a=1, b=1),   so   distance = |a| + |b| - 2 #(1, 1) = |b| + sum_k a_k b'_k:
// the query's own popcount plus one exact int8 product per bit - 32 train rows x 32 queries x 32 bits per
// v_mfma_i32_32x32x32_i8, eight of them per 32 x 32 tile of descriptor pairs.  A wavefront keeps the expanded fragments
// of 64 queries (two tiles) in registers; a lane (r = lane & 31, h = lane >> 5) loads half h (128 bits) of train row r
// of the current tile, expands 16 bits per step into 16 bytes (four bits per dword by one 24-bit multiply and a mask),
// and the accumulator tile comes back with the lane's own query in all 16 registers (column = lane & 31, row =
// (reg & 3) + 8 (reg >> 2) + 4 h), so the running minimum of key = distance << 16 | row is per lane and register: two
// vector instructions per descriptor pair of a lane instead of seventeen (they share the SIMD's issue with the
// products and set the pace: the matrix pipe is busy 40-45 % of the time).  Which bit sits in which slot of a fragment
// does not matter as long as both operands use the same rule (they share spread16).  Same arguments, same packed-key
// result and the same lowest-index-wins rule as k_bf_hamming.
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v16i_t __attribute__((ext_vector_type(16)));

__device__ __forceinline__ v4i_t spread16(unsigned hw) {   // 16 bits -> 16 bytes of 0 / 1
    v4i_t r;
    r.x = (int)(__umul24(hw & 0xfu, 0x204081u) & 0x01010101u);
    r.y = (int)(__umul24((hw >> 4) & 0xfu, 0x204081u) & 0x01010101u);
    r.z = (int)(__umul24((hw >> 8) & 0xfu, 0x204081u) & 0x01010101u);
    r.w = (int)(__umul24((hw >> 12) & 0xfu, 0x204081u) & 0x01010101u);
    return r;
}

constexpr int kMfmaQ = 2;                    // 32-query tiles per wavefront
constexpr int kMfmaSub = 1;                  // 32-row train tiles per barrier (2 and 4 measured the same)
constexpr int kMfmaBlockQ = 4 * 32 * kMfmaQ;  // queries per workgroup

__global__ __launch_bounds__(256, 2) void k_bf_hamming_mfma(const uint8_t* __restrict__ q, const int* __restrict__ nq_arr,
                                                         int nq_const, size_t q_stride, const uint8_t* __restrict__ t,
                                                         const int* __restrict__ nt_arr, int nt_const, size_t t_stride,
                                                         int chunk, unsigned long long* __restrict__ best,
                                                         size_t best_stride, int t_sets) {
    const int p = blockIdx.z;
    const int qs = t_sets > 0 ? p / t_sets : p, ts = t_sets > 0 ? p - qs * t_sets : p;
    // the per-set counts come from device memory another kernel wrote: never beyond the capacity the buffers were
    // sized for (nq_const / nt_const), never negative
    const int nq = nq_arr ? min(max(nq_arr[qs], 0), nq_const) : nq_const;
    const int nt = nt_arr ? min(max(nt_arr[ts], 0), nt_const) : nt_const;
    const int t0 = blockIdx.y * chunk;
    const int t1 = min(nt, t0 + chunk);
    if (blockIdx.x * kMfmaBlockQ >= nq || t0 >= t1) return;
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int q0 = blockIdx.x * kMfmaBlockQ + wv * (32 * kMfmaQ);
    const bool has_queries = q0 < nq;   // a wavefront without queries still expands its share of every train tile
    // The expanded train tiles are the same for the four wavefronts of the workgroup: each expands two of the eight
    // steps (one dword of its lanes' row halves) into LDS, fragment-major, so that writes and reads are whole
    // 16-byte words of consecutive lanes.  A stage is kMfmaSub tiles of 32 rows; two stage buffers, one barrier per
    // stage.
    __shared__ uint4 s_frag[2][kMfmaSub][8][64];
    v4i_t bfrag[kMfmaQ][8];
    unsigned base_q[kMfmaQ];
#pragma unroll
    for (int u = 0; u < kMfmaQ; u++) {
        const int qi = q0 + 32 * u + r;
        uint4 qa = make_uint4(0, 0, 0, 0), qb = qa;
        if (qi < nq) {
            const uint4* qp = (const uint4*)(q + (size_t)qs * q_stride + (size_t)qi * 32);
            qa = qp[0];
            qb = qp[1];
        }
        base_q[u] = (unsigned)(__popc(qa.x) + __popc(qa.y) + __popc(qa.z) + __popc(qa.w) + __popc(qb.x) + __popc(qb.y) +
                               __popc(qb.z) + __popc(qb.w)) << 16;
        const uint4 half = h ? qb : qa;
        const unsigned w[4] = {half.x, half.y, half.z, half.w};
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const v4i_t sp = spread16((w[s >> 1] >> (16 * (s & 1))) & 0xffffu);
            // 0 -> +1, 1 -> -1 in every byte (254 = 0xfe: no carry between bytes)
            bfrag[u][s].x = (int)((unsigned)sp.x * 254u + 0x01010101u);
            bfrag[u][s].y = (int)((unsigned)sp.y * 254u + 0x01010101u);
            bfrag[u][s].z = (int)((unsigned)sp.z * 254u + 0x01010101u);
            bfrag[u][s].w = (int)((unsigned)sp.w * 254u + 0x01010101u);
        }
    }
    unsigned best_k[kMfmaQ][16];
#pragma unroll
    for (int u = 0; u < kMfmaQ; u++)
#pragma unroll
        for (int j = 0; j < 16; j++) best_k[u][j] = 0xffffffffu;
    // dword wv of half h of train row (tile + r): steps 2 wv and 2 wv + 1 of this lane's fragments
    const uint8_t* tp = t + (size_t)ts * t_stride + 16 * h + 4 * wv;
    auto load_word = [&](int row) {
        unsigned v = 0u;
        if (row < t1) v = *(const unsigned*)(tp + (size_t)row * 32);
        return v;
    };
    auto expand_into = [&](int buf, int sub, unsigned word) {
        const v4i_t lo = spread16(word & 0xffffu), hi = spread16(word >> 16);
        s_frag[buf][sub][2 * wv][lane] = make_uint4((unsigned)lo.x, (unsigned)lo.y, (unsigned)lo.z, (unsigned)lo.w);
        s_frag[buf][sub][2 * wv + 1][lane] = make_uint4((unsigned)hi.x, (unsigned)hi.y, (unsigned)hi.z, (unsigned)hi.w);
    };
    constexpr int kStageRows = 32 * kMfmaSub;
    unsigned nxt[kMfmaSub];
#pragma unroll
    for (int sub = 0; sub < kMfmaSub; sub++) expand_into(0, sub, load_word(t0 + 32 * sub + r));
#pragma unroll
    for (int sub = 0; sub < kMfmaSub; sub++) nxt[sub] = load_word(t0 + kStageRows + 32 * sub + r);
    __syncthreads();
    int buf = 0;
    for (int sb = t0; sb < t1; sb += kStageRows, buf ^= 1) {
        unsigned word[kMfmaSub];
#pragma unroll
        for (int sub = 0; sub < kMfmaSub; sub++) {
            word[sub] = nxt[sub];
            nxt[sub] = load_word(sb + 2 * kStageRows + 32 * sub + r);
        }
        if (has_queries) {
#pragma unroll
            for (int sub = 0; sub < kMfmaSub; sub++) {
                const int tb = sb + 32 * sub;
                if (tb >= t1) break;   // workgroup-uniform
                v16i_t acc[kMfmaQ];
#pragma unroll
                for (int u = 0; u < kMfmaQ; u++)
#pragma unroll
                    for (int j = 0; j < 16; j++) acc[u][j] = 0;
                // all eight fragment reads of the tile are issued before the first product needs one
                v4i_t a[8];
#pragma unroll
                for (int s = 0; s < 8; s++) {
                    const uint4 f = s_frag[buf][sub][s][lane];
                    a[s].x = (int)f.x;
                    a[s].y = (int)f.y;
                    a[s].z = (int)f.z;
                    a[s].w = (int)f.w;
                }
#pragma unroll
                for (int s = 0; s < 8; s++)
#pragma unroll
                    for (int u = 0; u < kMfmaQ; u++)
                        acc[u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s], bfrag[u][s], acc[u], 0, 0, 0);
                // row of register j within the tile: (j & 3) + 8 (j >> 2) + 4 h; the part that depends on j is added
                // at the end
                const unsigned rel = (unsigned)(tb - t0) + 4u * (unsigned)h;
                unsigned key[kMfmaQ][16];
#pragma unroll
                for (int u = 0; u < kMfmaQ; u++) {
                    const unsigned base = base_q[u] + rel;
#pragma unroll
                    for (int j = 0; j < 16; j++) key[u][j] = ((unsigned)acc[u][j] << 16) + base;
                }
                if (tb + 32 > t1) {   // workgroup-uniform: only the last tile of a chunk can have rows past the end
#pragma unroll
                    for (int j = 0; j < 16; j++)
                        if (tb + (j & 3) + 8 * (j >> 2) + 4 * h >= t1) {
#pragma unroll
                            for (int u = 0; u < kMfmaQ; u++) key[u][j] = 0xffffffffu;
                        }
                }
#pragma unroll
                for (int u = 0; u < kMfmaQ; u++)
#pragma unroll
                    for (int j = 0; j < 16; j++) best_k[u][j] = min(best_k[u][j], key[u][j]);
            }
        }
        // the next stage goes into the other buffer, which every wavefront finished reading before the last barrier
        if (sb + kStageRows < t1) {
#pragma unroll
            for (int sub = 0; sub < kMfmaSub; sub++) expand_into(buf ^ 1, sub, word[sub]);
        }
        __syncthreads();
    }
    if (!has_queries) return;
#pragma unroll
    for (int u = 0; u < kMfmaQ; u++) {
        unsigned k = 0xffffffffu;
#pragma unroll
        for (int j = 0; j < 16; j++)
            if (best_k[u][j] != 0xffffffffu) k = min(k, best_k[u][j] + (unsigned)((j & 3) + 8 * (j >> 2)));
        const int qi = q0 + 32 * u + r;
        if (qi < nq && k != 0xffffffffu)
            atomicMin(&best[(size_t)p * best_stride + qi],
                      ((unsigned long long)(k >> 16) << 32) | (unsigned)(t0 + (int)(k & 0xffffu)));
    }
}

__global__ void k_unpack_best(const unsigned long long* __restrict__ best, size_t best_stride,
                              const int* __restrict__ nq_arr, int nq_const, int32_t* __restrict__ idx,
                              int32_t* __restrict__ dist, size_t out_stride, int t_sets) {
    const int p = blockIdx.y;
    const int nq = nq_arr ? min(max(nq_arr[t_sets > 0 ? p / t_sets : p], 0), nq_const) : nq_const;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const unsigned long long k = best[(size_t)p * best_stride + i];
    idx[(size_t)p * out_stride + i] = k == kNoMatch ? -1 : (int32_t)(k & 0xffffffffu);
    dist[(size_t)p * out_stride + i] = k == kNoMatch ? -1 : (int32_t)(k >> 32);
}

// BruteForceWithRadiusMatcher (orb_matcher.cpp:25-74): sum of absolute byte differences (v_sad_u8) over reference
// keypoints within the pixel radius; strict-less, first-wins, so the same packed-key minimum applies.
__global__ __launch_bounds__(256) void k_bf_sad_radius(const uint8_t* __restrict__ q, const float2* __restrict__ qxy,
                                                       int nq, const uint8_t* __restrict__ t,
                                                       const float2* __restrict__ txy, int nt, int chunk,
                                                       double max_sq, unsigned long long* __restrict__ best) {
    __shared__ uint4 tile[512];
    __shared__ float2 txy_s[256];
    const int qi = blockIdx.x * 256 + threadIdx.x;
    const int t0 = blockIdx.y * chunk;
    const int t1 = min(nt, t0 + chunk);
    if (t0 >= t1) return;
    uint4 qa = make_uint4(0, 0, 0, 0), qb = qa;
    float2 qp = make_float2(0.f, 0.f);
    if (qi < nq) {
        const uint4* p = (const uint4*)(q + (size_t)qi * 32);
        qa = p[0];
        qb = p[1];
        qp = qxy[qi];
    }
    const uint4* tp = (const uint4*)t;
    unsigned best_d = 0x7fffffffu;
    int best_i = -1;
    for (int tb = t0; tb < t1; tb += 256) {
        __syncthreads();
        const int j = tb + threadIdx.x;
        if (j < t1) {
            tile[2 * threadIdx.x] = tp[2 * (size_t)j];
            tile[2 * threadIdx.x + 1] = tp[2 * (size_t)j + 1];
            txy_s[threadIdx.x] = txy[j];
        }
        __syncthreads();
        const int cnt = min(256, t1 - tb);
        for (int k = 0; k < cnt; k++) {
            const float2 r = txy_s[k];
            const float dx = r.x - qp.x, dy = r.y - qp.y;
            const float sq = dx * dx + dy * dy;
            if ((double)sq <= max_sq) {
                const uint4 a = tile[2 * k], b = tile[2 * k + 1];
                unsigned d = 0;
                d = __builtin_amdgcn_sad_u8(qa.x, a.x, d);
                d = __builtin_amdgcn_sad_u8(qa.y, a.y, d);
                d = __builtin_amdgcn_sad_u8(qa.z, a.z, d);
                d = __builtin_amdgcn_sad_u8(qa.w, a.w, d);
                d = __builtin_amdgcn_sad_u8(qb.x, b.x, d);
                d = __builtin_amdgcn_sad_u8(qb.y, b.y, d);
                d = __builtin_amdgcn_sad_u8(qb.z, b.z, d);
                d = __builtin_amdgcn_sad_u8(qb.w, b.w, d);
                if (d < best_d) {
                    best_d = d;
                    best_i = tb + k;
                }
            }
        }
    }
    if (qi < nq && best_i >= 0) atomicMin(&best[qi], ((unsigned long long)best_d << 32) | (unsigned)best_i);
}

struct ProjArgs {
    orbx_grid g;
    unsigned gx_magic, gy_magic;  // ceil(2^32 / grid_size_x), ceil(2^32 / grid_size_y): see div_cell (0 when the size is 1)
    // per-frame arrays; the single-pair entry points use stride 0 and no index arrays
    const int32_t* frame_index;  // [pairs] or NULL (frame 0)
    const int32_t* ref_index;    // [pairs] or NULL (reference 0)
    const int32_t* counts;       // features per frame, or NULL (n / n_ref below)
    size_t kp_stride;            // keypoints (and cell items) per frame slot
    size_t cell_stride;          // cell_start entries per frame slot
    size_t pair_stride;          // entries per pair in the per-pair arrays
    size_t out_stride;           // entries per pair in matched / dists (0: pair_stride)
    const int32_t* n_ref_pair;   // reference points per PAIR (map-point mode, batched), or NULL
    int ref_desc_per_pair;       // reference descriptors are a per-pair array (map points), not a per-frame one
    const int32_t* cell_start;
    const int32_t* cell_items;
    const orbx_keypoint* kps;   // frame features
    int n;
    const uint8_t* desc;        // frame descriptors
    const orbx_keypoint* ref_kps;  // mode 0: reference features (octave, angle); mode 1: unused
    int n_ref;
    const uint8_t* ref_flag;    // mode 0: has map point; mode 1: in view; NULL = all set
    const float* proj_xy;
    const float* proj_z;        // mode 0 only; NULL = positive
    const int32_t* pred_level;  // mode 1 only
    const float* view_cos;      // mode 1 only
    const uint8_t* ref_desc;
    const float* scale_factors;
    int n_levels;
    int level_mode;
    float radius;
    float nn_ratio;
    int check_orientation;
    int32_t* matched;           // out [pairs][pair_stride]
    int32_t* dists;             // out [pairs][pair_stride]
    int32_t* pushes;            // scratch [pairs][2 * pair_stride]: (bin, feature) of every histogram push
    int32_t* status;            // out [pairs]: 1 = a window held more than kListCap candidates
    int32_t* n_matches;         // out [pairs] or NULL
    // pre-gathered search windows (K10a), or NULL: the resolve kernel gathers by itself
    int32_t* win_n;             // [pairs][pair_stride]
    int2* win_list;             // [pairs][pair_stride][kWinK] (feature index, Hamming distance)
    int win_stride;             // entries per reference point in win_list
    int32_t* win_overflow;      // [pairs]
    int32_t* need_slow;         // [pairs] or NULL: K10c sets it for a pair with an over-long window; K10b then runs
                                // only for flagged pairs
};

__device__ __forceinline__ int hamming32(const uint8_t* a, const uint8_t* b) {
    const uint4* pa = (const uint4*)a;
    const uint4* pb = (const uint4*)b;
    const uint4 a0 = pa[0], a1 = pa[1], b0 = pb[0], b1 = pb[1];
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// K10: projection matchers, one wavefront per frame.  Reference points are visited in order (the greedy
// "already matched" rule of orb_matcher.cpp:216,313 makes the outer loop sequential); for each one the 64 lanes
// gather the search window from the CSR bucket grid in the reference's cell-scan order (x-major cells, features in
// index order inside a cell), compute the Hamming distances in parallel, and lane 0 resolves the ordered list.
// MODE 0 = frame-to-frame (orb_matcher.cpp:115-255), MODE 1 = map points (orb_matcher.cpp:257-352).
constexpr int kWinK = 64;  // entries per reference point in the pre-gathered window lists

// Minimum over the 64 lanes with DPP row operations; every lane must be active.
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x141, 0xF, 0xF, false));  // row_half_mirror
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x140, 0xF, 0xF, false));  // row_mirror
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x142, 0xA, 0xF, false));  // row_bcast15
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x143, 0xC, 0xF, false));  // row_bcast31
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

struct Window {
    float px, py, radius_scaled;
    int min_level, max_level, left, up, ncx, ncy, ncells;
    bool min_only;
};

// num / d as C divides (towards zero), for the cell arithmetic of getBoxAroundPoint (frame.cpp:89-118): d is a grid
// cell size, the same for every lane, and magic = ceil(2^32 / d).  floor(a * magic / 2^32) = floor(a / d) while
// a * d < 2^32, so numerators below 2^20 and sizes below 2^11 take one multiply-high instead of a division routine.
__device__ __forceinline__ int div_cell(int num, int d, unsigned magic) {
    const unsigned a = (unsigned)(num < 0 ? -num : num);
    int q;
    if (__builtin_expect(a < (1u << 20) && d > 1 && d < (1 << 11), 1)) q = (int)__umulhi(a, magic);
    else q = (int)(a / (unsigned)d);
    return num < 0 ? -q : q;
}

inline void set_grid(ProjArgs& A, const orbx_grid* g) {
    A.g = *g;
    A.gx_magic = g->grid_size_x > 1 ? (unsigned)((0x100000000ull + (unsigned)g->grid_size_x - 1) / (unsigned)g->grid_size_x) : 0u;
    A.gy_magic = g->grid_size_y > 1 ? (unsigned)((0x100000000ull + (unsigned)g->grid_size_y - 1) / (unsigned)g->grid_size_y) : 0u;
}

// Points A's per-frame / per-pair views at pair `pair`.
__device__ __forceinline__ void resolve_views(ProjArgs& A, int pair) {
    const int fi = A.frame_index ? A.frame_index[pair] : 0;
    const int ri = A.ref_index ? A.ref_index[pair] : 0;
    // counts come from device memory another kernel wrote: whatever they hold, a frame has kp_stride slots
    A.n = A.counts ? min(max(A.counts[fi], 0), (int)A.kp_stride) : A.n;
    A.n_ref = A.n_ref_pair ? A.n_ref_pair[pair] : (A.counts ? min(max(A.counts[ri], 0), (int)A.kp_stride) : A.n_ref);
    A.kps += (size_t)fi * A.kp_stride;
    A.desc += (size_t)fi * A.kp_stride * 32;
    A.cell_start += (size_t)fi * A.cell_stride;
    A.cell_items += (size_t)fi * A.kp_stride;
    if (A.ref_kps) A.ref_kps += (size_t)ri * A.kp_stride;
    const size_t po = (size_t)pair * A.pair_stride;
    A.ref_desc += A.ref_desc_per_pair ? po * 32 : (size_t)ri * A.kp_stride * 32;
    A.n_ref = min(A.n_ref, A.pair_stride ? (int)A.pair_stride : A.n_ref);
    if (A.ref_flag) A.ref_flag += po;
    A.proj_xy += po * 2;
    if (A.proj_z) A.proj_z += po;
    if (A.pred_level) A.pred_level += po;
    if (A.view_cos) A.view_cos += po;
    const size_t oo = (size_t)pair * (A.out_stride ? A.out_stride : A.pair_stride);
    A.matched += oo;
    A.dists += oo;
    A.pushes += po * 2;
    A.status += pair;
    if (A.n_matches) A.n_matches += pair;
    if (A.need_slow) A.need_slow += pair;
    if (A.win_n) {
        A.win_n += po;
        A.win_list += po * (size_t)A.win_stride;
        A.win_overflow += pair;
    }
    A.n = min(A.n, kMaxFeatures);
}

// The per-reference-point tests and the search window (orb_matcher.cpp:146-209 / 268-303, frame.cpp:89-118).
template <int MODE>
__device__ __forceinline__ bool window_of(const ProjArgs& A, int ref_idx, Window& W) {
    const orbx_grid& g = A.g;
    if (A.ref_flag && !A.ref_flag[ref_idx]) return false;
    W.px = A.proj_xy[2 * ref_idx];
    W.py = A.proj_xy[2 * ref_idx + 1];
    W.min_only = false;
    int level;
    if (MODE == 0) {
        if (A.proj_z && A.proj_z[ref_idx] < 0) return false;
        if (!(W.px > g.min_x && W.px < g.max_x && W.py > g.min_y && W.py < g.max_y)) return false;
        level = A.ref_kps[ref_idx].octave;
        if (level < 0 || level >= A.n_levels) return false;
        W.radius_scaled = A.radius * A.scale_factors[level];
        if (A.level_mode == 1) { W.min_level = level; W.max_level = 0x7fffffff; W.min_only = true; }
        else if (A.level_mode == 2) { W.min_level = 0; W.max_level = level; }
        else { W.min_level = level - 1; W.max_level = level + 1; }
    } else {
        level = A.pred_level[ref_idx];
        if (level < 0 || level >= A.n_levels) return false;
        const float by_cos = A.view_cos[ref_idx] > 0.998 ? 2.5f : 4.0f;
        W.radius_scaled = A.radius * by_cos * A.scale_factors[level];
        W.min_level = level - 1;
        W.max_level = level;
    }
    W.left = max(0, div_cell((int)floorf(W.px - g.min_x - W.radius_scaled), g.grid_size_x, A.gx_magic));
    if (W.left >= g.grid_cols) return false;
    const int right = min(g.grid_cols - 1, div_cell((int)ceilf(W.px - g.min_x + W.radius_scaled), g.grid_size_x, A.gx_magic));
    if (right < 0) return false;
    W.up = max(0, div_cell((int)floorf(W.py - g.min_y - W.radius_scaled), g.grid_size_y, A.gy_magic));
    if (W.up >= g.grid_rows) return false;
    const int down = min(g.grid_rows - 1, div_cell((int)ceilf(W.py - g.min_y + W.radius_scaled), g.grid_size_y, A.gy_magic));
    if (down < 0) return false;
    W.ncx = right - W.left + 1;
    W.ncy = down - W.up + 1;
    W.ncells = W.ncx * W.ncy;
    return true;
}

// Feature test of Frame::getFeaturesAroundPoint: 4-argument overload (frame.cpp:151-183, compares with the CELL
// index as the reference does) when min_only, 5-argument overload (frame.cpp:185-218) otherwise.
__device__ __forceinline__ bool in_window(const orbx_keypoint& k, const Window& W, int cxi, int cyi) {
    if (W.min_only) {
        if (k.octave < W.min_level) return false;
        return fabsf(k.x - (float)cxi) < W.radius_scaled && fabsf(k.y - (float)cyi) < W.radius_scaled;
    }
    if (k.octave < W.min_level || k.octave > W.max_level) return false;
    return fabsf(k.x - W.px) < W.radius_scaled && fabsf(k.y - W.py) < W.radius_scaled;
}

// K10a: window gather, fully parallel: one wavefront per reference point collects every feature of its search
// window in the reference's scan order (x-major cells, index order inside a cell) with its Hamming distance.
// The "already matched" rule is NOT applied here (it is sequential); K10b applies it while resolving.
// win_n[ref] = -1: point skipped; > kWinK: list too long, K10b gathers that point's window itself.
template <int MODE>
__global__ __launch_bounds__(256) void k_window_gather(ProjArgs A) {
    // 16 lanes per reference point (a level-0 window spans ~30 grid cells): the work is a chain of dependent loads,
    // so four points per wavefront quadruple the loads in flight
    const int lane = threadIdx.x & 15;
    resolve_views(A, blockIdx.y);
    const int ref_idx = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (ref_idx >= A.n_ref) return;
    const orbx_grid& g = A.g;
    Window W;
    if (!window_of<MODE>(A, ref_idx, W)) {
        if (lane == 0) A.win_n[ref_idx] = -1;
        return;
    }
    const uint8_t* rd = A.ref_desc + (size_t)ref_idx * 32;
    int2* list = A.win_list + (size_t)ref_idx * kWinK;
    int list_n = 0;
    for (int cbase = 0; cbase < W.ncells && list_n <= kWinK; cbase += 16) {
        const int c = cbase + lane;
        int i0 = 0, i1 = 0, cxi = 0, cyi = 0;
        if (c < W.ncells) {
            cxi = W.left + c / W.ncy;
            cyi = W.up + c % W.ncy;
            const int cell = cxi * g.grid_rows + cyi;
            i0 = A.cell_start[cell];
            i1 = A.cell_start[cell + 1];
        }
        // accepted items of this lane's cell: a bit mask for the first 32, the rest re-tested when written
        unsigned acc = 0;
        int cnt = 0;
        for (int i = i0; i < i1; i++) {
            const bool ok = in_window(A.kps[A.cell_items[i]], W, cxi, cyi);
            if (ok && i - i0 < 32) acc |= 1u << (i - i0);
            cnt += ok;
        }
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            int v = __shfl_up(incl, o, 16);
            if (lane >= o) incl += v;
        }
        const int total = __shfl(incl, 15, 16);
        int pos = list_n + incl - cnt;
        list_n += total;
        if (list_n > kWinK) break;
        for (int i = i0; i < i1; i++) {
            const bool ok = i - i0 < 32 ? ((acc >> (i - i0)) & 1u) != 0 : in_window(A.kps[A.cell_items[i]], W, cxi, cyi);
            if (ok) {
                const int idx = A.cell_items[i];
                list[pos++] = make_int2(idx, hamming32(rd, A.desc + (size_t)idx * 32));
            }
        }
    }
    if (lane == 0) A.win_n[ref_idx] = list_n;
}

// K10b: projection matchers, one wavefront per frame pair.  Reference points are visited in order (the greedy
// "already matched" rule of orb_matcher.cpp:216,313 makes the outer loop sequential).  With pre-gathered window
// lists (K10a) each step is one coalesced list load (prefetched one point ahead), a flag lookup in LDS and a
// wave reduction; without them (single-pair entry points, or a window longer than kWinK) the 64 lanes gather the
// window from the CSR grid themselves.  MODE 0 = frame-to-frame (orb_matcher.cpp:115-255), MODE 1 = map points
// (orb_matcher.cpp:257-352).
template <int MODE>
__global__ __launch_bounds__(64) void k_projection_match(ProjArgs A) {
    extern __shared__ int32_t s_matched[];  // [n]
    __shared__ int32_t s_idx[kListCap];
    __shared__ int32_t s_dist[kListCap];
    __shared__ int32_t s_hist[30];
    const int lane = threadIdx.x;
    const orbx_grid g = A.g;
    if (A.need_slow && A.need_slow[blockIdx.x] == 0) return;
    resolve_views(A, blockIdx.x);
    {
        // flags cover the whole frame slot in the batched form: a stale grid entry (the reference's compute() may
        // erase keypoints after the grid was built) then reads "unmatched", never uninitialised memory
        const int n_init = A.kp_stride ? (int)A.kp_stride : A.n;
        for (int i = lane; i < n_init; i += 64) {
            s_matched[i] = -1;
            if (i < A.n || A.kp_stride) A.dists[i] = -1;
        }
    }
    if (lane < 30) s_hist[lane] = 0;
    __syncthreads();
    int n_push = 0;
    const float factor = 1.0f / 30;
    const bool use_lists = A.win_n != nullptr;
    // software prefetch, kPF reference points ahead: the list loads do not depend on anything the loop computes
    constexpr int kPF = 8;
    int pf_n[kPF];
    int2 pf_e[kPF];
#pragma unroll
    for (int k = 0; k < kPF; k++) {
        pf_n[k] = -1;
        pf_e[k] = make_int2(-1, 0);
        if (use_lists && k < A.n_ref) {
            pf_n[k] = A.win_n[k];
            pf_e[k] = A.win_list[(size_t)k * kWinK + lane];
        }
    }
    for (int ref_base = 0; ref_base < A.n_ref; ref_base += kPF) {
      int blk_n[kPF];
      int2 blk_e[kPF];
#pragma unroll
      for (int k = 0; k < kPF; k++) {
          blk_n[k] = pf_n[k];
          blk_e[k] = pf_e[k];
          const int nx = ref_base + kPF + k;
          if (use_lists && nx < A.n_ref) {
              pf_n[k] = A.win_n[nx];
              pf_e[k] = A.win_list[(size_t)nx * kWinK + lane];
          }
      }
#pragma unroll
      for (int k = 0; k < kPF; k++) {
        const int ref_idx = ref_base + k;
        if (ref_idx >= A.n_ref) break;
        int list_n = 0;
        int cur_n = 0;
        int2 cur_e = make_int2(-1, 0);
        if (use_lists) {
            cur_n = blk_n[k];
            cur_e = blk_e[k];
            if (cur_n <= 0) continue;
        }
        // a window longer than kWinK (rare) is gathered again here, sequentially, for this point only
        if (use_lists && cur_n <= kWinK) {
            const bool valid = lane < cur_n && !(s_matched[cur_e.x] > 0);
            if (MODE == 0) {
                // min distance, last in scan order wins ties (`<=` in orb_matcher.cpp:222): one 32-bit key per lane,
                // reduced with DPP row operations (no LDS round trips)
                const unsigned key = wave_min_u32(valid ? (((unsigned)cur_e.y << 8) | (unsigned)(63 - lane)) : ~0u);
                if (key == ~0u) continue;
                const int min_d = (int)(key >> 8);
                const int id = __builtin_amdgcn_readlane(cur_e.x, 63 - (int)(key & 0xffu));
                if (min_d <= 100) {
                    if (lane == 0) {
                        s_matched[id] = ref_idx;
                        A.dists[id] = min_d;
                        if (A.check_orientation) {  // the rotation bin is computed after the loop
                            A.pushes[2 * n_push] = ref_idx;
                            A.pushes[2 * n_push + 1] = id;
                            n_push++;
                        }
                    }
                    // one wavefront: LDS operations retire in program order, so the flag written above is seen by
                    // the next point's lookup without a barrier (and without waiting for the global stores)
                    __builtin_amdgcn_wave_barrier();
                }
                continue;
            }
            // MODE 1: compact the still-free entries in order, then lane 0 walks them (below)
            const unsigned long long bal = __ballot(valid);
            if (valid) {
                const int pos = __popcll(bal & ((1ull << lane) - 1ull));
                s_idx[pos] = cur_e.x;
                s_dist[pos] = cur_e.y;
            }
            list_n = __popcll(bal);
            __syncthreads();
        } else {
            Window W;
            if (!window_of<MODE>(A, ref_idx, W)) continue;
            // ordered gather of the window
            bool overflow = false;
            for (int cbase = 0; cbase < W.ncells; cbase += 64) {
                const int c = cbase + lane;
                int cnt = 0, i0 = 0, i1 = 0, cxi = 0, cyi = 0;
                if (c < W.ncells) {
                    cxi = W.left + c / W.ncy;
                    cyi = W.up + c % W.ncy;
                    const int cell = cxi * g.grid_rows + cyi;
                    i0 = A.cell_start[cell];
                    i1 = A.cell_start[cell + 1];
                }
                auto accept = [&](int idx) -> bool {
                    if (s_matched[idx] > 0) return false;
                    return in_window(A.kps[idx], W, cxi, cyi);
                };
                for (int i = i0; i < i1; i++) cnt += accept(A.cell_items[i]) ? 1 : 0;
                int incl = cnt;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    int v = __shfl_up(incl, o);
                    if (lane >= o) incl += v;
                }
                const int total = __shfl(incl, 63);
                int pos = list_n + incl - cnt;
                if (list_n + total > kListCap) overflow = true;
                else
                    for (int i = i0; i < i1; i++) {
                        const int idx = A.cell_items[i];
                        if (accept(idx)) s_idx[pos++] = idx;
                    }
                if (overflow) break;
                list_n += total;
            }
            if (overflow) {
                if (lane == 0) *A.status = 1;
                ref_base = A.n_ref;  // leave both loops
                break;
            }
            __syncthreads();
            const uint8_t* rd = A.ref_desc + (size_t)ref_idx * 32;
            for (int i = lane; i < list_n; i += 64) s_dist[i] = hamming32(rd, A.desc + (size_t)s_idx[i] * 32);
            __syncthreads();
        }
        if (lane == 0 && list_n > 0) {
            if (MODE == 0) {
                int min_d = 256, id = -1;
                for (int i = 0; i < list_n; i++)
                    if (s_dist[i] <= min_d) { min_d = s_dist[i]; id = s_idx[i]; }
                if (min_d <= 100 && id >= 0) {
                    s_matched[id] = ref_idx;
                    A.dists[id] = min_d;
                    if (A.check_orientation) {
                        A.pushes[2 * n_push] = ref_idx;
                        A.pushes[2 * n_push + 1] = id;
                        n_push++;
                    }
                }
            } else {
                int min_d = 256, id1 = -1, min2 = 256, id2 = -1;
                for (int i = 0; i < list_n; i++) {
                    const int d = s_dist[i], idx = s_idx[i];
                    if (d < min_d) { min2 = min_d; id2 = id1; min_d = d; id1 = idx; }
                    else if (d < min2) { min2 = d; id2 = idx; }
                    if (min_d <= 100) {
                        // no second candidate yet: the reference reads key_points[-1]; treated as "octaves differ"
                        if (id2 >= 0 && A.kps[id1].octave == A.kps[id2].octave && (float)min_d > A.nn_ratio * (float)min2)
                            continue;
                        s_matched[id1] = ref_idx;
                        A.dists[id1] = min_d;
                    }
                }
            }
        }
        __syncthreads();
      }
    }
    __syncthreads();
    if (MODE == 0 && A.check_orientation) {
        // MatcherBase::applyRotationConstraint + computeThreeMaxima (orb_matcher.h:57-79, utils.cpp:202-236)
        n_push = __shfl(n_push, 0);
        // rotation-difference bin of every accepted match (orb_matcher.cpp:231-243), deferred to here so that the
        // sequential loop above never waits on the two angle loads
        for (int i = lane; i < n_push; i += 64) {
            float rot = A.ref_kps[A.pushes[2 * i]].angle - A.kps[A.pushes[2 * i + 1]].angle;
            if (rot < 0.0f) rot += 360.0f;
            int bin = (int)roundf(rot * factor);
            if (bin == 30) bin = 0;
            if (bin < 0 || bin >= 30) bin = -1;
            else atomicAdd(&s_hist[bin], 1);
            A.pushes[2 * i] = bin;
        }
        __syncthreads();
        int ind1 = -1, ind2 = -1, ind3 = -1;
        int max1 = 0, max2 = 0, max3 = 0;
        for (int i = 0; i < 30; i++) {
            const int s = s_hist[i];
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
        for (int i = lane; i < n_push; i += 64) {
            const int bin = A.pushes[2 * i];
            if (bin >= 0 && bin != ind1 && bin != ind2 && bin != ind3) s_matched[A.pushes[2 * i + 1]] = 0;
        }
        __syncthreads();
    }
    int cnt = 0;
    for (int i = lane; i < A.n; i += 64) {
        const int m = s_matched[i];
        A.matched[i] = m;
        cnt += m > 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (lane == 0 && A.n_matches) *A.n_matches = cnt;
}

// Frame-to-frame rule for ONE reference point without a pre-gathered list: every lane scans grid cells of the
// window, keeps its best (distance, scan position) among the still-free features, and the wave takes the minimum
// distance with the LAST scan position on ties (`<=` in orb_matcher.cpp:222).  Returns (distance << 32 | feature)
// or -1 when the window holds no free feature.
__device__ __noinline__ long long best_in_window_scan(const ProjArgs& A, int ref_idx, const int32_t* s_matched,
                                                      int lane) {
    Window W;
    if (!window_of<0>(A, ref_idx, W)) return -1;
    const orbx_grid& g = A.g;
    const uint8_t* rd = A.ref_desc + (size_t)ref_idx * 32;
    unsigned long long best = ~0ull;  // dist << 44 | (inverted scan position: cell << 12 | item)
    int best_id = -1;
    for (int cbase = 0; cbase < W.ncells; cbase += 64) {
        const int c = cbase + lane;
        if (c >= W.ncells) continue;
        const int cxi = W.left + c / W.ncy, cyi = W.up + c % W.ncy;
        const int cell = cxi * g.grid_rows + cyi;
        const int i0 = A.cell_start[cell], i1 = A.cell_start[cell + 1];
        for (int i = i0; i < i1; i++) {
            const int idx = A.cell_items[i];
            if (s_matched[idx] > 0 || !in_window(A.kps[idx], W, cxi, cyi)) continue;
            const unsigned long long d = (unsigned)hamming32(rd, A.desc + (size_t)idx * 32);
            const unsigned long long pos = ((unsigned long long)c << 12) | (unsigned)min(i - i0, 4095);
            const unsigned long long key = (d << 44) | ((1ull << 44) - 1 - pos);
            if (key < best) { best = key; best_id = idx; }
        }
    }
    unsigned long long w = best;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(w, o);
        w = other < w ? other : w;
    }
    if (w == ~0ull) return -1;
    const unsigned long long bal = __ballot(best == w);
    const int src = __ffsll((long long)bal) - 1;
    const int id = __shfl(best_id, src);
    return (long long)(((w >> 44) << 32) | (unsigned)id);
}

// K10c: the batched frame-to-frame resolve in its lean form: window lists only (K10a), all per-pair state in LDS
// (matched flags, distances, histogram pushes), no global store inside the sequential loop, DPP wave minimum.
// A pair that meets a window longer than kWinK is flagged and redone by K10b.
__global__ __launch_bounds__(64) void k_resolve_lists(ProjArgs A) {
    extern __shared__ int32_t s_mem[];
    __shared__ int32_t s_hist[30];
    const int lane = threadIdx.x;
    const int pair = blockIdx.x;
    const int cap = (int)A.kp_stride;
    int32_t* s_matched = s_mem;            // [cap]
    int32_t* s_dists = s_mem + cap;        // [cap]
    int32_t* s_push = s_mem + 2 * cap;     // [2 * cap]
    resolve_views(A, pair);
    for (int i = lane; i < cap; i += 64) {
        s_matched[i] = -1;
        s_dists[i] = -1;
    }
    if (lane < 30) s_hist[lane] = 0;
    __syncthreads();
    int n_push = 0;
    constexpr int kPF = 8;
    int pf_n[kPF];
    int2 pf_e[kPF];
#pragma unroll
    for (int k = 0; k < kPF; k++) {
        pf_n[k] = -1;
        pf_e[k] = make_int2(-1, 0);
        if (k < A.n_ref) {
            pf_n[k] = A.win_n[k];
            pf_e[k] = A.win_list[(size_t)k * kWinK + lane];
        }
    }
    bool slow = false;
    for (int ref_base = 0; ref_base < A.n_ref && !slow; ref_base += kPF) {
        int blk_n[kPF];
        int2 blk_e[kPF];
#pragma unroll
        for (int k = 0; k < kPF; k++) {
            blk_n[k] = pf_n[k];
            blk_e[k] = pf_e[k];
        }
#pragma unroll
        for (int k = 0; k < kPF; k++) {
            const int nx = ref_base + kPF + k;
            if (nx < A.n_ref) {
                pf_n[k] = A.win_n[nx];
                pf_e[k] = A.win_list[(size_t)nx * kWinK + lane];
            }
        }
#pragma unroll
        for (int k = 0; k < kPF; k++) {
            const int ref_idx = ref_base + k;
            const int cur_n = blk_n[k];
            if (ref_idx >= A.n_ref || cur_n <= 0) continue;
            int min_d, id;
            if (cur_n > kWinK) {
                // window longer than the pre-gathered list (dense clusters): scan the grid for this point here
                const long long r = best_in_window_scan(A, ref_idx, s_matched, lane);
                if (r < 0) continue;
                min_d = (int)(r >> 32);
                id = (int)(r & 0xffffffff);
            } else {
                const int2 e = blk_e[k];
                const bool valid = lane < cur_n && !(s_matched[e.x] > 0);
                const unsigned key = wave_min_u32(valid ? (((unsigned)e.y << 8) | (unsigned)(63 - lane)) : ~0u);
                if (key == ~0u) continue;
                min_d = (int)(key >> 8);
                id = __builtin_amdgcn_readlane(e.x, 63 - (int)(key & 0xffu));
            }
            if (min_d > 100) continue;
            if (lane == 0) {
                s_matched[id] = ref_idx;
                s_dists[id] = min_d;
                s_push[2 * n_push] = ref_idx;
                s_push[2 * n_push + 1] = id;
                n_push++;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    (void)slow;
    __syncthreads();
    if (A.check_orientation) {
        n_push = __builtin_amdgcn_readlane(n_push, 0);
        const float factor = 1.0f / 30;
        for (int i = lane; i < n_push; i += 64) {
            float rot = A.ref_kps[s_push[2 * i]].angle - A.kps[s_push[2 * i + 1]].angle;
            if (rot < 0.0f) rot += 360.0f;
            int bin = (int)roundf(rot * factor);
            if (bin == 30) bin = 0;
            if (bin < 0 || bin >= 30) bin = -1;
            else atomicAdd(&s_hist[bin], 1);
            s_push[2 * i] = bin;
        }
        __syncthreads();
        int ind1 = -1, ind2 = -1, ind3 = -1;
        int max1 = 0, max2 = 0, max3 = 0;
        for (int i = 0; i < 30; i++) {
            const int s = s_hist[i];
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
        for (int i = lane; i < n_push; i += 64) {
            const int bin = s_push[2 * i];
            if (bin >= 0 && bin != ind1 && bin != ind2 && bin != ind3) s_matched[s_push[2 * i + 1]] = 0;
        }
        __syncthreads();
    }
    int cnt = 0;
    for (int i = lane; i < cap; i += 64) {
        const int m = i < A.n ? s_matched[i] : -1;
        A.matched[i] = m;
        A.dists[i] = s_dists[i];
        cnt += m > 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (lane == 0 && A.n_matches) *A.n_matches = cnt;
}

// ---- batched frame-to-frame matcher: records -> sorted window lists -> speculative resolve -------------------------

// One feature of a frame in the order of the CSR bucket grid, with everything the window test reads.
constexpr int kSortK = 64;   // entries per reference point in the sorted window lists (candidates within the bound)

struct ItemRec {
    float x, y;
    uint32_t idx_oct;   // feature index | (octave saturated to int8) << 24
    uint16_t cx, cy;    // grid cell
    uint32_t desc[8];   // the feature's descriptor: one record fetch gives the test inputs and the distance inputs
};
static_assert(sizeof(ItemRec) == 48, "ItemRec is read as three 16-byte words");

// K10r: CSR-ordered feature records of the frame of every pair (one thread per grid cell).
__global__ __launch_bounds__(256) void k_item_records(ProjArgs A, ItemRec* __restrict__ recs) {
    const int pair = blockIdx.y;
    const int cell = blockIdx.x * 256 + threadIdx.x;
    const int ncells = A.g.grid_cols * A.g.grid_rows;
    if (cell >= ncells) return;
    const int fi = A.frame_index ? A.frame_index[pair] : 0;
    const int32_t* cs = A.cell_start + (size_t)fi * A.cell_stride;
    const int32_t* ci = A.cell_items + (size_t)fi * A.kp_stride;
    const orbx_keypoint* kps = A.kps + (size_t)fi * A.kp_stride;
    const uint8_t* desc = A.desc + (size_t)fi * A.kp_stride * 32;
    ItemRec* out = recs + (size_t)pair * A.kp_stride;
    const int i0 = cs[cell], i1 = cs[cell + 1];
    for (int i = i0; i < i1; i++) {
        const int idx = ci[i];
        const orbx_keypoint k = kps[idx];
        const uint4* dp = (const uint4*)(desc + (size_t)idx * 32);
        const uint4 d0 = dp[0], d1 = dp[1];
        uint4* o = (uint4*)(out + i);
        o[0] = make_uint4(__float_as_uint(k.x), __float_as_uint(k.y),
                          (uint32_t)idx | ((uint32_t)(min(max(k.octave, -128), 127) & 0xff) << 24),
                          (uint32_t)(cell / A.g.grid_rows) | ((uint32_t)(cell % A.g.grid_rows) << 16));
        o[1] = d0;
        o[2] = d1;
    }
}

__device__ __forceinline__ bool in_window_rec(const ItemRec& r, const Window& W) {
    const int oct = (int)(int8_t)(r.idx_oct >> 24);
    if (W.min_only) {
        if (oct < W.min_level) return false;
        return fabsf(r.x - (float)r.cx) < W.radius_scaled && fabsf(r.y - (float)r.cy) < W.radius_scaled;
    }
    if (oct < W.min_level || oct > W.max_level) return false;
    return fabsf(r.x - W.px) < W.radius_scaled && fabsf(r.y - W.py) < W.radius_scaled;
}

// Inclusive prefix sum over the 16 lanes of a DPP row.
__device__ __forceinline__ int row16_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);  // row_shr:8
    return v;
}

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// K10s: window gather for the batched frame-to-frame matcher, 16 lanes per reference point.  The cells of one grid
// column of a window are adjacent in the CSR order, so a window is a handful of contiguous record runs; the lanes
// take the records of all runs side by side, test them, and append the accepted ones (with their Hamming distance)
// in scan order.  The list is then sorted by (distance, LAST scan position first) - the order in which the
// reference's `<=` minimum search (orb_matcher.cpp:222) would prefer them - so the resolve step only ever needs
// the first entry that is still free.
// win_n[ref] = -1: point skipped; > kSortK: list too long, the resolve step scans that window itself.
__global__ __launch_bounds__(256) void k_gather_sorted(ProjArgs A, const ItemRec* __restrict__ recs_all) {
    __shared__ int2 s_ent[16][kSortK + 1];  // + 1: the groups' lists start in different banks
    __shared__ int s_incl[16][16];
    __shared__ int s_first[16][16];
    __shared__ float s_sf[64];
    const int gl = threadIdx.x & 15;
    const int grp = threadIdx.x >> 4;
    const int sub = (threadIdx.x >> 4) & 3;  // group inside the wavefront
    // the per-level radius factors sit in LDS: one dependent memory round trip less per reference point.  Their load
    // is in flight while the pair's views are resolved, and a block past the pair's last reference point (the grid is
    // sized for the row capacity) leaves before the barrier
    float sf = 0.f;
    if ((int)threadIdx.x < min(A.n_levels, 64)) sf = A.scale_factors[threadIdx.x];
    const ItemRec* recs = recs_all + (size_t)blockIdx.y * A.kp_stride;
    resolve_views(A, blockIdx.y);
    if ((int)blockIdx.x * 16 >= A.n_ref) return;
    if (threadIdx.x < 64) s_sf[threadIdx.x] = sf;
    __syncthreads();
    A.scale_factors = s_sf;
    const int ref_idx = blockIdx.x * 16 + grp;
    if (ref_idx >= A.n_ref) return;
    const orbx_grid& g = A.g;
    Window W;
    if (!window_of<0>(A, ref_idx, W)) {
        if (gl == 0) A.win_n[ref_idx] = -1;
        return;
    }
    const uint4* rdp = (const uint4*)(A.ref_desc + (size_t)ref_idx * 32);
    const uint4 ra = rdp[0], rb = rdp[1];
    const int ncols = W.ncx;
    int list_n = 0;
    for (int cb = 0; cb < ncols && list_n <= kSortK; cb += 16) {
        int start = 0, len = 0;
        if (cb + gl < ncols) {
            const int c0 = (W.left + cb + gl) * g.grid_rows + W.up;
            start = A.cell_start[c0];
            len = A.cell_start[c0 + W.ncy] - start;
        }
        const int incl = row16_scan(len);
        const int total = __shfl(incl, 15, 16);
        s_incl[grp][gl] = incl;
        s_first[grp][gl] = start - (incl - len);  // record of item t of this run = first + t
        wave_lds_fence();
        // the first four runs of the chunk (a window is rarely wider) reach every lane of the group through DPP row
        // broadcasts; only a record beyond them looks its run up in LDS
        const int first = start - (incl - len);
        const int i0 = __builtin_amdgcn_update_dpp(0, incl, 0x150, 0xF, 0xF, false);
        const int i1 = __builtin_amdgcn_update_dpp(0, incl, 0x151, 0xF, 0xF, false);
        const int i2 = __builtin_amdgcn_update_dpp(0, incl, 0x152, 0xF, 0xF, false);
        const int i3 = __builtin_amdgcn_update_dpp(0, incl, 0x153, 0xF, 0xF, false);
        const int f0 = __builtin_amdgcn_update_dpp(0, first, 0x150, 0xF, 0xF, false);
        const int f1 = __builtin_amdgcn_update_dpp(0, first, 0x151, 0xF, 0xF, false);
        const int f2 = __builtin_amdgcn_update_dpp(0, first, 0x152, 0xF, 0xF, false);
        const int f3 = __builtin_amdgcn_update_dpp(0, first, 0x153, 0xF, 0xF, false);
        // two records per lane and step: both fetches are in flight together
        for (int tb = 0; tb < total && list_n <= kSortK; tb += 32) {
            uint4 w0[2], w1[2], w2[2];
            bool in[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int t = tb + 16 * u + gl;
                in[u] = t < total;
                w0[u] = w1[u] = w2[u] = make_uint4(0u, 0u, 0u, 0u);
                if (in[u]) {
                    int fr = t < i0 ? f0 : t < i1 ? f1 : t < i2 ? f2 : f3;
                    if (t >= i3) {
                        int run = 4;
                        while (t >= s_incl[grp][run]) run++;
                        fr = s_first[grp][run];
                    }
                    const uint4* rp = (const uint4*)(recs + fr + t);
                    w0[u] = rp[0];
                    w1[u] = rp[1];
                    w2[u] = rp[2];
                }
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                ItemRec r;
                r.x = __uint_as_float(w0[u].x);
                r.y = __uint_as_float(w0[u].y);
                r.idx_oct = w0[u].z;
                r.cx = (uint16_t)(w0[u].w & 0xffffu);
                r.cy = (uint16_t)(w0[u].w >> 16);
                // A candidate farther than the acceptance bound (orb_matcher.cpp:228, 100) can never be assigned: the
                // resolve step stops a point at its first free entry beyond the bound, which is the same as finding
                // no free entry among those within it.  Only the latter are listed (a few per window).
                bool ok = in[u] && in_window_rec(r, W);
                int d = 0;
                if (ok) {
                    d = __popc(ra.x ^ w1[u].x) + __popc(ra.y ^ w1[u].y) + __popc(ra.z ^ w1[u].z) +
                        __popc(ra.w ^ w1[u].w) + __popc(rb.x ^ w2[u].x) + __popc(rb.y ^ w2[u].y) +
                        __popc(rb.z ^ w2[u].z) + __popc(rb.w ^ w2[u].w);
                    ok = d <= 100;
                }
                const unsigned gm = (unsigned)(__ballot(ok) >> (16 * sub)) & 0xffffu;
                const int pos = list_n + __popc(gm & ((1u << gl) - 1u));
                if (ok && pos < kSortK) {
                    // the sort key goes in with the entry: distance << 8 | (255 - scan position), distinct per entry
                    s_ent[grp][pos] = make_int2((int)(r.idx_oct & 0xffffffu), (d << 8) | (255 - pos));
                }
                list_n += __popc(gm);
            }
        }
        wave_lds_fence();
    }
    if (gl == 0) A.win_n[ref_idx] = list_n;
    if (list_n > kSortK) return;
    wave_lds_fence();
    // rank sort: key = distance << 8 | (255 - scan position); keys are distinct
    int2* list = A.win_list + (size_t)ref_idx * kSortK;
    for (int i = gl; i < list_n; i += 16) {
        const int2 e = s_ent[grp][i];
        int rank = 0;
        for (int j = 0; j < list_n; j++) rank += s_ent[grp][j].y < e.y;
        list[rank] = make_int2(e.x, e.y >> 8);
    }
}

// K10t: resolve of the batched frame-to-frame matcher, one wavefront per pair, 64 reference points at a time.
// The reference visits its points in order and gives each the best feature not taken by an earlier point
// (orb_matcher.cpp:216-236).  Here every lane holds one point and proposes the first free entry of its sorted list;
// the lanes below the first lane whose proposal collides with a lower lane's have pairwise distinct proposals made
// from the same state, which is exactly what the sequential walk gives them, so they commit together; the others
// propose again.  A point without a free entry within the distance bound stays without a match whatever the
// earlier points do (taken features only accumulate).  Points with an over-long window split the block: they are
// resolved on their own, by a scan of the grid with all 64 lanes, when every point before them is done.
constexpr int kHead = 8;  // list entries per point held in registers; deeper ones are read from memory
__global__ __launch_bounds__(64) void k_resolve_sorted(ProjArgs A) {
    extern __shared__ int32_t s_mem[];
    __shared__ int32_t s_hist[30];
    const int lane = threadIdx.x;
    const int pair = blockIdx.x;
    const int cap = (int)A.kp_stride;
    int32_t* s_matched = s_mem;            // [cap]
    int32_t* s_dists = s_mem + cap;        // [cap]
    int32_t* s_claim = s_mem + 2 * cap;    // [cap]
    int32_t* s_push = s_mem + 3 * cap;     // [2 * cap]
    resolve_views(A, pair);
    for (int i = lane; i < cap; i += 64) {
        s_matched[i] = -1;
        s_dists[i] = -1;
        s_claim[i] = 0x7fffffff;
    }
    if (lane < 30) s_hist[lane] = 0;
    __syncthreads();
    int n_push = 0;
    // head of this lane's list for the next block, fetched one block ahead
    auto list_of = [&](int ref) { return A.win_list + (size_t)min(ref, max(A.n_ref - 1, 0)) * kSortK; };
    uint4 pf[kHead / 2];
    int pf_n = -1;
    if (A.n_ref > 0) {
        const uint4* lp = (const uint4*)list_of(lane);
#pragma unroll
        for (int j = 0; j < kHead / 2; j++) pf[j] = lp[j];
        if (lane < A.n_ref) pf_n = A.win_n[lane];
    }
    for (int base = 0; base < A.n_ref; base += 64) {
        const int n = pf_n;
        int ex[kHead], ed[kHead];
#pragma unroll
        for (int j = 0; j < kHead / 2; j++) {
            ex[2 * j] = (int)pf[j].x;
            ed[2 * j] = (int)pf[j].y;
            ex[2 * j + 1] = (int)pf[j].z;
            ed[2 * j + 1] = (int)pf[j].w;
        }
#pragma unroll
        for (int k = 0; k < kHead; k++)
            if (k >= n) ex[k] = 0;  // entries past the end are never chosen; keep their flag lookups in range
        if (base + 64 < A.n_ref) {
            const uint4* lp = (const uint4*)list_of(base + 64 + lane);
#pragma unroll
            for (int j = 0; j < kHead / 2; j++) pf[j] = lp[j];
            pf_n = base + 64 + lane < A.n_ref ? A.win_n[base + 64 + lane] : -1;
        }
        const int ref_idx = base + lane;
        const int2* my_list = list_of(ref_idx);
        int cursor = kHead;  // position in the part of the list that is read from memory
        bool done = !(n > 0);
        unsigned long long ovm = __ballot(n > kSortK);
        while (true) {
            const int limit = ovm ? __ffsll((long long)ovm) - 1 : 64;
            while (true) {
                bool prop = false;
                int f = 0, d = 0;
                if (lane < limit && !done) {
                    int fl[kHead];
#pragma unroll
                    for (int k = 0; k < kHead; k++) fl[k] = s_matched[ex[k]];
                    bool found = false;
#pragma unroll
                    for (int k = kHead - 1; k >= 0; k--)
                        if (k < n && !(fl[k] > 0)) { found = true; f = ex[k]; d = ed[k]; }
                    if (!found && n > kHead) {
                        while (cursor < n) {
                            const int2 e = my_list[cursor];
                            if (!(s_matched[e.x] > 0)) { found = true; f = e.x; d = e.y; break; }
                            cursor++;
                        }
                    }
                    if (!found || d > 100) done = true;
                    else prop = true;
                }
                if (__ballot(prop) == 0ull) break;
                if (prop) atomicMin(&s_claim[f], lane);
                wave_lds_fence();
                const bool conflict = prop && s_claim[f] != lane;
                const unsigned long long cm = __ballot(conflict);
                const int rstar = cm ? __ffsll((long long)cm) - 1 : 64;
                const bool commit = prop && lane < rstar;
                const unsigned long long km = __ballot(commit);
                wave_lds_fence();
                if (prop) s_claim[f] = 0x7fffffff;
                if (commit) {
                    const int pos = n_push + __popcll(km & ((1ull << lane) - 1ull));
                    s_matched[f] = ref_idx;
                    s_dists[f] = d;
                    s_push[2 * pos] = ref_idx;
                    s_push[2 * pos + 1] = f;
                    done = true;
                }
                n_push += __popcll(km);
                wave_lds_fence();
            }
            if (limit == 64) break;
            const long long r = best_in_window_scan(A, base + limit, s_matched, lane);
            if (r >= 0 && (int)(r >> 32) <= 100) {
                const int id = (int)(r & 0xffffffff);
                if (lane == 0) {
                    s_matched[id] = base + limit;
                    s_dists[id] = (int)(r >> 32);
                    s_push[2 * n_push] = base + limit;
                    s_push[2 * n_push + 1] = id;
                }
                n_push++;
            }
            if (lane == limit) done = true;
            ovm &= ovm - 1ull;
            wave_lds_fence();
        }
    }
    __syncthreads();
    if (A.check_orientation) {
        const float factor = 1.0f / 30;
        // four pushes per lane and step: the eight angle loads of a step are in flight together
        for (int i0 = 0; i0 < n_push; i0 += 256) {
            float ra[4], fa[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = i0 + 64 * u + lane;
                ra[u] = fa[u] = 0.f;
                if (i < n_push) {
                    ra[u] = A.ref_kps[s_push[2 * i]].angle;
                    fa[u] = A.kps[s_push[2 * i + 1]].angle;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = i0 + 64 * u + lane;
                if (i >= n_push) continue;
                float rot = ra[u] - fa[u];
                if (rot < 0.0f) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == 30) bin = 0;
                if (bin < 0 || bin >= 30) bin = -1;
                else atomicAdd(&s_hist[bin], 1);
                s_push[2 * i] = bin;
            }
        }
        __syncthreads();
        int ind1 = -1, ind2 = -1, ind3 = -1;
        int max1 = 0, max2 = 0, max3 = 0;
        for (int i = 0; i < 30; i++) {
            const int s = s_hist[i];
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
        for (int i = lane; i < n_push; i += 64) {
            const int bin = s_push[2 * i];
            if (bin >= 0 && bin != ind1 && bin != ind2 && bin != ind3) s_matched[s_push[2 * i + 1]] = 0;
        }
        __syncthreads();
    }
    int cnt = 0;
    for (int i = lane; i < cap; i += 64) {
        const int m = i < A.n ? s_matched[i] : -1;
        A.matched[i] = m;
        A.dists[i] = s_dists[i];
        cnt += m > 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (lane == 0 && A.n_matches) *A.n_matches = cnt;
}

// ---- BoW matcher (BowOrbMatcher::match, orb_matcher.cpp:386-497) ---------------------------------------------------
// The host intersects the two feature vectors (common nodes in ascending order) into a job list: one job per
// reference feature of a common node, with the span of the frame's features in that node.  One wavefront walks the
// jobs in order (the "already matched" rule makes them sequential); per job the lanes compute the Hamming distances
// of the still-free candidates in parallel and compact them in list order, then the best / second-best walk of the
// reference runs over that list with its own tie rules (`<=` for the best, which does not demote the old best).
struct BowArgs {
    const orbx_keypoint* kps;
    int n;
    const uint8_t* desc;
    const orbx_keypoint* ref_kps;
    const uint8_t* ref_desc;
    const uint8_t* ref_flag;
    const int32_t* jobs;   // [n_jobs][3]: reference feature, first candidate, number of candidates
    int n_jobs;
    const int32_t* items;  // candidate feature indices (the frame's feature vector, node after node)
    int low_threshold;
    float nn_ratio;
    int check_orientation;
    int32_t* matched;      // out [n]
    int32_t* dists;        // out [n]
    int32_t* pushes;       // scratch [2 * n_jobs]
};

__global__ __launch_bounds__(64) void k_bow_match(const BowArgs* __restrict__ args) {
    const BowArgs A = args[blockIdx.x];  // one frame pair per workgroup
    extern __shared__ int32_t s_matched[];  // [n]
    __shared__ int32_t s_idx[64];
    __shared__ int32_t s_dist[64];
    __shared__ int32_t s_hist[30];
    const int lane = threadIdx.x;
    for (int i = lane; i < A.n; i += 64) {
        s_matched[i] = -1;
        A.dists[i] = -1;
    }
    if (lane < 30) s_hist[lane] = 0;
    __syncthreads();
    int n_push = 0;
    for (int job = 0; job < A.n_jobs; job++) {
        const int ref_idx = A.jobs[3 * job], start = A.jobs[3 * job + 1], len = A.jobs[3 * job + 2];
        if (!A.ref_flag[ref_idx]) continue;
        const uint8_t* rd = A.ref_desc + (size_t)ref_idx * 32;
        int min_d = 256, min_d2 = 256, id = -1;  // the same on every lane
        for (int cb = 0; cb < len; cb += 64) {
            const int c = cb + lane;
            int idx = -1, d = 0;
            bool free_ = false;
            if (c < len) {
                idx = A.items[start + c];
                free_ = !(s_matched[idx] > 0);
                if (free_) d = hamming32(rd, A.desc + (size_t)idx * 32);
            }
            const unsigned long long bal = __ballot(free_);
            if (free_) {
                const int pos = __popcll(bal & ((1ull << lane) - 1ull));
                s_idx[pos] = idx;
                s_dist[pos] = d;
            }
            wave_lds_fence();
            const int cnt = __popcll(bal);
            for (int i = 0; i < cnt; i++) {
                const int di = s_dist[i];
                if (di <= min_d) { min_d = di; id = s_idx[i]; }
                else if (di < min_d2) min_d2 = di;
            }
            wave_lds_fence();
        }
        if (min_d <= A.low_threshold && (float)min_d < A.nn_ratio * (float)min_d2) {
            if (lane == 0) {
                s_matched[id] = ref_idx;
                A.dists[id] = min_d;
                if (A.check_orientation) {
                    A.pushes[2 * n_push] = ref_idx;
                    A.pushes[2 * n_push + 1] = id;
                }
            }
            n_push++;
            wave_lds_fence();
        }
    }
    __syncthreads();
    if (A.check_orientation) {
        __threadfence_block();
        const float factor = 1.0f / 30;
        for (int i = lane; i < n_push; i += 64) {
            float rot = A.ref_kps[A.pushes[2 * i]].angle - A.kps[A.pushes[2 * i + 1]].angle;
            if (rot < 0.0f) rot += 360.0f;
            int bin = (int)roundf(rot * factor);
            if (bin == 30) bin = 0;
            if (bin < 0 || bin >= 30) bin = -1;
            else atomicAdd(&s_hist[bin], 1);
            A.pushes[2 * i] = bin;
        }
        __syncthreads();
        int ind1 = -1, ind2 = -1, ind3 = -1;
        int max1 = 0, max2 = 0, max3 = 0;
        for (int i = 0; i < 30; i++) {
            const int s = s_hist[i];
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
        for (int i = lane; i < n_push; i += 64) {
            const int bin = A.pushes[2 * i];
            if (bin >= 0 && bin != ind1 && bin != ind2 && bin != ind3) s_matched[A.pushes[2 * i + 1]] = 0;
        }
        __syncthreads();
    }
    for (int i = lane; i < A.n; i += 64) A.matched[i] = s_matched[i];
}

// ---- MapPoint::computeBestDescriptor (map_point.cpp:109-166), batched ---------------------------------------------
// One wavefront per map point.  For every observation i the Hamming distances to all observations of the point are
// laid out in LDS; the median the reference reads after sorting, element (int)(0.5 * (n - 1)), is the value v with
// #(d < v) <= k < #(d <= v), found by counting; the first i with the smallest median wins (`<`).
constexpr int kBestMaxObs = 1024;
__global__ __launch_bounds__(64) void k_best_descriptor(const uint8_t* __restrict__ desc,
                                                        const int32_t* __restrict__ point_start, int n_points,
                                                        int32_t* __restrict__ best) {
    __shared__ int s_d[kBestMaxObs];
    const int p = blockIdx.x, lane = threadIdx.x;
    if (p >= n_points) return;
    const int first = point_start[p], n = point_start[p + 1] - first;
    if (n <= 0) {
        if (lane == 0) best[p] = -1;
        return;
    }
    const uint8_t* d = desc + (size_t)first * 32;
    const int k = (int)(0.5 * (n - 1));
    int best_median = 0x7fffffff, best_idx = 0;
    for (int i = 0; i < n; i++) {
        for (int j = lane; j < n; j += 64) s_d[j] = j == i ? 0 : hamming32(d + (size_t)i * 32, d + (size_t)j * 32);
        wave_lds_fence();
        int median = -1;
        for (int j = lane; j < n; j += 64) {
            const int v = s_d[j];
            int less = 0, leq = 0;
            for (int t = 0; t < n; t++) {
                const int o = s_d[t];
                less += o < v;
                leq += o <= v;
            }
            if (less <= k && k < leq) median = v;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) median = max(median, __shfl_xor(median, o));
        if (median < best_median) { best_median = median; best_idx = i; }
        wave_lds_fence();
    }
    if (lane == 0) best[p] = best_idx;
}

// ---- EpipolarConstraintWithBowMatcher::match (orb_matcher.cpp:520-661) -------------------------------------------
// One wavefront walks the common vocabulary nodes in order.  As the reference is written, the descriptor distance of
// its inner loop is taken between row idx of BOTH descriptor matrices (:573), so it is a per-feature constant d0
// (computed for all features in the prologue), and the candidate kept is the LAST one of the node's reference list
// that is still free (matched[ref_idx] <= 0, an array indexed by the frame's own features, :565), has no map point and
// passes the point-to-epipolar-line test (:628-661).  Per node the lanes hold up to 64 reference candidates in
// registers and the node's features are staged in LDS, so the sequential walk over the features touches LDS only.
struct EpiArgs {
    const orbx_keypoint* kps;
    int n;
    const uint8_t* desc;
    const orbx_keypoint* ref_kps;
    int n_ref;
    const uint8_t* ref_desc;
    const uint8_t* has_mp;
    const uint8_t* ref_has_mp;
    float F[9];
    const float* scale_sigmas;
    int n_levels;
    const int32_t* jobs;       // [n_jobs][4]: first feature, number of features, first candidate, number of candidates
    int n_jobs;
    const int32_t* items;      // the frame's feature vector, node after node
    const int32_t* ref_items;  // the reference frame's
    int low_threshold;
    int check_orientation;
    int32_t* matched;          // out [n]
    int32_t* dists;            // out [n]
    int32_t* pushes;           // scratch [2 * n]
};

__global__ __launch_bounds__(64) void k_epipolar_match(const EpiArgs* __restrict__ args) {
    const EpiArgs A = args[blockIdx.x];  // one key-frame pair per workgroup
    extern __shared__ int32_t s_mem[];
    int32_t* s_matched = s_mem;                            // [n]
    unsigned short* s_d0 = (unsigned short*)(s_mem + A.n);  // [n]: d0, or 0xffff when the feature cannot match
    __shared__ float s_fx[64], s_fy[64];
    __shared__ int32_t s_fid[64];
    __shared__ int32_t s_hist[30];
    const int lane = threadIdx.x;
    for (int i = lane; i < A.n; i += 64) {
        s_matched[i] = -1;
        A.dists[i] = -1;
        unsigned short d0 = 0xffffu;
        if (!A.has_mp[i] && i < A.n_ref) {
            const int d = hamming32(A.ref_desc + (size_t)i * 32, A.desc + (size_t)i * 32);
            if (d <= A.low_threshold) d0 = (unsigned short)d;
        }
        s_d0[i] = d0;
    }
    if (lane < 30) s_hist[lane] = 0;
    __syncthreads();
    int n_push = 0;
    for (int job = 0; job < A.n_jobs; job++) {
        const int f_start = A.jobs[4 * job], f_len = A.jobs[4 * job + 1];
        const int rf_start = A.jobs[4 * job + 2], rf_len = A.jobs[4 * job + 3];
        if (f_len <= 0 || rf_len <= 0) continue;
        const int n_rchunks = (rf_len + 63) >> 6;
        int r = 0;
        float rx = 0.f, ry = 0.f;
        double thr = 0.0;
        bool rok = false;
        auto load_chunk = [&](int rc) {
            const int c = rc * 64 + lane;
            rok = false;
            r = 0;
            if (c < rf_len) {
                r = A.ref_items[rf_start + c];
                const orbx_keypoint k = A.ref_kps[r];
                rx = k.x;
                ry = k.y;
                rok = !A.ref_has_mp[r] && k.octave >= 0 && k.octave < A.n_levels;
                if (rok) thr = 3.84 * A.scale_sigmas[k.octave];
            }
        };
        if (n_rchunks == 1) load_chunk(0);
        for (int fb = 0; fb < f_len; fb += 64) {
            wave_lds_fence();
            if (fb + lane < f_len) {
                const int idx = A.items[f_start + fb + lane];
                s_fid[lane] = idx;
                s_fx[lane] = A.kps[idx].x;
                s_fy[lane] = A.kps[idx].y;
            }
            wave_lds_fence();
            const int cnt = min(64, f_len - fb);
            for (int j = 0; j < cnt; j++) {
                const int idx = s_fid[j];
                const int d0 = s_d0[idx];
                if (d0 == 0xffff) continue;
                const float fx = s_fx[j], fy = s_fy[j];
                const float a = fx * A.F[0] + fy * A.F[3] + A.F[6];
                const float b = fx * A.F[1] + fy * A.F[4] + A.F[7];
                const float c = fx * A.F[2] + fy * A.F[5] + A.F[8];
                const float den = a * a + b * b;
                int found = -1;
                for (int rc = n_rchunks - 1; rc >= 0; rc--) {
                    if (n_rchunks > 1) load_chunk(rc);
                    bool pass = false;
                    if (rok && !(s_matched[r] > 0) && den != 0) {
                        const float num = a * rx + b * ry + c;
                        const float dsqr = num * num / den;
                        pass = (double)dsqr < thr;
                    }
                    const unsigned long long bal = __ballot(pass);
                    if (bal) {
                        found = __shfl(r, 63 - __clzll((long long)bal));
                        break;
                    }
                }
                if (found >= 0) {
                    if (lane == 0) {
                        s_matched[idx] = found;
                        A.dists[idx] = d0;
                        if (A.check_orientation) {
                            A.pushes[2 * n_push] = idx;
                            A.pushes[2 * n_push + 1] = found;
                        }
                    }
                    n_push++;
                    wave_lds_fence();
                }
            }
        }
    }
    __syncthreads();
    if (A.check_orientation) {
        __threadfence_block();
        const float factor = 1.0f / 30;
        for (int i = lane; i < n_push; i += 64) {
            float rot = A.kps[A.pushes[2 * i]].angle - A.ref_kps[A.pushes[2 * i + 1]].angle;  // :595-597
            if (rot < 0.0f) rot += 360.0f;
            int bin = (int)roundf(rot * factor);
            if (bin == 30) bin = 0;
            if (bin < 0 || bin >= 30) bin = -1;
            else atomicAdd(&s_hist[bin], 1);
            A.pushes[2 * i + 1] = bin;
        }
        __syncthreads();
        int ind1 = -1, ind2 = -1, ind3 = -1;
        int max1 = 0, max2 = 0, max3 = 0;
        for (int i = 0; i < 30; i++) {
            const int s = s_hist[i];
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
        for (int i = lane; i < n_push; i += 64) {
            const int bin = A.pushes[2 * i + 1];
            if (bin >= 0 && bin != ind1 && bin != ind2 && bin != ind3) s_matched[A.pushes[2 * i]] = 0;
        }
        __syncthreads();
    }
    for (int i = lane; i < A.n; i += 64) A.matched[i] = s_matched[i];
}

// ---- candidate search of ORBMatcher::fuse (orb_matcher.cpp:806-867) ----------------------------------------------
// `matched` is never written inside fuse, so the map points are independent: 16 lanes per point scan the search
// window (radius_multiplier * scale factor of the predicted level, levels [level-1, level]) in the reference's cell
// order, drop features whose squared reprojection error fails e2 * inv_sigma2[octave] <= 5.99 and keep the smallest
// descriptor distance, first in scan order on ties (`<`, :862).
struct FuseArgs {
    orbx_grid g;
    const int32_t* cell_start;
    const int32_t* cell_items;
    const orbx_keypoint* kps;
    const uint8_t* desc;
    int n_pts;
    const uint8_t* in_view;
    const float* proj_xy;
    const int32_t* pred_level;
    const uint8_t* pt_desc;
    const float* scale_factors;
    const float* inv_scale_sigmas;
    int n_levels;
    float radius_multiplier;
    int32_t* best_idx;
    int32_t* best_dist;
};

__global__ __launch_bounds__(256) void k_fuse_search(FuseArgs A) {
    const int lane = threadIdx.x & 15;
    const int p = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (p >= A.n_pts) return;
    const orbx_grid& g = A.g;
    unsigned long long best = ~0ull;  // distance << 40 | scan position (cell << 14 | item)
    int best_id = -1;
    const int level = A.pred_level[p];
    if (A.in_view[p] && level >= 0 && level < A.n_levels) {
        Window W;
        W.px = A.proj_xy[2 * p];
        W.py = A.proj_xy[2 * p + 1];
        W.min_only = false;
        W.radius_scaled = A.radius_multiplier * A.scale_factors[level];
        W.min_level = level - 1;
        W.max_level = level;
        W.left = max(0, (int)floorf(W.px - g.min_x - W.radius_scaled) / g.grid_size_x);
        const int right = min(g.grid_cols - 1, (int)ceilf(W.px - g.min_x + W.radius_scaled) / g.grid_size_x);
        W.up = max(0, (int)floorf(W.py - g.min_y - W.radius_scaled) / g.grid_size_y);
        const int down = min(g.grid_rows - 1, (int)ceilf(W.py - g.min_y + W.radius_scaled) / g.grid_size_y);
        if (W.left < g.grid_cols && right >= 0 && W.up < g.grid_rows && down >= 0) {
            W.ncy = down - W.up + 1;
            W.ncells = (right - W.left + 1) * W.ncy;
            const uint8_t* pd = A.pt_desc + (size_t)p * 32;
            for (int c = lane; c < W.ncells; c += 16) {
                const int cxi = W.left + c / W.ncy, cyi = W.up + c % W.ncy;
                const int cell = cxi * g.grid_rows + cyi;
                const int i0 = A.cell_start[cell], i1 = A.cell_start[cell + 1];
                for (int i = i0; i < i1; i++) {
                    const int idx = A.cell_items[i];
                    const orbx_keypoint k = A.kps[idx];
                    if (k.octave < 0 || !in_window(k, W, cxi, cyi)) continue;
                    const float ex = W.px - k.x, ey = W.py - k.y;
                    const float e_sqrd = ex * ex + ey * ey;
                    if ((double)(e_sqrd * A.inv_scale_sigmas[k.octave]) > 5.99) continue;
                    const unsigned long long d = (unsigned)hamming32(pd, A.desc + (size_t)idx * 32);
                    const unsigned long long key = (d << 40) | ((unsigned long long)c << 14) | (unsigned)min(i - i0, 16383);
                    if (key < best) { best = key; best_id = idx; }
                }
            }
        }
    }
    unsigned long long w = best;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(w, o, 16);
        w = other < w ? other : w;
    }
    // keys are unique per feature (scan positions differ), so exactly one lane owns the minimum
    if (w != ~0ull && best == w) {
        A.best_idx[p] = best_id;
        A.best_dist[p] = (int)(w >> 40);
    } else if (w == ~0ull && lane == 0) {
        A.best_idx[p] = -1;
        A.best_dist[p] = 256;
    }
}

}  // namespace

struct orbx_matcher {
    CallGate gate;  // one call at a time (ORBX_ENTER)
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    DevBuf<uint8_t> d_q, d_t, d_flags;
    DevBuf<unsigned long long> d_best;
    DevBuf<int32_t> d_idx, d_dist, d_ints, d_cells, d_items, d_pushes;
    DevBuf<float> d_f0, d_f1, d_f2;
    DevBuf<orbx_keypoint> d_kp, d_refkp;
    DevBuf<int32_t> d_win_n, d_win_over;
    DevBuf<int2> d_win_list;
    DevBuf<uint8_t> d_recs, d_args;
    size_t bow_lds_attr = 0, epi_lds_attr = 0;   // largest dynamic LDS the back-end matchers of this handle have asked for
    std::vector<unsigned long long> h_best;
    int hamming_kernel = 0;   // 0: matrix cores (k_bf_hamming_mfma), 1: vector pipes (k_bf_hamming)
};

namespace {

int pick_chunk(int n_train, int q_blocks, int pairs) {
    // enough workgroups to cover the chip: split the train set when queries x pairs alone cannot
    int want = 1024;
    int have = std::max(1, q_blocks * pairs);
    int chunks = std::min(std::max(1, want / have), std::max(1, n_train / 256));
    int chunk = div_up(std::max(n_train, 1), chunks);
    // at most 2^16 rows per chunk: k_bf_hamming keeps the row within the chunk in 16 bits of its key
    return std::min(div_up(chunk, 256) * 256, 65536);
}

// Pairs of descriptor sets through k_bf_hamming: pair p = (query set p, train set p), or with t_sets > 0 every query
// set against every train set (p = query set * t_sets + train set).  Outputs per pair, out_stride values apart.
orbx_status hamming_sets(orbx_matcher* h, const uint8_t* d_query, const int32_t* d_n_query, size_t query_stride,
                         const uint8_t* d_train, const int32_t* d_n_train, size_t train_stride, int n_pairs, int t_sets,
                         int max_query, int max_train, int32_t* d_train_idx, int32_t* d_distance, size_t out_stride) {
    if (!h || !d_query || !d_train || !d_train_idx || !d_distance) return ORBX_BAD_ARG;
    if (n_pairs <= 0 || n_pairs > 65535 || max_query < 0 || max_train < 0 || out_stride < (size_t)max_query)
        return ORBX_BAD_ARG;
    // descriptor rows are read as 16-byte words (queries) and by 64-byte scalar loads (train rows)
    if ((((uintptr_t)d_query | (uintptr_t)d_train | query_stride | train_stride) & 15) != 0) return ORBX_BAD_ARG;
    if (max_query == 0) return ORBX_OK;
    ORBX_HIP(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    size_t nb = (size_t)n_pairs * max_query;
    ORBX_TRY(h->d_best.ensure(nb));
    ORBX_HIP(hipMemsetAsync(h->d_best.p, 0xff, nb * sizeof(unsigned long long), s));
    const int qb = div_up(max_query, 256);
    if (max_train > 0) {
        const bool mfma = h->hamming_kernel == 0;
        const int hb = div_up(max_query, mfma ? kMfmaBlockQ : kHamBlockQ);
        const int chunk = pick_chunk(max_train, hb, n_pairs);
        dim3 grid(hb, div_up(max_train, chunk), n_pairs);
        hipLaunchKernelGGL(mfma ? k_bf_hamming_mfma : k_bf_hamming, grid, dim3(256), 0, s, d_query, d_n_query, max_query,
                           query_stride, d_train, d_n_train, max_train, train_stride, chunk, h->d_best.p,
                           (size_t)max_query, t_sets);
    }
    hipLaunchKernelGGL(k_unpack_best, dim3(qb, n_pairs), dim3(256), 0, s, h->d_best.p, (size_t)max_query, d_n_query,
                       max_query, d_train_idx, d_distance, out_stride, t_sets);
    ORBX_HIP(hipGetLastError());
    return ORBX_OK;
}

}  // namespace

extern "C" {

orbx_status orbx_matcher_create(int device, orbx_matcher** out) {
    if (!out) return ORBX_BAD_ARG;
    ORBX_HIP(hipSetDevice(device));
    orbx_matcher* h = new orbx_matcher();
    h->device = device;
    if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        return ORBX_DEVICE;
    }
    h->stream = h->own_stream;
    *out = h;
    return ORBX_OK;
}

void orbx_matcher_destroy(orbx_matcher* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->own_stream) (void)hipStreamSynchronize(h->own_stream);
    h->d_q.release(); h->d_t.release(); h->d_flags.release(); h->d_best.release(); h->d_idx.release();
    h->d_dist.release(); h->d_ints.release(); h->d_cells.release(); h->d_items.release(); h->d_pushes.release();
    h->d_f0.release(); h->d_f1.release(); h->d_f2.release(); h->d_kp.release(); h->d_refkp.release();
    h->d_win_n.release(); h->d_win_over.release(); h->d_win_list.release(); h->d_recs.release();
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

orbx_status orbx_matcher_set_stream(orbx_matcher* h, void* hip_stream) {
    ORBX_ENTER(h);
    if (!h) return ORBX_BAD_ARG;
    h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    return ORBX_OK;
}

orbx_status orbx_matcher_set_hamming_kernel(orbx_matcher* h, int kind) {
    ORBX_ENTER(h);
    if (!h || (kind != 0 && kind != 1)) return ORBX_BAD_ARG;
    h->hamming_kernel = kind;
    return ORBX_OK;
}

orbx_status orbx_match_hamming_batch(orbx_matcher* h, const uint8_t* d_query, const int32_t* d_n_query,
                                     size_t query_stride, const uint8_t* d_train, const int32_t* d_n_train,
                                     size_t train_stride, int n_pairs, int max_query, int max_train,
                                     int32_t* d_train_idx, int32_t* d_distance, size_t out_stride) {
    ORBX_ENTER(h);
    return hamming_sets(h, d_query, d_n_query, query_stride, d_train, d_n_train, train_stride, n_pairs, 0, max_query,
                        max_train, d_train_idx, d_distance, out_stride);
}

orbx_status orbx_match_hamming_cross(orbx_matcher* h, const uint8_t* d_query, const int32_t* d_n_query,
                                     size_t query_stride, int n_query_sets, const uint8_t* d_train,
                                     const int32_t* d_n_train, size_t train_stride, int n_train_sets, int max_query,
                                     int max_train, int32_t* d_train_idx, int32_t* d_distance, size_t out_stride) {
    ORBX_ENTER(h);
    if (n_query_sets <= 0 || n_train_sets <= 0 || (long long)n_query_sets * n_train_sets > 65535) return ORBX_BAD_ARG;
    return hamming_sets(h, d_query, d_n_query, query_stride, d_train, d_n_train, train_stride,
                        n_query_sets * n_train_sets, n_train_sets, max_query, max_train, d_train_idx, d_distance,
                        out_stride);
}

orbx_status orbx_match_hamming(orbx_matcher* h, const uint8_t* query, int n_query, const uint8_t* train, int n_train,
                               int32_t* train_idx, int32_t* distance) {
    ORBX_ENTER(h);
    if (!h || n_query < 0 || n_train < 0) return ORBX_BAD_ARG;
    if (n_query == 0) return ORBX_OK;
    if (!query || !train_idx || !distance || (n_train > 0 && !train)) return ORBX_BAD_ARG;
    ORBX_HIP(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    ORBX_TRY(h->d_q.ensure((size_t)n_query * 32));
    ORBX_TRY(h->d_t.ensure((size_t)std::max(n_train, 1) * 32));
    ORBX_TRY(h->d_idx.ensure(n_query));
    ORBX_TRY(h->d_dist.ensure(n_query));
    ORBX_HIP(hipMemcpyAsync(h->d_q.p, query, (size_t)n_query * 32, hipMemcpyHostToDevice, s));
    if (n_train) ORBX_HIP(hipMemcpyAsync(h->d_t.p, train, (size_t)n_train * 32, hipMemcpyHostToDevice, s));
    ORBX_TRY(orbx_match_hamming_batch(h, h->d_q.p, nullptr, 0, h->d_t.p, nullptr, 0, 1, n_query, n_train, h->d_idx.p,
                                      h->d_dist.p, n_query));
    ORBX_HIP(hipMemcpyAsync(train_idx, h->d_idx.p, sizeof(int32_t) * n_query, hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipMemcpyAsync(distance, h->d_dist.p, sizeof(int32_t) * n_query, hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipStreamSynchronize(s));
    return ORBX_OK;
}

orbx_status orbx_match_radius(orbx_matcher* h, const orbx_keypoint* kp, int n, const orbx_keypoint* ref_kp, int n_ref,
                              const uint8_t* desc, const uint8_t* ref_desc, double max_pixel_dist, orbx_match* matches,
                              int* count) {
    ORBX_ENTER(h);
    if (!h || !count || n < 0 || n_ref < 0) return ORBX_BAD_ARG;
    *count = 0;
    if (n == 0 || n_ref == 0) return ORBX_OK;
    if (!kp || !ref_kp || !desc || !ref_desc || !matches) return ORBX_BAD_ARG;
    ORBX_HIP(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    std::vector<float> xy(2 * (size_t)n), rxy(2 * (size_t)n_ref);
    for (int i = 0; i < n; i++) { xy[2 * i] = kp[i].x; xy[2 * i + 1] = kp[i].y; }
    for (int i = 0; i < n_ref; i++) { rxy[2 * i] = ref_kp[i].x; rxy[2 * i + 1] = ref_kp[i].y; }
    ORBX_TRY(h->d_q.ensure((size_t)n * 32));
    ORBX_TRY(h->d_t.ensure((size_t)n_ref * 32));
    ORBX_TRY(h->d_f0.ensure(xy.size()));
    ORBX_TRY(h->d_f1.ensure(rxy.size()));
    ORBX_TRY(h->d_best.ensure(n));
    ORBX_HIP(hipMemcpyAsync(h->d_q.p, desc, (size_t)n * 32, hipMemcpyHostToDevice, s));
    ORBX_HIP(hipMemcpyAsync(h->d_t.p, ref_desc, (size_t)n_ref * 32, hipMemcpyHostToDevice, s));
    ORBX_HIP(hipMemcpyAsync(h->d_f0.p, xy.data(), xy.size() * sizeof(float), hipMemcpyHostToDevice, s));
    ORBX_HIP(hipMemcpyAsync(h->d_f1.p, rxy.data(), rxy.size() * sizeof(float), hipMemcpyHostToDevice, s));
    ORBX_HIP(hipMemsetAsync(h->d_best.p, 0xff, (size_t)n * sizeof(unsigned long long), s));
    const int qb = div_up(n, 256);
    const int chunk = pick_chunk(n_ref, qb, 1);
    const double max_sq = max_pixel_dist * max_pixel_dist;
    hipLaunchKernelGGL(k_bf_sad_radius, dim3(qb, div_up(n_ref, chunk)), dim3(256), 0, s, h->d_q.p,
                       (const float2*)h->d_f0.p, n, h->d_t.p, (const float2*)h->d_f1.p, n_ref, chunk, max_sq,
                       h->d_best.p);
    ORBX_HIP(hipGetLastError());
    h->h_best.resize(n);
    ORBX_HIP(hipMemcpyAsync(h->h_best.data(), h->d_best.p, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipStreamSynchronize(s));
    int c = 0;
    for (int i = 0; i < n; i++) {
        const unsigned long long k = h->h_best[i];
        if (k == kNoMatch) continue;
        const double feature_dist = (double)(unsigned)(k >> 32) / 32;  // cv::sum(diff)[0] / descriptors.cols
        matches[c++] = orbx_match{i, (int32_t)(k & 0xffffffffu), -1, (float)feature_dist};
    }
    *count = c;
    return ORBX_OK;
}

static orbx_status run_projection(orbx_matcher* h, int mode, const orbx_grid* g, const int32_t* cell_start,
                                  const int32_t* cell_items, const orbx_keypoint* kps, int n, const uint8_t* desc,
                                  const orbx_keypoint* ref_kps, int n_ref, const uint8_t* flags, const float* proj_xy,
                                  const float* proj_z, const int32_t* pred_level, const float* view_cos,
                                  const uint8_t* ref_desc, const float* scale_factors, int n_levels, int level_mode,
                                  float radius, float nn_ratio, int check_orientation, orbx_match* matches,
                                  int* count) {
    if (!h || !g || !count || n < 0 || n_ref < 0) return ORBX_BAD_ARG;
    *count = 0;
    if (n == 0 || n_ref == 0) return ORBX_OK;
    if (!cell_start || !cell_items || !kps || !desc || !flags || !proj_xy || !ref_desc || !scale_factors || !matches)
        return ORBX_BAD_ARG;
    if (mode == 0 && (!ref_kps || !proj_z)) return ORBX_BAD_ARG;
    if (mode == 1 && (!pred_level || !view_cos)) return ORBX_BAD_ARG;
    if (g->grid_cols <= 0 || g->grid_rows <= 0 || g->grid_size_x <= 0 || g->grid_size_y <= 0 || n_levels <= 0)
        return ORBX_BAD_ARG;
    if (n > kMaxFeatures) return ORBX_CAPACITY;
    ORBX_HIP(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    const int ncell = g->grid_cols * g->grid_rows;
    ORBX_TRY(h->d_cells.ensure(ncell + 1));
    ORBX_TRY(h->d_items.ensure(n));
    ORBX_TRY(h->d_kp.ensure(n));
    ORBX_TRY(h->d_q.ensure((size_t)n * 32));
    ORBX_TRY(h->d_t.ensure((size_t)n_ref * 32));
    ORBX_TRY(h->d_flags.ensure(n_ref));
    ORBX_TRY(h->d_f0.ensure((size_t)2 * n_ref));
    ORBX_TRY(h->d_f1.ensure(n_ref));
    ORBX_TRY(h->d_f2.ensure(n_levels));
    ORBX_TRY(h->d_refkp.ensure(n_ref));
    ORBX_TRY(h->d_ints.ensure(n_ref + 1));
    ORBX_TRY(h->d_idx.ensure(n));
    ORBX_TRY(h->d_dist.ensure(n));
    ORBX_TRY(h->d_pushes.ensure((size_t)2 * n_ref + 2));
    ORBX_HIP(hipMemcpyAsync(h->d_cells.p, cell_start, sizeof(int32_t) * (ncell + 1), hipMemcpyHostToDevice, s));
    ORBX_HIP(hipMemcpyAsync(h->d_items.p, cell_items, sizeof(int32_t) * n, hipMemcpyHostToDevice, s));
    ORBX_HIP(hipMemcpyAsync(h->d_kp.p, kps, sizeof(orbx_keypoint) * n, hipMemcpyHostToDevice, s));
    ORBX_HIP(hipMemcpyAsync(h->d_q.p, desc, (size_t)n * 32, hipMemcpyHostToDevice, s));
    ORBX_HIP(hipMemcpyAsync(h->d_t.p, ref_desc, (size_t)n_ref * 32, hipMemcpyHostToDevice, s));
    ORBX_HIP(hipMemcpyAsync(h->d_flags.p, flags, n_ref, hipMemcpyHostToDevice, s));
    ORBX_HIP(hipMemcpyAsync(h->d_f0.p, proj_xy, sizeof(float) * 2 * n_ref, hipMemcpyHostToDevice, s));
    ORBX_HIP(hipMemcpyAsync(h->d_f2.p, scale_factors, sizeof(float) * n_levels, hipMemcpyHostToDevice, s));
    if (mode == 0) {
        ORBX_HIP(hipMemcpyAsync(h->d_f1.p, proj_z, sizeof(float) * n_ref, hipMemcpyHostToDevice, s));
        ORBX_HIP(hipMemcpyAsync(h->d_refkp.p, ref_kps, sizeof(orbx_keypoint) * n_ref, hipMemcpyHostToDevice, s));
    } else {
        ORBX_HIP(hipMemcpyAsync(h->d_f1.p, view_cos, sizeof(float) * n_ref, hipMemcpyHostToDevice, s));
        ORBX_HIP(hipMemcpyAsync(h->d_ints.p, pred_level, sizeof(int32_t) * n_ref, hipMemcpyHostToDevice, s));
    }
    int32_t* d_status = h->d_ints.p + n_ref;
    ORBX_HIP(hipMemsetAsync(d_status, 0, sizeof(int32_t), s));
    // host-side validation of what the kernel indexes with: CSR offsets and items
    if (cell_start[0] != 0 || cell_start[ncell] != n) return ORBX_BAD_ARG;
    for (int c = 0; c < ncell; c++)
        if (cell_start[c + 1] < cell_start[c]) return ORBX_BAD_ARG;
    for (int i = 0; i < n; i++)
        if (cell_items[i] < 0 || cell_items[i] >= n) return ORBX_BAD_ARG;

    ProjArgs A;
    std::memset(&A, 0, sizeof(A));
    set_grid(A, g);
    A.cell_start = h->d_cells.p;
    A.cell_items = h->d_items.p;
    A.kps = h->d_kp.p;
    A.n = n;
    A.desc = h->d_q.p;
    A.ref_kps = h->d_refkp.p;
    A.n_ref = n_ref;
    A.ref_flag = h->d_flags.p;
    A.proj_xy = h->d_f0.p;
    A.proj_z = h->d_f1.p;
    A.pred_level = h->d_ints.p;
    A.view_cos = h->d_f1.p;
    A.ref_desc = h->d_t.p;
    A.scale_factors = h->d_f2.p;
    A.n_levels = n_levels;
    A.level_mode = level_mode;
    A.radius = radius;
    A.nn_ratio = nn_ratio;
    A.check_orientation = check_orientation;
    A.matched = h->d_idx.p;
    A.dists = h->d_dist.p;
    A.pushes = h->d_pushes.p;
    A.status = d_status;
    A.n_matches = nullptr;
    A.pair_stride = n_ref;  // one pair: strides only place the window lists
    ORBX_TRY(h->d_win_n.ensure(n_ref));
    ORBX_TRY(h->d_win_list.ensure((size_t)n_ref * kWinK));
    ORBX_TRY(h->d_win_over.ensure(1));
    ORBX_HIP(hipMemsetAsync(h->d_win_over.p, 0, sizeof(int32_t), s));
    A.win_n = h->d_win_n.p;
    A.win_list = h->d_win_list.p;
    A.win_stride = kWinK;
    A.win_overflow = h->d_win_over.p;
    const size_t shmem = sizeof(int32_t) * (size_t)n;
    if (mode == 0) {
        hipLaunchKernelGGL(k_window_gather<0>, dim3(div_up(n_ref, 16), 1), dim3(256), 0, s, A);
        hipLaunchKernelGGL(k_projection_match<0>, dim3(1), dim3(64), shmem, s, A);
    } else {
        hipLaunchKernelGGL(k_window_gather<1>, dim3(div_up(n_ref, 16), 1), dim3(256), 0, s, A);
        hipLaunchKernelGGL(k_projection_match<1>, dim3(1), dim3(64), shmem, s, A);
    }
    ORBX_HIP(hipGetLastError());
    std::vector<int32_t> matched(n), dists(n);
    int32_t status = 0;
    ORBX_HIP(hipMemcpyAsync(matched.data(), h->d_idx.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipMemcpyAsync(dists.data(), h->d_dist.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipMemcpyAsync(&status, d_status, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipStreamSynchronize(s));
    if (status != 0) return ORBX_CAPACITY;
    // MatcherBase::createMatches (orb_matcher.h:81-93)
    int c = 0;
    for (int i = 0; i < n; i++)
        if (matched[i] > 0) matches[c++] = orbx_match{i, matched[i], -1, (float)dists[i]};
    *count = c;
    return ORBX_OK;
}

orbx_status orbx_match_projection_frame(orbx_matcher* h, const orbx_grid* g, const int32_t* cell_start,
                                        const int32_t* cell_items, const orbx_keypoint* keypoints, int n,
                                        const uint8_t* descriptors, const orbx_keypoint* ref_keypoints, int n_ref,
                                        const uint8_t* ref_has_point, const float* proj_xy, const float* proj_z,
                                        const uint8_t* ref_descriptors, const float* scale_factors, int n_levels,
                                        int level_mode, float radius, int check_orientation, orbx_match* matches,
                                        int* count) {
    ORBX_ENTER(h);
    return run_projection(h, 0, g, cell_start, cell_items, keypoints, n, descriptors, ref_keypoints, n_ref,
                          ref_has_point, proj_xy, proj_z, nullptr, nullptr, ref_descriptors, scale_factors, n_levels,
                          level_mode, radius, 0.f, check_orientation, matches, count);
}

orbx_status orbx_match_projection_frame_batch(orbx_matcher* h, const orbx_grid* g, int n_pairs,
                                              const int32_t* d_frame_index, const int32_t* d_ref_index,
                                              const orbx_keypoint* d_keypoints, const uint8_t* d_descriptors,
                                              const int32_t* d_counts, int kp_capacity,
                                              const int32_t* d_cell_start, const int32_t* d_cell_items,
                                              const uint8_t* d_ref_has_point, const float* d_proj_xy,
                                              const float* d_proj_z, const float* scale_factors, int n_levels,
                                              int level_mode, float radius, int check_orientation,
                                              int32_t* d_matched, int32_t* d_distance, int32_t* d_n_matches) {
    ORBX_ENTER(h);
    if (!h || !g || n_pairs <= 0 || !d_frame_index || !d_ref_index || !d_keypoints || !d_descriptors || !d_counts ||
        !d_cell_start || !d_cell_items || !d_proj_xy || !scale_factors || !d_matched || !d_distance)
        return ORBX_BAD_ARG;
    if (kp_capacity <= 0 || n_levels <= 0 || n_levels > 64) return ORBX_BAD_ARG;
    if (g->grid_cols <= 0 || g->grid_rows <= 0 || g->grid_size_x <= 0 || g->grid_size_y <= 0) return ORBX_BAD_ARG;
    if (kp_capacity > kMaxFeatures) return ORBX_CAPACITY;
    ORBX_HIP(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    ORBX_TRY(h->d_f2.ensure(n_levels));
    ORBX_TRY(h->d_pushes.ensure((size_t)2 * kp_capacity * n_pairs));
    ORBX_TRY(h->d_ints.ensure(n_pairs));
    ORBX_HIP(hipMemcpyAsync(h->d_f2.p, scale_factors, sizeof(float) * n_levels, hipMemcpyHostToDevice, s));
    ORBX_HIP(hipMemsetAsync(h->d_ints.p, 0, sizeof(int32_t) * n_pairs, s));
    ProjArgs A;
    std::memset(&A, 0, sizeof(A));
    set_grid(A, g);
    A.frame_index = d_frame_index;
    A.ref_index = d_ref_index;
    A.counts = d_counts;
    A.kp_stride = kp_capacity;
    A.cell_stride = (size_t)g->grid_cols * g->grid_rows + 1;
    A.pair_stride = kp_capacity;
    A.cell_start = d_cell_start;
    A.cell_items = d_cell_items;
    A.kps = d_keypoints;
    A.desc = d_descriptors;
    A.ref_kps = d_keypoints;
    A.ref_flag = d_ref_has_point;
    A.proj_xy = d_proj_xy;
    A.proj_z = d_proj_z;
    A.ref_desc = d_descriptors;
    A.scale_factors = h->d_f2.p;
    A.n_levels = n_levels;
    A.level_mode = level_mode;
    A.radius = radius;
    A.check_orientation = check_orientation;
    A.matched = d_matched;
    A.dists = d_distance;
    A.pushes = h->d_pushes.p;
    A.status = h->d_ints.p;
    A.n_matches = d_n_matches;
    ORBX_TRY(h->d_win_n.ensure((size_t)kp_capacity * n_pairs));
    ORBX_TRY(h->d_win_list.ensure((size_t)kp_capacity * n_pairs * (kp_capacity > 6144 ? kWinK : kSortK)));
    ORBX_TRY(h->d_win_over.ensure(n_pairs));
    ORBX_HIP(hipMemsetAsync(h->d_win_over.p, 0, sizeof(int32_t) * n_pairs, s));
    A.win_n = h->d_win_n.p;
    A.win_list = h->d_win_list.p;
    A.win_overflow = h->d_win_over.p;
    A.need_slow = h->d_win_over.p;
    A.win_stride = kSortK;
    if (kp_capacity > 6144) {
        A.win_stride = kWinK;
        // the speculative resolve keeps five words per feature in LDS; beyond that use the sequential one
        hipLaunchKernelGGL(k_window_gather<0>, dim3(div_up(kp_capacity, 16), n_pairs), dim3(256), 0, s, A);
        const size_t lds = sizeof(int32_t) * 4 * (size_t)kp_capacity;
        if (lds > 48 * 1024) ORBX_TRY(ensure_dynamic_lds((const void*)k_resolve_lists, lds));
        hipLaunchKernelGGL(k_resolve_lists, dim3(n_pairs), dim3(64), lds, s, A);
        ORBX_HIP(hipGetLastError());
        return ORBX_OK;
    }
    ORBX_TRY(h->d_recs.ensure((size_t)kp_capacity * n_pairs * sizeof(ItemRec) + 64));
    ItemRec* recs = (ItemRec*)h->d_recs.p;
    const int ncells = g->grid_cols * g->grid_rows;
    if (g->grid_cols > 65535 || g->grid_rows > 65535 || kp_capacity >= (1 << 24)) return ORBX_UNSUPPORTED;
    hipLaunchKernelGGL(k_item_records, dim3(div_up(ncells, 256), n_pairs), dim3(256), 0, s, A, recs);
    hipLaunchKernelGGL(k_gather_sorted, dim3(div_up(kp_capacity, 16), n_pairs), dim3(256), 0, s, A, recs);
    const size_t fast_lds = sizeof(int32_t) * 5 * (size_t)kp_capacity;
    if (fast_lds > 48 * 1024) ORBX_TRY(ensure_dynamic_lds((const void*)k_resolve_sorted, fast_lds));
    hipLaunchKernelGGL(k_resolve_sorted, dim3(n_pairs), dim3(64), fast_lds, s, A);
    ORBX_HIP(hipGetLastError());
    return ORBX_OK;
}

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
                                               int32_t* d_n_matches) {
    ORBX_ENTER(h);
    if (!h || !g || n_pairs <= 0 || !d_frame_index || !d_keypoints || !d_descriptors || !d_counts || !d_cell_start ||
        !d_cell_items || !d_n_points || !d_in_view || !d_proj_xy || !d_pred_level || !d_view_cos ||
        !d_point_descriptors || !scale_factors || !d_matched || !d_distance)
        return ORBX_BAD_ARG;
    if (kp_capacity <= 0 || points_capacity <= 0 || n_levels <= 0 || n_levels > 64) return ORBX_BAD_ARG;
    if (g->grid_cols <= 0 || g->grid_rows <= 0 || g->grid_size_x <= 0 || g->grid_size_y <= 0) return ORBX_BAD_ARG;
    if (kp_capacity > kMaxFeatures) return ORBX_CAPACITY;
    ORBX_HIP(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    ORBX_TRY(h->d_f2.ensure(n_levels));
    ORBX_TRY(h->d_pushes.ensure((size_t)2 * points_capacity * n_pairs + 2));
    ORBX_TRY(h->d_ints.ensure(n_pairs));
    ORBX_HIP(hipMemcpyAsync(h->d_f2.p, scale_factors, sizeof(float) * n_levels, hipMemcpyHostToDevice, s));
    ORBX_HIP(hipMemsetAsync(h->d_ints.p, 0, sizeof(int32_t) * n_pairs, s));
    ProjArgs A;
    std::memset(&A, 0, sizeof(A));
    set_grid(A, g);
    A.frame_index = d_frame_index;
    A.counts = d_counts;
    A.kp_stride = kp_capacity;
    A.cell_stride = (size_t)g->grid_cols * g->grid_rows + 1;
    A.pair_stride = points_capacity;
    A.out_stride = kp_capacity;
    A.n_ref_pair = d_n_points;
    A.ref_desc_per_pair = 1;
    A.cell_start = d_cell_start;
    A.cell_items = d_cell_items;
    A.kps = d_keypoints;
    A.desc = d_descriptors;
    A.ref_flag = d_in_view;
    A.proj_xy = d_proj_xy;
    A.pred_level = d_pred_level;
    A.view_cos = d_view_cos;
    A.ref_desc = d_point_descriptors;
    A.scale_factors = h->d_f2.p;
    A.n_levels = n_levels;
    A.radius = radius;
    A.nn_ratio = nn_ratio;
    A.matched = d_matched;
    A.dists = d_distance;
    A.pushes = h->d_pushes.p;
    A.status = h->d_ints.p;
    A.n_matches = d_n_matches;
    ORBX_TRY(h->d_win_n.ensure((size_t)points_capacity * n_pairs));
    ORBX_TRY(h->d_win_list.ensure((size_t)points_capacity * n_pairs * kWinK));
    ORBX_TRY(h->d_win_over.ensure(n_pairs));
    ORBX_HIP(hipMemsetAsync(h->d_win_over.p, 0, sizeof(int32_t) * n_pairs, s));
    A.win_n = h->d_win_n.p;
    A.win_list = h->d_win_list.p;
    A.win_stride = kWinK;
    A.win_overflow = h->d_win_over.p;
    // the window lists of all points of all pairs are gathered in parallel; the greedy "already matched" rule and the
    // ratio test inside the candidate loop (orb_matcher.cpp:313-345) are sequential per pair: one wavefront each
    hipLaunchKernelGGL(k_window_gather<1>, dim3(div_up(points_capacity, 16), n_pairs), dim3(256), 0, s, A);
    const size_t lds = sizeof(int32_t) * (size_t)kp_capacity;
    hipLaunchKernelGGL(k_projection_match<1>, dim3(n_pairs), dim3(64), lds, s, A);
    ORBX_HIP(hipGetLastError());
    return ORBX_OK;
}

orbx_status orbx_match_projection_points(orbx_matcher* h, const orbx_grid* g, const int32_t* cell_start,
                                         const int32_t* cell_items, const orbx_keypoint* keypoints, int n,
                                         const uint8_t* descriptors, int n_points, const uint8_t* in_view,
                                         const float* proj_xy, const int32_t* pred_level, const float* view_cos,
                                         const uint8_t* point_descriptors, const float* scale_factors, int n_levels,
                                         float radius, float nn_ratio, orbx_match* matches, int* count) {
    ORBX_ENTER(h);
    return run_projection(h, 1, g, cell_start, cell_items, keypoints, n, descriptors, nullptr, n_points, in_view,
                          proj_xy, nullptr, pred_level, view_cos, point_descriptors, scale_factors, n_levels, 0,
                          radius, nn_ratio, 0, matches, count);
}

// ---- back-end matchers over vocabulary nodes: many frame pairs per call -----------------------------------------
}  // extern "C"

namespace {

bool valid_feature_vector(const uint32_t* ids, const int32_t* start, const int32_t* items, int nn, int limit) {
    if (nn == 0) return true;
    if (!ids || !start || !items || start[0] != 0) return false;
    for (int i = 0; i < nn; i++) {
        if (start[i + 1] < start[i]) return false;
        if (i > 0 && !(ids[i - 1] < ids[i])) return false;
    }
    for (int i = 0; i < start[nn]; i++)
        if (items[i] < 0 || items[i] >= limit) return false;
    return true;
}

// Host-side concatenation of the per-pair arrays, uploaded with one copy per array.
struct NodePack {
    std::vector<orbx_keypoint> kp, refkp;
    std::vector<uint8_t> desc, refdesc, flags;
    std::vector<int32_t> items, jobs;
    struct Off { size_t kp, refkp, flags, items, ref_items, jobs; int n_jobs; };
    std::vector<Off> off;
};

template <typename T>
void append(std::vector<T>& v, const T* p, size_t n) { if (n) v.insert(v.end(), p, p + n); }

orbx_status upload_pack(orbx_matcher* h, const NodePack& P, size_t total_n, hipStream_t s) {
    ORBX_TRY(h->d_kp.ensure(std::max<size_t>(P.kp.size(), 1)));
    ORBX_TRY(h->d_refkp.ensure(std::max<size_t>(P.refkp.size(), 1)));
    ORBX_TRY(h->d_q.ensure(std::max<size_t>(P.desc.size(), 32)));
    ORBX_TRY(h->d_t.ensure(std::max<size_t>(P.refdesc.size(), 32)));
    ORBX_TRY(h->d_flags.ensure(std::max<size_t>(P.flags.size(), 1)));
    ORBX_TRY(h->d_items.ensure(std::max<size_t>(P.items.size(), 1)));
    ORBX_TRY(h->d_ints.ensure(std::max<size_t>(P.jobs.size(), 1)));
    ORBX_TRY(h->d_idx.ensure(total_n));
    ORBX_TRY(h->d_dist.ensure(total_n));
    ORBX_TRY(h->d_pushes.ensure(2 * total_n + 2 * P.jobs.size() + 2));
    auto up = [&](void* d, const void* src, size_t bytes) -> hipError_t {
        return bytes ? hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, s) : hipSuccess;
    };
    ORBX_HIP(up(h->d_kp.p, P.kp.data(), P.kp.size() * sizeof(orbx_keypoint)));
    ORBX_HIP(up(h->d_refkp.p, P.refkp.data(), P.refkp.size() * sizeof(orbx_keypoint)));
    ORBX_HIP(up(h->d_q.p, P.desc.data(), P.desc.size()));
    ORBX_HIP(up(h->d_t.p, P.refdesc.data(), P.refdesc.size()));
    ORBX_HIP(up(h->d_flags.p, P.flags.data(), P.flags.size()));
    ORBX_HIP(up(h->d_items.p, P.items.data(), P.items.size() * sizeof(int32_t)));
    ORBX_HIP(up(h->d_ints.p, P.jobs.data(), P.jobs.size() * sizeof(int32_t)));
    return ORBX_OK;
}

orbx_status raise_lds(const void* kernel, size_t bytes, size_t& current) {
    if (bytes > 48 * 1024) ORBX_TRY(ensure_dynamic_lds(kernel, bytes));
    current = std::max(current, bytes);
    return ORBX_OK;
}

// MatcherBase::createMatches (orb_matcher.h:81-93) per pair; pair p writes at matches + (features of earlier pairs).
void create_matches_batch(const std::vector<int32_t>& matched, const std::vector<int32_t>& dists, const int* n_of,
                          int n_pairs, orbx_match* matches, int* counts) {
    size_t base = 0;
    for (int p = 0; p < n_pairs; p++) {
        int c = 0;
        for (int i = 0; i < n_of[p]; i++)
            if (matched[base + i] > 0) matches[base + c++] = orbx_match{i, matched[base + i], -1, (float)dists[base + i]};
        counts[p] = c;
        base += n_of[p];
    }
}

}  // namespace

extern "C" {

orbx_status orbx_match_bow_batch(orbx_matcher* h, const orbx_bow_pair* pairs, int n_pairs, int low_threshold,
                                 float nn_ratio, int check_orientation, orbx_match* matches, int* counts) {
    ORBX_ENTER(h);
    if (!h || n_pairs < 0) return ORBX_BAD_ARG;
    if (n_pairs == 0) return ORBX_OK;
    if (!pairs || !counts) return ORBX_BAD_ARG;
    NodePack P;
    std::vector<int> n_of(n_pairs);
    size_t total_n = 0;
    int max_n = 0;
    for (int p = 0; p < n_pairs; p++) {
        const orbx_bow_pair& q = pairs[p];
        counts[p] = 0;
        if (q.n < 0 || q.n_ref < 0 || q.n_nodes < 0 || q.n_ref_nodes < 0) return ORBX_BAD_ARG;
        if (q.n > kMaxFeatures) return ORBX_CAPACITY;
        const bool live = q.n > 0 && q.n_ref > 0 && q.n_nodes > 0 && q.n_ref_nodes > 0;
        if (live && (!q.keypoints || !q.descriptors || !q.ref_keypoints || !q.ref_descriptors || !q.ref_has_point))
            return ORBX_BAD_ARG;
        if (live && (!valid_feature_vector(q.node_ids, q.node_start, q.node_items, q.n_nodes, q.n) ||
                     !valid_feature_vector(q.ref_node_ids, q.ref_node_start, q.ref_node_items, q.n_ref_nodes, q.n_ref)))
            return ORBX_BAD_ARG;
        NodePack::Off o{P.kp.size(), P.refkp.size(), P.flags.size(), P.items.size(), 0, P.jobs.size(), 0};
        n_of[p] = q.n;
        if (live) {
            // the reference's walk over the two std::maps with lower_bound jumps == intersection of the sorted id lists
            for (int f = 0, rf = 0; f < q.n_nodes && rf < q.n_ref_nodes;) {
                if (q.node_ids[f] == q.ref_node_ids[rf]) {
                    for (int a = q.ref_node_start[rf]; a < q.ref_node_start[rf + 1]; a++) {
                        P.jobs.push_back(q.ref_node_items[a]);
                        P.jobs.push_back(q.node_start[f]);
                        P.jobs.push_back(q.node_start[f + 1] - q.node_start[f]);
                    }
                    f++;
                    rf++;
                } else if (q.ref_node_ids[rf] < q.node_ids[f]) rf++;
                else f++;
            }
            o.n_jobs = (int)((P.jobs.size() - o.jobs) / 3);
            append(P.kp, q.keypoints, (size_t)q.n);
            append(P.desc, q.descriptors, (size_t)q.n * 32);
            append(P.refkp, q.ref_keypoints, (size_t)q.n_ref);
            append(P.refdesc, q.ref_descriptors, (size_t)q.n_ref * 32);
            append(P.flags, q.ref_has_point, (size_t)q.n_ref);
            append(P.items, q.node_items, (size_t)q.node_start[q.n_nodes]);
        } else {
            // nothing can match: the kernel still clears the pair's outputs
            P.kp.resize(P.kp.size() + q.n);
            P.desc.resize(P.desc.size() + (size_t)q.n * 32);
        }
        P.off.push_back(o);
        total_n += q.n;
        max_n = std::max(max_n, q.n);
    }
    if (total_n == 0) return ORBX_OK;
    if (!matches) return ORBX_BAD_ARG;
    ORBX_HIP(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    ORBX_TRY(upload_pack(h, P, total_n, s));
    std::vector<BowArgs> args(n_pairs);
    size_t out = 0;
    for (int p = 0; p < n_pairs; p++) {
        const NodePack::Off& o = P.off[p];
        BowArgs& A = args[p];
        A.kps = h->d_kp.p + o.kp;
        A.n = n_of[p];
        A.desc = h->d_q.p + o.kp * 32;
        A.ref_kps = h->d_refkp.p + o.refkp;
        A.ref_desc = h->d_t.p + o.refkp * 32;
        A.ref_flag = h->d_flags.p + o.flags;
        A.jobs = h->d_ints.p + o.jobs;
        A.n_jobs = o.n_jobs;
        A.items = h->d_items.p + o.items;
        A.low_threshold = low_threshold;
        A.nn_ratio = nn_ratio;
        A.check_orientation = check_orientation;
        A.matched = h->d_idx.p + out;
        A.dists = h->d_dist.p + out;
        A.pushes = h->d_pushes.p + 2 * (o.jobs / 3);
        out += n_of[p];
    }
    ORBX_TRY(h->d_args.ensure(sizeof(BowArgs) * (size_t)n_pairs));
    ORBX_HIP(hipMemcpyAsync(h->d_args.p, args.data(), sizeof(BowArgs) * (size_t)n_pairs, hipMemcpyHostToDevice, s));
    const size_t shmem = sizeof(int32_t) * (size_t)std::max(max_n, 1);
    ORBX_TRY(raise_lds((const void*)k_bow_match, shmem, h->bow_lds_attr));
    hipLaunchKernelGGL(k_bow_match, dim3(n_pairs), dim3(64), shmem, s, (const BowArgs*)h->d_args.p);
    ORBX_HIP(hipGetLastError());
    std::vector<int32_t> matched(total_n), dists(total_n);
    ORBX_HIP(hipMemcpyAsync(matched.data(), h->d_idx.p, sizeof(int32_t) * total_n, hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipMemcpyAsync(dists.data(), h->d_dist.p, sizeof(int32_t) * total_n, hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipStreamSynchronize(s));
    create_matches_batch(matched, dists, n_of.data(), n_pairs, matches, counts);
    return ORBX_OK;
}

orbx_status orbx_match_bow(orbx_matcher* h, const orbx_keypoint* keypoints, int n, const uint8_t* descriptors,
                           const orbx_keypoint* ref_keypoints, int n_ref, const uint8_t* ref_descriptors,
                           const uint8_t* ref_has_point, const uint32_t* node_ids, const int32_t* node_start,
                           const int32_t* node_items, int n_nodes, const uint32_t* ref_node_ids,
                           const int32_t* ref_node_start, const int32_t* ref_node_items, int n_ref_nodes,
                           int low_threshold, float nn_ratio, int check_orientation, orbx_match* matches,
                           int* count) {
    ORBX_ENTER(h);
    if (!h || !count) return ORBX_BAD_ARG;
    *count = 0;
    const orbx_bow_pair q{keypoints, n, descriptors, ref_keypoints, n_ref, ref_descriptors, ref_has_point,
                          node_ids, node_start, node_items, n_nodes, ref_node_ids, ref_node_start, ref_node_items,
                          n_ref_nodes};
    return orbx_match_bow_batch(h, &q, 1, low_threshold, nn_ratio, check_orientation, matches, count);
}

orbx_status orbx_match_epipolar_batch(orbx_matcher* h, const orbx_epipolar_pair* pairs, int n_pairs,
                                      const float* scale_sigmas, int n_levels, int low_threshold,
                                      int check_orientation, orbx_match* matches, int* counts) {
    ORBX_ENTER(h);
    if (!h || n_pairs < 0) return ORBX_BAD_ARG;
    if (n_pairs == 0) return ORBX_OK;
    if (!pairs || !counts || !scale_sigmas || n_levels <= 0) return ORBX_BAD_ARG;
    NodePack P;
    std::vector<int> n_of(n_pairs);
    size_t total_n = 0;
    int max_n = 0;
    for (int p = 0; p < n_pairs; p++) {
        const orbx_epipolar_pair& q = pairs[p];
        counts[p] = 0;
        if (q.n < 0 || q.n_ref < 0 || q.n_nodes < 0 || q.n_ref_nodes < 0) return ORBX_BAD_ARG;
        if (q.n > kMaxFeatures) return ORBX_CAPACITY;
        const bool live = q.n > 0 && q.n_ref > 0 && q.n_nodes > 0 && q.n_ref_nodes > 0;
        if (live && (!q.keypoints || !q.descriptors || !q.ref_keypoints || !q.ref_descriptors || !q.has_point ||
                     !q.ref_has_point || !q.fundamental))
            return ORBX_BAD_ARG;
        if (live && (!valid_feature_vector(q.node_ids, q.node_start, q.node_items, q.n_nodes, q.n) ||
                     !valid_feature_vector(q.ref_node_ids, q.ref_node_start, q.ref_node_items, q.n_ref_nodes, q.n_ref)))
            return ORBX_BAD_ARG;
        NodePack::Off o{P.kp.size(), P.refkp.size(), P.flags.size(), P.items.size(), 0, P.jobs.size(), 0};
        n_of[p] = q.n;
        if (live) {
            // common nodes in ascending order; where the reference would index out of range (row idx of the
            // reference descriptors, matched[ref_idx]) the call is rejected
            for (int f = 0, rf = 0; f < q.n_nodes && rf < q.n_ref_nodes;) {
                if (q.node_ids[f] == q.ref_node_ids[rf]) {
                    bool any_free = false;
                    for (int b = q.node_start[f]; b < q.node_start[f + 1]; b++) {
                        const int idx = q.node_items[b];
                        if (q.has_point[idx]) continue;
                        if (idx >= q.n_ref) return ORBX_BAD_ARG;
                        any_free = true;
                    }
                    if (any_free)
                        for (int a = q.ref_node_start[rf]; a < q.ref_node_start[rf + 1]; a++)
                            if (q.ref_node_items[a] >= q.n) return ORBX_BAD_ARG;
                    P.jobs.push_back(q.node_start[f]);
                    P.jobs.push_back(q.node_start[f + 1] - q.node_start[f]);
                    P.jobs.push_back(q.ref_node_start[rf]);
                    P.jobs.push_back(q.ref_node_start[rf + 1] - q.ref_node_start[rf]);
                    f++;
                    rf++;
                } else if (q.ref_node_ids[rf] < q.node_ids[f]) rf++;
                else f++;
            }
            o.n_jobs = (int)((P.jobs.size() - o.jobs) / 4);
            append(P.kp, q.keypoints, (size_t)q.n);
            append(P.desc, q.descriptors, (size_t)q.n * 32);
            append(P.refkp, q.ref_keypoints, (size_t)q.n_ref);
            append(P.refdesc, q.ref_descriptors, (size_t)q.n_ref * 32);
            append(P.flags, q.has_point, (size_t)q.n);
            append(P.flags, q.ref_has_point, (size_t)q.n_ref);
            append(P.items, q.node_items, (size_t)q.node_start[q.n_nodes]);
            o.ref_items = P.items.size();
            append(P.items, q.ref_node_items, (size_t)q.ref_node_start[q.n_ref_nodes]);
        } else {
            P.kp.resize(P.kp.size() + q.n);
            P.desc.resize(P.desc.size() + (size_t)q.n * 32);
            P.flags.resize(P.flags.size() + q.n, 1);  // "has a map point": the prologue then touches nothing else
        }
        P.off.push_back(o);
        total_n += q.n;
        max_n = std::max(max_n, q.n);
    }
    if (total_n == 0) return ORBX_OK;
    if (!matches) return ORBX_BAD_ARG;
    ORBX_HIP(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    ORBX_TRY(upload_pack(h, P, total_n, s));
    ORBX_TRY(h->d_f2.ensure(n_levels));
    ORBX_HIP(hipMemcpyAsync(h->d_f2.p, scale_sigmas, sizeof(float) * n_levels, hipMemcpyHostToDevice, s));
    std::vector<EpiArgs> args(n_pairs);
    size_t out = 0;
    for (int p = 0; p < n_pairs; p++) {
        const NodePack::Off& o = P.off[p];
        const orbx_epipolar_pair& q = pairs[p];
        EpiArgs& A = args[p];
        A.kps = h->d_kp.p + o.kp;
        A.n = n_of[p];
        A.desc = h->d_q.p + o.kp * 32;
        A.ref_kps = h->d_refkp.p + o.refkp;
        A.n_ref = o.n_jobs ? q.n_ref : 0;
        A.ref_desc = h->d_t.p + o.refkp * 32;
        A.has_mp = h->d_flags.p + o.flags;
        A.ref_has_mp = h->d_flags.p + o.flags + n_of[p];
        for (int i = 0; i < 9; i++) A.F[i] = q.fundamental ? q.fundamental[i] : 0.f;
        A.scale_sigmas = h->d_f2.p;
        A.n_levels = n_levels;
        A.jobs = h->d_ints.p + o.jobs;
        A.n_jobs = o.n_jobs;
        A.items = h->d_items.p + o.items;
        A.ref_items = h->d_items.p + o.ref_items;
        A.low_threshold = low_threshold;
        A.check_orientation = check_orientation;
        A.matched = h->d_idx.p + out;
        A.dists = h->d_dist.p + out;
        A.pushes = h->d_pushes.p + 2 * out;
        out += n_of[p];
    }
    ORBX_TRY(h->d_args.ensure(sizeof(EpiArgs) * (size_t)n_pairs));
    ORBX_HIP(hipMemcpyAsync(h->d_args.p, args.data(), sizeof(EpiArgs) * (size_t)n_pairs, hipMemcpyHostToDevice, s));
    const size_t shmem = (sizeof(int32_t) + sizeof(unsigned short)) * (size_t)std::max(max_n, 1);
    ORBX_TRY(raise_lds((const void*)k_epipolar_match, shmem, h->epi_lds_attr));
    hipLaunchKernelGGL(k_epipolar_match, dim3(n_pairs), dim3(64), shmem, s, (const EpiArgs*)h->d_args.p);
    ORBX_HIP(hipGetLastError());
    std::vector<int32_t> matched(total_n), dists(total_n);
    ORBX_HIP(hipMemcpyAsync(matched.data(), h->d_idx.p, sizeof(int32_t) * total_n, hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipMemcpyAsync(dists.data(), h->d_dist.p, sizeof(int32_t) * total_n, hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipStreamSynchronize(s));
    create_matches_batch(matched, dists, n_of.data(), n_pairs, matches, counts);
    return ORBX_OK;
}

orbx_status orbx_match_epipolar(orbx_matcher* h, const orbx_keypoint* keypoints, int n, const uint8_t* descriptors,
                                const orbx_keypoint* ref_keypoints, int n_ref, const uint8_t* ref_descriptors,
                                const uint8_t* has_point, const uint8_t* ref_has_point, const float* fundamental,
                                const float* scale_sigmas, int n_levels, const uint32_t* node_ids,
                                const int32_t* node_start, const int32_t* node_items, int n_nodes,
                                const uint32_t* ref_node_ids, const int32_t* ref_node_start,
                                const int32_t* ref_node_items, int n_ref_nodes, int low_threshold,
                                int check_orientation, orbx_match* matches, int* count) {
    ORBX_ENTER(h);
    if (!h || !count) return ORBX_BAD_ARG;
    *count = 0;
    const orbx_epipolar_pair q{keypoints, n, descriptors, ref_keypoints, n_ref, ref_descriptors, has_point,
                               ref_has_point, fundamental, node_ids, node_start, node_items, n_nodes, ref_node_ids,
                               ref_node_start, ref_node_items, n_ref_nodes};
    return orbx_match_epipolar_batch(h, &q, 1, scale_sigmas, n_levels, low_threshold, check_orientation, matches,
                                     count);
}

orbx_status orbx_fuse_search(orbx_matcher* h, const orbx_grid* g, const int32_t* cell_start, const int32_t* cell_items,
                             const orbx_keypoint* keypoints, int n, const uint8_t* descriptors, int n_points,
                             const uint8_t* in_view, const float* proj_xy, const int32_t* pred_level,
                             const uint8_t* point_descriptors, const float* scale_factors,
                             const float* inv_scale_sigmas, int n_levels, float radius_multiplier,
                             int32_t* best_index, int32_t* best_distance) {
    ORBX_ENTER(h);
    if (!h || !g || n < 0 || n_points < 0) return ORBX_BAD_ARG;
    if (n_points == 0) return ORBX_OK;
    if (!best_index || !best_distance) return ORBX_BAD_ARG;
    if (n == 0) {
        for (int p = 0; p < n_points; p++) { best_index[p] = -1; best_distance[p] = 256; }
        return ORBX_OK;
    }
    if (!cell_start || !cell_items || !keypoints || !descriptors || !in_view || !proj_xy || !pred_level ||
        !point_descriptors || !scale_factors || !inv_scale_sigmas)
        return ORBX_BAD_ARG;
    if (g->grid_cols <= 0 || g->grid_rows <= 0 || g->grid_size_x <= 0 || g->grid_size_y <= 0 || n_levels <= 0)
        return ORBX_BAD_ARG;
    const int ncell = g->grid_cols * g->grid_rows;
    if (cell_start[0] != 0 || cell_start[ncell] != n) return ORBX_BAD_ARG;
    for (int c = 0; c < ncell; c++)
        if (cell_start[c + 1] < cell_start[c]) return ORBX_BAD_ARG;
    for (int i = 0; i < n; i++) {
        if (cell_items[i] < 0 || cell_items[i] >= n) return ORBX_BAD_ARG;
        if (keypoints[i].octave >= n_levels) return ORBX_BAD_ARG;
    }
    ORBX_HIP(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    ORBX_TRY(h->d_cells.ensure(ncell + 1));
    ORBX_TRY(h->d_items.ensure(n));
    ORBX_TRY(h->d_kp.ensure(n));
    ORBX_TRY(h->d_q.ensure((size_t)n * 32));
    ORBX_TRY(h->d_t.ensure((size_t)n_points * 32));
    ORBX_TRY(h->d_flags.ensure(n_points));
    ORBX_TRY(h->d_f0.ensure((size_t)2 * n_points));
    ORBX_TRY(h->d_f2.ensure((size_t)2 * n_levels));
    ORBX_TRY(h->d_ints.ensure(n_points));
    ORBX_TRY(h->d_idx.ensure(n_points));
    ORBX_TRY(h->d_dist.ensure(n_points));
    ORBX_HIP(hipMemcpyAsync(h->d_cells.p, cell_start, sizeof(int32_t) * (ncell + 1), hipMemcpyHostToDevice, s));
    ORBX_HIP(hipMemcpyAsync(h->d_items.p, cell_items, sizeof(int32_t) * n, hipMemcpyHostToDevice, s));
    ORBX_HIP(hipMemcpyAsync(h->d_kp.p, keypoints, sizeof(orbx_keypoint) * n, hipMemcpyHostToDevice, s));
    ORBX_HIP(hipMemcpyAsync(h->d_q.p, descriptors, (size_t)n * 32, hipMemcpyHostToDevice, s));
    ORBX_HIP(hipMemcpyAsync(h->d_t.p, point_descriptors, (size_t)n_points * 32, hipMemcpyHostToDevice, s));
    ORBX_HIP(hipMemcpyAsync(h->d_flags.p, in_view, n_points, hipMemcpyHostToDevice, s));
    ORBX_HIP(hipMemcpyAsync(h->d_f0.p, proj_xy, sizeof(float) * 2 * n_points, hipMemcpyHostToDevice, s));
    ORBX_HIP(hipMemcpyAsync(h->d_f2.p, scale_factors, sizeof(float) * n_levels, hipMemcpyHostToDevice, s));
    ORBX_HIP(hipMemcpyAsync(h->d_f2.p + n_levels, inv_scale_sigmas, sizeof(float) * n_levels, hipMemcpyHostToDevice, s));
    ORBX_HIP(hipMemcpyAsync(h->d_ints.p, pred_level, sizeof(int32_t) * n_points, hipMemcpyHostToDevice, s));
    FuseArgs A;
    A.g = *g;
    A.cell_start = h->d_cells.p;
    A.cell_items = h->d_items.p;
    A.kps = h->d_kp.p;
    A.desc = h->d_q.p;
    A.n_pts = n_points;
    A.in_view = h->d_flags.p;
    A.proj_xy = h->d_f0.p;
    A.pred_level = h->d_ints.p;
    A.pt_desc = h->d_t.p;
    A.scale_factors = h->d_f2.p;
    A.inv_scale_sigmas = h->d_f2.p + n_levels;
    A.n_levels = n_levels;
    A.radius_multiplier = radius_multiplier;
    A.best_idx = h->d_idx.p;
    A.best_dist = h->d_dist.p;
    hipLaunchKernelGGL(k_fuse_search, dim3(div_up(n_points, 16)), dim3(256), 0, s, A);
    ORBX_HIP(hipGetLastError());
    ORBX_HIP(hipMemcpyAsync(best_index, h->d_idx.p, sizeof(int32_t) * n_points, hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipMemcpyAsync(best_distance, h->d_dist.p, sizeof(int32_t) * n_points, hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipStreamSynchronize(s));
    return ORBX_OK;
}

orbx_status orbx_best_descriptor_batch(orbx_matcher* h, const uint8_t* descriptors, const int32_t* point_start,
                                       int n_points, int32_t* best_index) {
    ORBX_ENTER(h);
    if (!h || !point_start || !best_index || n_points < 0) return ORBX_BAD_ARG;
    if (n_points == 0) return ORBX_OK;
    if (point_start[0] != 0) return ORBX_BAD_ARG;
    for (int p = 0; p < n_points; p++) {
        if (point_start[p + 1] < point_start[p]) return ORBX_BAD_ARG;
        if (point_start[p + 1] - point_start[p] > kBestMaxObs) return ORBX_CAPACITY;
    }
    const int total = point_start[n_points];
    if (total > 0 && !descriptors) return ORBX_BAD_ARG;
    ORBX_HIP(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    ORBX_TRY(h->d_q.ensure(std::max((size_t)total * 32, (size_t)32)));
    ORBX_TRY(h->d_ints.ensure((size_t)n_points + 1));
    ORBX_TRY(h->d_idx.ensure(n_points));
    if (total > 0) ORBX_HIP(hipMemcpyAsync(h->d_q.p, descriptors, (size_t)total * 32, hipMemcpyHostToDevice, s));
    ORBX_HIP(hipMemcpyAsync(h->d_ints.p, point_start, sizeof(int32_t) * (n_points + 1), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_best_descriptor, dim3(n_points), dim3(64), 0, s, h->d_q.p, h->d_ints.p, n_points, h->d_idx.p);
    ORBX_HIP(hipGetLastError());
    ORBX_HIP(hipMemcpyAsync(best_index, h->d_idx.p, sizeof(int32_t) * n_points, hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipStreamSynchronize(s));
    return ORBX_OK;
}

}  // extern "C"
