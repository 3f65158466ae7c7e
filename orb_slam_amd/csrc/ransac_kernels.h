// Two-view RANSAC kernels (gfx950): point normalisation, 8-point F / DLT H hypothesis solve, and scoring of
// every hypothesis against every correspondence.  One LANE owns one hypothesis, so every sum is accumulated in
// the reference's own order (src/Initializer.cpp:369-585) and the results are bit-identical to the scalar
// evaluation; parallelism comes from the iterations x pairs hypotheses in flight.
// Must be compiled with -ffp-contract=off: the reference build does not fuse multiply-adds.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace orbx {

// IEEE double hypot with the evaluation order of glibc 2.35's generic (non-FMA) __hypot, which is what
// cv::JacobiSVD's std::hypot resolves to on x86-64 Linux.
__device__ __forceinline__ double hypot_kernel(double ax, double ay) {
    double t1, t2;
    double h = sqrt(ax * ax + ay * ay);
    if (h <= 2.0 * ay) {
        double delta = h - ay;
        t1 = ax * (2.0 * delta - ax);
        t2 = (delta - 2.0 * (ax - ay)) * delta;
    } else {
        double delta = h - ax;
        t1 = 2.0 * delta * (ax - 2.0 * ay);
        t2 = (4.0 * delta - ay) * ay + delta * delta;
    }
    h -= (t1 + t2) / (2.0 * h);
    return h;
}

__device__ __forceinline__ double hypot_f64(double x, double y) {
    const double SCALE = 0x1p-600, LARGE_VAL = 0x1p+511, TINY_VAL = 0x1p-459, EPS = 0x1p-54;
    if (!isfinite(x) || !isfinite(y)) {
        if (isinf(x) || isinf(y)) return INFINITY;
        return x + y;
    }
    x = fabs(x);
    y = fabs(y);
    double ax = x < y ? y : x;
    double ay = x < y ? x : y;
    if (ax > LARGE_VAL) {
        if (ay <= ax * EPS) return ax + ay;
        return hypot_kernel(ax * SCALE, ay * SCALE) / SCALE;
    }
    if (ay < TINY_VAL) {
        if (ax >= ay / EPS) return ax + ay;
        return hypot_kernel(ax / SCALE, ay / SCALE) * SCALE;
    }
    if (ax >= ay / EPS) return ax + ay;
    return hypot_kernel(ax, ay);
}

// x * y + z for FLOAT-valued x and y, in double.  The product of two floats has at most 48 significant bits and an
// exponent far inside the double range: it is exact, so the separate multiply and add of the scalar code round once,
// in the addition, and the fused form rounds the same sum once - the same bits, one instruction instead of two.
// (Only for products of two float-valued operands; anywhere else fusing would change results.)
__device__ __forceinline__ double fma_exact_product(float x, float y, double z) {
    return __builtin_fma((double)x, (double)y, z);
}

// cv::JacobiSVDImpl_<float> (OpenCV 3.4 core/lapack.cpp): one-sided Jacobi on the N rows (length M) of At, with
// the rotations accumulated in Vt (N x N) when WANT_V.  Every loop over rows and columns is unrolled and every
// index is a compile-time constant, so At, Vt and W live in registers (one hypothesis per lane, no scratch memory);
// the operations and their order are those of the scalar loops.  On return W holds the squared row norms as the
// sweeps left them (OpenCV recomputes them next: see svd_finish).
// FUSE selects fma_exact_product for the dot products and row norms (same bits either way; whichever form leaves
// the caller's register allocation better off is used).
template <int M, int N, bool WANT_V, bool FUSE = true>
__device__ __forceinline__ void jacobi_sweeps(float (&At)[N * M], float (&Vt)[WANT_V ? N * N : 1], double (&W)[N]) {
    const float eps = 1.1920928955078125e-7f * 2;  // FLT_EPSILON * 2
    const int max_iter = M > 30 ? M : 30;
#pragma unroll
    for (int i = 0; i < N; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < M; k++) {
            const float t = At[i * M + k];
            sd = FUSE ? fma_exact_product(t, t, sd) : sd + (double)t * t;
        }
        W[i] = sd;
        if (WANT_V) {
#pragma unroll
            for (int k = 0; k < N; k++) Vt[i * N + k] = k == i ? 1.f : 0.f;
        }
    }
    for (int iter = 0; iter < max_iter; iter++) {
        bool changed = false;
#pragma unroll
        for (int i = 0; i < N - 1; i++) {
#pragma unroll
            for (int j = i + 1; j < N; j++) {
                double a = W[i], p = 0, b = W[j];
#pragma unroll
                for (int k = 0; k < M; k++)
                    p = FUSE ? fma_exact_product(At[i * M + k], At[j * M + k], p) : p + (double)At[i * M + k] * At[j * M + k];
                if (fabs(p) <= eps * sqrt((double)a * b)) continue;
                p *= 2;
                const double beta = a - b, gamma = hypot_f64((double)p, beta);
                float c, s;
                if (beta < 0) {
                    const double delta = (gamma - beta) * 0.5;
                    s = (float)sqrt(delta / gamma);
                    c = (float)(p / (gamma * s * 2));
                } else {
                    c = (float)sqrt((gamma + beta) / (gamma * 2));
                    s = (float)(p / (gamma * c * 2));
                }
                a = b = 0;
#pragma unroll
                for (int k = 0; k < M; k++) {
                    const float t0 = c * At[i * M + k] + s * At[j * M + k];
                    const float t1 = -s * At[i * M + k] + c * At[j * M + k];
                    At[i * M + k] = t0;
                    At[j * M + k] = t1;
                    a = FUSE ? fma_exact_product(t0, t0, a) : a + (double)t0 * t0;
                    b = FUSE ? fma_exact_product(t1, t1, b) : b + (double)t1 * t1;
                }
                W[i] = a;
                W[j] = b;
                changed = true;
                if (WANT_V) {
#pragma unroll
                    for (int k = 0; k < N; k++) {
                        const float t0 = c * Vt[i * N + k] + s * Vt[j * N + k];
                        const float t1 = -s * Vt[i * N + k] + c * Vt[j * N + k];
                        Vt[i * N + k] = t0;
                        Vt[j * N + k] = t1;
                    }
                }
            }
        }
        if (!changed) break;
    }
}

// The rest of JacobiSVDImpl_: singular values from the row norms, selection sort into descending order (first of
// equal maxima wins, as `W[j] < W[k]` decides), rows of At (SORT_A) and Vt (WANT_V) following.  perm[i] receives the
// original index of the row that ends at position i.
template <int M, int N, bool WANT_V, bool SORT_A>
__device__ __forceinline__ void svd_finish(float (&At)[N * M], float (&Vt)[WANT_V ? N * N : 1], double (&W)[N],
                                           int (&perm)[N]) {
#pragma unroll
    for (int i = 0; i < N; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < M; k++) {
            const float t = At[i * M + k];
            sd = fma_exact_product(t, t, sd);
        }
        W[i] = sqrt(sd);
        perm[i] = i;
    }
#pragma unroll
    for (int i = 0; i < N - 1; i++) {
        int j = i;
        double wj = W[i];
#pragma unroll
        for (int k = i + 1; k < N; k++)
            if (wj < W[k]) { j = k; wj = W[k]; }
#pragma unroll
        for (int k = i + 1; k < N; k++) {
            if (j != k) continue;
            const double tw = W[i]; W[i] = W[k]; W[k] = tw;
            const int tp = perm[i]; perm[i] = perm[k]; perm[k] = tp;
            if (SORT_A) {
#pragma unroll
                for (int q = 0; q < M; q++) { const float t = At[i * M + q]; At[i * M + q] = At[k * M + q]; At[k * M + q] = t; }
            }
            if (WANT_V && SORT_A) {
#pragma unroll
                for (int q = 0; q < N; q++) { const float t = Vt[i * N + q]; Vt[i * N + q] = Vt[k * N + q]; Vt[k * N + q] = t; }
            }
        }
    }
}

// Completion of the first N1 rows of At to an orthonormal set, as OpenCV does it: rows with a vanishing singular
// value, and rows >= N, are synthesised from cv::RNG(0x12345678) and Gram-Schmidt (for the 8x9 system that
// synthesised 9th row IS the null vector the reference takes).  At holds N1 rows; W the N sorted singular values.
template <int M, int N, int N1>
__device__ __forceinline__ void svd_complete(float (&At)[N1 * M], const double (&W)[N]) {
    const double minval = 1.17549435082228750797e-38;  // FLT_MIN
    const float eps = 1.1920928955078125e-7f * 2;
    unsigned long long rng = 0x12345678ull;
#pragma unroll
    for (int i = 0; i < N1; i++) {
        double sd = i < N ? W[i < N ? i : 0] : 0;
        for (int ii = 0; ii < 100 && sd <= minval; ii++) {
            const float val0 = (float)(1. / M);
#pragma unroll
            for (int k = 0; k < M; k++) {
                rng = (unsigned long long)(unsigned)rng * 4164903690U + (unsigned)(rng >> 32);
                At[i * M + k] = ((unsigned)rng & 256) != 0 ? val0 : -val0;
            }
            for (int iter = 0; iter < 2; iter++) {
#pragma unroll
                for (int j = 0; j < i; j++) {
                    sd = 0;
#pragma unroll
                    for (int k = 0; k < M; k++) sd += At[i * M + k] * At[j * M + k];
                    float asum = 0;
#pragma unroll
                    for (int k = 0; k < M; k++) {
                        const float t = (float)(At[i * M + k] - sd * At[j * M + k]);
                        At[i * M + k] = t;
                        asum += fabsf(t);
                    }
                    asum = asum > eps * 100 ? 1 / asum : 0;
#pragma unroll
                    for (int k = 0; k < M; k++) At[i * M + k] *= asum;
                }
            }
            sd = 0;
#pragma unroll
            for (int k = 0; k < M; k++) {
                const float t = At[i * M + k];
                sd += (double)t * t;
            }
            sd = sqrt(sd);
        }
        const float s = (float)(sd > minval ? 1 / sd : 0.);
#pragma unroll
        for (int k = 0; k < M; k++) At[i * M + k] *= s;
    }
}

// cv::gemm small-matrix path for 3x3 CV_32F without transpose flags: float dot products, left to right.
__device__ __forceinline__ void mul33(const float* A, const float* B, float* C) {
    float out[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) out[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
#pragma unroll
    for (int i = 0; i < 9; i++) C[i] = out[i];
}

// cv::invert 3x3 CV_32F (DECOMP_LU closed form, double determinant and cofactors).
__device__ __forceinline__ void inv33(const float* S, float* D) {
    double d = (double)S[0] * ((double)S[4] * S[8] - (double)S[5] * S[7]) -
               (double)S[1] * ((double)S[3] * S[8] - (double)S[5] * S[6]) +
               (double)S[2] * ((double)S[3] * S[7] - (double)S[4] * S[6]);
    if (d == 0) {
        for (int i = 0; i < 9; i++) D[i] = 0.f;
        return;
    }
    float t[9];
    d = 1. / d;
    t[0] = (float)(((double)S[4] * S[8] - (double)S[5] * S[7]) * d);
    t[1] = (float)(((double)S[2] * S[7] - (double)S[1] * S[8]) * d);
    t[2] = (float)(((double)S[1] * S[5] - (double)S[2] * S[4]) * d);
    t[3] = (float)(((double)S[5] * S[6] - (double)S[3] * S[8]) * d);
    t[4] = (float)(((double)S[0] * S[8] - (double)S[2] * S[6]) * d);
    t[5] = (float)(((double)S[2] * S[3] - (double)S[0] * S[5]) * d);
    t[6] = (float)(((double)S[3] * S[7] - (double)S[4] * S[6]) * d);
    t[7] = (float)(((double)S[1] * S[6] - (double)S[0] * S[7]) * d);
    t[8] = (float)(((double)S[0] * S[4] - (double)S[1] * S[3]) * d);
    for (int i = 0; i < 9; i++) D[i] = t[i];
}

// `mat *= 1.0 / mat(2,2)` on CV_32F: convertTo with the factor narrowed to float, dst = src * a + 0.f.
__device__ __forceinline__ void scale33(float* M, double alpha) {
    const float a = (float)alpha;
    for (int i = 0; i < 9; i++) M[i] = M[i] * a + 0.f;
}

// geometry::normalizePoints (reference src/geometry/utils.cpp:18-52) for one point set per wavefront.
// The float coordinate sums and the double sum of distances are accumulated by lane 0 in index order; the
// per-point arithmetic runs on all lanes.  Output: normalised points and T (row-major 3x3).
// With pd non-null the un-normalised coordinates are also stored widened to double, one (u1, v1, u2, v2) record per
// correspondence (reference point, then point): what the fundamental scorer reads, so that the widening happens once
// per correspondence and not once per hypothesis.
__global__ __launch_bounds__(64) void k_normalize(const float* __restrict__ pts, const float* __restrict__ ref, int n,
                                                  float* __restrict__ pts_n, float* __restrict__ ref_n,
                                                  float* __restrict__ T_all, double* __restrict__ pd) {
    __shared__ double s_norm[1024];
    __shared__ double s_bcast[3];
    const int pair = blockIdx.x, side = blockIdx.y, lane = threadIdx.x;
    const float* in = (side == 0 ? pts : ref) + (size_t)pair * n * 2;
    float* out = (side == 0 ? pts_n : ref_n) + (size_t)pair * n * 2;
    {
        // the coordinate sums run in index order on lane 0 (float addition is not associative); the wavefront
        // feeds it through LDS, 512 points at a time, so the serial part never waits on memory
        float2* s_xy = (float2*)s_norm;  // 1024 doubles = room for 1024 float2
        float sx = 0.f, sy = 0.f;
        for (int base = 0; base < n; base += 1024) {
            const int cnt = min(1024, n - base);
            for (int i = lane; i < cnt; i += 64) s_xy[i] = ((const float2*)in)[base + i];
            __syncthreads();
            if (lane == 0)
                for (int i = 0; i < cnt; i++) {
                    const float2 p = s_xy[i];
                    sx = sx + p.x;
                    sy = sy + p.y;
                }
            __syncthreads();
        }
        if (lane == 0) {
            const double n_inv = 1.0 / n;
            s_bcast[0] = sx * n_inv;
            s_bcast[1] = sy * n_inv;
        }
    }
    __syncthreads();
    const double mean_x = s_bcast[0], mean_y = s_bcast[1];
    if (pd) {
        double2* rec = (double2*)(pd + (size_t)pair * n * 4) + (side == 0 ? 1 : 0);  // side 0 = the points: (u2, v2)
        for (int i = lane; i < n; i += 64) rec[2 * i] = make_double2((double)in[2 * i], (double)in[2 * i + 1]);
    }
    double scale = 0.0;
    for (int base = 0; base < n; base += 1024) {
        const int cnt = min(1024, n - base);
        for (int i = lane; i < cnt; i += 64) {
            const float nx = (float)(in[2 * (base + i)] - mean_x), ny = (float)(in[2 * (base + i) + 1] - mean_y);
            out[2 * (base + i)] = nx;
            out[2 * (base + i) + 1] = ny;
            s_norm[i] = sqrt((double)nx * nx + (double)ny * ny);
        }
        __syncthreads();
        if (lane == 0)
            for (int i = 0; i < cnt; i++) scale += s_norm[i];
        __syncthreads();
    }
    if (lane == 0) {
        scale *= 1.0 / n;
        scale = sqrt(2.0) / scale;
        s_bcast[2] = scale;
    }
    __syncthreads();
    scale = s_bcast[2];
    for (int i = lane; i < n; i += 64) {
        out[2 * i] = (float)(out[2 * i] * scale);
        out[2 * i + 1] = (float)(out[2 * i + 1] * scale);
    }
    if (lane == 0) {
        float* T = T_all + ((size_t)pair * 2 + side) * 9;
        T[0] = (float)scale; T[1] = 0.f; T[2] = (float)(-mean_x * scale);
        T[3] = 0.f; T[4] = (float)scale; T[5] = (float)(-mean_y * scale);
        T[6] = 0.f; T[7] = 0.f; T[8] = 1.f;
    }
}

// K11F: one hypothesis per lane.  geometry::computeFundamentalMat (utils.cpp:61-103) on the 8 normalised
// correspondences of the set, then the de-normalisation and scaling of Initializer::findFundamentalMat
// (Initializer.cpp:336-345).
#ifndef ORBX_SOLVE_F_WAVES
#define ORBX_SOLVE_F_WAVES 2
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(ORBX_SOLVE_F_WAVES, ORBX_SOLVE_F_WAVES))) void k_solve_f(const float* __restrict__ pts_n, const float* __restrict__ ref_n, int n,
                                                const float* __restrict__ T_all, const uint32_t* __restrict__ sets8,
                                                int iterations, int pairs, float* __restrict__ F_all,
                                                int32_t* __restrict__ bad_sets) {
    // hypotheses of all pairs are numbered through, so wavefronts stay full when `iterations` is no multiple of 64
    const long long hyp = (long long)blockIdx.x * 64 + threadIdx.x;
    if (hyp >= (long long)pairs * iterations) return;
    const int pair = (int)(hyp / iterations), it = (int)(hyp - (long long)pair * iterations);
    const float* pn = pts_n + (size_t)pair * n * 2;
    const float* rn = ref_n + (size_t)pair * n * 2;
    const uint32_t* set = sets8 + ((size_t)pair * iterations + it) * 8;
    // 8x9 system used as is (rows = equations): N = 8 rows of length M = 9, completed to 9 rows
    // a set that names a correspondence outside [0, n) must not become an address: its index is replaced by 0 and
    // the pair is flagged, and k_select then reports the pair as invalid instead of a winner
    float At[81];
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t idx = set[i];
        if (idx >= (uint32_t)n) { idx = 0; bad = true; }
        const float u2 = pn[2 * idx], v2 = pn[2 * idx + 1], u1 = rn[2 * idx], v1 = rn[2 * idx + 1];
        At[i * 9 + 0] = u2 * u1; At[i * 9 + 1] = u2 * v1; At[i * 9 + 2] = u2;
        At[i * 9 + 3] = v2 * u1; At[i * 9 + 4] = v2 * v1; At[i * 9 + 5] = v2;
        At[i * 9 + 6] = u1; At[i * 9 + 7] = v1; At[i * 9 + 8] = 1;
    }
#pragma unroll
    for (int k = 0; k < 9; k++) At[72 + k] = 0.f;
    if (bad) bad_sets[pair] = 1;
    {
        float(&A8)[72] = *reinterpret_cast<float(*)[72]>(&At[0]);
        float no_v[1];
        double W[8];
        int perm[8];
        jacobi_sweeps<9, 8, false>(A8, no_v, W);
        svd_finish<9, 8, false, true>(A8, no_v, W, perm);
        svd_complete<9, 8, 9>(At, W);
    }
    // Fpre = vt.row(8) reshaped 3x3; the second SVD works on its transpose
    float B[9], V3[9];
    double W3[3];
    int perm3[3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) B[j * 3 + i] = At[72 + i * 3 + j];
    jacobi_sweeps<3, 3, true>(B, V3, W3);
    svd_finish<3, 3, true, true>(B, V3, W3, perm3);
    svd_complete<3, 3, 3>(B, W3);
    const float* T = T_all + (size_t)pair * 18;
    const float* refT = T + 9;
    const float w0 = (float)W3[0], w1 = (float)W3[1];
    float U[9], D[9] = {w0, 0, 0, 0, w1, 0, 0, 0, 0.f}, UD[9], Fn[9], tmp[9], Tt[9], F[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) U[j * 3 + i] = B[i * 3 + j];
    mul33(U, D, UD);
    mul33(UD, V3, Fn);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) Tt[j * 3 + i] = T[i * 3 + j];
    mul33(Tt, Fn, tmp);
    mul33(tmp, refT, F);
    if (fabs((double)F[8]) > 1e-8) scale33(F, 1.0 / F[8]);
    const size_t o = ((size_t)pair * iterations + it) * 9;
#pragma unroll
    for (int k = 0; k < 9; k++) F_all[o + k] = F[k];
}

constexpr bool kFuseSolveH = false;  // fused, the 9 x 16 system no longer fits 256 registers (it spills)

// K11H: one hypothesis per lane.  geometry::computeHomographyMat (utils.cpp:112-154), then the de-normalisation,
// inverse and scaling of Initializer::findHomographyMat (Initializer.cpp:476-486).
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_solve_h(const float* __restrict__ pts_n, const float* __restrict__ ref_n, int n,
                                                const float* __restrict__ T_all, const uint32_t* __restrict__ sets8,
                                                int iterations, int pairs, float* __restrict__ H_all,
                                                float* __restrict__ Hinv_all) {
    // hypotheses of all pairs are numbered through, so wavefronts stay full when `iterations` is no multiple of 64
    const long long hyp = (long long)blockIdx.x * 64 + threadIdx.x;
    if (hyp >= (long long)pairs * iterations) return;
    const int pair = (int)(hyp / iterations), it = (int)(hyp - (long long)pair * iterations);
    const float* pn = pts_n + (size_t)pair * n * 2;
    const float* rn = ref_n + (size_t)pair * n * 2;
    const uint32_t* set = sets8 + ((size_t)pair * iterations + it) * 8;
    // the 16x9 system is transposed: N = 9 rows of length M = 16; only Vt is needed
    float At[144], Vt[81];
    double W[9];
    int perm[9];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t idx = set[i];
        if (idx >= (uint32_t)n) idx = 0;  // k_solve_f flags the pair
        const float u2 = pn[2 * idx], v2 = pn[2 * idx + 1], u1 = rn[2 * idx], v1 = rn[2 * idx + 1];
        const int c0 = 2 * i, c1 = 2 * i + 1;
        At[0 * 16 + c0] = 0.0f; At[1 * 16 + c0] = 0.0f; At[2 * 16 + c0] = 0.0f;
        At[3 * 16 + c0] = -u1; At[4 * 16 + c0] = -v1; At[5 * 16 + c0] = -1;
        At[6 * 16 + c0] = v2 * u1; At[7 * 16 + c0] = v2 * v1; At[8 * 16 + c0] = v2;
        At[0 * 16 + c1] = u1; At[1 * 16 + c1] = v1; At[2 * 16 + c1] = 1;
        At[3 * 16 + c1] = 0.0f; At[4 * 16 + c1] = 0.0f; At[5 * 16 + c1] = 0.0f;
        At[6 * 16 + c1] = -u2 * u1; At[7 * 16 + c1] = -u2 * v1; At[8 * 16 + c1] = -u2;
    }
    jacobi_sweeps<16, 9, true, kFuseSolveH>(At, Vt, W);
    svd_finish<16, 9, true, false>(At, Vt, W, perm);
    // vt.row(8): the row the sort leaves last
    float Hn[9];
#pragma unroll
    for (int k = 0; k < 9; k++) Hn[k] = 0.f;
#pragma unroll
    for (int r = 0; r < 9; r++)
        if (perm[8] == r) {
#pragma unroll
            for (int k = 0; k < 9; k++) Hn[k] = Vt[r * 9 + k];
        }
    const float* T = T_all + (size_t)pair * 18;
    const float* refT = T + 9;
    float Tinv[9], tmp[9], H[9], Hinv[9];
    inv33(T, Tinv);
    mul33(Tinv, Hn, tmp);
    mul33(tmp, refT, H);
    inv33(H, Hinv);
    if (fabs((double)H[8]) > 1e-8) scale33(H, 1.0 / H[8]);
    const size_t o = ((size_t)pair * iterations + it) * 9;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        H_all[o + k] = H[k];
        Hinv_all[o + k] = Hinv[k];
    }
}

// ---- the public solvers for any number of correspondences ----------------------------------------------------
// geometry::computeFundamentalMat / computeHomographyMat (utils.cpp:61-154) accept any n; the systems are n x 9 and
// 2n x 9.  One wavefront works on one system.  cv::JacobiSVDImpl_<float> is a chain of row-pair rotations whose
// order cannot change, and every dot product and row norm is a double sum in index order; what IS independent is the
// per-element work, so the 64 lanes form the products (exact in double: float x float) and rotate the row elements
// side by side, park the products in LDS, and lane 0 adds them in index order.  At (N rows of length M, N <= 9)
// lives in global memory, Vt and W in LDS.
constexpr int kSvdChunk = 256;
struct SvdShared {
    double prod_a[kSvdChunk], prod_b[kSvdChunk];
    double W[9], bc[2];
    float Vt[81];
};

// sum_k x[k] * y[k], accumulated in double in index order; every lane returns the sum
__device__ __forceinline__ double wave_dot_ordered(const float* x, const float* y, int M, SvdShared& sh) {
    const int lane = threadIdx.x;
    double acc = 0;
    for (int base = 0; base < M; base += kSvdChunk) {
        const int cnt = min(kSvdChunk, M - base);
        for (int k = lane; k < cnt; k += 64) sh.prod_a[k] = (double)x[base + k] * y[base + k];
        __syncthreads();
        if (lane == 0)
            for (int k = 0; k < cnt; k++) acc += sh.prod_a[k];
        __syncthreads();
    }
    if (lane == 0) sh.bc[0] = acc;
    __syncthreads();
    acc = sh.bc[0];
    __syncthreads();
    return acc;
}

// JacobiSVDImpl_<float> on the N rows (length M) of At with Vt (N x N) accumulated in sh.Vt; on return sh.W holds
// the sorted singular values and Vt (and, with sort_rows, At) follow the sort.  With sort_rows and M == 9 the rows
// N .. 8 of At, and rows with a vanishing singular value, are completed to an orthonormal set as OpenCV does
// (At then has room for 9 rows).
__device__ void svd_rows_wave(float* At, int M, int N, bool sort_rows, SvdShared& sh) {
    const int lane = threadIdx.x;
    const float eps = 1.1920928955078125e-7f * 2;
    const int max_iter = M > 30 ? M : 30;
    for (int i = 0; i < N; i++) {
        const double w = wave_dot_ordered(At + (size_t)i * M, At + (size_t)i * M, M, sh);
        if (lane == 0) sh.W[i] = w;
    }
    for (int k = lane; k < N * N; k += 64) sh.Vt[k] = (k / N == k % N) ? 1.f : 0.f;
    __syncthreads();
    for (int iter = 0; iter < max_iter; iter++) {
        bool changed = false;
        for (int i = 0; i < N - 1; i++)
            for (int j = i + 1; j < N; j++) {
                float* Ai = At + (size_t)i * M;
                float* Aj = At + (size_t)j * M;
                double p = wave_dot_ordered(Ai, Aj, M, sh);
                double a = sh.W[i], b = sh.W[j];
                if (fabs(p) <= eps * sqrt((double)a * b)) continue;  // the same decision on every lane
                p *= 2;
                const double beta = a - b, gamma = hypot_f64((double)p, beta);
                float c, s;
                if (beta < 0) {
                    const double delta = (gamma - beta) * 0.5;
                    s = (float)sqrt(delta / gamma);
                    c = (float)(p / (gamma * s * 2));
                } else {
                    c = (float)sqrt((gamma + beta) / (gamma * 2));
                    s = (float)(p / (gamma * c * 2));
                }
                a = b = 0;
                for (int base = 0; base < M; base += kSvdChunk) {
                    const int cnt = min(kSvdChunk, M - base);
                    for (int k = lane; k < cnt; k += 64) {
                        const float x = Ai[base + k], y = Aj[base + k];
                        const float t0 = c * x + s * y;
                        const float t1 = -s * x + c * y;
                        Ai[base + k] = t0;
                        Aj[base + k] = t1;
                        sh.prod_a[k] = (double)t0 * t0;
                        sh.prod_b[k] = (double)t1 * t1;
                    }
                    __syncthreads();
                    if (lane == 0)
                        for (int k = 0; k < cnt; k++) {
                            a += sh.prod_a[k];
                            b += sh.prod_b[k];
                        }
                    __syncthreads();
                }
                if (lane == 0) {
                    sh.W[i] = a;
                    sh.W[j] = b;
                }
                changed = true;
                if (lane < N) {
                    const float x = sh.Vt[i * N + lane], y = sh.Vt[j * N + lane];
                    sh.Vt[i * N + lane] = c * x + s * y;
                    sh.Vt[j * N + lane] = -s * x + c * y;
                }
                __syncthreads();
            }
        if (!changed) break;
    }
    for (int i = 0; i < N; i++) {
        const double w = wave_dot_ordered(At + (size_t)i * M, At + (size_t)i * M, M, sh);
        if (lane == 0) sh.W[i] = sqrt(w);
    }
    __syncthreads();
    for (int i = 0; i < N - 1; i++) {
        int j = i;
        for (int k = i + 1; k < N; k++)
            if (sh.W[j] < sh.W[k]) j = k;
        __syncthreads();
        if (i != j) {
            if (lane == 0) { const double t = sh.W[i]; sh.W[i] = sh.W[j]; sh.W[j] = t; }
            if (lane < N) { const float t = sh.Vt[i * N + lane]; sh.Vt[i * N + lane] = sh.Vt[j * N + lane]; sh.Vt[j * N + lane] = t; }
            if (sort_rows)
                for (int k = lane; k < M; k += 64) {
                    const float t = At[(size_t)i * M + k];
                    At[(size_t)i * M + k] = At[(size_t)j * M + k];
                    At[(size_t)j * M + k] = t;
                }
        }
        __syncthreads();
    }
    if (!sort_rows || M != 9) return;
    if (lane == 0) {
        // the completion of cv::JacobiSVDImpl_ (n1 = 9 rows of length 9), a few hundred operations: one lane
        const double minval = 1.17549435082228750797e-38;  // FLT_MIN
        unsigned long long rng = 0x12345678ull;
        for (int i = 0; i < 9; i++) {
            double sd = i < N ? sh.W[i] : 0;
            for (int ii = 0; ii < 100 && sd <= minval; ii++) {
                const float val0 = (float)(1. / 9);
                for (int k = 0; k < 9; k++) {
                    rng = (unsigned long long)(unsigned)rng * 4164903690U + (unsigned)(rng >> 32);
                    At[i * 9 + k] = ((unsigned)rng & 256) != 0 ? val0 : -val0;
                }
                for (int it2 = 0; it2 < 2; it2++)
                    for (int j = 0; j < i; j++) {
                        sd = 0;
                        for (int k = 0; k < 9; k++) sd += At[i * 9 + k] * At[j * 9 + k];
                        float asum = 0;
                        for (int k = 0; k < 9; k++) {
                            const float t = (float)(At[i * 9 + k] - sd * At[j * 9 + k]);
                            At[i * 9 + k] = t;
                            asum += fabsf(t);
                        }
                        asum = asum > eps * 100 ? 1 / asum : 0;
                        for (int k = 0; k < 9; k++) At[i * 9 + k] *= asum;
                    }
                sd = 0;
                for (int k = 0; k < 9; k++) {
                    const float t = At[i * 9 + k];
                    sd += (double)t * t;
                }
                sd = sqrt(sd);
            }
            const float s = (float)(sd > minval ? 1 / sd : 0.);
            for (int k = 0; k < 9; k++) At[i * 9 + k] *= s;
        }
    }
    __syncthreads();
}

// geometry::computeFundamentalMat (model 0, utils.cpp:61-103) or computeHomographyMat (model 1, :112-154) on ALL n
// correspondences of a system: grid = systems, one wavefront each; system q reads n points from pts + q*n*2 and
// writes out9 + q*9.  work: 9 * max(rows, 9) floats per system, rows = n or 2n.
__global__ __launch_bounds__(64) void k_solve_npt(const float* __restrict__ pts, const float* __restrict__ ref, int n,
                                                  int model, float* __restrict__ work, float* __restrict__ out9) {
    __shared__ SvdShared sh;
    const int lane = threadIdx.x, q = blockIdx.x;
    const float2* p2 = (const float2*)(pts + (size_t)q * n * 2);
    const float2* r2 = (const float2*)(ref + (size_t)q * n * 2);
    const int rows = model == 0 ? n : 2 * n;
    // cv::SVDecomp works on the matrix whose rows are the SHORTER dimension: with 9 or more equations the system
    // is transposed (9 rows of length `rows`) and vt is the accumulated Vt; with fewer, the equations are the rows
    // (length 9), vt is the row set itself, completed to 9 rows
    const bool transposed = rows >= 9;
    const int M = transposed ? rows : 9, N = transposed ? 9 : rows;
    float* At = work + (size_t)q * 9 * (rows > 9 ? rows : 9);
    if (!transposed)
        for (int k = lane; k < 81; k += 64) At[k] = 0.f;
    __syncthreads();
    for (int i = lane; i < n; i += 64) {
        const float u1 = r2[i].x, v1 = r2[i].y, u2 = p2[i].x, v2 = p2[i].y;
        float e[2][9];
        if (model == 0) {
            e[0][0] = u2 * u1; e[0][1] = u2 * v1; e[0][2] = u2;
            e[0][3] = v2 * u1; e[0][4] = v2 * v1; e[0][5] = v2;
            e[0][6] = u1; e[0][7] = v1; e[0][8] = 1;
        } else {
            e[0][0] = 0.0f; e[0][1] = 0.0f; e[0][2] = 0.0f; e[0][3] = -u1; e[0][4] = -v1; e[0][5] = -1;
            e[0][6] = v2 * u1; e[0][7] = v2 * v1; e[0][8] = v2;
            e[1][0] = u1; e[1][1] = v1; e[1][2] = 1; e[1][3] = 0.0f; e[1][4] = 0.0f; e[1][5] = 0.0f;
            e[1][6] = -u2 * u1; e[1][7] = -u2 * v1; e[1][8] = -u2;
        }
        const int per = model == 0 ? 1 : 2;
        for (int r = 0; r < per; r++) {
            const int row = i * per + r;
#pragma unroll
            for (int cc = 0; cc < 9; cc++) {
                if (transposed) At[(size_t)cc * M + row] = e[r][cc];
                else At[row * 9 + cc] = e[r][cc];
            }
        }
    }
    __syncthreads();
    svd_rows_wave(At, M, N, !transposed, sh);
    if (lane != 0) return;
    float v8[9];
    for (int k = 0; k < 9; k++) v8[k] = transposed ? sh.Vt[8 * 9 + k] : At[8 * 9 + k];  // vt.row(8)
    float* out = out9 + (size_t)q * 9;
    if (model == 1) {
        for (int k = 0; k < 9; k++) out[k] = v8[k];
        return;
    }
    // rank-2 enforcement: SVD of the 3 x 3 matrix (worked on as its transpose), smallest singular value dropped
    float B[9], V3[9];
    double W3[3];
    int perm3[3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) B[j * 3 + i] = v8[i * 3 + j];
    jacobi_sweeps<3, 3, true>(B, V3, W3);
    svd_finish<3, 3, true, true>(B, V3, W3, perm3);
    svd_complete<3, 3, 3>(B, W3);
    const float w0 = (float)W3[0], w1 = (float)W3[1];
    float U[9], D[9] = {w0, 0, 0, 0, w1, 0, 0, 0, 0.f}, UD[9], Fn[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) U[j * 3 + i] = B[i * 3 + j];
    mul33(U, D, UD);
    mul33(UD, V3, Fn);
    for (int k = 0; k < 9; k++) out[k] = Fn[k];
}

// ---- scoring ---------------------------------------------------------------------------------------------------
// Initializer::checkFundamentalScore / checkHomographyScore (Initializer.cpp:369-444, 501-585) add one term per
// correspondence and side to a running score, in index order; that chain of additions is the only part whose order
// matters.  The terms themselves (two point-to-line or transfer distances, two IEEE divides: nearly all of the
// arithmetic) are independent.  A workgroup scores 64 hypotheses (lane = hypothesis).  Its four wavefronts each
// evaluate the terms of a quarter of a chunk of kScoreChunk correspondences for those 64 hypotheses and park them in
// LDS; after one barrier a single wavefront - they take turns, the running scores travel through LDS - adds the
// chunk's terms in index order while the other three already work on the next chunk (two term buffers).  Four times
// as many wavefronts as hypotheses / 64, so the SIMDs are evenly loaded whatever pairs x iterations is.
// A term that the reference skips (chi-square above the threshold) is parked as +0.0: the running score is never
// negative zero, so adding it changes nothing; a NaN distance is not "above" and is added, as in the reference.
// Hypotheses of all pairs are numbered through (hyp = pair * iterations + iteration): with iterations >= 63 the 64
// hypotheses of a workgroup lie in at most two pairs.  The correspondences are the same for all lanes of a pair: a
// wavefront whose lanes share one pair reads them through the scalar path.
#ifndef ORBX_SCORE_CHUNK_H
#define ORBX_SCORE_CHUNK_H 16
#endif
#ifndef ORBX_SCORE_CHUNK_F
#define ORBX_SCORE_CHUNK_F 16
#endif
constexpr int kScoreChunk = ORBX_SCORE_CHUNK_H;   // correspondences per chunk of the homography scorer (a quarter per wavefront)
constexpr int kScoreChunkF = ORBX_SCORE_CHUNK_F;  // the same for the fundamental scorer (its terms are doubles: twice the LDS)
constexpr int kScoreMinIterations = 63;

struct ScoreLane {
    int pair, it;
    bool live;
};

__device__ __forceinline__ ScoreLane score_lane(int iterations, int pairs) {
    ScoreLane sl;
    const long long total = (long long)pairs * iterations;
    const long long hyp = (long long)blockIdx.x * 64 + (threadIdx.x & 63);
    sl.live = hyp < total;
    const long long hc = sl.live ? hyp : total - 1;
    sl.pair = (int)(hc / iterations);
    sl.it = (int)(hc - (long long)sl.pair * iterations);
    return sl;
}

// The correspondences of a lane's pair.  UNIFORM: every lane of the wavefront has the same pair, whose index sits
// in a scalar register - the compiler then sees addresses that do not depend on the lane and reads the points through
// the scalar cache.  Otherwise each lane reads through its own pair's pointers.
template <bool UNIFORM>
struct PairPoints {
    const float2* p2;
    const float2* r2;
    __device__ __forceinline__ PairPoints(const float* pts, const float* ref, int n, int pair) {
        const int use = UNIFORM ? __builtin_amdgcn_readfirstlane(pair) : pair;
        p2 = (const float2*)(pts + (size_t)use * n * 2);
        r2 = (const float2*)(ref + (size_t)use * n * 2);
    }
};

// K12F: Initializer::checkFundamentalScore (Initializer.cpp:369-444); fp64.  pd: the correspondences widened to
// double by k_normalize, (u1, v1, u2, v2) per record.  f11 * u1 and its like are products of two float-valued
// doubles: exact, so `f11 * u1 + f12 * v1` is evaluated as one fused operation on the exact first product (see
// fma_exact_product); the remaining operations involve full doubles and stay un-fused as in the reference.
__global__ __launch_bounds__(256) void k_score_f(const double* __restrict__ pd, int n,
                                                 const float* __restrict__ F_all, int iterations, int pairs,
                                                 double sigma, float* __restrict__ score_out) {
    __shared__ double2 s_term[2][kScoreChunkF][64];
    __shared__ double s_score[64];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const ScoreLane sl = score_lane(iterations, pairs);
    const float* Fp = F_all + ((size_t)sl.pair * iterations + sl.it) * 9;
    const double f11 = Fp[0], f12 = Fp[1], f13 = Fp[2], f21 = Fp[3], f22 = Fp[4], f23 = Fp[5], f31 = Fp[6],
                 f32 = Fp[7], f33 = Fp[8];
    const double th = 3.841, th_score = 5.991;
    const double inv_sigma_square = 1.0 / (sigma * sigma);
    if (wv == 0) s_score[lane] = 0.0;
    constexpr int kPer = kScoreChunkF / 4;
    // the two terms of one correspondence: (u1, v1) in the reference frame, (u2, v2) in the current frame
    auto term = [&](double4 q) {
        const double u1 = q.x, v1 = q.y, u2 = q.z, v2 = q.w;
        const double a2 = __builtin_fma(f12, v1, f11 * u1) + f13;
        const double b2 = __builtin_fma(f22, v1, f21 * u1) + f23;
        const double c2 = __builtin_fma(f32, v1, f31 * u1) + f33;
        const double num2 = a2 * u2 + b2 * v2 + c2;
        const double sd1 = num2 * num2 / (a2 * a2 + b2 * b2);
        const double chi1 = sd1 * inv_sigma_square;
        const double a1 = __builtin_fma(f21, v2, f11 * u2) + f31;
        const double b1 = __builtin_fma(f22, v2, f12 * u2) + f32;
        const double c1 = __builtin_fma(f23, v2, f13 * u2) + f33;
        const double num1 = a1 * u1 + b1 * v1 + c1;
        const double sd2 = num1 * num1 / (a1 * a1 + b1 * b1);
        const double chi2 = sd2 * inv_sigma_square;
        return make_double2(chi1 > th ? 0.0 : th_score - chi1, chi2 > th ? 0.0 : th_score - chi2);
    };
    auto body = [&](auto uniform_tag) {
        // UNIFORM: every lane has the same pair, whose index sits in a scalar register - the addresses then do not
        // depend on the lane and the records come through the scalar cache
        const int use = decltype(uniform_tag)::value ? __builtin_amdgcn_readfirstlane(sl.pair) : sl.pair;
        const double4* rec = (const double4*)(pd + (size_t)use * n * 4);
        int chunk = 0;
        for (int base = 0; base < n; base += kScoreChunkF, chunk++) {
            const int buf = chunk & 1;
            const int c0 = base + wv * kPer;
            if (base + kScoreChunkF <= n) {
                // a whole chunk: the wavefront's records are fetched together, then evaluated
                double4 q[kPer];
#pragma unroll
                for (int j = 0; j < kPer; j++) q[j] = rec[c0 + j];
#pragma unroll
                for (int j = 0; j < kPer; j++) s_term[buf][wv * kPer + j][lane] = term(q[j]);
            } else {
#pragma unroll
                for (int j = 0; j < kPer; j++)
                    if (c0 + j < n) s_term[buf][wv * kPer + j][lane] = term(rec[c0 + j]);
            }
            __syncthreads();
            // this chunk's additions: the wavefronts take turns; the one whose turn it is finishes before it starts
            // on the next chunk, and so before the next barrier, after which the following wavefront takes over
            if (wv == (chunk & 3)) {
                double score = s_score[lane];
                const int cnt = min(kScoreChunkF, n - base);
                for (int cc = 0; cc < cnt; cc++) {
                    const double2 t = s_term[buf][cc][lane];
                    score += t.x;
                    score += t.y;
                }
                s_score[lane] = score;
            }
        }
    };
    // every wavefront of the workgroup meets the same barriers whichever form it takes: the trip counts are equal
    if (__all(sl.pair == __builtin_amdgcn_readfirstlane(sl.pair))) body(std::true_type{});
    else body(std::false_type{});
    __syncthreads();
    if (wv == 0 && sl.live) score_out[(size_t)sl.pair * iterations + sl.it] = (float)s_score[lane];
}

// 1.0f / x, correctly rounded.  v_rcp_f32 followed by one fused correction step gives exactly that whenever
// 2^-120 <= |x| <= 2^120 (checked on gfx950 for every one of the 2^32 bit patterns, tools/rcp_exact.hip): three
// instructions instead of the ten of the general division.  The short form is taken only when every lane of the
// wavefront is inside that range (one test per wavefront, no divergence); otherwise all lanes divide.
#ifndef ORBX_H_RCP
#define ORBX_H_RCP 1
#endif
__device__ __forceinline__ float recip_ieee(float x) {
#if ORBX_H_RCP
    const unsigned ab = __float_as_uint(x) & 0x7fffffffu;
    if (__all(ab - 0x03800000u <= 0x7b800000u - 0x03800000u)) {
        const float r = __builtin_amdgcn_rcpf(x);
        const float e = __builtin_fmaf(-x, r, 1.0f);
        return __builtin_fmaf(r, e, r);
    }
#endif
    return 1.0f / x;
}

// K12H: Initializer::checkHomographyScore (Initializer.cpp:501-585); fp32.  The reference writes the reciprocal as
// `1.0 / (float expression)` narrowed to float: a double division rounded twice.  The correctly rounded float
// division used here gives the same bits for every input: 1/x of a 24-bit x lies at least 2^-49 (relative) away
// from any midpoint between two floats (x * midpoint is an integer of at most 49 bits and never exactly a power of
// two unless 1/x is exact), far more than the 2^-53 the first rounding can move it, so the second rounding never
// changes side; zeros, infinities and out-of-range quotients behave alike in both forms.
__global__ __launch_bounds__(256) void k_score_h(const float* __restrict__ pts, const float* __restrict__ ref, int n,
                                                 const float* __restrict__ H_all, const float* __restrict__ Hinv_all,
                                                 int iterations, int pairs, double sigma,
                                                 float* __restrict__ score_out) {
    __shared__ float2 s_term[2][kScoreChunk][64];
    __shared__ float s_score[64];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const ScoreLane sl = score_lane(iterations, pairs);
    const float* Hp = H_all + ((size_t)sl.pair * iterations + sl.it) * 9;
    const float* Ip = Hinv_all + ((size_t)sl.pair * iterations + sl.it) * 9;
    const float h11 = Hp[0], h12 = Hp[1], h13 = Hp[2], h21 = Hp[3], h22 = Hp[4], h23 = Hp[5], h31 = Hp[6], h32 = Hp[7],
                h33 = Hp[8];
    const float i11 = Ip[0], i12 = Ip[1], i13 = Ip[2], i21 = Ip[3], i22 = Ip[4], i23 = Ip[5], i31 = Ip[6], i32 = Ip[7],
                i33 = Ip[8];
    const float th = 5.991;
    const float invSigmaSquare = 1.0 / (sigma * sigma);
    if (wv == 0) s_score[lane] = 0.f;
    constexpr int kPer = kScoreChunk / 4;
    auto term = [&](float2 rp, float2 cp) {
        const float u1 = rp.x, v1 = rp.y, u2 = cp.x, v2 = cp.y;
        const float w2in1inv = recip_ieee(i31 * u2 + i32 * v2 + i33);
        const float u2in1 = (i11 * u2 + i12 * v2 + i13) * w2in1inv;
        const float v2in1 = (i21 * u2 + i22 * v2 + i23) * w2in1inv;
        const float sq1 = (u1 - u2in1) * (u1 - u2in1) + (v1 - v2in1) * (v1 - v2in1);
        const float chi1 = sq1 * invSigmaSquare;
        const float w1in2inv = recip_ieee(h31 * u1 + h32 * v1 + h33);
        const float u1in2 = (h11 * u1 + h12 * v1 + h13) * w1in2inv;
        const float v1in2 = (h21 * u1 + h22 * v1 + h23) * w1in2inv;
        const float sq2 = (u2 - u1in2) * (u2 - u1in2) + (v2 - v1in2) * (v2 - v1in2);
        const float chi2 = sq2 * invSigmaSquare;
        return make_float2(chi1 > th ? 0.f : th - chi1, chi2 > th ? 0.f : th - chi2);
    };
    auto body = [&](auto uniform_tag) {
        const PairPoints<decltype(uniform_tag)::value> pp(pts, ref, n, sl.pair);
        int chunk = 0;
        for (int base = 0; base < n; base += kScoreChunk, chunk++) {
            const int buf = chunk & 1;
            const int c0 = base + wv * kPer;
            if (base + kScoreChunk <= n) {
                // a whole chunk: the wavefront's points are fetched together, then evaluated
                float2 rp[kPer], cp[kPer];
#pragma unroll
                for (int j = 0; j < kPer; j++) { rp[j] = pp.r2[c0 + j]; cp[j] = pp.p2[c0 + j]; }
#pragma unroll
                for (int j = 0; j < kPer; j++) s_term[buf][wv * kPer + j][lane] = term(rp[j], cp[j]);
            } else {
#pragma unroll
                for (int j = 0; j < kPer; j++)
                    if (c0 + j < n) s_term[buf][wv * kPer + j][lane] = term(pp.r2[c0 + j], pp.p2[c0 + j]);
            }
            __syncthreads();
            if (wv == (chunk & 3)) {
                float score = s_score[lane];
                const int cnt = min(kScoreChunk, n - base);
                for (int cc = 0; cc < cnt; cc++) {
                    const float2 t = s_term[buf][cc][lane];
                    score += t.x;
                    score += t.y;
                }
                s_score[lane] = score;
            }
        }
    };
    if (__all(sl.pair == __builtin_amdgcn_readfirstlane(sl.pair))) body(std::true_type{});
    else body(std::false_type{});
    __syncthreads();
    if (wv == 0 && sl.live) score_out[(size_t)sl.pair * iterations + sl.it] = s_score[lane];
}

// Winner selection and inlier flags (best_out: winning iteration, -1 when no hypothesis scored above 0, -2 when
// one of the pair's sets held an index >= n).  The reference keeps a hypothesis when (float)score > best (strict), best
// starting at 0: the winner is the earliest iteration that reaches the maximum, provided the maximum is > 0
// (Initializer.cpp:318,347-355,458,488-497).  grid = (pairs, 2): y = 0 fundamental, y = 1 homography.
__global__ __launch_bounds__(256) void k_select(const float* __restrict__ pts, const float* __restrict__ ref, int n,
                                                const float* __restrict__ F_all, const float* __restrict__ H_all,
                                                const float* __restrict__ Hinv_all,
                                                const float* __restrict__ score_f, const float* __restrict__ score_h,
                                                const int32_t* __restrict__ bad_sets, int iterations, double sigma,
                                                float* __restrict__ F_out,
                                                float* __restrict__ H_out, uint8_t* __restrict__ inl_f,
                                                uint8_t* __restrict__ inl_h, float* __restrict__ scores_out,
                                                int32_t* __restrict__ best_out) {
    __shared__ float s_val[256];
    __shared__ int s_it[256];
    const int pair = blockIdx.x, model = blockIdx.y, tid = threadIdx.x;
    const float* sc = (model == 0 ? score_f : score_h) + (size_t)pair * iterations;
    float bv = 0.f;
    int bi = -1;
    for (int i = tid; i < iterations; i += 256) {
        const float v = sc[i];
        if (v > bv) { bv = v; bi = i; }  // strict: the earliest of this thread's equal maxima stays
    }
    s_val[tid] = bv;
    s_it[tid] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            const float v2 = s_val[tid + o];
            const int i2 = s_it[tid + o];
            const float v1 = s_val[tid];
            const int i1 = s_it[tid];
            if (i2 >= 0 && (i1 < 0 || v2 > v1 || (v2 == v1 && i2 < i1))) { s_val[tid] = v2; s_it[tid] = i2; }
        }
        __syncthreads();
    }
    // a pair whose sets were invalid has no winner: best iteration -2, score 0, no inliers
    const int best = bad_sets[pair] ? -2 : s_it[0];
    const float best_score = s_val[0];
    if (tid == 0) {
        best_out[pair * 2 + model] = best;
        scores_out[pair * 2 + model] = best >= 0 ? best_score : 0.f;
    }
    uint8_t* inl = (model == 0 ? inl_f : inl_h) + (size_t)pair * n;
    if (best < 0) {
        for (int i = tid; i < n; i += 256) inl[i] = 0;
        return;
    }
    const float2* p2 = (const float2*)(pts + (size_t)pair * n * 2);
    const float2* r2 = (const float2*)(ref + (size_t)pair * n * 2);
    const size_t o = ((size_t)pair * iterations + best) * 9;
    if (model == 0) {
        const float* Fp = F_all + o;
        if (tid < 9) F_out[(size_t)pair * 9 + tid] = Fp[tid];
        const double f11 = Fp[0], f12 = Fp[1], f13 = Fp[2], f21 = Fp[3], f22 = Fp[4], f23 = Fp[5], f31 = Fp[6],
                     f32 = Fp[7], f33 = Fp[8];
        const double th = 3.841;
        const double inv_sigma_square = 1.0 / (sigma * sigma);
        for (int i = tid; i < n; i += 256) {
            const float u1 = r2[i].x, v1 = r2[i].y, u2 = p2[i].x, v2 = p2[i].y;
            bool good = true;
            const double a2 = f11 * u1 + f12 * v1 + f13;
            const double b2 = f21 * u1 + f22 * v1 + f23;
            const double c2 = f31 * u1 + f32 * v1 + f33;
            const double num2 = a2 * u2 + b2 * v2 + c2;
            const double chi1 = num2 * num2 / (a2 * a2 + b2 * b2) * inv_sigma_square;
            if (chi1 > th) good = false;
            const double a1 = f11 * u2 + f21 * v2 + f31;
            const double b1 = f12 * u2 + f22 * v2 + f32;
            const double c1 = f13 * u2 + f23 * v2 + f33;
            const double num1 = a1 * u1 + b1 * v1 + c1;
            const double chi2 = num1 * num1 / (a1 * a1 + b1 * b1) * inv_sigma_square;
            if (chi2 > th) good = false;
            inl[i] = good;
        }
    } else {
        const float* Hp = H_all + o;
        const float* Ip = Hinv_all + o;
        if (tid < 9) H_out[(size_t)pair * 9 + tid] = Hp[tid];
        const float h11 = Hp[0], h12 = Hp[1], h13 = Hp[2], h21 = Hp[3], h22 = Hp[4], h23 = Hp[5], h31 = Hp[6],
                    h32 = Hp[7], h33 = Hp[8];
        const float i11 = Ip[0], i12 = Ip[1], i13 = Ip[2], i21 = Ip[3], i22 = Ip[4], i23 = Ip[5], i31 = Ip[6],
                    i32 = Ip[7], i33 = Ip[8];
        const float th = 5.991;
        const float invSigmaSquare = 1.0 / (sigma * sigma);
        for (int i = tid; i < n; i += 256) {
            const float u1 = r2[i].x, v1 = r2[i].y, u2 = p2[i].x, v2 = p2[i].y;
            bool good = true;
            const float w2in1inv = 1.0 / (i31 * u2 + i32 * v2 + i33);
            const float u2in1 = (i11 * u2 + i12 * v2 + i13) * w2in1inv;
            const float v2in1 = (i21 * u2 + i22 * v2 + i23) * w2in1inv;
            const float sq1 = (u1 - u2in1) * (u1 - u2in1) + (v1 - v2in1) * (v1 - v2in1);
            if (sq1 * invSigmaSquare > th) good = false;
            const float w1in2inv = 1.0 / (h31 * u1 + h32 * v1 + h33);
            const float u1in2 = (h11 * u1 + h12 * v1 + h13) * w1in2inv;
            const float v1in2 = (h21 * u1 + h22 * v1 + h23) * w1in2inv;
            const float sq2 = (u2 - u1in2) * (u2 - u1in2) + (v2 - v1in2) * (v2 - v1in2);
            if (sq2 * invSigmaSquare > th) good = false;
            inl[i] = good;
        }
    }
}

}  // namespace orbx
