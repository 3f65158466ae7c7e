// Pose recovery after the F/H RANSAC (gfx950): Initializer::findRtWithFundamental and Initializer::triangulatePoints
// (reference src/Initializer.cpp:165-252), i.e. cv::recoverPose / cv::decomposeEssentialMat / cv::triangulatePoints
// of OpenCV 3.4 in double precision, batched over frame pairs.  One LANE owns one correspondence (a 4 x 4 Jacobi SVD
// per candidate pose, in registers) or one pair (the 3 x 3 SVD of the essential matrix), so every sum keeps the
// scalar order.  Must be compiled with -ffp-contract=off.
#pragma once

#include "ransac_kernels.h"

namespace orbx {

// cv::JacobiSVDImpl_<double> for a square N x N system held as the N rows of At (OpenCV 3.4 core/lapack.cpp), with
// V^T in Vt: sweeps, singular values, descending selection sort (rows of At and Vt follow) and the completion of
// rows with a vanishing singular value from cv::RNG(0x12345678) + Gram-Schmidt, then normalisation of every row.
// All indices are compile-time constants, so the three arrays live in registers.
template <int N>
__device__ __forceinline__ void jacobi_square_f64(double (&At)[N * N], double (&Vt)[N * N], double (&W)[N]) {
    const double eps = 2.220446049250313e-16 * 10;     // DBL_EPSILON * 10
    const double minval = 2.2250738585072014e-308;     // DBL_MIN
    const int max_iter = 30;
#pragma unroll
    for (int i = 0; i < N; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < N; k++) sd += At[i * N + k] * At[i * N + k];
        W[i] = sd;
#pragma unroll
        for (int k = 0; k < N; k++) Vt[i * N + k] = k == i ? 1.0 : 0.0;
    }
    for (int iter = 0; iter < max_iter; iter++) {
        bool changed = false;
#pragma unroll
        for (int i = 0; i < N - 1; i++) {
#pragma unroll
            for (int j = i + 1; j < N; j++) {
                double a = W[i], p = 0, b = W[j];
#pragma unroll
                for (int k = 0; k < N; k++) p += At[i * N + k] * At[j * N + k];
                if (fabs(p) <= eps * sqrt(a * b)) continue;
                p *= 2;
                const double beta = a - b, gamma = hypot_f64(p, beta);
                double c, s;
                if (beta < 0) {
                    const double delta = (gamma - beta) * 0.5;
                    s = sqrt(delta / gamma);
                    c = p / (gamma * s * 2);
                } else {
                    c = sqrt((gamma + beta) / (gamma * 2));
                    s = p / (gamma * c * 2);
                }
                a = b = 0;
#pragma unroll
                for (int k = 0; k < N; k++) {
                    const double t0 = c * At[i * N + k] + s * At[j * N + k];
                    const double t1 = -s * At[i * N + k] + c * At[j * N + k];
                    At[i * N + k] = t0;
                    At[j * N + k] = t1;
                    a += t0 * t0;
                    b += t1 * t1;
                }
                W[i] = a;
                W[j] = b;
                changed = true;
#pragma unroll
                for (int k = 0; k < N; k++) {
                    const double t0 = c * Vt[i * N + k] + s * Vt[j * N + k];
                    const double t1 = -s * Vt[i * N + k] + c * Vt[j * N + k];
                    Vt[i * N + k] = t0;
                    Vt[j * N + k] = t1;
                }
            }
        }
        if (!changed) break;
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < N; k++) sd += At[i * N + k] * At[i * N + k];
        W[i] = sqrt(sd);
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
#pragma unroll
            for (int q = 0; q < N; q++) {
                const double ta = At[i * N + q]; At[i * N + q] = At[k * N + q]; At[k * N + q] = ta;
                const double tv = Vt[i * N + q]; Vt[i * N + q] = Vt[k * N + q]; Vt[k * N + q] = tv;
            }
        }
    }
    unsigned long long rng = 0x12345678ull;
#pragma unroll
    for (int i = 0; i < N; i++) {
        double sd = W[i];
        for (int ii = 0; ii < 100 && sd <= minval; ii++) {
            const double val0 = 1. / N;
#pragma unroll
            for (int k = 0; k < N; k++) {
                rng = (unsigned long long)(unsigned)rng * 4164903690U + (unsigned)(rng >> 32);
                At[i * N + k] = ((unsigned)rng & 256) != 0 ? val0 : -val0;
            }
            for (int iter = 0; iter < 2; iter++) {
#pragma unroll
                for (int j = 0; j < i; j++) {
                    sd = 0;
#pragma unroll
                    for (int k = 0; k < N; k++) sd += At[i * N + k] * At[j * N + k];
                    double asum = 0;
#pragma unroll
                    for (int k = 0; k < N; k++) {
                        const double t = At[i * N + k] - sd * At[j * N + k];
                        At[i * N + k] = t;
                        asum += fabs(t);
                    }
                    asum = asum > eps * 100 ? 1 / asum : 0;
#pragma unroll
                    for (int k = 0; k < N; k++) At[i * N + k] *= asum;
                }
            }
            sd = 0;
#pragma unroll
            for (int k = 0; k < N; k++) sd += At[i * N + k] * At[i * N + k];
            sd = sqrt(sd);
        }
        const double s = sd > minval ? 1 / sd : 0.;
#pragma unroll
        for (int k = 0; k < N; k++) At[i * N + k] *= s;
    }
}

__device__ __forceinline__ double det3_f64(const double* m) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

// cv::gemm on CV_64F 3x3: sums in ascending k starting from 0.
__device__ __forceinline__ void mul33_f64(const double* A, const double* B, double* C) {
    double out[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) s += A[i * 3 + k] * B[k * 3 + j];
            out[i * 3 + j] = s;
        }
#pragma unroll
    for (int i = 0; i < 9; i++) C[i] = out[i];
}

// One correspondence of cv::triangulatePoints (calib3d triangulate.cpp): the 4 x 4 system of the two views, its SVD,
// the last row of V^T.  P1, P2 are 3 x 4 row-major.
__device__ __forceinline__ void triangulate_one(const double* P1, const double* P2, double x1, double y1, double x2,
                                                double y2, double (&q)[4]) {
    // the SVD front end works on the transpose: row r of At is column r of the system
    double At[16], Vt[16], W[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        At[k * 4 + 0] = x1 * P1[8 + k] - P1[k];
        At[k * 4 + 1] = y1 * P1[8 + k] - P1[4 + k];
        At[k * 4 + 2] = x2 * P2[8 + k] - P2[k];
        At[k * 4 + 3] = y2 * P2[8 + k] - P2[4 + k];
    }
    jacobi_square_f64<4>(At, Vt, W);
#pragma unroll
    for (int k = 0; k < 4; k++) q[k] = Vt[12 + k];
}

// Per pair: E = K^T F K as the reference forms it in CV_32F, cv::decomposeEssentialMat -> R1, R2, t.
// cand[pair] = R1 (9), R2 (9), t (3); norm[pair] = 1/f, -cx/f, -cy/f of cv::recoverPose's point normalisation.
__global__ __launch_bounds__(64) void k_pose_candidates(const float* __restrict__ F_all, const float* __restrict__ K9,
                                                        int pairs, double* __restrict__ cand, double* __restrict__ norm) {
    const int pair = blockIdx.x * 64 + threadIdx.x;
    if (pair >= pairs) return;
    const float* F = F_all + (size_t)pair * 9;
    float K[9];
#pragma unroll
    for (int i = 0; i < 9; i++) K[i] = K9[i];
    // K.t() * F carries a transpose flag: cv::gemm's general path, double sums rounded to float once;
    // (K^T F) * K has none: the 3 x 3 float path, left to right
    float KtF[9], Ef[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) s += (double)K[k * 3 + i] * (double)F[k * 3 + j];
            KtF[i * 3 + j] = (float)s;
        }
    mul33(KtF, K, Ef);
    // SVD front end: the work array holds E^T, its rows end up as the rows of U^T
    double At[9], Vt[9], W[3], U[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) At[j * 3 + i] = (double)Ef[i * 3 + j];
    jacobi_square_f64<3>(At, Vt, W);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) U[j * 3 + i] = At[i * 3 + j];
    if (det3_f64(U) < 0) {
#pragma unroll
        for (int i = 0; i < 9; i++) U[i] = U[i] * -1. + 0.;
    }
    if (det3_f64(Vt) < 0) {
#pragma unroll
        for (int i = 0; i < 9; i++) Vt[i] = Vt[i] * -1. + 0.;
    }
    const double Wm[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1}, Wt[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1};
    double UW[9], R1[9], R2[9];
    mul33_f64(U, Wm, UW);
    mul33_f64(UW, Vt, R1);
    mul33_f64(U, Wt, UW);
    mul33_f64(UW, Vt, R2);
    double* c = cand + (size_t)pair * 21;
#pragma unroll
    for (int i = 0; i < 9; i++) { c[i] = R1[i]; c[9 + i] = R2[i]; }
#pragma unroll
    for (int i = 0; i < 3; i++) c[18 + i] = U[i * 3 + 2] * 1.0;
    if (pair == 0) {
        const double focal = (double)(K[0] + K[4]) / 2.0;  // (focalX() + focalY()) / 2.0, Initializer.cpp:177-178
        const double inv = 1. / focal;
        norm[0] = inv;
        norm[1] = -(double)K[2] * inv;
        norm[2] = -(double)K[5] * inv;
    }
}

// cv::recoverPose's cheirality test: every correspondence is triangulated against each of the four candidate poses
// and counted when it lies in front of both cameras and nearer than 50.  points1 = reference points.
__global__ __launch_bounds__(256) void k_pose_cheirality(const double* __restrict__ pts, const double* __restrict__ ref,
                                                         const int32_t* __restrict__ pair_start,
                                                         const double* __restrict__ cand,
                                                         const double* __restrict__ norm, int32_t* __restrict__ good) {
    const int pair = blockIdx.y;
    const int first = pair_start[pair], n = pair_start[pair + 1] - first;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 >= n) return;
    const bool live = i < n;
    const double inv = norm[0], bx = norm[1], by = norm[2];
    double x1 = 0, y1 = 0, x2 = 0, y2 = 0;
    if (live) {
        const size_t o = 2 * (size_t)(first + i);
        x1 = ref[o] * inv + bx;
        y1 = ref[o + 1] * inv + by;
        x2 = pts[o] * inv + bx;
        y2 = pts[o + 1] * inv + by;
    }
    const double* c = cand + (size_t)pair * 21;
    const double P0[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    const double dist = 50.0;
#pragma unroll 1
    for (int k = 0; k < 4; k++) {
        const double* R = c + ((k & 1) ? 9 : 0);
        double P[12];
#pragma unroll
        for (int r = 0; r < 3; r++) {
#pragma unroll
            for (int q = 0; q < 3; q++) P[r * 4 + q] = R[r * 3 + q];
            P[r * 4 + 3] = k >= 2 ? -c[18 + r] : c[18 + r];
        }
        bool m = false;
        if (live) {
            double q[4];
            triangulate_one(P0, P, x1, y1, x2, y2, q);
            m = q[2] * q[3] > 0;
            const double w = q[3];
            q[0] /= w; q[1] /= w; q[2] /= w; q[3] /= w;
            m = (q[2] < dist) && m;
            double z = 0;
#pragma unroll
            for (int t = 0; t < 4; t++) z += P[8 + t] * q[t];
            m = (z > 0) && m;
            m = (z < dist) && m;
        }
        const unsigned long long bal = __ballot(m);
        if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&good[pair * 4 + k], __popcll(bal));
    }
}

// The choice among the four candidates, cv::recoverPose's own chain of >= comparisons.
__global__ __launch_bounds__(64) void k_pose_select(const double* __restrict__ cand, const int32_t* __restrict__ good,
                                                    int pairs, double* __restrict__ R, double* __restrict__ t,
                                                    int32_t* __restrict__ count) {
    const int pair = blockIdx.x * 64 + threadIdx.x;
    if (pair >= pairs) return;
    const int g0 = good[pair * 4], g1 = good[pair * 4 + 1], g2 = good[pair * 4 + 2], g3 = good[pair * 4 + 3];
    int pick;
    if (g0 >= g1 && g0 >= g2 && g0 >= g3) pick = 0;
    else if (g1 >= g0 && g1 >= g2 && g1 >= g3) pick = 1;
    else if (g2 >= g0 && g2 >= g1 && g2 >= g3) pick = 2;
    else pick = 3;
    const double* c = cand + (size_t)pair * 21;
    for (int i = 0; i < 9; i++) R[(size_t)pair * 9 + i] = c[((pick & 1) ? 9 : 0) + i];
    for (int i = 0; i < 3; i++) t[(size_t)pair * 3 + i] = pick >= 2 ? -c[18 + i] : c[18 + i];
    count[pair] = pick == 0 ? g0 : pick == 1 ? g1 : pick == 2 ? g2 : g3;
}

// Initializer::triangulatePoints (Initializer.cpp:201-252): projection = K [R | t], projection_ref = K [I | 0] in
// double, cv::triangulatePoints(projection, projection_ref, points, ref_points), x * (1 / w).
__global__ __launch_bounds__(256) void k_triangulate_initial(const float* __restrict__ K9, const double* __restrict__ R_all,
                                                             const double* __restrict__ t_all,
                                                             const double* __restrict__ pts, const double* __restrict__ ref,
                                                             const int32_t* __restrict__ pair_start,
                                                             double* __restrict__ out3) {
    const int pair = blockIdx.y;
    const int first = pair_start[pair], n = pair_start[pair + 1] - first;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double K[9], P[12], Pref[12];
#pragma unroll
    for (int k = 0; k < 9; k++) K[k] = (double)K9[k];
    const double* R = R_all + (size_t)pair * 9;
    const double* t = t_all + (size_t)pair * 3;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 4; c++) {
            double s = 0, sr = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                s += K[r * 3 + k] * (c < 3 ? R[k * 3 + c] : t[k]);
                sr += K[r * 3 + k] * (c == k ? 1.0 : 0.0);
            }
            P[r * 4 + c] = s;
            Pref[r * 4 + c] = sr;
        }
    const size_t o = 2 * (size_t)(first + i);
    double q[4];
    triangulate_one(P, Pref, pts[o], pts[o + 1], ref[o], ref[o + 1], q);
    const double alpha = 1. / q[3];
    double* dst = out3 + 3 * (size_t)(first + i);
    dst[0] = q[0] * alpha + 0.;
    dst[1] = q[1] * alpha + 0.;
    dst[2] = q[2] * alpha + 0.;
}

}  // namespace orbx
