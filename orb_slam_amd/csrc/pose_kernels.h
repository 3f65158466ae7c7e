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

// HomographyDecompInria::oppositeOfMinor (OpenCV 3.4 calib3d homography_decomp.cpp) with compile-time row / column.
template <int ROW, int COL>
__device__ __forceinline__ double opposite_of_minor(const double (&M)[9]) {
    constexpr int x1 = COL == 0 ? 1 : 0, x2 = COL == 2 ? 1 : 2;
    constexpr int y1 = ROW == 0 ? 1 : 0, y2 = ROW == 2 ? 1 : 2;
    return M[y1 * 3 + x2] * M[y2 * 3 + x1] - M[y1 * 3 + x1] * M[y2 * 3 + x2];
}

// HomographyDecompInria::findRmatFrom_tstar_n: R = Hnorm (I - (2 / v) tstar n^T), negated when det(R) < 0.
__device__ __forceinline__ void rmat_from_tstar_n(const double (&Hn)[9], const double (&tstar)[3], const double (&n)[3],
                                                  double v, double (&R)[9]) {
    const double alpha = 2 / v;
    double M[9];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const double ts = tstar[i] * alpha;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            double s = 0;
            s += ts * n[j];
            M[i * 3 + j] = (i == j ? 1.0 : 0.0) - s;
        }
    }
    mul33_f64(Hn, M, R);
    if (det3_f64(R) < 0) {
#pragma unroll
        for (int i = 0; i < 9; i++) R[i] = R[i] * -1;
    }
}

// Initializer::findRtWithHomography (reference src/Initializer.cpp:112-163) for one frame pair per wavefront:
// cv::decomposeHomographyMat (Hnorm = K^-1 H K scaled by its middle singular value, the analytic four-solution
// decomposition of S = Hnorm^T Hnorm - I) computed by every lane alike, then
// cv::filterHomographyDecompByVisibleRefpoints with the lanes striding over the correspondences (coordinates narrowed
// to float, as that function reads them), the verdict per solution formed by a ballot.
// sol[pair] = 4 x (R 9, t 3, n 3); info[pair] = number of solutions, bit mask of the visible ones.
__global__ __launch_bounds__(64) void k_pose_homography(const float* __restrict__ H_all, const float* __restrict__ K9,
                                                        const double* __restrict__ pts, const double* __restrict__ ref,
                                                        const int32_t* __restrict__ pair_start,
                                                        double* __restrict__ sol, int32_t* __restrict__ info) {
    const int pair = blockIdx.x;
    const int lane = threadIdx.x;
    double H[9], K[9], Kinv[9], Hn[9];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        H[i] = (double)H_all[(size_t)pair * 9 + i];
        K[i] = (double)K9[i];
    }
    // Matx33d::inv(): adjugate times 1 / det, zeros for a singular matrix
    double d = det3_f64(K);
    if (d == 0) {
#pragma unroll
        for (int i = 0; i < 9; i++) Kinv[i] = 0;
    } else {
        d = 1 / d;
        Kinv[0] = (K[4] * K[8] - K[5] * K[7]) * d;
        Kinv[1] = (K[2] * K[7] - K[1] * K[8]) * d;
        Kinv[2] = (K[1] * K[5] - K[2] * K[4]) * d;
        Kinv[3] = (K[5] * K[6] - K[3] * K[8]) * d;
        Kinv[4] = (K[0] * K[8] - K[2] * K[6]) * d;
        Kinv[5] = (K[2] * K[3] - K[0] * K[5]) * d;
        Kinv[6] = (K[3] * K[7] - K[4] * K[6]) * d;
        Kinv[7] = (K[1] * K[6] - K[0] * K[7]) * d;
        Kinv[8] = (K[0] * K[4] - K[1] * K[3]) * d;
    }
    mul33_f64(Kinv, H, Hn);
    mul33_f64(Hn, K, Hn);
    {
        // removeScale: the singular values of Hnorm (the SVD front end works on the transpose), Hnorm / W[1]
        double At[9], Vt[9], W[3];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) At[j * 3 + i] = Hn[i * 3 + j];
        jacobi_square_f64<3>(At, Vt, W);
        const double scale = 1.0 / W[1];
#pragma unroll
        for (int i = 0; i < 9; i++) Hn[i] = Hn[i] * scale;
    }
    double S[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) s += Hn[k * 3 + i] * Hn[k * 3 + j];
            S[i * 3 + j] = s;
        }
    S[0] -= 1.0;
    S[4] -= 1.0;
    S[8] -= 1.0;
    double ninf = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const double a = fabs(S[i]);
        ninf = ninf < a ? a : ninf;
    }
    // the four motions: R (9), t (3), n (3) each
    double Ra[9], Rb[9], ta[3], tb[3], na[3], nb[3];
    int nsol;
    if (ninf < 0.001) {
        nsol = 1;
#pragma unroll
        for (int i = 0; i < 9; i++) Ra[i] = Rb[i] = Hn[i];
#pragma unroll
        for (int i = 0; i < 3; i++) ta[i] = tb[i] = na[i] = nb[i] = 0;
    } else {
        nsol = 4;
        const double M00 = opposite_of_minor<0, 0>(S), M11 = opposite_of_minor<1, 1>(S);
        const double M22 = opposite_of_minor<2, 2>(S);
        const double rtM00 = sqrt(M00), rtM11 = sqrt(M11), rtM22 = sqrt(M22);
        const double M01 = opposite_of_minor<0, 1>(S), M12 = opposite_of_minor<1, 2>(S);
        const double M02 = opposite_of_minor<0, 2>(S);
        const int e12 = M12 >= 0 ? 1 : -1, e02 = M02 >= 0 ? 1 : -1, e01 = M01 >= 0 ? 1 : -1;
        const double nS00 = fabs(S[0]), nS11 = fabs(S[4]), nS22 = fabs(S[8]);
        int indx = 0;
        if (nS00 < nS11) {
            indx = 1;
            if (nS11 < nS22) indx = 2;
        } else {
            if (nS00 < nS22) indx = 2;
        }
        double npa[3], npb[3], Sii;
        if (indx == 0) {
            npa[0] = S[0], npb[0] = S[0];
            npa[1] = S[1] + rtM22, npb[1] = S[1] - rtM22;
            npa[2] = S[2] + e12 * rtM11, npb[2] = S[2] - e12 * rtM11;
            Sii = S[0];
        } else if (indx == 1) {
            npa[0] = S[1] + rtM22, npb[0] = S[1] - rtM22;
            npa[1] = S[4], npb[1] = S[4];
            npa[2] = S[5] - e02 * rtM00, npb[2] = S[5] + e02 * rtM00;
            Sii = S[4];
        } else {
            npa[0] = S[2] + e01 * rtM11, npb[0] = S[2] - e01 * rtM11;
            npa[1] = S[5] + rtM00, npb[1] = S[5] - rtM00;
            npa[2] = S[8], npb[2] = S[8];
            Sii = S[8];
        }
        const double traceS = S[0] + S[4] + S[8];
        const double v = 2.0 * sqrt(1 + traceS - M00 - M11 - M22);
        const double ESii = Sii >= 0 ? 1 : -1;
        const double r_2 = 2 + traceS + v, nt_2 = 2 + traceS - v;
        const double r = sqrt(r_2), n_t = sqrt(nt_2);
        double sa = 0, sb = 0;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            sa += npa[i] * npa[i];
            sb += npb[i] * npb[i];
        }
        const double ia = 1. / sqrt(sa), ib = 1. / sqrt(sb);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            na[i] = npa[i] * ia;
            nb[i] = npb[i] * ib;
        }
        const double half_nt = 0.5 * n_t, esii_t_r = ESii * r;
        double ta_star[3], tb_star[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            ta_star[i] = (nb[i] * esii_t_r - na[i] * n_t) * half_nt;
            tb_star[i] = (na[i] * esii_t_r - nb[i] * n_t) * half_nt;
        }
        rmat_from_tstar_n(Hn, ta_star, na, v, Ra);
        rmat_from_tstar_n(Hn, tb_star, nb, v, Rb);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            double s0 = 0, s1 = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                s0 += Ra[i * 3 + k] * ta_star[k];
                s1 += Rb[i * 3 + k] * tb_star[k];
            }
            ta[i] = s0;
            tb[i] = s1;
        }
    }
    // visibility: solution q has the normal sign[q] * (na | nb) and the rotated normal R (that normal)
    double rna[3], rnb[3], rnan[3], rnbn[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            s0 += Ra[i * 3 + k] * na[k];
            s1 += Ra[i * 3 + k] * (na[k] * -1);
            s2 += Rb[i * 3 + k] * nb[k];
            s3 += Rb[i * 3 + k] * (nb[k] * -1);
        }
        rna[i] = s0; rnan[i] = s1; rnb[i] = s2; rnbn[i] = s3;
    }
    const int first = pair_start[pair], n = pair_start[pair + 1] - first;
    bool ok0 = true, ok1 = true, ok2 = true, ok3 = true;
    for (int j = lane; j < n; j += 64) {
        const size_t o = 2 * (size_t)(first + j);
        const double px = (double)(float)pts[o], py = (double)(float)pts[o + 1];
        const double cx = (double)(float)ref[o], cy = (double)(float)ref[o + 1];
        ok0 = ok0 && !(na[0] * px + na[1] * py + na[2] <= 0 || rna[0] * cx + rna[1] * cy + rna[2] <= 0);
        ok1 = ok1 && !((na[0] * -1) * px + (na[1] * -1) * py + (na[2] * -1) <= 0 ||
                       rnan[0] * cx + rnan[1] * cy + rnan[2] <= 0);
        ok2 = ok2 && !(nb[0] * px + nb[1] * py + nb[2] <= 0 || rnb[0] * cx + rnb[1] * cy + rnb[2] <= 0);
        ok3 = ok3 && !((nb[0] * -1) * px + (nb[1] * -1) * py + (nb[2] * -1) <= 0 ||
                       rnbn[0] * cx + rnbn[1] * cy + rnbn[2] <= 0);
    }
    int mask = 0;
    if (__ballot(!ok0) == 0) mask |= 1;
    if (nsol == 4) {
        if (__ballot(!ok1) == 0) mask |= 2;
        if (__ballot(!ok2) == 0) mask |= 4;
        if (__ballot(!ok3) == 0) mask |= 8;
    }
    if (lane == 0) {
        double* o = sol + (size_t)pair * 60;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            o[i] = Ra[i];
            o[15 + i] = Ra[i];
            o[30 + i] = Rb[i];
            o[45 + i] = Rb[i];
        }
#pragma unroll
        for (int i = 0; i < 3; i++) {
            o[9 + i] = ta[i];
            o[12 + i] = na[i];
            o[24 + i] = ta[i] * -1;
            o[27 + i] = na[i] * -1;
            o[39 + i] = tb[i];
            o[42 + i] = nb[i];
            o[54 + i] = tb[i] * -1;
            o[57 + i] = nb[i] * -1;
        }
        info[pair * 2] = nsol;
        info[pair * 2 + 1] = mask;
    }
}

}  // namespace orbx
