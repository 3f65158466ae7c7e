/*
 * CPU oracle, pose recovery after the F/H RANSAC (TEST INFRASTRUCTURE ONLY; see oracle.h).
 * Follows src/Initializer.cpp:165-252 of the reference (findRtWithFundamental, triangulatePoints).
 * cv::recoverPose, cv::decomposeEssentialMat, cv::triangulatePoints, cv::SVD::compute on CV_64F, cv::gemm and the
 * cv::Mat expression arithmetic belong to OpenCV 3.4.9 (absent from the reference tree and from this image); they
 * are restated from the published 3.4 sources (calib3d five-point.cpp, triangulate.cpp, core lapack.cpp/matmul.cpp).
 * PARITY UNPINNED: no test or fixture of the reference records a rotation, translation or triangulated point.
 * Compile with -ffp-contract=off.
 */
#include "oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

/* core lapack.cpp JacobiSVDImpl_<double>: as the float form, with every intermediate in double,
   eps = DBL_EPSILON * 10 and minval = DBL_MIN. */
void jacobi_svd_f64(double* At, int astep, double* W, double* Vt, int vstep, int m, int n, int n1) {
    const double minval = DBL_MIN;
    const double eps = DBL_EPSILON * 10;
    int i, j, k, iter, max_iter = std::max(m, 30);
    double c, s, sd;
    for (i = 0; i < n; i++) {
        for (k = 0, sd = 0; k < m; k++) {
            double t = At[i * astep + k];
            sd += t * t;
        }
        W[i] = sd;
        if (Vt) {
            for (k = 0; k < n; k++) Vt[i * vstep + k] = 0;
            Vt[i * vstep + i] = 1;
        }
    }
    for (iter = 0; iter < max_iter; iter++) {
        bool changed = false;
        for (i = 0; i < n - 1; i++)
            for (j = i + 1; j < n; j++) {
                double *Ai = At + i * astep, *Aj = At + j * astep;
                double a = W[i], p = 0, b = W[j];
                for (k = 0; k < m; k++) p += Ai[k] * Aj[k];
                if (std::abs(p) <= eps * std::sqrt(a * b)) continue;
                p *= 2;
                double beta = a - b, gamma = hypot(p, beta);
                if (beta < 0) {
                    double delta = (gamma - beta) * 0.5;
                    s = std::sqrt(delta / gamma);
                    c = p / (gamma * s * 2);
                } else {
                    c = std::sqrt((gamma + beta) / (gamma * 2));
                    s = p / (gamma * c * 2);
                }
                a = b = 0;
                for (k = 0; k < m; k++) {
                    double t0 = c * Ai[k] + s * Aj[k];
                    double t1 = -s * Ai[k] + c * Aj[k];
                    Ai[k] = t0;
                    Aj[k] = t1;
                    a += t0 * t0;
                    b += t1 * t1;
                }
                W[i] = a;
                W[j] = b;
                changed = true;
                if (Vt) {
                    double *Vi = Vt + i * vstep, *Vj = Vt + j * vstep;
                    for (k = 0; k < n; k++) {
                        double t0 = c * Vi[k] + s * Vj[k];
                        double t1 = -s * Vi[k] + c * Vj[k];
                        Vi[k] = t0;
                        Vj[k] = t1;
                    }
                }
            }
        if (!changed) break;
    }
    for (i = 0; i < n; i++) {
        for (k = 0, sd = 0; k < m; k++) {
            double t = At[i * astep + k];
            sd += t * t;
        }
        W[i] = std::sqrt(sd);
    }
    for (i = 0; i < n - 1; i++) {
        j = i;
        for (k = i + 1; k < n; k++)
            if (W[j] < W[k]) j = k;
        if (i != j) {
            std::swap(W[i], W[j]);
            if (Vt) {
                for (k = 0; k < m; k++) std::swap(At[i * astep + k], At[j * astep + k]);
                for (k = 0; k < n; k++) std::swap(Vt[i * vstep + k], Vt[j * vstep + k]);
            }
        }
    }
    if (!Vt) return;
    uint64_t rng = 0x12345678;
    for (i = 0; i < n1; i++) {
        sd = i < n ? W[i] : 0;
        for (int ii = 0; ii < 100 && sd <= minval; ii++) {
            const double val0 = 1. / m;
            for (k = 0; k < m; k++) {
                rng = (uint64_t)(unsigned)rng * 4164903690U + (unsigned)(rng >> 32);
                At[i * astep + k] = ((unsigned)rng & 256) != 0 ? val0 : -val0;
            }
            for (iter = 0; iter < 2; iter++)
                for (j = 0; j < i; j++) {
                    sd = 0;
                    for (k = 0; k < m; k++) sd += At[i * astep + k] * At[j * astep + k];
                    double asum = 0;
                    for (k = 0; k < m; k++) {
                        double t = At[i * astep + k] - sd * At[j * astep + k];
                        At[i * astep + k] = t;
                        asum += std::abs(t);
                    }
                    asum = asum > eps * 100 ? 1 / asum : 0;
                    for (k = 0; k < m; k++) At[i * astep + k] *= asum;
                }
            sd = 0;
            for (k = 0; k < m; k++) {
                double t = At[i * astep + k];
                sd += t * t;
            }
            sd = std::sqrt(sd);
        }
        s = sd > minval ? 1 / sd : 0.;
        for (k = 0; k < m; k++) At[i * astep + k] *= s;
    }
}

/* SVD::compute(A, w, u, vt) for a square n x n CV_64F matrix, flags = 0 (_SVDcompute: A is transposed into the
   work array, whose rows become the rows of U^T; vt = V^T). */
void svd_square_f64(const double* A, int n, double* w, double* u, double* vt) {
    std::vector<double> a((size_t)n * n), v((size_t)n * n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) a[j * n + i] = A[i * n + j];
    jacobi_svd_f64(a.data(), n, w, v.data(), n, n, n, n);
    if (u)
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) u[j * n + i] = a[i * n + j];
    if (vt) std::memcpy(vt, v.data(), sizeof(double) * n * n);
}

double det3(const double* m) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

/* cv::gemm on CV_64F, C = A(ar x ac) * B(ac x bc): every element is the sum of the products in ascending k. */
void gemm_f64(const double* A, int ar, int ac, const double* B, int bc, double* C) {
    std::vector<double> out((size_t)ar * bc);
    for (int i = 0; i < ar; i++)
        for (int j = 0; j < bc; j++) {
            double s = 0;
            for (int k = 0; k < ac; k++) s += A[i * ac + k] * B[k * bc + j];
            out[i * bc + j] = s;
        }
    std::memcpy(C, out.data(), sizeof(double) * out.size());
}

/* calib3d triangulate.cpp (3.4): per point the 4 x 4 system of the two views, SVD, last row of V^T.
   P1, P2: 3 x 4 row-major; x1, x2: n points (x, y); out: n x 4 homogeneous points. */
void triangulate_points(const double* P1, const double* P2, const double* x1, const double* x2, int n, double* out) {
    const double* P[2] = {P1, P2};
    const double* X[2] = {x1, x2};
    for (int i = 0; i < n; i++) {
        double A[16], w[4], vt[16];
        for (int j = 0; j < 2; j++) {
            const double x = X[j][2 * i], y = X[j][2 * i + 1];
            for (int k = 0; k < 4; k++) {
                A[(j * 2 + 0) * 4 + k] = x * P[j][8 + k] - P[j][k];
                A[(j * 2 + 1) * 4 + k] = y * P[j][8 + k] - P[j][4 + k];
            }
        }
        svd_square_f64(A, 4, w, nullptr, vt);
        for (int k = 0; k < 4; k++) out[4 * i + k] = vt[12 + k];
    }
}

} // namespace

extern "C" {

void orc_svd_square_f64(const double* A, int n, double* w, double* u, double* vt) { svd_square_f64(A, n, w, u, vt); }

void orc_triangulate_points(const double* P1, const double* P2, const double* x1, const double* x2, int n,
                            double* out4) {
    triangulate_points(P1, P2, x1, x2, n, out4);
}

/* Initializer::findRtWithFundamental, Initializer.cpp:165-199: E = K^T F K, then
   cv::recoverPose(E, inlier_ref_points, inlier_points, R, t, focal, pp) with focal = (fx + fy) / 2 and
   pp = Point2f(cx, cy).  K9 and F9 are CV_32F as in the reference.  Returns the number of points in front of both
   cameras for the chosen solution (the return value of cv::recoverPose). */
int orc_recover_pose_fundamental(const float* F9, const float* K9, const double* inlier_points,
                                 const double* inlier_ref_points, int n, double* R9, double* t3) {
    /* K.t() * F: transpose flag -> cv::gemm's general path, products summed in double, rounded to float once;
       (K^T F) * K: no flags, 3 x 3 -> the small-matrix float path, left to right in float. */
    float KtF[9], Ef[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += (double)K9[k * 3 + i] * (double)F9[k * 3 + j];
            KtF[i * 3 + j] = (float)s;
        }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            Ef[i * 3 + j] = KtF[i * 3] * K9[j] + KtF[i * 3 + 1] * K9[3 + j] + KtF[i * 3 + 2] * K9[6 + j];
    double E[9];
    for (int i = 0; i < 9; i++) E[i] = Ef[i];
    const double focal = (double)(K9[0] + K9[4]) / 2.0; /* (focalX() + focalY()) / 2.0: float sum, double divide */
    const double fx = focal, fy = focal, cx = (double)K9[2], cy = (double)K9[5];

    /* recoverPose: points1 = inlier_ref_points, points2 = inlier_points, normalised as
       (col - c) / f == col * (1 / f) + (-c * (1 / f)) (MatExpr scaling, evaluated by convertTo). */
    std::vector<double> p1((size_t)2 * n), p2((size_t)2 * n);
    const double ifx = 1. / fx, ify = 1. / fy;
    const double bx = -cx * ifx, by = -cy * ify;
    for (int i = 0; i < n; i++) {
        p1[2 * i] = inlier_ref_points[2 * i] * ifx + bx;
        p2[2 * i] = inlier_points[2 * i] * ifx + bx;
        p1[2 * i + 1] = inlier_ref_points[2 * i + 1] * ify + by;
        p2[2 * i + 1] = inlier_points[2 * i + 1] * ify + by;
    }
    /* decomposeEssentialMat */
    double D[3], U[9], Vt[9];
    svd_square_f64(E, 3, D, U, Vt);
    if (det3(U) < 0)
        for (int i = 0; i < 9; i++) U[i] = U[i] * -1. + 0.;
    if (det3(Vt) < 0)
        for (int i = 0; i < 9; i++) Vt[i] = Vt[i] * -1. + 0.;
    const double W[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1}, Wt[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1};
    double UW[9], R1[9], R2[9], t[3];
    gemm_f64(U, 3, 3, W, 3, UW);
    gemm_f64(UW, 3, 3, Vt, 3, R1);
    gemm_f64(U, 3, 3, Wt, 3, UW);
    gemm_f64(UW, 3, 3, Vt, 3, R2);
    for (int i = 0; i < 3; i++) t[i] = U[i * 3 + 2] * 1.0;
    const double P0[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    const double* Rs[4] = {R1, R2, R1, R2};
    const double sgn[4] = {1, 1, -1, -1};
    const double dist = 50.0;
    int good[4];
    std::vector<double> Q((size_t)4 * std::max(n, 1));
    for (int c = 0; c < 4; c++) {
        double P[12];
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) P[i * 4 + j] = Rs[c][i * 3 + j] * 1.0;
            P[i * 4 + 3] = sgn[c] < 0 ? -t[i] * 1.0 : t[i] * 1.0;
        }
        triangulate_points(P0, P, p1.data(), p2.data(), n, Q.data());
        int cnt = 0;
        for (int i = 0; i < n; i++) {
            double* q = &Q[4 * i];
            bool m = q[2] * q[3] > 0;
            const double w = q[3];
            q[0] /= w; q[1] /= w; q[2] /= w; q[3] /= w;
            m = (q[2] < dist) && m;
            double z = 0;
            for (int k = 0; k < 4; k++) z += P[8 + k] * q[k];
            m = (z > 0) && m;
            m = (z < dist) && m;
            cnt += m;
        }
        good[c] = cnt;
    }
    int pick;
    if (good[0] >= good[1] && good[0] >= good[2] && good[0] >= good[3]) pick = 0;
    else if (good[1] >= good[0] && good[1] >= good[2] && good[1] >= good[3]) pick = 1;
    else if (good[2] >= good[0] && good[2] >= good[1] && good[2] >= good[3]) pick = 2;
    else pick = 3;
    std::memcpy(R9, Rs[pick], sizeof(double) * 9);
    for (int i = 0; i < 3; i++) t3[i] = sgn[pick] < 0 ? -t[i] : t[i];
    return good[pick];
}

/* Initializer::triangulatePoints, Initializer.cpp:201-252: projection = K * [R | t], projection_ref = K * [I | 0]
   (K widened to double), cv::triangulatePoints(projection, projection_ref, inlier_points, inlier_ref_points),
   then `x /= x(3)` (a multiplication by the reciprocal: Mat::convertTo with alpha = 1 / w). */
void orc_triangulate_initial(const float* K9, const double* R9, const double* t3, const double* inlier_points,
                             const double* inlier_ref_points, int n, double* points3) {
    double K[9], ext[12], ext_ref[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}, P[12], Pref[12];
    for (int i = 0; i < 9; i++) K[i] = K9[i];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) ext[i * 4 + j] = R9[i * 3 + j];
        ext[i * 4 + 3] = t3[i];
    }
    gemm_f64(K, 3, 3, ext_ref, 4, Pref);
    gemm_f64(K, 3, 3, ext, 4, P);
    std::vector<double> Q((size_t)4 * std::max(n, 1));
    triangulate_points(P, Pref, inlier_points, inlier_ref_points, n, Q.data());
    for (int i = 0; i < n; i++) {
        const double alpha = 1. / Q[4 * i + 3];
        for (int k = 0; k < 3; k++) points3[3 * i + k] = Q[4 * i + k] * alpha + 0.;
    }
}

} // extern "C"
