/*
 * CPU oracle, pose recovery after the F/H RANSAC (TEST INFRASTRUCTURE ONLY; see oracle.h).
 * Follows src/Initializer.cpp:112-252 of the reference (findRtWithHomography, findRtWithFundamental,
 * triangulatePoints).  cv::decomposeHomographyMat, cv::filterHomographyDecompByVisibleRefpoints, cv::recoverPose,
 * cv::decomposeEssentialMat, cv::triangulatePoints, cv::SVD::compute on CV_64F, cv::gemm and the cv::Mat / cv::Matx
 * expression arithmetic belong to OpenCV 3.4.9 (absent from the reference tree and from this image); they are
 * restated from the published 3.4 sources (calib3d homography_decomp.cpp, five-point.cpp, triangulate.cpp, core
 * lapack.cpp/matmul.cpp/matx.hpp).
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

/* cv::decomposeHomographyMat(H, K, rotations, translations, normals) of OpenCV 3.4.9 (calib3d homography_decomp.cpp,
   HomographyDecompInria after Malis & Vargas), as Initializer::findRtWithHomography calls it
   (Initializer.cpp:126-131) with the CV_32F homography and the camera's CV_32F intrinsic matrix, both widened to
   Matx33d.  Matx arithmetic: products are sums over ascending k that start from 0, a division of a vector by a
   scalar is a multiplication by its reciprocal.  R: nsol x 9, t: nsol x 3, nrm: nsol x 3; returns nsol (1 for a
   pure rotation, else 4: [Ra, ta, na], [Ra, -ta, -na], [Rb, tb, nb], [Rb, -tb, -nb]). */
static void matx_mul33(const double* A, const double* B, double* C) {
    double out[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += A[i * 3 + k] * B[k * 3 + j];
            out[i * 3 + j] = s;
        }
    std::memcpy(C, out, sizeof(out));
}

/* HomographyDecompInria::oppositeOfMinor */
static double opposite_of_minor(const double* M, int row, int col) {
    const int x1 = col == 0 ? 1 : 0, x2 = col == 2 ? 1 : 2;
    const int y1 = row == 0 ? 1 : 0, y2 = row == 2 ? 1 : 2;
    return M[y1 * 3 + x2] * M[y2 * 3 + x1] - M[y1 * 3 + x1] * M[y2 * 3 + x2];
}

static int signd(double x) { return x >= 0 ? 1 : -1; }

/* HomographyDecompInria::findRmatFrom_tstar_n: R = Hnorm * (I - (2 / v) * tstar * n^T), negated when det(R) < 0. */
static void rmat_from_tstar_n(const double* Hn, const double* tstar, const double* n, double v, double* R) {
    const double alpha = 2 / v;
    double ts[3], M[9];
    for (int i = 0; i < 3; i++) ts[i] = tstar[i] * alpha;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            s += ts[i] * n[j];
            M[i * 3 + j] = (i == j ? 1.0 : 0.0) - s;
        }
    matx_mul33(Hn, M, R);
    if (det3(R) < 0)
        for (int i = 0; i < 9; i++) R[i] = R[i] * -1;
}

int orc_decompose_homography(const float* H9, const float* K9, double* R, double* t, double* nrm) {
    double H[9], K[9], Kinv[9], KiH[9], Hn[9];
    for (int i = 0; i < 9; i++) {
        H[i] = H9[i];
        K[i] = K9[i];
    }
    /* K.inv(): Matx_FastInvOp<double, 3, 3> - adjugate times 1 / det, zeros when det == 0 */
    double d = det3(K);
    if (d == 0) {
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
    matx_mul33(Kinv, H, KiH);
    matx_mul33(KiH, K, Hn);
    /* removeScale: SVD::compute(Hnorm, W), Hnorm * (1 / W(1)) */
    double W[3];
    svd_square_f64(Hn, 3, W, nullptr, nullptr);
    const double scale = 1.0 / W[1];
    for (int i = 0; i < 9; i++) Hn[i] = Hn[i] * scale;
    /* decompose: S = Hnorm^T Hnorm - I */
    double Ht[9], S[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Ht[i * 3 + j] = Hn[j * 3 + i];
    matx_mul33(Ht, Hn, S);
    S[0] -= 1.0;
    S[4] -= 1.0;
    S[8] -= 1.0;
    double ninf = 0;
    for (int i = 0; i < 9; i++) ninf = std::max(ninf, std::abs(S[i]));
    if (ninf < 0.001) {
        for (int i = 0; i < 9; i++) R[i] = Hn[i];
        for (int i = 0; i < 3; i++) t[i] = nrm[i] = 0;
        return 1;
    }
    const double M00 = opposite_of_minor(S, 0, 0), M11 = opposite_of_minor(S, 1, 1), M22 = opposite_of_minor(S, 2, 2);
    const double rtM00 = std::sqrt(M00), rtM11 = std::sqrt(M11), rtM22 = std::sqrt(M22);
    const double M01 = opposite_of_minor(S, 0, 1), M12 = opposite_of_minor(S, 1, 2), M02 = opposite_of_minor(S, 0, 2);
    const int e12 = signd(M12), e02 = signd(M02), e01 = signd(M01);
    const double nS00 = std::abs(S[0]), nS11 = std::abs(S[4]), nS22 = std::abs(S[8]);
    int indx = 0;
    if (nS00 < nS11) {
        indx = 1;
        if (nS11 < nS22) indx = 2;
    } else {
        if (nS00 < nS22) indx = 2;
    }
    double npa[3], npb[3];
    switch (indx) {
        case 0:
            npa[0] = S[0], npb[0] = S[0];
            npa[1] = S[1] + rtM22, npb[1] = S[1] - rtM22;
            npa[2] = S[2] + e12 * rtM11, npb[2] = S[2] - e12 * rtM11;
            break;
        case 1:
            npa[0] = S[1] + rtM22, npb[0] = S[1] - rtM22;
            npa[1] = S[4], npb[1] = S[4];
            npa[2] = S[5] - e02 * rtM00, npb[2] = S[5] + e02 * rtM00;
            break;
        default:
            npa[0] = S[2] + e01 * rtM11, npb[0] = S[2] - e01 * rtM11;
            npa[1] = S[5] + rtM00, npb[1] = S[5] - rtM00;
            npa[2] = S[8], npb[2] = S[8];
            break;
    }
    const double traceS = S[0] + S[4] + S[8];
    const double v = 2.0 * std::sqrt(1 + traceS - M00 - M11 - M22);
    const double ESii = signd(S[indx * 3 + indx]);
    const double r_2 = 2 + traceS + v, nt_2 = 2 + traceS - v;
    const double r = std::sqrt(r_2), n_t = std::sqrt(nt_2);
    double na[3], nb[3], sa = 0, sb = 0;
    for (int i = 0; i < 3; i++) {
        sa += npa[i] * npa[i];
        sb += npb[i] * npb[i];
    }
    const double ia = 1. / std::sqrt(sa), ib = 1. / std::sqrt(sb);
    for (int i = 0; i < 3; i++) {
        na[i] = npa[i] * ia;
        nb[i] = npb[i] * ib;
    }
    const double half_nt = 0.5 * n_t, esii_t_r = ESii * r;
    double ta_star[3], tb_star[3];
    for (int i = 0; i < 3; i++) {
        ta_star[i] = (nb[i] * esii_t_r - na[i] * n_t) * half_nt;
        tb_star[i] = (na[i] * esii_t_r - nb[i] * n_t) * half_nt;
    }
    double Ra[9], Rb[9], ta[3], tb[3];
    rmat_from_tstar_n(Hn, ta_star, na, v, Ra);
    rmat_from_tstar_n(Hn, tb_star, nb, v, Rb);
    for (int i = 0; i < 3; i++) {
        double s0 = 0, s1 = 0;
        for (int k = 0; k < 3; k++) {
            s0 += Ra[i * 3 + k] * ta_star[k];
            s1 += Rb[i * 3 + k] * tb_star[k];
        }
        ta[i] = s0;
        tb[i] = s1;
    }
    for (int i = 0; i < 9; i++) {
        R[i] = R[9 + i] = Ra[i];
        R[18 + i] = R[27 + i] = Rb[i];
    }
    for (int i = 0; i < 3; i++) {
        t[i] = ta[i], t[3 + i] = ta[i] * -1, t[6 + i] = tb[i], t[9 + i] = tb[i] * -1;
        nrm[i] = na[i], nrm[3 + i] = na[i] * -1, nrm[6 + i] = nb[i], nrm[9 + i] = nb[i] * -1;
    }
    return 4;
}

/* cv::filterHomographyDecompByVisibleRefpoints(rotations, normals, beforePoints, afterPoints, possibleSolutions) of
   OpenCV 3.4.9 (calib3d homography_decomp.cpp), with the points it reads as CV_32FC2: a solution stays when
   n . (x, y, 1) > 0 for every before-point and (R n) . (x, y, 1) > 0 for every after-point.  The reference passes
   inlier_points as the before-points and inlier_ref_points as the after-points (Initializer.cpp:144-149) - as
   std::vector<cv::Point2d>, which the function's first CV_Assert (both types == CV_32FC2) rejects; this restates
   what it computes once the same coordinates arrive as floats.  Returns the number of solutions kept, their indices
   ascending in `solutions`. */
int orc_filter_homography_visible(const double* R, const double* nrm, int nsol, const float* before_points,
                                  const float* after_points, int npoints, int32_t* solutions) {
    int kept = 0;
    for (int i = 0; i < nsol; i++) {
        double rn[3];
        gemm_f64(R + 9 * i, 3, 3, nrm + 3 * i, 1, rn);
        const double* ni = nrm + 3 * i;
        bool ok = true;
        for (int j = 0; j < npoints && ok; j++) {
            const float px = before_points[2 * j], py = before_points[2 * j + 1];
            const float cx = after_points[2 * j], cy = after_points[2 * j + 1];
            const double prev = ni[0] * px + ni[1] * py + ni[2];
            const double curr = rn[0] * cx + rn[1] * cy + rn[2];
            if (prev <= 0 || curr <= 0) ok = false;
        }
        if (ok) solutions[kept++] = i;
    }
    return kept;
}

/* Initializer::findRtWithHomography, Initializer.cpp:112-163, as it would run if the filter accepted its points:
   decomposition, visibility filter on the inlier points narrowed to float, R and t of the first solution kept.
   Returns that solution's index, or -1 when none is kept (R9 and t3 are then left untouched, as the reference
   leaves its output matrices empty). */
int orc_recover_pose_homography(const float* H9, const float* K9, const double* inlier_points,
                                const double* inlier_ref_points, int n, double* R9, double* t3) {
    double R[36], t[12], nrm[12];
    const int nsol = orc_decompose_homography(H9, K9, R, t, nrm);
    std::vector<float> before((size_t)2 * std::max(n, 1)), after((size_t)2 * std::max(n, 1));
    for (int i = 0; i < 2 * n; i++) {
        before[i] = (float)inlier_points[i];
        after[i] = (float)inlier_ref_points[i];
    }
    int32_t sols[4];
    const int kept = orc_filter_homography_visible(R, nrm, nsol, before.data(), after.data(), n, sols);
    if (kept == 0) return -1;
    std::memcpy(R9, R + 9 * sols[0], sizeof(double) * 9);
    std::memcpy(t3, t + 3 * sols[0], sizeof(double) * 3);
    return sols[0];
}

} // extern "C"
