/*
 * CPU oracle, two-view geometry half (TEST INFRASTRUCTURE ONLY; see oracle.h).
 * Follows src/geometry/utils.cpp:18-154 and src/Initializer.cpp:254-585 of the reference.
 * cv::SVDecomp / cv::Mat products / cv::Mat::inv are OpenCV 3.4.9 (absent from the reference
 * tree): the stock (non-LAPACK) JacobiSVDImpl_<float>, cv::gemm's small-matrix float path for the 3x3
 * products and the closed-form 3x3 inverse are restated from the published algorithm.
 * Parity unpinned beyond the reference's 79-pair F/H goldens (see oracle.h).
 * Compile with -ffp-contract=off.
 */
#include "oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <numeric>
#include <random>
#include <vector>

namespace {

/* core lapack.cpp JacobiSVDImpl_<float>: one-sided Jacobi on the n rows (length m) of At; Vt is n x n.
   n1 rows of At are completed to an orthonormal set (rows >= n, or with a vanishing singular value). */
void jacobi_svd_f32(float* At, int astep, float* _W, float* Vt, int vstep, int m, int n, int n1) {
    const double minval = FLT_MIN;
    const float eps = FLT_EPSILON * 2;
    std::vector<double> W(n);
    int i, j, k, iter, max_iter = std::max(m, 30);
    float c, s;
    double sd;

    for (i = 0; i < n; i++) {
        for (k = 0, sd = 0; k < m; k++) {
            float t = At[i * astep + k];
            sd += (double)t * t;
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
                float *Ai = At + i * astep, *Aj = At + j * astep;
                double a = W[i], p = 0, b = W[j];
                for (k = 0; k < m; k++) p += (double)Ai[k] * Aj[k];
                if (std::abs(p) <= eps * std::sqrt((double)a * b)) continue;
                p *= 2;
                double beta = a - b, gamma = hypot((double)p, beta);
                if (beta < 0) {
                    double delta = (gamma - beta) * 0.5;
                    s = (float)std::sqrt(delta / gamma);
                    c = (float)(p / (gamma * s * 2));
                } else {
                    c = (float)std::sqrt((gamma + beta) / (gamma * 2));
                    s = (float)(p / (gamma * c * 2));
                }
                a = b = 0;
                for (k = 0; k < m; k++) {
                    float t0 = c * Ai[k] + s * Aj[k];
                    float t1 = -s * Ai[k] + c * Aj[k];
                    Ai[k] = t0;
                    Aj[k] = t1;
                    a += (double)t0 * t0;
                    b += (double)t1 * t1;
                }
                W[i] = a;
                W[j] = b;
                changed = true;
                if (Vt) {
                    float *Vi = Vt + i * vstep, *Vj = Vt + j * vstep;
                    for (k = 0; k < n; k++) {
                        float t0 = c * Vi[k] + s * Vj[k];
                        float t1 = -s * Vi[k] + c * Vj[k];
                        Vi[k] = t0;
                        Vj[k] = t1;
                    }
                }
            }
        if (!changed) break;
    }

    for (i = 0; i < n; i++) {
        for (k = 0, sd = 0; k < m; k++) {
            float t = At[i * astep + k];
            sd += (double)t * t;
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

    for (i = 0; i < n; i++) _W[i] = (float)W[i];
    if (!Vt) return;

    uint64_t rng = 0x12345678; /* cv::RNG: state = (uint32)state * 4164903690 + (state >> 32) */
    for (i = 0; i < n1; i++) {
        sd = i < n ? W[i] : 0;
        for (int ii = 0; ii < 100 && sd <= minval; ii++) {
            const float val0 = (float)(1. / m);
            for (k = 0; k < m; k++) {
                rng = (uint64_t)(unsigned)rng * 4164903690U + (unsigned)(rng >> 32);
                float val = ((unsigned)rng & 256) != 0 ? val0 : -val0;
                At[i * astep + k] = val;
            }
            for (iter = 0; iter < 2; iter++) {
                for (j = 0; j < i; j++) {
                    sd = 0;
                    for (k = 0; k < m; k++) sd += At[i * astep + k] * At[j * astep + k];
                    float asum = 0;
                    for (k = 0; k < m; k++) {
                        float t = (float)(At[i * astep + k] - sd * At[j * astep + k]);
                        At[i * astep + k] = t;
                        asum += std::abs(t);
                    }
                    asum = asum > eps * 100 ? 1 / asum : 0;
                    for (k = 0; k < m; k++) At[i * astep + k] *= asum;
                }
            }
            sd = 0;
            for (k = 0; k < m; k++) {
                float t = At[i * astep + k];
                sd += (double)t * t;
            }
            sd = std::sqrt(sd);
        }
        s = (float)(sd > minval ? 1 / sd : 0.);
        for (k = 0; k < m; k++) At[i * astep + k] *= s;
    }
}

/* cv::Mat 3x3 CV_32F product without transpose flags: cv::gemm's small-matrix path (matmul.cpp, len == 3),
   each dot product evaluated left to right in float. */
void mul33(const float* A, const float* B, float* C) {
    float out[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            float t = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
            out[i * 3 + j] = t;
        }
    std::memcpy(C, out, sizeof(out));
}

void transpose33(const float* A, float* T) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) T[j * 3 + i] = A[i * 3 + j];
}

/* core lapack.cpp cv::invert, 3x3 CV_32F, DECOMP_LU: closed form with a double determinant. */
void inv33(const float* S, float* D) {
    double d = (double)S[0] * ((double)S[4] * S[8] - (double)S[5] * S[7]) -
               (double)S[1] * ((double)S[3] * S[8] - (double)S[5] * S[6]) +
               (double)S[2] * ((double)S[3] * S[7] - (double)S[4] * S[6]);
    if (d == 0) { std::memset(D, 0, 9 * sizeof(float)); return; }
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
    std::memcpy(D, t, sizeof(t));
}

/* `mat *= 1.0 / mat(2,2)`: Mat::convertTo(-1, alpha) on CV_32F works in float: dst = src * (float)alpha + 0.f. */
void scale33(float* M, double alpha) {
    float a = (float)alpha;
    for (int i = 0; i < 9; i++) M[i] = M[i] * a + 0.f;
}

} // namespace

extern "C" {

/* cv::SVDecomp front end (core lapack.cpp _SVDcompute), CV_32F, FULL_UV. */
void orc_svd_f32(const float* A, int m, int n, float* w, float* u, float* vt) {
    bool at = false;
    int rows = m, cols = n;
    if (m < n) { std::swap(m, n); at = true; }
    int urows = m; /* FULL_UV */
    std::vector<float> temp_u((size_t)urows * m, 0.f), temp_v((size_t)n * n), temp_w(n);
    float* temp_a = temp_u.data(); /* n x m, shares storage with temp_u as in OpenCV */
    if (!at) {
        for (int i = 0; i < rows; i++)
            for (int j = 0; j < cols; j++) temp_a[j * m + i] = A[i * cols + j];
    } else {
        std::memcpy(temp_a, A, sizeof(float) * rows * cols);
    }
    jacobi_svd_f32(temp_a, m, temp_w.data(), temp_v.data(), n, m, n, urows);
    std::memcpy(w, temp_w.data(), sizeof(float) * n);
    if (!at) {
        if (u) for (int i = 0; i < urows; i++) for (int j = 0; j < m; j++) u[j * urows + i] = temp_u[i * m + j];
        if (vt) std::memcpy(vt, temp_v.data(), sizeof(float) * n * n);
    } else {
        if (u) for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) u[j * n + i] = temp_v[i * n + j];
        if (vt) std::memcpy(vt, temp_u.data(), sizeof(float) * urows * m);
    }
}

/* utils.cpp:18-52 */
void orc_normalize_points(const float* xy, int n, float* xy_norm, float* T9) {
    float sx = 0.f, sy = 0.f; /* std::accumulate over cv::Point2f */
    for (int i = 0; i < n; i++) { sx = sx + xy[2 * i]; sy = sy + xy[2 * i + 1]; }
    double n_inv = 1.0 / n;
    double mean_x = sx * n_inv, mean_y = sy * n_inv;
    double scale = 0.0;
    for (int i = 0; i < n; i++) {
        float nx = (float)(xy[2 * i] - mean_x), ny = (float)(xy[2 * i + 1] - mean_y);
        xy_norm[2 * i] = nx;
        xy_norm[2 * i + 1] = ny;
        scale += std::sqrt((double)nx * nx + (double)ny * ny); /* cv::norm(Point_<float>) */
    }
    scale *= n_inv;
    scale = std::sqrt(2.0) / scale;
    for (int i = 0; i < n; i++) { /* Point_<float> *= double -> saturate_cast<float>(x * scale) */
        xy_norm[2 * i] = (float)(xy_norm[2 * i] * scale);
        xy_norm[2 * i + 1] = (float)(xy_norm[2 * i + 1] * scale);
    }
    for (int i = 0; i < 9; i++) T9[i] = 0.f;
    T9[0] = (float)scale;
    T9[4] = (float)scale;
    T9[8] = 1.f;
    T9[2] = (float)(-mean_x * scale);
    T9[5] = (float)(-mean_y * scale);
}

/* utils.cpp:61-103 */
void orc_fundamental_8pt(const float* pts, const float* ref, int n, float* F9) {
    std::vector<float> A((size_t)n * 9);
    for (int i = 0; i < n; i++) {
        float u1 = ref[2 * i], v1 = ref[2 * i + 1], u2 = pts[2 * i], v2 = pts[2 * i + 1];
        float* r = &A[(size_t)i * 9];
        r[0] = u2 * u1; r[1] = u2 * v1; r[2] = u2;
        r[3] = v2 * u1; r[4] = v2 * v1; r[5] = v2;
        r[6] = u1; r[7] = v1; r[8] = 1;
    }
    std::vector<float> w(std::max(n, 9)), vt(81);
    orc_svd_f32(A.data(), n, 9, w.data(), nullptr, vt.data());
    float Fpre[9], w3[3], u3[9], vt3[9];
    std::memcpy(Fpre, &vt[8 * 9], sizeof(Fpre));
    orc_svd_f32(Fpre, 3, 3, w3, u3, vt3);
    w3[2] = 0;
    float D[9] = {w3[0], 0, 0, 0, w3[1], 0, 0, 0, w3[2]}, UD[9];
    mul33(u3, D, UD);
    mul33(UD, vt3, F9);
}

/* utils.cpp:112-154 */
void orc_homography_dlt(const float* pts, const float* ref, int n, float* H9) {
    std::vector<float> A((size_t)2 * n * 9);
    for (int i = 0; i < n; i++) {
        float u1 = ref[2 * i], v1 = ref[2 * i + 1], u2 = pts[2 * i], v2 = pts[2 * i + 1];
        float* r = &A[(size_t)(2 * i) * 9];
        r[0] = 0; r[1] = 0; r[2] = 0; r[3] = -u1; r[4] = -v1; r[5] = -1;
        r[6] = v2 * u1; r[7] = v2 * v1; r[8] = v2;
        r += 9;
        r[0] = u1; r[1] = v1; r[2] = 1; r[3] = 0; r[4] = 0; r[5] = 0;
        r[6] = -u2 * u1; r[7] = -u2 * v1; r[8] = -u2;
    }
    std::vector<float> w(9), vt(81);
    orc_svd_f32(A.data(), 2 * n, 9, w.data(), nullptr, vt.data());
    std::memcpy(H9, &vt[8 * 9], 9 * sizeof(float));
}

/* Initializer.cpp:369-444 */
double orc_score_fundamental(const float* F, const float* pts, const float* ref, int n, double sigma,
                             uint8_t* inliers) {
    double f11 = F[0], f12 = F[1], f13 = F[2], f21 = F[3], f22 = F[4], f23 = F[5], f31 = F[6], f32 = F[7], f33 = F[8];
    double score = 0;
    const double th = 3.841, th_score = 5.991;
    double inv_sigma_square = 1.0 / (sigma * sigma);
    for (int i = 0; i < n; i++) {
        bool good = true;
        const float u1 = ref[2 * i], v1 = ref[2 * i + 1], u2 = pts[2 * i], v2 = pts[2 * i + 1];
        const double a2 = f11 * u1 + f12 * v1 + f13;
        const double b2 = f21 * u1 + f22 * v1 + f23;
        const double c2 = f31 * u1 + f32 * v1 + f33;
        const double num2 = a2 * u2 + b2 * v2 + c2;
        const double sd1 = num2 * num2 / (a2 * a2 + b2 * b2);
        const double chi1 = sd1 * inv_sigma_square;
        if (chi1 > th) good = false; else score += th_score - chi1;
        const double a1 = f11 * u2 + f21 * v2 + f31;
        const double b1 = f12 * u2 + f22 * v2 + f32;
        const double c1 = f13 * u2 + f23 * v2 + f33;
        const double num1 = a1 * u1 + b1 * v1 + c1;
        const double sd2 = num1 * num1 / (a1 * a1 + b1 * b1);
        const double chi2 = sd2 * inv_sigma_square;
        if (chi2 > th) good = false; else score += th_score - chi2;
        if (inliers) inliers[i] = good;
    }
    return score;
}

/* Initializer.cpp:501-585 */
double orc_score_homography(const float* H, const float* Hi, const float* pts, const float* ref, int n, double sigma,
                            uint8_t* inliers) {
    const float h11 = H[0], h12 = H[1], h13 = H[2], h21 = H[3], h22 = H[4], h23 = H[5], h31 = H[6], h32 = H[7], h33 = H[8];
    const float i11 = Hi[0], i12 = Hi[1], i13 = Hi[2], i21 = Hi[3], i22 = Hi[4], i23 = Hi[5], i31 = Hi[6], i32 = Hi[7], i33 = Hi[8];
    float score = 0;
    const float th = 5.991;
    const float invSigmaSquare = 1.0 / (sigma * sigma);
    for (int i = 0; i < n; i++) {
        bool good = true;
        const float u1 = ref[2 * i], v1 = ref[2 * i + 1], u2 = pts[2 * i], v2 = pts[2 * i + 1];
        const float w2in1inv = 1.0 / (i31 * u2 + i32 * v2 + i33);
        const float u2in1 = (i11 * u2 + i12 * v2 + i13) * w2in1inv;
        const float v2in1 = (i21 * u2 + i22 * v2 + i23) * w2in1inv;
        const float sq1 = (u1 - u2in1) * (u1 - u2in1) + (v1 - v2in1) * (v1 - v2in1);
        const float chi1 = sq1 * invSigmaSquare;
        if (chi1 > th) good = false; else score += th - chi1;
        const float w1in2inv = 1.0 / (h31 * u1 + h32 * v1 + h33);
        const float u1in2 = (h11 * u1 + h12 * v1 + h13) * w1in2inv;
        const float v1in2 = (h21 * u1 + h22 * v1 + h23) * w1in2inv;
        const float sq2 = (u2 - u1in2) * (u2 - u1in2) + (v2 - v1in2) * (v2 - v1in2);
        const float chi2 = sq2 * invSigmaSquare;
        if (chi2 > th) good = false; else score += th - chi2;
        if (inliers) inliers[i] = good;
    }
    return score;
}

/* Initializer.cpp:261-277, with the per-iteration engine seeded deterministically (seed0 + it) in place of
   std::random_device; the index vector is shuffled cumulatively as in the reference. */
void orc_ransac_sets(int n, int iterations, uint32_t seed0, uint32_t* sets8) {
    std::vector<size_t> all(n);
    for (int i = 0; i < n; ++i) all[i] = i;
    for (int it = 0; it < iterations; it++) {
        std::shuffle(all.begin(), all.end(), std::mt19937{seed0 + (uint32_t)it});
        for (int j = 0; j < 8; j++) sets8[it * 8 + j] = (uint32_t)all[j];
    }
}

static void one_iteration(const float* pn, const float* rn, const float* T, const float* Tt, const float* Tinv,
                          const float* refT, const uint32_t* set, float* F, float* H, float* Hinv) {
    float ip[16], ir[16];
    for (int j = 0; j < 8; j++) {
        uint32_t idx = set[j];
        ip[2 * j] = pn[2 * idx]; ip[2 * j + 1] = pn[2 * idx + 1];
        ir[2 * j] = rn[2 * idx]; ir[2 * j + 1] = rn[2 * idx + 1];
    }
    (void)T;
    float fn[9], tmp[9];
    orc_fundamental_8pt(ip, ir, 8, fn);
    mul33(Tt, fn, tmp);
    mul33(tmp, refT, F); /* Initializer.cpp:341 */
    if (fabs(F[8]) > 1e-8) scale33(F, 1.0 / F[8]);
    float hn[9];
    orc_homography_dlt(ip, ir, 8, hn);
    mul33(Tinv, hn, tmp);
    mul33(tmp, refT, H); /* Initializer.cpp:481 */
    inv33(H, Hinv);      /* :482, before normalisation */
    if (fabs(H[8]) > 1e-8) scale33(H, 1.0 / H[8]);
}

void orc_ransac_fh_all(const float* pts, const float* ref, int n, const uint32_t* sets8, int iterations, double sigma,
                       float* F_all, float* H_all, float* Hinv_all, float* score_f, float* score_h) {
    std::vector<float> pn(2 * n), rn(2 * n);
    float T[9], refT[9], Tt[9], Tinv[9];
    orc_normalize_points(pts, n, pn.data(), T);
    orc_normalize_points(ref, n, rn.data(), refT);
    transpose33(T, Tt);
    inv33(T, Tinv);
    for (int it = 0; it < iterations; it++) {
        float F[9], H[9], Hi[9];
        one_iteration(pn.data(), rn.data(), T, Tt, Tinv, refT, sets8 + it * 8, F, H, Hi);
        if (F_all) std::memcpy(F_all + it * 9, F, sizeof(F));
        if (H_all) std::memcpy(H_all + it * 9, H, sizeof(H));
        if (Hinv_all) std::memcpy(Hinv_all + it * 9, Hi, sizeof(Hi));
        if (score_f) score_f[it] = (float)orc_score_fundamental(F, pts, ref, n, sigma, nullptr);
        if (score_h) score_h[it] = (float)orc_score_homography(H, Hi, pts, ref, n, sigma, nullptr);
    }
}

/* Initializer.cpp:254-367,446-499 */
void orc_ransac_fh(const float* pts, const float* ref, int n, const uint32_t* sets8, int iterations, double sigma,
                   float* F9, float* H9, uint8_t* inl_f, uint8_t* inl_h, double* score_f, double* score_h,
                   int* best_it) {
    std::vector<float> pn(2 * n), rn(2 * n);
    float T[9], refT[9], Tt[9], Tinv[9];
    orc_normalize_points(pts, n, pn.data(), T);
    orc_normalize_points(ref, n, rn.data(), refT);
    transpose33(T, Tt);
    inv33(T, Tinv);
    double f_score = 0.0, h_score = 0.0;
    std::memset(inl_f, 0, n);
    std::memset(inl_h, 0, n);
    best_it[0] = best_it[1] = -1;
    std::vector<uint8_t> cur(n);
    for (int it = 0; it < iterations; it++) {
        float F[9], H[9], Hi[9];
        one_iteration(pn.data(), rn.data(), T, Tt, Tinv, refT, sets8 + it * 8, F, H, Hi);
        float cs = (float)orc_score_fundamental(F, pts, ref, n, sigma, cur.data());
        if (cs > f_score) {
            std::memcpy(F9, F, sizeof(F));
            std::memcpy(inl_f, cur.data(), n);
            f_score = cs;
            best_it[0] = it;
        }
        cs = (float)orc_score_homography(H, Hi, pts, ref, n, sigma, cur.data());
        if (cs > h_score) {
            std::memcpy(H9, H, sizeof(H));
            std::memcpy(inl_h, cur.data(), n);
            h_score = cs;
            best_it[1] = it;
        }
    }
    *score_f = f_score;
    *score_h = h_score;
}

} // extern "C"
