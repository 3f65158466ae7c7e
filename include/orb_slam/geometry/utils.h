/*
 * Drop-in counterpart of the reference's include/orb_slam/geometry/utils.h: the same eight declarations in the same
 * namespaces (reference utils.h:24-79), declarations ONLY, as there.  The definitions live in one translation unit:
 *   - this repository's src/geometry/utils.cpp: normalizePoints / computeFundamentalMat / computeHomographyMat run
 *     on the GPU through orbx_normalize_points / orbx_fundamental_npt / orbx_homography_dlt (include/orbx.h); or
 *   - the reference's own src/geometry/utils.cpp, which compiles against this header unchanged (its OpenCV code
 *     then serves the public functions; Initializer and the matchers use the GPU either way).
 * Exactly one of the two goes into a build.
 */
#pragma once

#include <numeric>
#include <vector>

#include <orb_slam/orbx_compat.h>
#if ORBX_HAVE_OPENCV
#include <opencv2/highgui/highgui.hpp>
#include <opencv2/imgproc/imgproc.hpp>
#endif

namespace orb_slam {

namespace geometry {

/** Centres the points on their mean and scales them to a mean distance of sqrt(2); T is the 3x3 CV_32F transform. */
void normalizePoints(
    const std::vector<cv::Point2f>& points,
    std::vector<cv::Point2f>& normalized,
    cv::Mat& T);

/** Fundamental matrix (3x3 CV_32F, rank 2) of the n >= 1 correspondences by the linear eight-point method. */
void computeFundamentalMat(
    cv::Mat& f_mat,
    const std::vector<cv::Point2f>& points,
    const std::vector<cv::Point2f>& ref_points);

/** Homography (3x3 CV_32F) of the n >= 1 correspondences by the direct linear transform. */
void computeHomographyMat(
    cv::Mat& h_mat,
    const std::vector<cv::Point2f> &points,
    const std::vector<cv::Point2f> &ref_points);

/** Debug display of the epipolar lines of the inliers (GUI; a no-op in builds without OpenCV). */
void drawEpilines(
    std::vector<cv::Vec3f> lines,
    cv::Mat f_mat,
    cv::Mat image,
    std::vector<cv::Point2f> pts,
    std::vector<bool> inliers);

/** Hamming distance of two 256-bit descriptors (one row of 32 bytes each). */
int descriptorDistance(const cv::Mat &a, const cv::Mat &b);

/** Indices of the three fullest of the L histogram bins; the 2nd / 3rd become -1 below 10 % of the 1st. */
void computeThreeMaxima(
    std::vector<int>* histo, const int L, int &ind1, int &ind2, int &ind3);

} // namespace geometry

namespace utils {
    /** vec_mat[i] = row i of mat (headers sharing mat's storage). */
    void matToVectorMat(const cv::Mat& mat, std::vector<cv::Mat>& vec_mat);

    /** 3x3 CV_32F cross-product matrix of a 3-vector. */
    cv::Mat skew(const cv::Mat& vec);
} // namespace utils

} // namespace orb_slam
