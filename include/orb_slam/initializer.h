/*
 * Drop-in counterpart of the reference's Initializer (include/orb_slam/initializer.h:26-223, src/Initializer.cpp):
 * the same constructor (ref_frame, camera, sigma = 1.0, iterations = 200), tryToInitialize(frame, ...), the getters,
 * setPoints / setRefPoints / findFundamentalAndHomography (the reference's own test seam), findRtWithHomography,
 * findRtWithFundamental and triangulatePoints, so src/mono_tracker.cpp:81-87,127-134 and
 * src/tests/initializer_tests.cpp:118-124 compile against it unchanged.  The two-view RANSAC and the pose recovery
 * run on the device through orbx_ransac_* (include/orbx.h).
 *
 * The reference's header only forward-declares Frame and Camera and its .cpp sees the full types; this header has
 * no .cpp, so the constructor and tryToInitialize are templates over the pointer types: they are instantiated at
 * the call site, where the caller has the full types, and they use exactly the accessors the reference uses
 * (Frame::nMatches / matches / featuresUndist, Initializer.cpp:44-53; Camera::intrinsicMatrix, :124,174,218).
 *
 * Two additive seams: setRansacSets() injects the 8-index sets (without it they are drawn as in the reference:
 * std::shuffle seeded from std::random_device, cumulative over iterations, Initializer.cpp:261-277), and
 * setHomographyDecomposition() lets findRtWithHomography compute a pose where the reference always fails.
 */
#pragma once

#include <algorithm>
#include <functional>
#include <random>
#include <vector>

#include <orb_slam/orbx_compat.h>

namespace orb_slam {

namespace geometry {
    template <typename T>
    class Camera;
    template <typename T>
    using CameraPtr = std::shared_ptr<Camera<T>>;
}

class Frame;
using FramePtr = std::shared_ptr<Frame>;

class Initializer {
public:
    /**
     * Initializer::Initializer (Initializer.cpp:13-24).  ref_frame: the first frame to initialise against (may be
     * null: the reference only warns); camera: provides intrinsicMatrix().  `device` (additive, defaulted) picks the GPU.
     */
    template <typename FrameT, typename CameraT>
    Initializer(
        const std::shared_ptr<FrameT>& ref_frame,
        const std::shared_ptr<CameraT>& camera,
        double sigma = 1.0,
        int iterations = 200,
        int device = 0) :
        sigma_(sigma), iterations_(iterations)
    {
        sigma_squared_ = sigma * sigma;
        has_ref_frame_ = static_cast<bool>(ref_frame);
        if (ref_frame)
            ref_features_ = [ref_frame]() -> const std::vector<cv::KeyPoint>& { return ref_frame->featuresUndist(); };
        if (camera)
            intrinsic_of_camera_ = [camera]() { return toFloat33(camera->intrinsicMatrix()); };
        orbx_ransac* h = nullptr;
        orbx_compat::check(orbx_ransac_create(device, &h), "Initializer");
        handle_ = std::shared_ptr<orbx_ransac>(h, orbx_ransac_destroy);
    }

    ~Initializer() {}

    /**
     * Initializer::tryToInitialize (Initializer.cpp:30-110): gathers the matched undistorted points of `frame` and
     * of the reference frame (:44-53), finds F and H, picks the model by the score ratio (:59-92), recovers R, t and
     * triangulates.
     */
    template <typename FrameT>
    bool tryToInitialize(
        const std::shared_ptr<FrameT>& frame,
        cv::Mat& best_rot_mat,
        cv::Mat& best_trans_mat,
        std::vector<cv::Point2d>& inlier_points,
        std::vector<cv::Point2d>& inlier_ref_points,
        std::vector<cv::Point3d>& inlier_points_3d,
        std::vector<size_t>& inliers_idxs)
    {
        if (!has_ref_frame_) {
            // the reference warns and then dereferences the null frame; here the call fails instead
            return false;
        }
        const int n = frame->nMatches();
        const auto& matches = frame->matches();
        const auto& undist_key_points = frame->featuresUndist();
        const auto& undist_ref_key_points = ref_features_();
        ref_points_.resize(n);
        points_.resize(n);
        for (int i = 0; i < n; ++i) {
            ref_points_[i] = undist_ref_key_points[matches[i].trainIdx].pt;
            points_[i] = undist_key_points[matches[i].queryIdx].pt;
        }

        findFundamentalAndHomography();

        float r_score = h_score_ / (h_score_ + f_score_);
        const bool use_h = r_score > 0.40;
        const std::vector<bool>& inl = use_h ? inliers_h_ : inliers_f_;
        for (int i = 0; i < n; ++i) {
            if (!inl[i]) continue;
            const auto& p = points_[i];
            const auto& rp = ref_points_[i];
            inlier_points.push_back(cv::Point2d(p.x, p.y));
            inlier_ref_points.push_back(cv::Point2d(rp.x, rp.y));
            inliers_idxs.push_back(i);
        }
        if (use_h) {
            if (!findRtWithHomography(inlier_points, inlier_ref_points, best_rot_mat, best_trans_mat)) return false;
        } else {
            if (!findRtWithFundamental(inlier_points, inlier_ref_points, best_rot_mat, best_trans_mat)) return false;
        }
        return triangulatePoints(inlier_points, inlier_ref_points, best_rot_mat, best_trans_mat, inlier_points_3d);
    }

    void setRefPoints(const std::vector<cv::Point2f>& ref_points) { ref_points_ = ref_points; }
    void setPoints(const std::vector<cv::Point2f>& points) { points_ = points; }
    /** Additive: the 8-index sets of every iteration (ransac_sets_ of the reference, private there). */
    void setRansacSets(const std::vector<std::vector<size_t>>& sets) { ransac_sets_ = sets; injected_ = true; }

    /**
     * Initializer::findRtWithHomography (Initializer.cpp:112-163).  The reference hands std::vector<cv::Point2d>
     * to cv::filterHomographyDecompByVisibleRefpoints, whose first statement in OpenCV 3.4 (calib3d
     * homography_decomp.cpp) is CV_Assert(_beforeRectifiedPoints.type() == CV_32FC2 &&
     * _afterRectifiedPoints.type() == CV_32FC2); the cv::Exception is caught (:150-155) and the function returns
     * false, so the reference never initialises from a homography.  That outcome is the default here.
     *
     * Additive switch: after setHomographyDecomposition(true) the function computes what its statements describe -
     * cv::decomposeHomographyMat, the visibility filter on the same coordinates held as floats, R and t of the
     * first solution kept - on the device (orbx_recover_pose_homography_batch).  As in the reference it then
     * returns true even when no solution is visible, leaving R and t untouched (triangulatePoints fails on the
     * empty matrices).
     */
    void setHomographyDecomposition(bool enabled) { decompose_homography_ = enabled; }

    bool findRtWithHomography(
        const std::vector<cv::Point2d>& inlier_points,
        const std::vector<cv::Point2d>& inlier_ref_points,
        cv::Mat& R,
        cv::Mat& t)
    {
        if (!decompose_homography_) return false;
        const cv::Mat intrinsic_ = intrinsic();
        if (homography_mat_.empty() || intrinsic_.empty()) return false; // cv::Exception path of the reference
        const int32_t start[2] = {0, (int32_t)inlier_points.size()};
        double r9[9], t3[3];
        int32_t idx = -1;
        orbx_status st = orbx_recover_pose_homography_batch(
            handle_.get(), 1, homography_mat_.ptr<float>(), intrinsic_.ptr<float>(),
            reinterpret_cast<const double*>(inlier_points.data()),
            reinterpret_cast<const double*>(inlier_ref_points.data()), start, r9, t3, &idx, nullptr, nullptr);
        if (st != ORBX_OK) return false;
        if (idx >= 0) {
            R.create(3, 3, CV_64F);
            t.create(3, 1, CV_64F);
            std::memcpy(R.data, r9, sizeof(r9));
            std::memcpy(t.data, t3, sizeof(t3));
        }
        return true;
    }

    /** Initializer::findRtWithFundamental (Initializer.cpp:165-199): R (3x3) and t (3x1) as CV_64F. */
    bool findRtWithFundamental(
        const std::vector<cv::Point2d>& inlier_points,
        const std::vector<cv::Point2d>& inlier_ref_points,
        cv::Mat& R,
        cv::Mat& t)
    {
        const cv::Mat intrinsic_ = intrinsic();
        if (fundamental_mat_.empty() || intrinsic_.empty()) return false; // cv::Exception path of the reference
        const int32_t start[2] = {0, (int32_t)inlier_points.size()};
        double r9[9], t3[3];
        int32_t good = 0;
        orbx_status st = orbx_recover_pose_batch(
            handle_.get(), 1, fundamental_mat_.ptr<float>(), intrinsic_.ptr<float>(),
            reinterpret_cast<const double*>(inlier_points.data()),
            reinterpret_cast<const double*>(inlier_ref_points.data()), start, r9, t3, &good);
        if (st != ORBX_OK) return false;
        R.create(3, 3, CV_64F);
        t.create(3, 1, CV_64F);
        std::memcpy(R.data, r9, sizeof(r9));
        std::memcpy(t.data, t3, sizeof(t3));
        return true;
    }

    /** Initializer::triangulatePoints (Initializer.cpp:201-252). */
    bool triangulatePoints(
        const std::vector<cv::Point2d>& inlier_points,
        const std::vector<cv::Point2d>& inlier_ref_points,
        const cv::Mat& R,
        const cv::Mat& t,
        std::vector<cv::Point3d>& inlier_points_3d)
    {
        const cv::Mat intrinsic_ = intrinsic();
        if (R.empty() || t.empty() || intrinsic_.empty()) return false; // cv::hconcat / gemm would throw
        const int n = (int)inlier_points.size();
        const int32_t start[2] = {0, n};
        std::vector<cv::Point3d> out(n);
        orbx_status st = orbx_triangulate_batch(
            handle_.get(), 1, intrinsic_.ptr<float>(), R.ptr<double>(), t.ptr<double>(),
            reinterpret_cast<const double*>(inlier_points.data()),
            reinterpret_cast<const double*>(inlier_ref_points.data()), start,
            reinterpret_cast<double*>(out.data()));
        if (st != ORBX_OK) return false;
        inlier_points_3d.insert(inlier_points_3d.end(), out.begin(), out.end());
        return true;
    }

    cv::Mat getFundamentalMat() { return fundamental_mat_; }
    cv::Mat getHomographyMat() { return homography_mat_; }
    std::vector<bool> getInliersF() { return inliers_f_; }
    std::vector<bool> getInliersH() { return inliers_h_; }
    double fScore() const { return f_score_; }
    double hScore() const { return h_score_; }

    void findFundamentalAndHomography()
    {
        const size_t n = points_.size();
        if (!injected_) {
            std::vector<size_t> all_indices(n);
            for (size_t i = 0; i < n; ++i) all_indices[i] = i;
            ransac_sets_.assign(iterations_, std::vector<size_t>());
            for (int it = 0; it < iterations_; it++) {
                std::shuffle(all_indices.begin(), all_indices.end(), std::mt19937{std::random_device{}()});
                ransac_sets_[it].assign(all_indices.begin(), all_indices.begin() + 8);
            }
        }
        std::vector<uint32_t> sets8((size_t)ransac_sets_.size() * 8);
        for (size_t it = 0; it < ransac_sets_.size(); ++it)
            for (int j = 0; j < 8; ++j) sets8[it * 8 + j] = (uint32_t)ransac_sets_[it][j];
        float F[9] = {0}, H[9] = {0};
        std::vector<uint8_t> inl_f(n), inl_h(n);
        int32_t best[2] = {-1, -1};
        orbx_compat::check(
            orbx_ransac_fh(
                handle_.get(), reinterpret_cast<const float*>(points_.data()),
                reinterpret_cast<const float*>(ref_points_.data()), (int)n, sets8.data(), (int)ransac_sets_.size(),
                sigma_, F, H, inl_f.data(), inl_h.data(), &f_score_, &h_score_, best),
            "Initializer::findFundamentalAndHomography");
        inliers_f_.assign(inl_f.begin(), inl_f.end());
        inliers_h_.assign(inl_h.begin(), inl_h.end());
        // as in the reference, a matrix stays empty when no hypothesis scored above zero
        fundamental_mat_ = best[0] >= 0 ? toMat(F) : cv::Mat();
        homography_mat_ = best[1] >= 0 ? toMat(H) : cv::Mat();
    }

private:
    //! Camera::intrinsicMatrix() (a cv::Mat_<float> in the reference) as a continuous 3x3 CV_32F matrix
    template <typename MatT>
    static cv::Mat toFloat33(const MatT& K) {
        cv::Mat out(3, 3, CV_32F);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) out.at<float>(r, c) = K.template at<float>(r, c);
        return out;
    }
    cv::Mat intrinsic() const { return intrinsic_of_camera_ ? intrinsic_of_camera_() : cv::Mat(); }

    static cv::Mat toMat(const float* m) {
        cv::Mat out(3, 3, CV_32F);
        std::memcpy(out.data, m, 9 * sizeof(float));
        return out;
    }

    double sigma_, sigma_squared_;
    int iterations_;
    bool injected_ = false;
    bool decompose_homography_ = false;
    std::vector<cv::Point2f> points_, ref_points_;
    std::vector<std::vector<size_t>> ransac_sets_;
    cv::Mat fundamental_mat_, homography_mat_;
    bool has_ref_frame_ = false;
    std::function<const std::vector<cv::KeyPoint>&()> ref_features_;  // ref_frame_->featuresUndist()
    std::function<cv::Mat()> intrinsic_of_camera_;                    // camera_->intrinsicMatrix()
    std::vector<bool> inliers_f_, inliers_h_;
    double f_score_ = 0.0, h_score_ = 0.0;
    std::shared_ptr<orbx_ransac> handle_;
};

using InitializerPtr = std::unique_ptr<Initializer>;

} // namespace orb_slam
