/*
 * Drop-in counterpart of the reference's Initializer for the F/H RANSAC part (include/orb_slam/initializer.h:29-79,
 * src/Initializer.cpp:254-585): same member names for the pieces on the hot path.  The frame/camera constructor
 * arguments of the reference are not needed by this part and are left to the caller; setPoints / setRefPoints /
 * findFundamentalAndHomography / the getters are the reference's own test seam.  One additive seam:
 * setRansacSets() injects the 8-index sets; without it they are drawn as in the reference (std::shuffle seeded
 * from std::random_device, cumulative over iterations, Initializer.cpp:261-277).
 */
#pragma once

#include <algorithm>
#include <random>
#include <vector>

#include <orb_slam/orbx_compat.h>

namespace orb_slam {

class Initializer {
public:
    Initializer(double sigma = 1.0, int iterations = 200, int device = 0) :
        sigma_(sigma), iterations_(iterations)
    {
        sigma_squared_ = sigma * sigma;
        orbx_ransac* h = nullptr;
        orbx_compat::check(orbx_ransac_create(device, &h), "Initializer");
        handle_ = std::shared_ptr<orbx_ransac>(h, orbx_ransac_destroy);
    }

    void setRefPoints(const std::vector<cv::Point2f>& ref_points) { ref_points_ = ref_points; }
    void setPoints(const std::vector<cv::Point2f>& points) { points_ = points; }
    void setRansacSets(const std::vector<std::vector<size_t>>& sets) { ransac_sets_ = sets; injected_ = true; }

    const cv::Mat& getFundamentalMat() const { return fundamental_mat_; }
    const cv::Mat& getHomographyMat() const { return homography_mat_; }
    const std::vector<bool>& getInliersF() const { return inliers_f_; }
    const std::vector<bool>& getInliersH() const { return inliers_h_; }
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
    static cv::Mat toMat(const float* m) {
        cv::Mat out(3, 3, CV_32F);
        std::memcpy(out.data, m, 9 * sizeof(float));
        return out;
    }

    double sigma_, sigma_squared_;
    int iterations_;
    bool injected_ = false;
    std::vector<cv::Point2f> points_, ref_points_;
    std::vector<std::vector<size_t>> ransac_sets_;
    cv::Mat fundamental_mat_, homography_mat_;
    std::vector<bool> inliers_f_, inliers_h_;
    double f_score_ = 0.0, h_score_ = 0.0;
    std::shared_ptr<orbx_ransac> handle_;
};

} // namespace orb_slam
