/*
 * Drop-in replacement of the reference's include/orb_slam/geometry/orb_extractor.h: same class, namespace,
 * constructor, getters and detect/compute signatures (reference orb_extractor.h:25-150); the two cv::ORB objects
 * are replaced by one orbx_extractor handle (include/orbx.h), i.e. by the gfx950 kernels.
 */
#pragma once

#include <iostream>
#include <string>

#include <orb_slam/orbx_compat.h>

namespace orb_slam
{

namespace geometry
{

class ORBExtractor
{
public:
    ORBExtractor(const ros::NodeHandle& nh, int reserve_width = 640, int reserve_height = 480, int device = 0): nh_(nh) {
        std::string prefix = "orb_slam/orb_extractor/";
        // same keys and defaults as the reference (orb_extractor.h:29-37)
        nh_.param<int>(prefix + "n_key_points", n_key_points_, 8000);
        nh_.param<float>(prefix + "scale_factor", scale_factor_, 1.200000048F);
        nh_.param<int>(prefix + "level_pyramid", level_pyramid_, 4);
        nh_.param<int>(prefix + "edge_threshold", edge_threshold_, 31);
        nh_.param<int>(prefix + "first_level", first_level_, 0);
        nh_.param<int>(prefix + "wta_k", wta_k_, 2);
        nh_.param<int>(prefix + "patch_size", patch_size_, 31);
        nh_.param<int>(prefix + "score_threshold", score_threshold_, 20);
        // three additive keys the reference's cfg/orb_extractor.yaml does not have; absent, the extractor is cv::ORB's
        // (distribution 1: per-cell adaptive FAST threshold + quad-tree distribution, orbx_extractor_set_distribution)
        nh_.param<int>(prefix + "distribution", distribution_, 0);
        nh_.param<int>(prefix + "low_score_threshold", low_score_threshold_, 7);
        nh_.param<int>(prefix + "distribution_cell_size", distribution_cell_size_, 30);

        orbx_extractor_params p = {
            n_key_points_, scale_factor_, level_pyramid_, edge_threshold_, first_level_, wta_k_, patch_size_,
            score_threshold_};
        handle_ = orbx_compat::PerThreadHandle<orbx_extractor>(
            // the two sizes are what the handle reserves up front; an image of any other size makes it regrow
            [p, reserve_width, reserve_height, device, mode = distribution_, low = low_score_threshold_,
             cell = distribution_cell_size_]() {
                orbx_extractor* h = nullptr;
                orbx_compat::check(
                    orbx_extractor_create(&p, reserve_width, reserve_height, 1, device, &h), "ORBExtractor");
                if (mode != 0) {
                    const orbx_status s = orbx_extractor_set_distribution(h, mode, low, cell);
                    if (s != ORBX_OK) orbx_extractor_destroy(h);
                    orbx_compat::check(s, "ORBExtractor: distribution");
                }
                return h;
            },
            orbx_extractor_destroy);

        scale_factors_.resize(level_pyramid_);
        scale_sigma_sqrd_.resize(level_pyramid_);
        inv_scale_factors_.resize(level_pyramid_);
        inv_scale_sigma_sqrd_.resize(level_pyramid_);
        orbx_extractor_scale_tables(
            handle_.get(), scale_factors_.data(), inv_scale_factors_.data(), scale_sigma_sqrd_.data(),
            inv_scale_sigma_sqrd_.data(), &log_scale_factor_);
    }

    /**
     * Getters
     */
    const float& scaleFactor() const { return scale_factor_;}
    const float& logScaleFactor() const { return log_scale_factor_;}
    const std::vector<float>& scaleFactors() const { return scale_factors_; }
    const int& levels() const { return level_pyramid_; }
    const std::vector<float>& scaleSigmas() const { return scale_sigma_sqrd_; }
    const std::vector<float>& invScaleSigmas() const
        { return inv_scale_sigma_sqrd_; }

    /**
     * Extracts orb features from the input image (8UC1 or 8UC3 BGR)
     */
    void detect(
        const cv::Mat& image, std::vector<cv::KeyPoint>& key_points) const
    {
        key_points.clear();
        if (image.empty()) return; // cv::ORB returns silently on an empty image
        // the library writes into a buffer of this thread that is kept from call to call (ties can push the count
        // past n_key_points, so it starts with room to spare and grows when a frame needs more)
        static thread_local std::vector<orbx_keypoint> found;
        if (found.size() < (size_t)n_key_points_ + 1024) found.resize((size_t)n_key_points_ + 1024);
        int count = 0;
        for (;;) {
            orbx_status s = orbx_detect(
                handle_.get(), image.data, image.cols, image.rows, (int)image.step, image.channels(),
                found.data(), (int)found.size(), &count);
            if (s == ORBX_CAPACITY && (size_t)count > found.size()) { found.resize(count); continue; }
            orbx_compat::check(s, "ORBExtractor::detect");
            break;
        }
        key_points.resize(count);
        if (count > 0) std::memcpy(static_cast<void*>(key_points.data()), found.data(), (size_t)count * sizeof(orbx_keypoint));
    }

    /**
     * Computes orb descriptors from the input image and key points
     */
    void compute(
        const cv::Mat& image,
        std::vector<cv::KeyPoint>& key_points,
        cv::Mat& descriptors) const
    {
        if (image.empty() || key_points.empty()) { descriptors.release(); return; }
        int count = (int)key_points.size();
        cv::Mat d(count, 32, CV_8U);
        orbx_compat::check(
            orbx_compute(
                handle_.get(), image.data, image.cols, image.rows, (int)image.step, image.channels(),
                reinterpret_cast<orbx_keypoint*>(key_points.data()), &count, d.data),
            "ORBExtractor::compute");
        key_points.resize(count);
        if (count == 0) { descriptors.release(); return; }
        if (count == d.rows) descriptors = d;
        else {
            descriptors = cv::Mat(count, 32, CV_8U);
            std::memcpy(descriptors.data, d.data, (size_t)count * 32);
        }
    }

    ~ORBExtractor() {

    }

private:
    ros::NodeHandle nh_;

    // values of the orb_slam/orb_extractor/* keys
    int n_key_points_;
    float scale_factor_;
    int level_pyramid_;
    int edge_threshold_ = 31;
    int first_level_ = 0;
    int wta_k_ = 2;
    int patch_size_ = 31;
    int score_threshold_;
    int distribution_ = 0, low_score_threshold_ = 7, distribution_cell_size_ = 30;

    // per-level tables handed out by the getters; filled once from the device-side extractor
    float log_scale_factor_;
    std::vector<float> scale_factors_;
    std::vector<float> inv_scale_factors_;
    std::vector<float> scale_sigma_sqrd_;
    std::vector<float> inv_scale_sigma_sqrd_;

    // one orbx_extractor per calling thread stands where the reference holds its two cv::ORB objects; the object
    // itself is shared through static members (frame.h:520), so calls may arrive from more than one thread
    orbx_compat::PerThreadHandle<orbx_extractor> handle_;
};

using ORBExtractorPtr = std::shared_ptr<ORBExtractor>;
using ORBExtractorConstPtr = std::shared_ptr<const ORBExtractor>;

} // namespace geometry

} // namespace orb_slam
