/*
 * Drop-in counterpart of the reference's ORBMatcher facade (include/orb_slam/geometry/orb_matcher.h:342-470,
 * src/geometry/orb_matcher.cpp:679-791,890-924) on top of the C ABI in include/orbx.h.
 *
 * match() x2 and filterMatches() keep the reference signatures.  The projection matchers of the reference take
 * FramePtr / MapPointPtr, types that live outside this path; here they are templates over any frame type that
 * offers the accessors the reference's Frame has (featuresUndist(), descriptorsUndist(), obsMapPoints(),
 * worldToCamera<float>(), cameraToFrame<float,float>(), camera(), orbExtractor()), so the tracker's call sites
 * (src/frame.cpp:394-422) compile unchanged against them.
 */
#pragma once
#include <algorithm>

#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include <orb_slam/geometry/utils.h>
#include <orb_slam/orbx_compat.h>

namespace orb_slam
{

namespace geometry
{

//! Types of matchers (same enumerators as the reference)
enum OrbMatcherTypes {
    BF_WITH_RADIUS,
    BF_WITH_PROJ,
    BOW_ORB,
    EPIPOLAR_CONSTRAINT,
    CV_ORB,
    MATCHER_TYPES
};

class ORBMatcher {
public:
    ORBMatcher(const ros::NodeHandle& nh, int device = 0): nh_(nh) {
        nh_.getParam("orb_slam/orb_matcher/method", method_); // read and, as in the reference, not used
        nh_.getParam("orb_slam/brute_force_with_radius/max_matching_pixel_dist", max_matching_pixel_dist_);
        std::string type = "BruteForce-Hamming";
        nh_.getParam("orb_slam/cv_orb_matcher/type", type);
        if (type != "BruteForce-Hamming")
            throw std::runtime_error("ORBMatcher: only cv_orb_matcher/type BruteForce-Hamming is implemented");
        handle_ = orbx_compat::PerThreadHandle<orbx_matcher>(
            [device]() {
                orbx_matcher* h = nullptr;
                orbx_compat::check(orbx_matcher_create(device, &h), "ORBMatcher");
                return h;
            },
            orbx_matcher_destroy);
    }

    ~ORBMatcher() {}

    /** BruteForceWithRadiusMatcher path (orb_matcher.cpp:699-717). */
    void match(
        const std::vector<cv::KeyPoint>& key_points,
        const std::vector<cv::KeyPoint>& ref_key_points,
        const cv::Mat& descriptors,
        const cv::Mat& ref_descriptors,
        std::vector<cv::DMatch>& matches,
        bool filter_matches = true) const
    {
        std::vector<cv::DMatch> out(key_points.size());
        int count = 0;
        orbx_compat::check(
            orbx_match_radius(
                handle_.get(),
                reinterpret_cast<const orbx_keypoint*>(key_points.data()), (int)key_points.size(),
                reinterpret_cast<const orbx_keypoint*>(ref_key_points.data()), (int)ref_key_points.size(),
                descriptors.data, ref_descriptors.data, max_matching_pixel_dist_,
                reinterpret_cast<orbx_match*>(out.data()), &count),
            "ORBMatcher::match(radius)");
        out.resize(count);
        matches.insert(matches.end(), out.begin(), out.end()); // the reference push_backs
        if (filter_matches) filterMatches(descriptors, matches);
    }

    /** CVORBMatcher path = cv::BFMatcher(NORM_HAMMING) (orb_matcher.cpp:719-733). */
    void match(
        const cv::Mat& descriptors,
        const cv::Mat& ref_descriptors,
        std::vector<cv::DMatch>& matches,
        bool filter_matches = true) const
    {
        const int n = descriptors.rows;
        std::vector<int32_t> idx(n), dist(n);
        orbx_compat::check(
            orbx_match_hamming(
                handle_.get(), descriptors.data, n, ref_descriptors.data, ref_descriptors.rows,
                idx.data(), dist.data()),
            "ORBMatcher::match(hamming)");
        matches.clear();
        if (ref_descriptors.rows == 0) return;
        matches.reserve(n);
        for (int i = 0; i < n; ++i) {
            cv::DMatch m(i, idx[i], (float)dist[i]);
            m.imgIdx = 0; // cv::BFMatcher reports the index of the train image
            matches.push_back(m);
        }
        if (filter_matches) filterMatches(descriptors, matches);
    }

    /** Frame-to-frame projection matcher (orb_matcher.cpp:748-761, 76-93, 115-255). */
    template <typename FramePtrT>
    void matchByProjection(
        const FramePtrT& frame,
        const FramePtrT& ref_frame,
        std::vector<cv::DMatch>& matches,
        const bool check_orientation = true,
        const float radius = 1.0) const
    {
        const auto& key_points = frame->featuresUndist();
        const auto& ref_key_points = ref_frame->featuresUndist();
        const cv::Mat& descriptors = frame->descriptorsUndist();
        const auto& ref_map_points = ref_frame->obsMapPoints();
        const cv::Mat ref_T_f = ref_frame->worldInCameraT() * frame->cameraInWorldT();
        const float z = ref_T_f.template at<float>(2, 3);
        const int level_mode = z > 0.1f ? 1 : (-z > 0.1f ? 2 : 0);
        const int n = (int)key_points.size(), n_ref = (int)ref_key_points.size();
        std::vector<uint8_t> has(n_ref, 0);
        std::vector<float> xy(2 * (size_t)n_ref, 0.f), depth(n_ref, 1.f);
        cv::Mat ref_desc(n_ref, 32, CV_8U);
        for (int i = 0; i < n_ref; ++i) {
            const auto& mp = ref_map_points[i];
            if (!mp) continue;
            has[i] = 1;
            auto pc = frame->template worldToCamera<float>(mp->worldPos());
            depth[i] = pc.z;
            auto px = frame->template cameraToFrame<float, float>(pc);
            xy[2 * i] = px.x;
            xy[2 * i + 1] = px.y;
            std::memcpy(ref_desc.template ptr<uint8_t>(i), mp->bestDescriptor().data, 32);
        }
        orbx_grid g = gridOf(frame);
        std::vector<cv::KeyPoint> kps(key_points);
        std::vector<int32_t> cell_start((size_t)g.grid_cols * g.grid_rows + 1), cell_items(n);
        int kept = n;
        orbx_compat::check(
            orbx_assign_grid(&g, reinterpret_cast<orbx_keypoint*>(kps.data()), &kept, cell_start.data(), cell_items.data()),
            "matchByProjection(grid)");
        const auto& sf = frame->orbExtractor()->scaleFactors();
        std::vector<cv::DMatch> out(n);
        int count = 0;
        orbx_compat::check(
            orbx_match_projection_frame(
                handle_.get(), &g, cell_start.data(), cell_items.data(),
                reinterpret_cast<const orbx_keypoint*>(kps.data()), kept, descriptors.data,
                reinterpret_cast<const orbx_keypoint*>(ref_key_points.data()), n_ref, has.data(), xy.data(),
                depth.data(), ref_desc.data, sf.data(), (int)sf.size(), level_mode, radius, check_orientation ? 1 : 0,
                reinterpret_cast<orbx_match*>(out.data()), &count),
            "ORBMatcher::matchByProjection");
        matches.insert(matches.end(), out.begin(), out.begin() + count);
    }

    /**
     * BoW matcher (orb_matcher.cpp:735-746, 354-368, 386-497): features of `frame` against those of the key frame
     * that fall into the same vocabulary node; a reference feature takes part only if it has a map point that is
     * not bad.  The frame types provide featuresUndist(), descriptorsUndist() and bowFeatures() (an ordered map
     * node id -> feature indices, as DBoW2::FeatureVector is), the key frame frame() and obsMapPoints().
     */
    template <typename FramePtrT, typename KeyFramePtrT>
    void matchByBowFeatures(
        const FramePtrT& frame,
        const KeyFramePtrT& ref_key_frame,
        std::vector<cv::DMatch>& matches,
        const bool check_orientation = true,
        const float nn_ratio = 0.6) const
    {
        const auto& ref_frame = ref_key_frame->frame();
        const auto ref_map_points = ref_key_frame->obsMapPoints();
        const auto& key_points = frame->featuresUndist();
        const auto& ref_key_points = ref_frame->featuresUndist();
        const cv::Mat& descs = frame->descriptorsUndist();
        const cv::Mat& ref_descs = ref_frame->descriptorsUndist();
        const int n = (int)key_points.size(), n_ref = (int)ref_key_points.size();
        std::vector<uint8_t> has(n_ref, 0);
        for (int i = 0; i < n_ref && i < (int)ref_map_points.size(); ++i)
            has[i] = ref_map_points[i] && !ref_map_points[i]->isBad();
        std::vector<uint32_t> ids, ref_ids;
        std::vector<int32_t> start, items, ref_start, ref_items;
        flattenFeatureVector(frame->bowFeatures(), ids, start, items);
        flattenFeatureVector(ref_frame->bowFeatures(), ref_ids, ref_start, ref_items);
        std::vector<cv::DMatch> out(std::max(n, 1));
        int count = 0;
        const int low_threshold = 50;  // orb_matcher.h:201
        orbx_compat::check(
            orbx_match_bow(
                handle_.get(), reinterpret_cast<const orbx_keypoint*>(key_points.data()), n, descs.data,
                reinterpret_cast<const orbx_keypoint*>(ref_key_points.data()), n_ref, ref_descs.data, has.data(),
                ids.data(), start.data(), items.data(), (int)ids.size(), ref_ids.data(), ref_start.data(),
                ref_items.data(), (int)ref_ids.size(), low_threshold, nn_ratio, check_orientation ? 1 : 0,
                reinterpret_cast<orbx_match*>(out.data()), &count),
            "ORBMatcher::matchByBowFeatures");
        matches.insert(matches.end(), out.begin(), out.begin() + count);
    }

    /**
     * Map-point projection matcher (orb_matcher.cpp:763-778, 257-352).  With compute_track_info the track
     * properties are refreshed through the frame's own isInCameraView(mp, props, 0.5) exactly as the reference does
     * (:273-279); the search itself runs on the device.  The map point type provides resetTrackProperties(),
     * setTrackProperties(), trackProperties() (in_view_, proj_xy_, pred_scale_level_, view_cosine_), isBad() and
     * bestDescriptor().
     */
    template <typename FramePtrT, typename MapPointPtrT>
    void matchByProjection(
        const FramePtrT& frame,
        const std::vector<MapPointPtrT>& points_3d,
        std::vector<cv::DMatch>& matches,
        const bool compute_track_info,
        const float nn_ratio = 0.6,
        const float radius = 1.0) const
    {
        const auto& key_points = frame->featuresUndist();
        const cv::Mat& descriptors = frame->descriptorsUndist();
        const int n = (int)key_points.size(), n_pts = (int)points_3d.size();
        std::vector<uint8_t> in_view(std::max(n_pts, 1), 0);
        std::vector<float> xy(2 * (size_t)std::max(n_pts, 1), 0.f), view_cos(std::max(n_pts, 1), 0.f);
        std::vector<int32_t> pred(std::max(n_pts, 1), 0);
        cv::Mat pt_desc(std::max(n_pts, 1), 32, CV_8U);
        for (int i = 0; i < n_pts; ++i) {
            const auto& mp = points_3d[i];
            if (!mp) continue;
            if (compute_track_info) {
                typename std::decay<decltype(mp->trackProperties())>::type track_results;
                mp->resetTrackProperties();
                if (frame->isInCameraView(mp, track_results, 0.5))
                    mp->setTrackProperties(track_results);
            }
            const auto& props = mp->trackProperties();
            if (!props.in_view_ || mp->isBad()) continue;
            in_view[i] = 1;
            xy[2 * i] = props.proj_xy_.x;
            xy[2 * i + 1] = props.proj_xy_.y;
            pred[i] = props.pred_scale_level_;
            view_cos[i] = props.view_cosine_;
            std::memcpy(pt_desc.template ptr<uint8_t>(i), mp->bestDescriptor().data, 32);
        }
        orbx_grid g = gridOf(frame);
        std::vector<cv::KeyPoint> kps(key_points);
        std::vector<int32_t> cell_start((size_t)g.grid_cols * g.grid_rows + 1), cell_items(std::max(n, 1));
        int kept = n;
        orbx_compat::check(
            orbx_assign_grid(&g, reinterpret_cast<orbx_keypoint*>(kps.data()), &kept, cell_start.data(), cell_items.data()),
            "matchByProjection(grid)");
        const auto& sf = frame->orbExtractor()->scaleFactors();
        std::vector<cv::DMatch> out(std::max(n, 1));
        int count = 0;
        orbx_compat::check(
            orbx_match_projection_points(
                handle_.get(), &g, cell_start.data(), cell_items.data(),
                reinterpret_cast<const orbx_keypoint*>(kps.data()), kept, descriptors.data, n_pts, in_view.data(),
                xy.data(), pred.data(), view_cos.data(), pt_desc.data, sf.data(), (int)sf.size(), radius, nn_ratio,
                reinterpret_cast<orbx_match*>(out.data()), &count),
            "ORBMatcher::matchByProjection(points)");
        matches.insert(matches.end(), out.begin(), out.begin() + count);
    }

    /**
     * Epipolar matcher between two key frames (orb_matcher.cpp:780-791, 499-661).  The fundamental matrix comes
     * from the key frame's own computeFundamentalMat(F, ref_key_frame), as in the reference (:508-510).
     */
    template <typename KeyFramePtrT>
    void matchByEpipolarConstraint(
        const KeyFramePtrT& key_frame,
        const KeyFramePtrT& ref_key_frame,
        std::vector<cv::DMatch>& matches,
        const bool check_orientation = true) const
    {
        const auto map_points = key_frame->obsMapPoints();
        const auto ref_map_points = ref_key_frame->obsMapPoints();
        cv::Mat fundamental_mat;
        key_frame->computeFundamentalMat(fundamental_mat, ref_key_frame);
        const auto& frame = key_frame->frame();
        const auto& ref_frame = ref_key_frame->frame();
        const auto& key_points = frame->featuresUndist();
        const auto& ref_key_points = ref_frame->featuresUndist();
        const cv::Mat& descs = frame->descriptorsUndist();
        const cv::Mat& ref_descs = ref_frame->descriptorsUndist();
        const int n = (int)key_points.size(), n_ref = (int)ref_key_points.size();
        std::vector<uint8_t> has(std::max(n, 1), 0), ref_has(std::max(n_ref, 1), 0);
        for (int i = 0; i < n && i < (int)map_points.size(); ++i) has[i] = map_points[i] ? 1 : 0;
        for (int i = 0; i < n_ref && i < (int)ref_map_points.size(); ++i) ref_has[i] = ref_map_points[i] ? 1 : 0;
        std::vector<uint32_t> ids, ref_ids;
        std::vector<int32_t> start, items, ref_start, ref_items;
        flattenFeatureVector(frame->bowFeatures(), ids, start, items);
        flattenFeatureVector(ref_frame->bowFeatures(), ref_ids, ref_start, ref_items);
        float f9[9];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) f9[3 * r + c] = fundamental_mat.template at<float>(r, c);
        const auto& sigmas = ref_frame->orbExtractor()->scaleSigmas();
        std::vector<cv::DMatch> out(std::max(n, 1));
        int count = 0;
        const int low_threshold = 50;  // orb_matcher.h:201
        orbx_compat::check(
            orbx_match_epipolar(
                handle_.get(), reinterpret_cast<const orbx_keypoint*>(key_points.data()), n, descs.data,
                reinterpret_cast<const orbx_keypoint*>(ref_key_points.data()), n_ref, ref_descs.data, has.data(),
                ref_has.data(), f9, sigmas.data(), (int)sigmas.size(), ids.data(), start.data(), items.data(),
                (int)ids.size(), ref_ids.data(), ref_start.data(), ref_items.data(), (int)ref_ids.size(),
                low_threshold, check_orientation ? 1 : 0, reinterpret_cast<orbx_match*>(out.data()), &count),
            "ORBMatcher::matchByEpipolarConstraint");
        matches.insert(matches.end(), out.begin(), out.begin() + count);
    }

    /**
     * ORBMatcher::fuse (orb_matcher.cpp:793-888).  The per-point tests that read the map (:808-814) and the map
     * bookkeeping (:869-884) run here on the host through the caller's own types, in the reference's order; the
     * window search and descriptor comparison (:818-867) run on the device for all points at once.  The search of
     * a point depends on the map only through isBad()/isObservedInKeyFrame(), so those two are evaluated again at
     * the point's turn in the bookkeeping loop: the result equals the reference's sequential loop, duplicates in
     * points_3d included.
     */
    template <typename KeyFramePtrT, typename MapPointPtrT>
    int fuse(
        const KeyFramePtrT key_frame,
        const std::vector<MapPointPtrT>& points_3d,
        const float radius_multiplier = 3.0) const
    {
        const auto map_points = key_frame->obsMapPoints();
        const auto& frame = key_frame->frame();
        const auto& key_points = frame->featuresUndist();
        const cv::Mat& descriptors = frame->descriptorsUndist();
        const int n = (int)key_points.size(), n_pts = (int)points_3d.size();
        key_frame->updateWorldInCamLocal();
        std::vector<uint8_t> in_view(std::max(n_pts, 1), 0);
        std::vector<float> xy(2 * (size_t)std::max(n_pts, 1), 0.f);
        std::vector<int32_t> pred(std::max(n_pts, 1), 0);
        cv::Mat pt_desc(std::max(n_pts, 1), 32, CV_8U);
        for (int i = 0; i < n_pts; ++i) {
            const auto& mp = points_3d[i];
            if (!mp || mp->isBad() || mp->isObservedInKeyFrame(key_frame)) continue;
            typename std::decay<decltype(mp->trackProperties())>::type props;
            if (!key_frame->isInCameraViewLocal(mp, props, 0.5)) continue;
            in_view[i] = 1;
            xy[2 * i] = props.proj_xy_.x;
            xy[2 * i + 1] = props.proj_xy_.y;
            pred[i] = props.pred_scale_level_;
            std::memcpy(pt_desc.template ptr<uint8_t>(i), mp->bestDescriptor().data, 32);
        }
        orbx_grid g = gridOf(frame);
        std::vector<cv::KeyPoint> kps(key_points);
        std::vector<int32_t> cell_start((size_t)g.grid_cols * g.grid_rows + 1), cell_items(std::max(n, 1));
        int kept = n;
        orbx_compat::check(
            orbx_assign_grid(&g, reinterpret_cast<orbx_keypoint*>(kps.data()), &kept, cell_start.data(), cell_items.data()),
            "fuse(grid)");
        if (kept != n)
            throw std::runtime_error("ORBMatcher::fuse: key frame features outside the image bounds");
        const auto& sf = frame->orbExtractor()->scaleFactors();
        const auto& inv_sigmas = frame->orbExtractor()->invScaleSigmas();
        std::vector<int32_t> best_idx(std::max(n_pts, 1), -1), best_dist(std::max(n_pts, 1), 256);
        orbx_compat::check(
            orbx_fuse_search(
                handle_.get(), &g, cell_start.data(), cell_items.data(),
                reinterpret_cast<const orbx_keypoint*>(kps.data()), kept, descriptors.data, n_pts, in_view.data(),
                xy.data(), pred.data(), pt_desc.data, sf.data(), inv_sigmas.data(), (int)sf.size(), radius_multiplier,
                best_idx.data(), best_dist.data()),
            "ORBMatcher::fuse");
        const int low_threshold = 50;  // orb_matcher.h:201
        int n_fused = 0;
        for (int i = 0; i < n_pts; ++i) {
            if (!in_view[i] || best_idx[i] < 0) continue;
            // the reference skips a point with an empty window before the threshold test (:824-832) and counts
            // every point whose best distance passes it (:870-884)
            if (best_dist[i] > low_threshold) continue;
            const auto& mp = points_3d[i];
            if (mp->isBad() || mp->isObservedInKeyFrame(key_frame)) continue;
            const int matched_id = best_idx[i];
            auto mp_kf = map_points[matched_id];
            if (mp_kf) {
                if (!mp_kf->isBad()) {
                    if (mp_kf->nObservations() > mp->nObservations()) mp->replace(mp_kf);
                    else mp_kf->replace(mp);
                }
            } else {
                mp->addObservation(key_frame, matched_id);
                key_frame->setMapPointAt(mp, matched_id);
            }
            n_fused++;
        }
        return n_fused;
    }

    /** ORBMatcher::filterMatches (orb_matcher.cpp:890-924). */
    void filterMatches(
        const cv::Mat& descriptors,
        std::vector<cv::DMatch>& matches) const
    {
        int count = (int)matches.size();
        orbx_compat::check(
            orbx_filter_matches(reinterpret_cast<orbx_match*>(matches.data()), &count, descriptors.rows),
            "ORBMatcher::filterMatches");
        matches.resize(count);
    }

private:
    //! DBoW2::FeatureVector-like ordered map (node id -> feature indices) as the CSR arrays of the C ABI
    template <typename FeatureVectorT>
    static void flattenFeatureVector(const FeatureVectorT& fv, std::vector<uint32_t>& ids,
                                     std::vector<int32_t>& start, std::vector<int32_t>& items) {
        start.push_back(0);
        for (const auto& node : fv) {
            ids.push_back((uint32_t)node.first);
            for (auto i : node.second) items.push_back((int32_t)i);
            start.push_back((int32_t)items.size());
        }
    }

    template <typename FramePtrT>
    static orbx_grid gridOf(const FramePtrT& frame) {
        const auto& cam = frame->camera();
        orbx_grid g;
        g.min_x = cam->minX(); g.min_y = cam->minY(); g.max_x = cam->maxX(); g.max_y = cam->maxY();
        g.grid_cols = frame->gridCols(); g.grid_rows = frame->gridRows();
        g.grid_size_x = cam->undistWidth() / g.grid_cols;   // int = float / int, Frame::setupGrid
        g.grid_size_y = cam->undistHeight() / g.grid_rows;
        return g;
    }

    ros::NodeHandle nh_;
    std::string method_;
    double max_matching_pixel_dist_ = 0.0;
    //! one device-side matcher per calling thread: the reference shares this object between the tracker and the
    //! local-mapper thread (frame.h:521-522, map_point.h:240-241)
    orbx_compat::PerThreadHandle<orbx_matcher> handle_;
};

using ORBMatcherPtr = std::shared_ptr<ORBMatcher>;
using ORBMatcherConstPtr = std::shared_ptr<const ORBMatcher>;

} // namespace geometry

} // namespace orb_slam
