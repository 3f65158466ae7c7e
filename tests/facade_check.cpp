// Host-only compile/link check of the drop-in headers against the C ABI, plus a GPU run when asked (argv[1]=run).
#include <cmath>
#include <cstdio>
#include <orb_slam/geometry/orb_extractor.h>
#include <orb_slam/geometry/orb_matcher.h>
#include <orb_slam/initializer.h>

#include <map>
#include <memory>
#include <thread>

using namespace orb_slam;
using namespace orb_slam::geometry;

// the slice of Frame / KeyFrame / MapPoint / Camera that the ORBMatcher templates touch
struct MockTrackProperties {
    bool in_view_ = false;
    cv::Point2f proj_xy_;
    int pred_scale_level_ = 0;
    float view_cosine_ = 1.f;
};
struct MockKeyFrame;
struct MockMapPoint {
    bool bad = false;
    int n_obs = 1;
    cv::Point2f img;     // where the mock frustum test projects the point
    int level = 0;
    cv::Mat desc;
    MockTrackProperties props;
    std::vector<std::pair<const MockKeyFrame*, int>> obs;
    bool isBad() const { return bad; }
    void resetTrackProperties() { props.in_view_ = false; }
    void setTrackProperties(const MockTrackProperties& p) { props = p; }
    const MockTrackProperties& trackProperties() const { return props; }
    const cv::Mat& bestDescriptor() const { return desc; }
    int nObservations() const { return n_obs; }
    bool isObservedInKeyFrame(const std::shared_ptr<MockKeyFrame>& kf) const {
        for (auto& o : obs) if (o.first == kf.get()) return true;
        return false;
    }
    void addObservation(const std::shared_ptr<MockKeyFrame>& kf, int idx) { obs.emplace_back(kf.get(), idx); n_obs++; }
    void replace(const std::shared_ptr<MockMapPoint>&) { bad = true; }
};
struct MockCamera {
    cv::Mat K;
    MockCamera() {
        K.create(3, 3, CV_32F);
        const float kv[9] = {525.f, 0.f, 319.5f, 0.f, 525.f, 239.5f, 0.f, 0.f, 1.f};
        std::memcpy(K.data, kv, sizeof(kv));
    }
    const cv::Mat& intrinsicMatrix() const { return K; }
    float minX() const { return 0.f; }
    float minY() const { return 0.f; }
    float maxX() const { return 640.f; }
    float maxY() const { return 480.f; }
    float undistWidth() const { return 640.f; }
    float undistHeight() const { return 480.f; }
};
struct MockFrame {
    std::vector<cv::KeyPoint> kps;
    cv::Mat desc;
    std::map<unsigned, std::vector<unsigned>> bow;
    std::shared_ptr<MockCamera> cam = std::make_shared<MockCamera>();
    std::shared_ptr<ORBExtractor> extractor;
    std::vector<cv::DMatch> matches_;
    const std::vector<cv::DMatch>& matches() const { return matches_; }
    const int nMatches() const { return (int)matches_.size(); }
    const std::vector<cv::KeyPoint>& featuresUndist() const { return kps; }
    const cv::Mat& descriptorsUndist() const { return desc; }
    const std::map<unsigned, std::vector<unsigned>>& bowFeatures() const { return bow; }
    const std::shared_ptr<MockCamera>& camera() const { return cam; }
    const std::shared_ptr<ORBExtractor>& orbExtractor() const { return extractor; }
    int gridCols() const { return 64; }
    int gridRows() const { return 64; }
    bool isInCameraView(const std::shared_ptr<MockMapPoint>& mp, MockTrackProperties& p, float) const {
        p.in_view_ = true; p.proj_xy_ = mp->img; p.pred_scale_level_ = mp->level; p.view_cosine_ = 1.f;
        return true;
    }
};
struct MockKeyFrame {
    std::shared_ptr<MockFrame> f;
    std::vector<std::shared_ptr<MockMapPoint>> mps;
    const std::shared_ptr<MockFrame>& frame() const { return f; }
    std::vector<std::shared_ptr<MockMapPoint>> obsMapPoints() const { return mps; }
    void updateWorldInCamLocal() {}
    bool isInCameraViewLocal(const std::shared_ptr<MockMapPoint>& mp, MockTrackProperties& p, float lim) const {
        return f->isInCameraView(mp, p, lim);
    }
    void setMapPointAt(const std::shared_ptr<MockMapPoint>& mp, int idx) { mps[idx] = mp; }
    void computeFundamentalMat(cv::Mat& F, const std::shared_ptr<MockKeyFrame>&) const {
        F.create(3, 3, CV_32F);  // pure translation along x: epipolar lines are the image rows
        const float v[9] = {0, 0, 0, 0, 0, -1, 0, 1, 0};
        std::memcpy(F.data, v, sizeof(v));
    }
};

int main(int argc, char** argv) {
    ros::NodeHandle nh;
    if (argc > 2) {
        nh.loadYaml(std::string(argv[2]) + "/orb_extractor.yaml");
        nh.loadYaml(std::string(argv[2]) + "/orb_matcher.yaml");
    }
    int n_kp = -1;
    nh.getParam("orb_slam/orb_extractor/n_key_points", n_kp);
    std::printf("n_key_points=%d\n", n_kp);
    if (argc < 2 || std::string(argv[1]) != "run") return 0;
    ORBExtractor ex(nh, 640, 480);
    ORBMatcher mt(nh);
    cv::Mat img(480, 640, CV_8U);
    unsigned s = 1;
    for (int i = 0; i < 480 * 640; i++) { s = s * 1664525u + 1013904223u; img.data[i] = (i / 640 / 16 + i % 640 / 16) % 2 ? 40 + (s >> 28) : 200 - (s >> 28); }
    std::vector<cv::KeyPoint> k1, k2;
    cv::Mat d1, d2;
    ex.detect(img, k1);
    ex.compute(img, k1, d1);
    k2 = k1; d2 = d1.clone();
    std::vector<cv::DMatch> m;
    mt.match(d1, d2, m, true);
    std::printf("kp=%zu desc=%d matches=%zu levels=%d sf1=%.9g\n", k1.size(), d1.rows, m.size(), ex.levels(), ex.scaleFactors()[1]);
    {
        auto fa = std::make_shared<MockFrame>(), fb = std::make_shared<MockFrame>();
        fa->kps = k1; fa->desc = d1; fb->kps = k2; fb->desc = d2;
        for (unsigned i = 0; i < (unsigned)k1.size(); i++) { fa->bow[d1.data[32 * i] % 37u].push_back(i); fb->bow[d2.data[32 * i] % 37u].push_back(i); }
        auto kf = std::make_shared<MockKeyFrame>();
        kf->f = fb;
        for (size_t i = 0; i < k2.size(); i++) kf->mps.push_back(i % 5 ? std::make_shared<MockMapPoint>() : nullptr);
        std::vector<cv::DMatch> mb;
        mt.matchByBowFeatures(fa, kf, mb, true, 0.6f);
        std::printf("bow_matches=%zu\n", mb.size());
        // map-point projection matcher, epipolar matcher and fuse on the same features
        auto shared_ex = std::make_shared<ORBExtractor>(nh, 640, 480);
        fa->extractor = shared_ex; fb->extractor = shared_ex;
        std::vector<std::shared_ptr<MockMapPoint>> pts;
        for (size_t i = 0; i < k1.size(); i++) {
            auto mp = std::make_shared<MockMapPoint>();
            mp->img = k1[i].pt; mp->level = k1[i].octave;
            mp->desc.create(1, 32, CV_8U);
            std::memcpy(mp->desc.data, d1.data + 32 * i, 32);
            pts.push_back(i % 7 ? mp : nullptr);
        }
        std::vector<cv::DMatch> mp_matches;
        mt.matchByProjection(fa, pts, mp_matches, true, 0.6f, 1.0f);
        std::printf("point_matches=%zu\n", mp_matches.size());
        auto ka = std::make_shared<MockKeyFrame>(), kb = std::make_shared<MockKeyFrame>();
        ka->f = fa; kb->f = fb;
        ka->mps.assign(k1.size(), nullptr); kb->mps.assign(k2.size(), nullptr);
        std::vector<cv::DMatch> me;
        mt.matchByEpipolarConstraint(ka, kb, me, true);
        std::printf("epipolar_matches=%zu\n", me.size());
        const int fused = mt.fuse(ka, pts, 3.0f);
        size_t attached = 0;
        for (auto& m2 : ka->mps) attached += m2 ? 1 : 0;
        const int fused_again = mt.fuse(ka, pts, 3.0f);
        std::printf("fused=%d attached=%zu fused_again=%d\n", fused, attached, fused_again);
    }
    {
        // the reference shares one ORBMatcher / ORBExtractor between the tracker thread and the local-mapper thread
        // (frame.h:520-522): the same calls from two threads at once must give what they give alone
        std::vector<cv::DMatch> ma[2];
        std::vector<cv::KeyPoint> ka[2];
        cv::Mat da[2];
        bool same = true;
        for (int round = 0; round < 3 && same; round++) {
            std::thread th[2];
            for (int t = 0; t < 2; t++)
                th[t] = std::thread([&, t]() {
                    for (int rep = 0; rep < 4; rep++) {
                        ma[t].clear();
                        if (t == 0) mt.match(d1, d2, ma[t], true);
                        else mt.match(k1, k2, d1, d2, ma[t], false);
                        ex.detect(img, ka[t]);
                        ex.compute(img, ka[t], da[t]);
                    }
                });
            for (auto& x : th) x.join();
            std::vector<cv::DMatch> mr;
            mt.match(k1, k2, d1, d2, mr, false);
            same = ma[0].size() == m.size() && ma[1].size() == mr.size();
            for (size_t i = 0; same && i < m.size(); i++) same = ma[0][i].trainIdx == m[i].trainIdx && ma[0][i].queryIdx == m[i].queryIdx;
            for (size_t i = 0; same && i < mr.size(); i++) same = ma[1][i].trainIdx == mr[i].trainIdx && ma[1][i].distance == mr[i].distance;
            for (int t = 0; t < 2 && same; t++)
                same = ka[t].size() == k1.size() && da[t].rows == d1.rows &&
                       std::memcmp(da[t].data, d1.data, (size_t)d1.rows * 32) == 0 &&
                       std::memcmp(ka[t].data(), k1.data(), k1.size() * sizeof(cv::KeyPoint)) == 0;
        }
        std::printf("two_threads_same=%d\n", (int)same);
    }
    // Initializer through the reference's own constructor and tryToInitialize (mono_tracker.cpp:81-87,127-134)
    auto ref_frame = std::make_shared<MockFrame>(), cur_frame = std::make_shared<MockFrame>();
    auto camera = std::make_shared<MockCamera>();
    ref_frame->kps = k2;
    cur_frame->kps = k1;
    cur_frame->matches_ = m;
    InitializerPtr init = InitializerPtr(new Initializer(ref_frame, camera, 1.0, 50));
    if (m.size() >= 8) {
        cv::Mat R0, t0;
        std::vector<cv::Point2d> ip0, ir0;
        std::vector<cv::Point3d> x0;
        std::vector<size_t> idx0;
        // identical views: every hypothesis is degenerate in the same way for F; the call must come back either way
        const bool ok0 = init->tryToInitialize(cur_frame, R0, t0, ip0, ir0, x0, idx0);
        std::printf("H empty=%d init_identical=%d\n", (int)init->getHomographyMat().empty(), (int)ok0);
        // a synthetic rigid scene: matches, F/H RANSAC with injected sets, pose recovery, triangulation
        auto fr = std::make_shared<MockFrame>(), fc = std::make_shared<MockFrame>();
        unsigned q = 7;
        auto rnd = [&]() { q = q * 1664525u + 1013904223u; return (q >> 8) / 16777216.0; };
        for (int i = 0; i < 300; i++) {
            const double X = rnd() * 4 - 2, Y = rnd() * 3 - 1.5, Z = 2 + rnd() * 4, a = 0.05;
            const double x2 = std::cos(a) * X + std::sin(a) * Z + 0.3, y2 = Y + 0.02, z2 = -std::sin(a) * X + std::cos(a) * Z + 0.05;
            cv::KeyPoint kr, kc;
            kr.pt = cv::Point2f((float)(525 * X / Z + 319.5), (float)(525 * Y / Z + 239.5));
            kc.pt = cv::Point2f((float)(525 * x2 / z2 + 319.5), (float)(525 * y2 / z2 + 239.5));
            // the match list pairs query 299 - i of the current frame with train i of the reference frame
            fr->kps.push_back(kr);
            fc->kps.insert(fc->kps.begin(), kc);
            fc->matches_.push_back(cv::DMatch(299 - i, i, 0.f));
        }
        Initializer init2(fr, camera, 1.0, 200);
        std::vector<std::vector<size_t>> sets(200, std::vector<size_t>(8));
        for (auto& st : sets)
            for (int j = 0; j < 8; j++) {
                bool again = true;
                while (again) {
                    st[j] = (size_t)(rnd() * 300) % 300;
                    again = false;
                    for (int k = 0; k < j; k++) again = again || st[k] == st[j];
                }
            }
        init2.setRansacSets(sets);
        cv::Mat Rm, tm;
        std::vector<cv::Point2d> ip, ir;
        std::vector<cv::Point3d> xyz;
        std::vector<size_t> idx;
        const bool ok = init2.tryToInitialize(fc, Rm, tm, ip, ir, xyz, idx);
        int in_front = 0;
        for (auto& P3 : xyz) in_front += P3.z > 0;
        std::printf("pose ok=%d tri=%d inliers=%zu in_front=%d r00=%.4f tx=%.3f\n", (int)ok, (int)(xyz.size() == ip.size()),
                    ip.size(), in_front, ok ? Rm.at<double>(0, 0) : 0.0, ok ? tm.at<double>(0, 0) : 0.0);
        // the test seam of the reference: points set directly, then the public solver and the getters
        std::vector<cv::Point2f> sp, sr;
        for (auto& mm : fc->matches_) { sp.push_back(fc->kps[mm.queryIdx].pt); sr.push_back(fr->kps[mm.trainIdx].pt); }
        Initializer init3(fr, camera, 1.0, 200);
        init3.setRansacSets(sets);
        init3.setPoints(sp); init3.setRefPoints(sr);
        init3.findFundamentalAndHomography();
        cv::Mat F2 = init2.getFundamentalMat(), F3 = init3.getFundamentalMat();
        std::printf("seam_same_F=%d\n", (int)(!F2.empty() && !F3.empty() && std::memcmp(F2.data, F3.data, 36) == 0));
    }
    return 0;
}
