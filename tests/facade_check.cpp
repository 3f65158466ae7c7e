// Host-only compile/link check of the drop-in headers against the C ABI, plus a GPU run when asked (argv[1]=run).
#include <cmath>
#include <cstdio>
#include <orb_slam/geometry/orb_extractor.h>
#include <orb_slam/geometry/orb_matcher.h>
#include <orb_slam/geometry/utils.h>
#include <orb_slam/initializer.h>

#include <map>
#include <set>
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

// 3x3 CV_32F products / transpose / inverse for the transcribed reference tests (the stand-in cv::Mat has no operators)
static cv::Mat mul3(const cv::Mat& a, const cv::Mat& b) {
    cv::Mat c(3, 3, CV_32F);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double t = 0;
            for (int k = 0; k < 3; k++) t += (double)a.at<float>(i, k) * b.at<float>(k, j);
            c.at<float>(i, j) = (float)t;
        }
    return c;
}
static cv::Mat transpose3(const cv::Mat& a) {
    cv::Mat c(3, 3, CV_32F);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) c.at<float>(j, i) = a.at<float>(i, j);
    return c;
}
static cv::Mat inverse3(const cv::Mat& a) {
    double m[9], d;
    for (int i = 0; i < 9; i++) m[i] = a.at<float>(i / 3, i % 3);
    d = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
    const double c[9] = {m[4] * m[8] - m[5] * m[7], m[2] * m[7] - m[1] * m[8], m[1] * m[5] - m[2] * m[4],
                         m[5] * m[6] - m[3] * m[8], m[0] * m[8] - m[2] * m[6], m[2] * m[3] - m[0] * m[5],
                         m[3] * m[7] - m[4] * m[6], m[1] * m[6] - m[0] * m[7], m[0] * m[4] - m[1] * m[3]};
    cv::Mat out(3, 3, CV_32F);
    for (int i = 0; i < 9; i++) out.at<float>(i / 3, i % 3) = (float)(c[i] / d);
    return out;
}
static double distanceToGolden(cv::Mat m, const float* golden) {
    if (std::fabs(m.at<float>(2, 2)) > 1e-8)
        for (int i = 0; i < 9; i++) m.at<float>(i / 3, i % 3) *= (float)(1. / m.at<float>(2, 2));
    double s = 0;
    for (int i = 0; i < 9; i++) { const double d = (double)m.at<float>(i / 3, i % 3) - golden[i]; s += d * d; }
    return std::sqrt(s);
}
static void printBits(const char* name, const cv::Mat& m) {
    std::printf("%s=", name);
    for (int i = 0; i < 9; i++) { uint32_t w; std::memcpy(&w, &m.at<float>(i / 3, i % 3), 4); std::printf("%08x%s", w, i < 8 ? "," : "\n"); }
}

// The bodies of the reference's src/tests/geometry/utils_tests.cpp:71-147 against the facade's geometry/utils.h.
// `path` holds the reference's point tables and golden matrices: n, then n rows "px py rx ry", then F (9), then H (9).
static void utilsTests(const char* path) {
    std::ifstream f(path);
    int n = 0;
    f >> n;
    std::vector<cv::Point2f> points(n), ref_points(n);
    for (int i = 0; i < n; i++) f >> points[i].x >> points[i].y >> ref_points[i].x >> ref_points[i].y;
    float cv_f_mat[9], cv_h_mat[9];
    for (float& v : cv_f_mat) f >> v;
    for (float& v : cv_h_mat) f >> v;
    if (!f) { std::printf("utils_points unreadable\n"); return; }
    {   // TestFundamentalMatrix
        std::vector<cv::Point2f> points_norm, ref_points_norm;
        cv::Mat T, ref_T, f_norm;
        normalizePoints(points, points_norm, T);
        normalizePoints(ref_points, ref_points_norm, ref_T);
        computeFundamentalMat(f_norm, points_norm, ref_points_norm);
        cv::Mat f_mat = mul3(mul3(transpose3(T), f_norm), ref_T);
        std::printf("utils_F_err=%.9g n=%d\n", distanceToGolden(f_mat, cv_f_mat), n);
        printBits("utils_F_norm", f_norm);
        printBits("utils_T", T);
    }
    {   // TestHomographyMatrix
        std::vector<cv::Point2f> points_norm, ref_points_norm;
        cv::Mat T, ref_T, h_norm;
        normalizePoints(points, points_norm, T);
        normalizePoints(ref_points, ref_points_norm, ref_T);
        computeHomographyMat(h_norm, points_norm, ref_points_norm);
        cv::Mat h_mat = mul3(mul3(inverse3(T), h_norm), ref_T);
        std::printf("utils_H_err=%.9g\n", distanceToGolden(h_mat, cv_h_mat));
        printBits("utils_H_norm", h_norm);
    }
}

// TestComputeThreeMaxima (utils_tests.cpp:123-147) plus the two helpers of namespace utils; host only
static void utilsHostTests() {
    const int hist_l = 30;
    std::vector<int> hist[hist_l];
    const int fill[8] = {10, 1, 20, 15, 9, 8, 7, 5};
    for (int b = 0; b < 8; b++) hist[b].assign(fill[b], 1);
    int m1, m2, m3;
    computeThreeMaxima(hist, hist_l, m1, m2, m3);
    std::printf("maxima=%d,%d,%d\n", m1, m2, m3);
    cv::Mat d(4, 32, CV_8U);
    for (int i = 0; i < 4 * 32; i++) d.data[i] = (unsigned char)(i * 37 + 11);
    std::vector<cv::Mat> rows;
    orb_slam::utils::matToVectorMat(d, rows);   // as Frame::computeBow calls it (src/frame.cpp:316)
    bool shared = rows.size() == 4;
    for (int r = 0; shared && r < 4; r++) shared = rows[r].rows == 1 && rows[r].cols == 32 && rows[r].data == d.data + 32 * r;
    int bits = 0;
    for (int i = 0; i < 32; i++) bits += __builtin_popcount(d.data[i] ^ d.data[32 + i]);
    std::printf("rows_shared=%d distance_ok=%d\n", (int)shared, (int)(descriptorDistance(rows[0], rows[1]) == bits));
    cv::Mat v(3, 1, CV_32F);
    v.at<float>(0, 0) = 1.f; v.at<float>(1, 0) = 2.f; v.at<float>(2, 0) = 3.f;
    cv::Mat k = orb_slam::utils::skew(v);
    const float want[9] = {0, -3, 2, 3, 0, -1, -2, 1, 0};
    bool ok = true;
    for (int i = 0; i < 9; i++) ok = ok && k.at<float>(i / 3, i % 3) == want[i];
    std::printf("skew_ok=%d\n", (int)ok);
}

int main(int argc, char** argv) {
    ros::NodeHandle nh;
    if (argc > 2) {
        nh.loadYaml(std::string(argv[2]) + "/orb_extractor.yaml");
        nh.loadYaml(std::string(argv[2]) + "/orb_matcher.yaml");
    }
    int n_kp = -1;
    nh.getParam("orb_slam/orb_extractor/n_key_points", n_kp);
    std::printf("n_key_points=%d\n", n_kp);
    utilsHostTests();
    if (argc < 2 || std::string(argv[1]) != "run") return 0;
    if (argc > 3) utilsTests(argv[3]);
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
        // the three additive keys: the same class with per-cell adaptive thresholds and the quad-tree distribution
        ros::NodeHandle nh_q = nh;
        nh_q.set("orb_slam/orb_extractor/distribution", "1");
        nh_q.set("orb_slam/orb_extractor/low_score_threshold", "7");
        nh_q.set("orb_slam/orb_extractor/distribution_cell_size", "30");
        ORBExtractor exq(nh_q, 640, 480);
        std::vector<cv::KeyPoint> kq;
        cv::Mat dq;
        exq.detect(img, kq);
        exq.compute(img, kq, dq);
        std::set<int> cells_q, cells_o;
        for (auto& k : kq) cells_q.insert((int)k.pt.x / 40 + 100 * ((int)k.pt.y / 40));
        for (auto& k : k1) cells_o.insert((int)k.pt.x / 40 + 100 * ((int)k.pt.y / 40));
        float max_resp = 0;
        for (auto& k : kq) max_resp = std::max(max_resp, k.response);
        std::printf("dist_kp=%zu dist_desc=%d dist_cells=%zu orb_cells=%zu dist_max_response=%.0f\n", kq.size(), dq.rows,
                    cells_q.size(), cells_o.size(), max_resp);
    }
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
        // the homography branch: refused as in the reference unless the additive switch is set
        cv::Mat Rh, th;
        const bool h_default = init3.findRtWithHomography(ip, ir, Rh, th);
        const bool untouched = Rh.empty() && th.empty();
        init3.setHomographyDecomposition(true);
        const bool h_switch = init3.findRtWithHomography(ip, ir, Rh, th);
        double det = 1.0;
        if (!Rh.empty()) {
            const double* r = Rh.ptr<double>();
            det = r[0] * (r[4] * r[8] - r[5] * r[7]) - r[1] * (r[3] * r[8] - r[5] * r[6]) + r[2] * (r[3] * r[7] - r[4] * r[6]);
        }
        std::printf("h_default=%d untouched=%d h_switch=%d h_shape_ok=%d h_det_ok=%d\n", (int)h_default, (int)untouched,
                    (int)h_switch, (int)(Rh.empty() == th.empty()), (int)(std::fabs(det - 1.0) < 1e-9));
    }
    return 0;
}
