// Host-only compile/link check of the drop-in headers against the C ABI, plus a GPU run when asked (argv[1]=run).
#include <cstdio>
#include <orb_slam/geometry/orb_extractor.h>
#include <orb_slam/geometry/orb_matcher.h>
#include <orb_slam/initializer.h>

#include <map>
#include <memory>

using namespace orb_slam;
using namespace orb_slam::geometry;

// the slice of Frame / KeyFrame / MapPoint that ORBMatcher::matchByBowFeatures touches
struct MockMapPoint { bool bad = false; bool isBad() const { return bad; } };
struct MockFrame {
    std::vector<cv::KeyPoint> kps;
    cv::Mat desc;
    std::map<unsigned, std::vector<unsigned>> bow;
    const std::vector<cv::KeyPoint>& featuresUndist() const { return kps; }
    const cv::Mat& descriptorsUndist() const { return desc; }
    const std::map<unsigned, std::vector<unsigned>>& bowFeatures() const { return bow; }
};
struct MockKeyFrame {
    std::shared_ptr<MockFrame> f;
    std::vector<std::shared_ptr<MockMapPoint>> mps;
    const std::shared_ptr<MockFrame>& frame() const { return f; }
    std::vector<std::shared_ptr<MockMapPoint>> obsMapPoints() const { return mps; }
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
    }
    Initializer init(1.0, 50);
    std::vector<cv::Point2f> p, r;
    for (auto& mm : m) { p.push_back(k1[mm.queryIdx].pt); r.push_back(k2[mm.trainIdx].pt); }
    if (p.size() >= 8) {
        init.setPoints(p); init.setRefPoints(r);
        init.findFundamentalAndHomography();
        std::printf("H empty=%d score_h=%.3f\n", (int)init.getHomographyMat().empty(), init.hScore());
    }
    return 0;
}
