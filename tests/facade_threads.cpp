// Host-only ThreadSanitizer check of the drop-in facade classes: the reference keeps ONE ORBMatcher and ONE
// ORBExtractor in static members and calls them from the tracker thread and the local-mapper thread at once
// (include/orb_slam/frame.h:520-522, src/tracker.cpp:310, src/local_mapping/local_mapper.cpp:277,462,480), while a
// C-ABI handle is not re-entrant.  The C ABI is replaced here by stand-ins whose handles carry plain (non-atomic)
// state that every call reads and writes, so two threads inside calls on ONE handle are a data race that the
// sanitizer reports.  Built with -fsanitize=thread; no GPU, no liborbx.
//   facade_threads           two threads through one shared ORBMatcher + ORBExtractor: must be clean
//   facade_threads shared    the same calls on one raw handle from two threads: the sanitizer must object
#include <cstdio>
#include <cstring>
#include <thread>

#include <orb_slam/geometry/orb_extractor.h>
#include <orb_slam/geometry/orb_matcher.h>

struct orbx_matcher { long calls = 0; unsigned char buf[256] = {}; };
struct orbx_extractor { long calls = 0; unsigned char buf[256] = {}; };
static std::mutex g_count_mutex;
static int g_matchers = 0, g_extractors = 0;

template <typename H> static void touch(H* h, int salt) {
    h->calls++;
    for (int i = 0; i < 256; i++) h->buf[i] = (unsigned char)(h->buf[i] + salt + i);
}

extern "C" {
const char* orbx_status_string(orbx_status) { return "stub"; }
const char* orbx_last_error(void) { return ""; }
orbx_status orbx_matcher_create(int, orbx_matcher** out) {
    std::lock_guard<std::mutex> lock(g_count_mutex);
    g_matchers++;
    *out = new orbx_matcher();
    return ORBX_OK;
}
void orbx_matcher_destroy(orbx_matcher* h) { delete h; }
orbx_status orbx_match_hamming(orbx_matcher* h, const uint8_t*, int n_query, const uint8_t*, int, int32_t* idx,
                               int32_t* dist) {
    touch(h, 1);
    for (int i = 0; i < n_query; i++) { idx[i] = i; dist[i] = i % 40; }
    return ORBX_OK;
}
orbx_status orbx_match_radius(orbx_matcher* h, const orbx_keypoint*, int n, const orbx_keypoint*, int, const uint8_t*,
                              const uint8_t*, double, orbx_match* matches, int* count) {
    touch(h, 2);
    for (int i = 0; i < n; i++) matches[i] = orbx_match{i, i, 0, 1.f};
    *count = n;
    return ORBX_OK;
}
orbx_status orbx_filter_matches(orbx_match*, int*, int) { return ORBX_OK; }
orbx_status orbx_extractor_create(const orbx_extractor_params*, int, int, int, int, orbx_extractor** out) {
    std::lock_guard<std::mutex> lock(g_count_mutex);
    g_extractors++;
    *out = new orbx_extractor();
    return ORBX_OK;
}
void orbx_extractor_destroy(orbx_extractor* h) { delete h; }
orbx_status orbx_extractor_set_distribution(orbx_extractor*, int, int, int) { return ORBX_OK; }
orbx_status orbx_extractor_scale_tables(const orbx_extractor*, float* a, float* b, float* c, float* d, float* e) {
    a[0] = b[0] = c[0] = d[0] = 1.f;
    *e = 0.f;
    return ORBX_OK;
}
orbx_status orbx_detect(orbx_extractor* h, const uint8_t*, int, int, int, int, orbx_keypoint* kp, int cap, int* count) {
    touch(h, 3);
    *count = 10;
    for (int i = 0; i < 10 && i < cap; i++) kp[i] = orbx_keypoint{(float)i, (float)i, 31.f, 0.f, 1.f, 0, -1};
    return ORBX_OK;
}
orbx_status orbx_compute(orbx_extractor* h, const uint8_t*, int, int, int, int, orbx_keypoint*, int* count, uint8_t* d) {
    touch(h, 4);
    std::memset(d, 7, (size_t)*count * 32);
    return ORBX_OK;
}
}

using namespace orb_slam::geometry;

int main(int argc, char** argv) {
    const bool shared_raw = argc > 1 && std::string(argv[1]) == "shared";
    ros::NodeHandle nh;
    nh.set("orb_slam/orb_extractor/level_pyramid", "1");
    // the reference's pattern: shared_ptr<const T> singletons handed to every frame
    ORBMatcherConstPtr matcher(new ORBMatcher(nh));
    ORBExtractorConstPtr extractor(new ORBExtractor(nh, 64, 64));
    cv::Mat img(64, 64, CV_8U), d(10, 32, CV_8U);
    std::vector<cv::KeyPoint> kps(10);
    orbx_matcher* raw = nullptr;
    orbx_matcher_create(0, &raw);
    auto work = [&](int t) {
        for (int rep = 0; rep < 200; rep++) {
            if (shared_raw) {
                int32_t idx[10], dist[10];
                orbx_match_hamming(raw, d.data, 10, d.data, 10, idx, dist);
                continue;
            }
            std::vector<cv::DMatch> m;
            if (t == 0) matcher->match(d, d, m, true);            // tracker thread: Frame::match
            else matcher->match(kps, kps, d, d, m, false);        // a second caller on the same object
            std::vector<cv::KeyPoint> k;
            cv::Mat desc;
            extractor->detect(img, k);
            extractor->compute(img, k, desc);
        }
    };
    std::thread a(work, 0), b(work, 1);
    a.join();
    b.join();
    // threads that start after those two have ended take over their handles: no new handle is made, however many
    // short-lived threads pass through the shared objects
    for (int round = 0; round < 3 && !shared_raw; round++) {
        std::thread c(work, 0), e(work, 1);
        c.join();
        e.join();
    }
    orbx_matcher_destroy(raw);
    // one handle for the constructing thread and one per concurrently running worker thread (plus the raw one)
    std::printf("matchers=%d extractors=%d\n", g_matchers - 1, g_extractors);
    return (shared_raw || (g_matchers - 1 == 3 && g_extractors == 3)) ? 0 : 1;
}
