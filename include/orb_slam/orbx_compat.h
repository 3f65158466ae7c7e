/*
 * Compatibility layer of the drop-in headers.  With OpenCV and ROS present (the reference's own build) the real
 * cv:: and ros:: types are used.  Without them (this repository's test builds) minimal stand-ins with the same
 * member names and memory layout are defined, so the facade classes compile and run unchanged.
 */
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "orbx.h"

#if defined(ORBX_WITH_OPENCV) || (defined(__has_include) && __has_include(<opencv2/core/core.hpp>) && !defined(ORBX_NO_OPENCV))
#include <opencv2/core/core.hpp>
#include <opencv2/features2d.hpp>
#define ORBX_HAVE_OPENCV 1
#else
#define ORBX_HAVE_OPENCV 0
#define CV_8U 0
#define CV_32F 5
#define CV_64F 6
namespace cv {
template <typename T> struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T x_, T y_) : x(x_), y(y_) {}
};
typedef Point_<float> Point2f;
typedef Point_<double> Point2d;
template <typename T> struct Point3_ {
    T x, y, z;
    Point3_() : x(0), y(0), z(0) {}
    Point3_(T x_, T y_, T z_) : x(x_), y(y_), z(z_) {}
};
typedef Point3_<double> Point3d;
struct Vec3f {
    float val[3];
    Vec3f() : val{0.f, 0.f, 0.f} {}
    Vec3f(float a, float b, float c) : val{a, b, c} {}
    float& operator[](int i) { return val[i]; }
    const float& operator[](int i) const { return val[i]; }
};
struct KeyPoint {
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
    KeyPoint() : size(0), angle(-1), response(0), octave(0), class_id(-1) {}
};
struct DMatch {
    int queryIdx, trainIdx, imgIdx;
    float distance;
    DMatch() : queryIdx(-1), trainIdx(-1), imgIdx(-1), distance(3.402823466e+38f) {}
    DMatch(int q, int t, float d) : queryIdx(q), trainIdx(t), imgIdx(-1), distance(d) {}
};
/* Dense 2-D matrix with shared storage: the subset of cv::Mat the facade touches. */
class Mat {
public:
    int rows, cols;
    Mat() : rows(0), cols(0), type_(0), esz_(1) {}
    Mat(int r, int c, int type) { create(r, c, type); }
    void create(int r, int c, int type) {
        rows = r; cols = c; type_ = type & 7; ch_ = ((type >> 3) & 511) + 1;
        esz_ = (type_ == CV_64F ? 8 : type_ == CV_32F ? 4 : 1) * ch_;
        buf_ = std::make_shared<std::vector<unsigned char>>((size_t)r * c * esz_);
        data = buf_->data();
        step = (size_t)c * esz_;
    }
    bool empty() const { return rows == 0 || cols == 0; }
    int type() const { return type_ | ((ch_ - 1) << 3); }
    int channels() const { return ch_; }
    template <typename T> T* ptr(int r = 0) { return (T*)(data + step * r); }
    template <typename T> const T* ptr(int r = 0) const { return (const T*)(data + step * r); }
    template <typename T> T& at(int r, int c) { return ptr<T>(r)[c]; }
    template <typename T> const T& at(int r, int c) const { return ptr<T>(r)[c]; }
    Mat clone() const {
        Mat m(rows, cols, type());
        for (int r = 0; r < rows && cols > 0; r++) std::memcpy(m.data + m.step * r, data + step * r, (size_t)cols * esz_);
        return m;
    }
    /* header over row r, sharing this matrix' storage (cv::Mat::row) */
    Mat row(int r) const {
        Mat m;
        m.rows = 1; m.cols = cols; m.type_ = type_; m.ch_ = ch_; m.esz_ = esz_;
        m.buf_ = buf_; m.data = data + step * r; m.step = step;
        return m;
    }
    void release() { rows = cols = 0; buf_.reset(); data = nullptr; }
    unsigned char* data = nullptr;
    size_t step = 0;
private:
    int type_, ch_ = 1, esz_;
    std::shared_ptr<std::vector<unsigned char>> buf_;
};
}  // namespace cv
#define CV_8UC1 0
#define CV_8UC3 16
#endif

#if defined(ORBX_WITH_ROS) || (defined(__has_include) && __has_include(<ros/ros.h>) && !defined(ORBX_NO_ROS))
#include <ros/ros.h>
#define ORBX_HAVE_ROS 1
#else
#define ORBX_HAVE_ROS 0
namespace ros {
/* Parameter source standing in for ros::NodeHandle: a flat key -> scalar map, filled from the reference's own
   YAML files (cfg/orb_extractor.yaml, cfg/orb_matcher.yaml, cfg/tracker.yaml) with loadYaml(). Keys are the
   slash-joined paths the reference reads, e.g. "orb_slam/orb_extractor/n_key_points". */
class NodeHandle {
public:
    NodeHandle() : values_(std::make_shared<std::map<std::string, std::string>>()) {}
    bool loadYaml(const std::string& path) {
        std::ifstream f(path);
        if (!f) return false;
        std::vector<std::pair<int, std::string>> stack;
        std::string line;
        while (std::getline(f, line)) {
            size_t hash = line.find('#');
            if (hash != std::string::npos) line.erase(hash);
            size_t first = line.find_first_not_of(" \t\r");
            if (first == std::string::npos) continue;
            size_t colon = line.find(':', first);
            if (colon == std::string::npos) continue;
            std::string key = line.substr(first, colon - first);
            std::string val = line.substr(colon + 1);
            size_t a = val.find_first_not_of(" \t\r\""), b = val.find_last_not_of(" \t\r\"");
            val = a == std::string::npos ? "" : val.substr(a, b - a + 1);
            while (!stack.empty() && stack.back().first >= (int)first) stack.pop_back();
            if (val.empty() || val == "[") { stack.push_back({(int)first, key}); continue; }
            std::string full;
            for (auto& s : stack) full += s.second + "/";
            (*values_)[full + key] = val;
        }
        return true;
    }
    void set(const std::string& key, const std::string& value) { (*values_)[strip(key)] = value; }
    template <typename T> bool getParam(const std::string& key, T& out) const {
        auto it = values_->find(strip(key));
        if (it == values_->end()) return false;
        std::istringstream ss(it->second);
        ss >> out;
        return !ss.fail();
    }
    bool getParam(const std::string& key, std::string& out) const {
        auto it = values_->find(strip(key));
        if (it == values_->end()) return false;
        out = it->second;
        return true;
    }
    template <typename T> bool param(const std::string& key, T& out, const T& def) const {
        if (getParam(key, out)) return true;
        out = def;
        return false;
    }
private:
    static std::string strip(const std::string& k) { return !k.empty() && k[0] == '/' ? k.substr(1) : k; }
    std::shared_ptr<std::map<std::string, std::string>> values_;
};
}  // namespace ros
#endif

namespace orbx_compat {
static_assert(sizeof(cv::KeyPoint) == sizeof(orbx_keypoint), "cv::KeyPoint layout");
static_assert(sizeof(cv::DMatch) == sizeof(orbx_match), "cv::DMatch layout");
inline void check(orbx_status s, const char* where) {
    if (s != ORBX_OK)
        throw std::runtime_error(std::string(where) + ": " + orbx_status_string(s) + " " + orbx_last_error());
}

/*
 * One C-ABI handle per calling thread behind a facade object.  The reference keeps ONE ORBExtractor and ONE
 * ORBMatcher in static members (include/orb_slam/frame.h:520-522, map_point.h:240-241) and calls them from the
 * tracker thread and the local-mapper thread at the same time (src/tracker.cpp:310, src/local_mapping/
 * local_mapper.cpp:277,462,480); an orbx handle owns one stream and one set of device buffers and serves one call
 * at a time.  get() hands every thread its own handle, so concurrent calls through one facade object never share
 * device state and never wait for each other.  A handle is lent to a thread for the thread's lifetime: when the
 * thread ends, its handles go back to the idle list of the object they belong to and the next new thread takes one
 * from there, so a process that calls the facade from short-lived threads (pools, per-request workers) holds as many
 * handles as it ever had threads inside the facade at once, not one per thread ever started.  The state lives
 * behind a shared_ptr: copies of a facade object share the pool, as copies of the reference's classes share their
 * cv::Ptr; the handles are destroyed with the last copy.
 */
template <typename H>
class PerThreadHandle {
public:
    PerThreadHandle() {}
    PerThreadHandle(std::function<H*()> make, void (*destroy)(H*)) : state_(std::make_shared<State>()) {
        state_->make = std::move(make);
        state_->destroy = destroy;
        get();  // the constructing thread's handle: a missing GPU shows at construction, as a throw
    }
    H* get() const {
        const std::thread::id id = std::this_thread::get_id();
        H* h = nullptr;
        {
            std::lock_guard<std::mutex> lock(state_->mutex);
            auto it = state_->lent.find(id);
            if (it != state_->lent.end()) return it->second;
            if (!state_->idle.empty()) {
                h = state_->idle.back();
                state_->idle.pop_back();
            } else {
                h = state_->make();
            }
            state_->lent.emplace(id, h);
        }
        leases().states.push_back(state_);
        return h;
    }
    /* handles in existence (lent + idle) */
    size_t size() const {
        std::lock_guard<std::mutex> lock(state_->mutex);
        return state_->lent.size() + state_->idle.size();
    }
private:
    struct State {
        std::mutex mutex;
        std::function<H*()> make;
        void (*destroy)(H*) = nullptr;
        std::map<std::thread::id, H*> lent;
        std::vector<H*> idle;
        ~State() {
            for (auto& kv : lent) destroy(kv.second);
            for (H* h : idle) destroy(h);
        }
    };
    /* what the calling thread has borrowed; returned when the thread ends */
    struct Leases {
        std::vector<std::weak_ptr<State>> states;
        ~Leases() {
            const std::thread::id id = std::this_thread::get_id();
            for (auto& w : states)
                if (std::shared_ptr<State> s = w.lock()) {
                    std::lock_guard<std::mutex> lock(s->mutex);
                    auto it = s->lent.find(id);
                    if (it != s->lent.end()) {
                        s->idle.push_back(it->second);
                        s->lent.erase(it);
                    }
                }
        }
    };
    static Leases& leases() {
        static thread_local Leases l;
        return l;
    }
    std::shared_ptr<State> state_;
};
}  // namespace orbx_compat
