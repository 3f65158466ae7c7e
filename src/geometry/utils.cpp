/*
 * Definitions behind include/orb_slam/geometry/utils.h for a build that takes the GPU path: this file stands where
 * the reference's src/geometry/utils.cpp stands (compile one or the other, never both).
 *
 * normalizePoints, computeFundamentalMat and computeHomographyMat (reference utils.cpp:18-154) forward to the C ABI
 * (orbx_normalize_points, orbx_fundamental_npt, orbx_homography_dlt: HIP kernels that evaluate OpenCV's float Jacobi
 * SVD in its own order).  The remaining five are host helpers.  A failing device call surfaces as
 * std::runtime_error, where the reference lets cv::Exception escape.
 */
#include <orb_slam/geometry/utils.h>

#include <cstdlib>

namespace {

/* One solver handle per calling thread, made on first use and released when the thread ends.  The GPU is the one
   named by the environment variable ORBX_DEVICE (default 0). */
orbx_ransac* solverOfThisThread()
{
    struct Holder {
        orbx_ransac* h = nullptr;
        ~Holder() { if (h) orbx_ransac_destroy(h); }
    };
    thread_local Holder holder;
    if (!holder.h) {
        const char* dev = std::getenv("ORBX_DEVICE");
        orbx_compat::check(orbx_ransac_create(dev ? std::atoi(dev) : 0, &holder.h), "geometry utils");
    }
    return holder.h;
}

cv::Mat mat33(const float* v)
{
    cv::Mat m(3, 3, CV_32F);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) m.at<float>(r, c) = v[3 * r + c];
    return m;
}

/* element i of a 3-vector stored as a row or as a column (what cv::Mat::at<float>(i) reads) */
float element(const cv::Mat& vec, int i)
{
    return vec.rows == 1 ? vec.ptr<float>(0)[i] : vec.ptr<float>(i)[0];
}

static_assert(sizeof(cv::Point2f) == 2 * sizeof(float), "cv::Point2f is two packed floats");

} // namespace

namespace orb_slam {

namespace geometry {

void normalizePoints(
    const std::vector<cv::Point2f>& points,
    std::vector<cv::Point2f>& normalized,
    cv::Mat& T)
{
    normalized.resize(points.size());
    float t9[9];
    orbx_compat::check(
        orbx_normalize_points(
            solverOfThisThread(), reinterpret_cast<const float*>(points.data()), (int)points.size(),
            reinterpret_cast<float*>(normalized.data()), t9),
        "normalizePoints");
    T = mat33(t9);
}

void computeFundamentalMat(
    cv::Mat& f_mat,
    const std::vector<cv::Point2f>& points,
    const std::vector<cv::Point2f>& ref_points)
{
    if (ref_points.size() < points.size()) throw std::runtime_error("computeFundamentalMat: fewer reference points");
    float f9[9];
    orbx_compat::check(
        orbx_fundamental_npt(
            solverOfThisThread(), reinterpret_cast<const float*>(points.data()),
            reinterpret_cast<const float*>(ref_points.data()), (int)points.size(), 1, f9),
        "computeFundamentalMat");
    f_mat = mat33(f9);
}

void computeHomographyMat(
    cv::Mat& h_mat,
    const std::vector<cv::Point2f> &points,
    const std::vector<cv::Point2f> &ref_points)
{
    if (ref_points.size() < points.size()) throw std::runtime_error("computeHomographyMat: fewer reference points");
    float h9[9];
    orbx_compat::check(
        orbx_homography_dlt(
            solverOfThisThread(), reinterpret_cast<const float*>(points.data()),
            reinterpret_cast<const float*>(ref_points.data()), (int)points.size(), 1, h9),
        "computeHomographyMat");
    h_mat = mat33(h9);
}

void drawEpilines(
    std::vector<cv::Vec3f> lines,
    cv::Mat f_mat,
    cv::Mat image,
    std::vector<cv::Point2f> pts,
    std::vector<bool> inliers)
{
    (void)f_mat;
#if ORBX_HAVE_OPENCV
    const cv::Scalar green(0, 255, 0);
    for (std::size_t i = 0; i < lines.size(); ++i) {
        if (!inliers[i]) continue;
        // the line a x + b y + c = 0 drawn from the left to the right image border
        const float a = lines[i][0], b = lines[i][1], c = lines[i][2];
        const float x_left = 0.f, x_right = (float)image.cols;
        const float y_left = (-c - a * x_left) / b, y_right = (-c - a * x_right) / b;
        cv::circle(image, pts[i], 5, green, -1);
        cv::line(image, cv::Point((int)x_left, (int)y_left), cv::Point((int)x_right, (int)y_right), green, 1);
    }
    cv::imshow("image:", image);
    cv::waitKey(0);
#else
    (void)lines; (void)image; (void)pts; (void)inliers;  // no GUI without OpenCV
#endif
}

int descriptorDistance(const cv::Mat &a, const cv::Mat &b)
{
    const uint32_t* wa = a.ptr<uint32_t>();
    const uint32_t* wb = b.ptr<uint32_t>();
    int bits = 0;
    for (int w = 0; w < 8; ++w) bits += __builtin_popcount(wa[w] ^ wb[w]);
    return bits;
}

void computeThreeMaxima(
    std::vector<int>* histo, const int L, int &ind1, int &ind2, int &ind3)
{
    // a running top three by bin size; a bin only displaces a strictly smaller one, so the first of equals stays
    int top[3] = {0, 0, 0};
    for (int bin = 0; bin < L; ++bin) {
        const int size = (int)histo[bin].size();
        if (size > top[0]) {
            top[2] = top[1]; top[1] = top[0]; top[0] = size;
            ind3 = ind2; ind2 = ind1; ind1 = bin;
        } else if (size > top[1]) {
            top[2] = top[1]; top[1] = size;
            ind3 = ind2; ind2 = bin;
        } else if (size > top[2]) {
            top[2] = size;
            ind3 = bin;
        }
    }
    const float floor_size = 0.1f * (float)top[0];
    if (top[1] < floor_size) { ind2 = -1; ind3 = -1; }
    else if (top[2] < floor_size) { ind3 = -1; }
}

} // namespace geometry

namespace utils {

void matToVectorMat(const cv::Mat& mat, std::vector<cv::Mat>& vec_mat)
{
    vec_mat.clear();
    vec_mat.reserve(mat.rows);
    for (int r = 0; r < mat.rows; ++r) vec_mat.push_back(mat.row(r));
}

cv::Mat skew(const cv::Mat& vec)
{
    const float x = element(vec, 0), y = element(vec, 1), z = element(vec, 2);
    const float m[9] = {0.f, -z, y, z, 0.f, -x, -y, x, 0.f};
    return mat33(m);
}

} // namespace utils

} // namespace orb_slam
