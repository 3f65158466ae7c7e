// A translation unit laid out the way the reference lays out src/geometry/utils.cpp and its callers: the header is
// included first, then every function it declares is DEFINED (non-inline) in the same namespaces, and a caller uses
// utils::matToVectorMat the way Frame::computeBow does (reference src/frame.cpp:316).  It must compile against the
// facade header: a header that defined any of these functions inline would make this file a redefinition error.
// With DEFINE_LIKE_THE_REFERENCE unset only the caller is built and the definitions come from elsewhere
// (this repository's src/geometry/utils.cpp).  The bodies here are stand-ins: the test is about declarations.
#include <orb_slam/geometry/utils.h>

#include <cstdio>

#ifdef DEFINE_LIKE_THE_REFERENCE
namespace orb_slam {

namespace geometry {

void normalizePoints(
    const std::vector<cv::Point2f>& points,
    std::vector<cv::Point2f>& normalized,
    cv::Mat& T)
{
    normalized = points;
    T = cv::Mat(3, 3, CV_32F);
}

void computeFundamentalMat(
    cv::Mat& f_mat,
    const std::vector<cv::Point2f>& points,
    const std::vector<cv::Point2f>& ref_points)
{
    (void)points; (void)ref_points;
    f_mat = cv::Mat(3, 3, CV_32F);
}

void computeHomographyMat(
    cv::Mat& h_mat,
    const std::vector<cv::Point2f> &points,
    const std::vector<cv::Point2f> &ref_points)
{
    (void)points; (void)ref_points;
    h_mat = cv::Mat(3, 3, CV_32F);
}

void drawEpilines(
    std::vector<cv::Vec3f> lines,
    cv::Mat f_mat,
    cv::Mat image,
    std::vector<cv::Point2f> pts,
    std::vector<bool> inliers)
{
    (void)lines; (void)f_mat; (void)image; (void)pts; (void)inliers;
}

int descriptorDistance(const cv::Mat &a, const cv::Mat &b)
{
    int d = 0;
    for (int i = 0; i < 32; i++) d += __builtin_popcount(a.ptr<unsigned char>()[i] ^ b.ptr<unsigned char>()[i]);
    return d;
}

void computeThreeMaxima(
    std::vector<int>* histo, const int L, int &ind1, int &ind2, int &ind3)
{
    // slow and obvious: repeated arg-max with the earlier bin winning ties, then the 10 % rule
    int idx[3] = {-1, -1, -1}, size[3] = {0, 0, 0};
    for (int rank = 0; rank < 3; rank++)
        for (int i = 0; i < L; i++) {
            if (i == idx[0] || i == idx[1]) continue;
            if ((int)histo[i].size() > size[rank]) { size[rank] = (int)histo[i].size(); idx[rank] = i; }
        }
    ind1 = idx[0]; ind2 = idx[1]; ind3 = idx[2];
    if (size[1] < 0.1f * (float)size[0]) { ind2 = -1; ind3 = -1; }
    else if (size[2] < 0.1f * (float)size[0]) { ind3 = -1; }
}

} // namespace geometry

namespace utils {
    void matToVectorMat(const cv::Mat& mat, std::vector<cv::Mat>& vec_mat) {
        vec_mat.resize(mat.rows);
        for (int i = 0; i < mat.rows; ++i) vec_mat[i] = mat.row(i);
    }

    cv::Mat skew(const cv::Mat& vec)
    {
        (void)vec;
        return cv::Mat(3, 3, CV_32F);
    }
} // namespace utils

} // namespace orb_slam
#endif

namespace orb_slam {
// the slice of Frame that computeBow touches
class FrameLike {
public:
    FrameLike() : undist_descriptors_(3, 32, CV_8U) {}
    size_t computeBow() {
        std::vector<cv::Mat> vec_descriptors;
        utils::matToVectorMat(undist_descriptors_, vec_descriptors);
        return vec_descriptors.size();
    }
private:
    cv::Mat undist_descriptors_;
};
} // namespace orb_slam

int main()
{
    orb_slam::FrameLike frame;
    std::vector<int> hist[30];
    const int fill[8] = {10, 1, 20, 15, 9, 8, 7, 5};
    for (int b = 0; b < 8; b++) hist[b].assign(fill[b], 1);
    int m1 = -1, m2 = -1, m3 = -1;
    orb_slam::geometry::computeThreeMaxima(hist, 30, m1, m2, m3);
    std::printf("rows=%zu maxima=%d,%d,%d\n", frame.computeBow(), m1, m2, m3);
    return 0;
}
