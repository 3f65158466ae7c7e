/*
 * Drop-in counterpart of the reference's include/orb_slam/geometry/utils.h for the functions on the hot path
 * (reference src/geometry/utils.cpp:18-236).  descriptorDistance and computeThreeMaxima are scalar host helpers;
 * the 8-point / DLT solvers run on the GPU inside Initializer (include/orb_slam/initializer.h).
 */
#pragma once

#include <numeric>
#include <vector>

#include <orb_slam/orbx_compat.h>

namespace orb_slam {

namespace geometry {

/** Hamming distance of two 256-bit descriptors (utils.cpp:184-200). */
inline int descriptorDistance(const cv::Mat& a, const cv::Mat& b)
{
    const uint32_t* pa = a.ptr<uint32_t>();
    const uint32_t* pb = b.ptr<uint32_t>();
    int dist = 0;
    for (int i = 0; i < 8; i++) dist += __builtin_popcount(pa[i] ^ pb[i]);
    return dist;
}

/** Indices of the three fullest histogram bins; the 2nd/3rd are dropped below 10 % of the 1st (utils.cpp:202-236). */
inline void computeThreeMaxima(
    std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3)
{
    int best[3] = {0, 0, 0};
    for (int i = 0; i < L; i++) {
        const int s = (int)histo[i].size();
        if (s > best[0]) {
            best[2] = best[1]; best[1] = best[0]; best[0] = s;
            ind3 = ind2; ind2 = ind1; ind1 = i;
        } else if (s > best[1]) {
            best[2] = best[1]; best[1] = s;
            ind3 = ind2; ind2 = i;
        } else if (s > best[2]) {
            best[2] = s;
            ind3 = i;
        }
    }
    if (best[1] < 0.1f * (float)best[0]) { ind2 = -1; ind3 = -1; }
    else if (best[2] < 0.1f * (float)best[0]) { ind3 = -1; }
}

} // namespace geometry

} // namespace orb_slam
