// Element-for-element re-statement of the libstdc++ algorithms behind cv::KeyPointsFilter::retainBest:
// std::nth_element (introselect: median-of-three unguarded partition, heap-select fallback, final insertion
// sort) followed by std::partition (bidirectional form).  The ORDER these leave the kept keypoints in is part of
// the reference's observable result (it becomes the keypoint order, hence every match index), so the device path
// replays exactly the same sequence of comparisons and moves; compiled for the host as well, it is checked against
// the real std:: calls in the CPU tests.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define ORBX_HD __host__ __device__
#else
#define ORBX_HD
#endif

namespace orbx {
namespace stdsel {

struct Elem {
    float key;     // response
    int32_t idx;   // payload: candidate index
};

// comparator of retainBest: KeypointResponseGreater
ORBX_HD inline bool greater(const Elem& a, const Elem& b) { return a.key > b.key; }

ORBX_HD inline void swap_elem(Elem* a, Elem* b) {
    Elem t = *a;
    *a = *b;
    *b = t;
}

ORBX_HD inline int lg(int n) {  // std::__lg
    int k = 0;
    while (n > 1) { n >>= 1; k++; }
    return k;
}

ORBX_HD inline void move_median_to_first(Elem* result, Elem* a, Elem* b, Elem* c) {
    if (greater(*a, *b)) {
        if (greater(*b, *c)) swap_elem(result, b);
        else if (greater(*a, *c)) swap_elem(result, c);
        else swap_elem(result, a);
    } else if (greater(*a, *c)) swap_elem(result, a);
    else if (greater(*b, *c)) swap_elem(result, c);
    else swap_elem(result, b);
}

ORBX_HD inline Elem* unguarded_partition(Elem* first, Elem* last, Elem* pivot) {
    while (true) {
        while (greater(*first, *pivot)) ++first;
        --last;
        while (greater(*pivot, *last)) --last;
        if (!(first < last)) return first;
        swap_elem(first, last);
        ++first;
    }
}

ORBX_HD inline Elem* unguarded_partition_pivot(Elem* first, Elem* last) {
    Elem* mid = first + (last - first) / 2;
    move_median_to_first(first, first + 1, mid, last - 1);
    return unguarded_partition(first + 1, last, first);
}

ORBX_HD inline void unguarded_linear_insert(Elem* last) {
    Elem val = *last;
    Elem* next = last;
    --next;
    while (greater(val, *next)) {
        *last = *next;
        last = next;
        --next;
    }
    *last = val;
}

ORBX_HD inline void insertion_sort(Elem* first, Elem* last) {
    if (first == last) return;
    for (Elem* i = first + 1; i != last; ++i) {
        if (greater(*i, *first)) {
            Elem val = *i;
            for (Elem* p = i; p != first; --p) *p = *(p - 1);  // std::move_backward(first, i, i + 1)
            *first = val;
        } else
            unguarded_linear_insert(i);
    }
}

ORBX_HD inline void push_heap(Elem* first, long hole, long top, Elem value) {
    long parent = (hole - 1) / 2;
    while (hole > top && greater(first[parent], value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}

ORBX_HD inline void adjust_heap(Elem* first, long hole, long len, Elem value) {
    const long top = hole;
    long second = hole;
    while (second < (len - 1) / 2) {
        second = 2 * (second + 1);
        if (greater(first[second], first[second - 1])) second--;
        first[hole] = first[second];
        hole = second;
    }
    if ((len & 1) == 0 && second == (len - 2) / 2) {
        second = 2 * (second + 1);
        first[hole] = first[second - 1];
        hole = second - 1;
    }
    push_heap(first, hole, top, value);
}

ORBX_HD inline void make_heap(Elem* first, Elem* last) {
    if (last - first < 2) return;
    const long len = last - first;
    long parent = (len - 2) / 2;
    while (true) {
        Elem value = first[parent];
        adjust_heap(first, parent, len, value);
        if (parent == 0) return;
        parent--;
    }
}

ORBX_HD inline void pop_heap(Elem* first, Elem* last, Elem* result) {
    Elem value = *result;
    *result = *first;
    adjust_heap(first, 0, last - first, value);
}

ORBX_HD inline void heap_select(Elem* first, Elem* middle, Elem* last) {
    make_heap(first, middle);
    for (Elem* i = middle; i < last; ++i)
        if (greater(*i, *first)) pop_heap(first, middle, i);
}

// std::nth_element(first, nth, last, KeypointResponseGreater())
ORBX_HD inline void nth_element(Elem* first, Elem* nth, Elem* last) {
    if (first == last || nth == last) return;
    int depth_limit = lg((int)(last - first)) * 2;
    while (last - first > 3) {
        if (depth_limit == 0) {
            heap_select(first, nth + 1, last);
            swap_elem(first, nth);
            return;
        }
        --depth_limit;
        Elem* cut = unguarded_partition_pivot(first, last);
        if (cut <= nth) first = cut;
        else last = cut;
    }
    insertion_sort(first, last);
}

// std::partition(first, last, [t](kp){ return kp.response >= t; }), bidirectional-iterator form
ORBX_HD inline Elem* partition_ge(Elem* first, Elem* last, float t) {
    while (true) {
        while (true) {
            if (first == last) return first;
            else if (first->key >= t) ++first;
            else break;
        }
        --last;
        while (true) {
            if (first == last) return first;
            else if (!(last->key >= t)) --last;
            else break;
        }
        swap_elem(first, last);
        ++first;
    }
}

// cv::KeyPointsFilter::retainBest(keypoints, n_points): returns the new length.
ORBX_HD inline int retain_best(Elem* a, int n, int n_points) {
    if (n_points >= 0 && n > n_points) {
        if (n_points == 0) return 0;
        nth_element(a, a + n_points - 1, a + n);
        const float ambiguous = a[n_points - 1].key;
        Elem* new_end = partition_ge(a + n_points, a + n, ambiguous);
        return (int)(new_end - a);
    }
    return n;
}

#if defined(__HIPCC__)
// ---- the same algorithms carried out by one wavefront -------------------------------------------------------------
// The partitions are where the time goes, and their result can be formed without walking the two pointers: the
// left pointer stops exactly at the original positions whose element does not pass its test ("left stoppers", in
// ascending order), the right pointer at those that do not pass the other test ("right stoppers", descending); the
// serial loop swaps the k-th left stopper with the k-th right stopper for as long as the former lies left of the
// latter.  Ranks come from ballots, the swaps are independent, and the returned cut follows from the first pair
// that is not swapped.  Pivot choice, the final insertion sort and the heap-select fallback stay serial on lane 0
// (they touch a handful of elements, or never run on real data).  a, lp and rp are LDS arrays; every lane of the
// wavefront must call with the same arguments.

// (workgroup scope: the arrays may live in global memory, where the lanes' stores must have left the wavefront
// before other lanes load them; for LDS arrays this costs the same as the wavefront-scope form)
__device__ inline void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// MODE 0: std::__unguarded_partition(lo, hi, pivot) for KeypointResponseGreater, pivot key p; returns the cut.
// MODE 1: std::partition(lo, hi, key >= p), bidirectional form; returns the partition point.
// St: type of the stopper entries (offsets from lo): uint16_t while a range has fewer than 65536 elements.
template <int MODE, typename St>
__device__ inline int wave_partition(Elem* a, int lo, int hi, float p, St* lp, St* rp, int lane) {
    int nL = 0, nR = 0;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int base = lo; base < hi; base += 64) {
        const int i = base + lane;
        const bool in = i < hi;
        const float k = in ? a[i].key : 0.f;
        const bool is_l = in && (MODE == 0 ? !(k > p) : !(k >= p));
        const bool is_r = in && (MODE == 0 ? !(p > k) : (k >= p));
        const unsigned long long bl = __ballot(is_l), br = __ballot(is_r);
        if (is_l) lp[nL + __popcll(bl & below)] = (St)(i - lo);
        if (is_r) rp[nR + __popcll(br & below)] = (St)(i - lo);
        nL += __popcll(bl);
        nR += __popcll(br);
    }
    wave_sync();
    // k-th right stopper counted from the right end: rp[nR - 1 - k]
    const int nmin = nL < nR ? nL : nR;
    int K = 0;
    for (int base = 0; base < nmin; base += 64) {
        const int k = base + lane;
        const bool sw = k < nmin && lp[k] < rp[nR - 1 - k];
        const unsigned long long b = __ballot(sw);
        K += __popcll(b);
        if (b != ~0ull) break;  // the test is monotone in k
    }
    for (int base = 0; base < K; base += 64) {
        const int k = base + lane;
        if (k < K) {
            const int i = lo + lp[k], j = lo + rp[nR - 1 - k];
            const Elem t = a[i];
            a[i] = a[j];
            a[j] = t;
        }
    }
    int cut;
    if (MODE == 0) {
        const int l_next = K < nL ? (int)lp[K] : 0x7fffffff;
        const int r_prev = K > 0 ? (int)rp[nR - K] : 0x7fffffff;
        cut = lo + (l_next < r_prev ? l_next : r_prev);
    } else {
        cut = lo + nR;
    }
    wave_sync();
    return cut;
}

template <typename St>
__device__ inline void wave_nth_element(Elem* a, int first, int nth, int last, St* lp, St* rp, int lane) {
    if (first == last || nth == last) return;
    int depth_limit = lg(last - first) * 2;
    while (last - first > 3) {
        if (depth_limit == 0) {
            if (lane == 0) {
                heap_select(a + first, a + nth + 1, a + last);
                swap_elem(a + first, a + nth);
            }
            wave_sync();
            return;
        }
        --depth_limit;
        if (lane == 0) move_median_to_first(a + first, a + first + 1, a + first + (last - first) / 2, a + last - 1);
        wave_sync();
        const float pivot = a[first].key;
        const int cut = wave_partition<0, St>(a, first + 1, last, pivot, lp, rp, lane);
        if (cut <= nth) first = cut;
        else last = cut;
    }
    if (lane == 0) insertion_sort(a + first, a + last);
    wave_sync();
}

// cv::KeyPointsFilter::retainBest by one wavefront; lp / rp hold n entries each.  Returns the new length.
template <typename St>
__device__ inline int wave_retain_best(Elem* a, int n, int n_points, St* lp, St* rp, int lane) {
    if (n_points >= 0 && n > n_points) {
        if (n_points == 0) return 0;
        wave_nth_element(a, 0, n_points - 1, n, lp, rp, lane);
        const float ambiguous = a[n_points - 1].key;
        return wave_partition<1, St>(a, n_points, n, ambiguous, lp, rp, lane);
    }
    return n;
}
#endif  // __HIPCC__

}  // namespace stdsel
}  // namespace orbx
