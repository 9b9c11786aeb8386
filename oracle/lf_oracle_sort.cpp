/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see lf_oracle.h).
 * The seed order of OpenCV's later 3.x LSD, for QUANTIFYING a documented lead, not the default:
 * from the version that keeps `std::vector<normPoint> ordered_points` (imgproc/src/lsd.cpp, ll_angle: one normPoint
 * {Point2i p; int norm} per pixel pushed in raster order, norm = int(modgrad * bin_coef)) the list is ordered with
 *     std::sort(ordered_points.begin(), ordered_points.end(), compare_norm);      // n1.norm > n2.norm
 * which is not stable: inside a bin the order is whatever libstdc++'s introsort leaves.  This file runs that very
 * call with the C++ library of this image (g++ 11; the algorithm -- median-of-three introsort, threshold 16, final
 * insertion sort -- is unchanged since GCC 4) so that the oracle can be switched to it.  The OpenCV 3.0 form builds
 * per-bin linked lists in raster order instead, which is what the oracle (and the GPU) follow by default.
 */
#include <algorithm>
#include <cstdint>
#include <vector>

namespace {
struct Point2i { int x, y; };
struct normPoint { Point2i p; int norm; };
inline bool compare_norm(const normPoint& n1, const normPoint& n2) { return n1.norm > n2.norm; }
}

extern "C" void lfo_std_sort_seed_order(const int32_t* bin_of_pixel /* raster over (H-1) x (W-1) */, int H, int W, int32_t* order)
{
    std::vector<normPoint> pts;
    pts.reserve((size_t)(H - 1) * (W - 1));
    size_t k = 0;
    for (int y = 0; y < H - 1; ++y)
        for (int x = 0; x < W - 1; ++x) {
            normPoint q;
            q.p.x = x; q.p.y = y; q.norm = bin_of_pixel[k++];
            pts.push_back(q);
        }
    std::sort(pts.begin(), pts.end(), compare_norm);
    for (size_t i = 0; i < pts.size(); ++i) order[i] = pts[i].p.y * W + pts[i].p.x;
}

/* ---- test helpers for the device's sort emulation (lane_slam_amd/csrc/k_lsd_seed32.hip, lf_debug_std_sort) ------------- */

/* std::sort(compare_norm) of n keys in their given order: order[i] = index of the element left at place i; returns the number
 * of comparisons made */
extern "C" long long lfo_std_sort_keys(const int32_t* keys, int n, int32_t* order)
{
    std::vector<normPoint> pts((size_t)n);
    for (int i = 0; i < n; ++i) { pts[i].p.x = i; pts[i].p.y = 0; pts[i].norm = keys[i]; }
    long long ncmp = 0;
    std::sort(pts.begin(), pts.end(), [&ncmp](const normPoint& a, const normPoint& b) { ++ncmp; return a.norm > b.norm; });
    for (int i = 0; i < n; ++i) order[i] = pts[i].p.x;
    return ncmp;
}

/* M. D. McIlroy, "A Killer Adversary for Quicksort" (1999), played against THIS library's std::sort with the descending
 * comparator: the comparator decides the values while the sort runs ("gas" items are frozen to the next solid value when two of
 * them meet), which yields n distinct keys on which the same sort runs into its depth limit and finishes with heap sort.
 * keys[i] in [0, n): n <= 1024 for the bin range of the detector. */
extern "C" void lfo_antiqsort_keys(int n, int32_t* keys)
{
    std::vector<int> val((size_t)n, n - 1), ptr((size_t)n);
    const int gas = n - 1;
    int nsolid = 0, candidate = 0;
    for (int i = 0; i < n; ++i) ptr[i] = i;
    /* solid values count DOWN from the top for a descending sort: the adversary hands out the values the sort wants to see
     * first, so every pivot ends up at one end of its range */
    auto greater = [&](int x, int y) {
        if (val[x] == gas && val[y] == gas) { if (x == candidate) val[x] = nsolid++; else val[y] = nsolid++; }
        if (val[x] == gas) candidate = x; else if (val[y] == gas) candidate = y;
        return val[x] < val[y];            /* "x sorts before y": frozen (small) before gas, in the order of freezing */
    };
    std::sort(ptr.begin(), ptr.end(), greater);
    /* the order the adversary fixed is ascending in val; as keys for compare_norm (a.norm > b.norm) that is descending in key */
    for (int i = 0; i < n; ++i) keys[i] = (n - 1) - val[i];
}
