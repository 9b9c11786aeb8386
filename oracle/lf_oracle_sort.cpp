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
