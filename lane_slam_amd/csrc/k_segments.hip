// K_segments: per-segment stages a-5 .. a-8 fused, one lane per detected line.
//
// Reference (paths relative to /root/reference):
//   a-5 src/line_detector/include/line_detector/line_detector_lsd.py:74-125  _findNormal, _checkBounds,
//       _correctPixelOrdering (float32 numpy arithmetic, int truncation, float64 ordering test)
//   a-6 src/line_detector/src/line_detector_node.py:195-205,251-265  normalisation, colour order, float32 msg fields
//   a-7 src/ground_projection/include/ground_projection/GroundProjection.py:38-48,64-78  vector2pixel (+ the
//       v > ch-1 -> 0 quirk), rectifyPoint (cv2.undistortPoints, 5 iterations), homography
//   a-8 src/line_sanity/src/line_sanity_node.py:48-117  processSegmentList + fancyFilters
//
// Also compacts the fixed-capacity LSD slots [frame][colour][cap] into the frame-major,
// colour-minor SegmentList order.  Byte traffic is negligible (~130 B per segment).
#include "common.h"

namespace lf {

// overflow[0]: some problem has more lines than cap_lines.  overflow[1], [2]: problems whose defined pixels exceed the region
// growing kernel's LDS slice at its small / medium size (the host sizes the next batch's slices from them, k_lsd_grow.hip);
// overflow[3]: the largest problem's defined pixels; overflow[7]: the batch's segment total (so that ONE copy of the eight words
// takes everything lf_wait needs to the host)
__global__ void k_seg_offsets(int n_frames, int cap_lines, const int* __restrict__ counts, int* __restrict__ seg_offset,
                              int* __restrict__ frame_offset, int* __restrict__ overflow, const int* __restrict__ norder, int cap_small, int cap_medium)
{
    // single workgroup exclusive scan over n_frames*3 clipped counts
    __shared__ int carry;
    __shared__ int wsum[16];
    const int n = n_frames * 3;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) carry = 0;
    __syncthreads();
    int ovf = 0;
    for (int base = 0; base < n; base += blockDim.x) {
        int i = base + t;
        int v = 0;
        if (i < n) {
            v = counts[i]; if (v > cap_lines) { v = cap_lines; ovf = 1; }
            if (norder) { const int nd = norder[i]; if (nd > cap_small) atomicAdd(overflow + 1, 1); if (nd > cap_medium) atomicAdd(overflow + 2, 1); atomicMax(overflow + 3, nd); }
        }
        int inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { int o = __shfl_up(inc, d); if (lane >= d) inc += o; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int off = carry;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (i < n) {
            int ex = off + inc - v;
            seg_offset[i] = ex;
            if (i % 3 == 0) frame_offset[i / 3] = ex;
        }
        __syncthreads();
        if (t == blockDim.x - 1) carry = off + inc;
        __syncthreads();
    }
    if (t == 0) { seg_offset[n] = carry; frame_offset[n_frames] = carry; overflow[7] = carry; }
    if (ovf) atomicOr(overflow, 1);
}

void launch_seg_offsets(int n_frames, int cap_lines, const int* counts, int* seg_offset, int* frame_offset,
                        int* overflow, const int* norder, int cap_small, int cap_medium, hipStream_t s)
{
    // four waves: a 1024-thread workgroup waits for a CU with sixteen free wave slots in a busy pipeline
    hipLaunchKernelGGL(k_seg_offsets, dim3(1), dim3(256), 0, s, n_frames, cap_lines, counts, seg_offset,
                       frame_offset, overflow, norder, cap_small, cap_medium);
}

__device__ __forceinline__ int check_bounds(int v, int bound) { return v < 0 ? 0 : (v >= bound ? bound - 1 : v); }

__device__ void ground_point(const SegParams& p, double vx, double vy, double& gx, double& gy)
{
    double u = p.cw * vx, v = p.ch * vy;
    if (u < 0) u = 0;
    if (u > p.cw - 1) u = p.cw - 1;
    if (v < 0) v = 0;
    if (v > p.ch - 1) v = 0;
    const double fx = p.K[0], fy = p.K[4], cx = p.K[2], cy = p.K[5];
    const double ifx = 1. / fx, ify = 1. / fy;
    const double* k = p.D;
    double x = (u - cx) * ifx, y = (v - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; ++j) {
        double r2 = x * x + y * y;
        double icdist = 1 / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
        double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    const double* RR = p.RR;
    double xx = RR[0] * x + RR[1] * y + RR[2];
    double yy = RR[3] * x + RR[4] * y + RR[5];
    double ww = 1. / (RR[6] * x + RR[7] * y + RR[8]);
    double ur = xx * ww, vr = yy * ww;
    const double* H = p.H;
    double g0 = H[0] * ur + H[1] * vr + H[2] * 1.0;
    double g1 = H[3] * ur + H[4] * vr + H[5] * 1.0;
    double g2 = H[6] * ur + H[7] * vr + H[8] * 1.0;
    gx = g0 / g2;
    gy = g1 / g2;
}

__global__ void k_segments(SegParams p, int n_frames, const float* __restrict__ slot_lines,
                           const int* __restrict__ counts, const int* __restrict__ seg_offset,
                           const uint32_t* __restrict__ maskbits, int Ww, lf_segments out, int* __restrict__ seg_frame,
                           double* __restrict__ normals64, float* __restrict__ centers)
{
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;      // [pc][cap]
    const int pc = slot / p.cap_lines, i = slot - pc * p.cap_lines;
    if (pc >= n_frames * 3) return;
    int cnt = counts[pc];
    if (cnt > p.cap_lines) cnt = p.cap_lines;
    if (i >= cnt) return;
    const int idx = seg_offset[pc] + i;
    if (idx >= out.capacity) return;
    const int f = pc / 3, col = pc - 3 * f;
    const float* L = slot_lines + ((size_t)pc * p.cap_lines + i) * 4;
    float x1 = L[0], y1 = L[1], x2 = L[2], y2 = L[3];
    // a-5
    const float ex = x1 - x2, ey = y1 - y2;
    const float len = dm::fsqrt(ex * ex + ey * ey);
    const float dx = dm::fdiv(y2 - y1, len);
    const float dy = dm::fdiv(x1 - x2, len);
    const float cx = (x1 + x2) / 2, cy = (y1 + y2) / 2;
    int x3 = (int)(cx - 3.f * dx), y3 = (int)(cy - 3.f * dy);
    int x4 = (int)(cx + 3.f * dx), y4 = (int)(cy + 3.f * dy);
    x3 = check_bounds(x3, p.W); y3 = check_bounds(y3, p.Hc);
    x4 = check_bounds(x4, p.W); y4 = check_bounds(y4, p.Hc);
    // bw = the dilated colour mask, kept as a bit plane (k_pre)
    const uint32_t* bw = maskbits + (size_t)pc * p.Hc * Ww;
    const bool on3 = (bw[(size_t)y3 * Ww + (x3 >> 5)] >> (x3 & 31)) & 1u;
    const bool on4 = (bw[(size_t)y4 * Ww + (x4 >> 5)] >> (x4 & 31)) & 1u;
    const int sign = (on3 && !on4) ? 1 : -1;
    const double nx = (double)dx * sign, ny = (double)dy * sign;
    const double flag = (double)(x2 - x1) * ny - (double)(y2 - y1) * nx;
    if (flag > 0) { float tx = x1, ty = y1; x1 = x2; y1 = y2; x2 = tx; y2 = ty; }
    if (out.lines) { float* o = out.lines + 4 * (size_t)idx; o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2; }
    if (out.normals) { out.normals[2 * (size_t)idx] = (float)nx; out.normals[2 * (size_t)idx + 1] = (float)ny; }
    if (out.color) out.color[idx] = (uint8_t)col;
    if (normals64) { normals64[2 * (size_t)idx] = nx; normals64[2 * (size_t)idx + 1] = ny; }
    if (centers) { centers[2 * (size_t)idx] = cx; centers[2 * (size_t)idx + 1] = cy; }
    if (seg_frame) seg_frame[idx] = f;
    // a-6
    const float pn0 = (float)(((double)x1 + 0.0) * p.rx);
    const float pn1 = (float)(((double)y1 + p.cut) * p.ry);
    const float pn2 = (float)(((double)x2 + 0.0) * p.rx);
    const float pn3 = (float)(((double)y2 + p.cut) * p.ry);
    if (out.pixels_normalized) {
        float* o = out.pixels_normalized + 4 * (size_t)idx;
        o[0] = pn0; o[1] = pn1; o[2] = pn2; o[3] = pn3;
    }
    // a-7
    double p1x, p1y, p2x, p2y;
    ground_point(p, (double)pn0, (double)pn1, p1x, p1y);
    ground_point(p, (double)pn2, (double)pn3, p2x, p2y);
    if (out.ground) { double* o = out.ground + 4 * (size_t)idx; o[0] = p1x; o[1] = p1y; o[2] = p2x; o[3] = p2y; }
    // a-8
    if (out.keep) {
        int state = 0;
        const double gx_ = p2x - p1x, gy_ = p2y - p1y;
        const double nrm = dm::dsqrt(gx_ * gx_ + gy_ * gy_);
        const double tx = gx_ / nrm, ty = gy_ / nrm;
        const double hx = -ty, hy = tx;
        const double d1 = hx * p1x + hy * p1y;
        const double d2 = hx * p2x + hy * p2y;
        double d_i = (d1 + d2) / 2;
        double phi_i = dm::dasin(ty);
        if (col == LF_WHITE) {
            if (p1x > p2x) { d_i = d_i - p.linewidth_white; state = 1; }
            else { d_i = -d_i; phi_i = -phi_i; state = 2; }
            d_i = d_i - p.lanewidth / 2;
        } else if (col == LF_YELLOW) {
            if (p2x > p1x) { d_i = d_i - p.linewidth_yellow; phi_i = -phi_i; state = 3; }
            else { d_i = -d_i; state = 4; }
            d_i = p.lanewidth / 2 - d_i;
        }
        int k = 1;
        if (p1x < 0 || p2x < 0) k = 0;
        else if (col != LF_WHITE && col != LF_YELLOW) k = 0;
        else if (state == 0) k = 0;
        else if (d_i > p.d_max || d_i < p.d_min || phi_i < p.phi_min || phi_i > p.phi_max) k = 0;
        out.keep[idx] = (uint8_t)k;
    }
}

void launch_segments(const SegParams& p, int n_frames, const float* slot_lines, const int* counts,
                     const int* seg_offset, const uint32_t* maskbits, int Ww, lf_segments out, int* seg_frame,
                     double* normals64, float* centers, hipStream_t s)
{
    const int total = n_frames * 3 * p.cap_lines;
    hipLaunchKernelGGL(k_segments, dim3((total + 255) / 256), dim3(256), 0, s, p, n_frames, slot_lines, counts,
                       seg_offset, maskbits, Ww, out, seg_frame, normals64, centers);
}

}  // namespace lf
