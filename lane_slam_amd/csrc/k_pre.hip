// K_pre: fused a-1 + a-2(HSV) + a-3(inRange, dilate) -- one HBM pass over the camera frames.
//
// Reference call sites (paths relative to /root/reference):
//   src/line_detector/src/line_detector_node.py:163-175   resize-nearest, crop, AntiInstagram, convertScaleAbs
//   src/anti_instagram/include/anti_instagram/scale_and_shift.py:25-33
//   src/line_detector/include/line_detector/line_detector_lsd.py:38-53,138   BGR2HSV, inRange (red = OR), dilate
//
// Layout: each workgroup owns a 128x8 tile of the working image.  Phase 1 converts the
// tile plus its dilation halo to packed (b,g,r,maskbits) words in LDS (HSV is computed
// once per pixel, never written to HBM).  Phase 2: each lane owns 4 adjacent pixels,
// ORs the structuring element over the LDS mask bits and writes 12 B of corrected BGR
// and 4 B per colour plane -- dword stores, fully coalesced.
// Algorithmic bytes per working pixel: 3 read + 3 (bgr) + 3 (masks) written.
#include "common.h"

namespace lf {

constexpr int TW = 128, TH = 8, PRE_THREADS = 256;

__device__ __forceinline__ int hsv_bits(int b, int g, int r, const PreParams& p, const int* __restrict__ sdiv,
                                        const int* __restrict__ hdiv)
{
    // OpenCV RGB2HSV_b, hsv_shift = 12, hue range 180
    int v = max(b, max(g, r)), vmin = min(b, min(g, r));
    int diff = v - vmin;
    int vr = v == r ? -1 : 0;
    int vg = v == g ? -1 : 0;
    int s = (diff * sdiv[v] + (1 << 11)) >> 12;
    int h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
    h = (h * hdiv[diff] + (1 << 11)) >> 12;
    h += h < 0 ? 180 : 0;
    h = min(max(h, 0), 255);
    int in[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        in[k] = (h >= p.lo[k][0]) & (h <= p.hi[k][0]) & (s >= p.lo[k][1]) & (s <= p.hi[k][1]) &
                (v >= p.lo[k][2]) & (v <= p.hi[k][2]);
    return in[0] | (in[1] << 1) | ((in[2] | in[3]) << 2);
}

__global__ __launch_bounds__(PRE_THREADS) void k_pre(PreParams p, const uint8_t* __restrict__ frames,
                                                      uint8_t* __restrict__ bgr_out, uint8_t* __restrict__ masks,
                                                      const int* __restrict__ sdiv, const int* __restrict__ hdiv)
{
    __shared__ uint32_t tile[(TH + 2 * (kMaxKsize / 2)) * (TW + 2 * (kMaxKsize / 2))];
    const int r = p.r;
    const int tw = TW + 2 * r, th = TH + 2 * r;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, f = blockIdx.z;
    const uint8_t* src = frames + (size_t)f * p.in_rows * p.in_cols * 3;

    for (int idx = threadIdx.x; idx < tw * th; idx += PRE_THREADS) {
        int ty = idx / tw, tx = idx - ty * tw;
        int gx = x0 + tx - r, gy = y0 + ty - r;
        uint32_t packed = 0;
        if (gx >= 0 && gx < p.W && gy >= 0 && gy < p.Hc) {
            int yy = gy + p.top_cutoff;
            int sy = yy, sx = gx;
            if (p.resize) {
                sy = min(dm::ifloor(yy * p.ify), p.in_rows - 1);
                sx = min(dm::ifloor(gx * p.ifx), p.in_cols - 1);
            }
            const uint8_t* q = src + ((size_t)sy * p.in_cols + sx) * 3;
            int c[3];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                float v = (float)q[ch] * p.ai_scale[ch];
                v = v + p.ai_shift[ch];
                float a = v < 0 ? -v : v;
                int iv = dm::round_half_even((double)a);
                c[ch] = min(max(iv, 0), 255);
            }
            int bits = hsv_bits(c[0], c[1], c[2], p, sdiv, hdiv);
            packed = (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16) | ((uint32_t)bits << 24);
        }
        tile[idx] = packed;
    }
    __syncthreads();

    const int lx = (threadIdx.x & 31) * 4, ly = threadIdx.x >> 5;
    const int gx = x0 + lx, gy = y0 + ly;
    if (gx >= p.W || gy >= p.Hc) return;
    uint32_t px[4];
    uint32_t m[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        px[k] = tile[(ly + r) * tw + lx + r + k];
        uint32_t bits = 0;
        for (int i = 0; i < p.ksize; ++i)
            for (int j = p.j1[i]; j < p.j2[i]; ++j)
                bits |= tile[(ly + i) * tw + lx + k + j] >> 24;
        m[0] |= ((bits & 1u) ? 0xFFu : 0u) << (8 * k);
        m[1] |= ((bits & 2u) ? 0xFFu : 0u) << (8 * k);
        m[2] |= ((bits & 4u) ? 0xFFu : 0u) << (8 * k);
    }
    // 4 pixels = 12 bytes of BGR -> 3 dwords
    uint32_t w0 = (px[0] & 0xFFFFFFu) | (px[1] << 24);
    uint32_t w1 = ((px[1] >> 8) & 0xFFFFu) | (px[2] << 16);
    uint32_t w2 = ((px[2] >> 16) & 0xFFu) | (px[3] << 8);
    const size_t P = (size_t)p.Hc * p.W;
    const size_t pix = (size_t)gy * p.W + gx;
    uint32_t* bo = reinterpret_cast<uint32_t*>(bgr_out + ((size_t)f * P + pix) * 3);
    bo[0] = w0; bo[1] = w1; bo[2] = w2;
    uint8_t* mo = masks + (size_t)f * 3 * P + pix;
    *reinterpret_cast<uint32_t*>(mo) = m[0];
    *reinterpret_cast<uint32_t*>(mo + P) = m[1];
    *reinterpret_cast<uint32_t*>(mo + 2 * P) = m[2];
}

void launch_pre(const PreParams& p, const uint8_t* frames, int n_frames, uint8_t* bgr, uint8_t* masks,
                const int* sdiv, const int* hdiv, hipStream_t s)
{
    dim3 grid((p.W + TW - 1) / TW, (p.Hc + TH - 1) / TH, n_frames);
    hipLaunchKernelGGL(k_pre, grid, dim3(PRE_THREADS), 0, s, p, frames, bgr, masks, sdiv, hdiv);
}

}  // namespace lf
