// K_pre: fused a-1 + a-2(HSV) + a-3(inRange, dilate) -- one HBM pass over the camera frames.
//
// Reference call sites (paths relative to /root/reference):
//   src/line_detector/src/line_detector_node.py:163-175   resize-nearest, crop, AntiInstagram, convertScaleAbs
//   src/anti_instagram/include/anti_instagram/scale_and_shift.py:25-33
//   src/line_detector/include/line_detector/line_detector_lsd.py:38-53,138   BGR2HSV, inRange (red = OR), dilate
//
// Layout: each workgroup owns a 128x30 tile of the working image (32 rows with the 3x3 halo).  Phase 1 converts the
// tile plus its dilation halo to packed (b,g,r,maskbits) words in LDS (HSV is computed
// once per pixel, never written to HBM).  Phase 2: each lane owns 4 adjacent pixels,
// ORs the structuring element over the LDS mask bits and writes 16 B of corrected BGRX (byte 3 is not cleared; one dword
// per pixel, so the Canny stencil loads whole pixels), 4 B of gray (BGR2GRAY, 1 byte per pixel: what the LBD
// gradient stage reads) and the three dilated colour masks as BIT PLANES, 1 bit per pixel (8 lanes OR their
// nibbles into one word): the LSD stage ANDs them with the edge bit plane, _findNormal tests single bits, and the
// 0/255 byte form the reference's `Detections.area` wants is expanded from them on demand (k_edges_u8).
// Algorithmic bytes per working pixel: 3 read + 3 + 3 written; moved: 3 read + 4 (BGRX) + 1 (gray) +
// 3/8 (mask bits) written.
#include "common.h"

namespace lf {

#ifndef LF_PRE_TILE_H
#define LF_PRE_TILE_H 22       // (TH + 2) rows x 32 groups = 768 = 3 per lane; 30 (4 per lane, 8 KB more LDS) measured 6 % slower, 14 slower still
#endif
constexpr int TW = 128, TH = LF_PRE_TILE_H, PRE_THREADS = 256;   // (TH + 2) rows x 32 four-pixel groups = a whole number of passes of 256 lanes

struct PixOut { uint32_t packed; };

// scaleandshift2 (float32) + convertScaleAbs of one pixel, B | G << 8 | R << 16 in and out (anything above bit 23 is ignored)
__device__ __forceinline__ uint32_t correct_pixel(uint32_t p24, const PreParams& p)
{
    uint32_t pk = p24 & 0xFFFFFFu;
    if (!p.identity_ai) {                      // scale 1, shift 0 leaves every u8 value unchanged: the packed pixel is the output
        int c[3] = { (int)(p24 & 255u), (int)((p24 >> 8) & 255u), (int)((p24 >> 16) & 255u) };
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            float v = (float)c[ch] * p.ai_scale[ch];
            v = v + p.ai_shift[ch];
            float a = v < 0 ? -v : v;
            int iv = (int)__builtin_rintf(a);  // v_rndne_f32: round half to even, as cvRound
            c[ch] = min(max(iv, 0), 255);
        }
        pk = (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16);
    }
    return pk;
}

// BGR2GRAY (fixed point, as cvtColor): (B * 1868 + G * 9617 + R * 4899 + 2^13) >> 14 with the coefficients split into bytes: two
// 4 x u8 dot products per pixel (byte 3 meets a zero coefficient); the result is <= 255
__device__ __forceinline__ uint32_t gray_of(uint32_t pk)
{
    const uint32_t lo = __builtin_amdgcn_udot4(pk, 0x0023914Cu, 1u << 13, false);
    const uint32_t hi = __builtin_amdgcn_udot4(pk, 0x00132507u, 0u, false);
    return ((hi << 8) + lo) >> 14;
}

// p24: B | G << 8 | R << 16 as it sits in the frame (anything above bit 23 is ignored)
__device__ __forceinline__ uint32_t convert_pixel(uint32_t p24, const PreParams& p, const int* sdiv, const int* hdiv,
                                                  const uint8_t* boxes)
{
    // scaleandshift2 (float32) + convertScaleAbs, then OpenCV RGB2HSV_b (hsv_shift 12, hue range 180) + 4 inRange boxes
    const uint32_t pk = correct_pixel(p24, p);
    const int b = (int)(pk & 255u), g = (int)((pk >> 8) & 255u), r = (int)((pk >> 16) & 255u);
    int v = max(b, max(g, r)), vmin = min(b, min(g, r));
    // Every inRange box needs its V interval first: a pixel whose V (= max channel) lies in none of them is in no mask
    // whatever its hue and saturation are, and those are the expensive part (two table divisions).  On road images most
    // pixels are dark asphalt, so whole waves leave here.
    const int in_v = boxes[512 + v];
    if (in_v == 0) return pk;
    int diff = v - vmin;
    int vr = v == r ? -1 : 0;
    int vg = v == g ? -1 : 0;
    // 24-bit multiplies (full rate; the compiler's 32 x 32 -> 64 multiply-add is a quarter of that): diff <= 255, sdiv <= 255 << 12,
    // |h| <= 1530, hdiv <= 180 << 12 / 6 -- both products are exact in 32 bits
    int s = (__mul24(diff, sdiv[v]) + (1 << 11)) >> 12;
    int h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
    h = (__mul24(h, hdiv[diff]) + (1 << 11)) >> 12;
    h += h < 0 ? 180 : 0;
    h = min(max(h, 0), 255);
    // inRange against the 4 HSV boxes: per-channel acceptance masks (bit k = box k) from LDS tables
    const int in4 = boxes[h] & boxes[256 + s] & in_v;
    const int bits = (in4 & 3) | (((in4 >> 2) | (in4 >> 3)) & 1) << 2;       // white, yellow, red = red1 | red2
    return pk | ((uint32_t)bits << 24);
}

__global__ __launch_bounds__(PRE_THREADS) void k_pre(PreParams p, const uint8_t* __restrict__ frames,
                                                      uint32_t* __restrict__ bgrx_out, uint8_t* __restrict__ gray_out,
                                                      uint32_t* __restrict__ maskbits, const int* __restrict__ sdiv_g,
                                                      const int* __restrict__ hdiv_g)
{
    // LDS tile rows: [4 - r pad][r halo][TW interior, 16-byte aligned][r halo][pad]
    constexpr int tw = TW + 8, XO = 4;
    __shared__ __attribute__((aligned(16))) uint32_t tile[(TH + 2 * (kMaxKsize / 2)) * tw];
    __shared__ int sdiv[256], hdiv[256];
    __shared__ uint8_t boxes[3 * 256];           // boxes[ch*256 + value] bit k: lo[k][ch] <= value <= hi[k][ch]
    const int r = p.r;
    const int th = TH + 2 * r;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, f = blockIdx.z;
    const uint8_t* src = frames + (size_t)f * p.in_rows * p.in_cols * 3;
    // The frame pixels do not depend on the tables below: in the common geometry (no resize) every lane's three-dword
    // groups are requested first, so their trip to HBM runs under the table setup and the barrier.
    const bool fast = !p.resize && (p.in_cols & 3) == 0 && (p.W & 3) == 0;
    constexpr int MAXG = ((TH + 2 * (kMaxKsize / 2)) * (TW / 4) + PRE_THREADS - 1) / PRE_THREADS;
    uint32_t D[MAXG][3];
    if (fast) {
#pragma unroll
        for (int it = 0; it < MAXG; ++it) {
            const int g = (int)threadIdx.x + it * PRE_THREADS;
            const int ty = g / (TW / 4), cg = g - ty * (TW / 4);
            const int gx = x0 + 4 * cg, gy = y0 + ty - r;
            D[it][0] = D[it][1] = D[it][2] = 0u;
            if (g < th * (TW / 4) && gx < p.W && gy >= 0 && gy < p.Hc) {
                // byte offset inside the frame in 32 bits (24-bit multiply: rows and columns are far below 2^24)
                const uint32_t off = (__umul24((uint32_t)(gy + p.top_cutoff), (uint32_t)p.in_cols) + (uint32_t)gx) * 3u;
                const uint32_t* q = reinterpret_cast<const uint32_t*>(src + off);
                D[it][0] = q[0]; D[it][1] = q[1]; D[it][2] = q[2];
            }
        }
    }
    sdiv[threadIdx.x] = sdiv_g[threadIdx.x];
    hdiv[threadIdx.x] = hdiv_g[threadIdx.x];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        int mk = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) mk |= (int)((int)threadIdx.x >= p.lo[k][ch] && (int)threadIdx.x <= p.hi[k][ch]) << k;
        boxes[ch * 256 + threadIdx.x] = (uint8_t)mk;
    }
    __syncthreads();

    if (fast) {
        // interior columns: one lane = 4 pixels = 3 aligned dwords of the source row
#pragma unroll
        for (int it = 0; it < MAXG; ++it) {
            const int g = (int)threadIdx.x + it * PRE_THREADS;
            if (g >= th * (TW / 4)) break;
            const int ty = g / (TW / 4), cg = g - ty * (TW / 4);
            const int gx = x0 + 4 * cg, gy = y0 + ty - r;
            uint32_t o[4] = {0, 0, 0, 0};
            if (gx < p.W && gy >= 0 && gy < p.Hc) {
                const uint32_t d0 = D[it][0], d1 = D[it][1], d2 = D[it][2];
                o[0] = convert_pixel(d0, p, sdiv, hdiv, boxes);
                o[1] = convert_pixel((d0 >> 24) | (d1 << 8), p, sdiv, hdiv, boxes);
                o[2] = convert_pixel((d1 >> 16) | (d2 << 16), p, sdiv, hdiv, boxes);
                o[3] = convert_pixel(d2 >> 8, p, sdiv, hdiv, boxes);
            }
            *reinterpret_cast<uint4*>(tile + ty * tw + XO + 4 * cg) = make_uint4(o[0], o[1], o[2], o[3]);
        }
        // halo columns (2*r per row): single pixels
        for (int i = threadIdx.x; i < th * 2 * r; i += PRE_THREADS) {
            const int ty = i / (2 * r), j = i - ty * (2 * r);
            const int tx = j < r ? XO - r + j : XO + TW + (j - r);   // left r columns, right r columns
            const int gx = x0 + tx - XO, gy = y0 + ty - r;
            uint32_t packed = 0;
            if (gx >= 0 && gx < p.W && gy >= 0 && gy < p.Hc) {
                const uint8_t* q = src + ((size_t)(gy + p.top_cutoff) * p.in_cols + gx) * 3;
                packed = convert_pixel((uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16), p, sdiv, hdiv, boxes);
            }
            tile[ty * tw + tx] = packed;
        }
    } else {
        const int twr = TW + 2 * r;
        for (int idx = threadIdx.x; idx < twr * th; idx += PRE_THREADS) {
            int ty = idx / twr, tx = idx - ty * twr + XO - r;
            int gx = x0 + tx - XO, gy = y0 + ty - r;
            uint32_t packed = 0;
            if (gx >= 0 && gx < p.W && gy >= 0 && gy < p.Hc) {
                int yy = gy + p.top_cutoff;
                int sy = yy, sx = gx;
                if (p.resize) {
                    sy = min(dm::ifloor(yy * p.ify), p.in_rows - 1);
                    sx = min(dm::ifloor(gx * p.ifx), p.in_cols - 1);
                }
                const uint8_t* q = src + ((size_t)sy * p.in_cols + sx) * 3;
                packed = convert_pixel((uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16), p, sdiv, hdiv, boxes);
            }
            tile[ty * tw + tx] = packed;
        }
    }
    __syncthreads();

    const size_t P = (size_t)p.Hc * p.W;
    const int Ww = (p.W + 31) >> 5;
    const int sub = threadIdx.x & 7;
    // phase 2: TH rows x 32 groups; every lane of a wave takes part in the shuffles
    for (int g = threadIdx.x; g < ((TH * (TW / 4) + PRE_THREADS - 1) / PRE_THREADS) * PRE_THREADS; g += PRE_THREADS) {
        const int ly = g / (TW / 4), lx = (g - ly * (TW / 4)) * 4;
        const int gx = x0 + lx, gy = y0 + ly;
        const bool valid = ly < TH && gx < p.W && gy < p.Hc;
        uint32_t px[4] = {0, 0, 0, 0};
        uint32_t nib[3] = {0, 0, 0};
        if (valid) {
            const uint32_t* ctr = tile + (ly + r) * tw + XO + lx;
            const uint4 c4 = *reinterpret_cast<const uint4*>(ctr);
            px[0] = c4.x; px[1] = c4.y; px[2] = c4.z; px[3] = c4.w;
            uint32_t cross[4] = {0, 0, 0, 0};
            if (p.ksize == 3) {
                // 3x3 MORPH_ELLIPSE is the plus shape: five 16-byte / 4-byte LDS reads serve 4 pixels
                const uint4 up = *reinterpret_cast<const uint4*>(ctr - tw), dn = *reinterpret_cast<const uint4*>(ctr + tw);
                const uint32_t lft = ctr[-1], rgt = ctr[4];
                cross[0] = c4.x | lft | c4.y | up.x | dn.x;
                cross[1] = c4.y | c4.x | c4.z | up.y | dn.y;
                cross[2] = c4.z | c4.y | c4.w | up.z | dn.z;
                cross[3] = c4.w | c4.z | rgt | up.w | dn.w;
            }
            if (p.ksize != 3) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    uint32_t bits = 0;
                    for (int i = 0; i < p.ksize; ++i)
                        for (int j = p.j1[i]; j < p.j2[i]; ++j)
                            bits |= tile[(ly + i) * tw + XO - r + lx + k + j] >> 24;
                    cross[k] = bits << 24;
                }
            }
            // the four pixels' mask bytes in one word, then one dot product per colour gathers bit c of every byte into a
            // nibble: sum_k ((byte_k >> c) & 1) << k
            const uint32_t t4 = __builtin_amdgcn_perm(cross[1], cross[0], 0x0c0c0703u) | __builtin_amdgcn_perm(cross[3], cross[2], 0x07030c0cu);
#pragma unroll
            for (int c = 0; c < 3; ++c)
                nib[c] = __builtin_amdgcn_udot4(t4 & (0x01010101u << c), 0x08040201u, 0u, false) >> c;
        }
        // bit-plane copy of the dilated masks (1 bit / pixel): 8 lanes x 4 pixels = one 32-pixel word
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            uint32_t w = nib[c] << (4 * sub);
            w |= __shfl_xor(w, 1);
            w |= __shfl_xor(w, 2);
            w |= __shfl_xor(w, 4);
            nib[c] = w;
        }
        if (!valid) continue;
        if (sub == 0) {
            uint32_t* mb = maskbits + ((size_t)f * 3 * p.Hc + gy) * Ww + (gx >> 5);
            mb[0] = nib[0];
            mb[(size_t)p.Hc * Ww] = nib[1];
            mb[(size_t)2 * p.Hc * Ww] = nib[2];
        }
        // corrected working image as BGRX dwords: one 16-byte store for the lane's 4 pixels
        const size_t pix = (size_t)gy * p.W + gx;
        *reinterpret_cast<uint4*>(bgrx_out + (size_t)f * P + pix) =
            make_uint4(px[0], px[1], px[2], px[3]);      // byte 3 still holds the pixel's own mask bits: no reader looks at it
        // BGR2GRAY (fixed point, as cvtColor) of the same four pixels: the 1 byte/pixel plane the LBD gradient stage
        // reads instead of the 4 byte/pixel working image (binary_descriptor_custom.cpp:350-398 works on gray)
        // (B * 1868 + G * 9617 + R * 4899 + 2^13) >> 14 with the coefficients split into bytes: two 4 x u8 dot products
        // per pixel (byte 3 meets a zero coefficient); the result is <= 255
        uint32_t gq = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) gq |= gray_of(px[k]) << (8 * k);
        *reinterpret_cast<uint32_t*>(gray_out + (size_t)f * P + pix) = gq;
    }
}

// The gray working image alone -- resize-nearest / crop, colour correction, BGR2GRAY: what the octave detectors read
// (lf_keylines_batch, lf_lsd_keylines_batch from BGR frames; k_pre's masks, HSV and BGRX image were 0.11 ms per batch for
// nothing there).  One thread = four horizontally adjacent pixels; the same two functions as k_pre, so the same bytes.
__global__ __launch_bounds__(256) void k_pre_gray(PreParams p, const uint8_t* __restrict__ frames, uint8_t* __restrict__ gray_out)
{
    const int gx = (blockIdx.x * 256 + threadIdx.x) * 4, gy = blockIdx.y, f = blockIdx.z;
    if (gx >= p.W) return;
    const uint8_t* src = frames + (size_t)f * p.in_rows * p.in_cols * 3;
    uint32_t px[4] = {0, 0, 0, 0};
    const bool fast = !p.resize && (p.in_cols & 3) == 0 && (p.W & 3) == 0;
    if (fast) {
        const uint32_t off = (__umul24((uint32_t)(gy + p.top_cutoff), (uint32_t)p.in_cols) + (uint32_t)gx) * 3u;
        const uint32_t* q = reinterpret_cast<const uint32_t*>(src + off);
        const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
        px[0] = d0; px[1] = (d0 >> 24) | (d1 << 8); px[2] = (d1 >> 16) | (d2 << 16); px[3] = d2 >> 8;
    } else {
        const int yy = gy + p.top_cutoff;
        const int sy = p.resize ? min(dm::ifloor(yy * p.ify), p.in_rows - 1) : yy;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (gx + k >= p.W) break;
            const int sx = p.resize ? min(dm::ifloor((gx + k) * p.ifx), p.in_cols - 1) : gx + k;
            const uint8_t* q = src + ((size_t)sy * p.in_cols + sx) * 3;
            px[k] = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16);
        }
    }
    uint32_t gq = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) gq |= gray_of(correct_pixel(px[k], p)) << (8 * k);
    uint8_t* out = gray_out + (size_t)f * p.Hc * p.W + (size_t)gy * p.W + gx;
    if ((p.W & 3) == 0) *reinterpret_cast<uint32_t*>(out) = gq;
    else for (int k = 0; k < 4 && gx + k < p.W; ++k) out[k] = (uint8_t)(gq >> (8 * k));
}

void launch_pre_gray(const PreParams& p, const uint8_t* frames, int n_frames, uint8_t* gray, hipStream_t s)
{
    dim3 grid((p.W + 1023) / 1024, p.Hc, n_frames);
    hipLaunchKernelGGL(k_pre_gray, grid, dim3(256), 0, s, p, frames, gray);
}

void launch_pre(const PreParams& p, const uint8_t* frames, int n_frames, uint32_t* bgr, uint8_t* gray,
                uint32_t* maskbits, const int* sdiv, const int* hdiv, hipStream_t s)
{
    dim3 grid((p.W + TW - 1) / TW, (p.Hc + TH - 1) / TH, n_frames);
    hipLaunchKernelGGL(k_pre, grid, dim3(PRE_THREADS), 0, s, p, frames, bgr, gray, maskbits, sdiv, hdiv);
}

}  // namespace lf
