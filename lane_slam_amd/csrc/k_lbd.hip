// LBD line descriptor (a-9).
//
// Reference (never built by the reference itself; restated, see oracle/lf_oracle_lbd.c):
//   /root/reference/src/line_descriptor/src/binary_descriptor_custom.cpp
//     :350-398 computeSobel   (cvtColor BGR2GRAY, GaussianBlur 5x5 sigma 1, Sobel 3x3 -> s16)
//     :1026-1372 computeLBD   :401-412,653-667 binary code   :74-107 pair table
//   /root/reference/src/line_descriptor/src/LSDDetector_custom.cpp:73-102,169-197 KeyLine fields
//
// k_lbd_grad: 64x16 tile per workgroup; gray plane written by k_pre (+3 halo, BORDER_REFLECT_101) -> LDS, 5x5
//   fixed-point Gaussian {14,63,103,63,14}/256 twice -> LDS, Sobel -> s16 dx,dy.  Integer exact.
//   Algorithmic bytes per pixel (SURVEY 8d K_sobel_lbd): 1 read (gray), 4 written.
// k_lbd: ONE WAVE PER SEGMENT.  Lane r (< 63) walks row r of the 63 x len support region
//   with the reference's running float coordinates (rounded per step, clamped), gathering
//   s16 gradients; the per-row sums are scaled by the global Gaussian, staged in LDS, and
//   lanes 0..71 each accumulate one (band, statistic) sum in the reference's row order;
//   the three normalisations are wave reductions done in the reference's summation order.
#include "common.h"

namespace lf {

#ifndef LF_LBD_TILE_H
#define LF_LBD_TILE_H 64
#endif
constexpr int LT_W = 64, LT_H = LF_LBD_TILE_H;   // 64 rows: 9 % halo rows instead of 38 % at 16, and the three filter phases fill their last pass of 256 lanes better (0.119 -> 0.090 ms)

__device__ __forceinline__ int refl101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) { p = p < 0 ? -p : 2 * (n - 1) - p; }
    return p;
}

// Four horizontally adjacent outputs per lane in every phase: the LDS tiles are read as dwords /
// 16-byte vectors with sliding windows instead of one byte (or int) per tap.
__global__ __launch_bounds__(256) void k_lbd_grad(int Hc, int W, const uint8_t* __restrict__ gray_in,
                                                  uint32_t* __restrict__ dxyo)
{
    constexpr int GW = 72, GH = LT_H + 6;            // gray tile: 70 columns used (x0-3 .. x0+66), rows padded to dwords
    constexpr int RW = 68, RG = RW / 4;              // row-filtered: 66 columns used (x0-1 .. x0+64)
    constexpr int BW_ = 72, BH = LT_H + 2;           // blurred: column c <-> x0-1+c, 66 used
    __shared__ __attribute__((aligned(16))) uint8_t gray[GH * GW];
    __shared__ __attribute__((aligned(16))) uint16_t rowf[GH * RW];     // <= 257 * 255 = 65 535: exactly 16 bits
    __shared__ __attribute__((aligned(16))) uint8_t blur[BH * BW_];
    int tbx, tby, f;
    lf_xcd_tile(tbx, tby, f);
    const int x0 = tbx * LT_W, y0 = tby * LT_H;
    const int tid = threadIdx.y * 64 + threadIdx.x;
    const uint8_t* img = gray_in + (size_t)f * Hc * W;
    // gray tile from k_pre's 1 byte/pixel plane (BGR2GRAY is done there), 4 pixels per lane -> one dword store.
    // The tile starts at x0 - 3: interior groups take the two aligned dwords around their four bytes and shift,
    // groups that touch the image border reflect byte by byte (BORDER_REFLECT_101).
    for (int idx = tid; idx < GH * (GW / 4); idx += 256) {
        const int ty = idx / (GW / 4), g = idx - ty * (GW / 4);
        const int gy = refl101(y0 + ty - 3, Hc);
        const int xa = x0 + 4 * g - 4;                   // aligned dword below the group (W is a multiple of 32)
        uint32_t packed = 0;
        if (xa >= 0 && xa + 7 < W) {
            const uint32_t* q = reinterpret_cast<const uint32_t*>(img + (size_t)gy * W + xa);
            packed = (q[0] >> 8) | (q[1] << 24);         // bytes xa+1 .. xa+4 = x0 + 4g - 3 .. x0 + 4g
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) packed |= (uint32_t)img[(size_t)gy * W + refl101(x0 + 4 * g + k - 3, W)] << (8 * k);
        }
        *reinterpret_cast<uint32_t*>(gray + ty * GW + 4 * g) = packed;
    }
    __syncthreads();
    // horizontal 5-tap {14,63,103,63,14}: outputs c..c+3 (c = 4g) need gray[c .. c+7] = two dwords
    for (int idx = tid; idx < GH * RG; idx += 256) {
        const int ty = idx / RG, g = idx - ty * RG;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(gray + ty * GW + 4 * g);
        const uint32_t lo = src[0], hi = src[1];
        // two outputs per instruction in 16-bit lanes (the sums stay <= 65 535): E = (b0, b2), O = (b1, b3), ... ; the plane
        // keeps the pairs as they come out: word 0 = (out0, out2), word 1 = (out1, out3)
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
        const us2 E = __builtin_bit_cast(us2, lo & 0x00ff00ffu), O = __builtin_bit_cast(us2, (lo >> 8) & 0x00ff00ffu);
        const us2 E2 = __builtin_bit_cast(us2, hi & 0x00ff00ffu), O2 = __builtin_bit_cast(us2, (hi >> 8) & 0x00ff00ffu);
        const us2 P24 = { E.y, E2.x }, P35 = { O.y, O2.x };
        const us2 c14 = { 14, 14 }, c63 = { 63, 63 }, c103 = { 103, 103 };
        const us2 o02 = c14 * E + c63 * O + c103 * P24 + c63 * P35 + c14 * E2;
        const us2 o13 = c14 * O + c63 * P24 + c103 * P35 + c63 * E2 + c14 * O2;
        *reinterpret_cast<uint2*>(rowf + ty * RW + 4 * g) = make_uint2(__builtin_bit_cast(uint32_t, o02), __builtin_bit_cast(uint32_t, o13));
    }
    __syncthreads();
    // vertical 5-tap, (acc + 2^15) >> 16, saturate -> blurred u8
    for (int idx = tid; idx < BH * RG; idx += 256) {
        const int ty = idx / RG, g = idx - ty * RG;
        int r5[5][4];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const uint2 w = *reinterpret_cast<const uint2*>(rowf + (ty + j) * RW + 4 * g);
            r5[j][0] = (int)(w.x & 0xffffu); r5[j][2] = (int)(w.x >> 16); r5[j][1] = (int)(w.y & 0xffffu); r5[j][3] = (int)(w.y >> 16);
        }
        int v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = 14 * r5[0][k] + 63 * r5[1][k] + 103 * r5[2][k] + 63 * r5[3][k] + 14 * r5[4][k];
        uint32_t q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int t = (v[k] + (1 << 15)) >> 16;
            // ROCm 7.2 / gfx950: hipcc fuses "shift, clamp to u8, pack two" into v_ashr_pk_u8_i32 and then
            // ORs further bytes into the result as if its upper 16 bits were zero; on the MI355X they are
            // not (byte 2 of the packed word came out wrong, found by the parity test).  The empty asm
            // keeps the shift and the clamp apart so the fused instruction is never selected.
            asm volatile("" : "+v"(t));
            t = t < 0 ? 0 : (t > 255 ? 255 : t);
            q[k] = (uint32_t)t;
        }
        *reinterpret_cast<uint32_t*>(blur + ty * BW_ + 4 * g) = q[0] | (q[1] << 8) | (q[2] << 16) | (q[3] << 24);
    }
    __syncthreads();
    // Sobel 3x3 on the blurred tile: 4 outputs per lane from three rows of 6 bytes (two dwords each)
    for (int r0 = 0; r0 < LT_H; r0 += 16) {
        const int ry = r0 + (tid >> 4), g = tid & 15;  // 16 rows x 16 groups = 256 lanes per pass
        const int lx = 4 * g, gx = x0 + lx, gy = y0 + ry;
        if (gx < W && gy < Hc) {
            // packed 16-bit lanes again: per row the column pairs P02 = (c0, c2), P13, P24, P35 of its six bytes; column sums
            // S = r0 + 2 r1 + r2 and differences D = r2 - r0, then vx = S[k+2] - S[k], vy = D[k] + 2 D[k+1] + D[k+2]
            typedef short s2 __attribute__((ext_vector_type(2)));
            s2 P[3][4];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const uint32_t* src = reinterpret_cast<const uint32_t*>(blur + (ry + j) * BW_ + lx);   // columns lx .. lx+7 <-> x-1 ..
                const uint32_t lo = src[0], hi = src[1];
                const s2 E = __builtin_bit_cast(s2, lo & 0x00ff00ffu), O = __builtin_bit_cast(s2, (lo >> 8) & 0x00ff00ffu);
                const s2 E2 = __builtin_bit_cast(s2, hi & 0x00ff00ffu), O2 = __builtin_bit_cast(s2, (hi >> 8) & 0x00ff00ffu);
                P[j][0] = E; P[j][1] = O; P[j][2] = s2{ E.y, E2.x }; P[j][3] = s2{ O.y, O2.x };
            }
            const s2 two = { 2, 2 };
            s2 S[4], D[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) { S[c] = P[0][c] + two * P[1][c] + P[2][c]; D[c] = P[2][c] - P[0][c]; }
            const s2 vx02 = S[2] - S[0], vx13 = S[3] - S[1];
            const s2 vy02 = D[0] + two * D[1] + D[2], vy13 = D[1] + two * D[2] + D[3];
            const int vx[4] = { vx02.x, vx13.x, vx02.y, vx13.y }, vy[4] = { vy02.x, vy13.x, vy02.y, vy13.y };
            const size_t o = (size_t)f * Hc * W + (size_t)gy * W + gx;
            // dx and dy of a pixel share one dword (dx low, dy high): the descriptor kernel fetches both with one gather
            uint32_t w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] = (uint32_t)(uint16_t)vx[k] | ((uint32_t)(uint16_t)vy[k] << 16);
            if (gx + 3 < W && (W & 3) == 0) {
                *reinterpret_cast<uint4*>(dxyo + o) = make_uint4(w[0], w[1], w[2], w[3]);
            } else {
                for (int k = 0; k < 4 && gx + k < W; ++k) dxyo[o + k] = w[k];
            }
        }
    }
}

void launch_lbd_grad(int Hc, int W, int n_frames, const uint8_t* gray, uint32_t* dxy, hipStream_t s)
{
    dim3 grid((W + LT_W - 1) / LT_W, (Hc + LT_H - 1) / LT_H, n_frames);
    hipLaunchKernelGGL(k_lbd_grad, grid, dim3(64, 4), 0, s, Hc, W, gray, dxy);
}

__constant__ int c_comb[32][2] = {
    {0,1},{0,2},{0,3},{0,4},{0,5},{0,6},{1,2},{1,3},{1,4},{1,5},{1,6},{2,3},{2,4},{2,5},{2,6},{2,7},
    {2,8},{3,4},{3,5},{3,6},{3,7},{3,8},{4,5},{4,6},{4,7},{4,8},{5,6},{5,7},{5,8},{6,7},{6,8},{7,8} };

constexpr int NBANDS = 9, WBAND = 7, LSP_H = 63;
constexpr int LBD_STEPS = 8;      // support-region columns fetched per round trip

// KL = false: segments of the LSD front end (KeyLine fields worked out here, LSDDetector_custom.cpp:73-102,169-197, one
// octave).  KL = true: KeyLines of the EDLines / multi-octave path (k_edlines.hip, lf_describe_keylines): endpoints in the
// octave image, direction and numOfPixels are given, and every line names its octave's gradient plane (computeLBD,
// binary_descriptor_custom.cpp:1070-1100: edLineVec_[octave]->dxImg_ or dxImg_vector[octave]).
template <bool KL>
__global__ __launch_bounds__(256) void k_lbd(int Hc_, int W_, const int* __restrict__ n_seg_ptr,
                                             const float* __restrict__ lines, const int* __restrict__ seg_frame,
                                             const uint32_t* __restrict__ dxyi,
                                             const float* __restrict__ gauss_g /*63*/, const float* __restrict__ gauss_l /*21*/,
                                             float* __restrict__ desc, uint8_t* __restrict__ code,
                                             LbdPlanes planes, const float* __restrict__ kl_angle, const int* __restrict__ kl_npx,
                                             const int* __restrict__ kl_octave)
{
    __shared__ float rows[4][LSP_H][4];      // per wave: row sums pgdL, ngdL, pgdO, ngdO (already * coefG)
    __shared__ float dsc[4][72];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n_seg = *n_seg_ptr;
    // the segment count is only known on the device: a fixed grid of waves strides over the segments
  for (int seg = blockIdx.x * 4 + wave; seg < n_seg; seg += gridDim.x * 4) {
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const int f = seg_frame[seg];
    int Hc = Hc_, W = W_;
    const uint32_t* pdxy;
    float e0 = lines[4 * (size_t)seg], e1 = lines[4 * (size_t)seg + 1], e2 = lines[4 * (size_t)seg + 2], e3 = lines[4 * (size_t)seg + 3];
    int lengthOfLSP;
    float direction;
    if (KL) {
        const int oc = kl_octave[seg];
        Hc = planes.H[oc]; W = planes.W[oc];
        pdxy = planes.base[oc] + (size_t)f * Hc * W;
        lengthOfLSP = (int)(short)kl_npx[seg];                  // `short lengthOfLSP` (:1106)
        direction = kl_angle[seg];
    } else {
    pdxy = dxyi + (size_t)f * Hc * W;
    // KeyLine fields (LSDDetector_custom.cpp:73-102,169-197), octave 0
    if (e0 < 0) e0 = 0;
    if (e0 >= W) e0 = (float)W - 1.0f;
    if (e2 < 0) e2 = 0;
    if (e2 >= W) e2 = (float)W - 1.0f;
    if (e1 < 0) e1 = 0;
    if (e1 >= Hc) e1 = (float)Hc - 1.0f;
    if (e3 < 0) e3 = 0;
    if (e3 >= Hc) e3 = (float)Hc - 1.0f;
    const int ix0 = dm::round_half_even((double)e0), iy0 = dm::round_half_even((double)e1);
    const int ix1 = dm::round_half_even((double)e2), iy1 = dm::round_half_even((double)e3);
    lengthOfLSP = max(abs(ix1 - ix0), abs(iy1 - iy0)) + 1;
    const float ddy = e3 - e1, ddx = e2 - e0;
    direction = (float)dm::datan2((double)ddy, (double)ddx);
    }
    const int halfWidth = (lengthOfLSP - 1) / 2;
    const int halfHeight = (LSP_H - 1) / 2;
    const int imageWidth = W - 1, imageHeight = Hc - 1;
    const float midX = (float)(0.5 * (e0 + e2));
    const float midY = (float)(0.5 * (e1 + e3));
    double sn_, cs_;
    dm::dsincos((double)direction, sn_, cs_);
    const float dL0 = (float)cs_, dL1 = (float)sn_;
    const float dO0 = -dL1, dO1 = dL0;
    float sCorX0 = -dL0 * halfWidth + dL1 * halfHeight + midX;
    float sCorY0 = -dL1 * halfWidth - dL0 * halfHeight + midY;
    if (lane < LSP_H) {
        // the reference advances the row origin by repeated float updates: replay them
        for (int hh = 0; hh < lane; ++hh) { sCorX0 -= dL1; sCorY0 += dL0; }
        float sCorX = sCorX0, sCorY = sCorY0;
        float pgdL = 0, ngdL = 0, pgdO = 0, ngdO = 0;
        // coordinates never depend on the gathered data: LBD_STEPS steps of addresses first, 2 * LBD_STEPS loads
        // in flight per lane (the longest segment of a batch sets the kernel's duration: fewer, fuller round trips)
        for (int w0 = 0; w0 < lengthOfLSP; w0 += LBD_STEPS) {
            uint32_t dxyv[LBD_STEPS];
#pragma unroll
            for (int u = 0; u < LBD_STEPS; ++u) {
                // (a branch-free form of the rounding -- integer +-1 corrections by compare -- measured 30 % slower here)
                int tx = (int)dm::round_half_away_f(sCorX);
                int xCor = tx < 0 ? 0 : (tx > imageWidth ? imageWidth : tx);
                int ty = (int)dm::round_half_away_f(sCorY);
                int yCor = ty < 0 ? 0 : (ty > imageHeight ? imageHeight : ty);
                dxyv[u] = pdxy[yCor * W + xCor];
                sCorX += dL0;
                sCorY += dL1;
            }
#pragma unroll
            for (int u = 0; u < LBD_STEPS; ++u) {
                if (w0 + u >= lengthOfLSP) break;
                const float gx_ = (float)(int)(int16_t)(dxyv[u] & 0xffffu), gy_ = (float)((int)dxyv[u] >> 16);
                float gDL = gx_ * dL0 + gy_ * dL1;
                float gDO = gx_ * dO0 + gy_ * dO1;
                if (gDL > 0) pgdL += gDL; else ngdL -= gDL;
                if (gDO > 0) pgdO += gDO; else ngdO -= gDO;
            }
        }
        const float cg = gauss_g[lane];
        rows[wave][lane][0] = cg * pgdL;
        rows[wave][lane][1] = cg * ngdL;
        rows[wave][lane][2] = cg * pgdO;
        rows[wave][lane][3] = cg * ngdO;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // band accumulation: 72 sums (band b, statistic q in 0..7: pgdL ngdL pgdO ngdO pgdL2 ngdL2 pgdO2 ngdO2),
    // each the reference's ordered sum over rows hID of band b-1 (tap hID%7), b (tap +7), b+1 (tap +14)
    float* D = dsc[wave];
    for (int e = lane; e < 72; e += 64) {
        const int b = e >> 3, q = e & 7;
        const bool sq = q >= 4;
        const int src = q & 3;
        // rows (b - 1) * 7 .. (b + 2) * 7 - 1 inside the support region, in order; row h0 + i always meets tap i of
        // the 21-tap local Gaussian (bands b - 1, b, b + 1 use taps 0-6, 7-13, 14-20): uniform coefficients, no
        // division, the row sums at fixed LDS offsets
        const int h0 = (b - 1) * WBAND;
        const float* rp = &rows[wave][0][src] + 4 * h0;
        float acc = 0;
#pragma unroll
        for (int i = 0; i < 3 * WBAND; ++i) {
            const int hID = h0 + i;
            if (hID >= 0 && hID < LSP_H) {
                const float coef = gauss_l[i];
                const float v = rp[4 * i];
                if (!sq) acc += coef * v;
                else acc += coef * coef * (v * v);
            }
        }
        D[e] = acc;     // temporarily: band sums laid out [b][q]
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // mean / std per band (lanes 0..8), reference layout: desVec[b*8 + {0:pgdL,1:ngdL,2:pgdO,3:ngdO, 4..7 std}]
    float mean[4] = {0, 0, 0, 0}, sd[4] = {0, 0, 0, 0};
    if (lane < NBANDS) {
        const float invN = (lane == 0 || lane == NBANDS - 1) ? (float)(1.0 / (WBAND * 2.0)) : (float)(1.0 / (WBAND * 3.0));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float temp = D[lane * 8 + q] * invN;
            mean[q] = temp;
            sd[q] = (float)dm::dsqrt((double)(D[lane * 8 + 4 + q] * invN - temp * temp));
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (lane < NBANDS) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { D[lane * 8 + q] = mean[q]; D[lane * 8 + 4 + q] = sd[q]; }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // normalisation: sequential float sums in the reference's order (uniform across lanes)
    float tempM = 0, tempS = 0;
    for (int b = 0; b < NBANDS; ++b) {
        const float* v = D + 8 * b;
        tempM += v[0] * v[0]; tempM += v[1] * v[1]; tempM += v[2] * v[2]; tempM += v[3] * v[3];
        tempS += v[4] * v[4]; tempS += v[5] * v[5]; tempS += v[6] * v[6]; tempS += v[7] * v[7];
    }
    // binary_descriptor_custom.cpp:1301-1302: std::sqrt(float), then a float division -- two float roundings
    tempM = dm::fdiv(1.f, dm::fsqrt(tempM));
    tempS = dm::fdiv(1.f, dm::fsqrt(tempS));
    __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < 72; e += 64) {
        float v = D[e] * (((e & 7) < 4) ? tempM : tempS);
        if ((double)v > 0.4) v = (float)0.4;
        D[e] = v;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    float temp = 0;
    for (int i = 0; i < 72; ++i) temp += D[i] * D[i];
    temp = dm::fdiv(1.f, dm::fsqrt(temp));          // :1337
    __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < 72; e += 64) {
        float v = D[e] * temp;
        D[e] = v;
        if (desc) desc[(size_t)seg * 72 + e] = v;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (code && lane < 32) {
        const float* f1 = D + 8 * c_comb[lane][0];
        const float* f2 = D + 8 * c_comb[lane][1];
        unsigned r = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) if (f1[i] > f2[i]) r += 1u << i;
        code[(size_t)seg * 32 + lane] = (uint8_t)r;
    }
  }
}

void launch_lbd(int Hc, int W, int n_seg_cap, const int* n_seg, const float* lines, const int* seg_frame,
                const uint32_t* dxy, const float* gauss_g, const float* gauss_l,
                float* desc, uint8_t* code, hipStream_t s)
{
    if (n_seg_cap <= 0) return;
    int blocks = (n_seg_cap + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    LbdPlanes none;
    for (int i = 0; i < LF_MAX_OCTAVES; ++i) { none.base[i] = nullptr; none.W[i] = 0; none.H[i] = 0; }
    hipLaunchKernelGGL(k_lbd<false>, dim3(blocks), dim3(256), 0, s, Hc, W, n_seg, lines, seg_frame, dxy,
                       gauss_g, gauss_l, desc, code, none, nullptr, nullptr, nullptr);
}

void launch_lbd_keylines(const LbdPlanes& planes, int n_cap, const int* n_lines, const float* in_octave4, const float* angle, const int* npx,
                         const int* octave, const int* frame, const float* gauss_g, const float* gauss_l, float* desc, uint8_t* code,
                         hipStream_t s)
{
    if (n_cap <= 0) return;
    int blocks = (n_cap + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_lbd<true>, dim3(blocks), dim3(256), 0, s, 0, 0, n_lines, in_octave4, frame, nullptr, gauss_g, gauss_l, desc, code,
                       planes, angle, npx, octave);
}

// Debug only: the two s16 planes tests compare with the oracle's Sobel output.
__global__ void k_lbd_split_debug(size_t n, const uint32_t* __restrict__ dxy, int16_t* __restrict__ dx, int16_t* __restrict__ dy)
{
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t v = dxy[i];
    dx[i] = (int16_t)(v & 0xffffu);
    dy[i] = (int16_t)(v >> 16);
}

void launch_lbd_split_debug(size_t n, const uint32_t* dxy, int16_t* dx, int16_t* dy, hipStream_t s)
{
    hipLaunchKernelGGL(k_lbd_split_debug, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, dxy, dx, dy);
}

}  // namespace lf
