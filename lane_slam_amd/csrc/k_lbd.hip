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
static_assert(LT_H % 16 == 0, "the Sobel phase gives every lane LT_H / 16 rows of its column group");

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
    __shared__ __attribute__((aligned(16))) uint32_t rowp[(GH / 2) * RW];   // row-filtered, <= 257 * 255 = 65 535 = exactly 16 bits: (row 2m, row 2m + 1) per column
    __shared__ __attribute__((aligned(16))) uint8_t blur[BH * BW_];
    int tbx, tby, f;
    lf_xcd_tile(tbx, tby, f);
    const int x0 = tbx * LT_W, y0 = tby * LT_H;
    const int tid = threadIdx.y * 64 + threadIdx.x;
    const uint8_t* img = gray_in + (size_t)f * Hc * W;
    // gray tile from k_pre's 1 byte/pixel plane (BGR2GRAY is done there), 4 pixels per lane -> one dword store.
    // The tile starts at x0 - 3: a group takes the two aligned dwords around its four bytes and shifts.  Tiles that touch
    // the left or right image border (two of ten tile columns at 640) reflect byte by byte (BORDER_REFLECT_101); the
    // others skip every per-pixel border test -- this phase was 40 % of the kernel's vector instructions with the tests
    // in every group.  Rows reflect once (a tile reaches 3 rows past the image; refl101's loop only runs for images
    // shorter than that).
    const bool interior = x0 >= 4 && x0 + 72 <= W;           // wave-uniform: every group's dword pair lies inside the row
    auto reflect_row = [&](int ty) {
        int gy = y0 + ty - 3;
        gy = gy < 0 ? -gy : (gy >= Hc ? 2 * (Hc - 1) - gy : gy);
        if (gy < 0 || gy >= Hc) gy = refl101(y0 + ty - 3, Hc);
        return (uint32_t)gy * (uint32_t)W;
    };
    if (interior) {
        const uint8_t* base = img + (x0 - 4);
        for (int idx = tid; idx < GH * (GW / 4); idx += 256) {
            const int ty = idx / (GW / 4), g = idx - ty * (GW / 4);
            const uint32_t* q = reinterpret_cast<const uint32_t*>(base + (reflect_row(ty) + 4u * (uint32_t)g));
            *reinterpret_cast<uint32_t*>(gray + ty * GW + 4 * g) = (q[0] >> 8) | (q[1] << 24);   // bytes xa+1 .. xa+4 = x0 + 4g - 3 .. x0 + 4g
        }
    } else {
        for (int idx = tid; idx < GH * (GW / 4); idx += 256) {
            const int ty = idx / (GW / 4), g = idx - ty * (GW / 4);
            const uint32_t row = reflect_row(ty);
            const int xa = x0 + 4 * g - 4;               // aligned dword below the group (W is a multiple of 32)
            uint32_t packed = 0;
            if (xa >= 0 && xa + 7 < W) {
                const uint32_t* q = reinterpret_cast<const uint32_t*>(img + (row + (uint32_t)xa));
                packed = (q[0] >> 8) | (q[1] << 24);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) packed |= (uint32_t)img[row + (uint32_t)refl101(x0 + 4 * g + k - 3, W)] << (8 * k);
            }
            *reinterpret_cast<uint32_t*>(gray + ty * GW + 4 * g) = packed;
        }
    }
    __syncthreads();
    // horizontal 5-tap {14,63,103,63,14}: outputs c..c+3 (c = 4g) need gray[c .. c+7] = two dwords.  A lane filters the
    // SAME four columns of two consecutive rows and stores them paired by row -- rowp[m][c] = (row 2m, row 2m + 1) of
    // column c as two 16-bit halves -- which is the operand shape of v_dot2_u32_u16 in the vertical pass below.
    static_assert(GH % 2 == 0 && BH % 2 == 0, "rows are filtered in pairs");
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    for (int idx = tid; idx < (GH / 2) * RG; idx += 256) {
        const int m = idx / RG, g = idx - m * RG;
        uint32_t o02[2], o13[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(gray + (2 * m + r) * GW + 4 * g);
            const uint32_t lo = src[0], hi = src[1];
            // two outputs per instruction in 16-bit lanes (the sums stay <= 65 535): E = (b0, b2), O = (b1, b3), ...
            const us2 E = __builtin_bit_cast(us2, lo & 0x00ff00ffu), O = __builtin_bit_cast(us2, (lo >> 8) & 0x00ff00ffu);
            const us2 E2 = __builtin_bit_cast(us2, hi & 0x00ff00ffu), O2 = __builtin_bit_cast(us2, (hi >> 8) & 0x00ff00ffu);
            const us2 P24 = { E.y, E2.x }, P35 = { O.y, O2.x };
            const us2 c14 = { 14, 14 }, c63 = { 63, 63 }, c103 = { 103, 103 };
            o02[r] = __builtin_bit_cast(uint32_t, c14 * E + c63 * O + c103 * P24 + c63 * P35 + c14 * E2);      // (out0, out2)
            o13[r] = __builtin_bit_cast(uint32_t, c14 * O + c63 * P24 + c103 * P35 + c63 * E2 + c14 * O2);     // (out1, out3)
        }
        // (row 2m, row 2m + 1) of each column: low halves / high halves of the two rows' words
        const uint4 q = make_uint4(__builtin_amdgcn_perm(o02[1], o02[0], 0x05040100u), __builtin_amdgcn_perm(o13[1], o13[0], 0x05040100u),
                                   __builtin_amdgcn_perm(o02[1], o02[0], 0x07060302u), __builtin_amdgcn_perm(o13[1], o13[0], 0x07060302u));
        *reinterpret_cast<uint4*>(rowp + m * RW + 4 * g) = q;
    }
    __syncthreads();
    // vertical 5-tap, (acc + 2^15) >> 16, saturate -> blurred u8.  A lane walks DOWN a group of four columns, one row pair
    // at a time: the output rows 2m and 2m + 1 both live on the pairs m, m + 1, m + 2, three v_dot2_u32_u16 per pixel with
    // the rounding constant as the first one's accumulator (the kernel is bound by vector instructions,
    // profiles/r03_kernel_counters.json: this pass was 15 of them per pixel with one multiply-add per tap and unpacking).
    {
        constexpr int VPAIRS = 3, VSEG = (BH / 2 + VPAIRS - 1) / VPAIRS;       // 11 segments of 3 row pairs: 187 lanes = three waves
        static_assert(VSEG * RG <= 256, "one lane per (column group, segment)");
        const int g = tid % RG, seg = tid / RG;
        if (seg < VSEG) {
            const int m0 = seg * VPAIRS;
            uint4 P[3];
            auto fetch = [&](int m) { return *reinterpret_cast<const uint4*>(rowp + (m < GH / 2 ? m : GH / 2 - 1) * RW + 4 * g); };
            P[0] = fetch(m0); P[1] = fetch(m0 + 1);
#pragma unroll
            for (int i = 0; i < VPAIRS; ++i) {
                const int m = m0 + i;
                P[(i + 2) % 3] = fetch(m + 2);
                if (2 * m < BH) {
                    const uint4 A = P[i % 3], B = P[(i + 1) % 3], C = P[(i + 2) % 3];
                    const uint32_t a[4] = { A.x, A.y, A.z, A.w }, bb[4] = { B.x, B.y, B.z, B.w }, cc[4] = { C.x, C.y, C.z, C.w };
                    uint32_t even = 0, odd = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        // row 2m: 14 r[2m] + 63 r[2m+1] + 103 r[2m+2] + 63 r[2m+3] + 14 r[2m+4]; row 2m + 1: the same one row down
                        uint32_t e = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, a[k]), us2{ 14, 63 }, 1u << 15, false);
                        e = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, bb[k]), us2{ 103, 63 }, e, false);
                        e = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, cc[k]), us2{ 14, 0 }, e, false);
                        uint32_t o = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, a[k]), us2{ 0, 14 }, 1u << 15, false);
                        o = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, bb[k]), us2{ 63, 103 }, o, false);
                        o = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, cc[k]), us2{ 63, 14 }, o, false);
                        e >>= 16; o >>= 16;                                     // <= 257: saturate_cast<uchar>
                        even |= (e > 255u ? 255u : e) << (8 * k);
                        odd |= (o > 255u ? 255u : o) << (8 * k);
                    }
                    *reinterpret_cast<uint32_t*>(blur + (2 * m) * BW_ + 4 * g) = even;
                    *reinterpret_cast<uint32_t*>(blur + (2 * m + 1) * BW_ + 4 * g) = odd;
                }
            }
        }
    }
    __syncthreads();
    // Sobel 3x3 on the blurred tile: 4 outputs per lane and row from three rows of 6 bytes (two dwords each); a lane walks
    // down LT_H / 16 rows of its column group and unpacks every blurred row once
    {
        constexpr int SROWS = LT_H / 16;
        const int g = tid & 15, seg = tid >> 4;
        const int lx = 4 * g, gx = x0 + lx, ry0 = seg * SROWS;
        typedef short s2 __attribute__((ext_vector_type(2)));
        // packed 16-bit lanes again: per row the column pairs P02 = (c0, c2), P13, P24, P35 of its six bytes; column sums
        // S = r0 + 2 r1 + r2 and differences D = r2 - r0, then vx = S[k+2] - S[k], vy = D[k] + 2 D[k+1] + D[k+2]
        s2 P[3][4];
        auto unpack = [&](int row, s2 (&d)[4]) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(blur + row * BW_ + lx);   // columns lx .. lx+7 <-> x-1 ..
            const uint32_t lo = src[0], hi = src[1];
            const s2 E = __builtin_bit_cast(s2, lo & 0x00ff00ffu), O = __builtin_bit_cast(s2, (lo >> 8) & 0x00ff00ffu);
            const s2 E2 = __builtin_bit_cast(s2, hi & 0x00ff00ffu), O2 = __builtin_bit_cast(s2, (hi >> 8) & 0x00ff00ffu);
            d[0] = E; d[1] = O; d[2] = s2{ E.y, E2.x }; d[3] = s2{ O.y, O2.x };
        };
        unpack(ry0, P[0]);
        unpack(ry0 + 1, P[1]);
        // a + 2 b in both 16-bit halves, one instruction (the compiler emits a shift and an add)
        auto a_plus_2b = [](s2 a, s2 b) {
            s2 d;
            const uint32_t two = 0x00020002u;
            asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(d) : "v"(b), "v"(two), "v"(a));
            return d;
        };
        uint32_t* orow = dxyo + ((size_t)f * Hc + (size_t)(y0 + ry0)) * W + gx;      // one 64-bit address per lane, rows step by W
#pragma unroll
        for (int i = 0; i < SROWS; ++i) {
            const int ry = ry0 + i, gy = y0 + ry;
            unpack(ry + 2, P[(i + 2) % 3]);
            if (gx < W && gy < Hc) {
                s2 S[4], D[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) { S[c] = a_plus_2b(P[i % 3][c], P[(i + 1) % 3][c]) + P[(i + 2) % 3][c]; D[c] = P[(i + 2) % 3][c] - P[i % 3][c]; }
                const s2 vx02 = S[2] - S[0], vx13 = S[3] - S[1];
                const s2 vy02 = a_plus_2b(D[0], D[1]) + D[2], vy13 = a_plus_2b(D[1], D[2]) + D[3];
                uint32_t* op = orow + (size_t)i * W;
                // dx and dy of a pixel share one dword (dx low, dy high): the descriptor kernel fetches both with one gather
                const uint32_t x02 = __builtin_bit_cast(uint32_t, vx02), x13 = __builtin_bit_cast(uint32_t, vx13);
                const uint32_t y02 = __builtin_bit_cast(uint32_t, vy02), y13 = __builtin_bit_cast(uint32_t, vy13);
                const uint32_t w[4] = { __builtin_amdgcn_perm(y02, x02, 0x05040100u), __builtin_amdgcn_perm(y13, x13, 0x05040100u),
                                        __builtin_amdgcn_perm(y02, x02, 0x07060302u), __builtin_amdgcn_perm(y13, x13, 0x07060302u) };
                if (gx + 3 < W && (W & 3) == 0) {
                    *reinterpret_cast<uint4*>(op) = make_uint4(w[0], w[1], w[2], w[3]);
                } else {
                    for (int k = 0; k < 4 && gx + k < W; ++k) op[k] = w[k];
                }
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
// ANYW = true: BinaryDescriptor::setWidthOfBand (binary_descriptor_custom.cpp:134-176) with a width other than the default 7: the support
// region is 9 w rows (w <= LBD_MAXW), a lane takes rows lane, lane + 64, ...; the tables (9 w global, 3 w local weights) come from the host
// for that width.  The same statements otherwise -- a completeness path, the default width keeps its compile-time shape.
constexpr int LBD_MAXW = 21;
template <bool KL, bool ANYW = false>
__global__ __launch_bounds__(256) void k_lbd(int Hc_, int W_, const int* __restrict__ n_seg_ptr,
                                             const float* __restrict__ lines, const int* __restrict__ seg_frame,
                                             const uint32_t* __restrict__ dxyi,
                                             const float* __restrict__ gauss_g /*63*/, const float* __restrict__ gauss_l /*21*/,
                                             float* __restrict__ desc, uint8_t* __restrict__ code,
                                             LbdPlanes planes, const float* __restrict__ kl_angle, const int* __restrict__ kl_npx,
                                             const int* __restrict__ kl_octave, int n_cap, int n_frames, int wband_arg = WBAND)
{
    const int WBAND = ANYW ? wband_arg : lf::WBAND;          // (shadow the file's constants: runtime values in the ANYW copy)
    const int LSP_H = ANYW ? NBANDS * wband_arg : lf::LSP_H;
    __shared__ float rows[4][ANYW ? NBANDS * LBD_MAXW : lf::LSP_H][4];      // per wave: row sums pgdL, ngdL, pgdO, ngdO (already * coefG)
    __shared__ float dsc[4][72];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // never past the capacity the buffers were sized for: a count beyond it is the caller's LF_ERR_CAPACITY case
    const int n_seg = min(*n_seg_ptr, n_cap);
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
        // caller-supplied KeyLines (lf_describe_keylines with device arrays) are not validated on the host: a line that
        // names a plane that was not built gets a zero descriptor instead of a wild read
        if (oc < 0 || oc >= LF_MAX_OCTAVES || planes.base[oc] == nullptr || f < 0 || f >= n_frames) {
            if (desc) for (int i = lane; i < 72; i += 64) desc[(size_t)seg * 72 + i] = 0.f;
            if (code && lane < 32) code[(size_t)seg * 32 + lane] = 0;
            continue;
        }
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
    int row_done = 0;                                           // rows the running origin has been advanced by
    for (int row = lane; row < LSP_H; row += 64) {
        // the reference advances the row origin by repeated float updates: replay them
        for (; row_done < row; ++row_done) { sCorX0 -= dL1; sCorY0 += dL0; }
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
        const float cg = gauss_g[row];
        rows[wave][row][0] = cg * pgdL;
        rows[wave][row][1] = cg * ngdL;
        rows[wave][row][2] = cg * pgdO;
        rows[wave][row][3] = cg * ngdO;
        if (!ANYW) break;                                       // (63 rows: one per lane)
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
        for (int i = 0; i < (ANYW ? 3 * LBD_MAXW : 3 * lf::WBAND); ++i) {
            if (ANYW && i >= 3 * WBAND) break;
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
                float* desc, uint8_t* code, hipStream_t s, int wband)
{
    if (n_seg_cap <= 0) return;
    int blocks = (n_seg_cap + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    LbdPlanes none;
    for (int i = 0; i < LF_MAX_OCTAVES; ++i) { none.base[i] = nullptr; none.W[i] = 0; none.H[i] = 0; }
    if (wband == WBAND)
        hipLaunchKernelGGL((k_lbd<false, false>), dim3(blocks), dim3(256), 0, s, Hc, W, n_seg, lines, seg_frame, dxy,
                           gauss_g, gauss_l, desc, code, none, nullptr, nullptr, nullptr, n_seg_cap, 0, WBAND);
    else
        hipLaunchKernelGGL((k_lbd<false, true>), dim3(blocks), dim3(256), 0, s, Hc, W, n_seg, lines, seg_frame, dxy,
                           gauss_g, gauss_l, desc, code, none, nullptr, nullptr, nullptr, n_seg_cap, 0, wband);
}
int lbd_max_width_of_band() { return LBD_MAXW; }

void launch_lbd_keylines(const LbdPlanes& planes, int n_cap, int n_frames, const int* n_lines, const float* in_octave4, const float* angle, const int* npx,
                         const int* octave, const int* frame, const float* gauss_g, const float* gauss_l, float* desc, uint8_t* code,
                         hipStream_t s, int wband)
{
    if (n_cap <= 0) return;
    int blocks = (n_cap + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    if (wband == WBAND)
        hipLaunchKernelGGL((k_lbd<true, false>), dim3(blocks), dim3(256), 0, s, 0, 0, n_lines, in_octave4, frame, nullptr, gauss_g, gauss_l, desc, code,
                           planes, angle, npx, octave, n_cap, n_frames, WBAND);
    else
        hipLaunchKernelGGL((k_lbd<true, true>), dim3(blocks), dim3(256), 0, s, 0, 0, n_lines, in_octave4, frame, nullptr, gauss_g, gauss_l, desc, code,
                           planes, angle, npx, octave, n_cap, n_frames, wband);
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
