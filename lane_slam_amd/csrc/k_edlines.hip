// EDLines detector + multi-octave KeyLines: the reference's BinaryDescriptor::detect path,
//   /root/reference/src/line_descriptor/src/binary_descriptor_custom.cpp
//     :689-1024  OctaveKeyLines (Gaussian blur per octave, EDLines, resize by 1/sqrt 2, grouping of the octaves' lines)
//     :1442-2240 EdgeDrawing    :2242-2482 EDline    :2484-2643 LeastSquaresLineFit_    :2645-2726 LineValidation_
//     :455-513   detectImpl (KeyLine fields and order)
// re-designed for the MI355X; oracle/lf_oracle_edlines.c is the sequential statement these kernels are held to, bit
// for bit (tests/test_gpu_edlines.py).
//
//   k_ed_grad     streaming, one 64x64 tile per workgroup, built like k_lbd_grad (four pixels per lane, packed 16-bit
//                 filters, v_dot2 column filter) with the octave's run-time taps: 5x5 fixed-point Gaussian -> blurred u8
//                 plane (the next octave is resized from it), Sobel 3x3 -> dx | dy << 16 per pixel (the plane k_lbd
//                 gathers from: the descriptor of a detected line uses the DETECTOR's gradients, :1079-1090), and the
//                 thresholded gradient / 4 (round half to even) with the pixel's direction in bit 15 (u16 plane).
//                 HBM bound: reads P, writes P + 4P + 2P.
//   k_ed_resize   cv::resize INTER_LINEAR by 1/sqrt 2 on u8 (11-bit fixed-point weights from host-made tables), four
//                 pixels x four rows per thread.
//   k_ed_detect   one 512-thread workgroup per (frame, octave), three phases:
//                 ANCHORS by all waves: candidates tested row-major (coalesced, eight in flight per thread) into a bit
//                 plane in LDS indexed COLUMN-major -- the order the reference scans them in, which decides which anchor
//                 draws which edge -- then a ballot / prefix pass lists them.
//                 SMART ROUTING on ONE wave (sequential by definition: a walk stops at pixels earlier walks marked).  A
//                 lone wave pays ~13 cycles per dependent scalar instruction and 25 - 45 per taken branch (tools/probe/
//                 lone_wave_issue.hip), so the walk is organised to execute few of them: the wave holds an 8x8 window of
//                 the gradient plane, every lane precomputes the step a walk standing on its pixel would take, and the walk
//                 itself is a loop over lane numbers (ed_walk below); edge marks are a bit plane in LDS (global memory for
//                 octave images beyond the LDS budget), the chain is assembled in place (first part reversed by all 64
//                 lanes, second part written behind it as it is walked).
//                 LINE FITTING on ALL waves: the chains are independent, every wave takes the next one from a shared
//                 counter; the 64 lanes take the pixels of a run at once (distance tests by ballot and bit tricks, exact
//                 integer sums for the normal equations, ordered f64 sums where the reference's order matters), one
//                 lane-uniform evaluation of the NFA; lines go to temporary records and are put in chain order at the end,
//                 where the reference's running "too many lines" test is applied to the per-chain counts.
//   k_kl_assemble one workgroup per frame: groups the octaves' lines (each line of octave o against all lines of the
//                 octaves below, lanes = lines), orders KeyLines by (class, octave) and fills the output rows.
#include "common.h"
#include "k_edlines_types.h"

namespace lf {

constexpr int ET_W = 64, ET_H = 64;                 // k_ed_grad tile
constexpr int kHorizontal = 0x8000;                 // bit 15 of the g plane: |dx| < |dy|

__device__ __forceinline__ int ed_reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * n - 2 - p; }
    return p;
}

__device__ __forceinline__ int ed_div4_half_even(int v)
{
    const int q = v >> 2, r = v & 3;
    return r < 2 ? q : (r == 3 ? q + 1 : q + (q & 1));
}

// src u8 [B][H][W] -> blur u8, dxy u32, g u16.  taps: 5 ints, sum <= 257 (the row sums then fit 16 bits).
// Built like k_lbd_grad (k_lbd.hip), which computes the same blur + Sobel for the LSD path's descriptor: 64 x 64 tiles,
// four horizontally adjacent pixels per lane in every phase, two values per instruction in packed 16-bit halves, the
// row-filtered tile stored as ROW PAIRS per column -- the operand of v_dot2_u32_u16 in the column filter.  What differs:
// the taps are the octave's (run-time), the image is any size (octave widths are not multiples of four: rows are not
// dword-aligned, so the source is read with unaligned dword loads and the outputs fall back to narrower stores), and
// two more planes are written -- the blurred image (the next octave is resized from it) and the thresholded gradient
// magnitude / 4 (round half to even) with the direction in bit 15.  The blurred ring outside the image is the blur of the
// reflected source, which for a symmetric kernel IS the reflected blurred image the Sobel's BORDER_REFLECT_101 asks for.
// PACKED: the taps sum to <= 257, so a row sum (<= 255 * 257 = 65 535) fits 16 bits.  Rounding can make them sum to 258
// (sigma sqrt 2, the third octave: 29 61 78 61 29) -- those octaves take the copy with 32-bit row sums.
// Round 6: the ANCHORS of the detector (k_ed_detect: a candidate every second pixel in x and y whose gradient exceeds both neighbours
// across its edge direction by anchor_thr) are tested HERE, where the tile's gradients are at hand -- k_ed_detect read the whole plane
// back for it (105 MB per batch, HBM-bound: a tenth of its time).  A candidate's four neighbours lie in the tile or one pixel to its
// right / below: the blurred tile already reaches two columns further than the 64 x 64 outputs need and gets two more rows, the
// gradients of column 64 and row 64 are worked out beside the tile's own (never stored: the neighbouring tile writes them), all 65 x 65
// sit in LDS (where the row sums were), and every candidate column's 32 bits go to the frame's column-major candidate planes
// (aflags: [frame][2][n_cwords] -- candidate set, its pixel a horizontal-edge one) with two atomics each.  anchor planes: scan == 2 only.
struct EdAnchorOut { uint32_t* flags; int thr, nW, nH, n_cwords; };

template <bool PACKED>
__global__ __launch_bounds__(256) void k_ed_grad(int H, int W, const uint8_t* __restrict__ src, int t0, int t1, int t2, int t3, int t4,
                                                 int grad_threshold, uint8_t* __restrict__ bluro, uint32_t* __restrict__ dxyo,
                                                 uint16_t* __restrict__ go, EdAnchorOut an)
{
    constexpr int GW = 72, GH = ET_H + 8;            // source tile: 70 columns used (x0-3 .. x0+66), rows y0-3 .. y0+68, padded to dwords / pairs
    constexpr int RW = 68, RG = RW / 4;              // row-filtered: columns x0-1 .. x0+66
    constexpr int BW_ = 72, BH = ET_H + 4;           // blurred: column c <-> x0-1+c (68 computed), row r <-> y0-1+r (67 used: the gradient of row y0+64)
    constexpr int GLW = 68;                          // gradients of the tile and of column / row 64, 16 bits each (in rowp's place; rows 8-byte aligned)
    static_assert(65 * GLW * 2 <= (int)sizeof(uint32_t) * (PACKED ? (GH / 2) * RW : GH * RW), "the gradient square fits where the row sums were");
    __shared__ __attribute__((aligned(16))) uint8_t gray[GH * GW];
    __shared__ __attribute__((aligned(16))) uint32_t rowp[PACKED ? (GH / 2) * RW : GH * RW];   // PACKED: (row 2m, row 2m + 1) per column, 16 bits each; else a row sum per word
    __shared__ __attribute__((aligned(16))) uint8_t blur[BH * BW_];
    int tbx, tby, f;
    lf_xcd_tile(tbx, tby, f);
    const int x0 = tbx * ET_W, y0 = tby * ET_H;
    const int tid = threadIdx.x;
    const size_t fo = (size_t)f * H * W;
    const uint8_t* img = src + fo;
    auto reflect_row = [&](int ty) { return (uint32_t)ed_reflect101(y0 + ty - 3, H) * (uint32_t)W; };
    const bool interior = x0 >= 3 && x0 + 69 <= W;           // wave-uniform: all 72 columns x0-3 .. x0+68 lie inside the row
    if (interior) {
        const uint8_t* base = img + (x0 - 3);
        for (int idx = tid; idx < GH * (GW / 4); idx += 256) {
            const int ty = idx / (GW / 4), g = idx - ty * (GW / 4);
            uint32_t packed;
            __builtin_memcpy(&packed, base + (reflect_row(ty) + 4u * (uint32_t)g), 4);      // (not dword-aligned: W is any number)
            *reinterpret_cast<uint32_t*>(gray + ty * GW + 4 * g) = packed;
        }
    } else {
        for (int idx = tid; idx < GH * (GW / 4); idx += 256) {
            const int ty = idx / (GW / 4), g = idx - ty * (GW / 4);
            const uint32_t row = reflect_row(ty);
            const int xa = x0 + 4 * g - 3;
            uint32_t packed = 0;
            if (xa >= 0 && xa + 3 < W) __builtin_memcpy(&packed, img + (row + (uint32_t)xa), 4);
            else {
#pragma unroll
                for (int k = 0; k < 4; ++k) packed |= (uint32_t)img[row + (uint32_t)ed_reflect101(xa + k, W)] << (8 * k);
            }
            *reinterpret_cast<uint32_t*>(gray + ty * GW + 4 * g) = packed;
        }
    }
    __syncthreads();
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    if constexpr (PACKED) {
        // horizontal 5-tap: outputs c..c+3 (c = 4g) need gray[c .. c+7] = two dwords; the same four columns of two consecutive rows
        const us2 c0 = { (unsigned short)t0, (unsigned short)t0 }, c1 = { (unsigned short)t1, (unsigned short)t1 }, c2 = { (unsigned short)t2, (unsigned short)t2 },
                  c3 = { (unsigned short)t3, (unsigned short)t3 }, c4 = { (unsigned short)t4, (unsigned short)t4 };
        for (int idx = tid; idx < (GH / 2) * RG; idx += 256) {
            const int m = idx / RG, g = idx - m * RG;
            uint32_t o02[2], o13[2];
    #pragma unroll
            for (int r = 0; r < 2; ++r) {
                const uint32_t* sp = reinterpret_cast<const uint32_t*>(gray + (2 * m + r) * GW + 4 * g);
                const uint32_t lo = sp[0], hi = sp[1];
                const us2 E = __builtin_bit_cast(us2, lo & 0x00ff00ffu), O = __builtin_bit_cast(us2, (lo >> 8) & 0x00ff00ffu);
                const us2 E2 = __builtin_bit_cast(us2, hi & 0x00ff00ffu), O2 = __builtin_bit_cast(us2, (hi >> 8) & 0x00ff00ffu);
                const us2 P24 = { E.y, E2.x }, P35 = { O.y, O2.x };
                o02[r] = __builtin_bit_cast(uint32_t, c0 * E + c1 * O + c2 * P24 + c3 * P35 + c4 * E2);      // (out0, out2)
                o13[r] = __builtin_bit_cast(uint32_t, c0 * O + c1 * P24 + c2 * P35 + c3 * E2 + c4 * O2);     // (out1, out3)
            }
            const uint4 q = make_uint4(__builtin_amdgcn_perm(o02[1], o02[0], 0x05040100u), __builtin_amdgcn_perm(o13[1], o13[0], 0x05040100u),
                                       __builtin_amdgcn_perm(o02[1], o02[0], 0x07060302u), __builtin_amdgcn_perm(o13[1], o13[0], 0x07060302u));
            *reinterpret_cast<uint4*>(rowp + m * RW + 4 * g) = q;
        }
    } else {
        // the same with one 32-bit sum per pixel
        const uint32_t t[5] = { (uint32_t)t0, (uint32_t)t1, (uint32_t)t2, (uint32_t)t3, (uint32_t)t4 };
        for (int idx = tid; idx < GH * RG; idx += 256) {
            const int r = idx / RG, g = idx - r * RG;
            const uint32_t* sp = reinterpret_cast<const uint32_t*>(gray + r * GW + 4 * g);
            const uint32_t lo = sp[0], hi = sp[1];
            uint32_t bt[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) { bt[k] = (lo >> (8 * k)) & 0xffu; bt[4 + k] = (hi >> (8 * k)) & 0xffu; }
            uint32_t o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = t[0] * bt[k] + t[1] * bt[k + 1] + t[2] * bt[k + 2] + t[3] * bt[k + 3] + t[4] * bt[k + 4];
            *reinterpret_cast<uint4*>(rowp + r * RW + 4 * g) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
    __syncthreads();
    if constexpr (PACKED) {
        // vertical 5-tap, (acc + 2^15) >> 16, saturate -> blurred u8: rows 2m and 2m + 1 from the pairs m, m + 1, m + 2
        {
            constexpr int VPAIRS = 3, VSEG = (BH / 2 + VPAIRS - 1) / VPAIRS;
            static_assert(VSEG * RG <= 256, "one lane per (column group, segment)");
            const us2 k01 = { (unsigned short)t0, (unsigned short)t1 }, k23 = { (unsigned short)t2, (unsigned short)t3 }, k4_ = { (unsigned short)t4, 0 };
            const us2 k_0 = { 0, (unsigned short)t0 }, k12 = { (unsigned short)t1, (unsigned short)t2 }, k34 = { (unsigned short)t3, (unsigned short)t4 };
            const int g = tid % RG, seg = tid / RG;
            if (seg < VSEG) {
                const int m0 = seg * VPAIRS;
                uint4 Pp[3];
                auto fetch = [&](int m) { return *reinterpret_cast<const uint4*>(rowp + (m < GH / 2 ? m : GH / 2 - 1) * RW + 4 * g); };
                Pp[0] = fetch(m0); Pp[1] = fetch(m0 + 1);
    #pragma unroll
                for (int i = 0; i < VPAIRS; ++i) {
                    const int m = m0 + i;
                    Pp[(i + 2) % 3] = fetch(m + 2);
                    if (2 * m < BH) {
                        const uint4 A = Pp[i % 3], B = Pp[(i + 1) % 3], C = Pp[(i + 2) % 3];
                        const uint32_t a[4] = { A.x, A.y, A.z, A.w }, bb[4] = { B.x, B.y, B.z, B.w }, cc[4] = { C.x, C.y, C.z, C.w };
                        uint32_t even = 0, odd = 0;
    #pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            uint32_t e = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, a[k]), k01, 1u << 15, false);
                            e = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, bb[k]), k23, e, false);
                            e = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, cc[k]), k4_, e, false);
                            uint32_t o = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, a[k]), k_0, 1u << 15, false);
                            o = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, bb[k]), k12, o, false);
                            o = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, cc[k]), k34, o, false);
                            e >>= 16; o >>= 16;
                            even |= (e > 255u ? 255u : e) << (8 * k);
                            odd |= (o > 255u ? 255u : o) << (8 * k);
                        }
                        *reinterpret_cast<uint32_t*>(blur + (2 * m) * BW_ + 4 * g) = even;
                        *reinterpret_cast<uint32_t*>(blur + (2 * m + 1) * BW_ + 4 * g) = odd;
                    }
                }
            }
        }
    } else {
        const uint32_t t[5] = { (uint32_t)t0, (uint32_t)t1, (uint32_t)t2, (uint32_t)t3, (uint32_t)t4 };
        for (int idx = tid; idx < BH * RG; idx += 256) {
            const int r = idx / RG, g = idx - r * RG;
            uint32_t acc[4] = { 1u << 15, 1u << 15, 1u << 15, 1u << 15 };
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const uint4 q = *reinterpret_cast<const uint4*>(rowp + (r + j) * RW + 4 * g);
                acc[0] += t[j] * q.x; acc[1] += t[j] * q.y; acc[2] += t[j] * q.z; acc[3] += t[j] * q.w;
            }
            uint32_t out = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { const uint32_t e = acc[k] >> 16; out |= (e > 255u ? 255u : e) << (8 * k); }
            *reinterpret_cast<uint32_t*>(blur + r * BW_ + 4 * g) = out;
        }
    }
    __syncthreads();
    // Sobel 3x3 on the blurred tile, 4 outputs per lane and row; the three planes
    uint16_t* gl = reinterpret_cast<uint16_t*>(rowp);        // (the row sums are done with)
    {
        constexpr int SROWS = ET_H / 16;
        const int g = tid & 15, seg = tid >> 4;
        const int lx = 4 * g, gx = x0 + lx, ry0 = seg * SROWS;
        typedef short s2 __attribute__((ext_vector_type(2)));
        s2 Pq[3][4];
        uint32_t mid[3];                                    // the row's own four blurred bytes (columns lx+1 .. lx+4)
        auto unpack = [&](int row, s2 (&d)[4], uint32_t& centre) {
            const uint32_t* sp = reinterpret_cast<const uint32_t*>(blur + row * BW_ + lx);   // columns lx .. lx+7 <-> x-1 ..
            const uint32_t lo = sp[0], hi = sp[1];
            const s2 E = __builtin_bit_cast(s2, lo & 0x00ff00ffu), O = __builtin_bit_cast(s2, (lo >> 8) & 0x00ff00ffu);
            const s2 E2 = __builtin_bit_cast(s2, hi & 0x00ff00ffu), O2 = __builtin_bit_cast(s2, (hi >> 8) & 0x00ff00ffu);
            d[0] = E; d[1] = O; d[2] = s2{ E.y, E2.x }; d[3] = s2{ O.y, O2.x };
            centre = (lo >> 8) | (hi << 24);
        };
        unpack(ry0, Pq[0], mid[0]);
        unpack(ry0 + 1, Pq[1], mid[1]);
        auto a_plus_2b = [](s2 a, s2 b) {
            s2 d;
            const uint32_t two = 0x00020002u;
            asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(d) : "v"(b), "v"(two), "v"(a));
            return d;
        };
        const bool wide = (W & 3) == 0;                      // rows of the output planes are 4-pixel aligned
        const size_t o00 = fo + (size_t)(y0 + ry0) * W + gx;
#pragma unroll
        for (int i = 0; i < SROWS; ++i) {
            const int ry = ry0 + i, gy = y0 + ry;
            unpack(ry + 2, Pq[(i + 2) % 3], mid[(i + 2) % 3]);
            if (gx < W && gy < H) {
                s2 S[4], D[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) { S[c] = a_plus_2b(Pq[i % 3][c], Pq[(i + 1) % 3][c]) + Pq[(i + 2) % 3][c]; D[c] = Pq[(i + 2) % 3][c] - Pq[i % 3][c]; }
                const s2 vx02 = S[2] - S[0], vx13 = S[3] - S[1];
                const s2 vy02 = a_plus_2b(D[0], D[1]) + D[2], vy13 = a_plus_2b(D[1], D[2]) + D[3];
                const int vx[4] = { vx02.x, vx13.x, vx02.y, vx13.y }, vy[4] = { vy02.x, vy13.x, vy02.y, vy13.y };
                uint32_t w[4], gq[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    w[k] = ((uint32_t)vx[k] & 0xffffu) | ((uint32_t)vy[k] << 16);
                    const int ax = vx[k] < 0 ? -vx[k] : vx[k], ay = vy[k] < 0 ? -vy[k] : vy[k];
                    const int sum = ax + ay;
                    gq[k] = (uint32_t)(ed_div4_half_even(sum > grad_threshold + 1 ? sum : 0) | (ax < ay ? kHorizontal : 0));
                }
                if (an.flags) *reinterpret_cast<uint2*>(gl + ry * GLW + lx) = make_uint2(gq[0] | (gq[1] << 16), gq[2] | (gq[3] << 16));
                const size_t o = o00 + (size_t)i * W;
                const uint32_t bl = mid[(i + 1) % 3];
                if (gx + 3 < W) {
                    if (wide) {
                        *reinterpret_cast<uint4*>(dxyo + o) = make_uint4(w[0], w[1], w[2], w[3]);
                        *reinterpret_cast<uint2*>(go + o) = make_uint2(gq[0] | (gq[1] << 16), gq[2] | (gq[3] << 16));
                        *reinterpret_cast<uint32_t*>(bluro + o) = bl;
                    } else {                                   // the same stores at addresses that are not multiples of their size
                        const uint2 g2 = make_uint2(gq[0] | (gq[1] << 16), gq[2] | (gq[3] << 16));
                        __builtin_memcpy(dxyo + o, w, 16);
                        __builtin_memcpy(go + o, &g2, 8);
                        __builtin_memcpy(bluro + o, &bl, 4);
                    }
                } else {
                    for (int k = 0; k < 4 && gx + k < W; ++k) { dxyo[o + k] = w[k]; go[o + k] = (uint16_t)gq[k]; bluro[o + k] = (uint8_t)(bl >> (8 * k)); }
                }
            }
        }
    }
    if (!an.flags) return;
    // ---- the anchors of this tile's candidates
    {
        // column 64 (rows 0 .. 64) and row 64 (columns 0 .. 63): the same Sobel and quantisation, one pixel per thread
        const int t = tid;
        if (t < 129) {
            const int cx = t < 65 ? 64 : t - 65, cy = t < 65 ? t : 64;
            if (x0 + cx < W && y0 + cy < H) {
                const uint8_t* b = blur + (cy + 1) * BW_ + cx + 1;
                const int a00 = b[-BW_ - 1], a01 = b[-BW_], a02 = b[-BW_ + 1], a10 = b[-1], a12 = b[1], a20 = b[BW_ - 1], a21 = b[BW_], a22 = b[BW_ + 1];
                const int vx = (a02 + 2 * a12 + a22) - (a00 + 2 * a10 + a20), vy = (a20 + 2 * a21 + a22) - (a00 + 2 * a01 + a02);
                const int ax = vx < 0 ? -vx : vx, ay = vy < 0 ? -vy : vy, sum = ax + ay;
                gl[cy * GLW + cx] = (uint16_t)(ed_div4_half_even(sum > grad_threshold + 1 ? sum : 0) | (ax < ay ? kHorizontal : 0));
            }
        }
        uint32_t* s_col = reinterpret_cast<uint32_t*>(gray);      // [2][32]: a candidate column's 32 rows (the source tile is done with)
        if (t < 64) s_col[t] = 0u;
        __syncthreads();
        // thread = (candidate column, four of its 32 rows)
        const int cwl = t & 31, rg = t >> 5;
        const int cw = (x0 >> 1) + cwl, lx = 1 + 2 * cwl;
        uint32_t bits = 0u, hbits = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int chl = rg * 4 + j, ch = (y0 >> 1) + chl, ly = 1 + 2 * chl;
            if (cw < an.nW && ch < an.nH) {
                const uint32_t v = gl[ly * GLW + lx];
                const bool hz = (v & kHorizontal) != 0;
                const int gv = (int)(v & 0x7fffu);
                const int n1 = (int)(gl[hz ? (ly - 1) * GLW + lx : ly * GLW + lx - 1] & 0x7fffu), n2 = (int)(gl[hz ? (ly + 1) * GLW + lx : ly * GLW + lx + 1] & 0x7fffu);
                if (gv >= n1 + an.thr && gv >= n2 + an.thr) { bits |= 1u << chl; if (hz) hbits |= 1u << chl; }
            }
        }
        if (bits) atomicOr(&s_col[cwl], bits);
        if (hbits) atomicOr(&s_col[32 + cwl], hbits);
        __syncthreads();
        // a column's bits: plane positions cw * nH + ch0 .. + 31, two words
        if (t < 64) {
            const int c = t & 31, pl = t >> 5;
            const uint32_t v = s_col[t];
            const int cwc = (x0 >> 1) + c;
            if (v && cwc < an.nW) {
                const unsigned b0 = (unsigned)cwc * (unsigned)an.nH + (unsigned)(y0 >> 1);
                uint32_t* dst = an.flags + ((size_t)f * 2 + pl) * an.n_cwords;
                const unsigned w0 = b0 >> 5, sh = b0 & 31u;
                atomicOr(dst + w0, v << sh);
                if (sh && (v >> (32u - sh))) atomicOr(dst + w0 + 1, v >> (32u - sh));
            }
        }
    }
}

void launch_ed_grad(int H, int W, int n_frames, const uint8_t* src, const int* taps5, int grad_threshold, uint8_t* blur,
                    uint32_t* dxy, uint16_t* g, hipStream_t s, uint32_t* anchor_flags, int anchor_thr)
{
    dim3 grid((W + ET_W - 1) / ET_W, (H + ET_H - 1) / ET_H, n_frames);
    const int sum = taps5[0] + taps5[1] + taps5[2] + taps5[3] + taps5[4];
    EdAnchorOut an;
    an.flags = anchor_flags; an.thr = anchor_thr;
    an.nW = W > 2 ? (W - 2 + 1) / 2 : 0; an.nH = H > 2 ? (H - 2 + 1) / 2 : 0;      // scan interval 2 (ed_anchor_words)
    an.n_cwords = (an.nW * an.nH + 31) / 32;
    if (sum <= 257)
        hipLaunchKernelGGL(k_ed_grad<true>, grid, dim3(256), 0, s, H, W, src, taps5[0], taps5[1], taps5[2], taps5[3], taps5[4], grad_threshold, blur, dxy, g, an);
    else
        hipLaunchKernelGGL(k_ed_grad<false>, grid, dim3(256), 0, s, H, W, src, taps5[0], taps5[1], taps5[2], taps5[3], taps5[4], grad_threshold, blur, dxy, g, an);
}

// words of one candidate plane of a W x H octave image at scan interval 2 (k_ed_grad writes two per frame: set, horizontal-edge)
size_t ed_anchor_words(int W, int H)
{
    const int nW = W > 2 ? (W - 2 + 1) / 2 : 0, nH = H > 2 ? (H - 2 + 1) / 2 : 0;
    return ((size_t)nW * nH + 31) / 32;
}

// cv::GaussianBlur(src, dst, Size(k, k), sigma) on u8 for a kernel size OTHER than the reference's default 5 (Params::ksize_,
// binary_descriptor_custom.cpp:708): the same 8-bit fixed-point separable filter (row sums of taps x pixels, column sums of taps x row
// sums, (s + 2^15) >> 16, BORDER_REFLECT_101) with run-time taps, as two plain passes through an int32 plane.  A completeness path:
// k_ed_grad then runs with the identity taps (0 0 256 0 0), which hand the blurred pixels through unchanged.
struct EdTaps { int n; int k[31]; };
__global__ __launch_bounds__(256) void k_ed_blur_rows(int H, int W, const uint8_t* __restrict__ src, EdTaps t, int* __restrict__ tmp)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const uint8_t* row = src + ((size_t)blockIdx.z * H + y) * W;
    const int r = t.n / 2;
    int sum = 0;
    for (int j = -r; j <= r; ++j) sum += t.k[j + r] * (int)row[ed_reflect101(x + j, W)];
    tmp[((size_t)blockIdx.z * H + y) * W + x] = sum;
}
__global__ __launch_bounds__(256) void k_ed_blur_cols(int H, int W, const int* __restrict__ tmp, EdTaps t, uint8_t* __restrict__ dst)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const int* img = tmp + (size_t)blockIdx.z * H * W;
    const int r = t.n / 2;
    int sum = 0;
    for (int j = -r; j <= r; ++j) sum += t.k[j + r] * img[(size_t)ed_reflect101(y + j, H) * W + x];
    const int v = (sum + (1 << 15)) >> 16;
    dst[((size_t)blockIdx.z * H + y) * W + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
}
void launch_ed_blur_any(int H, int W, int n_frames, const uint8_t* src, const int* taps, int ksize, int* tmp, uint8_t* dst, hipStream_t s)
{
    EdTaps t;
    t.n = ksize;
    for (int i = 0; i < 31; ++i) t.k[i] = i < ksize ? taps[i] : 0;
    const dim3 grid((W + 255) / 256, H, n_frames);
    hipLaunchKernelGGL(k_ed_blur_rows, grid, dim3(256), 0, s, H, W, src, t, tmp);
    hipLaunchKernelGGL(k_ed_blur_cols, grid, dim3(256), 0, s, H, W, tmp, t, dst);
}

// cv::resize(src, dst, Size(), inv, inv), INTER_LINEAR, u8: 11-bit coefficients from float weights.  The coefficient
// tables -- per destination column (source column, its right neighbour, the two weights), per destination row (offsets of
// the two source rows, the two weights) -- depend on the sizes alone: the host computes them once per octave
// (ed_resize_tables, the reference's arithmetic) and the kernel is integer work: a thread makes four adjacent pixels of
// RS_ROWS rows and stores each row's four bytes as one dword.
void ed_resize_tables(int H, int W, int DH, int DW, double scale, int* tab)
{
    for (int dx = 0; dx < DW; ++dx) {
        float fx = (float)(((double)dx + 0.5) * scale - 0.5);
        int x = dm::ifloor((double)fx);
        fx -= (float)x;
        if (x < 0) { fx = 0.f; x = 0; }
        if (x >= W - 1) { fx = 0.f; x = W - 1; }
        int* t = tab + 4 * dx;
        t[0] = x; t[1] = x + 1 < W ? x + 1 : x;
        t[2] = dm::round_half_even((double)((1.f - fx) * 2048.f)); t[3] = dm::round_half_even((double)(fx * 2048.f));
    }
    for (int dy = 0; dy < DH; ++dy) {
        float fy = (float)(((double)dy + 0.5) * scale - 0.5);
        const int sy = dm::ifloor((double)fy);
        fy -= (float)sy;
        int y0 = sy, y1 = sy + 1;
        y0 = y0 >= 0 ? (y0 < H ? y0 : H - 1) : 0;
        y1 = y1 >= 0 ? (y1 < H ? y1 : H - 1) : 0;
        int* t = tab + 4 * (DW + dy);
        t[0] = y0 * W; t[1] = y1 * W;
        t[2] = dm::round_half_even((double)((1.f - fy) * 2048.f)); t[3] = dm::round_half_even((double)(fy * 2048.f));
    }
}

constexpr int RS_ROWS = 4, RS_THREADS = 128;
constexpr int RS_SPAN = 736;                     // source columns under RS_THREADS * 4 destination columns (sqrt 2 each) + slack, a multiple of 4
// The source rows of the workgroup's RS_ROWS destination rows (two each, as the row table names them) are staged in LDS
// with coalesced dword loads; the four samples of a pixel are LDS bytes.  (Gathering them from global memory one byte per
// lane made the kernel three times slower than its traffic.)
__global__ __launch_bounds__(RS_THREADS) void k_ed_resize(int H, int W, int DH, int DW, const int4* __restrict__ tab, const uint8_t* __restrict__ src,
                                                          uint8_t* __restrict__ dst)
{
    __shared__ __attribute__((aligned(4))) uint8_t rows[RS_ROWS][2][RS_SPAN];
    const int tid = threadIdx.x, dy0 = blockIdx.y * RS_ROWS;
    const int bx0 = blockIdx.x * RS_THREADS * 4;
    const int bx1 = min(bx0 + RS_THREADS * 4, DW) - 1;          // the workgroup's destination columns
    const int xlo = tab[bx0].x, xhi = tab[bx1].y;               // ... and the source columns under them
    const int span = xhi - xlo + 1;
    const uint8_t* S = src + (size_t)blockIdx.z * H * W;
    const int nrows = DH - dy0 < RS_ROWS ? DH - dy0 : RS_ROWS;
    if (span <= RS_SPAN) {
        for (int r = 0; r < nrows; ++r) {
            const int4 cy = tab[DW + dy0 + r];
            for (int h = 0; h < 2; ++h) {
                const uint8_t* p = S + (h ? cy.y : cy.x) + xlo;
                for (int i = 4 * tid; i < span; i += 4 * RS_THREADS) {
                    uint32_t w = 0;
                    if (i + 3 < span) __builtin_memcpy(&w, p + i, 4);      // (not dword-aligned: any width, any column)
                    else for (int k = 0; i + k < span; ++k) w |= (uint32_t)p[i + k] << (8 * k);
                    *reinterpret_cast<uint32_t*>(&rows[r][h][i]) = w;
                }
            }
        }
    }
    __syncthreads();
    const int dx0 = bx0 + tid * 4;
    if (dx0 >= DW) return;
    int4 cx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) cx[k] = tab[dx0 + k < DW ? dx0 + k : DW - 1];
    uint8_t* D = dst + (size_t)blockIdx.z * DH * DW;
#pragma unroll
    for (int r = 0; r < RS_ROWS; ++r) {
        if (r >= nrows) break;
        const int4 cy = tab[DW + dy0 + r];                     // (wave-uniform: a scalar load)
        uint32_t out = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int a, b_, c_, d;
            if (span <= RS_SPAN) {
                a = rows[r][0][cx[k].x - xlo]; b_ = rows[r][0][cx[k].y - xlo]; c_ = rows[r][1][cx[k].x - xlo]; d = rows[r][1][cx[k].y - xlo];
            } else {                                           // (a scale far from sqrt 2: straight from the image)
                a = S[cy.x + cx[k].x]; b_ = S[cy.x + cx[k].y]; c_ = S[cy.y + cx[k].x]; d = S[cy.y + cx[k].y];
            }
            // 24-bit multiplies (full rate; a 32-bit one is a quarter of that): pixels <= 255, weights <= 2048, S >> 4 <= 32 640
            const int S0 = __mul24(a, cx[k].z) + __mul24(b_, cx[k].w);
            const int S1 = __mul24(c_, cx[k].z) + __mul24(d, cx[k].w);
            int v = ((__mul24(cy.z, S0 >> 4) >> 16) + (__mul24(cy.w, S1 >> 4) >> 16) + 2) >> 2;
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
            out |= (uint32_t)v << (8 * k);
        }
        uint8_t* q = D + (size_t)(dy0 + r) * DW + dx0;
        if (dx0 + 3 < DW) __builtin_memcpy(q, &out, 4);             // (rows of an octave image are not dword-aligned)
        else for (int k = 0; k < 4 && dx0 + k < DW; ++k) q[k] = (uint8_t)(out >> (8 * k));
    }
}

void launch_ed_resize(int H, int W, int DH, int DW, const int* tab, int n_frames, const uint8_t* src, uint8_t* dst, hipStream_t s)
{
    dim3 grid((DW + 4 * RS_THREADS - 1) / (4 * RS_THREADS), (DH + RS_ROWS - 1) / RS_ROWS, n_frames);
    hipLaunchKernelGGL(k_ed_resize, grid, dim3(RS_THREADS), 0, s, H, W, DH, DW, reinterpret_cast<const int4*>(tab), src, dst);
}

// cv::pyrDown u8 -> (H/2, W/2): [1 4 6 4 1]^2, (sum + 128) >> 8, BORDER_REFLECT_101 (compute-only pyramid, :350-371)
__global__ void k_pyrdown(int H, int W, const uint8_t* __restrict__ src, uint8_t* __restrict__ dst)
{
    const int DW = W / 2, DH = H / 2;
    const int dx = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y;
    if (dx >= DW) return;
    const uint8_t* S = src + (size_t)blockIdx.z * H * W;
    const int k[5] = { 1, 4, 6, 4, 1 };
    int s = 0;
#pragma unroll
    for (int j = -2; j <= 2; ++j) {
        const uint8_t* row = S + (size_t)ed_reflect101(2 * dy + j, H) * W;
        int r = 0;
#pragma unroll
        for (int i = -2; i <= 2; ++i) r += k[i + 2] * (int)row[ed_reflect101(2 * dx + i, W)];
        s += k[j + 2] * r;
    }
    dst[((size_t)blockIdx.z * DH + dy) * DW + dx] = (uint8_t)((s + 128) >> 8);
}

void launch_pyrdown(int H, int W, int n_frames, const uint8_t* src, uint8_t* dst, hipStream_t s)
{
    dim3 grid((W / 2 + 255) / 256, H / 2, n_frames);
    if (W / 2 > 0 && H / 2 > 0) hipLaunchKernelGGL(k_pyrdown, grid, dim3(256), 0, s, H, W, src, dst);
}

// ------------------------------------------------------------------------------------------------ k_ed_detect
enum { UpDir = 1, RightDir = 2, DownDir = 3, LeftDir = 4 };

typedef __attribute__((address_space(3))) uint32_t ed_lds_u32;

// the edge marks: a bit plane in LDS (reached through the LDS address space: a flat access to LDS takes the slow path and
// waits on both memory counters), or in global memory for octave images beyond the LDS budget (there every access is an
// agent-scope atomic, so that none is served from a stale line of the vector L1).  A wave's LDS operations and its
// atomics on one address execute in order: a read after a set sees it.
struct EdMarks {
    uint32_t* p; bool in_lds;
    __device__ __forceinline__ bool get(int i) const
    {
        const uint32_t w = in_lds ? ((ed_lds_u32*)p)[i >> 5] : __hip_atomic_load(p + (i >> 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return (w >> (i & 31)) & 1u;
    }
    // per-lane i (the lanes of a wave may name bits of one word)
    __device__ __forceinline__ void set(int i) const
    {
        if (in_lds) __hip_atomic_fetch_or((ed_lds_u32*)p + (i >> 5), 1u << (i & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_fetch_or(p + (i >> 5), 1u << (i & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ __forceinline__ void clear(int i) const
    {
        if (in_lds) __hip_atomic_fetch_and((ed_lds_u32*)p + (i >> 5), ~(1u << (i & 31)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_fetch_and(p + (i >> 5), ~(1u << (i & 31)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};

// v_writelane_b32 (no clang builtin in this toolchain: the LLVM intrinsic by name): old with lane `l` replaced by the uniform v
extern "C" __device__ int ed_writelane(int v, int l, int old) __asm("llvm.amdgcn.writelane.i32");

// -DLF_ED_STAMP=k (diagnostic builds, tools/ed_stamps.py): one quantity per build, summed over the walking wave of a frame,
// returned in the frame's counts[3] (times in units of 4 cycles).  1: inside ed_walk, 2: of that, a window left until the
// walk goes on in the next, 3: a walk's start (window reused or fetched), 4 / 5: windows fetched on leaving one / at a start,
// 6: steps, 7: kernel start until the walk begins (10: until the candidates are tested), 8: the walking phase, 9: from its end to
// the kernel's
#ifdef LF_ED_STAMP
#define ED_T0(k) unsigned long long t_##k = (LF_ED_STAMP == k) ? __builtin_amdgcn_s_memtime() : 0ull
#define ED_T1(k, acc) do { if (LF_ED_STAMP == k) (acc) += __builtin_amdgcn_s_memtime() - t_##k; } while (0)
#define ED_CNT(k, acc) do { if (LF_ED_STAMP == k) (acc) += 4ull; } while (0)
#else
#define ED_T0(k) do { } while (0)
#define ED_T1(k, acc) do { } while (0)
#define ED_CNT(k, acc) do { } while (0)
#endif

struct EdWalk {
    unsigned long long diag;
    const uint16_t* g; EdMarks marks;
    int W, H;
    unsigned lastX, lastY;
    int wx0, wy0;           // the window the last walk ended in (an anchor's second walk starts in its first walk's window
    uint32_t nAB;           //   when that walk never left it: most walks are a few pixels long): per lane the two step records,
    uint64_t stopM, ngM, hzM;   // and per window the lanes a walk stops at (marked or no gradient), those without gradient, the horizontal-edge ones
    bool have;
};

// one walk of the smart routing (:1577-1720 and its three copies), executed uniformly by the wave; pixels go to out[]
// from off on (packed x | y << 16).  Returns false when the part array would overflow.
//
// The walk is a chain of dependent decisions, one pixel per step, and ONE wave per frame executes it: what it costs is
// the number of instructions per step (a lone wave pays 13 cycles per dependent scalar instruction, 5 per vector one, 25 -
// 45 per taken branch: tools/probe/lone_wave_issue.hip) and the round trips to the gradient plane.  Both are taken out
// of the step:
//   * the wave keeps an 8 x 8 WINDOW of the plane, one pixel per lane (lane = row * 8 + column), placed so that the
//     walk runs into it -- one column / row behind the current pixel, six ahead, shifted sideways when the last window
//     was left sideways; an anchor's FIRST walk starts in the middle of its window (three behind, four ahead), so that
//     the second walk, which leaves the anchor the other way, can start in the same window;
//   * the reference's direction logic needs, at a pixel, the walk's direction and whether the last step went right /
//     down (x > lastX, y > lastY).  That is two bits of STATE: the sign (right / down = 1) the walk takes at a
//     horizontal-edge pixel and the sign it takes at a vertical-edge one.  A step from a horizontal-edge pixel with sign
//     s leaves (s, dy > 0), one from a vertical-edge pixel (dx > 0, s);
//   * when a window is fetched, every lane works out FOR ITS PIXEL what a walk standing there would do, for both signs:
//     it compares the three pixels ahead (the neighbours' gradients AND direction bits by two DPP row shifts and two
//     ds_bpermute) and records, per sign, the LANE of the winner and the sign the walk will have THERE (the state after
//     the step is two bits, the winner's direction bit says which of them counts): 11 bits -- bit 0: the three ahead are
//     not all inside the window, bit 1: the reference stops at the image border first, bit 4: the next sign is minus
//     (as the shift that selects the minus record), bits 5 - 10: the next lane.  Both records in one word per lane
//     (plus << 0 | minus << 16); the lanes a walk stops at (marked, or no gradient) and the horizontal-edge lanes are
//     64-bit masks in scalar registers;
//   * inside a window the walk is a walk over NODES (lane, sign): a step tests and sets the lane's bit of the stop mask,
//     notes the lane's place in the visit (v_writelane), reads the lane's word (v_readlane), shifts by the sign's
//     selector and splits the record -- about a dozen scalar instructions (25 with the state and the records decoded per
//     step, round 5), no memory access, no coordinates;
//   * when the walk leaves the window (or ends), the visited lanes store their pixels at their places of the output
//     and set their edge marks in the bit plane, all at once.
// first: an anchor's first walk (its window is centred); else the walk may start in the window c holds.
__device__ __forceinline__ bool ed_walk(EdWalk& c, unsigned x0, unsigned y0, int lastDirection, bool first, uint32_t* __restrict__ out,
                                        unsigned base, unsigned& off_io, unsigned cap)
{
    const int W = c.W, H = c.H;
    const uint16_t* pg = c.g;
    const int lane = (int)(threadIdx.x & 63u), ldx = lane & 7, ldy = lane >> 3;
    int x = __builtin_amdgcn_readfirstlane((int)x0), y = __builtin_amdgcn_readfirstlane((int)y0);
    const int ld = __builtin_amdgcn_readfirstlane(lastDirection);
    // bit 1: the sign at a horizontal-edge pixel, bit 0: at a vertical-edge one.  Plain integers and bit operations: the
    // step stays on the scalar unit
    uint32_t st;
    {
        const uint32_t plus = (ld == RightDir || ld == DownDir) ? 1u : 0u;
        const uint32_t went_right = x > (int)c.lastX ? 1u : 0u, went_down = y > (int)c.lastY ? 1u : 0u;
        st = (ld & 1) == 0 ? (plus << 1) | went_down : (went_right << 1) | plus;
    }
    ED_T0(1); ED_T0(3);
    unsigned off = off_io;
    int wx0 = c.wx0, wy0 = c.wy0;
    uint32_t nAB = c.nAB;
    uint64_t stopM = c.stopM, ngM = c.ngM, hzM = c.hzM;
    int ord = -1, cnt = 0;                                  // this lane's place in the visit of the window; pixels visited in it
    int Lv = -1;                                            // the lane visited last
    bool ok = true;
    const bool inner_x = ldx >= 1 && ldx <= 6, inner_y = ldy >= 1 && ldy <= 6;
    // per lane, all ones when the three pixels ahead lie in the window: plus / minus at a horizontal-edge pixel, at a vertical-edge one
    const uint32_t reachA_h = (ldx <= 6 && inner_y) ? ~0u : 0u, reachB_h = (ldx >= 1 && inner_y) ? ~0u : 0u;
    const uint32_t reachA_v = (ldy <= 6 && inner_x) ? ~0u : 0u, reachB_v = (ldy >= 1 && inner_x) ? ~0u : 0u;
    // where the window goes when the walk leaves this one AT this lane, for a walk that has not drifted sideways since the
    // window was fetched (side 3): the lane's place in the new window is (back, side) along / across the walk -- one column
    // or row behind, six ahead.  As 1 | (ox + 8) << 5 | (oy + 8) << 10 with (ox, oy) the new origin relative to the old
    // one: plus / minus at a horizontal-edge pixel, at a vertical-edge one
    const uint32_t exitA_h = 1u | (uint32_t)(ldx - 1 + 8) << 5 | (uint32_t)(ldy - 3 + 8) << 10, exitB_h = 17u | (uint32_t)(ldx - 6 + 8) << 5 | (uint32_t)(ldy - 3 + 8) << 10;
    const uint32_t exitA_v = 1u | (uint32_t)(ldx - 3 + 8) << 5 | (uint32_t)(ldy - 1 + 8) << 10, exitB_v = 17u | (uint32_t)(ldx - 3 + 8) << 5 | (uint32_t)(ldy - 6 + 8) << 10;
    // the visited pixels of the window -> output and bit plane (the caller has checked that they fit)
    auto retire = [&]() {
        if (ord >= 0) {
            const int px = wx0 + ldx, py = wy0 + ldy;
            out[base + off + (unsigned)ord] = (uint32_t)px | ((uint32_t)py << 16);
            c.marks.set(py * W + px);
        }
        off += (unsigned)cnt;
        ord = -1; cnt = 0;
    };
    // the origin of the first window of a walk standing at (x, y) on a pixel of kind gh (1: horizontal-edge) with sign plus
    auto place = [&](uint32_t gh, uint32_t plus, bool centred, int& nx, int& ny) {
        const int back = centred ? (plus ? 3 : 4) : (plus ? 1 : 6);
        nx = x - (gh ? back : 3);
        ny = y - (gh ? 3 : back);
    };
    // the window at (nx, ny) for a walk standing at (x, y) and heading `go`.  Everything that does not need the loaded
    // values comes between the load and its first use: the old window's pixels retired, the exits of the new one
    auto fetch = [&](int nx, int ny, int go, bool with_retire) {
        const int px = nx + ldx, py = ny + ldy;
        const bool in = px >= 0 && px < W && py >= 0 && py < H;
        const int idx = in ? py * W + px : 0;
        const uint16_t v16 = pg[idx];
        if (with_retire) retire();
        wx0 = nx; wy0 = ny;
        bool mk = c.marks.get(idx);
        // a walk that leaves this window heading the way it entered, two or more pixels to the side of where it entered,
        // gets the next window shifted to that side (side 1 / 6 instead of 3): the one exit record of the four that this
        // concerns, by a uniform factor
        const int dY = py - y, dX = px - x;
        const int sY = (dY >= 2 ? 1 : (dY <= -2 ? 6 : 3)) - 3, sX = (dX >= 2 ? 1 : (dX <= -2 ? 6 : 3)) - 3;
        const uint32_t eA_h = exitA_h - (uint32_t)(sY * (go == RightDir ? 1 << 10 : 0)), eB_h = exitB_h - (uint32_t)(sY * (go == LeftDir ? 1 << 10 : 0));
        const uint32_t eA_v = exitA_v - (uint32_t)(sX * (go == DownDir ? 1 << 5 : 0)), eB_v = exitB_v - (uint32_t)(sX * (go == UpDir ? 1 << 5 : 0));
        __builtin_amdgcn_sched_barrier(0);                     // (left to itself the compiler waits for the load first)
        const uint32_t v = in ? (uint32_t)v16 : 0u;
        mk = in && mk;
        // the eight neighbours' gradients (as unsigned char, :1607-1609) and direction bits: west | own | east of this row
        // in one word (9 bits each), then the same word of the rows above and below.  Values from beyond the window's
        // edge are never used (bit 0 of the record)
        const int g9 = (int)((v & 0xffu) | ((v >> 7) & 0x100u));
        const int gW = __builtin_amdgcn_update_dpp(0, g9, 0x111, 0xf, 0xf, true);     // row_shr:1: from lane - 1
        const int gE = __builtin_amdgcn_update_dpp(0, g9, 0x101, 0xf, 0xf, true);     // row_shl:1: from lane + 1
        const int R = gW | (g9 << 9) | (gE << 18);
        const int RN = __builtin_amdgcn_ds_bpermute(((lane - 8) & 63) << 2, R), RS = __builtin_amdgcn_ds_bpermute(((lane + 8) & 63) << 2, R);
        const int nNW = RN & 0x1ff, nN = (RN >> 9) & 0x1ff, nNE = (RN >> 18) & 0x1ff, nSW = RS & 0x1ff, nS = (RS >> 9) & 0x1ff, nSE = (RS >> 18) & 0x1ff;
        const bool hz = (v & kHorizontal) != 0;
        // pixels 1 / 2 / 3 ahead: right NE E SE, left NW W SW, down SE S SW, up NE N NW; t = +1: pixel 1 wins, -1: pixel 3
        const int A1 = hz ? nNE : nSE, A2 = hz ? gE : nS, A3 = hz ? nSE : nSW;
        const int B1 = hz ? nNW : nNE, B2 = hz ? gW : nN, B3 = hz ? nSW : nNW;
        const int a1 = A1 & 0xff, a2 = A2 & 0xff, a3 = A3 & 0xff, b1 = B1 & 0xff, b2 = B2 & 0xff, b3 = B3 & 0xff;
        const int ta = (a1 >= a2 && a1 >= a3) ? 1 : ((a3 >= a2 && a3 >= a1) ? -1 : 0);
        const int tb = (b1 >= b2 && b1 >= b3) ? 1 : ((b3 >= b2 && b3 >= b1) ? -1 : 0);
        // From here on in integers: conditions combined as lane masks are scalar instructions, 13 cycles each on this wave's
        // one chain.  right / left: (+-1, -t); down / up: (t, +-1); the state after: (sign, dy > 0) resp. (dx > 0, sign); the
        // sign at the winner is the first of the two when its direction bit says horizontal: plus: hz ? h | t < 0 : !h | t > 0,
        // minus: hz ? !h & t < 0 : h & t > 0 -- as bits of two constants indexed by hz << 3 | h << 2 | t + 1
        const uint32_t hzi = (v >> 15) & 1u;
        const uint32_t wA = (uint32_t)(ta > 0 ? A1 : (ta < 0 ? A3 : A2)), wB = (uint32_t)(tb > 0 ? B1 : (tb < 0 ? B3 : B2));   // the winner
        const uint32_t iA = (hzi << 3) | ((wA >> 6) & 4u) | (uint32_t)(ta + 1), iB = (hzi << 3) | ((wB >> 6) & 4u) | (uint32_t)(tb + 1);
        const uint32_t nsA = (~0x7147u >> iA) & 1u, nsB = (~0x0140u >> iB) & 1u;          // 1: minus
        const int mul = hz ? -8 : 1, add = hz ? 1 : 8;
        const int laneA = lane + add + ta * mul, laneB = lane - add + tb * mul;
        const uint32_t recA = (((uint32_t)laneA << 5) & 0x7e0u) | (nsA << 4), recB = (((uint32_t)laneB << 5) & 0x7e0u) | (nsB << 4);
        // the three ahead inside the window: else the exit record
        uint32_t rA, rB;
        rA = (recA & (hz ? reachA_h : reachA_v)) | (~(hz ? reachA_h : reachA_v) & (hz ? eA_h : eA_v));
        rB = (recB & (hz ? reachB_h : reachB_v)) | (~(hz ? reachB_h : reachB_v) & (hz ? eB_h : eB_v));
        if (wx0 <= 0 || wy0 <= 0 || wx0 + 7 >= W - 1 || wy0 + 7 >= H - 1) {       // the window touches the image border
            const bool x_lo = px == 0, x_hi = px == W - 1, y_lo = py == 0, y_hi = py == H - 1;
            const bool brkA = hz ? (x_hi || y_lo || y_hi) : (x_lo || x_hi || y_hi);
            const bool brkB = hz ? (x_lo || y_lo || y_hi) : (x_lo || x_hi || y_lo);
            rA = brkA ? 2u : rA;
            rB = brkB ? 2u : rB;
        }
        nAB = rA | (rB << 16);
        ngM = __ballot((v & 0x7fffu) == 0u);
        stopM = ngM | __ballot(mk);
        hzM = __ballot(hz);
    };
    auto bit = [](uint64_t m, int l) { return (uint32_t)(m >> l) & 1u; };
    // the record of the walk's sign (as its selector sh: 0 plus, 16 minus) at lane l
    auto record = [&](int l, uint32_t sh_) { return ((uint32_t)__builtin_amdgcn_readlane((int)nAB, l) >> sh_) & 0xffffu; };
    int L;
    uint32_t sh;
    {
        bool reuse = false;
        if (!first && c.have && (unsigned)(x - wx0) < 8u && (unsigned)(y - wy0) < 8u) {
            // the anchor lies in the window the first walk ended in: start there unless its first step leaves the window.
            // The caller has cleared the anchor's mark in the bit plane: the same here
            L = (y - wy0) * 8 + (x - wx0);
            stopM = (stopM & ~(1ull << L)) | (ngM & (1ull << L));
            sh = ((st >> bit(hzM, L)) & 1u) ? 0u : 16u;
            reuse = (record(L, sh) & 1u) == 0u;
        }
        if (!reuse) {
            const uint32_t gh = (uint32_t)((ld & 1) ^ 1), plus = (st >> gh) & 1u;
            int nx, ny;
            place(gh, plus, first, nx, ny);
            fetch(nx, ny, gh ? (plus ? RightDir : LeftDir) : (plus ? DownDir : UpDir), false);
            ED_CNT(5, c.diag);
            L = (y - wy0) * 8 + (x - wx0);
            sh = ((st >> bit(hzM, L)) & 1u) ? 0u : 16u;
        }
    }
    ED_T1(3, c.diag);
    // The step loop has ONE exit test: the record's two flags or the next lane's stop bit (a flagged record names lane 0
    // and keeps the selector it was read with)
    if (bit(stopM, L) == 0u) {                                // neither marked nor without gradient
        for (;;) {
            uint32_t fl, e;
            do {
                stopM |= 1ull << L;
                ord = ed_writelane(cnt, L, ord);               // ord[lane L] = cnt
                ++cnt;
                Lv = L;
                e = record(L, sh);
                sh = e & 16u;
                L = (int)((e >> 5) & 63u);
                fl = e & 3u;
                ED_CNT(6, c.diag);
            } while ((fl | bit(stopM, L)) == 0u);
            if (fl != 1u) break;                               // the image border ahead (:1583, :1620, ...), or a pixel that stops the walk
            if (off + (unsigned)cnt > cap) { ok = false; break; }
            ED_T0(2); ED_CNT(4, c.diag);
            x = wx0 + (Lv & 7); y = wy0 + (Lv >> 3);           // the three ahead are not all in the window: the next one, where the record says
            const uint32_t hz = bit(hzM, Lv);
            fetch(wx0 + (int)((e >> 5) & 31u) - 8, wy0 + (int)((e >> 10) & 31u) - 8, hz ? (sh == 0u ? RightDir : LeftDir) : (sh == 0u ? DownDir : UpDir), true);
            L = (y - wy0) * 8 + (x - wx0);
            Lv = L;
            e = record(L, sh);
            ED_T1(2, c.diag);
            if (e & 3u) break;                                 // the border (bit 0 cannot happen: the window was placed around the three ahead)
            sh = e & 16u;
            L = (int)((e >> 5) & 63u);
            if (bit(stopM, L)) break;
        }
    }
    if (Lv >= 0) { c.lastX = (unsigned)(wx0 + (Lv & 7)); c.lastY = (unsigned)(wy0 + (Lv >> 3)); }
    if (ok && off + (unsigned)cnt > cap) ok = false;
    if (ok) retire();
    c.wx0 = wx0; c.wy0 = wy0; c.nAB = nAB; c.stopM = stopM; c.ngM = ngM; c.hzM = hzM; c.have = true;
    off_io = off;
    ED_T1(1, c.diag);
    return ok;
}

// ---- NFA (descriptor_custom.hpp:630-813), lane-uniform
__device__ __noinline__ double ed_log_gamma(double x)
{
    if (x > 15.0)
        return 0.918938533204673 + (x - 0.5) * dm::dlog(x) - x + 0.5 * x * dm::dlog(x * dm::dsinh_small(1 / x) + 1 / (810.0 * dm::dpow(x, 6.0)));
    const double q[7] = { 75122.6331530, 80916.6278952, 36308.2951477, 8687.24529705, 1168.92649479, 83.8676043424, 2.50662827511 };
    double a = (x + 0.5) * dm::dlog(x + 5.5) - (x + 5.5);
    double b = 0.0;
    for (int n = 0; n < 7; n++) {
        a -= dm::dlog(x + (double)n);
        b += q[n] * dm::dpow(x, (double)n);
    }
    return a + dm::dlog(b);
}

__device__ __forceinline__ bool ed_double_equal(double a, double b)
{
    if (a == b) return true;
    const double abs_diff = fabs(a - b), aa = fabs(a), bb = fabs(b);
    double abs_max = aa > bb ? aa : bb;
    if (abs_max < 2.2250738585072014e-308) abs_max = 2.2250738585072014e-308;
    return (abs_diff / abs_max) <= (100.0 * 2.2204460492503131e-16);
}

__device__ __noinline__ double ed_nfa(int n, int k, double p, double logNT)
{
    const double tolerance = 0.1;
    if (n == 0 || k == 0) return -logNT;
    if (n == k) return -logNT - (double)n * dm::dlog10(p);
    const double p_term = p / (1.0 - p);
    const double log1term = ed_log_gamma((double)n + 1.0) - ed_log_gamma((double)k + 1.0) - ed_log_gamma((double)(n - k) + 1.0)
                          + (double)k * dm::dlog(p) + (double)(n - k) * dm::dlog(1.0 - p);
    double term = dm::dexp(log1term);
    if (ed_double_equal(term, 0.0)) {
        if ((double)k > (double)n * p) return -log1term / 2.30258509299404568402 - logNT;
        return -logNT;
    }
    double bin_tail = term;
    for (int i = k + 1; i <= n; i++) {
        const double bin_term = (double)(n - i + 1) / (double)i;
        const double mult_term = bin_term * p_term;
        term *= mult_term;
        bin_tail += term;
        if (bin_term < 1.0) {
            const double err = term * ((1.0 - dm::dpow(mult_term, (double)(n - i + 1))) / (1.0 - mult_term) - 1.0);
            if (err < tolerance * fabs(-dm::dlog10(bin_tail) - logNT) * bin_tail) break;
        }
    }
    return -dm::dlog10(bin_tail) - logNT;
}

// sums over the wave: four DPP exchanges inside the rows of 16 lanes (lane pairs, pairs of pairs, mirrored halves, mirrored
// rows), then the four rows through scalar registers -- the six-step __shfl_xor butterfly is twelve trips through the LDS
// crossbar per 64-bit sum, and a line fit makes four of them
template <int CTRL>
__device__ __forceinline__ int ed_dpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ int wave_sum_i(int v)
{
    v += ed_dpp<0xB1>(v);            // quad_perm [1,0,3,2]
    v += ed_dpp<0x4E>(v);            // quad_perm [2,3,0,1]
    v += ed_dpp<0x141>(v);           // row_half_mirror
    v += ed_dpp<0x140>(v);           // row_mirror: every lane of a row holds the row's sum
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}
template <int CTRL>
__device__ __forceinline__ long long ed_dpp_ll(long long v)
{
    const int lo = ed_dpp<CTRL>((int)(v & 0xffffffffll)), hi = ed_dpp<CTRL>((int)(v >> 32));
    return ((long long)hi << 32) | (unsigned int)lo;
}
__device__ __forceinline__ long long ed_readlane_ll(long long v, int l)
{
    const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffll), l), hi = __builtin_amdgcn_readlane((int)(v >> 32), l);
    return ((long long)hi << 32) | (unsigned int)lo;
}
__device__ __forceinline__ long long wave_sum_ll(long long v)
{
    v += ed_dpp_ll<0xB1>(v);
    v += ed_dpp_ll<0x4E>(v);
    v += ed_dpp_ll<0x141>(v);
    v += ed_dpp_ll<0x140>(v);
    return ed_readlane_ll(v, 0) + ed_readlane_ll(v, 16) + ed_readlane_ll(v, 32) + ed_readlane_ll(v, 48);
}
__device__ __forceinline__ double ed_readlane_d(double v, int l) { return __longlong_as_double(ed_readlane_ll(__double_as_longlong(v), l)); }

struct EdFit { float ATA[4], ATV[2]; };

// [sum a^2, sum a; sum a, n], [sum a b, sum b] over chain[s, s + n): exact integer sums, rounded once to float
__device__ __forceinline__ void ed_sums(const uint32_t* __restrict__ chain, unsigned s, int n, bool horizontal, int lane, float* m4, float* v2)
{
    long long saa = 0, sa = 0, sab = 0, sb = 0;
    for (int i = lane; i < n; i += 64) {
        const uint32_t p = chain[s + i];
        const long long X = p & 0xffffu, Y = p >> 16;
        const long long A = horizontal ? X : Y, B = horizontal ? Y : X;
        saa += A * A; sa += A; sab += A * B; sb += B;
    }
    saa = wave_sum_ll(saa); sa = wave_sum_ll(sa); sab = wave_sum_ll(sab); sb = wave_sum_ll(sb);
    m4[0] = (float)saa; m4[1] = (float)sa; m4[2] = (float)sa; m4[3] = (float)n;
    v2[0] = (float)sab; v2[1] = (float)sb;
}

__device__ __forceinline__ void ed_solve(const EdFit& f, double& e0, double& e1)
{
    const float* A = f.ATA;
    const double coef = 1.0 / ((double)A[0] * (double)A[3] - (double)A[1] * (double)A[2]);
    e0 = coef * ((double)A[3] * (double)f.ATV[0] - (double)A[1] * (double)f.ATV[1]);
    e1 = coef * ((double)A[0] * (double)f.ATV[1] - (double)A[2] * (double)f.ATV[0]);
}

// 256 threads: one walking wave and three fitting ones (512 until the end of round 4: eight waves of one frame on one CU; four let
// twice as many frames share the device -- three octaves 0.714 -> 0.640 ms, pipelined KeyLines 204 k -> 242 k frames/s, the EDLines
// detector in the batch path 362 k -> 417 k; 128 / 192 / 1024 threads: 236 k / 240 k / 150 k)
#ifndef LF_ED_THREADS
#define LF_ED_THREADS 256
#endif
constexpr int ED_THREADS = LF_ED_THREADS, ED_WAVES = ED_THREADS / 64;

// the line records the fitting waves produce, in the order they finish them: per frame [max_lines] of each field
struct EdTemp {
    double* c; float* ep; float* dir; float* sal; int* npx; int* edge; int* idx;
    __device__ EdTemp(uint8_t* base, int n)
    {
        c = (double*)base; ep = (float*)(c + n); dir = ep + 4 * (size_t)n; sal = dir + n;
        npx = (int*)(sal + n); edge = npx + n; idx = edge + n;
    }
};

__global__ __launch_bounds__(ED_THREADS) void k_ed_detect(EdAll all, EdFitParams fp, int n_octaves)
{
    extern __shared__ uint32_t lds[];
    __shared__ int s_wave_count[ED_WAVES];
    __shared__ int s_base, s_edges, s_walked, s_fail, s_next_edge, s_temp_next, s_total;
    const int oc = blockIdx.y, f = blockIdx.x;
#ifdef LF_ED_STAMP
    __shared__ unsigned long long s_diag;
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
    unsigned long long t_walk_end = 0ull;
#endif
    const EdOct& o = all.o[oc];
    const int W = o.W, H = o.H, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t P = (size_t)W * H;
    const uint16_t* g = o.g + (size_t)f * P;
    const int scan = fp.scan;
    const int nW = W > 2 ? (W - 2 + scan - 1) / scan : 0, nH = H > 2 ? (H - 2 + scan - 1) / scan : 0;
    const int n_cand = nW * nH, n_cwords = (n_cand + 31) / 32;
    const int n_mwords = (int)((P + 31) / 32);
    uint32_t* flags = lds;                                   // candidate bits, column-major
    uint32_t* fhz = lds + n_cwords;                          // ... and whether the candidate's pixel is a horizontal-edge one
    uint32_t* ahz = lds + 2 * n_cwords;                      // the same bit per listed anchor
    uint32_t* marks = o.marks_in_lds ? lds + 3 * n_cwords : o.gmarks + (size_t)f * n_mwords;
    int* cnt = o.counts + 4 * (size_t)f;
    for (int i = tid; i < 3 * n_cwords; i += ED_THREADS) flags[i] = 0u;
    for (int i = tid; i < n_mwords; i += ED_THREADS) marks[i] = 0u;
    if (tid == 0) { s_base = 0; s_edges = 0; s_walked = 0; s_fail = 0; s_next_edge = 0; s_temp_next = 0; }
    __syncthreads();
    // ---- anchors (:1504-1532): tested row-major, recorded column-major.  All five loads of a candidate are issued
    // whatever its direction, 8 or 16 candidates per thread in flight: the phase is bound by the latency of the plane.  A
    // thread's candidates are ED_THREADS apart: (row, column) advance by that stride's quotient and remainder -- one
    // division per thread (two per candidate until round 6: 100 of its 150 instructions)
    if (o.aflags) {
        // the candidates k_ed_grad has tested (scan interval 2): its two planes into LDS
        const uint32_t* af = o.aflags + (size_t)f * 2 * n_cwords;
        for (int i = tid; i < n_cwords; i += ED_THREADS) { flags[i] = af[i]; fhz[i] = af[n_cwords + i]; }
    } else {
        const int step_q = ED_THREADS / (nW > 0 ? nW : 1), step_r = ED_THREADS - step_q * nW;
        int ch_n = tid / (nW > 0 ? nW : 1), cw_n = tid - ch_n * nW;
        auto test = [&](uint32_t v, uint32_t va, uint32_t vb, uint32_t vl, uint32_t vr, int b) {
            const int gv = (int)(v & 0x7fffu);
            const bool hz = (v & kHorizontal) != 0;
            const int n1 = (int)((hz ? va : vl) & 0x7fffu), n2 = (int)((hz ? vb : vr) & 0x7fffu);
            if (b >= 0 && gv >= n1 + fp.anchor_threshold && gv >= n2 + fp.anchor_threshold) {
                atomicOr(&flags[b >> 5], 1u << (b & 31));
                if (hz) atomicOr(&fhz[b >> 5], 1u << (b & 31));
            }
        };
        auto next = [&](int i, int& idx) {                     // the candidate's bit (-1 past the end) and pixel; on to the thread's next
            const bool live = i < n_cand;
            const int ch = live ? ch_n : 0, cw = live ? cw_n : 0;
            idx = (1 + scan * ch) * W + 1 + scan * cw;
            cw_n += step_r; ch_n += step_q;
            if (cw_n >= nW) { cw_n -= nW; ++ch_n; }
            return live ? cw * nH + ch : -1;
        };
        if (((scan | W) & 1) == 0) {
            // even columns to the left of every candidate: west | own | east as one 8-byte load at a 4-byte boundary, three loads
            // per candidate, sixteen candidates in flight
            struct __attribute__((packed, aligned(4))) Px4 { uint32_t lo, hi; };
            for (int i0 = tid; i0 < n_cand; i0 += ED_THREADS * 16) {
                Px4 q[16]; uint32_t va[16], vb[16]; int bit_i[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    int idx;
                    bit_i[u] = next(i0 + ED_THREADS * u, idx);
                    q[u] = *reinterpret_cast<const Px4*>(g + idx - 1); va[u] = g[idx - W]; vb[u] = g[idx + W];
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) test(q[u].lo >> 16, va[u], vb[u], q[u].lo & 0xffffu, q[u].hi & 0xffffu, bit_i[u]);
            }
        } else {
            for (int i0 = tid; i0 < n_cand; i0 += ED_THREADS * 8) {
                uint32_t v[8], va[8], vb[8], vl[8], vr[8]; int bit_i[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    int idx;
                    bit_i[u] = next(i0 + ED_THREADS * u, idx);
                    v[u] = g[idx]; va[u] = g[idx - W]; vb[u] = g[idx + W]; vl[u] = g[idx - 1]; vr[u] = g[idx + 1];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) test(v[u], va[u], vb[u], vl[u], vr[u], bit_i[u]);
            }
        }
    }
    __syncthreads();
#ifdef LF_ED_STAMP
    if (LF_ED_STAMP == 10 && tid == 0) s_diag = __builtin_amdgcn_s_memtime() - t_start;   // zeroing + candidates
#endif
    uint32_t* anchors = o.anchors + (size_t)f * o.cap;
    for (int start = 0; start < n_cwords; start += ED_THREADS) {
        const int wi = start + tid;
        const uint32_t word = wi < n_cwords ? flags[wi] : 0u;
        const int c = __popc(word);
        int incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
        if (lane == 63) s_wave_count[wave] = incl;
        __syncthreads();
        int off = s_base;
        for (int w2 = 0; w2 < wave; ++w2) off += s_wave_count[w2];
        off += incl - c;
        uint32_t bits = word;
        while (bits) {
            const int b = __ffs((int)bits) - 1;
            bits &= bits - 1;
            const int bi = wi * 32 + b;
            const int cw = bi / nH, ch = bi - cw * nH;
            if (off < o.cap) {
                anchors[off] = (uint32_t)(1 + scan * cw) | ((uint32_t)(1 + scan * ch) << 16);
                if ((fhz[wi] >> b) & 1u) atomicOr(&ahz[off >> 5], 1u << (off & 31));
            }
            ++off;
        }
        __syncthreads();
        if (tid == ED_THREADS - 1) s_base = off;
        __syncthreads();
    }
    const int n_anchors = __builtin_amdgcn_readfirstlane(s_base);
    uint32_t* part = o.part + (size_t)f * o.cap;
    uint32_t* chain = o.chain + (size_t)f * 2 * o.cap;
    uint32_t* sid = o.sid + (size_t)f * (o.max_edges + 2);
    // ================= smart routing: one wave (a walk stops at pixels earlier walks marked) =================
    if (wave == 0) {
#ifdef LF_ED_STAMP
        const unsigned long long t_walk = __builtin_amdgcn_s_memtime();
        if (LF_ED_STAMP == 7) s_diag = t_walk - t_start;
#endif
        int st = 0;
        unsigned ps = 0, cpos = 0;                            // edges, chain pixels
        if (n_anchors > o.cap) st = 1;                        // the reference returns -1 ("anchor size is larger than its maximal size")
        else {
            EdWalk wk; wk.g = g; wk.marks.p = marks; wk.marks.in_lds = o.marks_in_lds != 0; wk.W = W; wk.H = H; wk.lastX = 0; wk.lastY = 0; wk.wx0 = 0; wk.wy0 = 0; wk.diag = 0ull; wk.nAB = 0u; wk.stopM = 0ull; wk.ngM = 0ull; wk.hzM = 0ull; wk.have = false;
            const unsigned cap = (unsigned)o.cap;
            unsigned offF = 0, offS = 0;                      // kept first / second part pixels
            // the list, 64 anchors per load; every lane looks up its anchor's mark first: most anchors lie on an edge drawn
            // before their block is reached (marks are only ever added), the others are looked up again when their turn comes
            for (int a0 = 0; a0 < n_anchors && !st; a0 += 64) {
              const uint32_t ablk = a0 + lane < n_anchors ? anchors[a0 + lane] : 0u;
              uint64_t todo = __ballot(a0 + lane < n_anchors && !wk.marks.get((int)((ablk >> 16) * W + (ablk & 0xffffu))));
              while (todo) {
                const int k = (int)__builtin_ctzll(todo), a = a0 + k;
                todo &= todo - 1ull;
                const uint32_t an = (uint32_t)__builtin_amdgcn_readlane((int)ablk, k);
                const unsigned x = an & 0xffffu, y = an >> 16;
                const int i = (int)(y * W + x);
                if (__builtin_amdgcn_readfirstlane((int)wk.marks.get(i))) continue;
                if (ps > (unsigned)o.max_edges) { st = 2; break; }
                const bool horizontal = (__builtin_amdgcn_readfirstlane((int)ahz[a >> 5]) >> (a & 31)) & 1;
                unsigned nF = offF;
                if (!ed_walk(wk, x, y, horizontal ? RightDir : DownDir, true, part, 0u - offF, nF, cap)) { st = 2; break; }
                const unsigned lenF = nF - offF;
                wk.marks.clear(i);                                     // the anchor starts the second part as well
                // second part straight into the chain, behind the (still to be reversed) first part: entry t of the second
                // part lands at cpos + lenF + t - 1, i.e. its entry 0 (the anchor again) on top of the first part's last slot,
                // which the reversal below overwrites with the anchor anyway
                unsigned nS = offS;
                if (!ed_walk(wk, x, y, horizontal ? LeftDir : UpDir, false, chain, cpos + lenF - 1u - offS, nS, cap)) { st = 2; break; }
                const unsigned lenS = nS - offS;
                if ((int)(lenF + lenS) < fp.min_line_len + 1) continue;               // short chain: dropped, its marks stay
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                for (unsigned t = lane; t < lenF; t += 64) chain[cpos + t] = part[lenF - 1 - t];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                sid[ps] = cpos;
                cpos += lenF + lenS - 1;
                offF = nF; offS = nS;
                ++ps;
                sid[ps] = cpos;
                // the chain is complete: the fitting waves may have it
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __hip_atomic_store(&s_edges, (int)ps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              }
            }
            if (!st && ps > (unsigned)o.max_edges) st = 2;
#ifdef LF_ED_STAMP
            if (LF_ED_STAMP < 7) s_diag = wk.diag;
            if (LF_ED_STAMP == 11) s_diag = (unsigned long long)ps * 4ull;
#endif
        }
        if (!st) sid[ps] = cpos;
        __hip_atomic_store(&s_fail, st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __hip_atomic_store(&s_walked, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifdef LF_ED_STAMP
        t_walk_end = __builtin_amdgcn_s_memtime();
        if (LF_ED_STAMP == 8) s_diag = t_walk_end - t_walk;
#endif
    }
    // ================= EDline (:2242-2482): the chains are independent -- every wave takes the next one =================
    // ... WHILE the first wave is still walking: a chain is published (s_edges) as soon as it is complete, the other seven
    // waves fit it, and the walking wave joins them when it is done.  A wave appends its lines to the frame's temporary
    // records as it validates them and notes, per chain, how many it kept and how many it had kept when it last set out
    // on a line (what the reference's "too many lines" test looks at); the records are put in chain order afterwards.
    const uint32_t* dxy = o.dxy + (size_t)f * P;
    const int minLen = fp.min_line_len;
    const double thr = fp.fit_err;
    const double logNT = 2.0 * (dm::dlog10((double)(unsigned)W) + dm::dlog10((double)(unsigned)H));
    const EdTemp tl(o.tl + (size_t)f * o.tl_stride, o.max_lines);
    int* e_kept = o.ework + (size_t)f * 3 * (o.max_edges + 2);   // per chain: lines kept,
    int* e_last = e_kept + (o.max_edges + 2);                 // lines kept before the last line was begun (-1: none begun),
    int* e_base = e_last + (o.max_edges + 2);                 // index of the chain's first line in the frame
    for (;;) {
        int edgeID = 0;
        if (lane == 0) edgeID = atomicAdd(&s_next_edge, 1);
        edgeID = __builtin_amdgcn_readfirstlane(edgeID);
        // until the chain is published, or the walk is over without it
        int ready;
        for (;;) {
            ready = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&s_edges, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            if (edgeID < ready) break;
            if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&s_walked, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) {
                ready = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&s_edges, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                break;
            }
            __builtin_amdgcn_s_sleep(16);
        }
        if (edgeID >= ready) break;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        unsigned S = (unsigned)__builtin_amdgcn_readfirstlane((int)sid[edgeID]);
        const unsigned E = (unsigned)__builtin_amdgcn_readfirstlane((int)sid[edgeID + 1]);
        int kept = 0, kept_at_last = -1;
        EdFit fit;
        for (int k = 0; k < 4; ++k) fit.ATA[k] = 0.f;
        fit.ATV[0] = fit.ATV[1] = 0.f;
        double lineFitErr = 0, e0 = 0, e1 = 0;
        while (E > S + (unsigned)minLen) {
            // an initial segment of minLen pixels that fits
            while (E > S + (unsigned)minLen) {
                const uint32_t p0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)chain[S]);
                const bool hz = (__builtin_amdgcn_readfirstlane((int)g[(p0 >> 16) * W + (p0 & 0xffffu)]) & kHorizontal) != 0;
                ed_sums(chain, S, minLen, hz, lane, fit.ATA, fit.ATV);
                ed_solve(fit, e0, e1);
                // ordered sum of the squared residuals (:2519-2525)
                double c2 = 0;
                if (lane < minLen) {
                    const uint32_t p = chain[S + lane];
                    const double X = (double)(p & 0xffffu), Y = (double)(p >> 16);
                    const double c = hz ? (Y - X * e0 - e1) : (X - Y * e0 - e1);
                    c2 = c * c;
                }
                double err = 0;
                for (int i = 0; i < minLen; ++i) err += ed_readlane_d(c2, __builtin_amdgcn_readfirstlane(i));
                lineFitErr = dm::dsqrt(err);
                if (lineFitErr <= thr) break;
                S += 2;
            }
            if (lineFitErr > thr) break;
            kept_at_last = kept;                               // (:2290: the reference gives up when numOfLines >= limit here)
            const unsigned lstart = S;
            const uint32_t pl = (uint32_t)__builtin_amdgcn_readfirstlane((int)chain[S]);
            const bool horizontal = (__builtin_amdgcn_readfirstlane((int)g[(pl >> 16) * W + (pl & 0xffffu)]) & kHorizontal) != 0;
            double coef1 = 0;
            bool bExtended = true, bFirstTry = true;
            int numOfOutlier, tryTimes = 0;
            unsigned newS = 0;
            while (bExtended) {
                tryTimes++;
                if (bFirstTry) { bFirstTry = false; S += (unsigned)minLen; }
                else {
                    float m[4], v[2];
                    ed_sums(chain, newS, (int)(S - newS), horizontal, lane, m, v);
                    for (int k = 0; k < 4; ++k) fit.ATA[k] = fit.ATA[k] + m[k];
                    for (int k = 0; k < 2; ++k) fit.ATV[k] = fit.ATV[k] + v[k];
                    ed_solve(fit, e0, e1);
                }
                coef1 = horizontal ? 1 / dm::dsqrt(e0 * e0 + 1) : 1 / dm::dsqrt(1 + e0 * e0);
                numOfOutlier = 0;
                newS = S;
                // extend while fewer than four consecutive pixels are farther than the threshold: 64 pixels per step
                while (E > S) {
                    const unsigned rem = E - S;
                    const int nv = rem < 64u ? (int)rem : 64;
                    bool outl = false;
                    if (lane < nv) {
                        const uint32_t p = chain[S + lane];
                        const double X = (double)(p & 0xffffu), Y = (double)(p >> 16);
                        const double dis = horizontal ? fabs(e0 * X - Y + e1) * coef1 : fabs(X - e0 * Y - e1) * coef1;
                        outl = dis > thr;
                    }
                    const unsigned long long O = __ballot(outl);
                    int stop = -1;
                    const int lead = O == ~0ull ? 64 : __ffsll((long long)~O) - 1;      // outliers at the start of the chunk
                    if (numOfOutlier > 0 && numOfOutlier + lead >= 4) stop = 3 - numOfOutlier;
                    else {
                        const unsigned long long Q = O & (O >> 1) & (O >> 2) & (O >> 3);
                        if (Q) stop = __ffsll((long long)Q) - 1 + 3;
                    }
                    if (stop >= 0) { S += (unsigned)stop + 1u; numOfOutlier = 4; break; }
                    // the run of outliers the chunk ends with carries into the next
                    const unsigned long long valid = nv == 64 ? ~0ull : ((1ull << nv) - 1ull);
                    if ((O & valid) == valid) numOfOutlier += nv;
                    else {
                        const unsigned long long inv = ~O & valid;              // highest non-outlier position
                        numOfOutlier = nv - 1 - (63 - __clzll((long long)inv));
                    }
                    S += (unsigned)nv;
                }
                S -= (unsigned)numOfOutlier;
                if (!(S > newS && tryTimes < 6)) bExtended = false;
            }
            double q0, q1, q2;
            if (horizontal) { q0 = e0 * coef1; q1 = -1 * coef1; q2 = e1 * coef1; }
            else { q0 = 1 * coef1; q1 = -e0 * coef1; q2 = -e1 * coef1; }
            // ---- LineValidation_ (:2645-2726).  One pass over the line's pixels gathers the gradient sums AND the
            // salience (:2728-2751: the CV_16S plane sumDxDy / 4 read through an unsigned char pointer with a PIXEL index
            // -- byte (i & 1) of element i >> 1), so that the two gathers from the plane are in flight together
            const int n = (int)(S - lstart);
            int mgx = 0, mgy = 0, sal = 0;
            for (int i = lane; i < n; i += 64) {
                const uint32_t p = chain[lstart + i];
                const unsigned bi = (p >> 16) * (unsigned)W + (p & 0xffffu);
                const uint32_t d = dxy[bi], d2 = dxy[bi >> 1];
                mgx += (int)(int16_t)(d & 0xffffu);
                mgy += (int)d >> 16;
                const int vx = (int)(int16_t)(d2 & 0xffffu), vy = (int)d2 >> 16;
                const int gwo = ed_div4_half_even((vx < 0 ? -vx : vx) + (vy < 0 ? -vy : vy));
                sal += (bi & 1u) ? ((gwo >> 8) & 0xff) : (gwo & 0xff);
            }
            mgx = wave_sum_i(mgx); mgy = wave_sum_i(mgy);
            const double adx = fabs(q1), ady = fabs(q0);
            float direction = 0.f;
            bool ok = !(mgx == 0 && mgy == 0);
            if (ok) {
                if (mgx > 0 && mgy >= 0) direction = (float)dm::datan2(-ady, adx);
                if (mgx <= 0 && mgy > 0) direction = (float)dm::datan2(ady, adx);
                if (mgx < 0 && mgy <= 0) direction = (float)dm::datan2(ady, -adx);
                if (mgx >= 0 && mgy < 0) direction = (float)dm::datan2(-ady, -adx);
                const double PI = 3.14159265358979323846;
                if (fabs((double)direction) < 0.15 || PI - fabs((double)direction) < 0.15)
                    if (fabs(q2) < 10 || fabs((double)(unsigned)H - fabs(q2)) < 10) ok = false;
                if (ok && fabs(fabs((double)direction) - PI * 0.5) < 0.15)
                    if (fabs(q2) < 10 || fabs((double)(unsigned)W - fabs(q2)) < 10) ok = false;
                if (ok) {
                    int kk = 0;
                    for (int i = lane; i < n; i += 64) {
                        const uint32_t p = chain[lstart + i];
                        const uint32_t d = dxy[(p >> 16) * W + (p & 0xffffu)];
                        const double pd = dm::datan2(-(double)(int)(int16_t)(d & 0xffffu), (double)((int)d >> 16));
                        const double dis = fabs((double)direction - pd);
                        if (fabs(2 * PI - dis) < 0.392699 || dis < 0.392699) kk++;
                    }
                    kk = wave_sum_i(kk);
                    ok = __builtin_amdgcn_readfirstlane((int)(ed_nfa(n, kk, 0.125, logNT) > 0)) != 0;
                }
            }
            if (ok) {
                const double a1 = q1 * q1, a2 = q0 * q0, a3 = q0 * q1, a4 = q2 * q0, a5 = q2 * q1;
                const uint32_t pa = chain[lstart], pb = chain[S - 1];
                sal = wave_sum_i(sal);
                int slot = 0;
                if (lane == 0) slot = atomicAdd(&s_temp_next, 1);
                slot = __builtin_amdgcn_readfirstlane(slot);
                // (every lane stores the same record: a loop body that ends in work of lane 0 alone can be compiled so that
                // the other lanes run ahead to the loop's head, where the wave-uniform reads then see THEIR values)
                if (slot < o.max_lines) {                      // (more than max_lines: the frame fails below)
                    const unsigned Px = pa & 0xffffu, Py = pa >> 16, Qx = pb & 0xffffu, Qy = pb >> 16;
                    float* ep = tl.ep + 4 * (size_t)slot;
                    ep[0] = (float)(a1 * Px - a3 * Py - a4);
                    ep[1] = (float)(a2 * Py - a3 * Px - a5);
                    ep[2] = (float)(a1 * Qx - a3 * Qy - a4);
                    ep[3] = (float)(a2 * Qy - a3 * Qx - a5);
                    tl.c[slot] = q2;
                    tl.dir[slot] = direction;
                    tl.npx[slot] = n;
                    tl.sal[slot] = (float)sal;
                    tl.edge[slot] = edgeID;
                    tl.idx[slot] = kept;
                }
                kept++;
            }
        }
        e_kept[edgeID] = kept; e_last[edgeID] = kept_at_last;
    }
    __syncthreads();
    if (s_fail) {
        if (tid == 0) { cnt[0] = n_anchors; cnt[1] = -1; cnt[2] = 0; cnt[3] = s_fail; }
        return;
    }
    const int n_edges = __builtin_amdgcn_readfirstlane(s_edges);
    const unsigned limit = min(5u * (unsigned)n_edges, (unsigned)o.max_lines);
    __syncthreads();                                          // (s_fail is written again below)
    // ---- chain order: where each chain's lines start; the reference's test of the running count (:2290)
    if (wave == 0) {
        int run = 0;
        bool bad = s_temp_next > o.max_lines;
        for (int e0i = 0; e0i < n_edges; e0i += 64) {
            const int e = e0i + lane;
            const int c = e < n_edges ? e_kept[e] : 0;
            int incl = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
            const int base = run + incl - c;
            bool over = false;
            if (e < n_edges) {
                e_base[e] = base;
                const int kl = e_last[e];
                over = kl >= 0 && (unsigned)(base + kl) >= limit;
            }
            if (__ballot(over) != 0ull) bad = true;
            run += __shfl(incl, 63);
        }
        if (lane == 0) { s_total = run; s_fail = bad ? 3 : 0; }
    }
    __syncthreads();
    if (s_fail) {
        if (tid == 0) { cnt[0] = n_anchors; cnt[1] = -1; cnt[2] = 0; cnt[3] = s_fail; }
        return;
    }
    float* l_ep = o.l_ep + (size_t)f * o.max_lines * 4;
    double* l_c = o.l_c + (size_t)f * o.max_lines;
    float* l_dir = o.l_dir + (size_t)f * o.max_lines;
    int* l_npx = o.l_npx + (size_t)f * o.max_lines;
    float* l_sal = o.l_sal + (size_t)f * o.max_lines;
    const int total = s_total;
    for (int t = tid; t < total; t += ED_THREADS) {
        const int d = e_base[tl.edge[t]] + tl.idx[t];
        const float4 ep = *reinterpret_cast<const float4*>(tl.ep + 4 * (size_t)t);
        *reinterpret_cast<float4*>(l_ep + 4 * (size_t)d) = ep;
        l_c[d] = tl.c[t]; l_dir[d] = tl.dir[t]; l_npx[d] = tl.npx[t]; l_sal[d] = tl.sal[t];
    }
#ifdef LF_ED_STAMP
    if (tid == 0) { if (LF_ED_STAMP == 9) s_diag = __builtin_amdgcn_s_memtime() - t_walk_end; cnt[0] = n_anchors; cnt[1] = n_edges; cnt[2] = total; cnt[3] = (int)(s_diag >> 2); }
#else
    if (tid == 0) { cnt[0] = n_anchors; cnt[1] = n_edges; cnt[2] = total; cnt[3] = 0; }
#endif
}

size_t ed_detect_lds_bytes(int W, int H, int scan, bool* marks_in_lds)
{
    const int nW = W > 2 ? (W - 2 + scan - 1) / scan : 0, nH = H > 2 ? (H - 2 + scan - 1) / scan : 0;
    const size_t cw = 3 * (((size_t)nW * nH + 31) / 32), mw = ((size_t)W * H + 31) / 32;      // candidate bits, their directions, the anchors' directions
    const bool fits = (cw + mw) * 4 <= 150 * 1024;
    if (marks_in_lds) *marks_in_lds = fits;
    return (fits ? cw + mw : cw) * 4;
}

int launch_ed_detect(const EdAll& all, const EdFitParams& fp, int n_octaves, int n_frames, size_t lds_bytes, hipStream_t s)
{
    if (lds_bytes > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_ed_detect), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(k_ed_detect, dim3(n_frames, n_octaves), dim3(ED_THREADS), lds_bytes, s, all, fp, n_octaves);
    return 0;
}

// ------------------------------------------------------------------------------------------------ k_kl_assemble
// OctaveKeyLines' second half (:726-1022) + detectImpl (:478-509).  One workgroup per frame; lines of all octaves
// indexed 0 .. n-1 in octave order (= octaveLines).  LDS: per line octave / id-in-octave / class / length.
constexpr int kKlMaxLines = 4096;           // lines of one frame over all octaves that the grouping holds in LDS (more: global scratch)
constexpr int kKlHardMax = 32767;           // the reference's own limit: its line counters are `short` (binary_descriptor_custom.cpp:1029-1064)

__global__ __launch_bounds__(256) void k_kl_count(EdAll all, int n_octaves, int n_frames, int* __restrict__ frame_count, int* __restrict__ status)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frames) return;
    int total = 0, st = 0;
    for (int oc = 0; oc < n_octaves; ++oc) {
        const int* c = all.o[oc].counts + 4 * (size_t)f;
        if (c[1] < 0) st = c[3] ? c[3] : 2;
        total += c[2];
    }
    if (total > kKlHardMax) st = 4;
    frame_count[f] = st ? 0 : total;          // a failing detector yields no KeyLines for its frame (:465-468, return value ignored)
    status[f] = st;
}

// BIG = false: frames of up to kKlMaxLines lines, the grouping tables in LDS (every frame seen so far outside the tests).
// BIG = true: the frames beyond (up to the reference's own `short` limit), the same code with the tables in global scratch
// (big: [frame][big_stride] x (f32 + 3 x u16)); either launch leaves the other kind of frame alone.
template <bool BIG>
__global__ __launch_bounds__(256) void k_kl_assemble(EdAll all, int n_octaves, const int* __restrict__ frame_offset, int capacity, KlOut out,
                                                     uint8_t* __restrict__ big, int big_stride, int lds_lines)
{
    __shared__ float l_len[BIG ? 1 : kKlMaxLines];
    __shared__ uint16_t l_oct[BIG ? 1 : kKlMaxLines], l_lid[BIG ? 1 : kKlMaxLines], l_cls[BIG ? 1 : kKlMaxLines];
    __shared__ int s_ostart[LF_MAX_OCTAVES + 1];
    __shared__ int s_wave[4], s_next_class;
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int base = frame_offset[f], n = frame_offset[f + 1] - base;
    if (n <= 0 || base + n > capacity) return;
    if ((n > lds_lines) != BIG || (BIG && n > big_stride)) return;
    uint8_t* mine = BIG ? big + (size_t)f * big_stride * 10 : nullptr;
    float* s_len = BIG ? reinterpret_cast<float*>(mine) : l_len;
    uint16_t* s_oct = BIG ? reinterpret_cast<uint16_t*>(mine + (size_t)big_stride * 4) : l_oct;
    uint16_t* s_lid = BIG ? s_oct + big_stride : l_lid;
    uint16_t* s_cls = BIG ? s_lid + big_stride : l_cls;
    if (tid == 0) {
        int acc = 0;
        for (int oc = 0; oc < n_octaves; ++oc) { s_ostart[oc] = acc; acc += all.o[oc].counts[4 * (size_t)f + 2]; }
        s_ostart[n_octaves] = acc;
        s_next_class = 0;
    }
    __syncthreads();
    float scale[LF_MAX_OCTAVES];
    scale[0] = 1.f;
    {
        const double factor = dm::dsqrt(2.0);
        for (int oc = 1; oc < LF_MAX_OCTAVES; ++oc) scale[oc] = (float)(factor * scale[oc - 1]);
    }
    const double twoPI = 2 * 3.14159265358979323846, PI = 3.14159265358979323846;
    for (int oc = 0; oc < n_octaves; ++oc) {
        const EdOct& o = all.o[oc];
        const int o0 = s_ostart[oc], on = s_ostart[oc + 1] - o0;
        const float* ep_o = o.l_ep + (size_t)f * o.max_lines * 4;
        const double* c_o = o.l_c + (size_t)f * o.max_lines;
        const float* dir_o = o.l_dir + (size_t)f * o.max_lines;
        const int cls0 = s_next_class;
        __syncthreads();
        for (int start = 0; start < on; start += 256) {
            const int l = start + tid;
            bool fresh = false;
            int cls = 0;
            float length = 0.f;
            if (l < on) {
                const float* ep = ep_o + 4 * (size_t)l;
                float dx = (float)fabs((double)(ep[0] - ep[2])), dy = (float)fabs((double)(ep[1] - ep[3]));
                if (oc == 0) {
                    length = dm::fsqrt(dx * dx + dy * dy);
                    fresh = true;
                } else {
                    const float rho1 = (float)(scale[oc] * fabs(c_o[l]));
                    const float tempValue = (float)(rho1 * 0.0152);
                    float diffNearThreshold = (tempValue > 6) ? tempValue : 6;
                    diffNearThreshold = (diffNearThreshold < 12) ? diffNearThreshold : 12;
                    length = scale[oc] * dm::fsqrt(dx * dx + dy * dy);
                    float minEndPointDis = 12;
                    int closeLineID = 0;
                    const float lp0 = scale[oc] * ep[0], lp1 = scale[oc] * ep[1], lp2 = scale[oc] * ep[2], lp3 = scale[oc] * ep[3];
                    for (int nx = 0; nx < o0; ++nx) {
                        const int oid = s_oct[nx], lid = s_lid[nx];
                        const EdOct& q = all.o[oid];
                        const float direction = (float)fabs((double)(dir_o[l] - q.l_dir[(size_t)f * q.max_lines + lid]));
                        if (direction > 0.1745 && (twoPI - direction > 0.1745)) continue;
                        const float rho2 = (float)(scale[oid] * fabs(q.l_c[(size_t)f * q.max_lines + lid]));
                        const float diffNear = (float)fabs((double)(rho1 - rho2));
                        if (diffNear > diffNearThreshold) continue;
                        const float* np_ = q.l_ep + ((size_t)f * q.max_lines + lid) * 4;
                        const float np0 = scale[oid] * np_[0], np1 = scale[oid] * np_[1], np2 = scale[oid] * np_[2], np3 = scale[oid] * np_[3];
                        float endPointDis, minLocalDis, maxLocalDis;
                        dx = lp0 - np0; dy = lp1 - np1;
                        endPointDis = dm::fsqrt(dx * dx + dy * dy);
                        minLocalDis = endPointDis; maxLocalDis = endPointDis;
                        dx = lp2 - np2; dy = lp3 - np3;
                        endPointDis = dm::fsqrt(dx * dx + dy * dy);
                        minLocalDis = (endPointDis < minLocalDis) ? endPointDis : minLocalDis;
                        maxLocalDis = (endPointDis > maxLocalDis) ? endPointDis : maxLocalDis;
                        dx = lp0 - np2; dy = lp1 - np3;
                        endPointDis = dm::fsqrt(dx * dx + dy * dy);
                        minLocalDis = (endPointDis < minLocalDis) ? endPointDis : minLocalDis;
                        maxLocalDis = (endPointDis > maxLocalDis) ? endPointDis : maxLocalDis;
                        dx = lp2 - np0; dy = lp3 - np1;
                        endPointDis = dm::fsqrt(dx * dx + dy * dy);
                        minLocalDis = (endPointDis < minLocalDis) ? endPointDis : minLocalDis;
                        maxLocalDis = (endPointDis > maxLocalDis) ? endPointDis : maxLocalDis;
                        if (((double)maxLocalDis < 0.8 * (double)(length + s_len[nx])) && (minLocalDis < minEndPointDis)) {
                            minEndPointDis = minLocalDis;
                            closeLineID = nx;
                        }
                    }
                    if (minEndPointDis < 12) cls = s_cls[closeLineID];
                    else fresh = true;
                }
            }
            // new class ids in line order: exclusive count of the fresh lines before this one
            const unsigned long long bal = __ballot(fresh);
            const int before = __popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) s_wave[wave] = __popcll(bal);
            __syncthreads();
            int off = s_next_class;
            for (int w2 = 0; w2 < wave; ++w2) off += s_wave[w2];
            if (l < on) {
                if (fresh) cls = off + before;
                s_oct[o0 + l] = (uint16_t)oc; s_lid[o0 + l] = (uint16_t)l; s_cls[o0 + l] = (uint16_t)cls; s_len[o0 + l] = length;
            }
            __syncthreads();
            if (tid == 0) s_next_class += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
            __syncthreads();
        }
        (void)cls0;
    }
    __syncthreads();
    // detectImpl's order: class id, then octaveLines order inside a class
    for (int i = tid; i < n; i += 256) {
        const int ci = s_cls[i];
        int pos = 0;
        for (int j = 0; j < n; ++j) { const int cj = s_cls[j]; pos += (cj < ci) || (cj == ci && j < i); }
        const int oid = s_oct[i], lid = s_lid[i];
        const EdOct& q = all.o[oid];
        const size_t li = (size_t)f * q.max_lines + lid;
        const float direction = q.l_dir[li];
        const float* ep = q.l_ep + 4 * li;
        const float s1 = ep[0], s2 = ep[1], e1 = ep[2], e2 = ep[3];
        const float dx = e1 - s1, dy = e2 - s2;
        bool shouldChange = false;
        if (direction >= -0.75 * PI && direction < -0.25 * PI) { if (dy > 0) shouldChange = true; }
        if (direction >= -0.25 * PI && direction < 0.25 * PI) { if (dx < 0) shouldChange = true; }
        if (direction >= 0.25 * PI && direction < 0.75 * PI) { if (dy < 0) shouldChange = true; }
        if ((direction >= 0.75 * PI && direction < PI) || (direction >= -PI && direction < -0.75 * PI)) { if (dx > 0) shouldChange = true; }
        const float t = scale[oid];
        const size_t k = (size_t)base + pos;
        float io[4], st[4];
        if (shouldChange) { io[0] = e1; io[1] = e2; io[2] = s1; io[3] = s2; }
        else { io[0] = s1; io[1] = s2; io[2] = e1; io[3] = e2; }
        for (int c = 0; c < 4; ++c) st[c] = t * io[c];
        for (int c = 0; c < 4; ++c) { if (out.in_octave) out.in_octave[4 * k + c] = io[c]; if (out.start_end) out.start_end[4 * k + c] = st[c]; }
        if (out.angle) out.angle[k] = direction;
        if (out.num_pixels) out.num_pixels[k] = q.l_npx[li];
        if (out.line_length) out.line_length[k] = s_len[i];
        if (out.octave) out.octave[k] = oid;
        if (out.class_id) out.class_id[k] = ci;
        if (out.salience) out.salience[k] = q.l_sal[li];
        if (out.size) out.size[k] = (st[2] - st[0]) * (st[3] - st[1]);
        if (out.response) out.response[k] = s_len[i] / (float)(q.W > q.H ? q.W : q.H);
        if (out.pt) { out.pt[2 * k] = (st[2] + st[0]) / 2; out.pt[2 * k + 1] = (st[3] + st[1]) / 2; }
        if (out.frame) out.frame[k] = f;
    }
}

// The octave-0 lines of every frame -> the per-colour slot lists the segment stage reads (the EDLines detector behind the
// LineDetectorInterface: lf_set_image_edlines, and behind lf_process_batch: lf_set_detector).  One wave per frame; lines keep
// their order inside a colour; a frame on which the detector gave up has no lines and is counted in *failed.
__global__ __launch_bounds__(64) void k_ed_slots(EdAll all, const uint32_t* __restrict__ maskbits, int Ww, int cap_lines,
                                                float* __restrict__ slot_lines, int* __restrict__ counts, int* __restrict__ failed)
{
    const EdOct& o = all.o[0];
    const int lane = threadIdx.x, f = blockIdx.x;
    const int* fc = o.counts + 4 * (size_t)f;
    const bool gave_up = fc[1] < 0;
    const int n = gave_up ? 0 : fc[2];
    if (gave_up && failed && lane == 0) atomicAdd(failed, 1);
    const float* l_dir = o.l_dir + (size_t)f * o.max_lines;
    const float* l_ep = o.l_ep + (size_t)f * o.max_lines * 4;
    const uint32_t* mb = maskbits + (size_t)f * 3 * o.H * Ww;
    float* sl = slot_lines + (size_t)f * 3 * cap_lines * 4;
    const double PI = 3.14159265358979323846;
    int cnt[3] = { 0, 0, 0 };
    for (int start = 0; start < n; start += 64) {
        const int l = start + lane;
        float io[4] = { 0, 0, 0, 0 };
        int member = 0;
        if (l < n) {
            const float direction = l_dir[l];
            const float* ep = l_ep + 4 * (size_t)l;
            const float s1 = ep[0], s2 = ep[1], e1 = ep[2], e2 = ep[3];
            const float dx = e1 - s1, dy = e2 - s2;
            bool shouldChange = false;                                   // OctaveKeyLines :966-997
            if (direction >= -0.75 * PI && direction < -0.25 * PI) { if (dy > 0) shouldChange = true; }
            if (direction >= -0.25 * PI && direction < 0.25 * PI) { if (dx < 0) shouldChange = true; }
            if (direction >= 0.25 * PI && direction < 0.75 * PI) { if (dy < 0) shouldChange = true; }
            if ((direction >= 0.75 * PI && direction < PI) || (direction >= -PI && direction < -0.75 * PI)) { if (dx > 0) shouldChange = true; }
            if (shouldChange) { io[0] = e1; io[1] = e2; io[2] = s1; io[3] = s2; }
            else { io[0] = s1; io[1] = s2; io[2] = e1; io[3] = e2; }
            const float cx = (io[0] + io[2]) / 2, cy = (io[1] + io[3]) / 2;
            int ix = (int)cx, iy = (int)cy;
            ix = ix < 0 ? 0 : (ix > o.W - 1 ? o.W - 1 : ix);
            iy = iy < 0 ? 0 : (iy > o.H - 1 ? o.H - 1 : iy);
            for (int c = 0; c < 3; ++c)
                if ((mb[((size_t)c * o.H + iy) * Ww + (ix >> 5)] >> (ix & 31)) & 1u) member |= 1 << c;
        }
        for (int c = 0; c < 3; ++c) {
            const unsigned long long bal = __ballot((member >> c) & 1);
            const int pos = cnt[c] + __popcll(bal & ((1ull << lane) - 1ull));
            if (((member >> c) & 1) && pos < cap_lines)
                for (int q = 0; q < 4; ++q) sl[((size_t)c * cap_lines + pos) * 4 + q] = io[q];
            cnt[c] += __popcll(bal);
        }
    }
    if (lane < 3) counts[3 * f + lane] = cnt[lane];         // more than cap_lines: the segment stage reports the overflow
}

void launch_ed_slots(const EdAll& all, int n_frames, const uint32_t* maskbits, int Ww, int cap_lines, float* slot_lines, int* counts, int* failed, hipStream_t s)
{
    hipLaunchKernelGGL(k_ed_slots, dim3(n_frames), dim3(64), 0, s, all, maskbits, Ww, cap_lines, slot_lines, counts, failed);
}

void launch_kl_count(const EdAll& all, int n_octaves, int n_frames, int* frame_count, int* status, hipStream_t s)
{
    hipLaunchKernelGGL(k_kl_count, dim3((n_frames + 255) / 256), dim3(256), 0, s, all, n_octaves, n_frames, frame_count, status);
}

void launch_kl_assemble(const EdAll& all, int n_octaves, int n_frames, const int* frame_offset, int capacity, const KlOut& out, uint8_t* big, int big_stride,
                        int lds_lines, hipStream_t s)
{
    lds_lines = lds_lines < 1 ? 1 : (lds_lines > kKlMaxLines ? kKlMaxLines : lds_lines);
    hipLaunchKernelGGL(k_kl_assemble<false>, dim3(n_frames), dim3(256), 0, s, all, n_octaves, frame_offset, capacity, out, big, big_stride, lds_lines);
    // frames with more than kKlMaxLines lines (none so far outside the tests): the same grouping with its tables in global scratch
    int max_total = 0;
    for (int oc = 0; oc < n_octaves; ++oc) max_total += all.o[oc].max_lines;
    if (big && max_total > lds_lines)
        hipLaunchKernelGGL(k_kl_assemble<true>, dim3(n_frames), dim3(256), 0, s, all, n_octaves, frame_offset, capacity, out, big, big_stride, lds_lines);
}

// exclusive scan of the per-frame counts -> frame_offset [n_frames + 1]; total and overflow flag into pinned[0..1]
__global__ void k_kl_offsets(int n_frames, const int* __restrict__ frame_count, int capacity, int* __restrict__ frame_offset, int* __restrict__ totals)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int acc = 0;
    for (int f = 0; f < n_frames; ++f) { frame_offset[f] = acc; acc += frame_count[f]; }
    frame_offset[n_frames] = acc;
    totals[0] = acc;
    totals[1] = acc > capacity ? 1 : 0;
    totals[2] = acc > capacity ? 0 : acc;        // what the descriptor stage may walk: nothing when the output overflows
}

void launch_kl_offsets(int n_frames, const int* frame_count, int capacity, int* frame_offset, int* totals, hipStream_t s)
{
    hipLaunchKernelGGL(k_kl_offsets, dim3(1), dim3(64), 0, s, n_frames, frame_count, capacity, frame_offset, totals);
}

// ---- the `mask` argument of BinaryDescriptor::detect (ref: binary_descriptor_custom.cpp:509-519), with its loop as written:
//     for (keyCounter = 0; keyCounter < keylines.size(); keyCounter++)
//         if (mask(start) == 0 && mask(end) == 0) keylines.erase(keylines.begin() + keyCounter);
// -- no step back after the erase, so the KeyLine that slides into the erased place is never tested: in a run of consecutive
// KeyLines that fail the test the 1st, 3rd, 5th ... are erased and the 2nd, 4th ... survive.  With lastgood(j) = the last KeyLine
// in front of j that passes, erased(j) = fails(j) and (j - lastgood(j) - 1) even: a prefix maximum.  (LSDDetectorC::detect has the
// step back, LSDDetector_custom.cpp:203-213: every failing KeyLine goes; lanefront_lsdkl.inc.)
// k_kl_mask_flags: one workgroup per frame over the assembled (unmasked) KeyLines: erased[] and the frame's kept count.
__global__ __launch_bounds__(256) void k_kl_mask_flags(const int* __restrict__ fo, const float* __restrict__ start_end, const uint8_t* __restrict__ masks,
                                                      int rows, int cols, int capacity, uint8_t* __restrict__ erased, int* __restrict__ kept_count)
{
    __shared__ int s_w[4], s_k[4];
    const int f = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int base = fo[f], n = fo[f + 1] - base;
    if (n <= 0 || base + n > capacity) { if (t == 0) kept_count[f] = 0; return; }
    const uint8_t* mk = masks + (size_t)f * rows * cols;
    int carry = -1, kept = 0;                                              // last passing KeyLine so far (index in the frame)
    for (int j0 = 0; j0 < n; j0 += 256) {
        const int j = j0 + t;
        bool fails = false;
        if (j < n) {
            const float* e = start_end + 4 * (size_t)(base + j);
            // (clamped into the mask: a KeyLine end on the last row / column of an odd-sized image may round to rows / cols)
            const int sx = min(max((int)e[0], 0), cols - 1), sy = min(max((int)e[1], 0), rows - 1);
            const int ex = min(max((int)e[2], 0), cols - 1), ey = min(max((int)e[3], 0), rows - 1);
            fails = mk[(size_t)sy * cols + sx] == 0 && mk[(size_t)ey * cols + ex] == 0;
        }
        int lg = (j < n && !fails) ? j : -1;                                // inclusive prefix maximum over the chunk
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(lg, d); if (lane >= d) lg = max(lg, o); }
        if (lane == 63) s_w[wave] = lg;
        __syncthreads();
        int before = carry;
        for (int k = 0; k < wave; ++k) before = max(before, s_w[k]);
        lg = max(lg, before);
        const bool er = fails && (((j - lg - 1) & 1) == 0);
        if (j < n) erased[base + j] = er ? 1 : 0;
        const unsigned long long bk = __ballot(j < n && !er);
        if (lane == 0) s_k[wave] = __popcll(bk);
        carry = max(max(max(carry, s_w[0]), max(s_w[1], s_w[2])), s_w[3]);
        __syncthreads();
        kept += s_k[0] + s_k[1] + s_k[2] + s_k[3];
        __syncthreads();
    }
    if (t == 0) kept_count[f] = kept;
}

// k_kl_mask_move: the kept KeyLines of every frame, in order, from the assembled arrays to the final ones at the new offsets
__global__ __launch_bounds__(256) void k_kl_mask_move(const int* __restrict__ fo_src, const int* __restrict__ fo_dst, int capacity, const uint8_t* __restrict__ erased,
                                                     KlOut src, KlOut dst)
{
    __shared__ int s_k[4];
    const int f = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int base = fo_src[f], n = fo_src[f + 1] - base, out0 = fo_dst[f];
    if (n <= 0 || base + n > capacity || fo_dst[f + 1] > capacity) return;
    int done = 0;
    for (int j0 = 0; j0 < n; j0 += 256) {
        const int j = j0 + t;
        const bool keep = j < n && !erased[base + j];
        const unsigned long long bk = __ballot(keep);
        if (lane == 0) s_k[wave] = __popcll(bk);
        __syncthreads();
        int r = done + __popcll(bk & ((1ull << lane) - 1ull));
        for (int k = 0; k < wave; ++k) r += s_k[k];
        if (keep) {
            const size_t a = (size_t)base + j, b = (size_t)out0 + r;
#define MV4(fld) if (src.fld && dst.fld) { dst.fld[4 * b] = src.fld[4 * a]; dst.fld[4 * b + 1] = src.fld[4 * a + 1]; dst.fld[4 * b + 2] = src.fld[4 * a + 2]; dst.fld[4 * b + 3] = src.fld[4 * a + 3]; }
#define MV1(fld) if (src.fld && dst.fld) dst.fld[b] = src.fld[a];
            MV4(start_end) MV4(in_octave) MV1(angle) MV1(num_pixels) MV1(line_length) MV1(octave) MV1(class_id) MV1(response) MV1(size) MV1(salience) MV1(frame)
            if (src.pt && dst.pt) { dst.pt[2 * b] = src.pt[2 * a]; dst.pt[2 * b + 1] = src.pt[2 * a + 1]; }
#undef MV4
#undef MV1
        }
        done += s_k[0] + s_k[1] + s_k[2] + s_k[3];
        __syncthreads();
    }
}

void launch_kl_mask(int n_frames, const int* fo_src, int* fo_dst, int* totals, int capacity, const uint8_t* masks, int rows, int cols, uint8_t* erased,
                    int* kept_count, const KlOut& src, const KlOut& dst, hipStream_t s)
{
    hipLaunchKernelGGL(k_kl_mask_flags, dim3(n_frames), dim3(256), 0, s, fo_src, src.start_end, masks, rows, cols, capacity, erased, kept_count);
    launch_kl_offsets(n_frames, kept_count, capacity, fo_dst, totals, s);
    hipLaunchKernelGGL(k_kl_mask_move, dim3(n_frames), dim3(256), 0, s, fo_src, fo_dst, capacity, erased, src, dst);
}


}  // namespace lf
