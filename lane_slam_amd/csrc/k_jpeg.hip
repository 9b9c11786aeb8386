// K_jpeg: device half of the JPEG ingest stage -- everything libjpeg-turbo
// does after its entropy decoder, for a whole batch at once:
//   k_jpeg_idct   sparse quantised coefficients -> dequantise -> 8x8 inverse DCT -> component planes
//   k_jpeg_color  chroma upsampling + YCbCr -> BGR, written as the HWC u8 frames the front end consumes
// Replaces cv2.imdecode(data, IMREAD_COLOR) (ref: src/duckietown/include/duckietown_utils/jpg.py:21-31);
// arithmetic as restated in oracle/lf_oracle_jpeg.c (pinned against libjpeg-turbo vectors): the
// accurate 13-bit integer inverse DCT in two passes, triangle ("fancy") upsampling for 2x1 / 2x2
// chroma, 16-bit fixed-point colour conversion.  All integer, bit exact.
#include "common.h"
#include "jpeg_entropy.h"

namespace lf {

namespace {

__device__ __forceinline__ int32_t mulw(int32_t a, int32_t c) { return (int32_t)((uint32_t)a * (uint32_t)c); }
__device__ __forceinline__ int32_t addw(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
__device__ __forceinline__ int32_t subw(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }

// 8-point inverse DCT (Loeffler-Ligtenberg-Moschytz, 13-bit constants), result >> shift with rounding
__device__ __forceinline__ void idct8(const int32_t in[8], int32_t out[8], int shift)
{
    const int32_t z1 = mulw(addw(in[2], in[6]), 4433);
    const int32_t t2 = addw(z1, mulw(in[6], -15137));
    const int32_t t3 = addw(z1, mulw(in[2], 6270));
    const int32_t t0 = (int32_t)((uint32_t)addw(in[0], in[4]) << 13);
    const int32_t t1 = (int32_t)((uint32_t)subw(in[0], in[4]) << 13);
    const int32_t e0 = addw(t0, t3), e3 = subw(t0, t3), e1 = addw(t1, t2), e2 = subw(t1, t2);
    int32_t o0 = in[7], o1 = in[5], o2 = in[3], o3 = in[1];
    int32_t y1 = addw(o0, o3), y2 = addw(o1, o2), y3 = addw(o0, o2), y4 = addw(o1, o3);
    const int32_t y5 = mulw(addw(y3, y4), 9633);
    o0 = mulw(o0, 2446); o1 = mulw(o1, 16819); o2 = mulw(o2, 25172); o3 = mulw(o3, 12299);
    y1 = mulw(y1, -7373); y2 = mulw(y2, -20995);
    y3 = addw(mulw(y3, -16069), y5); y4 = addw(mulw(y4, -3196), y5);
    o0 = addw(o0, addw(y1, y3)); o1 = addw(o1, addw(y2, y4));
    o2 = addw(o2, addw(y2, y3)); o3 = addw(o3, addw(y1, y4));
    const int32_t r = (int32_t)1 << (shift - 1);
    out[0] = addw(addw(e0, o3), r) >> shift; out[7] = addw(subw(e0, o3), r) >> shift;
    out[1] = addw(addw(e1, o2), r) >> shift; out[6] = addw(subw(e1, o2), r) >> shift;
    out[2] = addw(addw(e2, o1), r) >> shift; out[5] = addw(subw(e2, o1), r) >> shift;
    out[3] = addw(addw(e3, o0), r) >> shift; out[4] = addw(subw(e3, o0), r) >> shift;
}

// libjpeg's post-IDCT range limit: table indexed by (x & 1023); x + 128 clamped for |x| < 384
__device__ __forceinline__ uint32_t idct_limit(int32_t x)
{
    const int idx = (int)((uint32_t)x & 1023u);
    return idx < 128 ? (uint32_t)(128 + idx) : idx < 512 ? 255u : idx < 896 ? 0u : (uint32_t)(idx - 896);
}

constexpr int kBlocksPerWg = 32;     // 8 threads per 8x8 block
constexpr int kBlockStride = 72;     // LDS words per block (64 + 8: column pass conflict free)

}  // namespace

// planes: [frame][component][Hp][Wp] u8
__global__ __launch_bounds__(256) void k_jpeg_idct(JpegGeom g, const jpeg::FrameHeader* __restrict__ hdrs,
                                                   const uint32_t* __restrict__ entries,
                                                   const uint32_t* __restrict__ block_end, uint8_t* __restrict__ planes)
{
    __shared__ int32_t ws[kBlocksPerWg * kBlockStride];
    const int f = blockIdx.y;
    const jpeg::FrameHeader* H = hdrs + f;
    const int t = threadIdx.x, lb = t >> 3, k = t & 7;
    const int b = blockIdx.x * kBlocksPerWg + lb;
    const int nblocks = H->valid ? H->nblocks : 0;
    if ((int)blockIdx.x * kBlocksPerWg >= nblocks) return;     // whole workgroup beyond this frame's scan
    const bool act = b < nblocks;
    int32_t* w = ws + lb * kBlockStride;
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i * 8 + k] = 0;
    __syncthreads();
    int comp = 0, bx = 0, by = 0;
    if (act) {
        const int hm = H->hmax, vm = H->vmax;
        const int luma = H->ncomp == 1 ? 1 : hm * vm;
        const int bpm = H->ncomp == 1 ? 1 : luma + 2;
        const int mcu = b / bpm, r = b - mcu * bpm;
        const int my = mcu / H->mcux, mx = mcu - my * H->mcux;
        if (r < luma) { comp = 0; bx = mx * hm + (r % hm); by = my * vm + (r / hm); }
        else { comp = r - luma + 1; bx = mx; by = my; }
        const uint16_t* q = H->qt[comp];
        const uint32_t* be = block_end + H->block_base;
        const uint32_t e0 = b ? be[b - 1] : 0u, e1 = be[b];
        const uint32_t* en = entries + H->entry_base;
        for (uint32_t e = e0 + (uint32_t)k; e < e1; e += 8) {
            const uint32_t v = en[e];
            const int pos = (int)(v >> 16) & 63;
            w[pos] = mulw((int32_t)(int16_t)(v & 0xffffu), (int32_t)q[pos]);
        }
    }
    __syncthreads();
    int32_t in[8], res[8];
    // pass 1: thread k owns column k (2 extra fraction bits kept)
#pragma unroll
    for (int r = 0; r < 8; ++r) in[r] = w[r * 8 + k];
    idct8(in, res, 13 - 2);
#pragma unroll
    for (int r = 0; r < 8; ++r) w[r * 8 + k] = res[r];
    __syncthreads();
    // pass 2: thread k owns row k; remove 2 + 3 bits, level shift, limit, store 8 samples
#pragma unroll
    for (int c = 0; c < 8; ++c) in[c] = w[k * 8 + c];
    idct8(in, res, 13 + 2 + 3);
    if (act) {
        const uint32_t lo = idct_limit(res[0]) | (idct_limit(res[1]) << 8) | (idct_limit(res[2]) << 16) | (idct_limit(res[3]) << 24);
        const uint32_t hi = idct_limit(res[4]) | (idct_limit(res[5]) << 8) | (idct_limit(res[6]) << 16) | (idct_limit(res[7]) << 24);
        uint8_t* pl = planes + ((size_t)f * 3 + comp) * g.Hp * g.Wp;
        *reinterpret_cast<uint2*>(pl + (size_t)(by * 8 + k) * g.Wp + bx * 8) = make_uint2(lo, hi);
    }
}

namespace {

// chroma sample for output pixel (x, y): libjpeg's fancy upsampling (jdsample.c h2v1 / h2v2),
// dw x dh = real size of the subsampled plane
__device__ __forceinline__ int chroma_at(const uint8_t* pl, int Wp, int hm, int vm, int dw, int dh, int x, int y)
{
    if (hm == 1) return pl[(size_t)y * Wp + x];
    const int cc = x >> 1;
    if (vm == 1) {
        const uint8_t* in = pl + (size_t)y * Wp;
        if (dw <= 2) return in[cc];
        if ((x & 1) == 0) return cc == 0 ? in[0] : (3 * in[cc] + in[cc - 1] + 1) >> 2;
        return cc == dw - 1 ? in[cc] : (3 * in[cc] + in[cc + 1] + 2) >> 2;
    }
    const int r = y >> 1;
    if (dw <= 2) return pl[(size_t)r * Wp + cc];
    int rn = (y & 1) ? r + 1 : r - 1;
    rn = rn < 0 ? 0 : (rn > dh - 1 ? dh - 1 : rn);
    const uint8_t* in0 = pl + (size_t)r * Wp;
    const uint8_t* in1 = pl + (size_t)rn * Wp;
    const int here = 3 * in0[cc] + in1[cc];
    if ((x & 1) == 0) return cc == 0 ? (here * 4 + 8) >> 4 : (here * 3 + 3 * in0[cc - 1] + in1[cc - 1] + 8) >> 4;
    return cc == dw - 1 ? (here * 4 + 7) >> 4 : (here * 3 + 3 * in0[cc + 1] + in1[cc + 1] + 7) >> 4;
}

__device__ __forceinline__ uint32_t clamp255(int v) { return (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// Eight consecutive chroma samples for output pixels x0 .. x0+7 of row y (x0 % 8 == 0, cols % 8 == 0, so
// the four source samples of a 2:1 plane are one aligned dword and only the two outer neighbours need
// single-byte loads).  Same arithmetic as chroma_at.
__device__ __forceinline__ void chroma8(const uint8_t* pl, int Wp, int hm, int vm, int dw, int dh, int x0, int y, int out[8])
{
    if (hm == 1) {
        const uint2 w = *reinterpret_cast<const uint2*>(pl + (size_t)y * Wp + x0);
#pragma unroll
        for (int i = 0; i < 4; ++i) { out[i] = (int)((w.x >> (8 * i)) & 255u); out[4 + i] = (int)((w.y >> (8 * i)) & 255u); }
        return;
    }
    const int c0 = x0 >> 1;
    int s[6];                                     // samples (2x1) or column sums 3*near + far (2x2) for c0-1 .. c0+4
    if (vm == 1) {
        const uint8_t* in = pl + (size_t)y * Wp;
        const uint32_t w = *reinterpret_cast<const uint32_t*>(in + c0);
#pragma unroll
        for (int i = 0; i < 4; ++i) s[1 + i] = (int)((w >> (8 * i)) & 255u);
        s[0] = c0 > 0 ? in[c0 - 1] : 0;
        s[5] = c0 + 4 < dw ? in[c0 + 4] : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cc = c0 + i;
            out[2 * i] = cc == 0 ? s[1] : (3 * s[1 + i] + s[i] + 1) >> 2;
            out[2 * i + 1] = cc == dw - 1 ? s[1 + i] : (3 * s[1 + i] + s[2 + i] + 2) >> 2;
        }
        return;
    }
    const int r = y >> 1;
    int rn = (y & 1) ? r + 1 : r - 1;
    rn = rn < 0 ? 0 : (rn > dh - 1 ? dh - 1 : rn);
    const uint8_t* in0 = pl + (size_t)r * Wp;
    const uint8_t* in1 = pl + (size_t)rn * Wp;
    const uint32_t w0 = *reinterpret_cast<const uint32_t*>(in0 + c0);
    const uint32_t w1 = *reinterpret_cast<const uint32_t*>(in1 + c0);
#pragma unroll
    for (int i = 0; i < 4; ++i) s[1 + i] = 3 * (int)((w0 >> (8 * i)) & 255u) + (int)((w1 >> (8 * i)) & 255u);
    s[0] = c0 > 0 ? 3 * in0[c0 - 1] + in1[c0 - 1] : 0;
    s[5] = c0 + 4 < dw ? 3 * in0[c0 + 4] + in1[c0 + 4] : 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int cc = c0 + i;
        out[2 * i] = cc == 0 ? (s[1] * 4 + 8) >> 4 : (s[1 + i] * 3 + s[i] + 8) >> 4;
        out[2 * i + 1] = cc == dw - 1 ? (s[1 + i] * 4 + 7) >> 4 : (s[1 + i] * 3 + s[2 + i] + 7) >> 4;
    }
}

}  // namespace

// frames: [frame][rows][cols][3] u8 BGR.  One thread = 8 horizontally adjacent pixels (24 output bytes).
// hdr_stride: bytes between the headers of consecutive frames (FrameHeader alone, or the head of a DevFrame)
__global__ __launch_bounds__(256) void k_jpeg_color(JpegGeom g, const jpeg::FrameHeader* __restrict__ hdrs, size_t hdr_stride,
                                                    const uint8_t* __restrict__ planes, uint8_t* __restrict__ frames)
{
    const int f = blockIdx.z, y = blockIdx.y + g.first_row;
    const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (x0 >= g.cols) return;
    const jpeg::FrameHeader* H = reinterpret_cast<const jpeg::FrameHeader*>(reinterpret_cast<const char*>(hdrs) + (size_t)f * hdr_stride);
    uint8_t* out = frames + ((size_t)f * g.rows + y) * g.cols * 3 + (size_t)x0 * 3;
    const int npx = g.cols - x0 < 8 ? g.cols - x0 : 8;
    uint32_t px[8][3];
    if (!H->valid) {
#pragma unroll
        for (int i = 0; i < 8; ++i) px[i][0] = px[i][1] = px[i][2] = 0u;
    } else {
        const size_t plane = (size_t)g.Hp * g.Wp;
        const uint8_t* pY = planes + (size_t)f * 3 * plane;
        const int hm = H->hmax, vm = H->vmax;
        const int dw = (g.cols + hm - 1) / hm, dh = (g.rows + vm - 1) / vm;
        int Y[8], u[8], v[8];
        const bool fast = (g.cols & 7) == 0 && dw > 2;
        if (fast) {
            const uint2 w = *reinterpret_cast<const uint2*>(pY + (size_t)y * g.Wp + x0);
#pragma unroll
            for (int i = 0; i < 4; ++i) { Y[i] = (int)((w.x >> (8 * i)) & 255u); Y[4 + i] = (int)((w.y >> (8 * i)) & 255u); }
            if (H->ncomp == 3) {
                chroma8(pY + plane, g.Wp, hm, vm, dw, dh, x0, y, u);
                chroma8(pY + 2 * plane, g.Wp, hm, vm, dw, dh, x0, y, v);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int x = x0 + i < g.cols ? x0 + i : g.cols - 1;
                Y[i] = pY[(size_t)y * g.Wp + x];
                u[i] = v[i] = 0;
                if (H->ncomp == 3) {
                    u[i] = chroma_at(pY + plane, g.Wp, hm, vm, dw, dh, x, y);
                    v[i] = chroma_at(pY + 2 * plane, g.Wp, hm, vm, dw, dh, x, y);
                }
            }
        }
        // libjpeg's constants: FIX(1.40200), FIX(1.77200), FIX(0.71414), FIX(0.34414) at 16 bits
        const int c_r = 91881, c_b = 116130, c_gr = 46802, c_gb = 22554;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (H->ncomp == 1) { px[i][0] = px[i][1] = px[i][2] = (uint32_t)Y[i]; continue; }
            if (H->is_rgb) { px[i][2] = (uint32_t)Y[i]; px[i][1] = (uint32_t)u[i]; px[i][0] = (uint32_t)v[i]; continue; }
            const int cb = u[i] - 128, cr = v[i] - 128;
            px[i][2] = clamp255(Y[i] + ((c_r * cr + 32768) >> 16));
            px[i][1] = clamp255(Y[i] + ((-c_gb * cb + 32768 - c_gr * cr) >> 16));
            px[i][0] = clamp255(Y[i] + ((c_b * cb + 32768) >> 16));
        }
    }
    if (npx == 8 && (g.cols & 7) == 0) {
        uint32_t o[6];                                           // 24 bytes, 8-byte aligned when cols % 8 == 0
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int b = 4 * q;
            o[3 * q + 0] = px[b][0] | (px[b][1] << 8) | (px[b][2] << 16) | (px[b + 1][0] << 24);
            o[3 * q + 1] = px[b + 1][1] | (px[b + 1][2] << 8) | (px[b + 2][0] << 16) | (px[b + 2][1] << 24);
            o[3 * q + 2] = px[b + 2][2] | (px[b + 3][0] << 8) | (px[b + 3][1] << 16) | (px[b + 3][2] << 24);
        }
        uint2* o2 = reinterpret_cast<uint2*>(out);
        o2[0] = make_uint2(o[0], o[1]);
        o2[1] = make_uint2(o[2], o[3]);
        o2[2] = make_uint2(o[4], o[5]);
    } else {
        for (int i = 0; i < npx; ++i) {
            out[3 * i + 0] = (uint8_t)px[i][0];
            out[3 * i + 1] = (uint8_t)px[i][1];
            out[3 * i + 2] = (uint8_t)px[i][2];
        }
    }
}

void launch_jpeg_decode(const JpegGeom& g, int n_frames, int max_blocks, const jpeg::FrameHeader* hdrs,
                        const uint32_t* entries, const uint32_t* block_end, uint8_t* planes, uint8_t* frames,
                        hipStream_t s)
{
    if (max_blocks > 0) {
        const dim3 grid((unsigned)((max_blocks + kBlocksPerWg - 1) / kBlocksPerWg), (unsigned)n_frames);
        hipLaunchKernelGGL(k_jpeg_idct, grid, dim3(256), 0, s, g, hdrs, entries, block_end, planes);
    }
    launch_jpeg_color(g, n_frames, hdrs, sizeof(jpeg::FrameHeader), planes, frames, s);
}

void launch_jpeg_color(const JpegGeom& g, int n_frames, const void* hdrs, size_t hdr_stride, const uint8_t* planes, uint8_t* frames, hipStream_t s)
{
    const int tx = 64;
    const dim3 cgrid((unsigned)((g.cols + 8 * tx - 1) / (8 * tx)), (unsigned)(g.rows - g.first_row), (unsigned)n_frames);
    hipLaunchKernelGGL(k_jpeg_color, cgrid, dim3(tx), 0, s, g, static_cast<const jpeg::FrameHeader*>(hdrs), hdr_stride, planes, frames);
}

}  // namespace lf
