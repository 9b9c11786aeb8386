// Associator (a-10 / a-11): nearest neighbour of every query descriptor in the live map.
//
// Reference semantics: BinaryDescriptorMatcher::match
// (/root/reference/src/line_descriptor/src/binary_descriptor_matcher.cpp:197-254) with
// Mihasher(256, 32), K = 1 (:635-753) and the popcount Hamming distance of
// bitops_custom.hpp:83-96: the exact Hamming nearest neighbour; candidates farther than
// D = 128 are never reported (:721).  The reference's multi-index hash is a CPU
// pointer-chasing structure rebuilt on every call; on MI355X the N x M distance matrix is
// one dense int8 contraction: bits -> +-64 bytes, dot / 4096 = 256 - 2*hamming, exactly.
//
// k_assoc_pack : 32-byte codes -> 256 int8 (+64 for bit 0, -64 for bit 1), zero padded rows.
// k_assoc      : 256 queries per workgroup (4 waves x 2 x 32 rows, A fragments resident in
//   VGPRs; a B fragment read from LDS feeds two MFMAs), map streamed through LDS in 64-entry tiles by LDS-DMA
//   (double buffered, source-swizzled: conflict-free ds_read_b128), v_mfma_i32_32x32x32_i8 over K = 256 plus one step that folds the column-block number into
//   the accumulator, so the running arg-max is one v_max3 per register pair (see the kernel) -- no
//   cross-lane traffic until the end; ties resolve to the lowest map index.  Map chunks are
//   spread over gridDim.y and merged with atomicMin on the packed word (order free).
//   Algorithmic ops: 2*N*M*256 int8.
// k_assoc_float: 72-d float LBD, Euclidean, on v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain).
#include "common.h"

namespace lf {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int QB = 2;          // 32-query row blocks per wave
constexpr int AQ = 128 * QB;   // queries per workgroup (4 waves)
constexpr int AM = 64;         // map entries per LDS tile

__global__ void k_assoc_pack(const uint8_t* __restrict__ codes, int n, int n_pad, int8_t* __restrict__ out)
{
    // one thread per (row, byte): 8 int8 = 2 dwords
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t total = (size_t)n_pad * 32;
    if (t >= total) return;
    size_t row = t >> 5;
    uint32_t lo = 0, hi = 0;
    if (row < (size_t)n) {
        uint32_t b = codes[t];
        // nibble -> 4 bytes of 0/1, then 0 -> 0x40 (+64), 1 -> 0xC0 (-64)
        uint32_t w0 = ((b & 15u) * 0x00204081u) & 0x01010101u;
        uint32_t w1 = ((b >> 4) * 0x00204081u) & 0x01010101u;
        lo = (w0 << 7) | 0x40404040u;
        hi = (w1 << 7) | 0x40404040u;
    }
    uint2* o = reinterpret_cast<uint2*>(out + t * 8);
    *o = make_uint2(lo, hi);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_assoc(const int8_t* __restrict__ qx, int nq, const int8_t* __restrict__ mx,
                                               int nm, int nm_pad, int m_chunk, unsigned int* __restrict__ best)
{
    __shared__ __attribute__((aligned(1024))) int8_t tile[2 * AM * 256];     // double buffered map tile, linear rows
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q0 = blockIdx.x * AQ + wave * (32 * QB);
    const int r32 = lane & 31, half = lane >> 5;
    // QB row blocks of 32 queries per wave: every B fragment read from LDS feeds QB MFMAs
    v4i A[QB][8];
#pragma unroll
    for (int b = 0; b < QB; ++b)
#pragma unroll
        for (int s = 0; s < 8; ++s)
            A[b][s] = *reinterpret_cast<const v4i*>(qx + (size_t)(q0 + 32 * b + r32) * 256 + 32 * s + 16 * half);
    // Arg-max inside the matrix core.  Codes are +-64 bytes, so a chain of 8 MFMAs leaves 4096 * dot in the
    // accumulator; a NINTH step multiplies the constant row fragment [64, 1, 0, ...] with the column fragment
    // [-(t >> 6), -(t & 63), 0, ...], which subtracts t, the running number of the 32-column block inside this
    // workgroup's map chunk.  key = 4096 * dot - t orders candidates by distance, then by column block, so the
    // whole epilogue is ONE v_max3 per pair of accumulator registers: no zeroing (the chain starts from the
    // inline constant 0), no packing, no select.  dot and t are recovered from the key at the very end.
    v4i AX;
    AX[0] = half == 0 ? 0x00000140 : 0;      // bytes: 64, 1, 0, 0
    AX[1] = AX[2] = AX[3] = 0;
    int running[QB][16];
#pragma unroll
    for (int b = 0; b < QB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) running[b][r] = (int)0x80000000;
    const int m_begin = blockIdx.y * m_chunk;
    const int m_end = min(nm_pad, m_begin + m_chunk);
    const int n_tiles = (m_end - m_begin) / AM;
    // Map tiles go global -> LDS directly (global_load_lds, no staging registers: this kernel lives at the
    // register cap), one tile ahead into the other buffer, so a tile's HBM/L2 latency hides behind the MFMAs of
    // the tile before it; one barrier per tile.  An LDS-DMA instruction writes 64 lanes x 16 B contiguously, so
    // the LDS image is linear (4 rows of 256 B per instruction) and the bank-conflict-free layout comes from
    // swizzling the SOURCE: 16-byte chunk c of row r is stored at chunk position c ^ (r & 15).
    auto glds_tile = [&](int k, int buf) {
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int piece = pass * 4 + wave;                // 1 KB piece of the 16 KB tile
            const int e = piece * 64 + lane, row = e >> 4, pos = e & 15;
            const int8_t* src = mx + (size_t)(m_begin + k * AM + row) * 256 + 16 * (pos ^ (row & 15));
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(tile + buf * (AM * 256) + piece * 1024),
                                             16, 0, 0);
        }
    };
    if (n_tiles > 0) {
        glds_tile(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    const v16i zero = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    for (int k = 0; k < n_tiles; ++k) {
        const int8_t* cur = tile + (k & 1) * (AM * 256);
        if (k + 1 < n_tiles) glds_tile(k + 1, (k + 1) & 1);     // every wave left that buffer at the last barrier
        const int t = 2 * k;                                      // 32-column block counter
        v4i Bf[2][8];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int row = cb * 32 + r32, c = 2 * s + half;
                Bf[cb][s] = *reinterpret_cast<const v4i*>(cur + row * 256 + ((c ^ (row & 15)) << 4));
            }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int tt = t + cb;
            v4i BX;
            BX[0] = half == 0 ? (((-(tt >> 6)) & 0xff) | (((-(tt & 63)) & 0xff) << 8)) : 0;
            BX[1] = BX[2] = BX[3] = 0;
            v16i acc[QB];
#pragma unroll
            for (int b = 0; b < QB; ++b) acc[b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(AX, BX, zero, 0, 0, 0);
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int b = 0; b < QB; ++b) acc[b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[b][s], Bf[cb][s], acc[b], 0, 0, 0);
#pragma unroll
            for (int b = 0; b < QB; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) running[b][r] = max(running[b][r], acc[b][r]);
        }
        // the next tile has landed (this wave's pieces) and every wave is done with the current one
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    // key -> (distance, column): dot = ceil(key / 4096), t = 4096 * dot - key; this lane's column inside block t
    // is r32.  Padding columns (>= nm; their rows are zero, i.e. "distance 128") are dropped here: a padding
    // column can only have displaced candidates with a negative dot, which are beyond 128 and never reported.
#pragma unroll
    for (int b = 0; b < QB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = running[b][r];
            int v = 0x7fffffff;
            if (key != (int)0x80000000) {
                const int dot4096 = (key + 4095) & ~4095;
                const int col = m_begin + 32 * (dot4096 - key) + r32;
                if (col < nm) v = (((256 << 12) - dot4096) << 9) | col;      // (256 - dot) << 21 | col
            }
#pragma unroll
            for (int d = 16; d >= 1; d >>= 1) v = min(v, __shfl_xor(v, d));
            if (r32 == 0) {
                int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                int q = q0 + 32 * b + row;
                if (q < nq) atomicMin(best + q, (unsigned int)v);
            }
        }
}

__global__ void k_assoc_finish(const unsigned int* __restrict__ best, int nq, int32_t* __restrict__ idx,
                               float* __restrict__ dist)
{
    int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    unsigned int v = best[q];
    int ham = (int)(v >> 22);
    if (v == 0x7fffffffu || ham > 128) { idx[q] = -1; dist[q] = -1.f; }
    else { idx[q] = (int)(v & 0x1fffffu); dist[q] = (float)ham; }
}

__global__ void k_fill_u32(unsigned int* p, int n, unsigned int v)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

void launch_assoc(const uint8_t* q, int nq, const uint8_t* m, int nm, int8_t* qx, int8_t* mx,
                  unsigned int* best, int32_t* idx, float* dist, hipStream_t s)
{
    const int nq_pad = (nq + AQ - 1) / AQ * AQ, nm_pad = (nm + AM - 1) / AM * AM;
    hipLaunchKernelGGL(k_assoc_pack, dim3(((size_t)nq_pad * 32 + 255) / 256), dim3(256), 0, s, q, nq, nq_pad, qx);
    hipLaunchKernelGGL(k_assoc_pack, dim3(((size_t)nm_pad * 32 + 255) / 256), dim3(256), 0, s, m, nm, nm_pad, mx);
    hipLaunchKernelGGL(k_fill_u32, dim3((nq + 255) / 256), dim3(256), 0, s, best, nq, 0x7fffffffu);
    const int qblocks = nq_pad / AQ;
    // 2 workgroups are resident per CU (234 VGPRs): split the map so that the grid is just under two full
    // rounds of the 512 slots -- long chunks amortise the A-fragment loads and the final cross-lane reduction
    int splits = (2 * 512) / qblocks;
    const int tiles = nm_pad / AM;
    const int min_splits = (nm_pad + (4096 * 32) - 1) / (4096 * 32);     // the in-accumulator block counter has 12 bits
    if (splits < min_splits) splits = min_splits;
    if (splits > tiles) splits = tiles;
    if (splits < 1) splits = 1;
    const int m_chunk = (tiles + splits - 1) / splits * AM;
    splits = (nm_pad + m_chunk - 1) / m_chunk;
    hipLaunchKernelGGL(k_assoc, dim3(qblocks, splits), dim3(256), 0, s, qx, nq, mx, nm, nm_pad, m_chunk, best);
    hipLaunchKernelGGL(k_assoc_finish, dim3((nq + 255) / 256), dim3(256), 0, s, best, nq, idx, dist);
}

// ---------------------------------------------------------------- float LBD (72-d)
// dist^2 = |q|^2 + |m|^2 - 2 q.m ; the dot product runs on the fp32-input MFMA (K = 2 per
// instruction, 36 steps).  One wave = 32 queries x 32 map entries per step.
__global__ void k_sqnorm72(const float* __restrict__ x, int n, float* __restrict__ out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0;
    for (int k = 0; k < 72; ++k) { float v = x[(size_t)i * 72 + k]; s += v * v; }
    out[i] = s;
}

__global__ __launch_bounds__(256) void k_assoc_float(const float* __restrict__ q, const float* __restrict__ qn, int nq,
                                                     const float* __restrict__ m, const float* __restrict__ mn, int nm,
                                                     int m_chunk, unsigned long long* __restrict__ best)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q0 = (blockIdx.x * 4 + wave) * 32;
    if (q0 >= nq) return;
    const int r32 = lane & 31, half = lane >> 5;
    const int qi = min(q0 + r32, nq - 1);
    float A[36];
#pragma unroll
    for (int s = 0; s < 36; ++s) A[s] = q[(size_t)qi * 72 + 2 * s + half];
    float qnr[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        qnr[r] = qn[min(q0 + row, nq - 1)];
    }
    unsigned long long running[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) running[r] = ~0ull;
    const int m_begin = blockIdx.y * m_chunk, m_end = min(nm, m_begin + m_chunk);
    for (int m0 = m_begin; m0 < m_end; m0 += 32) {
        const int col = m0 + r32;
        const int ci = min(col, nm - 1);
        v16f acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 36; ++s) {
            float B = m[(size_t)ci * 72 + 2 * s + half];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[s], B, acc, 0, 0, 0);
        }
        const float mnc = mn[ci];
        const bool valid = col < m_end;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float d2 = (qnr[r] + mnc) - 2.f * acc[r];
            d2 = d2 < 0.f ? 0.f : d2;
            unsigned long long v = valid ? (((unsigned long long)__float_as_uint(d2) << 32) | (unsigned int)col) : ~0ull;
            running[r] = v < running[r] ? v : running[r];
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        unsigned long long v = running[r];
#pragma unroll
        for (int d = 16; d >= 1; d >>= 1) {
            unsigned long long o = __shfl_xor(v, d);
            v = o < v ? o : v;
        }
        if (r32 == 0) {
            int row = (r & 3) + 8 * (r >> 2) + 4 * half;
            int qq = q0 + row;
            if (qq < nq) atomicMin(best + qq, v);
        }
    }
}

__global__ void k_assoc_float_finish(const unsigned long long* __restrict__ best, int nq, int32_t* __restrict__ idx,
                                     float* __restrict__ dist)
{
    int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    unsigned long long v = best[q];
    if (v == ~0ull) { idx[q] = -1; dist[q] = -1.f; }
    else { idx[q] = (int)(v & 0xffffffffull); dist[q] = dm::fsqrt(__uint_as_float((unsigned int)(v >> 32))); }
}

__global__ void k_fill_u64(unsigned long long* p, int n, unsigned long long v)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

void launch_assoc_float(const float* q, int nq, const float* m, int nm, float* qn, float* mn,
                        unsigned long long* best, int32_t* idx, float* dist, hipStream_t s)
{
    hipLaunchKernelGGL(k_sqnorm72, dim3((nq + 255) / 256), dim3(256), 0, s, q, nq, qn);
    hipLaunchKernelGGL(k_sqnorm72, dim3((nm + 255) / 256), dim3(256), 0, s, m, nm, mn);
    hipLaunchKernelGGL(k_fill_u64, dim3((nq + 255) / 256), dim3(256), 0, s, best, nq, ~0ull);
    const int qblocks = (nq + 127) / 128;
    int splits = (1024 + qblocks - 1) / qblocks;
    const int tiles = (nm + 31) / 32;
    if (splits > tiles) splits = tiles;
    if (splits < 1) splits = 1;
    const int m_chunk = (tiles + splits - 1) / splits * 32;
    splits = (nm + m_chunk - 1) / m_chunk;
    hipLaunchKernelGGL(k_assoc_float, dim3(qblocks, splits), dim3(256), 0, s, q, qn, nq, m, mn, nm, m_chunk, best);
    hipLaunchKernelGGL(k_assoc_float_finish, dim3((nq + 255) / 256), dim3(256), 0, s, best, nq, idx, dist);
}

}  // namespace lf
