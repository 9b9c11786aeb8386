// Associator (a-10 / a-11): nearest neighbour of every query descriptor in the live map.
//
// Reference semantics: BinaryDescriptorMatcher::match
// (/root/reference/src/line_descriptor/src/binary_descriptor_matcher.cpp:197-254) with
// Mihasher(256, 32), K = 1 (:635-753) and the popcount Hamming distance of
// bitops_custom.hpp:83-96: the exact Hamming nearest neighbour; candidates farther than
// D = 128 are never reported (:721).  The reference's multi-index hash is a CPU
// pointer-chasing structure rebuilt on every call; on MI355X the N x M distance matrix is
// one dense int8 contraction: query bits -> +-32, map bits -> +-16, so dot / 512 = 256 - 2*hamming,
// exactly.
//
// Packed operands (kept by the live map across calls, k_map.hip packs only the rows that change):
//   qx / mx   [rows][256] int8   the code, one byte per bit
//   qcx / mcx [rows][32]  int8   the operands of a NINTH MFMA step (K = 32) that folds two more terms into
//                                the accumulator:
//        k = 0, 1      query [16, 1]  x  map [-(t >> 4), -(t & 15)]  = -t, the number of the 32-column block inside
//                      the workgroup's map chunk (written by the kernel, the bytes are zero in memory)
//        k = 2 .. 31   three groups of ten, one per colour: query 127 in the group of ITS colour (only when colour
//                      gating is on), map -127 in the groups of the OTHER colours -> -161 290 when the colours
//                      differ, which is below every candidate within distance 128 (key >= -511)
//   key = 512 * dot - t - penalty orders candidates by distance, then by column block, so the running arg-max is
//   ONE v_max per accumulator register and colour gating (SURVEY a-11) costs nothing: it rides in the matrix core.
// k_assoc      : 256 queries per workgroup (4 waves x 2 x 32 rows, A fragments resident in VGPRs; a B fragment
//   read from LDS feeds two MFMAs), map streamed through LDS in 64-entry tiles by LDS-DMA (double buffered,
//   source-swizzled: conflict-free ds_read_b128), v_mfma_i32_32x32x32_i8 over K = 256 + 32; ties resolve to the
//   lowest map index.  Map chunks are spread over gridDim.y and merged with atomicMin on the packed word.
//   Algorithmic ops: 2*N*M*256 int8.
// k_assoc_float: 72-d float LBD, Euclidean, on v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain).
#include <cstdlib>
#include "common.h"

namespace lf {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int AQ = 512;        // query rows are padded to this (both kernel shapes divide it)
constexpr int AM = 64;         // map entries per LDS tile
constexpr int kMaxBlocksPerChunk = 512;   // the in-accumulator block counter t has 9 bits

// one thread per (row, code byte): 8 int8 = 2 dwords; thread 0 of a row also writes the row's ninth-step operand.
// side 0: query (+-32, [16, 1] counter weights, +127 in the own colour's group when gating)
// side 1: map   (+-16, zero counter bytes, -127 in the other colours' groups; colour >= 3 matches every colour)
__global__ void k_assoc_pack(const uint8_t* __restrict__ codes, const uint8_t* __restrict__ colors, int side, int gating,
                             int n, int n_pad, int8_t* __restrict__ out, int8_t* __restrict__ outc)
{
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t total = (size_t)n_pad * 32;
    if (t >= total) return;
    size_t row = t >> 5;
    uint32_t lo = 0, hi = 0;
    const bool live = row < (size_t)n;
    if (live) {
        uint32_t b = codes[t];
        // nibble -> 4 bytes of 0/1, then 0 -> +mag, 1 -> -mag
        uint32_t w0 = ((b & 15u) * 0x00204081u) & 0x01010101u;
        uint32_t w1 = ((b >> 4) * 0x00204081u) & 0x01010101u;
        if (side == 0) { lo = (w0 * 0xC0u) ^ 0x20202020u; hi = (w1 * 0xC0u) ^ 0x20202020u; }
        else { lo = (w0 * 0xE0u) ^ 0x10101010u; hi = (w1 * 0xE0u) ^ 0x10101010u; }
    }
    // queries: row major [row][256]; map: blocked by LDS tile (common.h assoc_map_offset)
    *reinterpret_cast<uint2*>(out + (side == 0 ? t * 8 : assoc_map_offset(row, (int)(t & 31) * 8))) = make_uint2(lo, hi);
    if ((t & 31) == 0) {
        uint32_t w[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        if (live) {
            const int c = colors ? colors[row] : 255;
            uint8_t* wb = reinterpret_cast<uint8_t*>(w);
            if (side == 0) {
                wb[0] = 16; wb[1] = 1;
                if (gating && c < 3) for (int k = 0; k < 10; ++k) wb[2 + 10 * c + k] = 127;
            } else if (c < 3) {
                for (int g = 0; g < 3; ++g)
                    if (g != c) for (int k = 0; k < 10; ++k) wb[2 + 10 * g + k] = (uint8_t)(-127);
            }
        }
        uint4* o = reinterpret_cast<uint4*>(outc + row * 32);
        o[0] = make_uint4(w[0], w[1], w[2], w[3]);
        o[1] = make_uint4(w[4], w[5], w[6], w[7]);
    }
}

// QB = 32-query row blocks per wave.  Two shapes:
//   QB = 2  256 queries per workgroup, 234 VGPRs, two workgroups per CU (small query counts: more workgroups)
//   QB = 4  512 queries per workgroup, one wave per SIMD with the whole register file: every B fragment read from LDS
//           feeds FOUR MFMAs (half the LDS reads and half the LDS-DMA traffic per MFMA, 72 MFMAs between barriers),
//           and a wave never shares its matrix core with a wave of another workgroup whose barrier phase differs
template <int QB, bool GATED>
__device__ __forceinline__ void assoc_body(const int8_t* __restrict__ qx, const int8_t* __restrict__ qcx, int nq,
                                           const int8_t* __restrict__ mx, const int8_t* __restrict__ mcx,
                                           int nm_bound, const int* __restrict__ nm_dev, int nm_pad, int m_chunk,
                                           unsigned int* __restrict__ best, int8_t* tile, int8_t* ctile)
{
    // nm_bound sized the grid on the host; when the exact size is only known on the device (the live map's size
    // after an update still in flight) it is read here.  Rows in [size, bound) are all-zero operands and are dropped
    // below exactly like padding rows, so the result does not depend on how loose the bound was.
    const int nm = nm_dev ? min(nm_bound, *nm_dev) : nm_bound;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q0 = blockIdx.x * (128 * QB) + wave * (32 * QB);
    const int r32 = lane & 31, half = lane >> 5;
    // QB row blocks of 32 queries per wave: every B fragment read from LDS feeds QB MFMAs
    v4i A[QB][8], AX[QB];
#pragma unroll
    for (int b = 0; b < QB; ++b) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
            A[b][s] = *reinterpret_cast<const v4i*>(qx + (size_t)(q0 + 32 * b + r32) * 256 + 32 * s + 16 * half);
        if (GATED) AX[b] = *reinterpret_cast<const v4i*>(qcx + (size_t)(q0 + 32 * b + r32) * 32 + 16 * half);
    }
    // Without colour gating the ninth MFMA step is not needed at all: the block counter enters as the chain's START
    // value (the C operand of the first MFMA), a register set that is decremented once per block -- 16 vector adds
    // that issue beside the MFMAs instead of one more MFMA per accumulator (8 instead of 9 matrix steps per block).
    v16i negt = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    if (!GATED) {
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(negt[r]));      // a vector, not 16 copies of a scalar
    }
    // Arg-max inside the matrix core (see the header): after the chain of 9 MFMAs the accumulator holds
    // key = 512 * dot - t - colour penalty, so the whole epilogue is ONE v_max per accumulator register: no zeroing
    // (the chain starts from the inline constant 0), no packing, no select.  dot and t are recovered from the key at
    // the very end.
    int running[QB][16];
#pragma unroll
    for (int b = 0; b < QB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) running[b][r] = (int)0x80000000;
    const int m_begin = blockIdx.y * m_chunk;
    const int m_end = min(nm_pad, m_begin + m_chunk);
    const int n_tiles = (m_end - m_begin) / AM;
    const unsigned tmask = half == 0 ? 0xffffu : 0u;          // the counter bytes live in k = 0, 1 (lanes 0..31)
    // Map tiles go global -> LDS directly (global_load_lds, no staging registers: this kernel lives at the register
    // cap).  The packed map is stored BLOCKED by tile in memory -- [tile of 64 rows][16-byte chunk c][row][16 B],
    // common.h assoc_map_offset -- so a tile is 16 KB of contiguous memory that goes to LDS as it is (an LDS-DMA
    // instruction writes 64 lanes x 16 B contiguously), and the 32 lanes of a half wave read 512 contiguous bytes per
    // fragment: no bank conflicts, no swizzle, one address register and immediate offsets on both sides.
    const uint32_t lane16 = (uint32_t)lane * 16u;
    auto glds_tile = [&](int k, int buf) {
        const int8_t* tbase = mx + (size_t)(m_begin + k * AM) * 256;          // uniform
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int piece = pass * 4 + wave;                // 1 KB piece of the 16 KB tile = chunk `piece` of all 64 rows
            const int8_t* src = tbase + piece * 1024 + lane16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(tile + buf * (AM * 256) + piece * 1024),
                                             16, 0, 0);
        }
        if (GATED && wave < 2) {
            const int8_t* src = mcx + (size_t)(m_begin + k * AM) * 32 + wave * 1024 + lane16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(ctile + buf * (AM * 32) + wave * 1024),
                                             16, 0, 0);
        }
    };
    // Software pipeline over HALF tiles (32 map rows = one B fragment set of 8 x 16 B per lane):
    //   step 2k    : read rows 32..63 of tile k into Bf[1]        | MFMAs on Bf[0] (rows 0..31 of tile k)
    //                wait: tile k + 1 landed, own reads of tile k done; BARRIER  -> tile k's buffer is free
    //                LDS-DMA of tile k + 3 into that buffer
    //   step 2k+1  : read rows 0..31 of tile k + 1 into Bf[0]      | MFMAs on Bf[1]
    // so a wave's LDS reads always run under its own MFMAs of the other half (the reads of all eight waves of a CU
    // after a barrier take ~500 cycles of LDS bandwidth: in front of the MFMAs they were dead time), and a tile has two
    // whole iterations to land.  Tile indices past the chunk are clamped (a redundant re-load of the last tile keeps
    // the outstanding-load count uniform: 4 LDS-DMA instructions per tile and wave, 5 with the ninth-step rows).
    const int last_tile = n_tiles - 1;
    // ONE fragment set: fragment s of the next half is read into the registers of fragment s of the current half as soon
    // as the MFMAs that consume it have issued, so it has 7/8 of a half step to arrive and the set costs 32 VGPRs, not 64.
    v4i Bf[8], BX;
    const int frag0 = half * 1024 + r32 * 16;                  // this lane inside chunk 2 s + half, row r32 (+ 32 cb)
    auto read_frag = [&](const int8_t* buf, int cb, int s) {
        Bf[s] = *reinterpret_cast<const v4i*>(buf + frag0 + s * 2048 + cb * 512);
    };
    auto read_x = [&](const int8_t* bufc, int cb) { BX = *reinterpret_cast<const v4i*>(bufc + (cb * 32 + r32) * 32 + 16 * half); };
    const v16i zero = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    // The arg-max of a half (one v_max per accumulator register) is folded in while the NEXT half's MFMAs run: two
    // accumulator sets, so no MFMA result is waited for and a wave's stream is MFMA, MFMA, LDS read, a few vector ops, ...
    // (the sched_group_barrier pattern) instead of a burst of 16 MFMAs followed by a burst of vector ops behind the
    // last MFMA's latency.
    v16i accs[2][QB];
#pragma unroll
    for (int b = 0; b < QB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) accs[1][b][r] = (int)0x80000000;
    // half step cb of a tile: MFMAs on the fragments in registers, which are refreshed from rows [32 ncb, 32 ncb + 32) of nbuf
    auto half_step = [&](int cb, int tt, const int8_t* nbuf, const int8_t* nbufc, int ncb) {
        if (GATED) {
            const unsigned tbytes = ((unsigned)(-(tt >> 4)) & 0xffu) | (((unsigned)(-(tt & 15)) & 0xffu) << 8);
            v4i bx = BX;
            bx[0] |= (int)(tbytes & tmask);
#pragma unroll
            for (int b = 0; b < QB; ++b) accs[cb][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(AX[b], bx, zero, 0, 0, 0);
            read_x(nbufc, ncb);
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#pragma unroll
            for (int b = 0; b < QB; ++b)
                accs[cb][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[b][s], Bf[s], (!GATED && s == 0) ? negt : accs[cb][b], 0, 0, 0);
            read_frag(nbuf, ncb, s);
        }
        if (!GATED) {
#pragma unroll
            for (int r = 0; r < 16; ++r) negt[r] -= 1;
        }
        // the OTHER half's keys (previous step)
#pragma unroll
        for (int b = 0; b < QB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                running[b][r] = max(running[b][r], accs[cb ^ 1][b][r]);
                // keep this v_max in THIS half step: fused with the next half's into one v_max3 it would need both
                // accumulator sets complete at once, which is exactly the wait this loop is built to avoid
                asm("" : "+v"(running[b][r]));
            }
        // issue order
#pragma unroll
        for (int i = 0; i < 8 + (GATED ? 1 : 0); ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, QB, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2 * QB, 0);
        }
    };
    if (n_tiles > 0) {
        glds_tile(0, 0);
        glds_tile(min(1, last_tile), 1);
        glds_tile(min(2, last_tile), 2);
        if (GATED && wave < 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int s = 0; s < 8; ++s) read_frag(tile, 0, s);
        if (GATED) read_x(ctile, 0);
    }
    int bcur = 0;
    for (int k = 0; k < n_tiles; ++k) {
        const int8_t* cur = tile + bcur * (AM * 256);
        const int8_t* curc = ctile + bcur * (AM * 32);
        const int bnext = bcur == 2 ? 0 : bcur + 1;
        half_step(0, 2 * k, cur, curc, 1);
        // tile k + 1 has landed (this wave's pieces; tile k + 2 may still be in flight), this wave holds the rest of tile k in registers
        if (GATED && wave < 2) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        glds_tile(min(k + 3, last_tile), bcur);
        half_step(1, 2 * k + 1, tile + bnext * (AM * 256), ctile + bnext * (AM * 32), 0);
        bcur = bnext;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int b = 0; b < QB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) running[b][r] = max(running[b][r], accs[1][b][r]);
    // key -> (distance, column): 512 * dot = ceil(key / 512) * 512, t = 512 * dot - key; this lane's column inside
    // block t is r32.  Padding columns (>= nm; their rows are zero, i.e. "distance 128") are dropped here: a padding
    // column can only have displaced candidates with a negative dot, which are beyond 128 and never reported.  A
    // candidate of another colour (gating) decodes to a distance beyond 128 and is dropped by k_assoc_finish.
#pragma unroll
    for (int b = 0; b < QB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = running[b][r];
            int v = 0x7fffffff;
            if (key != (int)0x80000000) {                             // this lane saw at least one column block
                const int dot512 = (key + 511) & ~511;
                const int col = m_begin + 32 * (dot512 - key) + r32;
                if (col < nm) v = (((256 << 9) - dot512) << 12) | col;      // hamming << 22 | col
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

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_assoc(const int8_t* __restrict__ qx, const int8_t* __restrict__ qcx, int nq,
                                               const int8_t* __restrict__ mx, const int8_t* __restrict__ mcx,
                                               int nm_bound, const int* __restrict__ nm_dev, int nm_pad, int m_chunk,
                                               unsigned int* __restrict__ best)
{
    __shared__ __attribute__((aligned(1024))) int8_t tile[3 * AM * 256];     // triple buffered map tile, linear rows
    __shared__ __attribute__((aligned(1024))) int8_t ctile[3 * AM * 32];     // the tiles' ninth-step operands
    assoc_body<2, true>(qx, qcx, nq, mx, mcx, nm_bound, nm_dev, nm_pad, m_chunk, best, tile, ctile);
}

// the same without colour gating: 8 MFMA steps per block, no ninth-step operands streamed
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_assoc_plain(const int8_t* __restrict__ qx, const int8_t* __restrict__ qcx, int nq,
                                               const int8_t* __restrict__ mx, const int8_t* __restrict__ mcx,
                                               int nm_bound, const int* __restrict__ nm_dev, int nm_pad, int m_chunk,
                                               unsigned int* __restrict__ best)
{
    __shared__ __attribute__((aligned(1024))) int8_t tile[3 * AM * 256];
    assoc_body<2, false>(qx, qcx, nq, mx, mcx, nm_bound, nm_dev, nm_pad, m_chunk, best, tile, nullptr);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_assoc4(const int8_t* __restrict__ qx, const int8_t* __restrict__ qcx, int nq,
                                               const int8_t* __restrict__ mx, const int8_t* __restrict__ mcx,
                                               int nm_bound, const int* __restrict__ nm_dev, int nm_pad, int m_chunk,
                                               unsigned int* __restrict__ best)
{
    __shared__ __attribute__((aligned(1024))) int8_t tile[3 * AM * 256];
    __shared__ __attribute__((aligned(1024))) int8_t ctile[3 * AM * 32];
    assoc_body<4, true>(qx, qcx, nq, mx, mcx, nm_bound, nm_dev, nm_pad, m_chunk, best, tile, ctile);
}

__global__ void k_assoc_finish(const unsigned int* __restrict__ best, int nq, int max_distance, int32_t* __restrict__ idx,
                               float* __restrict__ dist)
{
    int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    unsigned int v = best[q];
    int ham = (int)(v >> 22);
    if (v == 0x7fffffffu || ham > max_distance) { idx[q] = -1; dist[q] = -1.f; }
    else { idx[q] = (int)(v & 0x1fffffu); dist[q] = (float)ham; }
}

__global__ void k_fill_u32(unsigned int* p, int n, unsigned int v)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ void k_fill_nomatch(int n, int32_t* idx, float* dist)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { idx[i] = -1; dist[i] = -1.f; }
}

void launch_assoc_nomatch(int nq, int32_t* idx, float* dist, hipStream_t s)
{
    if (nq > 0) hipLaunchKernelGGL(k_fill_nomatch, dim3((nq + 255) / 256), dim3(256), 0, s, nq, idx, dist);
}

size_t assoc_rows_padded_q(int nq) { return ((size_t)nq + AQ - 1) / AQ * AQ; }
size_t assoc_rows_padded_m(int nm) { return ((size_t)nm + AM - 1) / AM * AM; }

void launch_assoc_pack(const uint8_t* codes, const uint8_t* colors, int side, int gating, int n, int n_pad, int8_t* x,
                       int8_t* cx, hipStream_t s)
{
    if (n_pad <= 0) return;
    hipLaunchKernelGGL(k_assoc_pack, dim3(((size_t)n_pad * 32 + 255) / 256), dim3(256), 0, s, codes, colors, side, gating, n,
                       n_pad, x, cx);
}

// association of packed queries against a packed map (the live map keeps its side packed across calls;
// lf_associate packs its caller's raw map first)
void launch_assoc_core(const int8_t* qx, const int8_t* qcx, int nq, const int8_t* mx, const int8_t* mcx, int nm,
                       const int* nm_dev, int gating, int max_distance, unsigned int* best, int32_t* idx, float* dist, hipStream_t s)
{
    const int nq_pad = (int)assoc_rows_padded_q(nq), nm_pad = (int)assoc_rows_padded_m(nm);
    hipLaunchKernelGGL(k_fill_u32, dim3((nq + 255) / 256), dim3(256), 0, s, best, nq, 0x7fffffffu);
    const int tiles = nm_pad / AM;
    const int min_splits = (nm_pad + (kMaxBlocksPerChunk * 32) - 1) / (kMaxBlocksPerChunk * 32);
    // Shape: 512-query workgroups (one per CU, QB = 4) once there are enough queries to give every CU a chunk of at
    // least 16 tiles; below that the 256-query shape, two workgroups per CU.
    const int qb4 = (nq_pad + 511) / 512;
    int sp4 = 256 / qb4;
    if (sp4 < 1) sp4 = 1;
    // measured on MI355X: the 512-query shape is SLOWER (34 % vs 40 % of the int8 peak at 16 k x 50 k): with one wave per
    // SIMD nothing covers a wave's LDS reads and barrier waits.  Kept for experiments (LF_ASSOC_QB4=1), not used.
    static const bool allow_big = getenv("LF_ASSOC_QB4") != nullptr;
    const bool big = allow_big && qb4 >= 4 && tiles / sp4 >= 16;
    if (big) {
        // one round of (at most) 256 workgroups; more query blocks than CUs: whole rounds
        int splits = sp4;
        if (splits < min_splits) splits = min_splits;
        if (splits > tiles) splits = tiles;
        const int m_chunk = (tiles + splits - 1) / splits * AM;
        splits = (nm_pad + m_chunk - 1) / m_chunk;
        hipLaunchKernelGGL(k_assoc4, dim3(qb4, splits), dim3(256), 0, s, qx, qcx, nq, mx, mcx, nm, nm_dev, nm_pad, m_chunk, best);
    } else {
        const int qblocks = nq_pad / 256;
        // 2 workgroups are resident per CU: split the map so that the grid is just under two full rounds of the 512
        // slots -- long chunks amortise the A-fragment loads and the final cross-lane reduction
        static const int slots = getenv("LF_ASSOC_SLOTS") ? atoi(getenv("LF_ASSOC_SLOTS")) : 512;   // one round of two workgroups per CU (measured best: 512 > 1024 > 768)
        int splits = slots / qblocks;
        if (splits < min_splits) splits = min_splits;
        if (splits > tiles) splits = tiles;
        if (splits < 1) splits = 1;
        const int m_chunk = (tiles + splits - 1) / splits * AM;
        splits = (nm_pad + m_chunk - 1) / m_chunk;
        if (gating) hipLaunchKernelGGL(k_assoc, dim3(qblocks, splits), dim3(256), 0, s, qx, qcx, nq, mx, mcx, nm, nm_dev, nm_pad, m_chunk, best);
        else hipLaunchKernelGGL(k_assoc_plain, dim3(qblocks, splits), dim3(256), 0, s, qx, qcx, nq, mx, mcx, nm, nm_dev, nm_pad, m_chunk, best);
    }
    hipLaunchKernelGGL(k_assoc_finish, dim3((nq + 255) / 256), dim3(256), 0, s, best, nq, max_distance, idx, dist);
}

void launch_assoc(const uint8_t* q, int nq, const uint8_t* m, int nm, int8_t* qx, int8_t* qcx, int8_t* mx, int8_t* mcx,
                  unsigned int* best, int32_t* idx, float* dist, hipStream_t s)
{
    launch_assoc_pack(m, nullptr, 1, 0, nm, (int)assoc_rows_padded_m(nm), mx, mcx, s);
    launch_assoc_pack(q, nullptr, 0, 0, nq, (int)assoc_rows_padded_q(nq), qx, qcx, s);
    launch_assoc_core(qx, qcx, nq, mx, mcx, nm, nullptr, 0, 128, best, idx, dist, s);
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
