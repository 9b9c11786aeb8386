// The matcher's tie rule (a-10 (iii)): WHICH of several equally near map codes BinaryDescriptorMatcher::match returns.
//
// Reference: Mihasher::query (/root/reference/src/line_descriptor/src/binary_descriptor_matcher.cpp:635-753) with
// B = 256, m = 32 eight-bit substrings, K = 1.  It enumerates, for radius s = 0, 1, .. and substring k = 0 .. 31, the buckets
// H[k][chunk_k(query) ^ bitstr] over every bitstr of weight s in the order of its combination loop (:681-741); a bucket
// lists its codes in insertion = train order (:806-819, :927-947).  The first time an index shows up its full distance is
// taken and the FIRST index seen per distance is kept (:716-722); the search stops after (s, k) once a code at distance
// s * 32 + k has been seen (:744-746), by when every code that near has been, so the result is the exact nearest
// neighbour and, among equally near ones, the one discovered first.  A candidate's discovery time is a function of the
// pair alone:   key = min over substrings k with h_k = popcount(q_k ^ c_k) <= 4 of (h_k, k, position of q_k ^ c_k in the
// enumeration of the weight-h_k strings),   then train index.  (A code within 128 bits always has a substring within 4.)
//
// On the MI355X the distance pass (k_assoc.hip) has already produced every query's minimum distance.  This second pass
// recomputes the N x M dot products on the FP4 matrix instruction from the SAME packed map operands, compares every
// accumulator value with its row's known optimum (one v_cmp per register, masks OR-ed on the scalar side), and only the
// (rare) equal ones -- the ties -- leave the matrix pipeline: their discovery key is worked out from the query's raw code and
// the map row's nibbles, both in LDS, and folded with a 64-bit atomic minimum (key << 32 | index), first in LDS, then once per query and map chunk in memory.
#include "common.h"
#include "mih_rank.h"

namespace lf {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__constant__ MihRank c_mih_rank = make_mih_rank();

void mih_rank_host(uint8_t out[5][256])
{
    constexpr MihRank t = make_mih_rank();
    for (int s = 0; s < 5; ++s) for (int i = 0; i < 256; ++i) out[s][i] = t.r[s][i];
}

constexpr int TQW = 256;          // queries per workgroup: 4 waves x 2 row blocks x 32
constexpr int TGROUP = 4;         // 64-row map tiles (8 KB of e2m1 rows each) per LDS buffer: a group's matrix work (~1.5 us) covers the next group's fetch
constexpr int TSETS = TGROUP * 4; // accumulator sets per group: tile x column block x row block

// eight e2m1 nibbles (0x2 = bit 0, 0xA = bit 1) -> the code byte they were expanded from (assoc_fp4_expand's inverse)
__device__ __forceinline__ uint32_t fp4_collapse(uint32_t w)
{
    uint32_t t = (w >> 3) & 0x11111111u;
    t = (t | (t >> 3)) & 0x03030303u;
    t = (t | (t >> 6)) & 0x000f000fu;
    return (t | (t >> 12)) & 0xffu;
}

// The lanes whose bit is set in `hits` hold, in accumulator register r, a candidate as near as its row's optimum: work out
// its discovery key -- the smallest (weight, substring) over the 32 byte substrings, then ONE table lookup for the place of that
// substring's difference in the enumeration -- from the query's raw code (LDS copy) and the map row's nibbles in the LDS tile,
// and fold it into the query's running minimum.  Rare; kept out of line so the matrix loop stays small.
template <bool GATED>
__device__ __noinline__ void ties_fold(uint32_t hits, int col, int col_in_tile, int row_base, const uint8_t* tile, const uint32_t* qraw,
                                       const uint8_t* __restrict__ qcolor, const uint8_t* __restrict__ mcolor, int q_first, int nq,
                                       unsigned long long* s_res)
{
    while (hits) {
        const int r = __ffs(hits) - 1;
        hits &= hits - 1;
        const int ql = row_base + (r & 3) + 8 * (r >> 2);
        const int qg = q_first + ql;
        if (qg >= nq) continue;
        if (GATED) {
            const int qc = qcolor[qg], mc = mcolor[col];
            if (qc < 3 && mc < 3 && qc != mc) continue;
        }
        uint32_t best = 0xffffffffu, bx = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) {                            // 16-byte chunk c of the row's 128 bytes = code bytes 4 c .. 4 c + 3
            const uint4 w = *reinterpret_cast<const uint4*>(tile + c * 1024 + col_in_tile * 16);
            const uint32_t mb = fp4_collapse(w.x) | (fp4_collapse(w.y) << 8) | (fp4_collapse(w.z) << 16) | (fp4_collapse(w.w) << 24);
            const uint32_t x = mb ^ qraw[ql * 8 + c];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint32_t xb = (x >> (8 * t)) & 255u;
                const uint32_t hk = (uint32_t)__popc(xb) * 32u + (uint32_t)(4 * c + t);        // (weight, substring)
                if (hk < best) { best = hk; bx = xb; }
            }
        }
        const uint32_t h = best >> 5;
        if (h > 4) continue;                                     // cannot happen within 128 bits; the reference would never meet it
        const uint32_t key = (best << 8) | c_mih_rank.r[h][bx];
        atomicMin(&s_res[ql], ((unsigned long long)key << 32) | (uint32_t)col);
    }
}

template <bool GATED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_assoc_ties(const uint8_t* __restrict__ q, const uint8_t* __restrict__ qcolor, int nq,
                                                    const int8_t* __restrict__ mx, const uint8_t* __restrict__ mcolor,
                                                    int nm_bound, const int* __restrict__ nm_dev,
                                                    int nm_pad, int m_chunk, const float* __restrict__ dist,
                                                    unsigned long long* __restrict__ res)
{
    // two buffers of TGROUP map tiles, filled by LDS-DMA (global_load_lds: 64 lanes x 16 bytes = 1 KB of a tile per instruction
    // and wave, no staging registers); separate arrays and a loop unrolled by two, so that the compiler can tell the buffer being
    // filled from the one being read
    __shared__ __attribute__((aligned(1024))) uint8_t tiles_a[TGROUP * 8192];
    __shared__ __attribute__((aligned(1024))) uint8_t tiles_b[TGROUP * 8192];
    __shared__ __attribute__((aligned(16))) uint32_t qraw[TQW * 8];
    __shared__ unsigned long long s_res[TQW];
    __shared__ uint32_t xtab[256];
    const int nm = nm_dev ? min(nm_bound, *nm_dev) : nm_bound;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q0 = blockIdx.x * TQW + wave * 64;
    const int r32 = lane & 31, half = lane >> 5;
    const int m_begin = blockIdx.y * m_chunk;
    const int m_end = min(nm_pad, m_begin + m_chunk);
    const int n_tiles = __builtin_amdgcn_readfirstlane((m_end - m_begin) / 64);
    if (n_tiles <= 0 || m_begin >= nm) return;
    const uint8_t* src = reinterpret_cast<const uint8_t*>(mx) + (size_t)(m_begin / 64) * 8192;
    const int n_groups = (n_tiles + TGROUP - 1) / TGROUP;
    auto fetch = [&](int g, uint8_t* dst) {
#pragma unroll
        for (int k = 0; k < TGROUP * 2; ++k) {
            const int off = k * 4096 + wave * 1024;
            if (g * TGROUP + (off >> 13) < n_tiles)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)g * (TGROUP * 8192) + off + lane * 16),
                                                 (__attribute__((address_space(3))) void*)(dst + off), 16, 0, 0);
        }
    };
    fetch(0, tiles_a);
    xtab[threadIdx.x] = assoc_fp4_expand(threadIdx.x);
    s_res[threadIdx.x] = ~0ull;
    {
        // the workgroup's 256 raw query codes (8 KB): thread t copies query t
        const int qi = blockIdx.x * TQW + threadIdx.x;
        uint4 c0 = make_uint4(0, 0, 0, 0), c1 = c0;
        if (qi < nq) { c0 = *reinterpret_cast<const uint4*>(q + (size_t)qi * 32); c1 = *reinterpret_cast<const uint4*>(q + (size_t)qi * 32 + 16); }
        *reinterpret_cast<uint4*>(&qraw[threadIdx.x * 8]) = c0;
        *reinterpret_cast<uint4*>(&qraw[threadIdx.x * 8 + 4]) = c1;
    }
    __syncthreads();
    // query operands: step s, k-half `half` = code dword 2 s + half, eight e2m1 nibbles per code byte (k_assoc.hip)
    v8i A[2][4];
    // Every chain STARTS at 0.5 - (the dot product a tie has = 256 - 2 * the row's minimum distance), so a tie ends at exactly
    // +0.5, every other candidate at k + 0.5 with k a non-zero integer (all exact in f32), and rows without a match near -3e38.
    // Read as unsigned integers, +0.5 (0x3f000000) is then the SMALLEST value an accumulator can hold -- positive floats order
    // like their bit patterns, negative ones lie above them all -- so "is there a tie among these registers" is an unsigned
    // minimum (v_min3_u32: two registers per instruction) and one compare, instead of a compare and a scalar OR per register.
    v16f start[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const uint32_t* c = &qraw[(wave * 64 + 32 * b + r32) * 8];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const uint32_t w = c[2 * s + half];
            A[b][s] = v8i{ (int)xtab[w & 0xffu], (int)xtab[(w >> 8) & 0xffu], (int)xtab[(w >> 16) & 0xffu], (int)xtab[w >> 24], 0, 0, 0, 0 };
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qr = q0 + 32 * b + (r & 3) + 8 * (r >> 2) + 4 * half;
            const float d = qr < nq ? dist[qr] : -1.f;
            start[b][r] = d >= 0.f ? 0.5f - (256.f - 2.f * d) : -3.0e38f;
        }
    }
    // accumulator sets: (tile tl of the group, column block cb, row block b) -> 16 dot products per lane; the two row blocks of a
    // column block share its fragments and run as two interleaved chains
    auto dots2 = [&](const uint8_t* tiles, int tl, int cb, v16f& acc0, v16f& acc1) {
        const uint8_t* fb = tiles + tl * 8192 + half * 1024 + (cb * 32 + r32) * 16;
        acc0 = start[0]; acc1 = start[1];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const v4i f = *reinterpret_cast<const v4i*>(fb + s * 2048);
            const v8i B = v8i{ f.x, f.y, f.z, f.w, 0, 0, 0, 0 };
            acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A[0][s], B, acc0, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A[1][s], B, acc1, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
    };
    constexpr uint32_t kTie = 0x3f000000u;          // +0.5
    auto umin16 = [](const v16f& a, uint32_t m) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) m = min(m, min(__float_as_uint(a[r]), __float_as_uint(a[r + 1])));
        return m;
    };
    auto group = [&](int g, const uint8_t* tiles, uint8_t* next) {
        if (g + 1 < n_groups) fetch(g + 1, next);
        // the matrix loop proper: no branch inside -- which column blocks saw a tie anywhere in the wave is kept as a scalar bit mask
        uint32_t sets = 0;
#pragma unroll
        for (int pair = 0; pair < TSETS / 2; ++pair) {
            v16f acc0, acc1;
            dots2(tiles, pair >> 1, pair & 1, acc0, acc1);
            const uint32_t m = umin16(acc1, umin16(acc0, 0xffffffffu));
            sets |= (__builtin_amdgcn_ballot_w64(m == kTie) != 0 ? 3u : 0u) << (2 * pair);
        }
        if (g * TGROUP + TGROUP > n_tiles) sets &= (1u << (4 * (n_tiles - g * TGROUP))) - 1u;      // a group's tiles past the chunk's end
        if (sets) {
            // candidates as near as their row's optimum (rare): recompute the set, find the registers, fold the keys
#pragma unroll
            for (int set = 0; set < TSETS; ++set) {
                if (!(sets & (1u << set))) continue;
                const int tl = set >> 2, cb = (set >> 1) & 1, b = set & 1;
                v16f acc0, acc1;
                dots2(tiles, tl, cb, acc0, acc1);
                const v16f acc = b ? acc1 : acc0;
                uint32_t hits = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) hits |= (__float_as_uint(acc[r]) == kTie ? 1u : 0u) << r;
                const int col = m_begin + (g * TGROUP + tl) * 64 + cb * 32 + r32;
                if (col >= nm) hits = 0;
                if (__builtin_amdgcn_ballot_w64(hits != 0))
                    ties_fold<GATED>(hits, col, cb * 32 + r32, wave * 64 + 32 * b + 4 * half, tiles + tl * 8192, qraw, qcolor, mcolor,
                                     blockIdx.x * TQW, nq, s_res);
            }
        }
        __syncthreads();
    };
    for (int g = 0; g < n_groups; g += 2) {
        group(g, tiles_a, tiles_b);
        if (g + 1 < n_groups) group(g + 1, tiles_b, tiles_a);
    }
    const unsigned long long mine = s_res[threadIdx.x];
    const int qg = blockIdx.x * TQW + threadIdx.x;
    if (mine != ~0ull && qg < nq) atomicMin(res + qg, mine);
}

__global__ void k_assoc_ties_finish(int nq, const unsigned long long* __restrict__ res, int32_t* __restrict__ idx)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const unsigned long long v = res[i];
    if (v != ~0ull && idx[i] >= 0) idx[i] = (int32_t)(uint32_t)v;
}

// After launch_assoc_core on the same stream: replaces idx (the lowest index among the equally near) by the index the
// reference's search finds first.  mx: the map's packed e2m1 rows (the raw bits are read back from them); mcolor: the map's raw
// colours (gating only; mcode is not read).  res: nq u64.
hipError_t launch_assoc_ties(const uint8_t* q, const uint8_t* qcolor, int nq, const int8_t* mx, const uint8_t* mcode, const uint8_t* mcolor,
                             int nm, const int* nm_dev, int gating, unsigned long long* res, int32_t* idx, const float* dist, hipStream_t s)
{
    if (nq <= 0 || nm <= 0) return hipSuccess;
    const int nm_pad = (int)assoc_rows_padded_m(nm);
    const int tiles = nm_pad / 64;
    const int qblocks = (nq + TQW - 1) / TQW;
    // two workgroups fit a CU (32 KB of tiles each): about two rounds of the chip, chunks of whole tile groups
    int splits = 1024 / qblocks;
    if (splits < 1) splits = 1;
    if (splits > (tiles + TGROUP - 1) / TGROUP) splits = (tiles + TGROUP - 1) / TGROUP;
    int m_chunk = (tiles + splits - 1) / splits;
    m_chunk = (m_chunk + TGROUP - 1) / TGROUP * TGROUP * 64;
    splits = (nm_pad + m_chunk - 1) / m_chunk;
    hipError_t e = hipMemsetAsync(res, 0xff, (size_t)nq * sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    if (gating) hipLaunchKernelGGL(k_assoc_ties<true>, dim3(qblocks, splits), dim3(256), 0, s, q, qcolor, nq, mx, mcolor, nm, nm_dev, nm_pad, m_chunk, dist, res);
    else hipLaunchKernelGGL(k_assoc_ties<false>, dim3(qblocks, splits), dim3(256), 0, s, q, qcolor, nq, mx, mcolor, nm, nm_dev, nm_pad, m_chunk, dist, res);
    hipLaunchKernelGGL(k_assoc_ties_finish, dim3((nq + 255) / 256), dim3(256), 0, s, nq, res, idx);
    return hipGetLastError();
}

}  // namespace lf
