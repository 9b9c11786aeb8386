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
// (rare) equal ones -- the ties -- leave the matrix pipeline: their discovery key is worked out from the raw codes and
// folded with a 64-bit atomic minimum (key << 32 | index), first in LDS, then once per query and map chunk in memory.
#include "common.h"

namespace lf {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

// Position of every 8-bit string among the strings of its weight, in the order the reference's combination loop visits them
// (:681-741 with curb = 8).  up[i] is one more than the place of the i-th one (up[i] == i: not placed yet); the ones from
// `mv` down are placed / moved, the string is reported, then every one that touches the one above it (the top one: the
// end of the byte) is taken off again and the first that does not is the next to move up a place.
struct MihRank { uint8_t r[5][256]; };
constexpr MihRank make_mih_rank()
{
    MihRank t{};
    for (int s = 0; s <= 4; ++s) {
        int up[6] = { 0, 1, 2, 3, 4, 5 };
        up[s] = 9;
        unsigned str = 0;
        int order = 0, mv = s - 1;
        for (;;) {
            for (; mv >= 0; --mv) {
                str ^= up[mv] == mv ? 1u << up[mv] : 3u << (up[mv] - 1);
                ++up[mv];
            }
            t.r[s][str & 255u] = (uint8_t)order++;
            int b = 0;
            for (; b < s && up[b] == up[b + 1] - 1; ++b) { str ^= 1u << (up[b] - 1); up[b] = b; }
            if (b == s) break;
            mv = b;
        }
    }
    return t;
}
__constant__ MihRank c_mih_rank = make_mih_rank();

void mih_rank_host(uint8_t out[5][256])
{
    constexpr MihRank t = make_mih_rank();
    for (int s = 0; s < 5; ++s) for (int i = 0; i < 256; ++i) out[s][i] = t.r[s][i];
}

__device__ __noinline__ uint32_t mih_discovery_key(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b)
{
    const uint4 a0 = *reinterpret_cast<const uint4*>(a), a1 = *reinterpret_cast<const uint4*>(a + 16);
    const uint4 b0 = *reinterpret_cast<const uint4*>(b), b1 = *reinterpret_cast<const uint4*>(b + 16);
    const uint32_t x[8] = { a0.x ^ b0.x, a0.y ^ b0.y, a0.z ^ b0.z, a0.w ^ b0.w, a1.x ^ b1.x, a1.y ^ b1.y, a1.z ^ b1.z, a1.w ^ b1.w };
    uint32_t best = 0xffffffffu;
#pragma unroll
    for (int d = 0; d < 8; ++d)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t xb = (x[d] >> (8 * t)) & 255u;
            const int h = __popc(xb);
            if (h <= 4) {
                const uint32_t kk = ((uint32_t)(h * 32 + 4 * d + t) << 8) | c_mih_rank.r[h][xb];
                best = kk < best ? kk : best;
            }
        }
    return best;
}

constexpr int TQW = 256;          // queries per workgroup: 4 waves x 2 row blocks x 32
constexpr int TGROUP = 2;         // 64-row map tiles (8 KB of e2m1 rows each) per LDS buffer

template <bool GATED>
__global__ __launch_bounds__(256) void k_assoc_ties(const uint8_t* __restrict__ q, const uint8_t* __restrict__ qcolor, int nq,
                                                    const int8_t* __restrict__ mx, const uint8_t* __restrict__ mcode,
                                                    const uint8_t* __restrict__ mcolor, int nm_bound, const int* __restrict__ nm_dev,
                                                    int nm_pad, int m_chunk, const float* __restrict__ dist,
                                                    unsigned long long* __restrict__ res)
{
    __shared__ __attribute__((aligned(16))) uint8_t tiles[2][TGROUP * 8192];
    __shared__ unsigned long long s_res[TQW];
    __shared__ uint32_t xtab[256];
    const int nm = nm_dev ? min(nm_bound, *nm_dev) : nm_bound;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q0 = blockIdx.x * TQW + wave * 64;
    const int r32 = lane & 31, half = lane >> 5;
    const int m_begin = blockIdx.y * m_chunk;
    const int m_end = min(nm_pad, m_begin + m_chunk);
    const int n_tiles = (m_end - m_begin) / 64;
    if (n_tiles <= 0 || m_begin >= nm) return;
    xtab[threadIdx.x] = assoc_fp4_expand(threadIdx.x);
    s_res[threadIdx.x] = ~0ull;
    // first group of tiles on its way while the query side is set up
    const uint8_t* src = reinterpret_cast<const uint8_t*>(mx) + (size_t)(m_begin / 64) * 8192;
    const int n_groups = (n_tiles + TGROUP - 1) / TGROUP;
    uint4 stage[TGROUP * 2];
    auto fetch = [&](int g) {
#pragma unroll
        for (int k = 0; k < TGROUP * 2; ++k) {
            const int off = k * 4096 + threadIdx.x * 16;
            stage[k] = (g * TGROUP + (off >> 13)) < n_tiles ? *reinterpret_cast<const uint4*>(src + (size_t)g * (TGROUP * 8192) + off) : make_uint4(0, 0, 0, 0);
        }
    };
    auto park = [&](int buf) {
#pragma unroll
        for (int k = 0; k < TGROUP * 2; ++k) *reinterpret_cast<uint4*>(&tiles[buf][k * 4096 + threadIdx.x * 16]) = stage[k];
    };
    fetch(0);
    __syncthreads();
    // query operands: step s, k-half `half` = code dword 2 s + half, eight e2m1 nibbles per code byte (k_assoc.hip)
    v8i A[2][4];
    float tgt[2][16];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int qi = q0 + 32 * b + r32;
        uint4 c0 = make_uint4(0, 0, 0, 0), c1 = c0;
        if (qi < nq) {
            c0 = *reinterpret_cast<const uint4*>(q + (size_t)qi * 32);
            c1 = *reinterpret_cast<const uint4*>(q + (size_t)qi * 32 + 16);
        }
        const uint32_t w4[4] = { half ? c0.y : c0.x, half ? c0.w : c0.z, half ? c1.y : c1.x, half ? c1.w : c1.z };
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const uint32_t w = w4[s];
            A[b][s] = v8i{ (int)xtab[w & 0xffu], (int)xtab[(w >> 8) & 0xffu], (int)xtab[(w >> 16) & 0xffu], (int)xtab[w >> 24], 0, 0, 0, 0 };
        }
        // the dot product a tie has: 256 - 2 * (the row's minimum distance); rows without a match never compare equal
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qr = q0 + 32 * b + (r & 3) + 8 * (r >> 2) + 4 * half;
            const float d = qr < nq ? dist[qr] : -1.f;
            tgt[b][r] = d >= 0.f ? 256.f - 2.f * d : 3.0e38f;
        }
    }
    park(0);
    __syncthreads();
    for (int g = 0; g < n_groups; ++g) {
        const int buf = g & 1;
        if (g + 1 < n_groups) fetch(g + 1);
#pragma unroll
        for (int tl = 0; tl < TGROUP; ++tl) {
            if (g * TGROUP + tl < n_tiles)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const uint8_t* fb = &tiles[buf][tl * 8192 + half * 1024 + (cb * 32 + r32) * 16];
                v8i B[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const v4i f = *reinterpret_cast<const v4i*>(fb + s * 2048);
                    B[s] = v8i{ f.x, f.y, f.z, f.w, 0, 0, 0, 0 };
                }
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    v16f acc = { 0 };
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A[b][s], B[s], acc, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                    unsigned long long any = 0;
#pragma unroll
                    for (int r = 0; r < 16; ++r) any |= __builtin_amdgcn_ballot_w64(acc[r] == tgt[b][r]);
                    if (any) {
                        // candidates as near as their row's optimum: which registers of this lane, then one at a time
                        uint32_t hits = 0;
#pragma unroll
                        for (int r = 0; r < 16; ++r) hits |= (acc[r] == tgt[b][r] ? 1u : 0u) << r;
                        const int col = m_begin + (g * TGROUP + tl) * 64 + cb * 32 + r32;
                        if (col >= nm) hits = 0;
                        while (hits) {
                            const int r = __ffs(hits) - 1;
                            hits &= hits - 1;
                            const int ql = wave * 64 + 32 * b + (r & 3) + 8 * (r >> 2) + 4 * half;
                            const int qg = blockIdx.x * TQW + ql;
                            if (qg >= nq) continue;
                            if (GATED) {
                                const int qc = qcolor[qg], mc = mcolor[col];
                                if (qc < 3 && mc < 3 && qc != mc) continue;
                            }
                            // its discovery key, from the raw codes
                            const uint32_t key = mih_discovery_key(q + (size_t)qg * 32, mcode + (size_t)col * 32);
                            atomicMin(&s_res[ql], ((unsigned long long)key << 32) | (uint32_t)col);
                        }
                    }
                }
            }
        }
        if (g + 1 < n_groups) park(buf ^ 1);
        __syncthreads();
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
// reference's search finds first.  mcode / mcolor: the map's RAW codes and colours (colours only with gating).  res: nq u64.
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
    if (gating) hipLaunchKernelGGL(k_assoc_ties<true>, dim3(qblocks, splits), dim3(256), 0, s, q, qcolor, nq, mx, mcode, mcolor, nm, nm_dev, nm_pad, m_chunk, dist, res);
    else hipLaunchKernelGGL(k_assoc_ties<false>, dim3(qblocks, splits), dim3(256), 0, s, q, qcolor, nq, mx, mcode, mcolor, nm, nm_dev, nm_pad, m_chunk, dist, res);
    hipLaunchKernelGGL(k_assoc_ties_finish, dim3((nq + 255) / 256), dim3(256), 0, s, nq, res, idx);
    return hipGetLastError();
}

}  // namespace lf
