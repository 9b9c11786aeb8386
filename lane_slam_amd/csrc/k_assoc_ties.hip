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
//
// Round 5: ONLY WHERE A TIE CAN BE.  The distance pass leaves, per query and map chunk, the chunk's best (distance, column) in
// part[] (k_assoc.hip: assoc_publish_and_merge).  A candidate as near as the query's optimum can only sit in a chunk whose best
// distance IS the optimum -- one chunk for most queries, two or three when the tie is real -- so the distance pass's merge step lists,
// per chunk, the queries that have to look there (their row of part[] against their optimum), and the matrix pass here runs on those
// (chunk, 256 listed queries) pieces alone: about 1 / (number of chunks) of the N x M products instead of all of them.  A chunk is
// cut into `subs` pieces of whole tile groups so that the few pieces still fill the chip.
#include <cstdlib>
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

// Two shapes of the pass (round 5).  SMALL: one 32-query row block per wave, one 8 KB map tile per LDS buffer -- 24 KB of LDS and 128
// registers, a workgroup that finds room on a chip full of region-growing workgroups (25 KB, 96 registers each) as soon as ONE of
// them retires; it is what a pipelined front end's map steps use (same bench, same call: 119 k -> 124 k frames/s against the big
// shape, although alone it is a third slower).  BIG: two row blocks per wave sharing every map fragment, two tiles per buffer (45 KB,
// 230 registers): the faster one alone, used for large query sets.

// eight e2m1 nibbles (0x2 = bit 0, 0xA = bit 1) -> the code byte they were expanded from (assoc_fp4_expand's inverse)
__device__ __forceinline__ uint32_t fp4_collapse(uint32_t w)
{
    uint32_t t = (w >> 3) & 0x11111111u;
    t = (t | (t >> 3)) & 0x03030303u;
    t = (t | (t >> 6)) & 0x000f000fu;
    return (t | (t >> 12)) & 0xffu;
}

// One candidate as near as its query's optimum: its discovery key -- the smallest (weight, substring) over the 32 byte substrings,
// then ONE table lookup for the place of that substring's difference in the enumeration -- from the two raw codes, folded into the
// query's result word with a 64-bit atomic minimum (key << 32 | index).  Round 5: the matrix loop only STAGES the candidates (a
// pair of numbers each, in LDS); their keys are worked out at the next flush, one candidate per thread -- inside the matrix loop a
// candidate cost ~500 instructions on one or two lanes of a wave, and since the pass only visits chunks that hold candidates
// that was most of its time.
__device__ __noinline__ void tie_eval(const uint8_t* __restrict__ q, const uint8_t* __restrict__ mcode, const uint8_t* __restrict__ qcolor,
                                         const uint8_t* __restrict__ mcolor, bool gated, int qg, int col, unsigned long long* __restrict__ res)
{
    if (gated) {
        const int qc = qcolor[qg], mc = mcolor[col];
        if (qc < 3 && mc < 3 && qc != mc) return;
    }
    const uint4 a0 = *reinterpret_cast<const uint4*>(q + (size_t)qg * 32), a1 = *reinterpret_cast<const uint4*>(q + (size_t)qg * 32 + 16);
    const uint4 b0 = *reinterpret_cast<const uint4*>(mcode + (size_t)col * 32), b1 = *reinterpret_cast<const uint4*>(mcode + (size_t)col * 32 + 16);
    const uint32_t x[8] = { a0.x ^ b0.x, a0.y ^ b0.y, a0.z ^ b0.z, a0.w ^ b0.w, a1.x ^ b1.x, a1.y ^ b1.y, a1.z ^ b1.z, a1.w ^ b1.w };
    uint32_t best = 0xffffffffu, bx = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t xb = (x[c] >> (8 * t)) & 255u;
            const uint32_t hk = (uint32_t)__popc(xb) * 32u + (uint32_t)(4 * c + t);        // (weight, substring)
            if (hk < best) { best = hk; bx = xb; }
        }
    }
    const uint32_t h = best >> 5;
    if (h > 4) return;                                       // cannot happen within 128 bits; the reference would never meet it
    const uint32_t key = (best << 8) | c_mih_rank.r[h][bx];
    atomicMin(res + qg, ((unsigned long long)key << 32) | (uint32_t)col);
}

constexpr int THCAP = 256;        // candidates a workgroup stages in LDS between two flushes (one per thread of a flush)
#define TQW (128 * TRB)
#define TSETS (TGROUP * 4)        // accumulator sets per group: tile x column block x row block

// Persistent workgroups over the WORK ITEMS of the pass: item = (map chunk c of the distance pass, slab of 256 of the queries
// listed for c, piece of c's columns).  The lists come from the distance pass's merge step (k_assoc.hip: pieces of up to 64 query
// numbers per (chunk, wave of a query block) and their counts); every workgroup adds the counts up for itself (a few thousand
// words) -- no list kernel, no prefix kernel, no early-exit workgroups.
template <bool GATED, int TRB, int TGROUP, int kTieUnroll>
__device__ __forceinline__ void assoc_ties_body(const uint8_t* __restrict__ q, const uint8_t* __restrict__ qcolor, int nq,
                                                    const int8_t* __restrict__ mx, const uint8_t* __restrict__ mcode, const uint8_t* __restrict__ mcolor,
                                                    int nm_bound, const int* __restrict__ nm_dev,
                                                    int nm_pad, int m_chunk, int splits, int subs, int sub_len, const int* __restrict__ pieces,
                                                    const int* __restrict__ counts, int n_pieces, const float* __restrict__ dist,
                                                    unsigned long long* __restrict__ res)
{
    // two buffers of TGROUP map tiles, filled by LDS-DMA (global_load_lds: 64 lanes x 16 bytes = 1 KB of a tile per instruction
    // and wave, no staging registers); separate arrays and a loop unrolled by two, so that the compiler can tell the buffer being
    // filled from the one being read
    __shared__ __attribute__((aligned(1024))) uint8_t tiles_a[TGROUP * 8192];
    __shared__ __attribute__((aligned(1024))) uint8_t tiles_b[TGROUP * 8192];
    __shared__ __attribute__((aligned(16))) uint32_t qraw[TQW * 8];
    __shared__ uint32_t xtab[256];
    __shared__ int s_qid[TQW];
    __shared__ int2 s_hits[THCAP];
    __shared__ int s_nh;
    __shared__ int s_flush[2];            // "flush after the next group", written by thread 0 alone (see group())
    __shared__ int s_items[129];          // items in front of chunk c (splits <= 128)
    __shared__ int s_wsum[4];
    const int nm = nm_dev ? min(nm_bound, *nm_dev) : nm_bound;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r32 = lane & 31, half = lane >> 5;
    int* s_pfx = reinterpret_cast<int*>(tiles_b);          // the current chunk's running piece counts: n_pieces + 1 words, gone before tiles_b is filled
    xtab[threadIdx.x] = assoc_fp4_expand(threadIdx.x);
    if (threadIdx.x == 0) { s_nh = 0; s_flush[0] = 0; s_flush[1] = 0; }
    // ---- the items: chunk c has ceil(listed(c) / 256) slabs x subs pieces
    for (int c = wave; c < splits; c += 4) {               // one wave per chunk adds its counts up
        int tot = 0;
        for (int p = lane; p < n_pieces; p += 64) tot += counts[(size_t)c * n_pieces + p];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) tot += __shfl_xor(tot, d);
        if (lane == 0) s_items[c + 1] = ((tot + TQW - 1) / TQW) * subs;
    }
    __syncthreads();
    if (threadIdx.x == 0) { s_items[0] = 0; for (int c = 0; c < splits; ++c) s_items[c + 1] += s_items[c]; }
    __syncthreads();
    const int n_items = s_items[splits];
    constexpr uint32_t kTie = 0x3f000000u;          // +0.5
    // the staged candidates' discovery keys, one candidate per thread
    auto flush = [&]() {
        const int n = min(s_nh, THCAP);
        if ((int)threadIdx.x < n) { const int2 hpair = s_hits[threadIdx.x]; tie_eval(q, mcode, qcolor, mcolor, GATED, hpair.x, hpair.y, res); }
        __syncthreads();
        if (threadIdx.x == 0) { s_nh = 0; s_flush[0] = 0; s_flush[1] = 0; }
        __syncthreads();
    };
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        int chunk = 0;
        while (s_items[chunk + 1] <= item) ++chunk;
        const int rel = item - s_items[chunk];
        const int slab = rel / subs, piece = rel - slab * subs;
        const int m_begin = chunk * m_chunk + piece * sub_len;
        const int m_end = min(min(nm_pad, (chunk + 1) * m_chunk), m_begin + sub_len);
        const int n_tiles = __builtin_amdgcn_readfirstlane((m_end - m_begin) / 64);
        if (n_tiles <= 0 || m_begin >= nm) continue;
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
        // the chunk's running piece counts (workgroup scan), then the slab's queries: listed query number li sits in the last piece
        // whose running count is <= li
        {
            const int* cnt = counts + (size_t)chunk * n_pieces;
            int carry = 0;
            for (int p0 = 0; p0 < n_pieces; p0 += 256) {
                const int p = p0 + threadIdx.x;
                const int v = p < n_pieces ? cnt[p] : 0;
                int inc = v;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const int n = __shfl_up(inc, d); if (lane >= d) inc += n; }
                if (lane == 63) s_wsum[wave] = inc;
                __syncthreads();
                int base = carry;
                for (int k = 0; k < wave; ++k) base += s_wsum[k];
                if (p < n_pieces) s_pfx[p] = base + inc - v;
                carry += s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
                __syncthreads();
            }
            if (threadIdx.x == 0) s_pfx[n_pieces] = carry;
            __syncthreads();
            const int n_listed = s_pfx[n_pieces];
            const int li = slab * TQW + threadIdx.x;
            int qi = -1;
            if (li < n_listed && (int)threadIdx.x < TQW) {
                int x = 0, y = n_pieces;
                while (y - x > 1) { const int mid = (x + y) >> 1; if (s_pfx[mid] <= li) x = mid; else y = mid; }
                qi = pieces[((size_t)chunk * n_pieces + x) * 64 + (li - s_pfx[x])];
            }
            if ((int)threadIdx.x < TQW) {
                s_qid[threadIdx.x] = qi;
                uint4 c0 = make_uint4(0, 0, 0, 0), c1 = c0;
                if (qi >= 0) { c0 = *reinterpret_cast<const uint4*>(q + (size_t)qi * 32); c1 = *reinterpret_cast<const uint4*>(q + (size_t)qi * 32 + 16); }
                *reinterpret_cast<uint4*>(&qraw[threadIdx.x * 8]) = c0;
                *reinterpret_cast<uint4*>(&qraw[threadIdx.x * 8 + 4]) = c1;
            }
        }
        __syncthreads();
        // query operands: step s, k-half `half` = code dword 2 s + half, eight e2m1 nibbles per code byte (k_assoc.hip)
        v8i A[TRB][4];
        // Every chain STARTS at 0.5 - (the dot product a tie has = 256 - 2 * the row's minimum distance), so a tie ends at exactly
        // +0.5 (0x3f000000), every other candidate at k + 0.5 with k a non-zero integer (all exact in f32), rows without a match near -3e38.
        v16f start[TRB];
#pragma unroll
        for (int b = 0; b < TRB; ++b) {
            const uint32_t* c = &qraw[(wave * (32 * TRB) + 32 * b + r32) * 8];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const uint32_t w = c[2 * s + half];
                A[b][s] = v8i{ (int)xtab[w & 0xffu], (int)xtab[(w >> 8) & 0xffu], (int)xtab[(w >> 16) & 0xffu], (int)xtab[w >> 24], 0, 0, 0, 0 };
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qr = s_qid[wave * (32 * TRB) + 32 * b + (r & 3) + 8 * (r >> 2) + 4 * half];
                const float d = qr >= 0 ? dist[qr] : -1.f;
                start[b][r] = d >= 0.f ? 0.5f - (256.f - 2.f * d) : -3.0e38f;
            }
        }
        // accumulator sets: (tile tl of the group, column block cb, row block b) -> 16 dot products per lane; the two row blocks of a
        // column block share its fragments and run as two interleaved chains
        auto dots2 = [&](const uint8_t* tiles, int tl, int cb, v16f& acc0, v16f& acc1) {
            const uint8_t* fb = tiles + tl * 8192 + half * 1024 + (cb * 32 + r32) * 16;
            acc0 = start[0]; acc1 = start[TRB - 1];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const v4i f = *reinterpret_cast<const v4i*>(fb + s * 2048);
                const v8i B = v8i{ f.x, f.y, f.z, f.w, 0, 0, 0, 0 };
                acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A[0][s], B, acc0, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                if (TRB == 2) acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A[TRB - 1][s], B, acc1, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            }
        };
        auto group = [&](int g, const uint8_t* tiles, uint8_t* next) {
            if (g + 1 < n_groups) fetch(g + 1, next);
            const int tiles_here = min(TGROUP, n_tiles - g * TGROUP);
#pragma unroll kTieUnroll
            for (int pair = 0; pair < TSETS / 2; ++pair) {
                const int tl = pair >> 1, cb = pair & 1;
                if (tl >= tiles_here) break;                                    // a group's tiles past the piece's end
                v16f acc0, acc1;
                dots2(tiles, tl, cb, acc0, acc1);
                const int col = m_begin + (g * TGROUP + tl) * 64 + cb * 32 + r32;
#pragma unroll
                for (int b = 0; b < TRB; ++b) {
                    uint32_t hits = 0;
#pragma unroll
                    for (int r = 0; r < 16; ++r) hits |= (__float_as_uint(b ? acc1[r] : acc0[r]) == kTie ? 1u : 0u) << r;
                    if (col >= nm) hits = 0;
                    if (__builtin_amdgcn_ballot_w64(hits != 0)) {
                        while (hits) {
                            const int r = __ffs(hits) - 1;
                            hits &= hits - 1;
                            const int qg = s_qid[wave * (32 * TRB) + 32 * b + 4 * half + (r & 3) + 8 * (r >> 2)];
                            if (qg < 0) continue;
                            const int pos = atomicAdd(&s_nh, 1);
                            if (pos < THCAP) s_hits[pos] = make_int2(qg, col);
                            else tie_eval(q, mcode, qcolor, mcolor, GATED, qg, col, res);      // (the stage is full: on the spot)
                        }
                    }
                }
            }
            // Flush when the stage is half full -- decided by ONE thread and acted on a group later.  (s_nh itself cannot be the
            // condition: a fast wave is already adding the next group's candidates while a slow one reads it here, the branch would
            // not be uniform and flush()'s barriers would pair up with the wrong ones.)  Thread 0 writes s_flush[g & 1] behind this
            // barrier; everybody reads it behind the NEXT group's barrier, and it is not written again before the one after that.
            // A stage that fills up in between evaluates on the spot (above), so the delay costs time only.
            __syncthreads();
            if (s_flush[(g + 1) & 1]) flush();
            if (threadIdx.x == 0) s_flush[g & 1] = s_nh > THCAP / 2;
        };
        for (int g = 0; g < n_groups; g += 2) {
            group(g, tiles_a, tiles_b);
            if (g + 1 < n_groups) group(g + 1, tiles_b, tiles_a);
        }
        flush();
    }
}

#undef TQW
#undef TSETS
#define LF_TIE_ARGS const uint8_t* __restrict__ q, const uint8_t* __restrict__ qcolor, int nq, const int8_t* __restrict__ mx, const uint8_t* __restrict__ mcode, \
    const uint8_t* __restrict__ mcolor, int nm_bound, const int* __restrict__ nm_dev, int nm_pad, int m_chunk, int splits, int subs, int sub_len, \
    const int* __restrict__ pieces, const int* __restrict__ counts, int n_pieces, const float* __restrict__ dist, unsigned long long* __restrict__ res
#define LF_TIE_PASS q, qcolor, nq, mx, mcode, mcolor, nm_bound, nm_dev, nm_pad, m_chunk, splits, subs, sub_len, pieces, counts, n_pieces, dist, res
template <bool GATED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_assoc_ties_small(LF_TIE_ARGS) { assoc_ties_body<GATED, 1, 1, 1>(LF_TIE_PASS); }
template <bool GATED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_assoc_ties_big(LF_TIE_ARGS) { assoc_ties_body<GATED, 2, 2, 2>(LF_TIE_PASS); }
#undef LF_TIE_ARGS
#undef LF_TIE_PASS

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
                             int nm, const int* nm_dev, int gating, AssocScratch& w, unsigned long long* res, int32_t* idx, const float* dist, hipStream_t s)
{
    if (nq <= 0 || nm <= 0) return hipSuccess;
    const int nm_pad = (int)assoc_rows_padded_m(nm);
    const int qblocks = (nq + 255) / 256;                                   // the distance pass's query blocks (256 queries each)
    const int splits = w.splits, m_chunk = w.m_chunk;                      // the distance pass's split of the map: its lists are laid out by it
    // the distance pass must have been launched with w.tie_res = res: its merge step wrote the lists and reset the result words
    if (splits < 1 || splits > 128 || w.qblocks != qblocks || w.tie_res != res || !w.tie_list) return hipErrorInvalidValue;
    // the shape: SMALL up to 12 288 queries (a front end's batch: the pass must find room between other kernels), BIG beyond
    // (LF_TIE_SHAPE=small|big overrides, for A/B runs)
    static const char* forced = getenv("LF_TIE_SHAPE");
    const bool small = forced ? forced[0] == 's' : nq <= 12288;
    const int trb = small ? 1 : 2, tgroup = small ? 1 : 2;
    if ((size_t)qblocks * 4 + 1 > (size_t)tgroup * 8192 / 4) return hipErrorInvalidValue;      // (the running piece counts of a chunk sit in one tile buffer)
    const int n_pieces = qblocks * 4;
    const int* pieces = w.tie_list;
    const int* counts = w.tie_list + (size_t)splits * n_pieces * 64;
    // every listed (chunk, slab) is cut into pieces of whole tile groups: about 1.5 slabs per 128 / 256 queries in all, 512 workgroup slots
    const int tqw = 128 * trb;
    const int tiles_chunk = m_chunk / 64;
    const int slabs = (nq + tqw - 1) / tqw;
    int subs = (768 + slabs + slabs / 2 - 1) / (slabs + slabs / 2);
    if (subs > (tiles_chunk + tgroup - 1) / tgroup) subs = (tiles_chunk + tgroup - 1) / tgroup;
    if (subs < 1) subs = 1;
    int sub_len = (tiles_chunk + subs - 1) / subs;
    sub_len = (sub_len + tgroup - 1) / tgroup * tgroup * 64;
    subs = (m_chunk + sub_len - 1) / sub_len;
    static const int grid = getenv("LF_TIE_GRID") ? atoi(getenv("LF_TIE_GRID")) : 512;
#define LF_TIE_LAUNCH(K) hipLaunchKernelGGL(K, dim3(grid), dim3(256), 0, s, q, qcolor, nq, mx, mcode, mcolor, nm, nm_dev, nm_pad, m_chunk, splits, subs, sub_len, pieces, counts, n_pieces, dist, res)
    if (small) { if (gating) LF_TIE_LAUNCH(k_assoc_ties_small<true>); else LF_TIE_LAUNCH(k_assoc_ties_small<false>); }
    else { if (gating) LF_TIE_LAUNCH(k_assoc_ties_big<true>); else LF_TIE_LAUNCH(k_assoc_ties_big<false>); }
#undef LF_TIE_LAUNCH
    hipLaunchKernelGGL(k_assoc_ties_finish, dim3((nq + 255) / 256), dim3(256), 0, s, nq, res, idx);
    return hipGetLastError();
}

}  // namespace lf
