// knnMatch / radiusMatch forms of the associator (SURVEY a-10):
//   BinaryDescriptorMatcher::knnMatch     /root/reference/src/line_descriptor/src/binary_descriptor_matcher.cpp:258-335
//   BinaryDescriptorMatcher::radiusMatch  :428-504
// Both run Mihasher(256, 32) with K = k / K = N (:756-819, 635-753): the K nearest train codes within D = 128 bits,
// nearest first; radiusMatch then keeps those within maxDistance.  Among equally near codes the reference lists them in
// its hash tables' discovery order; here (as for lf_associate, include/lanefront.h a-10) in index order.
//
// These are the secondary forms of the matcher -- the 1-NN association of the hot path runs on the matrix cores
// (k_assoc.hip); a list of neighbours per query has no arg-max epilogue to ride on, so this is the plain formulation:
// ONE LANE PER QUERY, the query code in eight registers, map codes staged through LDS 256 at a time (every lane reads the
// same address: a broadcast), XOR + popcount.  11 k queries x 66 k map codes = 1.2 ms; exact integer arithmetic.
#include "common.h"
#include "mih_rank.h"

namespace lf {

__constant__ MihRank c_knn_rank = make_mih_rank();

constexpr int KQ = 256;          // queries per workgroup (one per lane)
constexpr int KT = 256;          // map codes per LDS tile
constexpr int KMAXK = 16;

__device__ __forceinline__ int knn_hamming(const uint32_t (&q)[8], const uint32_t* __restrict__ t)
{
    int d = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) d += __popc(q[w] ^ t[w]);
    return d;
}

// The reference's order among equally near codes (LF_TIE_MIHASHER): (distance, discovery key, index) -- the key of a candidate
// is worked out only when its distance lets it into the list.  64-bit columns: distance << 40 | key (16 bits) << 24 | index.
__global__ __launch_bounds__(KQ) void k_knn_mih(const uint8_t* __restrict__ query, int nq, const uint8_t* __restrict__ map, int nm, int k,
                                                int max_distance, int32_t* __restrict__ idx, float* __restrict__ dist)
{
    __shared__ uint32_t tile[KT * 8];
    __shared__ unsigned long long best[KMAXK][KQ];
    const int t = threadIdx.x, qi = blockIdx.x * KQ + t;
    uint32_t q[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    if (qi < nq) {
        const uint4* p = reinterpret_cast<const uint4*>(query + (size_t)qi * 32);
        const uint4 a = p[0], b = p[1];
        q[0] = a.x; q[1] = a.y; q[2] = a.z; q[3] = a.w; q[4] = b.x; q[5] = b.y; q[6] = b.z; q[7] = b.w;
    }
    for (int j = 0; j < k; ++j) best[j][t] = ~0ull;
    unsigned long long worst = ~0ull;
    for (int base = 0; base < nm; base += KT) {
        __syncthreads();
        const int cnt = nm - base < KT ? nm - base : KT;
        for (int i = t; i < cnt * 2; i += KQ) reinterpret_cast<uint4*>(tile)[i] = reinterpret_cast<const uint4*>(map + (size_t)base * 32)[i];
        __syncthreads();
        if (qi >= nq) continue;
        for (int j = 0; j < cnt; ++j) {
            const uint32_t* tj = tile + 8 * j;
            const int d = knn_hamming(q, tj);
            if (d > max_distance || (unsigned long long)d > (worst >> 40)) continue;
            uint32_t x[8];
#pragma unroll
            for (int w = 0; w < 8; ++w) x[w] = q[w] ^ tj[w];
            const unsigned long long key = ((unsigned long long)d << 40) | ((unsigned long long)(mih_key_from_xor(x, c_knn_rank) & 0xffffu) << 24) | (unsigned long long)(base + j);
            if (key >= worst) continue;
            int pos = k - 1;
            while (pos > 0 && best[pos - 1][t] > key) { best[pos][t] = best[pos - 1][t]; --pos; }
            best[pos][t] = key;
            worst = best[k - 1][t];
        }
    }
    if (qi >= nq) return;
    for (int j = 0; j < k; ++j) {
        const unsigned long long key = best[j][t];
        const bool ok = key != ~0ull;
        idx[(size_t)qi * k + j] = ok ? (int32_t)(key & 0xffffffull) : -1;
        dist[(size_t)qi * k + j] = ok ? (float)(key >> 40) : -1.f;
    }
}

// radiusMatch in the reference's order: after k_radius_fill (distance ascending, index ascending inside a distance) every run of
// equal distance is put into (discovery key, index) order -- one lane per query, insertion sort of the run (runs are short: the
// codes at ONE exact distance of one query)
__global__ __launch_bounds__(KQ) void k_radius_order_mih(const uint8_t* __restrict__ query, int nq, const uint8_t* __restrict__ map,
                                                         const int32_t* __restrict__ offsets, int cap, int32_t* __restrict__ idx, const float* __restrict__ dist)
{
    const int qi = blockIdx.x * KQ + threadIdx.x;
    if (qi >= nq) return;
    uint32_t q[8];
    {
        const uint4* p = reinterpret_cast<const uint4*>(query + (size_t)qi * 32);
        const uint4 a = p[0], b = p[1];
        q[0] = a.x; q[1] = a.y; q[2] = a.z; q[3] = a.w; q[4] = b.x; q[5] = b.y; q[6] = b.z; q[7] = b.w;
    }
    const int a0 = offsets[qi], a1 = min(offsets[qi + 1], cap);
    auto key_of = [&](int j) {
        const uint4* p = reinterpret_cast<const uint4*>(map + (size_t)j * 32);
        const uint4 a = p[0], b = p[1];
        const uint32_t x[8] = { q[0] ^ a.x, q[1] ^ a.y, q[2] ^ a.z, q[3] ^ a.w, q[4] ^ b.x, q[5] ^ b.y, q[6] ^ b.z, q[7] ^ b.w };
        return ((unsigned long long)(mih_key_from_xor(x, c_knn_rank) & 0xffffu) << 32) | (unsigned int)j;
    };
    int r0 = a0;
    while (r0 < a1) {
        int r1 = r0 + 1;
        while (r1 < a1 && dist[r1] == dist[r0]) ++r1;
        for (int i = r0 + 1; i < r1; ++i) {                    // insertion sort by (key, index)
            const int ji = idx[i];
            const unsigned long long ki = key_of(ji);
            int p = i;
            while (p > r0 && key_of(idx[p - 1]) > ki) { idx[p] = idx[p - 1]; --p; }
            idx[p] = ji;
        }
        r0 = r1;
    }
}

// the k nearest within max_distance, (distance, index) ascending; missing slots: idx -1, dist -1
__global__ __launch_bounds__(KQ) void k_knn(const uint8_t* __restrict__ query, int nq, const uint8_t* __restrict__ map, int nm, int k,
                                            int max_distance, int32_t* __restrict__ idx, float* __restrict__ dist)
{
    __shared__ uint32_t tile[KT * 8];
    __shared__ uint32_t best[KMAXK][KQ];              // key = dist << 24 | index, per lane a sorted column
    const int t = threadIdx.x, qi = blockIdx.x * KQ + t;
    uint32_t q[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    if (qi < nq) {
        const uint4* p = reinterpret_cast<const uint4*>(query + (size_t)qi * 32);
        const uint4 a = p[0], b = p[1];
        q[0] = a.x; q[1] = a.y; q[2] = a.z; q[3] = a.w; q[4] = b.x; q[5] = b.y; q[6] = b.z; q[7] = b.w;
    }
    for (int j = 0; j < k; ++j) best[j][t] = 0xffffffffu;
    uint32_t worst = 0xffffffffu;                     // key of the k-th entry
    for (int base = 0; base < nm; base += KT) {
        __syncthreads();
        const int cnt = nm - base < KT ? nm - base : KT;
        for (int i = t; i < cnt * 2; i += KQ) reinterpret_cast<uint4*>(tile)[i] = reinterpret_cast<const uint4*>(map + (size_t)base * 32)[i];
        __syncthreads();
        if (qi >= nq) continue;
        for (int j = 0; j < cnt; ++j) {
            const int d = knn_hamming(q, tile + 8 * j);
            if (d > max_distance) continue;
            const uint32_t key = ((uint32_t)d << 24) | (uint32_t)(base + j);
            if (key >= worst) continue;
            // insertion from the back: equal distances keep index order because map codes arrive in index order
            int pos = k - 1;
            while (pos > 0 && best[pos - 1][t] > key) { best[pos][t] = best[pos - 1][t]; --pos; }
            best[pos][t] = key;
            worst = best[k - 1][t];
        }
    }
    if (qi >= nq) return;
    for (int j = 0; j < k; ++j) {
        const uint32_t key = best[j][t];
        const bool ok = key != 0xffffffffu;
        idx[(size_t)qi * k + j] = ok ? (int32_t)(key & 0xffffffu) : -1;
        dist[(size_t)qi * k + j] = ok ? (float)(key >> 24) : -1.f;
    }
}

// radius, pass 1: per query the number of map codes at every distance 0 .. max_distance (hist [nq][129])
__global__ __launch_bounds__(KQ) void k_radius_count(const uint8_t* __restrict__ query, int nq, const uint8_t* __restrict__ map, int nm,
                                                     int max_distance, int32_t* __restrict__ hist)
{
    __shared__ uint32_t tile[KT * 8];
    const int t = threadIdx.x, qi = blockIdx.x * KQ + t;
    uint32_t q[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    if (qi < nq) {
        const uint4* p = reinterpret_cast<const uint4*>(query + (size_t)qi * 32);
        const uint4 a = p[0], b = p[1];
        q[0] = a.x; q[1] = a.y; q[2] = a.z; q[3] = a.w; q[4] = b.x; q[5] = b.y; q[6] = b.z; q[7] = b.w;
    }
    int32_t* hq = hist + (size_t)qi * 129;
    for (int base = 0; base < nm; base += KT) {
        __syncthreads();
        const int cnt = nm - base < KT ? nm - base : KT;
        for (int i = t; i < cnt * 2; i += KQ) reinterpret_cast<uint4*>(tile)[i] = reinterpret_cast<const uint4*>(map + (size_t)base * 32)[i];
        __syncthreads();
        if (qi >= nq) continue;
        for (int j = 0; j < cnt; ++j) {
            const int d = knn_hamming(q, tile + 8 * j);
            if (d <= max_distance) hq[d] += 1;                 // the lane owns its query's row
        }
    }
}

// per query: exclusive scan of its histogram -> bucket cursors (in place) and its total
__global__ void k_radius_scan(int nq, int max_distance, int32_t* __restrict__ hist, int32_t* __restrict__ count)
{
    const int qi = blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= nq) return;
    int32_t* hq = hist + (size_t)qi * 129;
    int acc = 0;
    for (int d = 0; d <= max_distance; ++d) { const int c = hq[d]; hq[d] = acc; acc += c; }
    count[qi] = acc;
}

// offsets [nq + 1] from the counts (one workgroup; nq is a few ten thousand at most)
__global__ __launch_bounds__(1024) void k_radius_offsets(int nq, const int32_t* __restrict__ count, int32_t* __restrict__ offsets, int* __restrict__ total)
{
    __shared__ int wsum[16];
    __shared__ int carry;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nq; base += 1024) {
        const int i = base + t;
        const int v = i < nq ? count[i] : 0;
        int inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if (lane >= d) inc += o; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int off = carry;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (i < nq) offsets[i] = off + inc - v;
        __syncthreads();
        if (t == 1023) carry = off + inc;
        __syncthreads();
    }
    if (t == 0) { offsets[nq] = carry; *total = carry; }
}

// pass 2: every match goes to the cursor of its (query, distance) bucket: distance ascending, index ascending inside
__global__ __launch_bounds__(KQ) void k_radius_fill(const uint8_t* __restrict__ query, int nq, const uint8_t* __restrict__ map, int nm,
                                                    int max_distance, int32_t* __restrict__ hist, const int32_t* __restrict__ offsets, int cap,
                                                    int32_t* __restrict__ idx, float* __restrict__ dist)
{
    __shared__ uint32_t tile[KT * 8];
    const int t = threadIdx.x, qi = blockIdx.x * KQ + t;
    uint32_t q[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    int off = 0;
    if (qi < nq) {
        const uint4* p = reinterpret_cast<const uint4*>(query + (size_t)qi * 32);
        const uint4 a = p[0], b = p[1];
        q[0] = a.x; q[1] = a.y; q[2] = a.z; q[3] = a.w; q[4] = b.x; q[5] = b.y; q[6] = b.z; q[7] = b.w;
        off = offsets[qi];
    }
    int32_t* hq = hist + (size_t)qi * 129;
    for (int base = 0; base < nm; base += KT) {
        __syncthreads();
        const int cnt = nm - base < KT ? nm - base : KT;
        for (int i = t; i < cnt * 2; i += KQ) reinterpret_cast<uint4*>(tile)[i] = reinterpret_cast<const uint4*>(map + (size_t)base * 32)[i];
        __syncthreads();
        if (qi >= nq) continue;
        for (int j = 0; j < cnt; ++j) {
            const int d = knn_hamming(q, tile + 8 * j);
            if (d > max_distance) continue;
            const int pos = off + hq[d];
            hq[d] += 1;
            if (pos < cap) { idx[pos] = base + j; dist[pos] = (float)d; }
        }
    }
}

// The matcher's per-query mask (binary_descriptor_matcher.cpp:231-235, 305-309, 477-481: a DMatch is made only for queries whose
// mask byte is not 0, and carries its queryIdx): the unmasked queries, in order, with their original row numbers.  One workgroup;
// stable compaction by ballot ranks and a running offset.
__global__ __launch_bounds__(1024) void k_select_queries(const uint8_t* __restrict__ query, const uint8_t* __restrict__ mask, int nq,
                                                         uint8_t* __restrict__ out, int32_t* __restrict__ qidx, int* __restrict__ n_out)
{
    __shared__ int s_wave[16], s_base;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) s_base = 0;
    __syncthreads();
    for (int start = 0; start < nq; start += 1024) {
        const int i = start + t;
        const bool keep = i < nq && mask[i] != 0;
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) s_wave[wave] = __popcll(bal);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wave; ++w) off += s_wave[w];
        if (keep) {
            const int pos = off + __popcll(bal & ((1ull << lane) - 1ull));
            const uint4* src = reinterpret_cast<const uint4*>(query + (size_t)i * 32);
            uint4* dst = reinterpret_cast<uint4*>(out + (size_t)pos * 32);
            dst[0] = src[0]; dst[1] = src[1];
            qidx[pos] = i;
        }
        __syncthreads();
        if (t == 0) { int tot = 0; for (int w = 0; w < 16; ++w) tot += s_wave[w]; s_base += tot; }
        __syncthreads();
    }
    if (t == 0) *n_out = s_base;
}

void launch_select_queries(const uint8_t* q, const uint8_t* mask, int nq, uint8_t* out, int32_t* qidx, int* n_out, hipStream_t s)
{
    hipLaunchKernelGGL(k_select_queries, dim3(1), dim3(1024), 0, s, q, mask, nq, out, qidx, n_out);
}

void launch_knn(const uint8_t* q, int nq, const uint8_t* m, int nm, int k, int max_distance, int mih, int32_t* idx, float* dist, hipStream_t s)
{
    if (mih) { hipLaunchKernelGGL(k_knn_mih, dim3((nq + KQ - 1) / KQ), dim3(KQ), 0, s, q, nq, m, nm, k, max_distance, idx, dist); return; }
    hipLaunchKernelGGL(k_knn, dim3((nq + KQ - 1) / KQ), dim3(KQ), 0, s, q, nq, m, nm, k, max_distance, idx, dist);
}

void launch_radius(const uint8_t* q, int nq, const uint8_t* m, int nm, int max_distance, int32_t* hist, int32_t* count, int32_t* offsets,
                   int* total, int cap, int mih, int32_t* idx, float* dist, hipStream_t s)
{
    const int blocks = (nq + KQ - 1) / KQ;
    (void)hipMemsetAsync(hist, 0, (size_t)nq * 129 * sizeof(int32_t), s);
    hipLaunchKernelGGL(k_radius_count, dim3(blocks), dim3(KQ), 0, s, q, nq, m, nm, max_distance, hist);
    hipLaunchKernelGGL(k_radius_scan, dim3((nq + 255) / 256), dim3(256), 0, s, nq, max_distance, hist, count);
    hipLaunchKernelGGL(k_radius_offsets, dim3(1), dim3(1024), 0, s, nq, count, offsets, total);
    hipLaunchKernelGGL(k_radius_fill, dim3(blocks), dim3(KQ), 0, s, q, nq, m, nm, max_distance, hist, offsets, cap, idx, dist);
    if (mih && cap > 0) hipLaunchKernelGGL(k_radius_order_mih, dim3(blocks), dim3(KQ), 0, s, q, nq, m, offsets, cap, idx, dist);
}

}  // namespace lf
