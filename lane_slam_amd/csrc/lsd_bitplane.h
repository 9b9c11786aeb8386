// The defined pixels of one LSD problem as a bit plane with running counts, in LDS (round 4).
//   bits64 [words]      bit (y * W + x) set <=> the pixel's gradient is defined             (words = (H * W / 64 + 2) & ~1)
//   pref   [words / 2]  defined pixels in front of a PAIR of words = the compact index of the pair's first defined pixel
// The compact arrays list a problem's defined pixels in raster order, so a pixel's compact index is its RANK: the count of its
// pair + the set bits in front of it inside the pair.  Used by k_lsd_grow_bm (every neighbour lookup of the region growing,
// lsd_grow.h) and by k_lsd_label<true> (component labelling of the large problems).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace lf {

__host__ __device__ inline int bitplane_words(size_t Ps) { return ((int)(Ps >> 6) + 2) & ~1; }
// u32 words of LDS the plane and its counts take
__host__ __device__ inline size_t bitplane_lds_words(size_t Ps) { const size_t w = (size_t)bitplane_words(Ps); return 2 * w + (((w >> 1) + 1) >> 1); }

// Build both from the compact coordinate list (y << 16 | x), by all NT threads of the workgroup; wave_tot: NT / 64 ints of LDS.
// Ends with the tables complete for the calling thread's own writes only: the caller synchronises.
// XY = true: the list holds y << 16 | x (the compact coordinate list); false: raster addresses y * Ws + x (k_lsd_grad's records)
template <int NT, bool XY = true>
__device__ __forceinline__ void bitplane_build(uint32_t* lds, const uint32_t* __restrict__ gxy, int n_def, int Ws, size_t Ps, int* wave_tot)
{
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int words = bitplane_words(Ps);
    unsigned long long* bits64 = reinterpret_cast<unsigned long long*>(lds);
    uint16_t* pref = reinterpret_cast<uint16_t*>(lds + 2 * words);
    for (int i = tid; i < 2 * words; i += NT) lds[i] = 0u;
    __syncthreads();
    for (int i = tid; i < n_def; i += NT) {
        const uint32_t xy = gxy[i];
        const int pos = XY ? (int)(xy >> 16) * Ws + (int)(xy & 0xffffu) : (int)xy;
        atomicOr(lds + (pos >> 5), 1u << (pos & 31));
    }
    __syncthreads();
    // running counts, one per pair of words: every thread sums a run of consecutive pairs, the runs are scanned across the workgroup
    const int pairs = words >> 1;
    const int per = (pairs + NT - 1) / NT;
    const int w0 = tid * per < pairs ? tid * per : pairs, w1 = w0 + per < pairs ? w0 + per : pairs;
    int mine = 0;
    for (int w = w0; w < w1; ++w) mine += __builtin_popcountll(bits64[2 * w]) + __builtin_popcountll(bits64[2 * w + 1]);
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d, 64);
        if (lane >= d) incl += v;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int base = incl - mine;
    for (int k = 0; k < wave; ++k) base += wave_tot[k];
    for (int w = w0; w < w1; ++w) { pref[w] = (uint16_t)base; base += __builtin_popcountll(bits64[2 * w]) + __builtin_popcountll(bits64[2 * w + 1]); }
}

// the rank of raster position pos: the number of defined pixels in front of it
__device__ __forceinline__ uint32_t bitplane_rank(const uint32_t* lds, size_t Ps, int pos)
{
    const unsigned long long* bits64 = reinterpret_cast<const unsigned long long*>(lds);
    const uint16_t* pref = reinterpret_cast<const uint16_t*>(lds + 2 * bitplane_words(Ps));
    const int w = pos >> 6;
    const unsigned long long lo = bits64[w & ~1], cur = bits64[w];
    return (uint32_t)pref[w >> 1] + ((w & 1) ? (uint32_t)__builtin_popcountll(lo) : 0u) + (uint32_t)__builtin_popcountll(cur & ((1ull << (pos & 63)) - 1ull));
}

}  // namespace lf
