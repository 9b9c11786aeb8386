// The enumeration order of Mihasher::query's bit strings (shared by k_assoc_ties.hip and k_knn.hip).
#pragma once
#include <cstdint>

namespace lf {

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

// a candidate's discovery key from the XOR of the two codes: the smallest (weight, substring) over the 32 byte substrings, then the
// place of that substring's difference in the enumeration; 0xffffffff when no substring is within 4 bits (farther than 128 bits)
template <typename Table>
__device__ __forceinline__ uint32_t mih_key_from_xor(const uint32_t (&x)[8], const Table& rank)
{
    uint32_t best = 0xffffffffu, bx = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t xb = (x[c] >> (8 * t)) & 255u;
            const uint32_t hk = (uint32_t)__popc(xb) * 32u + (uint32_t)(4 * c + t);
            if (hk < best) { best = hk; bx = xb; }
        }
    const uint32_t h = best >> 5;
    if (h > 4) return 0xffffffffu;
    return (best << 8) | rank.r[h][bx];
}

}  // namespace lf
