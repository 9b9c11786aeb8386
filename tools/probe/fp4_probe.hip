// Probe: semantics of v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 (e2m1) operands on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void k(const uint32_t* a, const uint32_t* b, float* d, uint32_t sa, uint32_t sb)
{
    const int lane = threadIdx.x;
    v8i A = { (int)a[lane * 4], (int)a[lane * 4 + 1], (int)a[lane * 4 + 2], (int)a[lane * 4 + 3], 0, 0, 0, 0 };
    v8i B = { (int)b[lane * 4], (int)b[lane * 4 + 1], (int)b[lane * 4 + 2], (int)b[lane * 4 + 3], 0, 0, 0, 0 };
    v16f C = { 0 };
    // cbsz = 4 / blgp = 4: FP4 for A / B
    v16f D = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, C, 4, 4, 0, (int)sa, 0, (int)sb);
    for (int i = 0; i < 16; ++i) d[lane * 16 + i] = D[i];
}

int main()
{
    // A[row][k], B[k][col] as +-1; element k of lane (r, kh) = k_global = 32 kh + j, nibble j of the lane's 128 bits
    std::vector<int> Am(32 * 64), Bm(64 * 32);
    uint32_t seed = 12345;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (seed >> 16) & 1; };
    for (auto& v : Am) v = rnd() ? -1 : 1;
    for (auto& v : Bm) v = rnd() ? -1 : 1;
    std::vector<uint32_t> a(64 * 4, 0), b(64 * 4, 0);
    for (int lane = 0; lane < 64; ++lane) {
        const int r = lane & 31, kh = lane >> 5;
        for (int j = 0; j < 32; ++j) {
            const int ka = 32 * kh + j;
            const uint32_t na = Am[r * 64 + ka] < 0 ? 0xAu : 0x2u, nb = Bm[ka * 32 + r] < 0 ? 0xAu : 0x2u;
            a[lane * 4 + j / 8] |= na << (4 * (j % 8));
            b[lane * 4 + j / 8] |= nb << (4 * (j % 8));
        }
    }
    uint32_t *da, *db; float* dd;
    hipMalloc(&da, a.size() * 4); hipMalloc(&db, b.size() * 4); hipMalloc(&dd, 64 * 16 * 4);
    hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
    for (int pass = 0; pass < 2; ++pass) {
        const uint32_t sa = pass == 0 ? 0x7f7f7f7fu : 0x88888888u, sb = 0x7f7f7f7fu;
        k<<<1, 64>>>(da, db, dd, sa, sb);
        std::vector<float> d(64 * 16);
        hipMemcpy(d.data(), dd, d.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int lane = 0; lane < 64; ++lane)
            for (int i = 0; i < 16; ++i) {
                const int col = lane & 31, row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                int dot = 0;
                for (int kk = 0; kk < 64; ++kk) dot += Am[row * 64 + kk] * Bm[kk * 32 + col];
                const float want = (pass == 0 ? 1.f : 512.f) * dot;
                if (d[lane * 16 + i] != want) { if (bad < 5) printf("pass %d lane %d i %d got %g want %g\n", pass, lane, i, d[lane * 16 + i], want); ++bad; }
            }
        printf("pass %d (scale_a %s): %d mismatches of 1024\n", pass, pass ? "2^9" : "1", bad);
    }
    return 0;
}
