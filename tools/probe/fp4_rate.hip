// Probe: issue rate of v_mfma_scale_f32_32x32x64_f8f6f4 (FP4 operands) vs v_mfma_i32_32x32x32_i8, one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned long long* out, int iters, int seed)
{
    v8i A = { seed + (int)threadIdx.x, seed * 3, seed * 5, seed * 7, 0, 0, 0, 0 }, B = { seed * 11, seed * 13 + (int)threadIdx.x, seed * 17, seed * 19, 0, 0, 0, 0 };
    v4i A4 = { A[0], A[1], A[2], A[3] }, B4 = { B[0], B[1], B[2], B[3] };
    v16f C0 = { 0 }, C1 = { 1 }, C2 = { 2 }, C3 = { 3 };
    v16i I0 = { 0 }, I1 = { 1 }, I2 = { 2 }, I3 = { 3 };
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        asm volatile("" : "+v"(A), "+v"(B), "+v"(A4), "+v"(B4));
        if (KIND == 0) {
            C0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, C0, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            C1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, C1, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            C2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, C2, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            C3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, C3, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        } else {
            I0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A4, B4, I0, 0, 0, 0);
            I1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A4, B4, I1, 0, 0, 0);
            I2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A4, B4, I2, 0, 0, 0);
            I3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A4, B4, I3, 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0; int si = 0;
    for (int i = 0; i < 16; ++i) { s += C0[i] + C1[i] + C2[i] + C3[i]; si += I0[i] + I1[i] + I2[i] + I3[i]; }
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    reinterpret_cast<float*>(out + 512)[blockIdx.x * 256 + threadIdx.x] = s + (float)si;
}

int main()
{
    unsigned long long* d; hipMalloc(&d, 512 * 8 + 256 * 256 * 4);
    const int iters = 20000;
    for (int kind = 0; kind < 2; ++kind) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (kind == 0) k<0><<<256, 256>>>(d, iters, 3); else k<1><<<256, 256>>>(d, iters, 3);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            double cyc = 0; for (int i = 0; i < 256; ++i) cyc += (double)h[i]; cyc /= 256;
            const double nm = 4.0 * iters;
            printf("%s: %.3f ms, %.1f cycles per MFMA (s_memtime), %.2f GHz-equivalent, %.2f Pop/s chip\n", kind == 0 ? "fp4 32x32x64" : "i8  32x32x32", ms, cyc / nm,
                   cyc / (ms * 1e6), (kind == 0 ? 131072.0 : 65536.0) * nm * 1024 / (ms * 1e-3) / 1e15);
        }
    }
    return 0;
}
