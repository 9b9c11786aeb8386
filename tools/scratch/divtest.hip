#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
__device__ __forceinline__ uint32_t rng(uint32_t& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }
__global__ void k(unsigned long long* bad_div, unsigned long long* bad_sqrt, unsigned long long* bad_div2, int iters)
{
    uint32_t s = 0x9E3779B9u * (blockIdx.x * blockDim.x + threadIdx.x + 1);
    unsigned long long bd = 0, bs = 0, bd2 = 0;
    for (int i = 0; i < iters; ++i) {
        uint32_t ua = rng(s), ub = rng(s);
        float a = __uint_as_float(ua), b = __uint_as_float(ub);
        if (!(a == a) || !(b == b) || isinf(a) || isinf(b) || b == 0.f) continue;
        float ref = (float)((double)a / (double)b);
        float got = a / b;
        if (__float_as_uint(ref) != __float_as_uint(got) && !(ref != ref && got != got)) bd++;
        // the magnitudes fast_atan2 sees: small non-negative floats
        float c = fabsf(__uint_as_float((ua & 0x007fffffu) | 0x3f800000u)) * 300.f - 300.f, d = fabsf(__uint_as_float((ub & 0x007fffffu) | 0x3f800000u)) * 300.f - 299.f;
        float r2 = (float)((double)c / ((double)d + 2.2204460492503131e-16));
        float g2 = c / d;
        (void)r2; (void)g2;
        float ref2 = (float)((double)c / (double)d), got2 = c / d;
        if (__float_as_uint(ref2) != __float_as_uint(got2)) bd2++;
        float p = fabsf(a);
        float rs = (float)sqrt((double)p), gs = sqrtf(p);
        if (__float_as_uint(rs) != __float_as_uint(gs)) bs++;
    }
    atomicAdd(bad_div, bd); atomicAdd(bad_sqrt, bs); atomicAdd(bad_div2, bd2);
}
int main()
{
    unsigned long long *d, h[3] = {0, 0, 0};
    hipMalloc(&d, 24); hipMemset(d, 0, 24);
    hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, d, d + 1, d + 2, 2000);
    hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("pairs %llu: div mismatches %llu, atan-range div mismatches %llu, sqrt mismatches %llu\n", 1024ull * 256 * 2000, h[0], h[2], h[1]);
    return 0;
}
