// Introspection kernel: evaluates the deterministic math layer (detmath.h) on the device so
// tests can compare it bit-for-bit with the CPU oracle's copy.
#include "common.h"

namespace lf {

__global__ void k_detmath(int which, const double* __restrict__ a, const double* __restrict__ b,
                          double* __restrict__ y, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x = a[i], z = b ? b[i] : 0.0, r = 0.0;
    switch (which) {
    case 0: r = dm::dexp(x); break;
    case 1: r = dm::dlog(x); break;
    case 2: r = dm::dsin(x); break;
    case 3: r = dm::dcos(x); break;
    case 4: r = dm::datan(x); break;
    case 5: r = dm::dasin(x); break;
    case 6: r = dm::dlog10(x); break;
    case 7: r = dm::dsinh_small(x); break;
    case 8: r = dm::datan2(x, z); break;
    case 9: r = dm::dpow(x, z); break;
    case 10: r = dm::dsqrt(x); break;
    case 11: r = x / z; break;
    case 12: r = (double)dm::fast_atan2_deg((float)x, (float)z); break;
    case 13: r = (double)dm::fsqrt((float)x); break;
    case 14: r = (double)dm::fdiv((float)x, (float)z); break;
    default: r = 0.0;
    }
    y[i] = r;
}

}  // namespace lf

extern "C" int lf_debug_detmath(lf_handle* h, int which, const double* a, const double* b, double* y, int n)
{
    if (!h || !a || !y || n <= 0) return LF_ERR_BAD_ARG;
    double *da = nullptr, *db = nullptr, *dy = nullptr;
    size_t bytes = (size_t)n * sizeof(double);
    if (hipMalloc((void**)&da, bytes) != hipSuccess || hipMalloc((void**)&db, bytes) != hipSuccess ||
        hipMalloc((void**)&dy, bytes) != hipSuccess) {
        lf_set_error(h, LF_ERR_HIP, "lf_debug_detmath: hipMalloc failed");
        return LF_ERR_HIP;
    }
    (void)hipMemcpy(da, a, bytes, hipMemcpyHostToDevice);
    if (b) (void)hipMemcpy(db, b, bytes, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(lf::k_detmath, dim3((n + 255) / 256), dim3(256), 0, 0, which, da, b ? db : nullptr, dy, n);
    hipError_t e = hipMemcpy(y, dy, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dy);
    if (e != hipSuccess) { lf_set_error(h, LF_ERR_HIP, "lf_debug_detmath: %s", hipGetErrorString(e)); return LF_ERR_HIP; }
    return LF_OK;
}
