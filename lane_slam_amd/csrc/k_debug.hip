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

// Calibration probes for the FETCH_SIZE / WRITE_SIZE counters (MI355X_MICROARCH.md: "other access widths are
// uncalibrated: calibrate on a known byte count in your own access pattern").  Each kernel streams a buffer of known
// size once with one of the access shapes the pipeline's kernels use; tools/fetch_probe.py compares the counters of a
// `rocprofv3 --pmc` pass with the byte count.
template <int W>      // bytes per lane per access: 4 (dword), 8, 12 (three dwords, k_pre's source rows), 16
__global__ void k_probe_read(const uint32_t* __restrict__ src, size_t n_words, uint32_t* __restrict__ sink)
{
    constexpr int D = W / 4;
    uint32_t acc = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x * D;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * D; i + D <= n_words; i += stride) {
        if (D == 1) acc ^= src[i];
        else if (D == 2) { const uint2 v = *reinterpret_cast<const uint2*>(src + i); acc ^= v.x ^ v.y; }
        else if (D == 3) { acc ^= src[i] ^ src[i + 1] ^ src[i + 2]; }
        else { const uint4 v = *reinterpret_cast<const uint4*>(src + i); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x9e3779b9u) sink[0] = acc;          // keeps the loads alive, practically never taken
}

template <int W>
__global__ void k_probe_write(uint32_t* __restrict__ dst, size_t n_words)
{
    constexpr int D = W / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x * D;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * D; i + D <= n_words; i += stride) {
        if (D == 1) dst[i] = (uint32_t)i;
        else *reinterpret_cast<uint4*>(dst + i) = make_uint4((uint32_t)i, 1u, 2u, 3u);
    }
}

// k_canny_nms's access shape: the buffer as 640 x 320 dword images; a wave walks a 64-column strip down 62 rows and reads,
// per row, its column and both horizontal neighbours with three buffer loads (scalar row offset + fixed lane offset).
// Every byte of the buffer is needed exactly once (the +-1 columns overlap between the three loads and between strips).
__global__ __launch_bounds__(256) void k_probe_stencil(const uint32_t* __restrict__ src, int n_frames, uint32_t* __restrict__ sink)
{
    constexpr int W = 640, H = 320, R = 62, STRIPS = W / 64, BANDS = (H + R - 1) / R;
    const int u = (int)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int f = u / (STRIPS * BANDS), uf = u - f * (STRIPS * BANDS);
    if (f >= n_frames) return;
    const int strip = uf % STRIPS, band = uf / STRIPS, lane = (int)(threadIdx.x & 63u);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(src + (size_t)f * W * H), 0, W * H * 4, 0x00020000);
    const int x = strip * 64 + lane;
    const int xm = 4 * x, xl = 4 * max(x - 1, 0), xr = 4 * min(x + 1, W - 1);
    uint32_t acc = 0;
    for (int y = band * R; y < min(band * R + R, H); ++y) {
        const int row = y * W * 4;
        acc ^= __builtin_amdgcn_raw_buffer_load_b32(rs, xl, row, 0) ^ __builtin_amdgcn_raw_buffer_load_b32(rs, xm, row, 0) ^
               __builtin_amdgcn_raw_buffer_load_b32(rs, xr, row, 0);
    }
    if (acc == 0x9e3779b9u) sink[0] = acc;
}

}  // namespace lf

// stream `bytes` of a scratch buffer `reps` times with the given access shape (width 4 / 8 / 12 / 16 bytes per lane, 43 = k_canny_nms's stencil;
// write != 0: stores, widths 4 and 16 only).  Diagnostic entry for tools/fetch_probe.py.
extern "C" int lf_debug_probe(lf_handle* h, int width, int write, size_t bytes, int reps)
{
    if (!h || bytes < 4096 || reps < 1) return LF_ERR_BAD_ARG;
    uint32_t *buf = nullptr, *sink = nullptr;
    if (hipMalloc((void**)&buf, bytes) != hipSuccess || hipMalloc((void**)&sink, 256) != hipSuccess) {
        if (buf) (void)hipFree(buf);
        lf_set_error(h, LF_ERR_HIP, "lf_debug_probe: hipMalloc failed");
        return LF_ERR_HIP;
    }
    (void)hipMemset(buf, 1, bytes);
    const size_t nw = bytes / 4;
    const dim3 grid(256 * 16), block(256);
    for (int r = 0; r < reps; ++r) {
        if (write) {
            if (width == 4) hipLaunchKernelGGL(lf::k_probe_write<4>, grid, block, 0, 0, buf, nw);
            else hipLaunchKernelGGL(lf::k_probe_write<16>, grid, block, 0, 0, buf, nw);
        } else if (width == 43) {                     // k_canny_nms's three-column stencil shape (whole 640 x 320 images only)
            const int nf = (int)(bytes / (640 * 320 * 4));
            hipLaunchKernelGGL(lf::k_probe_stencil, dim3((nf * 60 + 3) / 4), block, 0, 0, buf, nf, sink);
        } else if (width == 4) hipLaunchKernelGGL(lf::k_probe_read<4>, grid, block, 0, 0, buf, nw, sink);
        else if (width == 8) hipLaunchKernelGGL(lf::k_probe_read<8>, grid, block, 0, 0, buf, nw, sink);
        else if (width == 12) hipLaunchKernelGGL(lf::k_probe_read<12>, grid, block, 0, 0, buf, nw, sink);
        else hipLaunchKernelGGL(lf::k_probe_read<16>, grid, block, 0, 0, buf, nw, sink);
    }
    const hipError_t e = hipDeviceSynchronize();
    (void)hipFree(buf); (void)hipFree(sink);
    if (e != hipSuccess) { lf_set_error(h, LF_ERR_HIP, "lf_debug_probe: %s", hipGetErrorString(e)); return LF_ERR_HIP; }
    return LF_OK;
}

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
