// K_msgs: SegmentList wire bodies <-> the struct-of-arrays segment block.
//
// The reference moves segments between its three nodes as duckietown_msgs/SegmentList messages and
// builds / walks them one Python object at a time:
//   line_detector_node.toSegmentMsg            (ref: src/line_detector/src/line_detector_node.py:251-265)
//   ground_projection_node.lineseglist_cb      (ref: src/ground_projection/src/ground_projection_node.py:55-65)
//   line_sanity_node.processSegmentList        (ref: src/line_sanity/src/line_sanity_node.py:48-72)
// ROS 1 serialisation of `duckietown_msgs/Segment[] segments` (ref: src/duckietown_msgs/msg/Segment.msg:1-8,
// Vector2D.msg:1-2, geometry_msgs/Point): little endian, u32 element count, then per segment 73 bytes:
//   u8 color | f32 pixels_normalized[0].x .y [1].x .y | f32 normal.x .y | f64 points[0].x .y .z [1].x .y .z
// Each stage fills the fields its node fills and leaves the others at the message default (0):
//   detector : color, pixels_normalized, normal          (points stay 0)
//   ground   : color, points (z = 0)                     (normal and pixels_normalized are dropped, :59-63)
//   filtered : the ground segments line sanity keeps, order preserved
// One thread per frame (a frame has tens of segments; the whole batch is < 1 MB).
#include "common.h"

namespace lf {

namespace {

constexpr int kSegBytes = 73;

__device__ __forceinline__ void put_bytes(uint8_t* dst, const void* src, int n)
{
    const uint8_t* s = static_cast<const uint8_t*>(src);
    for (int i = 0; i < n; ++i) dst[i] = s[i];
}
__device__ __forceinline__ void get_bytes(void* dst, const uint8_t* src, int n)
{
    uint8_t* d = static_cast<uint8_t*>(dst);
    for (int i = 0; i < n; ++i) d[i] = src[i];
}

}  // namespace

// counts[f] = segments frame f contributes; byte_offset[f] = start of its body (exclusive scan of 4 + 73 * count)
__global__ void k_msg_layout(int n_frames, int stage, const int* __restrict__ frame_offset, const uint8_t* __restrict__ keep,
                             int* __restrict__ counts, long long* __restrict__ byte_offset)
{
    // single workgroup; n_frames is a batch (hundreds), a serial carry across chunks of blockDim is fine
    __shared__ long long carry;
    __shared__ long long part[1024];
    const int t = threadIdx.x;
    if (t == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n_frames; base += blockDim.x) {
        const int f = base + t;
        long long bytes = 0;
        if (f < n_frames) {
            int n = frame_offset[f + 1] - frame_offset[f];
            if (stage == LF_MSG_FILTERED) {
                int k = 0;
                for (int i = frame_offset[f]; i < frame_offset[f + 1]; ++i) k += keep[i] ? 1 : 0;
                n = k;
            }
            counts[f] = n;
            bytes = 4 + (long long)kSegBytes * n;
        }
        part[t] = bytes;
        __syncthreads();
        if (t == 0) {
            long long run = carry;
            for (int i = 0; i < (int)blockDim.x; ++i) { const long long b = part[i]; part[i] = run; run += b; }
            carry = run;
        }
        __syncthreads();
        if (f < n_frames) byte_offset[f] = part[t];
        __syncthreads();
    }
    if (t == 0) byte_offset[n_frames] = carry;
}

__global__ void k_msg_write(int n_frames, int stage, const int* __restrict__ frame_offset, const uint8_t* __restrict__ color,
                            const float* __restrict__ pixels_normalized, const float* __restrict__ normals,
                            const double* __restrict__ ground, const uint8_t* __restrict__ keep,
                            const int* __restrict__ counts, const long long* __restrict__ byte_offset,
                            uint8_t* __restrict__ out)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frames) return;
    uint8_t* o = out + byte_offset[f];
    const uint32_t n = (uint32_t)counts[f];
    put_bytes(o, &n, 4);
    o += 4;
    for (int i = frame_offset[f]; i < frame_offset[f + 1]; ++i) {
        if (stage == LF_MSG_FILTERED && !keep[i]) continue;
        float pn[4] = { 0.f, 0.f, 0.f, 0.f }, nm[2] = { 0.f, 0.f };
        double pt[6] = { 0., 0., 0., 0., 0., 0. };
        if (stage == LF_MSG_DETECTOR) {
            for (int k = 0; k < 4; ++k) pn[k] = pixels_normalized[(size_t)i * 4 + k];
            nm[0] = normals[(size_t)i * 2];
            nm[1] = normals[(size_t)i * 2 + 1];
        } else {
            pt[0] = ground[(size_t)i * 4 + 0]; pt[1] = ground[(size_t)i * 4 + 1];
            pt[3] = ground[(size_t)i * 4 + 2]; pt[4] = ground[(size_t)i * 4 + 3];
        }
        o[0] = color[i];
        put_bytes(o + 1, pn, 16);
        put_bytes(o + 17, nm, 8);
        put_bytes(o + 25, pt, 48);
        o += kSegBytes;
    }
}

// ---- the other direction: bodies -> SoA (every field of the message is kept)
__global__ void k_msg_counts(int n_frames, const uint8_t* __restrict__ body, const long long* __restrict__ byte_offset,
                             int* __restrict__ frame_offset, int* __restrict__ bad)
{
    // single thread block, serial scan (tiny)
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int run = 0;
    for (int f = 0; f < n_frames; ++f) {
        uint32_t n = 0;
        get_bytes(&n, body + byte_offset[f], 4);
        const long long have = byte_offset[f + 1] - byte_offset[f];
        if (4 + (long long)kSegBytes * n != have) { *bad = 1; n = 0; }
        frame_offset[f] = run;
        run += (int)n;
    }
    frame_offset[n_frames] = run;
}

__global__ void k_msg_read(int n_frames, int capacity, const uint8_t* __restrict__ body, const long long* __restrict__ byte_offset,
                           const int* __restrict__ frame_offset, uint8_t* __restrict__ color,
                           float* __restrict__ pixels_normalized, float* __restrict__ normals, double* __restrict__ ground)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frames) return;
    const uint8_t* o = body + byte_offset[f] + 4;
    for (int i = frame_offset[f]; i < frame_offset[f + 1] && i < capacity; ++i, o += kSegBytes) {
        float pn[4], nm[2];
        double pt[6];
        get_bytes(pn, o + 1, 16);
        get_bytes(nm, o + 17, 8);
        get_bytes(pt, o + 25, 48);
        if (color) color[i] = o[0];
        if (pixels_normalized) for (int k = 0; k < 4; ++k) pixels_normalized[(size_t)i * 4 + k] = pn[k];
        if (normals) { normals[(size_t)i * 2] = nm[0]; normals[(size_t)i * 2 + 1] = nm[1]; }
        if (ground) {
            ground[(size_t)i * 4 + 0] = pt[0]; ground[(size_t)i * 4 + 1] = pt[1];
            ground[(size_t)i * 4 + 2] = pt[3]; ground[(size_t)i * 4 + 3] = pt[4];
        }
    }
}

void launch_msg_layout(int n_frames, int stage, const int* frame_offset, const uint8_t* keep, int* counts,
                       long long* byte_offset, hipStream_t s)
{
    hipLaunchKernelGGL(k_msg_layout, dim3(1), dim3(1024), 0, s, n_frames, stage, frame_offset, keep, counts, byte_offset);
}

void launch_msg_write(int n_frames, int stage, const int* frame_offset, const uint8_t* color, const float* pixels_normalized,
                      const float* normals, const double* ground, const uint8_t* keep, const int* counts,
                      const long long* byte_offset, uint8_t* out, hipStream_t s)
{
    hipLaunchKernelGGL(k_msg_write, dim3((n_frames + 63) / 64), dim3(64), 0, s, n_frames, stage, frame_offset, color,
                       pixels_normalized, normals, ground, keep, counts, byte_offset, out);
}

void launch_msg_read(int n_frames, int capacity, const uint8_t* body, const long long* byte_offset, int* frame_offset,
                     int* bad, uint8_t* color, float* pixels_normalized, float* normals, double* ground, hipStream_t s)
{
    hipLaunchKernelGGL(k_msg_counts, dim3(1), dim3(64), 0, s, n_frames, body, byte_offset, frame_offset, bad);
    hipLaunchKernelGGL(k_msg_read, dim3((n_frames + 63) / 64), dim3(64), 0, s, n_frames, capacity, body, byte_offset,
                       frame_offset, color, pixels_normalized, normals, ground);
}

}  // namespace lf
