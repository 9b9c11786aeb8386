// Live map of the associator (SURVEY a-11, 8f-3): device-resident entries + their packed MFMA operands, updated in
// place from the association results of a batch.
//
// The reference keeps its map as an append-only Python list of every received segment
// (/root/reference/src/show_map/src/show_map.py:28-42) in the robot frame of the moment, with the map -> duck
// transform published separately by odometry (/root/reference/src/odometry/src/odometry.py:110-120); its
// line_associator node is a stub (src/line_associator/src/line_associator_node.py:12-86).  The contract built here
// (include/lanefront.h, "live map") is therefore the build's own: segments are moved into the map frame with their
// frame's pose, matched against the map (k_assoc.hip) and then either APPENDED (show_map's behaviour) or, under the
// MERGE policy, used to REFRESH the entry they matched.
//
// Exchange format = a BLOCK: [1 + rows][80] bytes, row 0 the header {magic, count, step, n_frames}, then one row per
// segment in SegmentList order:
//   0..31 code | 32..63 ground x0 y0 x1 y1 (f64, map frame) | 64 idx i32 | 68 dist f32 | 72 colour | 73 keep | pad
// Blocks are what ranks all-gather: every rank applies the same blocks in rank order, so replicas stay
// identical; a single GPU applies its own block through the same kernels.
//
// Update semantics for the concatenation of the blocks' rows (s = 0, 1, ...), sequentially defined and executed in
// parallel with the same result (oracle/lf_oracle_map.c is the sequential statement):
//   eligible(s)  = keep[s] or not kept_only
//   refresh(s)   = eligible, policy MERGE, idx[s] >= 0 and dist[s] <= merge_distance
//   pass 1: every refresh in order: entry idx[s] <- code, colour, ground of s; hits += 1; last_seen = step
//           (parallel: the LAST s wins the entry -- atomicMax on a winner slot -- and hits take an atomicAdd each)
//   pass 2: every other eligible s in order: appended at (head + rank) mod capacity (RING; an append overwrites
//           whatever is there, including an entry refreshed in pass 1) or at size + rank while it fits (FULL_ERROR)
//           (parallel: rank = exclusive scan of the append flags)
#include "common.h"

namespace lf {

constexpr int kRow = LF_BLOCK_ROW_BYTES;
constexpr int kRefFlag = 0x40000000;
constexpr uint32_t kMagic = 0x4b42464cu;   // "LFBK"

struct MapDev {
    int capacity, policy, kept_only, merge_distance, when_full, fp4;
    uint8_t* code; uint8_t* color; double* ground; int* hits; int* last_seen; int* winner;
    int8_t* mx; int8_t* mcx;
    int* state;                    // [0] size [1] head [2] flags of the latest failing update (1 full, 2 bad header, 4 a peer's block
                                   // overflowed) [3] n_app [4] n_ref [5] old head [6] old size [7] step [8] failing updates so far
                                   // [9], [10] accumulators of the update in flight (refreshes, flags)
    unsigned long long* totals;    // [0] appended [1] refreshed
};

// block rows from a struct-of-arrays segment list; ground moved into the map frame with the frame's pose
// pose4: [n_frames][4] = x, y, cos(theta), sin(theta) or null (ground copied as is)
__global__ void k_map_pack_block(int n, int n_frames, const int* __restrict__ frame_offset, const uint8_t* __restrict__ code,
                                 const uint8_t* __restrict__ color, const uint8_t* __restrict__ keep,
                                 const double* __restrict__ ground, const int32_t* __restrict__ idx,
                                 const float* __restrict__ dist, const double* __restrict__ pose4, int step,
                                 uint8_t* __restrict__ block)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s == 0) {
        uint32_t* hd = reinterpret_cast<uint32_t*>(block);
        hd[0] = kMagic; hd[1] = (uint32_t)n; hd[2] = (uint32_t)step; hd[3] = (uint32_t)n_frames;
        for (int k = 4; k < kRow / 4; ++k) hd[k] = 0u;
    }
    if (s >= n) return;
    uint8_t* row = block + (size_t)(1 + s) * kRow;
    const uint4* c4 = reinterpret_cast<const uint4*>(code + (size_t)s * 32);
    uint4* r4 = reinterpret_cast<uint4*>(row);
    r4[0] = c4[0]; r4[1] = c4[1];
    double g[4] = { 0, 0, 0, 0 };
    if (ground) for (int k = 0; k < 4; ++k) g[k] = ground[(size_t)s * 4 + k];
    if (pose4 && frame_offset) {
        // frame of segment s: last f with frame_offset[f] <= s
        int lo = 0, hi = n_frames;
        while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (frame_offset[mid] <= s) lo = mid; else hi = mid; }
        const double x = pose4[4 * lo], y = pose4[4 * lo + 1], cs = pose4[4 * lo + 2], sn = pose4[4 * lo + 3];
        for (int e = 0; e < 2; ++e) {
            const double px = g[2 * e], py = g[2 * e + 1];
            const double a = cs * px, b = sn * py, c = sn * px, d = cs * py;
            g[2 * e] = x + (a - b);
            g[2 * e + 1] = y + (c + d);
        }
    }
    double* rg = reinterpret_cast<double*>(row + 32);
    for (int k = 0; k < 4; ++k) rg[k] = g[k];
    *reinterpret_cast<int32_t*>(row + 64) = idx ? idx[s] : -1;
    *reinterpret_cast<float*>(row + 68) = dist ? dist[s] : -1.f;
    row[72] = color ? color[s] : (uint8_t)255;
    row[73] = keep ? keep[s] : (uint8_t)1;
    row[74] = row[75] = row[76] = row[77] = row[78] = row[79] = 0;
}

// ---- update, step 1 of 3: classification.  kMapWg = 256 rows per workgroup (1024 until the end of round 4: a sixteen-wave
// workgroup waits for a CU with sixteen free wave slots, and next to other batches' region growing there is none -- map_update
// 0.016 ms alone, 0.39 ms in the pipeline), any number of workgroups (an 8-rank step is 8 x 16 Ki rows): every row becomes "refresh entry t", "append, rank r among this workgroup's appends" or nothing;
// refresh winners are elected (atomicMax) and hits counted here, the appends of each workgroup are counted for the
// scan in k_map_plan.  A block with a bad header, or one that carries the OVERFLOW marker (its rank had more
// segments than a block holds and sent only the header, lf_map_pack_block), makes EVERY workgroup skip the whole
// update: all replicas see the same blocks, so all of them skip the same step.
constexpr int kMapWg = 256;
__global__ __launch_bounds__(kMapWg) void k_map_classify(MapDev m, const uint8_t* __restrict__ blocks, int n_blocks, int block_rows,
                                                       int force_append, int* __restrict__ act, int* __restrict__ wg_count)
{
    __shared__ int wave_count[kMapWg / 64];
    __shared__ int bad_sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = block_rows - 1;
    const int total = n_blocks * G;
    const int size0 = m.state[0];
    if (tid == 0) bad_sh = 0;
    __syncthreads();
    for (int b = tid; b < n_blocks; b += kMapWg) {
        const uint32_t* hd = reinterpret_cast<const uint32_t*>(blocks + (size_t)b * block_rows * kRow);
        int f = 0;
        if (hd[0] != kMagic || hd[1] > (uint32_t)G) f |= 2;
        else if (hd[4] != 0u) f |= 4;
        if (f) atomicOr(&bad_sh, f);
    }
    __syncthreads();
    const int bad = bad_sh;
    const int s = blockIdx.x * kMapWg + tid;
    bool is_app = false, is_ref = false;
    int target = -1;
    if (!bad && s < total) {
        const int blk = s / G, r = s - blk * G;
        const uint8_t* bb = blocks + (size_t)blk * block_rows * kRow;
        const int count = (int)reinterpret_cast<const uint32_t*>(bb)[1];
        if (r < count) {
            const uint8_t* row = bb + (size_t)(1 + r) * kRow;
            const int idx = *reinterpret_cast<const int32_t*>(row + 64);
            const float dist = *reinterpret_cast<const float*>(row + 68);
            const bool eligible = !m.kept_only || row[73] != 0 || force_append;
            is_ref = eligible && !force_append && m.policy == LF_MAP_MERGE && idx >= 0 && idx < size0 && dist >= 0.f &&
                     dist <= (float)m.merge_distance;
            is_app = eligible && !is_ref;
            target = idx;
        }
    }
    const unsigned long long bal = __ballot(is_app);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_count[wave] = __popcll(bal);
    const unsigned long long rbal = __ballot(is_ref);
    __syncthreads();
    int off = 0, all = 0;
    for (int w = 0; w < kMapWg / 64; ++w) { const int c = wave_count[w]; if (w < wave) off += c; all += c; }
    if (s < total) act[s] = is_ref ? (target | kRefFlag) : (is_app ? off + before : -1);
    if (is_ref) {
        atomicMax(&m.winner[target], s);
        atomicAdd(&m.hits[target], 1);
    }
    if (lane == 0 && rbal) atomicAdd(&m.state[9], __popcll(rbal));
    if (tid == 0) {
        wg_count[blockIdx.x] = all;
        if (blockIdx.x == 0 && bad) m.state[10] = bad;
    }
}

// ---- step 2 of 3: one workgroup scans the workgroups' append counts (wg_count -> exclusive bases, in place) and
// advances the map's state
__global__ __launch_bounds__(kMapWg) void k_map_plan(MapDev m, const uint8_t* __restrict__ blocks, int n_blocks, int n_wg,
                                                   int* __restrict__ wg_count)
{
    __shared__ int wave_sum[kMapWg / 64];
    __shared__ int carry_sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_sh = 0;
    __syncthreads();
    for (int start = 0; start < n_wg; start += kMapWg) {
        const int i = start + tid;
        const int v = i < n_wg ? wg_count[i] : 0;
        int incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
        if (lane == 63) wave_sum[wave] = incl;
        __syncthreads();
        int off = carry_sh;
        for (int w = 0; w < wave; ++w) off += wave_sum[w];
        if (i < n_wg) wg_count[i] = off + incl - v;
        __syncthreads();
        if (tid == kMapWg - 1) carry_sh = off + incl;
        __syncthreads();
    }
    if (tid == 0) {
        // every block of one update carries the same step number; block 0's is taken
        const uint32_t* hd = reinterpret_cast<const uint32_t*>(blocks);
        const int step = n_blocks > 0 ? (int)hd[2] : 0;
        const int size0 = m.state[0], head0 = m.state[1];
        const int n_app = carry_sh, n_ref = m.state[9];
        int new_size, new_head, flags = m.state[10];
        if (m.when_full == LF_MAP_RING) {
            const long long sz = (long long)size0 + n_app;
            new_size = sz > m.capacity ? m.capacity : (int)sz;
            new_head = (int)(((long long)head0 + n_app) % m.capacity);
        } else {
            const int fit = n_app < m.capacity - size0 ? n_app : m.capacity - size0;
            if (fit < n_app) flags |= 1;
            new_size = size0 + fit;
            new_head = new_size % m.capacity;
        }
        m.state[3] = n_app; m.state[4] = n_ref; m.state[5] = head0; m.state[6] = size0; m.state[7] = step;
        m.state[0] = new_size; m.state[1] = new_head;
        // error events: the flags of the latest failing update + a running count the host compares with what it has
        // already reported (nothing is sticky: the map stays usable)
        if (flags) { m.state[2] = flags; m.state[8] += 1; }
        m.state[9] = 0; m.state[10] = 0;
        m.totals[0] += (unsigned long long)n_app;
        m.totals[1] += (unsigned long long)n_ref;
    }
}

// 32 threads per row (one per code byte): write the entry, its packed operands and, from thread 0, the rest
// ---- step 3 of 3
__global__ void k_map_apply(MapDev m, const uint8_t* __restrict__ blocks, int n_blocks, int block_rows,
                            const int* __restrict__ act, const int* __restrict__ wg_base)
{
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const int G = block_rows - 1;
    const size_t total = (size_t)n_blocks * G;
    const int s = (int)(t >> 5), b = (int)(t & 31);
    if ((size_t)s >= total) return;
    int a = act[s];
    if (a == -1) return;
    if (!(a & kRefFlag)) a += wg_base[s / kMapWg];
    const int n_app = m.state[3], head0 = m.state[5], size0 = m.state[6], step = m.state[7];
    const int cap = m.capacity;
    int pos;
    bool fresh;
    if (a & kRefFlag) {
        pos = a & ~kRefFlag;
        if (m.winner[pos] != s) return;
        bool overwritten = false;
        if (m.when_full == LF_MAP_RING) {
            int rel = pos - head0; if (rel < 0) rel += cap;
            overwritten = n_app >= cap || rel < n_app;
        }
        if (overwritten) { if (b == 0) m.winner[pos] = -1; return; }
        fresh = false;
    } else {
        if (m.when_full == LF_MAP_RING) {
            if (n_app > cap && a < n_app - cap) return;            // a later append of this very update lands here
            pos = (int)(((long long)head0 + a) % cap);
        } else {
            if (a >= cap - size0) return;                          // does not fit: overflow was flagged by the plan
            pos = size0 + a;
        }
        fresh = true;
    }
    const int blk = s / G, r = s - blk * G;
    const uint8_t* row = blocks + ((size_t)blk * block_rows + 1 + r) * kRow;
    const uint32_t byte = row[b];
    m.code[(size_t)pos * 32 + b] = (uint8_t)byte;
    const uint32_t w0 = ((byte & 15u) * 0x00204081u) & 0x01010101u;
    const uint32_t w1 = ((byte >> 4) * 0x00204081u) & 0x01010101u;
    if (m.fp4) *reinterpret_cast<uint32_t*>(m.mx + assoc_map_offset_fp4((size_t)pos, 4 * b)) = assoc_fp4_expand(byte);
    else *reinterpret_cast<uint2*>(m.mx + assoc_map_offset((size_t)pos, 8 * b)) = make_uint2((w0 * 0xE0u) ^ 0x10101010u, (w1 * 0xE0u) ^ 0x10101010u);
    if (b < 8 && m.fp4) {
        // colour row of the FP4 gated kernel: -6.0 (e2m1 0xF) in the nibbles 0..2 of the OTHER colours, first dword; zeros after
        const int c = row[72];
        uint32_t w = 0;
        if (b == 0 && c < 3)
            for (int g = 0; g < 3; ++g) if (g != c) w |= 0xFu << (4 * g);
        *reinterpret_cast<uint32_t*>(m.mcx + (size_t)pos * 32 + 4 * b) = w;
    }
    if (b < 8 && !m.fp4) {
        // the ninth-step operand of a map row: bytes 0,1 zero (block counter, filled in by k_assoc), then -127 in the
        // ten bytes of each OTHER colour's group
        const int c = row[72];
        uint32_t w = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int pos_k = 4 * b + k;
            const int g = pos_k >= 2 ? (pos_k - 2) / 10 : -1;
            if (g >= 0 && c < 3 && g != c) w |= 0x81u << (8 * k);
        }
        *reinterpret_cast<uint32_t*>(m.mcx + (size_t)pos * 32 + 4 * b) = w;
    }
    if (b == 0) {
        m.color[pos] = row[72];
        const double* rg = reinterpret_cast<const double*>(row + 32);
        for (int k = 0; k < 4; ++k) m.ground[(size_t)pos * 4 + k] = rg[k];
        m.last_seen[pos] = step;
        if (fresh) m.hits[pos] = 1; else m.winner[pos] = -1;
    }
}

// rows for lf_map_seed: plain arrays -> a block whose rows are all "append me"
__global__ void k_map_seed_block(int n, const uint8_t* __restrict__ code, const uint8_t* __restrict__ color,
                                 const double* __restrict__ ground, uint8_t* __restrict__ block)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s == 0) {
        uint32_t* hd = reinterpret_cast<uint32_t*>(block);
        hd[0] = kMagic; hd[1] = (uint32_t)n; hd[2] = (uint32_t)-1; hd[3] = 0u;
        for (int k = 4; k < kRow / 4; ++k) hd[k] = 0u;
    }
    if (s >= n) return;
    uint8_t* row = block + (size_t)(1 + s) * kRow;
    for (int k = 0; k < 32; ++k) row[k] = code[(size_t)s * 32 + k];
    double* rg = reinterpret_cast<double*>(row + 32);
    for (int k = 0; k < 4; ++k) rg[k] = ground ? ground[(size_t)s * 4 + k] : 0.0;
    *reinterpret_cast<int32_t*>(row + 64) = -1;
    *reinterpret_cast<float*>(row + 68) = -1.f;
    row[72] = color ? color[s] : (uint8_t)255;
    row[73] = 1;
    row[74] = row[75] = row[76] = row[77] = row[78] = row[79] = 0;
}

__global__ void k_fill_i32(int* p, size_t n, int v)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

void launch_fill_i32(int* p, size_t n, int v, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_fill_i32, dim3(1024), dim3(256), 0, s, p, n, v);
}

void launch_map_pack_block(int n, int n_frames, const int* frame_offset, const uint8_t* code, const uint8_t* color,
                           const uint8_t* keep, const double* ground, const int32_t* idx, const float* dist,
                           const double* pose4, int step, uint8_t* block, hipStream_t s)
{
    const int threads = n > 0 ? n : 1;
    hipLaunchKernelGGL(k_map_pack_block, dim3((threads + 255) / 256), dim3(256), 0, s, n, n_frames, frame_offset, code, color,
                       keep, ground, idx, dist, pose4, step, block);
}

__global__ void k_map_overflow_block(int n, int n_frames, int step, uint8_t* __restrict__ block)
{
    uint32_t* hd = reinterpret_cast<uint32_t*>(block);
    const int k = threadIdx.x;
    if (k < kRow / 4) hd[k] = k == 0 ? kMagic : k == 2 ? (uint32_t)step : k == 3 ? (uint32_t)n_frames : k == 4 ? (uint32_t)n : 0u;
}

void launch_map_overflow_block(int n, int n_frames, int step, uint8_t* block, hipStream_t s)
{
    hipLaunchKernelGGL(k_map_overflow_block, dim3(1), dim3(64), 0, s, n, n_frames, step, block);
}

void launch_map_seed_block(int n, const uint8_t* code, const uint8_t* color, const double* ground, uint8_t* block, hipStream_t s)
{
    const int threads = n > 0 ? n : 1;
    hipLaunchKernelGGL(k_map_seed_block, dim3((threads + 255) / 256), dim3(256), 0, s, n, code, color, ground, block);
}

void launch_map_update(const MapDevice& md, const uint8_t* blocks, int n_blocks, int block_rows, int force_append, int* act,
                       hipStream_t s)
{
    MapDev m;
    m.capacity = md.capacity; m.policy = md.policy; m.kept_only = md.kept_only; m.merge_distance = md.merge_distance;
    m.when_full = md.when_full; m.fp4 = md.fp4;
    m.code = md.code; m.color = md.color; m.ground = md.ground; m.hits = md.hits; m.last_seen = md.last_seen;
    m.winner = md.winner; m.mx = md.mx; m.mcx = md.mcx; m.state = md.state; m.totals = md.totals;
    // act: [rows] actions, then [n_wg] append counts / bases of the classification workgroups
    const size_t rows = (size_t)n_blocks * (block_rows - 1);
    const int n_wg = (int)((rows + kMapWg - 1) / kMapWg);
    int* wg_count = act + rows;
    if (n_wg)
        hipLaunchKernelGGL(k_map_classify, dim3(n_wg), dim3(kMapWg), 0, s, m, blocks, n_blocks, block_rows, force_append, act, wg_count);
    hipLaunchKernelGGL(k_map_plan, dim3(1), dim3(kMapWg), 0, s, m, blocks, n_blocks, n_wg, wg_count);
    const size_t threads = rows * 32;
    if (threads)
        hipLaunchKernelGGL(k_map_apply, dim3((threads + 255) / 256), dim3(256), 0, s, m, blocks, n_blocks, block_rows, act, wg_count);
}

}  // namespace lf
