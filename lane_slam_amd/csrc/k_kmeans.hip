// Anti-instagram colour clustering (k-means part).
//
// Reference: /root/reference/src/anti_instagram/include/anti_instagram/kmeans.py:14-47 -- runKMeans() =
// sklearn.cluster.KMeans(n_clusters, max_iter = 25, init = <array>).fit_predict on the B, G, R pixels of the frame's last
// 100 rows, then cluster_centers_, the label counts and score() = -inertia.  Arithmetic: scikit-learn's Lloyd iteration
// (centred data, tolerance = 1e-4 x mean feature variance, strict-convergence test on the labels, empty clusters re-seeded
// with the farthest samples), restated in oracle/lf_oracle_kmeans.c, which is pinned against the reference's own function
// (tests/golden/kmeans.npz).  This kernel computes the SAME fixed arithmetic as the oracle, so the two agree bit for bit:
// distances as csq + (-2 fma(x2, c2, fma(x1, c1, x0 c0))) (what the BLAS product gives, ties included), every sum over
// samples as an exact integer sum (order free: LDS atomics), the handful of float64 operations per iteration by one thread.
//
// One workgroup of 1024 threads runs the whole fit: a frame's strip is 16 000 .. 192 000 points, a Lloyd iteration over it
// a few microseconds, and the iterations are strictly sequential -- there is nothing for a second workgroup to do.
#include "common.h"

namespace lf {

constexpr int KM_T = 1024;
constexpr int KM_MAXK = 16;

__device__ __forceinline__ int km_label(double x0, double x1, double x2, const double* c, int k)
{
    int best = 0;
    double bd = 0.0;
    for (int j = 0; j < k; ++j) {
        const double c0 = c[3 * j], c1 = c[3 * j + 1], c2 = c[3 * j + 2];
        const double csq = c0 * c0 + c1 * c1 + c2 * c2;
        const double acc = fma(x2, c2, fma(x1, c1, x0 * c0));
        const double d = csq + (-2.0 * acc);
        if (j == 0 || d < bd) { bd = d; best = j; }
    }
    return best;
}


// Per-cluster integer sums without hammering three or four LDS words with one atomic per sample: for k <= 4 (the reference's
// inits) every thread keeps count + B, G, R sums per cluster in registers, a wave adds them up with lane shuffles and lane 0
// issues one LDS atomic per value; larger k falls back to an atomic per sample.
struct KmAcc {
    unsigned int v[16];        // [cluster][count, b, g, r]
    __device__ __forceinline__ void clear() { for (int i = 0; i < 16; ++i) v[i] = 0u; }
    __device__ __forceinline__ void add(int l, unsigned b0, unsigned b1, unsigned b2)
    {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool m = l == j;
            v[4 * j] += m ? 1u : 0u; v[4 * j + 1] += m ? b0 : 0u; v[4 * j + 2] += m ? b1 : 0u; v[4 * j + 3] += m ? b2 : 0u;
        }
    }
    __device__ __forceinline__ void flush(unsigned long long* cnt, unsigned long long* sum, int k)
    {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            unsigned int x = v[i];
            for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d);
            v[i] = x;
        }
        if ((threadIdx.x & 63) == 0)
            for (int j = 0; j < k && j < 4; ++j) {
                if (v[4 * j]) atomicAdd(&cnt[j], (unsigned long long)v[4 * j]);
                for (int d = 0; d < 3; ++d) if (v[4 * j + 1 + d]) atomicAdd(&sum[3 * j + d], (unsigned long long)v[4 * j + 1 + d]);
            }
    }
};

// out: [0 .. 3k) centres, [3k] inertia; counts: [k]; status: [0] iterations (or -1: a cluster stayed empty), lab: [n] scratch
__global__ __launch_bounds__(KM_T) void k_kmeans(const uint8_t* __restrict__ bgr, int n, int k, const double* __restrict__ init,
                                                 int max_iter, double tol_rel, uint8_t* __restrict__ lab, double* __restrict__ out,
                                                 long long* __restrict__ counts, int* __restrict__ status)
{
    __shared__ unsigned long long sum[KM_MAXK * 3], cnt[KM_MAXK], s12[6];
    __shared__ unsigned long long changed;
    __shared__ double c[KM_MAXK * 3], mean[3], cfin[KM_MAXK * 3];
    __shared__ double far_d[KM_T];
    __shared__ int far_i[KM_T], far_pick[KM_MAXK];
    __shared__ int stop, n_empty;
    const int tid = threadIdx.x;
    if (tid < 6) s12[tid] = 0ull;
    __syncthreads();
    {
        unsigned long long a[6] = { 0, 0, 0, 0, 0, 0 };
        for (int i = tid; i < n; i += KM_T)
            for (int d = 0; d < 3; ++d) { const unsigned long long v = bgr[3 * (size_t)i + d]; a[d] += v; a[3 + d] += v * v; }
        for (int d = 0; d < 6; ++d) if (a[d]) atomicAdd(&s12[d], a[d]);
    }
    __syncthreads();
    __shared__ double tol;
    if (tid == 0) {
        double var = 0.0;
        for (int d = 0; d < 3; ++d) { mean[d] = (double)(long long)s12[d] / (double)n; var += (double)(long long)s12[3 + d] / (double)n - mean[d] * mean[d]; }
        tol = tol_rel * (var / 3.0);
        for (int j = 0; j < k; ++j) for (int d = 0; d < 3; ++d) c[3 * j + d] = init[3 * j + d] - mean[d];
        stop = 0;
    }
    for (int i = tid; i < n; i += KM_T) lab[i] = 0xff;
    __syncthreads();
    int it = 0, strict = 0, bad = 0;
    for (it = 0; it < max_iter; ++it) {
        if (tid < KM_MAXK * 3) sum[tid] = 0ull;
        if (tid < KM_MAXK) cnt[tid] = 0ull;
        if (tid == 0) changed = 0ull;
        __syncthreads();
        {
            unsigned long long ch = 0;
            KmAcc acc; acc.clear();
            for (int i = tid; i < n; i += KM_T) {
                const unsigned b0 = bgr[3 * (size_t)i], b1 = bgr[3 * (size_t)i + 1], b2 = bgr[3 * (size_t)i + 2];
                const int l = km_label((double)b0 - mean[0], (double)b1 - mean[1], (double)b2 - mean[2], c, k);
                ch += l != (int)lab[i];
                lab[i] = (uint8_t)l;
                if (k <= 4) acc.add(l, b0, b1, b2);
                else {
                    atomicAdd(&cnt[l], 1ull);
                    atomicAdd(&sum[3 * l], (unsigned long long)b0);
                    atomicAdd(&sum[3 * l + 1], (unsigned long long)b1);
                    atomicAdd(&sum[3 * l + 2], (unsigned long long)b2);
                }
            }
            if (k <= 4) acc.flush(cnt, sum, k);
            if (ch) atomicAdd(&changed, ch);
        }
        __syncthreads();
        if (tid == 0) { int e = 0; for (int j = 0; j < k; ++j) e += cnt[j] == 0ull; n_empty = e; }
        __syncthreads();
        if (n_empty) {
            // re-seed the empty clusters with the samples farthest from their own old centre: largest distance first, the
            // lowest index among equals; one pick at a time (n_empty <= k)
            for (int e = 0; e < n_empty; ++e) {
                double bd = -1.0; int bi = -1;
                for (int i = tid; i < n; i += KM_T) {
                    bool taken = false;
                    for (int f = 0; f < e; ++f) taken |= far_pick[f] == i;
                    if (taken) continue;
                    const double* co = c + 3 * lab[i];
                    const double t0 = ((double)bgr[3 * (size_t)i] - mean[0]) - co[0], t1 = ((double)bgr[3 * (size_t)i + 1] - mean[1]) - co[1],
                                 t2 = ((double)bgr[3 * (size_t)i + 2] - mean[2]) - co[2];
                    const double dd = (t0 * t0 + t1 * t1) + t2 * t2;
                    if (dd > bd) { bd = dd; bi = i; }
                }
                far_d[tid] = bd; far_i[tid] = bi;
                __syncthreads();
                if (tid == 0) {
                    double gd = -1.0; int gi = -1;
                    for (int t = 0; t < KM_T; ++t)
                        if (far_i[t] >= 0 && (far_d[t] > gd || (far_d[t] == gd && far_i[t] < gi))) { gd = far_d[t]; gi = far_i[t]; }
                    far_pick[e] = gi;
                }
                __syncthreads();
            }
            if (tid == 0) {
                int e = 0;
                for (int j = 0; j < k; ++j) {
                    if (cnt[j] != 0ull) continue;
                    const int i = far_pick[e++];
                    if (i < 0) continue;
                    const int o = lab[i];
                    for (int d = 0; d < 3; ++d) { sum[3 * o + d] -= bgr[3 * (size_t)i + d]; sum[3 * j + d] = bgr[3 * (size_t)i + d]; }
                    cnt[o] -= 1ull; cnt[j] = 1ull;
                }
            }
            __syncthreads();
        }
        if (tid == 0) {
            double cn[KM_MAXK * 3];
            double shift_tot = 0.0;
            int b = 0;
            for (int j = 0; j < k; ++j) {
                if (cnt[j] == 0ull) { b = 1; break; }
                double q = 0.0;
                for (int d = 0; d < 3; ++d) {
                    cn[3 * j + d] = (double)(long long)sum[3 * j + d] / (double)(long long)cnt[j] - mean[d];
                    const double t = cn[3 * j + d] - c[3 * j + d];
                    q += t * t;
                }
                const double shift = dm::dsqrt(q);
                shift_tot += shift * shift;
            }
            if (b) stop = 3;
            else {
                for (int j = 0; j < 3 * k; ++j) c[j] = cn[j];
                stop = changed == 0ull ? 1 : (shift_tot <= tol ? 2 : 0);
            }
        }
        __syncthreads();
        const int st = stop;
        if (st == 3) { bad = 1; break; }
        if (st == 1) { strict = 1; ++it; break; }
        if (st == 2) { ++it; break; }
    }
    if (bad) { if (tid == 0) status[0] = -1; return; }
    // counts: the iteration's own labels when it stopped on unchanged labels, one more (centred) labelling otherwise
    if (tid < KM_MAXK) cnt[tid] = 0ull;
    if (tid < 3 * k) cfin[tid] = c[tid] + mean[tid % 3];
    __syncthreads();
    {
        KmAcc acc; acc.clear();
        for (int i = tid; i < n; i += KM_T) {
            int l = lab[i];
            if (!strict) l = km_label((double)bgr[3 * (size_t)i] - mean[0], (double)bgr[3 * (size_t)i + 1] - mean[1], (double)bgr[3 * (size_t)i + 2] - mean[2], c, k);
            if (k <= 4) acc.add(l, 0u, 0u, 0u); else atomicAdd(&cnt[l], 1ull);
        }
        if (k <= 4) acc.flush(cnt, sum, k);
    }
    __syncthreads();
    if (tid < k) counts[tid] = (long long)cnt[tid];
    __syncthreads();
    // score(): labels against the final, uncentred centres; inertia from the exact integer sums
    if (tid < KM_MAXK * 3) sum[tid] = 0ull;
    if (tid < KM_MAXK) cnt[tid] = 0ull;
    __syncthreads();
    {
        KmAcc acc; acc.clear();
        for (int i = tid; i < n; i += KM_T) {
            const unsigned b0 = bgr[3 * (size_t)i], b1 = bgr[3 * (size_t)i + 1], b2 = bgr[3 * (size_t)i + 2];
            const int l = km_label((double)b0, (double)b1, (double)b2, cfin, k);
            if (k <= 4) acc.add(l, b0, b1, b2);
            else {
                atomicAdd(&cnt[l], 1ull);
                atomicAdd(&sum[3 * l], (unsigned long long)b0);
                atomicAdd(&sum[3 * l + 1], (unsigned long long)b1);
                atomicAdd(&sum[3 * l + 2], (unsigned long long)b2);
            }
        }
        if (k <= 4) acc.flush(cnt, sum, k);
    }
    __syncthreads();
    if (tid == 0) {
        double in = (double)(long long)(s12[3] + s12[4] + s12[5]);
        for (int j = 0; j < k; ++j) {
            const double* cj = cfin + 3 * j;
            in -= 2.0 * (cj[0] * (double)(long long)sum[3 * j] + cj[1] * (double)(long long)sum[3 * j + 1] + cj[2] * (double)(long long)sum[3 * j + 2]);
            in += (double)(long long)cnt[j] * (cj[0] * cj[0] + cj[1] * cj[1] + cj[2] * cj[2]);
        }
        for (int j = 0; j < 3 * k; ++j) out[j] = cfin[j];
        out[3 * k] = in;
        status[0] = it;
    }
}

void launch_kmeans(const uint8_t* bgr, int n, int k, const double* init, int max_iter, double tol_rel, uint8_t* lab, double* out,
                   long long* counts, int* status, hipStream_t s)
{
    hipLaunchKernelGGL(k_kmeans, dim3(1), dim3(KM_T), 0, s, bgr, n, k, init, max_iter, tol_rel, lab, out, counts, status);
}

}  // namespace lf
