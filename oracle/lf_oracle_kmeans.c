/* ORACLE (test infrastructure only; never linked into the product).
 *
 * Anti-instagram colour clustering: /root/reference/src/anti_instagram/include/anti_instagram/kmeans.py:14-47
 *   runKMeans(cv_img, num_colors, init):  imgdata = pixels of the LAST 100 ROWS as [N][3] (B, G, R);
 *   KMeans(n_clusters = num_colors, max_iter = 25, n_init = 10, init = <array>).fit_predict(imgdata);
 *   returns cluster_centers_, the label counts and score(imgdata) = -inertia.
 * The arithmetic is scikit-learn's (a pip dependency of the reference, not vendored, version not pinned by the
 * reference; golden vectors were generated with the 1.7.2 installed in the build image, tests/golden/make_golden.py).
 * Restated from sklearn/cluster/_kmeans.py (KMeans.fit, _kmeans_single_lloyd, _tolerance, score), as published:
 *   - X -> float64; an explicit init array means ONE run (n_init is ignored);
 *   - tol = 1e-4 * mean over the features of the variance of X (population variance);
 *   - X and the init centres are centred on the feature means for the iterations, the centres are shifted back;
 *   - one Lloyd iteration: labels = argmin_k (|c_k|^2 - 2 x.c_k) (first minimum wins) against the OLD centres,
 *     new centres = mean of the members, shift_k = |new_k - old_k|.  A cluster WITHOUT members (frames that lack one of
 *     the reference's three or four colours: common) is re-seeded as _relocate_empty_clusters_dense does: the samples
 *     farthest from their own (old) centre, one per empty cluster in ascending cluster order, leave their cluster and
 *     become the empty cluster's only member.  sklearn picks them with numpy.argpartition, whose order among equal
 *     distances is an implementation detail; here: largest distance first, the lowest index among equals (on images
 *     equal distances come from equal pixels, so the result is the same);
 *   - stop when the labels did not change ("strict convergence"), or when sum_k shift_k^2 <= tol (then one more
 *     labelling pass against the final centres), or after max_iter iterations (same extra pass);
 *   - score = -(sum over the points of |x - c_label|^2) with labels recomputed against the final, uncentred centres.
 * Two places where scikit-learn's result depends on the ORDER of floating-point operations are given one fixed form here
 * (the product computes the same form, so product and oracle agree bit for bit):
 *   - the distances come out of BLAS dgemm there (pairwise = |c|^2 + (-2) X.C^T); with integer pixels and the reference's
 *     integer inits the first iteration has EXACT ties (36 of 16 000 points on a random image), which the rounding of that
 *     product breaks.  d = csq + (-2 * fma(x2, c2, fma(x1, c1, x0 * c0))) reproduces OpenBLAS's result on this host bit
 *     for bit (checked on the golden images: identical labels after one iteration);
 *   - the per-cluster sums: sklearn adds the centred float64 samples in chunks of 256 per thread; here the integer pixel
 *     sums are exact and new centre = S / count - mean (the difference is ~1e-13 relative; pinned to 1e-9 relative,
 *     label counts exactly).  The variance (only the tolerance depends on it) and the inertia use the exact integer sums
 *     the same way: var_d = sum(x^2) / n - mean_d^2, inertia = sum|x|^2 - 2 sum_k c_k.S_k + sum_k n_k |c_k|^2. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "lf_oracle.h"

static int label_of(const double* x, const double* c, int k)
{
    int best = 0;
    double bd = 0.0;
    for (int j = 0; j < k; ++j) {
        const double* cj = c + 3 * j;
        const double csq = cj[0] * cj[0] + cj[1] * cj[1] + cj[2] * cj[2];
        const double acc = fma(x[2], cj[2], fma(x[1], cj[1], x[0] * cj[0]));
        const double d = csq + (-2.0 * acc);
        if (j == 0 || d < bd) { bd = d; best = j; }
    }
    return best;
}

/* bgr: [n][3] u8 (the caller passes the last 100 rows).  init: [k][3].  Returns the number of iterations, or -1 when a
 * cluster runs empty / on bad arguments.  counts: [k] members per cluster; inertia: sum of squared distances. */
int lfo_kmeans(const uint8_t* bgr, int n, int k, const double* init, int max_iter, double tol_rel, double* centers,
               int64_t* counts, double* inertia)
{
    if (n < 1 || k < 1 || k > 16) return -1;
    uint8_t* lab = (uint8_t*)malloc((size_t)n);
    if (!lab) return -1;
    int64_t s1[3] = { 0, 0, 0 }, s2[3] = { 0, 0, 0 };
    for (int i = 0; i < n; ++i) for (int d = 0; d < 3; ++d) { const int64_t v = bgr[3 * i + d]; s1[d] += v; s2[d] += v * v; }
    double mean[3], var = 0.0;
    for (int d = 0; d < 3; ++d) { mean[d] = (double)s1[d] / (double)n; var += (double)s2[d] / (double)n - mean[d] * mean[d]; }
    const double tol = tol_rel * (var / 3.0);
    double c[16 * 3], cn[16 * 3];
    int64_t sum[16 * 3], cnt[16];
    for (int j = 0; j < k; ++j) for (int d = 0; d < 3; ++d) c[3 * j + d] = init[3 * j + d] - mean[d];
    memset(lab, 0xff, (size_t)n);
    int it = 0, strict = 0, bad = 0;
    for (it = 0; it < max_iter; ++it) {
        memset(sum, 0, sizeof(sum));
        memset(cnt, 0, sizeof(cnt));
        int64_t changed = 0;
        for (int i = 0; i < n; ++i) {
            const double x[3] = { (double)bgr[3 * i] - mean[0], (double)bgr[3 * i + 1] - mean[1], (double)bgr[3 * i + 2] - mean[2] };
            const int l = label_of(x, c, k);
            changed += l != (int)lab[i];
            lab[i] = (uint8_t)l;
            cnt[l] += 1;
            for (int d = 0; d < 3; ++d) sum[3 * l + d] += bgr[3 * i + d];
        }
        int n_empty = 0;
        for (int j = 0; j < k; ++j) n_empty += cnt[j] == 0;
        if (n_empty) {
            /* the n_empty samples farthest from their own old centre (descending distance, lowest index among equals) */
            int far[16];
            for (int e = 0; e < n_empty; ++e) {
                double bestd = -1.0; int besti = -1;
                for (int i = 0; i < n; ++i) {
                    int taken = 0;
                    for (int f = 0; f < e; ++f) taken |= far[f] == i;
                    if (taken) continue;
                    const double* co = c + 3 * lab[i];
                    const double t0 = ((double)bgr[3 * i] - mean[0]) - co[0], t1 = ((double)bgr[3 * i + 1] - mean[1]) - co[1], t2 = ((double)bgr[3 * i + 2] - mean[2]) - co[2];
                    const double dd = (t0 * t0 + t1 * t1) + t2 * t2;
                    if (dd > bestd) { bestd = dd; besti = i; }
                }
                far[e] = besti;
            }
            int e = 0;
            for (int j = 0; j < k; ++j) {
                if (cnt[j] != 0) continue;
                const int i = far[e++], o = lab[i];
                if (i < 0) { bad = 1; break; }
                for (int d = 0; d < 3; ++d) { sum[3 * o + d] -= bgr[3 * i + d]; sum[3 * j + d] = bgr[3 * i + d]; }
                cnt[o] -= 1; cnt[j] = 1;
            }
        }
        double shift_tot = 0.0;
        for (int j = 0; j < k; ++j) {
            if (cnt[j] == 0) { bad = 1; break; }
            double q = 0.0;
            for (int d = 0; d < 3; ++d) { cn[3 * j + d] = (double)sum[3 * j + d] / (double)cnt[j] - mean[d]; const double t = cn[3 * j + d] - c[3 * j + d]; q += t * t; }
            const double shift = sqrt(q);
            shift_tot += shift * shift;
        }
        if (bad) break;
        memcpy(c, cn, sizeof(double) * 3 * (size_t)k);
        if (changed == 0) { strict = 1; ++it; break; }
        if (shift_tot <= tol) { ++it; break; }
    }
    if (!bad) {
        for (int j = 0; j < k; ++j) for (int d = 0; d < 3; ++d) centers[3 * j + d] = c[3 * j + d] + mean[d];
        /* labels against the final centres: the iteration's own when it stopped on unchanged labels, one more centred pass
         * otherwise (fit); score() then labels once more against the uncentred centres, which is what inertia uses */
        memset(cnt, 0, sizeof(cnt));
        for (int i = 0; i < n; ++i) {
            int l = lab[i];
            if (!strict) {
                const double x[3] = { (double)bgr[3 * i] - mean[0], (double)bgr[3 * i + 1] - mean[1], (double)bgr[3 * i + 2] - mean[2] };
                l = label_of(x, c, k);
            }
            cnt[l] += 1;
        }
        for (int j = 0; j < k; ++j) counts[j] = cnt[j];
        memset(sum, 0, sizeof(sum));
        memset(cnt, 0, sizeof(cnt));
        for (int i = 0; i < n; ++i) {
            const double x[3] = { (double)bgr[3 * i], (double)bgr[3 * i + 1], (double)bgr[3 * i + 2] };
            const int l = label_of(x, centers, k);
            cnt[l] += 1;
            for (int d = 0; d < 3; ++d) sum[3 * l + d] += bgr[3 * i + d];
        }
        double in = (double)(s2[0] + s2[1] + s2[2]);
        for (int j = 0; j < k; ++j) {
            const double* cj = centers + 3 * j;
            in -= 2.0 * (cj[0] * (double)sum[3 * j] + cj[1] * (double)sum[3 * j + 1] + cj[2] * (double)sum[3 * j + 2]);
            in += (double)cnt[j] * (cj[0] * cj[0] + cj[1] * cj[1] + cj[2] * cj[2]);
        }
        *inertia = in;
    }
    free(lab);
    return bad ? -1 : it;
}
