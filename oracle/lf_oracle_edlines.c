/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see lf_oracle.h).
 * the EDLines detector and the multi-octave KeyLine / LBD path of the reference's line_descriptor
 * library, restated from the in-tree C++ (never built by the reference; it includes OpenCV headers, so it cannot be
 * compiled in this image without stand-ins -> PARITY UNPINNED):
 *   /root/reference/src/line_descriptor/src/binary_descriptor_custom.cpp
 *     :1374-1385  EDLineDetector defaults          :1442-2240  EdgeDrawing (gradient, anchors, smart routing)
 *     :2242-2482  EDline (line fitting / extension) :2484-2643  LeastSquaresLineFit_ (two forms)
 *     :2645-2726  LineValidation_ (direction, NFA)  :2728-2751  EDline(image): salience
 *     :689-1024   OctaveKeyLines                    :455-513    detectImpl (KeyLine fill)
 *     :263-301    operator() = detect + compute     :350-371    computeGaussianPyramid (compute-only path)
 *   /root/reference/src/line_descriptor/include/line_descriptor/descriptor_custom.hpp:443-445,630-813 (nfa, log_gamma)
 * The OpenCV calls inside (GaussianBlur, Sobel, threshold, Mat / 4, compare, resize INTER_LINEAR, pyrDown, Mat_<float>
 * products) are restated from the published OpenCV 3.0-3.3 implementation, also unpinned:
 *   GaussianBlur u8     getGaussianKernel(5, sigma, CV_32F) rounded to 8 fractional bits, rows then columns in int32,
 *                       (acc + 2^15) >> 16, BORDER_REFLECT_101 (the form lfo_gaussian5_u8 already follows)
 *   Mat(CV_16S) / 4     convertTo with alpha 0.25: round half to even
 *   resize INTER_LINEAR 11-bit fixed-point coefficients, ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2
 *   pyrDown u8          [1 4 6 4 1] x [1 4 6 4 1], (sum + 128) >> 8, BORDER_REFLECT_101
 *   Mat_<float> A * B.t(), A * v   cv::gemm with a double work type: every product sum here is a sum of integer
 *                       products (exact in double), rounded ONCE to float
 *
 * Deliberate deviations (the reference has undefined behaviour there, the oracle a defined failure):
 *   - EdgeDrawing writes its pixel / edge arrays before it checks their sizes (:1533-1537, :2185-2196); here a walk that
 *     would pass the array size fails the detection (-1), as the reference's later check would report.
 *   - lines.sId holds 5 * numOfEdges entries (:2257); more lines than that fail the detection (-1).
 *   - computeImpl's erase loop (:631-640) skips elements while erasing, so with three or more octaves it leaves
 *     "fictitious" lines behind whose octave index is out of range; here every real KeyLine gets its descriptor and
 *     nothing else does (what the loop does for one and two octaves).
 */
#include "lf_oracle.h"
#include "lf_detmath.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ED_HORIZONTAL 255   /* |dx| < |dy| */
#define ED_VERTICAL 0
enum { UpDir = 1, RightDir = 2, DownDir = 3, LeftDir = 4 };
#define TRY_TIME 6
#define SKIP_EDGE_POINT 2

static inline int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * n - 2 - p; }
    return p;
}

static int round_half_even_f(float v)          /* cvRound(float) */
{
    double d = (double)v;
    double f = floor(d);
    double r = d - f;
    long long i = (long long)f;
    if (r > 0.5 || (r == 0.5 && (i & 1))) i += 1;
    return (int)i;
}

static int round_half_even_d(double d)         /* cvRound(double) */
{
    double f = floor(d);
    double r = d - f;
    long long i = (long long)f;
    if (r > 0.5 || (r == 0.5 && (i & 1))) i += 1;
    return (int)i;
}

/* cv::getGaussianKernel(ksize, sigma, CV_32F) as the u8 separable filter uses it: taps * 256 rounded (createSeparableLinearFilter,
 * bits = 8).  sigma > 0 always here (OctaveKeyLines passes sqrt(curSigma2 - preSigma2), :707-708). */
void lfo_gaussian_taps_q8(int ksize, double sigma, int32_t* taps)
{
    float cf[32];
    const double sigmaX = sigma > 0 ? sigma : ((ksize - 1) * 0.5 - 1) * 0.3 + 0.8;
    const double scale2X = -0.5 / (sigmaX * sigmaX);
    double sum = 0;
    for (int i = 0; i < ksize; ++i) {
        const double x = i - (ksize - 1) * 0.5;
        const double t = lfo_exp(scale2X * x * x);
        cf[i] = (float)t;
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < ksize; ++i) {
        cf[i] = (float)(cf[i] * sum);
        taps[i] = round_half_even_f(cf[i] * 256.0f);
    }
}

/* cv::GaussianBlur(src, dst, Size(ksize, ksize), sigma) on u8 (binary_descriptor_custom.cpp:708) */
void lfo_gaussian_blur_u8(const uint8_t* src, int rows, int cols, int ksize, double sigma, uint8_t* dst)
{
    int32_t k[32];
    lfo_gaussian_taps_q8(ksize, sigma, k);
    const int r = ksize / 2;
    int* tmp = (int*)malloc((size_t)rows * cols * sizeof(int));
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            int s = 0;
            for (int j = -r; j <= r; ++j) s += k[j + r] * src[(size_t)y * cols + reflect101(x + j, cols)];
            tmp[(size_t)y * cols + x] = s;
        }
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            int s = 0;
            for (int j = -r; j <= r; ++j) s += k[j + r] * tmp[(size_t)reflect101(y + j, rows) * cols + x];
            int v = (s + (1 << 15)) >> 16;
            dst[(size_t)y * cols + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    free(tmp);
}

/* size cv::resize(src, dst, Size(), f, f) gives: saturate_cast<int>(n * f) */
void lfo_resize_size(int rows, int cols, double inv_scale, int* drows, int* dcols)
{
    *dcols = round_half_even_d(cols * inv_scale);
    *drows = round_half_even_d(rows * inv_scale);
}

/* cv::resize(src, dst, Size(), inv_scale, inv_scale) with INTER_LINEAR on u8 (binary_descriptor_custom.cpp:721) */
void lfo_resize_linear_u8(const uint8_t* src, int rows, int cols, double inv_scale, uint8_t* dst)
{
    int drows, dcols;
    lfo_resize_size(rows, cols, inv_scale, &drows, &dcols);
    const double scale = 1. / inv_scale;
    int* xofs = (int*)malloc((size_t)dcols * sizeof(int));
    short* ia = (short*)malloc((size_t)dcols * 2 * sizeof(short));
    for (int dx = 0; dx < dcols; ++dx) {
        float fx = (float)((dx + 0.5) * scale - 0.5);
        int sx = (int)floor((double)fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= cols - 1) { fx = 0; sx = cols - 1; }
        xofs[dx] = sx;
        ia[2 * dx] = (short)round_half_even_f((1.f - fx) * 2048.f);
        ia[2 * dx + 1] = (short)round_half_even_f(fx * 2048.f);
    }
    int* S0 = (int*)malloc((size_t)dcols * sizeof(int));
    int* S1 = (int*)malloc((size_t)dcols * sizeof(int));
    for (int dy = 0; dy < drows; ++dy) {
        float fy = (float)((dy + 0.5) * scale - 0.5);
        int sy = (int)floor((double)fy);
        fy -= sy;
        const short b0 = (short)round_half_even_f((1.f - fy) * 2048.f), b1 = (short)round_half_even_f(fy * 2048.f);
        int y0 = sy, y1 = sy + 1;
        y0 = y0 >= 0 ? (y0 < rows ? y0 : rows - 1) : 0;
        y1 = y1 >= 0 ? (y1 < rows ? y1 : rows - 1) : 0;
        for (int dx = 0; dx < dcols; ++dx) {
            const int sx = xofs[dx], sx1 = sx + 1 < cols ? sx + 1 : sx;
            S0[dx] = src[(size_t)y0 * cols + sx] * ia[2 * dx] + src[(size_t)y0 * cols + sx1] * ia[2 * dx + 1];
            S1[dx] = src[(size_t)y1 * cols + sx] * ia[2 * dx] + src[(size_t)y1 * cols + sx1] * ia[2 * dx + 1];
        }
        for (int dx = 0; dx < dcols; ++dx) {
            int v = (((b0 * (S0[dx] >> 4)) >> 16) + ((b1 * (S1[dx] >> 4)) >> 16) + 2) >> 2;
            dst[(size_t)dy * dcols + dx] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
    free(xofs); free(ia); free(S0); free(S1);
}

/* cv::pyrDown(src, dst, Size(cols / 2, rows / 2)) on u8 (binary_descriptor_custom.cpp:366, reductionRatio 2) */
void lfo_pyrdown_u8(const uint8_t* src, int rows, int cols, uint8_t* dst)
{
    const int drows = rows / 2, dcols = cols / 2;
    static const int k[5] = { 1, 4, 6, 4, 1 };
    int* row = (int*)malloc((size_t)rows * dcols * sizeof(int));
    for (int y = 0; y < rows; ++y)
        for (int dx = 0; dx < dcols; ++dx) {
            int s = 0;
            for (int j = -2; j <= 2; ++j) s += k[j + 2] * src[(size_t)y * cols + reflect101(2 * dx + j, cols)];
            row[(size_t)y * dcols + dx] = s;
        }
    for (int dy = 0; dy < drows; ++dy)
        for (int dx = 0; dx < dcols; ++dx) {
            int s = 0;
            for (int j = -2; j <= 2; ++j) s += k[j + 2] * row[(size_t)reflect101(2 * dy + j, rows) * dcols + dx];
            dst[(size_t)dy * dcols + dx] = (uint8_t)((s + 128) >> 8);
        }
    free(row);
}

/* ------------------------------------------------------------------ EdgeDrawing, gradient part (:1484-1496)
 * dx, dy = Sobel 3x3 (s16); sum = |dx| + |dy|; g = (sum > thr + 1 ? sum : 0) / 4; gwo = sum / 4 (both: round half to
 * even); dir = |dx| < |dy| ? 255 : 0 */
static inline short div4_half_even(int v)
{
    const int q = v >> 2, r = v & 3;
    return (short)(r < 2 ? q : r == 3 ? q + 1 : q + (q & 1));
}

void lfo_ed_gradient(const uint8_t* img, int rows, int cols, int gradient_threshold, int16_t* dx, int16_t* dy, int16_t* g,
                     int16_t* gwo, uint8_t* dir)
{
    lfo_sobel3_s16(img, rows, cols, dx, dy);
    for (size_t i = 0; i < (size_t)rows * cols; ++i) {
        const int ax = dx[i] < 0 ? -dx[i] : dx[i], ay = dy[i] < 0 ? -dy[i] : dy[i];
        const int sum = ax + ay;
        g[i] = div4_half_even(sum > gradient_threshold + 1 ? sum : 0);
        gwo[i] = div4_half_even(sum);
        dir[i] = ax < ay ? ED_HORIZONTAL : ED_VERTICAL;
    }
}

/* anchors, COLUMN-major scan (:1504-1532): returns the count, or -1 when it passes cap (= pixelNum / 5) */
int lfo_ed_anchors(const int16_t* g, const uint8_t* dir, int rows, int cols, int anchor_threshold, int scan, uint32_t* ax,
                   uint32_t* ay, int cap)
{
    int n = 0;
    for (unsigned w = 1; w + 1 < (unsigned)cols; w += scan)
        for (unsigned h = 1; h + 1 < (unsigned)rows; h += scan) {
            const int i = h * cols + w;
            int ok;
            if (dir[i] == ED_HORIZONTAL) ok = g[i] >= g[i - cols] + anchor_threshold && g[i] >= g[i + cols] + anchor_threshold;
            else ok = g[i] >= g[i - 1] + anchor_threshold && g[i] >= g[i + 1] + anchor_threshold;
            if (ok) {
                if (n >= cap) return -1;
                ax[n] = w; ay[n] = h; ++n;
            }
        }
    return n;
}

/* one walk of the smart routing (:1577-1720 and its three copies): marks and records pixels from (x, y) on while the
 * thresholded gradient is non-zero and the pixel is not an edge pixel yet */
typedef struct Walk {
    const int16_t* g; const uint8_t* dir; uint8_t* edge;
    int W, H;
    unsigned lastX, lastY;          /* live across walks and anchors, as in the reference */
} Walk;

static int walk(Walk* c, unsigned x, unsigned y, int lastDirection, uint32_t* px, uint32_t* py, unsigned* off, unsigned cap)
{
    const int W = c->W, H = c->H;
    const int16_t* pg = c->g;
    int i = y * W + x;
    while (pg[i] > 0 && !c->edge[i]) {
        c->edge[i] = 1;
        if (*off >= cap) return -1;
        px[*off] = x; py[(*off)++] = y;
        int should = 0, go;
        unsigned char g1, g2, g3;
        if (c->dir[i] == ED_HORIZONTAL) {
            if (lastDirection == UpDir || lastDirection == DownDir) should = x > c->lastX ? RightDir : LeftDir;
            c->lastX = x; c->lastY = y;
            if (lastDirection == RightDir || should == RightDir) go = RightDir;
            else if (lastDirection == LeftDir || should == LeftDir) go = LeftDir;
            else go = 0;
        } else {
            if (lastDirection == RightDir || lastDirection == LeftDir) should = y > c->lastY ? DownDir : UpDir;
            c->lastX = x; c->lastY = y;
            if (lastDirection == DownDir || should == DownDir) go = DownDir;
            else if (lastDirection == UpDir || should == UpDir) go = UpDir;
            else go = 0;
        }
        /* three neighbours ahead; the gradient values are compared as (unsigned char) of the short (:1607-1609) */
        if (go == RightDir) {
            if (x == (unsigned)W - 1 || y == 0 || y == (unsigned)H - 1) break;
            g1 = (unsigned char)pg[i - W + 1]; g2 = (unsigned char)pg[i + 1]; g3 = (unsigned char)pg[i + W + 1];
            if (g1 >= g2 && g1 >= g3) { x += 1; y -= 1; } else if (g3 >= g2 && g3 >= g1) { x += 1; y += 1; } else x += 1;
            lastDirection = RightDir;
        } else if (go == LeftDir) {
            if (x == 0 || y == 0 || y == (unsigned)H - 1) break;
            g1 = (unsigned char)pg[i - W - 1]; g2 = (unsigned char)pg[i - 1]; g3 = (unsigned char)pg[i + W - 1];
            if (g1 >= g2 && g1 >= g3) { x -= 1; y -= 1; } else if (g3 >= g2 && g3 >= g1) { x -= 1; y += 1; } else x -= 1;
            lastDirection = LeftDir;
        } else if (go == DownDir) {
            if (x == 0 || x == (unsigned)W - 1 || y == (unsigned)H - 1) break;
            g1 = (unsigned char)pg[i + W + 1]; g2 = (unsigned char)pg[i + W]; g3 = (unsigned char)pg[i + W - 1];
            if (g1 >= g2 && g1 >= g3) { x += 1; y += 1; } else if (g3 >= g2 && g3 >= g1) { x -= 1; y += 1; } else y += 1;
            lastDirection = DownDir;
        } else if (go == UpDir) {
            if (x == 0 || x == (unsigned)W - 1 || y == 0) break;
            g1 = (unsigned char)pg[i - W + 1]; g2 = (unsigned char)pg[i - W]; g3 = (unsigned char)pg[i - W - 1];
            if (g1 >= g2 && g1 >= g3) { x += 1; y -= 1; } else if (g3 >= g2 && g3 >= g1) { x -= 1; y -= 1; } else y -= 1;
            lastDirection = UpDir;
        }
        i = y * W + x;
    }
    return 0;
}

/* smart routing over the anchors in order + reassembly into chains (:1539-2236).  xcors / ycors [cap_px = pixelNum / 5
 * x 2], sid [max_edges + 2]; returns the number of edges or -1 */
int lfo_ed_link(const int16_t* g, const uint8_t* dir, int rows, int cols, const uint32_t* ax, const uint32_t* ay, int n_anchors,
                int min_line_len, uint32_t* xcors, uint32_t* ycors, uint32_t* sid, uint8_t* edge_out)
{
    const unsigned pixelNum = (unsigned)rows * cols, cap = pixelNum / 5, max_edges = cap / 20;
    uint8_t* edge = (uint8_t*)calloc(pixelNum, 1);
    uint32_t* fx = (uint32_t*)malloc((size_t)(cap + 1) * 4), *fy = (uint32_t*)malloc((size_t)(cap + 1) * 4);
    uint32_t* sx = (uint32_t*)malloc((size_t)(cap + 1) * 4), *sy = (uint32_t*)malloc((size_t)(cap + 1) * 4);
    uint32_t* fS = (uint32_t*)calloc((size_t)max_edges + 2, 4), *sS = (uint32_t*)calloc((size_t)max_edges + 2, 4);
    Walk c = { g, dir, edge, cols, rows, 0, 0 };
    unsigned offF = 0, offS = 0, ps = 0;
    int rc = 0;
    for (int a = 0; a < n_anchors && rc == 0; ++a) {
        const unsigned x = ax[a], y = ay[a];
        const int i = y * cols + x;
        if (edge[i]) continue;
        if (ps > max_edges) { rc = -1; break; }
        fS[ps] = offF;
        const int horizontal = dir[i] == ED_HORIZONTAL;
        if (walk(&c, x, y, horizontal ? RightDir : DownDir, fx, fy, &offF, cap)) { rc = -1; break; }
        edge[i] = 0;                        /* the anchor starts the second part as well */
        sS[ps] = offS;
        if (walk(&c, x, y, horizontal ? LeftDir : UpDir, sx, sy, &offS, cap)) { rc = -1; break; }
        const int lenF = (int)(offF - fS[ps]), lenS = (int)(offS - sS[ps]);
        if (lenF + lenS < min_line_len + 1) { offF = fS[ps]; offS = sS[ps]; }       /* short chain: dropped, its marks stay */
        else ps++;
    }
    if (rc == 0 && ps > max_edges) rc = -1;
    int n_edges = -1;
    if (rc == 0) {
        fS[ps] = offF; sS[ps] = offS;
        unsigned k = 0;
        for (unsigned e = 0; e < ps; ++e) {
            sid[e] = k;
            for (int t = (int)fS[e + 1] - 1; t >= (int)fS[e]; --t) { xcors[k] = fx[t]; ycors[k++] = fy[t]; }
            for (int t = (int)sS[e] + 1; t < (int)sS[e + 1]; ++t) { xcors[k] = sx[t]; ycors[k++] = sy[t]; }
        }
        sid[ps] = k;
        n_edges = (int)ps;
    }
    if (edge_out) memcpy(edge_out, edge, pixelNum);
    free(edge); free(fx); free(fy); free(sx); free(sy); free(fS); free(sS);
    return n_edges;
}

/* ------------------------------------------------------------------ nfa (descriptor_custom.hpp:630-813) */
static int double_equal(double a, double b)
{
    if (a == b) return 1;
    const double abs_diff = fabs(a - b), aa = fabs(a), bb = fabs(b);
    double abs_max = aa > bb ? aa : bb;
    if (abs_max < 2.2250738585072014e-308) abs_max = 2.2250738585072014e-308;
    return (abs_diff / abs_max) <= (100.0 * 2.2204460492503131e-16);
}

static double log_gamma_lanczos(double x)
{
    static const double q[7] = { 75122.6331530, 80916.6278952, 36308.2951477, 8687.24529705, 1168.92649479, 83.8676043424,
                                 2.50662827511 };
    double a = (x + 0.5) * lfo_log(x + 5.5) - (x + 5.5);
    double b = 0.0;
    for (int n = 0; n < 7; n++) {
        a -= lfo_log(x + (double)n);
        b += q[n] * lfo_pow(x, (double)n);
    }
    return a + lfo_log(b);
}

static double log_gamma_windschitl(double x)
{
    return 0.918938533204673 + (x - 0.5) * lfo_log(x) - x + 0.5 * x * lfo_log(x * lfo_sinh_small(1 / x) + 1 / (810.0 * lfo_pow(x, 6.0)));
}

static double log_gamma(double x) { return x > 15.0 ? log_gamma_windschitl(x) : log_gamma_lanczos(x); }

double lfo_ed_nfa(int n, int k, double p, double logNT)
{
    const double tolerance = 0.1;
    if (n == 0 || k == 0) return -logNT;
    if (n == k) return -logNT - (double)n * lfo_log10(p);
    const double p_term = p / (1.0 - p);
    const double log1term = log_gamma((double)n + 1.0) - log_gamma((double)k + 1.0) - log_gamma((double)(n - k) + 1.0)
                          + (double)k * lfo_log(p) + (double)(n - k) * lfo_log(1.0 - p);
    double term = lfo_exp(log1term);
    if (double_equal(term, 0.0)) {
        if ((double)k > (double)n * p) return -log1term / 2.30258509299404568402 - logNT;
        return -logNT;
    }
    double bin_tail = term;
    for (int i = k + 1; i <= n; i++) {
        const double bin_term = (double)(n - i + 1) / (double)i;
        const double mult_term = bin_term * p_term;
        term *= mult_term;
        bin_tail += term;
        if (bin_term < 1.0) {
            const double err = term * ((1.0 - lfo_pow(mult_term, (double)(n - i + 1))) / (1.0 - mult_term) - 1.0);
            if (err < tolerance * fabs(-lfo_log10(bin_tail) - logNT) * bin_tail) break;
        }
    }
    return -lfo_log10(bin_tail) - logNT;
}

/* ------------------------------------------------------------------ EDline (:2242-2482) */
typedef struct Fit {
    float ATA[4], ATV[2];          /* cv::Mat_<float> members: carried from the initial fit into the extensions */
} Fit;

static void sums_f32(const uint32_t* a, const uint32_t* b, unsigned s, int n, float* m4, float* v2)
{
    /* [sum a^2, sum a; sum a, n] and [sum a*b, sum b]: exact integer sums rounded once to float (cv::gemm, double work type) */
    long long saa = 0, sa = 0, sab = 0, sb = 0;
    for (int i = 0; i < n; ++i) { const long long A = a[s + i], B = b[s + i]; saa += A * A; sa += A; sab += A * B; sb += B; }
    m4[0] = (float)saa; m4[1] = (float)sa; m4[2] = (float)sa; m4[3] = (float)n;
    v2[0] = (float)sab; v2[1] = (float)sb;
}

static void solve(const Fit* f, double* eq)
{
    const float* A = f->ATA;
    const double coef = 1.0 / ((double)A[0] * (double)A[3] - (double)A[1] * (double)A[2]);
    eq[0] = coef * ((double)A[3] * (double)f->ATV[0] - (double)A[1] * (double)f->ATV[1]);
    eq[1] = coef * ((double)A[0] * (double)f->ATV[1] - (double)A[2] * (double)f->ATV[0]);
}

/* :2484-2553: fit minLineLen pixels from offsetS; returns the fit error */
static double fit_initial(Fit* f, const uint8_t* dir, int W, const uint32_t* xc, const uint32_t* yc, unsigned s, int min_len, double* eq)
{
    const int horizontal = dir[yc[s] * W + xc[s]] == ED_HORIZONTAL;
    const uint32_t* a = horizontal ? xc : yc, *b = horizontal ? yc : xc;
    sums_f32(a, b, s, min_len, f->ATA, f->ATV);
    solve(f, eq);
    double err = 0;
    for (int i = 0; i < min_len; ++i) {
        const double c = (double)b[s + i] - (double)a[s + i] * eq[0] - eq[1];
        err += c * c;
    }
    return sqrt(err);
}

/* :2555-2643: add the pixels [newS, e) of the line array to the normal equations */
static void fit_extend(Fit* f, const uint8_t* dir, int W, const uint32_t* xc, const uint32_t* yc, unsigned s, unsigned newS, unsigned e, double* eq)
{
    const int horizontal = dir[yc[s] * W + xc[s]] == ED_HORIZONTAL;
    const uint32_t* a = horizontal ? xc : yc, *b = horizontal ? yc : xc;
    float m[4], v[2];
    sums_f32(a, b, newS, (int)(e - newS), m, v);
    for (int k = 0; k < 4; ++k) f->ATA[k] = f->ATA[k] + m[k];
    for (int k = 0; k < 2; ++k) f->ATV[k] = f->ATV[k] + v[k];
    solve(f, eq);
}

/* :2645-2726 */
static int line_validation(const int16_t* pdx, const int16_t* pdy, int W, int H, const uint32_t* xc, const uint32_t* yc, unsigned s,
                           unsigned e, const double* lineEqu, double logNT, float* direction_out)
{
    const int n = (int)(e - s);
    int meanGradientX = 0, meanGradientY = 0;
    double* pointDirection = (double*)malloc((size_t)(n > 0 ? n : 1) * sizeof(double));
    for (int i = 0; i < n; ++i) {
        const int index = yc[s + i] * W + xc[s + i];
        meanGradientX += pdx[index];
        meanGradientY += pdy[index];
        pointDirection[i] = lfo_atan2(-(double)pdx[index], (double)pdy[index]);
    }
    const double dx = fabs(lineEqu[1]), dy = fabs(lineEqu[0]);
    float direction = 0;
    int ok = 1;
    if (meanGradientX == 0 && meanGradientY == 0) ok = 0;
    if (ok) {
        if (meanGradientX > 0 && meanGradientY >= 0) direction = (float)lfo_atan2(-dy, dx);
        if (meanGradientX <= 0 && meanGradientY > 0) direction = (float)lfo_atan2(dy, dx);
        if (meanGradientX < 0 && meanGradientY <= 0) direction = (float)lfo_atan2(dy, -dx);
        if (meanGradientX >= 0 && meanGradientY < 0) direction = (float)lfo_atan2(-dy, -dx);
        const double PI = 3.14159265358979323846;
        if (fabs((double)direction) < 0.15 || PI - fabs((double)direction) < 0.15)
            if (fabs(lineEqu[2]) < 10 || fabs((double)(unsigned)H - fabs(lineEqu[2])) < 10) ok = 0;
        if (ok && fabs(fabs((double)direction) - PI * 0.5) < 0.15)
            if (fabs(lineEqu[2]) < 10 || fabs((double)(unsigned)W - fabs(lineEqu[2])) < 10) ok = 0;
        if (ok) {
            int k = 0;
            for (int i = 0; i < n; ++i) {
                const double dis = fabs((double)direction - pointDirection[i]);
                if (fabs(2 * PI - dis) < 0.392699 || dis < 0.392699) k++;
            }
            ok = lfo_ed_nfa(n, k, 0.125, logNT) > 0;
        }
    }
    free(pointDirection);
    *direction_out = direction;
    return ok;
}

/* EDline over the chains: line pixels (lx, ly, lsid), equations (3 doubles), endpoints (4 floats), direction; returns
 * the number of lines, -1 on overflow of cap_lines */
int lfo_ed_lines(const int16_t* pdx, const int16_t* pdy, const uint8_t* dir, int rows, int cols, const uint32_t* xcors,
                 const uint32_t* ycors, const uint32_t* sid, int n_edges, int min_line_len, double fit_err_threshold,
                 uint32_t* lx, uint32_t* ly, uint32_t* lsid, double* equations3, float* endpoints4, float* direction, int cap_lines)
{
    const int W = cols, H = rows;
    const double logNT = 2.0 * (lfo_log10((double)(unsigned)W) + lfo_log10((double)(unsigned)H));
    unsigned numOfLines = 0, offL = 0, newOffsetS = 0;
    Fit f;
    memset(&f, 0, sizeof(f));
    double lineFitErr = 0, eq[2] = { 0, 0 };
    const unsigned limit = 5u * (unsigned)n_edges < (unsigned)cap_lines ? 5u * (unsigned)n_edges : (unsigned)cap_lines;
    for (int edgeID = 0; edgeID < n_edges; ++edgeID) {
        unsigned S = sid[edgeID];
        const unsigned E = sid[edgeID + 1];
        while (E > S + (unsigned)min_line_len) {
            while (E > S + (unsigned)min_line_len) {
                lineFitErr = fit_initial(&f, dir, W, xcors, ycors, S, min_line_len, eq);
                if (lineFitErr <= fit_err_threshold) break;
                S += SKIP_EDGE_POINT;
            }
            if (lineFitErr > fit_err_threshold) break;
            if (numOfLines >= limit) return -1;
            lsid[numOfLines] = offL;
            double coef1 = 0;
            int bExtended = 1, bFirstTry = 1, numOfOutlier, tryTimes = 0;
            const int horizontal = dir[ycors[S] * W + xcors[S]] == ED_HORIZONTAL;
            while (bExtended) {
                tryTimes++;
                if (bFirstTry) {
                    bFirstTry = 0;
                    for (int i = 0; i < min_line_len; ++i) { lx[offL] = xcors[S]; ly[offL++] = ycors[S++]; }
                } else {
                    fit_extend(&f, dir, W, lx, ly, lsid[numOfLines], newOffsetS, offL, eq);
                }
                coef1 = horizontal ? 1 / sqrt(eq[0] * eq[0] + 1) : 1 / sqrt(1 + eq[0] * eq[0]);
                numOfOutlier = 0;
                newOffsetS = offL;
                while (E > S) {
                    const double dis = horizontal ? fabs(eq[0] * (double)xcors[S] - (double)ycors[S] + eq[1]) * coef1
                                                  : fabs((double)xcors[S] - eq[0] * (double)ycors[S] - eq[1]) * coef1;
                    lx[offL] = xcors[S]; ly[offL++] = ycors[S++];
                    if (dis > fit_err_threshold) { numOfOutlier++; if (numOfOutlier > 3) break; }
                    else numOfOutlier = 0;
                }
                offL -= numOfOutlier;
                S -= numOfOutlier;
                if (!(offL - newOffsetS > 0 && tryTimes < TRY_TIME)) bExtended = 0;
            }
            double lineEqu[3];
            if (horizontal) { lineEqu[0] = eq[0] * coef1; lineEqu[1] = -1 * coef1; lineEqu[2] = eq[1] * coef1; }
            else { lineEqu[0] = 1 * coef1; lineEqu[1] = -eq[0] * coef1; lineEqu[2] = -eq[1] * coef1; }
            float dirn;
            if (line_validation(pdx, pdy, W, H, lx, ly, lsid[numOfLines], offL, lineEqu, logNT, &dirn)) {
                memcpy(equations3 + 3 * (size_t)numOfLines, lineEqu, sizeof(lineEqu));
                const double a1 = lineEqu[1] * lineEqu[1], a2 = lineEqu[0] * lineEqu[0], a3 = lineEqu[0] * lineEqu[1];
                const double a4 = lineEqu[2] * lineEqu[0], a5 = lineEqu[2] * lineEqu[1];
                float* ep = endpoints4 + 4 * (size_t)numOfLines;
                unsigned Px = lx[lsid[numOfLines]], Py = ly[lsid[numOfLines]];
                ep[0] = (float)(a1 * Px - a3 * Py - a4);
                ep[1] = (float)(a2 * Py - a3 * Px - a5);
                Px = lx[offL - 1]; Py = ly[offL - 1];
                ep[2] = (float)(a1 * Px - a3 * Py - a4);
                ep[3] = (float)(a2 * Py - a3 * Px - a5);
                direction[numOfLines] = dirn;
                numOfLines++;
            } else {
                offL = lsid[numOfLines];
            }
        }
    }
    lsid[numOfLines] = offL;
    return (int)numOfLines;
}

/* :2728-2751: gImgWO_ is a CV_16S image read through an unsigned char pointer with a PIXEL index -- the byte at
 * offset (y * W + x) of the little-endian s16 plane, not the pixel's value.  Restated as written. */
void lfo_ed_salience(const int16_t* gwo, int cols, const uint32_t* lx, const uint32_t* ly, const uint32_t* lsid, int n_lines,
                     float* salience)
{
    const unsigned char* pg = (const unsigned char*)gwo;
    for (int i = 0; i < n_lines; ++i) {
        int s = 0;
        for (unsigned k = lsid[i]; k < lsid[i + 1]; ++k) s += pg[ly[k] * (unsigned)cols + lx[k]];
        salience[i] = (float)s;
    }
}

/* ------------------------------------------------------------------ one octave, whole detector */
struct lfo_edlines {
    int rows, cols, n_anchors, n_edges, n_lines;
    int16_t *dx, *dy, *g, *gwo;
    uint8_t *dir, *edge;
    uint32_t *ax, *ay, *xcors, *ycors, *sid, *lx, *ly, *lsid;
    double* equations;
    float *endpoints, *direction, *salience;
};

void lfo_edlines_params_default(lfo_edlines_params* p)
{
    /* binary_descriptor_custom.cpp:1374-1385 */
    p->ksize = 15; p->sigma = 30.0f; p->gradient_threshold = 80; p->anchor_threshold = 8; p->scan_intervals = 2;
    p->min_line_len = 15; p->line_fit_err_threshold = 1.6;
}

void lfo_edlines_free(lfo_edlines* e)
{
    if (!e) return;
    free(e->dx); free(e->dy); free(e->g); free(e->gwo); free(e->dir); free(e->edge); free(e->ax); free(e->ay);
    free(e->xcors); free(e->ycors); free(e->sid); free(e->lx); free(e->ly); free(e->lsid); free(e->equations);
    free(e->endpoints); free(e->direction); free(e->salience);
    free(e);
}

/* EDLineDetector::EDline(image) on an already blurred u8 image; NULL when the detection fails (the reference's -1) */
lfo_edlines* lfo_edlines_run(const lfo_edlines_params* p, const uint8_t* img, int rows, int cols)
{
    lfo_edlines* e = (lfo_edlines*)calloc(1, sizeof(*e));
    const size_t np = (size_t)rows * cols;
    const unsigned cap = (unsigned)np / 5, max_edges = cap / 20;
    e->rows = rows; e->cols = cols;
    e->dx = (int16_t*)malloc(np * 2); e->dy = (int16_t*)malloc(np * 2); e->g = (int16_t*)malloc(np * 2); e->gwo = (int16_t*)malloc(np * 2);
    e->dir = (uint8_t*)malloc(np); e->edge = (uint8_t*)malloc(np);
    e->ax = (uint32_t*)malloc((size_t)(cap + 1) * 4); e->ay = (uint32_t*)malloc((size_t)(cap + 1) * 4);
    e->xcors = (uint32_t*)malloc((size_t)(2 * cap + 2) * 4); e->ycors = (uint32_t*)malloc((size_t)(2 * cap + 2) * 4);
    e->sid = (uint32_t*)calloc((size_t)max_edges + 3, 4);
    lfo_ed_gradient(img, rows, cols, p->gradient_threshold, e->dx, e->dy, e->g, e->gwo, e->dir);
    e->n_anchors = lfo_ed_anchors(e->g, e->dir, rows, cols, p->anchor_threshold, p->scan_intervals, e->ax, e->ay, (int)cap);
    if (e->n_anchors < 0) { lfo_edlines_free(e); return NULL; }
    e->n_edges = lfo_ed_link(e->g, e->dir, rows, cols, e->ax, e->ay, e->n_anchors, p->min_line_len, e->xcors, e->ycors, e->sid, e->edge);
    if (e->n_edges < 0) { lfo_edlines_free(e); return NULL; }
    const unsigned npx = e->sid[e->n_edges];
    const int cap_lines = 5 * e->n_edges + 2;
    e->lx = (uint32_t*)malloc((size_t)(npx + 1) * 4); e->ly = (uint32_t*)malloc((size_t)(npx + 1) * 4);
    e->lsid = (uint32_t*)calloc((size_t)cap_lines + 2, 4);
    e->equations = (double*)calloc((size_t)cap_lines * 3 + 3, sizeof(double));
    e->endpoints = (float*)calloc((size_t)cap_lines * 4 + 4, sizeof(float));
    e->direction = (float*)calloc((size_t)cap_lines + 1, sizeof(float));
    e->salience = (float*)calloc((size_t)cap_lines + 1, sizeof(float));
    e->n_lines = lfo_ed_lines(e->dx, e->dy, e->dir, rows, cols, e->xcors, e->ycors, e->sid, e->n_edges, p->min_line_len,
                              p->line_fit_err_threshold, e->lx, e->ly, e->lsid, e->equations, e->endpoints, e->direction, cap_lines);
    if (e->n_lines < 0) { lfo_edlines_free(e); return NULL; }
    lfo_ed_salience(e->gwo, cols, e->lx, e->ly, e->lsid, e->n_lines, e->salience);
    return e;
}

int lfo_edlines_counts(const lfo_edlines* e, int* n_anchors, int* n_edges, int* n_edge_pixels, int* n_lines, int* n_line_pixels)
{
    if (n_anchors) *n_anchors = e->n_anchors;
    if (n_edges) *n_edges = e->n_edges;
    if (n_edge_pixels) *n_edge_pixels = (int)e->sid[e->n_edges];
    if (n_lines) *n_lines = e->n_lines;
    if (n_line_pixels) *n_line_pixels = (int)e->lsid[e->n_lines];
    return 0;
}

const void* lfo_edlines_array(const lfo_edlines* e, int which)
{
    switch (which) {
    case 0: return e->dx; case 1: return e->dy; case 2: return e->g; case 3: return e->gwo; case 4: return e->dir;
    case 5: return e->edge; case 6: return e->ax; case 7: return e->ay; case 8: return e->xcors; case 9: return e->ycors;
    case 10: return e->sid; case 11: return e->lx; case 12: return e->ly; case 13: return e->lsid; case 14: return e->equations;
    case 15: return e->endpoints; case 16: return e->direction; case 17: return e->salience;
    default: return NULL;
    }
}

/* ------------------------------------------------------------------ OctaveKeyLines + detectImpl + LBD on the detector's
 * own gradients (operator() with useProvidedKeyLines = false, :263-301).  gray: u8 rows x cols.  Outputs, one row per
 * KeyLine in detectImpl's order (class id, then octave): see lfo_keylines_out in lf_oracle.h.  Returns the number of
 * KeyLines, -1 when a detector fails, -2 when cap is too small. */
int lfo_octave_keylines(const lfo_edlines_params* p, const uint8_t* gray, int rows, int cols, int n_octaves, int ksize, int cap,
                        lfo_keylines_out* out)
{
    lfo_edlines** det = (lfo_edlines**)calloc((size_t)n_octaves, sizeof(*det));
    int* ow = (int*)calloc((size_t)n_octaves, sizeof(int)), *oh = (int*)calloc((size_t)n_octaves, sizeof(int));
    uint8_t* image = (uint8_t*)malloc((size_t)rows * cols);
    memcpy(image, gray, (size_t)rows * cols);
    int iw = cols, ih = rows, rc = 0;
    float preSigma2 = 0, curSigma2 = 1.0f;
    const double factor = sqrt(2.0);
    unsigned numOfFinalLine = 0;
    for (int o = 0; o < n_octaves; ++o) {
        const float increaseSigma = (float)sqrt((double)(curSigma2 - preSigma2));       /* std::sqrt(float): correctly rounded */
        uint8_t* blur = (uint8_t*)malloc((size_t)iw * ih);
        lfo_gaussian_blur_u8(image, ih, iw, ksize, (double)increaseSigma, blur);
        ow[o] = iw; oh[o] = ih;
        det[o] = lfo_edlines_run(p, blur, ih, iw);
        if (!det[o]) { free(blur); rc = -1; break; }
        numOfFinalLine += (unsigned)det[o]->n_lines;
        int nh, nw;
        const double inv = (double)(1.f) / factor;
        lfo_resize_size(ih, iw, inv, &nh, &nw);
        free(image);
        image = (uint8_t*)malloc((size_t)(nh > 0 ? nh : 1) * (nw > 0 ? nw : 1));
        if (o + 1 < n_octaves) lfo_resize_linear_u8(blur, ih, iw, inv, image);
        free(blur);
        iw = nw; ih = nh;
        preSigma2 = curSigma2;
        curSigma2 = curSigma2 * 2;
    }
    free(image);
    int n_out = rc;
    if (rc == 0) {
        typedef struct { unsigned octaveCount, lineIDInOctave, lineIDInScaleLineVec; float lineLength; } OctaveLine;
        OctaveLine* ol = (OctaveLine*)calloc((size_t)numOfFinalLine + 1, sizeof(OctaveLine));
        unsigned nfl = 0, lineIDInScaleLineVec = 0;
        float dx, dy;
        for (int l = 0; l < det[0]->n_lines; ++l) {
            const float* ep = det[0]->endpoints + 4 * l;
            ol[nfl].octaveCount = 0; ol[nfl].lineIDInOctave = l; ol[nfl].lineIDInScaleLineVec = lineIDInScaleLineVec;
            dx = (float)fabs((double)(ep[0] - ep[2]));
            dy = (float)fabs((double)(ep[1] - ep[3]));
            ol[nfl].lineLength = (float)sqrt((double)(dx * dx + dy * dy));
            nfl++; lineIDInScaleLineVec++;
        }
        float* scale = (float*)malloc((size_t)n_octaves * sizeof(float));
        scale[0] = 1;
        for (int o = 1; o < n_octaves; ++o) scale[o] = (float)(factor * scale[o - 1]);
        const double twoPI = 2 * 3.14159265358979323846, PI = 3.14159265358979323846;
        for (int o = 1; o < n_octaves; ++o) {
            for (int l = 0; l < det[o]->n_lines; ++l) {
                const float* ep = det[o]->endpoints + 4 * l;
                const float rho1 = (float)(scale[o] * fabs(det[o]->equations[3 * l + 2]));
                const float tempValue = (float)(rho1 * 0.0152);
                float diffNearThreshold = (tempValue > 6) ? tempValue : 6;
                diffNearThreshold = (diffNearThreshold < 12) ? diffNearThreshold : 12;
                dx = (float)fabs((double)(ep[0] - ep[2]));
                dy = (float)fabs((double)(ep[1] - ep[3]));
                const float length = scale[o] * (float)sqrt((double)(dx * dx + dy * dy));
                float minEndPointDis = 12;
                unsigned closeLineID = 0;
                for (unsigned nx = 0; nx < nfl; ++nx) {
                    const unsigned oid = ol[nx].octaveCount;
                    if ((int)oid == o) break;
                    const unsigned lid = ol[nx].lineIDInOctave;
                    const float direction = (float)fabs((double)(det[o]->direction[l] - det[oid]->direction[lid]));
                    if (direction > 0.1745 && (twoPI - direction > 0.1745)) continue;
                    const float rho2 = (float)(scale[oid] * fabs(det[oid]->equations[3 * lid + 2]));
                    const float diffNear = (float)fabs((double)(rho1 - rho2));
                    if (diffNear > diffNearThreshold) continue;
                    const float* np_ = det[oid]->endpoints + 4 * lid;
                    const float lp0 = scale[o] * ep[0], lp1 = scale[o] * ep[1], lp2 = scale[o] * ep[2], lp3 = scale[o] * ep[3];
                    const float np0 = scale[oid] * np_[0], np1 = scale[oid] * np_[1], np2 = scale[oid] * np_[2], np3 = scale[oid] * np_[3];
                    float endPointDis, minLocalDis, maxLocalDis;
                    dx = lp0 - np0; dy = lp1 - np1;
                    endPointDis = (float)sqrt((double)(dx * dx + dy * dy));
                    minLocalDis = endPointDis; maxLocalDis = endPointDis;
                    dx = lp2 - np2; dy = lp3 - np3;
                    endPointDis = (float)sqrt((double)(dx * dx + dy * dy));
                    minLocalDis = (endPointDis < minLocalDis) ? endPointDis : minLocalDis;
                    maxLocalDis = (endPointDis > maxLocalDis) ? endPointDis : maxLocalDis;
                    dx = lp0 - np2; dy = lp1 - np3;
                    endPointDis = (float)sqrt((double)(dx * dx + dy * dy));
                    minLocalDis = (endPointDis < minLocalDis) ? endPointDis : minLocalDis;
                    maxLocalDis = (endPointDis > maxLocalDis) ? endPointDis : maxLocalDis;
                    dx = lp2 - np0; dy = lp3 - np1;
                    endPointDis = (float)sqrt((double)(dx * dx + dy * dy));
                    minLocalDis = (endPointDis < minLocalDis) ? endPointDis : minLocalDis;
                    maxLocalDis = (endPointDis > maxLocalDis) ? endPointDis : maxLocalDis;
                    if (((double)maxLocalDis < 0.8 * (double)(length + ol[nx].lineLength)) && (minLocalDis < minEndPointDis)) {
                        minEndPointDis = minLocalDis;
                        closeLineID = nx;
                    }
                }
                if (minEndPointDis < 12) ol[nfl].lineIDInScaleLineVec = ol[closeLineID].lineIDInScaleLineVec;
                else { ol[nfl].lineIDInScaleLineVec = lineIDInScaleLineVec; lineIDInScaleLineVec++; }
                ol[nfl].octaveCount = (unsigned)o; ol[nfl].lineIDInOctave = (unsigned)l; ol[nfl].lineLength = length;
                nfl++;
            }
        }
        /* keyLines[class].push_back(line) in octaveLines order, then detectImpl walks classes in order (:478-509):
         * a stable ordering by class id */
        if ((int)nfl > cap) n_out = -2;
        else {
            unsigned* start = (unsigned*)calloc((size_t)lineIDInScaleLineVec + 2, sizeof(unsigned));
            for (unsigned i = 0; i < nfl; ++i) start[ol[i].lineIDInScaleLineVec + 1]++;
            for (unsigned c = 0; c < lineIDInScaleLineVec; ++c) start[c + 1] += start[c];
            for (unsigned i = 0; i < nfl; ++i) {
                const unsigned k = start[ol[i].lineIDInScaleLineVec]++;
                const unsigned oid = ol[i].octaveCount, lid = ol[i].lineIDInOctave;
                const float direction = det[oid]->direction[lid];
                const float* ep = det[oid]->endpoints + 4 * lid;
                const float s1 = ep[0], s2 = ep[1], e1 = ep[2], e2 = ep[3];
                dx = e1 - s1; dy = e2 - s2;
                int shouldChange = 0;
                if (direction >= -0.75 * PI && direction < -0.25 * PI) { if (dy > 0) shouldChange = 1; }
                if (direction >= -0.25 * PI && direction < 0.25 * PI) { if (dx < 0) shouldChange = 1; }
                if (direction >= 0.25 * PI && direction < 0.75 * PI) { if (dy < 0) shouldChange = 1; }
                if ((direction >= 0.75 * PI && direction < PI) || (direction >= -PI && direction < -0.75 * PI)) { if (dx > 0) shouldChange = 1; }
                const float t = scale[oid];
                float* io = out->in_octave + 4 * (size_t)k;
                float* st = out->start_end + 4 * (size_t)k;
                if (shouldChange) { io[0] = e1; io[1] = e2; io[2] = s1; io[3] = s2; st[0] = t * e1; st[1] = t * e2; st[2] = t * s1; st[3] = t * s2; }
                else { io[0] = s1; io[1] = s2; io[2] = e1; io[3] = e2; st[0] = t * s1; st[1] = t * s2; st[2] = t * e1; st[3] = t * e2; }
                out->angle[k] = direction;
                out->num_pixels[k] = (int32_t)(det[oid]->lsid[lid + 1] - det[oid]->lsid[lid]);
                out->line_length[k] = ol[i].lineLength;
                out->octave[k] = (int32_t)oid;
                out->class_id[k] = (int32_t)ol[i].lineIDInScaleLineVec;
                if (out->salience) out->salience[k] = det[oid]->salience[lid];
                /* detectImpl :497-499 */
                if (out->size) out->size[k] = (st[2] - st[0]) * (st[3] - st[1]);
                const int mx = ow[oid] > oh[oid] ? ow[oid] : oh[oid];
                if (out->response) out->response[k] = ol[i].lineLength / mx;
                if (out->pt) { out->pt[2 * (size_t)k] = (st[2] + st[0]) / 2; out->pt[2 * (size_t)k + 1] = (st[3] + st[1]) / 2; }
                /* LBD on the detecting octave's own gradient images (computeLBD with useDetectionData, :1079-1090) */
                if (out->desc || out->code) {
                    float d72[72]; uint8_t c32[32];
                    lfo_lbd(det[oid]->dx, det[oid]->dy, oh[oid], ow[oid], io, &out->angle[k], &out->num_pixels[k], 1, d72, c32);
                    if (out->desc) memcpy(out->desc + 72 * (size_t)k, d72, sizeof(d72));
                    if (out->code) memcpy(out->code + 32 * (size_t)k, c32, 32);
                }
            }
            free(start);
            n_out = (int)nfl;
        }
        for (int o = 0; o < n_octaves && o < 8; ++o) {
            out->octave_rows[o] = oh[o]; out->octave_cols[o] = ow[o]; out->octave_lines[o] = det[o]->n_lines;
        }
        free(scale); free(ol);
    }
    for (int o = 0; o < n_octaves; ++o) lfo_edlines_free(det[o]);
    free(det); free(ow); free(oh);
    return n_out;
}

/* ------------------------------------------------------------------ compute-only path: descriptors of GIVEN KeyLines
 * (BinaryDescriptor::compute, :524-687 with useDetectionData = false): gradients from computeGaussianPyramid
 * (:350-371: GaussianBlur 5x5 sigma 1, then pyrDown per octave) + Sobel (:374-398) */
int lfo_describe_keylines(const uint8_t* gray, int rows, int cols, const float* in_octave4, const float* angle,
                          const int32_t* num_pixels, const int32_t* octave, int n, float* desc72, uint8_t* code32)
{
    int max_oct = -1;
    for (int i = 0; i < n; ++i) if (octave[i] > max_oct) max_oct = octave[i];
    if (max_oct < 0) return 0;
    if (max_oct > 7) return -1;
    uint8_t* cur = (uint8_t*)malloc((size_t)rows * cols);
    lfo_gaussian5_u8(gray, rows, cols, cur);
    int w = cols, h = rows;
    for (int o = 0; o <= max_oct; ++o) {
        if (o > 0) {
            uint8_t* nxt = (uint8_t*)malloc((size_t)(h / 2 > 0 ? h / 2 : 1) * (w / 2 > 0 ? w / 2 : 1));
            lfo_pyrdown_u8(cur, h, w, nxt);
            free(cur); cur = nxt; w /= 2; h /= 2;
        }
        int16_t* dx = (int16_t*)malloc((size_t)w * h * 2), *dy = (int16_t*)malloc((size_t)w * h * 2);
        lfo_sobel3_s16(cur, h, w, dx, dy);
        for (int i = 0; i < n; ++i)
            if (octave[i] == o)
                lfo_lbd(dx, dy, h, w, in_octave4 + 4 * (size_t)i, angle + i, num_pixels + i, 1, desc72 + 72 * (size_t)i, code32 + 32 * (size_t)i);
        free(dx); free(dy);
    }
    free(cur);
    return n;
}
