/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see lf_oracle.h).
 *
 * a-4: cv2.createLineSegmentDetector(_refine=LSD_REFINE_ADV).detect(edge)
 * (/root/reference/src/line_detector/include/line_detector/line_detector_lsd.py:64-72;
 *  the C++ wrapper /root/reference/src/line_descriptor/src/LSDDetector_custom.cpp:149,246-253
 *  calls the same OpenCV class).
 *
 * The arithmetic is OpenCV imgproc/lsd.cpp (third party, NOT in the reference tree,
 * not installed here) -> PARITY UNPINNED.  Restated from the published algorithm
 * (Grompone von Gioi, Jakubowicz, Morel, Randall, "LSD: a Line Segment Detector",
 * IPOL 2012) in the form OpenCV 3.0-3.3 implements it (the f64 image path that ROS
 * Kinetic/Melodic ship): u8 -> f64, GaussianBlur(sigma = sigma_scale/scale,
 * ksize = 1+2*ceil(sigma*sqrt(2*3*ln 10))), bilinear resize by `scale`,
 * 2x2 gradient with fastAtan2 level-line angle, 1024-bin pseudo-ordering
 * (descending bin, raster order inside a bin), 8-connected region growing with
 * running mean angle, rectangle fit, density refinement, NFA-driven improvement.
 * Known quirks of that implementation are kept where remembered (integer edge
 * steps and the y-vs-x comparisons in rect_nfa's scanline walk; the Gamma-less
 * "n+1" term in nfa()).  Transcendentals go through lf_detmath so the HIP path
 * can reproduce every bit.
 */
#include "lf_oracle.h"
#include "lf_detmath.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>

#define LSD_PI 3.14159265358979323846
#define M_3_2_PI ((3 * LSD_PI) / 2)
#define M_2__PI (2 * LSD_PI)
#define NOTDEF (-1024.0)
#define DEG_TO_RADS (LSD_PI / 180)
#define REL_ERR_FACTOR 100.0

static int cv_round(double v)
{
    double f = (double)(long long)v;
    double d = v - f;
    long long i = (long long)f;
    if (d > 0.5 || (d == 0.5 && (i & 1))) i += 1;
    else if (d < -0.5 || (d == -0.5 && (i & 1))) i -= 1;
    return (int)i;
}
static int cv_floor(double v) { int i = (int)v; return i - (v < (double)i ? 1 : 0); }
static int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * (n - 1) - p; }
    return p;
}

void lfo_lsd_scaled_size(const lfo_config* c, int rows, int cols, int* srows, int* scols)
{
    if (c->lsd_scale != 1.0) {
        *scols = cv_round(cols * c->lsd_scale);
        *srows = cv_round(rows * c->lsd_scale);
    } else { *scols = cols; *srows = rows; }
}

/* Gaussian kernel as cv::getGaussianKernel(n, sigma, CV_64F) computes it */
static int lsd_gauss_kernel(const lfo_config* c, double* k /* >= 64 */)
{
    const double sigma = (c->lsd_scale < 1) ? (c->lsd_sigma_scale / c->lsd_scale) : c->lsd_sigma_scale;
    const double sprec = 3;
    const unsigned h = (unsigned)ceil(sigma * sqrt(2 * sprec * lfo_log(10.0)));
    const int n = 1 + 2 * (int)h;
    const double scale2X = -0.5 / (sigma * sigma);
    double sum = 0;
    for (int i = 0; i < n; ++i) {
        double x = i - (n - 1) * 0.5;
        double t = lfo_exp(scale2X * x * x);
        k[i] = t;
        sum += t;
    }
    sum = 1.0 / sum;
    for (int i = 0; i < n; ++i) k[i] *= sum;
    return n;
}

void lfo_lsd_scaled_image(const lfo_config* c, const uint8_t* img, int rows, int cols, double* scaled)
{
    int srows, scols;
    lfo_lsd_scaled_size(c, rows, cols, &srows, &scols);
    if (c->lsd_scale == 1.0) {
        for (size_t i = 0; i < (size_t)rows * cols; ++i) scaled[i] = (double)img[i];
        return;
    }
    double k[64];
    const int n = lsd_gauss_kernel(c, k);
    const int h = n / 2;
    double* rowf = (double*)malloc((size_t)rows * cols * sizeof(double));
    double* blur = (double*)malloc((size_t)rows * cols * sizeof(double));
    /* cv::RowFilter<double,double>: s = kx[0]*S[0]; s += kx[k]*S[k] */
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            double s = k[0] * (double)img[(size_t)y * cols + reflect101(x - h, cols)];
            for (int j = 1; j < n; ++j) s += k[j] * (double)img[(size_t)y * cols + reflect101(x - h + j, cols)];
            rowf[(size_t)y * cols + x] = s;
        }
    /* cv::SymmColumnFilter: s = ky[0]*S[0] + delta; s += ky[k]*(S[k] + S[-k]) */
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            double s = k[h] * rowf[(size_t)y * cols + x] + 0.0;
            for (int j = 1; j <= h; ++j)
                s += k[h + j] * (rowf[(size_t)reflect101(y + j, rows) * cols + x] + rowf[(size_t)reflect101(y - j, rows) * cols + x]);
            blur[(size_t)y * cols + x] = s;
        }
    /* cv::resize(INTER_LINEAR) on CV_64F: float coefficients, double accumulation,
       horizontal pass first then vertical. */
    const double scale_x = 1.0 / c->lsd_scale, scale_y = 1.0 / c->lsd_scale;
    int* xofs = (int*)malloc(sizeof(int) * scols);
    float* xa = (float*)malloc(sizeof(float) * 2 * scols);
    int xmax = scols;
    for (int dx = 0; dx < scols; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= cols) {
            if (dx < xmax) xmax = dx;
            if (sx >= cols - 1) { fx = 0; sx = cols - 1; }
        }
        xofs[dx] = sx;
        xa[2 * dx] = 1.f - fx;
        xa[2 * dx + 1] = fx;
    }
    double* hbuf = (double*)malloc((size_t)rows * scols * sizeof(double));
    for (int y = 0; y < rows; ++y) {
        const double* S = blur + (size_t)y * cols;
        double* D = hbuf + (size_t)y * scols;
        for (int dx = 0; dx < scols; ++dx) {
            int sx = xofs[dx];
            if (dx < xmax) D[dx] = S[sx] * (double)xa[2 * dx] + S[sx + 1] * (double)xa[2 * dx + 1];
            else D[dx] = S[sx] * 1.0;
        }
    }
    for (int dy = 0; dy < srows; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        float b0 = 1.f - fy, b1 = fy;
        int y0 = sy < 0 ? 0 : (sy > rows - 1 ? rows - 1 : sy);
        int y1 = sy + 1 < 0 ? 0 : (sy + 1 > rows - 1 ? rows - 1 : sy + 1);
        const double* S0 = hbuf + (size_t)y0 * scols;
        const double* S1 = hbuf + (size_t)y1 * scols;
        for (int dx = 0; dx < scols; ++dx)
            scaled[(size_t)dy * scols + dx] = S0[dx] * (double)b0 + S1[dx] * (double)b1;
    }
    free(rowf); free(blur); free(xofs); free(xa); free(hbuf);
}

/* seed order inside a gradient bin: 0 = raster (OpenCV 3.0's per-bin lists; the default here and on the GPU),
 * 1 = what std::sort(ordered_points, compare_norm) of the later 3.x leaves (lf_oracle_sort.cpp): a switch for
 * counting how much that lead matters, never the product's behaviour */
static int g_seed_order = 0;
void lfo_lsd_set_seed_order(int mode) { g_seed_order = mode; }
void lfo_std_sort_seed_order(const int32_t* bin_of_pixel, int H, int W, int32_t* order);

int lfo_lsd_ll_angle(const lfo_config* c, const double* scaled, int H, int W,
                     double* angles, double* modgrad, int32_t* order)
{
    const double prec = LSD_PI * c->lsd_ang_th / 180;
    const double rho = c->lsd_quant / lfo_sin(prec);
    const int n_bins = c->lsd_n_bins;
    for (int x = 0; x < W; ++x) { angles[(size_t)(H - 1) * W + x] = NOTDEF; modgrad[(size_t)(H - 1) * W + x] = 0; }
    for (int y = 0; y < H; ++y) { angles[(size_t)y * W + W - 1] = NOTDEF; modgrad[(size_t)y * W + W - 1] = 0; }
    double max_grad = -1;
    for (int y = 0; y < H - 1; ++y)
        for (int x = 0; x < W - 1; ++x) {
            size_t a = (size_t)y * W + x;
            double DA = scaled[a + W + 1] - scaled[a];
            double BC = scaled[a + 1] - scaled[a + W];
            double gx = DA + BC;
            double gy = DA - BC;
            double norm = sqrt((gx * gx + gy * gy) / 4);
            modgrad[a] = norm;
            if (norm <= rho) angles[a] = NOTDEF;
            else {
                angles[a] = (double)lfo_fast_atan2_deg((float)gx, (float)(-gy)) * DEG_TO_RADS;
                if (norm > max_grad) max_grad = norm;
            }
        }
    /* pseudo-ordering: bins descending, raster order inside a bin */
    const double bin_coef = (max_grad > 0) ? (double)(n_bins - 1) / max_grad : 0;
    int* count = (int*)calloc((size_t)n_bins + 1, sizeof(int));
    for (int y = 0; y < H - 1; ++y)
        for (int x = 0; x < W - 1; ++x) {
            int i = (int)(modgrad[(size_t)y * W + x] * bin_coef);
            count[i]++;
        }
    int* start = (int*)malloc(sizeof(int) * n_bins);
    int acc = 0;
    for (int b = n_bins - 1; b >= 0; --b) { start[b] = acc; acc += count[b]; }
    for (int y = 0; y < H - 1; ++y)
        for (int x = 0; x < W - 1; ++x) {
            int i = (int)(modgrad[(size_t)y * W + x] * bin_coef);
            order[start[i]++] = y * W + x;
        }
    free(count); free(start);
    if (g_seed_order == 1 || c->lsd_seed_order == 1) {
        int32_t* bins = (int32_t*)malloc(sizeof(int32_t) * (size_t)(H - 1) * (W - 1));
        size_t k = 0;
        for (int y = 0; y < H - 1; ++y)
            for (int x = 0; x < W - 1; ++x) bins[k++] = (int)(modgrad[(size_t)y * W + x] * bin_coef);
        lfo_std_sort_seed_order(bins, H, W, order);
        free(bins);
    }
    return acc;
}

/* ------------------------------------------------------------------------- */
typedef struct { int x, y; double angle, modgrad; int addr; } RegPt;
typedef struct { double x1, y1, x2, y2, width, x, y, theta, dx, dy, prec, p; } Rect;
typedef struct {
    int W, H;
    const double* angles;
    const double* modgrad;
    uint8_t* used;
    double LOG_NT;
} Lsd;

static inline double distSq(double x1, double y1, double x2, double y2) { return (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1); }
static inline double dist(double x1, double y1, double x2, double y2) { return sqrt(distSq(x1, y1, x2, y2)); }

static inline double angle_diff_signed(double a, double b)
{
    double diff = a - b;
    while (diff <= -LSD_PI) diff += M_2__PI;
    while (diff > LSD_PI) diff -= M_2__PI;
    return diff;
}
static inline double angle_diff(double a, double b) { return fabs(angle_diff_signed(a, b)); }

static inline int double_equal(double a, double b)
{
    if (a == b) return 1;
    double abs_diff = fabs(a - b);
    double aa = fabs(a), bb = fabs(b);
    double abs_max = (aa > bb) ? aa : bb;
    if (abs_max < DBL_MIN) abs_max = DBL_MIN;
    return (abs_diff / abs_max) <= (REL_ERR_FACTOR * DBL_EPSILON);
}

static inline int is_aligned(const Lsd* L, int address, double theta, double prec)
{
    if (address < 0) return 0;
    double a = L->angles[address];
    if (a == NOTDEF) return 0;
    double n_theta = theta - a;
    if (n_theta < 0) n_theta = -n_theta;
    if (n_theta > M_3_2_PI) {
        n_theta -= M_2__PI;
        if (n_theta < 0) n_theta = -n_theta;
    }
    return n_theta <= prec;
}

static void region_grow(Lsd* L, int sx, int sy, RegPt* reg, int* reg_size, double* reg_angle, double prec)
{
    const int W = L->W, H = L->H;
    int n = 1;
    int addr = sx + sy * W;
    reg[0].x = sx; reg[0].y = sy; reg[0].addr = addr;
    *reg_angle = L->angles[addr];
    reg[0].angle = *reg_angle;
    reg[0].modgrad = L->modgrad[addr];
    float sumdx = (float)lfo_cos(*reg_angle);
    float sumdy = (float)lfo_sin(*reg_angle);
    L->used[addr] = 1;
    for (int i = 0; i < n; ++i) {
        const int px = reg[i].x, py = reg[i].y;
        int xx_min = px - 1 > 0 ? px - 1 : 0, xx_max = px + 1 < W - 1 ? px + 1 : W - 1;
        int yy_min = py - 1 > 0 ? py - 1 : 0, yy_max = py + 1 < H - 1 ? py + 1 : H - 1;
        for (int yy = yy_min; yy <= yy_max; ++yy) {
            int c_addr = xx_min + yy * W;
            for (int xx = xx_min; xx <= xx_max; ++xx, ++c_addr) {
                if (L->used[c_addr] != 1 && is_aligned(L, c_addr, *reg_angle, prec)) {
                    L->used[c_addr] = 1;
                    RegPt* rp = &reg[n];
                    rp->x = xx; rp->y = yy; rp->addr = c_addr;
                    rp->modgrad = L->modgrad[c_addr];
                    const double angle = L->angles[c_addr];
                    rp->angle = angle;
                    ++n;
                    /* cos(float(angle)) resolves to ::cos(double); the float sum takes the rounded double sum */
                    sumdx = (float)((double)sumdx + lfo_cos((double)(float)angle));
                    sumdy = (float)((double)sumdy + lfo_sin((double)(float)angle));
                    *reg_angle = (double)lfo_fast_atan2_deg(sumdy, sumdx) * DEG_TO_RADS;
                }
            }
        }
    }
    *reg_size = n;
}

static double get_theta(const RegPt* reg, int reg_size, double x, double y, double reg_angle, double prec)
{
    double Ixx = 0.0, Iyy = 0.0, Ixy = 0.0;
    for (int i = 0; i < reg_size; ++i) {
        const double regx = reg[i].x, regy = reg[i].y, weight = reg[i].modgrad;
        double dx = regx - x, dy = regy - y;
        Ixx += dy * dy * weight;
        Iyy += dx * dx * weight;
        Ixy -= dx * dy * weight;
    }
    double lambda = 0.5 * (Ixx + Iyy - sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
    double theta = (fabs(Ixx) > fabs(Iyy))
        ? (double)lfo_fast_atan2_deg((float)(lambda - Ixx), (float)Ixy)
        : (double)lfo_fast_atan2_deg((float)Ixy, (float)(lambda - Iyy));
    theta *= DEG_TO_RADS;
    if (angle_diff(theta, reg_angle) > prec) theta += LSD_PI;
    return theta;
}

static void region2rect(const RegPt* reg, int reg_size, double reg_angle, double prec, double p, Rect* rec)
{
    double x = 0, y = 0, sum = 0;
    for (int i = 0; i < reg_size; ++i) {
        const double weight = reg[i].modgrad;
        x += (double)reg[i].x * weight;
        y += (double)reg[i].y * weight;
        sum += weight;
    }
    x /= sum;
    y /= sum;
    double theta = get_theta(reg, reg_size, x, y, reg_angle, prec);
    double dx = lfo_cos(theta), dy = lfo_sin(theta);
    double l_min = 0, l_max = 0, w_min = 0, w_max = 0;
    for (int i = 0; i < reg_size; ++i) {
        double regdx = (double)reg[i].x - x;
        double regdy = (double)reg[i].y - y;
        double l = regdx * dx + regdy * dy;
        double w = -regdx * dy + regdy * dx;
        if (l > l_max) l_max = l; else if (l < l_min) l_min = l;
        if (w > w_max) w_max = w; else if (w < w_min) w_min = w;
    }
    rec->x1 = x + l_min * dx;
    rec->y1 = y + l_min * dy;
    rec->x2 = x + l_max * dx;
    rec->y2 = y + l_max * dy;
    rec->width = w_max - w_min;
    rec->x = x; rec->y = y; rec->theta = theta; rec->dx = dx; rec->dy = dy;
    rec->prec = prec; rec->p = p;
    if (rec->width < 1.0) rec->width = 1.0;
}

static int reduce_region_radius(Lsd* L, RegPt* reg, int* reg_size, double reg_angle, double prec, double p,
                                Rect* rec, double density, double density_th)
{
    double xc = (double)reg[0].x, yc = (double)reg[0].y;
    double radSq1 = distSq(xc, yc, rec->x1, rec->y1);
    double radSq2 = distSq(xc, yc, rec->x2, rec->y2);
    double radSq = radSq1 > radSq2 ? radSq1 : radSq2;
    while (density < density_th) {
        radSq *= 0.75 * 0.75;
        for (int i = 0; i < *reg_size; ++i) {
            if (distSq(xc, yc, (double)reg[i].x, (double)reg[i].y) > radSq) {
                L->used[reg[i].addr] = 0;
                RegPt t = reg[i]; reg[i] = reg[*reg_size - 1]; reg[*reg_size - 1] = t;
                --(*reg_size);
                --i;
            }
        }
        if (*reg_size < 2) return 0;
        region2rect(reg, *reg_size, reg_angle, prec, p, rec);
        density = (double)(*reg_size) / (dist(rec->x1, rec->y1, rec->x2, rec->y2) * rec->width);
    }
    return 1;
}

static int refine(Lsd* L, RegPt* reg, int* reg_size, double reg_angle, double prec, double p, Rect* rec, double density_th)
{
    double density = (double)(*reg_size) / (dist(rec->x1, rec->y1, rec->x2, rec->y2) * rec->width);
    if (density >= density_th) return 1;
    double xc = (double)reg[0].x, yc = (double)reg[0].y;
    const double ang_c = reg[0].angle;
    double sum = 0, s_sum = 0;
    int n = 0;
    for (int i = 0; i < *reg_size; ++i) {
        L->used[reg[i].addr] = 0;
        if (dist(xc, yc, (double)reg[i].x, (double)reg[i].y) < rec->width) {
            double ang_d = angle_diff_signed(reg[i].angle, ang_c);
            sum += ang_d;
            s_sum += ang_d * ang_d;
            ++n;
        }
    }
    double mean_angle = sum / (double)n;
    double tau = 2.0 * sqrt((s_sum - 2.0 * mean_angle * sum) / (double)n + mean_angle * mean_angle);
    region_grow(L, reg[0].x, reg[0].y, reg, reg_size, &reg_angle, tau);
    if (*reg_size < 2) return 0;
    region2rect(reg, *reg_size, reg_angle, prec, p, rec);
    density = (double)(*reg_size) / (dist(rec->x1, rec->y1, rec->x2, rec->y2) * rec->width);
    if (density < density_th) return reduce_region_radius(L, reg, reg_size, reg_angle, prec, p, rec, density, density_th);
    return 1;
}

static double log_gamma_lanczos(double x)
{
    static const double q[7] = { 75122.6331530, 80916.6278952, 36308.2951477, 8687.24529705,
                                 1168.92649479, 83.8676043424, 2.50662827511 };
    double a = (x + 0.5) * lfo_log(x + 5.5) - (x + 5.5);
    double b = 0;
    for (int n = 0; n < 7; ++n) {
        a -= lfo_log(x + (double)n);
        b += q[n] * lfo_pow(x, (double)n);
    }
    return a + lfo_log(b);
}
static double log_gamma_windschitl(double x)
{
    return 0.918938533204673 + (x - 0.5) * lfo_log(x) - x
         + 0.5 * x * lfo_log(x * lfo_sinh_small(1 / x) + 1 / (810.0 * lfo_pow(x, 6.0)));
}
static double log_gamma(double x) { return x > 15.0 ? log_gamma_windschitl(x) : log_gamma_lanczos(x); }

static double nfa(const Lsd* L, int n, int k, double p)
{
    const double LOG_NT = L->LOG_NT;
    if (n == 0 || k == 0) return -LOG_NT;
    if (n == k) return -LOG_NT - (double)n * lfo_log10(p);
    double p_term = p / (1 - p);
    double log1term = ((double)n + 1) - log_gamma((double)k + 1) - log_gamma((double)(n - k) + 1)
                    + (double)k * lfo_log(p) + (double)(n - k) * lfo_log(1.0 - p);
    double term = lfo_exp(log1term);
    if (double_equal(term, 0)) {
        if (k > n * p) return -log1term / 2.30258509299404568402 - LOG_NT;
        else return -LOG_NT;
    }
    double bin_tail = term;
    double tolerance = 0.1;
    for (int i = k + 1; i <= n; ++i) {
        double bin_term = (double)(n - i + 1) / (double)i;
        double mult_term = bin_term * p_term;
        term *= mult_term;
        bin_tail += term;
        if (bin_term < 1) {
            double err = term * ((1 - lfo_pow(mult_term, (double)(n - i + 1))) / (1 - mult_term) - 1);
            if (err < tolerance * fabs(-lfo_log10(bin_tail) - LOG_NT) * bin_tail) break;
        }
    }
    return -lfo_log10(bin_tail) - LOG_NT;
}

typedef struct { int x, y; int taken; } Edge;
static int edge_less(const Edge* a, const Edge* b)
{
    if (a->x == b->x) return a->y < b->y;
    return a->x < b->x;
}

static double rect_nfa(const Lsd* L, const Rect* rec)
{
    int total_pts = 0, alg_pts = 0;
    double half_width = rec->width / 2.0;
    double dyhw = rec->dy * half_width;
    double dxhw = rec->dx * half_width;
    Edge e[4];
    e[0].x = (int)(rec->x1 - dyhw); e[0].y = (int)(rec->y1 + dxhw); e[0].taken = 0;
    e[1].x = (int)(rec->x2 - dyhw); e[1].y = (int)(rec->y2 + dxhw); e[1].taken = 0;
    e[2].x = (int)(rec->x2 + dyhw); e[2].y = (int)(rec->y2 - dxhw); e[2].taken = 0;
    e[3].x = (int)(rec->x1 + dyhw); e[3].y = (int)(rec->y1 - dxhw); e[3].taken = 0;
    /* sort the 4 corners by (x, then y): insertion sort == std::sort result for a strict weak order
       up to the order of equal elements, which are identical points here */
    for (int i = 1; i < 4; ++i) {
        Edge t = e[i];
        int j = i - 1;
        while (j >= 0 && edge_less(&t, &e[j])) { e[j + 1] = e[j]; --j; }
        e[j + 1] = t;
    }
    Edge* min_y = &e[0];
    Edge* max_y = &e[0];
    for (int i = 1; i < 4; ++i) {
        if (min_y->y > e[i].y) min_y = &e[i];
        if (max_y->y < e[i].y) max_y = &e[i];
    }
    min_y->taken = 1;
    Edge* leftmost = 0;
    for (int i = 0; i < 4; ++i)
        if (!e[i].taken) { if (!leftmost) leftmost = &e[i]; else if (leftmost->x > e[i].x) leftmost = &e[i]; }
    leftmost->taken = 1;
    Edge* rightmost = 0;
    for (int i = 0; i < 4; ++i)
        if (!e[i].taken) { if (!rightmost) rightmost = &e[i]; else if (rightmost->x < e[i].x) rightmost = &e[i]; }
    rightmost->taken = 1;
    Edge* tailp = 0;
    for (int i = 0; i < 4; ++i)
        if (!e[i].taken) { if (!tailp) tailp = &e[i]; else if (tailp->x > e[i].x) tailp = &e[i]; }
    tailp->taken = 1;

    /* integer quotients, and the y-vs-x comparisons, as in OpenCV 3.x */
    double flstep = (min_y->y != leftmost->y) ? (double)((min_y->x - leftmost->x) / (min_y->y - leftmost->y)) : 0;
    double slstep = (leftmost->y != tailp->x) ? (double)((leftmost->x - tailp->x) / (leftmost->y - tailp->x)) : 0;
    double frstep = (min_y->y != rightmost->y) ? (double)((min_y->x - rightmost->x) / (min_y->y - rightmost->y)) : 0;
    double srstep = (rightmost->y != tailp->x) ? (double)((rightmost->x - tailp->x) / (rightmost->y - tailp->x)) : 0;
    double lstep = flstep, rstep = frstep;
    double left_x = min_y->x, right_x = min_y->x;
    int min_iter = min_y->y, max_iter = max_y->y;
    for (int y = min_iter; y <= max_iter; ++y) {
        if (y < 0 || y >= L->H) continue;
        int adx = y * L->W + (int)left_x;
        for (int x = (int)left_x; x <= (int)right_x; ++x, ++adx) {
            if (x < 0 || x >= L->W) continue;
            ++total_pts;
            if (is_aligned(L, adx, rec->theta, rec->prec)) ++alg_pts;
        }
        if (y >= leftmost->y) lstep = slstep;
        if (y >= rightmost->y) rstep = srstep;
        left_x += lstep;
        right_x += rstep;
    }
    return nfa(L, total_pts, alg_pts, rec->p);
}

static double rect_improve(const Lsd* L, Rect* rec, double LOG_EPS)
{
    double delta = 0.5;
    double delta_2 = delta / 2.0;
    double log_nfa = rect_nfa(L, rec);
    if (log_nfa > LOG_EPS) return log_nfa;
    Rect r = *rec;
    for (int n = 0; n < 5; ++n) {
        r.p /= 2;
        r.prec = r.p * LSD_PI;
        double v = rect_nfa(L, &r);
        if (v > log_nfa) { log_nfa = v; *rec = r; }
    }
    if (log_nfa > LOG_EPS) return log_nfa;
    r = *rec;
    for (int n = 0; n < 5; ++n) {
        if ((r.width - delta) >= 0.5) {
            r.width -= delta;
            double v = rect_nfa(L, &r);
            if (v > log_nfa) { *rec = r; log_nfa = v; }
        }
    }
    if (log_nfa > LOG_EPS) return log_nfa;
    r = *rec;
    for (int n = 0; n < 5; ++n) {
        if ((r.width - delta) >= 0.5) {
            r.x1 += -r.dy * delta_2;
            r.y1 += r.dx * delta_2;
            r.x2 += -r.dy * delta_2;
            r.y2 += r.dx * delta_2;
            r.width -= delta;
            double v = rect_nfa(L, &r);
            if (v > log_nfa) { *rec = r; log_nfa = v; }
        }
    }
    if (log_nfa > LOG_EPS) return log_nfa;
    r = *rec;
    for (int n = 0; n < 5; ++n) {
        if ((r.width - delta) >= 0.5) {
            r.x1 -= -r.dy * delta_2;
            r.y1 -= r.dx * delta_2;
            r.x2 -= -r.dy * delta_2;
            r.y2 -= r.dx * delta_2;
            r.width -= delta;
            double v = rect_nfa(L, &r);
            if (v > log_nfa) { *rec = r; log_nfa = v; }
        }
    }
    if (log_nfa > LOG_EPS) return log_nfa;
    r = *rec;
    for (int n = 0; n < 5; ++n) {
        if ((r.width - delta) >= 0.5) {
            r.p /= 2;
            r.prec = r.p * LSD_PI;
            double v = rect_nfa(L, &r);
            if (v > log_nfa) { *rec = r; log_nfa = v; }
        }
    }
    return log_nfa;
}

int lfo_lsd_detect(const lfo_config* c, const uint8_t* img, int rows, int cols,
                   float* lines4, double* extra3, int cap)
{
    int H, W;
    lfo_lsd_scaled_size(c, rows, cols, &H, &W);
    const size_t np = (size_t)H * W;
    double* scaled = (double*)malloc(np * sizeof(double));
    double* angles = (double*)malloc(np * sizeof(double));
    double* modgrad = (double*)malloc(np * sizeof(double));
    int32_t* order = (int32_t*)malloc(np * sizeof(int32_t));
    uint8_t* used = (uint8_t*)calloc(np, 1);
    RegPt* reg = (RegPt*)malloc(np * sizeof(RegPt));
    lfo_lsd_scaled_image(c, img, rows, cols, scaled);
    int n_order = lfo_lsd_ll_angle(c, scaled, H, W, angles, modgrad, order);

    const double prec = LSD_PI * c->lsd_ang_th / 180;
    const double p = c->lsd_ang_th / 180;
    Lsd L;
    L.W = W; L.H = H; L.angles = angles; L.modgrad = modgrad; L.used = used;
    L.LOG_NT = 5 * (lfo_log10((double)W) + lfo_log10((double)H)) / 2 + lfo_log10(11.0);
    const int min_reg_size = (int)(-L.LOG_NT / lfo_log10(p));
    int n_lines = 0;
    for (int i = 0; i < n_order; ++i) {
        int adx = order[i];
        if (used[adx] != 0 || angles[adx] == NOTDEF) continue;
        int reg_size;
        double reg_angle;
        region_grow(&L, adx % W, adx / W, reg, &reg_size, &reg_angle, prec);
        if (reg_size < min_reg_size) continue;
        Rect rec;
        region2rect(reg, reg_size, reg_angle, prec, p, &rec);
        double log_nfa = -1;
        if (c->lsd_refine > 0) {
            if (!refine(&L, reg, &reg_size, reg_angle, prec, p, &rec, c->lsd_density_th)) continue;
            if (c->lsd_refine >= 2) {
                log_nfa = rect_improve(&L, &rec, c->lsd_log_eps);
                if (log_nfa <= c->lsd_log_eps) continue;
            }
        }
        rec.x1 += 0.5; rec.y1 += 0.5; rec.x2 += 0.5; rec.y2 += 0.5;
        if (c->lsd_scale != 1) {
            rec.x1 /= c->lsd_scale; rec.y1 /= c->lsd_scale;
            rec.x2 /= c->lsd_scale; rec.y2 /= c->lsd_scale;
            rec.width /= c->lsd_scale;
        }
        if (n_lines < cap) {
            lines4[4 * n_lines + 0] = (float)rec.x1;
            lines4[4 * n_lines + 1] = (float)rec.y1;
            lines4[4 * n_lines + 2] = (float)rec.x2;
            lines4[4 * n_lines + 3] = (float)rec.y2;
            if (extra3) {
                extra3[3 * n_lines + 0] = rec.width;
                extra3[3 * n_lines + 1] = rec.p;
                extra3[3 * n_lines + 2] = log_nfa;
            }
        }
        ++n_lines;
    }
    free(scaled); free(angles); free(modgrad); free(order); free(used); free(reg);
    return n_lines < cap ? n_lines : cap;
}
