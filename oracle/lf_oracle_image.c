/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see lf_oracle.h).
 * Image stages a-1..a-3 and the LBD gradient inputs of a-9.
 * The arithmetic of cv2.* calls is restated from the published OpenCV 3.x
 * implementation (the version family ROS Kinetic/Melodic ship); OpenCV is not
 * in /root/reference, so these restatements are PARITY UNPINNED.
 */
#include "lf_oracle.h"
#include "lf_detmath.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* OpenCV cvRound: round half to even */
static int cv_round(double v)
{
    double f = (double)(long long)v;
    double d = v - f;
    long long i = (long long)f;
    if (d > 0.5 || (d == 0.5 && (i & 1))) i += 1;
    else if (d < -0.5 || (d == -0.5 && (i & 1))) i -= 1;
    return (int)i;
}
static int cv_floor(double v)
{
    int i = (int)v;
    return i - (v < (double)i ? 1 : 0);
}
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) {
        if (p < 0) p = -p;
        else p = 2 * (n - 1) - p;
    }
    return p;
}

/* ------------------------------------------------------------------ a-1
 * line_detector_node.py:163-175: cv2.resize(INTER_NEAREST) only when the size
 * differs, crop rows [top_cutoff:], AntiInstagram.applyTransform ->
 * scale_and_shift.py:25-33 (float32(px)*float32(scale[c]) + float32(shift[c])),
 * cv2.convertScaleAbs -> saturate_u8(round_half_even(|x|)).
 */
void lfo_preprocess(const lfo_config* c, const uint8_t* in, uint8_t* out)
{
    const int Hc = lfo_work_rows(c), W = lfo_work_cols(c);
    const int resize = (c->img_rows != c->in_rows) || (c->img_cols != c->in_cols);
    /* OpenCV resizeNN: ifx = 1/(dst/src); sx = min(floor(x*ifx), src-1) */
    const double fx = (double)c->img_cols / (double)c->in_cols;
    const double fy = (double)c->img_rows / (double)c->in_rows;
    const double ifx = 1.0 / fx, ify = 1.0 / fy;
    for (int y = 0; y < Hc; ++y) {
        int yy = y + c->top_cutoff;
        int sy = resize ? cv_floor(yy * ify) : yy;
        if (sy > c->in_rows - 1) sy = c->in_rows - 1;
        for (int x = 0; x < W; ++x) {
            int sx = resize ? cv_floor(x * ifx) : x;
            if (sx > c->in_cols - 1) sx = c->in_cols - 1;
            const uint8_t* p = in + ((size_t)sy * c->in_cols + sx) * 3;
            uint8_t* q = out + ((size_t)y * W + x) * 3;
            for (int ch = 0; ch < 3; ++ch) {
                float v = (float)p[ch] * c->ai_scale[ch];
                v = v + c->ai_shift[ch];
                float a = v < 0 ? -v : v;
                int r = cv_round((double)a);
                q[ch] = (uint8_t)clampi(r, 0, 255);
            }
        }
    }
}

/* ------------------------------------------------------------------ a-2
 * line_detector_lsd.py:138 cv2.cvtColor(bgr, COLOR_BGR2HSV), 8-bit:
 * OpenCV RGB2HSV_b fixed-point tables, hsv_shift = 12, hrange = 180.
 */
static int g_sdiv[256], g_hdiv180[256], g_hsv_init = 0;
static void hsv_tables(void)
{
    if (g_hsv_init) return;
    g_sdiv[0] = g_hdiv180[0] = 0;
    for (int i = 1; i < 256; ++i) {
        g_sdiv[i] = cv_round((255 << 12) / (1.0 * i));
        g_hdiv180[i] = cv_round((180 << 12) / (6.0 * i));
    }
    g_hsv_init = 1;
}

void lfo_bgr2hsv(const uint8_t* bgr, int npix, uint8_t* hsv)
{
    hsv_tables();
    for (int i = 0; i < npix; ++i) {
        int b = bgr[3 * i], g = bgr[3 * i + 1], r = bgr[3 * i + 2];
        int v = b, vmin = b;
        if (g > v) v = g;
        if (r > v) v = r;
        if (g < vmin) vmin = g;
        if (r < vmin) vmin = r;
        int diff = v - vmin;
        int vr = v == r ? -1 : 0;
        int vg = v == g ? -1 : 0;
        int s = (diff * g_sdiv[v] + (1 << 11)) >> 12;
        int h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
        h = (h * g_hdiv180[diff] + (1 << 11)) >> 12;
        h += h < 0 ? 180 : 0;
        hsv[3 * i] = (uint8_t)clampi(h, 0, 255);
        hsv[3 * i + 1] = (uint8_t)s;
        hsv[3 * i + 2] = (uint8_t)v;
    }
}

/* ------------------------------------------------------------------ a-3
 * line_detector_lsd.py:38-47: cv2.inRange inclusive on the three channels,
 * red = bitwise_or of two boxes.
 */
void lfo_color_masks(const lfo_config* c, const uint8_t* hsv, int npix, uint8_t* bw3)
{
    for (int i = 0; i < npix; ++i) {
        int in[4];
        for (int k = 0; k < 4; ++k) {
            int ok = 1;
            for (int ch = 0; ch < 3; ++ch) {
                int v = hsv[3 * i + ch];
                ok &= (v >= c->hsv_lo[k][ch]) & (v <= c->hsv_hi[k][ch]);
            }
            in[k] = ok;
        }
        bw3[i] = in[0] ? 255 : 0;
        bw3[npix + i] = in[1] ? 255 : 0;
        bw3[2 * npix + i] = (in[2] | in[3]) ? 255 : 0;
    }
}

/* line_detector_lsd.py:52-53: getStructuringElement(MORPH_ELLIPSE,(k,k)) + cv2.dilate.
 * OpenCV builds the ellipse row by row: dx = round(c*sqrt((r*r-dy*dy)/r^2)); a 3x3
 * "ellipse" is the plus shape.  Border = constant, ignored by the max.
 */
void lfo_dilate_ellipse(const uint8_t* src, int rows, int cols, int ksize, uint8_t* dst)
{
    int r = ksize / 2, cc = ksize / 2;
    double inv_r2 = r ? 1.0 / ((double)r * r) : 0.0;
    int* j1 = (int*)malloc(sizeof(int) * ksize);
    int* j2 = (int*)malloc(sizeof(int) * ksize);
    for (int i = 0; i < ksize; ++i) {
        int dy = i - r;
        j1[i] = 0; j2[i] = 0;
        if ((dy < 0 ? -dy : dy) <= r) {
            int dx = cv_round(cc * sqrt((r * r - dy * dy) * inv_r2));
            j1[i] = cc - dx > 0 ? cc - dx : 0;
            j2[i] = cc + dx + 1 < ksize ? cc + dx + 1 : ksize;
        }
    }
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            int m = 0;
            for (int i = 0; i < ksize; ++i) {
                int yy = y + i - r;
                if (yy < 0 || yy >= rows) continue;
                for (int j = j1[i]; j < j2[i]; ++j) {
                    int xx = x + j - cc;
                    if (xx < 0 || xx >= cols) continue;
                    int v = src[(size_t)yy * cols + xx];
                    if (v > m) m = v;
                }
            }
            dst[(size_t)y * cols + x] = (uint8_t)m;
        }
    free(j1); free(j2);
}

/* ------------------------------------------------------------------ a-2 (edges)
 * line_detector_lsd.py:60-62,139: cv2.Canny(bgr, lo, hi, apertureSize=3) on the
 * 3-channel image.  OpenCV: Sobel 3x3 to s16 with BORDER_REPLICATE per channel,
 * L1 magnitude, per pixel the channel with the largest magnitude (first wins
 * ties), non-maximum suppression with the TG22 fixed-point sector test,
 * hysteresis over 8-neighbours.  Output 0/255.
 */
void lfo_canny_bgr(const uint8_t* bgr, int rows, int cols, double lo_t, double hi_t, uint8_t* edges)
{
    if (lo_t > hi_t) { double t = lo_t; lo_t = hi_t; hi_t = t; }
    const int low = cv_floor(lo_t), high = cv_floor(hi_t);
    const size_t np = (size_t)rows * cols;
    int* mag = (int*)calloc((size_t)(rows + 2) * (cols + 2), sizeof(int));
    short* gx = (short*)malloc(np * sizeof(short));
    short* gy = (short*)malloc(np * sizeof(short));
    const int ms = cols + 2;
#define PX(y, x, ch) ((int)bgr[((size_t)clampi((y), 0, rows - 1) * cols + clampi((x), 0, cols - 1)) * 3 + (ch)])
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            int best = -1, bx = 0, by = 0;
            for (int ch = 0; ch < 3; ++ch) {
                int dx = (PX(y - 1, x + 1, ch) - PX(y - 1, x - 1, ch))
                       + 2 * (PX(y, x + 1, ch) - PX(y, x - 1, ch))
                       + (PX(y + 1, x + 1, ch) - PX(y + 1, x - 1, ch));
                int dy = (PX(y + 1, x - 1, ch) - PX(y - 1, x - 1, ch))
                       + 2 * (PX(y + 1, x, ch) - PX(y - 1, x, ch))
                       + (PX(y + 1, x + 1, ch) - PX(y - 1, x + 1, ch));
                int m = (dx < 0 ? -dx : dx) + (dy < 0 ? -dy : dy);
                if (m > best) { best = m; bx = dx; by = dy; }
            }
            mag[(size_t)(y + 1) * ms + x + 1] = best;
            gx[(size_t)y * cols + x] = (short)bx;
            gy[(size_t)y * cols + x] = (short)by;
        }
#undef PX
    /* map: 0 = candidate (weak), 1 = suppressed, 2 = edge */
    uint8_t* map = (uint8_t*)malloc(np);
    int32_t* stack = (int32_t*)malloc(np * sizeof(int32_t));
    size_t sp = 0;
    const int TG22 = (int)(0.4142135623730950488016887242097 * (1 << 15) + 0.5);
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            const int* pm = mag + (size_t)(y + 1) * ms + x + 1;
            int m = pm[0];
            int keep = 0;
            if (m > low) {
                int xs = gx[(size_t)y * cols + x], ys = gy[(size_t)y * cols + x];
                int ax = xs < 0 ? -xs : xs;
                int ay = (ys < 0 ? -ys : ys) << 15;
                int tg22x = ax * TG22;
                if (ay < tg22x) {
                    keep = (m > pm[-1] && m >= pm[1]);
                } else {
                    int tg67x = tg22x + (ax << 16);
                    if (ay > tg67x) {
                        keep = (m > pm[-ms] && m >= pm[ms]);
                    } else {
                        int s = (xs ^ ys) < 0 ? -1 : 1;
                        keep = (m > pm[-ms - s] && m > pm[ms + s]);
                    }
                }
            }
            size_t a = (size_t)y * cols + x;
            if (!keep) map[a] = 1;
            else if (m > high) { map[a] = 2; stack[sp++] = (int32_t)a; }
            else map[a] = 0;
        }
    while (sp) {
        int32_t a = stack[--sp];
        int y = a / cols, x = a % cols;
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                int yy = y + dy, xx = x + dx;
                if ((dx | dy) == 0 || yy < 0 || yy >= rows || xx < 0 || xx >= cols) continue;
                size_t b = (size_t)yy * cols + xx;
                if (map[b] == 0) { map[b] = 2; stack[sp++] = (int32_t)b; }
            }
    }
    for (size_t i = 0; i < np; ++i) edges[i] = map[i] == 2 ? 255 : 0;
    free(mag); free(gx); free(gy); free(map); free(stack);
}

/* ------------------------------------------------------------------ a-9 inputs
 * binary_descriptor_custom.cpp:546-547 cvtColor(BGR2GRAY): fixed point,
 * (B*1868 + G*9617 + R*4899 + 2^13) >> 14.
 */
void lfo_bgr2gray(const uint8_t* bgr, int npix, uint8_t* gray)
{
    for (int i = 0; i < npix; ++i) {
        int b = bgr[3 * i], g = bgr[3 * i + 1], r = bgr[3 * i + 2];
        gray[i] = (uint8_t)((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14);
    }
}

/* binary_descriptor_custom.cpp:358 cv::GaussianBlur(img, img, Size(5,5), 1) on u8.
 * OpenCV 3.x (< 3.4.1) separable fixed-point path: the float kernel
 * getGaussianKernel(5,1) is rounded to 8 fractional bits -> {14,63,103,63,14}
 * (sum 257), rows then columns in int32, result (acc + 2^15) >> 16 saturated;
 * BORDER_REFLECT_101.
 */
void lfo_gaussian5_u8(const uint8_t* src, int rows, int cols, uint8_t* dst)
{
    static const int k[5] = { 14, 63, 103, 63, 14 };
    int* tmp = (int*)malloc((size_t)rows * cols * sizeof(int));
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            int s = 0;
            for (int j = -2; j <= 2; ++j) s += k[j + 2] * src[(size_t)y * cols + reflect101(x + j, cols)];
            tmp[(size_t)y * cols + x] = s;
        }
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            int s = 0;
            for (int j = -2; j <= 2; ++j) s += k[j + 2] * tmp[(size_t)reflect101(y + j, rows) * cols + x];
            dst[(size_t)y * cols + x] = (uint8_t)clampi((s + (1 << 15)) >> 16, 0, 255);
        }
    free(tmp);
}

/* binary_descriptor_custom.cpp:395-396 cv::Sobel(img, d, CV_16SC1, 1,0,3) / (0,1,3); BORDER_REFLECT_101 */
void lfo_sobel3_s16(const uint8_t* src, int rows, int cols, int16_t* dx, int16_t* dy)
{
#define P(y, x) ((int)src[(size_t)reflect101((y), rows) * cols + reflect101((x), cols)])
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            int vx = (P(y - 1, x + 1) - P(y - 1, x - 1)) + 2 * (P(y, x + 1) - P(y, x - 1)) + (P(y + 1, x + 1) - P(y + 1, x - 1));
            int vy = (P(y + 1, x - 1) - P(y - 1, x - 1)) + 2 * (P(y + 1, x) - P(y - 1, x)) + (P(y + 1, x + 1) - P(y - 1, x + 1));
            dx[(size_t)y * cols + x] = (int16_t)vx;
            dy[(size_t)y * cols + x] = (int16_t)vy;
        }
#undef P
}
