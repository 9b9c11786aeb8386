/*
 * ORACLE (test infrastructure only).  See lf_detmath.h for the rationale.
 * Pure IEEE-754 double arithmetic, no FMA, no libm calls except sqrt (which is
 * correctly rounded on both sides).  Build with -ffp-contract=off.
 */
#include "lf_detmath.h"
#include <stdint.h>
#include <string.h>
#include <math.h>

static inline uint64_t d2u(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double u2d(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }

/* 2^k for k in [-1022, 1023] */
static inline double pow2i(int k) { return u2d((uint64_t)(k + 1023) << 52); }

/* ------------------------------------------------------------------ exp */
static const double LN2HI = 6.93147180369123816490e-01;
static const double LN2LO = 1.90821492927058770002e-10;
static const double INVLN2 = 1.44269504088896338700e+00;
static const double EP1 = 1.66666666666666019037e-01;
static const double EP2 = -2.77777777770155933842e-03;
static const double EP3 = 6.61375632143793436117e-05;
static const double EP4 = -1.65339022054652515390e-06;
static const double EP5 = 4.13813679705723846039e-08;

double lfo_exp(double x)
{
    if (x != x) return x;
    if (x > 709.782712893383973096) return INFINITY;
    if (x < -745.13321910194110842) return 0.0;
    double t = x * INVLN2 + (x < 0 ? -0.5 : 0.5);
    int k = (int)t;
    double fk = (double)k;
    double hi = x - fk * LN2HI;
    double lo = fk * LN2LO;
    double r = hi - lo;
    double z = r * r;
    double c = r - z * (EP1 + z * (EP2 + z * (EP3 + z * (EP4 + z * EP5))));
    double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    if (k >= -1021 && k <= 1023) return y * pow2i(k);
    if (k > 1023) return (y * pow2i(1023)) * pow2i(k - 1023);
    return (y * pow2i(k + 1000)) * pow2i(-1000);
}

/* ------------------------------------------------------------------ log */
static const double LG1 = 6.666666666666735130e-01;
static const double LG2 = 3.999999999940941908e-01;
static const double LG3 = 2.857142874366239149e-01;
static const double LG4 = 2.222219843214978396e-01;
static const double LG5 = 1.818357216161805012e-01;
static const double LG6 = 1.531383769920937332e-01;
static const double LG7 = 1.479819860511658591e-01;

double lfo_log(double x)
{
    if (x != x) return x;
    if (x < 0.0) return NAN;
    if (x == 0.0) return -INFINITY;
    if (x == INFINITY) return x;
    int k = 0;
    uint64_t u = d2u(x);
    if ((u >> 52) == 0) { /* subnormal */
        x = x * 18014398509481984.0; /* 2^54 */
        k -= 54;
        u = d2u(x);
    }
    uint32_t hx = (uint32_t)(u >> 32);
    uint32_t lx = (uint32_t)u;
    k += (int)(hx >> 20) - 1023;
    hx &= 0x000fffffu;
    uint32_t i = (hx + 0x95f64u) & 0x100000u;
    hx |= (i ^ 0x3ff00000u);
    k += (int)(i >> 20);
    double m = u2d(((uint64_t)hx << 32) | lx);
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double dk = (double)k;
    double z = s * s;
    double w = z * z;
    double t1 = w * (LG2 + w * (LG4 + w * LG6));
    double t2 = z * (LG1 + w * (LG3 + w * (LG5 + w * LG7)));
    double R = t2 + t1;
    double hfsq = 0.5 * f * f;
    if (k == 0) return f - (hfsq - s * (hfsq + R));
    return dk * LN2HI - ((hfsq - (s * (hfsq + R) + dk * LN2LO)) - f);
}

static const double LN10 = 2.30258509299404568402e+00;
double lfo_log10(double x) { return lfo_log(x) / LN10; }

/* -------------------------------------------------------------- sin/cos */
static const double INVPIO2 = 6.36619772367581382433e-01;
static const double PIO2_1 = 1.57079632673412561417e+00;
static const double PIO2_2 = 6.07710050630396597660e-11;
static const double PIO2_2T = 2.02226624879595063154e-21;
static const double S1 = -1.66666666666666324348e-01;
static const double S2 = 8.33333333332248946124e-03;
static const double S3 = -1.98412698298579493134e-04;
static const double S4 = 2.75573137070700676789e-06;
static const double S5 = -2.50507602534068634195e-08;
static const double S6 = 1.58969099521155010221e-10;
static const double C1 = 4.16666666666666019037e-02;
static const double C2 = -1.38888888888741095749e-03;
static const double C3 = 2.48015872894767294178e-05;
static const double C4 = -2.75573143513906633035e-07;
static const double C5 = 2.08757232129817482790e-09;
static const double C6 = -1.13596475577881948265e-11;

static inline double ksin(double x, double y)
{
    double z = x * x;
    double v = z * x;
    double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

static inline double kcos(double x, double y)
{
    double z = x * x;
    double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    double ax = x < 0 ? -x : x;
    if (ax < 0.3) return 1.0 - (0.5 * z - (z * r - x * y));
    double qx;
    if (ax > 0.78125) qx = 0.28125;
    else {
        uint32_t hi = (uint32_t)(d2u(ax) >> 32);
        qx = u2d((uint64_t)(hi - 0x00200000u) << 32);
    }
    double hz = 0.5 * z - qx;
    double a = 1.0 - qx;
    return a - (hz - (z * r - x * y));
}

/* reduce |x| <= ~1e6 to y0+y1 in [-pi/4, pi/4]; returns quadrant */
static inline int rem_pio2(double x, double* y0, double* y1)
{
    double t = x * INVPIO2;
    double fn = (double)(long long)(t + (t < 0 ? -0.5 : 0.5));
    double r = x - fn * PIO2_1;
    double w = fn * PIO2_2;
    double tt = r;
    r = tt - w;
    w = fn * PIO2_2T - ((tt - r) - w);
    *y0 = r - w;
    *y1 = (r - *y0) - w;
    return (int)((long long)fn & 3);
}

double lfo_sin(double x)
{
    if (x != x || x == INFINITY || x == -INFINITY) return NAN;
    double ax = x < 0 ? -x : x;
    if (ax < 0.78539816339744830962) {
        if (ax < 7.450580596923828125e-09) return x;
        return ksin(x, 0.0);
    }
    double y0, y1;
    int n = rem_pio2(x, &y0, &y1);
    switch (n) {
    case 0: return ksin(y0, y1);
    case 1: return kcos(y0, y1);
    case 2: return -ksin(y0, y1);
    default: return -kcos(y0, y1);
    }
}

double lfo_cos(double x)
{
    if (x != x || x == INFINITY || x == -INFINITY) return NAN;
    double ax = x < 0 ? -x : x;
    if (ax < 0.78539816339744830962) {
        if (ax < 7.450580596923828125e-09) return 1.0;
        return kcos(x, 0.0);
    }
    double y0, y1;
    int n = rem_pio2(x, &y0, &y1);
    switch (n) {
    case 0: return kcos(y0, y1);
    case 1: return -ksin(y0, y1);
    case 2: return -kcos(y0, y1);
    default: return ksin(y0, y1);
    }
}

/* ----------------------------------------------------------------- atan */
static const double ATANHI[4] = {
    4.63647609000806093515e-01, 7.85398163397448278999e-01,
    9.82793723247329054082e-01, 1.57079632679489655800e+00 };
static const double ATANLO[4] = {
    2.26987774529616870924e-17, 3.06161699786838301793e-17,
    1.39033110312309984516e-17, 6.12323399573676603587e-17 };
static const double AT[11] = {
    3.33333333333329318027e-01, -1.99999999998764832476e-01,
    1.42857142725034663711e-01, -1.11111104054623557880e-01,
    9.09088713343650656196e-02, -7.69187620504482999495e-02,
    6.66107313738753120669e-02, -5.83357013379057348645e-02,
    4.97687799461593236017e-02, -3.65315727442169155270e-02,
    1.62858201153657823623e-02 };

double lfo_atan(double x)
{
    if (x != x) return x;
    int neg = x < 0 || (x == 0 && (d2u(x) >> 63));
    double ax = neg ? -x : x;
    int id;
    if (ax >= 73786976294838206464.0) { /* 2^66 */
        double z = ATANHI[3] + ATANLO[3];
        return neg ? -z : z;
    }
    if (ax < 0.4375) {
        if (ax < 1.862645149230957e-09) return x; /* 2^-29 */
        id = -1;
    } else if (ax < 1.1875) {
        if (ax < 0.6875) { id = 0; ax = (2.0 * ax - 1.0) / (2.0 + ax); }
        else { id = 1; ax = (ax - 1.0) / (ax + 1.0); }
    } else {
        if (ax < 2.4375) { id = 2; ax = (ax - 1.5) / (1.0 + 1.5 * ax); }
        else { id = 3; ax = -1.0 / ax; }
    }
    double z = ax * ax;
    double w = z * z;
    double s1 = z * (AT[0] + w * (AT[2] + w * (AT[4] + w * (AT[6] + w * (AT[8] + w * AT[10])))));
    double s2 = w * (AT[1] + w * (AT[3] + w * (AT[5] + w * (AT[7] + w * AT[9]))));
    if (id < 0) { double r = ax - ax * (s1 + s2); return neg ? -r : r; }
    z = ATANHI[id] - ((ax * (s1 + s2) - ATANLO[id]) - ax);
    return neg ? -z : z;
}

static const double PI_D = 3.14159265358979311600e+00;
static const double PI_LO = 1.2246467991473531772e-16;
static const double PIO2_HI = 1.57079632679489655800e+00;

double lfo_atan2(double y, double x)
{
    if (x != x || y != y) return x + y;
    int sy = (int)(d2u(y) >> 63), sx = (int)(d2u(x) >> 63);
    if (y == 0.0) {
        if (!sx) return y;                 /* atan2(+-0, +x) = +-0 */
        return sy ? -PI_D : PI_D;          /* atan2(+-0, -x) = +-pi */
    }
    if (x == 0.0) return sy ? -PIO2_HI : PIO2_HI;
    if (x == INFINITY || x == -INFINITY) {
        if (y == INFINITY || y == -INFINITY) {
            double v = sx ? 3.0 * 0.78539816339744827900 : 0.78539816339744827900;
            return sy ? -v : v;
        }
        if (!sx) return sy ? -0.0 : 0.0;
        return sy ? -PI_D : PI_D;
    }
    if (y == INFINITY || y == -INFINITY) return sy ? -PIO2_HI : PIO2_HI;
    double ay = sy ? -y : y, ax = sx ? -x : x;
    double z = lfo_atan(ay / ax);
    if (!sx) return sy ? -z : z;
    z = PI_D - (z - PI_LO);
    return sy ? -z : z;
}

double lfo_asin(double x)
{
    if (x != x) return x;
    if (x > 1.0 || x < -1.0) return NAN;
    return lfo_atan2(x, sqrt((1.0 - x) * (1.0 + x)));
}

/* ----------------------------------------------------------- sinh, pow */
double lfo_sinh_small(double x)
{
    double z = x * x;
    double p = 1.0 / 6227020800.0;                 /* 1/13! */
    p = 1.0 / 39916800.0 + z * p;                  /* 1/11! */
    p = 1.0 / 362880.0 + z * p;
    p = 1.0 / 5040.0 + z * p;
    p = 1.0 / 120.0 + z * p;
    p = 1.0 / 6.0 + z * p;
    return x + x * (z * p);
}

double lfo_pow(double x, double y)
{
    double fy = (double)(long long)y;
    if (fy == y && y >= -64.0 && y <= 64.0) {
        int n = (int)y;
        int neg = n < 0;
        unsigned e = (unsigned)(neg ? -n : n);
        double r = 1.0, b = x;
        while (e) { if (e & 1u) r = r * b; b = b * b; e >>= 1; }
        return neg ? 1.0 / r : r;
    }
    return lfo_exp(y * lfo_log(x));
}

/* --------------------------------------------------- OpenCV fastAtan2 */
/* OpenCV 3.x core/mathfuncs_core: 7th-order odd polynomial on [0,1], degrees. */
float lfo_fast_atan2_deg(float y, float x)
{
    const float p1 = 0.9997878412794807f * (float)(180.0 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180.0 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180.0 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180.0 / 3.14159265358979323846);
    const float eps = (float)2.2204460492503131e-16;
    float ax = x < 0 ? -x : x, ay = y < 0 ? -y : y;
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* ------------------------------------------------------ vector harness */
void lfo_vec_unary(int which, const double* x, double* y, int n)
{
    for (int i = 0; i < n; ++i) {
        switch (which) {
        case 0: y[i] = lfo_exp(x[i]); break;
        case 1: y[i] = lfo_log(x[i]); break;
        case 2: y[i] = lfo_sin(x[i]); break;
        case 3: y[i] = lfo_cos(x[i]); break;
        case 4: y[i] = lfo_atan(x[i]); break;
        case 5: y[i] = lfo_asin(x[i]); break;
        case 6: y[i] = lfo_log10(x[i]); break;
        case 7: y[i] = lfo_sinh_small(x[i]); break;
        default: y[i] = NAN;
        }
    }
}

void lfo_vec_binary(int which, const double* a, const double* b, double* y, int n)
{
    for (int i = 0; i < n; ++i) {
        switch (which) {
        case 0: y[i] = lfo_atan2(a[i], b[i]); break;
        case 1: y[i] = lfo_pow(a[i], b[i]); break;
        default: y[i] = NAN;
        }
    }
}

void lfo_vec_fast_atan2(const float* y, const float* x, float* out, int n)
{
    for (int i = 0; i < n; ++i) out[i] = lfo_fast_atan2_deg(y[i], x[i]);
}
