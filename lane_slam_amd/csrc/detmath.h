// lanefront -- deterministic double-precision elementary functions for device code.
//
// The reference path evaluates cos/sin/atan2/log/exp/pow/asin through libm and
// OpenCV (e.g. /root/reference/src/line_descriptor/src/binary_descriptor_custom.cpp:1130-1131,
// /root/reference/src/line_sanity/src/line_sanity_node.py:95, OpenCV lsd.cpp behind
// /root/reference/src/line_detector/include/line_detector/line_detector_lsd.py:64-67).
// To make "segment endpoints bit-exact" a testable contract, every transcendental on
// the GPU path is evaluated by the fixed sequence of IEEE-754 double +,-,*,/ below
// (fdlibm-style reduction + polynomial kernels, <= 1-2 ULP of libm).  No FMA: the
// translation unit must be compiled with -ffp-contract=off.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define LF_HD __host__ __device__ __forceinline__
#else
#define LF_HD inline
#endif

namespace lf {
namespace dm {

LF_HD uint64_t d2u(double x) { return __builtin_bit_cast(uint64_t, x); }
LF_HD double u2d(uint64_t u) { return __builtin_bit_cast(double, u); }
LF_HD double pow2i(int k) { return u2d((uint64_t)(k + 1023) << 52); }
LF_HD double inf() { return u2d(0x7ff0000000000000ull); }
LF_HD double qnan() { return u2d(0x7ff8000000000000ull); }

LF_HD double dexp(double x)
{
    const double LN2HI = 6.93147180369123816490e-01, LN2LO = 1.90821492927058770002e-10;
    const double INVLN2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
                 P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
                 P5 = 4.13813679705723846039e-08;
    if (x != x) return x;
    if (x > 709.782712893383973096) return inf();
    if (x < -745.13321910194110842) return 0.0;
    double t = x * INVLN2 + (x < 0 ? -0.5 : 0.5);
    int k = (int)t;
    double fk = (double)k;
    double hi = x - fk * LN2HI;
    double lo = fk * LN2LO;
    double r = hi - lo;
    double z = r * r;
    double c = r - z * (P1 + z * (P2 + z * (P3 + z * (P4 + z * P5))));
    double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    if (k >= -1021 && k <= 1023) return y * pow2i(k);
    if (k > 1023) return (y * pow2i(1023)) * pow2i(k - 1023);
    return (y * pow2i(k + 1000)) * pow2i(-1000);
}

LF_HD double dlog(double x)
{
    const double LN2HI = 6.93147180369123816490e-01, LN2LO = 1.90821492927058770002e-10;
    const double G1 = 6.666666666666735130e-01, G2 = 3.999999999940941908e-01,
                 G3 = 2.857142874366239149e-01, G4 = 2.222219843214978396e-01,
                 G5 = 1.818357216161805012e-01, G6 = 1.531383769920937332e-01,
                 G7 = 1.479819860511658591e-01;
    if (x != x) return x;
    if (x < 0.0) return qnan();
    if (x == 0.0) return -inf();
    if (x == inf()) return x;
    int k = 0;
    uint64_t u = d2u(x);
    if ((u >> 52) == 0) {
        x = x * 18014398509481984.0;
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
    double t1 = w * (G2 + w * (G4 + w * G6));
    double t2 = z * (G1 + w * (G3 + w * (G5 + w * G7)));
    double R = t2 + t1;
    double hfsq = 0.5 * f * f;
    if (k == 0) return f - (hfsq - s * (hfsq + R));
    return dk * LN2HI - ((hfsq - (s * (hfsq + R) + dk * LN2LO)) - f);
}

LF_HD double dlog10(double x) { return dlog(x) / 2.30258509299404568402e+00; }

LF_HD double ksin(double x, double y)
{
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = x * x;
    double v = z * x;
    double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

LF_HD double kcos(double x, double y)
{
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
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

LF_HD int rem_pio2(double x, double& y0, double& y1)
{
    const double INVPIO2 = 6.36619772367581382433e-01, PIO2_1 = 1.57079632673412561417e+00,
                 PIO2_2 = 6.07710050630396597660e-11, PIO2_2T = 2.02226624879595063154e-21;
    double t = x * INVPIO2;
    double fn = (double)(long long)(t + (t < 0 ? -0.5 : 0.5));
    double r = x - fn * PIO2_1;
    double w = fn * PIO2_2;
    double tt = r;
    r = tt - w;
    w = fn * PIO2_2T - ((tt - r) - w);
    y0 = r - w;
    y1 = (r - y0) - w;
    return (int)((long long)fn & 3);
}

// sin and cos together (the callers always want both)
LF_HD void dsincos(double x, double& s, double& c)
{
    if (x != x || x == inf() || x == -inf()) { s = qnan(); c = qnan(); return; }
    double ax = x < 0 ? -x : x;
    if (ax < 0.78539816339744830962) {
        if (ax < 7.450580596923828125e-09) { s = x; c = 1.0; return; }
        s = ksin(x, 0.0);
        c = kcos(x, 0.0);
        return;
    }
    double y0, y1;
    int n = rem_pio2(x, y0, y1);
    double ks = ksin(y0, y1), kc = kcos(y0, y1);
    switch (n) {
    case 0: s = ks; c = kc; break;
    case 1: s = kc; c = -ks; break;
    case 2: s = -ks; c = -kc; break;
    default: s = -kc; c = ks; break;
    }
}
LF_HD double dsin(double x) { double s, c; dsincos(x, s, c); return s; }
LF_HD double dcos(double x) { double s, c; dsincos(x, s, c); return c; }

LF_HD double datan(double x)
{
    const double HI0 = 4.63647609000806093515e-01, HI1 = 7.85398163397448278999e-01,
                 HI2 = 9.82793723247329054082e-01, HI3 = 1.57079632679489655800e+00;
    const double LO0 = 2.26987774529616870924e-17, LO1 = 3.06161699786838301793e-17,
                 LO2 = 1.39033110312309984516e-17, LO3 = 6.12323399573676603587e-17;
    const double A0 = 3.33333333333329318027e-01, A1 = -1.99999999998764832476e-01,
                 A2 = 1.42857142725034663711e-01, A3 = -1.11111104054623557880e-01,
                 A4 = 9.09088713343650656196e-02, A5 = -7.69187620504482999495e-02,
                 A6 = 6.66107313738753120669e-02, A7 = -5.83357013379057348645e-02,
                 A8 = 4.97687799461593236017e-02, A9 = -3.65315727442169155270e-02,
                 A10 = 1.62858201153657823623e-02;
    if (x != x) return x;
    bool neg = (d2u(x) >> 63) != 0;
    double ax = neg ? -x : x;
    int id;
    double hi = 0.0, lo = 0.0;
    if (ax >= 73786976294838206464.0) {
        double z = HI3 + LO3;
        return neg ? -z : z;
    }
    if (ax < 0.4375) {
        if (ax < 1.862645149230957e-09) return x;
        id = -1;
    } else if (ax < 1.1875) {
        if (ax < 0.6875) { id = 0; hi = HI0; lo = LO0; ax = (2.0 * ax - 1.0) / (2.0 + ax); }
        else { id = 1; hi = HI1; lo = LO1; ax = (ax - 1.0) / (ax + 1.0); }
    } else {
        if (ax < 2.4375) { id = 2; hi = HI2; lo = LO2; ax = (ax - 1.5) / (1.0 + 1.5 * ax); }
        else { id = 3; hi = HI3; lo = LO3; ax = -1.0 / ax; }
    }
    double z = ax * ax;
    double w = z * z;
    double s1 = z * (A0 + w * (A2 + w * (A4 + w * (A6 + w * (A8 + w * A10)))));
    double s2 = w * (A1 + w * (A3 + w * (A5 + w * (A7 + w * A9))));
    if (id < 0) { double r = ax - ax * (s1 + s2); return neg ? -r : r; }
    z = hi - ((ax * (s1 + s2) - lo) - ax);
    return neg ? -z : z;
}

LF_HD double datan2(double y, double x)
{
    const double PI_D = 3.14159265358979311600e+00, PI_LO = 1.2246467991473531772e-16,
                 PIO2_HI = 1.57079632679489655800e+00, PIO4 = 0.78539816339744827900;
    if (x != x || y != y) return x + y;
    bool sy = (d2u(y) >> 63) != 0, sx = (d2u(x) >> 63) != 0;
    if (y == 0.0) {
        if (!sx) return y;
        return sy ? -PI_D : PI_D;
    }
    if (x == 0.0) return sy ? -PIO2_HI : PIO2_HI;
    if (x == inf() || x == -inf()) {
        if (y == inf() || y == -inf()) {
            double v = sx ? 3.0 * PIO4 : PIO4;
            return sy ? -v : v;
        }
        if (!sx) return sy ? -0.0 : 0.0;
        return sy ? -PI_D : PI_D;
    }
    if (y == inf() || y == -inf()) return sy ? -PIO2_HI : PIO2_HI;
    double ay = sy ? -y : y, ax = sx ? -x : x;
    double z = datan(ay / ax);
    if (!sx) return sy ? -z : z;
    z = PI_D - (z - PI_LO);
    return sy ? -z : z;
}

LF_HD double dsqrt(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __dsqrt_rn(x);
#else
    return __builtin_sqrt(x);
#endif
}

// Correctly rounded float sqrt / divide through double (53 >= 2*24+2 bits makes the second
// rounding innocuous).  HIP's __fsqrt_rn / __fdiv_rn intrinsics lower to approximate
// native instructions on gfx950 and are NOT usable for a bit-exact contract.
LF_HD float fsqrt(float x) { return (float)dsqrt((double)x); }
LF_HD float fdiv(float a, float b) { return (float)((double)a / (double)b); }

LF_HD double dasin(double x)
{
    if (x != x) return x;
    if (x > 1.0 || x < -1.0) return qnan();
    return datan2(x, dsqrt((1.0 - x) * (1.0 + x)));
}

LF_HD double dsinh_small(double x)
{
    double z = x * x;
    double p = 1.0 / 6227020800.0;
    p = 1.0 / 39916800.0 + z * p;
    p = 1.0 / 362880.0 + z * p;
    p = 1.0 / 5040.0 + z * p;
    p = 1.0 / 120.0 + z * p;
    p = 1.0 / 6.0 + z * p;
    return x + x * (z * p);
}

LF_HD double dpow(double x, double y)
{
    double fy = (double)(long long)y;
    if (fy == y && y >= -64.0 && y <= 64.0) {
        int n = (int)y;
        bool neg = n < 0;
        unsigned e = (unsigned)(neg ? -n : n);
        double r = 1.0, b = x;
        while (e) { if (e & 1u) r = r * b; b = b * b; e >>= 1; }
        return neg ? 1.0 / r : r;
    }
    return dexp(y * dlog(x));
}

// OpenCV 3.x fastAtan2: 7th-order odd polynomial, result in degrees [0,360).
LF_HD float fast_atan2_deg(float y, float x)
{
    const float s = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * s, p3 = -0.3258083974640975f * s,
                p5 = 0.1555786518463281f * s, p7 = -0.04432655554792128f * s;
    const float eps = (float)2.2204460492503131e-16;
    float ax = x < 0 ? -x : x, ay = y < 0 ? -y : y;
    // OpenCV divides the smaller magnitude by the larger (+ eps) in either branch: one division, selected
    // operands -- identical values, half the divide sequences in the region-growing accept chain
    const bool steep = !(ax >= ay);
    const float num = steep ? ax : ay, den = (steep ? ay : ax) + eps;
    const float c = num / den;
    const float c2 = c * c;
    float a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    if (steep) a = 90.f - a;
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// round-half-to-even of a float / double to int (OpenCV cvRound)
LF_HD int round_half_even(double v)
{
    double f = (double)(long long)v;            // trunc
    double d = v - f;
    long long i = (long long)f;
    if (d > 0.5 || (d == 0.5 && (i & 1))) i += 1;
    else if (d < -0.5 || (d == -0.5 && (i & 1))) i -= 1;
    return (int)i;
}

LF_HD int ifloor(double v)
{
    int i = (int)v;
    return i - (v < (double)i ? 1 : 0);
}

LF_HD int iceil(double v)
{
    int i = (int)v;
    return i + (v > (double)i ? 1 : 0);
}

// C round(): half away from zero (used by the LBD sampler, binary_descriptor_custom.cpp:1155)
LF_HD double round_half_away(double v)
{
    double f = (double)(long long)v;
    double d = v - f;
    if (d >= 0.5) return f + 1.0;
    if (d <= -0.5) return f - 1.0;
    return f;
}

// the same for a float argument, in float arithmetic: trunc and the remainder are exact for every finite
// float (|v| >= 2^23 has no fraction), so this equals round_half_away((double)v) without the f64 -> i64 round trip
LF_HD float round_half_away_f(float v)
{
    const float f = __builtin_truncf(v);
    const float d = v - f;
    if (d >= 0.5f) return f + 1.0f;
    if (d <= -0.5f) return f - 1.0f;
    return f;
}

}  // namespace dm
}  // namespace lf
