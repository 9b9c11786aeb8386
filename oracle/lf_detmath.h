/*
 * ORACLE (test infrastructure only -- never linked into the product library).
 *
 * Deterministic double-precision elementary functions.
 *
 * Why this exists: the reference path calls libm / OpenCV transcendental
 * functions (cos, sin, atan2, log, exp, pow, sinh, asin) in double precision
 * (see e.g. /root/reference/src/line_descriptor/src/binary_descriptor_custom.cpp:1130-1131,
 *  /root/reference/src/line_sanity/src/line_sanity_node.py:95, and OpenCV's
 *  lsd.cpp which the reference reaches through
 *  /root/reference/src/line_detector/include/line_detector/line_detector_lsd.py:64-67).
 * glibc and the GPU device library do not round these identically, so a
 * bit-exact CPU<->GPU comparison needs both sides to evaluate the SAME
 * sequence of IEEE-754 +,-,*,/,sqrt operations.  These functions are that
 * sequence (fdlibm-style argument reduction + polynomial kernels, no FMA;
 * compile with -ffp-contract=off).  The HIP product carries its own copy of the
 * same recipe in lane_slam_amd/csrc/detmath.h; tests/test_detmath.py checks
 * both against libm to a few ULP and (on the GPU) against each other bitwise.
 */
#ifndef LF_ORACLE_DETMATH_H
#define LF_ORACLE_DETMATH_H

#ifdef __cplusplus
extern "C" {
#endif

double lfo_exp(double x);
double lfo_log(double x);
double lfo_log10(double x);
double lfo_sin(double x);
double lfo_cos(double x);
double lfo_atan(double x);
double lfo_atan2(double y, double x);
double lfo_asin(double x);
double lfo_sinh_small(double x);      /* |x| <= 0.5, odd Taylor series */
double lfo_pow(double x, double y);   /* x > 0; exact-ish repeated squaring for small integer y */
float  lfo_fast_atan2_deg(float y, float x); /* OpenCV fastAtan2 (3.x polynomial form), degrees */

/* vector entry points for the ctypes test harness */
void lfo_vec_unary(int which, const double* x, double* y, int n);
void lfo_vec_binary(int which, const double* a, const double* b, double* y, int n);
void lfo_vec_fast_atan2(const float* y, const float* x, float* out, int n);

#ifdef __cplusplus
}
#endif
#endif
