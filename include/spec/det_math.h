/* det_math.h — deterministic scalar math shared by the HIP kernels and the CPU oracle.
 *
 * Why this exists: the reference (martingoe/physics) calls f32::sin / f32::cos / f32::asin /
 * f32::atan2 (rigid_body.rs:35 via nalgebra Quaternion::exp, fixed_orientation_constraint.rs:17 via
 * Rotation3::euler_angles), which resolve to the host libm. Device libm (OCML) and glibc differ in
 * the last ulp, which a chaotic contact stack amplifies. These functions evaluate in IEEE double
 * with fixed polynomials and fixed operation order (no FMA contraction: build with
 * -ffp-contract=off) and round once to float, so they are bit-identical on gfx950 and x86-64 and
 * agree with a correctly-rounded float libm except when the double result lies within ~1e-16
 * relative of a float rounding boundary. tests/test_det_math.py measures the disagreement with
 * glibc sinf/cosf/asinf/atan2f on millions of samples.
 *
 * Polynomial coefficients: the classic fdlibm (Sun, public domain) kernel_sin / kernel_cos /
 * e_asin / s_atan minimax sets.
 */
#ifndef PHYS_SPEC_DET_MATH_H
#define PHYS_SPEC_DET_MATH_H

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PHYS_HD __host__ __device__ __forceinline__
#define PHYS_UNROLL _Pragma("unroll")
#else
#define PHYS_HD static inline
#define PHYS_UNROLL
#endif

/* plain comparisons: no fminf/fmaxf (their NaN / signed-zero rules differ between libms) */
PHYS_HD float det_minf(float a, float b) { return a < b ? a : b; }
PHYS_HD float det_maxf(float a, float b) { return a > b ? a : b; }
PHYS_HD float det_absf(float a) { return a < 0.0f ? -a : a; }
PHYS_HD double det_absd(double a) { return a < 0.0 ? -a : a; }

/* sin and cos of r, |r| <= pi/4 (+ slack), double precision */
PHYS_HD double det_ksin(double r) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double z = r * r;
    const double p = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    return r + (r * z) * (S1 + z * p);
}
PHYS_HD double det_kcos(double r) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = r * r;
    const double p = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    return (1.0 - 0.5 * z) + z * p;
}

/* reduce x to r in [-pi/4, pi/4], returns quadrant q (0..3). Valid for |x| < ~2^20*pi/2; beyond
 * that the result is still deterministic but no longer accurate (the hot path feeds |w|*dt/2). */
PHYS_HD int det_rem_pio2(double x, double* r) {
    const double INV_PIO2 = 6.36619772367581382433e-01;
    const double PIO2_1 = 1.57079632673412561417e+00;  /* first 33 bits of pi/2 */
    const double PIO2_1T = 6.07710050650619224932e-11; /* pi/2 - PIO2_1 */
    const double t = x * INV_PIO2;
    const long long k = (long long)(t + (t >= 0.0 ? 0.5 : -0.5));
    const double kd = (double)k;
    *r = (x - kd * PIO2_1) - kd * PIO2_1T;
    return (int)(k & 3);
}

PHYS_HD float det_sinf(float xf) {
    double r;
    const int q = det_rem_pio2((double)xf, &r);
    double v;
    switch (q) {
        case 0: v = det_ksin(r); break;
        case 1: v = det_kcos(r); break;
        case 2: v = -det_ksin(r); break;
        default: v = -det_kcos(r); break;
    }
    return (float)v;
}

PHYS_HD float det_cosf(float xf) {
    double r;
    const int q = det_rem_pio2((double)xf, &r);
    double v;
    switch (q) {
        case 0: v = det_kcos(r); break;
        case 1: v = -det_ksin(r); break;
        case 2: v = -det_kcos(r); break;
        default: v = det_ksin(r); break;
    }
    return (float)v;
}

/* double sqrt by the compiler builtin: IEEE correctly rounded on both targets */
PHYS_HD double det_sqrtd(double x) { return __builtin_sqrt(x); }
PHYS_HD float det_sqrtf(float x) { return __builtin_sqrtf(x); }

/* atan on double (fdlibm s_atan structure: breakpoints 7/16, 11/16, 19/16, 39/16) */
PHYS_HD double det_atand(double x) {
    const double atanhi0 = 4.63647609000806093515e-01, atanhi1 = 7.85398163397448278999e-01,
                 atanhi2 = 9.82793723247329054082e-01, atanhi3 = 1.57079632679489655800e+00;
    const double atanlo0 = 2.26987774529616870924e-17, atanlo1 = 3.06161699786838301793e-17,
                 atanlo2 = 1.39033110312309984516e-17, atanlo3 = 6.12323399573676603587e-17;
    const double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01,
                 aT2 = 1.42857142725034663711e-01, aT3 = -1.11111104054623557880e-01,
                 aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
                 aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02,
                 aT8 = 4.97687799461593236017e-02, aT9 = -3.65315727442169155270e-02,
                 aT10 = 1.62858201153657823623e-02;
    const int neg = x < 0.0;
    double ax = neg ? -x : x;
    int id;
    double hi = 0.0, lo = 0.0;
    if (ax != ax) return x; /* NaN */
    if (ax >= 7.3786976294838206464e19) { /* 2^66: atan = pi/2 */
        const double v = atanhi3 + atanlo3;
        return neg ? -v : v;
    }
    if (ax < 0.4375) {
        id = -1;
    } else if (ax < 1.1875) {
        if (ax < 0.6875) { id = 0; ax = (2.0 * ax - 1.0) / (2.0 + ax); hi = atanhi0; lo = atanlo0; }
        else             { id = 1; ax = (ax - 1.0) / (ax + 1.0);       hi = atanhi1; lo = atanlo1; }
    } else {
        if (ax < 2.4375) { id = 2; ax = (ax - 1.5) / (1.0 + 1.5 * ax); hi = atanhi2; lo = atanlo2; }
        else             { id = 3; ax = -1.0 / ax;                     hi = atanhi3; lo = atanlo3; }
    }
    const double z = ax * ax;
    const double w = z * z;
    const double s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const double s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    double v;
    if (id < 0) v = ax - ax * (s1 + s2);
    else        v = hi - ((ax * (s1 + s2) - lo) - ax);
    return neg ? -v : v;
}

/* atan2f(y, x): IEEE special cases reduced to what finite inputs need, computed in double */
PHYS_HD float det_atan2f(float yf, float xf) {
    const double PI = 3.14159265358979311600e+00, PI_LO = 1.2246467991473531772e-16;
    const double y = (double)yf, x = (double)xf;
    if (x != x || y != y) return xf + yf;
    if (y == 0.0) {
        /* sign of zero: use bit test via division-free comparison on 1/x is not available for
         * x == 0; treat -0 as +0 except through the x < 0 branch (finite hot-path inputs). */
        if (x < 0.0) return (float)((1.0 / y) < 0.0 ? -PI : PI);
        return yf; /* +-0 */
    }
    if (x == 0.0) return (float)(y < 0.0 ? -0.5 * PI : 0.5 * PI);
    const double z = det_atand(det_absd(y / x));
    double v;
    if (x > 0.0) v = z;
    else         v = PI - (z - PI_LO);
    return (float)(y < 0.0 ? -v : v);
}

/* asinf via atan2 identity in double: asin(x) = atan2(x, sqrt((1-x)(1+x))) */
PHYS_HD float det_asinf(float xf) {
    const double x = (double)xf;
    const double PIO2 = 1.57079632679489655800e+00;
    if (x != x) return xf;
    if (x >= 1.0) return (float)(x > 1.0 ? (x - x) / (x - x) : PIO2);
    if (x <= -1.0) return (float)(x < -1.0 ? (x - x) / (x - x) : -PIO2);
    const double c = det_sqrtd((1.0 - x) * (1.0 + x));
    const double ax = det_absd(x);
    double v;
    if (ax <= c) v = det_atand(ax / c);
    else         v = PIO2 - (det_atand(c / ax) - 6.12323399573676603587e-17);
    return (float)(x < 0.0 ? -v : v);
}

#endif /* PHYS_SPEC_DET_MATH_H */
