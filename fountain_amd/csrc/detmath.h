/*
 * detmath.h -- deterministic elementary functions for the path-tracing core.
 *
 * The reference calls f32::sin/cos/acos/atan2/ln/tan/atan (-> the platform libm) on its hot path
 * (src/sampling.rs:16-19,39-43, src/shapes/sphere.rs:120-145, src/light/infinite.rs:106-152,
 * src/reflection/microfacet.rs:40-45,162-186, src/geometry/mod.rs:23-34).  glibc on the host and
 * OCML on the GPU do not round these identically, and a 1-ulp difference flips Russian-roulette /
 * lobe-choice / hit-miss branches, which moves a pixel by O(1/spp).  To make GPU results comparable
 * bit for bit with a CPU run we evaluate every transcendental here, in IEEE binary64 with a fixed
 * operation order (no FMA contraction: build with -ffp-contract=off), and round once to binary32.
 * The same source is compiled by hipcc for gfx950 and by g++ for the host; binary64 +,-,*,/ and sqrt
 * are correctly rounded on both, so the results are identical by construction.  Accuracy is a few
 * 1e-13 relative before the final rounding, i.e. the binary32 result is the correctly rounded one
 * except for roughly one argument in 1e5 (tests/test_detmath.py measures this against libm).
 *
 * Used by: the HIP kernels and the host-side constructors of this library.  The CPU oracle (oracle/)
 * follows the reference and calls libm; its optional "det" build includes this header so that a
 * bit-exact GPU-vs-CPU comparison is possible (tests compare against both builds).
 */
#ifndef FTN_DETMATH_H
#define FTN_DETMATH_H

#if defined(__HIPCC__) || defined(__HIP_DEVICE_COMPILE__)
#include <hip/hip_runtime.h>
#define FTN_HD __host__ __device__ inline
#else
#define FTN_HD inline
#endif

#include <math.h>
#include <stdint.h>
#include <string.h>

namespace ftn_det {

/* ---- bit casts that work on host and device */
FTN_HD uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
FTN_HD float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
FTN_HD uint64_t d2u(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
FTN_HD double u2d(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }

/* ---- sin / cos kernels on [-pi/4, pi/4] (Taylor, Horner in r^2) */
FTN_HD double ksin(double r) {
    const double z = r * r;
    double p = -7.6471637318198164759e-13;       /* -1/15! */
    p = p * z + 1.6059043836821614599e-10;       /*  1/13! */
    p = p * z - 2.5052108385441718775e-8;        /* -1/11! */
    p = p * z + 2.7557319223985890653e-6;        /*  1/9!  */
    p = p * z - 1.9841269841269841270e-4;        /* -1/7!  */
    p = p * z + 8.3333333333333333333e-3;        /*  1/5!  */
    p = p * z - 1.6666666666666666667e-1;        /* -1/3!  */
    return r + r * (z * p);
}
FTN_HD double kcos(double r) {
    const double z = r * r;
    double p = 4.7794773323873852974e-14;        /*  1/16! */
    p = p * z - 1.1470745597729724714e-11;       /* -1/14! */
    p = p * z + 2.0876756987868098979e-9;        /*  1/12! */
    p = p * z - 2.7557319223985890653e-7;        /* -1/10! */
    p = p * z + 2.4801587301587301587e-5;        /*  1/8!  */
    p = p * z - 1.3888888888888888889e-3;        /* -1/6!  */
    p = p * z + 4.1666666666666666667e-2;        /*  1/4!  */
    p = p * z - 0.5;
    return 1.0 + z * p;
}

/* quadrant reduction: x = k*(pi/2) + r, |r| <= pi/4 (two-constant Cody-Waite; |x| < 1e6) */
FTN_HD double reduce_pio2(double x, int* quadrant) {
    const double two_over_pi = 6.36619772367581382433e-01;
    const double pio2_hi = 1.57079632673412561417e+00;   /* first 33 bits of pi/2 */
    const double pio2_lo = 6.07710050650619224932e-11;   /* pi/2 - pio2_hi        */
    const double kf = floor(x * two_over_pi + 0.5);
    const double r = (x - kf * pio2_hi) - kf * pio2_lo;
    *quadrant = (int)((long long)kf & 3);
    return r;
}

FTN_HD float sinf_det(float xf) {
    if (!(xf == xf) || xf - xf != 0.0f) return xf - xf;   /* NaN / inf -> NaN */
    int q;
    const double r = reduce_pio2((double)xf, &q);
    double v;
    switch (q) {
        case 0: v = ksin(r); break;
        case 1: v = kcos(r); break;
        case 2: v = -ksin(r); break;
        default: v = -kcos(r); break;
    }
    return (float)v;
}

FTN_HD float cosf_det(float xf) {
    if (!(xf == xf) || xf - xf != 0.0f) return xf - xf;
    int q;
    const double r = reduce_pio2((double)xf, &q);
    double v;
    switch (q) {
        case 0: v = kcos(r); break;
        case 1: v = -ksin(r); break;
        case 2: v = -kcos(r); break;
        default: v = ksin(r); break;
    }
    return (float)v;
}

FTN_HD float tanf_det(float xf) {
    if (!(xf == xf) || xf - xf != 0.0f) return xf - xf;
    int q;
    const double r = reduce_pio2((double)xf, &q);
    const double s = ksin(r), c = kcos(r);
    return (float)((q & 1) ? (-c / s) : (s / c));
}

/* ---- atan on [0, +inf): argument reduction with 3 break points, odd Taylor on |t| <= ~0.2 */
FTN_HD double katan_pos(double x) {
    /* atan(x) = atan(c) + atan((x - c) / (1 + x c)) */
    double base, t;
    if (x > 5.0273394921258481045) {            /* tan(7pi/16): pi/2 - atan(1/x), 1/x < tan(pi/16) */
        base = 1.57079632679489661923;
        t = -1.0 / x;
    } else if (x > 1.4966057626654890176) {     /* tan(5pi/16) .. tan(7pi/16): c = tan(3pi/8) */
        const double c = 2.4142135623730950488;
        base = 1.17809724509617246442;          /* 3pi/8 */
        t = (x - c) / (1.0 + x * c);
    } else if (x > 0.66817863791929891999) {    /* tan(3pi/16) .. tan(5pi/16): c = 1 */
        base = 0.78539816339744830962;          /* pi/4 */
        t = (x - 1.0) / (1.0 + x);
    } else if (x > 0.19891236737965800691) {    /* tan(pi/16) .. tan(3pi/16): c = tan(pi/8) */
        const double c = 0.41421356237309504880;
        base = 0.39269908169872415481;          /* pi/8 */
        t = (x - c) / (1.0 + x * c);
    } else {
        base = 0.0;
        t = x;
    }
    const double z = t * t;
    double p = 1.0 / 19.0;
    p = -1.0 / 17.0 + z * p;
    p = 1.0 / 15.0 + z * p;
    p = -1.0 / 13.0 + z * p;
    p = 1.0 / 11.0 + z * p;
    p = -1.0 / 9.0 + z * p;
    p = 1.0 / 7.0 + z * p;
    p = -1.0 / 5.0 + z * p;
    p = 1.0 / 3.0 + z * p;
    return base + (t - t * (z * p));
}

FTN_HD float atanf_det(float xf) {
    if (!(xf == xf)) return xf;
    const double x = (double)xf;
    const double a = katan_pos(x < 0.0 ? -x : x);
    return (float)(x < 0.0 ? -a : a);
}

/* atan2 with the C / Rust special-case conventions for zeros; inf/NaN inputs are not produced by
 * the call sites and fall through to a finite-arithmetic answer. */
FTN_HD float atan2f_det(float yf, float xf) {
    if (!(yf == yf) || !(xf == xf)) return yf + xf;
    const double y = (double)yf, x = (double)xf;
    const double pi = 3.14159265358979323846;
    const bool yneg = (f2u(yf) >> 31) != 0;
    const bool xneg = (f2u(xf) >> 31) != 0;
    if (y == 0.0) {
        if (x == 0.0) {                      /* atan2(+-0, +-0) */
            const double r = xneg ? pi : 0.0;
            return (float)(yneg ? -r : r);
        }
        if (x > 0.0) return yneg ? -0.0f : 0.0f;
        return (float)(yneg ? -pi : pi);
    }
    if (x == 0.0) return (float)(yneg ? -0.5 * pi : 0.5 * pi);
    const double ay = yneg ? -y : y, ax = xneg ? -x : x;
    double a = katan_pos(ay / ax);
    if (xneg) a = pi - a;
    return (float)(yneg ? -a : a);
}

FTN_HD float acosf_det(float xf) {
    if (!(xf == xf)) return xf;
    const double x = (double)xf;
    if (x > 1.0 || x < -1.0) return u2f(0x7fc00000u);
    const double pi = 3.14159265358979323846;
    /* acos(x) = atan2(sqrt((1-x)(1+x)), x); (1-x) and (1+x) are exact in binary64 for a float x */
    const double s = sqrt((1.0 - x) * (1.0 + x));
    if (x == 0.0) return (float)(0.5 * pi);
    if (x > 0.0) return (float)katan_pos(s / x);
    return (float)(pi - katan_pos(s / -x));
}

/* ---- natural log: x = m * 2^e, m in [sqrt(1/2), sqrt(2)); log m = 2 atanh((m-1)/(m+1)) */
FTN_HD double klog_parts(double x, int* e_out) {
    uint64_t bits = d2u(x);                    /* floats (incl. subnormals) are normal doubles */
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    bits = (bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m = u2d(bits);                      /* [1, 2) */
    if (m > 1.41421356237309504880) { m = m * 0.5; e += 1; }
    const double s = (m - 1.0) / (m + 1.0);
    const double z = s * s;
    double p = 1.0 / 19.0;
    p = 1.0 / 17.0 + z * p;
    p = 1.0 / 15.0 + z * p;
    p = 1.0 / 13.0 + z * p;
    p = 1.0 / 11.0 + z * p;
    p = 1.0 / 9.0 + z * p;
    p = 1.0 / 7.0 + z * p;
    p = 1.0 / 5.0 + z * p;
    p = 1.0 / 3.0 + z * p;
    *e_out = e;
    return 2.0 * (s + s * (z * p));            /* ln(m) */
}

FTN_HD float logf_det(float xf) {
    if (!(xf == xf)) return xf;
    if (xf < 0.0f) return u2f(0x7fc00000u);
    if (xf == 0.0f) return u2f(0xff800000u);
    if (xf - xf != 0.0f) return xf;            /* +inf */
    int e;
    const double lm = klog_parts((double)xf, &e);
    const double ln2_hi = 6.93147180369123816490e-01;
    const double ln2_lo = 1.90821492927058770002e-10;
    const double ef = (double)e;
    return (float)((ef * ln2_hi + lm) + ef * ln2_lo);
}

/* log2 is only evaluated on the host (MIP level selection, src/mipmap.rs:247); exact on powers of two */
FTN_HD float log2f_det(float xf) {
    if (!(xf == xf)) return xf;
    if (xf < 0.0f) return u2f(0x7fc00000u);
    if (xf == 0.0f) return u2f(0xff800000u);
    if (xf - xf != 0.0f) return xf;
    int e;
    const double lm = klog_parts((double)xf, &e);
    return (float)((double)e + lm * 1.44269504088896340736);
}

}  // namespace ftn_det
#endif
