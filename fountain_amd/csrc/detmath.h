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
/* Kept inline everywhere: out-of-line calls were measured slower on gfx950 (calling-convention spills) than the code growth
 * they avoid. */
#define FTN_HD_NOINLINE FTN_HD

#include <math.h>
#include <stdint.h>
#include <string.h>

namespace ftn_det {

/* ---- bit casts that work on host and device */
FTN_HD uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
FTN_HD float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
FTN_HD uint64_t d2u(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
FTN_HD double u2d(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }

/* a*b + c as two correctly rounded binary64 operations (NOT fused: -ffp-contract=off).  Measured on MI355X: v_fma_f64 makes
 * these polynomials slower than v_mul_f64 + v_add_f64 (k_wf_shade 17.3 ms vs 15.2 ms per step), so no explicit fma here. */
FTN_HD double fma_(double a, double b, double c) { return a * b + c; }

/* ---- sin / cos kernels on [-pi/4, pi/4] (Taylor, Horner in r^2) */
FTN_HD double ksin(double r) {
    const double z = r * r;
    double p = -7.6471637318198164759e-13;       /* -1/15! */
    p = fma_(p, z, 1.6059043836821614599e-10);   /*  1/13! */
    p = fma_(p, z, -2.5052108385441718775e-8);   /* -1/11! */
    p = fma_(p, z, 2.7557319223985890653e-6);    /*  1/9!  */
    p = fma_(p, z, -1.9841269841269841270e-4);   /* -1/7!  */
    p = fma_(p, z, 8.3333333333333333333e-3);    /*  1/5!  */
    p = fma_(p, z, -1.6666666666666666667e-1);   /* -1/3!  */
    return fma_(r * z, p, r);
}
FTN_HD double kcos(double r) {
    const double z = r * r;
    double p = 4.7794773323873852974e-14;        /*  1/16! */
    p = fma_(p, z, -1.1470745597729724714e-11);  /* -1/14! */
    p = fma_(p, z, 2.0876756987868098979e-9);    /*  1/12! */
    p = fma_(p, z, -2.7557319223985890653e-7);   /* -1/10! */
    p = fma_(p, z, 2.4801587301587301587e-5);    /*  1/8!  */
    p = fma_(p, z, -1.3888888888888888889e-3);   /* -1/6!  */
    p = fma_(p, z, 4.1666666666666666667e-2);    /*  1/4!  */
    p = fma_(p, z, -0.5);
    return fma_(z, p, 1.0);
}

/* quadrant reduction: x = k*(pi/2) + r, |r| <= pi/4 (two-constant Cody-Waite; |x| < 1e6) */
FTN_HD double reduce_pio2(double x, int* quadrant) {
    const double two_over_pi = 6.36619772367581382433e-01;
    const double pio2_hi = 1.57079632673412561417e+00;   /* first 33 bits of pi/2 */
    const double pio2_lo = 6.07710050650619224932e-11;   /* pi/2 - pio2_hi        */
    const double kf = floor(fma_(x, two_over_pi, 0.5));
    const double r = fma_(-kf, pio2_lo, fma_(-kf, pio2_hi, x));
    *quadrant = (int)((long long)kf & 3);
    return r;
}

/* sin and cos of the same argument with one reduction; each equals the value of the single-function call */
FTN_HD_NOINLINE void sincosf_det(float xf, float* sn, float* cs) {
    if (!(xf == xf) || xf - xf != 0.0f) { *sn = xf - xf; *cs = xf - xf; return; }   /* NaN / inf -> NaN */
    int q;
    const double r = reduce_pio2((double)xf, &q);
    const double s = ksin(r), c = kcos(r);
    double vs, vc;
    switch (q) {
        case 0: vs = s; vc = c; break;
        case 1: vs = c; vc = -s; break;
        case 2: vs = -s; vc = -c; break;
        default: vs = -c; vc = s; break;
    }
    *sn = (float)vs; *cs = (float)vc;
}
FTN_HD float sinf_det(float xf) { float s, c; sincosf_det(xf, &s, &c); return s; }
FTN_HD float cosf_det(float xf) { float s, c; sincosf_det(xf, &s, &c); return c; }

FTN_HD_NOINLINE float tanf_det(float xf) {
    if (!(xf == xf) || xf - xf != 0.0f) return xf - xf;
    int q;
    const double r = reduce_pio2((double)xf, &q);
    const double s = ksin(r), c = kcos(r);
    return (float)((q & 1) ? (-c / s) : (s / c));
}

/* ---- atan on [0, 1]: atan(t) = t * H(t^2), H = degree-16 Chebyshev fit (tools/gen_detmath_coeffs.py, max error 7.5e-15) */
FTN_HD double katan01(double t) {
    const double u = t * t;
    double p = 7.0631426358880456444e-5;
    p = fma_(p, u, -6.7741053996450764411e-4);
    p = fma_(p, u, 3.0659957028108008166e-3);
    p = fma_(p, u, -8.7924355422176355454e-3);
    p = fma_(p, u, 1.8198570485395785522e-2);
    p = fma_(p, u, -2.9589012194344371458e-2);
    p = fma_(p, u, 4.0517789771250661081e-2);
    p = fma_(p, u, -4.9779966759550473653e-2);
    p = fma_(p, u, 5.7941923168818450958e-2);
    p = fma_(p, u, -6.64614579022125362e-2);
    p = fma_(p, u, 7.6888116624012939868e-2);
    p = fma_(p, u, -9.0904896477990322735e-2);
    p = fma_(p, u, 1.1111077584971771991e-1);
    p = fma_(p, u, -1.4285712646050400491e-1);
    p = fma_(p, u, 1.9999999957481824977e-1);
    p = fma_(p, u, -3.3333333332893979705e-1);
    p = fma_(p, u, 9.9999999999999241091e-1);
    return t * p;
}
FTN_HD double katan_pos(double x) {     /* x >= 0 */
    const double pio2 = 1.57079632679489661923;
    return x <= 1.0 ? katan01(x) : pio2 - katan01(1.0 / x);
}

FTN_HD_NOINLINE float atanf_det(float xf) {
    if (!(xf == xf)) return xf;
    const double x = (double)xf;
    const double a = katan_pos(x < 0.0 ? -x : x);
    return (float)(x < 0.0 ? -a : a);
}

/* atan2 with the C / Rust special-case conventions for zeros; one binary64 division. inf/NaN inputs are not produced by
 * the call sites and fall through to a finite-arithmetic answer. */
FTN_HD_NOINLINE float atan2f_det(float yf, float xf) {
    if (!(yf == yf) || !(xf == xf)) return yf + xf;
    const double y = (double)yf, x = (double)xf;
    const double pi = 3.14159265358979323846, pio2 = 1.57079632679489661923;
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
    if (x == 0.0) return (float)(yneg ? -pio2 : pio2);
    const double ay = yneg ? -y : y, ax = xneg ? -x : x;
    const bool swap = ay > ax;
    double a = katan01(swap ? ax / ay : ay / ax);
    if (swap) a = pio2 - a;
    if (xneg) a = pi - a;
    return (float)(yneg ? -a : a);
}

/* ---- asin on [0, 0.5]: asin(s) = s + s*u*G(u), u = s^2, G = degree-10 Chebyshev fit (max error 3.7e-15); no division */
FTN_HD double kasin_half(double s) {
    const double u = s * s;
    double p = 2.7871289137110144687e-2;
    p = fma_(p, u, -6.8220439806712631357e-3);
    p = fma_(p, u, 1.5445133336819307521e-2);
    p = fma_(p, u, 1.0289641123624901434e-2);
    p = fma_(p, u, 1.4140941807431191359e-2);
    p = fma_(p, u, 1.7337192543712075605e-2);
    p = fma_(p, u, 2.2373010066676287235e-2);
    p = fma_(p, u, 3.0381917485400309022e-2);
    p = fma_(p, u, 4.4642857578717756276e-2);
    p = fma_(p, u, 7.4999999997263017056e-2);
    p = fma_(p, u, 1.6666666666666949676e-1);
    return fma_(s * u, p, s);
}
FTN_HD_NOINLINE float acosf_det(float xf) {
    if (!(xf == xf)) return xf;
    const double x = (double)xf;
    if (x > 1.0 || x < -1.0) return u2f(0x7fc00000u);
    const double pi = 3.14159265358979323846, pio2 = 1.57079632679489661923;
    const double ax = x < 0.0 ? -x : x;
    if (ax <= 0.5) {                                  /* acos(x) = pi/2 - asin(x) */
        const double as = kasin_half(ax);
        return (float)(x < 0.0 ? pio2 + as : pio2 - as);
    }
    /* acos(|x|) = 2 asin(sqrt((1-|x|)/2)); (1-|x|)/2 is exact in binary64 for a float x */
    const double a = 2.0 * kasin_half(sqrt((1.0 - ax) * 0.5));
    return (float)(x < 0.0 ? pi - a : a);
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

FTN_HD_NOINLINE float logf_det(float xf) {
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

/* ---- exp on binary64: t = k ln2 + r, |r| <= ln2 / 2, Taylor to r^13 (remainder below 1e-17), scaled by 2^k */
FTN_HD double kexp(double t) {
    if (!(t == t)) return t;
    if (t > 709.0) return u2d(0x7ff0000000000000ULL);
    if (t < -745.0) return 0.0;
    const double kf = floor(t * 1.44269504088896340736 + 0.5);
    const double r = (t - kf * 6.93147180369123816490e-01) - kf * 1.90821492927058770002e-10;
    double p = 1.0 / 6227020800.0;               /* 1/13! */
    p = 1.0 / 479001600.0 + r * p;
    p = 1.0 / 39916800.0 + r * p;
    p = 1.0 / 3628800.0 + r * p;
    p = 1.0 / 362880.0 + r * p;
    p = 1.0 / 40320.0 + r * p;
    p = 1.0 / 5040.0 + r * p;
    p = 1.0 / 720.0 + r * p;
    p = 1.0 / 120.0 + r * p;
    p = 1.0 / 24.0 + r * p;
    p = 1.0 / 6.0 + r * p;
    p = 0.5 + r * p;
    p = 1.0 + r * p;
    p = 1.0 + r * p;
    /* 2^k in two factors so that results in the binary64 subnormal range (binary32 zero anyway) stay finite operations */
    const int k = (int)kf, k1 = k / 2, k2 = k - k1;
    return (p * u2d((uint64_t)(k1 + 1023) << 52)) * u2d((uint64_t)(k2 + 1023) << 52);
}

/* f32::powf for a positive base (the only use: imageio/mod.rs:173, ((v + 0.055) / 1.055).powf(2.4) with v > 0.04045): exp(y ln x)
 * in binary64, rounded once.  x <= 0, NaN and infinities follow IEEE pow for a non-integer y. */
FTN_HD_NOINLINE float powf_det(float xf, float yf) {
    if (!(xf == xf) || !(yf == yf)) return xf + yf;
    if (yf == 0.0f || xf == 1.0f) return 1.0f;
    if (xf < 0.0f) return u2f(0x7fc00000u);
    if (xf == 0.0f) return yf > 0.0f ? 0.0f : u2f(0x7f800000u);
    if (xf - xf != 0.0f) return yf > 0.0f ? xf : 0.0f;        /* +inf */
    int e;
    const double lm = klog_parts((double)xf, &e);
    const double ef = (double)e;
    const double lx = (ef * 6.93147180369123816490e-01 + lm) + ef * 1.90821492927058770002e-10;
    return (float)kexp((double)yf * lx);
}

}  // namespace ftn_det
#endif
