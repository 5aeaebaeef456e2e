/*
 * prt_detmath.h -- the numerics contract of the prt C-ABI.
 *
 * The radiance loop of the reference (kernels/main.cl render_kernel and everything it includes)
 * calls OpenCL C built-ins (sin cos tan atan2 acos exp log pow sqrt native_* fast_* ...) whose
 * results are implementation-defined within the OpenCL ULP bounds.  A path tracer is chaotic:
 * one flipped decision (Russian roulette, Fresnel choice, edge hit/miss) desynchronises a pixel
 * for ever, so "same picture as the reference" is only meaningful against ONE stated
 * implementation of those built-ins.  This header is that implementation.  Every function is
 * written with IEEE-754 binary32 (prt_pow: binary64) +,-,*,/,sqrt,fma and integer operations only, so that it
 * produces the same bits on x86-64 (gcc / clang, -ffp-contract=off) and on gfx950 (hipcc,
 * -ffp-contract=off, correctly rounded divide/sqrt which is hipcc's default).
 *
 * Users:  the HIP kernels in photorealistic-rendering-using-opencl_amd/csrc (device side),
 *         the host code of libprt, and -- as the stated OpenCL built-in library -- the test
 *         oracle under oracle/ (oracle/pt_oracle.c and the OpenCL runtime shim that lets the
 *         reference's own kernel text run on the host).
 *
 * Accuracy (tests/test_detmath.py, against float64 libm): sin/cos <= 2 ulp on |x| <= 1e4, tan <= 4 ulp,
 * exp/log <= 2 ulp, acos/atan2 <= 3 ulp, cbrt <= 2 ulp, pow(x,2) exact, general pow <= 0.51 ulp (computed in binary64).
 * All inside the OpenCL 1.2 full-profile bounds.  sin / cos / tan lose accuracy beyond |x| ~ 1e4 (three-constant argument
 * reduction): the path only passes angles in [0, 2 pi] and half the field of view.
 */
#ifndef PRT_DETMATH_H
#define PRT_DETMATH_H

#if defined(__HIP__)          /* clang in HIP mode: usable from host and device code */
#define PRT_HD __attribute__((host)) __attribute__((device)) static inline __attribute__((always_inline))
#else
#define PRT_HD static inline
#endif

/* ---- bit casts ------------------------------------------------------------------------- */
PRT_HD unsigned prt_f2u(float f) { unsigned u; __builtin_memcpy(&u, &f, 4); return u; }
PRT_HD float prt_u2f(unsigned u) { float f; __builtin_memcpy(&f, &u, 4); return f; }

/* ---- exact IEEE primitives -------------------------------------------------------------- */
PRT_HD float prt_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
PRT_HD float prt_sqrt(float x) { return __builtin_sqrtf(x); }
PRT_HD float prt_fabs(float x) { return prt_u2f(prt_f2u(x) & 0x7fffffffu); }
PRT_HD float prt_copysign(float x, float s) {
    return prt_u2f((prt_f2u(x) & 0x7fffffffu) | (prt_f2u(s) & 0x80000000u));
}
PRT_HD float prt_recip(float x) { return 1.0f / x; }           /* native_recip: IEEE divide */
PRT_HD int prt_isnan(float x) { return x != x; }

/* OpenCL fmin/fmax: a NaN operand is ignored; ties (+0,-0) return the FIRST operand. */
PRT_HD float prt_fmin(float a, float b) {
    if (b != b) return a;
    if (a != a) return b;
    return (b < a) ? b : a;
}
PRT_HD float prt_fmax(float a, float b) {
    if (b != b) return a;
    if (a != a) return b;
    return (a < b) ? b : a;
}

/* round-half-to-even to an integral float (|x| < 2^23 uses the add-magic trick, exact) */
PRT_HD float prt_rint(float x) {
    float ax = prt_fabs(x);
    if (!(ax < 8388608.0f)) return x;            /* already integral, inf or NaN */
    float r = (ax + 8388608.0f) - 8388608.0f;    /* RN-even in binary32 */
    return prt_copysign(r, x);
}
PRT_HD float prt_trunc(float x) {
    float ax = prt_fabs(x);
    if (!(ax < 8388608.0f)) return x;
    float r = (ax + 8388608.0f) - 8388608.0f;
    if (r > ax) r -= 1.0f;
    return prt_copysign(r, x);
}
PRT_HD float prt_floor(float x) {
    float t = prt_trunc(x);
    return (t > x) ? t - 1.0f : t;
}
/* OpenCL round(): halfway cases away from zero */
PRT_HD float prt_round(float x) {
    float t = prt_trunc(x);
    if (prt_fabs(x - t) >= 0.5f) t += prt_copysign(1.0f, x);
    return t;
}
PRT_HD float prt_fract(float x) {                 /* OpenCL fract: min(x - floor(x), 0x1.fffffep-1f) */
    float f = x - prt_floor(x);
    return prt_fmin(f, 0x1.fffffep-1f);
}
PRT_HD float prt_mix(float a, float b, float t) { return a + (b - a) * t; }

/* ---- sin / cos -------------------------------------------------------------------------- */
/* Cody-Waite reduction by pi/2 in three fma steps, Taylor kernels on [-pi/4, pi/4]. */
PRT_HD void prt_sincos_kernel(float x, float* s_out, float* c_out, int* q_out) {
    const float TWO_OVER_PI = 0x1.45f306p-1f;
    const float PIO2_H = 0x1.921fb6p+0f;
    const float PIO2_M = -0x1.777a5cp-25f;
    const float PIO2_L = -0x1.ee59dap-50f;
    float k = prt_rint(x * TWO_OVER_PI);
    float r = prt_fma(-k, PIO2_H, x);
    r = prt_fma(-k, PIO2_M, r);
    r = prt_fma(-k, PIO2_L, r);
    float s = r * r;
    /* sin r = r - r^3/3! + r^5/5! - r^7/7! + r^9/9! - r^11/11! */
    float ps = -0x1.ae6456p-26f;                    /* -1/39916800 */
    ps = prt_fma(ps, s, 0x1.71de3ap-19f);           /*  1/362880   */
    ps = prt_fma(ps, s, -0x1.a01a02p-13f);          /* -1/5040     */
    ps = prt_fma(ps, s, 0x1.111112p-7f);            /*  1/120      */
    ps = prt_fma(ps, s, -0x1.555556p-3f);           /* -1/6        */
    float sn = prt_fma(r * s, ps, r);
    /* cos r = 1 - s/2 + s^2/4! - s^3/6! + s^4/8! - s^5/10! + s^6/12! */
    float pc = 0x1.1eed8ep-29f;                     /*  1/479001600 */
    pc = prt_fma(pc, s, -0x1.27e4fcp-22f);          /* -1/3628800   */
    pc = prt_fma(pc, s, 0x1.a01a02p-16f);           /*  1/40320     */
    pc = prt_fma(pc, s, -0x1.6c16c2p-10f);          /* -1/720       */
    pc = prt_fma(pc, s, 0x1.555556p-5f);            /*  1/24        */
    float cs = prt_fma(s * s, pc, prt_fma(-0.5f, s, 1.0f));
    *s_out = sn;
    *c_out = cs;
    *q_out = (int)k;                                /* |k| < 2^31 for the supported domain */
}
PRT_HD float prt_sin(float x) {
    if (!(prt_fabs(x) < 1.0e9f)) return x - x;      /* inf, NaN, unsupported magnitude -> NaN/0 */
    float s, c; int q;
    prt_sincos_kernel(x, &s, &c, &q);
    float v = (q & 1) ? c : s;
    return (q & 2) ? -v : v;
}
PRT_HD float prt_cos(float x) {
    if (!(prt_fabs(x) < 1.0e9f)) return x - x;
    float s, c; int q;
    prt_sincos_kernel(x, &s, &c, &q);
    float v = (q & 1) ? s : c;
    return ((q + 1) & 2) ? -v : v;
}
PRT_HD float prt_tan(float x) {
    if (!(prt_fabs(x) < 1.0e9f)) return x - x;
    float s, c; int q;
    prt_sincos_kernel(x, &s, &c, &q);
    return (q & 1) ? (-c / s) : (s / c);
}

/* ---- exp -------------------------------------------------------------------------------- */
PRT_HD float prt_pow2i(int n) {                     /* 2^n, n in [-126,127] */
    return prt_u2f((unsigned)(n + 127) << 23);
}
PRT_HD float prt_exp(float x) {
    if (x != x) return x;
    if (x > 88.72284f) return prt_u2f(0x7f800000u);
    if (x < -103.98f) return 0.0f;
    const float LOG2E = 0x1.715476p+0f;
    const float LN2_H = 0x1.62e4p-1f;               /* 15 significant bits: n*LN2_H is exact */
    const float LN2_L = 0x1.7f7d1cp-20f;
    float n = prt_rint(x * LOG2E);
    float r = prt_fma(-n, LN2_H, x);
    r = prt_fma(-n, LN2_L, r);
    /* e^r, |r| <= ln2/2: 1 + r + r^2/2 + ... + r^8/8! */
    float p = 0x1.a01a02p-16f;                      /* 1/40320 */
    p = prt_fma(p, r, 0x1.a01a02p-13f);             /* 1/5040  */
    p = prt_fma(p, r, 0x1.6c16c2p-10f);             /* 1/720   */
    p = prt_fma(p, r, 0x1.111112p-7f);              /* 1/120   */
    p = prt_fma(p, r, 0x1.555556p-5f);              /* 1/24    */
    p = prt_fma(p, r, 0x1.555556p-3f);              /* 1/6     */
    p = prt_fma(p, r, 0.5f);
    p = prt_fma(p * r, r, r) + 1.0f;
    int ni = (int)n;
    int n1 = ni / 2, n2 = ni - n1;                  /* two-step scaling reaches the subnormals */
    return (p * prt_pow2i(n1)) * prt_pow2i(n2);
}

/* ---- log -------------------------------------------------------------------------------- */
PRT_HD float prt_log(float x) {
    if (x != x) return x;
    if (x < 0.0f) return prt_u2f(0x7fc00000u);
    if (x == 0.0f) return prt_u2f(0xff800000u);
    unsigned u = prt_f2u(x);
    if (u == 0x7f800000u) return x;
    int e = 0;
    if (u < 0x00800000u) { x *= 8388608.0f; u = prt_f2u(x); e = -23; }
    e += (int)(u >> 23) - 127;
    float m = prt_u2f((u & 0x007fffffu) | 0x3f800000u);   /* [1,2) */
    if (m > 0x1.6a09e6p+0f) { m *= 0.5f; e += 1; }         /* [sqrt(1/2), sqrt(2)] */
    float f = m - 1.0f;                                     /* exact */
    float s = f / (m + 1.0f);
    float z = s * s;
    /* ln m = 2s (1 + z/3 + z^2/5 + z^3/7 + z^4/9 + z^5/11) */
    float p = 0x1.745d18p-4f;                       /* 1/11 */
    p = prt_fma(p, z, 0x1.c71c72p-4f);              /* 1/9  */
    p = prt_fma(p, z, 0x1.24924ap-3f);              /* 1/7  */
    p = prt_fma(p, z, 0x1.99999ap-3f);              /* 1/5  */
    p = prt_fma(p, z, 0x1.555556p-2f);              /* 1/3  */
    float lm = prt_fma(2.0f * s * z, p, 2.0f * s);
    const float LN2_H = 0x1.62e4p-1f;
    const float LN2_L = 0x1.7f7d1cp-20f;
    float fe = (float)e;
    return prt_fma(fe, LN2_H, prt_fma(fe, LN2_L, lm));
}

/* ---- pow --------------------------------------------------------------------------------
 * General case in IEEE binary64 (+ - * / and integer operations only, like everything here: the same bits on x86-64 and gfx950):
 * x = 2^e m with m in [sqrt(1/2), sqrt(2)), log2 x = e + log2(e) * 2 atanh((m - 1) / (m + 1)) by its series, t = y log2 x,
 * 2^t = 2^n exp((t - n) ln 2) by its series, one rounding to binary32 at the end.  The error of t is below 2^-42 of the result's
 * exponent over the whole range (|t| <= 150), so the result is the correctly rounded one except for results within 2^-18 ulp of
 * a rounding boundary: at most 0.5 ulp + 4e-6 (measured on a 2^16 x 2^16 lattice of the arguments the Phong lobe produces,
 * kernels/bxdf/microfacet.cl:31-33,95-97: profiles/r03_detmath_ulp.txt).  Round 1-2 computed exp(y * log x) in binary32, whose
 * error grows with |y ln x| (tens of ulp for the Phong exponents 2 / r^2 - 2 of small roughness). */
PRT_HD float prt_pow(float x, float y) {
    if (y == 2.0f) return x * x;                    /* the only form the BASELINE configs use; exact */
    if (y == 0.0f || x == 1.0f) return 1.0f;
    if (x != x || y != y) return x + y;
    if (y == 1.0f) return x;
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : prt_u2f(0x7f800000u);
    if (x < 0.0f) return prt_u2f(0x7fc00000u);      /* negative bases are never used by the path */
    const float INF = prt_u2f(0x7f800000u);
    if (x == INF) return (y > 0.0f) ? INF : 0.0f;
    if (y == INF || y == -INF) return ((x > 1.0f) == (y > 0.0f)) ? INF : 0.0f;
    unsigned ux = prt_f2u(x);
    int e = 0;
    if (ux < 0x00800000u) { ux = prt_f2u(x * 16777216.0f); e = -24; }            /* subnormal base */
    e += (int)(ux >> 23) - 127;
    float m = prt_u2f((ux & 0x007fffffu) | 0x3f800000u);                         /* [1, 2) */
    if (m > 0x1.6a09e6p+0f) { m *= 0.5f; e += 1; }                               /* [sqrt(1/2), sqrt(2)) */
    const double f = (double)m - 1.0;
    const double s = f / (2.0 + f), z = s * s;
    double p = 1.0 / 21.0;                                                      /* atanh s = s (1 + z/3 + z^2/5 + ...), z <= 0.0295 */
    p = p * z + 1.0 / 19.0; p = p * z + 1.0 / 17.0; p = p * z + 1.0 / 15.0; p = p * z + 1.0 / 13.0; p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0; p = p * z + 1.0 / 7.0; p = p * z + 1.0 / 5.0; p = p * z + 1.0 / 3.0;
    const double ln_m = 2.0 * s + 2.0 * s * (z * p);
    const double t = (double)y * ((double)e + ln_m * 0x1.71547652b82fep+0);      /* log2(e) */
    if (!(t < 128.0)) return INF;
    if (t < -150.0) return 0.0f;
    const long long n = (long long)(t + (t >= 0.0 ? 0.5 : -0.5));
    const double w = (t - (double)n) * 0x1.62e42fefa39efp-1;                     /* ln 2; |w| <= 0.347 */
    double q = 1.0 / 6227020800.0;                                               /* exp w, degree 13 */
    q = q * w + 1.0 / 479001600.0; q = q * w + 1.0 / 39916800.0; q = q * w + 1.0 / 3628800.0; q = q * w + 1.0 / 362880.0;
    q = q * w + 1.0 / 40320.0; q = q * w + 1.0 / 5040.0; q = q * w + 1.0 / 720.0; q = q * w + 1.0 / 120.0; q = q * w + 1.0 / 24.0;
    q = q * w + 1.0 / 6.0; q = q * w + 0.5; q = q * w + 1.0; q = q * w + 1.0;
    unsigned long long sb = (unsigned long long)(n + 1023) << 52;                /* 2^n, n in [-150, 128] */
    double scale;
    __builtin_memcpy(&scale, &sb, 8);
    return (float)(q * scale);
}

/* ---- cbrt (kernels/phasefunctions/Rayleigh.cl:24) ------------------------------------------------------ */
PRT_HD float prt_cbrt(float x) {
    if (x != x || x == 0.0f) return x;
    unsigned ux = prt_f2u(x);
    const unsigned sign = ux & 0x80000000u;
    ux &= 0x7fffffffu;
    if (ux == 0x7f800000u) return x;
    float ax = prt_u2f(ux);
    int e = 0;
    if (ux < 0x00800000u) { ax *= 16777216.0f; ux = prt_f2u(ax); e = -8; }     /* subnormal: scale by 2^24 = (2^8)^3 */
    /* initial guess: exponent / 3 by integer arithmetic on the bit pattern, then 4 Newton steps y -= (y^3 - a) / (3 y^2) */
    float y = prt_u2f(ux / 3u + 0x2a5137a0u);
    for (int i = 0; i < 4; ++i) {
        const float y2 = y * y;
        y = y - prt_fma(y2, y, -ax) / (3.0f * y2);
    }
    if (e) y *= 0.00390625f;                                                    /* 2^-8 */
    return prt_u2f(prt_f2u(y) | sign);
}

/* ---- inverse trigonometry --------------------------------------------------------------- */
PRT_HD float prt_asin_poly(float z) {               /* asin z, 0 <= z <= 0.5 (Taylor, 11 terms) */
    float t = z * z;
    float p = 0x1.12ef3cp-7f;                       /* c10 = 46189/5505024 */
    p = prt_fma(p, t, 0x1.3fde50p-7f);              /* c9  = 12155/1245184 */
    p = prt_fma(p, t, 0x1.7a8788p-7f);              /* c8  = 6435/557056   */
    p = prt_fma(p, t, 0x1.c9999ap-7f);              /* c7  = 143/10240 */
    p = prt_fma(p, t, 0x1.1c4ec4p-6f);              /* c6  = 231/13312 */
    p = prt_fma(p, t, 0x1.6e8ba2p-6f);              /* c5  = 63/2816   */
    p = prt_fma(p, t, 0x1.f1c71cp-6f);              /* c4  = 35/1152   */
    p = prt_fma(p, t, 0x1.6db6dcp-5f);              /* c3  = 5/112     */
    p = prt_fma(p, t, 0x1.333334p-4f);              /* c2  = 3/40      */
    p = prt_fma(p, t, 0x1.555556p-3f);              /* c1  = 1/6       */
    return prt_fma(z * t, p, z);
}
PRT_HD float prt_acos(float x) {
    const float PIO2_H = 0x1.921fb6p+0f, PIO2_L = -0x1.777a5cp-25f;
    const float PI_H = 0x1.921fb6p+1f, PI_L = -0x1.777a5cp-24f;
    float ax = prt_fabs(x);
    if (!(ax <= 1.0f)) return prt_u2f(0x7fc00000u);
    if (ax <= 0.5f) return (PIO2_H - prt_asin_poly(ax) * prt_copysign(1.0f, x)) + PIO2_L;
    float z = prt_sqrt((1.0f - ax) * 0.5f);
    float a = 2.0f * prt_asin_poly(z);
    return (x > 0.0f) ? a : ((PI_H - a) + PI_L);
}
PRT_HD float prt_atan_poly(float z) {               /* atan z, |z| <= tan(pi/8) (Taylor, 10 terms) */
    float t = z * z;
    float p = -0x1.af286cp-5f;                      /* -1/19 */
    p = prt_fma(p, t, 0x1.e1e1e2p-5f);              /*  1/17 */
    p = prt_fma(p, t, -0x1.111112p-4f);             /* -1/15 */
    p = prt_fma(p, t, 0x1.3b13b2p-4f);              /*  1/13 */
    p = prt_fma(p, t, -0x1.745d18p-4f);             /* -1/11 */
    p = prt_fma(p, t, 0x1.c71c72p-4f);              /*  1/9  */
    p = prt_fma(p, t, -0x1.24924ap-3f);             /* -1/7  */
    p = prt_fma(p, t, 0x1.99999ap-3f);              /*  1/5  */
    p = prt_fma(p, t, -0x1.555556p-2f);             /* -1/3  */
    return prt_fma(z * t, p, z);
}
PRT_HD float prt_atan_pos(float t) {                /* atan t, t >= 0 (incl. +inf) */
    const float PIO2_H = 0x1.921fb6p+0f, PIO2_L = -0x1.777a5cp-25f;
    const float PIO4_H = 0x1.921fb6p-1f, PIO4_L = -0x1.777a5cp-26f;
    int inv = 0;
    if (t > 1.0f) { t = 1.0f / t; inv = 1; }
    float r;
    if (t > 0x1.a8279ap-2f) {                       /* tan(pi/8) */
        r = (PIO4_H + prt_atan_poly((t - 1.0f) / (t + 1.0f))) + PIO4_L;
    } else {
        r = prt_atan_poly(t);
    }
    return inv ? ((PIO2_H - r) + PIO2_L) : r;
}
PRT_HD float prt_atan2(float y, float x) {
    const float PI_H = 0x1.921fb6p+1f, PI_L = -0x1.777a5cp-24f;
    const float PIO2_H = 0x1.921fb6p+0f;
    if (x != x || y != y) return x + y;
    if (y == 0.0f) {
        unsigned neg = prt_f2u(x) & 0x80000000u;    /* x = -0 counts as negative, as in C */
        return neg ? prt_copysign(PI_H, y) : y;
    }
    if (x == 0.0f) return prt_copysign(PIO2_H, y);
    float ax = prt_fabs(x), ay = prt_fabs(y);
    float r;
    if (ax == ay) r = 0x1.921fb6p-1f;               /* also inf/inf */
    else r = prt_atan_pos(ay / ax);
    if (x < 0.0f) r = (PI_H - r) + PI_L;
    return prt_copysign(r, y);
}

#endif /* PRT_DETMATH_H */
