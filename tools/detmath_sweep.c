/* tools/detmath_sweep.c -- development tool: every one-argument function of include/prt_detmath.h over ALL 2^32 binary32
 * inputs against the float64 libm, on the CPU (the GPU gives the same bits: tests/test_detmath.py).  Prints, per
 * function and domain, the largest error in units of the last place of the correctly rounded result and where it occurs.
 *   gcc -O2 -ffp-contract=off -march=x86-64-v3 -Iinclude -o /tmp/detmath_sweep tools/detmath_sweep.c -lm -lpthread
 * and the two-argument ones (atan2, pow) on 2^16 x 2^16 lattices of the argument ranges the path produces.
 *   /tmp/detmath_sweep > profiles/r03_detmath_ulp.txt                       (about ten minutes on 8 cores) */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "prt_detmath.h"

typedef float (*fn1)(float);
typedef double (*ref1)(double);
typedef struct { const char* name; fn1 f; ref1 r; const char* domain; double lo, hi; int exact; } Case;

static float w_atan2_1(float x) { return prt_atan2(x, 1.0f); }
static double r_atan(double x) { return atan(x); }
static float w_pow2(float x) { return prt_pow(x, 2.0f); }
static double r_sq(double x) { return x * x; }
static double r_rint(double x) { return rint(x); }
static double r_recip(double x) { return 1.0 / x; }

static const Case CASES[] = {
    {"prt_sin", prt_sin, sin, "|x| <= 1e4", -1e4, 1e4, 0},
    {"prt_cos", prt_cos, cos, "|x| <= 1e4", -1e4, 1e4, 0},
    {"prt_tan", prt_tan, tan, "|x| <= 1e4", -1e4, 1e4, 0},
    /* (beyond ~1e4 the three-constant Cody-Waite reduction loses accuracy, as the header says: the path tracer's
     *  arguments are angles in [0, 2 pi] and fov / 2) */
    {"prt_exp", prt_exp, exp, "all x (result finite, normal)", -87.0, 88.7, 0},
    {"prt_log", prt_log, log, "x > 0 (normal)", 1.1754944e-38, INFINITY, 0},
    {"prt_acos", prt_acos, acos, "|x| <= 1", -1.0, 1.0, 0},
    {"prt_atan2(x, 1)", w_atan2_1, r_atan, "all finite x", -INFINITY, INFINITY, 0},
    {"prt_cbrt", prt_cbrt, cbrt, "all finite x", -INFINITY, INFINITY, 0},
    {"prt_sqrt", prt_sqrt, sqrt, "x >= 0", 0.0, INFINITY, 1},
    {"prt_recip", prt_recip, r_recip, "all finite x != 0", -INFINITY, INFINITY, 1},
    {"prt_pow(x, 2)", w_pow2, r_sq, "all finite x", -INFINITY, INFINITY, 1},
    {"prt_rint", prt_rint, r_rint, "all finite x", -INFINITY, INFINITY, 1},
    {"prt_floor", prt_floor, floor, "all finite x", -INFINITY, INFINITY, 1},
    {"prt_round", prt_round, round, "all finite x", -INFINITY, INFINITY, 1},
    {"prt_trunc", prt_trunc, trunc, "all finite x", -INFINITY, INFINITY, 1},
};
#define NT 8
typedef struct { const Case* c; int tid; double max_ulp; uint32_t arg; uint64_t n, wrong; } Job;

static double ulp_err(float got, double want) {
    if (isnan(want) || isinf(want)) return (isnan(want) ? isnan(got) : got == (float)want) ? 0.0 : INFINITY;
    const float rn = (float)want;                              /* correctly rounded result */
    if (isinf(rn)) return isinf(got) && (got > 0) == (rn > 0) ? 0.0 : INFINITY;
    int e;
    frexp(want == 0.0 ? 1e-45 : want, &e);
    double ulp = ldexp(1.0, e - 24);
    if (ulp < ldexp(1.0, -149)) ulp = ldexp(1.0, -149);        /* subnormal results: absolute spacing */
    return fabs((double)got - want) / ulp;
}

static void* work(void* arg) {
    Job* j = (Job*)arg;
    const Case* c = j->c;
    j->max_ulp = 0; j->n = 0; j->wrong = 0; j->arg = 0;
    for (uint64_t b = (uint64_t)j->tid; b < (1ull << 32); b += NT) {
        float x;
        uint32_t u = (uint32_t)b;
        memcpy(&x, &u, 4);
        if (isnan(x) || isinf(x)) continue;
        if (!((double)x >= c->lo && (double)x <= c->hi)) continue;
        const float got = c->f(x);
        const double want = c->r((double)x);
        if (isinf(want) || fabs(want) > 3.4028234e38 || (want != 0.0 && fabs(want) < 1.1754944e-38 && !c->exact)) continue;
        double err;
        if (c->exact) {
            const float rn = (float)want;
            err = (memcmp(&rn, &got, 4) == 0 || (rn == 0.0f && got == 0.0f)) ? 0.0 : ulp_err(got, want);
            if (err != 0.0) ++j->wrong;
        } else err = ulp_err(got, want);
        ++j->n;
        if (err > j->max_ulp) { j->max_ulp = err; j->arg = u; }
    }
    return 0;
}

/* ---- two-argument functions: 2^16 x 2^16 lattices of the ranges the radiance loop passes ------------------------------- */
typedef struct { const char* name; const char* domain; int kind; } Case2;
static const Case2 CASES2[] = {
    {"prt_atan2(y, x)", "x, y on a lattice of [-1, 1]^2 (utils.cl:46 envMapEquirect: atan2(dir.z, dir.x))", 0},
    {"prt_atan2(y, x)", "|x|, |y| = 2^a, a on a lattice of [-40, 20], all four sign pairs", 1},
    {"prt_pow(c, alpha)", "c on a lattice of (0, 1], alpha = 2 / r^2 - 2, r on a lattice of [1e-3, 1] (microfacet.cl:31-33 Phong D)", 2},
    {"prt_pow(xi, 1 / (alpha + 2))", "xi on a lattice of (0, 1), alpha as above (microfacet.cl:95-97 Phong sample)", 3},
    {"prt_pow(v, 1 / 2.2)", "v on a lattice of [0, 1] x 2^16 sub-steps (shaders/tonemapper.glsl gamma)", 4},
};
typedef struct { const Case2* c; int tid; double max_ulp; float ax, ay; uint64_t n; } Job2;
static void* work2(void* arg) {
    Job2* j = (Job2*)arg;
    j->max_ulp = 0; j->n = 0; j->ax = j->ay = 0;
    for (int i = j->tid; i < 65536; i += NT)
        for (int k = 0; k < 65536; ++k) {
            float x, y, got; double want;
            const float u = ((float)i + 0.5f) / 65536.0f, v = ((float)k + 0.5f) / 65536.0f;
            switch (j->c->kind) {
                case 0: x = 2.0f * u - 1.0f; y = 2.0f * v - 1.0f; got = prt_atan2(y, x); want = atan2((double)y, (double)x); break;
                case 1: { const int sg = (i & 1) | ((k & 1) << 1);
                          x = exp2f(-40.0f + 60.0f * u); y = exp2f(-40.0f + 60.0f * v); if (sg & 1) x = -x; if (sg & 2) y = -y;
                          got = prt_atan2(y, x); want = atan2((double)y, (double)x); break; }
                case 2: { const float r = 1e-3f + (1.0f - 1e-3f) * v; x = u; y = 2.0f / (r * r) - 2.0f; if (y == 2.0f) continue;
                          got = prt_pow(x, y); want = pow((double)x, (double)y); break; }
                case 3: { const float r = 1e-3f + (1.0f - 1e-3f) * v; x = u; y = 1.0f / ((2.0f / (r * r) - 2.0f) + 2.0f);
                          got = prt_pow(x, y); want = pow((double)x, (double)y); break; }
                default: x = u + v / 65536.0f; y = 1.0f / 2.2f; got = prt_pow(x, y); want = pow((double)x, (double)y); break;
            }
            if (want != 0.0 && fabs(want) < 1.1754944e-38) continue;            /* subnormal results are not priced */
            const double err = ulp_err(got, want);
            ++j->n;
            if (err > j->max_ulp) { j->max_ulp = err; j->ax = x; j->ay = y; }
        }
    return 0;
}

int main(int argc, char** argv) {
    const int only2 = argc > 1 && argv[1][0] == '2';
    printf("include/prt_detmath.h, one-argument functions, every binary32 input of the domain, against float64 libm (CPU, x86-64)\n");
    printf("%-18s %-34s %14s %12s %12s  %s\n", "function", "domain", "inputs", "max ulp", "at (bits)", "note");
    for (unsigned k = 0; !only2 && k < sizeof(CASES) / sizeof(CASES[0]); ++k) {
        pthread_t th[NT];
        Job jobs[NT];
        for (int t = 0; t < NT; ++t) { jobs[t].c = &CASES[k]; jobs[t].tid = t; pthread_create(&th[t], 0, work, &jobs[t]); }
        double mx = 0; uint32_t at = 0; uint64_t n = 0, wrong = 0;
        for (int t = 0; t < NT; ++t) { pthread_join(th[t], 0); n += jobs[t].n; wrong += jobs[t].wrong; if (jobs[t].max_ulp > mx) { mx = jobs[t].max_ulp; at = jobs[t].arg; } }
        char note[96] = "";
        if (CASES[k].exact) snprintf(note, sizeof note, "must be correctly rounded: %llu inputs are not", (unsigned long long)wrong);
        printf("%-18s %-34s %14llu %12.4f   0x%08x  %s\n", CASES[k].name, CASES[k].domain, (unsigned long long)n, mx, at, note);
        fflush(stdout);
    }
    printf("\ntwo-argument functions, 2^16 x 2^16 lattices of the argument ranges the path produces, against float64 libm\n");
    printf("%-30s %14s %10s  %-28s %s\n", "function", "pairs", "max ulp", "at (first, second argument)", "domain");
    for (unsigned k = 0; k < sizeof(CASES2) / sizeof(CASES2[0]); ++k) {
        pthread_t th[NT];
        Job2 jobs[NT];
        for (int t = 0; t < NT; ++t) { jobs[t].c = &CASES2[k]; jobs[t].tid = t; pthread_create(&th[t], 0, work2, &jobs[t]); }
        double mx = 0; float ax = 0, ay = 0; uint64_t n = 0;
        for (int t = 0; t < NT; ++t) { pthread_join(th[t], 0); n += jobs[t].n; if (jobs[t].max_ulp > mx) { mx = jobs[t].max_ulp; ax = jobs[t].ax; ay = jobs[t].ay; } }
        printf("%-30s %14llu %10.4f  (%-13a, %-13a) %s\n", CASES2[k].name, (unsigned long long)n, mx, ax, ay, CASES2[k].domain);
        fflush(stdout);
    }
    return 0;
}
