/* oracle/detmath_probe.c -- TEST INFRASTRUCTURE: host evaluation of include/prt_detmath.h, array
 * at a time, with the same function numbering as prt_selftest_math (include/prt.h). */
#include "prt_detmath.h"

void detmath_probe(int fn, const float* a, const float* b, float* out, int n) {
    for (int i = 0; i < n; ++i) {
        const float x = a[i], y = b[i];
        float r;
        switch (fn) {
            case 0: r = prt_sin(x); break;
            case 1: r = prt_cos(x); break;
            case 2: r = prt_tan(x); break;
            case 3: r = prt_exp(x); break;
            case 4: r = prt_log(x); break;
            case 5: r = prt_acos(x); break;
            case 6: r = prt_atan2(x, y); break;
            case 7: r = prt_pow(x, y); break;
            case 8: r = prt_sqrt(x); break;
            case 9: r = x / y; break;
            case 10: r = prt_fma(x, y, x); break;
            case 11: r = prt_fmin(x, y); break;
            case 12: r = prt_fmax(x, y); break;
            case 13: r = prt_round(x); break;
            case 14: r = prt_floor(x); break;
            case 16: r = prt_cbrt(x); break;
            default: r = prt_recip(x); break;
        }
        out[i] = r;
    }
}
