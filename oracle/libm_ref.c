/* libm_ref.c -- the C library's own sin / cos / atan2 / atan / acos / log / pow over arrays: what the reference's CPU path
 * calls (its Real is double: /root/reference/src/redner.h:46; call sites /root/reference/src/camera.h:142-191,
 * src/material.h, src/envmap.h, src/edge.cpp).  TEST INFRASTRUCTURE: the checker of redner_amd/csrc/libm_exact.h
 * (tests/test_libm_exact.py); nothing under redner_amd/ links or loads it.
 * Built by oracle/Makefile (or by the test itself) with -fno-builtin so that every value comes from libm at run time. */
#include <math.h>

void libm_ref_eval(int fn, const double *x, const double *y, double *out, long n) {
    for (long i = 0; i < n; ++i) {
        double a = x[i], b = y ? y[i] : 0.0, r;
        switch (fn) {
            case 0: r = sin(a); break;
            case 1: r = cos(a); break;
            case 2: r = atan2(a, b); break;
            case 3: r = atan(a); break;
            case 4: r = acos(a); break;
            case 5: r = log(a); break;
            default: r = pow(a, b); break;
        }
        out[i] = r;
    }
}
