/* TEST DOUBLE, not a product file and not a baseline: the five entry points of FFTW3f that oracle/ref_equiv.c looks up with
 * dlopen(), as a plain O(n^2) DFT in double precision.  It exists so that the "reference-equivalent" CPU leg of bench.py
 * (dead code on every box without FFTW3f and VOLK) is executed once by tests/test_refequiv_cpu.py; nothing is timed with it. */
#include <complex.h>
#include <math.h>
#include <stdlib.h>

typedef float _Complex cf;
struct plan { int n, sign; };

void *fftwf_malloc(size_t n) { return malloc(n); }
void fftwf_free(void *p) { free(p); }
void *fftwf_plan_dft_1d(int n, cf *in, cf *out, int sign, unsigned flags)
{
    (void)in; (void)out; (void)flags;
    struct plan *p = malloc(sizeof *p);
    p->n = n; p->sign = sign;
    return p;
}
void fftwf_destroy_plan(void *p) { free(p); }
void fftwf_execute_dft(const void *pv, cf *in, cf *out)
{
    const struct plan *p = pv;
    const int n = p->n;
    for (int k = 0; k < n; k++) {
        double re = 0.0, im = 0.0;
        for (int j = 0; j < n; j++) {
            const double a = (double)p->sign * 2.0 * M_PI * (double)(((long long)j * k) % n) / (double)n;
            const double c = cos(a), s = sin(a), xr = crealf(in[j]), xi = cimagf(in[j]);
            re += xr * c - xi * s; im += xr * s + xi * c;
        }
        out[k] = (float)re + (float)im * I;
    }
}
