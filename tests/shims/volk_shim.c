/* TEST DOUBLE (see fftw3f_shim.c): the two VOLK kernels oracle/ref_equiv.c uses, exported the way VOLK exports them — as
 * global dispatcher POINTERS named like the kernel. */
#include <complex.h>

typedef float _Complex cf;
static void mul(cf *out, const cf *a, const cf *b, unsigned int n) { for (unsigned int i = 0; i < n; i++) out[i] = a[i] * b[i]; }
static void smul(cf *out, const cf *a, const cf s, unsigned int n) { for (unsigned int i = 0; i < n; i++) out[i] = a[i] * s; }
void (*volk_32fc_x2_multiply_32fc)(cf *, const cf *, const cf *, unsigned int) = mul;
void (*volk_32fc_s32fc_multiply_32fc)(cf *, const cf *, const cf, unsigned int) = smul;
