/* Counts the device / pinned-host allocation calls a process makes through the HIP runtime, without LD_PRELOAD: loaded with RTLD_GLOBAL BEFORE
 * libfdc_amd.so, its definitions of hipMalloc & co. come first in the global lookup scope, so the library's calls bind here; each forwards to the
 * runtime's own (RTLD_NEXT).  tests/test_no_alloc_gpu.py reads the counters around 50 calls of every steady-state entry. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stddef.h>

static long n_malloc, n_free, n_host_malloc, n_host_free;
/* the runtime's own entry: next in the lookup order, or (the shim linked without the runtime: --as-needed drops an unreferenced library) by handle */
static void *real_sym(const char *name)
{
    void *f = dlsym(RTLD_NEXT, name);
    if (!f) {
        void *h = dlopen("libamdhip64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/libamdhip64.so", RTLD_NOW | RTLD_GLOBAL);
        if (h) f = dlsym(h, name);
    }
    return f;
}
typedef int (*malloc_fn)(void **, size_t);
typedef int (*free_fn)(void *);
typedef int (*host_malloc_fn)(void **, size_t, unsigned);

int hipMalloc(void **p, size_t n)
{
    static malloc_fn real;
    if (!real) real = (malloc_fn)real_sym("hipMalloc");
    __atomic_add_fetch(&n_malloc, 1, __ATOMIC_RELAXED);
    return real(p, n);
}
int hipFree(void *p)
{
    static free_fn real;
    if (!real) real = (free_fn)real_sym("hipFree");
    if (p) __atomic_add_fetch(&n_free, 1, __ATOMIC_RELAXED);      /* hipFree(NULL) is a no-op the destructors make freely */
    return real(p);
}
int hipHostMalloc(void **p, size_t n, unsigned flags)
{
    static host_malloc_fn real;
    if (!real) real = (host_malloc_fn)real_sym("hipHostMalloc");
    __atomic_add_fetch(&n_host_malloc, 1, __ATOMIC_RELAXED);
    return real(p, n, flags);
}
int hipHostFree(void *p)
{
    static free_fn real;
    if (!real) real = (free_fn)real_sym("hipHostFree");
    if (p) __atomic_add_fetch(&n_host_free, 1, __ATOMIC_RELAXED);
    return real(p);
}
void fdc_test_alloc_counts(long *v4) { v4[0] = n_malloc; v4[1] = n_free; v4[2] = n_host_malloc; v4[3] = n_host_free; }
