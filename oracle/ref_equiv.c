/* TEST INFRASTRUCTURE (bench.py's cpu_baseline leg only): the reference's throughput chain on the host cores with the
 * reference's OWN arithmetic libraries — FFTW3f for both transforms (gr-fft's fft_vcc is a thin wrapper over fftwf plans;
 * 4 threads on the forward transform, python/FrequencyDomainChannelizer.py:206) and VOLK for the window multiplication
 * (volk_32fc_x2_multiply_32fc, lib/phase_shifting_windowing_vcc_impl.cc:81) and the scalings — behind the reference's stage
 * boundaries (every block of the flowgraph writes its output buffer):
 *
 *   overlap_save (lib/overlap_save_impl.cc:62-81) -> fft_vcc(N, forward, shift) (py:206) -> multiply_const_cc(1/N) (py:214-216)
 *   -> per channel { vector_cut_vxx (lib/vector_cut_vxx_impl.cc:59-72) -> phase_shifting_windowing_vcc (:72-86)
 *   -> fft_vcc(l, inverse, shift) (py:228) -> vector_cut_vxx(l, l - lout, lout) (py:229) -> multiply_const_cc(l) (py:231) }
 *
 * Neither library is linked: both are looked for with dlopen() when the leg runs, so the file builds everywhere and the leg
 * runs only where the reference itself could run.  GNU Radio's scheduler gives every block its own thread; here the channel
 * branches are spread over the cores with OpenMP, which is the same degree of parallelism without the scheduler's overheads
 * (an upper bound for the reference, labelled "reference-equivalent", not "reference").
 * Nothing in the product links or loads this file. */
#include <complex.h>
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <omp.h>

typedef float _Complex cf;
typedef void *fplan;

static struct {
    void *h, *ht;
    int (*init_threads)(void);
    void (*plan_with_nthreads)(int);
    fplan (*plan_dft_1d)(int, cf *, cf *, int, unsigned);
    void (*execute_dft)(const fplan, cf *, cf *);
    void (*destroy_plan)(fplan);
    void *(*fmalloc)(size_t);
    void (*ffree)(void *);
} F;
static struct {
    void *h;
    void (**mul)(cf *, const cf *, const cf *, unsigned int);              /* volk_32fc_x2_multiply_32fc (a dispatcher pointer) */
    void (**smul)(cf *, const cf *, const cf, unsigned int);               /* volk_32fc_s32fc_multiply_32fc */
} V;

static void *open_first(const char *const *names)
{
    for (; *names; names++) { void *h = dlopen(*names, RTLD_NOW | RTLD_GLOBAL); if (h) return h; }
    return NULL;
}

/* 0 = both libraries found and every symbol resolved; otherwise a message in err */
int fdco_refequiv_probe(char *err, int errlen)
{
    static const char *const fn[] = {"libfftw3f.so.3", "libfftw3f.so", NULL};
    static const char *const ft[] = {"libfftw3f_threads.so.3", "libfftw3f_threads.so", "libfftw3f_omp.so.3", NULL};
    static const char *const vn[] = {"libvolk.so", "libvolk.so.3.1", "libvolk.so.3.0", "libvolk.so.2.5", "libvolk.so.2.4", "libvolk.so.2", "libvolk.so.1.4", "libvolk.so.1.3", NULL};
    if (!F.h) F.h = open_first(fn);
    if (!V.h) V.h = open_first(vn);
    if (!F.h || !V.h) {
        snprintf(err, (size_t)errlen, "%s%s%s not found by dlopen", F.h ? "" : "libfftw3f", (!F.h && !V.h) ? ", " : "", V.h ? "" : "libvolk");
        return 1;
    }
    if (!F.ht) F.ht = open_first(ft);                                     /* optional: without it the forward transform runs on one thread */
    F.plan_dft_1d = (fplan (*)(int, cf *, cf *, int, unsigned))dlsym(F.h, "fftwf_plan_dft_1d");
    F.execute_dft = (void (*)(const fplan, cf *, cf *))dlsym(F.h, "fftwf_execute_dft");
    F.destroy_plan = (void (*)(fplan))dlsym(F.h, "fftwf_destroy_plan");
    F.fmalloc = (void *(*)(size_t))dlsym(F.h, "fftwf_malloc");
    F.ffree = (void (*)(void *))dlsym(F.h, "fftwf_free");
    if (F.ht) {
        F.init_threads = (int (*)(void))dlsym(F.ht, "fftwf_init_threads");
        F.plan_with_nthreads = (void (*)(int))dlsym(F.ht, "fftwf_plan_with_nthreads");
    }
    V.mul = (void (**)(cf *, const cf *, const cf *, unsigned int))dlsym(V.h, "volk_32fc_x2_multiply_32fc");
    V.smul = (void (**)(cf *, const cf *, const cf, unsigned int))dlsym(V.h, "volk_32fc_s32fc_multiply_32fc");
    if (!F.plan_dft_1d || !F.execute_dft || !F.destroy_plan || !F.fmalloc || !F.ffree || !V.mul || !V.smul || !*V.mul || !*V.smul) {
        snprintf(err, (size_t)errlen, "libfftw3f / libvolk found but a symbol is missing");
        return 2;
    }
    return 0;
}

static double now_s(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

/* Times the chain on `nblocks` blocks of `x` (nblocks * (N - N/R) complex samples) until `budget` seconds have passed.
 * f, l: first bin and width of every channel; wins: all [R][l] window tables one behind the other, win_off[c] = start of
 * channel c's (complex elements).  fwd_threads: FFTW threads of the forward transform (the reference: 4).
 * Returns 0 and the input rate in *msps; the output of the last pass of channel 0 goes to out0 (nblocks * lout_0) if given. */
int fdco_refequiv_run(int N, int R, int nchan, const int *f, const int *l, const float *wins, const long *win_off, const float *x,
                      int nblocks, int fwd_threads, double budget, double *msps, int *passes, float *out0, char *err, int errlen)
{
    if (fdco_refequiv_probe(err, errlen)) return 1;
    const int ovl = N / R, H = N - ovl;
    const cf *xin = (const cf *)x;
    cf *blocks = F.fmalloc(sizeof(cf) * (size_t)nblocks * N), *spec = F.fmalloc(sizeof(cf) * (size_t)nblocks * N);
    cf *tmpN = F.fmalloc(sizeof(cf) * (size_t)N);
    if (!blocks || !spec || !tmpN) { snprintf(err, (size_t)errlen, "out of memory"); return 3; }
    if (F.init_threads && F.plan_with_nthreads) { F.init_threads(); F.plan_with_nthreads(fwd_threads > 0 ? fwd_threads : 1); }
    fplan pf = F.plan_dft_1d(N, blocks, tmpN, -1 /* FFTW_FORWARD */, 64 /* FFTW_ESTIMATE, like gr::fft::fft_complex's default wisdom-less plan */);
    if (F.plan_with_nthreads) F.plan_with_nthreads(1);
    fplan *pi = calloc((size_t)nchan, sizeof(fplan));
    int lmax = 0;
    for (int c = 0; c < nchan; c++) if (l[c] > lmax) lmax = l[c];
    cf *scratch = F.fmalloc(sizeof(cf) * 2 * (size_t)lmax);
    for (int c = 0; c < nchan; c++) {                                   /* one plan per distinct width would do; plans are cheap to share */
        int k = 0;
        for (; k < c; k++) if (l[k] == l[c]) break;
        pi[c] = k < c ? pi[k] : F.plan_dft_1d(l[c], scratch, scratch + lmax, +1 /* FFTW_BACKWARD */, 64);
    }
    const int nthr = omp_get_max_threads();
    cf **wk = calloc((size_t)nthr, sizeof(cf *));
    for (int t = 0; t < nthr; t++) wk[t] = F.fmalloc(sizeof(cf) * 3 * (size_t)lmax);
    cf **outs = calloc((size_t)nchan, sizeof(cf *));
    for (int c = 0; c < nchan; c++) outs[c] = F.fmalloc(sizeof(cf) * (size_t)nblocks * (size_t)(l[c] - l[c] / R));
    const cf invN = 1.0f / (float)N;
    int reps = 0;
    const double t0 = now_s();
    double dt;
    do {
        /* overlap_save: item m = history (zeros at stream start) + the new samples */
        for (int m = 0; m < nblocks; m++) {
            if (m == 0) memset(blocks, 0, sizeof(cf) * (size_t)ovl);
            else memcpy(blocks + (size_t)m * N, xin + (size_t)m * H - ovl, sizeof(cf) * (size_t)ovl);
            memcpy(blocks + (size_t)m * N + ovl, xin + (size_t)m * H, sizeof(cf) * (size_t)H);
        }
        /* fft_vcc(N, forward, shift): transform, then the halves of the output swapped; multiply_const_cc(1/N) */
        for (int m = 0; m < nblocks; m++) {
            F.execute_dft(pf, blocks + (size_t)m * N, tmpN);
            memcpy(spec + (size_t)m * N, tmpN + N / 2, sizeof(cf) * (size_t)(N / 2));
            memcpy(spec + (size_t)m * N + N / 2, tmpN, sizeof(cf) * (size_t)(N / 2));
            (*V.smul)(spec + (size_t)m * N, spec + (size_t)m * N, invN, (unsigned)N);
        }
        /* the channel branches */
#pragma omp parallel for schedule(dynamic, 1)
        for (int c = 0; c < nchan; c++) {
            cf *a = wk[omp_get_thread_num()], *b = a + lmax, *d = b + lmax;
            const int lc = l[c], lout = lc - lc / R, shift = ((f[c] % R) + R) % R;
            const cf *w = (const cf *)wins + win_off[c];
            int counter = 0;
            for (int m = 0; m < nblocks; m++) {
                memcpy(a, spec + (size_t)m * N + f[c], sizeof(cf) * (size_t)lc);                 /* vector_cut_vxx */
                (*V.mul)(b, a, w + (size_t)counter * lc, (unsigned)lc);                             /* phase_shifting_windowing_vcc */
                counter = (counter + shift) % R;
                memcpy(a, b + lc / 2, sizeof(cf) * (size_t)(lc / 2));                              /* fft_vcc(l, inverse, shift): input halves swapped */
                memcpy(a + lc / 2, b, sizeof(cf) * (size_t)(lc / 2));
                F.execute_dft(pi[c], a, d);
                (*V.smul)(outs[c] + (size_t)m * lout, d + (lc - lout), (cf)(float)lc, (unsigned)lout);   /* vector_cut_vxx + multiply_const_cc(l) */
            }
        }
        reps++;
        dt = now_s() - t0;
    } while (dt < budget && reps < 1000);
    *msps = (double)reps * nblocks * H / dt / 1e6;
    *passes = reps;
    if (out0) memcpy(out0, outs[0], sizeof(cf) * (size_t)nblocks * (size_t)(l[0] - l[0] / R));
    F.destroy_plan(pf);
    for (int c = 0; c < nchan; c++) { int k = 0; for (; k < c; k++) if (l[k] == l[c]) break; if (k == c) F.destroy_plan(pi[c]); }
    for (int c = 0; c < nchan; c++) F.ffree(outs[c]);
    for (int t = 0; t < nthr; t++) F.ffree(wk[t]);
    free(outs); free(wk); free(pi);
    F.ffree(scratch); F.ffree(blocks); F.ffree(spec); F.ffree(tmpN);
    return 0;
}
