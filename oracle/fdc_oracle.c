/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.  See fdc_oracle.h for scope, citations and pinning. */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "fdc_oracle.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define REAL double
#define SUF(x) d_##x
#include "fdc_fft_impl.inc"
#undef REAL
#undef SUF
#define REAL float
#define SUF(x) f_##x
#include "fdc_fft_impl.inc"
#undef REAL
#undef SUF

/* ---- python/FrequencyDomainChannelizer.py:37-40 ------------------------------------------- */
long fdco_nextpow2(double k)
{
    if (k < 1) return -1;
    return 1L << (int)ceil(log2(k));
}

/* ---- python/FrequencyDomainChannelizer.py:322-345 ----------------------------------------- */
int fdco_channel_params(int N, int R, double freq, double bw,
                        int *f, int *l, int *lout, double *pbw, double *sbw)
{
    double passsamps = (double)N * bw;                       /* :323 */
    long blocklen = fdco_nextpow2(passsamps);                /* :324 */
    if (blocklen < 0) return -1;
    if ((double)blocklen < 1.2 * passsamps) blocklen *= 2;   /* :326-327, 20% puffer */

    double passband = passsamps / (double)blocklen * 1.1;    /* :329 */
    double stopband = 1.0;
    if (passband >= 1.0) passband = 1.0;                     /* :331-332 */
    else if (passband < 0.7) stopband = passband + 0.25;     /* :333-334 */

    /* :336  int(round(freq*N)) % N — the reference is Python-2 code (python/__init__.py:29,34): its round()
     * goes half AWAY from zero = C round(); Python's % is a floored modulo.  Pinned by the rows of
     * tests/golden/channel_params.json, which come from running the reference's function (ties marked). */
    long fs = (long)round(freq * (double)N);
    fs = ((fs % N) + N) % N;
    double fsd = (double)fs - (double)blocklen / 2.0;        /* :337 (true division) */
    if (fsd < 0) fsd = fmod(fsd + (double)N, (double)N);     /* :338-339 */
    if (fsd + (double)blocklen > (double)N) fsd = (double)(N - blocklen); /* :340-341 */

    *f = (int)fsd;                                           /* :345 int() truncates */
    *l = (int)blocklen;
    *lout = (int)blocklen - (int)blocklen / R;               /* :343 */
    *pbw = passband;
    *sbw = stopband;
    return 0;
}

/* ---- lib/windows.h ------------------------------------------------------------------------- */
static void win_rect(int n, int lowsamps, int rampsamps, double *w, int normalize)
{   /* lib/windows.h:80-90 */
    double v = normalize ? 1.0 : 1.0 / (double)n;
    for (int i = 0; i < n; i++) w[i] = v;
    for (int i = 0; i < lowsamps + rampsamps / 2; i++) { w[i] = 0.0; w[n - 1 - i] = 0.0; }
}

static void win_ramp(int n, int lowsamps, int rampsamps, double *w, int normalize)
{   /* lib/windows.h:92-106 */
    double v = normalize ? 1.0 : 1.0 / (double)n;
    for (int i = 0; i < n; i++) w[i] = v;
    for (int i = 0; i < lowsamps; i++) { w[i] = 0.0; w[n - 1 - i] = 0.0; }
    for (int i = 0; i < rampsamps; i++) {
        w[lowsamps + i] = v * (double)(i + 1) / (double)(rampsamps + 1);
        w[n - lowsamps - 1 - i] = w[lowsamps + i];
    }
}

static void win_hann(int n, int lowsamps, int rampsamps, double *w, int normalize)
{   /* lib/windows.h:108-124 */
    double v = normalize ? 1.0 : 1.0 / (double)n;
    for (int i = 0; i < n; i++) w[i] = v;
    for (int i = 0; i < lowsamps; i++) { w[i] = 0.0; w[n - 1 - i] = 0.0; }
    for (int i = 0; i < rampsamps; i++) {
        double phi = (double)(i + 1) / (double)(rampsamps + 1) * M_PI;
        w[lowsamps + i] = v * (-cos(phi) / 2.0 + 0.5);
        w[n - lowsamps - 1 - i] = w[lowsamps + i];
    }
}

void fdco_window(int wintype, int blocksize, float passbw, float stopbw, int R, int step,
                 int normalize, float *w)
{
    /* lib/windows.h:41-55 (float arguments promoted to double exactly as the C++ does) */
    if (passbw >= 1.0) { passbw = 1.0f; stopbw = 1.0f; wintype = 0; }
    else if (stopbw >= 1.0) stopbw = 1.0f;
    int lowsamps = (int)((1.0 - stopbw) * (double)blocksize) / 2;
    int highsamps = (int)(passbw * (double)blocksize);
    int rampsamps = (blocksize - 2 * lowsamps - highsamps) / 2;

    /* lib/windows.h:57-78 */
    step = step % R;
    double *wd = (double *)malloc(sizeof(double) * (size_t)blocksize);
    if (wintype == 1) win_hann(blocksize, lowsamps, rampsamps, wd, normalize);
    else if (wintype == 2) win_ramp(blocksize, lowsamps, rampsamps, wd, normalize);
    else win_rect(blocksize, lowsamps, rampsamps, wd, normalize);
    int count = 0;
    for (int i = 0; i < R; i++) {
        double phi = 2.0 * M_PI * (double)count / (double)R;
        for (int k = 0; k < blocksize; k++) {
            /* std::polar(rho, theta) = (rho*cos(theta), rho*sin(theta)), then cast to float */
            w[2 * ((size_t)i * blocksize + k)]     = (float)(wd[k] * cos(phi));
            w[2 * ((size_t)i * blocksize + k) + 1] = (float)(wd[k] * sin(phi));
        }
        count = (count + step) % R;
    }
    free(wd);
}

/* ---- lib/overlap_save_impl.cc:62-81 ---------------------------------------------------------- */
void fdco_overlap_save(int itemsize, int outlen, int ovl, unsigned char *hist,
                       const void *in_, int nitems, void *out_)
{
    const unsigned char *in = (const unsigned char *)in_;
    unsigned char *out = (unsigned char *)out_;
    const size_t isz = (size_t)itemsize;
    const size_t inplen = (size_t)(outlen - ovl);
    if (nitems <= 0) return;
    memcpy(out, hist, isz * ovl);
    memcpy(out + isz * ovl, in, isz * inplen);
    for (int i = 1; i < nitems; i++) {
        memcpy(out + isz * i * outlen, in + isz * (i * inplen - ovl), isz * ovl);
        memcpy(out + isz * i * outlen + isz * ovl, in + isz * i * inplen, isz * inplen);
    }
    memcpy(hist, in + isz * (nitems * inplen - ovl), isz * ovl);
}

/* ---- lib/vector_cut_vxx_impl.cc:59-72 -------------------------------------------------------- */
void fdco_vector_cut(int itemsize, int veclen, int offset, int blocklen,
                     const void *in_, int nitems, void *out_)
{
    const unsigned char *in = (const unsigned char *)in_;
    unsigned char *out = (unsigned char *)out_;
    const size_t inplen = (size_t)itemsize * veclen, outplen = (size_t)itemsize * blocklen;
    const size_t shift = (size_t)offset * itemsize;
    for (int i = 0; i < nitems; i++) memcpy(out + i * outplen, in + i * inplen + shift, outplen);
}

/* ---- lib/phase_shifting_windowing_vcc_impl.cc:72-86 ------------------------------------------ */
static inline void cmul_f32(const float *a, const float *b, float *o)
{   /* volk_32fc_x2_multiply_32fc generic kernel: float complex product, no FMA contraction */
    float re = a[0] * b[0] - a[1] * b[1];
    float im = a[0] * b[1] + a[1] * b[0];
    o[0] = re; o[1] = im;
}

void fdco_phase_window(int l, int R, int shift, int *counter, const float *win,
                       const float *in, int nitems, float *out)
{
    for (int i = 0; i < nitems; i++) {
        const float *w = win + 2 * (size_t)(*counter) * l;
        for (int k = 0; k < l; k++)
            cmul_f32(in + 2 * ((size_t)i * l + k), w + 2 * k, out + 2 * ((size_t)i * l + k));
        *counter = (*counter + shift) % R;
    }
}

/* ---- fft_vcc semantics (python/FrequencyDomainChannelizer.py:206,228) ------------------------ */
void fdco_fft_vcc(int n, int forward, int shift, const float *in, int nitems, float *out)
{
    d_plan *p = d_plan_create(n, forward);
    const int h = n / 2;
    for (int it = 0; it < nitems; it++) {
        const float *x = in + 2 * (size_t)it * n;
        float *y = out + 2 * (size_t)it * n;
        for (int i = 0; i < n; i++) {
            /* inverse + shift: halves of the INPUT swapped before the transform */
            int s = (!forward && shift) ? (i + h) % n : i;
            p->a[i].re = x[2 * s]; p->a[i].im = x[2 * s + 1];
        }
        d_cpx *r = d_execute(p);
        for (int i = 0; i < n; i++) {
            /* forward + shift: halves of the OUTPUT swapped after the transform */
            int s = (forward && shift) ? (i + h) % n : i;
            y[2 * i] = (float)r[s].re; y[2 * i + 1] = (float)r[s].im;
        }
    }
    d_plan_destroy(p);
}

/* ---- full throughput chain -------------------------------------------------------------------- */
typedef struct { d_plan *d; f_plan *f; } any_plan;

static void any_exec(any_plan *p, int use_float, const float *in, int n, int rot_in, int rot_out,
                     float *out)
{
    /* in/out interleaved float; rot_in/rot_out are index rotations (0 or n/2) */
    if (use_float) {
        for (int i = 0; i < n; i++) {
            int s = (i + rot_in) % n;
            p->f->a[i].re = in[2 * s]; p->f->a[i].im = in[2 * s + 1];
        }
        f_cpx *r = f_execute(p->f);
        for (int i = 0; i < n; i++) {
            int s = (i + rot_out) % n;
            out[2 * i] = r[s].re; out[2 * i + 1] = r[s].im;
        }
    } else {
        for (int i = 0; i < n; i++) {
            int s = (i + rot_in) % n;
            p->d->a[i].re = in[2 * s]; p->d->a[i].im = in[2 * s + 1];
        }
        d_cpx *r = d_execute(p->d);
        for (int i = 0; i < n; i++) {
            int s = (i + rot_out) % n;
            out[2 * i] = (float)r[s].re; out[2 * i + 1] = (float)r[s].im;
        }
    }
}

static any_plan any_create(int n, int forward, int use_float)
{
    any_plan p = { 0, 0 };
    if (use_float) p.f = f_plan_create(n, forward); else p.d = d_plan_create(n, forward);
    return p;
}
static void any_destroy(any_plan *p) { d_plan_destroy(p->d); f_plan_destroy(p->f); }

int fdco_channelizer(int N, int R, int wintype, int C, const int *f, const int *l,
                     const float *pbw, const float *sbw, long first_block,
                     const float *prefix, const float *x, int nblocks,
                     float **out, float *spectrum, int use_float, int nthreads)
{
    const int ovl = N / R, H = N - ovl;
    if (nthreads < 1) nthreads = 1;
    /* window tables, one per channel: phase_shifting_windowing_vcc ctor (…_impl.cc:62) */
    float **win = (float **)calloc((size_t)(C > 0 ? C : 1), sizeof(float *));
    int maxl = 1;
    for (int c = 0; c < C; c++) {
        win[c] = (float *)malloc(sizeof(float) * 2 * (size_t)R * l[c]);
        fdco_window(wintype, l[c], pbw[c], sbw[c], R, 1, 0, win[c]);
        if (l[c] > maxl) maxl = l[c];
    }
    /* distinct IFFT lengths */
    int nlen = 0, lens[32];
    for (int c = 0; c < C; c++) {
        int k; for (k = 0; k < nlen; k++) if (lens[k] == l[c]) break;
        if (k == nlen && nlen < 32) lens[nlen++] = l[c];
    }
    int err = 0;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
    {
        any_plan fwd = any_create(N, 1, use_float);
        any_plan inv[32];
        for (int k = 0; k < nlen; k++) inv[k] = any_create(lens[k], 0, use_float);
        float *blk = (float *)malloc(sizeof(float) * 2 * (size_t)N);
        float *spec = (float *)malloc(sizeof(float) * 2 * (size_t)N);
        float *y = (float *)malloc(sizeof(float) * 2 * (size_t)maxl);
        float *z = (float *)malloc(sizeof(float) * 2 * (size_t)maxl);
        const float invN = 1.0f / (float)N;   /* multiply_const_cc(1/N), py:214-216 */
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int m = 0; m < nblocks; m++) {
            /* A.1 overlap-save blocking (lib/overlap_save_impl.cc:70-78) */
            for (int i = 0; i < N; i++) {
                long s = (long)m * H - ovl + i;
                if (s >= 0) { blk[2 * i] = x[2 * s]; blk[2 * i + 1] = x[2 * s + 1]; }
                else if (prefix) { blk[2 * i] = prefix[2 * (ovl + s)]; blk[2 * i + 1] = prefix[2 * (ovl + s) + 1]; }
                else { blk[2 * i] = 0.f; blk[2 * i + 1] = 0.f; }
            }
            /* A.2 forward fft_vcc(shift=True) then * 1/N, rounded to float at each stage */
            any_exec(&fwd, use_float, blk, N, 0, N / 2, spec);
            for (int i = 0; i < 2 * N; i++) spec[i] = spec[i] * invN;
            if (spectrum) memcpy(spectrum + 2 * (size_t)m * N, spec, sizeof(float) * 2 * (size_t)N);
            /* A.4 per channel */
            for (int c = 0; c < C; c++) {
                const int lc = l[c], lo = lc - lc / R;
                const int shift = ((f[c] % R) + R) % R;           /* …windowing_vcc_impl.cc:58 */
                const int counter = (int)((((first_block + m) % R) * shift) % R); /* closed form of :82 */
                const float *w = win[c] + 2 * (size_t)counter * lc;
                for (int i = 0; i < lc; i++) cmul_f32(spec + 2 * (size_t)(f[c] + i), w + 2 * i, y + 2 * i);
                int k; for (k = 0; k < nlen; k++) if (lens[k] == lc) break;
                any_exec(&inv[k], use_float, y, lc, lc / 2, 0, z);   /* fft_vcc(l, False, shift) py:228 */
                float *o = out[c] + 2 * (size_t)m * lo;
                const float scale = (float)lc;                      /* multiply_const_cc(N/dec) py:231 */
                for (int t = 0; t < lo; t++) {                      /* vector_cut(l, l-lout, lout) py:229 */
                    o[2 * t] = z[2 * (lc - lo + t)] * scale;
                    o[2 * t + 1] = z[2 * (lc - lo + t) + 1] * scale;
                }
            }
        }
        free(blk); free(spec); free(y); free(z);
        any_destroy(&fwd);
        for (int k = 0; k < nlen; k++) any_destroy(&inv[k]);
    }
    for (int c = 0; c < C; c++) free(win[c]);
    free(win);
    return err;
}
