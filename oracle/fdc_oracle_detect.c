/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * CPU restatements of the two stateful sink blocks that consume the normalised spectrum:
 *   PowerActivationChannel               lib/PowerActivationChannel_impl.{h,cc}
 *   activity_detection_channelizer_vcm   lib/activity_detection_channelizer_vcm_impl.{h,cc}
 * Each function cites the lines it follows.  PDUs (pmt::cons(dict, c32vector)) are returned as POD records; the
 * timestamp part of the ID strings (…_impl.cc get_current_time) is not modelled.
 * The inverse transforms use the oracle's double-precision FFT (fdc_oracle.c), rounded to float once, exactly like
 * the throughput chain.  Pinning: the reference holds no tests for these blocks; the only reference outputs available
 * are the PDU metadata recorded in SURVEY.md §8c (tests/golden/sink_known_answers.json).
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "fdc_oracle.h"

/* ---------------------------------------------------------------- PDU list */
static void pdu_push(fdco_pdu_list *L, const fdco_pdu *p, const float *samples, long nsamples)
{
    if (L->n == L->cap) {
        L->cap = L->cap ? 2 * L->cap : 16;
        L->pdu = (fdco_pdu *)realloc(L->pdu, sizeof(fdco_pdu) * (size_t)L->cap);
    }
    fdco_pdu *q = &L->pdu[L->n++];
    *q = *p;
    q->nsamples = nsamples;
    q->samples = (float *)malloc(sizeof(float) * 2 * (size_t)(nsamples > 0 ? nsamples : 1));
    if (nsamples > 0) memcpy(q->samples, samples, sizeof(float) * 2 * (size_t)nsamples);
}

void fdco_pdu_list_clear(fdco_pdu_list *L)
{
    for (int i = 0; i < L->n; i++) free(L->pdu[i].samples);
    free(L->pdu);
    L->pdu = 0; L->n = 0; L->cap = 0;
}

static int ipow2ceil(int k)     /* (int) pow(2, ceil(log2((double)k))) — PowerActivationChannel_impl.cc:396-401 */
{
    return (int)pow(2.0, ceil(log2((double)k)));
}

/* extraction shared by both blocks: PowerActivationChannel_impl.cc:260-284, …vcm_impl.cc:373-397 */
static void extract_block(const float *sig, int start, int w, const float *win /* w complex */, int skip,
                          float *out /* (w-skip) complex */)
{
    float *x = (float *)malloc(sizeof(float) * 2 * (size_t)w);
    float *sh = (float *)malloc(sizeof(float) * 2 * (size_t)w);
    float *y = (float *)malloc(sizeof(float) * 2 * (size_t)w);
    for (int i = 0; i < w; i++) {       /* volk_32fc_x2_multiply_32fc */
        const float ar = sig[2 * (start + i)], ai = sig[2 * (start + i) + 1], br = win[2 * i], bi = win[2 * i + 1];
        x[2 * i] = ar * br - ai * bi; x[2 * i + 1] = ar * bi + ai * br;
    }
    /* fftshift(): halves swapped; an odd size leaves the last element untouched (…cc:413-428 / :532-540) */
    int sz = w; if (sz % 2) sz -= 1;
    memcpy(sh, x, sizeof(float) * 2 * (size_t)w);
    if (sz > 0) {
        const int h = sz / 2;
        memcpy(sh, x + 2 * h, sizeof(float) * 2 * (size_t)h);
        memcpy(sh + 2 * h, x, sizeof(float) * 2 * (size_t)h);
    }
    if (w >= 2) fdco_fft_vcc(w, 0, 0, sh, 1, y);   /* gr::fft::fft_complex(w, false, 1): unnormalised backward DFT */
    else memcpy(y, sh, sizeof(float) * 2);
    memcpy(out, y + 2 * skip, sizeof(float) * 2 * (size_t)(w - skip));
    free(x); free(sh); free(y);
}

/* ================================================================= PowerActivationChannel */
struct fdco_pac {
    int blocklen, relinvovl, extract_start, extract_stop, extract_width, output_len, output_ovl_offset;
    int measure_start, measure_stop, maxblocks, deactivation_delay, ID;
    float thresh, lastpower;
    int active, count, phase, deltaphase, part, finished_channels, blockcount, msg_finished_index;
    float *windows;   /* [R][blocklen] */
    float *hist;      /* blocklen */
    float *blocks; long nblk, capblk;   /* buffered output blocks, output_len complex each */
};

/* lib/PowerActivationChannel_impl.cc:314-355 (set_startstop) + :357-375 (cr_windows) + :41-133 (ctor) */
fdco_pac *fdco_pac_create(int blocklen, float cfreq, float bw, int relinvovl, float thresh_db, int maxblocks,
                          int deactivation_delay, int ID)
{
    if (blocklen <= 0) return 0;                                              /* :64-65 */
    if (relinvovl <= 0 || relinvovl != ipow2ceil(relinvovl)) return 0;        /* :68-69 */
    bw = bw > 0.0f ? bw : -bw;                                                 /* :315 */
    if (bw > 1.0 || cfreq - bw / 2.0f < 0.0f || cfreq + bw / 2.0f > 1.0f) return 0;   /* :318-319 */
    if (thresh_db <= 0.0f) return 0;                                           /* :378-379 */
    fdco_pac *p = (fdco_pac *)calloc(1, sizeof(*p));
    p->blocklen = blocklen; p->relinvovl = relinvovl; p->ID = ID;
    int k = (int)ceil((double)bw * (double)blocklen);
    if (k <= 0) { free(p); return 0; }                                         /* nextpow2 throws, :397-399 */
    p->extract_width = ipow2ceil(k);                                           /* :322 */
    if (p->extract_width > blocklen) p->extract_width = blocklen;
    const int mid = (int)round((double)cfreq * (double)blocklen);              /* :326 */
    p->extract_start = mid - p->extract_width / 2;
    if (p->extract_start < 0) p->extract_start = 0;
    p->extract_stop = p->extract_start + p->extract_width;
    if (p->extract_stop > blocklen) {                                          /* :333-336, bug kept (App. B.2) */
        p->extract_stop = blocklen;
        p->extract_start = p->extract_stop - blocklen;
    }
    p->measure_start = (int)round((double)(cfreq - bw / 2.0f) * (double)blocklen);   /* :338-339 */
    p->measure_stop = (int)round((double)(cfreq + bw / 2.0f) * (double)blocklen);
    if (p->measure_start < p->extract_start) p->measure_start = p->extract_start;
    if (p->measure_stop > p->extract_stop) p->measure_stop = p->extract_stop;
    /* cr_windows, :357-375: tables of the FULL block length, float polar, rising sine edge mirrored to the far end */
    int rampsamps = (p->extract_stop - p->extract_start) - (p->measure_stop - p->measure_start);
    rampsamps /= 3;
    p->windows = (float *)malloc(sizeof(float) * 2 * (size_t)relinvovl * blocklen);
    for (int i = 0; i < relinvovl; i++) {
        const float ang = (float)(2.0f * M_PI * (double)i / (double)relinvovl);
        const float cr = cosf(ang), ci = sinf(ang);                            /* std::polar(1.0f, ang) */
        float *w = p->windows + 2 * (size_t)i * blocklen;
        for (int n = 0; n < blocklen; n++) { w[2 * n] = 1.0f * cr; w[2 * n + 1] = 1.0f * ci; }
    }
    for (int i = 0; i < rampsamps; i++)
        for (int r = 0; r < relinvovl; r++) {
            float *w = p->windows + 2 * (size_t)r * blocklen;
            const float f = (float)sin(0.5 * M_PI * (double)i / (double)(rampsamps + 1));
            w[2 * i] *= f; w[2 * i + 1] *= f;
            w[2 * (blocklen - i - 1)] = w[2 * i]; w[2 * (blocklen - i - 1) + 1] = w[2 * i + 1];
        }
    p->deltaphase = p->extract_start % relinvovl;                              /* :349 */
    p->phase = 0;
    p->output_ovl_offset = p->extract_width / relinvovl;
    p->output_len = p->extract_width - p->output_ovl_offset;
    p->thresh = (float)pow(10.0, (double)thresh_db / 10.0);                    /* :380 */
    p->maxblocks = maxblocks;
    p->deactivation_delay = deactivation_delay <= 0 ? 0 : deactivation_delay;  /* :83-86 (stored, never used: B.7) */
    p->hist = (float *)calloc(2 * (size_t)blocklen, sizeof(float));            /* :89 */
    p->lastpower = FLT_MAX;                                                    /* :92 */
    p->active = 0; p->blockcount = 1;                                          /* :94-96 */
    return p;
}

void fdco_pac_destroy(fdco_pac *p)
{
    if (!p) return;
    free(p->windows); free(p->hist); free(p->blocks); free(p);
}

void fdco_pac_params(const fdco_pac *p, int *v /* 8 ints */)
{
    v[0] = p->extract_start; v[1] = p->extract_stop; v[2] = p->extract_width; v[3] = p->measure_start;
    v[4] = p->measure_stop; v[5] = p->output_len; v[6] = p->output_ovl_offset; v[7] = p->deltaphase;
}

static void pac_process(fdco_pac *p, const float *sig)      /* :260-284 */
{
    if (p->nblk == p->capblk) {
        p->capblk = p->capblk ? 2 * p->capblk : 8;
        p->blocks = (float *)realloc(p->blocks, sizeof(float) * 2 * (size_t)p->capblk * p->output_len);
    }
    extract_block(sig, p->extract_start, p->extract_width, p->windows + 2 * (size_t)p->phase * p->blocklen,
                  p->output_ovl_offset, p->blocks + 2 * (size_t)p->nblk * p->output_len);
    p->nblk++;
    p->count++;
    p->phase = (p->phase + p->deltaphase) % p->relinvovl;
}

static void pac_emit(fdco_pac *p, int fin, fdco_pdu_list *L)    /* :212-258 */
{
    fdco_pdu d;
    memset(&d, 0, sizeof d);
    d.kind = 0; d.source = p->ID; d.chan_id = p->msg_finished_index;
    d.finalized = fin; d.part = p->part; d.has_part = 1;
    d.rel_cfreq = (double)(p->extract_start + p->extract_stop) / 2.0 / (double)p->blocklen;
    d.rel_bw = (double)p->extract_width / (double)p->blocklen;
    d.blockstart = p->blockcount - p->count; d.blockend = p->blockcount;
    d.vectorstart = p->extract_start; d.vectorend = p->extract_stop;   /* not in the PAC dict; kept for the log line */
    pdu_push(L, &d, p->blocks, p->nblk * p->output_len);
    p->nblk = 0;
    p->part++;
}

static int pac_measure(fdco_pac *p, const float *in)             /* :286-306 */
{
    float pwr = 0.0f;
    for (int i = p->measure_start; i < p->measure_stop; i++) {
        /* std::real(in[i] * std::conj(in[i])) : re*re - im*(-im) */
        const float re = in[2 * i], im = in[2 * i + 1];
        pwr += re * re - im * (-im);
    }
    if (pwr == 0.0f) pwr = FLT_MIN;
    if (!p->active && pwr / p->lastpower >= p->thresh) { p->lastpower = pwr; return 1; }
    else if (p->active && p->lastpower / pwr >= p->thresh) { p->lastpower = pwr; return 1; }
    p->lastpower = pwr;
    return 0;
}

void fdco_pac_work(fdco_pac *p, const float *in, int nitems, fdco_pdu_list *L)    /* :137-177 */
{
    if (nitems <= 0) return;
    const float *sighist = p->hist, *sig = in;
    for (int i = 0; i < nitems; i++) {
        sig = in + 2 * (size_t)i * p->blocklen;
        if (pac_measure(p, sig)) {
            if (!p->active) {                                   /* activate(), :198-210 */
                p->part = 0; p->count = 0; p->active = 1; p->phase = 0; p->nblk = 0;
                p->msg_finished_index = p->finished_channels;   /* create_ID(), :308-312 */
                pac_process(p, sighist);
                pac_process(p, sig);
            } else {
                pac_process(p, sig);
                p->active = 0;                                  /* deactivate(), :189-196 */
                pac_emit(p, 1, L);
                p->finished_channels++;
            }
        } else if (p->active) {
            pac_process(p, sig);
            if (p->maxblocks == 0 || (p->maxblocks > 0 && p->count % p->maxblocks == 0)) pac_emit(p, 0, L);
        }
        sighist = sig;
        p->blockcount++;
    }
    memcpy(p->hist, sig, sizeof(float) * 2 * (size_t)p->blocklen);   /* save_hist, :173,185-187 */
}

/* ================================================================= activity_detection_channelizer_vcm */
typedef struct {
    int ID, detect_start, detect_stop, extract_start, extract_stop, extract_width, extract_window, ovlskip,
        outputsamples, count, phase, phaseincrement, inactive, part;
    float *data; long nblk, capblk;
} vcm_chan;

typedef struct {
    int ID, active_channels_counter, start, stop, width;
    float *power; int npower;
    vcm_chan *ch; int nch, capch;
} vcm_seg;

struct fdco_vcm {
    int variant;           /* 0 = activity_detection_channelizer_vcm, 1 = SegmentDetection (lib/SegmentDetection_impl.cc) */
    int blocklen, relinvovl, maxblocks, delay, dec;
    float thresh; double puffer;
    vcm_seg *seg; int nseg;
    float **windows;       /* [log2 width][R][width] */
    int nwin;
    float *hist;
    unsigned blockcount;
};

static int ilog2d(double v) { return (int)log2(v); }

static float sd_mod_f(float x, float y)      /* SegmentDetection_impl.cc:700-703 */
{
    return (float)fmod(fmod((double)x, (double)y) + 1.0, (double)y);
}

/* SegmentDetection::make(ID, blocklen, relinvovl, seg_start, seg_stop, thresh, minchandist, window_flank_puffer,
 * maxblocks_to_emit, channel_deactivation_delay, …) — the single-segment twin the hier block instantiates
 * (python/FrequencyDomainChannelizer.py:261-278).  Differences to the vcm block: segment geometry
 * (lib/SegmentDetection_impl.cc:592-637), raw power sums and a plain divide (:178-193, :206), partial emission in a
 * separate pass (:346-365), block counter starting at 0 (:118), the segment ID given by the caller. */
fdco_vcm *fdco_sd_create(int ID, int blocklen, int relinvovl, float seg_start, float seg_stop, float thresh_db,
                         float minchandist, float window_flank_puffer, int maxblocks, int deactivation_delay)
{
    if (blocklen < 1 || blocklen != ipow2ceil(blocklen)) return 0;             /* :68-69 */
    if (relinvovl < 1 || relinvovl != ipow2ceil(relinvovl)) return 0;           /* :72-73 */
    if (thresh_db < 0.0f) return 0;                                             /* :75-76 */
    if ((double)window_flank_puffer < 0.0) return 0;                            /* :85-86 */
    minchandist = sd_mod_f(minchandist, 1.0f);                                  /* :594-596 */
    float start = sd_mod_f(seg_start, 1.0f), stop = sd_mod_f(seg_stop, 1.0f);
    if (start == stop) return 0;                                                /* :598-599 */
    if (start > stop) { float t = start; start = stop; stop = t; }
    fdco_vcm *v = (fdco_vcm *)calloc(1, sizeof(*v));
    v->variant = 1;
    v->blocklen = blocklen; v->relinvovl = relinvovl; v->maxblocks = maxblocks; v->delay = deactivation_delay;
    v->puffer = (double)window_flank_puffer;
    v->hist = (float *)calloc(2 * (size_t)blocklen, sizeof(float));
    const double dec = (double)blocklen * (double)minchandist / 2.0;            /* :608-614 */
    v->dec = dec < 2.0 ? 1 : (int)dec;
    v->thresh = (float)pow(10.0, (double)thresh_db / 10.0);
    size_t width = (size_t)((double)(stop - start) * (double)blocklen);         /* :617-621 */
    if (width % (size_t)v->dec) width += (size_t)v->dec - width % (size_t)v->dec;
    if (width > (size_t)blocklen) width = (size_t)(blocklen - (blocklen % v->dec));
    const size_t mid = (size_t)((double)(0.5f * (start + stop)) * (double)blocklen);   /* :626 */
    size_t dstart = mid < width / 2 ? 0 : mid - width / 2, dstop = dstart + width;
    if (dstop > (size_t)blocklen) { dstop = (size_t)blocklen; dstart = dstop - (size_t)blocklen; }   /* :629-632 (App. B.2) */
    v->seg = (vcm_seg *)calloc(1, sizeof(vcm_seg));
    v->nseg = 1;
    v->seg[0].ID = ID; v->seg[0].start = (int)dstart; v->seg[0].stop = (int)dstop; v->seg[0].width = (int)width;
    v->seg[0].npower = (int)width / v->dec;
    v->seg[0].power = (float *)calloc((size_t)(v->seg[0].npower > 0 ? v->seg[0].npower : 1), sizeof(float));
    if (dstart + width > (size_t)blocklen) { fdco_vcm_destroy(v); return 0; }   /* the reference would read past the block */
    /* cr_windows, :551-583 — identical to the vcm tables */
    v->nwin = ilog2d((double)blocklen) + 1;
    v->windows = (float **)calloc((size_t)v->nwin, sizeof(float *));
    for (int k = 0; k < v->nwin; k++) {
        const int ww = 1 << k;
        const int puffersamples = (int)(v->puffer * (double)ww);
        v->windows[k] = (float *)malloc(sizeof(float) * 2 * (size_t)relinvovl * ww);
        for (int i = 0; i < relinvovl; i++) {
            float *w = v->windows[k] + 2 * (size_t)i * ww;
            const double ang = 2.0 * M_PI * (double)i / (double)relinvovl;
            const float cr = (float)(1.0 * cos(ang)), ci = (float)(1.0 * sin(ang));
            for (int n = 0; n < ww; n++) { w[2 * n] = cr; w[2 * n + 1] = ci; }
            for (int q = 0; q < puffersamples; q++) {
                const float flank = 0.5f - 0.5f * (float)cos(M_PI * (double)q / (double)puffersamples);
                w[2 * q] *= flank; w[2 * q + 1] *= flank;
                w[2 * (ww - 1 - q)] *= flank; w[2 * (ww - 1 - q) + 1] *= flank;
            }
        }
    }
    v->blockcount = 0;                                                          /* :118 */
    return v;
}

fdco_vcm *fdco_vcm_create(int blocklen, int nseg, const float *segs /* nseg pairs */, float thresh_db, int relinvovl,
                          int maxblocks, float minchandist, int deactivation_delay, double window_flank_puffer)
{
    /* ctor, …vcm_impl.cc:82-190 */
    if (blocklen < 2 || blocklen != ipow2ceil(blocklen)) return 0;              /* :106-107 */
    if (minchandist <= 0.0f || minchandist >= 1.0) return 0;                    /* :231-232 */
    if (thresh_db < 0.0f) return 0;                                             /* :117-118 */
    if (relinvovl < 1 || relinvovl != ipow2ceil(relinvovl)) return 0;           /* :122-123 */
    if (deactivation_delay < 0) return 0;                                       /* :134-135 */
    if (window_flank_puffer < 0.0) return 0;                                    /* :139-140 */
    fdco_vcm *v = (fdco_vcm *)calloc(1, sizeof(*v));
    v->blocklen = blocklen; v->relinvovl = relinvovl; v->maxblocks = maxblocks; v->delay = deactivation_delay;
    v->puffer = window_flank_puffer;
    v->hist = (float *)calloc(2 * (size_t)blocklen, sizeof(float));
    const double dec = (double)blocklen * (double)minchandist / 2.0;            /* :234-240 */
    v->dec = dec < 2.0 ? 1 : (int)dec;
    v->thresh = (float)pow(10.0, (double)thresh_db / 10.0);                     /* :119 */
    v->seg = (vcm_seg *)calloc((size_t)(nseg > 0 ? nseg : 1), sizeof(vcm_seg));
    for (int s = 0; s < nseg; s++) {                                            /* create_segment, :248-279 */
        const float v0 = segs[2 * s], v1 = segs[2 * s + 1];
        if (v0 >= v1 || v0 < 0.0f || v1 > 1.0f) { fdco_vcm_destroy(v); return 0; }
        int mid = abs((int)round(((double)v1 + (double)v0) * 0.5 * (double)blocklen));
        int width = abs((int)round(((double)v1 - (double)v0) * (double)blocklen));
        width = (width % v->dec == 0) ? width : width + v->dec - width % v->dec;
        if (width >= blocklen) {
            if (blocklen % v->dec == 0) { fdco_vcm_destroy(v); return 0; }     /* the reference loops forever here */
            width = blocklen - (blocklen % v->dec);
        }
        int start = mid - width / 2 <= 0 ? 0 : mid - width / 2;
        int stop = start + width;
        if (stop > blocklen) { stop = blocklen; start = blocklen - width; }
        if (start < 0 || stop > blocklen) { fdco_vcm_destroy(v); return 0; }
        vcm_seg *g = &v->seg[v->nseg];
        g->ID = v->nseg; g->start = start; g->stop = stop; g->width = stop - start;
        g->npower = g->width / v->dec;
        g->power = (float *)calloc((size_t)(g->npower > 0 ? g->npower : 1), sizeof(float));
        v->nseg++;
    }
    /* cr_windows, :199-228 */
    v->nwin = ilog2d((double)blocklen) + 1;
    v->windows = (float **)calloc((size_t)v->nwin, sizeof(float *));
    for (int k = 0; k < v->nwin; k++) {
        const int ww = 1 << k;
        const int puffersamples = (int)(window_flank_puffer * (double)ww);
        v->windows[k] = (float *)malloc(sizeof(float) * 2 * (size_t)relinvovl * ww);
        for (int i = 0; i < relinvovl; i++) {
            float *w = v->windows[k] + 2 * (size_t)i * ww;
            const double ang = 2.0 * M_PI * (double)i / (double)relinvovl;
            const float cr = (float)(1.0 * cos(ang)), ci = (float)(1.0 * sin(ang));   /* gr_complex(std::polar(1.0, ang)) */
            for (int n = 0; n < ww; n++) { w[2 * n] = cr; w[2 * n + 1] = ci; }
            for (int q = 0; q < puffersamples; q++) {
                const float flank = 0.5f - 0.5f * (float)cos(M_PI * (double)q / (double)puffersamples);
                w[2 * q] *= flank; w[2 * q + 1] *= flank;
                w[2 * (ww - 1 - q)] *= flank; w[2 * (ww - 1 - q) + 1] *= flank;
            }
        }
    }
    v->blockcount = 1;                                                          /* :188 */
    return v;
}

void fdco_vcm_destroy(fdco_vcm *v)
{
    if (!v) return;
    for (int s = 0; s < v->nseg; s++) {
        for (int c = 0; c < v->seg[s].nch; c++) free(v->seg[s].ch[c].data);
        free(v->seg[s].ch); free(v->seg[s].power);
    }
    free(v->seg);
    if (v->windows) for (int k = 0; k < v->nwin; k++) free(v->windows[k]);
    free(v->windows); free(v->hist); free(v);
}

int fdco_vcm_segment_params(const fdco_vcm *v, int s, int *out /* start, stop, width, dec, npower */)
{
    if (s < 0 || s >= v->nseg) return -1;
    out[0] = v->seg[s].start; out[1] = v->seg[s].stop; out[2] = v->seg[s].width; out[3] = v->dec; out[4] = v->seg[s].npower;
    return 0;
}

typedef struct { float r; int pos; int ord; } edge;
static int edge_cmp(const void *a, const void *b)   /* fipair_sort: descending ratio (ties: std::sort is unstable) */
{
    const edge *x = (const edge *)a, *y = (const edge *)b;
    if (x->r > y->r) return -1;
    if (x->r < y->r) return 1;
    return x->ord - y->ord;
}

static void seg_detect(fdco_vcm *v, vcm_seg *g, const float *in)
{
    const int dec = v->dec, N = g->width / dec;
    /* measure_power, :630-650 */
    const float normfact = 1.0f / (float)dec;
    for (int i = 0; i < N; i++) {
        const int L = g->start + i * dec;
        float t = 0.0f;
        for (int k = 0; k < dec; k++) {
            const float re = in[2 * (L + k)], im = in[2 * (L + k) + 1];
            t += re * re - im * (-im);
        }
        /* SegmentDetection: volk_32fc_magnitude_squared_32f + volk_32f_accumulator_s32f, no normalisation (:185-190) */
        g->power[i] = v->variant == 1 ? t : t * normfact;
    }
    /* get_active_channels, :694-739 */
    edge *rise = (edge *)malloc(sizeof(edge) * (size_t)(N > 0 ? N : 1));
    int *fall = (int *)malloc(sizeof(int) * (size_t)(N > 0 ? N : 1));
    int nrise = 0, nfall = 0;
    const float inversethresh = 1.0f / v->thresh;
    for (int i = 1; i < N; i++) {
        float pd;
        if (v->variant == 1) pd = g->power[i] / g->power[i - 1];       /* volk_32f_x2_divide_32f, no zero guard (:206) */
        else if (g->power[i - 1] == 0.0f) pd = g->power[i] / FLT_MIN; else pd = g->power[i] / g->power[i - 1];
        if (pd > v->thresh) { rise[nrise].r = pd; rise[nrise].pos = (i - 1) * dec + g->start; rise[nrise].ord = nrise; nrise++; }
        else if (v->variant == 1) { if (pd < inversethresh) fall[nfall++] = i * dec + g->start; }   /* if / else if (:209-210) */
        if (v->variant == 0 && pd < inversethresh) fall[nfall++] = i * dec + g->start;
    }
    qsort(rise, (size_t)nrise, sizeof(edge), edge_cmp);
    int (*pc)[2] = (int (*)[2])malloc(sizeof(int[2]) * (size_t)(nrise > 0 ? nrise : 1));
    int npc = 0;
    for (int e = 0; e < nrise; e++) {
        const int ps = rise[e].pos;
        int ne = -1;                                   /* get_next_int, :678-692 */
        for (int k = 0; k < nfall; k++) if (fall[k] > ps) { ne = fall[k]; break; }
        if (ne <= ps) continue;
        int brk = 0;
        for (int k = 0; k < npc; k++) if (ps < pc[k][1] && ne >= pc[k][0]) { brk = 1; break; }
        if (brk) continue;
        pc[npc][0] = ps; pc[npc][1] = ne; npc++;
    }
    /* match_active_channels, :741-783 */
    if (npc == 0) {
        for (int c = 0; c < g->nch; c++) g->ch[c].inactive += 1;
    } else {
        for (int c = 0; c < g->nch; c++) {
            int inactive = 1, i = 0;
            while (i < npc) {
                if (pc[i][0] < g->ch[c].detect_stop && pc[i][1] >= g->ch[c].detect_start) {
                    g->ch[c].inactive = 0; inactive = 0;
                    memmove(&pc[i], &pc[i + 1], sizeof(int[2]) * (size_t)(npc - i - 1));
                    npc--;
                } else i++;
            }
            if (inactive) g->ch[c].inactive += 1;
        }
        for (int k = 0; k < npc; k++) {                /* activate, :785-841 */
            const int ds = pc[k][0], de = pc[k][1];
            const int dw = de - ds, emid = ds + dw / 2;
            const int ew = ipow2ceil((int)ceil((double)dw * (1.0 + 2.0 * v->puffer)));
            if (ew > v->blocklen) continue;            /* logged and skipped, :793-803 */
            int es = emid - ew / 2, ee = emid + ew / 2;
            if (es < 0) { es = 0; ee = ew; }
            if (ee > v->blocklen) { ee = v->blocklen; es = v->blocklen - ew; }
            if (g->nch == g->capch) {
                g->capch = g->capch ? 2 * g->capch : 8;
                g->ch = (vcm_chan *)realloc(g->ch, sizeof(vcm_chan) * (size_t)g->capch);
            }
            vcm_chan *c = &g->ch[g->nch++];
            memset(c, 0, sizeof *c);
            c->ID = g->active_channels_counter++;
            c->detect_start = ds; c->detect_stop = de; c->extract_start = es; c->extract_stop = ee; c->extract_width = ew;
            c->extract_window = ilog2d((double)ew);
            c->ovlskip = ew / v->relinvovl; c->outputsamples = ew - c->ovlskip;
            c->phaseincrement = es % v->relinvovl; c->inactive = -1;
        }
    }
    free(rise); free(fall); free(pc);
}

static void vcm_process(fdco_vcm *v, const float *sig, vcm_chan *c)     /* :373-397 */
{
    if (c->nblk == c->capblk) {
        c->capblk = c->capblk ? 2 * c->capblk : 8;
        c->data = (float *)realloc(c->data, sizeof(float) * 2 * (size_t)c->capblk * c->outputsamples);
    }
    extract_block(sig, c->extract_start, c->extract_width,
                  v->windows[c->extract_window] + 2 * (size_t)c->phase * c->extract_width, c->ovlskip,
                  c->data + 2 * (size_t)c->nblk * c->outputsamples);
    c->nblk++;
    c->count++;
    c->phase = (c->phase + c->phaseincrement) % v->relinvovl;
}

static void vcm_fill(const fdco_vcm *v, const vcm_seg *g, const vcm_chan *c, fdco_pdu *d)
{
    memset(d, 0, sizeof *d);
    d->kind = 1; d->source = g->ID; d->chan_id = c->ID;
    d->rel_bw = (double)c->extract_width / (double)v->blocklen;
    d->rel_cfreq = (double)(c->extract_start + c->extract_stop) / 2.0 / (double)v->blocklen;
    d->blockstart = (long)v->blockcount - c->count; d->blockend = (long)v->blockcount;
    d->vectorstart = c->extract_start; d->vectorend = c->extract_stop;
}

void fdco_vcm_work(fdco_vcm *v, const float *in, int nitems, fdco_pdu_list *L)      /* :542-576 */
{
    if (nitems <= 0) return;
    const float *sig = in, *sig_hist = v->hist;
    for (int it = 0; it < nitems; it++) {
        sig = in + 2 * (size_t)it * v->blocklen;
        for (int s = 0; s < v->nseg; s++) seg_detect(v, &v->seg[s], sig);           /* :558, :617-628 */
        /* extract_channels_in_segments_singlethread, :306-337 */
        for (int s = 0; s < v->nseg; s++) {
            vcm_seg *g = &v->seg[s];
            for (int k = 0; k < g->nch; k++) {
                vcm_chan *c = &g->ch[k];
                if (c->inactive < 0) { vcm_process(v, sig_hist, c); vcm_process(v, sig, c); c->inactive = 0; }  /* :399-403 */
                else if (c->inactive > v->delay) {                                  /* emit_channel, :406-452 */
                    fdco_pdu d; vcm_fill(v, g, c, &d);
                    d.finalized = 1; d.part = c->part; d.has_part = c->part > 0;
                    pdu_push(L, &d, c->data, c->nblk * c->outputsamples);
                    c->nblk = 0;
                } else vcm_process(v, sig, c);
                if (v->variant == 0 && v->maxblocks >= 0 && c->nblk >= v->maxblocks) {   /* emit_unfinished_channel, :454-510 */
                    const long ntx = v->maxblocks == 0 ? c->nblk : v->maxblocks;
                    if (ntx > 0) {
                        fdco_pdu d; vcm_fill(v, g, c, &d);
                        d.finalized = 0; d.part = c->part; d.has_part = 1;
                        pdu_push(L, &d, c->data, ntx * c->outputsamples);
                        memmove(c->data, c->data + 2 * (size_t)ntx * c->outputsamples,
                                sizeof(float) * 2 * (size_t)(c->nblk - ntx) * c->outputsamples);
                        c->nblk -= ntx;
                        c->part++;
                    }
                }
            }
        }
        if (v->variant == 1 && v->maxblocks >= 0)          /* SegmentDetection: partial emission after all channels, :359-362 */
            for (int k = 0; k < v->seg[0].nch; k++) {
                vcm_seg *g = &v->seg[0];
                vcm_chan *c = &g->ch[k];
                if (c->nblk < v->maxblocks) continue;
                const long ntx = v->maxblocks == 0 ? c->nblk : v->maxblocks;
                if (ntx <= 0) continue;
                fdco_pdu d; vcm_fill(v, g, c, &d);
                d.finalized = 0; d.part = c->part; d.has_part = 1;
                pdu_push(L, &d, c->data, ntx * c->outputsamples);
                memmove(c->data, c->data + 2 * (size_t)ntx * c->outputsamples,
                        sizeof(float) * 2 * (size_t)(c->nblk - ntx) * c->outputsamples);
                c->nblk -= ntx;
                c->part++;
            }
        for (int s = 0; s < v->nseg; s++) {                                         /* clear_inactive_channels, :512-524 */
            vcm_seg *g = &v->seg[s];
            int i = 0;
            while (i < g->nch) {
                if (g->ch[i].inactive > v->delay) {
                    free(g->ch[i].data);
                    memmove(&g->ch[i], &g->ch[i + 1], sizeof(vcm_chan) * (size_t)(g->nch - i - 1));
                    g->nch--;
                } else i++;
            }
        }
        sig_hist = sig;
        v->blockcount++;
    }
    memcpy(v->hist, sig, sizeof(float) * 2 * (size_t)v->blocklen);
}
