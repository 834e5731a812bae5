/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * CPU restatement ("oracle") of gr-FDC's hot path: overlap-save -> forward FFT (shifted, 1/N)
 * -> per-channel vector cut -> phase-shifting window -> inverse FFT (shifted) -> overlap discard -> * l.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and
 * only as the checker / the timed CPU baseline.  The product path (gr-fdc_amd/, include/fdc_amd.h)
 * never links, imports or calls anything in oracle/.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *  - window design (lib/windows.h): PINNED — compared bit-for-bit with the reference's own
 *    lib/windows.h compiled unmodified into oracle/_ref/ (oracle/Makefile, target ref).
 *  - channel parameter derivation (python/FrequencyDomainChannelizer.py:322-345): PINNED — compared with
 *    ~1600 outputs of the reference's own function, recorded by tests/golden/make_params_from_reference.py
 *    (which imports the reference's Python in the build container) in tests/golden/channel_params.json.
 *  - overlap_save / vector_cut_vxx / phase_shifting_windowing_vcc work(): restated from source;
 *    the reference .cc files need GNU Radio + VOLK headers that this image lacks => unbuildable here.
 *  - FFTs / multiply_const (GNU Radio + FFTW3f + VOLK, not under /root/reference, versions unpinned):
 *    PARITY UNPINNED by the reference (it holds no tests or vectors); restated from the published DFT
 *    definition, evaluated in double, cross-checked against numpy.fft (fixtures under tests/golden).
 */
#ifndef FDC_ORACLE_H
#define FDC_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* python/FrequencyDomainChannelizer.py:37-40.  Returns -1 where the reference raises ValueError. */
long fdco_nextpow2(double k);

/* python/FrequencyDomainChannelizer.py:322-345 (freq is the INTERNAL frequency in [0,1), i.e. after
 * get_freq (:70); bw after get_bw (:72)).  Returns 0, or -1 where the reference raises. */
int fdco_channel_params(int N, int R, double freq, double bw,
                        int *f, int *l, int *lout, double *pbw, double *sbw);

/* lib/windows.h:41-124.  w is [R][blocksize] interleaved complex float. */
void fdco_window(int wintype, int blocksize, float passbw, float stopbw, int R, int step,
                 int normalize, float *w);

/* lib/overlap_save_impl.cc:62-81 (byte-level; hist has itemsize*ovl bytes and is updated). */
void fdco_overlap_save(int itemsize, int outlen, int ovl, unsigned char *hist,
                       const void *in, int nitems, void *out);

/* lib/vector_cut_vxx_impl.cc:59-72 */
void fdco_vector_cut(int itemsize, int veclen, int offset, int blocklen,
                     const void *in, int nitems, void *out);

/* lib/phase_shifting_windowing_vcc_impl.cc:72-86.  win is [R][l]; *counter is the block's state. */
void fdco_phase_window(int l, int R, int shift, int *counter, const float *win,
                       const float *in, int nitems, float *out);

/* gr-fft fft_vcc(n, forward, rectangular window, shift, nthreads) as used at
 * python/FrequencyDomainChannelizer.py:206,228: forward => fftshift of the OUTPUT, inverse =>
 * ifftshift of the INPUT; unnormalised; computed in double, rounded to float once. */
void fdco_fft_vcc(int n, int forward, int shift, const float *in, int nitems, float *out);

/* Whole throughput chain, Appendix A.1-A.4 of SURVEY.md, float32 rounding at the reference's
 * stage boundaries (python/FrequencyDomainChannelizer.py:200-231, 283-299).
 *   prefix : ovl samples preceding x (NULL => zeros, lib/overlap_save_impl.cc:52)
 *   x      : nblocks*(N-N/R) new complex samples
 *   first_block : global index of the first block (phase counter = first_block*shift mod R)
 *   out[c] : nblocks*lout[c] complex samples
 *   spectrum (optional): nblocks*N normalised shifted spectrum (the hier block's debug port, :314)
 * use_float != 0 selects the float32 FFT arithmetic (the timed CPU baseline); nthreads: OpenMP. */
int fdco_channelizer(int N, int R, int wintype, int C, const int *f, const int *l,
                     const float *pbw, const float *sbw, long first_block,
                     const float *prefix, const float *x, int nblocks,
                     float **out, float *spectrum, int use_float, int nthreads);

/* ---- stateful sinks fed with the normalised spectrum (fdc_oracle_detect.c) ------------------------------------ */
typedef struct {
    int kind;            /* 0 = PowerActivationChannel, 1 = activity_detection_channelizer_vcm */
    int source;          /* PAC: ID argument; vcm: segment index                                    */
    int chan_id;         /* PAC: finished_channels at activation; vcm: channel counter in the segment (ID strings) */
    int finalized, part, has_part;
    double rel_bw, rel_cfreq;
    long blockstart, blockend, vectorstart, vectorend;
    long nsamples;
    float *samples;      /* interleaved complex, owned by the list */
} fdco_pdu;
typedef struct { fdco_pdu *pdu; int n, cap; } fdco_pdu_list;
void fdco_pdu_list_clear(fdco_pdu_list *L);

typedef struct fdco_pac fdco_pac;
/* PowerActivationChannel::make(blocklen, cfreq, bw, relinvovl, thresh, maxblocks, deactivation_delay, …, ID)
 * (include/FDC/PowerActivationChannel.h:49); NULL where the reference constructor throws. */
fdco_pac *fdco_pac_create(int blocklen, float cfreq, float bw, int relinvovl, float thresh_db, int maxblocks,
                          int deactivation_delay, int ID);
void fdco_pac_destroy(fdco_pac *p);
void fdco_pac_params(const fdco_pac *p, int *v8);   /* extract_start, extract_stop, extract_width, measure_start,
                                                       measure_stop, output_len, output_ovl_offset, deltaphase */
void fdco_pac_work(fdco_pac *p, const float *spectrum_items, int nitems, fdco_pdu_list *L);

typedef struct fdco_vcm fdco_vcm;
/* activity_detection_channelizer_vcm::make(blocklen, segments, thresh, relinvovl, maxblocks, …, minchandist,
 * channel_deactivation_delay, window_flank_puffer, …) (include/FDC/activity_detection_channelizer_vcm.h:49) */
fdco_vcm *fdco_vcm_create(int blocklen, int nseg, const float *segs, float thresh_db, int relinvovl, int maxblocks,
                          float minchandist, int deactivation_delay, double window_flank_puffer);
/* SegmentDetection::make(ID, blocklen, relinvovl, seg_start, seg_stop, thresh, minchandist, window_flank_puffer,
 * maxblocks_to_emit, channel_deactivation_delay, …) (include/FDC/SegmentDetection.h:49); same handle type. */
fdco_vcm *fdco_sd_create(int ID, int blocklen, int relinvovl, float seg_start, float seg_stop, float thresh_db,
                         float minchandist, float window_flank_puffer, int maxblocks, int deactivation_delay);
void fdco_vcm_destroy(fdco_vcm *v);
int fdco_vcm_segment_params(const fdco_vcm *v, int s, int *out5);   /* start, stop, width, dec, npower */
void fdco_vcm_work(fdco_vcm *v, const float *spectrum_items, int nitems, fdco_pdu_list *L);

#ifdef __cplusplus
}
#endif
#endif
