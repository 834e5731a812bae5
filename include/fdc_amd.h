/* fdc_amd.h — C-ABI of the MI355X (gfx950) frequency-domain channelizer.
 *
 * Drop-in boundary for the ONE hot path of gereonsuch/gr-FDC: overlap-save -> large forward FFT ->
 * per-channel vector cut / phase-shifting window -> per-channel inverse FFT -> overlap discard.
 * Plain pointers and sizes only; no C++ types, no exceptions, no torch types.  Every entry point names
 * the reference interface it replaces (paths relative to the reference tree).  How the reference's
 * blocks bind to these functions (C++ sync_block::work() bodies, ctypes stub) is in INTEGRATION.md.
 *
 * Conventions
 *  - samples are complex float32, interleaved (re, im): GNU Radio's gr_complex;
 *  - every function returns FDC_OK (0) or a negative fdc_status; fdc_last_error() gives the text;
 *  - a handle is used by one thread at a time (GNU Radio calls work() of one block instance from
 *    one thread); distinct handles may be used concurrently;
 *  - host pointers passed to *_work() are not retained after the call returns
 *    (the runtime owns them only for the duration of work(): lib/overlap_save_impl.cc:62-81).
 */
#ifndef FDC_AMD_H
#define FDC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    FDC_OK = 0,
    FDC_ERR_INVALID_ARGUMENT = -1, /* the predicates on which the reference ctors throw std::invalid_argument */
    FDC_ERR_HIP = -2,              /* a HIP runtime call failed */
    FDC_ERR_NO_DEVICE = -3,        /* no gfx950 device visible: the product path never falls back to the CPU */
    FDC_ERR_UNSUPPORTED = -4,
    FDC_ERR_NOMEM = -5
} fdc_status;

/* window shapes: lib/windows.h:28-32 */
enum { FDC_WIN_RECTANGULAR = 0, FDC_WIN_HANN = 1, FDC_WIN_RAMP = 2 };

const char *fdc_last_error(void);            /* thread-local text of the last failure */
/* No C++ exception leaves the library: every entry that can allocate on the host (std::vector, std::thread, new) runs behind one
 * barrier that turns std::bad_alloc / std::system_error into FDC_ERR_NOMEM and anything else into FDC_ERR_HIP (csrc/fdc_guard.hpp).
 * This entry throws each kind through that same barrier and returns FDC_OK when every one came back as the status it should
 * (needs no device; tests/test_abi_cpu.py). */
int fdc_selftest_exception_barrier(void);
const char *fdc_version(void);
int fdc_device_count(void);                  /* number of visible HIP devices (0 if none) */
/* Runs the headline geometry (65536-point blocks, R = 2, 256 channels of 256 bins, 96 blocks: the one-kernel path) through
 * fdc_pipeline_work() on EVERY visible device and compares a checksum of all output samples with device 0's (same kernels,
 * same input: the bytes must be identical), then once more through ONE multi-device handle over all visible devices
 * (fdc_pipeline_group below; with one device: two virtual members on it), the call cut in two so that the group's history is
 * used: again device 0's bytes.  Returns the number of devices verified, or a negative fdc_status with the device and the
 * mismatch in fdc_last_error().  A multi-GPU launcher calls it once before sharding block spans over the devices (the kernel
 * attributes are set per device: this is what exercises device_id > 0, and the group's worker threads on real peers). */
int fdc_selftest_devices(void);

/* ------------------------------------------------------------------------------------------------
 * Fused throughput pipeline = the chain python/FrequencyDomainChannelizer.py:200-231 wires:
 *   stream_to_vector -> overlap_save (lib/overlap_save_impl.cc:62-81) -> fft_vcc(N, fwd, shift) (:206)
 *   -> multiply_const_cc(1/N) (:214-216) -> per channel { vector_cut_vxx (lib/vector_cut_vxx_impl.cc:59-72)
 *   -> phase_shifting_windowing_vcc (lib/phase_shifting_windowing_vcc_impl.cc:72-86) -> fft_vcc(l, inv, shift)
 *   (:228) -> vector_cut_vxx(l, l-lout, lout) (:229) -> vector_to_stream (:230) -> multiply_const_cc(l) (:231) }.
 * The spectrum stays in HBM; only input samples and channel outputs cross the boundary.
 * ---------------------------------------------------------------------------------------------- */
typedef struct fdc_pipeline fdc_pipeline;

typedef struct {
    int32_t f;        /* first bin of the slice in the fftshifted spectrum (vector_cut_vxx offset) */
    int32_t l;        /* slice / inverse-FFT length, power of two (vector_cut_vxx blocklen)          */
    float passbw;     /* phase_shifting_windowing_vcc::make passbw                                   */
    float stopbw;     /* phase_shifting_windowing_vcc::make stopbw                                   */
} fdc_channel;

typedef struct {
    int32_t device_id;      /* HIP device ordinal */
    int32_t blocklen;       /* N, power of two (hier block "blocksize", py:138)                      */
    int32_t relinvovl;      /* R, power of two >= 2 (py:139); overlap = N/R                           */
    int32_t windowtype;     /* FDC_WIN_* (hier block "windowtype", py:227)                            */
    int32_t nchannels;
    const fdc_channel *channels;
    int32_t max_blocks;     /* largest nblocks a single work()/process call will carry; max_blocks x (sum of the channels' kept
                               samples per block) x 8 bytes must stay below 4 GiB (32-bit output offsets inside one call):
                               FDC_ERR_INVALID_ARGUMENT otherwise                                    */
    int32_t chunk_blocks;   /* blocks per internal launch group (0 = as many as a 2 GiB scratch budget
                               allows: long launches measured faster than cache-sized ones)          */
    int32_t keep_spectrum;  /* != 0: keep the whole normalised spectrum of a call (debug port, py:314) */
    int32_t flags;          /* FDC_PIPE_* bits, 0 = the library picks the fastest kernels for the plan            */
    int32_t min_block_launch; /* launch groups of fewer blocks than this do not take the one-block-per-compute-unit
                               kernels (0 = default 96: such a kernel gives a whole block to ONE compute unit, so a short
                               call would leave most of the device idle; 1 = always take them)                  */
    int32_t host_sub_blocks;  /* fdc_pipeline_work(): blocks per sub-batch whose transfers and kernels overlap (0 = about
                               8 MiB of input)                                                                */
} fdc_pipeline_cfg;
/* fdc_pipeline_cfg.flags: restrict the choice of kernels (A/B measurements, tests of the slower forms; results are the same
 * to rounding).  The library reads NO environment variable unless FDC_DEBUG_ENV=1 is set, and then only as a debugging
 * override of these fields (FDC_FORCE_GENERIC, FDC_NO_POLY, FDC_NO_BLOCK, FDC_BLOCK_HINTS, FDC_BLOCK_MIN_BLOCKS,
 * FDC_HOST_SUB, FDC_SINKS_THREADS, FDC_SINKS_TRACE, and the tile knobs of the two-launch kernels). */
enum {
    FDC_PIPE_FORCE_GENERIC = 1,   /* generic LDS Stockham kernels only                                              */
    FDC_PIPE_NO_POLY = 2,         /* no uniform-plan commutation: forward transform to a spectrum in memory + channel kernels */
    FDC_PIPE_NO_BLOCK = 4,        /* no one-block-per-compute-unit kernels (two-launch uniform path / two-pass transform)   */
    FDC_PIPE_PLAIN_STORES = 8,    /* block kernels: ordinary instead of streamed (nt) output stores                 */
    FDC_PIPE_NT_LOADS = 16,       /* block kernels: streamed (nt) input loads (not the 512- / 1024-bin kernels, which stage their loads) */
    FDC_PIPE_FULL_SPECTRUM = 32,  /* the handle's internal spectrum is written in full (default: only the 64-bin groups some channel reads) */
    FDC_PIPE_NO_FUSED = 128,      /* N = 4096: not the one-launch kernel that keeps the spectrum in LDS (A/B, tests); also off under FDC_PIPE_NO_POLY */
    FDC_PIPE_WIDE_UNIFORM = 64    /* uniform banks of ANY channel width (every channel l = L on the L-bin grid, one window) take the form without
                                     a spectrum: the width's block kernel where there is one (N = 16384 / 32768 / 65536: l = 64, 128, 256, 512, 1024), else two
                                     launches on generic kernels; default: only where that measured faster than the spectrum path (the block
                                     kernels; what the cost rule of csrc/fdc_plan_cost.hpp sends there: l = 1024 from 3 channels up) */
};

int fdc_pipeline_create(const fdc_pipeline_cfg *cfg, fdc_pipeline **out);
void fdc_pipeline_destroy(fdc_pipeline *p);

/* sizes derived from the configuration */
int64_t fdc_pipeline_input_samples(const fdc_pipeline *p, int nblocks);   /* nblocks*(N-N/R)            */
int64_t fdc_pipeline_output_samples(const fdc_pipeline *p, int nblocks);  /* nblocks*sum_c lout_c       */
int64_t fdc_pipeline_channel_offset(const fdc_pipeline *p, int channel, int nblocks); /* in d_out, samples */
int32_t fdc_pipeline_channel_lout(const fdc_pipeline *p, int channel);

/* work()-shaped entry: host buffers, stateful like the block chain (overlap history + window counters
 * persist across calls; lib/overlap_save_impl.h:33, lib/phase_shifting_windowing_vcc_impl.h:47).
 *   in          nblocks*(N-N/R) new samples (what stream_to_vector hands overlap_save)
 *   outs[c]     nblocks*lout_c samples for channel c (hier output port c)
 *   spectrum    NULL, or nblocks*N samples of the normalised spectrum (needs keep_spectrum)
 * Returns nblocks (items consumed, sync 1:1) or a negative fdc_status. */
int fdc_pipeline_work(fdc_pipeline *p, const void *in, int nblocks, void *const *outs, void *spectrum);
/* The same for a REAL input stream (float32 items, the hier block's "Float" input type: stream_to_vector(4, ...),
 * overlap_save(4, ...) and the fft_vfc the reference meant to put at python/FrequencyDomainChannelizer.py:207-208 — the branch
 * is unreachable there, :205-210): in = nblocks*(N-N/R) float32 samples, taken as the real part of the block; everything
 * behind the load is the complex path unchanged (fftshift, 1/N, channels).  History and block counter are shared with
 * fdc_pipeline_work; do not mix the two on one handle. */
int fdc_pipeline_work_real(fdc_pipeline *p, const void *in, int nblocks, void *const *outs, void *spectrum);

/* One contiguous SPAN of a longer stream, handed over by a dispatcher that cuts a work() call over several handles (the
 * members of an fdc_pipeline_group below; a user-written scheduler may call it directly).  Same as fdc_pipeline_work() except
 * that the two pieces of stream state come from the caller instead of from the handle:
 *   halo         the N/R samples in front of the span (the tail of the previous span; lib/overlap_save_impl.cc:62-81), or NULL
 *                for zeros (stream start, :52)
 *   first_block  global index of the span's first block (window phase = first_block*shift mod R,
 *                lib/phase_shifting_windowing_vcc_impl.cc:82)
 * Afterwards the handle's own state is that of a stream that ended with this span (a following fdc_pipeline_work() continues it).
 * _real: float32 samples, halo included (see fdc_pipeline_work_real). */
int fdc_pipeline_work_span(fdc_pipeline *p, const void *halo, const void *in, int64_t first_block, int nblocks, void *const *outs,
                           void *spectrum);
int fdc_pipeline_work_span_real(fdc_pipeline *p, const void *halo, const void *in, int64_t first_block, int nblocks, void *const *outs,
                                void *spectrum);

/* Optional: pin a host range that will be handed to fdc_pipeline_work() again and again (GNU Radio's circular buffers
 * live as long as the flowgraph: register them in start(), unregister in stop()).  A call whose `in` lies in a
 * registered range is DMA'd from it in place, and when every outs[c] does, the results are stored straight into
 * them; other buffers are staged through pinned memory inside the handle.  Results are identical either way.
 * The range must stay mapped until fdc_host_unregister(ptr) (same ptr as registered). */
int fdc_host_register(void *ptr, size_t bytes);
int fdc_host_unregister(void *ptr);
void fdc_pipeline_reset(fdc_pipeline *p);    /* history <- zeros, block counter <- 0 (fresh ctor state)  */

/* Device-resident entry (stateless; the form bench.py and a device-side flowgraph use).
 *   d_ring      device pointer: N/R halo samples preceding the span, then nblocks*(N-N/R) new samples
 *   first_block global index of the span's first block (window phase = first_block*shift mod R)
 *   d_out       device pointer, fdc_pipeline_output_samples() samples: channel c's stream of
 *               nblocks*lout_c samples starts at fdc_pipeline_channel_offset(p, c, nblocks)
 *   d_spectrum  NULL or device pointer for nblocks*N samples (needs keep_spectrum)
 *   stream      hipStream_t (NULL = the handle's own stream).  Asynchronous: returns after enqueue. */
int fdc_pipeline_process_device(fdc_pipeline *p, const void *d_ring, int64_t first_block, int nblocks,
                                void *d_out, void *d_spectrum, void *stream);
/* The same, and beside the spectrum the POWER OF ITS 16-BIN GROUPS (round 6): d_group_power receives nblocks x N/16 float32, entry [m][g] = the sum
 * of |S|^2 over bins 16 g .. 16 g + 15 of block m's normalised, shifted spectrum.  At N = 16384 / 32768 / 65536 the forward kernel sums them in its
 * epilogue, while the bins are in its registers; elsewhere (other block lengths, launch groups too short for the block kernel) a pass over the
 * spectrum does.  What a sink bank's power cells are summed from (fdc_sinks_group_power / fdc_sinks_prepare_from_groups below) instead of reading the
 * spectrum back: lib/PowerActivationChannel_impl.cc:286-306, lib/activity_detection_channelizer_vcm_impl.cc:630-650 sum |X|^2 over a bin range per
 * block.  Needs d_spectrum and N a multiple of 16; d_group_power = NULL is fdc_pipeline_process_device. */
int fdc_pipeline_process_device_power(fdc_pipeline *p, const void *d_ring, int64_t first_block, int nblocks, void *d_out, void *d_spectrum,
                                      void *d_group_power, void *stream);
int fdc_pipeline_synchronize(fdc_pipeline *p);
void *fdc_pipeline_stream(fdc_pipeline *p);  /* the handle's hipStream_t */
/* The one-block-per-compute-unit kernels are persistent: one workgroup per unit, all of its LDS.  A consumer that wants to run other
 * kernels BESIDE them (the sinks' look-ahead form, fdc_sinks_spectrum_ahead) asks for n units to be left out: the block kernels then
 * launch on (units - n) workgroups.  Returns that number — launch groups should hold a multiple of it in blocks, or the last round of
 * the persistent loop runs part of the machine — or a negative fdc_status.  n = 0 restores the default. */
int fdc_pipeline_reserve_compute_units(fdc_pipeline *p, int32_t n);
int32_t fdc_pipeline_chunk_blocks(const fdc_pipeline *p);   /* blocks per internal launch group */
/* Which kernels a process call without spectrum output runs: 0 = generic LDS Stockham kernels (any size),
 * 1 = spectrum in memory at N = 16384 / 32768 / 65536 (forward transform by the block kernel in one launch, ms[0]; channel kernels
 * ms[2]), 2 = uniform-plan path (all channels
 * l = 256 on the 256-bin grid: stage 1 = column FFT + window + IFFT, stage 2 = FFT across slots; timing
 * slots ms[0], ms[1] then hold stage 1 and stage 2 and ms[2] = 0), 3 = the uniform plan at N = 65536, R = 2 as ONE
 * kernel (one block per compute unit, nothing between the input rows and the output samples touches memory;
 * ms[0] = that kernel, ms[1] = ms[2] = 0).  Path 3 is every plan that is a line-up of up to FOUR BANKS, one block-kernel launch each: a bank is a
 * set of channels of one width l in {64, 128, 256, 512, 1024} on one grid f = l*slot + r with one window (l = 256: any r; 512 / 1024: r = 0 or l/2; 128 /
 * 64: r a multiple of l/4), banks of different widths may stand side by side (round 5), at N = 16384 / 32768 / 65536 (2 / 4 / 8 passes of the same kernels),
 * R = 2 or 4.  4 = a SPLIT plan at any of those three block lengths: the banks take path 3 (ms[0]), the rest — widths without a block kernel, odd offsets,
 * a fifth bank, banks the cost rule sends back — take the spectrum path on a partial spectrum that holds only what they read (forward transform ms[1], channel
 * kernels ms[2]).  Which it is, the cost rule decides (csrc/fdc_plan_cost.hpp: measured constants per launch and per band read; a plan goes to path 1 whole
 * when the sum says so); fdc_pipeline_describe / fdc_pipeline_plan_preview say what was chosen.
 * 5 (round 6) = N = 4096 — the block length of the reference's example flowgraph — in ONE launch (csrc/fdc_fused4096.hip; ms[0]): forward transform, channel
 * slices, windows and inverse transforms with the spectrum in LDS, nothing but the new input samples and the output samples crosses the memory interface.
 * Every plan of channels 16 ... 1024 bins wide (any offsets, windows, overlaps between them) of at least 512 and — about — at most 4096 bins in total
 * (exactly: the rows of a pair of blocks fit eight waves — two rows of 1024 bins, four of 512, eight of 256 or less per wave — and the two spectrum tiles);
 * a call that asks for the spectrum runs path 0 on the same handle.  FDC_PIPE_NO_FUSED / FDC_PIPE_NO_POLY: paths 0 / 2 as before (FDC_PIPE_WIDE_UNIFORM: path 5 for
 * plans under 512 bins too, which the two launches run a few per cent faster).  plan_preview: every channel -1.  The kernel takes one block per workgroup
 * where no channel is wider than 256 bins and a pair of blocks where one is (fdc_pipeline_describe says which). */
int32_t fdc_pipeline_path(const fdc_pipeline *p);
/* The same in words, for logs: which kernels the handle's plan was given ("N = 65536, R = 2, 512 channels; path 3: k_blknar, l = 128, bank of 511 half a
 * channel off the grid + bank of 1 on the grid (two launches)").  Writes at most n bytes including the terminator; returns the untruncated length, -1 for
 * bad arguments.  (No counterpart in the reference, which has one implementation.) */
int32_t fdc_pipeline_describe(const fdc_pipeline *p, char *buf, int32_t n);
/* What fdc_pipeline_create WOULD choose for cfg, without a device (round 5: the plan selector — csrc/fdc_api.hip classify_plan, the measured constants of
 * csrc/fdc_plan_cost.hpp — is host code): returns the path number (as fdc_pipeline_path; the block kernels are those of an MI355X), writes the description
 * (as fdc_pipeline_describe) into buf if given, and into assignment[c], if given (nchannels entries), where channel c goes: k >= 0 = bank k (one block-kernel
 * launch each, in launch order; path 2: the one bank of the two-launch form), -1 = the spectrum path (a split plan's remainder, or the whole plan),
 * -2 - c0 = a copy of channel c0's output (same slice, same window).  Negative return: the status fdc_pipeline_create would give for the arguments. */
int fdc_pipeline_plan_preview(const fdc_pipeline_cfg *cfg, char *buf, int32_t n, int32_t *assignment);

/* Timing of the kernels with HIP events recorded on the stream they are launched on (bench.py's
 * roofline leg).  While enabled, every process_device call brackets its launches with events; the readout
 * synchronises on them, sums all launch groups since enable / the previous readout and clears them:
 * ms[0] forward FFT pass A, ms[1] forward FFT pass B (or the single-pass kernel), ms[2] fused channel
 * kernel(s), ms[3] = number of launch groups summed (each group = one chunk of fdc_pipeline_chunk_blocks
 * blocks, the last of a call possibly shorter).  Needs n >= 4; returns the number of entries written. */
/* enable: 0 = off, 1 = every launch group, k > 1 = every k-th launch group (a sample: the event packets between the
 * kernels cost 7-17 us per group on MI355X, enough to show in the throughput of the region they time) */
int fdc_pipeline_enable_timing(fdc_pipeline *p, int enable);
int fdc_pipeline_last_kernel_ms(fdc_pipeline *p, float *ms, int n);

/* ------------------------------------------------------------------------------------------------
 * Multi-device handle: the same chain behind ONE work() — what the FrequencyDomainChannelizer hier block is to the scheduler
 * (python/FrequencyDomainChannelizer.py:283-315: one block, one work() thread) — spread over several GPUs of the node.  A call
 * of nblocks items is cut into contiguous spans, one per member device; every member copies its span's samples over its own
 * PCIe link (the halo of a span is the N/R samples in front of it in the caller's buffer; the first span's halo is the history
 * the group keeps), runs the kernels with the span's global first-block index and writes its results into the caller's
 * per-channel buffers at the span's block offset, all members concurrently.  No collective and no device-to-device traffic:
 * spans are independent given halo and block index.  Results are those of one handle fed the same stream
 * (bit for bit when the members run the kernels one handle would run: tests/test_group_gpu.py).
 * The reference's own parallelism inside one work(): 4 FFTW threads (python/...:206), one std::thread per segment / per
 * detected channel (lib/activity_detection_channelizer_vcm_impl.cc:293-304, :339-371).
 *   cfg              as for fdc_pipeline_create; device_id is ignored, max_blocks is the largest nblocks of a GROUP call
 *   devices[n]       HIP device ordinal of each member; an ordinal may appear more than once (virtual members on one GPU)
 *   min_span_blocks  a member is not given fewer blocks than this, so short calls use fewer members (0 = default 8)
 * One thread at a time per group, like every handle.  A call that fails on some member leaves the group unusable until
 * fdc_pipeline_group_reset() (the other members' spans of that call were written).
 * ---------------------------------------------------------------------------------------------- */
typedef struct fdc_pipeline_group fdc_pipeline_group;
int fdc_pipeline_group_create(const fdc_pipeline_cfg *cfg, const int32_t *devices, int ndevices, int min_span_blocks,
                              fdc_pipeline_group **out);
void fdc_pipeline_group_destroy(fdc_pipeline_group *g);
/* work()-shaped, same arguments and state semantics as fdc_pipeline_work() / fdc_pipeline_work_real() */
int fdc_pipeline_group_work(fdc_pipeline_group *g, const void *in, int nblocks, void *const *outs, void *spectrum);
int fdc_pipeline_group_work_real(fdc_pipeline_group *g, const void *in, int nblocks, void *const *outs, void *spectrum);
void fdc_pipeline_group_reset(fdc_pipeline_group *g);           /* history <- zeros, block counter <- 0, error state cleared */
int32_t fdc_pipeline_group_size(const fdc_pipeline_group *g);
fdc_pipeline *fdc_pipeline_group_member(fdc_pipeline_group *g, int i);   /* owned by the group (fdc_pipeline_path, sizes, timing) */
int32_t fdc_pipeline_group_device(const fdc_pipeline_group *g, int i);
int32_t fdc_pipeline_group_member_max_blocks(const fdc_pipeline_group *g);   /* the longest span a member can be given */
/* Host placement (round 6).  A group's worker thread for member i — the thread that copies the member's span over its PCIe link — pins itself
 * to the CPUs of the NUMA node its device hangs off (/sys/bus/pci/devices/<bdf>/numa_node, /sys/devices/system/node/node<k>/cpulist,
 * intersected with the process's own mask; member 0 runs on the calling thread, which is never touched: pin the scheduler thread of the
 * block yourself if you want it placed).  Pinned staging comes from the host pool closest to the member's device whatever thread asks
 * (hipHostMalloc without hipHostMallocNumaUser).  Best effort: unknown node (-1: single-node machines, most containers) = left alone.
 *   fdc_device_numa_node(d)                 the node of HIP device d, -1 if unknown or no such device
 *   fdc_selftest_worker_placement(node,..)  starts a worker exactly as a group would for a member on `node` and checks the mask it runs
 *                                           under: 1 = pinned inside the node's CPUs, 0 = left alone under the process's mask (node unknown
 *                                           or none of its CPUs usable), negative = failure; the two counts are the node's usable CPUs and the
 *                                           worker's.  Needs no device (tests/test_abi_cpu.py). */
int fdc_device_numa_node(int device_id);
int fdc_selftest_worker_placement(int node, int32_t *cpus_of_node, int32_t *cpus_of_worker);
/* the spans of the last call: first_block[i] (global index) and nblocks[i] (0 = member idle) for i < min(size, cap); returns size */
int fdc_pipeline_group_last_spans(const fdc_pipeline_group *g, int64_t *first_block, int32_t *nblocks, int cap);

/* ------------------------------------------------------------------------------------------------
 * Stateful sinks fed with the normalised spectrum (what the hier block connects to normalize_input,
 * python/FrequencyDomainChannelizer.py:301-312): a bank of
 *   gr::FDC::PowerActivationChannel::make(blocklen, cfreq, bw, relinvovl, thresh, maxblocks, deactivation_delay,
 *        msg, fileoutput, path, verbose, ID)                       — include/FDC/PowerActivationChannel.h:49
 * instances and the segments of one
 *   gr::FDC::activity_detection_channelizer_vcm::make(blocklen, segments, thresh, relinvovl, maxblocks, message,
 *        fileoutput, path, threads, minchandist, channel_deactivation_delay, window_flank_puffer, verbose)
 *                                                                  — include/FDC/activity_detection_channelizer_vcm.h:49
 * sharing ONE device-resident spectrum.  Power sums and extractions (window, half swap, inverse FFT, overlap discard)
 * run on the GPU for a whole batch; the per-block decisions run on the host (lib/PowerActivationChannel_impl.cc:137-306,
 * lib/activity_detection_channelizer_vcm_impl.cc:542-841).  What the reference publishes as pmt PDUs
 * (dict + c32vector, …vcm_impl.cc:415-430, PowerActivationChannel_impl.cc:222-233) comes back as POD records that the
 * C++ face turns into pmt; `msg`, `fileoutput`, `path`, `verbose` and `threads` stay with that face.
 * create() fails with FDC_ERR_INVALID_ARGUMENT exactly where the reference constructors throw.
 * Failure: the blocks are stateful and a batch advances that state on the device as it goes (block counter, channel state
 * machines, buffered blocks, the two-deep submission).  A work / submit / flush call that fails after it has started to do so
 * (a HIP error, out of memory while growing a landing buffer) cannot be rolled back or repeated: the handle is DEAD from then
 * on — every later work / submit / flush returns FDC_ERR_HIP naming the first failure, PDUs of the failed call are not handed
 * out (none twice, none with a wrong payload) — and must be destroyed.  Argument errors (a count out of range, a null buffer,
 * work_device while a batch is in flight) are refused before anything moves and leave the handle usable.
 * ---------------------------------------------------------------------------------------------- */
typedef struct fdc_sinks fdc_sinks;
typedef struct { float cfreq, bw; int32_t id; } fdc_pac_cfg;           /* INTERNAL frequency units, [0,1) */
typedef struct { float start, stop; } fdc_segment_cfg;                 /* INTERNAL frequency units        */
typedef struct {
    int32_t device_id, blocklen, relinvovl;
    int32_t npac; const fdc_pac_cfg *pac;
    float pac_thresh_db; int32_t pac_maxblocks, pac_deactivation_delay;
    int32_t nseg; const fdc_segment_cfg *seg;
    float det_thresh_db; int32_t det_maxblocks; float minchandist; int32_t det_deactivation_delay;
    double window_flank_puffer;
    int32_t max_blocks;                                                /* largest batch of one work call  */
    int32_t det_variant;   /* 0 = activity_detection_channelizer_vcm (all segments in one block); 1 = one
                              gr::FDC::SegmentDetection::make(ID, blocklen, relinvovl, seg_start, seg_stop, thresh,
                              minchandist, window_flank_puffer, maxblocks_to_emit, channel_deactivation_delay, …)
                              (include/FDC/SegmentDetection.h:49) per segment, ID = segment index — the twin the hier
                              block instantiates (python/FrequencyDomainChannelizer.py:261-278): its own segment geometry,
                              raw power sums, block counter from 0, partial emission after all channels */
    int32_t verbose;       /* the blocks' `verbose` argument (lib/PowerActivationChannel_impl.h:46-50): 0 = no log, 1 = to
                              stdout, 2 = to the reference's log files in the working directory — gr-FDC.PowActChan.<ID>.log
                              (PowerActivationChannel_impl.cc:54), gr-FDC.ActDetChan.log (…vcm_impl.cc:94),
                              gr-FDC.ActDetChan.ID_<n>.log (SegmentDetection_impl.cc:51); same lines: the derived geometry at
                              construction, one line per emitted PDU */
    int32_t det_id;        /* det_variant 1 with ONE segment: SegmentDetection's ID argument (names the log file and the
                              segment in the message IDs); < 0 = the segment index */
    int32_t flags;         /* FDC_SINKS_* bits, 0 = defaults */
    int32_t threads;       /* host engine only: worker threads of the decision phase (0 = chosen from the bank's size) */
    int32_t seg_id_base;   /* segment i of this bank is segment seg_id_base + i of the block it belongs to (the number in the message
                              IDs, fdc_pdu.source and the log lines): a bank that holds part of a block's segments — a member of an
                              fdc_sinks_group — still names them as the reference does; 0 for a whole block */
} fdc_sinks_cfg;
/* fdc_sinks_cfg.flags.  A bank runs on one of two engines.  DEVICE (default): the work() loops of the blocks
 * (lib/PowerActivationChannel_impl.cc:146-170, lib/activity_detection_channelizer_vcm_impl.cc:551-568) are device kernels —
 * per-channel state machines, edge detection and matching, the layout of the emitted payloads — so a batch is enqueued without a
 * host round trip and only finished PDUs cross PCIe.  HOST: the same decisions on host threads between two GPU phases (round 1/2
 * form); taken for verbose != 0 (the log lines are written while the decisions are made), for detection segments of more than
 * 1024 power cells, and on request.  Both engines emit the same PDUs in the same order (tests/test_sinks_gpu.py). */
enum {
    FDC_SINKS_HOST_DECISIONS = 1,   /* use the host engine */
    FDC_SINKS_DEVICE_PAYLOAD = 2,   /* device engine: fdc_pdu.samples are DEVICE pointers (no payload copy to the host); valid like
                                       the host pointers, until the next-but-one batch is submitted */
    FDC_SINKS_LOOKAHEAD = 4         /* a second spectrum buffer and a fill stream: the producer writes batch n + 1 while the bank decides
                                       batch n (fdc_sinks_spectrum_ahead / fdc_sinks_fill_stream / fdc_sinks_prepare, below) */
};
typedef struct {
    int32_t kind;        /* 0 = PowerActivationChannel, 1 = detected channel of a segment                       */
    int32_t source;      /* PAC: its ID argument; detection: segment index                                      */
    int32_t chan_id;     /* the running number inside the ID string: PAC finished_channels at activation
                            (…PowActChan.<ID>.<n>), detection: channel counter of the segment (…DETECTED.<seg>.<n>) */
    int32_t finalized, part, has_part;   /* has_part: whether the dict carries "part" (…vcm_impl.cc:419-420)   */
    double rel_bw, rel_cfreq;
    int64_t blockstart, blockend, vectorstart, vectorend;   /* vectorstart/end are in the dict for detection only */
    int64_t nsamples;
    const void *samples; /* complex float32, owned by the handle until its next work call                     */
    char id[72];         /* the message ID the reference builds when the channel is ACTIVATED:
                            "<YYYY-mm-dd-HH-MM-SS>.PowActChan.<ID>.<n>" (PowerActivationChannel_impl.cc:308-312; the dict and
                            the file name append ".fin" / ".part" / ".parted.<k>", :224, :237) and
                            "<YYYY-mm-dd-HH-MM-SS>.DETECTED.<seg>.<n>" (…vcm_impl.cc:526-530); local time, all PDUs of one
                            activation carry the same string */
} fdc_pdu;

/* Log lines of the sinks (verbose != 0) additionally go to this callback when one is set (process-wide; NULL = off). */
typedef void (*fdc_log_fn)(const char *line, void *user);
void fdc_set_log_callback(fdc_log_fn fn, void *user);

/* Same, for a hier block fed with items that are ALREADY transformed (inpveclen = blocksize: the front end is skipped,
 * python/FrequencyDomainChannelizer.py:201, :284-290): nblocks unnormalised, fftshifted spectrum items in; the 1/N of
 * multiply_const_cc (:213-216), the channel branches and, if given, the sinks run on the device.  sinks may be NULL. */
int fdc_pipeline_work_spectrum(fdc_pipeline *p, const void *in, int nblocks, void *const *outs, void *spectrum,
                               fdc_sinks *sinks);
int fdc_sinks_create(const fdc_sinks_cfg *cfg, fdc_sinks **out);
/* The whole hier block in one call: fdc_pipeline_work() whose spectrum lands directly in the sinks' device buffer,
 * followed by the sinks' work on it (python/FrequencyDomainChannelizer.py:283-312).  The pipeline needs keep_spectrum,
 * the same blocklen/device as the sinks and nblocks <= the sinks' max_blocks.
 * Two forms, chosen by the BANK:
 *  - a bank without FDC_SINKS_LOOKAHEAD (default): front end, then the sinks, inside the call; fdc_sinks_pdu*() afterwards give the PDUs
 *    the reference's blocks would have published for exactly these items (the reference's item-by-item behaviour, whatever the batch);
 *  - a bank created with FDC_SINKS_LOOKAHEAD: PIPELINED (round 6).  In the reference the front end and every sink are blocks of one
 *    flowgraph that GNU Radio's thread-per-block scheduler runs side by side (python/FrequencyDomainChannelizer.py:237-278); here one call
 *    copies and transforms ITS items into the bank's next-batch buffer on the bank's fill stream (power cells behind them), submits the
 *    batch the call BEFORE left there (fdc_sinks_submit_device: decision chains beside this call's copy and transform, then extractions)
 *    and hands out the PDUs of the batch before that one, whose payload copy ran meanwhile.  The call returns when its input has left the
 *    caller's buffer (and its stream outputs / debug spectrum, if any, have arrived): with pinned input and no stream outputs it costs what
 *    its input copy costs.  fdc_sinks_pdu*() after call n give the PDUs of the items of call n - fdc_pipeline_sinks_latency() (2 on the
 *    device engine, 1 on the host engine; none for the first calls) — same PDUs, same order, same bits as the serial form, later.
 *    fdc_pipeline_flush_sinks() hands out what is still inside: call it until it returns 0 when the stream ends (the block's stop()).
 *    The persistent forward kernel leaves ncu/8 compute units to the bank's chains when a call is long enough to take it through two
 *    rounds (fdc_pipeline_reserve_compute_units overrides).  Between the first pipelined call and the last flush the bank belongs to the
 *    pipeline: no fdc_sinks_work*() / submit / prepare on it from elsewhere; flush before destroying either handle
 *    (fdc_pipeline_reset resets the front end only — history and block counter; a batch already transformed into the bank stays there and its PDUs
 *    come out with the next call or a flush: the sink blocks keep their own state, as separate blocks do).  A pipelined call that fails after it has
 *    advanced the stream state (history, block counter, the bank's buffers: a HIP error, a dead bank) cannot be repeated or skipped: every later
 *    fdc_pipeline_work_sinks / fdc_pipeline_flush_sinks on that pipeline returns FDC_ERR_HIP — destroy both handles (as for a dead bank, below). */
int fdc_pipeline_work_sinks(fdc_pipeline *p, const void *in, int nblocks, void *const *outs, void *spectrum,
                            fdc_sinks *sinks);
/* Pipelined form: submits / finishes the oldest batch still inside and makes its PDUs the bank's current ones; returns its block count,
 * 0 when nothing is left (serial form: always 0), or a negative fdc_status. */
int fdc_pipeline_flush_sinks(fdc_pipeline *p, fdc_sinks *sinks);
/* How many calls later the PDUs of a call's items are handed out: 0 (serial form), 1 or 2 (pipelined; see above); -1 for a null handle. */
int32_t fdc_pipeline_sinks_latency(const fdc_pipeline *p, const fdc_sinks *sinks);
void fdc_sinks_destroy(fdc_sinks *s);
/* work()-shaped: nitems normalised-spectrum items of blocklen samples each on the host; returns nitems */
int fdc_sinks_work(fdc_sinks *s, const void *spectrum, int nitems);
/* the same when only the bins [bin_lo, bin_hi) of every item are needed by this bank's channels and segments: only those columns
 * cross PCIe (one strided copy), the rest of the device-side items keeps whatever it held.  The caller answers for the band covering
 * every bin the bank reads (fdc_sinks_read_band gives it); what an fdc_sinks_group member is fed with. */
int fdc_sinks_work_band(fdc_sinks *s, const void *spectrum, int nitems, int32_t bin_lo, int32_t bin_hi);
/* [lo, hi): every bin this bank can ever read — the measure and extract ranges of its PowerActivationChannels, its segments widened by
 * the widest extraction a detected channel can get (lib/activity_detection_channelizer_vcm_impl.cc:575-607: the next power of two above
 * width * (1 + 2 * window_flank_puffer), centred on the channel, clamped to the block) */
int fdc_sinks_read_band(const fdc_sinks *s, int32_t *lo, int32_t *hi);
/* device-resident: the producer (fdc_pipeline_process_device with d_spectrum = fdc_sinks_spectrum(s)) has written
 * nblocks spectra there on a stream it has synchronised; no PCIe traffic for the spectrum */
void *fdc_sinks_spectrum(fdc_sinks *s);
void *fdc_sinks_stream(fdc_sinks *s);
int32_t fdc_sinks_blocklen(const fdc_sinks *s);     /* N the bank was created for                                  */
int32_t fdc_sinks_max_blocks(const fdc_sinks *s);   /* capacity of its device-resident spectrum buffer, in blocks   */
int fdc_sinks_work_device(fdc_sinks *s, int nblocks);
/* Two-deep form of fdc_sinks_work_device for a producer that keeps the device busy: submit() enqueues the batch that sits in
 * the spectrum buffer and, if an earlier batch is still in flight, finishes that one first (waits for its payload copy, builds its
 * PDUs) while the device works on the new one.  Returns the number of blocks of the batch whose PDUs are readable now through
 * fdc_sinks_pdu*() (0 = none yet), or a negative fdc_status.  flush() finishes the batch in flight (returns its block count, 0 if
 * there was none).  The spectrum buffer may be overwritten by work enqueued on fdc_sinks_stream() as soon as submit() returns.
 * fdc_sinks_work_device(s, n) == submit(s, n) + flush(s); it refuses to run while a submitted batch is in flight.
 * PDUs (and their payload pointers) stay valid until the next submit / flush / work call on the handle. */
int fdc_sinks_submit_device(fdc_sinks *s, int nblocks);
int fdc_sinks_flush(fdc_sinks *s);
int32_t fdc_sinks_engine(const fdc_sinks *s);       /* 0 = host decisions, 1 = device decisions */
/* Look-ahead (banks created with FDC_SINKS_LOOKAHEAD; round 5).  The decision kernels of a batch are chains — one wave per
 * PowerActivationChannel, one workgroup per detection segment (lib/activity_detection_channelizer_vcm_impl.cc:741-841 is sequential over
 * blocks and channels) — that leave the device idle, and the host has to see their summary before it can size the extraction launches.
 * With two spectrum buffers the producer fills batch n + 1 beside them:
 *     fill(fdc_sinks_spectrum(s)) on fdc_sinks_fill_stream(s)                                 batch 0   (a look-ahead bank is filled on the fill stream only)
 *     loop:  fill(fdc_sinks_spectrum_ahead(s)) on fdc_sinks_fill_stream(s)                    batch n + 1   (never on fdc_sinks_stream)
 *            fdc_sinks_prepare(s, nblocks, 1)                          its power cells behind the fill, and the mark "batch n + 1 complete"
 *            fdc_sinks_submit_device(s, nblocks)                       batch n; afterwards fdc_sinks_spectrum(s) names batch n + 1's buffer
 * fdc_sinks_spectrum(s) is always the buffer the NEXT submit / work call reads; it alternates between the two, so ask again for every
 * batch.  The bank orders the streams itself: a submit waits for the mark fdc_sinks_prepare left behind ITS batch on the fill stream (not for
 * what the producer has enqueued there since: that is what runs beside the decisions) — without a prepare of the same block count, for
 * everything enqueued on the fill stream so far, which is correct and overlaps nothing; work enqueued on the fill stream after a submit
 * has returned waits until the batch before that submit's has been read for the last time.  The producer should
 * leave a few compute units free for the chains (fdc_pipeline_reserve_compute_units): the block kernels are persistent and fill every
 * unit's LDS.  Without the flag: spectrum_ahead and fill_stream return null, prepare refuses.
 * fdc_sinks_prepare(s, n, ahead): power cells of the n blocks in fdc_sinks_spectrum_ahead (ahead = 1) or fdc_sinks_spectrum (0) on the
 * fill stream; the submit of that batch then skips them if its block count is n.
 * Round 6: a batch prepared AHEAD is committed.  The submit in front of it, once it has placed its own tasks, enqueues the prepared batch's decision
 * chain at once (device engine) instead of when the caller returns with the next submit — the chain is the long pole of a detector's step and used
 * to start a host lap late — so the next submit / work_device call must be for exactly that batch (another block count: FDC_ERR_INVALID_ARGUMENT and the
 * handle is dead); fdc_sinks_work / work_band / a second prepare of the current buffer are refused until it has been submitted.  The extraction
 * streams run at the device's highest stream priority (own hardware queues: at equal priority they shared one with the fill stream and sat behind the
 * next batch's forward kernel). */
void *fdc_sinks_spectrum_ahead(fdc_sinks *s);
void *fdc_sinks_fill_stream(fdc_sinks *s);
int fdc_sinks_prepare(fdc_sinks *s, int nblocks, int ahead);
/* Power cells WITHOUT a pass over the spectrum (round 6).  fdc_sinks_group_power(s) / fdc_sinks_group_power_ahead(s): device buffers of max_blocks x N/16
 * float32 that go with fdc_sinks_spectrum(s) / fdc_sinks_spectrum_ahead(s) (they swap together; NULL for a bank without cells or N < 16): the producer
 * hands them to fdc_pipeline_process_device_power as d_group_power.  fdc_sinks_prepare_from_groups(s, n, ahead) then sums every cell from the groups
 * inside it plus the bins of the (at most two) groups it cuts, and marks the batch like fdc_sinks_prepare: the submit / work call of n blocks that follows
 * skips its own power pass.  ahead = 0 also on banks without FDC_SINKS_LOOKAHEAD (on the bank's stream).  The cells' float sums are formed in another
 * order than k_cell_power forms them (last bits); the PDUs' payloads do not depend on them.  fdc_pipeline_work_sinks does all of this itself. */
void *fdc_sinks_group_power(fdc_sinks *s);
void *fdc_sinks_group_power_ahead(fdc_sinks *s);
int fdc_sinks_prepare_from_groups(fdc_sinks *s, int nblocks, int ahead);
/* PDUs emitted by the last work call, in emission order */
int fdc_sinks_pdu_count(const fdc_sinks *s);
int fdc_sinks_pdu(const fdc_sinks *s, int i, fdc_pdu *out);
/* all of them at once: fills out[0 .. min(count, cap)) and returns the count.  Every payload is one contiguous run of a pinned
 * buffer of the handle (device engine: always; host engine: when all blocks of the PDU come from the last call). */
int fdc_sinks_pdus(const fdc_sinks *s, fdc_pdu *out, int cap);
/* for every PDU of the last call, the index (inside that call) of the item at which the block emitted it: what orders the PDUs of
 * several banks that saw the same items (fdc_sinks_group) the way one bank orders its own; returns the PDU count */
int fdc_sinks_pdu_emit_items(const fdc_sinks *s, int32_t *item, int cap);
/* the same, and for every PowerActivationChannel PDU the index of its channel in THIS bank's cfg->pac[] (-1 for a detection): a group that hands
 * its members the bank's channels in frequency order puts their PDUs back into the bank's own order with it (either array may be NULL) */
int fdc_sinks_pdu_emit_order(const fdc_sinks *s, int32_t *item, int32_t *pac, int cap);
/* derived geometry (for logs and tests): v[8] = extract_start, extract_stop, extract_width, measure_start,
 * measure_stop, output_len, output_ovl_offset, deltaphase;  v[5] = start, stop, width, decimation, power cells */
int fdc_sinks_pac_params(const fdc_sinks *s, int i, int32_t *v8);
int fdc_sinks_segment_params(const fdc_sinks *s, int i, int32_t *v5);

/* ------------------------------------------------------------------------------------------------
 * The sink blocks over several devices ("shard by channel / by segment").  A sink block's input is the
 * stream of normalised spectrum items (512 KiB each at N = 65536): fed from host memory, ONE device is bound by its PCIe link
 * long before its kernels matter (1024 items: 9.6 ms of copy, 0.5 ms of kernels).  The group cuts the bank BY FREQUENCY: the
 * PowerActivationChannels sorted by centre frequency (round 5: whatever order cfg->pac[] lists them in; the PDUs come back in the
 * bank's own order) and the segments as listed (their numbers in the message IDs are their places in cfg->seg[]: list them by
 * frequency), in runs of equal counts, one run per member device.  Every
 * member copies only the band of bins its run reads (fdc_sinks_work_band) over its own link and runs its state machines and
 * extractions on it, all members concurrently; their PDUs are merged into the order ONE bank emits them in (item by item:
 * PowerActivationChannels in bank order, then the segments in order).  Every channel / segment lives on exactly one member, so
 * no state is shared and nothing is exchanged between devices.  Same PDUs as one bank on one device, payload bits included
 * (tests/test_group_gpu.py).  The reference's counterpart: one std::thread per segment and per detected channel
 * (lib/activity_detection_channelizer_vcm_impl.cc:293-304, :339-371), one scheduler thread per PowerActivationChannel block.
 *   cfg       as for fdc_sinks_create; device_id is ignored; max_blocks = items per call, for every member
 *   devices   HIP ordinals, repeats allowed (virtual members)
 * One thread at a time per group.  PDUs (and payload pointers) stay valid until the next work call on the group. */
typedef struct fdc_sinks_group fdc_sinks_group;
int fdc_sinks_group_create(const fdc_sinks_cfg *cfg, const int32_t *devices, int ndevices, fdc_sinks_group **out);
void fdc_sinks_group_destroy(fdc_sinks_group *g);
int fdc_sinks_group_work(fdc_sinks_group *g, const void *spectrum, int nitems);      /* like fdc_sinks_work */
int fdc_sinks_group_pdu_count(const fdc_sinks_group *g);
int fdc_sinks_group_pdus(const fdc_sinks_group *g, fdc_pdu *out, int cap);           /* like fdc_sinks_pdus */
int32_t fdc_sinks_group_size(const fdc_sinks_group *g);
fdc_sinks *fdc_sinks_group_member(fdc_sinks_group *g, int i);                        /* owned by the group; NULL for a member without work */
/* member i: its device, the band [lo, hi) of bins it copies, how many PowerActivationChannels and segments it holds */
int fdc_sinks_group_member_info(const fdc_sinks_group *g, int i, int32_t *device, int32_t *lo, int32_t *hi, int32_t *npac, int32_t *nseg);

/* ------------------------------------------------------------------------------------------------
 * Single-block faces (same arithmetic as the fused pipeline, one reference block each).
 * ---------------------------------------------------------------------------------------------- */
/* gr::FDC::overlap_save::make(itemsize, outputlen, overlaplen) — include/FDC/overlap_save.h:49,
 * lib/overlap_save_impl.cc:41-81.  Type-agnostic byte copies on the device. */
typedef struct fdc_overlap_save fdc_overlap_save;
int fdc_overlap_save_create(int device_id, int itemsize, int outputlen, int overlaplen, fdc_overlap_save **out);
int fdc_overlap_save_work(fdc_overlap_save *b, const void *in, int nitems, void *out);
void fdc_overlap_save_destroy(fdc_overlap_save *b);

/* gr::FDC::vector_cut_vxx::make(itemsize, veclen, offset, blocklen) — include/FDC/vector_cut_vxx.h:49,
 * lib/vector_cut_vxx_impl.cc:41-72. */
typedef struct fdc_vector_cut fdc_vector_cut;
int fdc_vector_cut_create(int device_id, int itemsize, int veclen, int offset, int blocklen, fdc_vector_cut **out);
int fdc_vector_cut_work(fdc_vector_cut *b, const void *in, int nitems, void *out);
void fdc_vector_cut_destroy(fdc_vector_cut *b);

/* gr::FDC::phase_shifting_windowing_vcc::make(blocklen, numphasestates, shifts, passbw, stopbw, windowtype)
 * — include/FDC/phase_shifting_windowing_vcc.h:49, lib/phase_shifting_windowing_vcc_impl.cc:41-86.
 * create fails with FDC_ERR_INVALID_ARGUMENT on the ctor's predicates (:46-53). */
typedef struct fdc_phase_window fdc_phase_window;
int fdc_phase_window_create(int device_id, int blocklen, int numphasestates, int shifts, float passbw,
                            float stopbw, int windowtype, fdc_phase_window **out);
int fdc_phase_window_work(fdc_phase_window *b, const void *in, int nitems, void *out);
void fdc_phase_window_destroy(fdc_phase_window *b);

/* Window table generator used by the faces above (lib/windows.h:41-78 cr_win): w receives
 * [numphasestates][blocklen] complex float32.  Host-side, no device needed. */
int fdc_window_table(int windowtype, int blocklen, float passbw, float stopbw, int numphasestates,
                     int step, int normalize, float *w);

/* Stand-alone batched FFT with the semantics of gr-fft fft_vcc(n, forward, rectangular, shift, *)
 * (python/FrequencyDomainChannelizer.py:206,228) on host buffers: nitems transforms of length n. */
int fdc_fft_vcc(int device_id, int n, int forward, int shift, const void *in, int nitems, void *out);

#ifdef __cplusplus
}
#endif
#endif /* FDC_AMD_H */
