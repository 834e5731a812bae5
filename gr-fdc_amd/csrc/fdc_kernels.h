// Internal launcher interface between the C-ABI (fdc_api.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fdc {

constexpr int kThreads = 256;        // 4 wave64 per workgroup
constexpr int kMaxLdsFft = 8192;     // longest transform one workgroup does in LDS

// Per-channel record in device memory.
struct ChanDev {
    int32_t f;          // first bin in the shifted spectrum
    int32_t l;          // slice / IFFT length
    int32_t lout;       // l - l/R
    int32_t shift;      // ((f mod R)+R) mod R   (lib/phase_shifting_windowing_vcc_impl.cc:58)
    int64_t out_off;    // sum of lout over preceding channels (samples per block)
    int32_t win_off;    // offset of this channel's [R][l] table in the window pool (complex elements)
    int32_t pad;
};

// Sinks (PowerActivationChannel / activity_detection_channelizer_vcm): power cells and extraction tasks.
struct PowerCell { int32_t start, len; float scale; int32_t pad; };
struct ExtractTask {
    int32_t slot;       // spectrum block slot (0 = history block of the previous call, 1.. = blocks of this call)
    int32_t start;      // first bin of the slice
    int32_t win_off;    // offset of the (phase-resolved) window in the pool, complex elements
    int32_t pad;
    int64_t out_off;    // where the w - skip output samples go, complex elements
};

// Geometry of an LDS "column" FFT tile: TC independent transforms of length L, element (i, t) at i*ld + t.
struct TileGeom {
    int L, log2L, TC, log2TC, ld, NB;   // NB: 4096-point units per tile (2 only for L = 8192)
    size_t lds_bytes() const { return (size_t)L * ld * sizeof(float2); }
};
TileGeom tile_geom(int L);

// Two-pass split N = N1*N2 used above kMaxLdsFft.
struct BigGeom { int N, N1, N2; TileGeom a /* length N2 */, b /* length N1 */; };
BigGeom big_geom(int N);

hipError_t init_kernels();   // raises the dynamic-LDS limits once per process

// Forward/inverse batched FFT of `nitems` transforms of length N (any power of two >= 2).
//   in : item m starts at in + m*in_stride (complex elements); element n is read at (n + in_rot) mod N
//   out: item m at out + m*N; result bin k is stored at (k + out_rot) mod N, multiplied by scale
//   tw : exp(-2 pi i k / ntab), k in [0, ntab), ntab >= N, ntab % N == 0
//   tmp: nitems*N complex scratch (two-pass sizes only)
hipError_t launch_fft(const float2 *in, size_t in_stride, float2 *out, float2 *tmp, int N, int nitems,
                      bool inverse, int in_rot, int out_rot, float scale, const float2 *tw, int ntab,
                      hipStream_t s, hipEvent_t *ev /* null or 3 events: start, after pass A, end */,
                      const float2 *twf = nullptr /* two-pass sizes: [k2][n1] = exp(-2 pi i n1*k2/N), N entries, optional */,
                      bool generic_only = false /* no size-specific register kernels (FDC_FORCE_GENERIC handles) */,
                      unsigned long long keep4096 = ~0ull /* N = 4096: 64-bin groups of the output that are wanted (others may stay unwritten) */);

// Fused slice + phase + window + ifftshift + IFFT(l) + overlap discard + *l for one group of channels of
// equal l (<= kMaxLdsFft).  spec holds nb_chunk spectra; block m of the chunk is block mbase+m of the call.
hipError_t launch_channels(const float2 *spec, float2 *out, const ChanDev *chans, const int32_t *group,
                           int ngroup, int l, int N, int R, int nb_chunk, int mbase, int nb_call,
                           int64_t first_block, const float2 *wins, const float2 *tw, int ntab, hipStream_t s);

// ---- fast path (fdc_fast256.hip): N = 65536 forward transform, l = 256 channels ------------------------
//   tw256: exp(-2 pi i j/256), j in [0,256);  twf: [k2][n1] = exp(-2 pi i n1*k2/65536)
hipError_t init_fast_kernels();
hipError_t launch_fft65536(const float2 *in, size_t in_stride, float2 *out, float2 *tmp, int nitems, int out_rot,
                           float scale, const float2 *tw256, const float2 *twf, hipStream_t s, hipEvent_t *ev);
// needs an even discard length 256/R; aligned = all f even; out_aligned = all out_off and lout even
hipError_t launch_channels256(const float2 *spec, float2 *out, const ChanDev *chans, const int32_t *group,
                              int ngroup, bool aligned, bool out_aligned, int N, int R, int nb_chunk, int mbase,
                              int nb_call, int64_t first_block, const float2 *wins, const float2 *tw256,
                              hipStream_t s);

// the same for l = 512 and l = 1024 (fdc_chanwide.hip: 32 points per lane, DFT-32 layers); needs an even discard length l/R
hipError_t init_wide_kernels();
hipError_t launch_channels_wide(const float2 *spec, float2 *out, const ChanDev *chans, const int32_t *group, int ngroup, int l,
                                int N, int R, int nb_chunk, int mbase, int nb_call, int64_t first_block, const float2 *wins,
                                const float2 *tw, int ntab, hipStream_t s);

// 4096-point transforms in registers (fdc_chanwide.hip), same meaning of the arguments as launch_fft
hipError_t launch_fft4096(const float2 *in, size_t in_stride, float2 *out, int nitems, bool inverse, int in_rot, int out_rot,
                          float scale, const float2 *tw, int ntab, hipStream_t s,
                          unsigned long long keep = ~0ull /* 64-bin groups of the output some reader wants */);

// N = 4096 in one launch (fdc_fused4096.hip): forward transform + every channel's slice / window / inverse transform, the spectrum stays in LDS.
// A 512-thread workgroup takes a pair of blocks.  The schedule is the host's (fdc_api.hip plan_fused4096): rows[8 waves][8 slots], four bits per
// wave in wcls (0 = no rows, 1 = l = 256 slots 0..3, 2 = l = 256 slots 0..7, 3 = l = 512 slots 0..3, 4 = l = 1024 slots 0..1, 5 / 6 / 7 / 8 = l = 128 / 64 / 32 / 16
// slots 0..7); valid = 0: no row,
// 1 + k: a row of the pair's block k; xch = the row's exchange area in the two tiles (points), rows disjoint, all inside 2 fused4096_tile_points().
// Unused slots: everything 0 except lout.
struct F4Row { int32_t f, win_off, shift, xch, lout, valid; long long out_off; };
hipError_t init_fused4096_kernels();
int fused4096_tile_points();
hipError_t launch_fused4096(const float2 *in, size_t in_stride, float2 *out, int nb_chunk, int R, int mbase, int nb_call, int64_t first_block,
                            const float2 *tw, int ntab, const float2 *wins, const F4Row *rows, unsigned wcls, int teams /* blocks per workgroup: 1 (rows[4][8], no wide rows) or 2 (rows[8][8]) */, hipStream_t s);

// uniform plan (all channels l = 256, f = 256*slot, N = 256*N1): stage 1 + stage 2, no spectrum in memory.
//   twq[n1][q] = W_N^(16*n1*q), cbt[n1][b] = (-1)^n1 W_N^(n1*b)  (16 entries per n1 each), shn[k2] = shape[k2]/N;
//   slot_off[c] = per-block sample offset of the channel sitting in slot c, or -1;  g: nb_chunk*lout*N1 scratch
// one entry per channel: where the caller's (pinned, device-mapped) buffer of that channel is
struct ScatterEnt { float2 *dst; long long out_off; int lout; int pad; };
// dst[c][row0*lout_c + i] = src[nb*out_off_c + i], i < nb*lout_c: the [channel][nb*lout] result of one sub-batch is
// stored straight into the caller's per-channel host buffers (PCIe writes, 512 B per wave)
hipError_t launch_scatter_out(const float2 *src, const ScatterEnt *tab, int nchan, int nb, long long row0, hipStream_t s);

hipError_t launch_poly_stage1(const float2 *in, size_t in_stride, float2 *g, int N1, int R, int nb_chunk,
                              const float2 *tw256, const float2 *twq, const float2 *cbt, const float *shn,
                              int ncu /* compute units of the device */, hipStream_t s);
hipError_t launch_poly_stage2(const float2 *g, float2 *out, int N1 /* 256 or 1024 slots */, int R, int nb_chunk, int mbase,
                              int nb_call, const float2 *tw256, const float2 *tw1024 /* N1 = 1024 only */,
                              const long long *slot_off, unsigned out_bytes /* whole d_out, < 4 GiB */,
                              int ncu, hipStream_t s);

// stage 2 for any other slot count N1 = N/256 in [16, 4096] (generic LDS core, fdc_kernels.hip)
hipError_t launch_poly_stage2_generic(const float2 *g, float2 *out, int N1, int R, int nb_chunk, int mbase, int nb_call,
                                      const long long *slot_off, const float2 *tw, int ntab, hipStream_t s,
                                      int L = 256 /* channel width; other than 256: G in rows of N1 columns (launch_poly_stage1_generic) */);
// uniform plans of any power-of-two width L on the L-bin grid: stage 1 on the generic LDS core; g: nb_chunk * (L - L/R) * N/L points,
// shn[k2] = shape[k2] / N (L entries), tw = exp(-2 pi i k / ntab) with ntab = N
// t2[k2][t] = W_N^(t k2), t < poly_stage1_generic_tile_columns(N, L): the tile-local factor of the inter-pass twiddle
hipError_t launch_poly_stage1_generic(const float2 *in, size_t in_stride, float2 *g, int N, int L, int R, int nb_chunk, const float *shn,
                                      const float2 *tw, int ntab, const float2 *t2, hipStream_t s);
int poly_stage1_generic_tile_columns(int N, int L);

// uniform plan, N = 16384 / 32768 / 65536, R = 2 or 4: the whole path in one kernel, one block per CU, G kept in registers (fdc_block256.hip).
// hints: 1 = nt output stores, 2 = nt input loads.  ncu: compute units of the device (grid size).
hipError_t init_block_kernels();
hipError_t launch_poly_block(const float2 *in, size_t in_stride, float2 *out, int nb_chunk, int mbase, int nb_call,
                             const float2 *tw256, const float2 *twq, const float2 *cbt, const float *shn,
                             const long long *slot_off, unsigned out_bytes, int ncu, int hints, hipStream_t s,
                             unsigned long long *dbg = nullptr /* diagnostics: 8 x 4 x 32 cycle stamps of workgroup 0 */,
                             int r = 0 /* common offset f mod 256 of the channels; cbt must then hold (-1)^n1 W_N^(n1 (b + r)) */,
                             long long first_block = 0 /* global index of block 0 of this launch (window phase of odd r) */,
                             hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr /* timing: stamped by the dispatch itself */,
                             int R = 2 /* 2 or 4 */, float2 *scratch = nullptr /* R = 4: ncu x 32768 points */,
                             int N = 65536 /* block length: 16384, 32768 or 65536 (poly_block_supports) */);
bool poly_block_supports(int N);

// uniform plan of 512-bin channels on the 512-bin grid, N = 65536, R = 2 or 4: one kernel, one block per CU (fdc_block512.hip): the two
// parities of a column's 512 rows run the 256-point machinery side by side in the lanes of a quad.
//   tw512[k] = W_512^k (k < 256); twq[n1][q] = W_N^(16 n1 q) (128 x 16); cbt[n1][b + 16 h] = (-1)^n1 W_N^(n1 (b + 256 h)) (128 x 32);
//   shn[k2] = shape[k2] / N (512); slot_off[128]
hipError_t init_block512_kernels();
hipError_t launch_poly_block512(const float2 *in, size_t in_stride, float2 *out, int nb_chunk, int mbase, int nb_call, const float2 *tw256,
                                const float2 *tw512, const float2 *twq, const float2 *cbt, const float *shn, const long long *slot_off,
                                unsigned out_bytes, int ncu, int hints, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr,
                                int R = 2 /* 2 or 4 */, float2 *scratch = nullptr /* R = 4: ncu x 32768 points */,
                                bool half = false /* the bank at f = 512 slot + 256: shn and cbt with their halves swapped, W_N^(256 n1) in cbt */,
                                int N = 65536 /* block length: 16384, 32768 or 65536 (poly_block512_supports) */);
bool poly_block512_supports(int N, int R);

// uniform plan of 1024-bin channels on the 1024-bin grid, N = 65536, R = 2 or 4: one kernel, one block per CU (fdc_block1024.hip): the four phases of a
// column's 1024 rows run the 256-point machinery side by side in the lanes of a quad.
//   tw1024[k] = W_1024^k (k < 1024); twq[n1][q] = W_N^(16 n1 q) (64 x 16); cbt[n1][b + 16 i] = (-1)^n1 W_N^(n1 (b + 256 i)) (64 x 64);
//   shn[k2] = shape[k2] / N (1024); slot_off[64]
hipError_t init_block1024_kernels();
hipError_t launch_poly_block1024(const float2 *in, size_t in_stride, float2 *out, int nb_chunk, int mbase, int nb_call, const float2 *tw256,
                                 const float2 *tw1024, const float2 *twq, const float2 *cbt, const float *shn, const long long *slot_off,
                                 unsigned out_bytes, int ncu, int hints, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr,
                                 bool half = false /* the bank at f = 1024 slot + 512: the quarters of shn and cbt moved by two, W_N^(512 n1) in cbt */,
                                 int R = 2 /* 2 or 4 */, float2 *scratch = nullptr /* R = 4: ncu x 16384 points */,
                                 int N = 65536 /* block length: 16384, 32768 or 65536 (poly_block1024_supports) */);
bool poly_block1024_supports(int N, int R);

// uniform plan of narrow channels (l = 128 or 64 bins on the l-bin grid), N = 16384 / 32768 / 65536, R = 2 or 4: one kernel, one block per CU
// (fdc_blocknarrow.hip): S = 256/l adjacent columns interleaved into one 256-point virtual column, separated and re-joined in registers.
//   tab = the table image of poly_block_narrow_tables() (shn[k2] = shape[k2] / N, l values); cbt[V][b] = W_N^(S V b) (N/256 x 16); slot_off[N / l]
bool poly_block_narrow_supports(int N, int L, int R);
hipError_t init_block_narrow_kernels();
int poly_block_narrow_table_points(int L, int N);
void poly_block_narrow_tables(int L, int N, const float *shn, float2 *img, bool half = false /* the bank at f = l slot + l/2; cbt then W_N^(S V (b + l/2)) */,
                              int r = 0 /* l/4 or 3l/4 (not with half): the bank at f = l slot + r; cbt then W_N^(S V (b + r)) */);
hipError_t launch_poly_block_narrow(int L, const float2 *in, size_t in_stride, float2 *out, int nb_chunk, int mbase, int nb_call, const float2 *tab,
                                    const float2 *cbt, const long long *slot_off, unsigned out_bytes, int ncu, int hints, hipStream_t s,
                                    hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr, int R = 2 /* 2 or 4 */,
                                    float2 *scratch = nullptr /* R = 4: ncu x 16384 points */,
                                    int r = 0 /* the bank's offset from the l-bin grid: 0, l/4, l/2, 3l/4 (tables to match) */, int N = 65536);

// forward transform of N-sample blocks (N = 16384 / 32768 / 65536) with the block kernel (both halves of k2 in one launch): shifted, 1/N-scaled spectrum
hipError_t launch_block_fft(int N, const float2 *in, size_t in_stride, float2 *out, int nitems, const float2 *tw256,
                            const float2 *twq, const float2 *cbt0 /* r = 0 */, const float *shn1 /* 256 x 1/N */,
                            const long long *slot_off /* [N / 256]: 256 c */, float2 *scratch /* ncu x 32768 points */,
                            int ncu, int hints, hipStream_t s,
                            hipEvent_t *ev /* null or 3: start, end, end */,
                            const unsigned *keep = nullptr /* [N / 8192][4] words: 64-bin stores some channel reads (fdc_api.hip), or all */,
                            float *gpow = nullptr /* nitems x N / 16 floats: the power of every 16-bin group of the spectrum, summed in the epilogue */);
// the sinks' power cells from those group sums + the bins of the groups a cell cuts; the group sums of a spectrum in memory (launch groups the block kernel did not transform)
hipError_t launch_cell_power_groups(const float2 *spec, const float *gpow, int N, const PowerCell *cells, int ncells, int nblocks, float *out, hipStream_t s);
hipError_t launch_group_power(const float2 *spec, int N, int nblocks, float *gpow, hipStream_t s);

// real samples -> complex samples with zero imaginary part (the real-input front end)
hipError_t launch_real_to_complex(const float *in, float2 *out, size_t n, hipStream_t s);

hipError_t launch_scale(const float2 *in, float2 *out, size_t n, float k, hipStream_t s);

// sinks
hipError_t launch_cell_power(const float2 *spec, int N, const PowerCell *cells, int ncells, int nblocks, float *out,
                             hipStream_t s);
// one width class above 4096 points (two passes through `tmp`, ntasks * w points; slice * window read by pass A, kept samples
// written to their places by pass B)
hipError_t launch_extract_wide(const float2 *spec, int N, const ExtractTask *tasks, int ntasks, int w, int skip, const float2 *wins,
                               float2 *tmp, float2 *out, const float2 *tw, int ntab, hipStream_t s, float scale = 1.0f);
// the channels of one width above 4096 bins of a pipeline as such tasks: task m * ngroup + gi = channel group[gi] in block m of the launch group
hipError_t launch_wide_tasks(ExtractTask *tasks, const ChanDev *chans, const int32_t *group, int ngroup, int R, int nb_chunk, int mbase, int nb_call,
                             int64_t first_block, hipStream_t s);
// several width classes (w <= 4096 each) in one launch: class k = tasks [first[k], first[k] + cnt[k]) of width w[k]
constexpr int kMaxExtractClasses = 12;
struct ExtractClass { int32_t tile0, task0, ntasks, log2w, log2TC, ld, skip, pad; };
struct ExtractClasses { ExtractClass c[kMaxExtractClasses]; int32_t n; };
hipError_t launch_extract_multi(const float2 *spec, int N, const ExtractTask *tasks, const int *w, const size_t *first, const size_t *cnt,
                                int nclass, int R, const float2 *wins, float2 *out, const float2 *tw, int ntab, hipStream_t s);
hipError_t launch_extract(const float2 *spec, int N, const ExtractTask *tasks, int ntasks, int w, int skip,
                          const float2 *wins, float2 *out, const float2 *tw, int ntab, hipStream_t s);

// width 256 on the register kernel of the l = 256 channels (fdc_fast256.hip); tw256: exp(-2 pi i j/256)
hipError_t launch_extract256(const float2 *spec, int N, const ExtractTask *tasks, int ntasks, int skip, const float2 *wins, float2 *out,
                             const float2 *tw256, hipStream_t s);

// single-block faces
hipError_t launch_overlap_save(const unsigned char *ring, unsigned char *out, size_t in_item_bytes,
                               size_t out_item_bytes, int nitems, hipStream_t s);
hipError_t launch_vector_cut(const unsigned char *in, unsigned char *out, size_t in_item_bytes, size_t shift_bytes,
                             size_t out_item_bytes, int nitems, hipStream_t s);
hipError_t launch_phase_window(const float2 *in, float2 *out, const float2 *win, int l, int R, int shift,
                               int counter0, int nitems, hipStream_t s);

}  // namespace fdc
