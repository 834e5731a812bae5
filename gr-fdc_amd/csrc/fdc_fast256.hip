// gfx950 fast path for the headline geometry: N = 65536 = 256 x 256 forward transform and l = 256 channels.
//
// Every 256-point transform is two in-register DFT-16 layers (fdc_radix16.hpp) with ONE exchange through
// LDS in between; a 256-thread workgroup owns a tile of 8192 points (64 KiB of LDS, two workgroups per CU),
// each thread 32 points = two DFT-16 per layer.  Global traffic is 16 B per lane in 256-B (column kernels)
// or 128-B (row kernels) contiguous runs; LDS traffic is b128 reads and conflict-free b64/b128 writes.
//
//   k_a256  pass A: overlap-save gather + FFT over n2 for 32 columns n1, times W_N^(n1*k2), T[k2][n1]
//   k_b256  pass B: FFT over n1 for 32 rows k2, fftshift + 1/N folded into the store
//   k_c256  fused channel kernel: slice + phase/window + ifftshift + IFFT-256 + discard + *l, 32 (block,
//           channel) pairs per workgroup
// Index algebra: n = n1 + 256*n2, k = 256*k1 + k2 (same as the generic two-pass path in fdc_kernels.hip).
#include "fdc_kernels.h"
#include "fdc_radix16.hpp"
#include "fdc_devutil.hpp"
#include <cstdlib>

// k_c256's output stores streamed (nt) like those of k_c512 / k_c1024 (fdc_devutil.hpp st2_out): -2 % on the channel kernels of a mixed plan, -1.3 % on the
// example plan's, +1.4 % on a full band of 256-bin channels forced onto the spectrum path (profiles/r06/ch_nt_stores_ab.txt); -DFDC_C256_NT=0 builds the plain form
#ifndef FDC_C256_NT
#define FDC_C256_NT 1
#endif
namespace fdc {

extern __shared__ __attribute__((aligned(16))) unsigned char fdc_smem_fast[];

constexpr int kTileBytes = 256 * 32 * 8;          // 64 KiB of points
// Default: the streams that are touched once (stage-1 input rows, stage-2 output samples) carry the nt hint, so that G — written by
// stage 1 and read back by stage 2, about the size of the Infinity Cache per 1024-block launch — is what stays cached:
// measured 0.236 -> 0.212 ms per step (bits 4 and 8, hints on G itself, lose: profiles/r01/NOTES.md)
constexpr int kDefaultNtHints = 3;
constexpr int kP2kLds = 1024 * 16 * 8 + 68 * 18 * 8 + 4096;   // k_p2k: 16-row tile + twiddle rows + slot offsets

// ---- pass A -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_a256(const float2 *__restrict__ in, size_t in_stride,
                                                 float2 *__restrict__ tmp, const float2 *__restrict__ tw256,
                                                 const float2 *__restrict__ twf)
{
    float2 *tile = reinterpret_cast<float2 *>(fdc_smem_fast);
    float2 *w256 = reinterpret_cast<float2 *>(fdc_smem_fast + kTileBytes);
    const int tid = threadIdx.x, cp = tid & 15, b = tid >> 4;
    const int c0 = blockIdx.x * 32;
    const size_t m = blockIdx.y;
    const float2 *src = in + m * in_stride + c0 + 2 * cp;
    cf va[16], vb[16];
#pragma unroll
    for (int a = 0; a < 16; a++) {          // rows n2 = 16a+b, two adjacent columns per lane
        const float4 t = ld4(src + (size_t)(16 * a + b) * 256);
        va[a] = mk(t.x, t.y); vb[a] = mk(t.z, t.w);
    }
    w256[tid] = tw256[tid];
    dft16<false>(va); dft16<false>(vb);
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 16; p++) {          // W_256^(b*p), then row 16b+p of the exchange tile
        const cf w = ld2(&w256[b * p]);
        st4(&tile[(16 * b + p) * 32 + 2 * cp], cmul(va[rev16(p)], w), cmul(vb[rev16(p)], w));
    }
    __syncthreads();
#pragma unroll
    for (int bb = 0; bb < 16; bb++) {       // this thread now owns p' = b
        const float4 t = ld4(&tile[(16 * bb + b) * 32 + 2 * cp]);
        va[bb] = mk(t.x, t.y); vb[bb] = mk(t.z, t.w);
    }
    dft16<false>(va); dft16<false>(vb);
    float2 *dst = tmp + m * 65536 + c0 + 2 * cp;
    const float2 *twp = twf + c0 + 2 * cp;
#pragma unroll
    for (int q = 0; q < 16; q++) {          // k2 = b + 16q ; inter-pass twiddle W_N^(n1*k2)
        const int k2 = b + 16 * q;
        const float4 w = ld4(twp + k2 * 256);
        st4(dst + k2 * 256, cmul(va[rev16(q)], mk(w.x, w.y)), cmul(vb[rev16(q)], mk(w.z, w.w)));
    }
}

// ---- pass B -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_b256(const float2 *__restrict__ tmp, float2 *__restrict__ spec,
                                                 const float2 *__restrict__ tw256, int out_rot, float scale)
{
    float2 *tile = reinterpret_cast<float2 *>(fdc_smem_fast);
    float2 *w256 = reinterpret_cast<float2 *>(fdc_smem_fast + kTileBytes);
    const int tid = threadIdx.x;
    const int r0 = blockIdx.x * 32;
    const size_t m = blockIdx.y;
    cf va[16], vb[16];
    {
        const int r = tid >> 3, bp = tid & 7;          // row k2 = r0+r, columns n1 = 16a + {2bp, 2bp+1}
        const float2 *src = tmp + m * 65536 + (size_t)(r0 + r) * 256 + 2 * bp;
#pragma unroll
        for (int a = 0; a < 16; a++) {
            const float4 t = ld4(src + 16 * a);
            va[a] = mk(t.x, t.y); vb[a] = mk(t.z, t.w);
        }
        w256[tid] = tw256[tid];
        dft16<false>(va); dft16<false>(vb);
        __syncthreads();
        const int col = r ^ (bp << 1);                 // XOR swizzle: conflict-free b64 writes
#pragma unroll
        for (int p = 0; p < 16; p++) {
            st2(&tile[(p * 16 + 2 * bp) * 32 + col], cmul(va[rev16(p)], ld2(&w256[(2 * bp) * p])));
            st2(&tile[(p * 16 + 2 * bp + 1) * 32 + col], cmul(vb[rev16(p)], ld2(&w256[(2 * bp + 1) * p])));
        }
    }
    __syncthreads();
    {
        const int rp = tid & 15, p = tid >> 4;         // rows k2 = r0 + {2rp, 2rp+1}, output k1 = p + 16q
#pragma unroll
        for (int bb = 0; bb < 16; bb++) {
            const float4 t = ld4(&tile[(p * 16 + bb) * 32 + ((2 * rp) ^ ((bb >> 1) << 1))]);
            va[bb] = mk(t.x, t.y); vb[bb] = mk(t.z, t.w);
        }
        dft16<false>(va); dft16<false>(vb);
        float2 *dst = spec + m * 65536;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int k = ((p + 16 * q) << 8) + r0 + 2 * rp;          // bin 256*k1 + k2
            st4(dst + ((k + out_rot) & 65535), va[rev16(q)] * scale, vb[rev16(q)] * scale);
        }
    }
}

// ---- fused channel kernel, l = 256 ------------------------------------------------------------------------
// One row = one (block, channel) pair; 16 rows per workgroup, the 16 threads of a row are 16 neighbouring lanes, every
// thread holds 16 points in both layers: slice bins i = 16a + b (vector_cut_vxx) times W[cnt][i]
// (phase_shifting_windowing_vcc), ifftshift as a -> a ^ 8, DFT-16 over a, twiddle, 16x16 exchange INSIDE the row
// (LDS, row-padded so two rows of a half-wave use different bank halves), DFT-16 over b, keep t >= l/R, times l.
// Loads and stores are 128-byte runs per row; 32 KiB of LDS per workgroup, four workgroups per CU.
struct RowInfo { long long src; long long dst; int win; int valid; };
constexpr int kCRowPts = 272;
constexpr int kCTileBytes = 16 * kCRowPts * 8;

__global__ __launch_bounds__(256, 4) void k_c256(const float2 *__restrict__ spec, float2 *__restrict__ out,
                                                 const ChanDev *__restrict__ chans,
                                                 const int32_t *__restrict__ group, int ngroup, int N, int R,
                                                 int nb_chunk, int mbase, int nb_call, long long first_block,
                                                 const float2 *__restrict__ wins,
                                                 const float2 *__restrict__ tw256)
{
    __shared__ RowInfo rows[16];
    float2 *tile = reinterpret_cast<float2 *>(fdc_smem_fast);
    float2 *wrow = reinterpret_cast<float2 *>(fdc_smem_fast + kCTileBytes);              // [b][p] = W256^(b p), 16 x 18
    const int tid = threadIdx.x, r = tid >> 4, b = tid & 15;
    const long long ntrans = (long long)nb_chunk * ngroup;
    const int lout = 256 - 256 / R, skip = 256 - lout;
    if (tid < 16) {
        const long long t = (long long)blockIdx.x * 16 + tid;
        RowInfo ri{0, 0, 0, 0};
        if (t < ntrans) {
            const int m = (int)(t / ngroup);
            const ChanDev ch = chans[group[(int)(t - (long long)m * ngroup)]];
            const int cnt = (int)((((first_block + mbase + m) % R) * ch.shift) % R);
            ri.src = (long long)m * N + ch.f;
            ri.win = ch.win_off + cnt * 256;
            ri.dst = (long long)nb_call * ch.out_off + (long long)(mbase + m) * lout - skip;
            ri.valid = 1;
        }
        rows[tid] = ri;
    }
    wrow[(tid >> 4) * 18 + (tid & 15)] = tw256[((tid >> 4) * (tid & 15)) & 255];
    __syncthreads();
    const RowInfo ri = rows[r];
    cf v[16];
    {
        cf x[16], w[16];
#pragma unroll
        for (int a = 0; a < 16; a++) {
            x[a] = mk(0.f, 0.f); w[a] = x[a];
            if (ri.valid) { x[a] = ld2(spec + ri.src + 16 * a + b); w[a] = ld2(wins + ri.win + 16 * a + b); }
        }
#pragma unroll
        for (int a = 0; a < 16; a++) v[a ^ 8] = cmul(x[a], w[a]);
    }
    dft16<true>(v);
    float2 *row = tile + r * kCRowPts;
    {
        cf w[16];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float4 t = ld4(&wrow[b * 18 + 2 * i]);
            w[2 * i] = mk(t.x, t.y); w[2 * i + 1] = mk(t.z, t.w);
        }
        // element (b, p) of the row at p*16 + (b ^ p): writes of a fixed p and reads of a fixed b are both conflict-free
#pragma unroll
        for (int p = 0; p < 16; p++) st2(&row[p * 16 + (b ^ p)], cmulc(v[rev16(p)], w[p]));   // inverse: conjugate twiddles
    }
    __syncthreads();
#pragma unroll
    for (int bb = 0; bb < 16; bb++) v[bb] = ld2(&row[b * 16 + (bb ^ b)]);      // this thread now plays p = b
    dft16<true>(v);
    if (ri.valid) {
        // y[t], t = p + 16q; keep t >= l/R (vector_cut_vxx(l, l-lout, lout)), times l (multiply_const_cc)
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int tt = b + 16 * q;
            if (tt >= skip) {
#if FDC_C256_NT
                st2_out(out + ri.dst + tt, v[rev16(q)] * 256.f);
#else
                st2(out + ri.dst + tt, v[rev16(q)] * 256.f);
#endif
            }
        }
    }
}

// The same row machinery driven by extraction tasks of the sinks (width 256: a bank of 256-bin PowerActivationChannels, detected
// carriers of that class): slice of spectrum slot `slot` from bin `start`, the task's phase-resolved window, halves swapped,
// IFFT-256, the first 256/R samples dropped, results at the task's landing offset.  Replaces the generic LDS kernel k_extract<1>
// for this width (1.9 -> ~4 TB/s).
__global__ __launch_bounds__(256, 4) void k_x256(const float2 *__restrict__ spec, float2 *__restrict__ out,
                                                 const ExtractTask *__restrict__ tasks, int ntasks, int N, int skip,
                                                 const float2 *__restrict__ wins,
                                                 const float2 *__restrict__ tw256)
{
    __shared__ RowInfo rows[16];
    float2 *tile = reinterpret_cast<float2 *>(fdc_smem_fast);
    float2 *wrow = reinterpret_cast<float2 *>(fdc_smem_fast + kCTileBytes);              // [b][p] = W256^(b p), 16 x 18
    const int tid = threadIdx.x, r = tid >> 4, b = tid & 15;
    if (tid < 16) {
        const long long t = (long long)blockIdx.x * 16 + tid;
        RowInfo ri{0, 0, 0, 0};
        if (t < ntasks) {
            const ExtractTask tk = tasks[t];
            ri.src = (long long)tk.slot * N + tk.start;
            ri.win = tk.win_off;
            ri.dst = tk.out_off - skip;
            ri.valid = 1;
        }
        rows[tid] = ri;
    }
    wrow[(tid >> 4) * 18 + (tid & 15)] = tw256[((tid >> 4) * (tid & 15)) & 255];
    __syncthreads();
    const RowInfo ri = rows[r];
    cf v[16];
    {
        cf x[16], w[16];
#pragma unroll
        for (int a = 0; a < 16; a++) {
            x[a] = mk(0.f, 0.f); w[a] = x[a];
            if (ri.valid) { x[a] = ld2(spec + ri.src + 16 * a + b); w[a] = ld2(wins + ri.win + 16 * a + b); }
        }
#pragma unroll
        for (int a = 0; a < 16; a++) v[a ^ 8] = cmul(x[a], w[a]);
    }
    dft16<true>(v);
    float2 *row = tile + r * kCRowPts;
    {
        cf w[16];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float4 t = ld4(&wrow[b * 18 + 2 * i]);
            w[2 * i] = mk(t.x, t.y); w[2 * i + 1] = mk(t.z, t.w);
        }
        // element (b, p) of the row at p*16 + (b ^ p): writes of a fixed p and reads of a fixed b are both conflict-free
#pragma unroll
        for (int p = 0; p < 16; p++) st2(&row[p * 16 + (b ^ p)], cmulc(v[rev16(p)], w[p]));   // inverse: conjugate twiddles
    }
    __syncthreads();
#pragma unroll
    for (int bb = 0; bb < 16; bb++) v[bb] = ld2(&row[b * 16 + (bb ^ b)]);      // this thread now plays p = b
    dft16<true>(v);
    if (ri.valid) {
        // y[t], t = p + 16q; keep t >= w/R; no scaling (PowerActivationChannel_impl.cc:277-281, …vcm_impl.cc:390-394)
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int tt = b + 16 * q;
            if (tt >= skip) st2(out + ri.dst + tt, v[rev16(q)]);
        }
    }
}

// ---- uniform-plan path ("polyphase commutation") ------------------------------------------------------------
// When every channel has l = 256 and f = 256*slot (the tiled plans of BASELINE configs[1] and [3]), pass B of the
// forward transform (FFT over n1) commutes with the per-channel window + IFFT (which act on k2 only):
//   y_c[t] = (-1)^t sum_n1 W_N1^(n1*k1) G[n1][t],   G[n1][t] = IFFT_k2{ W0[k2] W_N^(n1*k2) A[n1][k2] },  k1 = c + N1/2
// so the spectrum never has to exist in memory.  Stage 1 (k_p1) does, per column n1: overlap-save gather,
// FFT-256 over n2, times the combined table (window * inter-pass twiddle * (-1)^n1 * l/N), ifftshift, IFFT-256,
// overlap discard, and stores G[t'][n1]; stage 2 (k_p2) is pass B on G: FFT over n1, result index = channel slot.
// Exact algebra (SURVEY.md App. A.2-A.4 substituted into each other); rounding differs from the 3-kernel
// path at the 1e-7 level.
// Persistent: each workgroup (TC*16 threads, one column of 16 points per thread and layer) owns ONE column tile and a run
// of CONSECUTIVE blocks, and issues the next block's global loads before it starts computing the current one (register
// double buffer), so HBM latency hides under the four DFT-16 layers and two LDS exchanges of the current tile.
// Consecutive blocks overlap by N/R samples = 16*KEEP rows of the tile: with KEEP > 0 (= 16/R, the launcher picks it
// when items are exactly N - N/R apart) those rows are handed over in registers — row 16a+b of the next block IS row
// 16(a + 16 - KEEP) + b of this one — and only 16 - KEEP rows per thread are loaded (R = 2: half the input reads;
// measured 0.144 -> 0.129 ms per 1024 blocks).  TC = 32: 512 threads, 2 workgroups/CU; TC = 16: 256 threads, 4/CU.
// ABL (diagnostic builds only, FDC_ABLATE env): 0 = real kernel, 1 = memory only (loads -> stores, no math, no LDS),
// 2 = math + LDS only (one load per thread, data kept live), 3 = loads + math, one store per thread, 4 = one load per thread + math + all
// stores; results are garbage for ABL != 0.
template <int TC, int ABL, int KEEP>
__global__ __launch_bounds__(TC * 16, 4) void k_p1(const float2 *__restrict__ in, size_t in_stride,
                                                   float2 *__restrict__ g, const float2 *__restrict__ tw256,
                                                   const float2 *__restrict__ twq, const float2 *__restrict__ cbt,
                                                   const float *__restrict__ shn, int N1, int log2ct, int nb, int bpg,
                                                   int qskip, int lout, int hints)
{
    constexpr int NT = TC * 16;
    float2 *tile = reinterpret_cast<float2 *>(fdc_smem_fast);                           // [256][TC]
    // tables, laid out so that a thread reads ITS 16 entries as 16-byte pairs (half the LDS instructions of 8-byte reads);
    // rows padded to 18 entries (144 B): the rows a wave touches then start in different bank groups
    float2 *wrow = reinterpret_cast<float2 *>(fdc_smem_fast + 256 * TC * 8);             // [b][p] = W256^(b p), 16 x 18
    float2 *tq = reinterpret_cast<float2 *>(fdc_smem_fast + 256 * TC * 8 + 2304);       // [col][q], TC x 18
    float *sh = reinterpret_cast<float *>(fdc_smem_fast + 256 * TC * 8 + 2304 + TC * 144);   // [b][q] = shape[b + 16 q] / N
    const int tid = threadIdx.x, col = tid & (TC - 1), b = tid / TC;
    // this workgroup: column tile (blockIdx mod ct) and runs of bpg consecutive blocks, run r = group, group + ngroups, ...
    const int c0 = (blockIdx.x & ((1 << log2ct) - 1)) * TC;
    const int grp = blockIdx.x >> log2ct, ngrp = gridDim.x >> log2ct;
    if (grp * bpg >= nb) return;
    // The factor the spectrum column is multiplied by,
    //   shape[k2]/N * (-1)^n1 * W_N^(n1*k2),  k2 = b + 16q,
    // is split into  shape[k2]/N (LDS, wave-uniform)  *  W_N^(16*n1*q) (LDS, TCx16 entries for this column tile)
    // *  (-1)^n1 W_N^(n1*b) (one register pair per thread, applied AFTER the first inverse DFT-16, which is
    // linear in it) — three short tables instead of a 64-register slice of the full N-entry table.
    for (int i = tid; i < 256; i += NT) {
        wrow[(i >> 4) * 18 + (i & 15)] = tw256[((i >> 4) * (i & 15)) & 255];
        sh[i] = shn[(i >> 4) + 16 * (i & 15)];
    }
    tq[col * 18 + b] = twq[(size_t)(c0 + col) * 16 + b];    // b plays q here
    cf cb = ld2(&cbt[(size_t)(c0 + col) * 16 + b]);
    vm_settle(cb);                                          // no compiler-visible load is pending when the loop starts (fdc_devutil.hpp)
    // per-lane byte offset inside a block (row b, column c0+col); rows 16a+b add a*16*N1*8 bytes (scalar)
    const unsigned voff = (unsigned)(b * N1 + c0 + col) * 8u;
    const unsigned rowstep = 16u * (unsigned)N1 * 8u;
    const unsigned inbytes = 256u * (unsigned)N1 * 8u;
    const unsigned gtile = (unsigned)lout * TC * 8u;           // bytes of one (block, column tile) piece of G
    const unsigned goff = (unsigned)(b * TC + col) * 8u;       // row b of a 16-row group, column col
    const unsigned gstep = 16u * TC * 8u;                      // 16 rows further
    // rows to discard, in units of 16: with the register hand-over it is the template constant (qskip == KEEP == 16/R), so
    // the store predicates fold and the unused outputs of the last DFT-16 are never computed
    const int qs = KEEP > 0 ? KEEP : qskip;
    // One tile: consume `cur` (block m), prefetch block mn (the next block of the run: overlap handed over in registers;
    // or the first block of this workgroup's next run: all rows loaded; or none, mn < 0) into `nbuf`.
    auto do_tile = [&](cf (&cur)[16], cf (&nbuf)[16], int m, int mn) {
        // The row loads are inline assembly and the wait for them is stated at the END of the tile, behind this tile's stores
        // (vm_wait, fdc_devutil.hpp): left to the compiler, the loop header waits with vmcnt(0) — for the eight stores of the tile
        // before, once per tile (worth 1.2 % at N = 65536, nothing at N = 262144: profiles/r05/NOTES.md section 2).
        if (mn >= 0) {
            const srd_t rin = make_srd(in + (size_t)mn * in_stride, inbytes);
            if (ABL == 2 || ABL == 4) {
                nbuf[0] = ald2<false>(rin, voff, 0);
#pragma unroll
                for (int a = 1; a < 16; a++) nbuf[a] = nbuf[0] * (float)a;
            } else if (KEEP > 0 && mn == m + 1) {
#pragma unroll
                for (int a = 0; a < KEEP; a++) nbuf[a] = cur[a + 16 - KEEP];       // the overlap, already on chip
                if (hints & 2) {
#pragma unroll
                    for (int a = KEEP; a < 16; a++) nbuf[a] = ald2<true>(rin, voff, a * rowstep);
                } else {
#pragma unroll
                    for (int a = KEEP; a < 16; a++) nbuf[a] = ald2<false>(rin, voff, a * rowstep);
                }
            } else {
#pragma unroll
                for (int a = 0; a < 16; a++) nbuf[a] = ald2<false>(rin, voff, a * rowstep);
            }
        }
        // G is stored tile-major, G[m][column tile][t'][TC]: this workgroup's whole output (lout*TC points) is
        // one contiguous run, and stage 2 reads it back in runs of TC rows x TC columns.
        const __amdgpu_buffer_rsrc_t rg = make_rsrc(g + ((size_t)m * (size_t)(N1 / TC) + (c0 / TC)) * (size_t)lout * TC, gtile);
        if (ABL == 1) {
#pragma unroll
            for (int q = 0; q < 16; q++)
                if (q >= qs) bst2(rg, goff, (unsigned)(q - qs) * gstep, cur[q]);
            vm_wait<0>(nbuf);
            return;
        }
        dft16<false>(cur);
        __syncthreads();                                        // tables ready / previous tile's LDS reads done
        // twiddles first, as one batch: a table read between two tile writes would wait (lgkmcnt is in-order)
        // for the write in front of it, once per element
        cf w[16];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float4 t = ld4(&wrow[b * 18 + 2 * i]);
            w[2 * i] = mk(t.x, t.y); w[2 * i + 1] = mk(t.z, t.w);
        }
#pragma unroll
        for (int p = 0; p < 16; p++) st2(&tile[(16 * b + p) * TC + col], cmul(cur[rev16(p)], w[p]));
        __syncthreads();
        cf v[16];
#pragma unroll
        for (int bb = 0; bb < 16; bb++) v[bb] = ld2(&tile[(16 * bb + b) * TC + col]);
        dft16<false>(v);                              // A[k2 = b + 16q] in v[rev16(q)]
        // window * twiddle, placed at the ifftshifted position (k2 ^ 128  <=>  q ^ 8).  The thread already
        // holds exactly the inputs of ITS first inverse DFT-16 (fixed low digit b, all high digits q).
        cf u[16];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float4 t0 = ld4(&tq[col * 18 + 4 * i]), t1 = ld4(&tq[col * 18 + 4 * i + 2]);
            const float4 sv = *reinterpret_cast<const float4 *>(&sh[b * 16 + 4 * i]);
            u[(4 * i) ^ 8] = cmul(v[rev16(4 * i)], mk(t0.x, t0.y)) * sv.x;
            u[(4 * i + 1) ^ 8] = cmul(v[rev16(4 * i + 1)], mk(t0.z, t0.w)) * sv.y;
            u[(4 * i + 2) ^ 8] = cmul(v[rev16(4 * i + 2)], mk(t1.x, t1.y)) * sv.z;
            u[(4 * i + 3) ^ 8] = cmul(v[rev16(4 * i + 3)], mk(t1.z, t1.w)) * sv.w;
        }
        dft16<true>(u);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float4 t = ld4(&wrow[b * 18 + 2 * i]);
            w[2 * i] = mk(t.x, t.y); w[2 * i + 1] = mk(t.z, t.w);
        }
#pragma unroll
        for (int p = 0; p < 16; p++) u[rev16(p)] = cmul(cmulc(u[rev16(p)], w[p]), cb);
        __syncthreads();                                // every thread has finished reading the first exchange
#pragma unroll
        for (int p = 0; p < 16; p++) st2(&tile[(16 * b + p) * TC + col], u[rev16(p)]);
        __syncthreads();
#pragma unroll
        for (int bb = 0; bb < 16; bb++) u[bb] = ld2(&tile[(16 * bb + b) * TC + col]);
        dft16<true>(u);                               // y[t = b + 16q]
        // row t - skip of this block's G, t = b + 16q; skip = 16*qskip, so the test is wave-uniform
        if (hints & 8) {
#pragma unroll
            for (int q = 0; q < 16; q++)
                if (q >= qs && ((ABL != 2 && ABL != 3) || q == 15)) bst2_nt(rg, goff, (unsigned)(q - qs) * gstep, u[rev16(q)]);
        } else {
#pragma unroll
            for (int q = 0; q < 16; q++)
                if (q >= qs && ((ABL != 2 && ABL != 3) || q == 15)) bst2(rg, goff, (unsigned)(q - qs) * gstep, u[rev16(q)]);
        }
        // the next block's rows were requested a tile ago: waited for HERE, behind the stores (vmcnt counts in issue order: "at most
        // the stores of this tile outstanding" = every row load has landed, no store is waited for)
        if (KEEP > 0 && ABL == 0) vm_wait<16 - KEEP>(nbuf);
        else if (ABL == 2 || ABL == 3) vm_wait<1>(nbuf);
        else vm_wait<0>(nbuf);
    };
    cf L[16];
    {
        const srd_t rin = make_srd(in + (size_t)(grp * bpg) * in_stride, inbytes);
#pragma unroll
        for (int a = 0; a < 16; a++) L[a] = ald2<false>(rin, voff, a * rowstep);
        vm_wait<0>(L);
    }
    for (int run = grp; run * bpg < nb; run += ngrp) {
        const int m0 = run * bpg, m1 = m0 + bpg < nb ? m0 + bpg : nb;
        const int nextrun = (run + ngrp) * bpg < nb ? (run + ngrp) * bpg : -1;
        for (int m = m0; m < m1; m++) {
            cf cur[16];
#pragma unroll
            for (int a = 0; a < 16; a++) cur[a] = L[a];
            do_tile(cur, L, m, m + 1 < m1 ? m + 1 : nextrun);
        }
    }
}

// Stage 2 for N1 = 256 slots: rows rho = m*lout + t' of G (256 contiguous n1 each), FFT over n1, bin = slot c.
// Persistent with next-tile prefetch like k_p1; a tile is TR consecutive rows (TR*2 KiB contiguous).
template <int TR, int TCG>
__global__ __launch_bounds__(TR * 16, 4) void k_p2(const float2 *__restrict__ g, float2 *__restrict__ out,
                                                   const float2 *__restrict__ tw256,
                                                   const long long *__restrict__ slot_off, long long nrows,
                                                   long long out_base, long long nb_call, unsigned out_bytes,
                                                   int ntiles, int lout, int hints)
{
    constexpr int NT = TR * 16;
    float2 *tile = reinterpret_cast<float2 *>(fdc_smem_fast);                           // [p][b][TR]
    float2 *w256 = reinterpret_cast<float2 *>(fdc_smem_fast + 256 * TR * 8);
    unsigned *soff = reinterpret_cast<unsigned *>(fdc_smem_fast + 256 * TR * 8 + 2048);
    const int tid = threadIdx.x;
    int tl = blockIdx.x;
    if (tl >= ntiles) return;
    for (int i = tid; i < 256; i += NT) {
        w256[i] = tw256[i];
        const long long o = slot_off[i];
        // byte offset of row 0 of the slot's stream (the launcher guarantees the whole output spans < 4 GiB)
        soff[i] = o >= 0 ? (unsigned)((o * nb_call + out_base) * 8) : 0xFFFFFFFFu;
    }
    const int r = tid >> 4, b = tid & 15;            // layer 1: row r, points n1 = 16a + b
    const int r2 = tid & (TR - 1), p2 = tid / TR;    // layer 2: row r2, outputs k1 = p2 + 16q
    // G is tile-major (k_p1): G[m][ct][t'][TCG] with TCG columns per column tile.  Point n1 = 16a + b of row t'
    // sits in column tile ct = n1 / TCG at column n1 % TCG.  A tile of this kernel = TR consecutive rows t' of one
    // block (lout is a multiple of TR), i.e. for every ct a contiguous run of TR*TCG points.
    const __amdgpu_buffer_rsrc_t rout = make_rsrc(out, out_bytes);
    const int tpb = lout / TR;                                 // tiles per block
    const unsigned ctstep = (unsigned)lout * TCG * 8u;         // bytes between column tiles of one block
    // TCG = 16: ct = a, column = b.  TCG = 32: ct = a >> 1, column = 16*(a & 1) + b.
    const unsigned voff = (unsigned)(r * TCG + b) * 8u;
    cf L[16];                                        // loads as inline assembly, waited for behind the stores: see k_p2k
    auto issue = [&](int t) __attribute__((always_inline)) {
        const size_t m = t / tpb;
        const int t0 = (t - (int)m * tpb) * TR;
        const srd_t rg = make_srd(g + m * (size_t)lout * 256 + (size_t)t0 * TCG, (unsigned)lout * 256u * 8u);
        if (hints & 4) {
#pragma unroll
            for (int a = 0; a < 16; a++)
                L[a] = ald2<true>(rg, voff + (TCG == 32 ? (unsigned)(a & 1) * 128u : 0u), (unsigned)(TCG == 32 ? a >> 1 : a) * ctstep);
        } else {
#pragma unroll
            for (int a = 0; a < 16; a++)
                L[a] = ald2<false>(rg, voff + (TCG == 32 ? (unsigned)(a & 1) * 128u : 0u), (unsigned)(TCG == 32 ? a >> 1 : a) * ctstep);
        }
    };
    issue(tl);
    vm_wait<0>(L);
    for (;;) {
        cf v[16];
#pragma unroll
        for (int a = 0; a < 16; a++) v[a] = L[a];
        const int nxt = tl + gridDim.x;
        if (nxt < ntiles) issue(nxt);
        dft16<false>(v);
        __syncthreads();
        // element (row r, b, p) at (p*16 + (b ^ (p&1)))*TR + (r ^ b) [mod TR]: conflict-free b64 writes (the 16 lanes
        // of a row r hit 16 different bank pairs) and reads (for TR = 16 a 32-lane read group spans two p, whose
        // 128-B rows are put on opposite halves of the 64 banks by the b ^ (p&1) term)
#pragma unroll
        for (int h = 0; h < 2; h++) {                       // two batches of 8: the full 16 cost 3-6 spilled VGPRs
            cf w[8];
#pragma unroll
            for (int p = 0; p < 8; p++) w[p] = ld2(&w256[b * (8 * h + p)]);
#pragma unroll
            for (int p = 0; p < 8; p++) {
                const int pv = 8 * h + p;
                st2(&tile[(pv * 16 + (b ^ (pv & 1))) * TR + ((r ^ b) & (TR - 1))], cmul(v[rev16(pv)], w[p]));
            }
        }
        __syncthreads();
#pragma unroll
        for (int bb = 0; bb < 16; bb++)
            v[bb] = ld2(&tile[(p2 * 16 + (bb ^ (p2 & 1))) * TR + ((r2 ^ bb) & (TR - 1))]);
        dft16<false>(v);
        const long long rho = (long long)tl * TR + r2;
        const bool live = rho < nrows;
        const unsigned rbytes = (unsigned)rho * 8u;
        unsigned so[16];                                       // always sixteen stores (unused slot / row beyond the call: dropped by the range check)
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const unsigned off = soff[p2 + 16 * q];             // start of the stream of the channel in this slot
            so[q] = (off == 0xFFFFFFFFu || !live) ? 0xFFFFFFF0u : off + rbytes;
        }
        if (hints & 1) {
#pragma unroll
            for (int q = 0; q < 16; q++) bst2_nt(rout, so[q], 0, v[rev16(q)]);
        } else {
#pragma unroll
            for (int q = 0; q < 16; q++) bst2(rout, so[q], 0, v[rev16(q)]);
        }
        if (nxt >= ntiles) break;
        vm_wait<16>(L);
        tl = nxt;
    }
}

// Stage 2 for N1 = 1024 slots (N = 262144, BASELINE configs[3]): FFT-1024 over n1 of every row (m, t') of G, as
// 16 x 16 x 4: n1 = 64a + b' (layer 1, DFT-16 over a -> p), b' = 4c + d (layer 2, DFT-16 over c -> s; layer 3, DFT-4
// over d -> u), slot k1 = p + 16 s + 256 u.  A tile is 16 consecutive rows t' of one block (128 KiB in LDS), one
// 1024-thread workgroup per CU, 16 points per thread in every layer; persistent with next-tile prefetch like k_p2.
// LDS exchange layouts (both conflict-free for b64 accesses, rows on the fast lanes at the store end):
//   exchange 1: (r, p, b')    at (p*64 + b')*16 + ((r ^ b') & 15)
//   exchange 2: (r, p, d, s)  at ((p*4 + d)*16 + (s ^ (d & 1)))*16 + r
// Round 6: 16-BYTE ACCESSES at both ends as a build variant (-DFDC_P2K_WIDE=1; the shipped form is the 8-byte one of rounds 2-5, see below).  The memory pattern alone runs at 6.3-6.6 TB/s
// with 16 bytes per lane where the 8-byte form of this kernel moved its 537 MB at 5.76 (profiles/r05/NOTES.md section 1), and what kept the wide shape
// out of k_p1 — lane pairs must trade halves so that a lane ends up with ONE column (loads) / one slot's TWO rows (stores) — costs one
// instruction per dword here: v_permlane16_swap_b32 (gfx950) trades the odd rows of one register with the even rows of another, so the lane bit
// that pairs two lanes is made bit 4 of the lane at both ends.
//   loads : lane (c8, rlo, ah, rmid) of wave (rhi, j) asks for columns 2 c8, 2 c8 + 1 of row r at a = 8 ah + i, i < 8 (sixteen bytes); after eight
//           swaps per component it holds column col = 2 c8 + ah at all sixteen a: the layer-1 roles with another lane numbering
//   stores: layer-2/3 roles numbered so that bit 4 of the lane is bit 0 of the row: lanes l and l ^ 16 hold rows r2, r2 ^ 1 of the same sixteen slots;
//           after the swaps the even row's lane has slots 0-7 of BOTH rows and the odd row's lane slots 8-15: eight 16-byte stores each.
// The exchanges stay conflict-free under the new lane numberings (worked out against the bank rules as in the comment above: exchange 2's row
// position is r2 ^ (d & 1) now, so that the two d of a 16-lane store group do not fall on the same 32 banks).
// MEASURED (profiles/r06/ab_p2k_wide.txt, same box, three alternations, N = 262144): 0.0990 / 0.1001 / 0.1010 ms per 256 blocks against 0.0956 / 0.0980 /
// 0.1062 for the 8-byte form: nothing.  Halving the instructions does not change what the memory system sees: G's column tiles are 16 columns wide
// (k_p1's layout) and a tile has 16 rows per slot (LDS: 128 KiB), so both ends stay at 128-byte pieces; the pattern bench's gain came with 256-byte
// pieces.  The 32 swaps per lane and tile cost nothing either.  Shipped: the 8-byte form (FDC_P2K_WIDE=0); the wide form passes the same parity tests.
#ifndef FDC_P2K_WIDE
#define FDC_P2K_WIDE 0
#endif
typedef unsigned u32x2p __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(1024, 1) void k_p2k(const float2 *__restrict__ g, float2 *__restrict__ out,
                                                  const float2 *__restrict__ tw1024,
                                                  const long long *__restrict__ slot_off, long long nrows,
                                                  long long out_base, long long nb_call, unsigned out_bytes,
                                                  int ntiles, int lout, int hints)
{
    constexpr int TR = 16;
    float2 *tile = reinterpret_cast<float2 *>(fdc_smem_fast);                           // 16 rows x 1024 points
    // twiddle tables as rows a thread reads 16 bytes at a time: tw1r[b'][p] = W_1024^(b' p) (64 rows), tw2r[d][s] = W_64^(d s)
    // (4 rows); rows padded to 18 entries so that the rows of a wave start in different bank groups
    float2 *tw1r = reinterpret_cast<float2 *>(fdc_smem_fast + 1024 * TR * 8);
    float2 *tw2r = reinterpret_cast<float2 *>(fdc_smem_fast + 1024 * TR * 8 + 64 * 18 * 8);
    unsigned *soff = reinterpret_cast<unsigned *>(fdc_smem_fast + 1024 * TR * 8 + 68 * 18 * 8);
    const int tid = threadIdx.x;
    int tl = blockIdx.x;
    if (tl >= ntiles) return;
    {
        tw1r[(tid >> 4) * 18 + (tid & 15)] = tw1024[((tid >> 4) * (tid & 15)) & 1023];
        if (tid < 64) tw2r[(tid >> 4) * 18 + (tid & 15)] = tw1024[(16 * (tid >> 4) * (tid & 15)) & 1023];
        const long long o = slot_off[tid];
        soff[tid] = o >= 0 ? (unsigned)((o * nb_call + out_base) * 8) : 0xFFFFFFFFu;
    }
#if FDC_P2K_WIDE
    const int ah = (tid >> 4) & 1;                                   // which half of a the lane loads; afterwards bit 0 of its column
    const int col = 2 * (tid & 7) + ah, r = ((tid >> 3) & 1) + 2 * ((tid >> 5) & 1) + 4 * ((tid >> 6) & 3), j = tid >> 8;
    const int bp = 16 * j + col;
    const int r2 = 2 * ((tid >> 1) & 7) + ((tid >> 4) & 1), d = (tid & 1) + 2 * ((tid >> 5) & 1), p2 = tid >> 6;
#else
    const int col = tid & 15, r = (tid >> 4) & 15, j = tid >> 8;     // layer 1: row r, b' = 16j + col
    const int bp = 16 * j + col;
    const int r2 = tid & 15, d = (tid >> 4) & 3, p2 = tid >> 6;      // layer 2: row r2, (p2, d); layer 3: (p2, e = d)
#endif
    const __amdgpu_buffer_rsrc_t rout = make_rsrc(out, out_bytes);
    const int tpb = lout / TR;                                       // tiles per block
    // G[m][ct][t'][16]: point n1 = 64a + b' sits in column tile ct = 4a + j at column col
    const unsigned astep = 4u * (unsigned)lout * 16u * 8u;
#if FDC_P2K_WIDE
    const unsigned voff = (unsigned)((j * lout + r) * 16 + (col & ~1)) * 8u + 8u * (unsigned)ah * astep;
#else
    const unsigned voff = (unsigned)((j * lout + r) * 16 + col) * 8u;
#endif
    // LDS addresses with the swizzles folded into a few base pointers (everything else is an immediate offset)
    float2 *const wr1 = tile + bp * 16 + ((r ^ bp) & 15);                          // + p*1024
    const float2 *rd1[4];                                                          // + c*64, base by c & 3
#pragma unroll
    for (int k = 0; k < 4; k++) rd1[k] = tile + (p2 * 64 + d) * 16 + ((r2 ^ d ^ (4 * k)) & 15);
    float2 *wr2[2];                                                                // s even / odd: + (s>>1)*32
    constexpr int kPosSw = FDC_P2K_WIDE ? 1 : 0;                                   // wide form: row position r2 ^ (d & 1)
    wr2[0] = tile + ((p2 * 4 + d) * 16 + (d & 1)) * 16 + (r2 ^ (kPosSw & d));
    wr2[1] = tile + ((p2 * 4 + d) * 16 + 1 - (d & 1)) * 16 + (r2 ^ (kPosSw & d));
    const float2 *rd2[2];                                                          // d' even / odd: + (d'*16 + 4f)*16
    rd2[0] = tile + (p2 * 64 + d) * 16 + r2;
    rd2[1] = tile + (p2 * 64 + (d ^ 1)) * 16 + (r2 ^ kPosSw);
    // the tile loads are inline assembly and waited for behind the tile's stores (vm_wait, fdc_devutil.hpp): the compiler's own wait at
    // the loop latch was vmcnt(0) — every wave of the one workgroup a compute unit has sat out the completion of its sixteen stores
#if FDC_P2K_WIDE
    u32x4 L4[8];
    auto issue = [&](int t) __attribute__((always_inline)) {
        const size_t m = t / tpb;
        const int t0 = (t - (int)m * tpb) * TR;
        const __amdgpu_buffer_rsrc_t rg = make_rsrc(g + m * (size_t)lout * 1024 + (size_t)t0 * 16, (unsigned)lout * 1024u * 8u);
#pragma unroll
        for (int i = 0; i < 8; i++) L4[i] = bld4(rg, voff, (unsigned)i * astep);
    };
    issue(tl);
#else
    cf L[16];
    auto issue = [&](int t) __attribute__((always_inline)) {
        const size_t m = t / tpb;
        const int t0 = (t - (int)m * tpb) * TR;
        const srd_t rg = make_srd(g + m * (size_t)lout * 1024 + (size_t)t0 * 16, (unsigned)lout * 1024u * 8u);
#pragma unroll
        for (int a = 0; a < 16; a++) L[a] = ald2<false>(rg, voff, (unsigned)a * astep);
    };
    issue(tl);
    vm_wait<0>(L);
#endif
    for (;;) {
        cf v[16];
#if FDC_P2K_WIDE
        // (columns 2 c8, 2 c8 + 1 at a = 8 ah + i) -> (column 2 c8 + ah at a = i and at a = 8 + i): the even row of a pair keeps its first column and
        // gets it at the other eight a from the odd row, which keeps its second column
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const u32x2p sx = __builtin_amdgcn_permlane16_swap(L4[i].x, L4[i].z, false, false);
            const u32x2p sy = __builtin_amdgcn_permlane16_swap(L4[i].y, L4[i].w, false, false);
            v[i] = mk(__uint_as_float(sx.x), __uint_as_float(sy.x));
            v[8 + i] = mk(__uint_as_float(sx.y), __uint_as_float(sy.y));
        }
#else
#pragma unroll
        for (int a = 0; a < 16; a++) v[a] = L[a];
#endif
        const int nxt = tl + gridDim.x;
        if (nxt < ntiles) issue(nxt);
        dft16<false>(v);                                            // Y_b'[p] in v[rev16(p)]
        __syncthreads();                                            // tables ready / previous tile's reads done
        // twiddles in batches of 8: a table read between two tile writes would wait for the write in front of it
#pragma unroll
        for (int h = 0; h < 2; h++) {
            cf w[8];
#pragma unroll
            for (int i = 0; i < 4; i++) {                                        // W_1024^(b' p)
                const float4 t = ld4(&tw1r[bp * 18 + 8 * h + 2 * i]);
                w[2 * i] = mk(t.x, t.y); w[2 * i + 1] = mk(t.z, t.w);
            }
#pragma unroll
            for (int p = 0; p < 8; p++) st2(wr1 + (8 * h + p) * 1024, cmul(v[rev16(8 * h + p)], w[p]));
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 16; c++) v[c] = ld2(rd1[c & 3] + c * 64);
        dft16<false>(v);                                            // Z[s] in v[rev16(s)]
        __syncthreads();                                            // every thread has read exchange 1
#pragma unroll
        for (int h = 0; h < 2; h++) {
            cf w[8];
#pragma unroll
            for (int i = 0; i < 4; i++) {                                        // W_64^(d s)
                const float4 t = ld4(&tw2r[d * 18 + 8 * h + 2 * i]);
                w[2 * i] = mk(t.x, t.y); w[2 * i + 1] = mk(t.z, t.w);
            }
#pragma unroll
            for (int sx = 0; sx < 8; sx++) {
                const int sv = 8 * h + sx;
                st2(wr2[sv & 1] + (sv >> 1) * 32, cmul(v[rev16(sv)], w[sx]));
            }
        }
        __syncthreads();
        // layer 3: this thread takes s = 4f + e (e = d), all four d of each
#pragma unroll
        for (int f = 0; f < 4; f++)
#pragma unroll
            for (int dd = 0; dd < 4; dd++) v[4 * f + dd] = ld2(rd2[dd & 1] + (dd * 16 + 4 * f) * 16);
#pragma unroll
        for (int f = 0; f < 4; f++) dft4<false>(v[4 * f], v[4 * f + 1], v[4 * f + 2], v[4 * f + 3]);
        // Always sixteen stores per lane: a slot the plan does not use, or a row beyond the call, gets an offset beyond the descriptor's
        // extent and is dropped by its range check (no branch per store; the count is what the wait below is stated in)
#if FDC_P2K_WIDE
        // rows r2 (even) and r2 + 1 of a slot are 16 contiguous bytes of its stream: the even row's lane takes slots v[0..7] of both rows, the odd
        // row's lane slots v[8..15]; always eight stores per lane (unused slots / rows beyond the call: beyond the descriptor's extent)
        const int odd = r2 & 1;
        const long long rho = (long long)tl * TR + (r2 & ~1);
        const bool live = rho < nrows;
        const unsigned rbytes = (unsigned)rho * 8u;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int f = i >> 2, u = i & 3;
            const unsigned off = soff[p2 + 16 * (4 * (f + 2 * odd) + d) + 256 * u];
            const unsigned o = (off == 0xFFFFFFFFu || !live) ? 0xFFFFFFF0u : off + rbytes;
            const u32x2p sx = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i].x), __float_as_uint(v[i + 8].x), false, false);
            const u32x2p sy = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i].y), __float_as_uint(v[i + 8].y), false, false);
            const cf lo = mk(__uint_as_float(sx.x), __uint_as_float(sy.x)), hi = mk(__uint_as_float(sx.y), __uint_as_float(sy.y));
            if (hints & 1) bst4<true>(rout, o, lo, hi); else bst4<false>(rout, o, lo, hi);
        }
        if (nxt >= ntiles) break;
        tl = nxt;
#else
        const long long rho = (long long)tl * TR + r2;
        const bool live = rho < nrows;
        const unsigned rbytes = (unsigned)rho * 8u;
        unsigned so[16];
#pragma unroll
        for (int f = 0; f < 4; f++)
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const unsigned off = soff[p2 + 16 * (4 * f + d) + 256 * u];
                so[4 * f + u] = (off == 0xFFFFFFFFu || !live) ? 0xFFFFFFF0u : off + rbytes;
            }
        if (hints & 1) {
#pragma unroll
            for (int i = 0; i < 16; i++) bst2_nt(rout, so[i], 0, v[i]);
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++) bst2(rout, so[i], 0, v[i]);
        }
        if (nxt >= ntiles) break;
        vm_wait<16>(L);                                              // the next tile has landed; this tile's stores are not waited for
        tl = nxt;
#endif
    }
}

static int nt_hints();
// ---- launchers ---------------------------------------------------------------------------------------------
hipError_t init_fast_kernels()
{
    hipError_t e = init_block_kernels();
    if (e != hipSuccess) return e;
    e = init_wide_kernels();
    if (e != hipSuccess) return e;
    e = init_fused4096_kernels();
    if (e != hipSuccess) return e;
    const int a = kTileBytes + 8192, c = kCTileBytes + 2304;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_a256), hipFuncAttributeMaxDynamicSharedMemorySize, a);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_b256), hipFuncAttributeMaxDynamicSharedMemorySize, a);
    if (e != hipSuccess) return e;
#define FDC_SETP1(T, A, K) \
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_p1<T, A, K>), hipFuncAttributeMaxDynamicSharedMemorySize, a); \
    if (e != hipSuccess) return e;
    FDC_SETP1(32, 0, 0) FDC_SETP1(16, 0, 0) FDC_SETP1(16, 1, 0) FDC_SETP1(16, 2, 0) FDC_SETP1(32, 1, 0) FDC_SETP1(32, 2, 0)
    FDC_SETP1(16, 3, 8) FDC_SETP1(16, 4, 0)
    FDC_SETP1(16, 0, 8) FDC_SETP1(16, 0, 4) FDC_SETP1(16, 0, 2) FDC_SETP1(16, 0, 1)
    FDC_SETP1(32, 0, 8) FDC_SETP1(32, 0, 4) FDC_SETP1(32, 0, 2) FDC_SETP1(32, 0, 1)
#undef FDC_SETP1
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_p2<32, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, a);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_p2<32, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, a);
    if (e != hipSuccess) return e;

    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_p2<16, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, a);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_p2k), hipFuncAttributeMaxDynamicSharedMemorySize, kP2kLds);
    if (e != hipSuccess) return e;
#define FDC_SETC(k) \
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, c); \
    if (e != hipSuccess) return e;
    FDC_SETC(k_c256) FDC_SETC(k_x256)
#undef FDC_SETC
    return hipSuccess;
}

hipError_t launch_fft65536(const float2 *in, size_t in_stride, float2 *out, float2 *tmp, int nitems, int out_rot,
                           float scale, const float2 *tw256, const float2 *twf, hipStream_t s, hipEvent_t *ev)
{
    hipError_t e;
    if (ev && (e = hipEventRecord(ev[0], s)) != hipSuccess) return e;
    for (int m0 = 0; m0 < nitems; m0 += 32768) {
        const int nb = nitems - m0 < 32768 ? nitems - m0 : 32768;
        hipLaunchKernelGGL(k_a256, dim3(8, nb), dim3(256), kTileBytes + 2048, s, in + (size_t)m0 * in_stride, in_stride,
                           tmp + (size_t)m0 * 65536, tw256, twf);
        if (m0 == 0 && ev && (e = hipEventRecord(ev[1], s)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_b256, dim3(8, nb), dim3(256), kTileBytes + 2048, s, tmp + (size_t)m0 * 65536,
                           out + (size_t)m0 * 65536, tw256, out_rot, scale);
    }
    if (ev && (e = hipEventRecord(ev[2], s)) != hipSuccess) return e;
    return hipGetLastError();
}

hipError_t launch_channels256(const float2 *spec, float2 *out, const ChanDev *chans, const int32_t *group,
                              int ngroup, bool aligned, bool out_aligned, int N, int R, int nb_chunk, int mbase, int nb_call,
                              int64_t first_block, const float2 *wins, const float2 *tw256, hipStream_t s)
{
    (void)aligned; (void)out_aligned;                       // 8-byte accesses: no alignment classes any more
    const long long ntrans = (long long)nb_chunk * ngroup;
    if (ntrans <= 0) return hipSuccess;
    dim3 grid((unsigned)((ntrans + 15) / 16));
    hipLaunchKernelGGL(k_c256, grid, dim3(256), kCTileBytes + 2304, s, spec, out, chans, group, ngroup, N, R, nb_chunk, mbase,
                       nb_call, (long long)first_block, wins, tw256);
    return hipGetLastError();
}

const char *debug_env(const char *name);     // fdc_api.hip: nullptr unless FDC_DEBUG_ENV=1
hipError_t launch_extract256(const float2 *spec, int N, const ExtractTask *tasks, int ntasks, int skip, const float2 *wins, float2 *out,
                             const float2 *tw256, hipStream_t s)
{
    if (ntasks <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_x256, dim3((unsigned)((ntasks + 15) / 16)), dim3(256), kCTileBytes + 2304, s, spec, out, tasks, ntasks, N, skip, wins, tw256);
    return hipGetLastError();
}

// FDC_NT bits (A/B testing): 1 = stage-2 output stores nt, 2 = stage-1 input loads nt, 4 = stage-2 G loads nt, 8 = stage-1 G stores nt
static int nt_hints()
{
    static int h = -1;
    if (h < 0) { const char *t = debug_env("FDC_NT"); h = t ? atoi(t) & 15 : kDefaultNtHints; }
    return h;
}
static int poly_tile(int lout)
{
    static int tcfg = -1;                                   // FDC_POLY_TILE=16|32 (A/B testing); default 16
    if (tcfg < 0) { const char *t = debug_env("FDC_POLY_TILE"); tcfg = (t && atoi(t) == 32) ? 32 : 16; }
    return (lout % tcfg) ? 16 : tcfg;                       // stage-2 tiles are TC whole rows of one block
}
hipError_t launch_poly_stage1(const float2 *in, size_t in_stride, float2 *g, int N1, int R, int nb_chunk,
                              const float2 *tw256, const float2 *twq, const float2 *cbt, const float *shn,
                              int ncu, hipStream_t s)
{
    const int skip = 256 / R, lout = 256 - skip;
    const int TC = N1 == 256 ? poly_tile(lout) : 16;        // k_p2k reads 16-column tiles
    const int ct = N1 / TC;
    int log2ct = 0;
    while ((1 << log2ct) < ct) log2ct++;
    const int maxwg = TC == 32 ? 2 : 4;                     // resident workgroups per CU: LDS-limited
    int slots = maxwg * (ncu > 0 ? ncu : 256);
    slots -= slots % ct;
    if (slots < ct) slots = ct;
    // every workgroup keeps one column tile and takes runs of bpg consecutive blocks (FDC_POLY_BPG: run length, A/B testing)
    static int bpg_cfg = -1;
    if (bpg_cfg < 0) { const char *t = debug_env("FDC_POLY_BPG"); bpg_cfg = t ? atoi(t) : 0; }
    int groups = slots / ct;
    if (groups > nb_chunk) groups = nb_chunk;
    int bpg = (nb_chunk + groups - 1) / groups;
    if (bpg_cfg > 0 && bpg_cfg < bpg) bpg = bpg_cfg;
    const int runs = (nb_chunk + bpg - 1) / bpg;
    if (groups > runs) groups = runs;
    const unsigned g1 = (unsigned)(groups * ct);
    const size_t lds1 = 256 * TC * 8 + 2304 + TC * 144 + 1024;
    static int abl = -1, noreuse = 0;
    if (abl < 0) {
        const char *t = debug_env("FDC_ABLATE"); abl = t ? atoi(t) : 0;
        const char *g4 = debug_env("FDC_POLY_NOREUSE"); noreuse = g4 ? atoi(g4) : 0;
    }
    // the overlap of consecutive blocks travels in registers when the items are exactly N - N/R apart
    const bool reuse = !noreuse && (abl == 0 || abl == 3) && in_stride == (size_t)256 * N1 - (size_t)256 * N1 / R && R <= 16;
#define FDC_LP1(T, A, K) \
    hipLaunchKernelGGL((k_p1<T, A, K>), dim3(g1), dim3(T * 16), lds1, s, in, in_stride, g, tw256, twq, cbt, shn, N1, log2ct, \
                       nb_chunk, bpg, skip / 16, lout, nt_hints())
#define FDC_LP1K(T) \
    do { if (!reuse) FDC_LP1(T, 0, 0); else if (R == 2) FDC_LP1(T, 0, 8); else if (R == 4) FDC_LP1(T, 0, 4); \
         else if (R == 8) FDC_LP1(T, 0, 2); else FDC_LP1(T, 0, 1); } while (0)
    if (TC == 32) { if (abl == 1) FDC_LP1(32, 1, 0); else if (abl == 2) FDC_LP1(32, 2, 0); else FDC_LP1K(32); }
    else { if (abl == 1) FDC_LP1(16, 1, 0); else if (abl == 2) FDC_LP1(16, 2, 0); else if (abl == 3 && reuse && R == 2) FDC_LP1(16, 3, 8);
           else if (abl == 4) FDC_LP1(16, 4, 0); else FDC_LP1K(16); }
#undef FDC_LP1K
#undef FDC_LP1
    return hipGetLastError();
}

hipError_t launch_poly_stage2(const float2 *g, float2 *out, int N1, int R, int nb_chunk, int mbase, int nb_call,
                              const float2 *tw256, const float2 *tw1024, const long long *slot_off, unsigned out_bytes,
                              int ncu, hipStream_t s)
{
    const int skip = 256 / R, lout = 256 - skip;
    if (N1 == 1024) {                                       // one 1024-thread workgroup per CU, 16-row tiles
        const long long nrows = (long long)nb_chunk * lout;
        const long long nt = (nrows + 15) / 16;
        const int slots = ncu > 0 ? ncu : 256;
        const unsigned gk = (unsigned)(nt < slots ? nt : slots);
        hipLaunchKernelGGL(k_p2k, dim3(gk), dim3(1024), kP2kLds, s, g, out, tw1024, slot_off, nrows,
                           (long long)mbase * lout, (long long)nb_call, out_bytes, (int)nt, lout, nt_hints());
        return hipGetLastError();
    }
    const int TCG = poly_tile(lout);                        // column-tile width stage 1 wrote G with
    static int rows_cfg = -1;                               // FDC_POLY_ROWS=16|32: rows per stage-2 tile (A/B testing)
    if (rows_cfg < 0) { const char *t = debug_env("FDC_POLY_ROWS"); rows_cfg = t ? atoi(t) : 0; }
    int TR = rows_cfg == 16 || rows_cfg == 32 ? rows_cfg : TCG;
    if (lout % TR) TR = 16;
    if (TCG == 32 && TR == 16) TR = 32;                     // 32-column G tiles are read 32 rows at a time
    const int maxwg = TR == 32 ? 2 : 4;
    const int slots = maxwg * (ncu > 0 ? ncu : 256);
    const size_t lds2 = 256 * TR * 8 + 2048 + 2048;
    const long long nrows = (long long)nb_chunk * lout;
    const long long nt2 = (nrows + TR - 1) / TR;
    const unsigned g2 = (unsigned)(nt2 < slots ? nt2 : slots);
#define FDC_LP2(A, B) \
    hipLaunchKernelGGL((k_p2<A, B>), dim3(g2), dim3(A * 16), lds2, s, g, out, tw256, slot_off, nrows, (long long)mbase * lout, \
                       (long long)nb_call, out_bytes, (int)nt2, lout, nt_hints())
    if (TR == 32 && TCG == 32) FDC_LP2(32, 32);
    else if (TR == 32) FDC_LP2(32, 16);
    else FDC_LP2(16, 16);
#undef FDC_LP2
    return hipGetLastError();
}

}  // namespace fdc
