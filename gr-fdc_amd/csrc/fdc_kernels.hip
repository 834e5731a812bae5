// gfx950 (MI355X, CDNA4) kernels of the frequency-domain channelizer — generic path.
//
// Everything here is wave64 / 256-thread workgroups, float32 complex (float2), Stockham autosort
// radix-4 (+ one radix-2 pass when log2 L is odd) with the transform index on the SLOW axis of an LDS
// tile: element i of transform t lives at lds[i*ld + t], lanes run over t (and then over butterflies),
// so every LDS access of a wave is unit-stride.  One LDS buffer serves both sides of a pass: all reads
// of a pass go to registers, barrier, then all writes (each point is read once and written once).
//
// Reference semantics implemented (paths in the reference tree):
//   overlap-save gather     lib/overlap_save_impl.cc:70-78           (fused into the first FFT load)
//   fft_vcc(N, fwd, shift)  python/FrequencyDomainChannelizer.py:206 (shift fused into the store index)
//   multiply_const(1/N)     python/FrequencyDomainChannelizer.py:214 (fused into the store)
//   vector_cut_vxx          lib/vector_cut_vxx_impl.cc:67-68         (fused into the channel load)
//   phase-shifting window   lib/phase_shifting_windowing_vcc_impl.cc:80-83
//   fft_vcc(l, inv, shift)  python/FrequencyDomainChannelizer.py:228 (ifftshift fused into the LDS index)
//   cut + *l                python/FrequencyDomainChannelizer.py:229-231
#include "fdc_kernels.h"
#include <cstdlib>
#include "fdc_radix16.hpp"

namespace fdc {

hipError_t init_sink_kernels();   // fdc_sinks_dev.hip

extern __shared__ __attribute__((aligned(16))) unsigned char fdc_smem[];

// A tile holds 4096*NB complex points (NB = 1: 32 KiB, four workgroups per CU; NB = 2 only for L = 8192), every
// thread owns PT = 16*NB of them in each pass and while a tile is loaded or stored (all PT global loads are issued
// before the first is used).
#define FDC_GENERIC_BOUNDS(NB) __launch_bounds__(kThreads, (NB) == 1 ? 4 : 2)
// the kernels that hold a tile's samples AND its window (or twiddle) values in flight: at four workgroups per compute unit (128 VGPRs)
// they spill 25 registers into the loops; three (168 VGPRs) hold everything
#define FDC_TILE_BOUNDS(NB) __launch_bounds__(kThreads, (NB) == 1 ? 3 : 2)

__device__ __forceinline__ float2 cmulf(float2 a, float2 b)
{
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

template <bool INV>
__device__ __forceinline__ float2 ldtw(const float2 *__restrict__ tw, int idx)
{
    float2 w = tw[idx];
    if (INV) w.y = -w.y;
    return w;
}
template <bool INV>
__device__ __forceinline__ cf ldtwc(const float2 *__restrict__ tw, int idx)
{
    const float2 w = tw[idx];
    return mk(w.x, INV ? -w.y : w.y);
}

// TC = 2^log2TC transforms of length L = 2^log2L held in lds as [L][ld].  twstride = ntab / L.
// Stockham autosort: radix-16 passes (two radix-4 layers in registers per LDS round trip, fdc_radix16.hpp), then
// one radix-4 and/or one radix-2 pass for what is left of log2 L.
template <bool INV, int NB>
__device__ __forceinline__ void fft_cols(float2 *lds, int log2L, int log2TC, int ld, const float2 *__restrict__ tw, int twstride)
{
    const int tid = threadIdx.x;
    const int L = 1 << log2L;
    const int cmask = (1 << log2TC) - 1;
    int log2ns = 0;
    for (; log2ns + 4 <= log2L; log2ns += 4) {
        const int ns = 1 << log2ns;
        const int q = L >> 4;
        const int total = q << log2TC;
        const int tstep = (L >> (log2ns + 4)) * twstride;
        cf o[NB][16];
#pragma unroll
        for (int i = 0; i < NB; i++) {
            const int b = tid + i * kThreads;
            if (b < total) {
                const int c = b & cmask, j = b >> log2TC, k = j & (ns - 1);
#pragma unroll
                for (int r = 0; r < 16; r++) o[i][r] = from2(lds[(j + r * q) * ld + c]);
                if (ns > 1) {
                    // input r = 4a + b2 takes W^(k r) = W^(4 a k) * W^(b2 k): six table reads instead of fifteen
                    const int kt = k * tstep;
                    cf A[4], B[4];
#pragma unroll
                    for (int x = 1; x < 4; x++) { A[x] = ldtwc<INV>(tw, 4 * x * kt); B[x] = ldtwc<INV>(tw, x * kt); }
#pragma unroll
                    for (int r = 1; r < 16; r++) {
                        if (r & 3) o[i][r] = cmul(o[i][r], B[r & 3]);
                        if (r >> 2) o[i][r] = cmul(o[i][r], A[r >> 2]);
                    }
                }
                dft16<INV>(o[i]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NB; i++) {
            const int b = tid + i * kThreads;
            if (b < total) {
                const int c = b & cmask, j = b >> log2TC, k = j & (ns - 1);
                const int j0 = ((j >> log2ns) << (log2ns + 4)) + k;
#pragma unroll
                for (int r = 0; r < 16; r++) lds[(j0 + r * ns) * ld + c] = to2(o[i][rev16(r)]);
            }
        }
        __syncthreads();
    }
    if (log2ns + 2 <= log2L) {
        const int ns = 1 << log2ns;
        const int q = L >> 2;
        const int total = q << log2TC;
        const int tstep = (L >> (log2ns + 2)) * twstride;
        cf o[4 * NB][4];
#pragma unroll
        for (int i = 0; i < 4 * NB; i++) {
            const int b = tid + i * kThreads;
            if (b < total) {
                const int c = b & cmask, j = b >> log2TC, k = j & (ns - 1);
#pragma unroll
                for (int r = 0; r < 4; r++) o[i][r] = from2(lds[(j + r * q) * ld + c]);
                if (ns > 1) {
#pragma unroll
                    for (int r = 1; r < 4; r++) o[i][r] = cmul(o[i][r], ldtwc<INV>(tw, r * k * tstep));
                }
                dft4<INV>(o[i][0], o[i][1], o[i][2], o[i][3]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4 * NB; i++) {
            const int b = tid + i * kThreads;
            if (b < total) {
                const int c = b & cmask, j = b >> log2TC, k = j & (ns - 1);
                const int j0 = ((j >> log2ns) << (log2ns + 2)) + k;
#pragma unroll
                for (int r = 0; r < 4; r++) lds[(j0 + r * ns) * ld + c] = to2(o[i][r]);
            }
        }
        __syncthreads();
        log2ns += 2;
    }
    if (log2ns < log2L) {   // final radix-2 pass
        const int ns = 1 << log2ns;
        const int q = L >> 1;
        const int total = q << log2TC;
        const int tstep = twstride;   // L / (2*ns) == 1 here
        cf o[8 * NB][2];
#pragma unroll
        for (int i = 0; i < 8 * NB; i++) {
            const int b = tid + i * kThreads;
            if (b < total) {
                const int c = b & cmask, j = b >> log2TC, k = j & (ns - 1);
                const cf v0 = from2(lds[j * ld + c]);
                cf v1 = from2(lds[(j + q) * ld + c]);
                if (ns > 1) v1 = cmul(v1, ldtwc<INV>(tw, k * tstep));
                o[i][0] = v0 + v1;
                o[i][1] = v0 - v1;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8 * NB; i++) {
            const int b = tid + i * kThreads;
            if (b < total) {
                const int c = b & cmask, j = b >> log2TC, k = j & (ns - 1);
                const int j0 = ((j >> log2ns) << (log2ns + 1)) + k;
                lds[j0 * ld + c] = to2(o[i][0]);
                lds[(j0 + ns) * ld + c] = to2(o[i][1]);
            }
        }
        __syncthreads();
    }
}

// ---- whole transform in one workgroup (N <= kMaxLdsFft): TC items per workgroup ------------------
template <bool INV, int NB>
__global__ FDC_GENERIC_BOUNDS(NB) void k_fft_small(const float2 *__restrict__ in, size_t in_stride,
                                                   float2 *__restrict__ out, int log2N, int log2TC, int ld,
                                                   int nitems, int in_rot, int out_rot, float scale,
                                                   const float2 *__restrict__ tw, int twstride)
{
    constexpr int PT = 16 * NB;
    float2 *lds = reinterpret_cast<float2 *>(fdc_smem);
    const int N = 1 << log2N, TC = 1 << log2TC, total = N << log2TC;
    const int m0 = blockIdx.x * TC;
    float2 v[PT];
#pragma unroll
    for (int u = 0; u < PT; u++) {
        const int e = threadIdx.x + u * kThreads;
        const int t = e >> log2N, i = e & (N - 1), m = m0 + t;
        v[u] = make_float2(0.f, 0.f);
        if (e < total && m < nitems) v[u] = in[(size_t)m * in_stride + ((i + in_rot) & (N - 1))];
    }
#pragma unroll
    for (int u = 0; u < PT; u++) {
        const int e = threadIdx.x + u * kThreads;
        if (e < total) lds[(e & (N - 1)) * ld + (e >> log2N)] = v[u];
    }
    __syncthreads();
    fft_cols<INV, NB>(lds, log2N, log2TC, ld, tw, twstride);
#pragma unroll
    for (int u = 0; u < PT; u++) {
        const int e = threadIdx.x + u * kThreads;
        const int t = e >> log2N, kp = e & (N - 1), m = m0 + t;
        if (e < total && m < nitems) {
            const float2 y = lds[((kp - out_rot) & (N - 1)) * ld + t];
            out[(size_t)m * N + kp] = make_float2(y.x * scale, y.y * scale);
        }
    }
}

// ---- two-pass transform N = N1*N2, n = n1 + N1*n2, k = N2*k1 + k2 -----------------------------------
// Pass A: TC columns n1, length-N2 transforms over n2 (row stride N1 in memory), times W_N^(n1*k2),
//         stored transposed-by-construction as T[k2][n1] (n1 contiguous).
// TASK: item m is extraction m of a width class (fdc_sinks): its input is the slice of the spectrum the task names, times the task's
// window (read where the transform reads it: no gathered copy), and pass B writes the kept part [skip, N) to the task's place.
template <bool INV, int NB, bool TASK = false>
__global__ FDC_TILE_BOUNDS(NB) void k_fft_pass_a(const float2 *__restrict__ in, size_t in_stride,
                                                    float2 *__restrict__ tmp, int log2N, int log2N1,
                                                    int log2TC, int ld, int in_rot,
                                                    const float2 *__restrict__ tw, int ntab,
                                                    const float2 *__restrict__ twf, const ExtractTask *__restrict__ tasks = nullptr,
                                                    const float2 *__restrict__ wins = nullptr)
{
    constexpr int PT = 16 * NB;
    float2 *lds = reinterpret_cast<float2 *>(fdc_smem);
    const int N = 1 << log2N, log2N2 = log2N - log2N1, N2 = 1 << log2N2;
    const int TC = 1 << log2TC, total = N2 << log2TC;
    const int c0 = blockIdx.x * TC;
    const size_t m = blockIdx.y;
    const float2 *src = in + m * in_stride, *win = nullptr;
    if (TASK) { const ExtractTask tk = tasks[m]; src = in + (size_t)tk.slot * in_stride + tk.start; win = wins + tk.win_off; }
    float2 v[PT];
#pragma unroll
    for (int u = 0; u < PT; u++) {
        const int e = threadIdx.x + u * kThreads;
        const int r = e >> log2TC, c = e & (TC - 1);
        v[u] = make_float2(0.f, 0.f);
        if (e < total) {
            const int idx = (c0 + c + (r << log2N1) + in_rot) & (N - 1);
            v[u] = src[idx];
            if (TASK) v[u] = cmulf(v[u], win[idx]);
        }
    }
#pragma unroll
    for (int u = 0; u < PT; u++) {
        const int e = threadIdx.x + u * kThreads;
        if (e < total) lds[(e >> log2TC) * ld + (e & (TC - 1))] = v[u];
    }
    __syncthreads();
    fft_cols<INV, NB>(lds, log2N2, log2TC, ld, tw, ntab >> log2N2);
    float2 *dst = tmp + m * (size_t)N;
    const int twn = ntab >> log2N;
    // W_N^(n1*k2): from the [k2][n1] table when the caller has one (coalesced like the store), else gathered
#pragma unroll
    for (int u = 0; u < PT; u++) {
        const int e = threadIdx.x + u * kThreads;
        const int k2 = e >> log2TC, c = e & (TC - 1), n1 = c0 + c;
        v[u] = make_float2(1.f, 0.f);
        if (e < total) {
            if (twf) { v[u] = twf[((size_t)k2 << log2N1) + n1]; if (INV) v[u].y = -v[u].y; }
            else v[u] = ldtw<INV>(tw, n1 * k2 * twn);
        }
    }
#pragma unroll
    for (int u = 0; u < PT; u++) {
        const int e = threadIdx.x + u * kThreads;
        const int k2 = e >> log2TC, c = e & (TC - 1), n1 = c0 + c;
        if (e < total) dst[((size_t)k2 << log2N1) + n1] = cmulf(lds[k2 * ld + c], v[u]);
    }
}

// Pass B: TR rows k2 of T, length-N1 transforms over n1 (contiguous in memory), result bin
//         k = N2*k1 + k2 stored at (k + out_rot) mod N, scaled.
template <bool INV, int NB, bool TASK = false>
__global__ FDC_GENERIC_BOUNDS(NB) void k_fft_pass_b(const float2 *__restrict__ tmp, float2 *__restrict__ out,
                                                    int log2N, int log2N1, int log2TR, int ld, int out_rot,
                                                    float scale, const float2 *__restrict__ tw, int ntab,
                                                    const ExtractTask *__restrict__ tasks = nullptr, int skip = 0)
{
    constexpr int PT = 16 * NB;
    float2 *lds = reinterpret_cast<float2 *>(fdc_smem);
    const int N = 1 << log2N, N1 = 1 << log2N1, log2N2 = log2N - log2N1;
    const int TR = 1 << log2TR, total = N1 << log2TR;
    const int r0 = blockIdx.x * TR;
    const size_t m = blockIdx.y;
    const float2 *src = tmp + m * (size_t)N;
    float2 v[PT];
#pragma unroll
    for (int u = 0; u < PT; u++) {
        const int e = threadIdx.x + u * kThreads;
        v[u] = make_float2(0.f, 0.f);
        if (e < total) v[u] = src[((size_t)(r0 + (e >> log2N1)) << log2N1) + (e & (N1 - 1))];
    }
#pragma unroll
    for (int u = 0; u < PT; u++) {
        const int e = threadIdx.x + u * kThreads;
        if (e < total) lds[(e & (N1 - 1)) * ld + (e >> log2N1)] = v[u];
    }
    __syncthreads();
    fft_cols<INV, NB>(lds, log2N1, log2TR, ld, tw, ntab >> log2N1);
    float2 *dst = TASK ? out + tasks[m].out_off - skip : out + m * (size_t)N;
#pragma unroll
    for (int u = 0; u < PT; u++) {
        const int e = threadIdx.x + u * kThreads;
        const int k1 = e >> log2TR, r = e & (TR - 1);
        if (e < total) {
            const int k = (k1 << log2N2) + r0 + r;
            const float2 y = lds[k1 * ld + r];
            const int kk = (k + out_rot) & (N - 1);
            if (!TASK || kk >= skip) dst[kk] = make_float2(y.x * scale, y.y * scale);
        }
    }
}

// ---- uniform path, stage 2 for any number of slots N1 = N/256 (fdc_fast256.hip has the 256- and 1024-slot forms) ----
// Rows rho = m*lout + t' of G (stage 1 writes them tile-major, G[m][ct][t'][16]), FFT over n1, bin k1 = channel slot.
// A tile is TR consecutive rows of one block; lanes run over the rows at the store end, so every channel stream
// receives TR*8-byte runs.
__global__ FDC_GENERIC_BOUNDS(1) void k_p2g(const float2 *__restrict__ g, float2 *__restrict__ out, int log2N1, int log2TR,
                                            int ld, int lout, const long long *__restrict__ slot_off, long long out_base,
                                            long long nb_call, const float2 *__restrict__ tw, int twstride, int rowmajor)
{
    constexpr int PT = 16;
    float2 *lds = reinterpret_cast<float2 *>(fdc_smem);
    const int N1 = 1 << log2N1, TR = 1 << log2TR, total = N1 << log2TR;
    const int tpb = lout >> log2TR;                                   // tiles per block
    const int m = blockIdx.x / tpb, t0 = (blockIdx.x - m * tpb) << log2TR;
    const float2 *src = g + (size_t)m * (size_t)lout * N1 + (size_t)t0 * 16;
    float2 v[PT];
#pragma unroll
    for (int u = 0; u < PT; u++) {
        const int e = threadIdx.x + u * kThreads;
        const int col = e & 15, r = (e >> 4) & (TR - 1), ct = e >> (4 + log2TR);
        v[u] = make_float2(0.f, 0.f);
        // G of the l = 256 stage 1 is tile-major ([column tile][row][16]); the generic-width stage 1 (k_p1g) writes rows of N1 columns
        if (e < total) v[u] = rowmajor ? g[((size_t)m * lout + t0 + r) * N1 + ct * 16 + col] : src[((size_t)ct * lout + r) * 16 + col];
    }
#pragma unroll
    for (int u = 0; u < PT; u++) {
        const int e = threadIdx.x + u * kThreads;
        const int col = e & 15, r = (e >> 4) & (TR - 1), ct = e >> (4 + log2TR);
        if (e < total) lds[(ct * 16 + col) * ld + r] = v[u];
    }
    __syncthreads();
    fft_cols<false, 1>(lds, log2N1, log2TR, ld, tw, twstride);
    const long long rho0 = (long long)m * lout + t0 + out_base;
#pragma unroll
    for (int u = 0; u < PT; u++) {
        const int e = threadIdx.x + u * kThreads;
        const int r = e & (TR - 1), k1 = e >> log2TR;
        if (e < total) {
            const long long o = slot_off[k1];                         // start of the stream of the channel in this slot
            // written once, never read back here: nt, so that G rather than the output stays cached (fdc_fast256.hip, nt_hints)
            if (o >= 0) __builtin_nontemporal_store(from2(lds[k1 * ld + r]), reinterpret_cast<cf *>(out + o * nb_call + rho0 + r));
        }
    }
}

// ---- fused channel kernel ------------------------------------------------------------------------
// Per tile column (one transform = one (block, channel) pair): where its slice, window row and output run start.
// Worked out once per column by the first TC threads, so the per-point loops carry no divisions or record fetches.
struct ColInfo { long long src; long long dst; int win; int valid; };

template <int NB>
__global__ FDC_TILE_BOUNDS(NB) void k_channels(const float2 *__restrict__ spec, float2 *__restrict__ out,
                                                  const ChanDev *__restrict__ chans,
                                                  const int32_t *__restrict__ group, int ngroup, int log2l,
                                                  int log2TC, int ld, int N, int R, int nb_chunk, int mbase,
                                                  int nb_call, long long first_block,
                                                  const float2 *__restrict__ wins,
                                                  const float2 *__restrict__ tw, int twstride)
{
    constexpr int PT = 16 * NB;
    __shared__ ColInfo col[32];
    float2 *lds = reinterpret_cast<float2 *>(fdc_smem);
    const int l = 1 << log2l, TC = 1 << log2TC, total = l << log2TC;
    const int lout = l - l / R, skip = l - lout;
    const long long ntrans = (long long)nb_chunk * ngroup;
    const long long t0 = (long long)blockIdx.x * TC;
    if ((int)threadIdx.x < TC) {
        const long long t = t0 + threadIdx.x;
        ColInfo ci{0, 0, 0, 0};
        if (t < ntrans) {
            const int m = (int)(t / ngroup), gi = (int)(t - (long long)m * ngroup);
            const ChanDev ch = chans[group[gi]];
            // vector_cut_vxx: bins [f, f+l) of block m's spectrum
            ci.src = (long long)m * N + ch.f;
            // phase_shifting_windowing_vcc: counter_m = (m*shift) mod R in closed form
            const int cnt = (int)((((first_block + mbase + m) % R) * ch.shift) % R);
            ci.win = ch.win_off + cnt * l;
            ci.dst = (long long)nb_call * ch.out_off + (long long)(mbase + m) * lout - skip;   // + output index i >= skip
            ci.valid = 1;
        }
        col[threadIdx.x] = ci;
    }
    __syncthreads();
    float2 v[PT], w[PT];
#pragma unroll
    for (int u = 0; u < PT; u++) {
        const int e = threadIdx.x + u * kThreads;
        const int tl = (e >> log2l) & (TC - 1), i = e & (l - 1);
        v[u] = make_float2(0.f, 0.f); w[u] = v[u];
        const ColInfo ci = col[tl];
        if (e < total && ci.valid) {
            v[u] = spec[ci.src + i];
            w[u] = wins[ci.win + i];
        }
    }
#pragma unroll
    for (int u = 0; u < PT; u++) {
        const int e = threadIdx.x + u * kThreads;
        const int tl = e >> log2l, i = e & (l - 1);
        if (e < total) lds[((i + (l >> 1)) & (l - 1)) * ld + tl] = cmulf(v[u], w[u]);     // ifftshift of the IFFT input
    }
    __syncthreads();
    fft_cols<true, NB>(lds, log2l, log2TC, ld, tw, twstride);
    const float scale = (float)l;
#pragma unroll
    for (int u = 0; u < PT; u++) {
        const int e = threadIdx.x + u * kThreads;
        const int tl = (e >> log2l) & (TC - 1), i = e & (l - 1);
        const ColInfo ci = col[tl];
        if (e < total && ci.valid && i >= skip) {                // overlap discard: the first l/R samples go
            const float2 y = lds[i * ld + tl];
            out[ci.dst + i] = make_float2(y.x * scale, y.y * scale);
        }
    }
}

// ---- sinks: power cells + task-list extraction (PowerActivationChannel / activity_detection_channelizer_vcm) ----
// One wave per (block, cell): sum |X|^2 over [start, start+len), times scale.
//   lib/PowerActivationChannel_impl.cc:289-291 (scale 1), lib/activity_detection_channelizer_vcm_impl.cc:641-648 (1/dec)
// A cell of a detector or a channel is a few hundred bins: one or two 16-byte loads per lane.  A wave takes its cell in kCellBlocks
// consecutive blocks at once (their loads are in flight together); every block's sum is formed exactly as before.
constexpr int kCellBlocks = 4;
__global__ __launch_bounds__(kThreads) void k_cell_power(const float2 *__restrict__ spec, int N,
                                                         const PowerCell *__restrict__ cells, int ncells, int nblocks,
                                                         float *__restrict__ out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cell = blockIdx.x * (kThreads / 64) + wave;
    const int mb = blockIdx.y * kCellBlocks;
    if (cell >= ncells) return;
    const PowerCell pc = cells[cell];
    // two bins per 16-byte load from the first even bin on (the spectrum and N are 16-byte aligned), two loads in flight per lane and block
    const int head = (pc.start & 1) && pc.len > 0 ? 1 : 0, n2 = (pc.len - head) >> 1;
    const bool tail = (pc.len - head) & 1;
    float acc[kCellBlocks], acc2[kCellBlocks];
    const float4 *x4[kCellBlocks];
#pragma unroll
    for (int j = 0; j < kCellBlocks; j++) {
        const int m = mb + j < nblocks ? mb + j : nblocks - 1;     // the surplus blocks of the last group repeat the last one (not stored)
        const float2 *x = spec + (size_t)m * N + pc.start;
        acc[j] = 0.f; acc2[j] = 0.f;
        if (lane == 0) {
            if (head) { const float2 v = x[0]; acc[j] += v.x * v.x + v.y * v.y; }
            if (tail) { const float2 v = x[pc.len - 1]; acc[j] += v.x * v.x + v.y * v.y; }
        }
        x4[j] = reinterpret_cast<const float4 *>(x + head);
    }
    int i = lane;
    for (; i + 64 < n2; i += 128) {
        float4 a[kCellBlocks], b[kCellBlocks];
#pragma unroll
        for (int j = 0; j < kCellBlocks; j++) { a[j] = x4[j][i]; b[j] = x4[j][i + 64]; }
#pragma unroll
        for (int j = 0; j < kCellBlocks; j++) {
            acc[j] += a[j].x * a[j].x + a[j].y * a[j].y + a[j].z * a[j].z + a[j].w * a[j].w;
            acc2[j] += b[j].x * b[j].x + b[j].y * b[j].y + b[j].z * b[j].z + b[j].w * b[j].w;
        }
    }
    if (i < n2) {
        float4 a[kCellBlocks];
#pragma unroll
        for (int j = 0; j < kCellBlocks; j++) a[j] = x4[j][i];
#pragma unroll
        for (int j = 0; j < kCellBlocks; j++) acc[j] += a[j].x * a[j].x + a[j].y * a[j].y + a[j].z * a[j].z + a[j].w * a[j].w;
    }
#pragma unroll
    for (int j = 0; j < kCellBlocks; j++) {
        float t = acc[j] + acc2[j];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
        if (lane == 0 && mb + j < nblocks) out[(size_t)(mb + j) * ncells + cell] = t * pc.scale;
    }
}

// Power cells from the sums of the spectrum's 16-bin groups (round 6): gpow[m][N / 16] is written by the forward transform while the spectrum is in its
// registers (fdc_block256.hip), so a cell is the groups that lie inside it plus the bins of the two groups it cuts — one or two cache lines of the
// spectrum per cell and block instead of all of it (a detector's cells tile 80 % of the band: k_cell_power read 444-472 MB per 1024 blocks back).
// Summation order differs from k_cell_power's in the last bits (the blocks' thresholds sit 6-10 dB above what they compare).
// Sixteen lanes per (cell, block) and kGrpBlocks blocks per lane group, all loads of a lane issued before the first sum: the first form (a wave per
// cell, its four blocks one after the other, a 64-lane reduction each) was a chain of dependent loads with a quarter of the lanes busy — 62-71 us
// per 1024 blocks, as long as the pass over the spectrum it replaced (profiles/r06/rocprof_kernel_stats_cfg3.csv, first collection).
constexpr int kGrpBlocks = 8;
__global__ __launch_bounds__(kThreads) void k_cell_power_groups(const float2 *__restrict__ spec, const float *__restrict__ gpow, int N,
                                                                const PowerCell *__restrict__ cells, int ncells, int nblocks, float *__restrict__ out)
{
    const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;      // 16 lane groups per workgroup: 16 cells
    const int cell = blockIdx.x * (kThreads / 16) + grp;
    const int mb = blockIdx.y * kGrpBlocks;
    const bool on = cell < ncells;
    const PowerCell pc = cells[on ? cell : 0];
    const int end = pc.start + pc.len, g0 = (pc.start + 15) >> 4, g1 = end >> 4;
    const bool any = g1 >= g0;                                     // a group boundary inside the cell
    const int hc = any ? 16 * g0 - pc.start : pc.len, tc = any ? end - 16 * g1 : 0, ng = any ? g1 - g0 : 0;    // hc < 31, tc < 16
    const int ngrp = N >> 4;
    float acc[kGrpBlocks];
    float2 h0[kGrpBlocks], h1[kGrpBlocks], t0[kGrpBlocks];
    float gq[kGrpBlocks];
#pragma unroll
    for (int j = 0; j < kGrpBlocks; j++) {
        const int m = mb + j < nblocks ? mb + j : nblocks - 1;
        const float2 *x = spec + (size_t)m * N;
        h0[j] = (on && sub < hc) ? x[pc.start + sub] : make_float2(0.f, 0.f);
        h1[j] = (on && sub + 16 < hc) ? x[pc.start + sub + 16] : make_float2(0.f, 0.f);
        t0[j] = (on && sub < tc) ? x[16 * g1 + sub] : make_float2(0.f, 0.f);
        gq[j] = (on && sub < ng) ? gpow[(size_t)m * ngrp + g0 + sub] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < kGrpBlocks; j++) {
        acc[j] = h0[j].x * h0[j].x + h0[j].y * h0[j].y + h1[j].x * h1[j].x + h1[j].y * h1[j].y + t0[j].x * t0[j].x + t0[j].y * t0[j].y + gq[j];
        if (ng > 16) {                                            // wide cells (a PowerActivationChannel's measure range can be any width)
            const int m = mb + j < nblocks ? mb + j : nblocks - 1;
            const float *gp = gpow + (size_t)m * ngrp + g0;
            for (int g = sub + 16; g < ng; g += 16) acc[j] += gp[g];
        }
    }
#pragma unroll
    for (int j = 0; j < kGrpBlocks; j++) {
        float t = acc[j];
        t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 4, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 1, 64);
        if (on && sub == 0 && mb + j < nblocks) out[(size_t)(mb + j) * ncells + cell] = t * pc.scale;
    }
}

// the same group sums from a spectrum in memory: launch groups the block kernel did not transform (short calls on the two-pass transform, block
// lengths without a block kernel)
__global__ __launch_bounds__(256) void k_group_power(const float4 *__restrict__ spec2, size_t ngroups, float *__restrict__ gpow)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < ngroups * 8; i += (size_t)gridDim.x * 256) {
        const float4 v = spec2[i];                                 // two bins; eight lanes per group
        float s = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
        if ((i & 7) == 0) gpow[i >> 3] = s;
    }
}

// Extraction of one width class: out = IFFT_w( halfswap( X[slot][start .. start+w) * win ) )[skip .. w)
//   lib/PowerActivationChannel_impl.cc:260-284, lib/activity_detection_channelizer_vcm_impl.cc:373-397
template <int NB>
__device__ __forceinline__ void extract_tile(const int tile, const float2 *__restrict__ spec, int N,
                                             const ExtractTask *__restrict__ tasks, int ntasks, int log2w,
                                             int log2TC, int ld, int skip, const float2 *__restrict__ wins,
                                             float2 *__restrict__ out, const float2 *__restrict__ tw,
                                             int twstride)
{
    float2 *lds = reinterpret_cast<float2 *>(fdc_smem);
    const int w = 1 << log2w, TC = 1 << log2TC;
    const int t0 = tile * TC;
    constexpr int PT = 16 * NB;
    __shared__ ColInfo col[32];
    const int total = w << log2TC;
    if ((int)threadIdx.x < TC) {
        const int t = t0 + threadIdx.x;
        ColInfo ci{0, 0, 0, 0};
        if (t < ntasks) {
            const ExtractTask tk = tasks[t];
            ci.src = (long long)tk.slot * N + tk.start; ci.win = tk.win_off; ci.dst = tk.out_off - skip; ci.valid = 1;
        }
        col[threadIdx.x] = ci;
    }
    __syncthreads();
    float2 v[PT], ww[PT];
#pragma unroll
    for (int u = 0; u < PT; u++) {
        const int e = threadIdx.x + u * kThreads;
        const int tl = (e >> log2w) & (TC - 1), i = e & (w - 1);
        v[u] = make_float2(0.f, 0.f); ww[u] = v[u];
        const ColInfo ci = col[tl];
        if (e < total && ci.valid) {
            v[u] = spec[ci.src + i];
            ww[u] = wins[ci.win + i];
        }
    }
#pragma unroll
    for (int u = 0; u < PT; u++) {
        const int e = threadIdx.x + u * kThreads;
        const int tl = e >> log2w, i = e & (w - 1);
        if (e < total) lds[((i + (w >> 1)) & (w - 1)) * ld + tl] = cmulf(v[u], ww[u]);       // fftshift(): halves swapped
    }
    __syncthreads();
    fft_cols<true, NB>(lds, log2w, log2TC, ld, tw, twstride);
#pragma unroll
    for (int u = 0; u < PT; u++) {
        const int e = threadIdx.x + u * kThreads;
        const int tl = (e >> log2w) & (TC - 1), i = e & (w - 1);
        const ColInfo ci = col[tl];
        if (e < total && ci.valid && i >= skip) out[ci.dst + i] = lds[i * ld + tl];
    }
}

template <int NB>
__global__ FDC_TILE_BOUNDS(NB) void k_extract(const float2 *__restrict__ spec, int N,
                                                      const ExtractTask *__restrict__ tasks, int ntasks, int log2w,
                                                      int log2TC, int ld, int skip, const float2 *__restrict__ wins,
                                                      float2 *__restrict__ out, const float2 *__restrict__ tw,
                                                      int twstride)
{
    extract_tile<NB>((int)blockIdx.x, spec, N, tasks, ntasks, log2w, log2TC, ld, skip, wins, out, tw, twstride);
}

// Channels wider than 4096 bins of a pipeline as extraction tasks (one per channel and block of the launch group), so that they take the
// task-addressed two-pass transform of the sinks: slice = bins [f, f + l) of block m's spectrum, window = the channel's table at the
// block's phase (phase_shifting_windowing_vcc's counter in closed form), output = the channel's stream at block mbase + m.
__global__ __launch_bounds__(256) void k_wide_tasks(ExtractTask *__restrict__ tasks, const ChanDev *__restrict__ chans, const int32_t *__restrict__ group,
                                                    int ngroup, int R, int nb_chunk, int mbase, int nb_call, long long first_block)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)nb_chunk * ngroup) return;
    const int m = (int)(t / ngroup), gi = (int)(t - (long long)m * ngroup);
    const ChanDev ch = chans[group[gi]];
    const int cnt = (int)((((first_block + mbase + m) % R) * ch.shift) % R);
    ExtractTask e{};
    e.slot = m; e.start = ch.f; e.win_off = ch.win_off + cnt * ch.l;
    e.out_off = (long long)nb_call * ch.out_off + (long long)(mbase + m) * ch.lout;
    tasks[t] = e;
}

hipError_t launch_wide_tasks(ExtractTask *tasks, const ChanDev *chans, const int32_t *group, int ngroup, int R, int nb_chunk, int mbase, int nb_call,
                             int64_t first_block, hipStream_t s)
{
    const long long n = (long long)nb_chunk * ngroup;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_wide_tasks, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, tasks, chans, group, ngroup, R, nb_chunk, mbase, nb_call,
                       (long long)first_block);
    return hipGetLastError();
}

// Several width classes (each up to 4096 points: the one-transform-per-workgroup tiling) in ONE launch: a bank of detected channels
// has a handful of classes with a few hundred to a few thousand extractions each, none of which fills the device on its own.
// A workgroup finds its class from the tile ranges.
__global__ FDC_TILE_BOUNDS(1) void k_extract_multi(const float2 *__restrict__ spec, int N, const ExtractTask *__restrict__ tasks,
                                                      const ExtractClasses cls, const float2 *__restrict__ wins, float2 *__restrict__ out,
                                                      const float2 *__restrict__ tw, int ntab)
{
    int c = 0;
    while (c + 1 < cls.n && (int)blockIdx.x >= cls.c[c + 1].tile0) c++;
    const ExtractClass ec = cls.c[c];
    extract_tile<1>((int)blockIdx.x - ec.tile0, spec, N, tasks + ec.task0, ec.ntasks, ec.log2w, ec.log2TC, ec.ld, ec.skip, wins, out, tw, ntab >> ec.log2w);
}

// multiply_const_cc(1/N, N) on spectrum items that arrive already transformed (hier block with inpveclen > 1,
// python/FrequencyDomainChannelizer.py:213-216, :289-290)
__global__ void k_scale(const float2 *__restrict__ in, float2 *__restrict__ out, size_t n, float k)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float2 v = in[i];
        out[i] = make_float2(v.x * k, v.y * k);
    }
}

// ---- single-block faces ----------------------------------------------------------------------------
__global__ void k_copy_items(const unsigned char *__restrict__ in, unsigned char *__restrict__ out,
                             size_t in_item_stride, size_t in_offset, size_t out_item_bytes)
{
    // one grid row per item; byte-granular so any itemsize works (the reference blocks are type-agnostic)
    const size_t m = blockIdx.y;
    const unsigned char *s = in + m * in_item_stride + in_offset;
    unsigned char *d = out + m * out_item_bytes;
    const size_t start = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    const size_t step = (size_t)gridDim.x * blockDim.x * 16;
    const bool aligned = ((((uintptr_t)s) | ((uintptr_t)d)) & 15) == 0;
    for (size_t b = start; b < out_item_bytes; b += step) {
        if (aligned && b + 16 <= out_item_bytes) {
            *reinterpret_cast<uint4 *>(d + b) = *reinterpret_cast<const uint4 *>(s + b);
        } else {
            const size_t e = b + 16 < out_item_bytes ? b + 16 : out_item_bytes;
            for (size_t k = b; k < e; k++) d[k] = s[k];
        }
    }
}

__global__ void k_phase_window(const float2 *__restrict__ in, float2 *__restrict__ out,
                               const float2 *__restrict__ win, int l, int R, int shift, int counter0)
{
    const int m = blockIdx.y;
    const int cnt = (int)(((long long)counter0 + (long long)(m % R) * shift) % R);
    const float2 *w = win + (size_t)cnt * l;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < l; i += gridDim.x * blockDim.x)
        out[(size_t)m * l + i] = cmulf(in[(size_t)m * l + i], w[i]);
}

// ---- host side ---------------------------------------------------------------------------------------
static int ilog2(int v) { int r = 0; while ((1 << r) < v) r++; return r; }

TileGeom tile_geom(int L)
{
    TileGeom g;
    g.L = L; g.log2L = ilog2(L);
    g.NB = L > 4096 ? 2 : 1;                    // tile = 4096*NB points
    int tc = 4096 * g.NB / L; if (tc > 32) tc = 32; if (tc < 1) tc = 1;
    g.TC = tc; g.log2TC = ilog2(tc);
    g.ld = tc > 1 ? tc + 1 : 1;
    return g;
}

BigGeom big_geom(int N)
{
    BigGeom g;
    const int lg = ilog2(N);
    const int lg1 = (lg + 1) / 2;          // N1 >= N2
    g.N = N; g.N1 = 1 << lg1; g.N2 = 1 << (lg - lg1);
    g.a = tile_geom(g.N2);
    g.b = tile_geom(g.N1);
    return g;
}

static hipError_t init_p1g_kernels();      // below, behind k_p1g

hipError_t init_kernels()
{
    const int maxlds = 72 * 1024;      // largest generic tile: 8192 points (64 KiB) + static per-column records
    hipError_t e;
#define FDC_SETLDS(k) \
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, maxlds); \
    if (e != hipSuccess) return e;
    FDC_SETLDS((k_fft_small<false, 2>)) FDC_SETLDS((k_fft_small<true, 2>))
    FDC_SETLDS(k_channels<2>) FDC_SETLDS(k_extract<2>) FDC_SETLDS(k_extract_multi)
    if ((e = init_p1g_kernels()) != hipSuccess) return e;
#undef FDC_SETLDS
    if ((e = init_sink_kernels()) != hipSuccess) return e;
    if ((e = init_block512_kernels()) != hipSuccess) return e;
    if ((e = init_block1024_kernels()) != hipSuccess) return e;
    if ((e = init_block_narrow_kernels()) != hipSuccess) return e;
    return init_fast_kernels();
}

hipError_t launch_fft(const float2 *in, size_t in_stride, float2 *out, float2 *tmp, int N, int nitems,
                      bool inverse, int in_rot, int out_rot, float scale, const float2 *tw, int ntab,
                      hipStream_t s, hipEvent_t *ev, const float2 *twf, bool generic_only, unsigned long long keep4096)
{
    if (nitems <= 0) return hipSuccess;
    hipError_t e;
    if (ev && (e = hipEventRecord(ev[0], s)) != hipSuccess) return e;
    if (N == 4096 && ntab % 4096 == 0 && !generic_only) {
        if ((e = launch_fft4096(in, in_stride, out, nitems, inverse, in_rot, out_rot, scale, tw, ntab, s, keep4096)) != hipSuccess) return e;
        if (ev && (e = hipEventRecord(ev[1], s)) != hipSuccess) return e;
    } else if (N <= kMaxLdsFft) {
        const TileGeom g = tile_geom(N);
        dim3 grid((nitems + g.TC - 1) / g.TC);
#define FDC_LS(I, B) \
    hipLaunchKernelGGL((k_fft_small<I, B>), grid, dim3(kThreads), g.lds_bytes(), s, in, in_stride, out, g.log2L, g.log2TC, g.ld, \
                       nitems, in_rot, out_rot, scale, tw, ntab / N)
        if (inverse) { if (g.NB == 2) FDC_LS(true, 2); else FDC_LS(true, 1); }
        else { if (g.NB == 2) FDC_LS(false, 2); else FDC_LS(false, 1); }
#undef FDC_LS
        if (ev && (e = hipEventRecord(ev[1], s)) != hipSuccess) return e;
    } else {
        const BigGeom g = big_geom(N);
        const int lgN = ilog2(N), lgN1 = ilog2(g.N1);
        for (int m0 = 0; m0 < nitems; m0 += 32768) {   // gridDim.y limit is 65535
            const int nb = nitems - m0 < 32768 ? nitems - m0 : 32768;
            dim3 ga(g.N1 / g.a.TC, nb), gb(g.N2 / g.b.TC, nb);
            const float2 *src = in + (size_t)m0 * in_stride;
            float2 *t = tmp + (size_t)m0 * N, *dst = out + (size_t)m0 * N;
            // N <= 2^24: both factors are <= 4096, one 4096-point tile per workgroup
            if (inverse)
                hipLaunchKernelGGL((k_fft_pass_a<true, 1>), ga, dim3(kThreads), g.a.lds_bytes(), s, src, in_stride, t, lgN,
                                   lgN1, g.a.log2TC, g.a.ld, in_rot, tw, ntab, twf);
            else
                hipLaunchKernelGGL((k_fft_pass_a<false, 1>), ga, dim3(kThreads), g.a.lds_bytes(), s, src, in_stride, t, lgN,
                                   lgN1, g.a.log2TC, g.a.ld, in_rot, tw, ntab, twf);
            if (m0 == 0 && ev && (e = hipEventRecord(ev[1], s)) != hipSuccess) return e;
            if (inverse)
                hipLaunchKernelGGL((k_fft_pass_b<true, 1>), gb, dim3(kThreads), g.b.lds_bytes(), s, t, dst, lgN, lgN1,
                                   g.b.log2TC, g.b.ld, out_rot, scale, tw, ntab);
            else
                hipLaunchKernelGGL((k_fft_pass_b<false, 1>), gb, dim3(kThreads), g.b.lds_bytes(), s, t, dst, lgN, lgN1,
                                   g.b.log2TC, g.b.ld, out_rot, scale, tw, ntab);
        }
    }
    if (ev && (e = hipEventRecord(ev[2], s)) != hipSuccess) return e;
    return hipGetLastError();
}

hipError_t launch_channels(const float2 *spec, float2 *out, const ChanDev *chans, const int32_t *group,
                           int ngroup, int l, int N, int R, int nb_chunk, int mbase, int nb_call,
                           int64_t first_block, const float2 *wins, const float2 *tw, int ntab, hipStream_t s)
{
    if (nb_chunk <= 0 || ngroup <= 0) return hipSuccess;
    const TileGeom g = tile_geom(l);
    const long long ntrans = (long long)nb_chunk * ngroup;
    dim3 grid((unsigned)((ntrans + g.TC - 1) / g.TC));
    if (g.NB == 2)
        hipLaunchKernelGGL(k_channels<2>, grid, dim3(kThreads), g.lds_bytes(), s, spec, out, chans, group, ngroup, g.log2L,
                           g.log2TC, g.ld, N, R, nb_chunk, mbase, nb_call, (long long)first_block, wins, tw, ntab / l);
    else
        hipLaunchKernelGGL(k_channels<1>, grid, dim3(kThreads), g.lds_bytes(), s, spec, out, chans, group, ngroup, g.log2L,
                           g.log2TC, g.ld, N, R, nb_chunk, mbase, nb_call, (long long)first_block, wins, tw, ntab / l);
    return hipGetLastError();
}

hipError_t launch_poly_stage2_generic(const float2 *g, float2 *out, int N1, int R, int nb_chunk, int mbase, int nb_call,
                                      const long long *slot_off, const float2 *tw, int ntab, hipStream_t s, int L)
{
    const int lout = L - L / R;
    int tr = 4096 / N1; if (tr > 32) tr = 32; if (tr < 1) tr = 1;
    while (lout % tr) tr >>= 1;                                       // tiles are whole rows of one block
    const TileGeom tg = tile_geom(N1);
    const int log2TR = ilog2(tr), ld = tr > 1 ? tr + 1 : 1;
    const long long ntiles = (long long)nb_chunk * (lout / tr);
    hipLaunchKernelGGL(k_p2g, dim3((unsigned)ntiles), dim3(kThreads), (size_t)N1 * ld * sizeof(float2), s, g, out, tg.log2L,
                       log2TR, ld, lout, slot_off, (long long)mbase * lout, (long long)nb_call, tw, ntab / N1, L != 256 ? 1 : 0);
    return hipGetLastError();
}

// ---- uniform plans of any width L (round 4): stage 1 on the generic LDS core -----------------------------------------------------
// Every channel l = L on the L-bin grid (f = L slot), one window: the commutation of fdc_fast256.hip does not depend on L = 256 —
//   y_c[t] = sum_n1 W_N1^(n1 k1) G[n1][t],   G[n1][t] = IFFT_L{ shape[k2]/N W_N^(n1 k2) (-1)^n1 FFT_L{ x[n1 + N1 n2] } },  N1 = N / L
// (ifftshift = the product lands at k2 ^ L/2; the discard keeps t >= L/R).  A workgroup takes TC columns n1 of one block: gather
// (rows of TC x 8 bytes, N1 samples apart), FFT over n2, table product, inverse FFT, kept rows to G[m][t'][n1] (rows of N1 columns:
// k_p2g reads them with rowmajor = 1).  Two launches, G through memory: the form k_p1 + k_p2 have for L = 256, without their
// register kernels — for uniform banks of other widths, which would otherwise take the spectrum path.
template <int NB>
__global__ FDC_TILE_BOUNDS(NB) void k_p1g(const float2 *__restrict__ in, size_t in_stride, float2 *__restrict__ g, int log2L, int log2TC,
                                          int ld, int log2N1, int skip, const float *__restrict__ shn, const float2 *__restrict__ tw,
                                          int ntab, const float2 *__restrict__ t2)
{
    float2 *lds = reinterpret_cast<float2 *>(fdc_smem);
    const int L = 1 << log2L, TC = 1 << log2TC, N1 = 1 << log2N1, total = L << log2TC, lout = L - skip;
    float2 *t1 = lds + L * ld;                                            // [L]: shape[k2]/N W_N^(c0 k2), c0 = this tile's first column
    const int c0 = blockIdx.x << log2TC;
    const size_t m = blockIdx.y;
    const float2 *src = in + m * in_stride + c0;
    const int twn = ntab >> (log2L + log2N1);                             // table entries per step of W_N
    const int nmask = (ntab / twn) - 1;                                   // N - 1
    // gather, 16 elements per thread in flight at a time (all offsets fit 32 bits: < N)
#pragma unroll
    for (int h = 0; h < NB; h++) {
        float2 v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int e = threadIdx.x + (16 * h + u) * kThreads;
            v[u] = make_float2(0.f, 0.f);
            if (e < total) v[u] = src[((e >> log2TC) << log2N1) + (e & (TC - 1))];
        }
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int e = threadIdx.x + (16 * h + u) * kThreads;
            if (e < total) lds[(e >> log2TC) * ld + (e & (TC - 1))] = v[u];
        }
    }
    // W_N^(n1 k2) = W_N^(c0 k2) W_N^(t k2): the first factor is this tile's (L values, gathered once into LDS together with the
    // window), the second is the same for every tile and block: t2[k2][t], read in the tile's own order (coalesced)
    for (int k2 = threadIdx.x; k2 < L; k2 += kThreads) {
        const float2 w = tw[(int)(((long long)c0 * k2) & nmask) * twn];
        const float sc = shn[k2];
        t1[k2] = make_float2(w.x * sc, w.y * sc);
    }
    __syncthreads();
    fft_cols<false, NB>(lds, log2L, log2TC, ld, tw, ntab >> log2L);
    // shape[k2]/N * (-1)^n1 * W_N^(n1 k2), placed at the ifftshifted position: an element trades places with its partner k2 ^ L/2,
    // which the SAME thread holds 8 * NB steps further on (e and e + total/2 differ by kThreads * 8 * NB exactly when total = 4096 * NB)
    if (total == 4096 * NB) {
#pragma unroll
        for (int u = 0; u < 8 * NB; u++) {
            const int e = threadIdx.x + u * kThreads, e2 = e + (total >> 1);
            const int t = e & (TC - 1), k2 = e >> log2TC, k3 = e2 >> log2TC;
            const float sg = ((c0 + t) & 1) ? -1.f : 1.f;
            const float2 a = cmulf(cmulf(lds[k2 * ld + t], t1[k2]), t2[e]);
            const float2 b = cmulf(cmulf(lds[k3 * ld + t], t1[k3]), t2[e2]);
            lds[k3 * ld + t] = make_float2(a.x * sg, a.y * sg);
            lds[k2 * ld + t] = make_float2(b.x * sg, b.y * sg);
        }
    } else {                                                              // small tiles (fewer columns than a full tile): two phases
        float2 v[16 * NB];
#pragma unroll
        for (int u = 0; u < 16 * NB; u++) {
            const int e = threadIdx.x + u * kThreads;
            const int t = e & (TC - 1), k2 = e >> log2TC;
            if (e < total) {
                const float sg = ((c0 + t) & 1) ? -1.f : 1.f;
                const float2 a = cmulf(cmulf(lds[k2 * ld + t], t1[k2]), t2[e]);
                v[u] = make_float2(a.x * sg, a.y * sg);
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 16 * NB; u++) {
            const int e = threadIdx.x + u * kThreads;
            if (e < total) lds[((e >> log2TC) ^ (L >> 1)) * ld + (e & (TC - 1))] = v[u];
        }
    }
    __syncthreads();
    fft_cols<true, NB>(lds, log2L, log2TC, ld, tw, ntab >> log2L);
    float2 *dst = g + m * (size_t)lout * N1 + c0;
#pragma unroll
    for (int u = 0; u < 16 * NB; u++) {
        const int e = threadIdx.x + u * kThreads;
        const int t = e & (TC - 1), tt = e >> log2TC;
        if (e < total && tt >= skip) dst[((tt - skip) << log2N1) + t] = lds[tt * ld + t];
    }
}

static hipError_t init_p1g_kernels()
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_p1g<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_p1g<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    return e;
}

// tile of the generic-width stage 1: columns per workgroup
static TileGeom p1g_geom(int N, int L)
{
    TileGeom tg = tile_geom(L);
    // (8192-point tiles for L = 512 / 1024 — 128-byte row pieces, but 256 VGPRs and two waves per SIMD — measured slower than these
    // 4096-point ones: stage 1 0.83 against 0.73 ms per 2048 blocks at L = 512)
    if (tg.TC > N / L) { tg.TC = N / L; tg.log2TC = ilog2(tg.TC); tg.ld = tg.TC > 1 ? tg.TC + 1 : 1; }
    return tg;
}
int poly_stage1_generic_tile_columns(int N, int L) { return p1g_geom(N, L).TC; }

hipError_t launch_poly_stage1_generic(const float2 *in, size_t in_stride, float2 *g, int N, int L, int R, int nb_chunk, const float *shn,
                                      const float2 *tw, int ntab, const float2 *t2, hipStream_t s)
{
    if (nb_chunk <= 0) return hipSuccess;
    const int N1 = N / L;
    const TileGeom tg = p1g_geom(N, L);
    for (int m0 = 0; m0 < nb_chunk; m0 += 32768) {                        // gridDim.y limit
        const int nb = nb_chunk - m0 < 32768 ? nb_chunk - m0 : 32768;
        dim3 grid((unsigned)(N1 / tg.TC), (unsigned)nb);
#define FDC_L1G(B) \
        hipLaunchKernelGGL((k_p1g<B>), grid, dim3(kThreads), tg.lds_bytes() + (size_t)L * sizeof(float2), s, in + (size_t)m0 * in_stride, in_stride, \
                           g + (size_t)m0 * (size_t)(L - L / R) * N1, tg.log2L, tg.log2TC, tg.ld, ilog2(N1), L / R, shn, tw, ntab, t2)
        if (tg.NB == 2) FDC_L1G(2); else FDC_L1G(1);
#undef FDC_L1G
    }
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_scatter_out(const float2 *__restrict__ src, const ScatterEnt *__restrict__ tab,
                                                     int nb, long long row0)
{
    const ScatterEnt e = tab[blockIdx.y];
    if (!e.dst) return;
    const size_t n = (size_t)nb * e.lout;
    const float2 *s = src + (size_t)nb * e.out_off;
    float2 *d = e.dst + (size_t)row0 * e.lout;
    if ((((uintptr_t)s | (uintptr_t)d) & 15) == 0) {            // 16 bytes per lane: 1 KiB PCIe write runs per wave
        const size_t n2 = n >> 1;
        const float4 *s4 = reinterpret_cast<const float4 *>(s);
        float4 *d4 = reinterpret_cast<float4 *>(d);
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) d4[i] = s4[i];
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) d[n - 1] = s[n - 1];
    } else {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = s[i];
    }
}

hipError_t launch_scatter_out(const float2 *src, const ScatterEnt *tab, int nchan, int nb, long long row0, hipStream_t s)
{
    if (nchan <= 0 || nb <= 0) return hipSuccess;
    for (int c0 = 0; c0 < nchan; c0 += 32768) {
        const int nc = nchan - c0 < 32768 ? nchan - c0 : 32768;
        hipLaunchKernelGGL(k_scatter_out, dim3(4, nc), dim3(256), 0, s, src, tab + c0, nb, row0);
    }
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_real_to_complex(const float *__restrict__ in, float2 *__restrict__ out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = make_float2(in[i], 0.f);
}

hipError_t launch_real_to_complex(const float *in, float2 *out, size_t n, hipStream_t s)
{
    if (!n) return hipSuccess;
    size_t g = (n + 255) / 256; if (g > 8192) g = 8192;
    hipLaunchKernelGGL(k_real_to_complex, dim3((unsigned)g), dim3(256), 0, s, in, out, n);
    return hipGetLastError();
}

hipError_t launch_scale(const float2 *in, float2 *out, size_t n, float k, hipStream_t s)
{
    if (!n) return hipSuccess;
    size_t g = (n + 255) / 256; if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_scale, dim3((unsigned)g), dim3(256), 0, s, in, out, n, k);
    return hipGetLastError();
}

hipError_t launch_cell_power(const float2 *spec, int N, const PowerCell *cells, int ncells, int nblocks, float *out,
                             hipStream_t s)
{
    if (ncells <= 0 || nblocks <= 0) return hipSuccess;
    for (int m0 = 0; m0 < nblocks; m0 += 32768) {
        const int nb = nblocks - m0 < 32768 ? nblocks - m0 : 32768;
        hipLaunchKernelGGL(k_cell_power, dim3((ncells + 3) / 4, (nb + kCellBlocks - 1) / kCellBlocks), dim3(kThreads), 0, s, spec + (size_t)m0 * N, N, cells,
                           ncells, nb, out + (size_t)m0 * ncells);
    }
    return hipGetLastError();
}

hipError_t launch_cell_power_groups(const float2 *spec, const float *gpow, int N, const PowerCell *cells, int ncells, int nblocks, float *out, hipStream_t s)
{
    if (ncells <= 0 || nblocks <= 0) return hipSuccess;
    if (N & 15) return hipErrorInvalidValue;
    for (int m0 = 0; m0 < nblocks; m0 += 32768) {
        const int nb = nblocks - m0 < 32768 ? nblocks - m0 : 32768;
        hipLaunchKernelGGL(k_cell_power_groups, dim3((ncells + kThreads / 16 - 1) / (kThreads / 16), (nb + kGrpBlocks - 1) / kGrpBlocks), dim3(kThreads), 0, s, spec + (size_t)m0 * N,
                           gpow + (size_t)m0 * (N >> 4), N, cells, ncells, nb, out + (size_t)m0 * ncells);
    }
    return hipGetLastError();
}

hipError_t launch_group_power(const float2 *spec, int N, int nblocks, float *gpow, hipStream_t s)
{
    if (nblocks <= 0) return hipSuccess;
    if (N & 15) return hipErrorInvalidValue;
    const size_t ngroups = (size_t)nblocks * (size_t)(N >> 4);
    size_t g = (ngroups * 8 + 255) / 256; if (g > 16384) g = 16384;
    hipLaunchKernelGGL(k_group_power, dim3((unsigned)g), dim3(256), 0, s, reinterpret_cast<const float4 *>(spec), ngroups, gpow);
    return hipGetLastError();
}

hipError_t launch_extract(const float2 *spec, int N, const ExtractTask *tasks, int ntasks, int w, int skip,
                          const float2 *wins, float2 *out, const float2 *tw, int ntab, hipStream_t s)
{
    if (ntasks <= 0) return hipSuccess;
    const TileGeom g = tile_geom(w);
    if (g.NB == 2)
        hipLaunchKernelGGL(k_extract<2>, dim3((ntasks + g.TC - 1) / g.TC), dim3(kThreads), g.lds_bytes(), s, spec, N, tasks, ntasks,
                           g.log2L, g.log2TC, g.ld, skip, wins, out, tw, ntab / w);
    else
        hipLaunchKernelGGL(k_extract<1>, dim3((ntasks + g.TC - 1) / g.TC), dim3(kThreads), g.lds_bytes(), s, spec, N, tasks, ntasks,
                           g.log2L, g.log2TC, g.ld, skip, wins, out, tw, ntab / w);
    return hipGetLastError();
}

// One width class above 4096 points: the two-pass inverse transform with the slice * window read in pass A and the kept samples
// written to their landing offsets by pass B.  tmp: ntasks * w points.
hipError_t launch_extract_wide(const float2 *spec, int N, const ExtractTask *tasks, int ntasks, int w, int skip, const float2 *wins,
                               float2 *tmp, float2 *out, const float2 *tw, int ntab, hipStream_t s, float scale)
{
    if (ntasks <= 0) return hipSuccess;
    const BigGeom g = big_geom(w);
    const int lgN = ilog2(w), lgN1 = ilog2(g.N1);
    if (g.a.NB != 1 || g.b.NB != 1) return hipErrorInvalidValue;
    for (int m0 = 0; m0 < ntasks; m0 += 32768) {       // gridDim.y limit is 65535
        const int nb = ntasks - m0 < 32768 ? ntasks - m0 : 32768;
        dim3 ga(g.N1 / g.a.TC, nb), gb(g.N2 / g.b.TC, nb);
        hipLaunchKernelGGL((k_fft_pass_a<true, 1, true>), ga, dim3(kThreads), g.a.lds_bytes(), s, spec, (size_t)N, tmp, lgN, lgN1, g.a.log2TC,
                           g.a.ld, w / 2, tw, ntab, static_cast<const float2 *>(nullptr), tasks + m0, wins);
        hipLaunchKernelGGL((k_fft_pass_b<true, 1, true>), gb, dim3(kThreads), g.b.lds_bytes(), s, tmp, out, lgN, lgN1, g.b.log2TC, g.b.ld, 0,
                           scale, tw, ntab, tasks + m0, skip);
    }
    return hipGetLastError();
}

hipError_t launch_extract_multi(const float2 *spec, int N, const ExtractTask *tasks, const int *w, const size_t *first, const size_t *cnt,
                                int nclass, int R, const float2 *wins, float2 *out, const float2 *tw, int ntab, hipStream_t s)
{
    ExtractClasses cls{};
    size_t lds = 0;
    int tiles = 0;
    for (int k = 0; k < nclass && cls.n < kMaxExtractClasses; k++) {
        if (!cnt[k]) continue;
        const TileGeom g = tile_geom(w[k]);
        if (g.NB != 1) return hipErrorInvalidValue;
        const int nt = (int)((cnt[k] + (size_t)g.TC - 1) / (size_t)g.TC);
        cls.c[cls.n++] = ExtractClass{tiles, (int)first[k], (int)cnt[k], g.log2L, g.log2TC, g.ld, w[k] / R, 0};
        tiles += nt;
        lds = std::max(lds, g.lds_bytes());
    }
    if (!tiles) return hipSuccess;
    hipLaunchKernelGGL(k_extract_multi, dim3((unsigned)tiles), dim3(kThreads), lds, s, spec, N, tasks, cls, wins, out, tw, ntab);
    return hipGetLastError();
}

hipError_t launch_overlap_save(const unsigned char *ring, unsigned char *out, size_t in_item_bytes,
                               size_t out_item_bytes, int nitems, hipStream_t s)
{
    // item i = ring bytes [i*in_item_bytes, i*in_item_bytes + out_item_bytes): the ring already starts
    // with the history (lib/overlap_save_impl.cc:70-76)
    if (nitems <= 0) return hipSuccess;
    unsigned gx = (unsigned)((out_item_bytes / 16 + 255) / 256); if (gx < 1) gx = 1; if (gx > 64) gx = 64;
    for (int m0 = 0; m0 < nitems; m0 += 32768) {
        const int nb = nitems - m0 < 32768 ? nitems - m0 : 32768;
        hipLaunchKernelGGL(k_copy_items, dim3(gx, nb), dim3(256), 0, s, ring + (size_t)m0 * in_item_bytes,
                           out + (size_t)m0 * out_item_bytes, in_item_bytes, (size_t)0, out_item_bytes);
    }
    return hipGetLastError();
}

hipError_t launch_vector_cut(const unsigned char *in, unsigned char *out, size_t in_item_bytes, size_t shift_bytes,
                             size_t out_item_bytes, int nitems, hipStream_t s)
{
    if (nitems <= 0) return hipSuccess;
    unsigned gx = (unsigned)((out_item_bytes / 16 + 255) / 256); if (gx < 1) gx = 1; if (gx > 64) gx = 64;
    for (int m0 = 0; m0 < nitems; m0 += 32768) {
        const int nb = nitems - m0 < 32768 ? nitems - m0 : 32768;
        hipLaunchKernelGGL(k_copy_items, dim3(gx, nb), dim3(256), 0, s, in + (size_t)m0 * in_item_bytes,
                           out + (size_t)m0 * out_item_bytes, in_item_bytes, shift_bytes, out_item_bytes);
    }
    return hipGetLastError();
}

hipError_t launch_phase_window(const float2 *in, float2 *out, const float2 *win, int l, int R, int shift,
                               int counter0, int nitems, hipStream_t s)
{
    if (nitems <= 0) return hipSuccess;
    unsigned gx = (unsigned)((l + 255) / 256); if (gx > 64) gx = 64;
    for (int m0 = 0; m0 < nitems; m0 += 32768) {
        const int nb = nitems - m0 < 32768 ? nitems - m0 : 32768;
        const int c0 = (int)(((long long)counter0 + (long long)(m0 % R) * shift) % R);
        hipLaunchKernelGGL(k_phase_window, dim3(gx, nb), dim3(256), 0, s, in + (size_t)m0 * l, out + (size_t)m0 * l, win,
                           l, R, shift, c0);
    }
    return hipGetLastError();
}


}  // namespace fdc
