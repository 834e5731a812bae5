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

namespace fdc {

extern __shared__ __attribute__((aligned(16))) unsigned char fdc_smem_fast[];

constexpr int kTileBytes = 256 * 32 * 8;          // 64 KiB of points
constexpr int kCTileBytes = 32 * 272 * 8;         // channel kernel: 32 rows padded to 272 points

__device__ __forceinline__ float4 ld4(const float2 *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float2 *p, float2 a, float2 b)
{
    *reinterpret_cast<float4 *>(p) = make_float4(a.x, a.y, b.x, b.y);
}

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
    float2 va[16], vb[16];
#pragma unroll
    for (int a = 0; a < 16; a++) {          // rows n2 = 16a+b, two adjacent columns per lane
        const float4 t = ld4(src + (size_t)(16 * a + b) * 256);
        va[a] = make_float2(t.x, t.y); vb[a] = make_float2(t.z, t.w);
    }
    w256[tid] = tw256[tid];
    dft16<false>(va); dft16<false>(vb);
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 16; p++) {          // W_256^(b*p), then row 16b+p of the exchange tile
        const float2 w = w256[b * p];
        st4(&tile[(16 * b + p) * 32 + 2 * cp], cmul(va[rev16(p)], w), cmul(vb[rev16(p)], w));
    }
    __syncthreads();
#pragma unroll
    for (int bb = 0; bb < 16; bb++) {       // this thread now owns p' = b
        const float4 t = ld4(&tile[(16 * bb + b) * 32 + 2 * cp]);
        va[bb] = make_float2(t.x, t.y); vb[bb] = make_float2(t.z, t.w);
    }
    dft16<false>(va); dft16<false>(vb);
    float2 *dst = tmp + m * 65536 + c0 + 2 * cp;
    const float2 *twp = twf + c0 + 2 * cp;
#pragma unroll
    for (int q = 0; q < 16; q++) {          // k2 = b + 16q ; inter-pass twiddle W_N^(n1*k2)
        const int k2 = b + 16 * q;
        const float4 w = ld4(twp + k2 * 256);
        st4(dst + k2 * 256, cmul(va[rev16(q)], make_float2(w.x, w.y)), cmul(vb[rev16(q)], make_float2(w.z, w.w)));
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
    float2 va[16], vb[16];
    {
        const int r = tid >> 3, bp = tid & 7;          // row k2 = r0+r, columns n1 = 16a + {2bp, 2bp+1}
        const float2 *src = tmp + m * 65536 + (size_t)(r0 + r) * 256 + 2 * bp;
#pragma unroll
        for (int a = 0; a < 16; a++) {
            const float4 t = ld4(src + 16 * a);
            va[a] = make_float2(t.x, t.y); vb[a] = make_float2(t.z, t.w);
        }
        w256[tid] = tw256[tid];
        dft16<false>(va); dft16<false>(vb);
        __syncthreads();
        const int col = r ^ (bp << 1);                 // XOR swizzle: conflict-free b64 writes
#pragma unroll
        for (int p = 0; p < 16; p++) {
            tile[(p * 16 + 2 * bp) * 32 + col] = cmul(va[rev16(p)], w256[(2 * bp) * p]);
            tile[(p * 16 + 2 * bp + 1) * 32 + col] = cmul(vb[rev16(p)], w256[(2 * bp + 1) * p]);
        }
    }
    __syncthreads();
    {
        const int rp = tid & 15, p = tid >> 4;         // rows k2 = r0 + {2rp, 2rp+1}, output k1 = p + 16q
#pragma unroll
        for (int bb = 0; bb < 16; bb++) {
            const float4 t = ld4(&tile[(p * 16 + bb) * 32 + ((2 * rp) ^ ((bb >> 1) << 1))]);
            va[bb] = make_float2(t.x, t.y); vb[bb] = make_float2(t.z, t.w);
        }
        dft16<false>(va); dft16<false>(vb);
        float2 *dst = spec + m * 65536;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int k = ((p + 16 * q) << 8) + r0 + 2 * rp;          // bin 256*k1 + k2
            st4(dst + ((k + out_rot) & 65535), cscale(va[rev16(q)], scale), cscale(vb[rev16(q)], scale));
        }
    }
}

// ---- fused channel kernel, l = 256 ------------------------------------------------------------------------
// One row = one (block, channel) pair.  ALIGNED: every channel's f is even (16-B aligned slice loads);
// OUT_ALIGNED: every channel's output offset and lout are even (16-B aligned stores).  Needs an even discard
// length l/R (the launcher falls back to the generic kernel otherwise).
template <bool ALIGNED, bool OUT_ALIGNED>
__global__ __launch_bounds__(256, 2) void k_c256(const float2 *__restrict__ spec, float2 *__restrict__ out,
                                                 const ChanDev *__restrict__ chans,
                                                 const int32_t *__restrict__ group, int ngroup, int N, int R,
                                                 int nb_chunk, int mbase, int nb_call, long long first_block,
                                                 const float2 *__restrict__ wins,
                                                 const float2 *__restrict__ tw256)
{
    float2 *tile = reinterpret_cast<float2 *>(fdc_smem_fast);
    float2 *w256 = reinterpret_cast<float2 *>(fdc_smem_fast + kCTileBytes);
    const int tid = threadIdx.x;
    const int r = tid >> 3, lane8 = tid & 7;
    const long long ntrans = (long long)nb_chunk * ngroup;
    const long long t = (long long)blockIdx.x * 32 + r;
    const bool live = t < ntrans;
    int m = 0;
    ChanDev ch;
    ch.f = 0; ch.l = 256; ch.lout = 128; ch.shift = 0; ch.out_off = 0; ch.win_off = 0; ch.pad = 0;
    if (live) {
        m = (int)(t / ngroup);
        ch = chans[group[(int)(t - (long long)m * ngroup)]];
    }
    float2 va[16], vb[16];
    {
        // slice bins i = 16a + {2bp, 2bp+1} (vector_cut_vxx), times W[cnt][i] (phase_shifting_windowing_vcc),
        // stored at the ifftshifted position i ^ 128 — i.e. a -> a ^ 8
        const int bp = lane8;
        const int cnt = (int)((((first_block + mbase + m) % R) * ch.shift) % R);
        const float2 *src = spec + (size_t)m * N + ch.f + 2 * bp;
        const float2 *wsrc = wins + ch.win_off + cnt * 256 + 2 * bp;
#pragma unroll
        for (int a = 0; a < 16; a++) {
            float2 x0 = make_float2(0.f, 0.f), x1 = x0;
            if (live) {
                if (ALIGNED) {
                    const float4 tt = ld4(src + 16 * a);
                    x0 = make_float2(tt.x, tt.y); x1 = make_float2(tt.z, tt.w);
                } else {
                    x0 = src[16 * a]; x1 = src[16 * a + 1];
                }
            }
            const float4 w = ld4(wsrc + 16 * a);
            va[a ^ 8] = cmul(x0, make_float2(w.x, w.y));
            vb[a ^ 8] = cmul(x1, make_float2(w.z, w.w));
        }
        w256[tid] = tw256[tid];
        dft16<true>(va); dft16<true>(vb);
        __syncthreads();
        // exchange layout: row r padded to 272 points; element (b, p) at b*16 + p, p-pairs swizzled by bp
        float2 *row = tile + r * 272;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float2 w0 = w256[(2 * bp) * (2 * j)], w1 = w256[(2 * bp) * (2 * j + 1)];
            w0.y = -w0.y; w1.y = -w1.y;                       // inverse transform: conjugate twiddles
            st4(row + (2 * bp) * 16 + 2 * (j ^ bp), cmul(va[rev16(2 * j)], w0), cmul(va[rev16(2 * j + 1)], w1));
            float2 w2 = w256[(2 * bp + 1) * (2 * j)], w3 = w256[(2 * bp + 1) * (2 * j + 1)];
            w2.y = -w2.y; w3.y = -w3.y;
            st4(row + (2 * bp + 1) * 16 + 2 * (j ^ bp), cmul(vb[rev16(2 * j)], w2), cmul(vb[rev16(2 * j + 1)], w3));
        }
    }
    __syncthreads();
    {
        const int pp = lane8;                                  // this thread owns p = 2pp, 2pp+1 of row r
        const float2 *row = tile + r * 272;
#pragma unroll
        for (int bb = 0; bb < 16; bb++) {
            const float4 tt = ld4(row + bb * 16 + 2 * (pp ^ (bb >> 1)));
            va[bb] = make_float2(tt.x, tt.y); vb[bb] = make_float2(tt.z, tt.w);
        }
        dft16<true>(va); dft16<true>(vb);
        if (live) {
            // y[t], t = p + 16q; keep t >= l/R (vector_cut_vxx(l, l-lout, lout)), times l (multiply_const_cc)
            const int skip = 256 - ch.lout;
            float2 *dst = out + (size_t)nb_call * ch.out_off + (size_t)(mbase + m) * ch.lout;
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int tt = 2 * pp + 16 * q;
                if (tt >= skip) {
                    if (OUT_ALIGNED) {
                        st4(dst + (tt - skip), cscale(va[rev16(q)], 256.f), cscale(vb[rev16(q)], 256.f));
                    } else {
                        dst[tt - skip] = cscale(va[rev16(q)], 256.f);
                        dst[tt - skip + 1] = cscale(vb[rev16(q)], 256.f);
                    }
                }
            }
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
__global__ __launch_bounds__(256, 2) void k_p1(const float2 *__restrict__ in, size_t in_stride,
                                               float2 *__restrict__ g, const float2 *__restrict__ tw256,
                                               const float2 *__restrict__ twm, int N1, int skip, int lout)
{
    float2 *tile = reinterpret_cast<float2 *>(fdc_smem_fast);
    float2 *w256 = reinterpret_cast<float2 *>(fdc_smem_fast + kTileBytes);
    const int tid = threadIdx.x, cp = tid & 15, b = tid >> 4;
    const int c0 = blockIdx.x * 32;
    const size_t m = blockIdx.y;
    const float2 *src = in + m * in_stride + c0 + 2 * cp;
    float2 va[16], vb[16];
#pragma unroll
    for (int a = 0; a < 16; a++) {
        const float4 t = ld4(src + (size_t)(16 * a + b) * N1);
        va[a] = make_float2(t.x, t.y); vb[a] = make_float2(t.z, t.w);
    }
    w256[tid] = tw256[tid];
    dft16<false>(va); dft16<false>(vb);
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 16; p++) {
        const float2 w = w256[b * p];
        st4(&tile[(16 * b + p) * 32 + 2 * cp], cmul(va[rev16(p)], w), cmul(vb[rev16(p)], w));
    }
    __syncthreads();
#pragma unroll
    for (int bb = 0; bb < 16; bb++) {
        const float4 t = ld4(&tile[(16 * bb + b) * 32 + 2 * cp]);
        va[bb] = make_float2(t.x, t.y); vb[bb] = make_float2(t.z, t.w);
    }
    dft16<false>(va); dft16<false>(vb);          // A[k2 = b + 16q] in va/vb[rev16(q)]
    // window * twiddle * sign * scale, placed at the ifftshifted position (k2 ^ 128  <=>  q ^ 8).  The thread
    // already holds exactly the inputs of ITS first inverse DFT-16 (fixed low digit b, all high digits q).
    float2 ua[16], ub[16];
    const float2 *twp = twm + c0 + 2 * cp;
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const float4 w = ld4(twp + (size_t)(b + 16 * q) * N1);
        ua[q ^ 8] = cmul(va[rev16(q)], make_float2(w.x, w.y));
        ub[q ^ 8] = cmul(vb[rev16(q)], make_float2(w.z, w.w));
    }
    dft16<true>(ua); dft16<true>(ub);
    __syncthreads();                                // every thread has finished reading the first exchange
#pragma unroll
    for (int p = 0; p < 16; p++) {
        float2 w = w256[b * p];
        w.y = -w.y;
        st4(&tile[(16 * b + p) * 32 + 2 * cp], cmul(ua[rev16(p)], w), cmul(ub[rev16(p)], w));
    }
    __syncthreads();
#pragma unroll
    for (int bb = 0; bb < 16; bb++) {
        const float4 t = ld4(&tile[(16 * bb + b) * 32 + 2 * cp]);
        ua[bb] = make_float2(t.x, t.y); ub[bb] = make_float2(t.z, t.w);
    }
    dft16<true>(ua); dft16<true>(ub);             // y[t = b + 16q]
    float2 *dst = g + m * (size_t)lout * N1 + c0 + 2 * cp;
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const int t = b + 16 * q;
        if (t >= skip) st4(dst + (size_t)(t - skip) * N1, ua[rev16(q)], ub[rev16(q)]);
    }
}

// Stage 2 for N1 = 256 slots: rows rho = m*lout + t' of G (256 contiguous n1 each), FFT over n1, bin = slot c.
__global__ __launch_bounds__(256, 2) void k_p2(const float2 *__restrict__ g, float2 *__restrict__ out,
                                               const float2 *__restrict__ tw256,
                                               const long long *__restrict__ slot_off, long long nrows,
                                               long long out_base, long long nb_call)
{
    float2 *tile = reinterpret_cast<float2 *>(fdc_smem_fast);
    float2 *w256 = reinterpret_cast<float2 *>(fdc_smem_fast + kTileBytes);
    const int tid = threadIdx.x;
    const long long r0 = (long long)blockIdx.x * 32;
    float2 va[16], vb[16];
    {
        const int r = tid >> 3, bp = tid & 7;
        const bool live = r0 + r < nrows;
        const float2 *src = g + (size_t)(r0 + r) * 256 + 2 * bp;
#pragma unroll
        for (int a = 0; a < 16; a++) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live) t = ld4(src + 16 * a);
            va[a] = make_float2(t.x, t.y); vb[a] = make_float2(t.z, t.w);
        }
        w256[tid] = tw256[tid];
        dft16<false>(va); dft16<false>(vb);
        __syncthreads();
        const int col = r ^ (bp << 1);
#pragma unroll
        for (int p = 0; p < 16; p++) {
            tile[(p * 16 + 2 * bp) * 32 + col] = cmul(va[rev16(p)], w256[(2 * bp) * p]);
            tile[(p * 16 + 2 * bp + 1) * 32 + col] = cmul(vb[rev16(p)], w256[(2 * bp + 1) * p]);
        }
    }
    __syncthreads();
    {
        const int rp = tid & 15, p = tid >> 4;
#pragma unroll
        for (int bb = 0; bb < 16; bb++) {
            const float4 t = ld4(&tile[(p * 16 + bb) * 32 + ((2 * rp) ^ ((bb >> 1) << 1))]);
            va[bb] = make_float2(t.x, t.y); vb[bb] = make_float2(t.z, t.w);
        }
        dft16<false>(va); dft16<false>(vb);
        const long long rho = r0 + 2 * rp;
        if (rho < nrows) {
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const long long off = slot_off[p + 16 * q];     // per-block sample offset of the channel in this slot
                if (off >= 0) st4(out + off * nb_call + out_base + rho, va[rev16(q)], vb[rev16(q)]);
            }
        }
    }
}

// ---- launchers ---------------------------------------------------------------------------------------------
hipError_t init_fast_kernels()
{
    hipError_t e;
    const int a = kTileBytes + 2048, c = kCTileBytes + 2048;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_a256), hipFuncAttributeMaxDynamicSharedMemorySize, a);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_b256), hipFuncAttributeMaxDynamicSharedMemorySize, a);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_p1), hipFuncAttributeMaxDynamicSharedMemorySize, a);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_p2), hipFuncAttributeMaxDynamicSharedMemorySize, a);
    if (e != hipSuccess) return e;
#define FDC_SETC(k) \
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, c); \
    if (e != hipSuccess) return e;
    FDC_SETC((k_c256<true, true>)) FDC_SETC((k_c256<true, false>)) FDC_SETC((k_c256<false, true>)) FDC_SETC((k_c256<false, false>))
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
    const long long ntrans = (long long)nb_chunk * ngroup;
    if (ntrans <= 0) return hipSuccess;
    dim3 grid((unsigned)((ntrans + 31) / 32));
#define FDC_LC(A, B) \
    hipLaunchKernelGGL((k_c256<A, B>), grid, dim3(256), kCTileBytes + 2048, s, spec, out, chans, group, ngroup, N, R, \
                       nb_chunk, mbase, nb_call, (long long)first_block, wins, tw256)
    if (aligned && out_aligned) FDC_LC(true, true);
    else if (aligned) FDC_LC(true, false);
    else if (out_aligned) FDC_LC(false, true);
    else FDC_LC(false, false);
#undef FDC_LC
    return hipGetLastError();
}

hipError_t launch_poly256(const float2 *in, size_t in_stride, float2 *g, float2 *out, int N1, int R, int nb_chunk,
                          int mbase, int nb_call, const float2 *tw256, const float2 *twm,
                          const long long *slot_off, hipStream_t s, hipEvent_t *ev)
{
    hipError_t e;
    const int skip = 256 / R, lout = 256 - skip;
    if (ev && (e = hipEventRecord(ev[0], s)) != hipSuccess) return e;
    for (int m0 = 0; m0 < nb_chunk; m0 += 32768) {
        const int nb = nb_chunk - m0 < 32768 ? nb_chunk - m0 : 32768;
        hipLaunchKernelGGL(k_p1, dim3(N1 / 32, nb), dim3(256), kTileBytes + 2048, s, in + (size_t)m0 * in_stride, in_stride,
                           g + (size_t)m0 * lout * N1, tw256, twm, N1, skip, lout);
    }
    if (ev && (e = hipEventRecord(ev[1], s)) != hipSuccess) return e;
    const long long nrows = (long long)nb_chunk * lout;
    hipLaunchKernelGGL(k_p2, dim3((unsigned)((nrows + 31) / 32)), dim3(256), kTileBytes + 2048, s, g, out, tw256, slot_off,
                       nrows, (long long)mbase * lout, (long long)nb_call);
    if (ev && (e = hipEventRecord(ev[2], s)) != hipSuccess) return e;
    return hipGetLastError();
}

}  // namespace fdc
