// Fused channel kernels for l = 512 and l = 1024 on gfx950 — the register form of k_c256 (fdc_fast256.hip) for the two
// wider channel classes that mixed plans use most (the reference's example flowgraph has one of each): slice
// (vector_cut_vxx) + phase/window (phase_shifting_windowing_vcc) + ifftshift + IFFT-l + overlap discard + *l in one pass,
// one (block, channel) pair per row of lanes, every lane holding 32 points.
//
//   l = 1024 = 32 x 32: a row is 32 lanes; DFT-32 over a (points 32a + b), twiddle W_1024^(-b p), 32 x 32 exchange inside
//       the row (LDS, element (b, p) at p*32 + (b ^ p): conflict-free for the 16-lane store groups and the 32-lane read
//       groups), DFT-32 over b; 8 rows per workgroup.
//   l =  512 = 32 x 16: a row is 16 lanes; DFT-32 over a (points 16a + b), twiddle W_512^(-b p), exchange, then every lane
//       does the DFT-16 over b for p = lane and p = lane + 16; 16 rows per workgroup.
// The generic LDS kernel (k_channels, fdc_kernels.hip) stays for every other width and for odd discard lengths.
#include "fdc_kernels.h"
#include "fdc_radix16.hpp"
#include "fdc_devutil.hpp"

namespace fdc {

extern __shared__ __attribute__((aligned(16))) unsigned char fdc_smem_wide[];

namespace {
struct RowInfoW { long long src; long long dst; int win; int valid; };
// dft32 leaves X[k0 + 2 k1] in v[16 k0 + rev16(k1)]
__device__ __forceinline__ constexpr int pos32(int k) { return 16 * (k & 1) + rev16(k >> 1); }

template <int L, int ROWS>
__device__ __forceinline__ void row_setup(RowInfoW *rows, const ChanDev *__restrict__ chans, const int32_t *__restrict__ group,
                                          int ngroup, int N, int R, long long ntrans, int mbase, int nb_call, long long first_block)
{
    const int tid = threadIdx.x;
    if (tid < ROWS) {
        const long long t = (long long)blockIdx.x * ROWS + tid;
        RowInfoW ri{0, 0, 0, 0};
        if (t < ntrans) {
            const int m = (int)(t / ngroup);
            const ChanDev ch = chans[group[(int)(t - (long long)m * ngroup)]];
            const int cnt = (int)((((first_block + mbase + m) % R) * ch.shift) % R);     // phase counter in closed form
            const int lout = L - L / R;
            ri.src = (long long)m * N + ch.f;
            ri.win = ch.win_off + cnt * L;
            ri.dst = (long long)nb_call * ch.out_off + (long long)(mbase + m) * lout - (L - lout);
            ri.valid = 1;
        }
        rows[tid] = ri;
    }
}
}  // namespace

constexpr int kW1024RowPts = 1024 + 32;          // rows of one half-wave pair on different bank halves
constexpr int kW1024Tile = 8 * kW1024RowPts * 8;
constexpr int kW512RowPts = 512 + 1;             // two rows of a 32-lane read group on complementary banks
constexpr int kW512Tile = 16 * kW512RowPts * 8;

__global__ __launch_bounds__(256, 2) void k_c1024(const float2 *__restrict__ spec, float2 *__restrict__ out,
                                                  const ChanDev *__restrict__ chans, const int32_t *__restrict__ group,
                                                  int ngroup, int N, int R, int nb_chunk, int mbase, int nb_call,
                                                  long long first_block, const float2 *__restrict__ wins,
                                                  const float2 *__restrict__ tw, int twstride /* ntab / 1024 */)
{
    __shared__ RowInfoW rows[8];
    float2 *tile = reinterpret_cast<float2 *>(fdc_smem_wide);
    float2 *wt = reinterpret_cast<float2 *>(fdc_smem_wide + kW1024Tile);                 // [b][p] = W_1024^(b p), 32 x 34
    const int tid = threadIdx.x, r = tid >> 5, b = tid & 31;
    const int skip = 1024 / R;
    row_setup<1024, 8>(rows, chans, group, ngroup, N, R, (long long)nb_chunk * ngroup, mbase, nb_call, first_block);
    for (int i = tid; i < 1024; i += 256) wt[(i >> 5) * 34 + (i & 31)] = tw[(((i >> 5) * (i & 31)) & 1023) * twstride];
    __syncthreads();
    const RowInfoW ri = rows[r];
    cf v[32];
    {   // all 64 loads of the lane (spectrum slice, window row) in flight before the first is used
        cf x[32];
#pragma unroll
        for (int a = 0; a < 32; a++) x[a] = ri.valid ? ld2(spec + ri.src + 32 * a + b) : mk(0.f, 0.f);
        cf w[32];
#pragma unroll
        for (int a = 0; a < 32; a++) w[a] = ri.valid ? ld2(wins + ri.win + 32 * a + b) : mk(0.f, 0.f);
#pragma unroll
        for (int a = 0; a < 32; a++) v[a ^ 16] = cmul(x[a], w[a]);       // ifftshift of the IFFT input: i -> i + l/2
    }
    dft32<true>(v);                                              // index p in v[pos32(p)]
    float2 *row = tile + r * kW1024RowPts;
    {
        const float2 *wr = wt + b * 34;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const float4 t = ld4(&wr[2 * i]);
            st2(&row[(2 * i) * 32 + (b ^ (2 * i))], cmulc(v[pos32(2 * i)], mk(t.x, t.y)));          // inverse: conjugate twiddles
            st2(&row[(2 * i + 1) * 32 + (b ^ (2 * i + 1))], cmulc(v[pos32(2 * i + 1)], mk(t.z, t.w)));
        }
    }
    __syncthreads();
#pragma unroll
    for (int bb = 0; bb < 32; bb++) v[bb] = ld2(&row[b * 32 + (bb ^ b)]);   // this lane now plays p = b
    dft32<true>(v);                                              // y[t = b + 32 q] in v[pos32(q)]
    if (ri.valid) {
#pragma unroll
        for (int q = 0; q < 32; q++) {
            const int tt = b + 32 * q;
            if (tt >= skip) st2_out(out + ri.dst + tt, v[pos32(q)] * 1024.f);       // overlap discard, multiply_const(l)
        }
    }
}

__global__ __launch_bounds__(256, 2) void k_c512(const float2 *__restrict__ spec, float2 *__restrict__ out,
                                                 const ChanDev *__restrict__ chans, const int32_t *__restrict__ group,
                                                 int ngroup, int N, int R, int nb_chunk, int mbase, int nb_call,
                                                 long long first_block, const float2 *__restrict__ wins,
                                                 const float2 *__restrict__ tw, int twstride /* ntab / 512 */)
{
    __shared__ RowInfoW rows[16];
    float2 *tile = reinterpret_cast<float2 *>(fdc_smem_wide);
    float2 *wt = reinterpret_cast<float2 *>(fdc_smem_wide + kW512Tile);                  // [b][p] = W_512^(b p), 16 x 34
    const int tid = threadIdx.x, r = tid >> 4, b = tid & 15;
    const int skip = 512 / R;
    row_setup<512, 16>(rows, chans, group, ngroup, N, R, (long long)nb_chunk * ngroup, mbase, nb_call, first_block);
    for (int i = tid; i < 512; i += 256) wt[(i >> 5) * 34 + (i & 31)] = tw[(((i >> 5) * (i & 31)) & 511) * twstride];
    __syncthreads();
    const RowInfoW ri = rows[r];
    cf v[32];
    {   // all 64 loads of the lane (spectrum slice, window row) in flight before the first is used
        cf x[32];
#pragma unroll
        for (int a = 0; a < 32; a++) x[a] = ri.valid ? ld2(spec + ri.src + 16 * a + b) : mk(0.f, 0.f);
        cf w[32];
#pragma unroll
        for (int a = 0; a < 32; a++) w[a] = ri.valid ? ld2(wins + ri.win + 16 * a + b) : mk(0.f, 0.f);
#pragma unroll
        for (int a = 0; a < 32; a++) v[a ^ 16] = cmul(x[a], w[a]);       // i = 16 a + b -> i + 256
    }
    dft32<true>(v);                                              // over a: index p (0..31) in v[pos32(p)]
    float2 *row = tile + r * kW512RowPts;
    {
        const float2 *wr = wt + b * 34;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const float4 t = ld4(&wr[2 * i]);
            st2(&row[(2 * i) * 16 + (b ^ ((2 * i) & 15))], cmulc(v[pos32(2 * i)], mk(t.x, t.y)));
            st2(&row[(2 * i + 1) * 16 + (b ^ ((2 * i + 1) & 15))], cmulc(v[pos32(2 * i + 1)], mk(t.z, t.w)));
        }
    }
    __syncthreads();
    // second layer: DFT-16 over b for p = lane and p = lane + 16; y[t = p + 32 q], q < 16
    cf u0[16], u1[16];
#pragma unroll
    for (int bb = 0; bb < 16; bb++) {
        u0[bb] = ld2(&row[b * 16 + (bb ^ b)]);
        u1[bb] = ld2(&row[(b + 16) * 16 + (bb ^ b)]);
    }
    dft16<true>(u0); dft16<true>(u1);
    if (ri.valid) {
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int t0 = b + 32 * q, t1 = t0 + 16;
            if (t0 >= skip) st2_out(out + ri.dst + t0, u0[rev16(q)] * 512.f);
            if (t1 >= skip) st2_out(out + ri.dst + t1, u1[rev16(q)] * 512.f);
        }
    }
}

// ---- 4096-point transform, one block per workgroup, in registers ------------------------------------------------------------
// N = 4096 = 16 x 16 x 16, n = a + 16 b + 256 c, k = k0 + 16 k1 + 256 k2.  256 threads, 16 points each, three DFT-16 layers:
//   layer 1 over c (thread = a + 16 b; the loads are 2 KiB runs), times W_256^(b k0);
//   exchange [k0][b][a] (rows 272 apart) -> thread = a + 16 k0, layer 2 over b, times W_4096^(a k0) W_256^(a k1);
//   exchange [k1][k0][a ^ k0] (rows 257 apart) -> thread = k0 + 16 k1, layer 3 over a: bins k0 + 16 k1 + 256 k2, 2 KiB runs.
// Both exchanges are conflict-free for the 16-lane store groups and the 32-lane read groups.  The twiddles are two 16 x 16
// tables in LDS (the generic core of fdc_kernels.hip reads six table entries per butterfly and pass from global memory).
// Same interface as k_fft_small (input rotation, output rotation = fftshift, scale, item stride = overlap-save gather).
constexpr int kF4096Tile = 16 * 272 * 8;                       // 34816 >= 16 * 257 * 8
template <bool INV>
__global__ __launch_bounds__(256, 4) void k_fft4096(const float2 *__restrict__ in, size_t in_stride, float2 *__restrict__ out,
                                                    int nitems, int in_rot, int out_rot, float scale,
                                                    const float2 *__restrict__ tw, int twstride /* ntab / 4096 */,
                                                    unsigned long long keep /* 64-bin groups of the output that are wanted */)
{
    float2 *tile = reinterpret_cast<float2 *>(fdc_smem_wide);
    float2 *t256 = reinterpret_cast<float2 *>(fdc_smem_wide + kF4096Tile);               // [x][y] = W_256^(x y), 16 x 18
    float2 *t4k = t256 + 16 * 18;                                                         // [x][y] = W_4096^(x y), 16 x 18
    const int tid = threadIdx.x, lo = tid & 15, hi = tid >> 4;
    const size_t m = blockIdx.x;
    t256[hi * 18 + lo] = tw[((16 * hi * lo) & 4095) * twstride];
    t4k[hi * 18 + lo] = tw[(hi * lo) * twstride];
    cf v[16];
#pragma unroll
    for (int c = 0; c < 16; c++) v[c] = ld2(in + m * in_stride + ((tid + 256 * c + in_rot) & 4095));
    __syncthreads();
    dft16<INV>(v);                                               // layer 1 over c: k0 in v[rev16(k0)]; thread = (a = lo, b = hi)
    {
        cf w[16];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float4 t = ld4(&t256[hi * 18 + 2 * i]);
            w[2 * i] = mk(t.x, t.y); w[2 * i + 1] = mk(t.z, t.w);
        }
#pragma unroll
        for (int k0 = 0; k0 < 16; k0++)
            st2(&tile[k0 * 272 + tid], k0 == 0 ? v[rev16(0)] : (INV ? cmulc(v[rev16(k0)], w[k0]) : cmul(v[rev16(k0)], w[k0])));
    }
    __syncthreads();
    // thread = (a = lo, k0 = hi): points over b
#pragma unroll
    for (int b = 0; b < 16; b++) v[b] = ld2(&tile[hi * 272 + b * 16 + lo]);
    dft16<INV>(v);                                               // layer 2 over b: k1 in v[rev16(k1)]
    __syncthreads();                                             // every read of exchange 1 is done
    {
        const cf s = ld2(&t4k[lo * 18 + hi]);                    // W_4096^(a k0)
        cf w[16];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float4 t = ld4(&t256[lo * 18 + 2 * i]);        // W_256^(a k1)
            w[2 * i] = mk(t.x, t.y); w[2 * i + 1] = mk(t.z, t.w);
        }
#pragma unroll
        for (int k1 = 0; k1 < 16; k1++) {
            const cf t = k1 == 0 ? s : cmul(s, w[k1]);
            st2(&tile[k1 * 257 + hi * 16 + (lo ^ hi)], INV ? cmulc(v[rev16(k1)], t) : cmul(v[rev16(k1)], t));
        }
    }
    __syncthreads();
    // thread = (k0 = lo, k1 = hi): points over a
#pragma unroll
    for (int a = 0; a < 16; a++) v[a] = ld2(&tile[hi * 257 + lo * 16 + (a ^ lo)]);
    dft16<INV>(v);                                               // layer 3 over a: k2 in v[rev16(k2)]
    // a wave's store is 64 consecutive bins (out_rot is a multiple of 64 whenever `keep` is not all ones): groups nobody reads stay unwritten
    const int g0 = __builtin_amdgcn_readfirstlane(((tid & ~63) + out_rot) >> 6);
#pragma unroll
    for (int k2 = 0; k2 < 16; k2++)
        if ((keep >> ((g0 + 4 * k2) & 63)) & 1ull) st2(out + m * 4096 + ((tid + 256 * k2 + out_rot) & 4095), v[rev16(k2)] * scale);
    (void)nitems;
}

hipError_t launch_fft4096(const float2 *in, size_t in_stride, float2 *out, int nitems, bool inverse, int in_rot, int out_rot,
                          float scale, const float2 *tw, int ntab, hipStream_t s, unsigned long long keep)
{
    if (out_rot & 63) keep = ~0ull;
    if (nitems <= 0) return hipSuccess;
    const int lds = kF4096Tile + 2 * 16 * 18 * 8;
    if (inverse)
        hipLaunchKernelGGL(k_fft4096<true>, dim3((unsigned)nitems), dim3(256), lds, s, in, in_stride, out, nitems, in_rot, out_rot,
                           scale, tw, ntab / 4096, keep);
    else
        hipLaunchKernelGGL(k_fft4096<false>, dim3((unsigned)nitems), dim3(256), lds, s, in, in_stride, out, nitems, in_rot, out_rot,
                           scale, tw, ntab / 4096, keep);
    return hipGetLastError();
}

hipError_t init_wide_kernels()
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_c1024), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       kW1024Tile + 32 * 34 * 8);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(k_c512), hipFuncAttributeMaxDynamicSharedMemorySize,
                               kW512Tile + 16 * 34 * 8);
}

hipError_t launch_channels_wide(const float2 *spec, float2 *out, const ChanDev *chans, const int32_t *group, int ngroup, int l,
                                int N, int R, int nb_chunk, int mbase, int nb_call, int64_t first_block, const float2 *wins,
                                const float2 *tw, int ntab, hipStream_t s)
{
    const long long ntrans = (long long)nb_chunk * ngroup;
    if (ntrans <= 0) return hipSuccess;
    if (l == 1024)
        hipLaunchKernelGGL(k_c1024, dim3((unsigned)((ntrans + 7) / 8)), dim3(256), kW1024Tile + 32 * 34 * 8, s, spec, out, chans, group,
                           ngroup, N, R, nb_chunk, mbase, nb_call, (long long)first_block, wins, tw, ntab / 1024);
    else
        hipLaunchKernelGGL(k_c512, dim3((unsigned)((ntrans + 15) / 16)), dim3(256), kW512Tile + 16 * 34 * 8, s, spec, out, chans, group,
                           ngroup, N, R, nb_chunk, mbase, nb_call, (long long)first_block, wins, tw, ntab / 512);
    return hipGetLastError();
}

}  // namespace fdc
