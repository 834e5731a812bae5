// gfx950 one-block-per-CU form of the uniform-plan path for channels of l = 128 bins (N = 65536 = 128 rows x 512 columns, R = 2, every
// channel on the 128-bin grid, one window): the machinery of fdc_block256.hip with TWO ADJACENT COLUMNS INTERLEAVED into one 256-point
// "virtual column".
//
// A 128-point column in the lane layout of fdc_block256.hip would leave half of the 16 x 16 exchange idle and need 512 columns' worth of
// passes.  Instead virtual column V carries the real columns n1 = 2V and 2V + 1 as the even and odd samples of ONE 256-point sequence
// z[2 mu + e] = x[(2V + e) + 512 mu]; the 256-point transforms of the old stage 1 run on z unchanged, and the radix-2 layers that separate and
// re-join the two columns pair the registers q and q + 8 OF THE SAME LANE (k = b + 16 q  <->  k + 128): no exchange, no extra pass.
//     forward:  Z[kap] = A0[kap] + W_256^kap A1[kap],  Z[kap + 128] = A0[kap] - W_256^kap A1[kap]           kap = 0 .. 127
//               => 2 A0 = Z[kap] + Z[kap + 128],   2 A1 = (Z[kap] - Z[kap + 128]) conj(W_256^kap)
//     product:  U_e[kap] = A_e[kap] shape[kap]/N (-1)^n1 W_N^(n1 kap), n1 = 2V + e; the ifftshift of the 128-point inverse is kap ^ 64 (q ^ 4)
//     inverse:  Z'[kap] = U0' + W_256^kap U1',  Z'[kap + 128] = U0' - W_256^kap U1'  (U' = shifted)  => IFFT256{Z'}[2 m + e] = 2 g_e[m]
//     kept:     m >= 64  <=>  t = 2 m + e >= 128: the same eight registers per lane and pass as for l = 256: G is 8 passes x 8 = 128 VGPRs.
// 256 virtual columns = 8 passes of 32: stage 1 has the cost of the l = 256 kernel plus the in-lane layers (+16 complex multiplications and
// 32 additions per lane and pass).  Stage 2 is the FFT-512 over n1 = 2V + e of every kept row m': the rows t' = 2 m' + e of the old stage 2
// (FFT-256 over V: DFT-8 over the pass index in registers, one trip through LDS, DFT-32 over c5 in the lane that owns the row) are the two
// columns' partial sums F_e, held by ADJACENT lanes, and the last radix-2 layer is an exchange with lane ^ 1 (DPP quad_perm [1, 0, 3, 2]):
//     Y[k] = F0[k] + W_512^k F1[k]  (even lane, slot k),   Y[k + 256] = F0[k] - W_512^k F1[k]  (odd lane, slot k + 256),   k = klo + 8 khi
// A wave's store is 32 consecutive samples of slot k and 32 of slot k + 256 (two 256-byte runs).
//
// All constants of stage 1 come from ONE host-built table image (fdc_api.hip: double precision, rounded once).  The arithmetic is that of
// k_p1g + k_p2g (fdc_kernels.hip) regrouped; parity against the oracle: tests/test_parity_gpu.py.
#include <hip/hip_ext.h>
#include <cmath>
#include <type_traits>
#include "fdc_kernels.h"
#include "fdc_radix16.hpp"
#include "fdc_devutil.hpp"

namespace fdc {

extern __shared__ __attribute__((aligned(16))) unsigned char fdc_smem_b128[];

typedef unsigned long long h8v __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned long long pack128(cf v) { return ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x); }
__device__ __forceinline__ cf unpack128(unsigned long long u) { return mk(__uint_as_float((unsigned)u), __uint_as_float((unsigned)(u >> 32))); }
#if defined(__HIP_DEVICE_COMPILE__)
#define FDC_PLAIN_DS128 __attribute__((target("no-load-store-opt")))
#else
#define FDC_PLAIN_DS128
#endif

// the value of the neighbouring lane (lane ^ 1): the other column of the same virtual column's row pair
__device__ __forceinline__ cf swap_neighbour(cf x)
{
    return mk(__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x.x), 0xB1, 0xF, 0xF, true)),
              __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x.y), 0xB1, 0xF, 0xF, true)));
}

// LDS map (bytes): strips and trip buffer as in fdc_block256.hip (P = 8); the tables are one image, laid out by the host in this order
constexpr int k1ScrPts = 1084;
constexpr int k1Ld = 262;                                        // stage-2 trip rows: [8 klo][32 c5] + 6 (12 dwords mod 64)
constexpr int k1Trip = 64 * k1Ld * 8;                            // 134144
constexpr int k1OffTab = 136960;
// table image, in points (float2) from k1OffTab; rows of 10 (20 dwords: the sixteen b rows of a 16-byte read cover all 64 banks)
constexpr int k1TWrow = 0;                                       // [16 b][18]       W_256^(b p)
constexpr int k1TB = k1TWrow + 16 * 18;                          // [32 c5][10]      W_N^(32 c5 q), q < 8
constexpr int k1TSA = k1TB + 32 * 10;                            // [8 pass][16 b][10]  shape[b + 16 q]/(4N) W_N^(1024 pass q)
constexpr int k1TDI = k1TSA + 8 * 16 * 10;                       // [16 b][10]       -conj(W_256^kap) W_N^kap W_256^(kap ^ 64), kap = b + 16 q
constexpr int k1TWp = k1TDI + 16 * 10;                           // [2 parity][8 klo][32 register]  parity 1: W_512^(klo + 8 khi); parity 0: 1
constexpr int k1TCt = k1TWp + 2 * 8 * 32;                        // [32 c5][8 klo]   W_256^(c5 klo)
constexpr int k1TabPts = k1TCt + 32 * 8;                         // 2816 points = 22528 B
constexpr int k1OffSoff = k1OffTab + k1TabPts * 8;               // [8 klo][2 parity][32 register] output offsets (bytes)
constexpr int k1Lds = k1OffSoff + 512 * 4;                       // 161536
static_assert(k1Trip <= k1OffTab && 8 * k1ScrPts * 8 <= k1OffTab && k1Lds <= 160 * 1024, "LDS budget");

int poly_block128_table_points() { return k1TabPts; }

// The table image (host side; N = 65536, l = 128).  shn4[kap] = shape[kap] / (4 N) (the two radix-2 layers each carry a factor 2).
void poly_block128_tables(const float *shn4 /* [128] */, float2 *img /* [poly_block128_table_points()] */)
{
    const double N = 65536.0;
    auto hrev = [](int k) { return 4 * (k & 3) + (k >> 2); };
    auto W = [](double num, double den) { const double a = -2.0 * M_PI * num / den; return make_float2(float(std::cos(a)), float(std::sin(a))); };
    for (int b = 0; b < 16; b++)
        for (int p = 0; p < 18; p++) img[k1TWrow + b * 18 + p] = p < 16 ? W(double((b * p) & 255), 256.0) : make_float2(0.f, 0.f);
    for (int c = 0; c < 32; c++)
        for (int q = 0; q < 10; q++) img[k1TB + c * 10 + q] = q < 8 ? W(double((32 * c * q) & 65535), N) : make_float2(0.f, 0.f);
    for (int ps = 0; ps < 8; ps++)
        for (int b = 0; b < 16; b++)
            for (int q = 0; q < 10; q++) {
                float2 t = make_float2(0.f, 0.f);
                if (q < 8) {
                    const double a = -2.0 * M_PI * double((1024 * ps * q) & 65535) / N, s = double(shn4[b + 16 * q]);
                    t = make_float2(float(s * std::cos(a)), float(s * std::sin(a)));
                }
                img[k1TSA + (ps * 16 + b) * 10 + q] = t;
            }
    for (int b = 0; b < 16; b++)
        for (int q = 0; q < 10; q++) {
            float2 t = make_float2(0.f, 0.f);
            if (q < 8) {
                // column e = 1: (-1)^n1 = -1, W_N^(e kap), the separation's conj(W_256^kap), and the re-join's W_256^(kap') at the shifted place kap' = kap ^ 64
                const int kap = b + 16 * q, kp = kap ^ 64;
                const double a = -2.0 * M_PI * (double(kap) / N - double(kap) / 256.0 + double(kp) / 256.0);
                t = make_float2(float(-std::cos(a)), float(-std::sin(a)));
            }
            img[k1TDI + b * 10 + q] = t;
        }
    for (int klo = 0; klo < 8; klo++)
        for (int r = 0; r < 32; r++) {
            const int khi = (r >> 4) + 2 * hrev(r & 15);                // register 16 k0 + rev16(k1) holds khi = k0 + 2 k1
            img[k1TWp + klo * 32 + r] = make_float2(1.f, 0.f);
            img[k1TWp + (8 + klo) * 32 + r] = W(double(klo + 8 * khi), 512.0);
        }
    for (int c = 0; c < 32; c++)
        for (int klo = 0; klo < 8; klo++) img[k1TCt + c * 8 + klo] = W(double((c * klo) & 255), 256.0);
}

template <bool NT>
__global__ FDC_PLAIN_DS128 __launch_bounds__(512) void k_blk128(const float2 *__restrict__ in, size_t in_stride, float2 *__restrict__ out,
                                                      const float2 *__restrict__ tab /* the table image */,
                                                      const float2 *__restrict__ cbt /* [256 V][16 b]  W_N^(2 V b) */,
                                                      const long long *__restrict__ slot_off /* [512] */, long long out_base, long long nb_call,
                                                      unsigned out_bytes, int nb, int hints)
{
    float2 *scr = reinterpret_cast<float2 *>(fdc_smem_b128);
    float2 *tbl = reinterpret_cast<float2 *>(fdc_smem_b128 + k1OffTab);
    unsigned *soff = reinterpret_cast<unsigned *>(fdc_smem_b128 + k1OffSoff);
    const int tid = threadIdx.x;
    // lane = col + 4 b: virtual column c5 = 4 wave + col of the pass, rows nu = 16 a + b of its 256-point sequence
    const int w = tid >> 6, lane = tid & 63, col = lane & 3, b = lane >> 2, c5 = 4 * w + col;

    const int grid = gridDim.x, per = grid >> 3;
    const bool xmap = (grid & 7) == 0;
    const int first = xmap ? (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (first >= nb) return;

    constexpr unsigned inbytes = 65536u * 8u;
    // z[nu = 16 a + b] = x[(2V + (b & 1)) + 512 (8 a + (b >> 1))]: a adds 4096 samples = 32 KiB, a pass 64 columns = 512 B
    const unsigned voff = (unsigned)(512 * (b >> 1) + 2 * c5 + (b & 1)) * 8u;
    const __amdgpu_buffer_rsrc_t rcb = make_rsrc(cbt, 256u * 16u * 8u);
    const unsigned voffc = (unsigned)(c5 * 16 + b) * 8u;
    cf LA[16], LB[16], cbA, cbB;
    {
        const __amdgpu_buffer_rsrc_t rin = make_rsrc(in + (size_t)first * in_stride, inbytes);
#pragma unroll
        for (int a = 0; a < 16; a++) LA[a] = bld2(rin, voff, (unsigned)a * 32768u);
        cbA = bld2(rcb, voffc, 0);
    }
    for (int i = tid; i < k1TabPts; i += 512) tbl[i] = tab[i];
    {
        // slot i < 256 = klo + 8 khi (khi = k0 + 2 k1) is entry [klo][0][16 k0 + rev16(k1)], slot i + 256 entry [klo][1][same]
        const int i = tid & 255, par = tid >> 8;
        const long long o = slot_off[i + 256 * par];
        soff[((i & 7) * 2 + par) * 32 + ((i >> 3) & 1) * 16 + rev16(i >> 4)] = o >= 0 ? (unsigned)((o * nb_call + out_base) * 8) : 0xFFFFFFFFu;
    }
    __syncthreads();

    float2 *const scrw = scr + w * k1ScrPts + lane;
    const float2 *const scrr = scr + w * k1ScrPts + col + 68 * b;
    const float2 *const wr = tbl + k1TWrow + b * 18;
    const float2 *const btr = tbl + k1TB + c5 * 10;
    const float2 *const sab = tbl + k1TSA + b * 10;                      // + pass * 160
    const float2 *const tdr = tbl + k1TDI + b * 10;
    const __amdgpu_buffer_rsrc_t rout = make_rsrc(out, out_bytes);

    for (int m = first; m < nb; m += grid) {
        const int mnext = m + grid < nb ? m + grid : m;
        h8v G[8];
        auto one_pass = [&](const int ps, cf (&cur)[16], const cf cb, cf (&L)[16], cf &cbn) __attribute__((always_inline)) {
            {
                const int pn = ps < 7 ? ps + 1 : 0;
                const int mb = ps < 7 ? m : mnext;
                const __amdgpu_buffer_rsrc_t rin = make_rsrc(in + (size_t)mb * in_stride + 64 * pn, inbytes);
                if (hints & 2) {
#pragma unroll
                    for (int a = 0; a < 16; a++) L[a] = bld2_nt(rin, voff, (unsigned)a * 32768u);
                } else {
#pragma unroll
                    for (int a = 0; a < 16; a++) L[a] = bld2(rin, voff, (unsigned)a * 32768u);
                }
                cbn = bld2(rcb, voffc, (unsigned)pn * 4096u);
            }
            // ---- the 256-point forward transform of the virtual column: exactly the old stage 1
            dft16<false>(cur);
            {
                cf tw[16];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const float4 t = ld4(&wr[2 * i]);
                    tw[2 * i] = mk(t.x, t.y); tw[2 * i + 1] = mk(t.z, t.w);
                }
                st2(&scrw[0], cur[rev16(0)]);
#pragma unroll
                for (int p = 1; p < 16; p++) st2(&scrw[68 * p], cmul(cur[rev16(p)], tw[p]));
            }
            __builtin_amdgcn_wave_barrier();
            cf v[16];
#pragma unroll
            for (int bb = 0; bb < 16; bb++) v[bb] = ld2(&scrr[4 * bb]);
            dft16<false>(v);                                      // Z[k = b + 16 q] in v[rev16(q)]
            // ---- separate the two columns (registers q, q + 8), product, shift, re-join
            cf u[16];
            {
                const float2 *sar = sab + ps * 160;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float4 t0 = ld4(&btr[2 * i]), t1 = ld4(&sar[2 * i]), t2 = ld4(&tdr[2 * i]);
                    const cf bt[2] = {mk(t0.x, t0.y), mk(t0.z, t0.w)}, sa[2] = {mk(t1.x, t1.y), mk(t1.z, t1.w)}, td[2] = {mk(t2.x, t2.y), mk(t2.z, t2.w)};
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        const int q = 2 * i + e;
                        const cf z0 = v[rev16(q)], z1 = v[rev16(q + 8)];
                        const cf f = cmul(bt[e], sa[e]);                      // shape/(4N) W_N^(2V 16 q) without the lane's W_N^(2V b) (cb, below)
                        const cf u0 = cmul(z0 + z1, f);                       // column 2V
                        const cf x1 = cmul(cmul(z0 - z1, td[e]), f);          // column 2V + 1, with the re-join's twiddle
                        u[q ^ 4] = u0 + x1;
                        u[(q ^ 4) + 8] = u0 - x1;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // ---- the 256-point inverse transform of the virtual column (no q ^ 8: the shift was inside the 128-point halves)
            dft16<true>(u);
            {
                cf tw[16];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const float4 t = ld4(&wr[2 * i]);
                    tw[2 * i] = mk(t.x, t.y); tw[2 * i + 1] = mk(t.z, t.w);
                }
                u[rev16(0)] = cmul(u[rev16(0)], cb);
#pragma unroll
                for (int p = 1; p < 16; p++) u[rev16(p)] = cmul(cmulc(u[rev16(p)], tw[p]), cb);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int p = 0; p < 16; p++) st2(&scrw[68 * p], u[rev16(p)]);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int bb = 0; bb < 16; bb++) u[bb] = ld2(&scrr[4 * bb]);
            dft16<true>(u);                                       // y[t = b + 16 q] in u[rev16(q)]; keep q >= 8
#pragma unroll
            for (int j = 0; j < 8; j++) G[j][ps] = pack128(u[rev16(8 + j)]);
        };
#pragma nounroll
        for (int pp = 0; pp < 8; pp += 2) {
            one_pass(pp, LA, cbA, LB, cbB);
            one_pass(pp + 1, LB, cbB, LA, cbA);
        }
        // ---------------- stage 2: FFT-512 over n1 = 2 (32 pass + c5) + e of every row t' = b + 16 j = 2 m' + e ----------------
        {
            __syncthreads();                                          // every wave is done with its strip
            int t2 = tid;
            asm volatile("" : "+v"(t2));
            const int lane2 = t2 & 63, w2 = __builtin_amdgcn_readfirstlane(t2 >> 6), b_2 = lane2 >> 2, c5_2 = 4 * w2 + (lane2 & 3), par2 = lane2 & 1;
            float2 *const gw0 = scr + b_2 * k1Ld + c5_2;              // element (row b + 16 jj, klo) at + 16 jj kLd + 32 klo
            int rowjb = 32 * k1Ld;
            asm volatile("" : "+v"(rowjb));
            float2 *const gw1 = gw0 + rowjb;
            const float2 *const gr = scr + lane2 * k1Ld + 32 * w2;    // row = lane, klo = wave: 32 consecutive points
            const float2 *const wpr = tbl + k1TWp + (8 * par2 + w2) * 32;
            const uint4 *const sow = reinterpret_cast<const uint4 *>(soff + (2 * w2 + par2) * 32);
            const float fsgn = par2 ? -1.0f : 1.0f;
#pragma unroll
            for (int tr = 0; tr < 2; tr++) {
                cf ct[8];
                {
                    const float2 *ctr = tbl + k1TCt + c5_2 * 8;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const float4 t = ld4(&ctr[2 * i]);
                        ct[2 * i] = mk(t.x, t.y); ct[2 * i + 1] = mk(t.z, t.w);
                    }
                }
#pragma unroll
                for (int jj = 0; jj < 4; jj++) {
                    cf a[8];
#pragma unroll
                    for (int ps = 0; ps < 8; ps++) a[ps] = unpack128(G[4 * tr + jj][ps]);
                    dft8<false>(a);                                   // klo = k0 + 2 k1 in a[4 k0 + k1]
                    float2 *const gw = (jj < 2 ? gw0 : gw1) + (jj & 1) * 16 * k1Ld;
                    st2(&gw[0], a[0]);
#pragma unroll
                    for (int k = 1; k < 8; k++) st2(&gw[32 * k], cmul(a[4 * (k & 1) + (k >> 1)], ct[k]));
                }
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();                                      // the trip is in LDS
                cf v[32];
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const float4 t = ld4(&gr[2 * i]);
                    v[2 * i] = mk(t.x, t.y); v[2 * i + 1] = mk(t.z, t.w);
                }
                __syncthreads();                                      // every read of the trip is done
                __builtin_amdgcn_sched_barrier(0);
                dft32<false>(v);                                      // F_e[klo + 8 khi], khi = k0 + 2 k1, in v[16 k0 + rev16(k1)]
                // the last radix-2 layer over e: the neighbour lane holds the other column's sum (table row of ones for the even lanes)
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const float4 t = ld4(&wpr[2 * i]);
                    const cf x0 = cmul(v[2 * i], mk(t.x, t.y)), x1 = cmul(v[2 * i + 1], mk(t.z, t.w));
                    v[2 * i] = swap_neighbour(x0) + x0 * fsgn;        // even lane: F0 + W F1 (slot k); odd lane: F0 - W F1 (slot k + 256)
                    v[2 * i + 1] = swap_neighbour(x1) + x1 * fsgn;
                    __builtin_amdgcn_sched_barrier(0);                // one table read at a time: the phase has no registers for more
                }
                // row m' = (64 tr + lane) / 2 of slot k (even lanes) or k + 256 (odd lanes)
                const unsigned rb = (unsigned)(m * 64 + 32 * tr + (lane2 >> 1)) * 8u;
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const uint4 t = sow[q];
                    const unsigned so[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                    for (int e = 0; e < 4; e++) bst2t<NT>(rout, (so[e] == 0xFFFFFFFFu ? 0xFFFFFFF0u : so[e] + rb), v[4 * q + e]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the trip region (= the strips) was last read before the barrier above: the next block starts without one
    }
}

hipError_t init_block128_kernels()
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_blk128<true>), hipFuncAttributeMaxDynamicSharedMemorySize, k1Lds);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_blk128<false>), hipFuncAttributeMaxDynamicSharedMemorySize, k1Lds);
    return e;
}

hipError_t launch_poly_block128(const float2 *in, size_t in_stride, float2 *out, int nb_chunk, int mbase, int nb_call, const float2 *tab,
                                const float2 *cbt, const long long *slot_off, unsigned out_bytes, int ncu, int hints, hipStream_t s,
                                hipEvent_t ev_start, hipEvent_t ev_stop)
{
    if (nb_chunk <= 0) return hipSuccess;
    int grid = ncu > 0 ? ncu : 256;
    if (grid > nb_chunk) grid = nb_chunk;
#define FDC_L128(A) \
    hipExtLaunchKernelGGL((k_blk128<A>), dim3((unsigned)grid), dim3(512), k1Lds, s, ev_start, ev_stop, 0u, in, in_stride, out, tab, cbt, slot_off, \
                          (long long)mbase * 64, (long long)nb_call, out_bytes, nb_chunk, hints)
    if (hints & 1) FDC_L128(true); else FDC_L128(false);
#undef FDC_L128
    return hipGetLastError();
}

}  // namespace fdc
