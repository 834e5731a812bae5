// gfx950 one-block-per-CU form of the uniform-plan path for NARROW channels: l = 128 or 64 bins (N = 65536 = l rows x 65536/l columns,
// R = 2, every channel on the l-bin grid, one window): the machinery of fdc_block256.hip with S = 256/l ADJACENT COLUMNS INTERLEAVED into
// one 256-point "virtual column".
//
// An l-point column in the lane layout of fdc_block256.hip would leave most of the 16 x 16 exchange idle and need S times the passes.
// Instead virtual column V carries the real columns n1 = S V + e, e < S, as the samples nu = S mu + e of ONE 256-point sequence
// z[S mu + e] = x[(S V + e) + N1 mu]; the 256-point transforms of the old stage 1 run on z unchanged, and the radix-S layers that separate
// and re-join the columns combine the registers q0, q0 + 16/S, ... OF THE SAME LANE (k = b + 16 q  <->  k + l): no exchange, no extra pass.
//     forward:  Z[kap + l i] = sum_e W_S^(i e) (W_256^(kap e) A_e[kap])           kap < l, i < S
//               => S W_256^(kap e) A_e[kap] = sum_i W_S^(-i e) Z[kap + l i]         (an inverse DFT-S over the registers)
//     product:  U_e[kap] = A_e[kap] shape[kap]/N (-1)^n1 W_N^(n1 kap), n1 = S V + e; the ifftshift of the l-point inverse is kap ^ l/2
//     inverse:  Z'[kap + l i] = sum_e W_S^(i e) (W_256^(kap e) U_e'[kap])  (U' = shifted)  => IFFT256{Z'}[S m + e] = S g_e[m]
//     kept:     m >= l/2  <=>  t = S m + e >= 128: the same eight registers per lane and pass as for l = 256: G is 8 passes x 8 = 128 VGPRs.
// 256 virtual columns = 8 passes of 32: stage 1 has the cost of the l = 256 kernel plus the in-lane layers.  Stage 2 is the FFT over
// n1 = S V + e (256 S slots) of every kept row m': the rows t' = S m' + e of the old stage 2 (FFT-256 over V: DFT-8 over the pass index in
// registers, one trip through LDS, DFT-32 over c5 in the lane that owns the row) are the S columns' partial sums F_e, held by S ADJACENT
// lanes, and the last radix-S layer is a DPP exchange inside the quad:
//     Y[k + 256 i] = sum_e W_S^(i e) (W_{256 S}^(k e) F_e[k]),   k = klo + 8 khi
// A wave's store is 64/S consecutive samples of each of S slots (S = 2: two 256-byte runs; S = 4: four 128-byte runs).
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

extern __shared__ __attribute__((aligned(16))) unsigned char fdc_smem_nar[];

__device__ __forceinline__ unsigned long long pack_nar(cf v) { return ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x); }
__device__ __forceinline__ cf unpack_nar(unsigned long long u) { return mk(__uint_as_float((unsigned)u), __uint_as_float((unsigned)(u >> 32))); }
#if defined(__HIP_DEVICE_COMPILE__)
#define FDC_PLAIN_DSN __attribute__((target("no-load-store-opt")))
#else
#define FDC_PLAIN_DSN
#endif

// the value of lane ^ 1 / lane ^ 2 of the quad
__device__ __forceinline__ cf quad_xor1(cf x)
{
    return mk(__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x.x), 0xB1, 0xF, 0xF, true)),
              __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x.y), 0xB1, 0xF, 0xF, true)));
}
__device__ __forceinline__ cf quad_xor2(cf x)
{
    return mk(__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x.x), 0x4E, 0xF, 0xF, true)),
              __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x.y), 0x4E, 0xF, 0xF, true)));
}

// LDS map (bytes): strips and trip buffer as in fdc_block256.hip; the tables are one image, laid out by the host in the order below.
// S = columns per virtual column (2: l = 128, 4: l = 64); QG = 16/S registers q0 per lane carry kap = b + 16 q0 < l.
// P = passes of 32 virtual columns (round 5): N = 256 rows x 32 P virtual columns = 8192 P (P = 8: 65536; 4: 32768; 2: 16384), N1 = 32 P S slots.
// Stage 2 is the FFT-32P over V: DFT-P over the pass index in registers, one trip through LDS, DFT-32 over c5.  A trip holds up to 8 / P blocks of 64 rows
// ([rows][P klo][32 c5]): a wave reads (klo = wave mod P, block = wave div P); a run of fewer blocks than a trip holds (R = 2: 128 kept rows = two blocks;
// the rows that come back from the scratch at R = 4: one) leaves the other waves without a row for the DFT-32.
constexpr int kNarScrPts = 1084;
template <int S, int P = 8>
struct NarGeom {
    static_assert(S == 2 || S == 4, "two or four columns per virtual column");
    static_assert(P == 2 || P == 4 || P == 8, "passes of 32 virtual columns: N = 16384, 32768 or 65536");
    static constexpr int kL = 256 / S;                           // channel width
    static constexpr int kNV = 32 * P;                           // virtual columns
    static constexpr int kN1 = kNV * S;                          // columns = channel slots
    static constexpr int kN = 256 * kNV;
    static constexpr int kLout = kL / 2;                         // kept samples per block and channel (R = 2)
    static constexpr int kQG = 16 / S;
    static constexpr int kLd = 32 * P + 6;                       // stage-2 trip rows: [P klo][32 c5] + 6 (12 dwords mod 64)
    static constexpr int kTripBlocks = (8 / P) < 2 ? (8 / P) : 2;   // 64-row blocks a trip ever holds (a run has two at most)
    static constexpr int kTrip = 64 * kTripBlocks * kLd * 8;     // P = 8: 134144; 4: 137216; 2: 71680
    static constexpr int kStrips = 8 * kNarScrPts * 8;
    static constexpr int kOffTab = P == 8 ? 136960 : (kTrip > kStrips ? kTrip : kStrips);
    // table rows read 16 bytes at a time by the sixteen b rows of a wave: row strides of 20 (S = 2), 12 and 28 (S = 4) dwords put the sixteen
    // reads on sixteen different bank quartets
    static constexpr int kRowQ = kQG + 2;                        // Bt, SA rows: 10 / 6 points
    static constexpr int kRowT = (S - 1) * kQG + 2;              // TD rows: 10 / 14 points
    // table image, in points (float2) from kOffTab
    static constexpr int kTWrow = 0;                             // [16 b][18]            W_256^(b p)
    static constexpr int kTB = kTWrow + 16 * 18;                 // [32 c5][kRowQ]        W_N^(16 S c5 q0)
    static constexpr int kTSA = kTB + 32 * kRowQ;                // [P pass][16 b][kRowQ] shape[b + 16 q0]/(S^2 N) W_N^(512 S pass q0)
    static constexpr int kTTD = kTSA + P * 16 * kRowQ;           // [16 b][e - 1][q0]     (-1)^e W_N^(e kap) conj(W_256^(e kap)) W_256^(e (kap ^ l/2)), kap = b + 16 q0
    static constexpr int kTWp = kTTD + 16 * kRowT;               // [S e][P klo][32 register]  W_{32 P S}^(e (klo + P khi))
    static constexpr int kTCt = kTWp + S * P * 32;               // [32 c5][P klo]        W_{32 P}^(c5 klo)
    static constexpr int kTabPts = kTCt + 32 * P;
    static constexpr int kOffSoff = kOffTab + kTabPts * 8;       // [P klo][S lane][32 register] output offsets (bytes)
    static constexpr int kLds = kOffSoff + kN1 * 4;              // P = 8: S = 2: 161536; S = 4: 163072
    static_assert(kTrip <= kOffTab && kStrips <= kOffTab && kOffTab % 16 == 0 && kLds <= 160 * 1024, "LDS budget");
    static_assert((kLd * 2) % 64 == 12, "trip rows 12 dwords apart mod 64");
};

bool poly_block_narrow_supports(int N, int L, int R) { return (N == 65536 || N == 32768 || N == 16384) && (R == 2 || R == 4) && (L == 128 || L == 64); }
int poly_block_narrow_table_points(int L, int N)
{
    const bool two = L == 128;
    switch (N) {
    case 65536: return two ? NarGeom<2, 8>::kTabPts : NarGeom<4, 8>::kTabPts;
    case 32768: return two ? NarGeom<2, 4>::kTabPts : NarGeom<4, 4>::kTabPts;
    default: return two ? NarGeom<2, 2>::kTabPts : NarGeom<4, 2>::kTabPts;
    }
}

// The table image (host side).  shn[kap] = shape[kap] / N (l values); the factor 1/S^2 of the two radix-S layers is applied here.
// half: the bank sits half a channel higher (f = l slot + l/2: channels centred on multiples of l).  That is the on-grid plan of the block
// modulated by exp(-2 pi i (l/2) n / N) = W_N^((l/2) n1) (-1)^n2: the (-1)^n2 moves every column's spectrum by l/2 bins, which together with the
// ifftshift of the inverse is the identity — the data stays in its registers and only the tables move (entries of kap ^ l/2), the separation's
// and the re-join's twiddles cancel, and the per-column constant W_N^((l/2) n1) goes into TD (its e part) and cbt (its V part, fdc_api.hip).
template <int S, int P>
static void narrow_tables(const float *shn, float2 *img, bool half, int r)
{
    typedef NarGeom<S, P> GM;
    const double N = double(GM::kN);
    const long long Ni = GM::kN;
    auto hrev = [](int k) { return 4 * (k & 3) + (k >> 2); };
    auto W = [](double num, double den) { const double a = -2.0 * M_PI * num / den; return make_float2(float(std::cos(a)), float(std::sin(a))); };
    for (int i = 0; i < GM::kTabPts; i++) img[i] = make_float2(0.f, 0.f);
    for (int b = 0; b < 16; b++)
        for (int p = 0; p < 16; p++) img[GM::kTWrow + b * 18 + p] = W(double((b * p) & 255), 256.0);
    for (int c = 0; c < 32; c++)
        for (int q = 0; q < GM::kQG; q++) {
            const int qt = half ? q ^ (GM::kQG / 2) : q;                 // the table entry of kap ^ l/2
            img[GM::kTB + c * GM::kRowQ + q] = W(double((16ll * S * c * qt) % Ni), N);
        }
    for (int ps = 0; ps < P; ps++)
        for (int b = 0; b < 16; b++)
            for (int q = 0; q < GM::kQG; q++) {
                const int qt = half ? q ^ (GM::kQG / 2) : q;
                const double a = -2.0 * M_PI * double((512ll * S * ps * qt) % Ni) / N, s = double(shn[b + 16 * qt]) / double(S * S);
                img[GM::kTSA + (ps * 16 + b) * GM::kRowQ + q] = make_float2(float(s * std::cos(a)), float(s * std::sin(a)));
            }
    for (int b = 0; b < 16; b++)
        for (int e = 1; e < S; e++)
            for (int q = 0; q < GM::kQG; q++) {
                // column e: (-1)^n1 = (-1)^e, W_N^(e kap), the separation's conj(W_256^(e kap)) and the re-join's W_256^(e kap') at the shifted place
                const int kap = b + 16 * q, kp = kap ^ (GM::kL / 2);
                // on the grid: W_N^(e kap), separation at kap, re-join at kap'; half: W_N^(e kap') W_N^(e l/2), separation and re-join both at kap
                // r (a quarter or three quarters of a channel; not with half): column e's constant W_N^(r e) conj(W_256^(r e)) with it
                const double a = half ? -2.0 * M_PI * double(e) * (double(kp) + double(GM::kL / 2)) / N
                                      : -2.0 * M_PI * double(e) * ((double(kap) + double(r)) / N - (double(kap) + double(r)) / 256.0 + double(kp) / 256.0);
                const double sg = (e & 1) ? -1.0 : 1.0;
                img[GM::kTTD + b * GM::kRowT + (e - 1) * GM::kQG + q] = make_float2(float(sg * std::cos(a)), float(sg * std::sin(a)));
            }
    for (int e = 0; e < S; e++)
        for (int klo = 0; klo < P; klo++)
            for (int r = 0; r < 32; r++) {
                const int khi = (r >> 4) + 2 * hrev(r & 15);             // register 16 k0 + rev16(k1) holds khi = k0 + 2 k1
                img[GM::kTWp + (e * P + klo) * 32 + r] = W(double(e * (klo + P * khi)), double(GM::kNV * S));
            }
    for (int c = 0; c < 32; c++)
        for (int klo = 0; klo < P; klo++) img[GM::kTCt + c * P + klo] = W(double((c * klo) % GM::kNV), double(GM::kNV));
}
void poly_block_narrow_tables(int L, int N, const float *shn, float2 *img, bool half, int r)
{
    const bool two = L == 128;
    switch (N) {
    case 65536: if (two) narrow_tables<2, 8>(shn, img, half, r); else narrow_tables<4, 8>(shn, img, half, r); break;
    case 32768: if (two) narrow_tables<2, 4>(shn, img, half, r); else narrow_tables<4, 4>(shn, img, half, r); break;
    default: if (two) narrow_tables<2, 2>(shn, img, half, r); else narrow_tables<4, 2>(shn, img, half, r); break;
    }
}

template <int P> __device__ __forceinline__ constexpr int nar_pass_idx(int k) { return P == 8 ? 4 * (k & 1) + (k >> 1) : k; }
template <int P>
__device__ __forceinline__ void nar_pass_dft(cf (&a)[P])
{
    if constexpr (P == 8) dft8<false>(a);                          // klo = k0 + 2 k1 in a[4 k0 + k1]
    else if constexpr (P == 4) dft4<false>(a[0], a[1], a[2], a[3]);
    else { const cf s0 = a[0] + a[1], d0 = a[0] - a[1]; a[0] = s0; a[1] = d0; }
}

// R4 = true: relinvovl = 4 (the reference's default overlap): three quarters of every inverse transform are kept.  The rows t >= 128 of the
// virtual column stay in the G registers as for R = 2; the rows 64 <= t < 128 go to 128 KiB of per-workgroup scratch ([pass][q - 4][thread]: the
// L2 holds it) and come back for a second, 64-row run of stage 2 (the first 64/S output rows of the block), as in fdc_block256.hip.
// HALF = true: the bank half a channel higher (tables: narrow_tables(half)); in the kernel only the place the re-joined values go to changes.
// ROT != 0: the bank a QUARTER of a channel off the grid (f = l slot + r, r = l/4 or 3l/4: doubled slices centred between the raster points).  The block
// modulated by exp(-2 pi i r n / N): on the virtual column that is W_256^(r nu) — its 256-point spectrum moved by r bins, i.e. by ROT = r/16 REGISTERS of the
// same lane — times one constant per column: W_N^(r e) conj(W_256^(r e)) (in TD, host) and W_N^(r S V) (in cbt).  The kernel differs in which registers the
// separation reads.
template <int S, bool NT, bool R4, bool HALF, int ROT = 0, int P = 8>
__global__ FDC_PLAIN_DSN __launch_bounds__(512) void k_blknar(const float2 *__restrict__ in, size_t in_stride, float2 *__restrict__ out,
                                                    const float2 *__restrict__ tab /* the table image */,
                                                    const float2 *__restrict__ cbt /* [32 P V][16 b]  W_N^(S V b) */,
                                                    const long long *__restrict__ slot_off /* [32 P S] */, long long out_base, long long nb_call,
                                                    unsigned out_bytes, int nb, int hints, float2 *__restrict__ scratch)
{
    typedef NarGeom<S, P> GM;
    static_assert(!(HALF && ROT), "half a channel off the grid is its own form");
    constexpr int kNarLd = GM::kLd;
    constexpr int kRows = R4 ? 3 * GM::kL / 4 : GM::kL / 2;        // kept samples per block and channel: 3 l / 4 or l / 2
    constexpr int kQG = GM::kQG, kLS = S == 2 ? 1 : 2;
    float2 *scr = reinterpret_cast<float2 *>(fdc_smem_nar);
    float2 *tbl = reinterpret_cast<float2 *>(fdc_smem_nar + GM::kOffTab);
    unsigned *soff = reinterpret_cast<unsigned *>(fdc_smem_nar + GM::kOffSoff);
    const int tid = threadIdx.x;
    // lane = col + 4 b: virtual column c5 = 4 wave + col of the pass, rows nu = 16 a + b of its 256-point sequence
    const int w = tid >> 6, lane = tid & 63, col = lane & 3, b = lane >> 2, c5 = 4 * w + col;

    const int grid = gridDim.x, per = grid >> 3;
    const bool xmap = (grid & 7) == 0;
    const int first = xmap ? (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (first >= nb) return;

    constexpr unsigned inbytes = (unsigned)GM::kN * 8u;
    constexpr unsigned kAStep = 16u * (unsigned)GM::kNV * 8u;      // a adds 16 NV samples (P = 8: 4096 = 32 KiB)
    // z[nu = 16 a + b] = x[(S V + (b mod S)) + N1 ((16/S) a + b div S)]: a adds 16 NV samples, a pass 32 S columns
    const unsigned voff = (unsigned)(GM::kN1 * (b >> kLS) + S * c5 + (b & (S - 1))) * 8u;
    const __amdgpu_buffer_rsrc_t rcb = make_rsrc(cbt, (unsigned)GM::kNV * 16u * 8u);
    const unsigned voffc = (unsigned)(c5 * 16 + b) * 8u;
    cf LA[16], LB[16], cbA, cbB;
    {
        const __amdgpu_buffer_rsrc_t rin = make_rsrc(in + (size_t)first * in_stride, inbytes);
#pragma unroll
        for (int a = 0; a < 16; a++) LA[a] = bld2(rin, voff, (unsigned)a * kAStep);
        cbA = bld2(rcb, voffc, 0);
    }
    for (int i = tid; i < GM::kTabPts; i += 512) tbl[i] = tab[i];
    for (int i = tid; i < GM::kN1; i += 512) {
        // slot k + 32 P i2 (k = klo + P khi, khi = k0 + 2 k1) is produced by the quad lane lam(i2) in register 16 k0 + rev16(k1):
        // entry [klo][lam][register]; S = 2: lam = i2; S = 4: lam = the two bits of i2 swapped (the order the DPP layer leaves)
        const int k = i % GM::kNV, i2 = i / GM::kNV, lam = S == 2 ? i2 : ((i2 & 1) << 1 | (i2 >> 1));
        const int klo = k % P, khi = k / P;
        const long long o = slot_off[i];
        soff[(klo * S + lam) * 32 + (khi & 1) * 16 + rev16(khi >> 1)] = o >= 0 ? (unsigned)((o * nb_call + out_base) * 8) : 0xFFFFFFFFu;
    }
    __syncthreads();

    float2 *const scrw = scr + w * kNarScrPts + lane;
    const float2 *const scrr = scr + w * kNarScrPts + col + 68 * b;
    const float2 *const wr = tbl + GM::kTWrow + b * 18;
    const float2 *const btr = tbl + GM::kTB + c5 * GM::kRowQ;
    const float2 *const sab = tbl + GM::kTSA + b * GM::kRowQ;             // + pass * 16 rows
    const float2 *const tdr = tbl + GM::kTTD + b * GM::kRowT;
    const __amdgpu_buffer_rsrc_t rout = make_rsrc(out, out_bytes);
    const __amdgpu_buffer_rsrc_t rscr = make_rsrc(R4 ? scratch + (size_t)blockIdx.x * 16384 : scratch, R4 ? 16384u * 8u : 0u);

    for (int m = first; m < nb; m += grid) {
        const int mnext = m + grid < nb ? m + grid : m;
        typedef unsigned long long gvec __attribute__((ext_vector_type(P)));
        gvec G[8];
        auto one_pass = [&](const int ps, cf (&cur)[16], const cf cb, cf (&L)[16], cf &cbn) __attribute__((always_inline)) {
            {
                const int pn = ps < P - 1 ? ps + 1 : 0;
                const int mb = ps < P - 1 ? m : mnext;
                const __amdgpu_buffer_rsrc_t rin = make_rsrc(in + (size_t)mb * in_stride + 32 * S * pn, inbytes);
                if (hints & 2) {
#pragma unroll
                    for (int a = 0; a < 16; a++) L[a] = bld2_nt(rin, voff, (unsigned)a * kAStep);
                } else {
#pragma unroll
                    for (int a = 0; a < 16; a++) L[a] = bld2(rin, voff, (unsigned)a * kAStep);
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
            // ---- separate the S columns (registers q0 + (16/S) i), product, shift, re-join
            cf u[16];
            {
                const float2 *sar = sab + ps * 16 * GM::kRowQ;
#pragma unroll
                for (int g = 0; g < kQG / 2; g++) {
                    const float4 t0 = ld4(&btr[2 * g]), t1 = ld4(&sar[2 * g]);
                    const cf bt[2] = {mk(t0.x, t0.y), mk(t0.z, t0.w)}, sa[2] = {mk(t1.x, t1.y), mk(t1.z, t1.w)};
                    cf td[S - 1][2];
#pragma unroll
                    for (int e = 1; e < S; e++) {
                        const float4 t2 = ld4(&tdr[(e - 1) * kQG + 2 * g]);
                        td[e - 1][0] = mk(t2.x, t2.y); td[e - 1][1] = mk(t2.z, t2.w);
                    }
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const int q = 2 * g + h, qs = HALF ? q : q ^ (kQG / 2);   // the shifted place (HALF: the two shifts cancel)
                        const cf f = cmul(bt[h], sa[h]);                      // shape/(S^2 N) W_N^(S V 16 q) without the lane's W_N^(S V b) (cb, below)
                        if constexpr (S == 2) {
                            const cf z0 = v[rev16((q + ROT) & 15)], z1 = v[rev16((q + 8 + ROT) & 15)];
                            const cf u0 = cmul(z0 + z1, f);                   // column 2V
                            const cf x1 = cmul(cmul(z0 - z1, td[0][h]), f);   // column 2V + 1, with the re-join's twiddle
                            u[qs] = u0 + x1;
                            u[qs + 8] = u0 - x1;
                        } else {
                            cf z0 = v[rev16((q + ROT) & 15)], z1 = v[rev16((q + 4 + ROT) & 15)], z2 = v[rev16((q + 8 + ROT) & 15)], z3 = v[rev16((q + 12 + ROT) & 15)];
                            dft4<true>(z0, z1, z2, z3);                       // 4 W_256^(kap e) A_e[kap], e = 0 .. 3
                            z0 = cmul(z0, f);
                            z1 = cmul(cmul(z1, td[0][h]), f);
                            z2 = cmul(cmul(z2, td[1][h]), f);
                            z3 = cmul(cmul(z3, td[2][h]), f);
                            dft4<false>(z0, z1, z2, z3);                      // Z'[kap' + 64 i], i = 0 .. 3
                            u[qs] = z0; u[qs + 4] = z1; u[qs + 8] = z2; u[qs + 12] = z3;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // ---- the 256-point inverse transform of the virtual column (no q ^ 8: the shift was inside the l-point parts)
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
            for (int j = 0; j < 8; j++) G[j][ps] = pack_nar(u[rev16(8 + j)]);
            if constexpr (R4) {                                   // R = 4 keeps q >= 4: rows 64..127 go to the scratch, [pass][q - 4][thread]
#pragma unroll
                for (int j = 0; j < 4; j++) bst2(rscr, (unsigned)tid * 8u + (unsigned)j * 4096u, (unsigned)ps * 16384u, u[rev16(4 + j)]);
            }
        };
#pragma nounroll
        for (int pp = 0; pp < P; pp += 2) {
            one_pass(pp, LA, cbA, LB, cbB);
            one_pass(pp + 1, LB, cbB, LA, cbA);
        }
        // ---------------- stage 2: the FFT over n1 = S (32 pass + c5) + e of every row t' = b + 16 j = S m' + e ----------------
        // get(j, pass): the value of row group j; rowbase: first output row of the run; nblkc: its number of 64-row blocks (four j each)
        auto stage2 = [&](auto get, const int rowbase, auto nblkc) __attribute__((always_inline)) {
            constexpr int kNBlk = decltype(nblkc)::value, kTB = 8 / P;            // blocks of the run; blocks a trip has waves for
            constexpr int kNTrip = (kNBlk + kTB - 1) / kTB;
            __syncthreads();                                          // every wave is done with its strip / the previous run's trip
            int t2 = tid;
            asm volatile("" : "+v"(t2));
            const int lane2 = t2 & 63, w2 = __builtin_amdgcn_readfirstlane(t2 >> 6), b_2 = lane2 >> 2, c5_2 = 4 * w2 + (lane2 & 3);
            const int klo2 = P == 8 ? w2 : w2 % P, rh2 = P == 8 ? 0 : w2 / P;   // reader: klo, block of the trip
            constexpr int kPairs = 2 * (kNBlk < kTB ? kNBlk : kTB);   // pairs of row groups in a trip
            float2 *gwb[kPairs];                                      // element (row b + 16 jj, klo) at gwb[jj / 2] + 16 (jj mod 2) kLd + 32 klo
            gwb[0] = scr + b_2 * kNarLd + c5_2;
            int rowjb = 32 * kNarLd;
            asm volatile("" : "+v"(rowjb));
#pragma unroll
            for (int i = 1; i < kPairs; i++) gwb[i] = gwb[i - 1] + rowjb;
            const float2 *const gr = scr + (64 * rh2 + lane2) * kNarLd + 32 * klo2;  // row = 64 block + lane, klo: 32 consecutive points
#pragma unroll
            for (int tr = 0; tr < kNTrip; tr++) {
                const int blk = kNBlk - kTB * tr < kTB ? kNBlk - kTB * tr : kTB;   // blocks in this trip (a constant once unrolled)
                cf ct[P];
                {
                    const float2 *ctr = tbl + GM::kTCt + c5_2 * P;
#pragma unroll
                    for (int i = 0; i < P / 2; i++) {
                        const float4 t = ld4(&ctr[2 * i]);
                        ct[2 * i] = mk(t.x, t.y); ct[2 * i + 1] = mk(t.z, t.w);
                    }
                }
                cf src[4 * (kNBlk < kTB ? kNBlk : kTB)][P];           // (runs that come back from the scratch: all loads in flight at once)
#pragma unroll
                for (int jj = 0; jj < 4 * blk; jj++)
#pragma unroll
                    for (int ps = 0; ps < P; ps++) src[jj][ps] = get(4 * kTB * tr + jj, ps);
#pragma unroll
                for (int jj = 0; jj < 4 * blk; jj++) {
                    cf a[P];
#pragma unroll
                    for (int ps = 0; ps < P; ps++) a[ps] = src[jj][ps];
                    nar_pass_dft<P>(a);
                    float2 *const gw = gwb[jj >> 1] + (jj & 1) * 16 * kNarLd;
                    st2(&gw[0], a[0]);
#pragma unroll
                    for (int k = 1; k < P; k++) st2(&gw[32 * k], cmul(a[nar_pass_idx<P>(k)], ct[k]));
                }
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();                                      // the trip is in LDS
                const bool active = kTB == 1 || rh2 < blk;            // wave-uniform: a run shorter than a trip leaves waves without a block
                cf v[32];
                if (active) {
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const float4 t = ld4(&gr[2 * i]);
                        v[2 * i] = mk(t.x, t.y); v[2 * i + 1] = mk(t.z, t.w);
                    }
                }
                __syncthreads();                                      // every read of the trip is done
                __builtin_amdgcn_sched_barrier(0);
                if (active) {
                    dft32<false>(v);                                  // F_e[klo + P khi], khi = k0 + 2 k1, in v[16 k0 + rev16(k1)]
                    // the lane's place in its quad and what hangs on it are worked out here, behind the DFT-32, which has no registers to carry them
                    int t3 = tid;
                    asm volatile("" : "+v"(t3));
                    const int lam = t3 & (S - 1);
                    const float2 *const wpr = tbl + GM::kTWp + (P * lam + klo2) * 32;
                    const uint4 *const sow = reinterpret_cast<const uint4 *>(soff + (S * klo2 + lam) * 32);
                    // the last radix-S layer over e: the quad's lanes hold the S columns' sums.  Own term: + in the lower lane of a pair, - in the upper
                    const float sg1 = (lam & 1) ? -1.0f : 1.0f;
                    [[maybe_unused]] const float sg2 = (lam & 2) ? -1.0f : 1.0f;
                    [[maybe_unused]] const bool rot = lam == 3;
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const float4 t = ld4(&wpr[2 * i]);
                        cf x0 = cmul(v[2 * i], mk(t.x, t.y)), x1 = cmul(v[2 * i + 1], mk(t.z, t.w));
                        if constexpr (S == 4) {
                            // lanes 0, 1: s_e = x_e + x_(e+2); lanes 2, 3: d_e = x_e - x_(e+2); lane 3: d_1 -> -j d_1
                            x0 = quad_xor2(x0) + x0 * sg2; x1 = quad_xor2(x1) + x1 * sg2;
                            x0 = rot ? mk(x0.y, -x0.x) : x0; x1 = rot ? mk(x1.y, -x1.x) : x1;
                        }
                        // S = 2: even lane F0 + W F1 (slot k), odd lane F0 - W F1 (slot k + 32 P)
                        // S = 4: lane 0: s0 + s1 (i2 = 0), lane 1: s0 - s1 (i2 = 2), lane 2: d0 - j d1 (i2 = 1), lane 3: d0 + j d1 (i2 = 3)
                        v[2 * i] = quad_xor1(x0) + x0 * sg1;
                        v[2 * i + 1] = quad_xor1(x1) + x1 * sg1;
                        __builtin_amdgcn_sched_barrier(0);            // one table read at a time: the phase has no registers for more
                    }
                    // row m' = (64 (block of the run) + lane) / S of the lane's slot
                    const unsigned rb = (unsigned)(m * kRows + rowbase + (64 / S) * (kTB * tr + rh2) + ((t3 & 63) >> kLS)) * 8u;
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const uint4 t = sow[q];
                        const unsigned so[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                        for (int e = 0; e < 4; e++) bst2t<NT>(rout, (so[e] == 0xFFFFFFFFu ? 0xFFFFFFF0u : so[e] + rb), v[4 * q + e]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        stage2([&](int j, int ps) { return unpack_nar(G[j][ps]); }, R4 ? 64 / S : 0, std::integral_constant<int, 2>{});
        if constexpr (R4) {
            // rows 64..127 of the inverse transforms = the first output rows: this lane's own stores, served by the L2
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stage2([&](int j, int ps) { return bld2_sc1(rscr, (unsigned)tid * 8u + (unsigned)(j * 4096 + ps * 16384), 0u); }, 0,
                   std::integral_constant<int, 1>{});
        }
        // the trip region (= the strips) was last read before the barrier above: the next block starts without one
    }
}

hipError_t init_block_narrow_kernels()
{
    hipError_t e = hipSuccess;
#define FDC_SETN(S, A, B, H, Q, P) \
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_blknar<S, A, B, H, Q, P>), hipFuncAttributeMaxDynamicSharedMemorySize, NarGeom<S, P>::kLds);
#define FDC_SETNS(S, H, Q, P) FDC_SETN(S, true, false, H, Q, P) FDC_SETN(S, false, false, H, Q, P) FDC_SETN(S, true, true, H, Q, P) FDC_SETN(S, false, true, H, Q, P)
#define FDC_SETNP(P) \
    FDC_SETNS(2, false, 0, P) FDC_SETNS(2, true, 0, P) FDC_SETNS(2, false, 2, P) FDC_SETNS(2, false, 6, P) \
    FDC_SETNS(4, false, 0, P) FDC_SETNS(4, true, 0, P) FDC_SETNS(4, false, 1, P) FDC_SETNS(4, false, 3, P)
    FDC_SETNP(8) FDC_SETNP(4) FDC_SETNP(2)
#undef FDC_SETNP
#undef FDC_SETNS
#undef FDC_SETN
    return e;
}

// r: the bank's offset from the l-bin grid: 0, l/4, l/2 (= half), 3l/4
template <int P>
static hipError_t launch_narrow_p(int L, const float2 *in, size_t in_stride, float2 *out, int nb_chunk, int mbase, int nb_call, const float2 *tab,
                                  const float2 *cbt, const long long *slot_off, unsigned out_bytes, int grid, int hints, hipStream_t s,
                                  hipEvent_t ev_start, hipEvent_t ev_stop, int R, float2 *scratch, int r)
{
    const int rows = R == 4 ? 3 * L / 4 : L / 2, quarter = r / (L / 4);
#define FDC_LNAR(S, A, B, H, Q) \
    hipExtLaunchKernelGGL((k_blknar<S, A, B, H, Q, P>), dim3((unsigned)grid), dim3(512), NarGeom<S, P>::kLds, s, ev_start, ev_stop, 0u, in, in_stride, out, tab, cbt, \
                          slot_off, (long long)mbase * rows, (long long)nb_call, out_bytes, nb_chunk, hints, B ? scratch : (float2 *)nullptr)
#define FDC_LNARH(S, H, Q) \
    do { \
        if (R == 4) { if (hints & 1) FDC_LNAR(S, true, true, H, Q); else FDC_LNAR(S, false, true, H, Q); } \
        else { if (hints & 1) FDC_LNAR(S, true, false, H, Q); else FDC_LNAR(S, false, false, H, Q); } \
    } while (0)
    if (L == 128) {
        if (quarter == 2) FDC_LNARH(2, true, 0); else if (quarter == 1) FDC_LNARH(2, false, 2); else if (quarter == 3) FDC_LNARH(2, false, 6); else FDC_LNARH(2, false, 0);
    } else {
        if (quarter == 2) FDC_LNARH(4, true, 0); else if (quarter == 1) FDC_LNARH(4, false, 1); else if (quarter == 3) FDC_LNARH(4, false, 3); else FDC_LNARH(4, false, 0);
    }
#undef FDC_LNARH
#undef FDC_LNAR
    return hipGetLastError();
}

hipError_t launch_poly_block_narrow(int L, const float2 *in, size_t in_stride, float2 *out, int nb_chunk, int mbase, int nb_call, const float2 *tab,
                                    const float2 *cbt, const long long *slot_off, unsigned out_bytes, int ncu, int hints, hipStream_t s,
                                    hipEvent_t ev_start, hipEvent_t ev_stop, int R, float2 *scratch, int r, int N)
{
    if (nb_chunk <= 0) return hipSuccess;
    if (!poly_block_narrow_supports(N, L, R) || (R == 4 && !scratch) || r < 0 || r >= L || (r % (L / 4))) return hipErrorInvalidValue;
    int grid = ncu > 0 ? ncu : 256;
    if (grid > nb_chunk) grid = nb_chunk;
    switch (N) {
    case 65536: return launch_narrow_p<8>(L, in, in_stride, out, nb_chunk, mbase, nb_call, tab, cbt, slot_off, out_bytes, grid, hints, s, ev_start, ev_stop, R, scratch, r);
    case 32768: return launch_narrow_p<4>(L, in, in_stride, out, nb_chunk, mbase, nb_call, tab, cbt, slot_off, out_bytes, grid, hints, s, ev_start, ev_stop, R, scratch, r);
    default: return launch_narrow_p<2>(L, in, in_stride, out, nb_chunk, mbase, nb_call, tab, cbt, slot_off, out_bytes, grid, hints, s, ev_start, ev_stop, R, scratch, r);
    }
}

}  // namespace fdc
