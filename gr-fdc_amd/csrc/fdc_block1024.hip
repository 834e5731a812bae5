// gfx950 one-block-per-CU form of the uniform-plan path for channels of l = 1024 bins (N = 65536 = 1024 rows x 64 columns, R = 2, every
// channel on the 1024-bin grid, one window): fdc_block512.hip taken one step further — a column's 1024 rows split into FOUR PHASES
// n2 = 4 mu + rho, each a 256-point sub-sequence that is exactly a "column" of fdc_block256.hip.  A quad of lanes is the four phases of ONE
// column; all four run the 256-point transforms of the old stage 1 in lockstep, and the radix-4 layers that join them are two DPP exchanges
// inside the quad (lane ^ 2, a -j / +j on lane 3, lane ^ 1):
//     forward (decimation in time):  A[kap + 256 i] = sum_rho W_4^(i rho) (W_1024^(kap rho) E_rho[kap])          kap = 0 .. 255
//                                    natural phase order in, lane lam holds i = bit-swapped lam afterwards
//     product: shape[k2]/N (-1)^n1 W_N^(n1 k2), k2 = kap + 256 i, all of it before the inverse layer (its i part differs between the lanes that
//              are about to be added); the ifftshift of the 1024-point inverse (k2 ^ 512: i ^ 2) is a factor (-1)^rho' on the result
//     inverse (decimation in frequency):  g[4 m + rho'] = IFFT256{ (-1)^rho' conj(W_1024^(kap rho')) sum_i W_4^(-i rho') U[kap + 256 i] }[m]
//                                    bit-swapped lane order in, lane rho' holds phase rho' afterwards
//     kept: t = 4 m + rho' >= 512  <=>  m >= 128: eight values per lane and pass, as before: G is 8 passes x 8 = 128 VGPRs.
// A wave owns one column per pass, the workgroup 8: 8 passes for the 64 columns.  A wave's own load instruction would be 64 single 8-byte samples, one
// per row (64 cache lines): the rows are fetched by the workgroup as whole 64-byte row segments and handed over through LDS instead (STAGED, below).
// Stage 2 is the FFT-64 over the columns n1 = 8 pass + c3: DFT-8 over the pass index in registers, W_64^(c3 klo), one trip through LDS
// ([128 rows][8 klo][8 c3], four trips for the 512 kept rows), DFT-8 over c3 in the lane that owns (row, klo); a wave's store is 64 consecutive
// samples of one channel.
//
// The arithmetic is that of k_p1g + k_p2g (fdc_kernels.hip) regrouped; parity against the oracle: tests/test_parity_gpu.py.
#include <hip/hip_ext.h>
#include <type_traits>
#include "fdc_kernels.h"
#include "fdc_radix16.hpp"
#include "fdc_devutil.hpp"

namespace fdc {

extern __shared__ __attribute__((aligned(16))) unsigned char fdc_smem_b1k[];

__device__ __forceinline__ unsigned long long pack1k(cf v) { return ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x); }
__device__ __forceinline__ cf unpack1k(unsigned long long u) { return mk(__uint_as_float((unsigned)u), __uint_as_float((unsigned)(u >> 32))); }
#if defined(__HIP_DEVICE_COMPILE__)
#define FDC_PLAIN_DS1K __attribute__((target("no-load-store-opt")))
#else
#define FDC_PLAIN_DS1K
#endif

__device__ __forceinline__ cf q1k_xor1(cf x)
{
    return mk(__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x.x), 0xB1, 0xF, 0xF, true)),
              __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x.y), 0xB1, 0xF, 0xF, true)));
}
__device__ __forceinline__ cf q1k_xor2(cf x)
{
    return mk(__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x.x), 0x4E, 0xF, 0xF, true)),
              __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x.y), 0x4E, 0xF, 0xF, true)));
}

// LDS map (bytes).  The block length is a template parameter (round 5): N = 1024 rows x (8 P) columns, P = 2, 4, 8 passes of 8 columns: N = 16384, 32768,
// 65536; the channel slots are the columns N1 = 8 P.  Stage 1 depends on P through the row pitch and the table sizes only; G is 8 P registers per lane;
// stage 2 is a DFT-P over the pass index in registers, one trip through LDS ([1024 / P rows][P klo][8 c3]: P = 8: 128 rows, four trips for the 512 kept
// rows; P = 2: all 512 in one) and the same DFT-8 over c3; a wave reads (klo = wave mod P, 128-row block = wave div P).
constexpr int kKScrPts = 1084;                                   // per-wave exchange strip, as in fdc_block256.hip
constexpr int kKOffX = 8 * kKScrPts * 8;                         // 69376: end of the strips
// Round 5 (41 % of this kernel's LDS cycles were bank conflicts, profiles/r05/NOTES.md section 5): the four lanes of a quad read the rows (rho, b) /
// (i, b) of these two tables in ONE instruction, and with rho / i strides of whole multiples of 64 dwords all four fell on the same banks (4-way).
// T1k: rho stride 296 points (592 dwords = 4 windows of 4 dwords mod 16): the 4 x 4 rows of a 16-byte read's lane group on 16 different windows.
// Sh: rows of 18 floats, i stride 304 floats (16 mod 64): the 4 x 8 rows of an 8-byte read's 32 lanes on all 64 banks once.
constexpr int kKT1kRho = 16 * 18 + 8;
constexpr int kKShRow = 18, kKShQ = 16 * kKShRow + 16;
// STAGED loads (below): the next pass's 8 columns x 1024 rows as [column][row] planes, 2 points of padding per plane: a store instruction's sixteen
// contiguous lanes are 4 column pairs x 4 rows, and an 8-byte STORE is banked on 32 dwords (MI355X_MICROARCH.md, LDS table): dword 2 (2 cp 1026 +
// row) = 8 cp + 2 row mod 32 — four windows of 8 dwords, conflict-free (round 5; 1032-point planes put all four pairs on the same banks: 4-way).
// A wave reads 64 consecutive rows of its one column: conflict-free either way.
constexpr int kKStagePlane = 1024 + 2;
template <int P>
struct B1kGeom {
    static_assert(P == 2 || P == 4 || P == 8, "passes of 8 columns: N = 16384, 32768 or 65536");
    static constexpr int kN1 = 8 * P;                             // columns = channel slots
    static constexpr int kN = 1024 * kN1;
    static constexpr int kTripRows = 1024 / P;                    // rows of one stage-2 trip
    static constexpr int kJT = 16 / P;                            // 64-row groups per trip
    static constexpr int kLd = P * 8 + 2;                         // trip rows: [P klo][8 c3] + 2 (row stride 132 / 68 / 36 dwords: 4 x an odd number: a 16-byte read's
                                                                  // consecutive rows on different 4-dword windows)
    static constexpr int kTrip = kTripRows * kLd * 8;             // P = 8: 67584 (below the strips' end); 4: 69632; 2: 73728
    static constexpr int kOffCt = kTrip > kKOffX ? kTrip : kKOffX;   // [8 c3][P klo]  W_N1^(c3 klo): behind the strips and the trip buffer
    static constexpr int kOffWrow = kOffCt + 8 * P * 8;           // [16][18]  W_256^(b p)
    static constexpr int kOffT1k = kOffWrow + 16 * 18 * 8;        // [4 rho][16 b][18]  W_1024^(rho (b + 16 q))
    static constexpr int kOffB = kOffT1k + 4 * kKT1kRho * 8;      // [N1 n1][18]  W_N^(16 n1 q)
    static constexpr int kOffSh = kOffB + kN1 * 18 * 8;           // [4 i][16 b][18] floats: shape[b + 16 q + 256 i] / N
    static constexpr int kOffSoff = kOffSh + 4 * kKShQ * 4;       // [P klo][8] output offsets (bytes)
    static constexpr int kLds = kOffSoff + kN1 * 4;               // P = 8: 96000
    static constexpr int kOffStage = kLds;
    static constexpr int kLdsStaged = kOffStage + 8 * kKStagePlane * 8;   // P = 8: 162048 <= 163840
    static_assert(kOffCt % 16 == 0 && kOffB % 16 == 0 && kOffSh % 16 == 0 && kOffSoff % 16 == 0 && kOffStage % 16 == 0, "aligned table reads");
    static_assert(kLdsStaged <= 160 * 1024, "LDS budget");
    static_assert((kLd * 2) % 8 == 4, "trip rows an odd number of 4-dword windows apart");
};
template <int P> __device__ __forceinline__ constexpr int b1k_pass_idx(int k) { return P == 8 ? 4 * (k & 1) + (k >> 1) : k; }
template <int P>
__device__ __forceinline__ void b1k_pass_dft(cf (&a)[P])
{
    if constexpr (P == 8) dft8<false>(a);                          // klo = k0 + 2 k1 in a[4 k0 + k1]
    else if constexpr (P == 4) dft4<false>(a[0], a[1], a[2], a[3]);
    else { const cf s0 = a[0] + a[1], d0 = a[0] - a[1]; a[0] = s0; a[1] = d0; }
}

// STAGED = true: the rows reach the lanes through LDS.  One column per wave means a wave's own load instruction is 64 single samples from 64 rows
// (64 cache lines); staged, the workgroup's eight waves fetch the pass's 8 columns x 1024 rows in 16-byte pieces of whole 64-byte row segments (wave w:
// rows 128 w .., 16 rows x 4 column pairs per instruction: 16 lines, half the instructions), park them in registers for a pass as before, write them to
// [column][row] planes in LDS at the pass boundary and read their own column back: two workgroup barriers per pass for an eighth of the line requests.
// R4 = true: relinvovl = 4 (the reference's default overlap): 768 of the 1024 samples of every inverse transform are kept.  The rows m >= 128 of all four
// phases stay in the G registers as for R = 2 (output rows 256 ..); the rows 64 <= m < 128 go to 128 KiB of per-workgroup scratch ([pass][q - 4][thread]:
// the L2 holds it) and come back for a second, 256-row run of stage 2 (output rows 0 .. 255), as in fdc_block512.hip.
template <bool NT, bool STAGED, bool R4, int P = 8>
__global__ FDC_PLAIN_DS1K __launch_bounds__(512) void k_blk1024(const float2 *__restrict__ in, size_t in_stride, float2 *__restrict__ out,
                                                      const float2 *__restrict__ tw256, const float2 *__restrict__ tw1024 /* W_1024^k, k < 1024 */,
                                                      const float2 *__restrict__ twq /* [n1][16] W_N^(16 n1 q) */,
                                                      const float2 *__restrict__ cbt /* [n1][64] (-1)^n1 W_N^(n1 (b + 256 i)) at b + 16 i */,
                                                      const float *__restrict__ shn /* [1024] shape / N */,
                                                      const long long *__restrict__ slot_off, long long out_base, long long nb_call,
                                                      unsigned out_bytes, int nb, int hints, int half, float2 *__restrict__ scratch)
{
    // half: the bank 512 bins higher (f = 1024 slot + 512).  The block modulated by exp(-2 pi i 512 n / N) = W_N^(512 n1) (-1)^n2 moves every column's
    // spectrum by half its length: the lane that holds quarter i of k2 holds quarter i ^ 2 of the modulated column (the host moves the quarters of shn
    // and cbt and puts W_N^(512 n1) into cbt), and the ifftshift — the same move again — no longer leaves a sign.
    typedef B1kGeom<P> GM;
    constexpr int kN1 = GM::kN1, kKLd = GM::kLd;
    float2 *scr = reinterpret_cast<float2 *>(fdc_smem_b1k);
    float2 *ctab = reinterpret_cast<float2 *>(fdc_smem_b1k + GM::kOffCt);
    float2 *wrow = reinterpret_cast<float2 *>(fdc_smem_b1k + GM::kOffWrow);
    float2 *t1k = reinterpret_cast<float2 *>(fdc_smem_b1k + GM::kOffT1k);
    float2 *Bt = reinterpret_cast<float2 *>(fdc_smem_b1k + GM::kOffB);
    float *Sh = reinterpret_cast<float *>(fdc_smem_b1k + GM::kOffSh);
    unsigned *soff = reinterpret_cast<unsigned *>(fdc_smem_b1k + GM::kOffSoff);
    const int tid = threadIdx.x;
    // lane = rho + 4 b: rho = phase of the column's rows, b = row group of the 256-point sub-transform; the wave is the column of the pass
    const int w = tid >> 6, lane = tid & 63, rho = lane & 3, b = lane >> 2;
    const int iq = ((rho & 1) << 1) | (rho >> 1);                         // the quarter i of k2 this lane holds between the two radix-4 layers

    const int grid = gridDim.x, per = grid >> 3;
    const bool xmap = (grid & 7) == 0;
    const int first = xmap ? (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (first >= nb) return;

    constexpr unsigned inbytes = (unsigned)GM::kN * 8u;
    constexpr unsigned kRow64 = 64u * (unsigned)kN1 * 8u;          // 64 rows further on, bytes (P = 8: 32 KiB)
    constexpr unsigned kRow16 = 16u * (unsigned)kN1 * 8u;          // 16 rows (staged loads: one instruction further on)
    // row n2 = 4 (16 a + b) + rho of column 8 pass + w: kN1 columns per row; a adds 64 rows, a pass 8 columns = 64 B
    const unsigned voff = (unsigned)((4 * b + rho) * kN1 + w) * 8u;
    const __amdgpu_buffer_rsrc_t rcb = make_rsrc(cbt, (unsigned)kN1 * 64u * 8u);
    const unsigned voffc = (unsigned)(w * 64 + b + 16 * iq) * 8u;
    cf LA[16], LB[16], cbA, cbB;
    // staged: wave w fetches rows 128 w + 16 i + (lane >> 2), columns 2 (lane & 3), + 1 of the pass (16 bytes); instruction i adds 16 rows = 8 KiB
    float2 *stg = reinterpret_cast<float2 *>(fdc_smem_b1k + GM::kOffStage);
    const unsigned voffs = (unsigned)((128 * w + (lane >> 2)) * kN1 + 2 * (lane & 3)) * 8u;
    float2 *const stw = stg + 2 * (lane & 3) * kKStagePlane + 128 * w + (lane >> 2);       // + 16 i rows; second column: + one plane
    const float2 *const strd = stg + w * kKStagePlane + lane;                              // this lane's rows 64 a + lane of column w
    u32x4 PF[8];
    auto stage_load = [&](int mb, int pn) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rin = make_rsrc(in + (size_t)mb * in_stride + 8 * pn, inbytes);
#pragma unroll
        for (int i = 0; i < 8; i++) PF[i] = bld4(rin, voffs, (unsigned)i * kRow16);
    };
    auto stage_write = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            st2(&stw[16 * i], mk(__uint_as_float(PF[i].x), __uint_as_float(PF[i].y)));
            st2(&stw[kKStagePlane + 16 * i], mk(__uint_as_float(PF[i].z), __uint_as_float(PF[i].w)));
        }
    };
    if constexpr (STAGED) {
        stage_load(first, 0);
        cbA = bld2(rcb, voffc, 0);
        stage_write();                                           // pass 0 of the first block: visible after the barrier behind the tables
        stage_load(first, 1);
    } else {
        const __amdgpu_buffer_rsrc_t rin = make_rsrc(in + (size_t)first * in_stride, inbytes);
#pragma unroll
        for (int a = 0; a < 16; a++) LA[a] = bld2(rin, voff, (unsigned)a * kRow64);
        cbA = bld2(rcb, voffc, 0);
    }
    // ---- tables
    for (int i = tid; i < 256; i += 512) wrow[(i >> 4) * 18 + (i & 15)] = tw256[((i >> 4) * (i & 15)) & 255];
    for (int i = tid; i < 1024; i += 512) {
        const int r = i >> 8, kap = i & 255;                              // [rho][b][q], kap = b + 16 q
        t1k[r * kKT1kRho + (kap & 15) * 18 + (kap >> 4)] = tw1024[(r * kap) & 1023];
        Sh[(i >> 8) * kKShQ + (i & 15) * kKShRow + ((i >> 4) & 15)] = shn[i];   // i = b + 16 q + 256 quarter
    }
    for (int i = tid; i < kN1 * 16; i += 512) Bt[(i >> 4) * 18 + (i & 15)] = twq[i];     // [n1 = i >> 4][q]: twq is [n1][16] already
    for (int i = tid; i < kN1; i += 512) {
        const long long o = slot_off[i];                                  // slot i = klo + P khi, khi = k0 + 2 k1, is entry [klo][4 k0 + k1]
        const int klo = i % P, khi = i / P;
        soff[klo * 8 + 4 * (khi & 1) + (khi >> 1)] = o >= 0 ? (unsigned)((o * nb_call + out_base) * 8) : 0xFFFFFFFFu;
        ctab[i] = tw256[((256 / kN1) * khi * klo) & 255];                 // [c3 = i / P][klo] = W_N1^(c3 klo)
    }
    __syncthreads();

    float2 *const scrw = scr + w * kKScrPts + lane;
    const float2 *const scrr = scr + w * kKScrPts + rho + 68 * b;
    const float2 *const wr = wrow + b * 18;
    const float2 *const t1r = t1k + rho * kKT1kRho + b * 18;
    const float2 *const btw = Bt + w * 18;                                // + pass * 8 rows
    const float *const shr = Sh + iq * kKShQ + b * kKShRow;
    // the sign of a lane's own term in the lane ^ 1 and lane ^ 2 layers; lane 3 turns its value by -j (forward) / +j (inverse) between them
    const float sg1 = (rho & 1) ? -1.0f : 1.0f, sg2 = (rho & 2) ? -1.0f : 1.0f;
    const bool rot = rho == 3;
    const float osg = half ? 1.0f : sg1;                                  // (-1)^rho: the ifftshift (half: cancelled)
    const __amdgpu_buffer_rsrc_t rout = make_rsrc(out, out_bytes);
    const __amdgpu_buffer_rsrc_t rscr = make_rsrc(R4 ? scratch + (size_t)blockIdx.x * 16384 : scratch, R4 ? 16384u * 8u : 0u);
    constexpr int kRows = R4 ? 768 : 512;                                 // kept samples per block and channel

    for (int m = first; m < nb; m += grid) {
        const int mnext = m + grid < nb ? m + grid : m;
        typedef unsigned long long gvec __attribute__((ext_vector_type(P)));
        gvec G[8];
        auto one_pass = [&](const int ps, cf (&cur)[16], const cf cb, cf (&L)[16], cf &cbn) __attribute__((always_inline)) {
            if constexpr (STAGED) {
                // the pass's rows are in the planes (written a pass ago, or by the prologue): take this lane's sixteen, then hand the planes over to
                // the rows that arrived in the meantime (pass + 1) and request pass + 2
                __syncthreads();
#pragma unroll
                for (int a = 0; a < 16; a++) cur[a] = ld2(&strd[64 * a]);
                __syncthreads();
                stage_write();
                stage_load(ps < P - 2 ? m : mnext, (ps + 2) & (P - 1));
                cbn = bld2(rcb, voffc, (unsigned)((ps + 1) & (P - 1)) * 4096u);
            } else {
                const int pn = ps < P - 1 ? ps + 1 : 0;
                const int mb = ps < P - 1 ? m : mnext;
                const __amdgpu_buffer_rsrc_t rin = make_rsrc(in + (size_t)mb * in_stride + 8 * pn, inbytes);
                if (hints & 2) {
#pragma unroll
                    for (int a = 0; a < 16; a++) L[a] = bld2_nt(rin, voff, (unsigned)a * kRow64);
                } else {
#pragma unroll
                    for (int a = 0; a < 16; a++) L[a] = bld2(rin, voff, (unsigned)a * kRow64);
                }
                cbn = bld2(rcb, voffc, (unsigned)pn * 4096u);
            }
            // ---- the 256-point forward transform of this lane's phase: exactly the old stage 1
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
            dft16<false>(v);                                      // E_rho at kap = b + 16 q in v[rev16(q)]
            // ---- forward radix-4 layer, product, inverse radix-4 layer: two values at a time, 16-byte table reads
            cf u[16];
            {
                const float2 *bpr = btw + ps * (8 * 18);
#pragma unroll
                for (int g2 = 0; g2 < 8; g2++) {
                    const float4 ta = ld4(&t1r[2 * g2]), ba = ld4(&bpr[2 * g2]);
                    const float2 sh = *reinterpret_cast<const float2 *>(&shr[2 * g2]);
                    const cf w1s[2] = {mk(ta.x, ta.y), mk(ta.z, ta.w)}, bps[2] = {mk(ba.x, ba.y), mk(ba.z, ba.w)};
                    const float shs[2] = {sh.x, sh.y};
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        const int q = 2 * g2 + e;
                        cf x = cmul(v[rev16(q)], w1s[e]);                            // E_rho W_1024^(kap rho)
                        x = q1k_xor2(x) + x * sg2;                                   // lanes 0, 1: x_rho + x_(rho+2); lanes 2, 3: x_(rho-2) - x_rho
                        x = rot ? mk(x.y, -x.x) : x;                                 // lane 3: -j
                        x = q1k_xor1(x) + x * sg1;                                   // A[kap + 256 iq]
                        // shape / N, W_N^(16 n1 q), (-1)^n1 W_N^(n1 (b + 256 iq)) = cb
                        cf y = cmul(cmul(x, bps[e]), cb) * shs[e];
                        y = q1k_xor1(y) + y * sg1;                                   // lanes (0, 1) hold quarters 0, 2; lanes (2, 3) quarters 1, 3
                        y = rot ? mk(-y.y, y.x) : y;                                 // lane 3: +j
                        y = q1k_xor2(y) + y * sg2;                                   // sum_i W_4^(-i rho) U[kap + 256 i] in lane rho
                        u[q] = cmulc(y, w1s[e]) * osg;                               // conj(W_1024^(kap rho)), and (-1)^rho: the ifftshift
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // ---- the 256-point inverse transform of this lane's phase (no q ^ 8: the shift was the sign above)
            dft16<true>(u);
            {
                cf tw[16];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const float4 t = ld4(&wr[2 * i]);
                    tw[2 * i] = mk(t.x, t.y); tw[2 * i + 1] = mk(t.z, t.w);
                }
#pragma unroll
                for (int p = 1; p < 16; p++) u[rev16(p)] = cmulc(u[rev16(p)], tw[p]);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int p = 0; p < 16; p++) st2(&scrw[68 * p], u[rev16(p)]);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int bb = 0; bb < 16; bb++) u[bb] = ld2(&scrr[4 * bb]);
            dft16<true>(u);                                       // g[4 m + rho], m = b + 16 q in u[rev16(q)]; keep q >= 8
#pragma unroll
            for (int j = 0; j < 8; j++) G[j][ps] = pack1k(u[rev16(8 + j)]);
            if constexpr (R4) {                                   // R = 4 keeps q >= 4: m = 64 .. 127 go to the scratch, [pass][q - 4][thread]
#pragma unroll
                for (int j = 0; j < 4; j++) bst2(rscr, (unsigned)tid * 8u + (unsigned)j * 4096u, (unsigned)ps * 16384u, u[rev16(4 + j)]);
            }
        };
#pragma nounroll
        for (int pp = 0; pp < P; pp += 2) {
            one_pass(pp, LA, cbA, LB, cbB);
            one_pass(pp + 1, LB, cbB, LA, cbA);
        }
        // ---------------- stage 2: FFT-N1 over n1 = 8 pass + c3 of every row t' = rowbase + 4 (b + 16 j) + rho = rowbase + lane + 64 j ----------------
        // get(j, pass): the value of row group j; rowbase: first output row of the run; njc: its number of 64-row groups (a trip holds kJT of them:
        // 1024 / P rows; a run of fewer — the rows that come back from the scratch at N = 16384 — leaves the upper row blocks' waves without rows)
        auto stage2 = [&](auto get, const int rowbase, auto njc) __attribute__((always_inline)) {
            constexpr int kNJ = decltype(njc)::value, kJT = GM::kJT, kNTrip = (kNJ + kJT - 1) / kJT, kJA = kNJ < kJT ? kNJ : kJT;
            static_assert(kNJ % kJA == 0, "whole trips");
            __syncthreads();                                          // every wave is done with its strip / the previous run's trip
            int t2 = tid;
            asm volatile("" : "+v"(t2));
            const int lane2 = t2 & 63, w2 = __builtin_amdgcn_readfirstlane(t2 >> 6);
            const int klo2 = w2 % P, rh2 = w2 / P;                    // reader: klo, 128-row block of the trip (0 for P = 8)
            float2 *const gw = scr + lane2 * kKLd + w2;               // element (row lane + 64 jj, klo) at + 64 jj kLd + 8 klo
            const float2 *const gr = scr + (128 * rh2 + lane2) * kKLd + 8 * klo2;     // row = 128 rh + lane (+ 64 hh), klo: 8 consecutive points
            const uint4 *const sow = reinterpret_cast<const uint4 *>(soff + 8 * klo2);
            cf ct[P];
            {
                const float2 *ctr = reinterpret_cast<const float2 *>(fdc_smem_b1k + GM::kOffCt) + w2 * P;
#pragma unroll
                for (int i = 0; i < P / 2; i++) {
                    const float4 t = ld4(&ctr[2 * i]);
                    ct[2 * i] = mk(t.x, t.y); ct[2 * i + 1] = mk(t.z, t.w);
                }
            }
#pragma unroll
            for (int tr = 0; tr < kNTrip; tr++) {
                cf src[kJA][P];
#pragma unroll
                for (int jj = 0; jj < kJA; jj++)
#pragma unroll
                    for (int ps = 0; ps < P; ps++) src[jj][ps] = get(kJA * tr + jj, ps);
#pragma unroll
                for (int jj = 0; jj < kJA; jj++) {
                    cf a[P];
#pragma unroll
                    for (int ps = 0; ps < P; ps++) a[ps] = src[jj][ps];
                    b1k_pass_dft<P>(a);
                    float2 *const g = gw + jj * 64 * kKLd;
                    st2(&g[0], a[0]);
#pragma unroll
                    for (int k = 1; k < P; k++) st2(&g[8 * k], cmul(a[b1k_pass_idx<P>(k)], ct[k]));
                }
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();                                      // the trip is in LDS
                const bool active = kJA == kJT || 2 * rh2 < kJA;      // wave-uniform (kJA even): this wave's 128 rows exist in the trip
                cf v[2][8];
                if (active) {
#pragma unroll
                    for (int hh = 0; hh < 2; hh++)
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const float4 t = ld4(&gr[hh * 64 * kKLd + 2 * i]);
                            v[hh][2 * i] = mk(t.x, t.y); v[hh][2 * i + 1] = mk(t.z, t.w);
                        }
                }
                __syncthreads();                                      // every read of the trip is done
                __builtin_amdgcn_sched_barrier(0);
                if (active) {
                    const uint4 s0 = sow[0], s1 = sow[1];
                    const unsigned so[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
                    for (int hh = 0; hh < 2; hh++) {
                        dft8<false>(v[hh]);                           // khi = k0 + 2 k1 in v[4 k0 + k1]
                        const unsigned rb = (unsigned)(m * kRows + rowbase + 64 * kJA * tr + 128 * rh2 + 64 * hh + lane2) * 8u;
#pragma unroll
                        for (int e = 0; e < 8; e++) bst2t<NT>(rout, (so[e] == 0xFFFFFFFFu ? 0xFFFFFFF0u : so[e] + rb), v[hh][e]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        stage2([&](int j, int ps) { return unpack1k(G[j][ps]); }, R4 ? 256 : 0, std::integral_constant<int, 8>{});
        if constexpr (R4) {
            // m = 64 .. 127 = output rows 0 .. 255: this lane's own stores, served by the L2
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stage2([&](int j, int ps) { return bld2_sc1(rscr, (unsigned)tid * 8u + (unsigned)(j * 4096 + ps * 16384), 0u); }, 0,
                   std::integral_constant<int, 4>{});
        }
    }
}

#ifndef FDC_1K_STAGED
#define FDC_1K_STAGED 1
#endif

bool poly_block1024_supports(int N, int R)
{
    return (N == 65536 || N == 32768 || N == 16384) && (R == 2 || R == 4);
}

hipError_t init_block1024_kernels()
{
    hipError_t e = hipSuccess;
#define FDC_SET1K(A, B, C, P) \
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_blk1024<A, B, C, P>), hipFuncAttributeMaxDynamicSharedMemorySize, B ? B1kGeom<P>::kLdsStaged : B1kGeom<P>::kLds);
    FDC_SET1K(true, false, false, 8) FDC_SET1K(false, false, false, 8) FDC_SET1K(true, true, false, 8) FDC_SET1K(false, true, false, 8)
    FDC_SET1K(true, false, true, 8) FDC_SET1K(false, false, true, 8) FDC_SET1K(true, true, true, 8) FDC_SET1K(false, true, true, 8)
    FDC_SET1K(true, true, false, 4) FDC_SET1K(false, true, false, 4) FDC_SET1K(true, true, true, 4) FDC_SET1K(false, true, true, 4)
    FDC_SET1K(true, true, false, 2) FDC_SET1K(false, true, false, 2) FDC_SET1K(true, true, true, 2) FDC_SET1K(false, true, true, 2)
#undef FDC_SET1K
    return e;
}

hipError_t launch_poly_block1024(const float2 *in, size_t in_stride, float2 *out, int nb_chunk, int mbase, int nb_call, const float2 *tw256,
                                 const float2 *tw1024, const float2 *twq, const float2 *cbt, const float *shn, const long long *slot_off,
                                 unsigned out_bytes, int ncu, int hints, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop, bool half, int R,
                                 float2 *scratch, int N)
{
    if (nb_chunk <= 0) return hipSuccess;
    if (!poly_block1024_supports(N, R) || (R == 4 && !scratch)) return hipErrorInvalidValue;
    int grid = ncu > 0 ? ncu : 256;
    if (grid > nb_chunk) grid = nb_chunk;
    constexpr bool kStaged = FDC_1K_STAGED != 0;
#define FDC_L1K(A, S, C, P) \
    hipExtLaunchKernelGGL((k_blk1024<A, S, C, P>), dim3((unsigned)grid), dim3(512), S ? B1kGeom<P>::kLdsStaged : B1kGeom<P>::kLds, s, ev_start, ev_stop, 0u, in, in_stride, \
                          out, tw256, tw1024, twq, cbt, shn, slot_off, (long long)mbase * (C ? 768 : 512), (long long)nb_call, out_bytes, nb_chunk, hints, \
                          half ? 1 : 0, C ? scratch : (float2 *)nullptr)
    const bool nt = (hints & 1) != 0;
    if (N == 65536) {
        if (R == 4) { if (nt) FDC_L1K(true, kStaged, true, 8); else FDC_L1K(false, kStaged, true, 8); }
        else { if (nt) FDC_L1K(true, kStaged, false, 8); else FDC_L1K(false, kStaged, false, 8); }
    } else if (N == 32768) {
        if (R == 4) { if (nt) FDC_L1K(true, true, true, 4); else FDC_L1K(false, true, true, 4); }
        else { if (nt) FDC_L1K(true, true, false, 4); else FDC_L1K(false, true, false, 4); }
    } else {
        if (R == 4) { if (nt) FDC_L1K(true, true, true, 2); else FDC_L1K(false, true, true, 2); }
        else { if (nt) FDC_L1K(true, true, false, 2); else FDC_L1K(false, true, false, 2); }
    }
#undef FDC_L1K
    return hipGetLastError();
}

}  // namespace fdc
