// gfx950 one-block-per-CU form of the uniform-plan path for channels of l = 512 bins (N = 65536 = 512 rows x 128 columns, R = 2,
// every channel on the 512-bin grid, one window): the machinery of fdc_block256.hip with each column's 512 rows split by PARITY.
//
// Why not "the same kernel with a bigger number": G is 256 KiB whatever l is, but a 512-point column in the lane layout of
// fdc_block256.hip (16 lanes x 16 row registers per column) would need 32 row registers per lane and pass — 64 VGPRs beside the 64 of
// the prefetch and the 128 of G.  Instead the column's rows n2 = 2 mu + par are two 256-point sub-sequences, each of which is exactly a
// "column" of the 256 kernel: a QUAD of lanes is (column c, parity 0), (c + 1, 0), (c, 1), (c + 1, 1), both parities run the 256-point
// transforms of the old stage 1 in lockstep on the same instructions, and the radix-2 layer that joins them is an exchange with lane ^ 2
// (DPP quad_perm [2, 3, 0, 1]):
//     forward (decimation in time):   A[kap]       = E[kap] + W_512^kap O[kap]          kap = 0 .. 255
//                                     A[kap + 256] = E[kap] - W_512^kap O[kap]          -> parity-0 lanes hold the lower half of k2, parity-1 the upper
//     product: shape[k2]/N (-1)^n1 W_N^(n1 k2), k2 = kap + 256 h — all of it BEFORE the inverse layer (its h part differs between the lanes
//              that are about to be added); the ifftshift of the 512-point inverse (k2 ^ 256) swaps the roles of the two lanes of a pair
//     inverse (decimation in frequency): with P = the value the h = 1 lane holds (index kap after the shift), Q = the h = 0 lane's (kap + 256):
//                                     even samples g[2m]     = IFFT256{ P + Q }[m]                       in the parity-0 lanes
//                                     odd samples  g[2m + 1] = IFFT256{ (P - Q) conj(W_512^kap) }[m]     in the parity-1 lanes
//     kept: t = 2m + par >= 256 <=> m >= 128: eight values per lane and pass, as before: G is 8 passes x 8 = 128 VGPRs.
// A wave owns 2 columns per pass, the workgroup 16: 8 passes for the 128 columns.  Stage 2 is the FFT-128 over the columns n1 = 16 pass + c4:
// DFT-8 over the pass index in registers, W_128^(c4 klo), one trip through LDS ([64 rows][8 klo][16 c4], four trips for the 256 kept rows),
// DFT-16 over c4 in the lane that owns (row, klo); a wave's store is 64 consecutive samples of one channel.
//
// The arithmetic is that of k_p1g + k_p2g (fdc_kernels.hip) regrouped; parity against the oracle: tests/test_parity_gpu.py.
#include <hip/hip_ext.h>
#include <type_traits>
#include "fdc_kernels.h"
#include "fdc_radix16.hpp"
#include "fdc_devutil.hpp"

namespace fdc {

extern __shared__ __attribute__((aligned(16))) unsigned char fdc_smem_b512[];

__device__ __forceinline__ unsigned long long pack512(cf v) { return ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x); }
__device__ __forceinline__ cf unpack512(unsigned long long u) { return mk(__uint_as_float((unsigned)u), __uint_as_float((unsigned)(u >> 32))); }
#if defined(__HIP_DEVICE_COMPILE__)
#define FDC_PLAIN_DS512 __attribute__((target("no-load-store-opt")))
#else
#define FDC_PLAIN_DS512
#endif

// the value of the lane two further on in the quad (lane ^ 2): the other parity of the same column
__device__ __forceinline__ cf swap_parity(cf x)
{
    return mk(__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x.x), 0x4E, 0xF, 0xF, true)),
              __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x.y), 0x4E, 0xF, 0xF, true)));
}

// LDS map (bytes).  The block length is a template parameter (round 5): N = 512 rows x (16 P) columns, P = 2, 4, 8 passes of 16 columns: N = 16384, 32768,
// 65536; the channel slots are the columns N1 = 16 P.  Stage 1 does not depend on P except through the row pitch and the table sizes; G is 16 P
// registers per lane; stage 2 is a DFT-P over the pass index in registers, one trip through LDS and the same DFT-16 over c4.  A trip holds 512 / P rows
// ([rows][P klo][16 c4]; P = 8: 64 rows, four trips for the 256 kept rows; P = 2: all 256 in one) and a wave reads (klo = wave mod P, row block = wave div P).
constexpr int k5ScrPts = 1084;                                   // per-wave exchange strip, as in fdc_block256.hip
constexpr int k5OffX = 8 * k5ScrPts * 8;                         // 69376: end of the strips
// W_512^(b + 16 q) for the parity-1 lanes as [16 b][18] from point k5T512Tab on; the parity-0 lanes' factor is 1: ONE row of ones at point 0, read by all
// of them (a broadcast).  The table starts 92 dwords behind the ones (= 28 mod 64 + 64): a 16-byte read's lane group then finds the ones and its
// four b rows on five different 4-dword windows (round 5: two tables [2 par][16][18] put (0, b) and (1, b) on the same banks — 2-way on every read)
constexpr int k5T512Tab = 46;
constexpr int k5ShPar = 16 * 16 + 4;                             // floats between the two halves of Sh: the rows (0, b) and (1, b) of a 16-byte read 4 dwords apart
// STAGED loads: the next pass's 16 columns x 512 rows as [column][row] planes.  Round 5 (the counters said a third of this kernel's LDS
// cycles were bank conflicts, profiles/r05/NOTES.md section 5): planes 528 points apart and the rows of column pair cp turned by 2 cp —
//   store (16 contiguous lanes = 8 column pairs x 2 rows, 8 bytes each): dword 2 (col 528 + row + 2 cp) = 32 col + 2 row + 4 cp mod 64: the eight pairs
//         on eight different 4-dword windows, the two columns of a pair (separate instructions) on the two bank halves: conflict-free;
//   load  (32 lanes = 8 row groups x two columns x two parities, consecutive rows): 32 creal + 4 b + 2 par + const: all 64 banks once.
// (516-point planes before: the pairs cp and cp + 4 of a store on the same banks, the two columns of a load 8 dwords apart: 2-way both.)
constexpr int k5StagePlane = 512 + 16;
template <int P>
struct B5Geom {
    static_assert(P == 2 || P == 4 || P == 8, "passes of 16 columns: N = 16384, 32768 or 65536");
    static constexpr int kN1 = 16 * P;                            // columns = channel slots
    static constexpr int kN = 512 * kN1;
    static constexpr int kTripRows = 512 / P;                     // rows of one stage-2 trip
    static constexpr int kJT = 16 / P;                            // 32-row groups (two parities x 16 b) per trip
    static constexpr int kLd = P * 16 + 6;                        // trip rows: [P klo][16 c4] + 6 points (row stride = 12 dwords mod 64 for P = 8, 4, 2: 268, 140, 76)
    static constexpr int kTrip = kTripRows * kLd * 8;             // P = 8: 68608; 4: 71680; 2: 77824
    static constexpr int kOffCt = kTrip > k5OffX ? kTrip : k5OffX;   // [16 c4][P klo]  W_N1^(c4 klo): behind the strips and the trip buffer
    static constexpr int kOffWrow = kOffCt + 16 * P * 8;          // [16][18]  W_256^(b p)
    static constexpr int kOffT512 = kOffWrow + 16 * 18 * 8;
    static constexpr int kOffB = kOffT512 + (k5T512Tab + 16 * 18) * 8;   // [P pass][16 c4][16]  W_N^(16 n1 q), n1 = 16 pass + c4 (rows unpadded: a wave reads two of them, broadcast)
    static constexpr int kOffSh = kOffB + P * 16 * 16 * 8;        // [2 h][16 b][16] floats: shape[b + 16 q + 256 h] / N
    static constexpr int kOffSoff = kOffSh + (k5ShPar + 16 * 16) * 4;    // [P klo][16] output offsets (bytes)
    static constexpr int kLds = kOffSoff + kN1 * 4;               // P = 8: 94336
    static constexpr int kOffStage = kLds;
    static constexpr int kLdsStaged = kOffStage + 16 * k5StagePlane * 8;   // P = 8: 161920
    static_assert(kOffB % 16 == 0 && kOffSh % 16 == 0 && kOffSoff % 16 == 0 && kOffCt % 16 == 0, "16-byte table reads");
    static_assert(kLdsStaged <= 160 * 1024, "LDS budget");
    static_assert((kLd * 2) % 64 == 12, "trip rows 12 dwords apart mod 64: the lane groups of ds_read_b128 on all banks");
};
// DFT over the pass index (the register index of G): P points in place; X[k] is read through b5_pass_idx<P>(k)
template <int P> __device__ __forceinline__ constexpr int b5_pass_idx(int k) { return P == 8 ? 4 * (k & 1) + (k >> 1) : k; }
template <int P>
__device__ __forceinline__ void b5_pass_dft(cf (&a)[P])
{
    if constexpr (P == 8) dft8<false>(a);                          // klo = k0 + 2 k1 in a[4 k0 + k1]
    else if constexpr (P == 4) dft4<false>(a[0], a[1], a[2], a[3]);
    else { const cf s0 = a[0] + a[1], d0 = a[0] - a[1]; a[0] = s0; a[1] = d0; }
}

// R4 = true: relinvovl = 4 (the reference's default overlap): 384 of the 512 samples of every inverse transform are kept.  The rows m >= 128 of
// both parities stay in the G registers as for R = 2 (output rows 128 ..); the rows 64 <= m < 128 go to 128 KiB of per-workgroup scratch
// ([pass][q - 4][thread]: the L2 holds it) and come back for a third, 128-row run of stage 2 (output rows 0 .. 127), as in fdc_block256.hip.
// STAGED = true: the rows reach the lanes through LDS, as in fdc_block1024.hip.  A wave's own load instruction is 32 rows x 16 bytes (32 cache lines);
// staged, wave w fetches rows 64 w .. of the pass's 16 columns in 16-byte pieces of whole 128-byte row segments (8 rows per instruction: 8 lines, half the
// instructions), parks them in registers for a pass, writes them to [column][row] planes in LDS at the pass boundary and reads its own columns' rows back:
// two workgroup barriers per pass for an eighth of the line requests.
template <bool NT, bool R4, bool STAGED, int P = 8>
__global__ FDC_PLAIN_DS512 __launch_bounds__(512) void k_blk512(const float2 *__restrict__ in, size_t in_stride, float2 *__restrict__ out,
                                                      const float2 *__restrict__ tw256, const float2 *__restrict__ tw512 /* W_512^k, k < 256 */,
                                                      const float2 *__restrict__ twq /* [n1][16] W_N^(16 n1 q) */,
                                                      const float2 *__restrict__ cbt /* [n1][32] (-1)^n1 W_N^(n1 (b + 256 h)) at b + 16 h */,
                                                      const float *__restrict__ shn /* [512] shape / N */,
                                                      const long long *__restrict__ slot_off, long long out_base, long long nb_call,
                                                      unsigned out_bytes, int nb, int hints, float2 *__restrict__ scratch, int half)
{
    // half: the bank 256 bins higher (f = 512 slot + 256).  The block modulated by exp(-2 pi i 256 n / N) = W_N^(256 n1) (-1)^n2 moves every column's
    // spectrum by half its length: the lane that holds half h of k2 holds half h ^ 1 of the modulated column.  The host swaps the halves of the
    // tables (shn, cbt) and puts W_N^(256 n1) into cbt; with the ifftshift (another swap) the inverse layer sees its halves in place, which leaves
    // one sign in the parity-1 lanes.
    typedef B5Geom<P> GM;
    constexpr int kN1 = GM::kN1, k5Ld = GM::kLd;
    float2 *scr = reinterpret_cast<float2 *>(fdc_smem_b512);
    float2 *ctab = reinterpret_cast<float2 *>(fdc_smem_b512 + GM::kOffCt);
    float2 *wrow = reinterpret_cast<float2 *>(fdc_smem_b512 + GM::kOffWrow);
    float2 *t512 = reinterpret_cast<float2 *>(fdc_smem_b512 + GM::kOffT512);
    float2 *Bt = reinterpret_cast<float2 *>(fdc_smem_b512 + GM::kOffB);
    float *Sh = reinterpret_cast<float *>(fdc_smem_b512 + GM::kOffSh);
    unsigned *soff = reinterpret_cast<unsigned *>(fdc_smem_b512 + GM::kOffSoff);
    const int tid = threadIdx.x;
    // lane = vc + 4 b: vc = creal + 2 par (a quad = two columns x two parities), b = row group of the 256-point sub-transform
    const int w = tid >> 6, lane = tid & 63, vc = lane & 3, creal = vc & 1, par = vc >> 1, b = lane >> 2, c4 = 2 * w + creal;

    const int grid = gridDim.x, per = grid >> 3;
    const bool xmap = (grid & 7) == 0;
    const int first = xmap ? (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (first >= nb) return;

    constexpr unsigned inbytes = (unsigned)GM::kN * 8u;
    constexpr unsigned kRow32 = 32u * (unsigned)kN1 * 8u;          // 32 rows further on, bytes (P = 8: 32 KiB)
    constexpr unsigned kRow8 = 8u * (unsigned)kN1 * 8u;            // 8 rows (staged loads: one instruction further on)
    // row n2 = 2 (16 a + b) + par of column 16 pass + c4: kN1 columns per row; a adds 32 rows, a pass 16 columns = 128 B
    const unsigned voff = (unsigned)((2 * b + par) * kN1 + c4) * 8u;
    const __amdgpu_buffer_rsrc_t rcb = make_rsrc(cbt, (unsigned)kN1 * 32u * 8u);
    const unsigned voffc = (unsigned)(c4 * 32 + b + 16 * par) * 8u;
    cf LA[16], LB[16], cbA, cbB;
    // staged: wave w fetches rows 64 w + 8 i + (lane >> 3), columns 2 (lane & 7), + 1 of the pass (16 bytes); instruction i adds 8 rows = 8 KiB
    [[maybe_unused]] float2 *stg = reinterpret_cast<float2 *>(fdc_smem_b512 + GM::kOffStage);
    [[maybe_unused]] const unsigned voffs = (unsigned)((64 * w + (lane >> 3)) * kN1 + 2 * (lane & 7)) * 8u;
    [[maybe_unused]] float2 *const stw = stg + 2 * (lane & 7) * k5StagePlane + 64 * w + (lane >> 3) + 2 * (lane & 7);   // + 8 i rows; second column: + one plane
    [[maybe_unused]] const float2 *const strd = stg + c4 * k5StagePlane + 2 * b + par + 2 * w;            // this lane's rows 32 a + 2 b + par of column c4 (pair w: turned by 2 w)
    [[maybe_unused]] u32x4 PF[8];
    [[maybe_unused]] auto stage_load = [&](int mb, int pn) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rin = make_rsrc(in + (size_t)mb * in_stride + 16 * pn, inbytes);
#pragma unroll
        for (int i = 0; i < 8; i++) PF[i] = bld4(rin, voffs, (unsigned)i * kRow8);
    };
    [[maybe_unused]] auto stage_write = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            st2(&stw[8 * i], mk(__uint_as_float(PF[i].x), __uint_as_float(PF[i].y)));
            st2(&stw[k5StagePlane + 8 * i], mk(__uint_as_float(PF[i].z), __uint_as_float(PF[i].w)));
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
        for (int a = 0; a < 16; a++) LA[a] = bld2(rin, voff, (unsigned)a * kRow32);
        cbA = bld2(rcb, voffc, 0);
    }
    // ---- tables
    for (int i = tid; i < 256; i += 512) {
        wrow[(i >> 4) * 18 + (i & 15)] = tw256[((i >> 4) * (i & 15)) & 255];
        t512[k5T512Tab + (i & 15) * 18 + (i >> 4)] = tw512[i];           // [b][q] = W_512^(b + 16 q), i = b + 16 q
        if (i < 18) t512[i] = make_float2(1.f, 0.f);                     // the parity-0 lanes' row of ones
    }
    for (int i = tid; i < kN1 * 16; i += 512) Bt[i] = twq[i];                             // [n1 = i >> 4][q]: twq is [n1][16] already
    for (int i = tid; i < kN1; i += 512) {
        const long long o = slot_off[i];                                  // slot i = klo + P khi is entry [klo][rev16(khi)]
        soff[(i % P) * 16 + rev16(i / P)] = o >= 0 ? (unsigned)((o * nb_call + out_base) * 8) : 0xFFFFFFFFu;
        ctab[i] = tw256[((256 / kN1) * (i / P) * (i % P)) & 255];         // [c4][klo] = W_N1^(c4 klo)
    }
    for (int i = tid; i < 512; i += 512) Sh[(i >> 8) * k5ShPar + (i & 15) * 16 + ((i >> 4) & 15)] = shn[i];   // i = b + 16 q + 256 h
    __syncthreads();

    float2 *const scrw = scr + w * k5ScrPts + lane;
    const float2 *const scrr = scr + w * k5ScrPts + vc + 68 * b;
    const float2 *const wr = wrow + b * 18;
    const float2 *const t5r = par ? t512 + k5T512Tab + b * 18 : t512;
    const float2 *const btr = Bt + c4 * 16;                              // + pass * 16 rows
    const float *const shr = Sh + par * k5ShPar + b * 16;
    const float fsgn = par ? -1.0f : 1.0f;                                // the sign of a lane's own term in both radix-2 layers
    const float hsgn = (half && par) ? -1.0f : 1.0f;
    const __amdgpu_buffer_rsrc_t rout = make_rsrc(out, out_bytes);
    const __amdgpu_buffer_rsrc_t rscr = make_rsrc(R4 ? scratch + (size_t)blockIdx.x * 32768 : scratch, R4 ? 32768u * 8u : 0u);
    constexpr int kRows = R4 ? 384 : 256;                                 // kept samples per block and channel

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
                for (int a = 0; a < 16; a++) cur[a] = ld2(&strd[32 * a]);
                __syncthreads();
                stage_write();
                stage_load(ps < P - 2 ? m : mnext, (ps + 2) & (P - 1));
                cbn = bld2(rcb, voffc, (unsigned)((ps + 1) & (P - 1)) * 4096u);
            } else {
                const int pn = ps < P - 1 ? ps + 1 : 0;
                const int mb = ps < P - 1 ? m : mnext;
                const __amdgpu_buffer_rsrc_t rin = make_rsrc(in + (size_t)mb * in_stride + 16 * pn, inbytes);
                if (hints & 2) {
#pragma unroll
                    for (int a = 0; a < 16; a++) L[a] = bld2_nt(rin, voff, (unsigned)a * kRow32);
                } else {
#pragma unroll
                    for (int a = 0; a < 16; a++) L[a] = bld2(rin, voff, (unsigned)a * kRow32);
                }
                cbn = bld2(rcb, voffc, (unsigned)pn * 4096u);
            }
            // ---- the 256-point forward transform of this lane's parity: exactly the old stage 1
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
            dft16<false>(v);                                      // E (par 0) / O (par 1) at kap = b + 16 q in v[rev16(q)]
            // ---- forward radix-2 layer, product, inverse radix-2 layer (the parity-0 lanes' "twiddle" is a table row of ones: no branches).
            // One value at a time, 8-byte table reads: the kernel has no registers for wider ones.
            cf u[16];
            {
                const float2 *bpr = btr + ps * (16 * 16);
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++) {                                     // four values at a time: 16-byte table reads, not hoisted further
                    const float4 ta = ld4(&t5r[4 * g4]), tb = ld4(&t5r[4 * g4 + 2]), ba = ld4(&bpr[4 * g4]), bb4 = ld4(&bpr[4 * g4 + 2]);
                    const float4 sh = *reinterpret_cast<const float4 *>(&shr[4 * g4]);
                    const cf w5s[4] = {mk(ta.x, ta.y), mk(ta.z, ta.w), mk(tb.x, tb.y), mk(tb.z, tb.w)};
                    const cf bps[4] = {mk(ba.x, ba.y), mk(ba.z, ba.w), mk(bb4.x, bb4.y), mk(bb4.z, bb4.w)};
                    const float shs[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int q = 4 * g4 + e;
                        const cf x = cmul(v[rev16(q)], w5s[e]);                      // par 1: O W_512^kap; par 0: E
                        const cf a = swap_parity(x) + x * fsgn;                      // par 0: E + O' = A[kap]; par 1: E - O' = A[kap + 256]
                        // shape / N, W_N^(16 n1 q), (-1)^n1 W_N^(n1 (b + 256 h)) = cb
                        const cf y = cmul(cmul(a, bps[e]), cb) * shs[e];
                        // ifftshift: the h = 1 lane's value is P (index kap), the h = 0 lane's is Q (index kap + 256)
                        const cf z = y + swap_parity(y) * fsgn;                      // par 0: Q + P; par 1: P - Q
                        u[q] = cmulc(z, w5s[e]) * hsgn;                              // par 1: conj(W_512^kap)
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // ---- the 256-point inverse transform of this lane's parity (no q ^ 8: the shift was the half swap)
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
            dft16<true>(u);                                       // g[2 m + par], m = b + 16 q in u[rev16(q)]; keep q >= 8
#pragma unroll
            for (int j = 0; j < 8; j++) G[j][ps] = pack512(u[rev16(8 + j)]);
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
        // ---------------- stage 2: FFT-N1 over n1 = 16 pass + c4 of every row t' = rowbase + 2 (b + 16 j) + par ----------------
        // get(j, pass): the value of row group j; rowbase: first output row of the run; njc: its number of 32-row groups (a trip holds kJT of them:
        // 512 / P rows; a run of fewer — the rows that come back from the scratch at N = 16384 — leaves the upper row blocks' waves without rows)
        auto stage2 = [&](auto get, const int rowbase, auto njc) __attribute__((always_inline)) {
            constexpr int kNJ = decltype(njc)::value, kJT = GM::kJT, kNTrip = (kNJ + kJT - 1) / kJT, kJA = kNJ < kJT ? kNJ : kJT;
            static_assert(kNJ % kJA == 0 && kJA % 2 == 0, "whole trips of an even number of row groups");
            __syncthreads();                                          // every wave is done with its strip / the previous run's trip
            int t2 = tid;
            asm volatile("" : "+v"(t2));
            const int lane2 = t2 & 63, w2 = __builtin_amdgcn_readfirstlane(t2 >> 6);
            const int vc2 = lane2 & 3, b_2 = lane2 >> 2, c4_2 = 2 * w2 + (vc2 & 1), par2 = vc2 >> 1;
            const int klo2 = w2 % P, rh2 = w2 / P;                    // reader: klo, 64-row block of the trip (0 for P = 8)
            float2 *const gw = scr + (2 * b_2 + par2) * k5Ld + c4_2;             // element (row 2 b + par + 32 jj, klo) at + 32 jj kLd + 16 klo
            const float2 *const gr = scr + (64 * rh2 + lane2) * k5Ld + 16 * klo2;   // row = 64 rh + lane, klo: 16 consecutive points
            const uint4 *const sow = reinterpret_cast<const uint4 *>(soff + 16 * klo2);
            cf ct[P];
            {
                const float2 *ctr = reinterpret_cast<const float2 *>(fdc_smem_b512 + GM::kOffCt) + c4_2 * P;
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
                    b5_pass_dft<P>(a);
                    float2 *const g = gw + jj * 32 * k5Ld;
                    st2(&g[0], a[0]);
#pragma unroll
                    for (int k = 1; k < P; k++) st2(&g[16 * k], cmul(a[b5_pass_idx<P>(k)], ct[k]));
                }
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();                                      // the trip is in LDS
                const bool active = kJA == kJT || 2 * rh2 < kJA;      // wave-uniform: this wave's 64 rows exist in the trip
                cf v[16];
                if (active) {
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const float4 t = ld4(&gr[2 * i]);
                        v[2 * i] = mk(t.x, t.y); v[2 * i + 1] = mk(t.z, t.w);
                    }
                }
                __syncthreads();                                      // every read of the trip is done
                __builtin_amdgcn_sched_barrier(0);
                if (active) {
                    dft16<false>(v);                                  // khi in v[rev16(khi)]
                    const unsigned rb = (unsigned)(m * kRows + rowbase + 32 * kJA * tr + 64 * rh2 + lane2) * 8u;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const uint4 t = sow[q];
                        const unsigned so[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                        for (int e = 0; e < 4; e++) bst2t<NT>(rout, (so[e] == 0xFFFFFFFFu ? 0xFFFFFFF0u : so[e] + rb), v[4 * q + e]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        stage2([&](int j, int ps) { return unpack512(G[j][ps]); }, R4 ? 128 : 0, std::integral_constant<int, 8>{});
        if constexpr (R4) {
            // m = 64 .. 127 = output rows 0 .. 127: this lane's own stores, served by the L2
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stage2([&](int j, int ps) { return bld2_sc1(rscr, (unsigned)tid * 8u + (unsigned)(j * 4096 + ps * 16384), 0u); }, 0,
                   std::integral_constant<int, 4>{});
        }
    }
}

#ifndef FDC_512_STAGED
#define FDC_512_STAGED 1
#endif

bool poly_block512_supports(int N, int R)
{
    return (N == 65536 || N == 32768 || N == 16384) && (R == 2 || R == 4);
}

hipError_t init_block512_kernels()
{
    hipError_t e = hipSuccess;
#define FDC_SET5(A, B, C, P) \
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_blk512<A, B, C, P>), hipFuncAttributeMaxDynamicSharedMemorySize, C ? B5Geom<P>::kLdsStaged : B5Geom<P>::kLds);
    FDC_SET5(true, false, false, 8) FDC_SET5(false, false, false, 8) FDC_SET5(true, true, false, 8) FDC_SET5(false, true, false, 8)
    FDC_SET5(true, false, true, 8) FDC_SET5(false, false, true, 8) FDC_SET5(true, true, true, 8) FDC_SET5(false, true, true, 8)
    FDC_SET5(true, false, true, 4) FDC_SET5(false, false, true, 4) FDC_SET5(true, true, true, 4) FDC_SET5(false, true, true, 4)
    FDC_SET5(true, false, true, 2) FDC_SET5(false, false, true, 2) FDC_SET5(true, true, true, 2) FDC_SET5(false, true, true, 2)
#undef FDC_SET5
    return e;
}

hipError_t launch_poly_block512(const float2 *in, size_t in_stride, float2 *out, int nb_chunk, int mbase, int nb_call, const float2 *tw256,
                                const float2 *tw512, const float2 *twq, const float2 *cbt, const float *shn, const long long *slot_off,
                                unsigned out_bytes, int ncu, int hints, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop, int R, float2 *scratch,
                                bool half, int N)
{
    if (nb_chunk <= 0) return hipSuccess;
    if (!poly_block512_supports(N, R) || (R == 4 && !scratch)) return hipErrorInvalidValue;
    int grid = ncu > 0 ? ncu : 256;
    if (grid > nb_chunk) grid = nb_chunk;
    constexpr bool kStaged = FDC_512_STAGED != 0;
#define FDC_L512(A, B, S, P) \
    hipExtLaunchKernelGGL((k_blk512<A, B, S, P>), dim3((unsigned)grid), dim3(512), S ? B5Geom<P>::kLdsStaged : B5Geom<P>::kLds, s, ev_start, ev_stop, 0u, in, in_stride, out, \
                          tw256, tw512, twq, cbt, shn, slot_off, (long long)mbase * (B ? 384 : 256), (long long)nb_call, out_bytes, nb_chunk, hints, \
                          B ? scratch : (float2 *)nullptr, half ? 1 : 0)
    const bool nt = (hints & 1) != 0;
    if (N == 65536) {
        if (R == 4) { if (nt) FDC_L512(true, true, kStaged, 8); else FDC_L512(false, true, kStaged, 8); }
        else { if (nt) FDC_L512(true, false, kStaged, 8); else FDC_L512(false, false, kStaged, 8); }
    } else if (N == 32768) {
        if (R == 4) { if (nt) FDC_L512(true, true, true, 4); else FDC_L512(false, true, true, 4); }
        else { if (nt) FDC_L512(true, false, true, 4); else FDC_L512(false, false, true, 4); }
    } else {
        if (R == 4) { if (nt) FDC_L512(true, true, true, 2); else FDC_L512(false, true, true, 2); }
        else { if (nt) FDC_L512(true, false, true, 2); else FDC_L512(false, false, true, 2); }
    }
#undef FDC_L512
    return hipGetLastError();
}

}  // namespace fdc
