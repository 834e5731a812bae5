// gfx950 "one block per CU" form of the uniform-plan path for N = 65536 = 256 x 256, l = 256, R = 2
// (BASELINE configs[1] / [2]): overlap-save gather, forward FFT, window, per-channel IFFT, overlap discard and the
// FFT over the channel slots in ONE kernel, with the intermediate G (fdc_fast256.hip: 128 rows t' x 256 columns n1
// per block = 256 KiB) never leaving the compute unit.
//
// Why: the two-launch form (k_p1 + k_p2) moves G out to memory and back, 2.03x the algorithmic bytes, and both of
// its kernels sit at the copy rate of their own traffic (profiles/r01/NOTES.md); only moving fewer bytes helps.
// G does not fit the 160 KiB of LDS, but it fits the register file: ONE 512-thread workgroup per CU (8 waves, 2 per
// SIMD, 256 VGPRs each) keeps a whole block's G in 128 VGPRs per lane.
//
//   stage 1, 8 passes of 32 columns: every wave owns 4 columns per pass, lane = col + 4*b holds the 16 rows
//       n2 = 16a + b of its column.  Both 16 x 16 exchanges of the FFT-256 / IFFT-256 pair stay inside the wave
//       (a private 8.5 KiB LDS scratch, in-order LDS queue, no s_barrier), so the eight waves drift apart and one
//       wave's LDS phases overlap the other waves' DFT-16 arithmetic.  The next pass's rows are loaded into
//       registers before the current pass is computed.  The 8 kept outputs t = b + 16q, q >= 8, of every pass go
//       into the G registers (indexed by the pass: s_set_gpr_idx).
//   stage 2, 2 chunks of 64 rows t': the FFT-256 over n1 = 32 pass + c5 starts with a DFT-8 over the pass index, which
//       is the register index of G (no exchange), then W_256^(c5 klo), then ONE trip through LDS ([row][klo][c5]) to the
//       lane that owns (row, klo) and transforms the remaining 32 points over c5 in registers.  Slot klo + 8 khi of 64
//       consecutive rows: every wave store is a 512-byte run of one channel.  Two s_barriers per chunk.
// Every LDS access is base register + immediate offset; all layouts are padded (not XOR-swizzled) so that no
// per-element address arithmetic is left, and conflict-free for the lane groups of ds_write_b64 (16 lanes) and
// ds_read_b64 (32 lanes) (MI355X_MICROARCH.md, LDS table).
//
// Input rows are read in 32-byte pieces per wave (4 columns x 8 B); the 8 waves of the workgroup cover 256
// contiguous bytes of each row in the same pass.  Consecutive blocks overlap by half (R = 2): the workgroups of one
// XCD take CONSECUTIVE blocks in the same round, so the shared half is fetched from memory once and served to the
// neighbour from that XCD's L2.
//
// The arithmetic is the uniform-plan commutation of fdc_fast256.hip (same tables, same rounding points), so the
// result matches k_p1 + k_p2 to the last few ulps; parity against the oracle: tests/test_parity_gpu.py.
#include <hip/hip_ext.h>
#include <type_traits>
#include "fdc_kernels.h"
#include "fdc_radix16.hpp"
#include "fdc_devutil.hpp"

namespace fdc {

extern __shared__ __attribute__((aligned(16))) unsigned char fdc_smem_blk[];

typedef unsigned long long u8v __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned long long pack_cf(cf v) { return ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x); }
__device__ __forceinline__ cf unpack_cf(unsigned long long u) { return mk(__uint_as_float((unsigned)u), __uint_as_float((unsigned)(u >> 32))); }
// The SI load/store optimizer would pair the exchange reads into ds_read2_b64, which moves 128 B/clk where
// ds_read_b64 moves 256 (MI355X_MICROARCH.md, LDS table): switched off for this kernel (device pass only).
#if defined(__HIP_DEVICE_COMPILE__)
#define FDC_PLAIN_DS __attribute__((target("no-load-store-opt")))
#else
#define FDC_PLAIN_DS
#endif

// The block length is a template parameter: N = 256 rows x (32 P) columns with P = 2, 4, 8 passes of 32 columns, i.e. N = 16384,
// 32768, 65536; the number of channel slots is the number of columns N1 = 32 P.  Stage 1 does not depend on P except through the row
// pitch and the table sizes (a pass is a pass); G is 16 P registers per lane; stage 2 is a DFT-P over the pass index in registers, one
// trip through LDS and the same DFT-32 over c5.  With P < 8 a trip holds all 128 rows of a run (P = 8: 64) and a wave reads
// (klo = wave mod P, row half = wave div P).
//
// LDS map (bytes).  Stage-1 scratch: per wave 68*15 + 64 = 1084 points (element (p; lane) at lane + 68 p).
constexpr int kBlkScrPts = 1084;
constexpr int kBlkOffX = 8 * kBlkScrPts * 8;                      // 69376: end of the stage-1 strips
template <int P>
struct BlkGeom {
    static_assert(P == 2 || P == 4 || P == 8, "passes of 32 columns: N = 16384, 32768 or 65536");
    static constexpr int kN1 = 32 * P;                            // columns = channel slots
    static constexpr int kN = 256 * kN1;
    static constexpr int kJT = P == 8 ? 4 : 8;                    // 16-row groups per stage-2 trip (64 or 128 rows)
    static constexpr int kJB = P == 8 ? 2 : P == 4 ? 4 : 8;       // 16-row groups within reach of one ds base register (16-bit byte offset)
    // stage-2 trip: [rows][32 P + 6] (P klo x 32 c5 + 6).  The row stride in dwords is 12 mod 64 for every P: the 16-lane groups of
    // ds_read_b128 (lanes {0-3,12-15,20-27}, ...; MI355X_MICROARCH.md, LDS table) then cover all 64 banks, and the 16 contiguous lanes
    // of a ds_write_b64 group (4 rows x 4 columns) all 32 (a stride of 8 mod 64 is conflict-free for the stores only: 1.46 M conflict
    // cycles at P = 8)
    // P = 2: 32 P + 2 (4 dwords mod 64: as clean for the reads), which brings the workgroup under half of the LDS: two workgroups per CU
    static constexpr int kLd = P == 2 ? 32 * P + 2 : 32 * P + 6;
    static constexpr int kTripBytes = 16 * kJT * kLd * 8;         // P = 8: 134144
    static constexpr int kOffCt = kTripBytes > kBlkOffX ? kTripBytes : kBlkOffX;   // stage-2 twiddles [32][P], behind the trip buffer and the strips
    static constexpr int kOffWrow = P == 8 ? 136960 : kOffCt + 32 * P * 8;   // tables: above the strips and the trip buffer
    static constexpr int kOffB = kOffWrow + 16 * 18 * 8;
    static constexpr int kOffSA = kOffB + 32 * 18 * 8;
    static constexpr int kOffSoff = kOffSA + 16 * P * 18 * 8;
    static constexpr int kLds = kOffSoff + kN1 * 4;               // P = 8: 163328 <= 163840
    static constexpr int kOffSoffOff = kOffSA + 16 * P * 16 * 8;  // offset plans: unpadded SA rows, then soff, then wrowF
    static constexpr int kOffWrowF = kOffSoffOff + kN1 * 4;
    static constexpr int kLdsOff = kOffWrowF + 16 * 18 * 8;       // P = 8: 163584 <= 163840
    static_assert(kLdsOff <= 160 * 1024 && kLds <= 160 * 1024, "LDS budget");
    // STG (P = 8): the pass's 32 columns x 256 rows staged in LDS as [row][34] (rows 68 dwords apart: the sixteen lanes of a 16-byte store cover a row's
    // 64 banks, the sixteen row groups x four columns of an 8-byte read fall on every bank twice — the minimum for 512 bytes), behind the strips;
    // the tables move up by 2 KiB and the SA rows give up their padding for it
    static constexpr int kPlLd = 34;
    static constexpr int kOffPl = kBlkOffX;
    static constexpr int kPlBytes = 256 * kPlLd * 8;              // 69632
    static constexpr int kOffWrowS = kOffPl + kPlBytes;           // 139008
    static constexpr int kOffBS = kOffWrowS + 16 * 18 * 8;
    static constexpr int kOffSAS = kOffBS + 32 * 18 * 8;
    static constexpr int kOffSoffS = kOffSAS + 16 * P * 16 * 8;
    static constexpr int kLdsS = kOffSoffS + kN1 * 4;             // P = 8: 163328
    static_assert(P != 8 || kLdsS <= 160 * 1024, "LDS budget (staged)");
    static_assert(kOffCt >= kBlkOffX && kOffCt + 32 * P * 8 <= kOffWrow, "stage-2 trip buffer and twiddles below the tables, tables above the strips");
    static_assert((16 * (kJB - 1) * kLd + 32 * (P - 1)) * 8 < 65536, "ds offsets of a base register");
};

// FWD = false: the channelizer (above).  FWD = true: the same machinery as a plain forward transform of the block (no window,
// no inverse transform).  Stage 1 stops after the forward FFT-256 of a column: T[k2][n1] = A[k2] W_N^(n1 k2) / N, 512 KiB per block —
// twice what the G registers hold.  Rounds 2-5 kept the half k2 < 128 in G and sent the other half through 256 KiB of per-workgroup
// scratch: the counters (profiles/r06/pmc_summary_fwd.txt) say that trip is real traffic — WRITE 807 MB + FETCH 560 MB per 1024 blocks
// for 805 MB of work, at 5.5 TB/s: the kernel was bound by the memory system on bytes of which 39 % were its own scratch.
// Round 6 built the alternative (-DFDC_FWD_TWO_WG=1; not shipped, it measured 5 % slower — see the macro below): TWO WORKGROUPS PER BLOCK,
// no scratch.  A work item is (block, half h): the workgroup computes the columns' forward
// FFT-256 for the rows k2 = b + 16 q with q = 2 r + h only — bit 4 of k2, so that a half is every other 16-bin run = every other
// 128-byte line of the spectrum: both halves store whole lines — 128 rows, exactly G.  The split is a decimation in frequency of
// the SECOND DFT-16 (over the exchanged index bb): q even: DFT-8 of v[bb] + v[bb + 8]; q odd: DFT-8 of (v[bb] - v[bb + 8]) W_16^bb;
// first DFT-16, twiddle and exchange are done in full by both halves (stage-1 arithmetic per half: 0.7 of the whole).  The two
// halves of a block are neighbouring slots of ONE XCD in the same round (h = slot & 1 is a constant of the workgroup): the input
// rows are fetched from memory once and the second reader hits that XCD's L2, as the overlap half of the next block always did.
// Stage 2 is unchanged and runs once per item; its "slots" are the k1 of the spectrum: bins 256 c + k2 of the SHIFTED spectrum (the
// (-1)^n1 of cbt moves k1 by 128 = fftshift); a wave store is four 16-bin runs 32 bins apart.  This is what plans that need a
// spectrum in memory (mixed channel plans, the sinks, the debug port) use instead of two passes through a scratch of the whole batch.
// R4 = true: the channelizer at relinvovl = 4 (the reference's default overlap, grc/FDC_FrequencyDomainChannelizer.xml:61): three
// quarters of every inverse transform are kept, G is 192 rows x N1 columns.  The rows t >= 128 stay in the G registers
// as for R = 2; the rows 64 <= t < 128 take the route of the forward-transform variant: per-workgroup scratch (L2),
// read back for a third, 64-row run of stage 2.  Off the grid (OFF) the window phase counter runs: a constant j^p per block in cb.

// DFT over the pass index (the register index of G): P points in place; the result X[k] is read through blk_pass_idx<P>(k)
template <int P> __device__ __forceinline__ constexpr int blk_pass_idx(int k) { return P == 8 ? 4 * (k & 1) + (k >> 1) : k; }
template <int P>
__device__ __forceinline__ void blk_pass_dft(cf (&a)[P])
{
    if constexpr (P == 8) dft8<false>(a);                          // klo = k0 + 2 k1 in a[4 k0 + k1]
    else if constexpr (P == 4) {                                   // natural order
        dft4<false>(a[0], a[1], a[2], a[3]);
    } else {
        const cf s0 = a[0] + a[1], d0 = a[0] - a[1];
        a[0] = s0; a[1] = d0;
    }
}

// STG = true (P = 8, the plain channelizer): the input rows reach the lanes through LDS.  A wave's own load instruction is 16 rows x 32 bytes (16 cache
// lines, each of which four waves ask for); staged, wave w fetches rows 32 w .. of the pass's 32 columns in 16-byte pieces of whole 256-byte row
// segments (4 rows per instruction: 8 lines, half the instructions: a quarter of the line requests), keeps them in registers for a pass as before, writes
// them to [row][34] in LDS at the pass boundary and reads its own column's sixteen rows back: two workgroup barriers per pass (profiles/r04/NOTES.md
// section 11; the same move took k_blk1024 from 0.20 to 0.30; here it costs 1.5 %: a build variant, -DFDC_BLK_STAGED=1).
// HALF = true: every channel half a slot higher (f = 256 slot + 128: a bank centred on multiples of 256 bins) WITHOUT the offset machinery.  The block
// modulated by exp(-2 pi i 128 n / N) = W_N^(128 n1) (-1)^n2: the (-1)^n2 moves every column's spectrum by half its length, which together with the
// ifftshift of the inverse is the identity — the value stays in its register, the tables are read at k2 ^ 128 (q ^ 8), W_N^(128 n1) is in cbt (the host
// builds it for r = 128 as for any offset).  No second twiddle table, no rotated exchange, both row sets: the on-grid rate, and relinvovl 4 too
// (128 mod 4 = 0: the window phase stays 0).
// -DFDC_FWD_TWO_WG=1 builds the forward-transform variant as TWO WORKGROUPS PER BLOCK without scratch (round 6, tools/build_variant.sh); the
// shipped form is the round-2..5 one (one workgroup per block, the half k2 >= 128 through per-workgroup scratch).  Measured on one box,
// alternating (profiles/r06/fwd_ab.txt): full band 0.2525 against 0.2405 ms per 1024 blocks, nothing written 0.2017 against 0.188 — the
// two-workgroup form moves 805 MB instead of 1367 and is SLOWER: its passes are a third of the channelizer's (no inverse transform), so the
// rows requested one pass ahead are not there when the pass starts, and it asks for every row twice (16 cache lines per load instruction).
#ifndef FDC_FWD_TWO_WG
#define FDC_FWD_TWO_WG 0
#endif
template <int P, bool NT, bool OFF, bool FWD, bool R4 = false, bool STG = false, bool HALF = false>
__global__ FDC_PLAIN_DS __launch_bounds__(512, (P == 2 && !OFF && !R4 && !FWD) ? 4 : 2) void k_blk256(const float2 *__restrict__ in, size_t in_stride, float2 *__restrict__ out,
                                                const float2 *__restrict__ tw256, const float2 *__restrict__ twq,
                                                const float2 *__restrict__ cbt, const float *__restrict__ shn,
                                                const long long *__restrict__ slot_off, long long out_base,
                                                long long nb_call, unsigned out_bytes, int nb, int hints,
                                                unsigned long long *__restrict__ dbg, int roff, long long first_block,
                                                float2 *__restrict__ fwd_scratch, const unsigned *__restrict__ keep,
                                                float *__restrict__ gpow)
{
    typedef BlkGeom<P> GM;
    static_assert(!STG || (P == 8 && !OFF && !R4), "staged loads: the plain channelizer and the forward transform at N = 65536");
    static_assert(!HALF || (!OFF && !FWD && !STG), "the half-slot form is a variant of the on-grid channelizer");
    constexpr int kN1 = GM::kN1, kLd = GM::kLd, kJT = GM::kJT, kJB = GM::kJB;
    constexpr bool FW2 = FWD && FDC_FWD_TWO_WG;                                  // forward transform, two workgroups per block
    float2 *scr = reinterpret_cast<float2 *>(fdc_smem_blk);                     // stage 1: 8 wave scratches; stage 2: the trip buffer
    float2 *wrow = reinterpret_cast<float2 *>(fdc_smem_blk + (STG ? GM::kOffWrowS : GM::kOffWrow));     // [b][p] = W256^(b p), rows of 18
    float2 *Bt = reinterpret_cast<float2 *>(fdc_smem_blk + (STG ? GM::kOffBS : GM::kOffB));             // [c5][q] = W_N^(16 c5 q)
    float2 *SA = reinterpret_cast<float2 *>(fdc_smem_blk + (STG ? GM::kOffSAS : GM::kOffSA));           // [pass][b][q] = shape[b+16q]/N * W_N^(512 pass q)
    // offset plans need a second twiddle table, staged loads 2 KiB more for the rows: the SA rows give up their padding (2-way conflicts on 8 reads per pass)
    constexpr int kSaLd = (OFF || STG) ? 16 : 18;
    unsigned *soff = reinterpret_cast<unsigned *>(fdc_smem_blk + (STG ? GM::kOffSoffS : OFF ? GM::kOffSoffOff : GM::kOffSoff));
    float2 *ctab = reinterpret_cast<float2 *>(fdc_smem_blk + GM::kOffCt);       // [c5][klo] = W_N1^(c5 klo): stage 2, after the DFT-P
    const int tid = threadIdx.x;
    // stage-1 roles
    const int w = tid >> 6, lane = tid & 63, col = lane & 3, b = lane >> 2, c5 = 4 * w + col;
    // stage-2 roles: writer = the stage-1 role (column c5, rows b + 16 j); reader: row = lane (+ 64 row half), klo = wave mod P
    // FWD with a plan that reads part of the spectrum only: which of this wave's 64-bin stores some channel reads at all
    // ([klo = wave mod P][k2 / 64], bit = register index of the slot); the others are dropped (offset beyond the descriptor's extent)
    unsigned mqs[4] = {~0u, ~0u, ~0u, ~0u};
    if (FWD && keep) {
        const int wu = __builtin_amdgcn_readfirstlane(w) % P;
#pragma unroll
        for (int q = 0; q < 4; q++) mqs[q] = keep[wu * 4 + q];
    }

    // block order: round rho, XCD x = workgroup mod 8 (round-robin dispatch), slot = workgroup / 8:
    // block = rho*grid + x*(grid/8) + slot, i.e. one XCD works on grid/8 consecutive blocks at a time
    const int grid = gridDim.x, per = grid >> 3;
    // FWD: the work items are (block, half); neighbouring slots of an XCD share a block (the launcher makes the grid even), so a round is
    // grid / 2 blocks and an XCD still walks a contiguous run of them
    const bool xmap = FW2 ? (grid & 15) == 0 : (grid & 7) == 0;
    const int fhalf = FW2 ? (xmap ? (int)(blockIdx.x >> 3) & 1 : (int)blockIdx.x & 1) : 0;      // which rows this workgroup computes: k2 bit 4
    const int mstride = FW2 ? grid >> 1 : grid;
    const int first = FW2 ? (xmap ? (int)(blockIdx.x & 7) * (per >> 1) + (int)(blockIdx.x >> 4) : (int)(blockIdx.x >> 1))
                          : (xmap ? (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3) : (int)blockIdx.x);
    if (first >= nb) return;
#ifdef FDC_BLK_WGTIMES
    // diagnostics (tools/wg_times.py): when every workgroup starts and ends, 100 MHz clock: [0..255] starts, [256..511] ends
    if (dbg && tid == 0) dbg[blockIdx.x] = wall_clock64();
#endif
#ifdef FDC_BLK_STAGGER
    // experiment: half of the workgroups (by slot parity inside their XCD) start FDC_BLK_STAGGER x 8128 cycles late, so that the
    // store bursts of one half meet the load phases of the other
    if ((blockIdx.x >> 3) & 1) for (int i = 0; i < FDC_BLK_STAGGER; i++) __builtin_amdgcn_s_sleep(127);
#endif

    constexpr unsigned inbytes = (unsigned)GM::kN * 8u;
    constexpr unsigned kRowGrp = (unsigned)kN1 * 128u;            // 16 rows of kN1 columns, bytes (P = 8: 32 KiB)
    const unsigned voff = (unsigned)(b * kN1 + c5) * 8u;          // row b, column c5 of pass 0; pass adds 256 B, row group a 16 rows
    // the first block's rows are requested before the tables are built: their latency hides behind the table set-up
    const __amdgpu_buffer_rsrc_t rcb = make_rsrc(cbt, (unsigned)kN1 * 16u * 8u);      // cbt[n1][b], n1 = 32 pass + c5
    const unsigned voffc = (unsigned)(c5 * 16 + b) * 8u;
    // two row sets: a pass computes on one while the rows of the next pass arrive in the other (no register copies between passes)
    cf LA[16], LB[16], cbA, cbB;
    // staged: wave w fetches rows 32 w + 4 i + (lane >> 4), columns 2 (lane & 15), + 1 of the pass (16 bytes); instruction i adds 4 rows = 8 KiB
    [[maybe_unused]] float2 *pl = reinterpret_cast<float2 *>(fdc_smem_blk + GM::kOffPl);
    [[maybe_unused]] const unsigned voffs = (unsigned)((32 * w + (lane >> 4)) * kN1 + 2 * (lane & 15)) * 8u;
    [[maybe_unused]] float2 *const plw = pl + (32 * w + (lane >> 4)) * GM::kPlLd + 2 * (lane & 15);      // + 4 i rows
    [[maybe_unused]] const float2 *const plr = pl + b * GM::kPlLd + c5;                                   // this lane's rows 16 a + b of column c5
    [[maybe_unused]] u32x4 PF[8];
    [[maybe_unused]] auto stage_load = [&](int mb, int pn) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rin = make_rsrc(in + (size_t)mb * in_stride + 32 * pn, inbytes);
#pragma unroll
        for (int i = 0; i < 8; i++) PF[i] = bld4(rin, voffs, (unsigned)i * 4u * (unsigned)kN1 * 8u);
    };
    if constexpr (STG) {
        stage_load(first, 0);
        cbA = bld2(rcb, voffc, 0);
    } else {
        const __amdgpu_buffer_rsrc_t rin = make_rsrc(in + (size_t)first * in_stride, inbytes);
#pragma unroll
        for (int a = 0; a < 16; a++) LA[a] = bld2(rin, voff, (unsigned)a * kRowGrp);
        cbA = bld2(rcb, voffc, 0);
    }
    // ---- tables (once per workgroup; the workgroup is persistent)
    // Offset plans (every channel at f = 256*slot + r, OFF): the block is modulated by exp(-2 pi i r n / N), n = n1 + N1 (16a + b),
    // without a single extra multiplication.  W_16^(r a) rotates the outputs of the first DFT-16 (index p reads Z[(p + r) mod 16]:
    // the exchange slot of register Z[p] becomes (p - r) mod 16), W_256^(r b) joins the forward twiddle (table wrowF), W_N^(r n1)
    // sits in cbt (host), and for odd r the window phase (-1)^block (phase_shifting_windowing_vcc_impl.cc:82, R = 2) in cb.
    float2 *wrowF = OFF ? reinterpret_cast<float2 *>(fdc_smem_blk + GM::kOffWrowF) : wrow;
    const int r16 = roff & 15;
    for (int i = tid; i < 256; i += 512) {
        wrow[(i >> 4) * 18 + (i & 15)] = tw256[((i >> 4) * (i & 15)) & 255];
        if (OFF)     // entry [b][p]: the twiddle of register Z[p], whose true index is pt = (p - r) mod 16: W_256^(b (pt + r))
            wrowF[(i >> 4) * 18 + (i & 15)] = tw256[((i >> 4) * ((((i & 15) - r16) & 15) + roff)) & 255];
    }
    for (int i = tid; i < kN1; i += 512) {
        const long long o = slot_off[i];                  // slot i = klo + P khi, khi = k0 + 2 k1, is entry [klo][16 k0 + rev16(k1)]
        soff[(i % P) * 32 + ((i / P) & 1) * 16 + rev16((i / P) >> 1)] = o >= 0 ? (unsigned)((o * nb_call + out_base) * 8) : 0xFFFFFFFFu;
    }
    for (int i = tid; i < 32 * P; i += 512) ctab[i] = tw256[((i / P) * (i % P) * (8 / P)) & 255];     // [c5][klo] = W_N1^(c5 klo)
    // FWD: entry r < 8 of a row is the one of q = 2 r + h (the rows this workgroup computes); entries 8 .. 15 are not read
    Bt[(tid >> 4) * 18 + (tid & 15)] = twq[HALF ? tid ^ 8 : FW2 ? (tid & ~15) + ((2 * (tid & 15) + fhalf) & 15) : tid];   // c5 = tid >> 4 < 32, q = tid & 15 (HALF: the entry of q ^ 8)
    for (int i = tid; i < 256 * P; i += 512) {
        const int ps = i >> 8, bb = (i >> 4) & 15, q = i & 15, qt = HALF ? q ^ 8 : FW2 ? (2 * q + fhalf) & 15 : q;
        const float2 t = twq[(size_t)(32 * ps) * 16 + qt];                       // W_N^(16 * 32 ps * q)
        const float s = shn[bb + 16 * qt];
        SA[(ps * 16 + bb) * kSaLd + q] = make_float2(t.x * s, t.y * s);
    }
    __syncthreads();

    float2 *const scrw = scr + w * kBlkScrPts + lane;             // exchange write base: element p at + 68 p
    const float2 *const scrr = scr + w * kBlkScrPts + col + 68 * b;   // exchange read base: element bb at + 4 bb
    const float2 *const wr = wrow + b * 18;
    const float2 *const wrf = wrowF + b * 18;
    const float2 *const btr = Bt + c5 * 18;
    const __amdgpu_buffer_rsrc_t rout = make_rsrc(out, out_bytes);
    // FWD / R4: this workgroup's scratch, [pass][j][thread]
    constexpr bool kScr = (FWD && !FW2) || R4;
    const __amdgpu_buffer_rsrc_t rscr = make_rsrc(kScr ? fwd_scratch + (size_t)blockIdx.x * 32768 : fwd_scratch, kScr ? 32768u * 8u : 0u);

    // Two waves share a SIMD (waves w and w + 4).  The older one wins the issue arbitration and finishes stage 1 ~10 k cycles
    // earlier; s_setprio (either half favoured, or alternating per pass) changes nothing about that (profiles/r02/NOTES.md).

    // diagnostics (FDC_BLOCK_DEBUG=1): cycle stamps of workgroup 0, per wave: [wave][block round][24]
    [[maybe_unused]] int dbgk = 0;
    // (only in the -DFDC_BLK_STAMPS build, tools/block_probe.py: the stamps cost half a dozen registers the kernel does not have)
#ifdef FDC_BLK_STAMPS
    // the stamps are taken into scalar registers (s_memtime) and written out once per block: no vector register is held
#define FDC_STAMP(i) do { st[i] = __builtin_readcyclecounter(); } while (0)
#else
#define FDC_STAMP(i) do { } while (0)
#endif
#ifdef FDC_BLK_L2PF
    unsigned pfd = 0;
#endif
    for (int m = first; m < nb; m += mstride) {
        const int mnext = m + mstride < nb ? m + mstride : m;
#ifdef FDC_BLK_STAMPS
        unsigned long long st[32] = {};
#endif
        const float sgn = (OFF && !R4 && (roff & 1) && ((first_block + m) & 1)) ? -1.0f : 1.0f;
        // R = 4 off the grid: the window phase counter (phase_shifting_windowing_vcc_impl.cc:82) is (block * (f mod 4)) mod 4, and phase p of the window
        // table is the window times exp(2 pi i p / 4) = j^p: one constant per block
        cf phs = mk(1.f, 0.f);
        if constexpr (OFF && R4) {
            const int pc = (int)((((first_block + m) & 3) * (roff & 3)) & 3);
            phs = mk(pc == 0 ? 1.f : pc == 2 ? -1.f : 0.f, pc == 1 ? 1.f : pc == 3 ? -1.f : 0.f);
        }
        FDC_STAMP(0);
        // G[j][pass]: row t' = b + 16 j, column 32 pass + c5.  One complex value = one 64-bit vector element (two floats packed
        // into an integer): the element index is the pass number at run time, and with 64-bit elements the compiler brackets
        // all sixteen moves of a pass with ONE s_set_gpr_idx_on / off pair (a pair per dword with 32-bit elements).  Integer,
        // not double, elements: bit-casting an extracted double to two floats read element 0 for every pass (seen in the ISA).
        typedef unsigned long long gvec __attribute__((ext_vector_type(P)));
        gvec G[8];
#define FDC_GGET(j, ps) unpack_cf(G[j][ps])
#define FDC_GPUT(j, ps, val) G[j][ps] = pack_cf(val)
        // ---------------- stage 1 ----------------
        // One pass per trip.  The 16 rows of this lane's column were requested a whole pass ago into L; the rows of the next
        // pass (of this block, or pass 0 of this workgroup's next block; after the last block: the same rows again, unused)
        // are requested first, unconditionally (a conditional request costs a second set of register copies).
        auto one_pass = [&](const int ps, cf (&cur)[16], const cf cbc, cf (&L)[16], cf &cbn) __attribute__((always_inline)) {
#ifdef FDC_BLK_L2PF
            // experiment: touch every 128-byte line of the pass after next (this block's, or the next block's) once, two passes ahead
            {
                asm volatile("" :: "v"(pfd));
                const int p2 = (ps + 2) & (P - 1);
                const int mb2 = ps < P - 2 ? m : mnext;
                const __amdgpu_buffer_rsrc_t rpf = make_rsrc(in + (size_t)mb2 * in_stride + 32 * p2, inbytes);
                pfd = __builtin_amdgcn_raw_buffer_load_b32(rpf, (unsigned)((tid >> 1) * (kN1 * 8) + (tid & 1) * 128), 0u, 0);
            }
#endif
            const cf cb = (OFF && R4) ? cmul(cbc, phs) : OFF ? cbc * sgn : cbc;
            if constexpr (STG) {
                // the rows of this pass arrived in PF during the pass before: into the planes, this lane's sixteen back out, then the next pass's
                // rows (of this block, or pass 0 of this workgroup's next block, kept in PF across stage 2) are requested
#pragma unroll
                for (int i = 0; i < 8; i++) *reinterpret_cast<u32x4 *>(&plw[4 * i * GM::kPlLd]) = PF[i];
                __syncthreads();
#pragma unroll
                for (int a = 0; a < 16; a++) cur[a] = ld2(&plr[16 * a * GM::kPlLd]);
                __syncthreads();                                  // every lane has its rows: the planes may be rewritten
                stage_load(ps < P - 1 ? m : mnext, ps < P - 1 ? ps + 1 : 0);
                cbn = bld2(rcb, voffc, (unsigned)(ps < P - 1 ? ps + 1 : 0) * 4096u);
                if (ps == P - 1) {
                    // stage 2's twiddles live where the planes' last rows were: rebuilt for every block (stage 2 reads them behind its first barrier)
                    for (int i = tid; i < 32 * P; i += 512) ctab[i] = tw256[((i / P) * (i % P) * (8 / P)) & 255];
                }
            } else {
                const int pn = ps < P - 1 ? ps + 1 : 0;
                const int mb = ps < P - 1 ? m : mnext;
                // the pass offset (32 columns) sits in the descriptor's base: every pass uses the same per-lane offset and the
                // same 16 scalar row offsets
                const __amdgpu_buffer_rsrc_t rin = make_rsrc(in + (size_t)mb * in_stride + 32 * pn, inbytes);
                if (hints & 2) {
#pragma unroll
#ifdef FDC_BLK_SC1LOADS
                    for (int a = 0; a < 16; a++) L[a] = bld2_sc1(rin, voff, (unsigned)a * kRowGrp);
#else
                    for (int a = 0; a < 16; a++) L[a] = bld2_nt(rin, voff, (unsigned)a * kRowGrp);
#endif
                } else {
#pragma unroll
                    for (int a = 0; a < 16; a++) L[a] = bld2(rin, voff, (unsigned)a * kRowGrp);
                }
                cbn = bld2(rcb, voffc, (unsigned)pn * 4096u);
            }
            dft16<false>(cur);                                    // in place, over a: index p in cur[rev16(p)]
            cf tw[16];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const float4 t = ld4(&wrf[2 * i]);
                tw[2 * i] = mk(t.x, t.y); tw[2 * i + 1] = mk(t.z, t.w);
            }
            if (OFF) {
#pragma unroll
                for (int p = 0; p < 16; p++) st2(&scrw[68 * ((p - r16) & 15)], cmul(cur[rev16(p)], tw[p]));
            } else {
                st2(&scrw[0], cur[rev16(0)]);                     // W256^0 = 1
#pragma unroll
                for (int p = 1; p < 16; p++) st2(&scrw[68 * p], cmul(cur[rev16(p)], tw[p]));
            }
            __builtin_amdgcn_wave_barrier();                      // same wave, in-order LDS queue: no s_barrier
            cf v[16];
#pragma unroll
            for (int bb = 0; bb < 16; bb++) v[bb] = ld2(&scrr[4 * bb]);
            if constexpr (FWD && !FW2) {
                // round-2..5 form: T[k2 = b + 16 q][n1] = A[k2] W_N^(n1 k2) / N.  The half q < 8 stays in the G registers, the half q >= 8 goes to this
                // workgroup's 256 KiB of scratch ([pass][j][thread]: 512-byte wave stores) and comes back for the second run of stage 2
                dft16<false>(v);
                const float2 *sar = SA + (ps * 16 + b) * kSaLd;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const float4 t0 = ld4(&btr[2 * i]), t1 = ld4(&sar[2 * i]);
                    const cf y0 = cmul(cmul(cmul(v[rev16(2 * i)], mk(t0.x, t0.y)), mk(t1.x, t1.y)), cb);
                    const cf y1 = cmul(cmul(cmul(v[rev16(2 * i + 1)], mk(t0.z, t0.w)), mk(t1.z, t1.w)), cb);
                    if (i < 4) { FDC_GPUT(2 * i, ps, y0); FDC_GPUT(2 * i + 1, ps, y1); }
                    else {
                        bst2(rscr, (unsigned)tid * 8u + (unsigned)(2 * i - 8) * 4096u, (unsigned)ps * 32768u, y0);
                        bst2(rscr, (unsigned)tid * 8u + (unsigned)(2 * i - 7) * 4096u, (unsigned)ps * 32768u, y1);
                    }
                }
            } else if constexpr (FW2) {
                // forward only: T[k2 = b + 16 q][n1] = A[k2] W_N^(n1 k2) / N for this workgroup's rows q = 2 r + h: the second DFT-16 decimated
                // in frequency — a fold of v[bb] with v[bb + 8] and ONE DFT-8 (h is uniform: a scalar branch)
                cf e[8];
                if (fhalf) {
                    e[0] = v[0] - v[8];
                    e[1] = mul_w16<false, 1>(v[1] - v[9]);
                    e[2] = mul_w16<false, 2>(v[2] - v[10]);
                    e[3] = mul_w16<false, 3>(v[3] - v[11]);
                    e[4] = mul_w16<false, 4>(v[4] - v[12]);
                    e[5] = mul_w16<false, 5>(v[5] - v[13]);
                    e[6] = mul_w16<false, 6>(v[6] - v[14]);
                    e[7] = mul_w16<false, 7>(v[7] - v[15]);
                } else {
#pragma unroll
                    for (int i = 0; i < 8; i++) e[i] = v[i] + v[i + 8];
                }
                dft8<false>(e);                                   // A[b + 16 (2 r + h)], r = k0 + 2 k1, in e[4 k0 + k1]
                const float2 *sar = SA + (ps * 16 + b) * kSaLd;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float4 t0 = ld4(&btr[2 * i]), t1 = ld4(&sar[2 * i]);
                    constexpr int kRev8[8] = {0, 4, 1, 5, 2, 6, 3, 7};          // r -> 4 (r & 1) + (r >> 1)
                    FDC_GPUT(2 * i, ps, cmul(cmul(cmul(e[kRev8[2 * i]], mk(t0.x, t0.y)), mk(t1.x, t1.y)), cb));
                    FDC_GPUT(2 * i + 1, ps, cmul(cmul(cmul(e[kRev8[2 * i + 1]], mk(t0.z, t0.w)), mk(t1.z, t1.w)), cb));
                }
            } else {
                dft16<false>(v);                                  // A[k2 = b + 16 q] in v[rev16(q)]
                cf u[16];
                {
                    const float2 *sar = SA + (ps * 16 + b) * kSaLd;
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const float4 t0 = ld4(&btr[2 * i]), t1 = ld4(&sar[2 * i]);
                        // window * inter-pass twiddle, placed at the ifftshifted position (k2 ^ 128 <=> q ^ 8)
                        u[HALF ? 2 * i : (2 * i) ^ 8] = cmul(cmul(v[rev16(2 * i)], mk(t0.x, t0.y)), mk(t1.x, t1.y));
                        u[HALF ? 2 * i + 1 : (2 * i + 1) ^ 8] = cmul(cmul(v[rev16(2 * i + 1)], mk(t0.z, t0.w)), mk(t1.z, t1.w));
                    }
                }
                dft16<true>(u);
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const float4 t = ld4(&wr[2 * i]);
                    tw[2 * i] = mk(t.x, t.y); tw[2 * i + 1] = mk(t.z, t.w);
                }
                u[rev16(0)] = cmul(u[rev16(0)], cb);
#pragma unroll
                for (int p = 1; p < 16; p++) u[rev16(p)] = cmul(cmulc(u[rev16(p)], tw[p]), cb);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int p = 0; p < 16; p++) st2(&scrw[68 * p], u[rev16(p)]);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int bb = 0; bb < 16; bb++) u[bb] = ld2(&scrr[4 * bb]);
                dft16<true>(u);                                       // y[t = b + 16 q] in u[rev16(q)]; keep q >= 8 (R = 2)
#pragma unroll
                for (int j = 0; j < 8; j++) FDC_GPUT(j, ps, u[rev16(8 + j)]);
                if constexpr (R4) {                                   // R = 4 keeps q >= 4: rows 64..127 go to the scratch, [pass][q - 4][thread]
#pragma unroll
                    for (int j = 0; j < 4; j++) bst2(rscr, (unsigned)tid * 8u + (unsigned)j * 4096u, (unsigned)ps * 16384u, u[rev16(4 + j)]);
                }
            }
            FDC_STAMP(1 + ps);
        };
        // two passes per trip: the row sets swap roles (the pass count is even for every P).  The offset-plan variant at P = 8 has no
        // registers for the second set's live range (32 bytes of scratch): it copies the rows at the top of a pass as before.
#ifndef FDC_BLK_COPYFORM
#define FDC_BLK_COPYFORM 0
#endif
        if constexpr ((OFF || (FDC_BLK_COPYFORM && !STG)) && P == 8) {
#pragma nounroll
            for (int ps = 0; ps < P; ps++) {
                cf cur[16];
#pragma unroll
                for (int a = 0; a < 16; a++) cur[a] = LA[a];
                one_pass(ps, cur, cbA, LA, cbA);
            }
        } else {
#pragma nounroll
            for (int pp = 0; pp < P; pp += 2) {
                one_pass(pp, LA, cbA, LB, cbB);
                one_pass(pp + 1, LB, cbB, LA, cbA);
            }
        }
        // ---------------- stage 2 ----------------
        // FFT-N1 over n1 = 32 pass + c5 of every row t' = b + 16 j.  The P passes of a column sit in ONE lane: a DFT-P over
        // the pass index needs no exchange at all (k1 = klo + P khi: W_N1^(n1 k1) = W_P^(pass klo) W_N1^(c5 klo) W_32^(c5 khi)).
        // What is left is a DFT-32 over c5 = 4 wave + col, i.e. across the whole workgroup: ONE trip through LDS per value
        // (trips of 16 kJT rows: [row][klo][c5], rows kLd apart), read back as whole 32-point runs by lane = row (+ 64 for the
        // upper row half when P < 8), klo = wave mod P, transformed in registers.  A wave's store is 64 consecutive samples of one
        // channel (512 B).
        // rowbase: first output row (sample index inside the block's lout rows, or bin offset 128 h of a forward transform) of the
        // run; njc: its number of 16-row groups (8 = the 128 rows in the G registers, 4 = a 64-row run from the scratch)
        auto stage2 = [&](auto get, const int rowbase, auto njc) __attribute__((always_inline)) {
            constexpr int kNJ = decltype(njc)::value;
            constexpr int kNTrip = (kNJ + kJT - 1) / kJT;
            __syncthreads();                                          // every wave is done with its stage-1 scratch
            FDC_STAMP(9);
            // the stage-2 roles are worked out here, from a thread index the compiler cannot trace back: loop-invariant address
            // registers would otherwise stay live across stage 1, which has none to spare
            int t2 = tid;
            asm volatile("" : "+v"(t2));
            const int lane2 = t2 & 63, w2 = __builtin_amdgcn_readfirstlane(t2 >> 6), b_2 = lane2 >> 2, c5_2 = 4 * w2 + (lane2 & 3);
            const int klo2 = w2 % P, rh2 = w2 / P;                    // reader: klo, row half (0 for P = 8)
            float2 *const gw0 = scr + b_2 * kLd + c5_2;               // element (row b + 16 jj, klo) at + 16 jj * kLd + 32 klo
            // rows beyond 16 kJB are out of reach of the 16-bit ds offset from gw0: a second base, opaque to the constant folder (it
            // would otherwise materialise one address register per write)
            int rowjb = 16 * kJB * kLd;
            asm volatile("" : "+v"(rowjb));
            float2 *const gw1 = gw0 + rowjb;
            const float2 *const gr = scr + (64 * rh2 + lane2) * kLd + 32 * klo2;   // 32 consecutive points of one row
            const uint4 *const sow = reinterpret_cast<const uint4 *>(soff + 32 * klo2);
#pragma unroll
            for (int tr = 0; tr < kNTrip; tr++) {
                const int ja = kNJ - kJT * tr < kJT ? kNJ - kJT * tr : kJT;   // 16-row groups of this trip (compile time after unrolling)
                cf ct[P];                                             // W_N1^(c5 klo): read per trip, not held across the DFT-32 phase
                {
                    const float2 *ctr = reinterpret_cast<const float2 *>(fdc_smem_blk + GM::kOffCt) + c5_2 * P;
#pragma unroll
                    for (int i = 0; i < P / 2; i++) {
                        const float4 t = ld4(&ctr[2 * i]);
                        ct[2 * i] = mk(t.x, t.y); ct[2 * i + 1] = mk(t.z, t.w);
                    }
                }
                // the values of this trip: register reads, or (runs that come back from the scratch) all loads in flight at once
                cf src[kJT][P];
#pragma unroll
                for (int jj = 0; jj < kJT; jj++)
#pragma unroll
                    for (int ps = 0; ps < P; ps++) if (jj < ja) src[jj][ps] = get(kJT * tr + jj, ps);
#pragma unroll
                for (int jj = 0; jj < kJT; jj++) {
                    if (jj >= ja) continue;
                    cf a[P];
#pragma unroll
                    for (int ps = 0; ps < P; ps++) a[ps] = src[jj][ps];
                    blk_pass_dft<P>(a);
                    float2 *const gw = (jj < kJB ? gw0 : gw1) + (jj % kJB) * 16 * kLd;
                    st2(&gw[0], a[0]);
#pragma unroll
                    for (int k = 1; k < P; k++) st2(&gw[32 * k], cmul(a[blk_pass_idx<P>(k)], ct[k]));
                }
                FDC_STAMP(10 + 5 * tr);
                __builtin_amdgcn_sched_barrier(0);                    // keep the next phase's arithmetic (and its registers) behind
                __syncthreads();                                      // the trip is in LDS
                FDC_STAMP(11 + 5 * tr);
                const bool reads = P == 8 || 4 * rh2 < ja;            // P < 8: waves whose row half the trip does not have sit the phase out
                cf v[32];
                if (reads) {
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const float4 t = ld4(&gr[2 * i]);
                        v[2 * i] = mk(t.x, t.y); v[2 * i + 1] = mk(t.z, t.w);
                    }
                }
                __syncthreads();                                      // every read of the trip is done: the region may be rewritten
                __builtin_amdgcn_sched_barrier(0);
                FDC_STAMP(12 + 5 * tr);
                if (reads) {
                    dft32<false>(v);                                  // khi = k0 + 2 k1 in v[16 k0 + rev16(k1)]
                    FDC_STAMP(13 + 5 * tr);
                    // Stores: slot klo + P khi of row t' = 16 kJT tr + 64 rh + lane.  The 32 stream offsets are the same for the whole
                    // wave (table laid out [klo][register]).  Unused slots: the byte offset is pushed beyond the buffer's extent and the
                    // store is dropped by the range check of the descriptor (no branch per store).
                    // FWD: [block][N bins]; row t' = b + 16 j of the run is bin k2 = b + 16 (2 j + h): four 16-bin runs per wave store, 32 bins apart
                    const int trow0 = 16 * kJT * tr + 64 * rh2, trow = trow0 + lane2;
                    const unsigned rb = FW2 ? (unsigned)(m * GM::kN + (trow & 15) + 32 * (trow >> 4) + 16 * fhalf) * 8u
                                            : (unsigned)(m * (FWD ? GM::kN : (R4 ? 192 : 128)) + rowbase + trow) * 8u;
                    // two workgroups per block: the two 64-bin groups of k2 a wave's store touches (lanes 0-31 / 32-63): kept if some channel reads either
                    unsigned mq = FW2 ? (mqs[(trow0 >> 5) & 3] | mqs[((trow0 >> 5) + 1) & 3])
                                      : FWD ? mqs[((rowbase + trow0) >> 6) & 3] : ~0u;
                    if constexpr (FWD) asm volatile("" : "+s"(mq));   // the 32 scalar terms below are worked out here, not held from block to block
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const uint4 t = sow[q];
                        const unsigned so[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            // FWD: a store nobody reads gets the offset of an unused slot (all ones: a scalar term, one OR per store)
                            const unsigned o = FWD ? (so[e] | (((mq >> (4 * q + e)) & 1u) - 1u)) : so[e];
                            bst2t<NT>(rout, (o == 0xFFFFFFFFu ? 0xFFFFFFF0u : o + rb), v[4 * q + e]);
                        }
                    }
                    if constexpr (FWD && !FW2) {
                        // Power of every 16-bin group of the spectrum while it is in the registers (round 6: what the sinks' power cells are summed from —
                        // k_cell_power used to read the whole spectrum back, 444-472 MB per 1024 blocks).  This lane holds bins 256 slot + k2 of 32
                        // slots, its row of 16 lanes is one 16-bin group of each: 32 sums over 16 lanes as ONE transposed reduction — every step
                        // halves the registers and pairs the lanes of one bit (bits 3 and 2 of the lane: DPP row rotations / shifts whose bank mask
                        // picks which half of the lanes writes; bits 1 and 0: quad permutations) — 66 additions instead of 128, and lane j of the
                        // row ends with the sums of registers 2 j and 2 j + 1.
                        if (gpow) {
                            float t16[16];
#pragma unroll
                            for (int i = 0; i < 16; i++) {
                                const float s0 = __builtin_fmaf(v[i].y, v[i].y, v[i].x * v[i].x);
                                const float s1 = __builtin_fmaf(v[i + 16].y, v[i + 16].y, v[i + 16].x * v[i + 16].x);
                                // (s_nop 1: a DPP operand written by the VALU instruction just before needs two wait states; the compiler's hazard
                                // recogniser does not look into inline assembly)
                                asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                                    "v_add_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc" : "=&v"(t16[i]) : "v"(s0), "v"(s1));
                            }
                            float t8[8];
#pragma unroll
                            for (int i = 0; i < 8; i++)
                                asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                                    "v_add_f32_dpp %0, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa" : "=&v"(t8[i]) : "v"(t16[i]), "v"(t16[i + 8]));
                            const bool b1 = (lane2 & 2) != 0, b0 = (lane2 & 1) != 0;
                            float t4[4], t2[2];
#pragma unroll
                            for (int i = 0; i < 4; i++) {
                                const float keepv = b1 ? t8[i + 4] : t8[i], send = b1 ? t8[i] : t8[i + 4];
                                t4[i] = keepv + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0x4E /* quad_perm [2,3,0,1] */, 0xF, 0xF, false));
                            }
#pragma unroll
                            for (int i = 0; i < 2; i++) {
                                const float keepv = b0 ? t4[i + 2] : t4[i], send = b0 ? t4[i] : t4[i + 2];
                                t2[i] = keepv + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, false));
                            }
                            // Lane (row g, j) now holds the sums of registers 2 j, 2 j + 1 over row g = group g of this trip's four.  Stored from there a
                            // wave's 128 values would be 128 separate 4-byte write requests (the four groups of a slot sit in four different rows of the
                            // wave; measured: + 31 us per 896 blocks, the request rate, not the bytes).  One hop through the LDS crossbar (ds_bpermute: no
                            // LDS memory) first: lane (rho, i) takes register R = 8 rho + (i >> 1), groups 2 (i & 1) and + 1, so that two NEIGHBOURING lanes
                            // write the 16 contiguous bytes of a slot's four groups: 32 requests per wave and trip.
                            const int i16 = lane2 & 15, rho = lane2 >> 4, gsrc = 2 * (i16 & 1);
                            const int a0 = (16 * gsrc + 4 * rho + (i16 >> 2)) * 4, a1 = a0 + 64;
                            const int z0 = __builtin_bit_cast(int, t2[0]), z1 = __builtin_bit_cast(int, t2[1]);
                            const int f0a = __builtin_amdgcn_ds_bpermute(a0, z0), f0b = __builtin_amdgcn_ds_bpermute(a0, z1);
                            const int f1a = __builtin_amdgcn_ds_bpermute(a1, z0), f1b = __builtin_amdgcn_ds_bpermute(a1, z1);
                            const bool odd = (i16 & 2) != 0;                      // R & 1: which of the source lane's two registers
                            float2 pr;
                            pr.x = __builtin_bit_cast(float, odd ? f0b : f0a);
                            pr.y = __builtin_bit_cast(float, odd ? f1b : f1a);
                            // register R -> slot (soff is laid out [klo][register]: register R is khi = (R >> 4) + 2 rev16(R & 15), slot klo + P khi);
                            // gpow is [block][slot][16 groups of the slot's 256 bins]
                            const int reg = 8 * rho + (i16 >> 1), slot = klo2 + P * ((reg >> 4) + 2 * (4 * (reg & 3) + ((reg & 15) >> 2)));
                            *reinterpret_cast<float2 *>(gpow + (size_t)m * (GM::kN / 16) + 16 * slot + (rowbase + trow0) / 16 + gsrc) = pr;
                        }
                    }
                }
                FDC_STAMP(14 + 5 * tr);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        stage2([&](int j, int ps) { return FDC_GGET(j, ps); }, R4 ? 64 : 0, std::integral_constant<int, 8>{});
        if constexpr (FWD && !FW2) {
            // second half of k2: the values stage 1 put aside are this lane's own stores; sc1 loads are served by the L2
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stage2([&](int j, int ps) { return bld2_sc1(rscr, (unsigned)tid * 8u + (unsigned)(j * 4096 + ps * 32768), 0u); }, 128,
                   std::integral_constant<int, 8>{});
        }
        if constexpr (R4) {
            // rows 64..127 of the inverse transforms = output rows 0..63: this lane's own stores, served by the L2
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stage2([&](int j, int ps) { return bld2_sc1(rscr, (unsigned)tid * 8u + (unsigned)(j * 4096 + ps * 16384), 0u); }, 0,
                   std::integral_constant<int, 4>{});
        }
        FDC_STAMP(30);
#ifdef FDC_BLK_STAMPS
        if (dbg && blockIdx.x == 0 && lane == 0 && dbgk < 4)
            for (int i = 0; i < 32; i++) dbg[(w * 4 + dbgk) * 32 + i] = st[i];
#endif
        dbgk++;
        // the trip region (= stage-1 scratch) was last read before the barrier above: the next block starts without one
    }
#ifdef FDC_BLK_WGTIMES
    if (dbg && tid == 0) dbg[256 + blockIdx.x] = wall_clock64();
#endif
}

// -DFDC_BLK_STAGED=1: the plain channelizer at N = 65536 with its loads staged through LDS (the STG variant above): measured 1.5 % SLOWER than the
// shipped form (profiles/r04/NOTES.md section 11), kept as a build variant
#ifndef FDC_BLK_STAGED
#define FDC_BLK_STAGED 0
#endif
// -DFDC_FWD_STAGED=1: the forward-transform variant at N = 65536 with its loads staged through LDS (experiment, round 6)
#ifndef FDC_FWD_STAGED
#define FDC_FWD_STAGED 0
#endif

hipError_t init_block_kernels()
{
    hipError_t e;
#define FDC_SETB(P, A, B, F, R4) \
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_blk256<P, A, B, F, R4>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                            B ? BlkGeom<P>::kLdsOff : BlkGeom<P>::kLds); \
    if (e != hipSuccess) return e;
#define FDC_SETP(P) \
    FDC_SETB(P, true, false, false, false) FDC_SETB(P, false, false, false, false) FDC_SETB(P, true, true, false, false) \
    FDC_SETB(P, false, true, false, false) FDC_SETB(P, true, false, false, true) FDC_SETB(P, false, false, false, true) \
    FDC_SETB(P, true, true, false, true) FDC_SETB(P, false, true, false, true)
    FDC_SETP(2) FDC_SETP(4) FDC_SETP(8)
    FDC_SETB(8, true, false, true, false) FDC_SETB(8, false, false, true, false) FDC_SETB(4, true, false, true, false) FDC_SETB(4, false, false, true, false)
    FDC_SETB(2, true, false, true, false) FDC_SETB(2, false, false, true, false)
#undef FDC_SETP
#undef FDC_SETB
#define FDC_SETH(P, A, R4) \
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_blk256<P, A, false, false, R4, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, BlkGeom<P>::kLds); \
    if (e != hipSuccess) return e;
#define FDC_SETHP(P) FDC_SETH(P, true, false) FDC_SETH(P, false, false) FDC_SETH(P, true, true) FDC_SETH(P, false, true)
    FDC_SETHP(2) FDC_SETHP(4) FDC_SETHP(8)
#undef FDC_SETHP
#undef FDC_SETH
#if FDC_FWD_STAGED
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_blk256<8, true, false, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, BlkGeom<8>::kLdsS);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_blk256<8, false, false, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, BlkGeom<8>::kLdsS);
    if (e != hipSuccess) return e;
#endif
#if FDC_BLK_STAGED
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_blk256<8, true, false, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, BlkGeom<8>::kLdsS);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_blk256<8, false, false, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, BlkGeom<8>::kLdsS);
    if (e != hipSuccess) return e;
#endif
    return hipSuccess;
}

bool poly_block_supports(int N) { return N == 16384 || N == 32768 || N == 65536; }

hipError_t launch_poly_block(const float2 *in, size_t in_stride, float2 *out, int nb_chunk, int mbase, int nb_call,
                             const float2 *tw256, const float2 *twq, const float2 *cbt, const float *shn,
                             const long long *slot_off, unsigned out_bytes, int ncu, int hints, hipStream_t s,
                             unsigned long long *dbg, int r, long long first_block, hipEvent_t ev_start, hipEvent_t ev_stop, int R,
                             float2 *scratch, int N)
{
    if (nb_chunk <= 0) return hipSuccess;
    const bool halfslot = (r & 255) == 128;                 // half a slot: the on-grid kernel with its tables moved (HALF), R = 2 and 4
    if (!poly_block_supports(N) || (R != 2 && R != 4) || (R == 4 && !scratch)) return hipErrorInvalidValue;
    int grid = ncu > 0 ? ncu : 256;                         // one 512-thread workgroup per CU (LDS: up to 159.5 KiB each)
    // N = 16384 on the grid at R = 2: 126 registers and 79.75 KiB of LDS per workgroup: two workgroups per CU, one's stage 2 beside the
    // other's stage 1
    if (N == 16384 && R == 2 && (!(r & 255) || halfslot)) grid *= 2;
    if (grid > nb_chunk) grid = nb_chunk;
    // output samples are written once and never read back here: streamed (nt) stores, measured 0.186 -> 0.172 ms (hints bit 0)
    // ev_start / ev_stop (timing): the dispatch packet's own begin / end time stamps (hipExtLaunchKernel) — no barrier packet
    // in front of or behind the kernel, unlike hipEventRecord (measured 7-17 us per bracketed launch)
    // R = 4: three quarters of every inverse transform kept: 192 rows per block, 64 of them via the scratch
#define FDC_LB(P, A, B, R4) \
    hipExtLaunchKernelGGL((k_blk256<P, A, B, false, R4>), dim3((unsigned)grid), dim3(512), B ? BlkGeom<P>::kLdsOff : BlkGeom<P>::kLds, s, ev_start, \
                          ev_stop, 0u, in, in_stride, out, tw256, twq, cbt, shn, slot_off, (long long)mbase * (R4 ? 192 : 128), (long long)nb_call, \
                          out_bytes, nb_chunk, hints, dbg, r & 255, first_block, R4 ? scratch : (float2 *)nullptr, (const unsigned *)nullptr, (float *)nullptr)
#define FDC_LH(P, A, R4) \
    hipExtLaunchKernelGGL((k_blk256<P, A, false, false, R4, false, true>), dim3((unsigned)grid), dim3(512), BlkGeom<P>::kLds, s, ev_start, \
                          ev_stop, 0u, in, in_stride, out, tw256, twq, cbt, shn, slot_off, (long long)mbase * (R4 ? 192 : 128), (long long)nb_call, \
                          out_bytes, nb_chunk, hints, dbg, 0, first_block, R4 ? scratch : (float2 *)nullptr, (const unsigned *)nullptr, (float *)nullptr)
#define FDC_LP(P) \
    do { \
        if (halfslot) { \
            if (R == 4) { if (hints & 1) FDC_LH(P, true, true); else FDC_LH(P, false, true); } \
            else { if (hints & 1) FDC_LH(P, true, false); else FDC_LH(P, false, false); } \
        } else if (R == 4 && (r & 255)) { if (hints & 1) FDC_LB(P, true, true, true); else FDC_LB(P, false, true, true); } \
        else if (R == 4) { if (hints & 1) FDC_LB(P, true, false, true); else FDC_LB(P, false, false, true); } \
        else if (r & 255) { if (hints & 1) FDC_LB(P, true, true, false); else FDC_LB(P, false, true, false); } \
        else { if (hints & 1) FDC_LB(P, true, false, false); else FDC_LB(P, false, false, false); } \
    } while (0)
#if FDC_BLK_STAGED
    if (N == 65536 && R == 2 && !(r & 255)) {
        // the plain channelizer with its loads staged through LDS
#define FDC_LS(A) \
        hipExtLaunchKernelGGL((k_blk256<8, A, false, false, false, true>), dim3((unsigned)grid), dim3(512), BlkGeom<8>::kLdsS, s, ev_start, ev_stop, 0u, in, \
                              in_stride, out, tw256, twq, cbt, shn, slot_off, (long long)mbase * 128, (long long)nb_call, out_bytes, nb_chunk, hints, dbg, 0, \
                              first_block, (float2 *)nullptr, (const unsigned *)nullptr, (float *)nullptr)
        if (hints & 1) FDC_LS(true); else FDC_LS(false);
#undef FDC_LS
    } else
#endif
    if (N == 65536) FDC_LP(8); else if (N == 32768) FDC_LP(4); else FDC_LP(2);
#undef FDC_LP
#undef FDC_LH
#undef FDC_LB
    return hipGetLastError();
}

// Forward transform of nitems blocks of 65536 samples (item m at in + m*in_stride) into the shifted, 1/N-scaled spectrum
// out[m][N] with the block kernel (N = 16384 / 32768 / 65536: k_blk256<P, ..., FWD>).  slot_off[c] = 256 c, c < N / 256; shn1[k2] = 1/N; cbt0 = the r = 0 table.
// ev: null or 3 events: start and end of the kernel (dispatch stamps), and an event recorded behind it (the 3-event protocol
// of the two-pass transform: its second interval is empty here).
hipError_t launch_block_fft(int N, const float2 *in, size_t in_stride, float2 *out, int nitems, const float2 *tw256,
                            const float2 *twq, const float2 *cbt0, const float *shn1, const long long *slot_off,
                            float2 *scratch /* ncu x 32768 points */, int ncu, int hints, hipStream_t s, hipEvent_t *ev,
                            const unsigned *keep, float *gpow /* null, or nitems x N/16 floats: the power of every 16-bin group of the spectrum */)
{
    if (!poly_block_supports(N)) return hipErrorInvalidValue;
    const int per = (int)(((size_t)1 << 28) / (size_t)N);  // 32-bit byte offsets inside one launch: at most 2 GiB of spectrum (N = 65536: 4096 blocks)
    for (int m0 = 0; m0 < nitems; m0 += per) {
        const int nb = nitems - m0 < per ? nitems - m0 : per;
        int grid = ncu > 0 ? ncu : 256;
#if FDC_FWD_TWO_WG
        grid &= ~1;                                        // two workgroups per block (the halves of k2): an even grid, at most two per block
        if (grid > 2 * nb) grid = 2 * nb;
#else
        if (grid > nb) grid = nb;
#endif
        hipEvent_t e0 = ev && m0 == 0 ? ev[0] : nullptr, e2 = ev && m0 + nb >= nitems ? ev[1] : nullptr;
#define FDC_LF(P, A) \
        hipExtLaunchKernelGGL((k_blk256<P, A, false, true>), dim3((unsigned)grid), dim3(512), BlkGeom<P>::kLds, s, e0, e2, 0u, in + (size_t)m0 * in_stride, \
                              in_stride, out + (size_t)m0 * (size_t)N, tw256, twq, cbt0, shn1, slot_off, 0ll, 1ll, \
                              (unsigned)((size_t)nb * (size_t)N * 8), nb, hints, (unsigned long long *)nullptr, 0, 0ll, scratch, keep, \
                              gpow ? gpow + (size_t)m0 * (size_t)(N / 16) : (float *)nullptr)
        const bool nt = (hints & 1) != 0;
#if FDC_FWD_STAGED
#define FDC_LFS(A) \
        hipExtLaunchKernelGGL((k_blk256<8, A, false, true, false, true>), dim3((unsigned)grid), dim3(512), BlkGeom<8>::kLdsS, s, e0, e2, 0u, in + (size_t)m0 * in_stride, \
                              in_stride, out + (size_t)m0 * (size_t)N, tw256, twq, cbt0, shn1, slot_off, 0ll, 1ll, \
                              (unsigned)((size_t)nb * (size_t)N * 8), nb, hints, (unsigned long long *)nullptr, 0, 0ll, scratch, keep, \
                              gpow ? gpow + (size_t)m0 * (size_t)(N / 16) : (float *)nullptr)
        if (N == 65536) { if (nt) FDC_LFS(true); else FDC_LFS(false); } else
#undef FDC_LFS
#endif
        if (N == 65536) { if (nt) FDC_LF(8, true); else FDC_LF(8, false); }
        else if (N == 32768) { if (nt) FDC_LF(4, true); else FDC_LF(4, false); }
        else { if (nt) FDC_LF(2, true); else FDC_LF(2, false); }
#undef FDC_LF
    }
#if FDC_FWD_TWO_WG
    // (variant build: the epilogue reduction exists in the shipped form only)
    if (gpow) { hipError_t e = launch_group_power(out, N, nitems, gpow, s); if (e != hipSuccess) return e; }
#endif
    if (ev) { hipError_t e = hipEventRecord(ev[2], s); if (e != hipSuccess) return e; }
    return hipGetLastError();
}

}  // namespace fdc
