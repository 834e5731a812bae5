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
//   stage 2, 4 chunks of 32 rows t': G registers -> LDS [row][n1] -> DFT-16 over a (n1 = 16a + b2) -> twiddle ->
//       LDS [p2][b2][row] -> DFT-16 over b2 -> the 256 slot outputs of 32 consecutive rows: 256-byte runs per channel.
//       Two LDS buffers, two s_barriers per chunk.
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

// LDS map (bytes).  Stage-1 scratch: per wave 68*15 + 64 = 1084 points (element (p; lane) at lane + 68 p).
constexpr int kBlkScrPts = 1084;
constexpr int kBlkGbufLd = 260;                                   // stage-2 G chunk: [32 rows][260]
constexpr int kBlkXbufPts = 8448;                                 // stage-2 exchange: r + 33 b2 + 528 p2
constexpr int kBlkOffX = 8 * kBlkScrPts * 8;                      // 69376 (>= 32*260*8 = 66560)
constexpr int kBlkOffWrow = kBlkOffX + kBlkXbufPts * 8;           // 136960
constexpr int kBlkOffB = kBlkOffWrow + 16 * 18 * 8;               // 139264
constexpr int kBlkOffSA = kBlkOffB + 32 * 18 * 8;                 // 143872
constexpr int kBlkOffSoff = kBlkOffSA + 128 * 18 * 8;             // 162304
constexpr int kBlkLds = kBlkOffSoff + 256 * 4;                    // 163328 <= 163840
constexpr int kBlkOffSoffOff = kBlkOffSA + 128 * 16 * 8;          // offset plans: unpadded SA rows, then soff, then wrowF
constexpr int kBlkOffWrowF = kBlkOffSoffOff + 256 * 4;
constexpr int kBlkLdsOff = kBlkOffWrowF + 16 * 18 * 8;           // 163584 <= 163840
static_assert(kBlkLdsOff <= 160 * 1024, "LDS budget of the offset-plan variant");
static_assert(kBlkLds <= 160 * 1024, "LDS budget");
static_assert(kBlkOffX >= 32 * kBlkGbufLd * 8, "G chunk must fit the scratch region");

template <bool NT, bool OFF>
__global__ FDC_PLAIN_DS __launch_bounds__(512) void k_blk256(const float2 *__restrict__ in, size_t in_stride, float2 *__restrict__ out,
                                                const float2 *__restrict__ tw256, const float2 *__restrict__ twq,
                                                const float2 *__restrict__ cbt, const float *__restrict__ shn,
                                                const long long *__restrict__ slot_off, long long out_base,
                                                long long nb_call, unsigned out_bytes, int nb, int hints,
                                                unsigned long long *__restrict__ dbg, int roff, long long first_block)
{
    float2 *scr = reinterpret_cast<float2 *>(fdc_smem_blk);                     // stage 1: 8 wave scratches; stage 2: G chunk
    float2 *xbuf = reinterpret_cast<float2 *>(fdc_smem_blk + kBlkOffX);
    float2 *wrow = reinterpret_cast<float2 *>(fdc_smem_blk + kBlkOffWrow);      // [b][p] = W256^(b p), rows of 18
    float2 *Bt = reinterpret_cast<float2 *>(fdc_smem_blk + kBlkOffB);           // [c5][q] = W_N^(16 c5 q)
    float2 *SA = reinterpret_cast<float2 *>(fdc_smem_blk + kBlkOffSA);          // [pass][b][q] = shape[b+16q]/N * W_N^(512 pass q)
    // offset plans need a second twiddle table: the SA rows give up their padding for it (2-way conflicts on 8 reads per pass)
    constexpr int kSaLd = OFF ? 16 : 18;
    unsigned *soff = reinterpret_cast<unsigned *>(fdc_smem_blk + (OFF ? kBlkOffSoffOff : kBlkOffSoff));
    const int tid = threadIdx.x;
    // stage-1 roles
    const int w = tid >> 6, lane = tid & 63, col = lane & 3, b = lane >> 2, c5 = 4 * w + col;
    // stage-2 roles.  Layer 1: row r (the two rows of a 32-lane read group are 4 apart: their 16-point runs then sit on
    // opposite halves of the 64 banks), points n1 = 16a + b2.  Layer 2: row r2 (fast: stores are 256-B runs), outputs p2 + 16q.
    const int rr = tid >> 4, b2 = tid & 15, r1 = (rr >> 3) * 8 + ((rr >> 1) & 3) + 4 * (rr & 1);
    const int r2 = tid & 31, p2 = tid >> 5;

    // ---- tables (once per workgroup; the workgroup is persistent)
    // Offset plans (every channel at f = 256*slot + r, OFF): the block is modulated by exp(-2 pi i r n / N), n = n1 + 256 (16a + b),
    // without a single extra multiplication.  W_16^(r a) rotates the outputs of the first DFT-16 (index p reads Z[(p + r) mod 16]:
    // the exchange slot of register Z[p] becomes (p - r) mod 16), W_256^(r b) joins the forward twiddle (table wrowF), W_N^(r n1)
    // sits in cbt (host), and for odd r the window phase (-1)^block (phase_shifting_windowing_vcc_impl.cc:82, R = 2) in cb.
    float2 *wrowF = OFF ? reinterpret_cast<float2 *>(fdc_smem_blk + kBlkOffWrowF) : wrow;
    const int r16 = roff & 15;
    for (int i = tid; i < 256; i += 512) {
        wrow[(i >> 4) * 18 + (i & 15)] = tw256[((i >> 4) * (i & 15)) & 255];
        if (OFF)     // entry [b][p]: the twiddle of register Z[p], whose true index is pt = (p - r) mod 16: W_256^(b (pt + r))
            wrowF[(i >> 4) * 18 + (i & 15)] = tw256[((i >> 4) * ((((i & 15) - r16) & 15) + roff)) & 255];
        const long long o = slot_off[i];                  // slot i = p2 + 16 q, q = 2k + odd, is entry [p2][odd][k]
        soff[(i & 15) * 16 + ((i >> 4) & 1) * 8 + (i >> 5)] = o >= 0 ? (unsigned)((o * nb_call + out_base) * 8) : 0xFFFFFFFFu;
    }
    Bt[(tid >> 4) * 18 + (tid & 15)] = twq[tid];                                // c5 = tid >> 4 < 32, q = tid & 15
    for (int i = tid; i < 2048; i += 512) {
        const int ps = i >> 8, bb = (i >> 4) & 15, q = i & 15;
        const float2 t = twq[(size_t)(32 * ps) * 16 + q];                        // W_N^(16 * 32 ps * q)
        const float s = shn[bb + 16 * q];
        SA[(ps * 16 + bb) * kSaLd + q] = make_float2(t.x * s, t.y * s);
    }
    __syncthreads();

    // block order: round rho, XCD x = workgroup mod 8 (round-robin dispatch), slot = workgroup / 8:
    // block = rho*grid + x*(grid/8) + slot, i.e. one XCD works on grid/8 consecutive blocks at a time
    const int grid = gridDim.x, per = grid >> 3;
    const bool xmap = (grid & 7) == 0;
    const int first = xmap ? (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (first >= nb) return;

    const unsigned inbytes = 65536u * 8u;
    const unsigned voff = (unsigned)(b * 256 + c5) * 8u;          // row b, column c5 of pass 0; pass adds 256 B, row group a 32 KiB
    float2 *const scrw = scr + w * kBlkScrPts + lane;             // exchange write base: element p at + 68 p
    const float2 *const scrr = scr + w * kBlkScrPts + col + 68 * b;   // exchange read base: element bb at + 4 bb
    const float2 *const wr = wrow + b * 18;
    const float2 *const wrf = wrowF + b * 18;
    const float2 *const btr = Bt + c5 * 18;
    const __amdgpu_buffer_rsrc_t rout = make_rsrc(out, out_bytes);

    // Two waves share a SIMD (waves w and w + 4).  With equal priority they convoy: both do their DFT-16 arithmetic at
    // half speed together and then wait for their LDS exchanges together.  Unequal priority breaks the tie: the
    // favoured wave runs its arithmetic at full rate, and the other one fills the gaps its waits leave.
    // hints bit 2: static (waves 0-3 favoured); bit 3: the favoured half alternates every pass.
    if (hints & 4) { if (w < 4) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0); }

    const __amdgpu_buffer_rsrc_t rcb = make_rsrc(cbt, 256u * 16u * 8u);      // cbt[n1][b], n1 = 32 pass + c5
    const unsigned voffc = (unsigned)(c5 * 16 + b) * 8u;
    cf L[16], cbn;
    {
        const __amdgpu_buffer_rsrc_t rin = make_rsrc(in + (size_t)first * in_stride, inbytes);
#pragma unroll
        for (int a = 0; a < 16; a++) L[a] = bld2(rin, voff, (unsigned)a * 32768u);
        cbn = bld2(rcb, voffc, 0);
    }
    // diagnostics (FDC_BLOCK_DEBUG=1): cycle stamps of workgroup 0, per wave: [wave][block round][24]
    int dbgk = 0;
#define FDC_STAMP(i) do { if (dbg && blockIdx.x == 0 && lane == 0 && dbgk < 4) dbg[(w * 4 + dbgk) * 32 + (i)] = __builtin_readcyclecounter(); } while (0)
    for (int m = first; m < nb; m += grid) {
        const int mnext = m + grid < nb ? m + grid : m;
        const float sgn = (OFF && (roff & 1) && ((first_block + m) & 1)) ? -1.0f : 1.0f;
        FDC_STAMP(0);
        // G[j][pass]: row t' = b + 16 j, column 32 pass + c5.  One complex value = one 64-bit vector element: the element index
        // is the pass number at run time, and with 64-bit elements the compiler brackets all sixteen moves of a pass with
        // one s_set_gpr_idx_on / off pair (with 32-bit elements it emits a pair per dword).
        // G[j][pass]: row t' = b + 16 j, column 32 pass + c5.  One complex value = one 64-bit vector element (two floats packed
        // into an integer): the element index is the pass number at run time, and with 64-bit elements the compiler brackets
        // all sixteen moves of a pass with ONE s_set_gpr_idx_on / off pair (a pair per dword with 32-bit elements).  Integer,
        // not double, elements: bit-casting an extracted double to two floats read element 0 for every pass (seen in the ISA).
        u8v G[8];
#define FDC_GGET(j, ps) unpack_cf(G[j][ps])
#define FDC_GPUT(j, ps, val) G[j][ps] = pack_cf(val)
        // ---------------- stage 1 ----------------
        // One pass per trip.  The 16 rows of this lane's column were requested a whole pass ago into L; the rows of the next
        // pass (of this block, or pass 0 of this workgroup's next block; after the last block: the same rows again, unused)
        // are requested first, unconditionally (a conditional request costs a second set of register copies).
#pragma nounroll
        for (int ps = 0; ps < 8; ps++) {
            if (hints & 8) { if (((w >> 2) ^ ps) & 1) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(2); }
            const cf cb = OFF ? cbn * sgn : cbn;
            cf cur[16];
#pragma unroll
            for (int a = 0; a < 16; a++) cur[a] = L[a];
            {
                const int pn = ps < 7 ? ps + 1 : 0;
                const int mb = ps < 7 ? m : mnext;
                // the pass offset (32 columns) sits in the descriptor's base: every pass uses the same per-lane offset and the
                // same 16 scalar row offsets
                const __amdgpu_buffer_rsrc_t rin = make_rsrc(in + (size_t)mb * in_stride + 32 * pn, inbytes);
                if (hints & 2) {
#pragma unroll
                    for (int a = 0; a < 16; a++) L[a] = bld2_nt(rin, voff, (unsigned)a * 32768u);
                } else {
#pragma unroll
                    for (int a = 0; a < 16; a++) L[a] = bld2(rin, voff, (unsigned)a * 32768u);
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
            dft16<false>(v);                                      // A[k2 = b + 16 q] in v[rev16(q)]
            cf u[16];
            {
                const float2 *sar = SA + (ps * 16 + b) * kSaLd;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const float4 t0 = ld4(&btr[2 * i]), t1 = ld4(&sar[2 * i]);
                    // window * inter-pass twiddle, placed at the ifftshifted position (k2 ^ 128 <=> q ^ 8)
                    u[(2 * i) ^ 8] = cmul(cmul(v[rev16(2 * i)], mk(t0.x, t0.y)), mk(t1.x, t1.y));
                    u[(2 * i + 1) ^ 8] = cmul(cmul(v[rev16(2 * i + 1)], mk(t0.z, t0.w)), mk(t1.z, t1.w));
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
            FDC_STAMP(1 + ps);
        }
        // ---------------- stage 2 ----------------
        __syncthreads();                                          // every wave is done with its stage-1 scratch
        FDC_STAMP(9);
        float2 *const gw = scr + b * kBlkGbufLd + c5;             // G chunk write base: (row b + 16 jj, column 32 pass + c5)
        const float2 *const gr = scr + r1 * kBlkGbufLd + b2;      // layer-1 read base: point n1 = 16 a + b2
        float2 *const xw = xbuf + r1 + 33 * b2;                   // exchange write base: element p at + 528 p
        const float2 *const xr = xbuf + r2 + 528 * p2;            // exchange read base: element bb at + 33 bb
        const float2 *const wr2 = wrow + b2 * 18;
        const unsigned rowb = (unsigned)(m * 128 + r2) * 8u;
#pragma unroll
        for (int jj = 0; jj < 2; jj++)
#pragma unroll
            for (int ps = 0; ps < 8; ps++) st2(&gw[jj * 16 * kBlkGbufLd + 32 * ps], FDC_GGET(jj, ps));
#pragma unroll
        for (int c = 0; c < 4; c++) {
            __syncthreads();                                      // chunk c of G is in LDS; every read of xbuf (chunk c-1) is done
            FDC_STAMP(10 + 5 * c);
            cf v[16];
#pragma unroll
            for (int a = 0; a < 16; a++) v[a] = ld2(&gr[16 * a]);
            dft16<false>(v);
            {
                cf tw[16];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const float4 t = ld4(&wr2[2 * i]);
                    tw[2 * i] = mk(t.x, t.y); tw[2 * i + 1] = mk(t.z, t.w);
                }
#pragma unroll
                for (int p = 0; p < 16; p++) st2(&xw[528 * p], cmul(v[rev16(p)], tw[p]));
            }
            FDC_STAMP(11 + 5 * c);
            __syncthreads();                                      // exchange written; every read of the G chunk is done
            FDC_STAMP(12 + 5 * c);
            if (c < 3) {
#pragma unroll
                for (int jj = 0; jj < 2; jj++)
#pragma unroll
                    for (int ps = 0; ps < 8; ps++)
                        st2(&gw[jj * 16 * kBlkGbufLd + 32 * ps], FDC_GGET(2 * c + 2 + jj, ps));
            }
            FDC_STAMP(13 + 5 * c);
#pragma unroll
            for (int bb = 0; bb < 16; bb++) v[bb] = ld2(&xr[33 * bb]);
            dft16<false>(v);                                      // slot k1 = p2 + 16 q in v[rev16(q)], row t' = 32 c + r2
            FDC_STAMP(14 + 5 * c);
            // Stores.  A lane holds the 16 slot outputs p2 + 16 q of ONE row; neighbouring lanes hold neighbouring rows.  The pair
            // trades halves (lane ^ 1, DPP): the even lane ends up with both rows of the even q, the odd lane with both rows of
            // the odd q, and every store is 16 bytes — half as many store instructions (their issue, 16 per wave and chunk at
            // one per ~100 cycles, was the longest part of this phase).  Unused slots: the byte offset is pushed beyond the
            // buffer's extent and the store is dropped by the range check of the descriptor (no branch per store).
            const bool oddrow = (tid & 1) != 0;
            const unsigned rb = rowb + (unsigned)c * 256u - (oddrow ? 8u : 0u);        // the even row of the pair
            // this lane's 8 stream offsets: slots p2 + 16 (2k + odd), one b128 pair (table laid out [p2][odd][k])
            unsigned so[8];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const uint4 t = *reinterpret_cast<const uint4 *>(&soff[p2 * 16 + (oddrow ? 8 : 0) + 4 * i]);
                so[4 * i] = t.x; so[4 * i + 1] = t.y; so[4 * i + 2] = t.z; so[4 * i + 3] = t.w;
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const cf A = v[rev16(2 * k)], B = v[rev16(2 * k + 1)];
                const cf send = oddrow ? A : B;
                const cf recv = mk(swap_pair(send.x), swap_pair(send.y));
                const cf lo = oddrow ? recv : A, hi = oddrow ? B : recv;
                bst4<NT>(rout, so[k] == 0xFFFFFFFFu ? 0xFFFFFFF0u : so[k] + rb, lo, hi);
            }
        }
        FDC_STAMP(30);
        dbgk++;
        // the last chunk's xbuf reads may still be in flight in other waves: xbuf is not touched by stage 1, and the G chunk
        // region (= stage-1 scratch) was last read before the barrier above, so the next block starts without a barrier
    }
}

hipError_t init_block_kernels()
{
    hipError_t e;
#define FDC_SETB(A, B) \
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_blk256<A, B>), hipFuncAttributeMaxDynamicSharedMemorySize, B ? kBlkLdsOff : kBlkLds); \
    if (e != hipSuccess) return e;
    FDC_SETB(true, false) FDC_SETB(false, false) FDC_SETB(true, true) FDC_SETB(false, true)
#undef FDC_SETB
    return hipSuccess;
}

hipError_t launch_poly_block(const float2 *in, size_t in_stride, float2 *out, int nb_chunk, int mbase, int nb_call,
                             const float2 *tw256, const float2 *twq, const float2 *cbt, const float *shn,
                             const long long *slot_off, unsigned out_bytes, int ncu, int hints, hipStream_t s,
                             unsigned long long *dbg, int r, long long first_block)
{
    if (nb_chunk <= 0) return hipSuccess;
    int grid = ncu > 0 ? ncu : 256;                         // one 512-thread workgroup per CU (LDS: 159.5 KiB each)
    if (grid > nb_chunk) grid = nb_chunk;
    // output samples are written once and never read back here: streamed (nt) stores, measured 0.186 -> 0.172 ms (hints bit 0)
#define FDC_LB(A, B) \
    hipLaunchKernelGGL((k_blk256<A, B>), dim3((unsigned)grid), dim3(512), B ? kBlkLdsOff : kBlkLds, s, in, in_stride, out, tw256, twq, cbt, shn, \
                       slot_off, (long long)mbase * 128, (long long)nb_call, out_bytes, nb_chunk, hints, dbg, r & 255, first_block)
    if (r & 255) { if (hints & 1) FDC_LB(true, true); else FDC_LB(false, true); }
    else { if (hints & 1) FDC_LB(true, false); else FDC_LB(false, false); }
#undef FDC_LB
    return hipGetLastError();
}

}  // namespace fdc
