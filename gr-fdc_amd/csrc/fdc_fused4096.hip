// N = 4096 in ONE launch (round 6): overlap-save gather + forward transform + every channel's slice, phase / window, ifftshift,
// inverse transform, overlap discard and * l — the spectrum of a block never leaves its compute unit.  This is the block length of the
// reference's own example flowgraph (examples/FDC_example.grc: l = 256 / 512 / 1024 / 512 at N = 4096) and of BASELINE configs[0]; the
// two-launch spectrum path it replaces there (k_fft4096 + k_c256 / k_c512 / k_c1024) writes the bins some channel reads to memory and reads
// them back: python/FrequencyDomainChannelizer.py:206 (fft_vcc) -> :214-216 (vector_cut_vxx, phase_shifting_windowing_vcc) -> :218-226
// (ifft, vector_cut_vxx, multiply_const) per channel.
//
//   a workgroup takes T blocks, T teams of four waves: each team transforms one block (4096 points = 32 KiB, the same three DFT-16 layers as
//   k_fft4096, fdc_chanwide.hip) and stores the shifted, 1/N-scaled spectrum to LDS over its exchange tile (32 KiB); every one of the 4 T
//   WAVES then owns rows — (block of the workgroup, channel) — of ONE channel width: the plan's schedule, made by the host (fdc_api.hip,
//   plan_fused4096).  T = 2 (512 threads, two workgroups on a unit) for plans with wide rows: a wave's instructions cost the same for one
//   row as for a full wave of them, and the reference's example plan fills its 1024- and 512-bin waves only with the rows of two blocks;
//   T = 1 (256 threads, FOUR workgroups on a unit: four barrier domains instead of two, 11 - 12 % faster) where no row is wide.
//       l =  256: 16 lanes x 16 points per row, up to eight rows (two sets of four) per wave   (the row machinery of k_c256)
//       l =  512: 16 lanes x 32 points, up to four rows                                        (k_c512)
//       l = 1024: 32 lanes x 32 points, up to two rows                                         (k_c1024)
//       l =  128:  8 lanes x 16 points, eight rows; l = 64 / 32 / 16: 4 / 2 / 1 lanes, eight rows   (DFT-16 over a, exchange inside the row, DFT-8 / 4 / 2 over b)
//     a row reads its slice from the LDS spectrum and its window row (phase counter in closed form) from memory, runs its first DFT
//     layer in registers; ONE workgroup barrier (every slice has been read) and the tile is free for the rows' own exchanges, which stay
//     inside a wave: no further barrier.  Sum of the rows' exchange areas <= the T tiles, at most 4 T waves of rows: plans with more
//     (channels that overlap to more than 4096 bins in total, more than 32 narrow channels) stay on the spectrum path.
// Bytes per block: H = N - N/R new input samples + sum(lout) output samples, nothing else (the window rows and tables are cache hits).
#include "fdc_kernels.h"
#include "fdc_radix16.hpp"
#include "fdc_devutil.hpp"

namespace fdc {

extern __shared__ __attribute__((aligned(16))) unsigned char fdc_smem_f4[];

// diagnostics (tools/build_variant.sh -DF4_EXP=bits): 1 = every input load from one cached line, 2 = every window load from one cached line,
// 4 = no output stores (unless a value the data never takes): what each stream's latency costs the kernel (profiles/r06/NOTES.md section 8);
// 8 = three of the forward transform's five workgroup barriers left out (WRONG results): what they cost
#ifndef F4_EXP
#define F4_EXP 0
#endif
// streamed (nt) accesses: bit 0 = output stores, bit 1 = input loads (A/B: profiles/r06/NOTES.md section 8)
#ifndef F4_NT
#define F4_NT 1
#endif
namespace {
constexpr int kF4TilePts = 16 * 272;                         // exchange tile of one block's forward transform; then its spectrum; then rows' exchanges
// LDS image for T blocks (teams of four waves) per workgroup: [T tiles][W_256^(x y) 16 x 18][W_4096^(x y) 16 x 18][schedule: 4 T waves x 8 slots]
// and, with wide rows, [W_1024^(x p) 16 x 34][W_64^(c p) 2 x 34]
constexpr int f4_off_t256(int T) { return T * kF4TilePts * 8; }
constexpr int f4_off_t4k(int T) { return f4_off_t256(T) + 16 * 18 * 8; }
constexpr int f4_off_rows(int T) { return f4_off_t4k(T) + 16 * 18 * 8; }
constexpr int f4_lds(int T) { return f4_off_rows(T) + 32 * T * 32; }                 // T = 2: 76288 (two workgroups per unit); T = 1: 40448 (four)
constexpr int f4_off_w1k(int T) { return f4_lds(T); }
constexpr int f4_off_w64(int T) { return f4_off_w1k(T) + 16 * 34 * 8; }
constexpr int f4_lds_wide(int T) { return f4_off_w64(T) + 2 * 34 * 8; }              // T = 2: 81184
static_assert(sizeof(F4Row) == 32, "schedule rows are copied 16 bytes at a time");
static_assert(2 * f4_lds_wide(2) <= 160 * 1024 && 4 * f4_lds(1) <= 160 * 1024 && 3 * f4_lds_wide(1) <= 160 * 1024, "LDS budget");

__device__ __forceinline__ constexpr int pos32(int k) { return 16 * (k & 1) + rev16(k >> 1); }   // dft32 leaves X[k0 + 2 k1] in v[16 k0 + rev16(k1)]

// stores of some lanes of a wave, then loads of others of the SAME wave: LDS serves a wave's accesses in order, the compiler is told not to move them
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void out_st(float2 *p, cf v)
{
#if F4_NT & 1
    if (!(F4_EXP & 4) || v.x == 1.2345e30f) __builtin_nontemporal_store(v, reinterpret_cast<cf *>(p));
#else
    if (!(F4_EXP & 4) || v.x == 1.2345e30f) st2(p, v);
#endif
}

struct RowAt { const float2 *win; const float2 *spec; long long dst; bool on; };
// ri.valid: 0 = no row, 1 + k = a row of the workgroup's block k
// fbm = (first block of the call + first block of the launch) mod R
__device__ __forceinline__ RowAt row_at(const F4Row &ri, int L, int m0, int nb, int mbase, int fbm, int R, const float2 *wins,
                                        long long nb_call, const float2 *tiles)
{
    const int k = ri.valid ? ri.valid - 1 : 0, m = m0 + k;
    const int cnt = (int)(((unsigned)(fbm + m) % (unsigned)R) * (unsigned)ri.shift % (unsigned)R);     // phase counter in closed form (lib/phase_shifting_windowing_vcc_impl.cc:58,83-89)
    return RowAt{wins + ((F4_EXP & 2) ? 0 : ri.win_off + cnt * L), tiles + k * kF4TilePts + ri.f, nb_call * ri.out_off + ((long long)mbase + m) * ri.lout - (L - ri.lout),
                 ri.valid != 0 && m < nb};
}
}  // namespace

// TEAMS teams of four waves (256 TEAMS threads): team t transforms block TEAMS g + t of workgroup g's blocks; then every wave runs the rows the schedule gives it.
// TEAMS = 2 wherever rows of ONE block would leave waves part empty (all wide rows: the reference's example fills its 1024- and 512-bin waves only with the
// rows of two blocks); TEAMS = 1 — four independent workgroups per unit instead of two of twice the size — where the schedule of one block fits four waves.
// wcls: four bits per wave: 0 = no rows, 1 = l = 256 (slots 0..3), 2 = l = 256 two sets (slots 0..7), 3 = l = 512 (slots 0..3), 4 = l = 1024 (slots 0..1),
// 5 = l = 128, 6 = l = 64, 7 = l = 32, 8 = l = 16 (slots 0..7 each)
template <bool WIDE, int TEAMS>
__global__ __launch_bounds__(256 * TEAMS, 4 /* waves per SIMD */) void k_f4096(const float2 *__restrict__ in, size_t in_stride, float2 *__restrict__ out, int nb, int R,
                                                  int mbase, int nb_call, int fbm /* (first block of the call + mbase) mod R */, const float2 *__restrict__ tw,
                                                  int twstride /* ntab / 4096 */, const float2 *__restrict__ wins,
                                                  const F4Row *__restrict__ rows, unsigned wcls)
{
    float2 *tiles = reinterpret_cast<float2 *>(fdc_smem_f4);
    float2 *t256 = reinterpret_cast<float2 *>(fdc_smem_f4 + f4_off_t256(TEAMS));
    float2 *t4k = reinterpret_cast<float2 *>(fdc_smem_f4 + f4_off_t4k(TEAMS));
    const F4Row *srows = reinterpret_cast<const F4Row *>(fdc_smem_f4 + f4_off_rows(TEAMS));
    const int team = TEAMS == 1 ? 0 : threadIdx.x >> 8, tid = threadIdx.x & 255, lo = tid & 15, hi = tid >> 4;
    // neighbouring blocks share R - 1 of R input samples: workgroup ids go round the eight XCDs, so XCD x takes the x-th eighth of the launch
    // and the shared samples are hits in ITS L2
    const int ngroups = (nb + TEAMS - 1) / TEAMS, per = (ngroups + 7) >> 3;
    const int grp = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (grp >= ngroups) return;
    const int m0 = TEAMS * grp, m = m0 + team;
    float2 *tile = tiles + team * kF4TilePts;
    if (team == 0) {
        t256[hi * 18 + lo] = tw[((16 * hi * lo) & 4095) * twstride];
        t4k[hi * 18 + lo] = tw[(hi * lo) * twstride];
    }
    if (team == TEAMS - 1) {
        if (tid < 64 * TEAMS) reinterpret_cast<float4 *>(fdc_smem_f4 + f4_off_rows(TEAMS))[tid] = reinterpret_cast<const float4 *>(rows)[tid];
        if constexpr (WIDE) {
            float2 *w1k = reinterpret_cast<float2 *>(fdc_smem_f4 + f4_off_w1k(TEAMS)), *w64 = reinterpret_cast<float2 *>(fdc_smem_f4 + f4_off_w64(TEAMS));
            for (int i = tid; i < 512; i += 256) w1k[(i >> 5) * 34 + (i & 31)] = tw[((i >> 5) * (i & 31)) * (4 * twstride)];
            if (tid < 64) w64[(tid >> 5) * 34 + (tid & 31)] = tid < 32 ? make_float2(1.0f, 0.0f) : tw[(tid & 31) * (64 * twstride)];
        }
    }
    // ---- forward transform: n = a + 16 b + 256 c, k = k0 + 16 k1 + 256 k2 (k_fft4096, fdc_chanwide.hip) ---------------------------------
    cf v[32];
    {
        cf (&u)[16] = reinterpret_cast<cf (&)[16]>(v[0]);
#pragma unroll
        for (int c = 0; c < 16; c++) u[c] = m < nb ? ((F4_NT & 2) ? __builtin_nontemporal_load(reinterpret_cast<const cf *>(in + (size_t)m * in_stride + (tid + 256 * c)))
                                                                   : ld2(in + ((F4_EXP & 1) ? (size_t)(tid & 15) + 16 * c : (size_t)m * in_stride + (tid + 256 * c)))) : mk(0.f, 0.f);
        __syncthreads();
        dft16<false>(u);                                             // layer 1 over c: k0 in u[rev16(k0)]; thread = (a = lo, b = hi)
        {
            cf w[16];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const float4 t = ld4(&t256[hi * 18 + 2 * i]);
                w[2 * i] = mk(t.x, t.y); w[2 * i + 1] = mk(t.z, t.w);
            }
#pragma unroll
            for (int k0 = 0; k0 < 16; k0++) st2(&tile[k0 * 272 + tid], k0 == 0 ? u[rev16(0)] : cmul(u[rev16(k0)], w[k0]));
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < 16; b++) u[b] = ld2(&tile[hi * 272 + b * 16 + lo]);      // thread = (a = lo, k0 = hi)
        dft16<false>(u);                                             // layer 2 over b: k1 in u[rev16(k1)]
        if constexpr (F4_EXP & 8) __builtin_amdgcn_wave_barrier(); else
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
            for (int k1 = 0; k1 < 16; k1++) st2(&tile[k1 * 257 + hi * 16 + (lo ^ hi)], cmul(u[rev16(k1)], k1 == 0 ? s : cmul(s, w[k1])));
        }
        if constexpr (F4_EXP & 8) __builtin_amdgcn_wave_barrier(); else
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 16; a++) u[a] = ld2(&tile[hi * 257 + lo * 16 + (a ^ lo)]);   // thread = (k0 = lo, k1 = hi)
        dft16<false>(u);                                             // layer 3 over a: bin k0 + 16 k1 + 256 k2 in u[rev16(k2)]
        if constexpr (F4_EXP & 8) __builtin_amdgcn_wave_barrier(); else
        __syncthreads();                                             // every read of exchange 2 is done
        // the shifted spectrum (fftshift: bin k at k + N/2; python/FrequencyDomainChannelizer.py:206 fft_vcc(..., shift = True)), times 1/N
#pragma unroll
        for (int k2 = 0; k2 < 16; k2++) st2(&tile[tid + 256 * (k2 ^ 8)], u[rev16(k2)] * (1.0f / 4096.0f));
    }
    __syncthreads();
    // ---- the rows of this wave ---------------------------------------------------------------------------------------------------------
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const unsigned cls = (wcls >> (4 * wave)) & 0xfu;
    const F4Row *wr = srows + 8 * wave;
    F4Row r0{}, r1{};
    RowAt a0{}, a1{};
    if (cls == 1 || cls == 2) {
        const int b = lane & 15;
        r0 = wr[lane >> 4];
        a0 = row_at(r0, 256, m0, nb, mbase, fbm, R, wins, nb_call, tiles);
        {
            cf w[16];
#pragma unroll
            for (int a = 0; a < 16; a++) w[a] = ld2(a0.win + ((F4_EXP & 2) ? 0 : 16 * a) + b);
#pragma unroll
            for (int a = 0; a < 16; a++) v[a ^ 8] = cmul(ld2(a0.spec + 16 * a + b), w[a]);      // ifftshift of the slice: i -> i + l/2
        }
        if (cls == 2) {
            r1 = wr[4 + (lane >> 4)];
            a1 = row_at(r1, 256, m0, nb, mbase, fbm, R, wins, nb_call, tiles);
            cf w[16];
#pragma unroll
            for (int a = 0; a < 16; a++) w[a] = ld2(a1.win + ((F4_EXP & 2) ? 0 : 16 * a) + b);
#pragma unroll
            for (int a = 0; a < 16; a++) v[16 + (a ^ 8)] = cmul(ld2(a1.spec + 16 * a + b), w[a]);
        }
        dft16<true>(reinterpret_cast<cf (&)[16]>(v[0]));
        if (cls == 2) dft16<true>(reinterpret_cast<cf (&)[16]>(v[16]));
    }
    if constexpr (WIDE) {
        if (cls == 3 || cls == 4) {
            const int L = cls == 3 ? 512 : 1024, lg = cls == 3 ? 4 : 5;
            const int b = lane & ((1 << lg) - 1);
            r0 = wr[lane >> lg];
            a0 = row_at(r0, L, m0, nb, mbase, fbm, R, wins, nb_call, tiles);
            cf w[32];
#pragma unroll
            for (int a = 0; a < 32; a++) w[a] = ld2(a0.win + ((F4_EXP & 2) ? 0 : (a << lg)) + b);
#pragma unroll
            for (int a = 0; a < 32; a++) v[a ^ 16] = cmul(ld2(a0.spec + (a << lg) + b), w[a]);
            dft32<true>(v);                                          // over a: index p in v[pos32(p)]
        }
    }
    if (cls >= 5) {
        // l = 128 (8 lanes x 16 points per row), 64 (4 lanes), 32 (2 lanes), 16 (one lane): eight rows on the first 8 * lanes lanes of the wave; slice (a << lg) + b
        const int lg = 8 - (int)cls, b = lane & ((1 << lg) - 1);
        r0 = wr[(lane >> lg) & 7];
        a0 = row_at(r0, 16 << lg, m0, nb, mbase, fbm, R, wins, nb_call, tiles);
        if ((lane >> lg) >= 8) a0.on = false;
        cf w[16];
#pragma unroll
        for (int a = 0; a < 16; a++) w[a] = ld2(a0.win + ((F4_EXP & 2) ? 0 : (a << lg)) + b);
#pragma unroll
        for (int a = 0; a < 16; a++) v[a ^ 8] = cmul(ld2(a0.spec + (a << lg) + b), w[a]);          // ifftshift of the slice: i -> i + l/2 = a -> a ^ 8
        dft16<true>(reinterpret_cast<cf (&)[16]>(v[0]));             // over a: index p in v[rev16(p)]
    }
    __syncthreads();                                                 // every slice has been read: the tiles belong to the rows' exchanges
    if (cls == 1 || cls == 2) {
        const int b = lane & 15;
        cf w[16];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float4 t = ld4(&t256[b * 18 + 2 * i]);
            w[2 * i] = mk(t.x, t.y); w[2 * i + 1] = mk(t.z, t.w);
        }
        // element (b, p) of a row at p * 16 + (b ^ p): stores of one p and loads of one b are conflict-free (k_c256)
        float2 *row0 = tiles + r0.xch, *row1 = tiles + r1.xch;
        if (a0.on) {
#pragma unroll
            for (int p = 0; p < 16; p++) st2(&row0[p * 16 + (b ^ p)], cmulc(v[rev16(p)], w[p]));
        }
        if (cls == 2 && a1.on) {
#pragma unroll
            for (int p = 0; p < 16; p++) st2(&row1[p * 16 + (b ^ p)], cmulc(v[16 + rev16(p)], w[p]));
        }
        wave_sync();
#pragma unroll
        for (int bb = 0; bb < 16; bb++) v[bb] = ld2(&row0[b * 16 + (bb ^ b)]);
        if (cls == 2) {
#pragma unroll
            for (int bb = 0; bb < 16; bb++) v[16 + bb] = ld2(&row1[b * 16 + (bb ^ b)]);
        }
        dft16<true>(reinterpret_cast<cf (&)[16]>(v[0]));
        if (cls == 2) dft16<true>(reinterpret_cast<cf (&)[16]>(v[16]));
        // y[t], t = b + 16 q; keep t >= l/R (vector_cut_vxx(l, l - lout, lout)), times l (multiply_const_cc)
        const int skip = 256 - r0.lout;
        if (a0.on) {
#pragma unroll
            for (int q = 0; q < 16; q++) if (b + 16 * q >= skip) out_st(out + a0.dst + b + 16 * q, v[rev16(q)] * 256.f);
        }
        if (cls == 2 && a1.on) {
#pragma unroll
            for (int q = 0; q < 16; q++) if (b + 16 * q >= skip) out_st(out + a1.dst + b + 16 * q, v[16 + rev16(q)] * 256.f);
        }
    }
    if constexpr (WIDE) {
        // the inter-layer twiddles of both wide forms from ONE table: W_1024^(x p) = W_1024^((x & 15) p) W_64^((x >> 4) p), x < 32;
        // l = 1024: x = b; l = 512: W_512^(b p) = W_1024^(2 b p), x = 2 b
        if (cls == 3 || cls == 4) {
            const int lg = cls == 3 ? 4 : 5, b = lane & ((1 << lg) - 1), x = cls == 3 ? 2 * b : b;
            const float2 *wa = reinterpret_cast<const float2 *>(fdc_smem_f4 + f4_off_w1k(TEAMS)) + (x & 15) * 34;
            const float2 *wb = reinterpret_cast<const float2 *>(fdc_smem_f4 + f4_off_w64(TEAMS)) + (x >> 4) * 34;
            float2 *row = tiles + r0.xch;
            // element (b, p) of a row at p * lanes + (b ^ (p mod lanes)) (k_c1024, k_c512)
            if (a0.on) {
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const float4 t = ld4(&wa[2 * i]), c = ld4(&wb[2 * i]);
                    st2(&row[((2 * i) << lg) + (b ^ ((2 * i) & ((1 << lg) - 1)))], cmulc(v[pos32(2 * i)], cmul(mk(t.x, t.y), mk(c.x, c.y))));
                    st2(&row[((2 * i + 1) << lg) + (b ^ ((2 * i + 1) & ((1 << lg) - 1)))], cmulc(v[pos32(2 * i + 1)], cmul(mk(t.z, t.w), mk(c.z, c.w))));
                }
            }
            wave_sync();
            if (cls == 4) {
#pragma unroll
                for (int bb = 0; bb < 32; bb++) v[bb] = ld2(&row[b * 32 + (bb ^ b)]);
                dft32<true>(v);                                      // y[t = b + 32 q] in v[pos32(q)]
                const int skip = 1024 - r0.lout;
                if (a0.on) {
#pragma unroll
                    for (int q = 0; q < 32; q++) if (b + 32 * q >= skip) out_st(out + a0.dst + b + 32 * q, v[pos32(q)] * 1024.f);
                }
            } else {
                // l = 512 = 32 x 16: DFT-16 over b for p = lane and p = lane + 16; y[t = p + 32 q]
#pragma unroll
                for (int bb = 0; bb < 16; bb++) {
                    v[bb] = ld2(&row[b * 16 + (bb ^ b)]);
                    v[16 + bb] = ld2(&row[(b + 16) * 16 + (bb ^ b)]);
                }
                dft16<true>(reinterpret_cast<cf (&)[16]>(v[0]));
                dft16<true>(reinterpret_cast<cf (&)[16]>(v[16]));
                const int skip = 512 - r0.lout;
                if (a0.on) {
#pragma unroll
                    for (int q = 0; q < 16; q++) {
                        const int t0 = b + 32 * q, t1 = t0 + 16;
                        if (t0 >= skip) out_st(out + a0.dst + t0, v[rev16(q)] * 512.f);
                        if (t1 >= skip) out_st(out + a0.dst + t1, v[16 + rev16(q)] * 512.f);
                    }
                }
            }
        }
    }
    if (cls >= 5) {
        // y[t = p + 16 q] = sum_b W_l^(-b p) W_(l/16)^(-b q) (DFT-16 over a)[p]: twiddle W_l^(b p) = W_256^((256 / l) b p) from the 256 table, an exchange inside the
        // row — element (b, p) at p * lanes + (b ^ (p mod lanes)) — then every lane runs the DFT-(l/16) over b for its 16 / lanes values of p
        const int lg = 8 - (int)cls, lanes = 1 << lg, b = lane & (lanes - 1);
        cf w[16];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float4 t = ld4(&t256[(b << (4 - lg)) * 18 + 2 * i]);
            w[2 * i] = mk(t.x, t.y); w[2 * i + 1] = mk(t.z, t.w);
        }
        float2 *row = tiles + r0.xch;
        if (a0.on) {
#pragma unroll
            for (int p = 0; p < 16; p++) st2(&row[(p << lg) + (b ^ (p & (lanes - 1)))], cmulc(v[rev16(p)], w[p]));
        }
        wave_sync();
        if (cls == 5) {
            // p = b and p = b + 8: two DFT-8 (dft8 leaves X[k0 + 2 k1] in [4 k0 + k1])
#pragma unroll
            for (int bb = 0; bb < 8; bb++) {
                v[bb] = ld2(&row[(b << 3) + (bb ^ b)]);
                v[8 + bb] = ld2(&row[((b + 8) << 3) + (bb ^ b)]);
            }
            dft8<true>(reinterpret_cast<cf (&)[8]>(v[0]));
            dft8<true>(reinterpret_cast<cf (&)[8]>(v[8]));
            const int skip = 128 - r0.lout;
            if (a0.on) {
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int t0 = b + 16 * q, t1 = t0 + 8;
                    if (t0 >= skip) out_st(out + a0.dst + t0, v[4 * (q & 1) + (q >> 1)] * 128.f);
                    if (t1 >= skip) out_st(out + a0.dst + t1, v[8 + 4 * (q & 1) + (q >> 1)] * 128.f);
                }
            }
        } else if (cls == 7) {
            // l = 32: p = b + 2 j, j < 8: eight DFT-2; y[t = p + 16 q], q < 2
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const cf e = ld2(&row[((b + 2 * j) << 1) + b]), o = ld2(&row[((b + 2 * j) << 1) + (1 ^ b)]);
                v[2 * j] = e + o; v[2 * j + 1] = e - o;
            }
            const int skip = 32 - r0.lout;
            if (a0.on) {
#pragma unroll
                for (int j = 0; j < 8; j++) {
#pragma unroll
                    for (int q = 0; q < 2; q++) if (b + 2 * j + 16 * q >= skip) out_st(out + a0.dst + b + 2 * j + 16 * q, v[2 * j + q] * 32.f);
                }
            }
        } else if (cls == 8) {
            // l = 16: the DFT-16 over a is the whole transform (the trip through the row's exchange area only puts y[p] into register p)
#pragma unroll
            for (int p = 0; p < 16; p++) v[p] = ld2(&row[p]);
            const int skip = 16 - r0.lout;
            if (a0.on) {
#pragma unroll
                for (int p = 0; p < 16; p++) if (p >= skip) out_st(out + a0.dst + p, v[p] * 16.f);
            }
        } else {
            // p = b + 4 j, j < 4: four DFT-4
#pragma unroll
            for (int j = 0; j < 4; j++) {
#pragma unroll
                for (int bb = 0; bb < 4; bb++) v[4 * j + bb] = ld2(&row[((b + 4 * j) << 2) + (bb ^ b)]);
                dft4<true>(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
            }
            const int skip = 64 - r0.lout;
            if (a0.on) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
#pragma unroll
                    for (int q = 0; q < 4; q++) if (b + 4 * j + 16 * q >= skip) out_st(out + a0.dst + b + 4 * j + 16 * q, v[4 * j + q] * 64.f);
                }
            }
        }
    }
}

hipError_t init_fused4096_kernels()
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_f4096<false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, f4_lds(2));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_f4096<false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, f4_lds(1));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_f4096<true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, f4_lds_wide(1));
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(k_f4096<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, f4_lds_wide(2));
}

int fused4096_tile_points() { return kF4TilePts; }

// teams: blocks per workgroup the schedule was made for (1: rows[4 waves][8]; 2: rows[8 waves][8])
hipError_t launch_fused4096(const float2 *in, size_t in_stride, float2 *out, int nb_chunk, int R, int mbase, int nb_call, int64_t first_block,
                            const float2 *tw, int ntab, const float2 *wins, const F4Row *rows, unsigned wcls, int teams, hipStream_t s)
{
    if (nb_chunk <= 0) return hipSuccess;
    bool wide = false;
    for (int w = 0; w < 8; w++) wide = wide || ((wcls >> (4 * w)) & 0xfu) == 3 || ((wcls >> (4 * w)) & 0xfu) == 4;
    if (teams != 2 && teams != 1) return hipErrorInvalidValue;
    const int ngroups = (nb_chunk + teams - 1) / teams;
    const dim3 grid((unsigned)(8 * ((ngroups + 7) / 8)));
    const int fbm = (int)((first_block + mbase) % R);
    if (wide && teams == 1)
        hipLaunchKernelGGL((k_f4096<true, 1>), grid, dim3(256), f4_lds_wide(1), s, in, in_stride, out, nb_chunk, R, mbase, nb_call, fbm, tw, ntab / 4096, wins, rows, wcls);
    else if (wide)
        hipLaunchKernelGGL((k_f4096<true, 2>), grid, dim3(512), f4_lds_wide(2), s, in, in_stride, out, nb_chunk, R, mbase, nb_call, fbm, tw, ntab / 4096, wins, rows, wcls);
    else if (teams == 2)
        hipLaunchKernelGGL((k_f4096<false, 2>), grid, dim3(512), f4_lds(2), s, in, in_stride, out, nb_chunk, R, mbase, nb_call, fbm, tw, ntab / 4096, wins, rows, wcls);
    else
        hipLaunchKernelGGL((k_f4096<false, 1>), grid, dim3(256), f4_lds(1), s, in, in_stride, out, nb_chunk, R, mbase, nb_call, fbm, tw, ntab / 4096, wins, rows, wcls);
    return hipGetLastError();
}

}  // namespace fdc
