// Uniform path, both stages in ONE persistent launch (N = 65536, 256 slots): "merged" form of k_p1 + k_p2.
//
// Sixteen workgroups that share an XCD (blockIdx mod 8 is the XCD) form a group; a group owns a run of consecutive
// blocks.  Workgroup c of the group does stage 1 of column tile c for every block of the run (overlap handed over in
// registers exactly as in k_p1) and, after its stage-1 tile of block m, one 16-row stage-2 tile of block m - 1 when the
// rotation gives it one: tile t of block mm goes to member (t + mm*T2) mod 16, so over two blocks (R = 2: T2 = 8) every
// member does two stage-1 tiles and one stage-2 tile.  G of a block is written by the group with plain stores (they
// stay in the XCD's L2), counted on a per-block arrival counter after every storing wave has drained its stores, and
// read back one block-time later by members of the same group with L1-bypassing (sc1) loads — same XCD, same L2.
// The streams that are touched once (input rows, output samples) carry the nt hint so that they do not push G out.
// No global barrier, no queue: the only waits are on the 16 arrivals of an earlier block (`lag` blocks behind; bounded
// spin; on time-out the error word is set and fdc_pipeline_synchronize reports it).
//
// EXPERIMENT, off by default (FDC_POLY_MERGED=1; FDC_MERGED_HINTS bits: 1 nt input, 2 nt output, 4 stage 1 only, 8 do not
// wait (wrong results), 16/32 lag - 2, 64 print poll statistics).  Measured on MI355X, 1024 blocks: stage 1 alone inside
// this kernel 0.132 ms (= k_p1); both stages WITHOUT waiting 0.201 ms — the ceiling of the idea, 15 % under the
// 0.237 ms of k_p1 + k_p2; with the waits 0.328 / 0.282 / 0.266 / 0.259 ms at lag 2 / 3 / 4 / 5.  A poll of the arrival
// counter costs ~3 us under load and a group's members run one to two tasks apart (the stage-2 duty alternates), so
// most waits poll at least once; reading the counter one tile early made it worse (0.30-0.36).  The two launches stay.
#include "fdc_kernels.h"
#include "fdc_devutil.hpp"

namespace fdc {

extern __shared__ __attribute__((aligned(16))) unsigned char fdc_smem_merged[];

constexpr int kMergedLds = 256 * 16 * 8 + 2304 + 2304 + 1024 + 1024;     // tile, wrow, tq, sh, soff = 39424 B

template <int KEEP>
__global__ __launch_bounds__(256, 4) void k_pm(const float2 *__restrict__ in, size_t in_stride, float2 *__restrict__ g,
                                               float2 *__restrict__ out, const float2 *__restrict__ tw256,
                                               const float2 *__restrict__ twq, const float2 *__restrict__ cbt,
                                               const float *__restrict__ shn, const long long *__restrict__ slot_off,
                                               int *__restrict__ done, int *__restrict__ err, int nb, int bpg,
                                               int qskip, int lout, long long out_base, long long nb_call,
                                               unsigned out_bytes, int hints)
{
    constexpr int TC = 16, N1 = 256;
    float2 *tile = reinterpret_cast<float2 *>(fdc_smem_merged);                          // 4096 points, both stages
    float2 *wrow = reinterpret_cast<float2 *>(fdc_smem_merged + 32768);                  // [b][p] = W256^(b p), 16 x 18
    float2 *tq = reinterpret_cast<float2 *>(fdc_smem_merged + 32768 + 2304);             // [col][q], 16 x 18
    float *sh = reinterpret_cast<float *>(fdc_smem_merged + 32768 + 4608);               // [b][q]
    unsigned *soff = reinterpret_cast<unsigned *>(fdc_smem_merged + 32768 + 4608 + 1024);
    const int tid = threadIdx.x;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int member = j & 15, gid = xcd * (gridDim.x >> 7) + (j >> 4);                 // gridDim = 8 XCDs x groups x 16
    const int m0 = gid * bpg, m1 = m0 + bpg < nb ? m0 + bpg : nb;
    if (m0 >= m1) return;
    const int T2 = lout >> 4;                              // stage-2 tiles (16 rows) per block
    // ---- stage-1 constants (k_p1 with TC = 16)
    const int col = tid & 15, b = tid >> 4, c0 = member * TC;
    for (int i = tid; i < 256; i += 256) {
        wrow[(i >> 4) * 18 + (i & 15)] = tw256[((i >> 4) * (i & 15)) & 255];
        sh[i] = shn[(i >> 4) + 16 * (i & 15)];
        const long long o = slot_off[i];
        soff[i] = o >= 0 ? (unsigned)((o * nb_call + out_base) * 8) : 0xFFFFFFFFu;
    }
    tq[col * 18 + b] = twq[(size_t)(c0 + col) * 16 + b];
    const cf cb = ld2(&cbt[(size_t)(c0 + col) * 16 + b]);
    const unsigned voff = (unsigned)(b * N1 + c0 + col) * 8u;
    const unsigned rowstep = 16u * (unsigned)N1 * 8u;
    const unsigned inbytes = 256u * (unsigned)N1 * 8u;
    const unsigned gtile = (unsigned)lout * TC * 8u;
    const unsigned goff = (unsigned)(b * TC + col) * 8u;
    const unsigned gstep = 16u * TC * 8u;
    // ---- stage-2 constants (k_p2 with TR = 16)
    const int r = tid >> 4, bq = tid & 15;                 // layer 1: row r, points n1 = 16a + bq
    const int r2 = tid & 15, p2 = tid >> 4;                // layer 2: row r2, outputs k1 = p2 + 16q
    const unsigned ctstep = (unsigned)lout * 16u * 8u;     // bytes between column tiles of one block
    const unsigned v2off = (unsigned)(r * 16 + bq) * 8u;
    const __amdgpu_buffer_rsrc_t rout = make_rsrc(out, out_bytes);
    const bool nt_in = hints & 1, nt_out = hints & 2;
    bool dead = false;                                     // a wait timed out: stop waiting, results are invalid
    int spin_total = 0, nwait = 0;                         // diagnostics (lane 0): polls that found the counter short
    const int lag = 2 + ((hints >> 4) & 3);                // blocks between a stage-1 tile and the stage-2 tiles that read it

    auto stage1 = [&](cf (&cur)[16], cf (&nbuf)[16], int m, int pend) {
        if (m + 1 < m1) {
            const __amdgpu_buffer_rsrc_t rin = make_rsrc(in + (size_t)(m + 1) * in_stride, inbytes);
#pragma unroll
            for (int a = 0; a < KEEP; a++) nbuf[a] = cur[a + 16 - KEEP];
            if (nt_in) {
#pragma unroll
                for (int a = KEEP; a < 16; a++) nbuf[a] = bld2_nt(rin, voff, a * rowstep);
            } else {
#pragma unroll
                for (int a = KEEP; a < 16; a++) nbuf[a] = bld2(rin, voff, a * rowstep);
            }
        }
        const __amdgpu_buffer_rsrc_t rg = make_rsrc(g + ((size_t)m * 16 + member) * (size_t)lout * TC, gtile);
        dft16<false>(cur);
        __syncthreads();
        cf w[16];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float4 t = ld4(&wrow[b * 18 + 2 * i]);
            w[2 * i] = mk(t.x, t.y); w[2 * i + 1] = mk(t.z, t.w);
        }
#pragma unroll
        for (int p = 0; p < 16; p++) st2(&tile[(16 * b + p) * TC + col], cmul(cur[rev16(p)], w[p]));
        __syncthreads();
        cf v[16];
#pragma unroll
        for (int bb = 0; bb < 16; bb++) v[bb] = ld2(&tile[(16 * bb + b) * TC + col]);
        dft16<false>(v);
        cf u[16];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float4 t0 = ld4(&tq[col * 18 + 4 * i]), t1 = ld4(&tq[col * 18 + 4 * i + 2]);
            const float4 sv = *reinterpret_cast<const float4 *>(&sh[b * 16 + 4 * i]);
            u[(4 * i) ^ 8] = cmul(v[rev16(4 * i)], mk(t0.x, t0.y)) * sv.x;
            u[(4 * i + 1) ^ 8] = cmul(v[rev16(4 * i + 1)], mk(t0.z, t0.w)) * sv.y;
            u[(4 * i + 2) ^ 8] = cmul(v[rev16(4 * i + 2)], mk(t1.x, t1.y)) * sv.z;
            u[(4 * i + 3) ^ 8] = cmul(v[rev16(4 * i + 3)], mk(t1.z, t1.w)) * sv.w;
        }
        dft16<true>(u);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float4 t = ld4(&wrow[b * 18 + 2 * i]);
            w[2 * i] = mk(t.x, t.y); w[2 * i + 1] = mk(t.z, t.w);
        }
#pragma unroll
        for (int p = 0; p < 16; p++) u[rev16(p)] = cmul(cmulc(u[rev16(p)], w[p]), cb);
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 16; p++) st2(&tile[(16 * b + p) * TC + col], u[rev16(p)]);
        __syncthreads();
#pragma unroll
        for (int bb = 0; bb < 16; bb++) u[bb] = ld2(&tile[(16 * bb + b) * TC + col]);
        dft16<true>(u);
        // deferred arrival of the PREVIOUS tile: its stores were issued a whole tile ago, so draining them costs nothing
        // now; every storing wave drains, then one lane counts that tile in
        if (pend >= 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(&done[pend], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int q = 0; q < 16; q++)
            if (q >= qskip) bst2(rg, goff, (unsigned)(q - qskip) * gstep, u[rev16(q)]);
    };

    auto stage2 = [&](int mm, int t) {
        if (hints & 4) return;                               // diagnostics: stage 1 alone
        // all 16 column tiles of block mm have arrived?  (hints & 8: diagnostics, do not wait — wrong results)
        if (tid == 0 && !dead && !(hints & 8)) {
            int spins = 0;
            nwait++;
            while (__hip_atomic_load(&done[mm], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 16) {
                __builtin_amdgcn_s_sleep(8);
                spin_total++;
                if (++spins > (1 << 18)) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); dead = true; break; }
            }
        }
        __syncthreads();                                     // also: stage 1's last LDS reads are done
        const __amdgpu_buffer_rsrc_t rg = make_rsrc(g + (size_t)mm * (size_t)lout * 256 + (size_t)t * 16 * 16, (unsigned)lout * 256u * 8u);
        cf v[16];
#pragma unroll
        for (int a = 0; a < 16; a++) v[a] = bld2_sc1(rg, v2off, (unsigned)a * ctstep);
        dft16<false>(v);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            cf w[8];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float4 tt = ld4(&wrow[bq * 18 + 8 * h + 2 * i]);
                w[2 * i] = mk(tt.x, tt.y); w[2 * i + 1] = mk(tt.z, tt.w);
            }
#pragma unroll
            for (int p = 0; p < 8; p++) {
                const int pv = 8 * h + p;
                st2(&tile[(pv * 16 + (bq ^ (pv & 1))) * 16 + ((r ^ bq) & 15)], cmul(v[rev16(pv)], w[p]));
            }
        }
        __syncthreads();
#pragma unroll
        for (int bb = 0; bb < 16; bb++) v[bb] = ld2(&tile[(p2 * 16 + (bb ^ (p2 & 1))) * 16 + ((r2 ^ bb) & 15)]);
        dft16<false>(v);
        const unsigned rbytes = (unsigned)(((long long)mm * lout + t * 16 + r2) * 8);
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const unsigned off = soff[p2 + 16 * q];
            if (off != 0xFFFFFFFFu) {
                if (nt_out) bst2_nt(rout, off + rbytes, 0, v[rev16(q)]);
                else bst2(rout, off + rbytes, 0, v[rev16(q)]);
            }
        }
    };

    cf L[16];
    {
        const __amdgpu_buffer_rsrc_t rin = make_rsrc(in + (size_t)m0 * in_stride, inbytes);
#pragma unroll
        for (int a = 0; a < 16; a++) L[a] = bld2(rin, voff, a * rowstep);
    }
    for (int m = m0; m < m1; m++) {
        {
            cf cur[16];
#pragma unroll
            for (int a = 0; a < 16; a++) cur[a] = L[a];
            stage1(cur, L, m, m > m0 ? m - 1 : -1);
        }
        if (m >= m0 + lag) {
            const int mm = m - lag, t = (member - mm * T2) & 15;    // my stage-2 tile of the block `lag` behind, if any
            if (t < T2) stage2(mm, t);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the last tile's arrival
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(&done[m1 - 1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int mm = (m1 - lag > m0 ? m1 - lag : m0); mm < m1; mm++) {
        const int t = (member - mm * T2) & 15;
        if (t < T2) stage2(mm, t);
    }
    if (tid == 0 && (hints & 64)) { atomicAdd(&err[1], spin_total); atomicAdd(&err[2], nwait); }
}

hipError_t init_merged_kernels()
{
    hipError_t e;
#define FDC_SETM(K) \
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_pm<K>), hipFuncAttributeMaxDynamicSharedMemorySize, kMergedLds); \
    if (e != hipSuccess) return e;
    FDC_SETM(0) FDC_SETM(8) FDC_SETM(4) FDC_SETM(2) FDC_SETM(1)
#undef FDC_SETM
    return hipSuccess;
}

// nb_chunk blocks in one launch; `done` holds nb_chunk ints (zeroed here), `err` one int the caller reads back later.
hipError_t launch_poly_merged(const float2 *in, size_t in_stride, float2 *g, float2 *out, int R, int nb_chunk, int mbase,
                              int nb_call, const float2 *tw256, const float2 *twq, const float2 *cbt, const float *shn,
                              const long long *slot_off, unsigned out_bytes, int *done, int *err, int ncu, int hints,
                              hipStream_t s)
{
    const int skip = 256 / R, lout = 256 - skip;
    // one group = 16 workgroups of one XCD; 4 workgroups per CU resident: groups per XCD = ncu*4 / (8*16)
    int gpx = ncu * 4 / 128;
    if (gpx < 1) return hipErrorInvalidConfiguration;
    int groups = gpx * 8;
    if (groups > nb_chunk) {                                // short launch: fewer groups per XCD (at least one each)
        gpx = (nb_chunk + 7) / 8;
        groups = gpx * 8;
    }
    const int bpg = (nb_chunk + groups - 1) / groups;
    hipError_t e = hipMemsetAsync(done, 0, sizeof(int) * (size_t)nb_chunk, s);
    if (e != hipSuccess) return e;
    const bool reuse = in_stride == (size_t)65536 - (size_t)65536 / R;
    const dim3 grid((unsigned)(groups * 16));
#define FDC_LM(K) \
    hipLaunchKernelGGL(k_pm<K>, grid, dim3(256), kMergedLds, s, in, in_stride, g, out, tw256, twq, cbt, shn, slot_off, done, \
                       err, nb_chunk, bpg, skip / 16, lout, (long long)mbase * lout, (long long)nb_call, out_bytes, hints)
    if (!reuse) FDC_LM(0); else if (R == 2) FDC_LM(8); else if (R == 4) FDC_LM(4); else if (R == 8) FDC_LM(2); else FDC_LM(1);
#undef FDC_LM
    return hipGetLastError();
}

}  // namespace fdc
