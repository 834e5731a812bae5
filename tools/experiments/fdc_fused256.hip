// Uniform-plan path as an XCD-local dataflow: stage 1 and stage 2 run CONCURRENTLY (two persistent kernels on two
// streams, 3 + 1 workgroups per CU) and hand the intermediate G over through the 4-MiB L2 of the XCD they share.
//
// Why: measured on MI355X (tools/ubench), the Infinity Cache serves streaming reads no faster than HBM, so the only place
// where G (lout*N1 points per block, half of all bytes the two-launch form moves through HBM) can live cheaply is L2 —
// and the eight L2s are private to their XCDs.  So every block is owned by ONE XCD (block m -> XCD m mod 8): workgroups
// read their XCC_ID (HW_REG_XCC_ID) and serve only their XCD's queues; G of a block is written with plain stores (stays
// in that L2, write-back), read back a few microseconds later by a stage-2 workgroup of the same XCD, and its ring slot
// is overwritten before it is ever evicted.  HBM then carries only new input samples in and channel samples out.
//
// Hand-off inside an XCD (L2 is the coherence point of its CUs; L1 is write-through and never refreshed):
//   producer (stage-1 tile):  plain G stores -> every wave `s_waitcnt vmcnt(0)` -> __syncthreads() -> lane 0 agent-scope
//                             fetch_add on the block's counter
//   consumer (stage-2 tile):  lane 0 polls the counter (relaxed, agent) until 16 -> __syncthreads() -> every load of G is an
//                             sc1 load (bypasses this CU's L1, served by the XCD's L2)
//   ring reuse:               a stage-1 tile of local block j first waits for stage 2 of local block j - ring.
// Same-XCD is guaranteed by construction (the hardware id, not by assuming a dispatch order), so no L2 write-back
// (release) is needed; a workgroup never touches another XCD's G.  Queues are popped in order and waits only ever point
// at earlier queue entries, which running workgroups already hold, so progress does not depend on residency beyond
// "each kernel has at least one workgroup on each XCD"; every spin is bounded and raises the error word instead of
// hanging, and fdc_pipeline_synchronize() reports it.
// (An earlier form of this file — one kernel, one global task queue, sc1 write-through hand-off valid across XCDs, G in
// the Infinity Cache — was correct but no faster than two launches: profiles/r01/NOTES.md.)
#include "fdc_kernels.h"
#include "fdc_radix16.hpp"
#include "fdc_devutil.hpp"

namespace fdc {

extern __shared__ __attribute__((aligned(16))) unsigned char fdc_smem_fused[];

constexpr int kFTC = 16;                     // columns per stage-1 tile = rows per stage-2 tile
constexpr unsigned kSpinLimit = 1u << 21;    // bounded spins (~1 s), then the error word is set

struct FusedCtl {
    unsigned q1[8][16];                      // per XCD: next stage-1 tile (one counter per 64-B line)
    unsigned q2[8][16];                      // per XCD: next stage-2 block
    unsigned error, pad[15];
    unsigned flags[1];                       // [nb] stage-1 tiles finished, then [nb] stage-2 finished
};

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u; }   // HW_REG_XCC_ID

__device__ __forceinline__ bool wait_geq(unsigned *p, unsigned want)
{
    for (unsigned i = 0; i < kSpinLimit; i++) {
        if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
        __builtin_amdgcn_s_sleep(4);
    }
    return false;
}

// ---- stage 1: tiles of this XCD's blocks, in queue order, next tile's loads in flight while the current one computes
__global__ __launch_bounds__(256, 4) void k_p1x(const float2 *__restrict__ in, size_t in_stride,
                                                float2 *__restrict__ gring, const float2 *__restrict__ tw256,
                                                const float2 *__restrict__ twq, const float2 *__restrict__ cbt,
                                                const float *__restrict__ shn, FusedCtl *ctl, int nb, int ringx,
                                                int qskip, int lout)
{
    constexpr int TC = kFTC, N1 = 256;
    float2 *tile = reinterpret_cast<float2 *>(fdc_smem_fused);                           // 256 x 16 points
    float2 *w256 = reinterpret_cast<float2 *>(fdc_smem_fused + 256 * TC * 8);
    float2 *tq = reinterpret_cast<float2 *>(fdc_smem_fused + 256 * TC * 8 + 2048);        // [q][col]
    float *sh = reinterpret_cast<float *>(fdc_smem_fused + 256 * TC * 8 + 2048 + 2048);
    int *bcast = reinterpret_cast<int *>(fdc_smem_fused + 256 * TC * 8 + 2048 + 2048 + 1024);
    const int tid = threadIdx.x, col = tid & (TC - 1), b = tid / TC;
    const int x = (int)xcc_id();
    const int nbx = (nb - x + 7) >> 3;                               // blocks owned by this XCD: m = 8j + x
    const int ntl = nbx * 16;
    w256[tid] = tw256[tid];
    sh[tid] = shn[tid];
    unsigned *s1done = ctl->flags, *s2done = ctl->flags + nb;
    const unsigned gtile = (unsigned)lout * TC * 8u, gblock = gtile * (N1 / TC);
    unsigned char *gx = reinterpret_cast<unsigned char *>(gring) + (size_t)x * ringx * gblock;
    const unsigned vrow = (unsigned)b * N1 * 8u + (unsigned)col * 8u, rowstep = 16u * N1 * 8u;
    const unsigned goff = (unsigned)(b * TC + col) * 8u, gstep = 16u * TC * 8u;
    auto pop = [&]() -> int {                                        // uniform; two barriers
        __syncthreads();
        if (tid == 0) bcast[0] = (int)__hip_atomic_fetch_add(&ctl->q1[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        return bcast[0];
    };
    cf L[16];
    auto issue = [&](int t) {
        const int m = 8 * (t >> 4) + x, ct = t & 15;
        const __amdgpu_buffer_rsrc_t rin = make_rsrc(in + (size_t)m * in_stride, 256u * N1 * 8u);
        const unsigned vo = vrow + (unsigned)ct * TC * 8u;
#pragma unroll
        for (int a = 0; a < 16; a++) L[a] = bld2(rin, vo, a * rowstep);
    };
    int t = pop();
    if (t < ntl) issue(t);
    while (t < ntl) {
        const int j = t >> 4, ct = t & 15, m = 8 * j + x, c0 = ct * TC;
        // this tile's slice of the short tables first (the in-order vmcnt then leaves the prefetch below in flight)
        const cf tqv = ld2(&twq[(size_t)(c0 + col) * 16 + b]);
        const cf cb = ld2(&cbt[(size_t)(c0 + col) * 16 + b]);
        const int tn = pop();
        cf v[16];
#pragma unroll
        for (int a = 0; a < 16; a++) v[a] = L[a];
        if (tn < ntl) issue(tn);                                     // prefetch
        st2(&tq[tid], tqv);                                          // tq[q*TC + col] with q = b
        if (j >= ringx && tid == 0 && !wait_geq(&s2done[8 * (j - ringx) + x], (unsigned)(lout / TC)))   // ring slot read out?
            __hip_atomic_store(&ctl->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        dft16<false>(v);
        cf w[16];
#pragma unroll
        for (int p = 0; p < 16; p++) w[p] = ld2(&w256[b * p]);
#pragma unroll
        for (int p = 0; p < 16; p++) st2(&tile[(16 * b + p) * TC + col], cmul(v[rev16(p)], w[p]));
        __syncthreads();
#pragma unroll
        for (int bb = 0; bb < 16; bb++) v[bb] = ld2(&tile[(16 * bb + b) * TC + col]);
        dft16<false>(v);
        cf u[16];
#pragma unroll
        for (int q = 0; q < 16; q++) u[q ^ 8] = cmul(v[rev16(q)], ld2(&tq[q * TC + col])) * sh[b + 16 * q];
        dft16<true>(u);
#pragma unroll
        for (int p = 0; p < 16; p++) w[p] = ld2(&w256[b * p]);
#pragma unroll
        for (int p = 0; p < 16; p++) u[rev16(p)] = cmul(cmulc(u[rev16(p)], w[p]), cb);
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 16; p++) st2(&tile[(16 * b + p) * TC + col], u[rev16(p)]);
        __syncthreads();
#pragma unroll
        for (int bb = 0; bb < 16; bb++) u[bb] = ld2(&tile[(16 * bb + b) * TC + col]);
        dft16<true>(u);
        const __amdgpu_buffer_rsrc_t rg = make_rsrc(gx + (size_t)(j % ringx) * gblock + (size_t)ct * gtile, gtile);
#pragma unroll
        for (int q = 0; q < 16; q++)
            if (q >= qskip) bst2(rg, goff, (unsigned)(q - qskip) * gstep, u[rev16(q)]);   // plain: stays in this XCD's L2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // every storing wave, before the barrier
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(&s1done[m], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t = tn;
    }
}

// ---- stage 2: row tiles (16 rows t' of one block) of this XCD, in queue order; G comes out of the XCD's L2 through
// sc1 loads (L1 bypass), so no fence is needed and a block's eight tiles run on eight workgroups at once — the
// stage-2 latency of a block, which bounds how small the G ring can be, is one tile time.
__global__ __launch_bounds__(256, 4) void k_p2x(const float2 *__restrict__ gring, float2 *__restrict__ out,
                                                const float2 *__restrict__ tw256,
                                                const long long *__restrict__ slot_off, FusedCtl *ctl, int nb,
                                                int ringx, int lout, long long out_base, long long nb_call,
                                                unsigned out_bytes)
{
    constexpr int TC = kFTC, N1 = 256;
    float2 *tile = reinterpret_cast<float2 *>(fdc_smem_fused);
    float2 *w256 = reinterpret_cast<float2 *>(fdc_smem_fused + 256 * TC * 8);
    unsigned *soff = reinterpret_cast<unsigned *>(fdc_smem_fused + 256 * TC * 8 + 2048);
    int *bcast = reinterpret_cast<int *>(fdc_smem_fused + 256 * TC * 8 + 2048 + 1024);
    const int tid = threadIdx.x;
    const int x = (int)xcc_id();
    const int nbx = (nb - x + 7) >> 3;
    const int s2tiles = lout / TC;
    const int ntl = nbx * s2tiles;
    w256[tid] = tw256[tid];
    {
        const long long o = slot_off[tid];
        soff[tid] = o >= 0 ? (unsigned)((o * nb_call + out_base) * 8) : 0xFFFFFFFFu;
    }
    unsigned *s1done = ctl->flags, *s2done = ctl->flags + nb;
    const unsigned gtile = (unsigned)lout * TC * 8u, gblock = gtile * (N1 / TC);
    const unsigned char *gx = reinterpret_cast<const unsigned char *>(gring) + (size_t)x * ringx * gblock;
    const __amdgpu_buffer_rsrc_t rout = make_rsrc(out, out_bytes);
    const int r = tid >> 4, b = tid & 15;                            // layer 1: row r, points n1 = 16a + b (ct = a)
    const int r2 = tid & (TC - 1), p2 = tid / TC;                    // layer 2: row r2, outputs k1 = p2 + 16q
    const unsigned voff = (unsigned)(r * TC + b) * 8u;
    for (;;) {
        __syncthreads();
        if (tid == 0) {
            const int t = (int)__hip_atomic_fetch_add(&ctl->q2[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t < ntl && !wait_geq(&s1done[8 * (t / s2tiles) + x], 16u))
                __hip_atomic_store(&ctl->error, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bcast[0] = t;
        }
        __syncthreads();
        const int t = bcast[0];
        if (t >= ntl) break;
        const int j = t / s2tiles, tt = t - j * s2tiles, m = 8 * j + x;
        const __amdgpu_buffer_rsrc_t rg = make_rsrc(gx + (size_t)(j % ringx) * gblock + (size_t)tt * TC * TC * 8, gblock);
        cf v[16];
#pragma unroll
        for (int a = 0; a < 16; a++) v[a] = bld2_sc1(rg, voff, (unsigned)a * gtile);
        dft16<false>(v);
        cf w[16];
#pragma unroll
        for (int p = 0; p < 16; p++) w[p] = ld2(&w256[b * p]);
#pragma unroll
        for (int p = 0; p < 16; p++)
            st2(&tile[(p * 16 + (b ^ (p & 1))) * TC + ((r ^ b) & (TC - 1))], cmul(v[rev16(p)], w[p]));
        __syncthreads();                                             // all G loads of the tile have been consumed too
        if (tid == 0) __hip_atomic_fetch_add(&s2done[m], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int bb = 0; bb < 16; bb++)
            v[bb] = ld2(&tile[(p2 * 16 + (bb ^ (p2 & 1))) * TC + ((r2 ^ bb) & (TC - 1))]);
        dft16<false>(v);
        const unsigned rbytes = (unsigned)((long long)m * lout + tt * TC + r2) * 8u;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const unsigned off = soff[p2 + 16 * q];
            if (off != 0xFFFFFFFFu) bst2(rout, off + rbytes, 0, v[rev16(q)]);
        }
    }
}

hipError_t init_fused_kernels()
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_p1x), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(k_p2x), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
}

size_t fused_ctl_bytes(int nb) { return sizeof(FusedCtl) + sizeof(unsigned) * 2 * (size_t)nb; }
size_t fused_ring_bytes(int R, int ringx) { return (size_t)8 * ringx * (size_t)(256 - 256 / R) * 256 * sizeof(float2); }

// Stage 1 on s1, stage 2 on s2 (ordered after everything s1 had before this call; s1 is joined on s2 afterwards).
// ctl holds fused_ctl_bytes(nb_chunk) bytes and is zeroed here on s1 (memset node, every call) BEFORE the fork event.
hipError_t launch_poly_fused(const float2 *in, size_t in_stride, float2 *gring, float2 *out, int R, int nb_chunk,
                             int mbase, int nb_call, const float2 *tw256, const float2 *twq, const float2 *cbt,
                             const float *shn, const long long *slot_off, unsigned out_bytes, void *ctl, int ringx,
                             int wg1_per_cu, hipStream_t s1, hipStream_t s2, hipEvent_t fork, hipEvent_t join,
                             hipEvent_t *ev /* null or 4: s1 start, s1 end, s2 start, s2 end */)
{
    const int skip = 256 / R, lout = 256 - skip;
    hipError_t e = hipMemsetAsync(ctl, 0, fused_ctl_bytes(nb_chunk), s1);
    if (e != hipSuccess) return e;
    if ((e = hipEventRecord(fork, s1)) != hipSuccess) return e;
    if ((e = hipStreamWaitEvent(s2, fork, 0)) != hipSuccess) return e;
    int dev = 0, ncu = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        ncu = prop.multiProcessorCount;
    if (wg1_per_cu < 1 || wg1_per_cu > 3) wg1_per_cu = 3;
    const size_t lds1 = 256 * kFTC * 8 + 2048 + 2048 + 1024 + 64, lds2 = 256 * kFTC * 8 + 2048 + 1024 + 64;
    if (ev && (e = hipEventRecord(ev[0], s1)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_p1x, dim3((unsigned)(wg1_per_cu * ncu)), dim3(256), lds1, s1, in, in_stride, gring, tw256, twq, cbt,
                       shn, static_cast<FusedCtl *>(ctl), nb_chunk, ringx, skip / 16, lout);
    if (ev && (e = hipEventRecord(ev[1], s1)) != hipSuccess) return e;
    if (ev && (e = hipEventRecord(ev[2], s2)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_p2x, dim3((unsigned)((4 - wg1_per_cu) * ncu)), dim3(256), lds2, s2, gring, out, tw256, slot_off,
                       static_cast<FusedCtl *>(ctl), nb_chunk, ringx, lout, (long long)mbase * lout, (long long)nb_call,
                       out_bytes);
    if (ev && (e = hipEventRecord(ev[3], s2)) != hipSuccess) return e;
    if ((e = hipEventRecord(join, s2)) != hipSuccess) return e;
    if ((e = hipStreamWaitEvent(s1, join, 0)) != hipSuccess) return e;
    return hipGetLastError();
}

}  // namespace fdc
