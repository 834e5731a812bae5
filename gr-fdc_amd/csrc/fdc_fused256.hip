// Uniform-plan path as ONE persistent dataflow kernel (N = 256*N1 with N1 = 256, all channels l = 256 on the grid).
//
// The two-launch form (k_p1 then k_p2, fdc_fast256.hip) sends the stage-1 output G (lout*N1 points per block) through
// HBM: a launch must be thousands of tiles long to be efficient, so G of a launch never fits a cache.  Here both
// stages live in one launch and workgroups pull TASKS from a global queue in an order that keeps a block's G young:
//     step s:   16 stage-1 tasks (the column tiles of block s),  then ONE stage-2 task (all rows of block s - D)
// so stage 2 of a block runs D blocks (a few MiB of traffic) after its stage 1, and G lives in a small ring that stays in
// the Infinity Cache / L2.  HBM then carries only the compulsory bytes (new input samples in, channel samples out).
//
// Inter-workgroup hand-off (placement independent; /opt/skills/guides/cdna_hip_programming.md Guideline 16):
//   producer (stage-1 task): G stores are sc1 (device-scope write-through) -> every wave `s_waitcnt vmcnt(0)` ->
//       __syncthreads() -> lane 0 relaxed agent-scope fetch_add on the block's counter            (no release fence needed)
//   consumer (stage-2 task): lane 0 polls the counter (relaxed, agent) until 16 -> agent-scope acquire fence ->
//       `s_waitcnt vmcnt(0)` -> __syncthreads() -> plain loads of G
//   ring reuse: a stage-1 task of block m first waits until the stage-2 task of block m - ring has finished reading.
// Tasks are taken strictly in queue order and a task only ever waits for tasks EARLIER in the queue, which have already
// been taken by running workgroups, so the scheme cannot deadlock whatever the residency; every spin is bounded and
// raises an error word instead of hanging.
#include "fdc_kernels.h"
#include "fdc_radix16.hpp"
#include "fdc_devutil.hpp"

namespace fdc {

extern __shared__ __attribute__((aligned(16))) unsigned char fdc_smem_fused[];

constexpr int kFTC = 16;                     // columns per stage-1 tile = rows per stage-2 tile
constexpr unsigned kSpinLimit = 1u << 22;    // bounded spins: ~seconds, then the error word is set

struct FusedCtl {
    unsigned next_task, error, pad0, pad1;
    unsigned flags[1];                       // [nb] stage-1 tiles finished, then [nb] stage-2 finished
};

__device__ __forceinline__ bool wait_geq(unsigned *p, unsigned want)
{
    for (unsigned i = 0; i < kSpinLimit; i++) {
        if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
        __builtin_amdgcn_s_sleep(8);
    }
    return false;
}

// One workgroup = a stream of TILES: a stage-1 task is one tile, a stage-2 task is lout/16 tiles.  While a tile is being
// computed, the loads of the workgroup's next tile are already in flight (register double buffer) whenever that tile
// has no unmet dependency: the next tile of the same stage-2 task, or the single tile of a stage-1 task.
__global__ __launch_bounds__(256, 3) void k_pf(const float2 *__restrict__ in, size_t in_stride,
                                               float2 *__restrict__ gring, float2 *__restrict__ out,
                                               const float2 *__restrict__ tw256, const float2 *__restrict__ twq,
                                               const float2 *__restrict__ cbt, const float *__restrict__ shn,
                                               const long long *__restrict__ slot_off, FusedCtl *ctl, int nb, int D,
                                               int ring, int qskip, int lout, long long out_base, long long nb_call,
                                               unsigned out_bytes)
{
    constexpr int TC = kFTC, N1 = 256;
    float2 *tile = reinterpret_cast<float2 *>(fdc_smem_fused);                           // 256 x 16 points
    float2 *w256 = reinterpret_cast<float2 *>(fdc_smem_fused + 256 * TC * 8);
    float2 *tq = reinterpret_cast<float2 *>(fdc_smem_fused + 256 * TC * 8 + 2048);        // [q][col]
    float *sh = reinterpret_cast<float *>(fdc_smem_fused + 256 * TC * 8 + 2048 + 2048);
    unsigned *soff = reinterpret_cast<unsigned *>(fdc_smem_fused + 256 * TC * 8 + 2048 + 2048 + 1024);
    int *bcast = reinterpret_cast<int *>(fdc_smem_fused + 256 * TC * 8 + 2048 + 2048 + 1024 + 1024);
    const int tid = threadIdx.x;
    w256[tid] = tw256[tid];
    sh[tid] = shn[tid];
    {
        const long long o = slot_off[tid];
        soff[tid] = o >= 0 ? (unsigned)((o * nb_call + out_base) * 8) : 0xFFFFFFFFu;
    }
    unsigned *s1done = ctl->flags, *s2done = ctl->flags + nb;
    const unsigned gtile = (unsigned)lout * TC * 8u;                 // bytes of one (block, column tile) piece of G
    const unsigned gblock = gtile * (N1 / TC);                       // bytes of one block of G
    const int n1full = 16 * D;                                       // tasks before the first stage-2 task
    const int total = 17 * nb;                                       // 16 stage-1 + 1 stage-2 task per block
    const int s2tiles = lout / TC;
    const __amdgpu_buffer_rsrc_t rout = make_rsrc(out, out_bytes);
    // thread roles
    const int col = tid & (TC - 1), b1 = tid / TC;                   // stage 1: column col, rows 16a + b1
    const int r = tid >> 4, b2 = tid & 15;                           // stage 2 layer 1: row r, points n1 = 16a + b2
    const int r2 = tid & (TC - 1), p2 = tid / TC;                    // stage 2 layer 2: row r2, outputs k1 = p2 + 16q
    const unsigned v1row = (unsigned)b1 * N1 * 8u + (unsigned)col * 8u, rowstep = 16u * N1 * 8u;
    const unsigned v2off = (unsigned)(r * TC + b2) * 8u;

    // queue order: steps s = 0 .. nb+D-1; step s holds S1(s, 0..15) if s < nb, then S2(s - D) if s >= D
    auto decode = [&](int t, int &m, int &idx) {
        if (t < n1full) { m = t >> 4; idx = t & 15; return; }
        const int u = t - n1full, mid = 17 * (nb - D);
        if (u < mid) {
            const int s = D + u / 17, rr = u - (s - D) * 17;
            if (rr < 16) { m = s; idx = rr; } else { m = s - D; idx = 16; }
        } else { m = nb - D + (u - mid); idx = 16; }
    };
    auto fetch_task = [&]() -> int {                                 // uniform result; two barriers
        __syncthreads();
        if (tid == 0) bcast[0] = (int)__hip_atomic_fetch_add(&ctl->next_task, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        return bcast[0];
    };
    cf L[16];
    auto load_s1 = [&](int m, int idx) {
        const __amdgpu_buffer_rsrc_t rin = make_rsrc(in + (size_t)m * in_stride, 256u * N1 * 8u);
        const unsigned vo = v1row + (unsigned)idx * TC * 8u;
#pragma unroll
        for (int a = 0; a < 16; a++) L[a] = bld2(rin, vo, a * rowstep);
    };
    auto load_s2 = [&](int m, int tt) {
        const unsigned char *gb = reinterpret_cast<const unsigned char *>(gring) + (size_t)(m % ring) * gblock;
        const __amdgpu_buffer_rsrc_t rg = make_rsrc(gb + (size_t)tt * TC * TC * 8, gblock);
#pragma unroll
        for (int a = 0; a < 16; a++) L[a] = bld2(rg, v2off, (unsigned)a * gtile);
    };

    int t = fetch_task();
    bool have = false;                                               // L already holds the loads of the coming tile
    while (t < total) {
        int m, idx;
        decode(t, m, idx);
        if (idx < 16) {
            // ============================================================ stage-1 task: column tile idx of block m
            const int c0 = idx * TC;
            if (!have) load_s1(m, idx);
            tq[tid] = twq[(size_t)(c0 + col) * 16 + b1];
            const cf cb = ld2(&cbt[(size_t)(c0 + col) * 16 + b1]);
            if (m >= ring && tid == 0 && !wait_geq(&s2done[m - ring], 1u))      // the ring slot must have been read out
                __hip_atomic_store(&ctl->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int tn = fetch_task();                             // next task; also orders the tq writes
            int mn = 0, idn = 16;
            if (tn < total) decode(tn, mn, idn);
            cf v[16];
#pragma unroll
            for (int a = 0; a < 16; a++) v[a] = L[a];
            have = tn < total && idn < 16;
            if (have) load_s1(mn, idn);                              // prefetch: a stage-1 tile has no dependency to wait for
            dft16<false>(v);
            cf w[16];
#pragma unroll
            for (int p = 0; p < 16; p++) w[p] = ld2(&w256[b1 * p]);
#pragma unroll
            for (int p = 0; p < 16; p++) st2(&tile[(16 * b1 + p) * TC + col], cmul(v[rev16(p)], w[p]));
            __syncthreads();
#pragma unroll
            for (int bb = 0; bb < 16; bb++) v[bb] = ld2(&tile[(16 * bb + b1) * TC + col]);
            dft16<false>(v);
            cf u[16];
#pragma unroll
            for (int q = 0; q < 16; q++) u[q ^ 8] = cmul(v[rev16(q)], ld2(&tq[q * TC + col])) * sh[b1 + 16 * q];
            dft16<true>(u);
#pragma unroll
            for (int p = 0; p < 16; p++) w[p] = ld2(&w256[b1 * p]);
#pragma unroll
            for (int p = 0; p < 16; p++) u[rev16(p)] = cmul(cmulc(u[rev16(p)], w[p]), cb);
            __syncthreads();
#pragma unroll
            for (int p = 0; p < 16; p++) st2(&tile[(16 * b1 + p) * TC + col], u[rev16(p)]);
            __syncthreads();
#pragma unroll
            for (int bb = 0; bb < 16; bb++) u[bb] = ld2(&tile[(16 * bb + b1) * TC + col]);
            dft16<true>(u);
            const __amdgpu_buffer_rsrc_t rg = make_rsrc(reinterpret_cast<unsigned char *>(gring) + (size_t)(m % ring) * gblock + (size_t)idx * gtile, gtile);
            const unsigned goff = (unsigned)(b1 * TC + col) * 8u, gstep = 16u * TC * 8u;
#pragma unroll
            for (int q = 0; q < 16; q++)
                if (q >= qskip) bst2_sc1(rg, goff, (unsigned)(q - qskip) * gstep, u[rev16(q)]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave, before the barrier
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(&s1done[m], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            t = tn;
        } else {
            // ============================================================ stage-2 task: all rows of block m
            if (tid == 0) {
                if (!wait_geq(&s1done[m], 16u)) __hip_atomic_store(&ctl->error, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
            load_s2(m, 0);
            int tn = total, mn = 0, idn = 16;
            for (int tt = 0; tt < s2tiles; tt++) {
                cf v[16];
#pragma unroll
                for (int a = 0; a < 16; a++) v[a] = L[a];
                have = false;
                if (tt + 1 < s2tiles) load_s2(m, tt + 1);            // next tile of this task
                else {
                    tn = fetch_task();
                    if (tn < total) decode(tn, mn, idn);
                    have = tn < total && idn < 16;
                    if (have) load_s1(mn, idn);
                }
                dft16<false>(v);
                __syncthreads();
                cf w[16];
#pragma unroll
                for (int p = 0; p < 16; p++) w[p] = ld2(&w256[b2 * p]);
#pragma unroll
                for (int p = 0; p < 16; p++)
                    st2(&tile[(p * 16 + (b2 ^ (p & 1))) * TC + ((r ^ b2) & (TC - 1))], cmul(v[rev16(p)], w[p]));
                __syncthreads();
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
            __syncthreads();                                         // every wave's G loads have returned (consumed)
            if (tid == 0) __hip_atomic_store(&s2done[m], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            t = tn;
        }
    }
}

hipError_t init_fused_kernels()
{
    return hipFuncSetAttribute(reinterpret_cast<const void *>(k_pf), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
}

size_t fused_ctl_bytes(int nb) { return sizeof(unsigned) * (4 + 2 * (size_t)nb); }

// One launch for nb_chunk blocks.  ctl must hold fused_ctl_bytes(nb_chunk) bytes and is zeroed here (memset node on
// the stream, every call); gring holds `ring` blocks of lout*256 points.
hipError_t launch_poly_fused(const float2 *in, size_t in_stride, float2 *gring, float2 *out, int R, int nb_chunk,
                             int mbase, int nb_call, const float2 *tw256, const float2 *twq, const float2 *cbt,
                             const float *shn, const long long *slot_off, unsigned out_bytes, void *ctl, int D, int ring,
                             hipStream_t s)
{
    const int skip = 256 / R, lout = 256 - skip;
    hipError_t e = hipMemsetAsync(ctl, 0, fused_ctl_bytes(nb_chunk), s);
    if (e != hipSuccess) return e;
    int dev = 0, ncu = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        ncu = prop.multiProcessorCount;
    const long long total = 17ll * nb_chunk;
    const unsigned grid = (unsigned)(total < 3ll * ncu ? total : 3ll * ncu);   // 3 workgroups per CU: 168-VGPR budget
    const size_t lds = 256 * kFTC * 8 + 2048 + 2048 + 1024 + 1024 + 64;
    hipLaunchKernelGGL(k_pf, dim3(grid), dim3(256), lds, s, in, in_stride, gring, out, tw256, twq, cbt, shn, slot_off,
                       static_cast<FusedCtl *>(ctl), nb_chunk, D, ring, skip / 16, lout, (long long)mbase * lout,
                       (long long)nb_call, out_bytes);
    return hipGetLastError();
}

}  // namespace fdc
