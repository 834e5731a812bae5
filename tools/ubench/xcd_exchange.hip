// Microbenchmark for the XCD-local form of the N = 262144 shape: can GS workgroups that share one XCD's L2 hand a
// 1 MiB intermediate (G: 128 rows x 1024 columns of float2) to each other through that L2 — column slices in, row slices out — faster
// than the round trip through memory that the two-launch form pays?
//   * one 512-thread workgroup per compute unit (LDS request), workgroup w on XCD w % 8 (checked with HW_REG_XCC_ID);
//   * the 32 workgroups of an XCD form 32 / GS groups; per "block" a workgroup writes its 1024 / GS columns of all 128 rows (plain
//     stores, runs of 8192 / GS bytes), announces it (one atomic per workgroup on the group's counter — atomics execute in the L2),
//     waits for the group, reads its 128 / GS rows (sc1 loads: past the compute unit's L1) and checks a checksum;
//   * buffers are double: a workgroup may start writing block b + 2 only when the group has read block b.
// No agent-scope release between the writes and the flag (that would write the whole L2 back, which is what the experiment wants to
// avoid): the stores are only waited for (they have then reached the L2), which is enough INSIDE one XCD and nowhere else.
// Every wait is bounded (the kernel gives up and reports it rather than hang).
// Build and run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/ubench/xcd_exchange.hip -o /tmp/xcd_exchange && /tmp/xcd_exchange
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

constexpr int kRows = 128, kCols = 1024;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// 16-byte load past the compute unit's L1 (sc1), from the XCD's L2
__device__ __forceinline__ u32x4 ld16_sc1(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16); }
__device__ __forceinline__ unsigned long long word_of(int b, int r, int c) { return (unsigned long long)(b * 1315423911u + r * 2654435761u + c); }

template <int GS>
__global__ __launch_bounds__(512) void k_exchange(unsigned long long *gbuf /* [groups][2][128][1024] */, unsigned *arrive, unsigned *done,
                                                  unsigned *xcc, unsigned *fail, unsigned long long *sums, int nblk)
{
    extern __shared__ unsigned char pad[];                         // forces one workgroup per compute unit
    __shared__ int stop;                                           // a give-up is seen by the whole workgroup at once
    if (threadIdx.x == 0) { pad[0] = 0; stop = 0; }
    __syncthreads();
    const int w = blockIdx.x, x = w & 7, slot = w >> 3, gi = slot / GS, mi = slot % GS;
    const int group = x * (32 / GS) + gi;
    if (threadIdx.x == 0) xcc[w] = __builtin_amdgcn_s_getreg(20 | (3 << 11));
    unsigned long long *G = gbuf + (size_t)group * 2 * kRows * kCols;
    unsigned *arr = arrive + (size_t)group * nblk, *dn = done + (size_t)group * nblk;
    constexpr int cw = kCols / GS, rw = kRows / GS;
    unsigned long long acc = 0;
    for (int b = 0; b < nblk; b++) {
        unsigned long long *buf = G + (size_t)(b & 1) * kRows * kCols;
        if (b >= 2) {                                               // the buffer is free once the group has read block b - 2
            int spins = 0;
            if (threadIdx.x == 0) while (__hip_atomic_load(&dn[b - 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < GS && ++spins < 2000000) __builtin_amdgcn_s_sleep(2);
            if (threadIdx.x == 0 && (spins >= 2000000 || *(volatile unsigned *)fail)) { *fail = 1; stop = 1; }
            __syncthreads();
            if (stop) return;
        }
        // column slice: rows 0..127, columns mi * cw .. + cw - 1, two words (16 bytes) per lane and store
        for (int e = threadIdx.x; e < kRows * cw / 2; e += 512) {
            const int r = e / (cw / 2), c = mi * cw + 2 * (e % (cw / 2));
            const unsigned long long a0 = word_of(b, r, c), a1 = word_of(b, r, c + 1);
            *reinterpret_cast<ulonglong2 *>(&buf[(size_t)r * kCols + c]) = make_ulonglong2(a0, a1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the stores have reached the L2
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(&arr[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (__hip_atomic_load(&arr[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < GS && ++spins < 2000000) __builtin_amdgcn_s_sleep(2);
            if (spins >= 2000000 || *(volatile unsigned *)fail) { *fail = 2; stop = 1; }
        }
        __syncthreads();
        if (stop) return;
        // row slice: rows mi * rw .. + rw - 1, all 1024 columns
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, kRows * kCols * 8, 0x00020000);
        for (int e = threadIdx.x; e < rw * kCols / 2; e += 512) {
            const int r = mi * rw + e / (kCols / 2), c = 2 * (e % (kCols / 2));
            const u32x4 v = ld16_sc1(rs, (unsigned)(((size_t)r * kCols + c) * 8));
            acc += (((unsigned long long)v.y << 32 | v.x) ^ word_of(b, r, c)) + (((unsigned long long)v.w << 32 | v.z) ^ word_of(b, r, c + 1));   // 0 when coherent
        }
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(&dn[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // wave sum of the mismatch accumulator
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&sums[w], acc);
}

template <int GS>
static void run(int nblk)
{
    const int grid = 256, groups = grid / GS;
    unsigned long long *gbuf, *sums;
    unsigned *arrive, *done, *xcc, *fail;
    hipMalloc(&gbuf, sizeof(unsigned long long) * (size_t)groups * 2 * kRows * kCols);
    hipMalloc(&arrive, sizeof(unsigned) * (size_t)groups * nblk); hipMalloc(&done, sizeof(unsigned) * (size_t)groups * nblk);
    hipMalloc(&xcc, sizeof(unsigned) * grid); hipMalloc(&fail, sizeof(unsigned)); hipMalloc(&sums, sizeof(unsigned long long) * grid);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_exchange<GS>), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipMemset(arrive, 0, sizeof(unsigned) * (size_t)groups * nblk); hipMemset(done, 0, sizeof(unsigned) * (size_t)groups * nblk);
        hipMemset(fail, 0, sizeof(unsigned)); hipMemset(sums, 0, sizeof(unsigned long long) * grid);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_exchange<GS>, dim3(grid), dim3(512), 120 * 1024, 0, gbuf, arrive, done, xcc, fail, sums, nblk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned> hx(grid); unsigned hf; std::vector<unsigned long long> hs(grid);
    hipMemcpy(hx.data(), xcc, sizeof(unsigned) * grid, hipMemcpyDeviceToHost); hipMemcpy(&hf, fail, sizeof hf, hipMemcpyDeviceToHost);
    hipMemcpy(hs.data(), sums, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost);
    int badx = 0; unsigned long long mism = 0;
    for (int w = 0; w < grid; w++) { if ((hx[w] & 15) != (hx[w & 7] & 15)) badx++; mism += hs[w]; }
    const double bytes = (double)groups * nblk * kRows * kCols * 8.0;      // G per block, once in, once out
    printf("GS %2d (%d blocks in flight per XCD): %.3f ms for %d blocks per group -> G exchanged at %.2f TB/s in + %.2f TB/s out; "
           "workgroups off their XCD: %d, give-ups: %u, mismatching words: %llu\n", GS, 32 / GS, ms, nblk, bytes / ms / 1e9, bytes / ms / 1e9, badx, hf, mism);
    hipFree(gbuf); hipFree(arrive); hipFree(done); hipFree(xcc); hipFree(fail); hipFree(sums);
}

int main()
{
    run<8>(64);
    run<16>(64);
    run<32>(64);
    return 0;
}
