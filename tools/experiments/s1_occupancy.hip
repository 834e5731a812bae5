// Experiment (MI355X): what stage 1 of the block kernel (fdc_block256.hip) would gain from four waves per SIMD.
// The same per-pass work — 16 row loads per lane, DFT-16, exchange, DFT-16, window/twiddle, DFT-16, exchange, DFT-16, eight
// values kept — for one 65536-sample block per workgroup, stage 1 ONLY (no stage 2, the kept values are summed and stored
// once), in two shapes:
//   W = 8  : 512 threads, 8 passes of 32 columns, 2 waves per SIMD, next pass prefetched into registers (the shipped shape);
//   W = 16 : 1024 threads, 4 passes of 64 columns, 4 waves per SIMD, 128 VGPRs per lane: no room for a prefetch, the waves
//            cover each other's load latency instead.  G would take 64 of those 128 registers (it does here: the kept values
//            are held in a 4-element vector per row group, as the kernel would).
// Tables hold unit-magnitude values; the result is not a channelizer output, only the instruction stream is the same.
//   hipcc -O3 --offload-arch=gfx950 -I gr-fdc_amd/csrc tools/experiments/s1_occupancy.hip -o tools/experiments/s1_occupancy.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include "fdc_radix16.hpp"
#include "fdc_devutil.hpp"

using namespace fdc;

extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
typedef unsigned long long u8v __attribute__((ext_vector_type(8)));
typedef unsigned long long u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned long long pk(cf v) { return ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x); }
__device__ __forceinline__ cf upk(unsigned long long u) { return mk(__uint_as_float((unsigned)u), __uint_as_float((unsigned)(u >> 32))); }

constexpr int kScrPts = 1084;

// DMA = true (W = 8 only, round 3): the rows of the next pass do not wait in 32 VGPRs but in LDS — every wave fills a private
// 8-KiB strip ([256 rows][4 columns]) with eight `buffer_load_dwordx4 ... lds` (two lanes per row: 32 rows per instruction) and
// reads its 16 values per lane back at the top of the next pass.  Private strips: no workgroup barrier, the waves keep drifting.
template <int W, bool DMA = false>
__global__ __attribute__((target("no-load-store-opt"))) __launch_bounds__(64 * W) void k_s1(const float2 *__restrict__ in, size_t in_stride,
                                                                                                 float2 *__restrict__ out, const float2 *__restrict__ tab,
                                                                                                 int nb)
{
    constexpr int PASSES = 256 / (4 * W), COLS = 4 * W;
    using GV = typename std::conditional<W == 8, u8v, u4v>::type;
    float2 *scr = reinterpret_cast<float2 *>(smem);
    float2 *wrow = scr + W * kScrPts;                       // [16][18]
    float2 *Bt = wrow + 16 * 18;                            // [COLS][18]
    float2 *SA = Bt + COLS * 18;                            // [PASSES][16][18]
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, col = lane & 3, b = lane >> 2, c5 = 4 * w + col;
    for (int i = tid; i < 16 * 18; i += 64 * W) wrow[i] = tab[i & 255];
    for (int i = tid; i < COLS * 18; i += 64 * W) Bt[i] = tab[(i * 7) & 255];
    for (int i = tid; i < PASSES * 16 * 18; i += 64 * W) SA[i] = tab[(i * 3) & 255];
    __syncthreads();
    const int grid = gridDim.x, per = grid >> 3;
    const int first = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    const unsigned voff = (unsigned)(b * 256 + c5) * 8u;
    float2 *const scrw = scr + w * kScrPts + lane;
    const float2 *const scrr = scr + w * kScrPts + col + 68 * b;
    const float2 *const wr = wrow + b * 18;
    const float2 *const btr = Bt + c5 * 18;
    cf acc = mk(0.f, 0.f);
    cf L[16];
    // DMA: this wave's strip behind the tables; lane -> (row inside an instruction's 32 rows, which half of the 32 bytes)
    float2 *const dstrip = SA + PASSES * 16 * 18 + w * 1024;
    const unsigned voffd = (unsigned)((lane >> 1) * 2048 + (lane & 1) * 16);
    auto dma_issue = [&](int mb, int pn) {
        const __amdgpu_buffer_rsrc_t rin = make_rsrc(in + (size_t)mb * in_stride + COLS * pn + 4 * w, 65536u * 8u);
#pragma unroll
        for (int i = 0; i < 8; i++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void *)(dstrip + i * 128), 16, voffd, (unsigned)i * 65536u, 0, 0);
    };
    if (DMA) dma_issue(first, 0);
    else if (W == 8) {
        const __amdgpu_buffer_rsrc_t rin = make_rsrc(in + (size_t)first * in_stride, 65536u * 8u);
#pragma unroll
        for (int a = 0; a < 16; a++) L[a] = bld2(rin, voff, (unsigned)a * 32768u);
    }
    for (int m = first; m < nb; m += grid) {
        const int mnext = m + grid < nb ? m + grid : m;
        GV G[8];
#pragma nounroll
        for (int ps = 0; ps < PASSES; ps++) {
            cf cur[16];
            if (DMA) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the strip is filled
#pragma unroll
                for (int a = 0; a < 16; a++) cur[a] = ld2(&dstrip[(16 * a + b) * 4 + col]);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // ... and read out: the next pass may overwrite it
                dma_issue(ps < PASSES - 1 ? m : mnext, ps < PASSES - 1 ? ps + 1 : 0);
            } else if (W == 8) {
#pragma unroll
                for (int a = 0; a < 16; a++) cur[a] = L[a];
                const int pn = ps < PASSES - 1 ? ps + 1 : 0;
                const __amdgpu_buffer_rsrc_t rin = make_rsrc(in + (size_t)(ps < PASSES - 1 ? m : mnext) * in_stride + COLS * pn, 65536u * 8u);
#pragma unroll
                for (int a = 0; a < 16; a++) L[a] = bld2(rin, voff, (unsigned)a * 32768u);
            } else {
                const __amdgpu_buffer_rsrc_t rin = make_rsrc(in + (size_t)m * in_stride + COLS * ps, 65536u * 8u);
#pragma unroll
                for (int a = 0; a < 16; a++) cur[a] = bld2(rin, voff, (unsigned)a * 32768u);
            }
            dft16<false>(cur);
            st2(&scrw[0], cur[rev16(0)]);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const float4 t = ld4(&wr[2 * i]);
                if (i) st2(&scrw[68 * (2 * i)], cmul(cur[rev16(2 * i)], mk(t.x, t.y)));
                st2(&scrw[68 * (2 * i + 1)], cmul(cur[rev16(2 * i + 1)], mk(t.z, t.w)));
            }
            __builtin_amdgcn_wave_barrier();
            cf v[16];
#pragma unroll
            for (int bb = 0; bb < 16; bb++) v[bb] = ld2(&scrr[4 * bb]);
            dft16<false>(v);
            cf u[16];
            {
                const float2 *sar = SA + (ps * 16 + b) * 18;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const float4 t0 = ld4(&btr[2 * i]), t1 = ld4(&sar[2 * i]);
                    u[(2 * i) ^ 8] = cmul(cmul(v[rev16(2 * i)], mk(t0.x, t0.y)), mk(t1.x, t1.y));
                    u[(2 * i + 1) ^ 8] = cmul(cmul(v[rev16(2 * i + 1)], mk(t0.z, t0.w)), mk(t1.z, t1.w));
                }
            }
            dft16<true>(u);
            const cf cb = mk(0.8f, 0.6f);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const float4 t = ld4(&wr[2 * i]);
                st2(&scrw[68 * (2 * i)], i ? cmul(cmulc(u[rev16(2 * i)], mk(t.x, t.y)), cb) : cmul(u[rev16(0)], cb));
                st2(&scrw[68 * (2 * i + 1)], cmul(cmulc(u[rev16(2 * i + 1)], mk(t.z, t.w)), cb));
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int bb = 0; bb < 16; bb++) u[bb] = ld2(&scrr[4 * bb]);
            dft16<true>(u);
#pragma unroll
            for (int j = 0; j < 8; j++) G[j][ps] = pk(u[rev16(8 + j)]);
        }
#pragma unroll
        for (int j = 0; j < 8; j++)
#pragma unroll
            for (int ps = 0; ps < PASSES; ps++) acc += upk(G[j][ps]);
    }
    out[(size_t)blockIdx.x * 64 * W + tid] = to2(acc);
}

template <int W, bool DMA = false>
static void run(const char *name, const float2 *in, float2 *out, const float2 *tab, int nb)
{
    const int lds = (W * kScrPts + 16 * 18 + 4 * W * 18 + (256 / (4 * W)) * 16 * 18 + (DMA ? 8 * 1024 : 0)) * 8;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_s1<W, DMA>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k_s1<W, DMA>), dim3(256), dim3(64 * W), lds, 0, in, (size_t)32768, out, tab, nb);
    hipEventRecord(e0, 0);
    const int reps = 20;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_s1<W, DMA>), dim3(256), dim3(64 * W), lds, 0, in, (size_t)32768, out, tab, nb);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s LDS %6d B   %.4f ms per %d blocks (stage 1 only)   %s\n", name, lds, ms / reps, nb, hipGetErrorString(hipGetLastError()));
}

int main()
{
    const int nb = 1024;
    float2 *in, *out, *tab;
    hipMalloc(&in, sizeof(float2) * ((size_t)nb * 32768 + 32768)); hipMalloc(&out, sizeof(float2) * 256 * 1024); hipMalloc(&tab, sizeof(float2) * 256);
    std::vector<float2> h((size_t)nb * 32768 + 32768);
    for (size_t i = 0; i < h.size(); i++) h[i] = make_float2((float)((i * 2654435761u) & 1023) / 1024.f - 0.5f, (float)((i * 40503u) & 1023) / 1024.f - 0.5f);
    hipMemcpy(in, h.data(), sizeof(float2) * h.size(), hipMemcpyHostToDevice);
    std::vector<float2> t(256);
    for (int i = 0; i < 256; i++) t[i] = make_float2((float)std::cos(2 * M_PI * i / 256), (float)-std::sin(2 * M_PI * i / 256));
    hipMemcpy(tab, t.data(), sizeof(float2) * 256, hipMemcpyHostToDevice);
    run<8>("W = 8  (512 threads, 2 waves per SIMD)", in, out, tab, nb);
    run<16>("W = 16 (1024 threads, 4 waves per SIMD)", in, out, tab, nb);
    run<8, true>("W = 8, next pass by LDS-DMA (no prefetch VGPRs)", in, out, tab, nb);
    run<8>("W = 8  again", in, out, tab, nb);
    return 0;
}
