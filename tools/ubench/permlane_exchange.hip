// Microbenchmark (MI355X, gfx950): the 16 x 16 exchange of stage 1 of the block kernels — lane (col, b), register p  ->  lane (col, p),
// register b, over the sixteen lanes col + 4 b of a column — through LDS as shipped (16 ds_write_b64 to a padded strip, 16 ds_read_b64, same wave: no
// barrier) against the same transpose WITHOUT LDS: the gfx950-only v_permlane16_swap_b32 / v_permlane32_swap_b32 for the two lane bits that select a
// DPP row (b bits 2, 3 = lane bits 4, 5: one instruction per dword pair) and DPP row shifts with bank masks for lane bits 2, 3 (three per dword pair),
// and against the "half" form (lane bits 4, 5 by swaps, the rest through LDS).  Each alone, and with a block of packed FMAs per exchange as in a pass
// (the real question: the swaps issue in the VALU, which is the busier pipe of that loop; the LDS route runs beside it).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/permlane_exchange.hip -o /tmp/permlane_exchange && /tmp/permlane_exchange
// Checks the transposes against each other before timing.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#pragma clang diagnostic ignored "-Wunused-value"
typedef float cf __attribute__((ext_vector_type(2)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void swap16(float &a, float &b) { const u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false); a = __uint_as_float(r.x); b = __uint_as_float(r.y); }
__device__ __forceinline__ void swap32(float &a, float &b) { const u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false); a = __uint_as_float(r.x); b = __uint_as_float(r.y); }
// lanes whose bit `LB` (2 or 3) is 1 trade their a with the b of the lane 2^LB below: tmp = a; a <- partner's b (upper lanes); b <- partner's old a (lower lanes)
template <int LB> __device__ __forceinline__ void swap_row(float &a, float &b)
{
    float t = a;
    if constexpr (LB == 2) {
        asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %2 row_shr:4 row_mask:0xf bank_mask:0xa\n\tv_mov_b32_dpp %1, %3 row_shl:4 row_mask:0xf bank_mask:0x5" : "+v"(a), "+v"(b) : "v"(b), "v"(t));
    } else {
        asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %2 row_shr:8 row_mask:0xf bank_mask:0xc\n\tv_mov_b32_dpp %1, %3 row_shl:8 row_mask:0xf bank_mask:0x3" : "+v"(a), "+v"(b) : "v"(b), "v"(t));
    }
}
// the transpose over the lane bits of b: bit k of b pairs registers p and p | (1 << k)
template <int K, class F> __device__ __forceinline__ void layer(cf (&v)[16], F f)
{
#pragma unroll
    for (int p = 0; p < 16; p++) if (!(p & (1 << K))) {
        float ax = v[p].x, ay = v[p].y, bx = v[p | (1 << K)].x, by = v[p | (1 << K)].y;
        f(ax, bx); f(ay, by);
        v[p] = cf{ax, ay}; v[p | (1 << K)] = cf{bx, by};
    }
}
__device__ __forceinline__ void exch_swaps(cf (&v)[16])
{
    layer<0>(v, [](float &a, float &b) { swap_row<2>(a, b); });
    layer<1>(v, [](float &a, float &b) { swap_row<3>(a, b); });
    layer<2>(v, [](float &a, float &b) { swap16(a, b); });
    layer<3>(v, [](float &a, float &b) { swap32(a, b); });
}
__device__ __forceinline__ void exch_lds(cf (&v)[16], float2 *strip, int lane)
{
    const int col = lane & 3, b = lane >> 2;
#pragma unroll
    for (int p = 0; p < 16; p++) *reinterpret_cast<cf *>(&strip[68 * p + lane]) = v[p];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int bb = 0; bb < 16; bb++) v[bb] = *reinterpret_cast<const cf *>(&strip[col + 68 * b + 4 * bb]);
    __builtin_amdgcn_wave_barrier();
}
// half: lane bits 4, 5 by swaps, then a 4 x 4 exchange of register groups among the lanes b & 3 through the strip (every value still makes the trip)
__device__ __forceinline__ void exch_half(cf (&v)[16], float2 *strip, int lane)
{
    layer<2>(v, [](float &a, float &b) { swap16(a, b); });
    layer<3>(v, [](float &a, float &b) { swap32(a, b); });
    const int col = lane & 3, b = lane >> 2;
#pragma unroll
    for (int p = 0; p < 16; p++) *reinterpret_cast<cf *>(&strip[68 * p + lane]) = v[p];
    __builtin_amdgcn_wave_barrier();
    // register p = (p_hi, p_lo): after the swaps p_hi already equals the source's b_hi; what is left: lane bits b_lo <-> p_lo
#pragma unroll
    for (int p = 0; p < 16; p++) v[p] = *reinterpret_cast<const cf *>(&strip[68 * ((p & 12) | (b & 3)) + col + 4 * ((b & 12) | (p & 3))]);
    __builtin_amdgcn_wave_barrier();
}

template <int MODE, int FMA>
__global__ __launch_bounds__(512) void k(float2 *out, int iters, int check)
{
    extern __shared__ float2 smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float2 *strip = smem + w * 1084;
    cf v[16];
#pragma unroll
    for (int p = 0; p < 16; p++) v[p] = cf{(float)(threadIdx.x * 16 + p), (float)(blockIdx.x + 1)};
    const cf c = {0.9999f, 1e-4f};
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) exch_lds(v, strip, lane);
        if (MODE == 1) exch_swaps(v);
        if (MODE == 2) exch_half(v, strip, lane);
#pragma unroll
        for (int r = 0; r < FMA; r++)
#pragma unroll
            for (int p = 0; p < 16; p++) asm volatile("v_pk_fma_f32 %0, %0, %1, %0 op_sel_hi:[1,1,1]" : "+v"(v[p]) : "v"(c));
    }
    if (check || iters)
#pragma unroll
        for (int p = 0; p < 16; p++) out[((size_t)blockIdx.x * 512 + threadIdx.x) * 16 + p] = make_float2(v[p].x, v[p].y);
}

template <int MODE, int FMA> float run(float2 *out, int iters)
{
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE, FMA>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 1084 * 8);
    hipLaunchKernelGGL((k<MODE, FMA>), dim3(256), dim3(512), 8 * 1084 * 8, 0, out, iters, 0);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, FMA>), dim3(256), dim3(512), 8 * 1084 * 8, 0, out, iters, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    float2 *out;
    const size_t n = (size_t)256 * 512 * 16;
    hipMalloc(&out, n * 8);
    // correctness: one exchange each; lane (col, b) register p must hold what lane (col, p) register b started with
    std::vector<float2> h(n);
    int bad[3] = {0, 0, 0};
    for (int mode = 0; mode < 3; mode++) {
        if (mode == 0) hipLaunchKernelGGL((k<0, 0>), dim3(1), dim3(512), 8 * 1084 * 8, 0, out, 1, 1);
        if (mode == 1) hipLaunchKernelGGL((k<1, 0>), dim3(1), dim3(512), 8 * 1084 * 8, 0, out, 1, 1);
        if (mode == 2) hipLaunchKernelGGL((k<2, 0>), dim3(1), dim3(512), 8 * 1084 * 8, 0, out, 1, 1);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), out, 512 * 16 * 8, hipMemcpyDeviceToHost);
        for (int t = 0; t < 512; t++)
            for (int p = 0; p < 16; p++) {
                const int lane = t & 63, col = lane & 3, b = lane >> 2, src = (t & ~63) + col + 4 * p;
                if (h[(size_t)t * 16 + p].x != (float)(src * 16 + b)) bad[mode]++;
            }
    }
    printf("transposes checked: lds %s, swaps + dpp %s, swaps + lds %s\n", bad[0] ? "WRONG" : "ok", bad[1] ? "WRONG" : "ok", bad[2] ? "WRONG" : "ok");
    const int iters = 2000;
    const double per = 1e6 / iters;            // ns per exchange-iteration of the 8 waves of a CU
    printf("one 512-thread workgroup per CU (2 waves per SIMD), %d exchanges per wave; ns per iteration:\n", iters);
    printf("  %-44s alone %7.1f   + 64 pk_fma / lane %7.1f   + 256 pk_fma / lane %7.1f\n", "LDS strip (16 ds_write_b64 + 16 ds_read_b64)", run<0, 0>(out, iters) * per, run<0, 4>(out, iters) * per, run<0, 16>(out, iters) * per);
    printf("  %-44s alone %7.1f   + 64 pk_fma / lane %7.1f   + 256 pk_fma / lane %7.1f\n", "32 permlane swaps + 96 dpp moves, no LDS", run<1, 0>(out, iters) * per, run<1, 4>(out, iters) * per, run<1, 16>(out, iters) * per);
    printf("  %-44s alone %7.1f   + 64 pk_fma / lane %7.1f   + 256 pk_fma / lane %7.1f\n", "32 permlane swaps + the LDS strip", run<2, 0>(out, iters) * per, run<2, 4>(out, iters) * per, run<2, 16>(out, iters) * per);
    printf("  %-44s       %7s   + 64 pk_fma / lane %7.1f   + 256 pk_fma / lane %7.1f\n", "(the FMAs alone)", "", run<3, 4>(out, iters) * per, run<3, 16>(out, iters) * per);
    return (bad[0] || bad[1] || bad[2]) ? 1 : 0;
}
