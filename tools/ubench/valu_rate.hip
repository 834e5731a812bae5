// Microbenchmark (MI355X): issue cost in cycles of the vector instructions the channelizer kernels are made of,
// measured with s_memtime around an unrolled run of independent instructions, 1 and 2 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float cf __attribute__((ext_vector_type(2)));

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

template <int OP>
__global__ __launch_bounds__(512) void k(float2 *out, unsigned long long *cyc, int iters)
{
    cf a0 = {1.0f + threadIdx.x, 0.5f}, a1 = a0 * 1.1f, a2 = a0 * 1.2f, a3 = a0 * 1.3f, a4 = a0 * 1.4f, a5 = a0 * 1.5f, a6 = a0 * 1.6f, a7 = a0 * 1.7f;
    const cf b = {0.999f, 0.001f};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        if (OP == 0) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %1, %1, %8, %1\n v_pk_fma_f32 %2, %2, %8, %2\n v_pk_fma_f32 %3, %3, %8, %3\n v_pk_fma_f32 %4, %4, %8, %4\n v_pk_fma_f32 %5, %5, %8, %5\n v_pk_fma_f32 %6, %6, %8, %6\n v_pk_fma_f32 %7, %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 1) { REP8(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 2) { REP8(asm volatile("v_pk_mul_f32 %0, %0, %8 op_sel_hi:[0,1]\n v_pk_mul_f32 %1, %1, %8 op_sel_hi:[0,1]\n v_pk_mul_f32 %2, %2, %8 op_sel_hi:[0,1]\n v_pk_mul_f32 %3, %3, %8 op_sel_hi:[0,1]\n v_pk_mul_f32 %4, %4, %8 op_sel_hi:[0,1]\n v_pk_mul_f32 %5, %5, %8 op_sel_hi:[0,1]\n v_pk_mul_f32 %6, %6, %8 op_sel_hi:[0,1]\n v_pk_mul_f32 %7, %7, %8 op_sel_hi:[0,1]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 3) { REP8(asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7" : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(b.x));) }
        if (OP == 4) { REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8" : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(b.x));) }
        if (OP == 5) { REP8(asm volatile("v_pk_add_f32 %0, %0, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %1, %1, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %2, %2, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %3, %3, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %4, %4, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %5, %5, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %6, %6, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %7, %7, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 6) { REP8(asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8" : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(b.x));) }
        if (OP == 7) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %0, %0, %8, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    a0 += a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = make_float2(a0.x, a0.y);
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
void run(const char *name, float2 *out, unsigned long long *cyc)
{
    const int iters = 4000;
    for (int threads : {256, 512}) {            // one workgroup per CU (LDS not used; grid = 256): 1 or 2 waves per SIMD
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[8];
        hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
        const double n = 64.0 * iters;
        printf("%-34s %d waves/SIMD: %.2f cycles/instr per wave (s_memtime), kernel %.3f ms -> %.2f GHz-cycles per SIMD instr slot\n", name,
               threads / 256, (double)h[0] / n, ms, ms * 1e-3 * 2.4e9 / (n * (threads / 256)));
    }
}

int main()
{
    float2 *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 512 * 8); hipMalloc(&cyc, 256 * 8 * 8);
    run<0>("v_pk_fma_f32", out, cyc);
    run<1>("v_pk_add_f32", out, cyc);
    run<2>("v_pk_mul_f32 op_sel_hi", out, cyc);
    run<5>("v_pk_add_f32 op_sel+neg (add_mj)", out, cyc);
    run<3>("v_fma_f32", out, cyc);
    run<4>("v_add_f32", out, cyc);
    run<6>("v_mov_b32", out, cyc);
    run<7>("v_pk_fma_f32 dependent chain", out, cyc);
    return 0;
}
