// Microbenchmark (MI355X): do VALU work and LDS traffic from DIFFERENT waves of one CU overlap?
// mode 0: all waves VALU only; 1: all waves LDS only; 2: even workgroup-waves VALU, odd waves LDS (half work each)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float cf __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256, 4) void k(int mode, int iters, float2 *out)
{
    __shared__ float2 lds[256 * 16];
    const int tid = threadIdx.x, wave = tid >> 6;
    cf a = {1.0f + tid, 0.5f}, b = {0.99f, 0.01f}, c = {0.f, 0.f};
    cf v[8];
    for (int i = 0; i < 8; i++) v[i] = a * (float)(i + 1);
    const bool do_valu = mode == 0 || (mode == 2 && (wave & 1) == 0);
    const bool do_lds = mode == 1 || (mode == 2 && (wave & 1) == 1);
    for (int it = 0; it < iters; it++) {
        if (do_valu) {
#pragma unroll
            for (int r = 0; r < 16; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = __builtin_elementwise_fma(v[i], b, a);
        }
        if (do_lds) {
#pragma unroll
            for (int r = 0; r < 16; r++) *reinterpret_cast<cf *>(&lds[(r * 16 + (tid >> 4)) * 16 + (tid & 15)]) = v[r & 7];
            __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
            for (int r = 0; r < 16; r++) c += *reinterpret_cast<cf *>(&lds[((tid >> 4) * 16 + r) * 16 + (tid & 15)]);
        }
    }
    for (int i = 0; i < 8; i++) c += v[i];
    out[blockIdx.x * 256 + tid] = make_float2(c.x, c.y);
}
int main()
{
    float2 *out; hipMalloc(&out, 1024 * 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; mode++) {
        const int iters = 2000;
        hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, mode, 10, out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, mode, iters, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // per iteration per wave: VALU 128 pk_fma; LDS 16 b64 writes + 16 b64 reads
        printf("mode %d: %.3f ms  (%.1f cycles/iter at 2.1 GHz, 4 WG/CU x 4 waves)\n", mode, ms, ms * 1e-3 * 2.1e9 / iters);
    }
    return 0;
}
