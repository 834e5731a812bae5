// Microbenchmark: a batch of complex DFT-16 the way the kernels do it (in-register butterflies, packed FP32 VALU: fdc_radix16.hpp) against
// the same DFT as a dense matrix product on the f32 matrix pipe.  [Yr; Yi] = [Fr -Fi; Fi Fr] [Xr; Xi] is a 32 x 32 real matrix times a
// 32 x 32 batch (32 transforms per wave and product): sixteen v_mfma_f32_32x32x2f32 (K = 2 each), the operand layout being the one the
// instruction wants (a: row = lane % 32, k = lane / 32; b: col = lane % 32, k = lane / 32; 16 accumulators per lane) — i.e. WITHOUT the
// cost of getting a kernel's data (one transform's 16 points in ONE lane) into that layout and back.
// Build and run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -I gr-fdc_amd/csrc tools/ubench/mfma_dft16.hip -o /tmp/mfma_dft16 && /tmp/mfma_dft16
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "fdc_radix16.hpp"

using fdc::cf;
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k_valu(float2 *out, int iters)
{
    cf v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = fdc::mk(0.001f * (float)(threadIdx.x + i), 0.002f * (float)i);
    for (int it = 0; it < iters; it++) {
        fdc::dft16<false>(v);
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = v[i] * 0.25f;                 // keeps the values bounded; one packed multiply per point
    }
    cf s = v[0];
#pragma unroll
    for (int i = 1; i < 16; i++) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = make_float2(s.x, s.y);
}

__global__ __launch_bounds__(256) void k_mfma(const float *amat /* [16 steps][64 lanes] */, float2 *out, int iters)
{
    const int lane = threadIdx.x & 63;
    float a[16], b[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { a[k] = amat[k * 64 + lane]; b[k] = 0.001f * (float)(lane + k); }
    f16v acc0 = {0}, acc1 = {0};
    for (int it = 0; it < iters; it++) {
        // two independent batches per trip (two accumulator sets): 2 x 32 transforms per wave
#pragma unroll
        for (int k = 0; k < 16; k++) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[k], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[(k + 1) & 15], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 16; k++) b[k] = b[k] * 0.25f + 1e-3f * acc0[k];   // the next batch depends on the result (as a transform chain would)
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) s += acc0[i] + acc1[i];
    out[blockIdx.x * 256 + threadIdx.x] = make_float2(s, 0.f);
}

int main()
{
    const int grid = 256 * 8, iters = 2000;
    float2 *d_out;
    float *d_a;
    hipMalloc(&d_out, sizeof(float2) * grid * 256);
    std::vector<float> am(16 * 64);
    for (int k2 = 0; k2 < 16; k2++)
        for (int lane = 0; lane < 64; lane++) {
            const int row = lane % 32, k = 2 * k2 + lane / 32;                 // element (row, k) of [Fr -Fi; Fi Fr]
            const int r = row % 16, c = k % 16;
            const double ang = -2.0 * M_PI * r * c / 16.0;
            const double fr = std::cos(ang), fi = std::sin(ang);
            am[k2 * 64 + lane] = (float)((row < 16) == (k < 16) ? fr : (row < 16 ? -fi : fi));
        }
    hipMalloc(&d_a, sizeof(float) * am.size());
    hipMemcpy(d_a, am.data(), sizeof(float) * am.size(), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0); hipLaunchKernelGGL(k_valu, dim3(grid), dim3(256), 0, 0, d_out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        const double nv = (double)grid * 256 * iters;
        if (rep) printf("VALU butterflies: %.3f ms for %.3g DFT-16 -> %.1f G DFT-16/s\n", ms, nv, nv / ms / 1e6);
        hipEventRecord(e0); hipLaunchKernelGGL(k_mfma, dim3(grid), dim3(256), 0, 0, d_a, d_out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        const double nm = (double)grid * 4 * 64 * iters;                        // 4 waves x 2 batches x 32 transforms per trip
        if (rep) printf("MFMA 32x32x2 f32: %.3f ms for %.3g DFT-16 -> %.1f G DFT-16/s\n", ms, nm, nm / ms / 1e6);
    }
    return 0;
}
