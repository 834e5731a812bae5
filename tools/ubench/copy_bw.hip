// Calibration: plain streaming copy rates on this MI355X (float4 grid-stride copy; read-only sum; write-only fill)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_copy(const float4 *a, float4 *b, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i]; }
__global__ void k_read(const float4 *a, float4 *b, size_t n) { float4 s = {0,0,0,0}; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = a[i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; } if (s.x == 1234.5f) b[0] = s; }
__global__ void k_fill(float4 *b, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = make_float4(1, 2, 3, 4); }
int main()
{
    const size_t bytes = 1ull << 30, n = bytes / 16;
    float4 *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {2048, 8192, 65536}) {
        for (int mode = 0; mode < 3; mode++) {
            float best = 1e9;
            for (int it = 0; it < 5; it++) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n);
                else if (mode == 1) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, b, n);
                else hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, 0, b, n);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            const double moved = mode == 0 ? 2.0 * bytes : bytes;
            printf("grid %6d %s: %.3f ms  %.2f TB/s\n", grid, mode == 0 ? "copy (r+w)" : mode == 1 ? "read only " : "write only", best, moved / best / 1e9);
        }
    }
    return 0;
}
