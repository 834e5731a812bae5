// Does data WRITTEN by one kernel get served from the Infinity Cache when the next kernel reads it? (MI355X)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_read(const float4 *a, float4 *b, size_t n) { float4 s = {0,0,0,0}; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = a[i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; } if (s.x == 1234.5f) b[0] = s; }
__global__ void k_fill(float4 *b, size_t n, float x) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = make_float4(x, 2, 3, 4); }
static float timed(void (*f)(), hipEvent_t e0, hipEvent_t e1) { hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms; }
int main()
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float4 *big, *buf, *sink; hipMalloc(&big, 1ull << 30); hipMalloc(&sink, 4096);
    hipMemset(big, 1, 1ull << 30);
    for (size_t mib : {16, 32, 64, 128, 192}) {
        const size_t bytes = mib << 20, n = bytes / 16;
        hipMalloc(&buf, bytes); hipMemset(buf, 0, bytes);
        float ms; const int grid = 4096;
        // (a) cold read: flush caches by streaming 1 GiB first
        hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, big, sink, (1ull << 30) / 16); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, buf, sink, n); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        const float cold = bytes / ms / 1e9;
        // (b) warm re-read
        hipEventRecord(e0); hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, buf, sink, n); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        const float warm = bytes / ms / 1e9;
        // (c) read right after a write by another kernel (after flushing)
        hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, big, sink, (1ull << 30) / 16); hipDeviceSynchronize();
        hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, 0, buf, n, 1.5f);
        hipEventRecord(e0); hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, buf, sink, n); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        const float afterw = bytes / ms / 1e9;
        // (d) write speed into a buffer that is cache-warm vs cold
        hipEventRecord(e0); hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, 0, buf, n, 2.5f); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        const float wwarm = bytes / ms / 1e9;
        printf("%4zu MiB: read cold %.2f TB/s | re-read %.2f | read after write %.2f | write (warm) %.2f\n", mib, cold, warm, afterw, wwarm);
        hipFree(buf);
    }
    return 0;
}
