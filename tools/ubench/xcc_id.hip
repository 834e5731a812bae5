// Probe: HW_REG_XCC_ID per workgroup (placement of consecutive blockIdx over the 8 XCDs) and workgroups per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(unsigned *out)
{
    __shared__ float2 pad[4000];   // ~32 KiB so that 4 workgroups fit a CU like the real kernels
    pad[threadIdx.x] = make_float2(0, 0);
    if (threadIdx.x == 0) {
        const unsigned x = __builtin_amdgcn_s_getreg(20 | (3 << 11));      // HW_REG_XCC_ID[3:0]
        const unsigned hw = __builtin_amdgcn_s_getreg(4 | (31 << 11));     // HW_REG_HW_ID
        out[2 * blockIdx.x] = x; out[2 * blockIdx.x + 1] = hw;
    }
}
int main()
{
    const int n = 1024;
    unsigned *d, h[2 * n]; hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(n), dim3(256), 0, 0, d); hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int cnt[16] = {0};
    for (int i = 0; i < n; i++) cnt[h[2 * i] & 15]++;
    printf("workgroups per XCC id:"); for (int i = 0; i < 16; i++) if (cnt[i]) printf(" [%d]=%d", i, cnt[i]); printf("\n");
    printf("first 24 blockIdx -> xcc:"); for (int i = 0; i < 24; i++) printf(" %u", h[2 * i]); printf("\n");
    printf("hw_id of blockIdx 0,8,16,...,56 (same XCC):"); for (int i = 0; i < 64; i += 8) printf(" %08x", h[2 * i + 1]); printf("\n");
    return 0;
}
