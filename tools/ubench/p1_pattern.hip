// Microbenchmark (MI355X): the MEMORY side of stage 1 at BASELINE configs[3] (N = 262144 = 1024 columns x 256 rows, R = 2)
// with nothing else in the loop.  A persistent workgroup owns a column tile of TC columns and a run of consecutive blocks; per
// block it loads the 128 NEW rows of its tile (row pitch 8 KiB, TC*8 bytes per row piece), one block ahead, and stores 128 rows of
// G tile-major (one contiguous run of 128*TC*8 bytes) — what k_p1<TC,0,8> moves.  Variants:
//   TC     columns per tile (16 = shipped at this size: 128-byte row pieces)
//   VW     complex values per lane and access (1 = 8-byte accesses as shipped, 2 = 16-byte accesses)
//   MAP    0 = blockIdx -> column tile fastest (shipped: an XCD sees the tiles ct = xcd mod 8 of EVERY block)
//          1 = blockIdx -> group fastest (an XCD sees whole rows of ITS blocks)
//   DEPTH  blocks of loads in flight ahead of the stores (1 = shipped)
//   NTL    input loads carry the nt hint (shipped: yes)
// Reported: microseconds per 256 blocks and TB/s of the 554 MB moved; a float4 copy of the same bytes for the box's copy rate.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/p1_pattern.hip -o /tmp/p1_pattern && /tmp/p1_pattern
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
}
template <int VW> struct Vec;
template <> struct Vec<1> { typedef u32x2 T; };
template <> struct Vec<2> { typedef u32x4 T; };
template <int VW, bool NT>
__device__ __forceinline__ typename Vec<VW>::T ldv(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    if constexpr (VW == 1) return __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, NT ? 2 : 0);
    else return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, NT ? 2 : 0);
}
template <int VW>
__device__ __forceinline__ void stv(typename Vec<VW>::T v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    if constexpr (VW == 1) __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, 0);
    else __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, 0);
}

constexpr int N1 = 1024, ROWS = 256, LOUT = 128;

template <int TC, int VW, int MAP, int DEPTH, bool NTL>
__global__ __launch_bounds__(16 * TC / VW) void k(const float2 *in, size_t stride, float2 *g, int nb, int bpg)
{
    typedef typename Vec<VW>::T V;
    constexpr int LPR = TC / VW;                       // lanes per row piece
    constexpr int CT = N1 / TC;
    const int tid = threadIdx.x, cl = tid % LPR, b = tid / LPR;
    const int ngrp = gridDim.x / CT;
    const int cti = MAP == 0 ? blockIdx.x % CT : blockIdx.x / ngrp;
    const int grp = MAP == 0 ? blockIdx.x / CT : blockIdx.x % ngrp;
    const unsigned voff = (unsigned)(b * N1 + cti * TC + cl * VW) * 8u;
    const unsigned rowstep = 16u * N1 * 8u;
    const unsigned goff = (unsigned)(b * TC + cl * VW) * 8u, gstep = 16u * TC * 8u;
    const int m0 = grp * bpg, m1 = m0 + bpg < nb ? m0 + bpg : nb;
    V buf[DEPTH + 1][8];
    auto issue = [&](int m, V (&d)[8]) {
        const __amdgpu_buffer_rsrc_t r = rsrc(in + (size_t)m * stride, (unsigned)ROWS * N1 * 8u);
#pragma unroll
        for (int a = 0; a < 8; a++) d[a] = ldv<VW, NTL>(r, voff, (unsigned)(a + 8) * rowstep);
    };
#pragma unroll
    for (int d = 0; d < DEPTH; d++) if (m0 + d < m1) issue(m0 + d, buf[d]);
    for (int m = m0; m < m1; m += DEPTH + 1) {
#pragma unroll
        for (int s = 0; s <= DEPTH; s++) {             // slot s holds block m + s; slot (s + DEPTH) % (DEPTH + 1) is free for m + s + DEPTH
            const int mm = m + s;
            if (mm < m1) {
                if (mm + DEPTH < m1) issue(mm + DEPTH, buf[(s + DEPTH) % (DEPTH + 1)]);
                const __amdgpu_buffer_rsrc_t rg = rsrc(g + ((size_t)mm * CT + cti) * (size_t)LOUT * TC, (unsigned)LOUT * TC * 8u);
#pragma unroll
                for (int q = 0; q < 8; q++) stv<VW>(buf[s][q], rg, goff, (unsigned)q * gstep);
            }
        }
    }
}

__global__ void k_copy(const float4 *a, float4 *b, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

static float2 *g_in[3], *g_g;
static const int nb = 256;
static const size_t H = 131072;

template <int TC, int VW, int MAP, int DEPTH, bool NTL>
static void run()
{
    constexpr int NT = 16 * TC / VW, CT = N1 / TC;
    const int wgs = (1024 / NT) * 256;
    int groups = wgs / CT; if (groups > nb) groups = nb;
    const int bpg = (nb + groups - 1) / groups;
    const unsigned grid = (unsigned)(groups * CT);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 6; i++) hipLaunchKernelGGL((k<TC, VW, MAP, DEPTH, NTL>), dim3(grid), dim3(NT), 0, 0, g_in[i % 3], H, g_g, nb, bpg);
    hipEventRecord(e0, 0);
    const int reps = 30;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k<TC, VW, MAP, DEPTH, NTL>), dim3(grid), dim3(NT), 0, 0, g_in[i % 3], H, g_g, nb, bpg);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double bytes = (double)nb * (H + (double)LOUT * N1) * 8;
    printf("TC %3d  %2d B/lane  map %d  depth %d  nt %d  threads %4d  grid %4u  bpg %2d : %7.1f us  %.2f TB/s\n", TC, 8 * VW, MAP, DEPTH,
           (int)NTL, NT, grid, bpg, ms * 1e3, bytes / ms / 1e9);
    fflush(stdout);
}

int main()
{
    for (int i = 0; i < 3; i++) { hipMalloc(&g_in[i], sizeof(float2) * ((size_t)nb * H + 2 * H)); hipMemset(g_in[i], 1, sizeof(float2) * ((size_t)nb * H + 2 * H)); }
    hipMalloc(&g_g, sizeof(float2) * (size_t)nb * LOUT * N1);
    {   // the box's copy rate on the same bytes
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const size_t n = (size_t)nb * H / 2;
        for (int i = 0; i < 6; i++) hipLaunchKernelGGL(k_copy, dim3(256 * 8), dim3(256), 0, 0, (const float4 *)g_in[i % 3], (float4 *)g_g, n);
        hipEventRecord(e0, 0);
        for (int i = 0; i < 30; i++) hipLaunchKernelGGL(k_copy, dim3(256 * 8), dim3(256), 0, 0, (const float4 *)g_in[i % 3], (float4 *)g_g, n);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 30;
        printf("float4 copy of 268 MB -> 268 MB: %.1f us  %.2f TB/s\n", ms * 1e3, 2.0 * n * 16 / ms / 1e9);
    }
    run<16, 1, 0, 1, true>();      // shipped shape
    run<16, 1, 0, 1, false>();
    run<16, 1, 1, 1, true>();      // XCD by group
    run<16, 1, 0, 2, true>();      // two blocks ahead
    run<16, 1, 1, 2, true>();
    run<16, 1, 0, 3, true>();
    run<32, 1, 0, 1, true>();      // 256-byte pieces
    run<32, 1, 1, 1, true>();
    run<32, 1, 0, 2, true>();
    run<64, 1, 0, 1, true>();      // 512-byte pieces, 1024 threads
    run<64, 1, 1, 1, true>();
    run<32, 2, 0, 1, true>();      // 16-byte accesses
    run<32, 2, 1, 1, true>();
    run<32, 2, 0, 2, true>();
    run<64, 2, 0, 1, true>();
    run<64, 2, 1, 1, true>();
    run<64, 2, 1, 2, true>();
    run<128, 2, 0, 1, true>();     // 1-KiB pieces
    run<128, 2, 1, 1, true>();
    run<128, 2, 1, 2, true>();
    run<16, 1, 0, 1, true>();      // shipped shape again (drift check)
    return 0;
}
