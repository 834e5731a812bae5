// Small device helpers shared by the gfx950 fast kernels: 16-B / 8-B accesses on float2 arrays and buffer (SRD)
// addressing with an optional cache-policy word (aux: 0 = default, 16 = sc1 = device-scope write-through / L1 bypass).
#pragma once
#include <hip/hip_runtime.h>
#include "fdc_radix16.hpp"

namespace fdc {

__device__ __forceinline__ float4 ld4(const float2 *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float2 *p, cf a, cf b)
{
    *reinterpret_cast<float4 *>(p) = make_float4(a.x, a.y, b.x, b.y);
}
__device__ __forceinline__ cf ld2(const float2 *p) { return *reinterpret_cast<const cf *>(p); }

// Buffer addressing (SRD in SGPRs + 32-bit per-lane byte offset + scalar offset): no 64-bit VGPR address per access.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ cf bld2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return mk(__uint_as_float(t.x), __uint_as_float(t.y));
}
// sc1 load: bypasses this CU's L1 and is served by the XCD's L2 (MI355X_MICROARCH.md, "__hip_atomic_load/store ... sc1")
__device__ __forceinline__ cf bld2_sc1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 16);
    return mk(__uint_as_float(t.x), __uint_as_float(t.y));
}
__device__ __forceinline__ void bst2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, cf v)
{
    u32x2 t;
    t.x = __float_as_uint(v.x); t.y = __float_as_uint(v.y);
    __builtin_amdgcn_raw_buffer_store_b64(t, r, voff, soff, 0);
}
// sc1 store: written through to the device-coherent level and dropped from this XCD's L2 (MI355X_MICROARCH.md,
// "stores of each flavour"); with every storing wave's s_waitcnt vmcnt(0) it needs no release fence before a flag.
__device__ __forceinline__ void bst2_sc1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, cf v)
{
    u32x2 t;
    t.x = __float_as_uint(v.x); t.y = __float_as_uint(v.y);
    __builtin_amdgcn_raw_buffer_store_b64(t, r, voff, soff, 16);
}
// nt = streamed once: the line is not kept against data that will be re-read (aux bit 1)
__device__ __forceinline__ cf bld2_nt(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 2);
    return mk(__uint_as_float(t.x), __uint_as_float(t.y));
}
__device__ __forceinline__ void bst2_nt(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, cf v)
{
    u32x2 t;
    t.x = __float_as_uint(v.x); t.y = __float_as_uint(v.y);
    __builtin_amdgcn_raw_buffer_store_b64(t, r, voff, soff, 2);
}
template <bool NT>
__device__ __forceinline__ void bst2t(__amdgpu_buffer_rsrc_t r, unsigned voff, cf v)
{
    u32x2 t;
    t.x = __float_as_uint(v.x); t.y = __float_as_uint(v.y);
    __builtin_amdgcn_raw_buffer_store_b64(t, r, voff, 0, NT ? 2 : 0);
}
__device__ __forceinline__ void st2(float2 *p, cf a) { *reinterpret_cast<cf *>(p) = a; }

// 16-byte buffer store of two complex values (NT: streamed, aux bit 1)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// 16-byte buffer load (two complex values)
__device__ __forceinline__ u32x4 bld4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
}
template <bool NT>
__device__ __forceinline__ void bst4(__amdgpu_buffer_rsrc_t r, unsigned voff, cf a, cf b)
{
    u32x4 t;
    t.x = __float_as_uint(a.x); t.y = __float_as_uint(a.y); t.z = __float_as_uint(b.x); t.w = __float_as_uint(b.y);
    __builtin_amdgcn_raw_buffer_store_b128(t, r, voff, 0, NT ? 2 : 0);
}
// the value of the neighbouring lane (lane ^ 1): DPP quad_perm [1, 0, 3, 2]
__device__ __forceinline__ float swap_pair(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true));
}


}  // namespace fdc
