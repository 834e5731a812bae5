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
// ---- loads the compiler's s_waitcnt pass does not see (round 5) -------------------------------------------------------------------
// hipcc places s_waitcnt vmcnt(N) from a per-register scoreboard that is MERGED over all predecessors of a loop header.  In a persistent
// loop of the form { request the NEXT tile's rows; compute; store this tile } the preheader (nothing younger than the first rows) and
// the back edge (the tile's stores are younger than the prefetched rows) disagree, the merge takes the smaller count, and the loop waits
// with vmcnt(0) — for its own stores of the tile before, every tile; in the block kernels it is vmcnt(1..9) right behind the sixteen
// prefetch loads a pass has just issued.  vmcnt counts loads and stores of one wave in issue order on gfx9 (LLVM's own model: one event
// type on the counter), so the exact wait is known at every point: "at most the number of vector-memory instructions issued after the
// ones I need".  These helpers issue the loads as inline assembly (the scoreboard sees no pending load) and the kernel states the wait
// itself with vm_wait<N>(values), which also ties the values so that no use can be scheduled in front of it.  Rules for a kernel that
// uses them: no compiler-visible load may be pending at a loop header inside which they are used (force such values with vm_settle
// before the loop), and ONE unconditional vm_wait per set of loads, in front of the loop's back edge (two alternative waits make the
// compiler copy the tied registers ahead of the wait: it reads rows that have not landed).
// MEASURED (profiles/r05/NOTES.md section 2, same box, three alternations): the exact waits are worth 1.2 % on the two-launch form at
// N = 65536 and nothing at N = 262144 or on the headline block kernel (tried there, reverted) — a wave that waits early is covered by the
// other waves of its SIMD.  Kept in k_p1 / k_p2 / k_p2k, where it also removes a branch per store; -DFDC_AUTO_WAITS=1 builds the old form.
typedef int srd_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ srd_t make_srd(const void *base, unsigned bytes)
{
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    srd_t r;
    r.x = (int)(unsigned)a; r.y = (int)((unsigned)(a >> 32) & 0xFFFFu); r.z = (int)bytes; r.w = 0x00020000;   // as make_rsrc: stride 0, raw
    return r;
}
// ROUND 6: the SHIPPED build is -DFDC_AUTO_WAITS=1 — compiler-visible loads, the compiler's own waits.  The hand-stated waits are correct only
// while two things hold that nothing checks: the register allocator never copies or spills an asm "=v" result between the load and the
// vm_wait (it believes the value is ready when the asm statement ends), and the number of vector-memory instructions issued behind the loads
// is at least the N of the wait (vm_wait<16 - KEEP> relies on the run-time row skip equalling the template KEEP, vm_wait<16> on exactly 16
// unconditional stores).  A compiler update or a small edit of a kernel would corrupt data silently, for 1.2 % on one path that is not the
// headline.  -DFDC_AUTO_WAITS=0 (tools/build_variant.sh) builds the exact-wait form for A/B runs.
#ifndef FDC_AUTO_WAITS
#define FDC_AUTO_WAITS 1
#endif
template <bool NT>
__device__ __forceinline__ cf ald2(srd_t r, unsigned voff, unsigned soff)
{
#if FDC_AUTO_WAITS
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void *>(((unsigned long long)(unsigned)(r.y & 0xFFFF) << 32) | (unsigned)r.x), 0, r.z, 0x00020000);
    const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rr, voff, soff, NT ? 2 : 0);
    return mk(__uint_as_float(t.x), __uint_as_float(t.y));
#else
    cf v;
    if constexpr (NT) asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen nt" : "=v"(v) : "v"(voff), "s"(r), "s"(soff) : "memory");
    else asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(r), "s"(soff) : "memory");
    return v;
#endif
}
// wait until at most N vector-memory instructions of this wave are outstanding; the listed values are the ones the wait is for
#if FDC_AUTO_WAITS
#define FDC_VMWAIT ""
#else
#define FDC_VMWAIT "s_waitcnt vmcnt(%8)"
#endif
template <int N> __device__ __forceinline__ void vm_wait(cf (&a)[8])
{
    asm volatile(FDC_VMWAIT : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void vm_wait(cf (&a)[16])
{
    asm volatile(FDC_VMWAIT : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "n"(N) : "memory");
    asm volatile("" : "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) :: "memory");
}
// a value from a compiler-visible load, made ready HERE (the compiler puts its own wait in front of this use)
__device__ __forceinline__ void vm_settle(cf &v) { asm volatile("" : "+v"(v) :: "memory"); }

template <bool NT>
__device__ __forceinline__ void bst2t(__amdgpu_buffer_rsrc_t r, unsigned voff, cf v)
{
    u32x2 t;
    t.x = __float_as_uint(v.x); t.y = __float_as_uint(v.y);
    __builtin_amdgcn_raw_buffer_store_b64(t, r, voff, 0, NT ? 2 : 0);
}
__device__ __forceinline__ void st2(float2 *p, cf a) { *reinterpret_cast<cf *>(p) = a; }
// output samples of the wide channel kernels (k_c512 / k_c1024): written once, read by nobody on the device: streamed (nt) stores, round 6:
// -4 % on those kernels (profiles/r06/ch_nt_stores_ab.txt; -DFDC_CH_NT=0 builds the plain form)
#ifndef FDC_CH_NT
#define FDC_CH_NT 1
#endif
__device__ __forceinline__ void st2_out(float2 *p, cf a)
{
#if FDC_CH_NT
    __builtin_nontemporal_store(a, reinterpret_cast<cf *>(p));
#else
    *reinterpret_cast<cf *>(p) = a;
#endif
}

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
