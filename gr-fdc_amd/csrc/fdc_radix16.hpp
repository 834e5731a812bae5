// In-register DFT-16 (two radix-4 layers) for the gfx950 fast kernels.
//
// dft16<INV>(v): v[n], n = 4a+b, is replaced by the 16-point DFT X[k], k = p+4q, which is left in
// DIGIT-REVERSED register order: X[k] sits in v[rev16(k)] with rev16(k) = 4*(k&3) + (k>>2).  Every index
// below is a compile-time constant after unrolling, so the arrays live in VGPRs.
#pragma once
#include <hip/hip_runtime.h>

namespace fdc {

__device__ __forceinline__ constexpr int rev16(int k) { return 4 * (k & 3) + (k >> 2); }

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cscale(float2 a, float s) { return make_float2(a.x * s, a.y * s); }

template <bool INV>
__device__ __forceinline__ void dft4(float2 &a0, float2 &a1, float2 &a2, float2 &a3)
{
    const float2 s0 = cadd(a0, a2), d0 = csub(a0, a2), s1 = cadd(a1, a3), d1 = csub(a1, a3);
    // forward: -j*d1 ; inverse: +j*d1
    const float2 jd = INV ? make_float2(-d1.y, d1.x) : make_float2(d1.y, -d1.x);
    a0 = cadd(s0, s1); a1 = cadd(d0, jd); a2 = csub(s0, s1); a3 = csub(d0, jd);
}

// multiply by exp(-/+ j*2*pi*e/16), e compile-time
template <bool INV, int E>
__device__ __forceinline__ float2 mul_w16(float2 v)
{
    constexpr float C1 = 0.92387953251128673848f, S1 = 0.38268343236508978178f, H = 0.70710678118654752440f;
    constexpr int e = E & 15;
    if constexpr (e == 0) return v;
    // forward twiddle = (c, -s); inverse = (c, +s)
    constexpr float c = e == 1 ? C1 : e == 2 ? H : e == 3 ? S1 : e == 4 ? 0.f : e == 6 ? -H : e == 9 ? -C1 : 0.f;
    constexpr float s0 = e == 1 ? S1 : e == 2 ? H : e == 3 ? C1 : e == 4 ? 1.f : e == 6 ? H : e == 9 ? -S1 : 0.f;
    static_assert(e == 1 || e == 2 || e == 3 || e == 4 || e == 6 || e == 9, "unused twiddle");
    constexpr float s = INV ? s0 : -s0;   // imaginary part of the twiddle
    if constexpr (e == 4) return INV ? make_float2(-v.y, v.x) : make_float2(v.y, -v.x);
    return make_float2(v.x * c - v.y * s, v.x * s + v.y * c);
}

template <bool INV>
__device__ __forceinline__ void dft16(float2 (&v)[16])
{
    // layer 1: DFT-4 over a for each b; u[b][p] lands in v[4p+b]
#pragma unroll
    for (int b = 0; b < 4; b++) dft4<INV>(v[b], v[4 + b], v[8 + b], v[12 + b]);
    // W16^(b*p)
    v[4 * 1 + 1] = mul_w16<INV, 1>(v[4 * 1 + 1]);
    v[4 * 1 + 2] = mul_w16<INV, 2>(v[4 * 1 + 2]);
    v[4 * 1 + 3] = mul_w16<INV, 3>(v[4 * 1 + 3]);
    v[4 * 2 + 1] = mul_w16<INV, 2>(v[4 * 2 + 1]);
    v[4 * 2 + 2] = mul_w16<INV, 4>(v[4 * 2 + 2]);
    v[4 * 2 + 3] = mul_w16<INV, 6>(v[4 * 2 + 3]);
    v[4 * 3 + 1] = mul_w16<INV, 3>(v[4 * 3 + 1]);
    v[4 * 3 + 2] = mul_w16<INV, 6>(v[4 * 3 + 2]);
    v[4 * 3 + 3] = mul_w16<INV, 9>(v[4 * 3 + 3]);
    // layer 2: DFT-4 over b for each p; X[p+4q] lands in v[4p+q]
#pragma unroll
    for (int p = 0; p < 4; p++) dft4<INV>(v[4 * p], v[4 * p + 1], v[4 * p + 2], v[4 * p + 3]);
}

}  // namespace fdc
