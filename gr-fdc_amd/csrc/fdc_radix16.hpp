// In-register DFT-16 (two radix-4 layers) for the gfx950 fast kernels.
//
// dft16<INV>(v): v[n], n = 4a+b, is replaced by the 16-point DFT X[k], k = p+4q, which is left in
// DIGIT-REVERSED register order: X[k] sits in v[rev16(k)] with rev16(k) = 4*(k&3) + (k>>2).  Every index
// below is a compile-time constant after unrolling, so the arrays live in VGPRs.
//
// Complex values are clang 2-vectors (cf = float x2) so that complex add/sub/scale lower to the packed
// v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 of gfx950 (one instruction for re and im).
#pragma once
#include <hip/hip_runtime.h>

namespace fdc {

typedef float cf __attribute__((ext_vector_type(2)));

__device__ __forceinline__ constexpr int rev16(int k) { return 4 * (k & 3) + (k >> 2); }

__device__ __forceinline__ cf mk(float x, float y) { cf r; r.x = x; r.y = y; return r; }
__device__ __forceinline__ cf from2(float2 a) { return mk(a.x, a.y); }
__device__ __forceinline__ float2 to2(cf a) { return make_float2(a.x, a.y); }

// (a.x + j a.y)(b.x + j b.y) as two packed ops: t = a.xx*b ; r = a.yy*(-b.y, b.x) + t.  The half swap and the
// sign of the second product are op_sel / neg modifiers (hipcc's C lowering spends a v_xor + v_mov on them).
__device__ __forceinline__ cf cmul(cf a, cf b)
{
    cf r;   // one asm block: hipcc pads an s_nop between separate asm statements
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=&v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ cf cmulc(cf a, cf b)   // a * conj(b)
{
    cf r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]"
        : "=&v"(r) : "v"(a), "v"(b));
    return r;
}
// multiply by -j (forward) / +j (inverse)
template <bool INV>
__device__ __forceinline__ cf mulj(cf d) { return INV ? mk(-d.y, d.x) : mk(d.y, -d.x); }

// a + (-j)*d = (a.x + d.y, a.y - d.x)  and  a - (-j)*d = (a.x - d.y, a.y + d.x) in ONE packed add each: the
// half swap of d is an op_sel, the sign a neg modifier (hipcc emits v_xor + v_mov + v_pk_add for the C form).
__device__ __forceinline__ cf add_mj(cf a, cf d)     // a + (-j) d
{
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(d));
    return r;
}
__device__ __forceinline__ cf add_pj(cf a, cf d)     // a + (+j) d
{
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(d));
    return r;
}

template <bool INV>
__device__ __forceinline__ void dft4(cf &a0, cf &a1, cf &a2, cf &a3)
{
    const cf s0 = a0 + a2, d0 = a0 - a2, s1 = a1 + a3, d1 = a1 - a3;
    a0 = s0 + s1; a2 = s0 - s1;
    // forward: a1 = d0 - j d1, a3 = d0 + j d1 ; inverse: the other way round
    a1 = INV ? add_pj(d0, d1) : add_mj(d0, d1);
    a3 = INV ? add_mj(d0, d1) : add_pj(d0, d1);
}

// multiply by exp(-/+ j*2*pi*e/16), e compile-time
template <bool INV, int E>
__device__ __forceinline__ cf mul_w16(cf v)
{
    constexpr float C1 = 0.92387953251128673848f, S1 = 0.38268343236508978178f, H = 0.70710678118654752440f;
    constexpr int e = E & 15;
    static_assert(e == 0 || e == 1 || e == 2 || e == 3 || e == 4 || e == 5 || e == 6 || e == 7 || e == 9, "unused twiddle");
    if constexpr (e == 0) return v;
    else if constexpr (e == 4) return mulj<INV>(v);
    else {
        // forward twiddle = (c, -s0); inverse = (c, +s0)
        constexpr float c = e == 1 ? C1 : e == 2 ? H : e == 3 ? S1 : e == 5 ? -S1 : e == 6 ? -H : -C1;          // cos(2 pi e / 16); e = 7, 9: -C1
        constexpr float s0 = e == 1 ? S1 : e == 2 ? H : e == 3 ? C1 : e == 5 ? C1 : e == 6 ? H : e == 7 ? S1 : -S1;   // sin(2 pi e / 16)
        constexpr float s = INV ? s0 : -s0;
        return __builtin_elementwise_fma(v.yy, mk(-s, c), v.xx * mk(c, s));
    }
}

template <bool INV>
__device__ __forceinline__ void dft16(cf (&v)[16])
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

// In-register DFT-8: v[n], n = 4 n1 + n0, becomes X[k], k = k0 + 2 k1, left in v[4 k0 + k1].
template <bool INV>
__device__ __forceinline__ void dft8(cf (&v)[8])
{
#pragma unroll
    for (int n0 = 0; n0 < 4; n0++) {
        const cf a = v[n0] + v[n0 + 4], d = v[n0] - v[n0 + 4];
        v[n0] = a; v[n0 + 4] = d;
    }
    v[5] = mul_w16<INV, 2>(v[5]);          // W8^(n0 k0), k0 = 1
    v[6] = mul_w16<INV, 4>(v[6]);
    v[7] = mul_w16<INV, 6>(v[7]);
    dft4<INV>(v[0], v[1], v[2], v[3]);
    dft4<INV>(v[4], v[5], v[6], v[7]);
}

// multiply by exp(-/+ j*2*pi*E/32), 0 <= E < 16 compile-time
template <bool INV, int E>
__device__ __forceinline__ cf mul_w32(cf v)
{
    static_assert(E >= 0 && E < 16, "first half of the circle only");
    if constexpr (E == 0) return v;
    else if constexpr (E == 8) return mulj<INV>(v);
    else {
        constexpr float C[16] = {1.0f, 0.98078528040323044913f, 0.92387953251128673848f, 0.83146961230254523708f,
                                 0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508978178f,
                                 0.19509032201612826785f, 0.0f, -0.19509032201612826785f, -0.38268343236508978178f,
                                 -0.55557023301960222474f, -0.70710678118654752440f, -0.83146961230254523708f,
                                 -0.92387953251128673848f, -0.98078528040323044913f};
        constexpr float c = C[E], s0 = C[(E + 24) & 15] * ((E + 24) & 16 ? -1.0f : 1.0f);   // sin(x) = cos(x - pi/2)
        constexpr float s = INV ? s0 : -s0;
        return __builtin_elementwise_fma(v.yy, mk(-s, c), v.xx * mk(c, s));
    }
}

// In-register DFT-32: v[n], n = 16 n1 + n0, becomes X[k], k = k0 + 2 k1, left in v[16 k0 + rev16(k1)].
template <bool INV, int N0 = 0>
__device__ __forceinline__ void dft32_layer1(cf (&v)[32])
{
    if constexpr (N0 < 16) {
        const cf a = v[N0] + v[N0 + 16], d = v[N0] - v[N0 + 16];
        v[N0] = a; v[N0 + 16] = mul_w32<INV, N0>(d);
        dft32_layer1<INV, N0 + 1>(v);
    }
}
template <bool INV>
__device__ __forceinline__ void dft32(cf (&v)[32])
{
    dft32_layer1<INV>(v);
    dft16<INV>(reinterpret_cast<cf (&)[16]>(v[0]));
    dft16<INV>(reinterpret_cast<cf (&)[16]>(v[16]));
}

}  // namespace fdc
