// Operand helpers shared by the streamed-operand GEMMs (pw_b16.hip, pw_ws.hip): bf16 fragment types, the exact three-way split of
// an fp32 value (k_pw_bx3's decomposition), and a lane's A operand of one K slab as loaded from global memory.
#pragma once
#include "pw_common.h"

namespace bnhip {

typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 b16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32v4 __attribute__((ext_vector_type(4)));
typedef unsigned u32v2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ b16x8 b16_cvt8(const float4& a, const float4& b) {       // round to nearest even, 8 values
    u32v4 h;
    h[0] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a.x, a.y}, b16x2));
    h[1] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a.z, a.w}, b16x2));
    h[2] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){b.x, b.y}, b16x2));
    h[3] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){b.z, b.w}, b16x2));
    return __builtin_bit_cast(b16x8, h);
}
// fp32 -> three bf16 pieces, exactly k_pw_bx3's decomposition (hi = RNE(x), mid = RNE(x - hi), lo = RNE(x - hi - mid); the
// subtractions are exact and kept scalar: packed they cost ~13 cycles beside MFMAs against ~4)
__device__ __forceinline__ float b16_sub(float a, float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void b16_split8(const float4& a, const float4& b, b16x8* hi, b16x8* mid, b16x8* lo) {
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    u32v4 h, m, l;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const f32x2 v = {x[2 * q], x[2 * q + 1]};
        const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(v, b16x2));
        const f32x2 r = {b16_sub(v[0], __uint_as_float(hb << 16)), b16_sub(v[1], __uint_as_float(hb & 0xffff0000u))};
        const unsigned mb = __builtin_bit_cast(unsigned, __builtin_convertvector(r, b16x2));
        const f32x2 t = {b16_sub(r[0], __uint_as_float(mb << 16)), b16_sub(r[1], __uint_as_float(mb & 0xffff0000u))};
        h[q] = hb; m[q] = mb; l[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(t, b16x2));
    }
    *hi = __builtin_bit_cast(b16x8, h); *mid = __builtin_bit_cast(b16x8, m); *lo = __builtin_bit_cast(b16x8, l);
}
__device__ __forceinline__ float4 b16_unpack4(const u32v2& r) {
    return make_float4(__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xffff0000u), __uint_as_float(r[1] << 16),
                       __uint_as_float(r[1] & 0xffff0000u));
}

// one slab's worth of a lane's A operand as loaded: k = 32 s + 4 kq .. + 3 (lo) and 32 s + 16 + 4 kq .. + 3 (hi) of its row - the
// order of the weight image's slots, i.e. k_pw_bx3's fragment order, so that every product sits at the same position of the MFMA
// in both kernels and their sums round alike
template <bool ABF> struct ARaw;
template <> struct ARaw<true> { u32v2 lo, hi; };
template <> struct ARaw<false> { float4 lo, hi; };

}  // namespace bnhip
