// Device helpers shared by the translation units that hold GEMM-shaped kernels (kernels.hip, pw_b16.hip): vector typedefs,
// activations, bf16 storage accessors, the XCD-aware block order, constant division, and the pointwise epilogue.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "kernels.h"

namespace bnhip {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// sigmoid via the hardware exp2/rcp units (v_exp_f32 / v_rcp_f32, ~1 ulp each).  TFLite's own LOGISTIC
// kernels are polynomial approximations of similar accuracy, so this stays inside fp32 noise.
// (__frcp_rn is NOT v_rcp_f32: it expands to the 10-instruction correctly-rounded division sequence.)
__device__ __forceinline__ float fast_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case ACT_RELU: return fmaxf(v, 0.0f);
        case ACT_RELU6: return fminf(fmaxf(v, 0.0f), 6.0f);
        case ACT_SWISH: return v * fast_sigmoid(v);          // LOGISTIC then MUL, as the graph does
        case ACT_SIGMOID: return fast_sigmoid(v);
        case ACT_HARD_SWISH: return v * fminf(fmaxf(v + 3.0f, 0.0f), 6.0f) / 6.0f;
        default: return v;
    }
}

// The activation code is uniform per launch.  Calling apply_act per element makes the compiler emit the whole
// switch (compare/branch tree, plus hard-swish's IEEE division sequence) once per element; with_act() branches
// once per call site and hands the body a branch-free functor (found in the ISA: ~170 s_branch per kernel before).
template <typename Body>
__device__ __forceinline__ void with_act(int act, Body&& body) {
    if (act == ACT_SWISH) body([](float v) { return v * fast_sigmoid(v); });
    else if (act == ACT_NONE) body([](float v) { return v; });
    else body([act](float v) { return apply_act(v, act); });
}

// Four-wide swish with the non-transcendental steps on packed-f32 instructions (v_pk_mul_f32 / v_pk_add_f32): the scalar
// form compiles to 5 VALU instructions per element (ISA check), this one to 4 - the two transcendentals stay scalar.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 swish2(f32x2 v) {
    const f32x2 t = v * (f32x2){-1.4426950408889634f, -1.4426950408889634f};
    f32x2 e = (f32x2){__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
    e = e + (f32x2){1.0f, 1.0f};
    const f32x2 r = (f32x2){__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
    return v * r;
}
__device__ __forceinline__ f32x4 swish4(f32x4 v) {
#ifdef BNHIP_SWISH_SCALAR
    // the same five operations per element on plain VALU instructions (same bits): a packed-f32 instruction costs ~13 cycles
    // beside MFMAs against ~4 for a plain one (MI355X_MICROARCH.md, "price of one filler") - A/B build switch
    f32x4 r;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float t = v[i] * -1.4426950408889634f;
        const float e = __builtin_amdgcn_exp2f(t) + 1.0f;
        r[i] = v[i] * __builtin_amdgcn_rcpf(e);
    }
    return r;
#else
    f32x2 lo = swish2((f32x2){v[0], v[1]}), hi = swish2((f32x2){v[2], v[3]});
    return (f32x4){lo[0], lo[1], hi[0], hi[1]};
#endif
}

// bf16 activation storage ("precision":"bf16" engines, engine.cpp mark_bf16_storage): a value whose producer and consumers all
// understand it is kept as bf16 in HBM - the 6x-expanded tensors between expand, depthwise and projection, which are what the
// HBM-bound layers move.  Round to nearest even on the way out (v_cvt_pk_bf16_f32), a 16-bit shift on the way in; arithmetic
// and accumulation stay fp32.  Four channels = one 8-byte access instead of a 16-byte one.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 bf16x4_load(const float* base, size_t quad) {       // quad: index in units of 4 elements
    const uint2 r = reinterpret_cast<const uint2*>(base)[quad];
    return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u));
}
__device__ __forceinline__ void bf16x4_store(float* base, size_t quad, const float4& v) {
    uint2 r;
    r.x = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){v.x, v.y}, bf16x2_t));
    r.y = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){v.z, v.w}, bf16x2_t));
    reinterpret_cast<uint2*>(base)[quad] = r;
}

// XCD-aware logical block id: the dispatcher places block b on XCD b % 8; remapping so that each XCD walks a
// contiguous range of logical blocks keeps halo rows / shared operand panels in ONE XCD's L2 (bijective form).
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
    unsigned q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Division of a block-uniform index by a launch constant without the float-reciprocal sequence the compiler emits for
// "uniform / uniform" (a dozen VALU instructions plus a readfirstlane each - measured as ~35 of the 540-1240 VALU
// instructions of a k_expand_dw wave): q = (n * M) >> 40 with M = floor(2^40 / d) + 1, exact for n * d < 2^40 (block and
// row indices times tile / pixel counts: < 2^38 at batch 2048), evaluated as a few scalar multiplies/adds on the 41-bit M
// split into lo (32 bits) and hi (<= 256).
struct FDiv { unsigned lo, hi, d; };
static FDiv make_fdiv(unsigned d) {
    const unsigned long long M = (1ull << 40) / d + 1;
    return FDiv{(unsigned)(M & 0xffffffffull), (unsigned)(M >> 32), d};
}
__device__ __forceinline__ unsigned fdiv(unsigned n, const FDiv& f) {
    return (unsigned)(((unsigned long long)n * f.hi + __umulhi(n, f.lo)) >> 8);      // 64-bit sum: any 32-bit n
}

#define PW_BM 128      // largest row tile (WM = 2); WM = 1 gives 64-row tiles for small grids
#ifndef PW_BK
#define PW_BK 32     // K slab; PW_LS = PW_BK + 8 keeps the (row, k-quad) slots conflict-free for 32 and 64
#endif
#define PW_LS (PW_BK + 8)
#define PW_C4 (PW_BK / 4)
// Epilogue shared by the k_pw_gemm variants.  D[i = n 4*kq + r][j = m li]: the lane holds 4 consecutive channels of one
// row.  Bias and activation are applied in registers, the 16 x (16*NT) sub-tile is staged through this wave's private LDS
// slice, and written out row-contiguously (full 64*NT-byte runs per row instead of 64-byte pieces); the residual is read
// with the same coalesced pattern.  `lds` must hold 4 x 16 x (16 NT + 4) floats and be free of operand data.
template <int NT, int WM>
__device__ __forceinline__ void pw_epilogue(const PwParams& p, f32x4 (&acc)[NT][WM], float* lds, int m0, int n0) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    constexpr int BN = NT * 16;
    constexpr int CS = BN + 4;                    // staging row stride (floats), keeps 16-byte alignment
    float* stage = lds + wave * (16 * CS);        // 4 waves x 16 x CS floats fits in one operand tile
    const bool vec_ok = (p.N & 3) == 0;
    if (p.bias) {
#pragma unroll
        for (int t = 0; t < NT; t++) {
            int n = n0 + 16 * t + 4 * kq;
            f32x4 bq = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (vec_ok && n + 3 < p.N) { float4 t4 = *reinterpret_cast<const float4*>(p.bias + n); bq = (f32x4){t4.x, t4.y, t4.z, t4.w}; }
            else {
#pragma unroll
                for (int r = 0; r < 4; r++) if (n + r < p.N) bq[r] = p.bias[n + r];
            }
#pragma unroll
            for (int mt = 0; mt < WM; mt++) acc[t][mt] += bq;
        }
    }
    if (p.act == ACT_SWISH) {
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int mt = 0; mt < WM; mt++) acc[t][mt] = swish4(acc[t][mt]);
    } else {
        with_act(p.act, [&](auto f) {
#pragma unroll
            for (int t = 0; t < NT; t++)
#pragma unroll
                for (int mt = 0; mt < WM; mt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) acc[t][mt][r] = f(acc[t][mt][r]);
        });
    }
#pragma unroll
    for (int mt = 0; mt < WM; mt++) {
#pragma unroll
        for (int t = 0; t < NT; t++)
            *reinterpret_cast<f32x4*>(&stage[li * CS + 16 * t + 4 * kq]) = acc[t][mt];
        // wave-private region: the wave's own LDS writes are visible to it once the LDS counter drains
        __builtin_amdgcn_s_waitcnt(0xc07f);       // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
        const int mbase = m0 + 16 * WM * wave + 16 * mt;
#pragma unroll
        for (int q = 0; q < (16 * (BN / 4) + 63) / 64; q++) {
            int idx = lane + 64 * q;
            int row = idx / (BN / 4), c4 = idx % (BN / 4);
            int m = mbase + row, n = n0 + 4 * c4;
            if (row < 16 && m < p.M && n < p.N) {
                f32x4 v = *reinterpret_cast<const f32x4*>(&stage[row * CS + 4 * c4]);
                float* op = p.out + (size_t)m * p.N + n;
                if (p.out_bf16 || p.res_bf16) {           // (planner: only with N % 4 == 0) bf16 residual stream / activation storage
                    if (p.res) {
                        const float4 rv = p.res_bf16 ? bf16x4_load(p.res, ((size_t)m * p.N + n) >> 2) : *reinterpret_cast<const float4*>(p.res + (size_t)m * p.N + n);
                        v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                    }
                    if (p.out_bf16) bf16x4_store(p.out, ((size_t)m * p.N + n) >> 2, make_float4(v[0], v[1], v[2], v[3]));
                    else *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
                } else if (vec_ok) {
                    if (p.res) { float4 rv = *reinterpret_cast<const float4*>(p.res + (size_t)m * p.N + n); v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w; }
                    *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (n + r < p.N) op[r] = v[r] + (p.res ? p.res[(size_t)m * p.N + n + r] : 0.f);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}


// ---- split-bf16 operand helpers (k_pw_bx3 in kernels.hip, the BX / bf16 phase 1 of the fused kernels in expdw.hip)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// exact fp32 subtraction kept scalar: under -O3 the compiler SLP-packs the two remainders of a pair into v_pk_add_f32, which
// costs ~13 cycles beside MFMAs on this chip (MI355X_MICROARCH.md, "price of one filler") against ~4 for a plain v_sub_f32
__device__ __forceinline__ float bx3_sub(float a, float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void bx3_split8(const f32x4& a, const f32x4& b, bf16x8* hi, bf16x8* mid, bf16x8* lo) {
    const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    u32x4 h, m, l;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        // v_cvt_pk_bf16_f32 (round to nearest even) per pair, remainders by exact fp32 subtraction
        const f32x2 v = {x[2 * q], x[2 * q + 1]};
        const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
        const f32x2 r = {bx3_sub(v[0], __uint_as_float(hb << 16)), bx3_sub(v[1], __uint_as_float(hb & 0xffff0000u))};
        const unsigned mb = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
        const f32x2 t = {bx3_sub(r[0], __uint_as_float(mb << 16)), bx3_sub(r[1], __uint_as_float(mb & 0xffff0000u))};
        h[q] = hb; m[q] = mb; l[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
    }
    *hi = __builtin_bit_cast(bf16x8, h); *mid = __builtin_bit_cast(bf16x8, m); *lo = __builtin_bit_cast(bf16x8, l);
}

// plain bf16 operands (PwParams::prec = 1): round to nearest even, no remainders
__device__ __forceinline__ bf16x8 bx1_cvt8(const f32x4& a, const f32x4& b) {
    u32x4 h;
    h[0] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a[0], a[1]}, bf16x2));
    h[1] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a[2], a[3]}, bf16x2));
    h[2] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){b[0], b[1]}, bf16x2));
    h[3] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){b[2], b[3]}, bf16x2));
    return __builtin_bit_cast(bf16x8, h);
}


}  // namespace bnhip
