// gfx950 pointwise / dense GEMM of the "precision":"bf16" engines (BASELINE configs[4], Perch-style deployments):
//   out[M,N] = act(bf16(A[M,K] (* scale[b,K])) . bf16(W[N,K])^T + bias[N]) (+ res[M,N]),   fp32 accumulate.
// The arithmetic is exactly that of k_pw_bx3 with PwParams::prec = 1 (one v_mfma_f32_16x16x32_bf16 product per operand pair,
// operands rounded to nearest even, slabs of 32 along K in order, the same K order inside a slab), so the two kernels agree
// bit for bit (tests/test_perch_like.py); what differs is how the operands travel.  k_pw_bx3 is built for the six-product
// fp32-equivalent form, where 6 MFMAs amortise each operand fragment: it stages A as fp32 through LDS and converts at the
// fragment read.  With ONE product per fragment that traffic is the kernel (PMC, Perch bf16: 8-47 VALU per MFMA, 3-22 % of
// the bf16 pipe busy, LDS bank conflicts on up to 90 % of LDS instructions).  Here:
//   * A never touches LDS.  Wave w owns rows [32 w, 32 w + 32) of the block's 128-row tile and no other wave reads them, so a
//     lane loads its own fragments straight from global memory (bf16 storage: two 8-byte loads per 16 x 32 fragment; fp32
//     activations: two 16-byte loads), two slabs ahead; the squeeze-excite scale is multiplied in and the product rounded to
//     bf16 in registers (4 v_cvt_pk_bf16_f32 per fragment), or - bf16 storage without scale - the loaded bits ARE the fragment.
//   * W comes from the plan-time image k_pw_bx3 uses (hi plane = RNE bf16, already in fragment order [slab][plane][kq][row][8]):
//     a slab's tile is a straight 16-byte-per-thread copy into LDS, double-buffered, read back conflict-free (lane = slot).
//   * one barrier per slab (the weight buffer swap); bias / activation / residual / bf16 or fp32 store = pw_epilogue.
// Block = 128 rows x 16 NT columns, 4 waves, accumulators [NT][2] x f32x4.
// Also in this file: the six-product (fp32-equivalent) form of the same kernel for the fp32 engines, 64- or 128-row tiles
// (template parameters SIX, WM), and k_pw_b16s - skinny projections with the whole weight matrix in registers.
#include "pw_split.h"

#include <algorithm>
#include <atomic>

namespace bnhip {

// B16_TRACE (tools/ubench/b16_trace.hip only): lane 0 of every wave of the first 64 logical blocks stamps the shader clock at the
// phase boundaries, to see where a wave's time goes.  Compiled out of the library.
#ifdef B16_TRACE
__device__ long long* g_b16_trace = nullptr;      // [64 blocks][4 waves][B16_TRACE_SLOTS]
#define B16_TRACE_SLOTS 160
#define B16_T(i) do { if (lane == 0 && L < 64u && (i) < B16_TRACE_SLOTS) g_b16_trace[((size_t)L * 4 + wave) * B16_TRACE_SLOTS + (i)] = clock64(); } while (0)
#else
#define B16_T(i) do { } while (0)
#endif

template <bool ABF, bool SCR> struct ASet { ARaw<ABF> a[2]; };
template <bool ABF> struct ASet<ABF, true> { ARaw<ABF> a[2]; float4 slo[2], shi[2]; };

// SCL: the squeeze-excite scale of a slab goes through LDS once per block ([clips of the row tile][32] floats) instead of being
// loaded per row - possible when a 16-row MFMA tile never straddles two clips (HW % 16 == 0); otherwise each lane loads its own.
// Loads run two slabs ahead for A (two register sets, the loop is unrolled by two) and one iteration ahead for the weight tile
// (loaded in iteration s - 1, written to the idle LDS buffer at the top of iteration s, read in s + 1): with one slab of
// prefetch every slab paid a memory latency (16 MFMAs = 256 cycles of cover against ~1.5 us).
// SIX: the fp32-equivalent form (PwParams::prec == 0, fp32 engines): A split into three bf16 pieces in registers, all three
// weight planes of the slab in LDS, six products per operand pair in k_pw_bx3's order.  WM: 16-row tiles per wave (rows per
// block = 64 WM).
template <int NT, bool SC, bool ABF, bool SCL = false, int WM = 2, bool SIX = false>
__global__ __launch_bounds__(256) void k_pw_b16(PwParams p, const uint16_t* __restrict__ Wimg, int Npad, int nblk_n, unsigned nblk,
                                                 FDiv dn, FDiv dhw) {
    static_assert(!SCL || SC, "SCL is a form of SC");
    static_assert(!SIX || !ABF, "fp32 engines keep fp32 activations");
    constexpr bool SCR = SC && !SCL;                           // scale in registers, per lane
    constexpr int BM = 64 * WM, BN = 16 * NT;
    constexpr int NP = SIX ? 3 : 1;                            // weight planes used (hi | hi, mid, lo)
    constexpr int WSLOTS = NP * 4 * BN;                        // 16-byte slots of a slab's weight tile: [plane NP][kq 4][row BN]
    constexpr int WQ = (WSLOTS + 255) / 256;
    constexpr int SCLIPS = BM / 16 + 1;                        // clips a 128-row tile can touch when HW >= 16
    constexpr int SBUF = SCL ? SCLIPS * 32 : 0;                // floats of one scale buffer
    constexpr int OBUF = WSLOTS * 4 + SBUF;                    // floats of one operand buffer
    constexpr int STG = 4 * 16 * (BN + 4);                     // pw_epilogue's staging area (floats)
    constexpr int LDSN = 2 * OBUF > STG ? 2 * OBUF : STG;
    __shared__ __attribute__((aligned(16))) float lds[LDSN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const unsigned L = xcd_remap(blockIdx.x, nblk);
    const int mblk = (int)fdiv(L, dn);
    const int m0 = mblk * BM;
    const int n0 = ((int)L - mblk * nblk_n) * BN;
    const int K = p.K;
    const int nslab = (K + 31) >> 5;

    // element offsets of the lane's two rows (rows beyond M are clamped: their results are never stored)
    unsigned aoff[WM], soff[SC ? WM : 1];
    const int b_first = SCL ? (int)fdiv((unsigned)m0, dhw) : 0;                 // first clip of the row tile
#pragma unroll
    for (int mt = 0; mt < WM; mt++) {
        const int m = min(m0 + 16 * WM * wave + 16 * mt + li, p.M - 1);
        aoff[mt] = (unsigned)m * (unsigned)K + 4u * (unsigned)kq;
        if (SC) {
            const unsigned b = fdiv((unsigned)m, dhw);
            soff[SC ? mt : 0] = SCL ? (unsigned)(((int)b - b_first) * 32 + 4 * kq)   // float offset inside the LDS scale buffer
                                    : b * (unsigned)K + 4u * (unsigned)kq;
        }
    }
    unsigned woff[WQ];
#pragma unroll
    for (int q = 0; q < WQ; q++) {
        const int slot = min(tid + 256 * q, WSLOTS - 1);
        const int kqs = slot / BN, r = slot - kqs * BN;                                   // kqs = plane * 4 + kq
        woff[q] = (unsigned)kqs * (unsigned)Npad + (unsigned)min(n0 + r, Npad - 1);       // 16-byte units inside a slab of the image
    }
    const u32v4* W16 = reinterpret_cast<const u32v4*>(Wimg);
    const uint16_t* A16 = reinterpret_cast<const uint16_t*>(p.A);
    // scale tile loader: thread t -> clip t / 8, k quad t % 8
    const int sclip = tid >> 3, sk4 = tid & 7;
    const int nclips_blk = SCL ? (int)fdiv((unsigned)(min(m0 + BM, p.M) - 1), dhw) - b_first + 1 : 0;

    ASet<ABF, SCR> set0, set1;
    u32v4 wreg[WQ];
    float4 sreg;
    auto aload = [&](int sl, auto& st) {
        const int k0 = sl * 32;
        const bool inlo = k0 + 4 * kq < K, inhi = k0 + 16 + 4 * kq < K;      // K tail: columns beyond K are zeros (their weights too)
#pragma unroll
        for (int mt = 0; mt < WM; mt++) {
            if constexpr (ABF) {
                const uint16_t* ap = A16 + aoff[mt] + k0;
                st.a[mt].lo = inlo ? *reinterpret_cast<const u32v2*>(ap) : (u32v2){0u, 0u};
                st.a[mt].hi = inhi ? *reinterpret_cast<const u32v2*>(ap + 16) : (u32v2){0u, 0u};
            } else {
                const float* ap = p.A + aoff[mt] + k0;
                st.a[mt].lo = inlo ? *reinterpret_cast<const float4*>(ap) : make_float4(0.f, 0.f, 0.f, 0.f);
                st.a[mt].hi = inhi ? *reinterpret_cast<const float4*>(ap + 16) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if constexpr (SCR) {
                const float* sp = p.ascale + soff[mt] + k0;
                st.slo[mt] = inlo ? *reinterpret_cast<const float4*>(sp) : make_float4(0.f, 0.f, 0.f, 0.f);
                st.shi[mt] = inhi ? *reinterpret_cast<const float4*>(sp + 16) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto wload = [&](int sl) {
        const int k0 = sl * 32;
        if constexpr (SCL) {
            sreg = (sclip < nclips_blk && k0 + 4 * sk4 < K)
                       ? *reinterpret_cast<const float4*>(p.ascale + (size_t)(b_first + sclip) * K + k0 + 4 * sk4)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const u32v4* Ws = W16 + (size_t)sl * 12 * Npad;                       // 3 planes x 4 kq x Npad slots per slab; plane 0 = hi
#pragma unroll
        for (int q = 0; q < WQ; q++)
            if (tid + 256 * q < WSLOTS) wreg[q] = Ws[woff[q]];
    };
    auto wstore = [&](int buf) {
        float* base = lds + buf * OBUF;
        u32v4* Wl = reinterpret_cast<u32v4*>(base);
#pragma unroll
        for (int q = 0; q < WQ; q++)
            if (tid + 256 * q < WSLOTS) Wl[tid + 256 * q] = wreg[q];
        if constexpr (SCL) {
            if (sclip < SCLIPS) *reinterpret_cast<float4*>(base + WSLOTS * 4 + sclip * 32 + 4 * sk4) = sreg;
        }
    };
    // the lane's A values of one 16-row tile for this slab as fp32 (scale applied), or directly as a bf16 fragment
    auto avals = [&](int mt, const auto& st, const float* sbuf, float4& v0, float4& v1) {
        if constexpr (ABF) { v0 = b16_unpack4(st.a[mt].lo); v1 = b16_unpack4(st.a[mt].hi); }
        else { v0 = st.a[mt].lo; v1 = st.a[mt].hi; }
        if constexpr (SC) {
            float4 s0, s1;
            if constexpr (SCL) {
                s0 = *reinterpret_cast<const float4*>(sbuf + soff[mt]);
                s1 = *reinterpret_cast<const float4*>(sbuf + soff[mt] + 16);
            } else { s0 = st.slo[mt]; s1 = st.shi[mt]; }
            v0.x *= s0.x; v0.y *= s0.y; v0.z *= s0.z; v0.w *= s0.w;
            v1.x *= s1.x; v1.y *= s1.y; v1.z *= s1.z; v1.w *= s1.w;
        }
    };

    f32x4 acc[NT][WM];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int mt = 0; mt < WM; mt++) acc[t][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // one slab: the weight tile of the NEXT slab (in wreg since the previous iteration) goes to the idle buffer and wreg takes the
    // slab after it; this slab's A fragments come out of `st`, which then takes slab sl + 2
    auto slab = [&](int sl, auto& st) {
        B16_T(8 + 4 * sl);                                 // slab begins
        if (sl + 1 < nslab) wstore((sl + 1) & 1);          // (every wave finished reading that buffer before the last barrier)
        B16_T(9 + 4 * sl);                                 // next weight tile in LDS (its global loads had landed)
        if (sl + 2 < nslab) wload(sl + 2);
        const float* obuf = lds + (sl & 1) * OBUF;
        b16x8 ah[WM], am[SIX ? WM : 1], al[SIX ? WM : 1];
#pragma unroll
        for (int mt = 0; mt < WM; mt++) {
            if constexpr (ABF && !SC) {
                ah[mt] = __builtin_bit_cast(b16x8, (u32v4){st.a[mt].lo[0], st.a[mt].lo[1], st.a[mt].hi[0], st.a[mt].hi[1]});
            } else {
                float4 v0, v1;
                avals(mt, st, obuf + WSLOTS * 4, v0, v1);
                if constexpr (SIX) b16_split8(v0, v1, &ah[mt], &am[SIX ? mt : 0], &al[SIX ? mt : 0]);
                else ah[mt] = b16_cvt8(v0, v1);
            }
        }
        if (sl + 2 < nslab) aload(sl + 2, st);
        const u32v4* Wl = reinterpret_cast<const u32v4*>(obuf);
        if constexpr (SIX) {
            // weight fragments of tile t + 1 are requested before the MFMAs of tile t
            u32v4 wfr[2][3];
#pragma unroll
            for (int pl3 = 0; pl3 < 3; pl3++) wfr[0][pl3] = Wl[(pl3 * 4 + kq) * BN + li];
#pragma unroll
            for (int t = 0; t < NT; t++) {
                if (t + 1 < NT) {
#pragma unroll
                    for (int pl3 = 0; pl3 < 3; pl3++) wfr[(t + 1) & 1][pl3] = Wl[(pl3 * 4 + kq) * BN + 16 * (t + 1) + li];
                }
                const b16x8 wh = __builtin_bit_cast(b16x8, wfr[t & 1][0]);
                const b16x8 wm = __builtin_bit_cast(b16x8, wfr[t & 1][1]);
                const b16x8 wl = __builtin_bit_cast(b16x8, wfr[t & 1][2]);
#pragma unroll
                for (int mt = 0; mt < WM; mt++) {
                    f32x4 c = acc[t][mt];                        // smallest terms first (k_pw_bx3's order)
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, ah[mt], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, al[SIX ? mt : 0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, am[SIX ? mt : 0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, ah[mt], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, am[SIX ? mt : 0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, ah[mt], c, 0, 0, 0);
                    acc[t][mt] = c;
                }
            }
        } else {
            b16x8 wf[NT];
#pragma unroll
            for (int t = 0; t < NT; t++) wf[t] = __builtin_bit_cast(b16x8, Wl[kq * BN + 16 * t + li]);
#pragma unroll
            for (int t = 0; t < NT; t++)
#pragma unroll
                for (int mt = 0; mt < WM; mt++) acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t], ah[mt], acc[t][mt], 0, 0, 0);
        }
        B16_T(10 + 4 * sl);                                // MFMAs issued
        __syncthreads();
        B16_T(11 + 4 * sl);                                // barrier passed
    };
    B16_T(0);
    aload(0, set0);
    if (nslab > 1) aload(1, set1);
    wload(0);
    wstore(0);
    if (nslab > 1) wload(1);
    B16_T(1);                                              // first weight tile stored (its loads had landed)
    __syncthreads();
    B16_T(2);
    for (int sl = 0; sl < nslab; sl += 2) {
        slab(sl, set0);
        if (sl + 1 < nslab) slab(sl + 1, set1);
    }
    B16_T(3);
    pw_epilogue<NT, WM>(p, acc, lds, m0, n0);
    B16_T(4);
}

static std::atomic<long> g_pw_b16_launches{0};      // diagnostics (tests assert that this path, not k_pw_bx3's, ran); the workers of a multi-device handle launch concurrently

// ---- skinny layers: weights stationary in registers
// The early projections of an MBConv stack have K and N of a few dozen (24 -> 24, 40 -> 24, 144 -> 32, 192 -> 32) against millions of
// rows: a 128-row tile there is one or a handful of slabs wrapped in a prologue, an LDS epilogue and barriers (Perch b2/project:
// 458 us for 1.2 GB, 2.6 TB/s).  Here the whole weight matrix lives in a wave's registers as fragments (NS slabs x NT 16-column tiles
// x 4 registers, at most 48), a wave walks 16-row tiles of its own row range with nothing but loads, NS x NT MFMAs and stores per
// tile - no LDS, no barriers - the next tile's A fragments requested before this tile's MFMAs, and the epilogue goes straight from
// the accumulators (a lane holds four consecutive channels of one row: one 8- or 16-byte access).  Same image, same K order, one
// product per pair: bit-identical to k_pw_b16 / k_pw_bx3.  Squeeze-excite scale: per-clip fragments reloaded when the tile's clip
// changes (HW % 16 == 0: a tile never straddles two clips).
template <int NT, int NS, bool SC, bool ABF>
__global__ __launch_bounds__(256) void k_pw_b16s(PwParams p, const uint16_t* __restrict__ Wimg, int Npad, int tiles_per_wave, FDiv dhw) {
    const int lane = threadIdx.x & 63, li = lane & 15, kq = lane >> 4;
    const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int K = p.K, N = p.N;
    const int ntiles = (p.M + 15) >> 4;
    const int t0 = wave_g * tiles_per_wave, t1 = min(t0 + tiles_per_wave, ntiles);
    if (t0 >= t1) return;
    // weights: plane 0 of the image, slot (slab, kq, row 16 t + li)
    const u32v4* W16 = reinterpret_cast<const u32v4*>(Wimg);
    b16x8 wf[NS][NT];
#pragma unroll
    for (int ns = 0; ns < NS; ns++)
#pragma unroll
        for (int t = 0; t < NT; t++) wf[ns][t] = __builtin_bit_cast(b16x8, W16[((size_t)ns * 12 + kq) * Npad + min(16 * t + li, Npad - 1)]);
    f32x4 bias[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) {
        bias[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int n = 16 * t + 4 * kq;
        if (p.bias && n + 3 < N) { const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n); bias[t] = (f32x4){b4.x, b4.y, b4.z, b4.w}; }
    }
    const uint16_t* A16 = reinterpret_cast<const uint16_t*>(p.A);
    bool inlo[NS], inhi[NS];
#pragma unroll
    for (int ns = 0; ns < NS; ns++) { inlo[ns] = 32 * ns + 4 * kq < K; inhi[ns] = 32 * ns + 16 + 4 * kq < K; }
    ARaw<ABF> cur[NS], nxt[NS];
    auto aload = [&](int tile, ARaw<ABF> (&dst)[NS]) {
        const int m = min(16 * tile + li, p.M - 1);
        const size_t off = (size_t)m * K + 4 * kq;
#pragma unroll
        for (int ns = 0; ns < NS; ns++) {
            if constexpr (ABF) {
                dst[ns].lo = inlo[ns] ? *reinterpret_cast<const u32v2*>(A16 + off + 32 * ns) : (u32v2){0u, 0u};
                dst[ns].hi = inhi[ns] ? *reinterpret_cast<const u32v2*>(A16 + off + 32 * ns + 16) : (u32v2){0u, 0u};
            } else {
                dst[ns].lo = inlo[ns] ? *reinterpret_cast<const float4*>(p.A + off + 32 * ns) : make_float4(0.f, 0.f, 0.f, 0.f);
                dst[ns].hi = inhi[ns] ? *reinterpret_cast<const float4*>(p.A + off + 32 * ns + 16) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    float4 slo[SC ? NS : 1], shi[SC ? NS : 1];
    int sclip = -1;
    aload(t0, cur);
    for (int tile = t0; tile < t1; tile++) {
        if (tile + 1 < t1) aload(tile + 1, nxt);
        if constexpr (SC) {
            const int b = (int)fdiv((unsigned)min(16 * tile, p.M - 1), dhw);           // wave-uniform
            if (b != sclip) {
                sclip = b;
                const float* sp = p.ascale + (size_t)b * K + 4 * kq;
#pragma unroll
                for (int ns = 0; ns < NS; ns++) {
                    slo[ns] = inlo[ns] ? *reinterpret_cast<const float4*>(sp + 32 * ns) : make_float4(0.f, 0.f, 0.f, 0.f);
                    shi[ns] = inhi[ns] ? *reinterpret_cast<const float4*>(sp + 32 * ns + 16) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ns = 0; ns < NS; ns++) {
            b16x8 af;
            if constexpr (ABF && !SC) {
                af = __builtin_bit_cast(b16x8, (u32v4){cur[ns].lo[0], cur[ns].lo[1], cur[ns].hi[0], cur[ns].hi[1]});
            } else {
                float4 v0, v1;
                if constexpr (ABF) { v0 = b16_unpack4(cur[ns].lo); v1 = b16_unpack4(cur[ns].hi); }
                else { v0 = cur[ns].lo; v1 = cur[ns].hi; }
                if constexpr (SC) {
                    const float4 s0 = slo[ns], s1 = shi[ns];
                    v0.x *= s0.x; v0.y *= s0.y; v0.z *= s0.z; v0.w *= s0.w;
                    v1.x *= s1.x; v1.y *= s1.y; v1.z *= s1.z; v1.w *= s1.w;
                }
                af = b16_cvt8(v0, v1);
            }
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ns][t], af, acc[t], 0, 0, 0);
        }
        // epilogue from the accumulators: row m = 16 tile + li, channels 16 t + 4 kq .. + 3
        const int m = 16 * tile + li;
        if (m < p.M) {
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const int n = 16 * t + 4 * kq;
                if (n + 3 < N) {                                   // (N % 4 == 0: a quad is wholly inside or outside)
                    f32x4 v = acc[t] + bias[t];
                    if (p.act == ACT_SWISH) v = swish4(v);
                    else if (p.act != ACT_NONE) { v[0] = apply_act(v[0], p.act); v[1] = apply_act(v[1], p.act); v[2] = apply_act(v[2], p.act); v[3] = apply_act(v[3], p.act); }
                    const size_t o = (size_t)m * N + n;
                    if (p.res) {
                        const float4 rv = p.res_bf16 ? bf16x4_load(p.res, o >> 2) : *reinterpret_cast<const float4*>(p.res + o);
                        v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                    }
                    if (p.out_bf16) bf16x4_store(p.out, o >> 2, make_float4(v[0], v[1], v[2], v[3]));
                    else *reinterpret_cast<float4*>(p.out + o) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
        if (tile + 1 < t1) {
#pragma unroll
            for (int ns = 0; ns < NS; ns++) cur[ns] = nxt[ns];
        }
    }
}

// layers the weights-stationary kernel takes: one-product engines, N a multiple of 4 up to 32, K up to 192 (NS x NT <= 12 fragments),
// squeeze-excite scale only when a 16-row tile cannot straddle clips
bool pw_b16s_ok(const PwParams& p) {
    if (p.sw & PW_SW_B16S_OFF) return false;
    if (p.prec != 1 || (p.K & 3) || p.K < 16 || (p.N & 3) || p.N > 32) return false;
    const int ns = (p.K + 31) / 32, nt = (p.N + 15) / 16;
    if (ns * nt > 12 || ns > 6) return false;
    if (p.ascale && !(p.HW >= 16 && (p.HW & 15) == 0)) return false;
    return true;
}
void launch_pw_b16s(const PwParams& p, const uint16_t* Wimg, int Npad, hipStream_t s) {
    g_pw_b16_launches.fetch_add(1, std::memory_order_relaxed);
    const int ns = (p.K + 31) / 32, nt = (p.N + 15) / 16;
    const int ntiles = (p.M + 15) / 16;
    // enough waves to fill the chip several times over, few enough that a wave amortises its weight fragments over many tiles
    int tpw = std::max(1, std::min(64, ntiles / (256 * 4 * 8)));
    const int waves = (ntiles + tpw - 1) / tpw;
    const unsigned nblk = (unsigned)((waves + 3) / 4);
    const FDiv dhw = make_fdiv((unsigned)std::max(p.HW, 1));
    const bool sc = p.ascale != nullptr, abf = p.a_bf16 != 0;
#define B16S_LAUNCH(NT_, NS_, SC_, ABF_) hipLaunchKernelGGL((k_pw_b16s<NT_, NS_, SC_, ABF_>), dim3(nblk), dim3(256), 0, s, p, Wimg, Npad, tpw, dhw)
#define B16S_FLAV(NT_, NS_) do { if (sc) { if (abf) B16S_LAUNCH(NT_, NS_, true, true); else B16S_LAUNCH(NT_, NS_, true, false); } \
                                 else { if (abf) B16S_LAUNCH(NT_, NS_, false, true); else B16S_LAUNCH(NT_, NS_, false, false); } } while (0)
#define B16S_NS(NT_) switch (ns) { case 1: B16S_FLAV(NT_, 1); break; case 2: B16S_FLAV(NT_, 2); break; case 3: B16S_FLAV(NT_, 3); break; \
                                   case 4: B16S_FLAV(NT_, 4); break; case 5: B16S_FLAV(NT_, 5); break; default: B16S_FLAV(NT_, 6); break; }
    if (nt == 1) B16S_NS(1) else B16S_NS(2)
#undef B16S_LAUNCH
#undef B16S_FLAV
#undef B16S_NS
}

// the switches (BNHIP_PW_B16 / BNHIP_PW_B16S: 0 = candidate taken away, 2 = forced) are read once per engine and travel in PwParams::sw
bool pw_b16_ok(int prec, int K, int sw) { return (prec == 0 || prec == 1) && (K & 3) == 0 && K >= 16 && !(sw & PW_SW_B16_OFF); }

// prec 1 (one product): 128-row tiles; prec 0 (six products): 64- or 128-row tiles (wm = 1 | 2)
void launch_pw_b16(const PwParams& p, const uint16_t* Wimg, int nt, int wm, int Npad, int nblk_n, unsigned nblk, hipStream_t s) {
    g_pw_b16_launches.fetch_add(1, std::memory_order_relaxed);
    const FDiv dn = make_fdiv((unsigned)nblk_n), dhw = make_fdiv((unsigned)std::max(p.HW, 1));
    const bool sc = p.ascale != nullptr, abf = p.a_bf16 != 0, six = p.prec == 0;
    const bool scl = sc && p.HW >= 16 && (p.HW & 15) == 0;      // a 16-row tile never straddles two clips: scale through LDS
    dim3 grid(nblk);
#define B16_LAUNCH(NT_, SC_, ABF_, SCL_, WM_, SIX_) hipLaunchKernelGGL((k_pw_b16<NT_, SC_, ABF_, SCL_, WM_, SIX_>), grid, dim3(256), 0, s, p, Wimg, Npad, nblk_n, nblk, dn, dhw)
#define B16_FLAV(NT_, WM_, SIX_, ABF_) do { if (scl) B16_LAUNCH(NT_, true, ABF_, true, WM_, SIX_); else if (sc) B16_LAUNCH(NT_, true, ABF_, false, WM_, SIX_); \
                                            else B16_LAUNCH(NT_, false, ABF_, false, WM_, SIX_); } while (0)
#define B16_CASE(NT_) case NT_: \
        if (six) { if (wm == 1) B16_FLAV(NT_, 1, true, false); else B16_FLAV(NT_, 2, true, false); } \
        else if (abf) B16_FLAV(NT_, 2, false, true); \
        else B16_FLAV(NT_, 2, false, false); \
        break;
    switch (nt) { B16_CASE(1) B16_CASE(2) B16_CASE(3) B16_CASE(4) B16_CASE(5) B16_CASE(6) B16_CASE(7) default: B16_CASE(8) }
#undef B16_LAUNCH
#undef B16_FLAV
#undef B16_CASE
}

}  // namespace bnhip

extern "C" long bnhip_debug_pw_b16_launches(void) { return bnhip::g_pw_b16_launches.load(std::memory_order_relaxed); }
