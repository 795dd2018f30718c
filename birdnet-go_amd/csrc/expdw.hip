// gfx950 fused MBConv front half (expand 1x1 + depthwise k x k + squeeze-excite sums) and the LDS-staged plain depthwise form:
// k_expand_dw, k_expand_dw_sk and their launchers, tile-shape tables and plan-time images.  Split out of kernels.hip (the
// heaviest templates of the library: their own translation unit compiles in parallel with the rest).
#include "kernels.h"
#include "pw_common.h"

#include <algorithm>
#include <type_traits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace bnhip {

// ------------------------------------------------------------------------------------------ fused expand + depthwise
// MBConv front half in one kernel: y = act_d(dw_kxk(act_e(x * We^T + be)) + bd), plus the per-tile channel sums
// the squeeze-excite mean needs.  The 6x-expanded tensor (the largest activation of every block, ~40 % of all
// HBM traffic when materialised) lives only in LDS.
//   block = (clip, TOH x TOW output tile, 32-channel chunk of the expanded width)
//   phase 1: E[pixel][32] = expand over the tile's input footprint ((TOH-1)S+K) x ((TOW-1)S+K), as an f32-MFMA
//            GEMM whose rows are the *in-image* footprint pixels (halo outside the image is zero padding of the
//            expanded tensor and is never computed); K-slabs of 32 input channels staged like k_pw_gemm.
//   phase 2: depthwise taps from LDS, 32 thread-tiles (4 x 8) x 8 channel quads, bias + activation, store, sums.
struct ExpDwParams {
    const float* x; const float* we; const float* be; const float* wd; const float* bd;
    float* y; float* partial;
    int B, H, W, Cin, Cmid, Ho, Wo, pt, pl, act_e, act_d, tiles_h, tiles_w, cchunks, Kw, Cp;
    // STEM variant: x is the raw [B, Hin, Win, 2] image and the "expand" is the 3x3 stride-2 stem conv seen as an implicit
    // GEMM (K layout of k_stem_mfma); H, W above are then the stem's output size
    int Hin = 0, Win = 0, pts = 0, pls = 0;
    const uint16_t* wep = nullptr; int Kp = 0;   // BX variant: split-bf16 expand weights [Cp][3][Kp]
    int prec = 0;                                // BX variant: 1 = plain bf16 operands (one product), as PwParams::prec
    // pixel strides of the input / output image as the kernel's (row, column) walk them: (W, 1) / (Wo, 1) in image orientation,
    // (1, W') / (1, Wo') when rows and columns are swapped (tr = 1: H, W, Ho, Wo, pt, pl above are then the swapped values and
    // the depthwise taps are read transposed)
    int xsh = 0, xsw = 1, ysh = 0, ysw = 1, tr = 0;
    int in_bf16 = 0, out_bf16 = 0;      // bf16 activation storage: x (COPY form only) / y hold bf16 values (see bf16x4_load)
    FDiv d_bpc{}, d_cch{}, d_tw{};      // blocks per clip, channel chunks, tiles per row (set by the launcher)
    int cpp = 0;                        // chunk-loop form: channel chunks per block (0 = all: one block per (clip, tile)); small calls cut the loop into parts
    FDiv d_part{};                      // ... and the divisor by the parts per tile
};
#define ED_ES 36     // E row stride (floats)
// Phase 1 feeds the MFMA straight from global memory: every footprint pixel row belongs to exactly one wave
// (not shared across waves), so staging it through LDS would only add a write pass and two barriers per slab
// (the guide's "operand streamed once per block and not shared -> load straight to VGPRs" case).  Each lane
// loads its own fragment: pixel li of tile jt, input channels 16*t16 + 4*kq .. +3 (one float4); the K order
// inside a 16-wide slab is permuted exactly as in k_pw_gemm.  The 32 x K weight panel is tiny and L1/L2 resident.
// The planner hands over padded parameters (expand weights [Cp][Kw], Kw = Cin rounded up to 8 with zero
// columns, Cp = Cmid rounded up to 32; biases and taps padded to Cp) so that every load in the kernel is
// unconditional: pixels outside the image read a clamped (valid) address and are masked when E is written, the
// K tail multiplies finite activations by zero weights.  All small parameter loads (biases, taps) are issued at
// the top so their latency overlaps phase 1 (ISA check: they used to sit behind s_waitcnt vmcnt(0) mid-kernel).
// TRH = footprint rows held in LDS.  Only in-image rows are computed and stored (compacted), so a tile that spans the
// whole image height has no vertical halo at all; phase 2 skips the taps that fall on padding rows (the row test is
// wave-uniform: a wave owns one row group of the tile).
// Phase-2 lane -> (thread-tile column tx, channel quad c4) assignment.  ds_read_b128 is serviced in four fixed groups of 16
// lanes ({0-3,12-15,20-27}, {4-11,16-19,28-31}, and the same + 32; MI355X_MICROARCH.md, LDS), each conflict-free only if its
// lanes cover 16 distinct 16-byte slots of a 256-byte row.  With the E row stride of 36 floats the slot of a lane is
// (a * tx + c4) mod 16, a = SW * S (2 or 4 for every instantiated shape); the natural lane = tx * 8 + c4 order put three
// lanes of a group on one slot (PMC: up to 25 % of CU cycles in LDS bank conflicts).  These permutations give each group
// two tx values whose slot ranges are disjoint.
// Every group of four lanes moves as a unit, so a permutation is 16 nibbles (lane group -> lane group) in one 64-bit
// constant, decoded with a shift and a mask - a table in memory cost every block a dependent global load right before its
// first operand loads.
#define ED_PERM2 0xfdce5764b98a1320ull
#define ED_PERM4 0xfdce9ba875461320ull
__device__ __forceinline__ int ed_perm(unsigned long long magic, int lane) {
    return (int)((magic >> ((lane >> 2) * 4)) & 15ull) * 4 + (lane & 3);
}
constexpr bool ed_perm_c4_rule(unsigned long long magic) {
    for (int g = 0; g < 16; g++)
        if (((magic >> (4 * g)) & 1ull) != (unsigned long long)((g >> 1) & 1)) return false;
    return true;
}
// BX: phase 1 on the split-bf16 MFMA (see k_pw_bx3): the expand weights come pre-split ([Cp][3 planes][Kp] bf16, Kp = K rounded
// up to 32), the lane's 8 input channels of a 32-wide slab are split in registers.  The split is amortised over only two
// 16-channel tiles here (a block owns one 32-channel chunk), so it pays where the f32 MFMA dominates the wave (Cin >= 40:
// 16 MFMAs x 32 cycles per tile and slab become 12 x 16 + ~36 VALU) and not in the VALU-bound early layers: the
// create-time autotuner picks per layer.
// (the BX instantiations of the common shapes come out 2-4 registers above 128: the occupancy bound keeps them at four
// waves per SIMD like their f32 twins)
constexpr int expdw_min_waves(int K, int S, int TOW, int TRH) {
    const int tiw = (TOW - 1) * S + K, jt = (TRH * tiw + 15) / 16, jtw = (jt + 3) / 4;
    return jtw <= 4 ? 4 : (jtw == 5 ? 3 : 2);
}
// Waves per SIMD the small-K form is compiled for (the VGPR budget; measured against the register allocator's spills): what
// stays live across the chunk loop - the footprint's input operands, the prefetched chunk parameters - plus the larger of
// the two phases' working sets.  A spill here is worse than a lost wave: the reload (scratch is VMEM) waits on vmcnt behind
// the prefetch.
#ifndef EXPDW_NW8_WAVES
#define EXPDW_NW8_WAVES (est <= 76 ? 6 : 4)
#endif
constexpr int expdw_sk_waves(int K, int S, int TOH, int TOW, int TRH, int KW, bool one_chunk, int NW = 4) {
    const int tiw = (TOW - 1) * S + K, jt = (TRH * tiw + 15) / 16, jtw = (jt + NW - 1) / NW;
    if (one_chunk) return jtw <= 3 ? 5 : expdw_min_waves(K, S, TOW, TRH);
    const int sh = TOH / NW, sw = TOW / 8, rw = (sw - 1) * S + K;
    const int p1 = jtw * 8 + 8, p2 = sh * sw * 4 + rw * 4 + K * 4;
    const int est = jtw * KW / 4 + (KW / 2 + 16) + 24 + (p1 > p2 ? p1 : p2);
    // eight-wave blocks (two waves per SIMD share one footprint): waves per SIMD come in pairs
    if (NW == 8) return EXPDW_NW8_WAVES;
    return est <= 120 ? 4 : (est <= 160 ? 3 : 2);
}
// COPY: no expand at all - phase 1 only stages the tile's input footprint (32 channels of x itself) in LDS and phase 2 runs
// as above: a plain depthwise convolution whose taps read LDS instead of L1/L2 (k_dwconv_t re-reads every input value
// (TIH x TIW) / (TH x TW) = 6x for a 5 x 5 filter), with the fused kernel's tile shapes, orientations and per-tile sums.
// ---- phase 2 of the fused kernel (shared by its forms): depthwise taps from the expanded footprint in LDS, bias + activation,
// store; returns the lane's sum of what it stored (for the squeeze-excite mean).  ty is the wave index (scalar): every row
// test is wave-uniform.
// pre_store() runs after the taps and before the first global store (k_expand_dw_sk makes its prefetched loads land there).
template <int K, int S, int TOH, int TOW, int TRH, int NW = 4, typename PreStore>
__device__ __forceinline__ float4 ed_phase2(const ExpDwParams& p, const float* E, const float4* wds, int b, int oh0, int ow0, int vr0,
                                            int vr1, int ty, int tx, int c4, int n_base, const float4& bv, PreStore&& pre_store) {
    constexpr int TIW = (TOW - 1) * S + K;
    static_assert(TOH % NW == 0, "a wave owns TOH / NW output rows");
    constexpr int SH = TOH / NW, SW = TOW / 8;                // outputs per thread (thread-tiles are NW x 8)
    constexpr int RW = (SW - 1) * S + K;
    const int n = n_base + 4 * c4;
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 acc2[SH][SW];
#pragma unroll
    for (int a = 0; a < SH; a++)
#pragma unroll
        for (int c = 0; c < SW; c++) acc2[a][c] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < p.Cmid) {
        const float* e0 = E + (tx * SW * S) * ED_ES + 4 * c4;
#pragma unroll 1
        for (int i = 0; i < K; i++) {                 // kernel row (kept rolled: bounds the live weight registers)
            float4 w[K];
#pragma unroll
            for (int j = 0; j < K; j++) w[j] = wds[(i * K + j) * 8 + c4];
#pragma unroll
            for (int a = 0; a < SH; a++) {
                const int fr = (ty * SH + a) * S + i;                       // footprint row of this tap
                if (fr < vr0 || fr >= vr1 || oh0 + ty * SH + a >= p.Ho) continue;   // padding row / no such output row
                float4 xr[RW];
#pragma unroll
                for (int c = 0; c < RW; c++) xr[c] = *reinterpret_cast<const float4*>(e0 + ((fr - vr0) * TIW + c) * ED_ES);
#pragma unroll
                for (int j = 0; j < K; j++) {
#pragma unroll
                    for (int c = 0; c < SW; c++) {
                        const float4 xv = xr[c * S + j];
                        acc2[a][c].x = fmaf(xv.x, w[j].x, acc2[a][c].x); acc2[a][c].y = fmaf(xv.y, w[j].y, acc2[a][c].y);
                        acc2[a][c].z = fmaf(xv.z, w[j].z, acc2[a][c].z); acc2[a][c].w = fmaf(xv.w, w[j].w, acc2[a][c].w);
                    }
                }
            }
        }
    }
    pre_store();                                          // (on every path: outside the per-lane channel test)
    if (n < p.Cmid) {
        // store address = block-uniform 64-bit base (scalar registers) + 32-bit element offset inside the clip's image: one
        // VGPR per lane instead of a 64-bit pointer that the chunk loop of k_expand_dw_sk would have to keep (or spill)
        float* const yclip = p.y + (size_t)b * p.Ho * p.Wo * p.Cmid;
        const unsigned ylane = (unsigned)(((oh0 + ty * SH) * p.ysh + (ow0 + tx * SW) * p.ysw) * p.Cmid + n);
#pragma unroll
        for (int a = 0; a < SH; a++) {
            int oh = oh0 + ty * SH + a;
            if (oh >= p.Ho) continue;
            if (p.act_d == ACT_SWISH) {
#pragma unroll
                for (int c = 0; c < SW; c++) {
                    float4& v = acc2[a][c];
                    const f32x4 r = swish4((f32x4){v.x + bv.x, v.y + bv.y, v.z + bv.z, v.w + bv.w});
                    v = make_float4(r[0], r[1], r[2], r[3]);
                }
            } else {
                with_act(p.act_d, [&](auto f) {
#pragma unroll
                    for (int c = 0; c < SW; c++) {
                        float4& v = acc2[a][c];
                        v.x = f(v.x + bv.x); v.y = f(v.y + bv.y); v.z = f(v.z + bv.z); v.w = f(v.w + bv.w);
                    }
                });
            }
#pragma unroll
            for (int c = 0; c < SW; c++) {
                int ow = ow0 + tx * SW + c;
                if (ow >= p.Wo) continue;
                float4 v = acc2[a][c];
                const unsigned yo = ylane + (unsigned)((a * p.ysh + c * p.ysw) * p.Cmid);
                if (p.out_bf16) bf16x4_store(p.y, ((size_t)b * p.Ho * p.Wo * p.Cmid + yo) >> 2, v);   // (bf16 image: same element offsets)
                else *reinterpret_cast<float4*>(yclip + yo) = v;
                sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
            }
        }
    }
    return sum;
}
// Per-tile channel sums, first half: both lane permutations keep c4 = 4 * (physical lane bit 3) + (lane & 3), so the lanes that
// share a channel quad differ in physical lane bits 2, 4 and 5: two ds_swizzle xor steps (immediate pattern - no
// partner-address arithmetic, no inverse permutation) leave the sum of each half wave in its lanes, and the eight
// (wave, half) partials meet in LDS (red: [4 waves][2 halves][8 quads]).  Second half, after a barrier: ed_sums_out.
__device__ __forceinline__ void ed_sums_lanes(float4 sum, float4* red, int wave, int lane, int c4) {
    static_assert(ed_perm_c4_rule(ED_PERM2) && ed_perm_c4_rule(ED_PERM4), "lane permutation: c4 bit 2 must be physical lane bit 3");
    auto xsum = [&](auto pat) {
        constexpr int P = decltype(pat)::value;      // ds_swizzle bit mode: and 0x1f, or 0, xor (P >> 10)
        sum.x += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, sum.x), P));
        sum.y += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, sum.y), P));
        sum.z += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, sum.z), P));
        sum.w += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, sum.w), P));
    };
    xsum(std::integral_constant<int, 0x101f>{});     // xor 4
    xsum(std::integral_constant<int, 0x401f>{});     // xor 16
    if ((lane & 0x14) == 0) red[(wave * 2 + (lane >> 5)) * 8 + c4] = sum;
}
template <int NW = 4>
__device__ __forceinline__ void ed_sums_out(const ExpDwParams& p, const float4* red, int tid, size_t tile_index, int n_base) {
    if (tid < 8 && n_base + 4 * tid < p.Cmid) {
        float4 t = red[tid];
#pragma unroll
        for (int w = 1; w < 2 * NW; w++) { float4 v = red[w * 8 + tid]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        unsigned l4 = 4u * (unsigned)tid;
        asm volatile("" : "+v"(l4));                     // (opaque: scalar base + 32-bit lane offset, nothing 64-bit per lane to hoist)
        *reinterpret_cast<float4*>(p.partial + tile_index * p.Cmid + n_base + l4) = t;
    }
}

template <int K, int S, int TOH, int TOW, int TRH, bool H8 = false, bool BX = false, bool COPY = false>
__global__ __launch_bounds__(256, BX ? expdw_min_waves(K, S, TOW, TRH) : 1) void k_expand_dw(ExpDwParams p, unsigned nblk) {
    constexpr int TIH = (TOH - 1) * S + K, TIW = (TOW - 1) * S + K;
    static_assert(TRH <= TIH, "TRH is a cap on the footprint rows");
    constexpr int NPIX = TRH * TIW, NPIXP = (NPIX + 15) / 16 * 16;
    constexpr int JT = NPIXP / 16, JTW = (JT + 3) / 4;
    constexpr int SW = TOW / 8;                               // output columns per thread in phase 2 (thread-tiles are 4 x 8)
    __shared__ __attribute__((aligned(16))) float lds[NPIX * ED_ES + 256 + K * K * 32];
    float* E = lds;                                                      // [<=TRH rows][TIW][36] expanded footprint
    float4* red = reinterpret_cast<float4*>(lds + NPIX * ED_ES);         // [4 waves][2 half waves][8] sum scratch
    float4* wds = reinterpret_cast<float4*>(lds + NPIX * ED_ES + 256);   // [K*K][8] depthwise taps of this chunk
    // (the wave index through readfirstlane: the compiler then keeps every "which tiles / rows does this wave own" test
    // on the scalar unit instead of comparing per lane and branching on exec)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;

    const unsigned L = xcd_remap(blockIdx.x, nblk);
    const int tiles = p.tiles_h * p.tiles_w;
    const int bpc = tiles * p.cchunks;
    const int b = (int)fdiv(L, p.d_bpc), rest = (int)L - b * bpc;
    const int tile = (int)fdiv((unsigned)rest, p.d_cch), cc = rest - tile * p.cchunks;
    const int trow = (int)fdiv((unsigned)tile, p.d_tw);
    const int oh0 = trow * TOH, ow0 = (tile - trow * p.tiles_w) * TOW;
    const int ih0 = oh0 * S - p.pt, iw0 = ow0 * S - p.pl;
    // footprint rows are compacted to the in-image range [vr0, vr1) (host guarantees vr1 - vr0 <= TRH); columns keep
    // the compile-time width TIW (out-of-image columns are masked): GEMM row j <-> footprint pixel
    // (vr0 + j / TIW, j % TIW), stored at E[j]
    const int vr0 = max(ih0, 0) - ih0, vr1 = min(ih0 + TIH, p.H) - ih0;
    const int nvalid = (vr1 - vr0) * TIW;
    const int jtv = (nvalid + 15) >> 4;
    const int Cin = p.Cin, Kw = p.Kw;
    const int n_base = cc * 32;

    // ---- small parameters first (registers; the taps go to LDS after phase 1)
    static_assert(SW * S == 2 || SW * S == 4, "lane permutation tables cover SW*S in {2, 4}");
    constexpr unsigned long long PERM = SW * S == 2 ? ED_PERM2 : ED_PERM4;
    const int pl = ed_perm(PERM, lane);                  // logical lane: tx * 8 + c4
    const int c4 = pl & 7, tx = pl >> 3;
    float4 wdreg = make_float4(0.f, 0.f, 0.f, 0.f);
    // (COPY: the tap table and bias are the graph's own unpadded tensors - row stride Cp = C, loads guarded)
    if (tid < K * K * 8 && (!COPY || n_base + 4 * (tid & 7) < p.Cmid)) {
        const int tap = tid >> 3, tsrc = p.tr ? (tap % K) * K + tap / K : tap;
        wdreg = *reinterpret_cast<const float4*>(p.wd + (size_t)tsrc * p.Cp + n_base + 4 * (tid & 7));
    }
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 bq0 = COPY ? zero4 : *reinterpret_cast<const float4*>(p.be + n_base + 4 * kq);
    const float4 bq1 = COPY ? zero4 : *reinterpret_cast<const float4*>(p.be + n_base + 16 + 4 * kq);
    const float4 bv = (COPY && (!p.bd || n_base + 4 * c4 >= p.Cmid)) ? zero4 : *reinterpret_cast<const float4*>(p.bd + n_base + 4 * c4);

    if constexpr (COPY) {
        // ---- phase 1 (COPY): E[j][0..31] = x[pixel j][n_base ..], zero outside the image; 8 lanes x 16 bytes per pixel
        const float4* x4 = reinterpret_cast<const float4*>(p.x);
        const int C4 = p.Cmid >> 2, q4 = (n_base >> 2) + (tid & 7);
        const bool cin = q4 < C4;
        const size_t img = (size_t)b * p.H * p.W;
#pragma unroll 4
        for (int j = tid >> 3; j < nvalid; j += 32) {
            const int r = j / TIW, c = j - r * TIW;
            const int iw = iw0 + c, ih = ih0 + vr0 + r;
            float4 v = zero4;
            if (cin && iw >= 0 && iw < p.W) {
                const size_t quad = (img + (size_t)ih * p.xsh + (size_t)iw * p.xsw) * C4 + q4;
                v = p.in_bf16 ? bf16x4_load(p.x, quad) : x4[quad];
            }
            *reinterpret_cast<float4*>(&E[j * ED_ES + 4 * (tid & 7)]) = v;
        }
        if (tid < K * K * 8) wds[tid] = wdreg;
        __syncthreads();
    } else {

    // this lane's pixel per owned tile (a): clamped global offset + validity
    int xoff[JTW];
    bool xin[JTW];
#pragma unroll
    for (int a = 0; a < JTW; a++) {
        int j = 16 * (wave + 4 * a) + li;
        int r = j / TIW, c = j - r * TIW;
        int iw = iw0 + c;
        xin[a] = j < nvalid && iw >= 0 && iw < p.W;
        int ihc = min(ih0 + vr0 + r, p.H - 1), iwc = min(max(iw, 0), p.W - 1);
        xoff[a] = (b * p.H * p.W + ihc * p.xsh + iwc * p.xsw) * Cin + (BX ? 8 : 4) * kq;
    }
    const float* wrow0 = p.we + (size_t)(n_base + li) * Kw + 4 * kq;
    const float* wrow1 = wrow0 + (size_t)16 * Kw;

    // the accumulators start at the bias (the lane's four rows are channels 4 kq .. + 3 of its pixel): the first MFMA of a
    // tile reads it as its C operand, which removes both the zero fill and the bias add of the epilogue
    f32x4 acc[JTW][2];
#pragma unroll
    for (int a = 0; a < JTW; a++) { acc[a][0] = (f32x4){bq0.x, bq0.y, bq0.z, bq0.w}; acc[a][1] = (f32x4){bq1.x, bq1.y, bq1.z, bq1.w}; }

    auto fload = [&](int k0, f32x4& wf0, f32x4& wf1, f32x4 (&xf)[JTW]) {
        float4 t0 = *reinterpret_cast<const float4*>(wrow0 + k0), t1 = *reinterpret_cast<const float4*>(wrow1 + k0);
        wf0 = (f32x4){t0.x, t0.y, t0.z, t0.w}; wf1 = (f32x4){t1.x, t1.y, t1.z, t1.w};
        const int kx = (k0 + 4 * kq < Cin) ? k0 : -4 * kq;     // K tail: any in-bounds address (its weights are zero)
#pragma unroll
        for (int a = 0; a < JTW; a++) {
            float4 t = *reinterpret_cast<const float4*>(p.x + (size_t)xoff[a] + kx);
            xf[a] = (f32x4){t.x, t.y, t.z, t.w};
        }
    };
    auto fmma = [&](const f32x4& wf0, const f32x4& wf1, const f32x4 (&xf)[JTW]) {
#pragma unroll
        for (int a = 0; a < JTW; a++) {
            if (wave + 4 * a < jtv) {
#pragma unroll
                for (int sidx = 0; sidx < 4; sidx++) {
                    acc[a][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf0[sidx], xf[a][sidx], acc[a][0], 0, 0, 0);
                    acc[a][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf1[sidx], xf[a][sidx], acc[a][1], 0, 0, 0);
                }
            }
        }
    };
    // 8-wide half slab (Kw = 8 mod 16): lane kq holds k = k0 + 2 kq, + 1 - one float2 per operand, two MFMA steps
    auto fload8 = [&](int k0, f32x2& wh0, f32x2& wh1, f32x2 (&xh)[JTW]) {
        const float2 t0 = *reinterpret_cast<const float2*>(wrow0 - 2 * kq + k0), t1 = *reinterpret_cast<const float2*>(wrow1 - 2 * kq + k0);
        wh0 = (f32x2){t0.x, t0.y}; wh1 = (f32x2){t1.x, t1.y};
        const int kx = (k0 + 2 * kq + 1 < Cin) ? k0 - 2 * kq : -4 * kq;     // K tail: any in-bounds address (its weights are zero)
#pragma unroll
        for (int a = 0; a < JTW; a++) {
            const float2 t = *reinterpret_cast<const float2*>(p.x + (size_t)xoff[a] + kx);
            xh[a] = (f32x2){t.x, t.y};
        }
    };
    auto fmma8 = [&](const f32x2& wh0, const f32x2& wh1, const f32x2 (&xh)[JTW]) {
#pragma unroll
        for (int a = 0; a < JTW; a++) {
            if (wave + 4 * a < jtv) {
#pragma unroll
                for (int sidx = 0; sidx < 2; sidx++) {
                    acc[a][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wh0[sidx], xh[a][sidx], acc[a][0], 0, 0, 0);
                    acc[a][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wh1[sidx], xh[a][sidx], acc[a][1], 0, 0, 0);
                }
            }
        }
    };
    // H8 (compile time: the half slab costs ~10 VGPRs when it is a run-time option, which drops every shape at 120
    // VGPRs from four waves per SIMD to three) = Kw is 8 mod 16
    const int Kfull = Kw & ~15;
    if (BX) {
        const u32x4* wimg = reinterpret_cast<const u32x4*>(p.wep);          // 16-byte units: (n * 3 + plane) * Kp / 8 + k / 8
        const unsigned kp8 = (unsigned)p.Kp >> 3;
        const unsigned wr0 = (unsigned)(n_base + li) * 3u * kp8 + (unsigned)kq, wr1 = wr0 + 48u * kp8;
        for (int s32 = 0; s32 < (p.Kp >> 5); s32++) {
            // this lane's 8 channels lie inside the real K, or their weights are zero and any valid address will do
            const int kx = (32 * s32 + 8 * kq + 7 < Cin) ? 32 * s32 : -8 * kq;
            if (p.prec == 1) {                               // plain bf16 operands: hi plane only, one product
                const bf16x8 w0 = __builtin_bit_cast(bf16x8, wimg[wr0 + 4 * s32]), w1 = __builtin_bit_cast(bf16x8, wimg[wr1 + 4 * s32]);
#pragma unroll
                for (int a = 0; a < JTW; a++) {
                    if (wave + 4 * a < jtv) {
                        const float* xq = p.x + (size_t)xoff[a] + kx;
                        const float4 t0 = *reinterpret_cast<const float4*>(xq), t1 = *reinterpret_cast<const float4*>(xq + 4);
                        const bf16x8 xh = bx1_cvt8((f32x4){t0.x, t0.y, t0.z, t0.w}, (f32x4){t1.x, t1.y, t1.z, t1.w});
                        acc[a][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, xh, acc[a][0], 0, 0, 0);
                        acc[a][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, xh, acc[a][1], 0, 0, 0);
                    }
                }
                continue;
            }
            u32x4 wq[2][3];
#pragma unroll
            for (int pl3 = 0; pl3 < 3; pl3++) { wq[0][pl3] = wimg[wr0 + pl3 * kp8 + 4 * s32]; wq[1][pl3] = wimg[wr1 + pl3 * kp8 + 4 * s32]; }
#pragma unroll
            for (int a = 0; a < JTW; a++) {
                if (wave + 4 * a < jtv) {
                    const float* xq = p.x + (size_t)xoff[a] + kx;
                    const float4 t0 = *reinterpret_cast<const float4*>(xq), t1 = *reinterpret_cast<const float4*>(xq + 4);
                    bf16x8 xh, xm, xl;
                    bx3_split8((f32x4){t0.x, t0.y, t0.z, t0.w}, (f32x4){t1.x, t1.y, t1.z, t1.w}, &xh, &xm, &xl);
#pragma unroll
                    for (int t = 0; t < 2; t++) {
                        const bf16x8 wh = __builtin_bit_cast(bf16x8, wq[t][0]), wm = __builtin_bit_cast(bf16x8, wq[t][1]),
                                     wl = __builtin_bit_cast(bf16x8, wq[t][2]);
                        f32x4 c = acc[a][t];
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xm, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xh, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xm, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, c, 0, 0, 0);
                        acc[a][t] = c;
                    }
                }
            }
        }
    } else if (!H8) {
        // (a rolling two-slab register prefetch was measured here: the extra VGPRs cost more occupancy than it buys)
        for (int k0 = 0; k0 < Kw; k0 += 16) {
            f32x4 wf0, wf1, xf[JTW];
            fload(k0, wf0, wf1, xf);
            fmma(wf0, wf1, xf);
        }
    } else {
        for (int k0 = 0; k0 < Kfull; k0 += 16) {
            f32x4 wf0, wf1, xf[JTW];
            fload(k0, wf0, wf1, xf);
            fmma(wf0, wf1, xf);
        }
        f32x2 hA0, hA1, hx[JTW];
        fload8(Kfull, hA0, hA1, hx);
        fmma8(hA0, hA1, hx);
    }

    // ---- E <- act_e(acc) at compacted footprint coordinates (masked columns are zero).  Only a tile on the left / right
    // image border has columns to mask (block-uniform test): interior tiles store without the eight selects per 16 pixels
    const bool border = iw0 < 0 || iw0 + TIW > p.W;
#pragma unroll
    for (int a = 0; a < JTW; a++) {
        if (wave + 4 * a < jtv) {                         // wave-uniform: tiles beyond the valid rows cost nothing
            if (p.act_e == ACT_SWISH) {
                acc[a][0] = swish4(acc[a][0]);
                acc[a][1] = swish4(acc[a][1]);
            } else {
                with_act(p.act_e, [&](auto f) {
                    f32x4& v0 = acc[a][0];
                    f32x4& v1 = acc[a][1];
                    v0[0] = f(v0[0]); v0[1] = f(v0[1]); v0[2] = f(v0[2]); v0[3] = f(v0[3]);
                    v1[0] = f(v1[0]); v1[1] = f(v1[1]); v1[2] = f(v1[2]); v1[3] = f(v1[3]);
                });
            }
            int j = 16 * (wave + 4 * a) + li;
            if (j < nvalid) {
                int e = j * ED_ES + 4 * kq;
                if (border) {
                    const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
                    *reinterpret_cast<f32x4*>(&E[e]) = xin[a] ? acc[a][0] : z;
                    *reinterpret_cast<f32x4*>(&E[e + 16]) = xin[a] ? acc[a][1] : z;
                } else {
                    *reinterpret_cast<f32x4*>(&E[e]) = acc[a][0];
                    *reinterpret_cast<f32x4*>(&E[e + 16]) = acc[a][1];
                }
            }
        }
    }
    if (tid < K * K * 8) wds[tid] = wdreg;
    __syncthreads();
    }   // !COPY

    // ---- depthwise from LDS, per-tile channel sums
    const float4 sum = ed_phase2<K, S, TOH, TOW, TRH, 4>(p, E, wds, b, oh0, ow0, vr0, vr1, wave, tx, c4, n_base, bv, [] {});
    if (p.partial) {
        ed_sums_lanes(sum, red, wave, lane, c4);
        __syncthreads();
        ed_sums_out(p, red, tid, (size_t)b * tiles + tile, n_base);
    }
}

// Small-K form (Kw = 16, 24 or 32: the stem - an implicit GEMM over its 3 x 4 x 2 window - and the early blocks).  These layers
// are the VALU-bound ones: a wave spends more issue slots on the swish of the expanded tensor and on set-up (pixel offsets,
// validity, parameter addresses) than on its MFMAs, and with three to five waves per SIMD little of one hides behind the
// other.  So here
//   - a block owns (clip, tile) and walks ALL the 32-channel chunks of the expanded width: the pixel set-up, the footprint's
//     input operands (the whole K range: 4-8 registers per owned 16-pixel tile) and their memory latency are paid once per
//     block instead of once per chunk; per chunk only the 32 x Kw weight panel, biases and taps are fetched - requested one
//     chunk ahead, right after the barrier that publishes E, so they arrive under phase 2;
//   - tiles are walked tile-outer with the activation + LDS store of tile a - 1 issued under the MFMAs of tile a
//     (sched_group_barrier: per MFMA - 8 passes = 32 cycles of the matrix pipe - two plain and two transcendental VALU ops);
//   - two barriers per chunk, as before: "E free" sits after the first tile's MFMAs of the next chunk (they need no LDS), and
//     the per-tile channel sums of chunk c are written out by wave 0 between the two barriers of chunk c + 1.
// (LOOP = false - the stem, whose expanded width is normally a single chunk: one block per (clip, tile, chunk) as in
// k_expand_dw; without the chunk loop's live ranges it keeps the registers for five waves per SIMD)
// NW = 8 (chunk-loop form only): an eight-wave block - TWO waves per SIMD share one expanded footprint in LDS.  The early layers
// are bound by unhidden latency at the three blocks per CU their 40-50 KB footprints allow (a wave waits two thirds of its
// life); with the same LDS the CU then holds twice the waves, each owning half the pixel tiles in phase 1 and half the output
// rows in phase 2 (so the registers that stay live across the chunk loop halve too).
// PH: the pipe phase 1 runs on.  0: f32 MFMA (v_mfma_f32_16x16x4_f32).  1 ("precision":"bf16" engines, BASELINE configs[4] "bf16
// MFMA conv"): v_mfma_f32_16x16x32_bf16 with one product per operand pair - the lane's 8 input channels of each zero-padded
// 32-wide slab rounded to bf16 once per block, the weights from plane 0 of the split image (expdw_bx_image: [Cp][3][Kp] bf16,
// natural k order): 2 MFMAs of 16 cycles per 16-pixel tile, chunk and slab where the fp32 form issues 12-16 of 32 cycles
// (Perch b3-b6: -37...40 %).  (A six-product fp32-equivalent form of the same idea - operands split once per block into three
// exact bf16 pieces - was built for b2-b4 of the fp32 engines in round 4 and measured neutral: DESIGN.md section 12, git history.)
// NS (PH = 1 only): 32-wide slabs of K the block keeps resident - a bf16 fragment is 4 registers per tile and slab, so layers
// with 48 ... 256 input channels fit the chunk-loop form that the f32 operands (8 registers per 32 channels) reserve for K <= 32
// (instantiated: 2, 3, 5, 8 slabs = 64, 96, 160, 256 padded channels).
template <int K, int S, int TOH, int TOW, int TRH, bool STEM, int KW, bool LOOP = !STEM, int NW = 4, int PH = 0, int NS = 1>
__global__ __launch_bounds__(64 * NW, expdw_sk_waves(K, S, TOH, TOW, TRH, PH == 1 ? 16 * NS : KW, !LOOP, NW)) void k_expand_dw_sk(ExpDwParams p, unsigned nblk) {
    static_assert(PH == 0 || PH == 1, "phase 1 on the f32 or on the bf16 pipe");
    constexpr bool B16 = PH == 1;                             // operands are bf16 fragments of 32-wide slabs
    static_assert(!B16 || (LOOP && !STEM), "bf16 phase 1: chunk-loop form");
    static_assert(NS == 1 || B16, "several resident slabs: bf16 operands only");
    constexpr int KP = 32 * NS;                               // row length of one plane of the split image (expdw_kp)
    static_assert(PH != 1 || KW == 24 || KW == 32, "one-product form: layers of the bf16 engines");
    static_assert(NW == 4 || (NW == 8 && LOOP && TOH % 8 == 0), "eight-wave blocks: chunk-loop form, tile height a multiple of 8");
    static_assert(KW == 16 || KW == 24 || KW == 32, "one or two K slabs, or a slab and a half");
    static_assert(!STEM || KW == 24, "the stem's window is 3 rows x 4 columns x 2 channels");
    constexpr int TIH = (TOH - 1) * S + K, TIW = (TOW - 1) * S + K;
    static_assert(TRH <= TIH, "TRH is a cap on the footprint rows");
    constexpr int NPIX = TRH * TIW, NPIXP = (NPIX + 15) / 16 * 16;
    constexpr int JT = NPIXP / 16, JTW = (JT + NW - 1) / NW;
    constexpr int SW = TOW / 8;
    constexpr int NMMA = B16 ? 2 * NS : KW / 2;               // MFMAs per 16-pixel tile (two 16-channel halves)
    __shared__ __attribute__((aligned(16))) float lds[NPIX * ED_ES + 64 * NW + K * K * 32];
    float* E = lds;
    float4* red = reinterpret_cast<float4*>(lds + NPIX * ED_ES);
    float4* wds = reinterpret_cast<float4*>(lds + NPIX * ED_ES + 64 * NW);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;

    const unsigned L = xcd_remap(blockIdx.x, nblk);
    const int tiles = p.tiles_h * p.tiles_w;
    int b, tile, cc0 = 0;
    int cpp = p.cchunks;
    if constexpr (LOOP) {
        if (p.cpp > 0) {
            // small calls: a (clip, tile)'s chunk loop cut into parts of cpp chunks, one block each - the same chunks computed the same way
            // (and their per-tile sums written by whoever computed them), only on more CUs (launch_expand_dw)
            cpp = p.cpp;
            const int nparts = (p.cchunks + cpp - 1) / cpp;
            b = (int)fdiv(L, p.d_bpc);
            const int rest = (int)L - b * tiles * nparts;
            tile = (int)fdiv((unsigned)rest, p.d_part);
            cc0 = (rest - tile * nparts) * cpp;
        } else {
            b = (int)fdiv(L, p.d_bpc); tile = (int)L - b * tiles;           // (d_bpc divides by tiles here)
        }
    } else {
        b = (int)fdiv(L, p.d_bpc);
        const int rest = (int)L - b * tiles * p.cchunks;
        tile = (int)fdiv((unsigned)rest, p.d_cch); cc0 = rest - tile * p.cchunks;
    }
    const int trow = (int)fdiv((unsigned)tile, p.d_tw);
    const int oh0 = trow * TOH, ow0 = (tile - trow * p.tiles_w) * TOW;
    const int ih0 = oh0 * S - p.pt, iw0 = ow0 * S - p.pl;
    const int vr0 = max(ih0, 0) - ih0, vr1 = min(ih0 + TIH, p.H) - ih0;
    const int nvalid = (vr1 - vr0) * TIW;
    const int jtv = (nvalid + 15) >> 4;
    const int Cin = p.Cin;
    const int ncc = LOOP ? min(cpp, p.cchunks - cc0) : 1;
    const size_t tile_index = (size_t)b * tiles + tile;

    static_assert(SW * S == 2 || SW * S == 4, "lane permutation tables cover SW*S in {2, 4}");
    constexpr unsigned long long PERM = SW * S == 2 ? ED_PERM2 : ED_PERM4;
    const int pl = ed_perm(PERM, lane);                  // logical lane: tx * 8 + c4
    const int c4 = pl & 7, tx = pl >> 3;

    // ---- once per block: this lane's pixel per owned tile, and its input operands for the whole K range
    bool xin[JTW];
    f32x4 xA[B16 ? 1 : JTW];                             // k = 4 kq .. + 3 (slab 0)
    f32x4 xB[(KW == 32 && !B16) ? JTW : 1];              // k = 16 + 4 kq .. (slab 1)
    f32x2 xH[(KW == 24 && !B16) ? JTW : 1];              // k = 16 + 2 kq, + 1 (half slab)
    bf16x8 xb[B16 ? JTW : 1][NS];                        // B16: k = 32 ns + 8 kq .. + 7 of each slab, as bf16
    auto load_x = [&]() {
        const float* xbase = STEM ? p.x + (size_t)b * p.Hin * p.Win * 2 : p.x;
#pragma unroll
        for (int a = 0; a < JTW; a++) {
            const int j = 16 * (wave + NW * a) + li;
            const int r = j / TIW, c = j - r * TIW;
            const int iw = iw0 + c;
            xin[a] = j < nvalid && iw >= 0 && iw < p.W;
            const int ihc = min(ih0 + vr0 + r, p.H - 1), iwc = min(max(iw, 0), p.W - 1);
            if constexpr (STEM) {
                // the pixel (ihc, iwc) of the stem's output: its window starts at input (2 ihc - pts, 2 iwc - pls).  K layout of
                // k_stem_mfma: slab 0 lane group kq = window row kq >> 1, columns 2 (kq & 1), + 1, both channels; the half slab
                // = window row 2, column kq, both channels.  Taps outside the input image are zero (the stem's own padding).
                const int row0 = ihc * 2 - p.pts, col0 = iwc * 2 - p.pls;
                auto tap2 = [&](int row, int col) {
                    const bool v = row >= 0 && row < p.Hin && col >= 0 && col < p.Win;
                    const float2 u = *reinterpret_cast<const float2*>(xbase + ((size_t)min(max(row, 0), p.Hin - 1) * p.Win + min(max(col, 0), p.Win - 1)) * 2);
                    return (f32x2){v ? u.x : 0.f, v ? u.y : 0.f};
                };
                const f32x2 u = tap2(row0 + (kq >> 1), col0 + (kq & 1) * 2), w = tap2(row0 + (kq >> 1), col0 + (kq & 1) * 2 + 1);
                xA[a] = (f32x4){u[0], u[1], w[0], w[1]};
                xH[a] = tap2(row0 + 2, col0 + kq);
            } else if constexpr (B16) {
                const float* xp = xbase + (size_t)((b * p.H * p.W + ihc * p.xsh + iwc * p.xsw) * Cin);
                // K tail (Cin = 24: lane group 3): any in-bounds address - the image's weights are zero there
                if (p.in_bf16) {
                    // bf16 residual stream: the lane's 8 channels of a slab are 16 bytes that ARE the fragment
                    const uint16_t* xh = reinterpret_cast<const uint16_t*>(xbase) + (size_t)((b * p.H * p.W + ihc * p.xsh + iwc * p.xsw) * Cin);
#pragma unroll
                    for (int ns = 0; ns < NS; ns++)
                        xb[a][ns] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(xh + (32 * ns + 8 * kq < Cin ? 32 * ns + 8 * kq : 0)));
                } else
#pragma unroll
                for (int ns = 0; ns < NS; ns++) {
                    const float* xq = xp + (32 * ns + 8 * kq < Cin ? 32 * ns + 8 * kq : 0);
                    const float4 t0 = *reinterpret_cast<const float4*>(xq), t1 = *reinterpret_cast<const float4*>(xq + 4);
                    xb[a][ns] = bx1_cvt8((f32x4){t0.x, t0.y, t0.z, t0.w}, (f32x4){t1.x, t1.y, t1.z, t1.w});
                }
            } else {
                const float* xp = xbase + (size_t)((b * p.H * p.W + ihc * p.xsh + iwc * p.xsw) * Cin);
                // K tail: lanes whose channels lie beyond Cin read any in-bounds address (their weights are zero)
                const float4 t = *reinterpret_cast<const float4*>(xp + (4 * kq < Cin ? 4 * kq : 0));
                xA[a] = (f32x4){t.x, t.y, t.z, t.w};
                if constexpr (KW == 32) {
                    const float4 t2 = *reinterpret_cast<const float4*>(xp + (16 + 4 * kq < Cin ? 16 + 4 * kq : 0));
                    xB[a] = (f32x4){t2.x, t2.y, t2.z, t2.w};
                }
                if constexpr (KW == 24) {
                    const float2 t2 = *reinterpret_cast<const float2*>(xp + (16 + 2 * kq + 1 < Cin ? 16 + 2 * kq : 0));
                    xH[a] = (f32x2){t2.x, t2.y};
                }
            }
        }
    };
    load_x();

    // ---- per chunk: weight panel rows n_base + li and n_base + 16 + li (this lane's k range), biases, taps
    struct Chunk {
        f32x4 wA0, wA1, wB0, wB1;
        f32x2 wH0, wH1;
        bf16x8 wb0[NS], wb1[NS];                         // B16: rows n_base + li / + 16 + li of the image's plane 0, k = 32 ns + 8 kq .. + 7
        float4 bq0, bq1, bv, wd;
    };
    // (addresses as block-uniform base + 32-bit lane offset: nothing 64-bit per lane stays live across the chunk loop)
    const unsigned wlane = B16 ? (unsigned)(li * 3 * KP + 8 * kq) : (unsigned)(li * KW + 4 * kq);  // (B16: uint16 units inside the split image)
    const int tap = tid >> 3, tsrc = p.tr ? (tap % K) * K + tap / K : tap;
    const unsigned wdlane = (unsigned)(tsrc * p.Cp + 4 * (tid & 7));
    auto fetch = [&](int cc, Chunk& q) {
        const int n_base = cc * 32;
        // (the empty asm keeps the lane offsets opaque per call: left alone, the compiler hoists "pointer + lane offset" out of
        // the chunk loop as 64-bit per-lane values, runs out of registers at the four-wave budget and reloads them from
        // scratch - a vmcnt(0) wait right behind the prefetch it has just issued)
        unsigned wl = wlane, wdl = wdlane;
        asm volatile("" : "+v"(wl), "+v"(wdl));
        if constexpr (B16) {
            // image row n: 3 planes x KP bf16; plane 0, this lane's 16 bytes of each slab
            const uint16_t* w0 = p.wep + (size_t)n_base * (3 * KP) + wl;
#pragma unroll
            for (int ns = 0; ns < NS; ns++) {
                q.wb0[ns] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(w0 + 32 * ns));
                q.wb1[ns] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(w0 + 16 * 3 * KP + 32 * ns));
            }
        } else {
        const float* w0 = p.we + (size_t)n_base * KW + wl;
        const float* w1 = w0 + 16 * KW;
        const float4 t0 = *reinterpret_cast<const float4*>(w0), t1 = *reinterpret_cast<const float4*>(w1);
        q.wA0 = (f32x4){t0.x, t0.y, t0.z, t0.w}; q.wA1 = (f32x4){t1.x, t1.y, t1.z, t1.w};
        if constexpr (KW == 32) {
            const float4 u0 = *reinterpret_cast<const float4*>(w0 + 16), u1 = *reinterpret_cast<const float4*>(w1 + 16);
            q.wB0 = (f32x4){u0.x, u0.y, u0.z, u0.w}; q.wB1 = (f32x4){u1.x, u1.y, u1.z, u1.w};
        }
        if constexpr (KW == 24) {
            const float2 u0 = *reinterpret_cast<const float2*>(w0 - 2 * kq + 16), u1 = *reinterpret_cast<const float2*>(w1 - 2 * kq + 16);
            q.wH0 = (f32x2){u0.x, u0.y}; q.wH1 = (f32x2){u1.x, u1.y};
        }
        }
        q.bq0 = *reinterpret_cast<const float4*>(p.be + n_base + 4 * kq);
        q.bq1 = *reinterpret_cast<const float4*>(p.be + n_base + 16 + 4 * kq);
        q.bv = *reinterpret_cast<const float4*>(p.bd + n_base + 4 * c4);
        if (tid < K * K * 8) q.wd = *reinterpret_cast<const float4*>(p.wd + n_base + wdl);
    };
    Chunk q;
    q.wd = make_float4(0.f, 0.f, 0.f, 0.f);
    fetch(cc0, q);
    auto land = [&] {
        if constexpr (B16) {
#pragma unroll
            for (int ns = 0; ns < NS; ns++) asm volatile("" :: "v"(q.wb0[ns]), "v"(q.wb1[ns]));
            asm volatile("" :: "v"(q.bq0.x), "v"(q.bq1.x), "v"(q.bv.x), "v"(q.wd.x));
        } else {
        asm volatile("" :: "v"(q.wA0), "v"(q.wA1), "v"(q.bq0.x), "v"(q.bq1.x), "v"(q.bv.x), "v"(q.wd.x));
        if constexpr (KW == 32) asm volatile("" :: "v"(q.wB0), "v"(q.wB1));
        if constexpr (KW == 24) asm volatile("" :: "v"(q.wH0), "v"(q.wH1));
        }
    };
    if constexpr (LOOP) land();     // (also on the way in: the wait at the loop head would otherwise be shared with the back edge)

    const bool border = iw0 < 0 || iw0 + TIW > p.W;       // block-uniform: only such tiles have columns to mask in E
    const int e_lane = li * ED_ES + 4 * kq;

    for (int ci = 0; ci < ncc; ci++) {
        const int cc = cc0 + ci, n_base = cc * 32;
        // the accumulators start at the bias (the lane's four rows are channels 4 kq .. + 3 of its pixel): the first MFMA of a
        // tile reads it as its C operand - no zero fill, no bias add in the epilogue
        f32x4 acc[JTW][2];
#pragma unroll
        for (int a = 0; a < JTW; a++) {
            acc[a][0] = (f32x4){q.bq0.x, q.bq0.y, q.bq0.z, q.bq0.w};
            acc[a][1] = (f32x4){q.bq1.x, q.bq1.y, q.bq1.z, q.bq1.w};
        }
        // (hi = the chunk's upper 16 channels exist: false only for the tail chunk of a width that is 16 mod 32 - b3 / b4's
        // 144 - which then skips half of its MFMAs, activations and LDS stores; phase 2 never reads those E columns)
        auto tile_mma = [&](int a, auto hi) {
            constexpr bool HI = decltype(hi)::value;
            if constexpr (B16) {
#pragma unroll
                for (int ns = 0; ns < NS; ns++) {
                    acc[a][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q.wb0[ns], xb[a][ns], acc[a][0], 0, 0, 0);
                    if (HI) acc[a][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q.wb1[ns], xb[a][ns], acc[a][1], 0, 0, 0);
                }
                return;
            }
#pragma unroll
            for (int sidx = 0; sidx < 4; sidx++) {
                acc[a][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(q.wA0[sidx], xA[B16 ? 0 : a][sidx], acc[a][0], 0, 0, 0);
                if (HI) acc[a][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(q.wA1[sidx], xA[B16 ? 0 : a][sidx], acc[a][1], 0, 0, 0);
            }
            if constexpr (KW == 32) {
#pragma unroll
                for (int sidx = 0; sidx < 4; sidx++) {
                    acc[a][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(q.wB0[sidx], xB[B16 ? 0 : a][sidx], acc[a][0], 0, 0, 0);
                    if (HI) acc[a][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(q.wB1[sidx], xB[B16 ? 0 : a][sidx], acc[a][1], 0, 0, 0);
                }
            }
            if constexpr (KW == 24) {
#pragma unroll
                for (int sidx = 0; sidx < 2; sidx++) {
                    acc[a][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(q.wH0[sidx], xH[B16 ? 0 : a][sidx], acc[a][0], 0, 0, 0);
                    if (HI) acc[a][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(q.wH1[sidx], xH[B16 ? 0 : a][sidx], acc[a][1], 0, 0, 0);
                }
            }
        };
        // E <- act_e(acc) for tile a, at compacted footprint coordinates (masked columns are zero)
        auto tile_out = [&](int a, auto masked, auto hi, auto&& act4) {
            constexpr bool HI = decltype(hi)::value;
            acc[a][0] = act4(acc[a][0]);
            if (HI) acc[a][1] = act4(acc[a][1]);
            if (16 * (wave + NW * a) + li < nvalid) {
                float* e = E + (16 * (wave + NW * a)) * ED_ES + e_lane;
                if constexpr (decltype(masked)::value) {
                    const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
                    *reinterpret_cast<f32x4*>(e) = xin[a] ? acc[a][0] : z;
                    if (HI) *reinterpret_cast<f32x4*>(e + 16) = xin[a] ? acc[a][1] : z;
                } else {
                    *reinterpret_cast<f32x4*>(e) = acc[a][0];
                    if (HI) *reinterpret_cast<f32x4*>(e + 16) = acc[a][1];
                }
            }
        };
        auto phase1 = [&](auto masked, auto hi, auto&& act4) {
            if (wave < jtv) tile_mma(0, hi);
            if (LOOP && ci > 0) {
                __syncthreads();                           // every wave is through phase 2 of the previous chunk: E, taps and sums
                if (p.partial) ed_sums_out<NW>(p, red, tid, tile_index, n_base - 32);
            }
#pragma unroll
            for (int a = 1; a < JTW; a++) {
                if (wave + NW * a < jtv) {                // (tile a valid => tile a - 1 valid)
                    tile_mma(a, hi);
                    tile_out(a - 1, masked, hi, act4);
#pragma unroll
                    for (int m = 0; m < (decltype(hi)::value ? NMMA : NMMA / 2); m++) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
                    }
                } else if (wave + NW * (a - 1) < jtv) tile_out(a - 1, masked, hi, act4);
            }
            if (wave + NW * (JTW - 1) < jtv) tile_out(JTW - 1, masked, hi, act4);
        };
        auto phase1_act = [&](auto masked, auto hi) {
            // (the chunk-loop instantiations are launched for swish layers only - the others take k_expand_dw - so that the
            // generic activation's per-element switch is not compiled into 48 more kernels)
            if (LOOP || p.act_e == ACT_SWISH) phase1(masked, hi, [](f32x4 v) { return swish4(v); });
            else {
                const int act = p.act_e;
                phase1(masked, hi, [act](f32x4 v) {
                    return (f32x4){apply_act(v[0], act), apply_act(v[1], act), apply_act(v[2], act), apply_act(v[3], act)};
                });
            }
        };
        if (LOOP && n_base + 16 >= p.Cmid) {
            if (border) phase1_act(std::true_type{}, std::false_type{});
            else phase1_act(std::false_type{}, std::false_type{});
        } else {
            if (border) phase1_act(std::true_type{}, std::true_type{});
            else phase1_act(std::false_type{}, std::true_type{});
        }
        if (tid < K * K * 8) wds[tid] = q.wd;
        const float4 bv = q.bv;
        __syncthreads();
        if (LOOP && ci + 1 < ncc) fetch(cc + 1, q);               // next chunk's parameters: in flight during phase 2

        // The prefetched registers are "used" before phase 2's stores are issued: vmcnt retires in order, so a wait for those
        // loads placed after the stores (the top of the next chunk is where the compiler would put it) is a wait for the
        // stores' write acknowledgements as well - measured as vmcnt(0) at the loop head.  Here it only covers loads that
        // have had the whole tap loop to arrive.
        const float4 sum = ed_phase2<K, S, TOH, TOW, TRH, NW>(p, E, wds, b, oh0, ow0, vr0, vr1, wave, tx, c4, n_base, bv, land);
        if (p.partial) ed_sums_lanes(sum, red, wave, lane, c4);
    }
    if (p.partial) {
        __syncthreads();
        ed_sums_out<NW>(p, red, tid, tile_index, (cc0 + ncc - 1) * 32);
    }
}

// K of the expand GEMM as the kernel walks it: 16-wide slabs plus, when Cin <= 8 mod 16, one 8-wide half slab (two MFMA
// steps instead of four: Cin = 24 / 40 would otherwise spend 25 % / 17 % of their MFMAs on zero columns)
int expdw_kw(int Cin) { return (Cin & 1) ? (Cin + 15) / 16 * 16 : (Cin + 7) / 8 * 8; }
// K width of the small-K chunk-loop form when the layer takes it (swish expand, one or two K slabs, not the stem), else 0
int expdw_skw(int Cin, int act_e, bool stem) {
    const int kw = expdw_kw(Cin);
    return !stem && act_e == ACT_SWISH && (kw == 16 || kw == 24 || kw == 32) ? kw : 0;
}
int expdw_cp(int Cmid) { return (Cmid + 31) / 32 * 32; }
// Instantiated tile shapes.  The chooser takes, per layer, the shape that computes the fewest expanded pixels
// (rows x TIW summed over the tiles of one image; halo recompute and masked padding columns both count) among those
// whose in-image footprint rows fit TRH.
struct ExpDwShape { int k, s, toh, tow, trh, nw = 4; };
static const ExpDwShape kExpDwShapes[] = {
    {3, 1, 8, 16, 10}, {3, 1, 4, 16, 6}, {3, 1, 8, 32, 6}, {3, 1, 8, 32, 10},
    {5, 1, 8, 16, 12}, {5, 1, 4, 16, 8}, {5, 1, 8, 32, 6}, {5, 1, 12, 16, 12},
    {3, 2, 4, 8, 9}, {3, 2, 8, 8, 12}, {3, 2, 8, 8, 17},
    {5, 2, 4, 8, 11}, {5, 2, 4, 16, 6}, {5, 2, 8, 8, 19},   // the last one computes fewer pixels on b4 but measured 27 % slower
                                                             // there (52 KB of LDS, 6 MFMA tiles per wave): hence the autotuner
    // eight-wave blocks of the small-K chunk-loop form (k_expand_dw_sk<..., NW = 8>): the 8-row tiles again, two waves per SIMD
    // sharing one footprint; only offered to layers that take that form (ExpDwGeo::skw)
    {3, 1, 8, 16, 10, 8}, {3, 1, 8, 32, 6, 8}, {3, 1, 8, 32, 10, 8}, {5, 1, 8, 16, 12, 8}, {5, 1, 8, 32, 6, 8},
    {3, 2, 8, 8, 12, 8}, {3, 2, 8, 8, 17, 8}, {5, 2, 8, 8, 19, 8},
};
static long expdw_cost(const ExpDwShape& sh, int H, int Ho, int Wo, int pt, bool* fits) {
    const int tih = (sh.toh - 1) * sh.s + sh.k, tiw = (sh.tow - 1) * sh.s + sh.k;
    const int th = (Ho + sh.toh - 1) / sh.toh, tw = (Wo + sh.tow - 1) / sh.tow;
    long rows = 0;
    *fits = true;
    for (int t = 0; t < th; t++) {
        int ih0 = t * sh.toh * sh.s - pt;
        int v = std::min(ih0 + tih, H) - std::max(ih0, 0);
        if (v > sh.trh) *fits = false;
        rows += std::max(v, 0);
    }
    return rows * tiw * tw;
}
static const int kNumExpDwShapes = (int)(sizeof(kExpDwShapes) / sizeof(kExpDwShapes[0]));
int expdw_num_shapes() { return 2 * kNumExpDwShapes; }
// the geometry as the kernel walks it for shape index idx (rows and columns swapped for idx >= n)
static ExpDwGeo expdw_oriented(int idx, const ExpDwGeo& g) {
    if (idx < kNumExpDwShapes) return g;
    ExpDwGeo t = g;
    t.H = g.W; t.W = g.H; t.Ho = g.Wo; t.Wo = g.Ho; t.pt = g.pl; t.pl = g.pt;
    return t;
}
bool expdw_shape_fits(int idx, const ExpDwGeo& g0, bool planning) {
    if (idx < 0 || idx >= 2 * kNumExpDwShapes) return false;
    if (idx >= kNumExpDwShapes && g0.stem) return false;
    if (!g0.stem && planning) {                          // BNHIP_EXPDW_ORIENT = n | t: one orientation only (tests, A/B runs; plan / tune time only)
        const char* oe = getenv("BNHIP_EXPDW_ORIENT");
        if (oe && ((oe[0] == 'n' && idx >= kNumExpDwShapes) || (oe[0] == 't' && idx < kNumExpDwShapes))) return false;
    }
    const ExpDwShape& sh = kExpDwShapes[idx % kNumExpDwShapes];
    if (sh.k != g0.k || sh.s != g0.s) return false;
    if (sh.nw == 8) {                                     // eight-wave blocks exist for the chunk-loop small-K form only
        static const bool off = getenv("BNHIP_NO_EXPDW_NW8") != nullptr;
        if (off || g0.stem || !(g0.skw == 16 || g0.skw == 24 || g0.skw == 32)) return false;
    }
    const ExpDwGeo g = expdw_oriented(idx, g0);
    bool fits;
    (void)expdw_cost(sh, g.H, g.Ho, g.Wo, g.pt, &fits);
    return fits;
}
int expdw_shape_slabs(int idx, const ExpDwGeo& g0) {
    const ExpDwShape& sh = kExpDwShapes[idx % kNumExpDwShapes];
    const ExpDwGeo g = expdw_oriented(idx, g0);
    return ((g.Ho + sh.toh - 1) / sh.toh) * ((g.Wo + sh.tow - 1) / sh.tow);
}
// cost-model choice (fewest expanded pixels); the engine's create-time autotuner may override it per layer
int expdw_default_shape(const ExpDwGeo& g0) {
    int best = -1;
    long best_cost = 0;
    for (int i = 0; i < 2 * kNumExpDwShapes; i++) {
        if (!expdw_shape_fits(i, g0)) continue;
        const ExpDwGeo g = expdw_oriented(i, g0);
        const ExpDwShape& sh = kExpDwShapes[i % kNumExpDwShapes];
        bool fits;
        long c = expdw_cost(sh, g.H, g.Ho, g.Wo, g.pt, &fits);
        const long lds = (long)sh.trh * ((sh.tow - 1) * sh.s + sh.k) * ED_ES * 4;
        if (lds > 51 * 1024) c += c / 3;                // fewer than three blocks per CU: measured to outweigh a smaller halo
        if (best < 0 || c < best_cost) { best = i; best_cost = c; }
    }
    return best;
}
int expdw_sum_slabs(const ExpDwGeo& g) {
    int idx = expdw_default_shape(g);
    return idx < 0 ? 0 : expdw_shape_slabs(idx, g);
}
int expdw_max_slabs(const ExpDwGeo& g) {
    int mx = 0;
    for (int i = 0; i < 2 * kNumExpDwShapes; i++)
        if (expdw_shape_fits(i, g)) mx = std::max(mx, expdw_shape_slabs(i, g));
    return mx;
}
// split-bf16 expand weights: [Cp][3][Kp] bf16 in natural k order (a lane reads 8 consecutive k = one 16-byte unit per plane)
bool expdw_bx_ok(int Cin) { return (Cin & 7) == 0 && Cin >= 16; }
int expdw_kp(int Cin) { return (Cin + 31) / 32 * 32; }
std::vector<uint16_t> expdw_bx_image(const float* We /*[Cmid][Cin]*/, int Cmid, int Cin) {
    const int Cp = expdw_cp(Cmid), Kp = expdw_kp(Cin);
    std::vector<uint16_t> img((size_t)Cp * 3 * Kp, 0);
    auto rne = [](float f) -> uint16_t {
        unsigned u; memcpy(&u, &f, 4);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
        return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    };
    auto widen = [](uint16_t h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; };
    for (int n = 0; n < Cmid; n++)
        for (int k = 0; k < Cin; k++) {
            const float x = We[(size_t)n * Cin + k];
            const uint16_t h = rne(x); const float r = x - widen(h);
            const uint16_t m = rne(r); const float q = r - widen(m);
            img[((size_t)n * 3 + 0) * Kp + k] = h; img[((size_t)n * 3 + 1) * Kp + k] = m; img[((size_t)n * 3 + 2) * Kp + k] = rne(q);
        }
    return img;
}
// phase 1 of the chunk-loop form on the bf16 matrix pipe (k_expand_dw_sk<PH = 1>): "precision":"bf16" engines with the split
// image on hand, swish expand, K of 24 ... 160 channels (1, 2, 3 or 5 slabs of 32).  BNHIP_EXPDW_B16=0: never.
bool expdw_sk_pipe16(int Cin, int act_e, bool stem, int prec, bool have_image) {
    static const bool off = getenv("BNHIP_EXPDW_B16") && atoi(getenv("BNHIP_EXPDW_B16")) == 0;
    static const int max_ns = getenv("BNHIP_EXPDW_B16_NS") ? atoi(getenv("BNHIP_EXPDW_B16_NS")) : 5;     // A/B switch: resident slabs allowed
    if (off || !have_image || stem || act_e != ACT_SWISH || !expdw_bx_ok(Cin)) return false;
    const int kw = expdw_skw(Cin, act_e, stem), ns = expdw_kp(Cin) / 32;
    // one product: resident slabs.  (Eight slabs - 232 -> 1392 channels on 16 x 4 images, one block per clip - were measured
    // slower than the unfused pair, 266 vs 65 + 90 us per block, and are not instantiated.)
    return prec == 1 && ((kw == 24 || kw == 32) || (kw == 0 && (ns == 2 || ns == 3 || ns == 5) && ns <= max_ns));
}
bool expdw_supported(int k, int s, int Cin, int Cmid, int act_e, int prec) {
    // measured on MI355X at batch 256: beyond ~128 input channels the unpipelined K loop of the fused kernel loses
    // to the separate pw_gemm + dwconv pair (b13-b16 of the B0 stack: 126 us vs 176 us), so those stay unfused ...
    if (!((k == 3 || k == 5) && (s == 1 || s == 2) && (Cin & 3) == 0 && (Cmid & 3) == 0)) return false;
    if (Cin <= 128) return true;
    // ... except in "precision":"bf16" engines when the layer fits the chunk-loop form with resident bf16 operands
    // (k_expand_dw_sk<PH = 1, NS = 5 | 8>: 160 / 256 padded input channels, 2 NS MFMAs of 16 cycles per tile and chunk)
    return prec == 1 && expdw_sk_pipe16(Cin, act_e, false, prec, true);
}
void launch_expand_dw(const float* x, const float* we, const float* be, const float* wd, const float* bd, float* y,
                      float* partial, int B, int H, int W, int Cin, int Cmid, int Ho, int Wo, int k, int s, int pt,
                      int pl, int act_e, int act_d, int shape, const StemGeom* stem, hipStream_t st, const uint16_t* wep, int prec, int out_bf16,
                      int in_bf16) {
    // (a layer whose phase 1 runs on the bf16 pipe never takes an eight-wave shape - those exist in the f32 form only - so that
    // the arithmetic of a layer does not depend on which tile the tuner preferred: skw = 0 withholds them)
    const bool pipe16 = expdw_sk_pipe16(Cin, act_e, stem != nullptr, prec, wep != nullptr);
    const ExpDwGeo g0{k, s, H, W, Ho, Wo, pt, pl, stem != nullptr, pipe16 ? 0 : expdw_skw(Cin, act_e, stem != nullptr)};
    if (!expdw_shape_fits(shape, g0, false)) shape = expdw_default_shape(g0);
    if (shape < 0) return;                             // the planner only fuses layers some shape accepts
    const ExpDwShape* sh = &kExpDwShapes[shape % kNumExpDwShapes];
    const bool tr = shape >= kNumExpDwShapes;
    const ExpDwGeo g = expdw_oriented(shape, g0);
    ExpDwParams p{x, we, be, wd, bd, y, partial, B, g.H, g.W, Cin, Cmid, g.Ho, g.Wo, g.pt, g.pl, act_e, act_d,
                  (g.Ho + sh->toh - 1) / sh->toh, (g.Wo + sh->tow - 1) / sh->tow, (Cmid + 31) / 32, expdw_kw(Cin), expdw_cp(Cmid)};
    // the kernel's rows are image columns when tr: one kernel row step = one pixel, one kernel column step = an image row
    p.xsh = tr ? 1 : W; p.xsw = tr ? W : 1; p.ysh = tr ? 1 : Wo; p.ysw = tr ? Wo : 1; p.tr = tr ? 1 : 0;
    p.out_bf16 = out_bf16;
    unsigned nblk = (unsigned)B * p.tiles_h * p.tiles_w * p.cchunks;
    p.d_bpc = make_fdiv((unsigned)(p.tiles_h * p.tiles_w * p.cchunks));
    p.d_cch = make_fdiv((unsigned)p.cchunks);
    p.d_tw = make_fdiv((unsigned)p.tiles_w);
    if (stem) p.Kw = 24;                               // 3 rows x 4 columns x 2 channels
    // the small-K form (f32 MFMA) also serves the bf16x3 = 2 / "precision":"bf16" engines: with one or two K slabs the MFMAs
    // are a small part of the wave either way, and the chunk loop is worth more than the cheaper products (fp32 products where
    // bf16 ones were asked for are never less accurate)
    const bool sk = stem || pipe16 || (act_e == ACT_SWISH && (p.Kw == 16 || p.Kw == 24 || p.Kw == 32));
    const bool bx = wep != nullptr && !sk && expdw_bx_ok(Cin);
    if (bx) { p.wep = wep; p.Kp = expdw_kp(Cin); p.prec = prec; }
    // ... except where phase 1 runs on the bf16 pipe (expdw_sk_pipe16: "precision":"bf16" engines, one product per pair - 2 MFMAs
    // instead of 12-16 per tile, chunk and slab)
    const bool b16 = pipe16;
    if (pipe16) { p.wep = wep; p.Kp = expdw_kp(Cin); p.prec = prec; p.in_bf16 = in_bf16; }     // (the planner marks x bf16 for this form only)
    if (sk && !stem) {
        // small-K form: a block owns (clip, tile) and walks the channel chunks itself
        const int tiles = p.tiles_h * p.tiles_w;
        nblk = (unsigned)B * tiles;
        p.d_bpc = make_fdiv((unsigned)tiles);
        // ... unless the call is so small that the chip would idle behind a few serial chunk loops (one clip: 12-24 blocks walking
        // five chunks each, 19-24 us): then the loop is cut into parts until the grid has about one block per CU
        const int cus = device_cus();
        if (p.cchunks > 1 && (long)B * tiles < cus / 2) {
            const int want = (int)std::min<long>(p.cchunks, (cus + (long)B * tiles - 1) / ((long)B * tiles));
            const int cpp = (p.cchunks + want - 1) / want, nparts = (p.cchunks + cpp - 1) / cpp;
            if (nparts > 1) {
                p.cpp = cpp;
                p.d_part = make_fdiv((unsigned)nparts);
                nblk = (unsigned)B * tiles * nparts;
                p.d_bpc = make_fdiv((unsigned)(tiles * nparts));
            }
        }
    }
    if (stem) {
        p.Hin = stem->Hin; p.Win = stem->Win; p.pts = stem->pt; p.pls = stem->pl;
#define ED_STEM(TH_, TW_, TR_)                                                                                \
    if (sh->k == 3 && sh->s == 1 && sh->toh == TH_ && sh->tow == TW_ && sh->trh == TR_) {                     \
        hipLaunchKernelGGL((k_expand_dw_sk<3, 1, TH_, TW_, TR_, true, 24>), dim3(nblk), dim3(256), 0, st, p, nblk);  \
        return;                                                                                               \
    }
        ED_STEM(8, 16, 10) ED_STEM(4, 16, 6) ED_STEM(8, 32, 6) ED_STEM(8, 32, 10)
#undef ED_STEM
        return;
    }
#define ED_CASE8(K_, S_, TH_, TW_, TR_)                                                                       \
    if (sh->nw == 8 && sh->k == K_ && sh->s == S_ && sh->toh == TH_ && sh->tow == TW_ && sh->trh == TR_) {    \
        if (p.Kw == 16) hipLaunchKernelGGL((k_expand_dw_sk<K_, S_, TH_, TW_, TR_, false, 16, true, 8>), dim3(nblk), dim3(512), 0, st, p, nblk); \
        else if (p.Kw == 24) hipLaunchKernelGGL((k_expand_dw_sk<K_, S_, TH_, TW_, TR_, false, 24, true, 8>), dim3(nblk), dim3(512), 0, st, p, nblk); \
        else hipLaunchKernelGGL((k_expand_dw_sk<K_, S_, TH_, TW_, TR_, false, 32, true, 8>), dim3(nblk), dim3(512), 0, st, p, nblk); \
        return;                                                                                               \
    }
    if (sh->nw == 8) {
        if (!sk || stem) return;                       // (expdw_shape_fits never offers these to other layers)
        ED_CASE8(3, 1, 8, 16, 10) ED_CASE8(3, 1, 8, 32, 6) ED_CASE8(3, 1, 8, 32, 10) ED_CASE8(5, 1, 8, 16, 12) ED_CASE8(5, 1, 8, 32, 6)
        ED_CASE8(3, 2, 8, 8, 12) ED_CASE8(3, 2, 8, 8, 17) ED_CASE8(5, 2, 8, 8, 19)
        return;
    }
#undef ED_CASE8
#define ED_CASE(K_, S_, TH_, TW_, TR_)                                                                        \
    if (sh->k == K_ && sh->s == S_ && sh->toh == TH_ && sh->tow == TW_ && sh->trh == TR_) {                   \
        if (bx) hipLaunchKernelGGL((k_expand_dw<K_, S_, TH_, TW_, TR_, false, true>), dim3(nblk), dim3(256), 0, st, p, nblk); \
        else if (b16 && p.Kp == 64) hipLaunchKernelGGL((k_expand_dw_sk<K_, S_, TH_, TW_, TR_, false, 32, true, 4, 1, 2>), dim3(nblk), dim3(256), 0, st, p, nblk); \
        else if (b16 && p.Kp == 96) hipLaunchKernelGGL((k_expand_dw_sk<K_, S_, TH_, TW_, TR_, false, 32, true, 4, 1, 3>), dim3(nblk), dim3(256), 0, st, p, nblk); \
        else if (b16 && p.Kp == 160) hipLaunchKernelGGL((k_expand_dw_sk<K_, S_, TH_, TW_, TR_, false, 32, true, 4, 1, 5>), dim3(nblk), dim3(256), 0, st, p, nblk); \
        else if (b16 && p.Kw == 24) hipLaunchKernelGGL((k_expand_dw_sk<K_, S_, TH_, TW_, TR_, false, 24, true, 4, 1>), dim3(nblk), dim3(256), 0, st, p, nblk); \
        else if (b16) hipLaunchKernelGGL((k_expand_dw_sk<K_, S_, TH_, TW_, TR_, false, 32, true, 4, 1>), dim3(nblk), dim3(256), 0, st, p, nblk); \
        else if (sk && p.Kw == 16) hipLaunchKernelGGL((k_expand_dw_sk<K_, S_, TH_, TW_, TR_, false, 16>), dim3(nblk), dim3(256), 0, st, p, nblk); \
        else if (sk && p.Kw == 24) hipLaunchKernelGGL((k_expand_dw_sk<K_, S_, TH_, TW_, TR_, false, 24>), dim3(nblk), dim3(256), 0, st, p, nblk); \
        else if (sk && p.Kw == 32) hipLaunchKernelGGL((k_expand_dw_sk<K_, S_, TH_, TW_, TR_, false, 32>), dim3(nblk), dim3(256), 0, st, p, nblk); \
        else if (p.Kw & 8) hipLaunchKernelGGL((k_expand_dw<K_, S_, TH_, TW_, TR_, true>), dim3(nblk), dim3(256), 0, st, p, nblk); \
        else hipLaunchKernelGGL((k_expand_dw<K_, S_, TH_, TW_, TR_>), dim3(nblk), dim3(256), 0, st, p, nblk);  \
        return;                                                                                               \
    }
    ED_CASE(3, 1, 8, 16, 10) ED_CASE(3, 1, 4, 16, 6) ED_CASE(3, 1, 8, 32, 6) ED_CASE(3, 1, 8, 32, 10)
    ED_CASE(5, 1, 8, 16, 12) ED_CASE(5, 1, 4, 16, 8) ED_CASE(5, 1, 8, 32, 6) ED_CASE(5, 1, 12, 16, 12)
    ED_CASE(3, 2, 4, 8, 9) ED_CASE(3, 2, 8, 8, 12) ED_CASE(3, 2, 8, 8, 17)
    ED_CASE(5, 2, 4, 8, 11) ED_CASE(5, 2, 4, 16, 6) ED_CASE(5, 2, 8, 8, 19)
#undef ED_CASE
}

// plain depthwise conv staged through LDS: the fused kernel in COPY mode (its phase 2 alone), same shape indices as above
bool dwconv_lds_supported(const DwParams& p) {
    return (p.C & 3) == 0 && p.kh == p.kw && p.sh == p.sw && (p.kh == 3 || p.kh == 5) && (p.sh == 1 || p.sh == 2);
}
void launch_dwconv_lds(const DwParams& q, float* partial, int shape, hipStream_t st) {
    const ExpDwGeo g0{q.kh, q.sh, q.H, q.W, q.Ho, q.Wo, q.pt, q.pl, false};
    if (!expdw_shape_fits(shape, g0, false)) shape = expdw_default_shape(g0);
    if (shape < 0) return;
    const ExpDwShape* sh = &kExpDwShapes[shape % kNumExpDwShapes];
    const bool tr = shape >= kNumExpDwShapes;
    const ExpDwGeo g = expdw_oriented(shape, g0);
    ExpDwParams p{q.in, nullptr, nullptr, q.w, q.bias, q.out, partial, q.B, g.H, g.W, q.C, q.C, g.Ho, g.Wo, g.pt, g.pl, ACT_NONE, q.act,
                  (g.Ho + sh->toh - 1) / sh->toh, (g.Wo + sh->tow - 1) / sh->tow, (q.C + 31) / 32, 0, q.C};
    p.xsh = tr ? 1 : q.W; p.xsw = tr ? q.W : 1; p.ysh = tr ? 1 : q.Wo; p.ysw = tr ? q.Wo : 1; p.tr = tr ? 1 : 0;
    p.in_bf16 = q.in_bf16; p.out_bf16 = q.out_bf16;
    unsigned nblk = (unsigned)q.B * p.tiles_h * p.tiles_w * p.cchunks;
    p.d_bpc = make_fdiv((unsigned)(p.tiles_h * p.tiles_w * p.cchunks));
    p.d_cch = make_fdiv((unsigned)p.cchunks);
    p.d_tw = make_fdiv((unsigned)p.tiles_w);
#define DL_CASE(K_, S_, TH_, TW_, TR_)                                                                        \
    if (sh->k == K_ && sh->s == S_ && sh->toh == TH_ && sh->tow == TW_ && sh->trh == TR_) {                   \
        hipLaunchKernelGGL((k_expand_dw<K_, S_, TH_, TW_, TR_, false, false, true>), dim3(nblk), dim3(256), 0, st, p, nblk); \
        return;                                                                                               \
    }
    DL_CASE(3, 1, 8, 16, 10) DL_CASE(3, 1, 4, 16, 6) DL_CASE(3, 1, 8, 32, 6) DL_CASE(3, 1, 8, 32, 10)
    DL_CASE(5, 1, 8, 16, 12) DL_CASE(5, 1, 4, 16, 8) DL_CASE(5, 1, 8, 32, 6) DL_CASE(5, 1, 12, 16, 12)
    DL_CASE(3, 2, 4, 8, 9) DL_CASE(3, 2, 8, 8, 12) DL_CASE(3, 2, 8, 8, 17)
    DL_CASE(5, 2, 4, 8, 11) DL_CASE(5, 2, 4, 16, 6) DL_CASE(5, 2, 8, 8, 19)
#undef DL_CASE
}


}  // namespace bnhip
