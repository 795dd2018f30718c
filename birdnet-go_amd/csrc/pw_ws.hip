// gfx950 late-layer GEMMs (round 5): out[M,N] = act((A[M,K] (* scale[b,K])) . W[N,K]^T + bias[N]) (+ res[M,N]) on
// v_mfma_f32_16x16x32_bf16 with k_pw_bx3's arithmetic - the plan-time weight image (three exact bf16 planes in fragment order),
// the same K order inside a slab, the same six products per accumulator in the same order (or one product, "precision":"bf16") -
// so both kernels here agree with k_pw_bx3 / k_pw_b16 bit for bit; what differs is how operands travel and who waits for whom.
//
// What the wave-lifetime trace of the tiled kernels showed (profiles/r04_b16_trace.txt): at batch 256 the late layers have
// M = 12 288 rows - a 128 x 64 tile is 288 MFMA-cycles of work per K slab and wave against ~2 000 cycles of load latency and
// barrier, co-resident blocks run identical phases, a third of a wave's life issues MFMAs.
//
//  * k_pw_ws - short K, wide N, no squeeze-excite scale (the 6x expands 192 -> 1152, the 320 -> 1024 layer in front of the
//    pooling).  A block's weight columns (NS slabs x 3 planes x 16 NT columns) are loaded into LDS ONCE; after that barrier its
//    waves are independent: each owns a contiguous range of 16-row tiles of the block's row range and walks it two tiles at a
//    time (one tile at an odd tail: the split is even to one 16-row tile per wave), A streamed straight from global memory
//    into a ring of R register slab sets that runs R slabs ahead ACROSS tiles, the exact three-way split of slab s + 1 spread
//    between the MFMAs of slab s, the two accumulator chains of a tile pair interleaved (a dependent MFMA never follows its
//    producer), bias from LDS, activation and the store straight from the accumulators (a lane holds 4 consecutive channels
//    of one row: 64-byte segments per row and instruction).  No tile writes, no per-slab barrier, no prologue per tile.
//
// (A second form - k_pw_deep: 512-thread blocks, every wave 32 rows x NT column tiles of a 256-row x whole-N or 128-row x two-half
// block, the slab's weight tile double-buffered in LDS, A through a four-slab register ring, the block's squeeze-excite rows
// resident in LDS - was built for the long-K projections in the same round, bit-identical, and measured 10-25 % SLOWER than the
// best tiled candidate on every projection and on the dense head: profiles/r05_pw_lab_all_candidates.txt; removed)
#include "pw_split.h"

#include <algorithm>
#include <atomic>

namespace bnhip {

// exact split of a pair of fp32 values into three packed bf16 pairs (k_pw_bx3's decomposition: hi = RNE(x), mid = RNE(x - hi),
// lo = RNE(x - hi - mid); the subtractions are exact)
__device__ __forceinline__ void ws_split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    const f32x2 v = {x0, x1};
    const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(v, b16x2));
    const f32x2 r = {b16_sub(v[0], __uint_as_float(hb << 16)), b16_sub(v[1], __uint_as_float(hb & 0xffff0000u))};
    const unsigned mb = __builtin_bit_cast(unsigned, __builtin_convertvector(r, b16x2));
    const f32x2 t = {b16_sub(r[0], __uint_as_float(mb << 16)), b16_sub(r[1], __uint_as_float(mb & 0xffff0000u))};
    h = hb; m = mb; l = __builtin_bit_cast(unsigned, __builtin_convertvector(t, b16x2));
}
// pair q (0..3) of a lane's eight values of a slab: (lo.x lo.y) (lo.z lo.w) (hi.x hi.y) (hi.z hi.w)
__device__ __forceinline__ void ws_pair(const float4& lo, const float4& hi, int q, float& x0, float& x1) {
    x0 = q == 0 ? lo.x : q == 1 ? lo.z : q == 2 ? hi.x : hi.z;
    x1 = q == 0 ? lo.y : q == 1 ? lo.w : q == 2 ? hi.y : hi.w;
}

struct WsFrag { u32v4 h, m, l; };      // a 16 x 32 operand fragment as three bf16 planes (one-product kernels use h only)

// the six products of a (column tile, slab) on the WM accumulators of a tile pair, k_pw_bx3's order per accumulator
// (smallest terms first), the chains interleaved so that an MFMA never reads the result of the one issued just before it
template <int WM, bool SIX>
__device__ __forceinline__ void ws_mfma(f32x4 (&c)[WM], const u32v4 (&w)[SIX ? 3 : 1], const WsFrag (&f)[2]) {
    const b16x8 wh = __builtin_bit_cast(b16x8, w[0]);
    if constexpr (SIX) {
        const b16x8 wm = __builtin_bit_cast(b16x8, w[1]), wl = __builtin_bit_cast(b16x8, w[2]);
#pragma unroll
        for (int mt = 0; mt < WM; mt++) c[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, __builtin_bit_cast(b16x8, f[mt].h), c[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < WM; mt++) c[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, __builtin_bit_cast(b16x8, f[mt].l), c[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < WM; mt++) c[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, __builtin_bit_cast(b16x8, f[mt].m), c[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < WM; mt++) c[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, __builtin_bit_cast(b16x8, f[mt].h), c[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < WM; mt++) c[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, __builtin_bit_cast(b16x8, f[mt].m), c[mt], 0, 0, 0);
    }
#pragma unroll
    for (int mt = 0; mt < WM; mt++) c[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, __builtin_bit_cast(b16x8, f[mt].h), c[mt], 0, 0, 0);
}

// bias + activation + residual + store of one accumulator quad: row m, channels n .. n + 3 (bias: nullable), pw_epilogue's
// arithmetic in pw_epilogue's order.  N % 4 == 0: a quad is wholly in or out and 16-byte aligned; otherwise element by element
// (fp32 storage only).
__device__ __forceinline__ void ws_store(const PwParams& p, f32x4 v, const float* bias, int m, int n) {
    const bool vec = (p.N & 3) == 0;
    if (n >= p.N) return;
    if (bias) {
        if (vec) { const float4 b4 = *reinterpret_cast<const float4*>(bias + n); v += (f32x4){b4.x, b4.y, b4.z, b4.w}; }
        else {
#pragma unroll
            for (int r = 0; r < 4; r++) if (n + r < p.N) v[r] += bias[n + r];
        }
    }
    if (p.act == ACT_SWISH) v = swish4(v);
    else if (p.act != ACT_NONE) { v[0] = apply_act(v[0], p.act); v[1] = apply_act(v[1], p.act); v[2] = apply_act(v[2], p.act); v[3] = apply_act(v[3], p.act); }
    const size_t o = (size_t)m * p.N + n;
    if (vec) {
        if (p.res) {
            const float4 rv = p.res_bf16 ? bf16x4_load(p.res, o >> 2) : *reinterpret_cast<const float4*>(p.res + o);
            v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
        }
        if (p.out_bf16) bf16x4_store(p.out, o >> 2, make_float4(v[0], v[1], v[2], v[3]));
        else *reinterpret_cast<float4*>(p.out + o) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int r = 0; r < 4; r++)
            if (n + r < p.N) p.out[o + r] = v[r] + (p.res ? p.res[o + r] : 0.f);
    }
}

static std::atomic<long> g_pw_ws_launches{0};      // diagnostics: tests assert that this path ran

// ================================================================================================ weight columns stationary in LDS
template <int NT, int NS, int R, bool SIX, bool ABF, int NW>
__global__ __launch_bounds__(64 * NW) void k_pw_ws(PwParams p, const uint16_t* __restrict__ Wimg, int Npad, int nblk_n, unsigned nblk,
                                                    unsigned GW /*waves per column block = groups x NW*/, FDiv dn) {
    static_assert(NS % R == 0 && R < NS, "the register ring must divide the slab count (slab s of every tile lives in set s % R)");
    static_assert(!SIX || !ABF, "fp32 engines keep fp32 activations");
    constexpr int BN = 16 * NT, NP = SIX ? 3 : 1;
    constexpr int WSLOTS = NS * NP * 4 * BN;                 // 16-byte slots: [slab][plane][kq][column]
    extern __shared__ __attribute__((aligned(16))) unsigned char ws_lds[];
    u32v4* Wl = reinterpret_cast<u32v4*>(ws_lds);
    float* Bl = reinterpret_cast<float*>(ws_lds + (size_t)WSLOTS * 16);       // the block's bias columns
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);         // (uniform: the tile bookkeeping below stays on the scalar unit)
    const unsigned L = xcd_remap(blockIdx.x, nblk);          // consecutive L: the column blocks of one row range, on one XCD (they share A)
    const int g = (int)fdiv(L, dn);
    const int n0 = ((int)L - g * nblk_n) * BN;
    const int K = p.K, N = p.N;
    // WS_TRACE (tools/ubench/ws_trace.hip only; compiled out of the library): lane 0 of every wave stamps the shader clock at the
    // phase boundaries into the buffer that travels in p.res
#ifdef WS_TRACE
    long long* trc = reinterpret_cast<long long*>(const_cast<float*>(p.res)) + ((size_t)blockIdx.x * NW + wave) * 32;
    int tslot = 0;
    p.res = nullptr;
    if (lane == 0) trc[30] = wall_clock64();
#define WS_T() do { if (lane == 0 && tslot < 29) trc[tslot] = clock64(); tslot++; } while (0)
#else
#define WS_T() do { } while (0)
#endif
    WS_T();
    const u32v4* W16 = reinterpret_cast<const u32v4*>(Wimg);
    {   // the block's weight columns: every load of a thread requested before the first store (a load - store loop paid a memory
        // latency per trip: 11 k of a wave's 66 k cycles in the first timing of this kernel)
        constexpr int WQ = (WSLOTS + 64 * NW - 1) / (64 * NW);
        u32v4 wv[WQ];
#pragma unroll
        for (int i = 0; i < WQ; i++) {
            const int slot = min(tid + 64 * NW * i, WSLOTS - 1);
            const int r = slot % BN, q = slot / BN;          // q = (slab * NP + plane) * 4 + kq
            const int kqs = q & 3, pl = (q >> 2) % NP, sl = (q >> 2) / NP;
            wv[i] = W16[((size_t)sl * 12 + pl * 4 + kqs) * Npad + min(n0 + r, Npad - 1)];
        }
#pragma unroll
        for (int i = 0; i < WQ; i++)
            if (tid + 64 * NW * i < WSLOTS) Wl[tid + 64 * NW * i] = wv[i];
    }
    for (int c = tid; c < BN; c += 64 * NW) Bl[c] = (p.bias && n0 + c < N) ? p.bias[n0 + c] : 0.f;
    __syncthreads();
    WS_T();

    // this wave's 16-row tiles: an even split of the row range over the GW waves that share the column block
    const unsigned mt16 = (unsigned)(p.M + 15) >> 4;
    const unsigned gw = (unsigned)g * NW + (unsigned)wave;
    const int t0 = (int)(gw * mt16 / GW), t1 = (int)((gw + 1) * mt16 / GW);
    if (t0 >= t1) return;

    const uint16_t* A16 = reinterpret_cast<const uint16_t*>(p.A);
    auto rows = [&](int tile, unsigned (&off)[2]) {          // element offsets of the lane's rows of a tile pair (clamped: never stored)
#pragma unroll
        for (int mt = 0; mt < 2; mt++) off[mt] = (unsigned)min(16 * (tile + mt) + li, p.M - 1) * (unsigned)K + 4u * (unsigned)kq;
    };
    // (K tail, K % 32 != 0: only the last slab can be short - its missing columns are read from the slab's first quad instead and
    // zeroed by a select, branch-free: a predicated load would cut the MFMA stream into basic blocks; their weights are zeros too)
    auto aload = [&](const unsigned (&off)[2], int ns, ARaw<ABF> (&dst)[2]) {
        const bool last = ns == NS - 1;
        const bool inlo = !last || 32 * ns + 4 * kq < K, inhi = !last || 32 * ns + 16 + 4 * kq < K;
        const int klo = inlo ? 32 * ns : 32 * ns - 4 * kq, khi = inhi ? 32 * ns + 16 : 32 * ns - 4 * kq;      // (column 32 ns of the row always exists)
#pragma unroll
        for (int mt = 0; mt < 2; mt++) {
            if constexpr (ABF) {
                const uint16_t* ap = A16 + off[mt];
                const u32v2 lo = *reinterpret_cast<const u32v2*>(ap + klo), hi = *reinterpret_cast<const u32v2*>(ap + khi);
                dst[mt].lo = inlo ? lo : (u32v2){0u, 0u};
                dst[mt].hi = inhi ? hi : (u32v2){0u, 0u};
            } else {
                const float* ap = p.A + off[mt];
                const float4 lo = *reinterpret_cast<const float4*>(ap + klo), hi = *reinterpret_cast<const float4*>(ap + khi);
                dst[mt].lo = inlo ? lo : make_float4(0.f, 0.f, 0.f, 0.f);
                dst[mt].hi = inhi ? hi : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    // piece q (0..3) of the fragment of one 16-row tile and slab (six products: one pair split per piece; otherwise the whole
    // fragment at piece 0)
    auto piece = [&](const ARaw<ABF>& a, int q, WsFrag& f) {
        if constexpr (SIX) {
            float x0, x1;
            ws_pair(a.lo, a.hi, q, x0, x1);
            unsigned h, m, l;
            ws_split2(x0, x1, h, m, l);
            f.h[q] = h; f.m[q] = m; f.l[q] = l;
        } else if (q == 0) {
            if constexpr (ABF) f.h = (u32v4){a.lo[0], a.lo[1], a.hi[0], a.hi[1]};
            else f.h = __builtin_bit_cast(u32v4, b16_cvt8(a.lo, a.hi));
        }
    };
    auto wfrag = [&](int idx, u32v4 (&dst)[NP]) {            // step idx = slab * NT + column tile
        const int ns = idx / NT, t = idx - ns * NT;
#pragma unroll
        for (int pl = 0; pl < NP; pl++) dst[pl] = Wl[((ns * NP + pl) * 4 + kq) * BN + 16 * t + li];
    };

    ARaw<ABF> ring[R][2];
    WsFrag fr[2][2];
    unsigned aoff[2], noff[2];
    rows(t0, aoff);
#pragma unroll
    for (int r = 0; r < R; r++) aload(aoff, r, ring[r]);
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int q = 0; q < 4; q++) piece(ring[0][mt], q, fr[0][mt]);
    aload(aoff, R, ring[0]);
    WS_T();

    // one tile pair (WM = 2) or single tile (WM = 1): fr[0] holds slab 0's fragments, the ring slabs 1 .. R (slab R in set 0);
    // `nxt` is the tile that follows (-1: none) - its slabs enter the ring as this tile's leave it, its slab-0 fragments are
    // split between the MFMAs of this tile's last slab
    auto body = [&](auto wm_tag, int tile, int nxt) {
        constexpr int WM = decltype(wm_tag)::value;
        constexpr int P = 8;                                  // fragment pieces per slab (2 tiles x 4 pairs; a single tile splits its phantom partner too)
        rows(nxt >= 0 ? nxt : tile, noff);                    // (no successor: the run-ahead loads and splits re-read this tile - harmless, and the loop stays branch-free)
        f32x4 acc[NT][WM];
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int mt = 0; mt < WM; mt++) acc[t][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        u32v4 wfr[2][NP];
        wfrag(0, wfr[0]);
#pragma unroll
        for (int ns = 0; ns < NS; ns++) {
            const int cur = ns & 1, nx = cur ^ 1;
            const int rs = (ns + 1) % R;                      // its ring set
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const int idx = ns * NT + t;
                if (idx + 1 < NS * NT) wfrag(idx + 1, wfr[(idx + 1) & 1]);
                ws_mfma<WM, SIX>(acc[t], wfr[idx & 1], fr[cur]);
#pragma unroll
                for (int i = 0; i < P; i++)
                    if (i * NT / P == t) piece(ring[rs][i >> 2], i & 3, fr[nx][i >> 2]);
            }
            // the set just split takes the slab R further on: of this tile, or of the next one
            const int lin = ns + 1 + R;
            if (lin < NS) aload(aoff, lin, ring[rs]);
            else aload(noff, lin - NS, ring[rs]);
        }
        if constexpr (NS & 1) {                               // slab 0 of the next tile was split into fr[1]: it is read from fr[0]
#pragma unroll
            for (int mt = 0; mt < 2; mt++) fr[0][mt] = fr[1][mt];
        }
        WS_T();
        // epilogue from the accumulators: row m = 16 (tile + mt) + li, channels n0 + 16 t + 4 kq .. + 3
#pragma unroll
        for (int mt = 0; mt < WM; mt++) {
            const int m = 16 * (tile + mt) + li;
            if (m < p.M) {
#pragma unroll
                for (int t = 0; t < NT; t++)
                    ws_store(p, acc[t][mt] + *reinterpret_cast<const f32x4*>(&Bl[16 * t + 4 * kq]), nullptr, m, n0 + 16 * t + 4 * kq);
            }
        }
        WS_T();
        aoff[0] = noff[0]; aoff[1] = noff[1];
    };
    // Phase stagger: every wave has the same work per item, so waves that start together compute together and store together.
    // The second wave of each SIMD (waves NW/2 ..) opens with a SINGLE tile and runs half an item out of phase with its partner
    // from then on (measured neutral on the layer alone, profiles/r05_ws_lab.txt; kept: it costs nothing).
    int tile = t0;
    if (wave >= NW / 2 && t1 - t0 >= 3) { body(std::integral_constant<int, 1>{}, tile, tile + 1); tile++; }
    for (; tile + 2 <= t1; tile += 2) body(std::integral_constant<int, 2>{}, tile, tile + 2 < t1 ? tile + 2 : -1);
    if (tile < t1) body(std::integral_constant<int, 1>{}, tile, -1);
#ifdef WS_TRACE
    if (lane == 0) { trc[31] = wall_clock64(); trc[29] = clock64(); }
#endif
#undef WS_T
}

// ring depth by slab count: divides it and is smaller (the set of slab s + 1 is refilled with slab s + 1 + R of this tile or of the next)
static constexpr int ws_ring(int ns) { return ns == 4 ? 2 : ns == 6 ? 3 : ns == 8 ? 4 : ns == 9 ? 3 : ns == 10 ? 5 : ns == 12 ? 4 : 1; }
static bool ws_slabs_ok(int ns, bool six) { return ns == 3 || ns == 4 || ns == 5 || ns == 6 || (six && (ns == 8 || ns == 10)); }      // (one-product forms beyond six slabs spill)
static constexpr int WS_NW = 8;                              // waves per block (two per SIMD: one's loads, split and epilogue under the other's MFMAs)
static size_t ws_lds_bytes(int ns, int nt, bool six) { return (size_t)ns * (six ? 3 : 1) * 4 * 16 * nt * 16 + (size_t)16 * nt * 4; }
// column-block width (16-column units): what the caller asks for when the kernel has it and the columns fit LDS, else the widest that does
static int ws_pick_nt(const PwParams& p) {
    const int ns = (p.K + 31) / 32;
    const bool six = p.prec == 0;
    if (!six) return 4;                                      // one-product form: 64-column blocks only
    const int want = (p.nt == 4 || p.nt == 6 || p.nt == 8) ? p.nt : 8;
    for (int nt : {want, 8, 6, 4})
        if (nt <= want && ws_lds_bytes(ns, nt, six) <= 160 * 1024) return nt;
    return 0;
}
// layers it takes: 3 .. 6 (six products: also 8, 10) slabs whose column block fits LDS, N >= 64, no squeeze-excite scale on A (the
// expands have none), at least one tile pair per wave of one block
bool pw_ws_ok(const PwParams& p) {
    if (p.sw & PW_SW_WS_OFF) return false;
    if (p.prec != 0 && p.prec != 1) return false;
    if ((p.K & 3) || (p.N & 3) || p.N < 64 || p.ascale) return false;
    if (p.a_bf16 && (p.prec != 1 || (p.K & 7))) return false;
    const int ns = (p.K + 31) / 32;
    if (!ws_slabs_ok(ns, p.prec == 0)) return false;
    if (!ws_pick_nt(p)) return false;
    return (p.M + 15) / 16 >= 2 * WS_NW && (long)((p.M + 15) / 16) * (2 * WS_NW * 256) < (1l << 31);
}
// ... and the calls it is worth launching for: blocks on at least half of the CUs with a tile pair per wave (smaller calls - a few
// clips, the 64-clip chunks of a blocking host call - keep the tiled kernels with their shrinking grids: same bits)
bool pw_ws_fills(const PwParams& p) {
    const int nt = ws_pick_nt(p);
    if (!nt) return false;
    const int nblk_n = (p.N + 16 * nt - 1) / (16 * nt), mt16 = (p.M + 15) / 16;
    const int cus = device_cus();
    const int G = std::min(std::max(1, cus / nblk_n), mt16 / (2 * WS_NW));
    return G * nblk_n >= cus / 2;
}
void launch_pw_ws(const PwParams& p, const uint16_t* Wimg, int Npad, hipStream_t s) {
    g_pw_ws_launches.fetch_add(1, std::memory_order_relaxed);
    const int ns = (p.K + 31) / 32;
    const bool six = p.prec == 0, abf = p.a_bf16 != 0;
    const int nt = ws_pick_nt(p);
    const int BN = 16 * nt;
    const size_t lds = ws_lds_bytes(ns, nt, six);
    const int nblk_n = (p.N + BN - 1) / BN, mt16 = (p.M + 15) / 16;
    int G = std::max(1, device_cus() / nblk_n);              // one resident generation of blocks (one per CU: 512 threads at up to 256 registers): each loads its columns once
    G = std::min(G, std::max(1, mt16 / (2 * WS_NW)));        // (at least one tile pair per wave)
    const unsigned nblk = (unsigned)G * (unsigned)nblk_n;
    const unsigned GW = (unsigned)G * WS_NW;
    const FDiv dn = make_fdiv((unsigned)nblk_n);
#define WS_LAUNCH(NT_, NS_, SIX_, ABF_) do { \
        constexpr auto kern = &k_pw_ws<NT_, NS_, ws_ring(NS_), SIX_, ABF_, WS_NW>; \
        /* (the attribute belongs to the function ON THE CURRENT DEVICE: once per function and device, kernels.h lds_limit_once) */ \
        lds_limit_once<kern>(160 * 1024); \
        hipLaunchKernelGGL(kern, dim3(nblk), dim3(64 * WS_NW), lds, s, p, Wimg, Npad, nblk_n, nblk, GW, dn); } while (0)
#define WS_SIX(NT_) switch (ns) { case 3: WS_LAUNCH(NT_, 3, true, false); break; case 4: WS_LAUNCH(NT_, 4, true, false); break; \
                                   case 5: WS_LAUNCH(NT_, 5, true, false); break; default: WS_LAUNCH(NT_, 6, true, false); break; }
#define WS_ONE(ABF_) switch (ns) { case 3: WS_LAUNCH(4, 3, false, ABF_); break; case 4: WS_LAUNCH(4, 4, false, ABF_); break; \
                                   case 5: WS_LAUNCH(4, 5, false, ABF_); break; default: WS_LAUNCH(4, 6, false, ABF_); break; }
    if (six) {
        if (ns == 10) WS_LAUNCH(4, 10, true, false);         // (ten slabs: 64 columns fill LDS)
        else if (ns == 8) { if (nt == 4) WS_LAUNCH(4, 8, true, false); else WS_LAUNCH(6, 8, true, false); }
        else if (nt == 4) WS_SIX(4) else if (nt == 6) WS_SIX(6) else WS_SIX(8)
    } else if (abf) WS_ONE(true) else WS_ONE(false)
#undef WS_LAUNCH
#undef WS_SIX
#undef WS_ONE
}

// (k_pw_f32s - the fp32 twin of k_pw_b16s for the HBM-bound early projections 32 -> 16, 96 -> 24, 144 -> 24: the whole weight matrix
// as f32-MFMA operands in a wave's registers, 16-byte loads, MFMAs and 16-byte stores, no LDS, no barrier, bit-identical to
// k_pw_gemm - was built and measured in round 5: 128 / 92 / 151 us against 104 / 81 / 137 us for the tiled kernel, whatever the
// number of waves; those layers already move 4.4-5.4 TB/s.  Removed: profiles/r05_pw_lab_f32_family.txt)

// ================================================================================================ small calls: the long-K layers as a latency problem
// One clip per Predict is the product's own call pattern (cmd/benchmark/benchmark.go:99-133, orchestrator.go:531).  There the
// projections (K = 480 .. 1152) and the dense head (K = 1024) are a handful of tiles whose K loop is a chain of memory latencies: the
// tiled kernels fetch a slab one or two ahead and meet at a barrier per slab - 36 slabs x ~0.8 us = 28.5 us for 1.4 MFLOP at one clip, the
// same at eight (tools/latency_small.py).  k_pw_lat: a block is one 16-row tile x four column groups of 16 NT columns (one per wave).
//   * weight fragments come straight from the plan-time image (which is in fragment order) into a per-wave ring of DW register stages
//     that runs DW slabs ahead - no LDS, nobody to wait for;
//   * the rows' operand work is shared: K is walked in groups of four slabs, wave w loads (two groups ahead), scales and splits slab
//     4 g + w of the group and publishes its three bf16 planes to LDS, ONE barrier per group, then every wave runs the group's
//     4 x 6 NT MFMAs from those fragments (the first form of this kernel had every wave split every slab: 52 VALU per 6 MFMAs on
//     one in-order wave - 16 us a projection; tools/ubench/pw_lab '@' shapes);
//   * epilogue straight from the accumulators.
// Same image, same K order, same six products per accumulator in the same order: bit-identical to k_pw_bx3 - a clip's logits still
// do not depend on the call's size.
template <int NT, bool SC, int DW>
__global__ __launch_bounds__(256) void k_pw_lat(PwParams p, const uint16_t* __restrict__ Wimg, int Npad, int ngroups, FDiv dhw) {
    __shared__ __attribute__((aligned(16))) u32v4 frl[2][4][3][64];       // [group parity][slab of the group][plane][lane]
    const int lane = threadIdx.x & 63, li = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int gblocks = (ngroups + 3) >> 2;                  // blocks per row tile
    const int rt = (int)blockIdx.x / gblocks;
    const int cg = min(((int)blockIdx.x - rt * gblocks) * 4 + wave, ngroups - 1);       // (a surplus wave repeats the last group: it shares the split work)
    const bool live = ((int)blockIdx.x - rt * gblocks) * 4 + wave < ngroups;
    const int K = p.K, nslab = (K + 31) >> 5, ngrp = (nslab + 3) >> 2;
    const int m = min(16 * rt + li, p.M - 1), n0 = 16 * NT * cg;
    const float* arow = p.A + (size_t)m * K + 4 * kq;
    const float* srow = SC ? p.ascale + (size_t)fdiv((unsigned)m, dhw) * K + 4 * kq : nullptr;
    const u32v4* W16 = reinterpret_cast<const u32v4*>(Wimg);
    unsigned wcol[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) wcol[t] = (unsigned)kq * (unsigned)Npad + (unsigned)min(n0 + 16 * t + li, Npad - 1);

    struct AStage { float4 alo, ahi, slo, shi; };
    AStage ast[2];                                           // this wave's slab of groups g and g + 1 (loaded two groups ahead)
    auto aload = [&](int sl, AStage& s_) {
        // (a slab beyond K, or the missing quads of a short last slab: read the row's first quad instead and zero by a select)
        const bool inlo = 32 * sl + 4 * kq < K, inhi = 32 * sl + 16 + 4 * kq < K;
        const int klo = inlo ? 32 * sl : -4 * kq, khi = inhi ? 32 * sl + 16 : -4 * kq;
        const float4 lo = *reinterpret_cast<const float4*>(arow + klo), hi = *reinterpret_cast<const float4*>(arow + khi);
        s_.alo = inlo ? lo : make_float4(0.f, 0.f, 0.f, 0.f);
        s_.ahi = inhi ? hi : make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (SC) {
            s_.slo = *reinterpret_cast<const float4*>(srow + klo);
            s_.shi = *reinterpret_cast<const float4*>(srow + khi);
        }
    };
    u32v4 wst[DW][NT][3];
    auto wload = [&](int sl, u32v4 (&dst)[NT][3]) {
        const u32v4* Ws = W16 + (size_t)sl * 12 * Npad;
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int pl = 0; pl < 3; pl++) dst[t][pl] = Ws[wcol[t] + (unsigned)(pl * 4) * (unsigned)Npad];
    };
    auto publish = [&](const AStage& s_, int par) {          // scale, split exactly, three planes to LDS
        float4 v0 = s_.alo, v1 = s_.ahi;
        if constexpr (SC) {
            v0.x *= s_.slo.x; v0.y *= s_.slo.y; v0.z *= s_.slo.z; v0.w *= s_.slo.w;
            v1.x *= s_.shi.x; v1.y *= s_.shi.y; v1.z *= s_.shi.z; v1.w *= s_.shi.w;
        }
        WsFrag f;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            float x0, x1;
            ws_pair(v0, v1, q, x0, x1);
            unsigned h, mm, l;
            ws_split2(x0, x1, h, mm, l);
            f.h[q] = h; f.m[q] = mm; f.l[q] = l;
        }
        frl[par][wave][0][lane] = f.h; frl[par][wave][1][lane] = f.m; frl[par][wave][2][lane] = f.l;
    };
    f32x4 acc[NT][1];
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f};

    aload(wave, ast[0]);
    aload(4 + wave, ast[1]);
#pragma unroll
    for (int d = 0; d < DW; d++) if (d < nslab) wload(d, wst[d]);
    static_assert(DW % 8 == 0, "groups are walked DW / 4 at a time: stage and A-set indices are static when that count is even");
    // groups are walked DW / 4 at a time so that every register stage index is static
    for (int g0 = 0; g0 < ngrp; g0 += DW / 4) {
#pragma unroll
        for (int gg = 0; gg < DW / 4; gg++) {
            const int g = g0 + gg;
            if (g < ngrp) {
                publish(ast[gg & 1], gg & 1);
                aload(4 * (g + 2) + wave, ast[gg & 1]);     // (beyond K: zeros from a valid address)
                __syncthreads();
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const int sl = 4 * g + d;
                    if (sl < nslab) {
                        WsFrag f[2];
                        f[0].h = frl[gg & 1][d][0][lane]; f[0].m = frl[gg & 1][d][1][lane]; f[0].l = frl[gg & 1][d][2][lane];
#pragma unroll
                        for (int t = 0; t < NT; t++) ws_mfma<1, true>(acc[t], wst[4 * gg + d][t], f);
                        if (sl + DW < nslab) wload(sl + DW, wst[4 * gg + d]);
                    }
                }
            }
        }
    }
    const int mrow = 16 * rt + li;
    if (live && mrow < p.M) {
#pragma unroll
        for (int t = 0; t < NT; t++) ws_store(p, acc[t][0], p.bias, mrow, n0 + 16 * t + 4 * kq);
    }
}

static std::atomic<long> g_pw_lat_launches{0};
// calls it takes: six-product engines, fp32 activations, K of at least eight slabs, few enough (row tile, column tile) pairs that the
// tiled kernels would be a chain of latencies rather than a full chip
bool pw_lat_ok(const PwParams& p) {
    if (p.sw & PW_SW_LAT_OFF) return false;
    if (p.prec != 0 || p.a_bf16 || (p.K & 3) || p.K < 256) return false;
    if ((p.N & 3) && (p.out_bf16 || p.res_bf16)) return false;
    // Every row tile streams the whole weight image through its waves' registers, so the call must be small in tiles AND in
    // re-read weight bytes (measured crossover, tools/ubench/pw_lab '@' shapes: 1152 -> 192 wins up to 32 clips = 96 row tiles x 1.3 MB,
    // loses at 64; the dense head (40 MB of image) wins at 8 clips = one row tile, loses at 64 = four).  BNHIP_PW_LAT_TILES: lab sweeps, read once.
    static const long max_tiles = getenv("BNHIP_PW_LAT_TILES") ? atol(getenv("BNHIP_PW_LAT_TILES")) : 2048;
    const long rt = (p.M + 15) / 16, ct = (p.N + 15) / 16;
    return rt * ct <= max_tiles && (double)rt * (double)p.K * (double)(16 * ct) * 6.0 <= 130e6 * (double)max_tiles / 2048.0;      // (any HW: every lane reads the scale row of its own clip)
}
void launch_pw_lat(const PwParams& p, const uint16_t* Wimg, int Npad, hipStream_t s) {
    g_pw_lat_launches.fetch_add(1, std::memory_order_relaxed);
    const int rt = (p.M + 15) / 16, ct = (p.N + 15) / 16;
    // two column tiles per wave once one each already gives the chip a wave per SIMD
    const int nt = (long)rt * ct >= 2048 / 2 ? 2 : 1;
    const int ngroups = (ct + nt - 1) / nt;
    const unsigned nblk = (unsigned)rt * (unsigned)((ngroups + 3) / 4);
    const FDiv dhw = make_fdiv((unsigned)std::max(p.HW, 1));
    if (p.ascale) {
        if (nt == 1) hipLaunchKernelGGL((k_pw_lat<1, true, 8>), dim3(nblk), dim3(256), 0, s, p, Wimg, Npad, ngroups, dhw);
        else hipLaunchKernelGGL((k_pw_lat<2, true, 8>), dim3(nblk), dim3(256), 0, s, p, Wimg, Npad, ngroups, dhw);
    } else {
        if (nt == 1) hipLaunchKernelGGL((k_pw_lat<1, false, 8>), dim3(nblk), dim3(256), 0, s, p, Wimg, Npad, ngroups, dhw);
        else hipLaunchKernelGGL((k_pw_lat<2, false, 8>), dim3(nblk), dim3(256), 0, s, p, Wimg, Npad, ngroups, dhw);
    }
}

}  // namespace bnhip

extern "C" long bnhip_debug_pw_ws_launches(void) { return bnhip::g_pw_ws_launches.load(std::memory_order_relaxed); }
extern "C" long bnhip_debug_pw_lat_launches(void) { return bnhip::g_pw_lat_launches.load(std::memory_order_relaxed); }
