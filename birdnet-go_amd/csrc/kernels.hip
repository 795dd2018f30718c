// gfx950 (CDNA4 / MI355X) kernels for the BirdNET inference hot path.  fp32 throughout (the
// reference documents f16 as fatal for v2.4: internal/classifier/model_openvino.go:99-103);
// contractions run on the f32-input MFMA (v_mfma_f32_16x16x4_f32), whose result is bit-for-bit a
// k-ordered fmaf chain, so numerics are those of a plain fp32 CPU kernel.
//
// Layouts: activations NHWC fp32; pointwise/FC weights [N][K] (TFLite OHWI with 1x1 == [Cout][Cin]),
// depthwise weights [kh][kw][C], stem weights re-laid to [kh][kw][Cin][Cout] at plan time.
#include "kernels.h"
#include "pw_common.h"
#include "fft_r8.h"

#include <algorithm>
#include <type_traits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace bnhip {

// ------------------------------------------------------------------------------------------ ingest
// internal/analysis/process.go:491-495: float32(int16)/32768
__global__ void k_pcm16_to_f32(const int16_t* __restrict__ pcm, float* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = (float)pcm[i] / 32768.0f;
}
// internal/audiocore/convert/pcm.go:242-268: 24-bit little-endian with two's-complement sign extension / 8388608,
// 32-bit / 2147483648 (float32(int32) rounds to nearest even in Go and here; the divisors are powers of two)
__global__ void k_pcm24_to_f32(const uint8_t* __restrict__ pcm, float* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        int32_t v = (int32_t)pcm[3 * i] | ((int32_t)pcm[3 * i + 1] << 8) | ((int32_t)pcm[3 * i + 2] << 16);
        if (v & 0x00800000) v |= ~0x00FFFFFF;
        out[i] = (float)v / 8388608.0f;
    }
}
__global__ void k_pcm32_to_f32(const int32_t* __restrict__ pcm, float* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = (float)pcm[i] / 2147483648.0f;
}
void launch_pcm_to_f32(const void* pcm, int bits, float* out, size_t n, hipStream_t s) {
    int blocks = (int)((n + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    if (bits == 16) hipLaunchKernelGGL(k_pcm16_to_f32, dim3(blocks), dim3(256), 0, s, static_cast<const int16_t*>(pcm), out, n);
    else if (bits == 24) hipLaunchKernelGGL(k_pcm24_to_f32, dim3(blocks), dim3(256), 0, s, static_cast<const uint8_t*>(pcm), out, n);
    else hipLaunchKernelGGL(k_pcm32_to_f32, dim3(blocks), dim3(256), 0, s, static_cast<const int32_t*>(pcm), out, n);
}

// ------------------------------------------------------------------------------------------ front-end
// One block per clip: min(x) and fl(max(x)-min)+eps, i.e. REDUCE_MIN -> SUB -> REDUCE_MAX -> ADD eps.
__global__ __launch_bounds__(1024) void k_clip_minmax(const float* __restrict__ x, int n_samples, float eps,
                                                      float2* __restrict__ mm) {
    const float* xc = x + (size_t)blockIdx.x * n_samples;
    float mn = INFINITY, mx = -INFINITY;
    if ((n_samples & 3) == 0 && ((((size_t)blockIdx.x * n_samples) & 3) == 0)) {
        const float4* x4 = reinterpret_cast<const float4*>(xc);
        // batches of 8 independent loads: a rolled loop walks the clip one L2/HBM round trip at a time (24 us for one clip)
        const int n4 = n_samples / 4;
        for (int i0 = threadIdx.x; i0 < n4; i0 += 8 * blockDim.x) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                int i = i0 + u * blockDim.x;
                v[u] = i < n4 ? x4[i] : x4[i0];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                mn = fminf(fminf(mn, v[u].x), fminf(v[u].y, fminf(v[u].z, v[u].w)));
                mx = fmaxf(fmaxf(mx, v[u].x), fmaxf(v[u].y, fmaxf(v[u].z, v[u].w)));
            }
        }
    } else {
        for (int i = threadIdx.x; i < n_samples; i += blockDim.x) { float v = xc[i]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
    }
    for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_down(mn, o, 64)); mx = fmaxf(mx, __shfl_down(mx, o, 64)); }
    __shared__ float smn[16], smx[16];
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { smn[w] = mn; smx[w] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int nw = blockDim.x >> 6;
        for (int i = 1; i < nw; i++) { mn = fminf(mn, smn[i]); mx = fmaxf(mx, smx[i]); }
        float range = mx - mn;          // == max_i fl(x_i - mn): rounding is monotone
        mm[blockIdx.x] = make_float2(mn, range + eps);
    }
}
// Small calls (one clip per Predict is the product's call pattern): one block walks a 576 KB clip in ~5 dependent round trips
// (17-20 us at one clip).  G blocks per clip take a contiguous part each, publish (min, max) with agent-scope atomic stores, and the
// block that arrives last at the clip's counter combines the G pairs - min / max are exact whatever the grouping, so the result
// is the one-block kernel's bit for bit.  scratch: [clip][2 G + 2] floats of the plan's arena that nothing else ever uses, zeroed
// once (the counter resets itself).  Cross-XCD visibility: payload and counter are agent-scope atomics on both sides
// (MI355X_MICROARCH.md, "valid forms").
__global__ __launch_bounds__(1024) void k_clip_minmax_parts(const float* __restrict__ x, int n_samples, float eps, int G, float* __restrict__ scratch,
                                                            float2* __restrict__ mm) {
    const int clip = blockIdx.x / G, part = blockIdx.x - clip * G;
    const float4* x4 = reinterpret_cast<const float4*>(x + (size_t)clip * n_samples);
    const int n4 = n_samples / 4, per = (n4 + G - 1) / G, lo = part * per, hi = min(lo + per, n4);
    float mn = INFINITY, mx = -INFINITY;
    for (int i0 = lo + threadIdx.x; i0 < hi; i0 += 4 * blockDim.x) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int i = i0 + u * blockDim.x; v[u] = i < hi ? x4[i] : x4[i0]; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            mn = fminf(fminf(mn, v[u].x), fminf(v[u].y, fminf(v[u].z, v[u].w)));
            mx = fmaxf(fmaxf(mx, v[u].x), fmaxf(v[u].y, fmaxf(v[u].z, v[u].w)));
        }
    }
    for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_down(mn, o, 64)); mx = fmaxf(mx, __shfl_down(mx, o, 64)); }
    __shared__ float smn[16], smx[16];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { smn[w] = mn; smx[w] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < (int)(blockDim.x >> 6); i++) { mn = fminf(mn, smn[i]); mx = fmaxf(mx, smx[i]); }
        float* sc = scratch + (size_t)clip * (2 * G + 2);
        __hip_atomic_store(sc + 2 * part, mn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sc + 2 * part + 1, mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        unsigned* cnt = reinterpret_cast<unsigned*>(sc + 2 * G);
        if (atomicAdd(cnt, 1u) == (unsigned)(G - 1)) {           // every other part of this clip is published
            __threadfence();
            for (int g = 0; g < G; g++) {
                mn = fminf(mn, __hip_atomic_load(sc + 2 * g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                mx = fmaxf(mx, __hip_atomic_load(sc + 2 * g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            }
            mm[clip] = make_float2(mn, (mx - mn) + eps);
            __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next call on this arena
        }
    }
}
void launch_clip_minmax(const float* x, int n_clips, int n_samples, float eps, float2* mm, float* scratch, hipStream_t s) {
    // (the parts kernel needs whole, aligned quads; above 16 clips one block per clip already fills enough of the chip)
    if (scratch && n_clips <= 16 && (n_samples & 3) == 0 && n_samples >= 16 * 4096) {
        hipLaunchKernelGGL(k_clip_minmax_parts, dim3(n_clips * kMinMaxParts), dim3(1024), 0, s, x, n_samples, eps, kMinMaxParts, scratch, mm);
        return;
    }
    hipLaunchKernelGGL(k_clip_minmax, dim3(n_clips), dim3(1024), 0, s, x, n_samples, eps, mm);
}

// Fused normalise -> frame -> window -> (real-DFT * mel) -> x^p1 -> x^p2 -> NHWC store.
// Because the graph keeps only the REAL part of the STFT (CAST complex64->float32) and applies the
// mel matrix before squaring, everything between the window multiply and the first POW is linear:
//   mel[f, m] = sum_n fl32(xn[f*hop + n] * w[n]) * G[n, m],   G[n, m] = sum_k cos(2*pi*k*n/N) * Mel[k, m]
// TFLite evaluates RFFT2D in double precision (rfft2d.cc runs Ooura fft2d on doubles), so bins that
// cancel to ~0 really are ~0 there; the subsequent power-law compression (x^0.45) amplifies any
// accumulation noise in such bins by orders of magnitude (digital silence - the reference benchmark's
// own input, cmd/benchmark/benchmark.go:99-101 - is the extreme case).  The contraction therefore
// runs on the f64 MFMA (v_mfma_f64_16x16x4_f64) with G held in fp64, while the window product is
// rounded to fp32 first exactly as the graph's MUL does.  A rows are overlapping windows of the
// LDS-resident clip segment (never materialised), B = G streamed from L2 in 32-row chunks.
// cos(2*pi*k*(N-n)/N) = cos(2*pi*k*n/N) makes G symmetric in n, so the windowed frame is folded first,
//   a[n'] = double(fl32(x[n']*w[n'])) + double(fl32(x[N-n']*w[N-n']))   (exact in fp64), n' = 0..N/2,
// halving the contraction length (K = N/2+1) at no cost in accuracy.
// Block: 64 frames x (16*NT) mel columns, 4 waves, wave w owns frames [16w,16w+16) x all NT tiles.
typedef double f64x4 __attribute__((ext_vector_type(4)));
// Tile parameters: FT frames x all mel tiles per block, waves = (FT/16 frame groups) x (WN mel groups), G streamed
// in KC-row chunks.  Two shapes are instantiated:
//   <FT 32, KC 16>: ~80 KB of LDS -> two blocks per CU (measured 3 % faster than the 64-frame shape);
//   <FT 64, KC 32|16>: fallback when the smaller shape's LDS does not allow two blocks anyway.
// Measured ceilings on MI355X (tools/ubench/mfma_f64*.hip): the f64 MFMA sustains 68 TF with two waves per SIMD and
// nothing else; FP VALU work does NOT overlap it (2 v_mul_f32 per MFMA -> 54 TF, 8 -> 49 TF; integer VALU is free) and
// an LDS read consumed right away costs far more (1 per MFMA -> 57 TF, 4 -> 38 TF).  This kernel needs 5 FP ops per
// 3 MFMAs to build the folded A operand, which bounds it near 55 TF; it reaches 39 TF (ch0) / 33 TF (ch1).
template <int NT, int WN, int FT, int KC>
__global__ __launch_bounds__(64 * (FT / 16) * WN) void k_frontend(FrontendParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NTP = NT * 16;
    constexpr int NTW = NT / WN;                 // mel tiles per wave
    constexpr int FG = FT / 16;                  // frame groups
    constexpr int NTHR = 64 * FG * WN;
    constexpr int GS = NTP + 16;                 // LDS row stride (doubles): k-rows land 32 banks apart for ds_read_b64
    constexpr int GQ = (KC * (NTP / 4) + NTHR - 1) / NTHR;   // double4 (32 B) per thread per chunk
    const int seg_len = (FT - 1) * p.hop + p.Lfft + 4;      // +4: the n'=0 mirror reads one past the frame (weight 0)
    float* seg = smem;
    float* win = smem + ((seg_len + 3) & ~3);                // [Kp] window at n'
    float* win2 = win + p.Kp;                                // [Kp] window at the mirror index (0 where there is none)
    double* Gs = reinterpret_cast<double*>(win2 + p.Kp);     // Kp is a multiple of 16 -> 16-byte aligned

    const int b = blockIdx.y;
    const int f0 = blockIdx.x * FT;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = (tid >> 6) % FG, nh = (tid >> 6) / FG;
    const int li = lane & 15, kq = lane >> 4;

    // ---- stage + normalise the clip segment ((x - min) / (range+eps) - 0.5) * 2, exactly the graph's op order
    {
        const float2 mm = p.mm[b];
        const float* xc = p.x + (size_t)b * p.n_samples;
        const int s0 = f0 * p.hop;
        // all of this thread's loads are issued before the first use (a rolled one-load-per-iteration loop exposes
        // the global latency once per element)
        auto norm = [&](float x) { float t = x - mm.x; t = t / mm.y; t = t - p.norm_sub; return t * p.norm_mul; };
        const int lim = min(seg_len - 4, p.n_samples - s0);      // samples of this segment that exist
        if ((((size_t)xc & 15) | (s0 & 3)) == 0) {
            constexpr int UN = 10;
            const int nq = (seg_len + 3) >> 2;
            for (int q0 = tid; q0 < nq; q0 += NTHR * UN) {
                float4 v[UN];
#pragma unroll
                for (int u = 0; u < UN; u++) {
                    int q = q0 + u * NTHR;
                    v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (4 * q + 3 < lim) v[u] = *reinterpret_cast<const float4*>(xc + s0 + 4 * q);
                    else if (4 * q < lim) {
                        v[u].x = xc[s0 + 4 * q];
                        if (4 * q + 1 < lim) v[u].y = xc[s0 + 4 * q + 1];
                        if (4 * q + 2 < lim) v[u].z = xc[s0 + 4 * q + 2];
                    }
                }
#pragma unroll
                for (int u = 0; u < UN; u++) {
                    int q = q0 + u * NTHR;
                    if (q < nq) {
                        float4 o;
                        o.x = 4 * q < lim ? norm(v[u].x) : 0.f; o.y = 4 * q + 1 < lim ? norm(v[u].y) : 0.f;
                        o.z = 4 * q + 2 < lim ? norm(v[u].z) : 0.f; o.w = 4 * q + 3 < lim ? norm(v[u].w) : 0.f;
                        *reinterpret_cast<float4*>(seg + 4 * q) = o;
                    }
                }
            }
        } else {
            for (int i = tid; i < seg_len; i += NTHR) seg[i] = i < lim ? norm(xc[s0 + i]) : 0.0f;
        }
        for (int i = tid; i < p.Kp; i += NTHR) { win[i] = p.window[i]; win2[i] = p.window[p.Kp + i]; }
    }

    // G chunks travel global -> registers (two stages: the load for chunk c+3 is issued while chunk c computes, so it
    // has two full iterations to arrive; one iteration is shorter than the L2 latency under load) -> LDS (2 buffers)
    const double4* G4 = reinterpret_cast<const double4*>(p.G);
    double4 greg[2][GQ];
    auto gload = [&](int chunk, double4 (&gr)[GQ]) {
#pragma unroll
        for (int q = 0; q < GQ; q++) {
            int idx = tid + NTHR * q;
            double4 v = make_double4(0., 0., 0., 0.);
            if (idx < KC * (NTP / 4)) v = G4[(size_t)chunk * KC * (NTP / 4) + idx];
            gr[q] = v;
        }
    };
    auto gstore = [&](int buf, const double4 (&gr)[GQ]) {
#pragma unroll
        for (int q = 0; q < GQ; q++) {
            int idx = tid + NTHR * q;
            if (idx < KC * (NTP / 4)) {
                int r = idx / (NTP / 4), c4 = idx % (NTP / 4);
                *reinterpret_cast<double4*>(&Gs[buf * KC * GS + r * GS + 4 * c4]) = gr[q];
            }
        }
    };

    f64x4 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; t++) acc[t] = (f64x4){0., 0., 0., 0.};

    // Register pipeline, one chunk deep: while the KS*NTW MFMAs of chunk ch run from registers, the operands of
    // chunk ch+1 are fetched from LDS (folded window products + G fragments) and G(ch+2) travels global -> regs ->
    // LDS.  (ISA of the first version: every k-step was ds_read -> s_waitcnt lgkmcnt(0) -> mfma, i.e. the LDS
    // latency was exposed once per k-step with only two waves per SIMD to cover it.)
    constexpr int KS = KC / 4;
    const int nchunks = p.Kp / KC;
    const float* arow = seg + (16 * wave + li) * p.hop + kq;                 // x[f*hop + n']
    const float* mrow = seg + (16 * wave + li) * p.hop + p.Lfft - kq;        // x[f*hop + N - n']
    auto fetch = [&](int ch, double (&av)[KS], double (&bm)[KS][NTW]) {
        const double* gb = Gs + (ch & 1) * KC * GS + kq * GS + nh * NTW * 16 + li;
        const float* ab = arow + ch * KC;
        const float* mb = mrow - ch * KC;
        const float* wb = win + ch * KC + kq;
        const float* wb2 = win2 + ch * KC + kq;
#pragma unroll
        for (int kk = 0; kk < KS; kk++) {
            float xw = ab[kk * 4] * wb[kk * 4];          // fp32 products, rounded like the graph's window MUL
            float xm = mb[-kk * 4] * wb2[kk * 4];
            av[kk] = (double)xw + (double)xm;            // exact fold in fp64
#pragma unroll
            for (int t = 0; t < NTW; t++) bm[kk][t] = gb[kk * 4 * GS + t * 16];
        }
    };
    auto mma = [&](const double (&av)[KS], const double (&bm)[KS][NTW]) {
#pragma unroll
        for (int kk = 0; kk < KS; kk++)
#pragma unroll
            for (int t = 0; t < NTW; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], bm[kk][t], acc[t], 0, 0, 0);
    };
    double a0[KS], b0[KS][NTW], a1[KS], b1[KS][NTW];
    gload(0, greg[0]);
    if (nchunks > 1) gload(1, greg[1]);
    gstore(0, greg[0]);
    if (nchunks > 2) gload(2, greg[0]);
    __syncthreads();
    fetch(0, a0, b0);
    if (nchunks > 1) gstore(1, greg[1]);
    __syncthreads();
    // iteration invariant: (ac,bc) = chunk ch in registers, LDS buffer (ch+1)&1 = G(ch+1) visible to all
    // (ac,bc) = chunk ch in registers, LDS buffer (ch+1)&1 = G(ch+1) visible to all, gc = G(ch+2) in flight/registers
    auto iter = [&](int ch, double (&ac)[KS], double (&bc)[KS][NTW], double (&an)[KS], double (&bn)[KS][NTW],
                    double4 (&gc)[GQ], double4 (&gn)[GQ]) {
        if (ch + 3 < nchunks) gload(ch + 3, gn);
        // unconditional (the last iteration re-reads its own chunk, unused) so that fetch and the MFMA burst share a
        // basic block; the group barriers then interleave them: each 64-cycle f64 MFMA leaves 15 issue slots
        fetch(min(ch + 1, nchunks - 1), an, bn);
        mma(ac, bc);
#pragma unroll
        for (int i = 0; i < KS * NTW; i++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // 2 LDS reads
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);   // 3 VALU
        }
        if (ch + 2 < nchunks) gstore(ch & 1, gc);   // buffer ch&1 was last read (chunk ch) before the previous barrier
        __syncthreads();
    };
    for (int ch = 0; ch < nchunks; ch += 2) {
        iter(ch, a0, b0, a1, b1, greg[0], greg[1]);
        if (ch + 1 < nchunks) iter(ch + 1, a1, b1, a0, b0, greg[1], greg[0]);
    }

    // ---- epilogue: f64 C/D layout D[row = kq + 4*r][col = li]  (row = frame, col = mel)
#pragma unroll
    for (int t = 0; t < NTW; t++) {
        int m = (nh * NTW + t) * 16 + li;
        if (m >= p.n_mels) continue;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int f = f0 + 16 * wave + kq + 4 * r;
            if (f >= p.F) continue;
            float v = (float)acc[t][r];
            float y = (p.p1 == 2.0f) ? v * v : powf(v, p.p1);
            if (p.p2 != 1.0f) y = powf(y, p.p2);
            p.out[(((size_t)b * p.n_mels + m) * p.F + f) * p.C + p.c] = y;
        }
    }
}

static size_t fe_lds_bytes(int FT, int KC, int Lfft, int Kp, int hop, int NTP) {
    int seg_len = (FT - 1) * hop + Lfft + 4;
    return (size_t)(((seg_len + 3) & ~3) + 2 * Kp) * sizeof(float) + (size_t)2 * KC * (NTP + 16) * sizeof(double);
}
// two blocks of the 32-frame shape must fit in the CU's 160 KB, otherwise the 64-frame shape is used
static bool fe_small_shape(int Lfft, int Kp, int hop, int NTP) {
    return 2 * (fe_lds_bytes(32, 16, Lfft, Kp, hop, NTP) + 512) <= 160 * 1024;
}
int frontend_kc(int Lfft, int hop, int NTP) {
    int Kp16 = (Lfft / 2 + 1 + 15) / 16 * 16;
    return fe_small_shape(Lfft, Kp16, hop, NTP) ? 16 : 32;
}
size_t frontend_lds_bytes(int Lfft, int Kp, int hop, int NTP) {
    return fe_small_shape(Lfft, Kp, hop, NTP) ? fe_lds_bytes(32, 16, Lfft, Kp, hop, NTP) : fe_lds_bytes(64, Kp % 32 ? 16 : 32, Lfft, Kp, hop, NTP);
}

template <int NT, int WN, int FT, int KC>
static void launch_frontend_shape(const FrontendParams& p, hipStream_t s) {
    size_t lds = fe_lds_bytes(FT, KC, p.Lfft, p.Kp, p.hop, p.NTP);
    // (per launch: the limit is an attribute of the function on the CURRENT device - see launch_stft_bins)
    lds_limit_once<&k_frontend<NT, WN, FT, KC>>(160 * 1024);
    dim3 grid((p.F + FT - 1) / FT, p.n_clips);
    hipLaunchKernelGGL((k_frontend<NT, WN, FT, KC>), grid, dim3(64 * (FT / 16) * WN), lds, s, p);
}
template <int NT, int WN>
static void launch_frontend_nt(const FrontendParams& p, hipStream_t s) {
    static const char* force = getenv("BNHIP_FE_SHAPE");       // experiment switch: "64" forces the large shape
    bool small = fe_small_shape(p.Lfft, p.Kp, p.hop, p.NTP) && p.Kp % 16 == 0 && !(force && atoi(force) == 64);
    if (small) launch_frontend_shape<NT, WN, 32, 16>(p, s);
    else if (p.Kp % 32 == 0) launch_frontend_shape<NT, WN, 64, 32>(p, s);
    else launch_frontend_shape<NT, WN, 64, 16>(p, s);
}
void launch_frontend(const FrontendParams& p, hipStream_t s) {
    // even tile counts run 8 waves (two mel halves): two waves per SIMD keep the f64 matrix pipe fed while the
    // partner waits on LDS
    switch (p.NTP / 16) {
        case 1: launch_frontend_nt<1, 1>(p, s); break; case 2: launch_frontend_nt<2, 2>(p, s); break;
        case 3: launch_frontend_nt<3, 1>(p, s); break; case 4: launch_frontend_nt<4, 2>(p, s); break;
        case 5: launch_frontend_nt<5, 1>(p, s); break; case 6: launch_frontend_nt<6, 2>(p, s); break;
        case 7: launch_frontend_nt<7, 1>(p, s); break; case 8: launch_frontend_nt<8, 2>(p, s); break;
        default: break;
    }
}

// ------------------------------------------------------------------------------------------ direct conv (stem)
// thread = (output pixel, group of 4 output channels); weights [kh][kw][Cin][Cout].
__global__ __launch_bounds__(256) void k_conv_direct(ConvParams p) {
    const int C4 = p.Cout >> 2;
    size_t total = (size_t)p.B * p.Ho * p.Wo * C4;
    size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    int c4 = (int)(idx % C4);
    size_t pix = idx / C4;
    int wo = (int)(pix % p.Wo);
    int ho = (int)((pix / p.Wo) % p.Ho);
    int b = (int)(pix / ((size_t)p.Wo * p.Ho));
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* w4 = reinterpret_cast<const float4*>(p.w);
    for (int i = 0; i < p.kh; i++) {
        int hi = ho * p.sh - p.pt + i;
        if (hi < 0 || hi >= p.H) continue;
        for (int j = 0; j < p.kw; j++) {
            int wi = wo * p.sw - p.pl + j;
            if (wi < 0 || wi >= p.W) continue;
            const float* ip = p.in + (((size_t)b * p.H + hi) * p.W + wi) * p.Cin;
            const float4* wp = w4 + (size_t)((i * p.kw + j) * p.Cin) * C4 + c4;
            for (int ci = 0; ci < p.Cin; ci++) {
                float x = ip[ci];
                float4 w = wp[(size_t)ci * C4];
                acc.x = fmaf(x, w.x, acc.x); acc.y = fmaf(x, w.y, acc.y);
                acc.z = fmaf(x, w.z, acc.z); acc.w = fmaf(x, w.w, acc.w);
            }
        }
    }
    if (p.bias) {
        float4 bv = reinterpret_cast<const float4*>(p.bias)[c4];
        acc.x += bv.x; acc.y += bv.y; acc.z += bv.z; acc.w += bv.w;
    }
    acc.x = apply_act(acc.x, p.act); acc.y = apply_act(acc.y, p.act);
    acc.z = apply_act(acc.z, p.act); acc.w = apply_act(acc.w, p.act);
    reinterpret_cast<float4*>(p.out)[idx] = acc;
}
// Compile-time (KH, KW, CIN) variant: every tap load is issued up front (the generic loop above is a chain of
// dependent L1/L2 round trips: 18 serial loads per thread for the 3x3x2 stem made it latency-bound at 1 TB/s).
template <int KH, int KW, int CIN>
__global__ __launch_bounds__(256) void k_conv_direct_t(ConvParams p) {
    const int C4 = p.Cout >> 2;
    size_t total = (size_t)p.B * p.Ho * p.Wo * C4;
    size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    int c4 = (int)(idx % C4);
    size_t pix = idx / C4;
    int wo = (int)(pix % p.Wo);
    int ho = (int)((pix / p.Wo) % p.Ho);
    int b = (int)(pix / ((size_t)p.Wo * p.Ho));
    float x[KH][KW][CIN];
#pragma unroll
    for (int i = 0; i < KH; i++) {
        int hi = ho * p.sh - p.pt + i;
#pragma unroll
        for (int j = 0; j < KW; j++) {
            int wi = wo * p.sw - p.pl + j;
            bool ok = hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
            const float* ip = p.in + (((size_t)b * p.H + (ok ? hi : 0)) * p.W + (ok ? wi : 0)) * CIN;
#pragma unroll
            for (int ci = 0; ci < CIN; ci++) x[i][j][ci] = ok ? ip[ci] : 0.f;
        }
    }
    float4 acc = p.bias ? reinterpret_cast<const float4*>(p.bias)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 a2 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* w4 = reinterpret_cast<const float4*>(p.w) + c4;
#pragma unroll
    for (int i = 0; i < KH; i++)
#pragma unroll
        for (int j = 0; j < KW; j++)
#pragma unroll
            for (int ci = 0; ci < CIN; ci++) {
                float4 w = w4[(size_t)((i * KW + j) * CIN + ci) * C4];
                float xv = x[i][j][ci];
                a2.x = fmaf(xv, w.x, a2.x); a2.y = fmaf(xv, w.y, a2.y); a2.z = fmaf(xv, w.z, a2.z); a2.w = fmaf(xv, w.w, a2.w);
            }
    // same association as the generic kernel: sum of products first, bias added last
    with_act(p.act, [&](auto f) {
        acc.x = f(a2.x + acc.x); acc.y = f(a2.y + acc.y); acc.z = f(a2.z + acc.z); acc.w = f(a2.w + acc.w);
    });
    reinterpret_cast<float4*>(p.out)[idx] = acc;
}
// PX consecutive output columns per thread (same 4 output channels): the weight quad is loaded once per tap for
// PX pixels and the overlapping input columns once per thread.  PMC on the one-pixel version of the 3x3x2 stem:
// 329 VALU instructions per thread for 72 FMAs - address arithmetic and 72 scalar loads dominated.
template <int KH, int KW, int CIN, int S, int PX>
__global__ __launch_bounds__(256) void k_conv_direct_px(ConvParams p, int wgroups) {
    constexpr int NCOL = (PX - 1) * S + KW;
    const int C4 = p.Cout >> 2;
    size_t total = (size_t)p.B * p.Ho * wgroups * C4;
    size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    int c4 = (int)(idx % C4);
    size_t g = idx / C4;
    int wg = (int)(g % wgroups);
    int ho = (int)((g / wgroups) % p.Ho);
    int b = (int)(g / ((size_t)wgroups * p.Ho));
    const int wo0 = wg * PX, wi0 = wo0 * S - p.pl;
    float x[KH][NCOL][CIN];
#pragma unroll
    for (int i = 0; i < KH; i++) {
        int hi = ho * S - p.pt + i;
        bool rok = hi >= 0 && hi < p.H;
        const float* rp = p.in + ((size_t)b * p.H + (rok ? hi : 0)) * p.W * CIN;
#pragma unroll
        for (int c = 0; c < NCOL; c++) {
            int wi = wi0 + c;
            bool ok = rok && wi >= 0 && wi < p.W;
            const float* ip = rp + (size_t)(ok ? wi : 0) * CIN;
#pragma unroll
            for (int ci = 0; ci < CIN; ci++) x[i][c][ci] = ok ? ip[ci] : 0.f;
        }
    }
    const float4 bv = p.bias ? reinterpret_cast<const float4*>(p.bias)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 a2[PX];
#pragma unroll
    for (int q = 0; q < PX; q++) a2[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* w4 = reinterpret_cast<const float4*>(p.w) + c4;
#pragma unroll
    for (int i = 0; i < KH; i++)
#pragma unroll
        for (int j = 0; j < KW; j++)
#pragma unroll
            for (int ci = 0; ci < CIN; ci++) {
                float4 w = w4[(size_t)((i * KW + j) * CIN + ci) * C4];
#pragma unroll
                for (int q = 0; q < PX; q++) {
                    float xv = x[i][q * S + j][ci];
                    a2[q].x = fmaf(xv, w.x, a2[q].x); a2[q].y = fmaf(xv, w.y, a2[q].y);
                    a2[q].z = fmaf(xv, w.z, a2[q].z); a2[q].w = fmaf(xv, w.w, a2[q].w);
                }
            }
    // same association as the generic kernel: sum of products first, bias added last
    with_act(p.act, [&](auto f) {
#pragma unroll
        for (int q = 0; q < PX; q++) {
            a2[q].x = f(a2[q].x + bv.x); a2[q].y = f(a2[q].y + bv.y); a2[q].z = f(a2[q].z + bv.z); a2[q].w = f(a2[q].w + bv.w);
        }
    });
    const size_t o0 = (((size_t)b * p.Ho + ho) * p.Wo + wo0) * C4 + c4;
    float4* op = reinterpret_cast<float4*>(p.out) + o0;
#pragma unroll
    for (int q = 0; q < PX; q++)
        if (wo0 + q < p.Wo) {
            if (p.out_bf16) bf16x4_store(p.out, o0 + (size_t)q * C4, a2[q]);
            else op[(size_t)q * C4] = a2[q];
        }
}
// (bf16 activation storage: only the 4-pixel kernels below write bf16)
bool conv_direct_bf16_ok(const ConvParams& p) {
    return p.kh == 3 && p.kw == 3 && (p.Cin == 1 || p.Cin == 2) && p.sh == 2 && p.sw == 2 && (p.Cout & 3) == 0;
}
void launch_conv_direct(const ConvParams& p, hipStream_t s) {
    if (p.kh == 3 && p.kw == 3 && p.Cin == 2 && p.sh == 2 && p.sw == 2 && (p.Cout & 3) == 0) {
        constexpr int PX = 4;
        int wgroups = (p.Wo + PX - 1) / PX;
        size_t tot = (size_t)p.B * p.Ho * wgroups * (p.Cout >> 2);
        hipLaunchKernelGGL((k_conv_direct_px<3, 3, 2, 2, PX>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, p, wgroups);
        return;
    }
    if (p.kh == 3 && p.kw == 3 && p.Cin == 1 && p.sh == 2 && p.sw == 2 && (p.Cout & 3) == 0) {      // one-channel image (log-mel stem)
        constexpr int PX = 4;
        int wgroups = (p.Wo + PX - 1) / PX;
        size_t tot = (size_t)p.B * p.Ho * wgroups * (p.Cout >> 2);
        hipLaunchKernelGGL((k_conv_direct_px<3, 3, 1, 2, PX>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, p, wgroups);
        return;
    }
    size_t total = (size_t)p.B * p.Ho * p.Wo * (p.Cout >> 2);
    dim3 grid((unsigned)((total + 255) / 256));
    if (p.kh == 3 && p.kw == 3 && p.Cin == 2) hipLaunchKernelGGL((k_conv_direct_t<3, 3, 2>), grid, dim3(256), 0, s, p);
    else if (p.kh == 3 && p.kw == 3 && p.Cin == 1) hipLaunchKernelGGL((k_conv_direct_t<3, 3, 1>), grid, dim3(256), 0, s, p);
    else if (p.kh == 3 && p.kw == 3 && p.Cin == 3) hipLaunchKernelGGL((k_conv_direct_t<3, 3, 3>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(k_conv_direct, grid, dim3(256), 0, s, p);
}

// Stem as an implicit GEMM on the f32 MFMA (3x3 stride-2 conv, Cin = 2): out[pixel][n] = sum_kk A[pixel][kk] W[n][kk]
// with kk = i*8 + jj*2 + ch over a 3 x 4 x 2 window (the fourth column is a zero-weight pad, so a k-group of 4 is two
// adjacent input pixels x 2 channels = 4 contiguous floats).  K = 24 -> two 16-wide slabs in the fragment order of
// k_expand_dw (lane kq of slab s holds k = 16 s + 4 kq .. +3).  The VALU version above spends 288 FMAs + addressing per
// 4 pixels x 4 channels (58 % VALU-busy at 188 us); here the FMAs move to the matrix pipe.
struct StemParams {
    const float* in; const float* wm /*[Cout][32]*/; const float* bias /*[Cout], zeros if absent*/; float* out;
    int B, H, W, Ho, Wo, Cout, pt, pl, act;
    unsigned total_px, tiles_per_wave;
};
template <int NTILES>
__global__ __launch_bounds__(256) void k_stem_mfma(StemParams p) {
    constexpr int CS = NTILES * 16 + 4;                          // staging row stride (floats)
    __shared__ __attribute__((aligned(16))) float stage[4 * 16 * CS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, kq = lane >> 4;
    // weight fragments + bias of this lane (constant for the block)
    f32x4 wf[2][NTILES];
    float4 bq[NTILES];
#pragma unroll
    for (int t = 0; t < NTILES; t++) {
#pragma unroll
        for (int sl = 0; sl < 2; sl++) {
            float4 w = *reinterpret_cast<const float4*>(p.wm + (size_t)(16 * t + li) * 32 + 16 * sl + 4 * kq);
            wf[sl][t] = (f32x4){w.x, w.y, w.z, w.w};
        }
        bq[t] = *reinterpret_cast<const float4*>(p.bias + 16 * t + 4 * kq);
    }
    // window position of this lane's k-groups: slab 0 -> row i = kq >> 1, slab 1 -> row 2 (kq >= 2: zero weights)
    const int i0 = kq >> 1, j0 = (kq & 1) * 2;
    const unsigned tile0 = (blockIdx.x * 4u + wave) * p.tiles_per_wave;
    const unsigned hw = (unsigned)p.Ho * p.Wo;
    // pixel coordinates of this lane: decoded once, then advanced by 16 columns per tile (carry into row / clip)
    unsigned px = tile0 * 16u + li;
    int b, oh, ow;
    {
        const unsigned pc = min(px, p.total_px - 1);
        b = pc / hw;
        const unsigned rem = pc - (unsigned)b * hw;
        oh = rem / p.Wo; ow = rem - oh * p.Wo;
    }
    for (unsigned tt = 0; tt < p.tiles_per_wave; tt++, px += 16u) {
        if ((tile0 + tt) * 16u >= p.total_px) break;                    // wave-uniform
        if (tt) {
            ow += 16;
            while (ow >= p.Wo) { ow -= p.Wo; if (++oh == p.Ho) { oh = 0; b++; } }
            if (b >= p.B) { b = p.B - 1; oh = p.Ho - 1; ow = p.Wo - 1; }  // lanes past the end: any valid pixel
        }
        const float* xb = p.in + (size_t)b * p.H * p.W * 2;
        f32x4 xf[2];
        const int r0 = oh * 2 - p.pt, c0 = ow * 2 - p.pl + j0;
        // interior pixels (all but the image border) need neither clamps nor masks: one wave-uniform test
        const bool inner = r0 >= 0 && r0 + 2 < p.H && c0 >= 0 && c0 + 1 < p.W;
        if (__builtin_amdgcn_ballot_w64(!inner) == 0) {
#pragma unroll
            for (int sl = 0; sl < 2; sl++) {
                const float* q = xb + ((size_t)(r0 + (sl == 0 ? i0 : 2)) * p.W + c0) * 2;
                const float2 a = *reinterpret_cast<const float2*>(q), c = *reinterpret_cast<const float2*>(q + 2);
                xf[sl] = (f32x4){a.x, a.y, c.x, c.y};
            }
        } else {
#pragma unroll
            for (int sl = 0; sl < 2; sl++) {
                const int row = r0 + (sl == 0 ? i0 : 2), col = c0;
                const bool rv = row >= 0 && row < p.H;
                const bool v0 = rv && col >= 0 && col < p.W, v1 = rv && col + 1 >= 0 && col + 1 < p.W;
                const int rc = min(max(row, 0), p.H - 1);
                const float2 a = *reinterpret_cast<const float2*>(xb + ((size_t)rc * p.W + min(max(col, 0), p.W - 1)) * 2);
                const float2 c = *reinterpret_cast<const float2*>(xb + ((size_t)rc * p.W + min(max(col + 1, 0), p.W - 1)) * 2);
                xf[sl] = (f32x4){v0 ? a.x : 0.f, v0 ? a.y : 0.f, v1 ? c.x : 0.f, v1 ? c.y : 0.f};
            }
        }
        f32x4 acc[NTILES];
#pragma unroll
        for (int t = 0; t < NTILES; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sl = 0; sl < 2; sl++)
#pragma unroll
            for (int sidx = 0; sidx < 4; sidx++)
#pragma unroll
                for (int t = 0; t < NTILES; t++)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[sl][t][sidx], xf[sl][sidx], acc[t], 0, 0, 0);
        with_act(p.act, [&](auto f) {
#pragma unroll
            for (int t = 0; t < NTILES; t++) {
                acc[t][0] = f(acc[t][0] + bq[t].x); acc[t][1] = f(acc[t][1] + bq[t].y);
                acc[t][2] = f(acc[t][2] + bq[t].z); acc[t][3] = f(acc[t][3] + bq[t].w);
            }
        });
        // the lane holds 4 channels of one pixel per n-tile: stage the 16 x Cout tile through this wave's LDS slice and write
        // it out as one contiguous run (16 pixels x Cout floats) instead of 64-byte pieces
        float* stg = stage + wave * (16 * CS);
#pragma unroll
        for (int t = 0; t < NTILES; t++) *reinterpret_cast<f32x4*>(&stg[li * CS + 16 * t + 4 * kq]) = acc[t];
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        const unsigned px0 = (tile0 + tt) * 16u;
#pragma unroll
        for (int q = 0; q < (16 * NTILES * 4) / 64; q++) {
            const int idx = lane + 64 * q;                       // float4 index inside the 16 x Cout tile
            const int row = idx / (NTILES * 4), c4 = idx % (NTILES * 4);
            if (px0 + row < p.total_px)
                *reinterpret_cast<f32x4*>(p.out + (size_t)(px0 + row) * p.Cout + 4 * c4) = *reinterpret_cast<const f32x4*>(&stg[row * CS + 4 * c4]);
        }
        __builtin_amdgcn_wave_barrier();
    }
}
bool stem_mfma_supported(const ConvParams& p) {
    return p.kh == 3 && p.kw == 3 && p.sh == 2 && p.sw == 2 && p.Cin == 2 && (p.Cout == 32 || p.Cout == 64);
}
void launch_stem_mfma(const ConvParams& c, const float* wm, const float* bias_p, hipStream_t s) {
    static const int stem_tpw = getenv("BNHIP_STEM_TPW") ? std::max(atoi(getenv("BNHIP_STEM_TPW")), 1) : 8;   // tiles per wave
    StemParams p{c.in, wm, bias_p, c.out, c.B, c.H, c.W, c.Ho, c.Wo, c.Cout, c.pt, c.pl, c.act,
                 (unsigned)((size_t)c.B * c.Ho * c.Wo), (unsigned)stem_tpw};
    unsigned tiles = (p.total_px + 15) / 16;
    unsigned blocks = (tiles + 4 * p.tiles_per_wave - 1) / (4 * p.tiles_per_wave);
    if (c.Cout == 32) hipLaunchKernelGGL((k_stem_mfma<2>), dim3(blocks), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((k_stem_mfma<4>), dim3(blocks), dim3(256), 0, s, p);
}

// ------------------------------------------------------------------------------------------ pointwise GEMM
// out[M,N] = act(A[M,K] * W[N,K]^T + bias[N]) (+ res[M,N]); optional per-(batch,k) scale on A (squeeze-excite
// MUL folded into the consumer's operand load).  f32 MFMA 16x16x4 with the roles swapped (W rows feed
// the MFMA "A" side, activation rows the "B" side) so each lane ends up holding 4 consecutive output
// channels of one row -> one 16-byte store per tile.
// Block 128 rows x (16*NT) cols, 4 waves, wave w owns rows [32w,32w+32).  BK = 32, LDS row stride 40 floats:
// with ds_read_b128 fragment loads the (row*10 + kq) 16-byte slot pattern is conflict-free for every
// 16-lane service group.  K is consumed in permuted order inside each 16-wide slab (lane kq holds
// k = 4kq..4kq+3, MFMA step s pairs element s of both operands) - a fixed reordering of the fp32 sum.
// PW_TRACE (tools/ubench/pw_trace.hip only): lane 0 of every wave of the first 64 logical blocks stamps the shader clock at
// the phase boundaries of each K slab, to see where a wave's time goes.  Compiled out of the library.
#ifdef PW_TRACE
__device__ long long* g_pw_trace = nullptr;      // [64 blocks][4 waves][PW_TRACE_SLOTS]
#define PW_TRACE_SLOTS 128
#define PW_T(i) do { if (lane == 0 && L < 64u && (i) < PW_TRACE_SLOTS && !((i) >= 60 && (i) < 64)) g_pw_trace[((size_t)L * 4 + wave) * PW_TRACE_SLOTS + (i)] = clock64(); } while (0)
#else
#define PW_T(i) do { } while (0)
#endif
// IM (implicit GEMM): the same kernel as a general convolution.  A is never materialised: row m is output pixel
// (b, oh, ow), column k = (i * kw + j) * Cin + ci is input value x[b][oh s - pt + i d][ow s - pl + j d][ci] (zero outside the
// image), and the file's OHWI weights are already the [N][K] matrix.  Cin % 4 == 0, so a float4 of K never straddles a tap.
struct ImGeo {
    int H, W, Cin, kw, sh, sw, dh, dw, pt, pl, Wo;
    FDiv d_cin, d_kw, d_wo, d_howo;      // k -> tap, tap -> row, pixel -> (oh, ow), m -> clip
};
template <int NT, bool SC, int WM, bool IM = false>
__global__ __launch_bounds__(256) void k_pw_gemm(PwParams p, int nblk_n, unsigned nblk, FDiv dn, FDiv dhw, ImGeo g) {
    constexpr int BM = 64 * WM;                  // rows per block: 4 waves x (16*WM) rows
    constexpr int XQ = BM * PW_C4 / 256;         // float4 per thread for the activation tile
    // operand tiles (the epilogue re-uses the array as per-wave output staging).  SC: squeeze-excite scale on A.
    constexpr bool DB = false;   // double-buffering measured neutral on this chip for these shapes; kept for experiments
    constexpr int TILE = (BM + NT * 16) * PW_LS;
    // the epilogue stages 4 waves x 16 rows x (16 NT + 4) floats through the same array: with 64-row tiles and NT >= 7
    // that is MORE than the operand tile (found by bench.py's batch-vs-small-batch check when the work-based tuner first
    // picked <7, *, 1>: rows of neighbouring waves overwrote each other)
    constexpr int STG = 4 * 16 * (NT * 16 + 4);
    constexpr int LDSN = (DB ? 2 : 1) * TILE > STG ? (DB ? 2 : 1) * TILE : STG;
    __shared__ __attribute__((aligned(16))) float lds[LDSN];
    constexpr int WQ = (NT * 16 * PW_C4 + 255) / 256;   // float4 per thread for the W tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    // XCD-aware order: N-blocks fastest so the blocks that share an activation tile sit on one XCD's L2
    const unsigned L = xcd_remap(blockIdx.x, nblk);
    const int mblk = (int)fdiv(L, dn);
    const int m0 = mblk * BM;
    const int n0 = ((int)L - mblk * nblk_n) * (NT * 16);
    const int K = p.K;

    float4 xreg[XQ], wreg[WQ], sreg[SC ? XQ : 1];
    int srow[SC ? XQ : 1];                        // batch index of each staged row (for the per-(batch,k) scale)
    if (SC) {
#pragma unroll
        for (int q = 0; q < XQ; q++) {
            int m = m0 + ((tid + 256 * q) / PW_C4);
            srow[SC ? q : 0] = (int)fdiv((unsigned)(m < p.M ? m : 0), dhw);
        }
    }
    // IM: top-left input coordinate and image base of each staged row (fixed for the block)
    int ih0[IM ? XQ : 1], iw0[IM ? XQ : 1]; unsigned ibase[IM ? XQ : 1];
    if (IM) {
#pragma unroll
        for (int q = 0; q < XQ; q++) {
            const unsigned m = (unsigned)min(m0 + (tid + 256 * q) / PW_C4, p.M - 1);
            const unsigned b = fdiv(m, g.d_howo), pix = m - b * g.d_howo.d;
            const unsigned oh = fdiv(pix, g.d_wo), ow = pix - oh * (unsigned)g.Wo;
            ih0[IM ? q : 0] = (int)oh * g.sh - g.pt; iw0[IM ? q : 0] = (int)ow * g.sw - g.pl;
            ibase[IM ? q : 0] = b * (unsigned)(g.H * g.W);
        }
    }
    // all global loads of a slab are issued back-to-back (scale included); the multiply happens at LDS-store time
    auto gload = [&](int k0) {
#pragma unroll
        for (int q = 0; q < XQ; q++) {
            int idx = tid + 256 * q;
            int row = idx / PW_C4, c4 = idx % PW_C4;
            int m = m0 + row, k = k0 + 4 * c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f), sc = make_float4(0.f, 0.f, 0.f, 0.f);
            if (IM) {
                if (m < p.M && k < K) {
                    const unsigned tap = fdiv((unsigned)k, g.d_cin), ci = (unsigned)k - tap * (unsigned)g.Cin;
                    const unsigned ti = fdiv(tap, g.d_kw), tj = tap - ti * (unsigned)g.kw;
                    const int ih = ih0[IM ? q : 0] + (int)ti * g.dh, iw = iw0[IM ? q : 0] + (int)tj * g.dw;
                    if (ih >= 0 && ih < g.H && iw >= 0 && iw < g.W)
                        v = *reinterpret_cast<const float4*>(p.A + ((size_t)ibase[IM ? q : 0] + (size_t)ih * g.W + iw) * g.Cin + ci);
                }
            } else if (m < p.M && k < K) {
                v = *reinterpret_cast<const float4*>(p.A + (size_t)m * K + k);
                if (SC) sc = *reinterpret_cast<const float4*>(p.ascale + (size_t)srow[SC ? q : 0] * K + k);
            }
            xreg[q] = v;
            if (SC) sreg[SC ? q : 0] = sc;
        }
#pragma unroll
        for (int q = 0; q < WQ; q++) {
            int idx = tid + 256 * q;
            int row = idx / PW_C4, c4 = idx % PW_C4;
            int n = n0 + row, k = k0 + 4 * c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < NT * 16 && n < p.N && k < K) v = *reinterpret_cast<const float4*>(p.W + (size_t)n * K + k);
            wreg[q] = v;
        }
    };
    auto lstore = [&](int buf) {
        float* Xs = lds + buf * TILE;
        float* Ws = Xs + BM * PW_LS;
#pragma unroll
        for (int q = 0; q < XQ; q++) {
            int idx = tid + 256 * q;
            float4 v = xreg[q];
            if (SC) { float4 sc = sreg[SC ? q : 0]; v.x *= sc.x; v.y *= sc.y; v.z *= sc.z; v.w *= sc.w; }
            *reinterpret_cast<float4*>(&Xs[(idx / PW_C4) * PW_LS + 4 * (idx % PW_C4)]) = v;
        }
#pragma unroll
        for (int q = 0; q < WQ; q++) {
            int idx = tid + 256 * q;
            if ((idx / PW_C4) < NT * 16) *reinterpret_cast<float4*>(&Ws[(idx / PW_C4) * PW_LS + 4 * (idx % PW_C4)]) = wreg[q];
        }
    };

    f32x4 acc[NT][WM];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int mt = 0; mt < WM; mt++) acc[t][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // software pipeline: slab s computes from LDS buffer s&1 while slab s+1 moves registers -> the other
    // buffer and slab s+2's global loads are in flight; one barrier per slab.
    const int nslab = (K + PW_BK - 1) / PW_BK;
    PW_T(0);
    gload(0);
    lstore(0);
    if (nslab > 1) gload(PW_BK);
    __syncthreads();
    PW_T(1);
    for (int sl = 0; sl < nslab; sl++) {
        PW_T(4 + 4 * sl);
        const float* Xs = lds + (DB ? (sl & 1) : 0) * TILE;
        const float* Ws = Xs + BM * PW_LS;
#pragma unroll
        for (int t16 = 0; t16 < PW_BK / 16; t16++) {
            f32x4 xf[WM], wf[NT];
#pragma unroll
            for (int mt = 0; mt < WM; mt++)
                xf[mt] = *reinterpret_cast<const f32x4*>(&Xs[(16 * WM * wave + 16 * mt + li) * PW_LS + 16 * t16 + 4 * kq]);
#pragma unroll
            for (int t = 0; t < NT; t++)
                wf[t] = *reinterpret_cast<const f32x4*>(&Ws[(16 * t + li) * PW_LS + 16 * t16 + 4 * kq]);
#pragma unroll
            for (int sidx = 0; sidx < 4; sidx++) {
#pragma unroll
                for (int t = 0; t < NT; t++)
#pragma unroll
                    for (int mt = 0; mt < WM; mt++)
                        acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t][sidx], xf[mt][sidx], acc[t][mt], 0, 0, 0);
            }
        }
        PW_T(5 + 4 * sl);                      // MFMAs issued
        if (sl + 1 < nslab) {
            if (!DB) __syncthreads();          // single buffer: everyone must be done reading it
            PW_T(6 + 4 * sl);                  // all waves done with the slab
#ifdef PW_TRACE
            __builtin_amdgcn_s_waitcnt(0x0f70);    // vmcnt(0): the next slab's global loads have landed
            PW_T(64 + 2 * sl);
#endif
            lstore(DB ? ((sl + 1) & 1) : 0);
#ifdef PW_TRACE
            __builtin_amdgcn_s_waitcnt(0xc07f);    // lgkmcnt(0): LDS stores done
            PW_T(65 + 2 * sl);
#endif
            if (sl + 2 < nslab) gload((sl + 2) * PW_BK);
            PW_T(7 + 4 * sl);                  // next slab stored (its global loads had landed), slab + 2 requested
        }
        __syncthreads();
    }
    PW_T(2);

    pw_epilogue<NT, WM>(p, acc, lds, m0, n0);
    PW_T(3);
}

// Software-pipelined variant for K % PW_BK == 0.  The phase trace of k_pw_gemm (tools/ubench/pw_trace.hip) shows a wave
// spending only ~1/3 of a slab period issuing MFMAs: the rest is two barrier waits and the refill of the single operand
// buffer, whose address VALU, LDS stores and global-load issue crawl because they compete with the other waves' MFMAs
// for issue slots.  Here the refill rides in the wave's OWN MFMA shadow instead: two LDS operand buffers, the stores of
// slab s+1 are interleaved with the MFMAs of the first half of slab s and the global loads of slab s+2 with those of the
// second half (sched_group_barrier), one barrier per slab.  All per-slab address arithmetic is gone: loads use per-thread
// offsets computed once (rows clamped into range: out-of-range rows produce values the epilogue never stores) plus k0.
template <int NT, bool SC, int WM>
__global__ __launch_bounds__(256) void k_pw_pipe(PwParams p, int nblk_n, unsigned nblk, FDiv dn, FDiv dhw) {
    constexpr int BM = 64 * WM;
    constexpr int XQ = BM * PW_C4 / 256, WQ = (NT * 16 * PW_C4 + 255) / 256;
    constexpr int TILE = (BM + NT * 16) * PW_LS;
    __shared__ __attribute__((aligned(16))) float lds[2 * TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const unsigned L = xcd_remap(blockIdx.x, nblk);
    const int mblk = (int)fdiv(L, dn);
    const int m0 = mblk * BM;
    const int n0 = ((int)L - mblk * nblk_n) * (NT * 16);
    const int K = p.K;

    unsigned xoff[XQ], soff[SC ? XQ : 1], woff[WQ];
    // LDS store slots: thread idx = tid + 256 q -> row idx / PW_C4, k-quad idx % PW_C4 (256 / PW_C4 rows further per q)
    const int lbase = (tid / PW_C4) * PW_LS + 4 * (tid % PW_C4);
    constexpr int LQ = (256 / PW_C4) * PW_LS;
#pragma unroll
    for (int q = 0; q < XQ; q++) {
        const int idx = tid + 256 * q, row = idx / PW_C4, c4 = idx % PW_C4;
        const int m = min(m0 + row, p.M - 1);
        xoff[q] = (unsigned)m * (unsigned)K + 4 * c4;
        if (SC) soff[SC ? q : 0] = fdiv((unsigned)m, dhw) * (unsigned)K + 4 * c4;
    }
#pragma unroll
    for (int q = 0; q < WQ; q++) {
        const int idx = tid + 256 * q, row = min(idx / PW_C4, NT * 16 - 1), c4 = idx % PW_C4;
        woff[q] = (unsigned)min(n0 + row, p.N - 1) * (unsigned)K + 4 * c4;
    }
    float4 xreg[XQ], wreg[WQ], sreg[SC ? XQ : 1];
    auto gload = [&](int k0) {
        const float* Ak = p.A + k0;
        const float* Wk = p.W + k0;
#pragma unroll
        for (int q = 0; q < XQ; q++) {
            xreg[q] = *reinterpret_cast<const float4*>(Ak + xoff[q]);
            if (SC) sreg[SC ? q : 0] = *reinterpret_cast<const float4*>(p.ascale + k0 + soff[SC ? q : 0]);
        }
#pragma unroll
        for (int q = 0; q < WQ; q++) wreg[q] = *reinterpret_cast<const float4*>(Wk + woff[q]);
    };
    auto lstore = [&](float* buf) {
#pragma unroll
        for (int q = 0; q < XQ; q++) {
            float4 v = xreg[q];
            if (SC) { const float4 sc = sreg[SC ? q : 0]; v.x *= sc.x; v.y *= sc.y; v.z *= sc.z; v.w *= sc.w; }
            *reinterpret_cast<float4*>(&buf[lbase + q * LQ]) = v;
        }
#pragma unroll
        for (int q = 0; q < WQ; q++)
            if ((NT * 16 * PW_C4) % 256 == 0 || (tid + 256 * q) / PW_C4 < NT * 16)
                *reinterpret_cast<float4*>(&buf[BM * PW_LS + lbase + q * LQ]) = wreg[q];
    };

    f32x4 acc[NT][WM];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int mt = 0; mt < WM; mt++) acc[t][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nslab = K / PW_BK;
    gload(0);
    lstore(lds);
    if (nslab > 1) gload(PW_BK);
    __syncthreads();
    // one slab: DS = also store slab sl + 1 (held in registers) into the other buffer, DL = also request slab sl + 2
    auto slab = [&](int sl, auto DS, auto DL) {
        const float* Xs = lds + (sl & 1) * TILE;
        const float* Ws = Xs + BM * PW_LS;
        float* nxt = lds + ((sl + 1) & 1) * TILE;
#pragma unroll
        for (int t16 = 0; t16 < PW_BK / 16; t16++) {
            f32x4 xf[WM], wf[NT];
#pragma unroll
            for (int mt = 0; mt < WM; mt++)
                xf[mt] = *reinterpret_cast<const f32x4*>(&Xs[(16 * WM * wave + 16 * mt + li) * PW_LS + 16 * t16 + 4 * kq]);
#pragma unroll
            for (int t = 0; t < NT; t++)
                wf[t] = *reinterpret_cast<const f32x4*>(&Ws[(16 * t + li) * PW_LS + 16 * t16 + 4 * kq]);
            if (t16 == 0 && decltype(DS)::value) lstore(nxt);
            if (t16 == PW_BK / 16 - 1 && decltype(DL)::value) gload((sl + 2) * PW_BK);
#pragma unroll
            for (int sidx = 0; sidx < 4; sidx++) {
#pragma unroll
                for (int t = 0; t < NT; t++)
#pragma unroll
                    for (int mt = 0; mt < WM; mt++)
                        acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t][sidx], xf[mt][sidx], acc[t][mt], 0, 0, 0);
            }
            // interleave: fragment reads first, then the refill instructions spread between the MFMAs
            constexpr int NMF = 4 * NT * WM, NREF = XQ + WQ, STEP = NMF / (NREF + 1) > 0 ? NMF / (NREF + 1) : 1;
            __builtin_amdgcn_sched_group_barrier(0x100, WM + NT, 0);
            if (t16 == 0 && decltype(DS)::value) {
#pragma unroll
                for (int r = 0; r < NREF; r++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, STEP, 0);
                    if (SC) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
            }
            if (t16 == PW_BK / 16 - 1 && decltype(DL)::value) {
#pragma unroll
                for (int r = 0; r < NREF + (SC ? XQ : 0); r++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, SC ? 1 : STEP, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
            }
        }
        __syncthreads();
    };
    int sl = 0;
    for (; sl + 2 < nslab; sl++) slab(sl, std::true_type{}, std::true_type{});
    if (nslab >= 2) { slab(sl, std::true_type{}, std::false_type{}); sl++; }
    slab(sl, std::false_type{}, std::false_type{});

    pw_epilogue<NT, WM>(p, acc, lds, m0, n0);
}

// ------------------------------------------------------------------------------------------ split-bf16 pointwise GEMM
// The same GEMM on the bf16 matrix pipe with fp32-equivalent products.  The f32-input MFMA runs at the vector rate (157 TF);
// v_mfma_f32_16x16x32_bf16 is 16x faster per MAC, and an fp32 value splits EXACTLY into three bf16 pieces by truncation:
//     x = hi + mid + lo,   hi = bf16(x), mid = bf16(x - hi), lo = x - hi - mid      (round to nearest even)
// (each subtraction is exact in fp32: the remainder of an 8-significant-bit rounding has at most 16 significant bits, the
// next one at most 8, so lo is a bf16 too; |mid| <= 2^-8 |x|, |lo| <= 2^-16 |x|).  Then
//     x * w = hi*whi + (hi*wmid + mid*whi) + (hi*wlo + mid*wmid + lo*whi)  +  [mid*wlo + lo*wmid + lo*wlo]
// where every bf16 x bf16 product is exact in the MFMA's fp32 accumulator and the bracketed terms are <= 2^-23 |x w|
// (the rounding of an fp32 product itself is <= 2^-24 |x w|): six products per k instead of one reproduce the fp32 product
// to within two units of its own rounding.  Accumulation stays fp32.  (|x| above the largest bf16, 3.39e38, is the one
// range the split cannot represent: hi overflows to infinity.)  The weights are split once at plan time (pw_bx3_image); activations are split in registers right after the
// fragment read - every activation row belongs to exactly one wave, so nothing is split twice, and the f32 operand tile in
// LDS (and the squeeze-excite multiply at store time) stays as in k_pw_gemm.
// K order: lane kq of a 32-wide slab holds k = 4kq..4kq+3 and 16+4kq..16+4kq+3 (the two conflict-free b128 slots of the
// f32 tile); the weight image uses the same order.
static int pick_nt(int M, int N);
template <int NT, bool SC, int WM>
__global__ __launch_bounds__(256) void k_pw_bx3(PwParams p, const uint16_t* __restrict__ Wimg, int Npad, int nblk_n, unsigned nblk,
                                                 FDiv dn, FDiv dhw) {
    constexpr int BM = 64 * WM;
    constexpr int XQ = BM * PW_C4 / 256;
    constexpr int WSLOTS = 12 * NT * 16;                   // 16-byte slots of the weight tile: [plane 3][kq 4][row NT*16]
    constexpr int WQ = (WSLOTS + 255) / 256;
    constexpr int TILE_F = BM * PW_LS + WSLOTS * 4;        // floats
    constexpr int STG = 4 * 16 * (NT * 16 + 4);
    constexpr int LDSN = TILE_F > STG ? TILE_F : STG;
    __shared__ __attribute__((aligned(16))) float lds[LDSN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const unsigned L = xcd_remap(blockIdx.x, nblk);
    const int mblk = (int)fdiv(L, dn);
    const int m0 = mblk * BM;
    const int n0 = ((int)L - mblk * nblk_n) * (NT * 16);
    const int K = p.K;

    unsigned xoff[XQ], soff[SC ? XQ : 1], woff[WQ];
    const bool one = p.prec == 1;                          // plain bf16: only the hi plane of the weights, one product
    const int wslots = one ? WSLOTS / 3 : WSLOTS;
    const int kc4 = 4 * (tid % PW_C4);                     // this thread's column inside a slab (the same for every q: 256 % PW_C4 == 0)
    const int lbase = (tid / PW_C4) * PW_LS + 4 * (tid % PW_C4);
    constexpr int LQ = (256 / PW_C4) * PW_LS;
#pragma unroll
    for (int q = 0; q < XQ; q++) {
        const int idx = tid + 256 * q, row = idx / PW_C4, c4 = idx % PW_C4;
        const int m = min(m0 + row, p.M - 1);
        xoff[q] = (unsigned)m * (unsigned)K + 4 * c4;
        if (SC) soff[SC ? q : 0] = fdiv((unsigned)m, dhw) * (unsigned)K + 4 * c4;
    }
#pragma unroll
    for (int q = 0; q < WQ; q++) {
        const int slot = min(tid + 256 * q, WSLOTS - 1);
        const int plkq = slot / (NT * 16), r = slot - plkq * (NT * 16);
        woff[q] = (unsigned)plkq * (unsigned)Npad + (unsigned)min(n0 + r, Npad - 1);          // in 16-byte units
    }
    float4 xreg[XQ], sreg[SC ? XQ : 1];
    u32x4 wreg[WQ];
    const u32x4* W16 = reinterpret_cast<const u32x4*>(Wimg);
    // bf16 activation storage (p.a_bf16, K % 8 == 0): a thread fetches 8 channels = the same 16 bytes per load instruction as
    // the fp32 path, in half as many instructions (4 threads per 32-wide row instead of 8; the first XQ / 2 entries of the
    // same offset / register arrays, so the fp32 path pays nothing for it).  The first form of this path kept the fp32 thread
    // mapping with 8-byte loads and made every projection 20-30 % SLOWER: these layers are bound by loads in flight per wave,
    // not by bytes.
    constexpr int XH = XQ / 2;
    constexpr int LQH = 64 * PW_LS;
    if (p.a_bf16) {
#pragma unroll
        for (int q = 0; q < XH; q++) {
            const int row = (tid >> 2) + 64 * q;
            const int m = min(m0 + row, p.M - 1);
            xoff[q] = (unsigned)m * (unsigned)K + 8u * (unsigned)(tid & 3);
            if (SC) soff[SC ? q : 0] = fdiv((unsigned)m, dhw) * (unsigned)K + 8u * (unsigned)(tid & 3);
        }
    }
    auto gload = [&](int sl) {
        const float* Ak = p.A + sl * PW_BK;
        const bool kin = sl * PW_BK + kc4 < K;             // K tail (K % 32 != 0): columns beyond K are zeros (their weights too)
        if (p.a_bf16) {
            const bool kinh = sl * PW_BK + 8 * (tid & 3) < K;
            const uint16_t* A16 = reinterpret_cast<const uint16_t*>(p.A) + sl * PW_BK;
#pragma unroll
            for (int q = 0; q < XH; q++) {
                xreg[q] = kinh ? *reinterpret_cast<const float4*>(A16 + xoff[q]) : make_float4(0.f, 0.f, 0.f, 0.f);   // (8 raw bf16)
                if (SC) {
                    const float* sp = p.ascale + sl * PW_BK + soff[SC ? q : 0];
                    sreg[SC ? 2 * q : 0] = kinh ? *reinterpret_cast<const float4*>(sp) : make_float4(0.f, 0.f, 0.f, 0.f);
                    sreg[SC ? 2 * q + 1 : 0] = kinh ? *reinterpret_cast<const float4*>(sp + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < XQ; q++) {
                xreg[q] = kin ? *reinterpret_cast<const float4*>(Ak + xoff[q]) : make_float4(0.f, 0.f, 0.f, 0.f);
                if (SC) sreg[SC ? q : 0] = kin ? *reinterpret_cast<const float4*>(p.ascale + sl * PW_BK + soff[SC ? q : 0]) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        const u32x4* Ws = W16 + (size_t)sl * 12 * Npad;
#pragma unroll
        for (int q = 0; q < WQ; q++)
            if (tid + 256 * q < wslots) wreg[q] = Ws[woff[q]];
    };
    u32x4* Wl = reinterpret_cast<u32x4*>(lds + BM * PW_LS);
    auto lstore = [&]() {
        if (p.a_bf16) {
#pragma unroll
            for (int q = 0; q < XH; q++) {
                const unsigned r[4] = {__float_as_uint(xreg[q].x), __float_as_uint(xreg[q].y), __float_as_uint(xreg[q].z), __float_as_uint(xreg[q].w)};
                float4 v0 = make_float4(__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xffff0000u), __uint_as_float(r[1] << 16), __uint_as_float(r[1] & 0xffff0000u));
                float4 v1 = make_float4(__uint_as_float(r[2] << 16), __uint_as_float(r[2] & 0xffff0000u), __uint_as_float(r[3] << 16), __uint_as_float(r[3] & 0xffff0000u));
                if (SC) {
                    const float4 s0 = sreg[SC ? 2 * q : 0], s1 = sreg[SC ? 2 * q + 1 : 0];
                    v0.x *= s0.x; v0.y *= s0.y; v0.z *= s0.z; v0.w *= s0.w; v1.x *= s1.x; v1.y *= s1.y; v1.z *= s1.z; v1.w *= s1.w;
                }
                *reinterpret_cast<float4*>(&lds[((tid >> 2) * PW_LS + 8 * (tid & 3)) + q * LQH]) = v0;
                *reinterpret_cast<float4*>(&lds[((tid >> 2) * PW_LS + 8 * (tid & 3)) + q * LQH + 4]) = v1;
            }
        } else {
#pragma unroll
        for (int q = 0; q < XQ; q++) {
            float4 v = xreg[q];
            if (SC) { const float4 sc = sreg[SC ? q : 0]; v.x *= sc.x; v.y *= sc.y; v.z *= sc.z; v.w *= sc.w; }
            *reinterpret_cast<float4*>(&lds[lbase + q * LQ]) = v;
        }
        }
#pragma unroll
        for (int q = 0; q < WQ; q++)
            if (tid + 256 * q < wslots) Wl[tid + 256 * q] = wreg[q];
    };

    f32x4 acc[NT][WM];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int mt = 0; mt < WM; mt++) acc[t][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nslab = (K + PW_BK - 1) / PW_BK;
    gload(0);
    lstore();
    if (nslab > 1) gload(1);
    __syncthreads();
    for (int sl = 0; sl < nslab; sl++) {
        bf16x8 ah[WM], am[WM], al[WM];
        if (one) {
#pragma unroll
            for (int mt = 0; mt < WM; mt++) {
                const float* xr = &lds[(16 * WM * wave + 16 * mt + li) * PW_LS + 4 * kq];
                ah[mt] = bx1_cvt8(*reinterpret_cast<const f32x4*>(xr), *reinterpret_cast<const f32x4*>(xr + 16));
            }
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const bf16x8 wh = __builtin_bit_cast(bf16x8, Wl[kq * (NT * 16) + 16 * t + li]);
#pragma unroll
                for (int mt = 0; mt < WM; mt++) acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, ah[mt], acc[t][mt], 0, 0, 0);
            }
        } else {
#pragma unroll
        for (int mt = 0; mt < WM; mt++) {
            const float* xr = &lds[(16 * WM * wave + 16 * mt + li) * PW_LS + 4 * kq];
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(xr), x1 = *reinterpret_cast<const f32x4*>(xr + 16);
            bx3_split8(x0, x1, &ah[mt], &am[mt], &al[mt]);
        }
        // weight fragments of tile t + 1 are requested before the MFMAs of tile t (the ISA of the first version waited a full
        // LDS round trip in front of every tile: with ~2 waves per SIMD on the late layers nothing else covered it)
        u32x4 wfr[2][3];
#pragma unroll
        for (int pl3 = 0; pl3 < 3; pl3++) wfr[0][pl3] = Wl[(pl3 * 4 + kq) * (NT * 16) + li];
#pragma unroll
        for (int t = 0; t < NT; t++) {
            if (t + 1 < NT) {
#pragma unroll
                for (int pl3 = 0; pl3 < 3; pl3++) wfr[(t + 1) & 1][pl3] = Wl[(pl3 * 4 + kq) * (NT * 16) + 16 * (t + 1) + li];
            }
            const bf16x8 wh = __builtin_bit_cast(bf16x8, wfr[t & 1][0]);
            const bf16x8 wm = __builtin_bit_cast(bf16x8, wfr[t & 1][1]);
            const bf16x8 wl = __builtin_bit_cast(bf16x8, wfr[t & 1][2]);
#pragma unroll
            for (int mt = 0; mt < WM; mt++) {
                f32x4 c = acc[t][mt];                        // smallest terms first
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, ah[mt], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, al[mt], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, am[mt], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, ah[mt], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, am[mt], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, ah[mt], c, 0, 0, 0);
                acc[t][mt] = c;
            }
        }
        }
        if (sl + 1 < nslab) {
            __syncthreads();                 // everyone is done reading the single operand buffer
            lstore();
            if (sl + 2 < nslab) gload(sl + 2);
        }
        __syncthreads();
    }
    pw_epilogue<NT, WM>(p, acc, lds, m0, n0);
}

// Software-pipelined form (the k_pw_pipe structure): the late layers have M = 12 288 rows at batch 256, i.e. only ~2 blocks
// per CU and 1-2 waves per SIMD, so nothing hides a wave's own global -> LDS refill; with the bf16 MFMA phase 2.5x shorter
// than the fp32 one that refill dominated the slab period of k_pw_bx3.  Here it rides in the wave's MFMA shadow: two LDS
// operand buffers (dynamic LDS: up to 80 KB), slab s+1 is stored while slab s computes, slab s+2 is in flight in registers,
// one barrier per slab.
template <int NT, bool SC, int WM>
__global__ __launch_bounds__(256) void k_pw_bx3p(PwParams p, const uint16_t* __restrict__ Wimg, int Npad, int nblk_n, unsigned nblk,
                                                  FDiv dn, FDiv dhw) {
    constexpr int BM = 64 * WM;
    constexpr int XQ = BM * PW_C4 / 256;
    constexpr int WSLOTS = 12 * NT * 16;
    constexpr int WQ = (WSLOTS + 255) / 256;
    constexpr int TILE_F = BM * PW_LS + WSLOTS * 4;
    extern __shared__ __attribute__((aligned(16))) float bx3p_lds[];  // 2 * TILE_F floats (>= the epilogue staging area)
    float* lds = bx3p_lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const unsigned L = xcd_remap(blockIdx.x, nblk);
    const int mblk = (int)fdiv(L, dn);
    const int m0 = mblk * BM;
    const int n0 = ((int)L - mblk * nblk_n) * (NT * 16);
    const int K = p.K;

    unsigned xoff[XQ], soff[SC ? XQ : 1], woff[WQ];
    const bool one = p.prec == 1;                          // plain bf16: only the hi plane of the weights, one product
    const int wslots = one ? WSLOTS / 3 : WSLOTS;
    const int kc4 = 4 * (tid % PW_C4);                     // this thread's column inside a slab (the same for every q: 256 % PW_C4 == 0)
    const int lbase = (tid / PW_C4) * PW_LS + 4 * (tid % PW_C4);
    constexpr int LQ = (256 / PW_C4) * PW_LS;
#pragma unroll
    for (int q = 0; q < XQ; q++) {
        const int idx = tid + 256 * q, row = idx / PW_C4, c4 = idx % PW_C4;
        const int m = min(m0 + row, p.M - 1);
        xoff[q] = (unsigned)m * (unsigned)K + 4 * c4;
        if (SC) soff[SC ? q : 0] = fdiv((unsigned)m, dhw) * (unsigned)K + 4 * c4;
    }
#pragma unroll
    for (int q = 0; q < WQ; q++) {
        const int slot = min(tid + 256 * q, WSLOTS - 1);
        const int plkq = slot / (NT * 16), r = slot - plkq * (NT * 16);
        woff[q] = (unsigned)plkq * (unsigned)Npad + (unsigned)min(n0 + r, Npad - 1);
    }
    float4 xreg[XQ], sreg[SC ? XQ : 1];
    u32x4 wreg[WQ];
    const u32x4* W16 = reinterpret_cast<const u32x4*>(Wimg);
    // bf16 activation storage (p.a_bf16, K % 8 == 0): a thread fetches 8 channels = the same 16 bytes per load instruction as
    // the fp32 path, in half as many instructions (4 threads per 32-wide row instead of 8; the first XQ / 2 entries of the
    // same offset / register arrays, so the fp32 path pays nothing for it).  The first form of this path kept the fp32 thread
    // mapping with 8-byte loads and made every projection 20-30 % SLOWER: these layers are bound by loads in flight per wave,
    // not by bytes.
    constexpr int XH = XQ / 2;
    constexpr int LQH = 64 * PW_LS;
    if (p.a_bf16) {
#pragma unroll
        for (int q = 0; q < XH; q++) {
            const int row = (tid >> 2) + 64 * q;
            const int m = min(m0 + row, p.M - 1);
            xoff[q] = (unsigned)m * (unsigned)K + 8u * (unsigned)(tid & 3);
            if (SC) soff[SC ? q : 0] = fdiv((unsigned)m, dhw) * (unsigned)K + 8u * (unsigned)(tid & 3);
        }
    }
    auto gload = [&](int sl) {
        const float* Ak = p.A + sl * PW_BK;
        const bool kin = sl * PW_BK + kc4 < K;             // K tail (K % 32 != 0): columns beyond K are zeros (their weights too)
        if (p.a_bf16) {
            const bool kinh = sl * PW_BK + 8 * (tid & 3) < K;
            const uint16_t* A16 = reinterpret_cast<const uint16_t*>(p.A) + sl * PW_BK;
#pragma unroll
            for (int q = 0; q < XH; q++) {
                xreg[q] = kinh ? *reinterpret_cast<const float4*>(A16 + xoff[q]) : make_float4(0.f, 0.f, 0.f, 0.f);   // (8 raw bf16)
                if (SC) {
                    const float* sp = p.ascale + sl * PW_BK + soff[SC ? q : 0];
                    sreg[SC ? 2 * q : 0] = kinh ? *reinterpret_cast<const float4*>(sp) : make_float4(0.f, 0.f, 0.f, 0.f);
                    sreg[SC ? 2 * q + 1 : 0] = kinh ? *reinterpret_cast<const float4*>(sp + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < XQ; q++) {
                xreg[q] = kin ? *reinterpret_cast<const float4*>(Ak + xoff[q]) : make_float4(0.f, 0.f, 0.f, 0.f);
                if (SC) sreg[SC ? q : 0] = kin ? *reinterpret_cast<const float4*>(p.ascale + sl * PW_BK + soff[SC ? q : 0]) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        const u32x4* Ws = W16 + (size_t)sl * 12 * Npad;
#pragma unroll
        for (int q = 0; q < WQ; q++)
            if (tid + 256 * q < wslots) wreg[q] = Ws[woff[q]];
    };
    auto lstore = [&](float* buf) {
        if (p.a_bf16) {
#pragma unroll
            for (int q = 0; q < XH; q++) {
                const unsigned r[4] = {__float_as_uint(xreg[q].x), __float_as_uint(xreg[q].y), __float_as_uint(xreg[q].z), __float_as_uint(xreg[q].w)};
                float4 v0 = make_float4(__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xffff0000u), __uint_as_float(r[1] << 16), __uint_as_float(r[1] & 0xffff0000u));
                float4 v1 = make_float4(__uint_as_float(r[2] << 16), __uint_as_float(r[2] & 0xffff0000u), __uint_as_float(r[3] << 16), __uint_as_float(r[3] & 0xffff0000u));
                if (SC) {
                    const float4 s0 = sreg[SC ? 2 * q : 0], s1 = sreg[SC ? 2 * q + 1 : 0];
                    v0.x *= s0.x; v0.y *= s0.y; v0.z *= s0.z; v0.w *= s0.w; v1.x *= s1.x; v1.y *= s1.y; v1.z *= s1.z; v1.w *= s1.w;
                }
                *reinterpret_cast<float4*>(&buf[((tid >> 2) * PW_LS + 8 * (tid & 3)) + q * LQH]) = v0;
                *reinterpret_cast<float4*>(&buf[((tid >> 2) * PW_LS + 8 * (tid & 3)) + q * LQH + 4]) = v1;
            }
        } else {
#pragma unroll
        for (int q = 0; q < XQ; q++) {
            float4 v = xreg[q];
            if (SC) { const float4 sc = sreg[SC ? q : 0]; v.x *= sc.x; v.y *= sc.y; v.z *= sc.z; v.w *= sc.w; }
            *reinterpret_cast<float4*>(&buf[lbase + q * LQ]) = v;
        }
        }
        u32x4* Wl = reinterpret_cast<u32x4*>(buf + BM * PW_LS);
#pragma unroll
        for (int q = 0; q < WQ; q++)
            if (tid + 256 * q < wslots) Wl[tid + 256 * q] = wreg[q];
    };

    f32x4 acc[NT][WM];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int mt = 0; mt < WM; mt++) acc[t][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nslab = (K + PW_BK - 1) / PW_BK;
    gload(0);
    lstore(lds);
    if (nslab > 1) gload(1);
    __syncthreads();
    auto slab = [&](int sl, auto DS, auto DL) {
        const float* Xs = lds + (sl & 1) * TILE_F;
        const u32x4* Wl = reinterpret_cast<const u32x4*>(Xs + BM * PW_LS);
        float* nxt = lds + ((sl + 1) & 1) * TILE_F;
        bf16x8 ah[WM], am[WM], al[WM];
        if (one) {
#pragma unroll
            for (int mt = 0; mt < WM; mt++) {
                const float* xr = &Xs[(16 * WM * wave + 16 * mt + li) * PW_LS + 4 * kq];
                ah[mt] = bx1_cvt8(*reinterpret_cast<const f32x4*>(xr), *reinterpret_cast<const f32x4*>(xr + 16));
            }
            bf16x8 wh1[NT];
#pragma unroll
            for (int t = 0; t < NT; t++) wh1[t] = __builtin_bit_cast(bf16x8, Wl[kq * (NT * 16) + 16 * t + li]);
            if (decltype(DS)::value) lstore(nxt);
            if (decltype(DL)::value) gload(sl + 2);
#pragma unroll
            for (int t = 0; t < NT; t++)
#pragma unroll
                for (int mt = 0; mt < WM; mt++) acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh1[t], ah[mt], acc[t][mt], 0, 0, 0);
            __syncthreads();
            return;
        }
#pragma unroll
        for (int mt = 0; mt < WM; mt++) {
            const float* xr = &Xs[(16 * WM * wave + 16 * mt + li) * PW_LS + 4 * kq];
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(xr), x1 = *reinterpret_cast<const f32x4*>(xr + 16);
            bx3_split8(x0, x1, &ah[mt], &am[mt], &al[mt]);
        }
        u32x4 wfr[2][3];
#pragma unroll
        for (int pl3 = 0; pl3 < 3; pl3++) wfr[0][pl3] = Wl[(pl3 * 4 + kq) * (NT * 16) + li];
#pragma unroll
        for (int t = 0; t < NT; t++) {
            if (t + 1 < NT) {
#pragma unroll
                for (int pl3 = 0; pl3 < 3; pl3++) wfr[(t + 1) & 1][pl3] = Wl[(pl3 * 4 + kq) * (NT * 16) + 16 * (t + 1) + li];
            }
            const bf16x8 wh = __builtin_bit_cast(bf16x8, wfr[t & 1][0]);
            const bf16x8 wm = __builtin_bit_cast(bf16x8, wfr[t & 1][1]);
            const bf16x8 wl = __builtin_bit_cast(bf16x8, wfr[t & 1][2]);
            if (t == 0 && decltype(DS)::value) lstore(nxt);
            if (t == NT - 1 && decltype(DL)::value) gload(sl + 2);
#pragma unroll
            for (int mt = 0; mt < WM; mt++) {
                f32x4 c = acc[t][mt];
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, ah[mt], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, al[mt], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, am[mt], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, ah[mt], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, am[mt], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, ah[mt], c, 0, 0, 0);
                acc[t][mt] = c;
            }
        }
        __syncthreads();
    };
    int sl = 0;
    for (; sl + 2 < nslab; sl++) slab(sl, std::true_type{}, std::true_type{});
    if (nslab >= 2) { slab(sl, std::true_type{}, std::false_type{}); sl++; }
    slab(sl, std::false_type{}, std::false_type{});

    pw_epilogue<NT, WM>(p, acc, lds, m0, n0);
}
static size_t bx3p_lds_bytes(int nt, int wm) { return 2 * ((size_t)64 * wm * PW_LS + (size_t)12 * nt * 16 * 4) * sizeof(float); }
bool pw_bx3p_ok(int nt, int wm, int K) { return pw_bx3_ok(K) && K > PW_BK && bx3p_lds_bytes(nt, wm) <= 80 * 1024; }

// Plan-time weight image for k_pw_bx3: W [N][K] fp32 -> uint16 [ceil(K/32) slabs][3 planes][4 kq][Npad rows][8], Npad = N rounded
// up to 16 (rows beyond N are zeros), the 8 values of a (row, kq) slot being k = 32 s + 4 kq + (0..3) and 32 s + 16 + 4 kq + (0..3);
// a K tail (K % 32 != 0, K % 4 == 0) is zero weights against zero-filled operand columns.
bool pw_bx3_ok(int K) { return K >= 16 && K % 4 == 0; }
int pw_bx3_npad(int N) { return (N + 15) / 16 * 16; }
std::vector<uint16_t> pw_bx3_image(const float* W, int N, int K) {
    const int Npad = pw_bx3_npad(N), nslab = (K + PW_BK - 1) / PW_BK;
    std::vector<uint16_t> img((size_t)nslab * 12 * Npad * 8, 0);
    for (int n = 0; n < N; n++)
        for (int k = 0; k < K; k++) {
            // round-to-nearest-even bf16 pieces with exact fp32 remainders (same decomposition as bx3_split8)
            auto rne = [](float f) -> uint16_t {
                unsigned u; memcpy(&u, &f, 4);
                if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);      // NaN stays NaN
                return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
            };
            auto widen = [](uint16_t h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; };
            const float x = W[(size_t)n * K + k];
            uint16_t piece[3];
            piece[0] = rne(x);
            const float r = x - widen(piece[0]);
            piece[1] = rne(r);
            const float q = r - widen(piece[1]);
            piece[2] = rne(q);
            const int s = k / PW_BK, kk = k % PW_BK, half = kk / 16, kq = (kk % 16) / 4, j = half * 4 + (kk % 4);
            for (int pl = 0; pl < 3; pl++)
                img[((((size_t)s * 3 + pl) * 4 + kq) * Npad + n) * 8 + j] = piece[pl];
        }
    return img;
}

// The tile of a contraction is tuned at batch size.  A call with fewer clips (one clip per Predict; the 64-clip chunks of a
// blocking 256-clip host call) can leave most of the 256 CUs without a workgroup with that tile: shrink it - 64-row tiles first,
// then narrower column tiles - until the grid has at least one workgroup per CU or the smallest tile is reached.  Pipelined
// kernel forms keep their own constraints, so they fall back to the plain form when the tile changes.
static inline bool pw_fill_grid(int M, int N, int* nt, int* wm, unsigned* nblk, int* nblk_n) {
    static const int target_env = getenv("BNHIP_PW_FILL") ? atoi(getenv("BNHIP_PW_FILL")) : -1;
    const int target = target_env >= 0 ? target_env : device_cus();
    bool changed = false;
    auto count = [&]() { *nblk_n = (N + *nt * 16 - 1) / (*nt * 16); *nblk = (unsigned)((M + 64 * *wm - 1) / (64 * *wm)) * (unsigned)*nblk_n; };
    count();
    while ((int)*nblk < target && (*wm > 1 || *nt > 1)) {
        if (*wm > 1) *wm = 1;
        else *nt = (*nt + 1) / 2;
        changed = true;
        count();
    }
    return changed;
}
void launch_pw_bx3(const PwParams& p, const uint16_t* Wimg, hipStream_t s) {
    if (!(p.sw & (PW_SW_B16_FORCE | PW_SW_B16S_FORCE | PW_SW_WS_FORCE)) && pw_lat_ok(p)) { launch_pw_lat(p, Wimg, pw_bx3_npad(p.N), s); return; }   // small calls, long K (same bits; a parity test's forced kernel goes first)
    if ((p.wm == 11 || (p.sw & PW_SW_B16S_FORCE)) && pw_b16s_ok(p)) { launch_pw_b16s(p, Wimg, pw_bx3_npad(p.N), s); return; }   // skinny layers: weights in registers
    if (((p.wm == 12 && pw_ws_fills(p)) || (p.sw & PW_SW_WS_FORCE)) && pw_ws_ok(p)) { launch_pw_ws(p, Wimg, pw_bx3_npad(p.N), s); return; }   // short K, wide N: weight columns in LDS
    // (a layer tuned onto one of those forms whose call is too small for it - a few clips - takes a tiled kernel: same bits)
    int nt = (p.nt >= 1 && p.nt <= 8) ? p.nt : pick_nt(p.M, p.N);
    int wm = (p.wm == 5 || p.wm == 7 || p.wm == 10) ? 1 : 2;   // PwParams::wm 5 / 6: 64- / 128-row tiles on the split-bf16 kernel, 7 / 8: pipelined,
                                                           // 10 / 9: 64- / 128-row tiles on k_pw_b16 (pw_b16.hip; one-product engines: 128 only)
    bool pipe = (p.wm == 7 || p.wm == 8) && pw_bx3p_ok(nt, wm, p.K);
    int bm = 64 * wm;
    int nblk_n = (p.N + nt * 16 - 1) / (nt * 16);
    unsigned nblk = (unsigned)((p.M + bm - 1) / bm) * nblk_n;
    if (pw_fill_grid(p.M, p.N, &nt, &wm, &nblk, &nblk_n)) { bm = 64 * wm; pipe = false; }
    (void)bm;
    const int Npad = pw_bx3_npad(p.N);
    // "precision":"bf16" engines: 128-row tiles run on the kernel built for one product per operand fragment (pw_b16.hip) -
    // the same arithmetic, A straight from global memory into fragments
    if ((p.wm == 9 || p.wm == 10 || (p.sw & PW_SW_B16_FORCE)) && (wm == 2 || p.prec == 0) && pw_b16_ok(p.prec, p.K, p.sw)) {
        launch_pw_b16(p, Wimg, nt, wm, Npad, nblk_n, nblk, s);
        return;
    }
    const FDiv dn = make_fdiv((unsigned)nblk_n), dhw = make_fdiv((unsigned)std::max(p.HW, 1));
    const bool sc = p.ascale != nullptr;
    dim3 grid(nblk);
#define BX_LAUNCH(NT_, SC_, WM_) hipLaunchKernelGGL((k_pw_bx3<NT_, SC_, WM_>), grid, dim3(256), 0, s, p, Wimg, Npad, nblk_n, nblk, dn, dhw)
#define BX_CASE(NT_) case NT_: if (sc) { if (wm == 1) BX_LAUNCH(NT_, true, 1); else BX_LAUNCH(NT_, true, 2); } \
                     else { if (wm == 1) BX_LAUNCH(NT_, false, 1); else BX_LAUNCH(NT_, false, 2); } break;
    if (pipe) {
        const size_t ldsb = bx3p_lds_bytes(nt, wm);
#define BP_LAUNCH(NT_, SC_, WM_) do { lds_limit_once<&k_pw_bx3p<NT_, SC_, WM_>>(80 * 1024); \
        hipLaunchKernelGGL((k_pw_bx3p<NT_, SC_, WM_>), grid, dim3(256), ldsb, s, p, Wimg, Npad, nblk_n, nblk, dn, dhw); } while (0)
#define BP_CASE(NT_) case NT_: if (sc) { if (wm == 1) BP_LAUNCH(NT_, true, 1); else BP_LAUNCH(NT_, true, 2); } \
                     else { if (wm == 1) BP_LAUNCH(NT_, false, 1); else BP_LAUNCH(NT_, false, 2); } break;
        switch (nt) { BP_CASE(1) BP_CASE(2) BP_CASE(3) BP_CASE(4) BP_CASE(5) BP_CASE(6) BP_CASE(7) default: BP_CASE(8) }
#undef BP_LAUNCH
#undef BP_CASE
        return;
    }
    switch (nt) { BX_CASE(1) BX_CASE(2) BX_CASE(3) BX_CASE(4) BX_CASE(5) BX_CASE(6) BX_CASE(7) default: BX_CASE(8) }
#undef BX_LAUNCH
#undef BX_CASE
}

// scalar fallback for K not a multiple of 4 (never hit by EfficientNet-style graphs; kept for drop-in safety)
__global__ void k_pw_naive(PwParams p) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)p.M * p.N) return;
    int n = (int)(idx % p.N);
    size_t m = idx / p.N;
    float acc = 0.f;
    for (int k = 0; k < p.K; k++) {
        float a = p.A[m * p.K + k];
        if (p.ascale) a *= p.ascale[(m / p.HW) * p.K + k];
        acc = fmaf(a, p.W[(size_t)n * p.K + k], acc);
    }
    if (p.bias) acc += p.bias[n];
    acc = apply_act(acc, p.act);
    if (p.res) acc += p.res[idx];
    p.out[idx] = acc;
}

// Tile-width choice.  Measured on MI355X (tests/micro sweep, late-layer shapes at batch 256): NT <= 4 keeps
// the kernel at <= 112 VGPRs (4 waves/SIMD) and beats the wider tiles (134-158 VGPRs, 2 waves/SIMD) by
// 10-45 % even where they pad less, so: widest NT in 1..4 whose padded width is within 15 % of the best.
// k_pw_pipe needs whole K slabs and both operand buffers inside the 64 KB of static LDS
bool pw_pipe_ok(int nt, int wm, int K) { return K % PW_BK == 0 && nt >= 1 && nt <= 4 && 2 * (64 * wm + 16 * nt) * PW_LS * 4 <= 64 * 1024; }
static int pick_nt(int M, int N) {
    (void)M;
    long best_cols = -1;
    for (int nt = 1; nt <= 4; nt++) {
        long cols = (long)((N + nt * 16 - 1) / (nt * 16)) * nt * 16;
        if (best_cols < 0 || cols < best_cols) best_cols = cols;
    }
    int best = 1;
    for (int nt = 1; nt <= 4; nt++) {
        long cols = (long)((N + nt * 16 - 1) / (nt * 16)) * nt * 16;
        if (cols * 100 <= best_cols * 115) best = nt;
    }
    static const int ov = getenv("BNHIP_PW_NT") ? atoi(getenv("BNHIP_PW_NT")) : 0;      // experiment switch
    if (ov >= 1 && ov <= 8) best = ov;
    return best;
}

int pw_default_nt(int M, int N) { return pick_nt(M, N); }

void launch_pw_gemm(const PwParams& p, hipStream_t s) {
    if ((p.K & 3) != 0) {
        size_t total = (size_t)p.M * p.N;
        hipLaunchKernelGGL(k_pw_naive, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p);
        return;
    }
    int nt = (p.nt >= 1 && p.nt <= 8) ? p.nt : pick_nt(p.M, p.N);
    // wm 3 / 4 = the software-pipelined kernel with 64 / 128-row tiles (needs whole K slabs and nt <= 4)
    static const int wm_env = getenv("BNHIP_PW_WM") ? atoi(getenv("BNHIP_PW_WM")) : 0;      // test switch: force the row tile / kernel
    const int wm_req = wm_env >= 1 && wm_env <= 4 ? wm_env : p.wm;
    bool pipe = (wm_req == 3 || wm_req == 4) && pw_pipe_ok(nt, wm_req - 2, p.K);
    int wm = (wm_req == 1 || wm_req == 3) ? 1 : 2;
    int bm = 64 * wm;
    int nblk_n = (p.N + nt * 16 - 1) / (nt * 16);
    unsigned nblk = (unsigned)((p.M + bm - 1) / bm) * nblk_n;
    // the tile was tuned at batch size; a call with a handful of clips would leave most CUs idle with it (one clip:
    // 1-5 workgroups each walking the whole K loop): fall back to the smallest tile to get workgroups
    static const bool forced = getenv("BNHIP_PW_NT") != nullptr || getenv("BNHIP_PW_WM") != nullptr;   // tests pin the tile
    if (!forced && pw_fill_grid(p.M, p.N, &nt, &wm, &nblk, &nblk_n)) { bm = 64 * wm; pipe = false; }
    (void)bm;
    dim3 grid(nblk);
    const FDiv dn = make_fdiv((unsigned)nblk_n), dhw = make_fdiv((unsigned)std::max(p.HW, 1));
    const bool sc = p.ascale != nullptr;
#define PW_LAUNCH(NT_, SC_, WM_) hipLaunchKernelGGL((k_pw_gemm<NT_, SC_, WM_>), grid, dim3(256), 0, s, p, nblk_n, nblk, dn, dhw, ImGeo{})
#define PW_CASE(NT_) case NT_: if (sc) { if (wm == 1) PW_LAUNCH(NT_, true, 1); else PW_LAUNCH(NT_, true, 2); } \
                     else { if (wm == 1) PW_LAUNCH(NT_, false, 1); else PW_LAUNCH(NT_, false, 2); } break;
#define PP_LAUNCH(NT_, SC_, WM_) hipLaunchKernelGGL((k_pw_pipe<NT_, SC_, WM_>), grid, dim3(256), 0, s, p, nblk_n, nblk, dn, dhw)
#define PP_CASE(NT_) case NT_: if (sc) { if (wm == 1) PP_LAUNCH(NT_, true, 1); else PP_LAUNCH(NT_, true, 2); } \
                     else { if (wm == 1) PP_LAUNCH(NT_, false, 1); else PP_LAUNCH(NT_, false, 2); } break;
    if (pipe) {
        switch (nt) { PP_CASE(1) PP_CASE(2) PP_CASE(3) default: PP_CASE(4) }
        return;
    }
#undef PP_LAUNCH
#undef PP_CASE
    switch (nt) { PW_CASE(1) PW_CASE(2) PW_CASE(3) PW_CASE(4) PW_CASE(5) PW_CASE(6) PW_CASE(7) default: PW_CASE(8) }
#undef PW_LAUNCH
#undef PW_CASE
}

// General convolution as an implicit GEMM on the f32 MFMA (k_pw_gemm<.., IM = true>): any kernel size, stride, dilation and
// explicit top / left padding, Cin % 4 == 0.  Weights are the file's OHWI tensor as it is.  Bias, activation and the output
// burst are the pointwise kernel's epilogue.
bool conv_igemm_supported(int Cin, int Cout, int kh, int kw) { return (Cin & 3) == 0 && Cin >= 4 && kh * kw * Cin >= 32 && Cout >= 1; }
void launch_conv_igemm(const float* in, const float* w_ohwi, const float* bias, float* out, int B, int H, int W, int Cin, int Ho, int Wo,
                       int Cout, int kh, int kw, int sh, int sw, int dh, int dw, int pt, int pl, int act, int nt_req, int wm_req,
                       hipStream_t s) {
    PwParams p{in, w_ohwi, bias, nullptr, nullptr, out, B * Ho * Wo, Cout, kh * kw * Cin, Ho * Wo, act, nt_req, wm_req};
    int nt = (p.nt >= 1 && p.nt <= 8) ? p.nt : pick_nt(p.M, p.N);
    int wm = p.wm == 1 ? 1 : 2;
    int bm = 64 * wm;
    int nblk_n = (p.N + nt * 16 - 1) / (nt * 16);
    unsigned nblk = (unsigned)((p.M + bm - 1) / bm) * nblk_n;
    if (nblk < 64 && (nt > 1 || wm > 1)) {
        nt = 1; wm = 1; bm = 64;
        nblk_n = (p.N + 15) / 16;
        nblk = (unsigned)((p.M + bm - 1) / bm) * nblk_n;
    }
    ImGeo g{H, W, Cin, kw, sh, sw, dh, dw, pt, pl, Wo, make_fdiv((unsigned)Cin), make_fdiv((unsigned)kw), make_fdiv((unsigned)Wo),
            make_fdiv((unsigned)(Ho * Wo))};
    const FDiv dn = make_fdiv((unsigned)nblk_n), dhw = make_fdiv((unsigned)std::max(p.HW, 1));
    dim3 grid(nblk);
#define IG_LAUNCH(NT_, WM_) hipLaunchKernelGGL((k_pw_gemm<NT_, false, WM_, true>), grid, dim3(256), 0, s, p, nblk_n, nblk, dn, dhw, g)
#define IG_CASE(NT_) case NT_: if (wm == 1) IG_LAUNCH(NT_, 1); else IG_LAUNCH(NT_, 2); break;
    switch (nt) { IG_CASE(1) IG_CASE(2) IG_CASE(3) IG_CASE(4) IG_CASE(5) IG_CASE(6) IG_CASE(7) default: IG_CASE(8) }
#undef IG_LAUNCH
#undef IG_CASE
}

// ------------------------------------------------------------------------------------------ depthwise conv
// Register-tiled depthwise conv: a thread owns 4 channels x (TH x TW) output pixels, so each input
// row segment it loads is reused across the TW horizontal and up to K vertical taps (HBM/L2 traffic per
// output drops from K*K loads to ~((TH-1)S+K)((TW-1)S+K)/(TH*TW)).  Threads are laid out channel-fastest
// (coalesced float4), tiles row-major; blocks are XCD-remapped so vertically adjacent tiles of a clip share an
// L2.  Optionally emits deterministic per-block channel sums for the squeeze-excite mean (no second pass over
// the tensor, no float atomics): partial[b][tile_chunk][c].
// IN16 (bf16 activation storage on the input): the thread's whole input patch - RH x RW quads of 8 bytes - is requested before
// the first tap is applied, so a block pays one memory round trip instead of one per input row (the row loop otherwise issues
// RW loads, waits, multiplies, and only then issues the next row's: b1 / b2 of the Perch stack ran at 2.0 / 3.2 TB/s).
template <int K, int S, int TH, int TW, bool IN16 = false>
__global__ __launch_bounds__(256) void k_dwconv_t(DwParams p, int CX, int PY, int tiles_w, int tiles, int tchunks,
                                                  int cchunks, unsigned nblk, float* __restrict__ partial) {
    __shared__ float red[256 * 4];
    const unsigned L = xcd_remap(blockIdx.x, nblk);
    const int bpc = tchunks * cchunks;
    const int b = L / bpc;
    const int rest = L % bpc;
    const int tc = rest / cchunks, cc = rest % cchunks;
    const int tx = threadIdx.x % CX, ty = threadIdx.x / CX;
    const int c4 = cc * CX + tx;
    const int C4 = p.C >> 2;
    const int tile = tc * PY + ty;
    const bool live = c4 < C4 && tile < tiles && ty < PY;
    float4 acc[TH][TW];
#pragma unroll
    for (int a = 0; a < TH; a++)
#pragma unroll
        for (int c = 0; c < TW; c++) acc[a][c] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int th0 = live ? (tile / tiles_w) * TH : 0, tw0 = live ? (tile % tiles_w) * TW : 0;
    if (live) {
        constexpr int RW = (TW - 1) * S + K;       // input columns per row segment
        constexpr int RH = (TH - 1) * S + K;       // input rows
        const int hi0 = th0 * S - p.pt, wi0 = tw0 * S - p.pl;
        const float4* in4 = reinterpret_cast<const float4*>(p.in) + (size_t)b * p.H * p.W * C4 + c4;
        const float4* w4 = reinterpret_cast<const float4*>(p.w) + c4;
        uint2 raw[IN16 ? RH : 1][IN16 ? RW : 1];
        if constexpr (IN16) {
            const uint2* in2 = reinterpret_cast<const uint2*>(p.in) + (size_t)b * p.H * p.W * C4 + c4;
#pragma unroll
            for (int r = 0; r < RH; r++)
#pragma unroll
                for (int c = 0; c < RW; c++) {
                    const int hi = hi0 + r, wi = wi0 + c;
                    const bool in = hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
                    raw[r][c] = in ? in2[((size_t)hi * p.W + wi) * C4] : make_uint2(0u, 0u);
                }
        }
#pragma unroll
        for (int r = 0; r < RH; r++) {
            const int hi = hi0 + r;
            if (hi < 0 || hi >= p.H) continue;
            float4 x[RW];
#pragma unroll
            for (int c = 0; c < RW; c++) {
                int wi = wi0 + c;
                if constexpr (IN16) {
                    const uint2 rr = raw[r][c];
                    x[c] = make_float4(__uint_as_float(rr.x << 16), __uint_as_float(rr.x & 0xffff0000u), __uint_as_float(rr.y << 16), __uint_as_float(rr.y & 0xffff0000u));
                } else if (!(wi >= 0 && wi < p.W)) x[c] = make_float4(0.f, 0.f, 0.f, 0.f);
                else if (p.in_bf16) x[c] = bf16x4_load(p.in, ((size_t)b * p.H * p.W + (size_t)hi * p.W + wi) * C4 + c4);
                else x[c] = in4[((size_t)hi * p.W + wi) * C4];
            }
#pragma unroll
            for (int a = 0; a < TH; a++) {
                const int i = r - a * S;            // kernel row feeding output row a from input row r
                if (i < 0 || i >= K) continue;
#pragma unroll
                for (int j = 0; j < K; j++) {
                    const float4 w = w4[(size_t)(i * K + j) * C4];
#pragma unroll
                    for (int c = 0; c < TW; c++) {
                        const float4 xv = x[c * S + j];
                        acc[a][c].x = fmaf(xv.x, w.x, acc[a][c].x); acc[a][c].y = fmaf(xv.y, w.y, acc[a][c].y);
                        acc[a][c].z = fmaf(xv.z, w.z, acc[a][c].z); acc[a][c].w = fmaf(xv.w, w.w, acc[a][c].w);
                    }
                }
            }
        }
    }
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        float4 bv = p.bias ? reinterpret_cast<const float4*>(p.bias)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
        float4* out4 = reinterpret_cast<float4*>(p.out) + (size_t)b * p.Ho * p.Wo * C4 + c4;
        if (p.act == ACT_SWISH) {
#pragma unroll
            for (int a = 0; a < TH; a++)
#pragma unroll
                for (int c = 0; c < TW; c++) {
                    float4& v = acc[a][c];
                    const f32x4 r = swish4((f32x4){v.x + bv.x, v.y + bv.y, v.z + bv.z, v.w + bv.w});
                    v = make_float4(r[0], r[1], r[2], r[3]);
                }
        } else {
            with_act(p.act, [&](auto f) {
#pragma unroll
                for (int a = 0; a < TH; a++)
#pragma unroll
                    for (int c = 0; c < TW; c++) {
                        float4& v = acc[a][c];
                        v.x = f(v.x + bv.x); v.y = f(v.y + bv.y); v.z = f(v.z + bv.z); v.w = f(v.w + bv.w);
                    }
            });
        }
#pragma unroll
        for (int a = 0; a < TH; a++) {
            int ho = th0 + a;
            if (ho >= p.Ho) continue;
#pragma unroll
            for (int c = 0; c < TW; c++) {
                int wo = tw0 + c;
                if (wo >= p.Wo) continue;
                float4 v = acc[a][c];
                if (p.out_bf16) bf16x4_store(p.out, ((size_t)b * p.Ho * p.Wo + (size_t)ho * p.Wo + wo) * C4 + c4, v);
                else out4[((size_t)ho * p.Wo + wo) * C4] = v;
                sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
            }
        }
    }
    if (partial) {
        reinterpret_cast<float4*>(red)[threadIdx.x] = sum;
        __syncthreads();
        if (ty == 0 && c4 < C4) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int y = 0; y < PY; y++) {
                float4 v = reinterpret_cast<float4*>(red)[y * CX + tx];
                t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
            }
            reinterpret_cast<float4*>(partial)[((size_t)b * tchunks + tc) * C4 + c4] = t;
        }
    }
}

// generic fallback: thread = (output pixel, 4 or 1 channels)
template <int VEC>
__global__ __launch_bounds__(256) void k_dwconv(DwParams p) {
    const int CV = p.C / VEC;
    size_t total = (size_t)p.B * p.Ho * p.Wo * CV;
    size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    int cv = (int)(idx % CV);
    size_t pix = idx / CV;
    int wo = (int)(pix % p.Wo);
    int ho = (int)((pix / p.Wo) % p.Ho);
    int b = (int)(pix / ((size_t)p.Wo * p.Ho));
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; v++) acc[v] = 0.f;
    for (int i = 0; i < p.kh; i++) {
        int hi = ho * p.sh - p.pt + i;
        if (hi < 0 || hi >= p.H) continue;
        for (int j = 0; j < p.kw; j++) {
            int wi = wo * p.sw - p.pl + j;
            if (wi < 0 || wi >= p.W) continue;
            const float* ip = p.in + (((size_t)b * p.H + hi) * p.W + wi) * p.C + (size_t)cv * VEC;
            const float* wp = p.w + (size_t)(i * p.kw + j) * p.C + (size_t)cv * VEC;
#pragma unroll
            for (int v = 0; v < VEC; v++) acc[v] = fmaf(ip[v], wp[v], acc[v]);
        }
    }
#pragma unroll
    for (int v = 0; v < VEC; v++) {
        float x = acc[v];
        if (p.bias) x += p.bias[cv * VEC + v];
        p.out[idx * VEC + v] = apply_act(x, p.act);
    }
}

static void dw_geometry(const DwParams& p, int TH, int TW, int* CX, int* PY, int* tiles_w, int* tiles, int* tchunks,
                        int* cchunks) {
    int C4 = p.C / 4;
    *CX = C4 < 64 ? C4 : 64;
    *PY = 256 / *CX;
    *tiles_w = (p.Wo + TW - 1) / TW;
    *tiles = *tiles_w * ((p.Ho + TH - 1) / TH);
    if (*PY > *tiles) *PY = *tiles;
    *tchunks = (*tiles + *PY - 1) / *PY;
    *cchunks = (C4 + *CX - 1) / *CX;
}
static bool dw_tiled_shape(const DwParams& p, int* TH, int* TW) {
    if ((p.C & 3) || p.kh != p.kw || p.sh != p.sw) return false;
    if (p.kh == 3 && p.sh == 1) { *TH = 2; *TW = 4; return true; }
    if (p.kh == 3 && p.sh == 2) { *TH = 2; *TW = 2; return true; }
    if (p.kh == 5 && p.sh == 1) { *TH = 2; *TW = 4; return true; }
    if (p.kh == 5 && p.sh == 2) { *TH = 1; *TW = 2; return true; }
    return false;
}
int dwconv_sum_slabs(const DwParams& p) {
    int TH, TW, CX, PY, tw, t, tch, cch;
    if (!dw_tiled_shape(p, &TH, &TW)) return 0;
    dw_geometry(p, TH, TW, &CX, &PY, &tw, &t, &tch, &cch);
    return tch;
}
void launch_dwconv(const DwParams& p, float* partial, hipStream_t s) {
    int TH, TW;
    if (dw_tiled_shape(p, &TH, &TW)) {
        int CX, PY, tiles_w, tiles, tchunks, cchunks;
        dw_geometry(p, TH, TW, &CX, &PY, &tiles_w, &tiles, &tchunks, &cchunks);
        unsigned nblk = (unsigned)p.B * tchunks * cchunks;
        dim3 block(CX * PY < 64 ? 64 : CX * PY);
        static const bool pre16 = !(getenv("BNHIP_DW_PREFETCH") && atoi(getenv("BNHIP_DW_PREFETCH")) == 0);
#define DW_LAUNCH(K_, S_, TH_, TW_) do { if (p.in_bf16 && pre16) hipLaunchKernelGGL((k_dwconv_t<K_, S_, TH_, TW_, true>), dim3(nblk), block, 0, s, p, CX, PY, \
                                                       tiles_w, tiles, tchunks, cchunks, nblk, partial); \
                                         else hipLaunchKernelGGL((k_dwconv_t<K_, S_, TH_, TW_>), dim3(nblk), block, 0, s, p, CX, PY, \
                                                       tiles_w, tiles, tchunks, cchunks, nblk, partial); } while (0)
        if (p.kh == 3 && p.sh == 1) DW_LAUNCH(3, 1, 2, 4);
        else if (p.kh == 3) DW_LAUNCH(3, 2, 2, 2);
        else if (p.sh == 1) DW_LAUNCH(5, 1, 2, 4);
        else DW_LAUNCH(5, 2, 1, 2);
#undef DW_LAUNCH
        return;
    }
    if ((p.C & 3) == 0) {
        size_t total = (size_t)p.B * p.Ho * p.Wo * (p.C / 4);
        hipLaunchKernelGGL(k_dwconv<4>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p);
    } else {
        size_t total = (size_t)p.B * p.Ho * p.Wo * p.C;
        hipLaunchKernelGGL(k_dwconv<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p);
    }
}

// ------------------------------------------------------------------------------------------ spatial mean
// Deterministic two-stage reduction (no float atomics): partial[b][s][c] = sum over the s-th pixel slab.
#define MEAN_SLAB 512
int mean_splits(int HW) { return (HW + MEAN_SLAB - 1) / MEAN_SLAB; }

__global__ __launch_bounds__(256) void k_mean_partial(const float* __restrict__ in, float* __restrict__ partial,
                                                      int HW, int C, int S) {
    __shared__ float red[256];
    const int b = blockIdx.z, sp = blockIdx.y;
    const int CW = blockDim.x, PY = blockDim.y;
    const int c = blockIdx.x * CW + threadIdx.x;
    const int p0 = sp * MEAN_SLAB, p1 = min(HW, p0 + MEAN_SLAB);
    float acc = 0.f;
    if (c < C)
        for (int px = p0 + threadIdx.y; px < p1; px += PY) acc += in[((size_t)b * HW + px) * C + c];
    red[threadIdx.y * CW + threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.y == 0 && c < C) {
        float sum = 0.f;
        for (int y = 0; y < PY; y++) sum += red[y * CW + threadIdx.x];
        partial[((size_t)b * S + sp) * C + c] = sum;
    }
}
void launch_mean_partial(const float* in, float* partial, int B, int HW, int C, int S, hipStream_t s) {
    int CW = C < 64 ? C : 64;
    int PY = 256 / CW; if (PY < 1) PY = 1;
    dim3 grid((C + CW - 1) / CW, S, B);
    hipLaunchKernelGGL(k_mean_partial, grid, dim3(CW, PY), 0, s, in, partial, HW, C, S);
}
__global__ void k_mean_finish(const float* __restrict__ partial, float* __restrict__ out, int B, int HW, int C, int S) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * C) return;
    int c = (int)(idx % C); size_t b = idx / C;
    float sum = 0.f;
    for (int sidx = 0; sidx < S; sidx++) sum += partial[(b * S + sidx) * C + c];
    out[idx] = sum / (float)HW;
}
void launch_mean_finish(const float* partial, float* out, int B, int HW, int C, int S, hipStream_t s) {
    size_t total = (size_t)B * C;
    hipLaunchKernelGGL(k_mean_finish, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, partial, out, B, HW, C, S);
}

// ------------------------------------------------------------------------------------------ squeeze-excite
// One block (16 waves) per clip: mean -> FC(Cr)+act1 -> FC(C)+act2 -> scale[b][c].
// w1 [Cr][C] is read wave-per-output with coalesced rows; w2t is the second FC transposed to [Cr][C] at plan
// time so thread c reads it coalesced.  (Splitting a clip over 4 blocks that each redo mean+FC1 measured 2x slower:
// the pass over the per-slab sums dominates.)
__global__ __launch_bounds__(1024) void k_se(SeParams p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* mean = sm;            // [C]
    float* r = sm + p.C;         // [Cr]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NTHR = blockDim.x, NW = NTHR >> 6;      // block size is a launch parameter (see launch_se)
    const float inv = 1.0f / (float)p.HW;
    // the per-slab sums: P threads per channel walk interleaved slab subsets (a 96-slab layer with 96 channels used to
    // be 96 threads x 96 serial loads), then fold through LDS in a fixed order
    int P = 1;
    while (P * 2 * p.C <= NTHR && P * 2 <= p.S) P *= 2;
    float* part = sm + ((p.C + p.Cr + 3) & ~3);      // [P][C] / float4 [G][C/4] scratch, 16-byte aligned
    if (P > 1) {
        const int c = tid % p.C, q = tid / p.C;
        if (q < P) {
            float sum = 0.f;
#pragma unroll 4
            for (int sidx = q; sidx < p.S; sidx += P) sum += p.partial[((size_t)b * p.S + sidx) * p.C + c];
            part[q * p.C + c] = sum;
        }
        __syncthreads();
        for (int c2 = tid; c2 < p.C; c2 += NTHR) {
            float sum = 0.f;
            for (int q2 = 0; q2 < P; q2++) sum += part[q2 * p.C + c2];
            mean[c2] = sum * inv;
        }
    } else {
        for (int c = tid; c < p.C; c += NTHR) {
            float sum = 0.f;
#pragma unroll 4
            for (int sidx = 0; sidx < p.S; sidx++) sum += p.partial[((size_t)b * p.S + sidx) * p.C + c];
            mean[c] = sum * inv;
        }
    }
    __syncthreads();
    const bool v4 = (p.C & 3) == 0;
    // FC1: one wave per output, 16-byte loads (the scalar form was ~54 dependent 4-byte loads per thread per FC).  A wave takes
    // its outputs four at a time: the kernel is bound by round trips to the weights, so four rows' loads are in flight together
    // (one clip: 21.7 -> 15.7 us on a 1152-channel layer, 11.9 -> 8.3 us on the first); each output's own sum keeps its order
    // (a lane's columns ascending, then the butterfly), so no bit changes.
    for (int j0 = wave; j0 < p.Cr; j0 += 4 * NW) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (v4) {
            const float4* m4 = reinterpret_cast<const float4*>(mean);
            const float4* w4 = reinterpret_cast<const float4*>(p.w1);
            const int C4 = p.C / 4;
#pragma unroll 2
            for (int c = lane; c < C4; c += 64) {
                const float4 mv = m4[c];
                float4 w[4];
#pragma unroll
                for (int u = 0; u < 4; u++) w[u] = w4[(size_t)min(j0 + u * NW, p.Cr - 1) * C4 + c];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    acc[u] = fmaf(w[u].x, mv.x, acc[u]); acc[u] = fmaf(w[u].y, mv.y, acc[u]);
                    acc[u] = fmaf(w[u].z, mv.z, acc[u]); acc[u] = fmaf(w[u].w, mv.w, acc[u]);
                }
            }
        } else {
            for (int c = lane; c < p.C; c += 64) {
                const float mv = mean[c];
#pragma unroll
                for (int u = 0; u < 4; u++) acc[u] = fmaf(p.w1[(size_t)min(j0 + u * NW, p.Cr - 1) * p.C + c], mv, acc[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            float a = acc[u];
            for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
            const int j = j0 + u * NW;
            if (lane == 0 && j < p.Cr) r[j] = apply_act(a + (p.b1 ? p.b1[j] : 0.f), p.act1);
        }
    }
    __syncthreads();
    // FC2: a thread owns 4 channels; G thread groups split the Cr range and fold through LDS in a fixed order
    if (v4) {
        const int C4 = p.C / 4;
        int G = 1;
        while (G * 2 * C4 <= NTHR && G * 2 <= p.Cr) G *= 2;
        float4* part4 = reinterpret_cast<float4*>(part);           // [G][C4] (G * C <= 4096 floats <= scratch? see launch)
        // (C4 <= NTHR is guaranteed by launch_se: a block is never smaller than the channel-quad count)
        const int c4 = tid % C4, g = tid / C4;
        if (g < G) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4* w4 = reinterpret_cast<const float4*>(p.w2) + c4;
#pragma unroll 8
            for (int j = g; j < p.Cr; j += G) {
                float4 w = w4[(size_t)j * C4];
                float rj = r[j];
                acc.x = fmaf(w.x, rj, acc.x); acc.y = fmaf(w.y, rj, acc.y); acc.z = fmaf(w.z, rj, acc.z); acc.w = fmaf(w.w, rj, acc.w);
            }
            if (G > 1) part4[g * C4 + c4] = acc;
            else {
                float4 bb = p.b2 ? reinterpret_cast<const float4*>(p.b2)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
                float4 o = make_float4(apply_act(acc.x + bb.x, p.act2), apply_act(acc.y + bb.y, p.act2),
                                       apply_act(acc.z + bb.z, p.act2), apply_act(acc.w + bb.w, p.act2));
                reinterpret_cast<float4*>(p.scale + (size_t)b * p.C)[c4] = o;
            }
        }
        if (G > 1) {
            __syncthreads();
            for (int c = tid; c < C4; c += NTHR) {
                float4 acc = part4[c];
                for (int g2 = 1; g2 < G; g2++) { float4 v = part4[g2 * C4 + c]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
                float4 bb = p.b2 ? reinterpret_cast<const float4*>(p.b2)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
                float4 o = make_float4(apply_act(acc.x + bb.x, p.act2), apply_act(acc.y + bb.y, p.act2),
                                       apply_act(acc.z + bb.z, p.act2), apply_act(acc.w + bb.w, p.act2));
                reinterpret_cast<float4*>(p.scale + (size_t)b * p.C)[c] = o;
            }
        }
    } else {
        for (int c = tid; c < p.C; c += NTHR) {
            float acc = 0.f;
#pragma unroll 8
            for (int j = 0; j < p.Cr; j++) acc = fmaf(p.w2[(size_t)j * p.C + c], r[j], acc);
            p.scale[(size_t)b * p.C + c] = apply_act(acc + (p.b2 ? p.b2[c] : 0.f), p.act2);
        }
    }
}
void launch_se(const SeParams& p, hipStream_t s) {
    // Block size: 1024 threads finish a clip fastest when the kernel owns the GPU, but a 16-wave workgroup needs a whole
    // CU's worth of free wave slots and, beside another context's kernels, waited for one 5-30x its own run time
    // (rocprofv3, timed window: avg 50 us, max 330 us against 6-18 us alone) - stalling the dependent projection GEMM.
    // A pipelined engine therefore launches 4-wave blocks that slot in anywhere (BNHIP_SE_THREADS overrides).
    static const int env = getenv("BNHIP_SE_THREADS") ? atoi(getenv("BNHIP_SE_THREADS")) : 0;
    int thr = env ? env : (p.threads ? p.threads : 1024);
    thr = std::max(64, std::min(1024, thr / 64 * 64));
    while (thr < 1024 && (p.C + 3) / 4 > thr) thr *= 2;              // FC2 maps one thread to a channel quad
    // mean, r, fold scratch: [P][C] floats with P C <= threads, or [G][C / 4] float4 with G C / 4 <= threads - i.e. at most one
    // float4 per thread.  (Sized for 1024 threads whatever the block, a 4-wave block asked for 16 KB more LDS than it can use and,
    // beside another context's LDS-heavy kernels, waited for it: rocprofv3, Perch bf16 pipelined: avg 62 us, max 777 us.)
    const size_t lds = (size_t)(p.C + p.Cr + 4 * thr + 16) * sizeof(float);
    hipLaunchKernelGGL(k_se, dim3(p.B), dim3(thr), lds, s, p);
}

// ------------------------------------------------------------------------------------------ generic elementwise
__global__ void k_unary(const float* __restrict__ in, float* __restrict__ out, size_t n, int act) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = apply_act(in[i], act);
}
void launch_unary(const float* in, float* out, size_t n, int act, hipStream_t s) {
    size_t blocks = (n + 255) / 256; if (blocks > 16384) blocks = 16384; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_unary, dim3((unsigned)blocks), dim3(256), 0, s, in, out, n, act);
}
__global__ void k_binary(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, size_t n,
                         int op, int mode, int HW, int C, int act) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float y = mode == 0 ? b[i] : (mode == 2 ? b[0] : b[(i / ((size_t)HW * C)) * C + (i % C)]);
        float x = a[i];
        float v = op == 0 ? x + y : (op == 1 ? x * y : x - y);
        out[i] = apply_act(v, act);
    }
}
void launch_binary(const float* a, const float* b, float* out, size_t n, int op, int mode, int HW, int C, int act,
                   hipStream_t s) {
    size_t blocks = (n + 255) / 256; if (blocks > 16384) blocks = 16384; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_binary, dim3((unsigned)blocks), dim3(256), 0, s, a, b, out, n, op, mode, HW, C, act);
}

// ------------------------------------------------------------------------------------------ post-processing
// classifier/analyze.go:113-115,197-208 (mode 0); onnx/postprocess.go:8-10 (mode 2)
__global__ void k_sigmoid(const float* __restrict__ x, float* __restrict__ out, size_t n, int mode, double sens) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (mode == 0) out[i] = (float)(1.0 / (1.0 + exp(-sens * (double)x[i])));
    else out[i] = 1.0f / (1.0f + (float)exp((double)(-x[i])));
}
// classifier/perch_onnx.go:315-335: f32 max, e = float32(exp(float64(x - m))), f32 sum in index order, divide
__global__ __launch_bounds__(256) void k_softmax(const float* __restrict__ x, float* __restrict__ out, int n) {
    __shared__ float red[4];
    __shared__ float s_sum;
    const float* xr = x + (size_t)blockIdx.x * n;
    float* orow = out + (size_t)blockIdx.x * n;
    float m = -INFINITY;
    for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, xr[i]);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    for (int i = threadIdx.x; i < n; i += 256) orow[i] = (float)exp((double)(xr[i] - m));
    __syncthreads();
    if (threadIdx.x == 0) {          // sequential f32 sum in index order == the Go loop, bit for bit
        float sum = 0.f;
        for (int i = 0; i < n; i++) sum += orow[i];
        s_sum = sum;
    }
    __syncthreads();
    float sum = s_sum;
    for (int i = threadIdx.x; i < n; i += 256) orow[i] = orow[i] / sum;
}
void launch_activation(const float* logits, float* conf, int n_clips, int n_classes, int activation, double sens,
                       hipStream_t s) {
    if (activation == 1) {
        hipLaunchKernelGGL(k_softmax, dim3(n_clips), dim3(256), 0, s, logits, conf, n_classes);
    } else {
        size_t n = (size_t)n_clips * n_classes;
        hipLaunchKernelGGL(k_sigmoid, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, logits, conf, n, activation, sens);
    }
}

// top-k by confidence, descending; ties resolved to the lower label index (the reference's order for
// ties is implementation-defined: analyze.go:120-124 uses the unstable sort.Slice).
__global__ __launch_bounds__(256) void k_topk(const float* __restrict__ conf, int n, int k, float* __restrict__ oc,
                                              int32_t* __restrict__ oi) {
    extern __shared__ float v[];
    __shared__ float rv[4];
    __shared__ int ri[4];
    const float* row = conf + (size_t)blockIdx.x * n;
    // NaN confidences (non-finite logits) rank below everything: loaded as -inf, so every emitted index is in [0, n)
    for (int i = threadIdx.x; i < n; i += 256) { float x = row[i]; v[i] = x != x ? -INFINITY : x; }
    __syncthreads();
    int kk = k < n ? k : n;
    for (int r = 0; r < kk; r++) {
        float best = -INFINITY; int bi = 0x7fffffff;
        for (int i = threadIdx.x; i < n; i += 256) {
            float x = v[i];
            if (x > best || (x == best && i < bi)) { best = x; bi = i; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            float ob = __shfl_down(best, o, 64); int oidx = __shfl_down(bi, o, 64);
            if (ob > best || (ob == best && oidx < bi)) { best = ob; bi = oidx; }
        }
        if ((threadIdx.x & 63) == 0) { rv[threadIdx.x >> 6] = best; ri[threadIdx.x >> 6] = bi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; w++)
                if (rv[w] > best || (rv[w] == best && ri[w] < bi)) { best = rv[w]; bi = ri[w]; }
            oc[(size_t)blockIdx.x * k + r] = best;
            oi[(size_t)blockIdx.x * k + r] = bi;
            if (bi >= 0 && bi < n) v[bi] = __builtin_nanf("");      // taken: NaN never compares > or ==, so -inf ties stay selectable
        }
        __syncthreads();
    }
}
void launch_topk(const float* conf, int n_clips, int n_classes, int k, float* out_conf, int32_t* out_idx,
                 hipStream_t s) {
    if ((size_t)n_classes * sizeof(float) > 48 * 1024)      // class counts above 12 K need more than the default dynamic LDS (per device: not cached)
        lds_limit_once<&k_topk>(160 * 1024 - 256);
    hipLaunchKernelGGL(k_topk, dim3(n_clips), dim3(256), (size_t)n_classes * sizeof(float), s, conf, n_classes, k,
                       out_conf, out_idx);
}

// ------------------------------------------------------------------------------------------ ultrasonic frame-CV
// internal/audiocore/ultrasonic/filter.go:20-66 in float64.  One block per (frame, clip); the 8192-point
// complex128 FFT lives entirely in LDS (128 KiB of the CU's 160 KiB): bit-reversed load with the symmetric
// Hann window applied, radix-2 DIT stages with directly evaluated twiddles (the Go code's w *= wn recurrence
// only adds rounding noise), then the one-sided power sum above the split bin.
// T = double (the reference's float64 samples) or int16_t (raw PCM: int16 / 32768 as float64, convert/pcm.go:108-113)
template <typename T>
__global__ __launch_bounds__(1024) void k_us_frame_power(const T* __restrict__ samples, int n, int fft, int hop,
                                                         int frames, int split_bin, int log2n,
                                                         const double2* __restrict__ tw /* [fft/2] (cos, -sin)(2 pi j / fft), then the Hann window [fft] */,
                                                         double* __restrict__ powers) {
    extern __shared__ __attribute__((aligned(16))) double lds[];   // re[fft], im[fft]
    double* re = lds; double* im = lds + fft;
    __shared__ double red[16];
    const int frame = blockIdx.x, clip = blockIdx.y;
    const T* x = samples + (size_t)clip * n + (size_t)frame * hop;
    const double* __restrict__ hann = reinterpret_cast<const double*>(tw + fft / 2);    // symmetric Hann window, filter.go:139-145
    for (int i = threadIdx.x; i < fft; i += blockDim.x) {
        double w = hann[i];
        unsigned j = __brev((unsigned)i) >> (32 - log2n);
        const double xv = std::is_same<T, int16_t>::value ? (double)x[i] / 32768.0 : (double)x[i];
        re[j] = xv * w; im[j] = 0.0;
    }
    __syncthreads();
    // twiddles from a plan-time table (L1 / L2 resident, 64 KiB for 8192 points): evaluating sincos in float64 per butterfly
    // was ~5x the butterfly arithmetic itself (2.0 ms for 256 x 34 frames; the Go code's w *= wn recurrence only adds noise)
    int shift = log2n - 1;
    for (int size = 2; size <= fft; size <<= 1, shift--) {
        int half = size >> 1;
        for (int t = threadIdx.x; t < fft / 2; t += blockDim.x) {
            int k = t & (half - 1);
            int i0 = ((t - k) << 1) + k, i1 = i0 + half;
            const double2 w = tw[k << shift];
            const double c = w.x, s = w.y;
            double vr = c * re[i1] - s * im[i1], vi = c * im[i1] + s * re[i1];
            double ur = re[i0], ui = im[i0];
            re[i0] = ur + vr; im[i0] = ui + vi; re[i1] = ur - vr; im[i1] = ui - vi;
        }
        __syncthreads();
    }
    const int nyq = fft / 2;
    double pw = 0.0;
    for (int b = split_bin + threadIdx.x; b <= nyq; b += blockDim.x) {
        double q = re[b] * re[b] + im[b] * im[b];
        if (b > 0 && b < nyq) q *= 2.0;
        pw += q;
    }
    for (int o = 32; o > 0; o >>= 1) pw += __shfl_down(pw, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = pw;
    __syncthreads();
    if (threadIdx.x == 0) {
        double sum = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); w++) sum += red[w];
        powers[(size_t)clip * frames + frame] = sum;
    }
}
// 8192-point frames (the reference's default, filter.go:20-66) the fast way.  The frame is real: its even / odd samples are
// packed into 4096 complex points, transformed by four in-place radix-8 decimation-in-frequency passes (512 threads, one 8-point
// butterfly per thread and pass, the 8-point DFT in registers: 4 LDS round trips instead of the 13 of the radix-2 kernel above,
// on half the data), and the spectrum of the real frame is recovered on the fly in the power sum: X[k] = E[k] + W^k O[k] with
// E, O from Z[k] and conj(Z[4096 - k]).  DIF leaves Z[k] at the base-8 digit-reversed index.  LDS index i lives at i + (i >> 3)
// (one pad per 8 doubles: the last pass walks the array with stride 8).  73.7 KB of LDS per frame: two frames per CU.
#define US8_N2 4096
#define US8_PHYS(i) ((i) + ((i) >> 3))
template <typename T>
__global__ __launch_bounds__(512) void k_us_frame_power8(const T* __restrict__ samples, int n, int hop, int frames, int split_bin,
                                                         const double2* __restrict__ tw /* [4096] (cos, -sin)(2 pi j / 8192), then Hann [8192] */,
                                                         double* __restrict__ powers) {
    extern __shared__ __attribute__((aligned(16))) double us8_lds[];
    double* zr = us8_lds; double* zi = us8_lds + US8_PHYS(US8_N2);
    __shared__ double red[8];
    const int tid = threadIdx.x, frame = blockIdx.x, clip = blockIdx.y;
    const T* x = samples + (size_t)clip * n + (size_t)frame * hop;
    const double* __restrict__ hann = reinterpret_cast<const double*>(tw + US8_N2);
    auto sample = [&](int i) -> double { return std::is_same<T, int16_t>::value ? (double)x[i] / 32768.0 : (double)x[i]; };
    for (int i = tid; i < US8_N2; i += 512) {
        zr[US8_PHYS(i)] = sample(2 * i) * hann[2 * i];
        zi[US8_PHYS(i)] = sample(2 * i + 1) * hann[2 * i + 1];
    }
    __syncthreads();
#pragma unroll 1
    for (int st = 0; st < 4; st++) {
        const int L = US8_N2 >> (3 * st), span = L >> 3;
        const int j = tid & (span - 1), base = ((tid - j) << 3) + j;  // (tid / span) * L + j
        double re[8], im[8];
#pragma unroll
        for (int m = 0; m < 8; m++) { const int i = US8_PHYS(base + m * span); re[m] = zr[i]; im[m] = zi[i]; }
        fft_r8_dft8(re, im);
        const int tstep = j * (8192 / L);                            // W_L^(j q) = W_8192^(q * tstep)
#pragma unroll
        for (int sl = 0; sl < 8; sl++) {
            const int q = kFftR8Slot[sl];
            double yr = re[sl], yi = im[sl];
            if (q != 0 && st < 3) {                                  // the last pass has span 1: j = 0, no twiddles
                int e = q * tstep;                                   // < 7168
                double2 w = tw[e & 4095];
                if (e >= 4096) { w.x = -w.x; w.y = -w.y; }
                const double tr = yr * w.x - yi * w.y, ti = yr * w.y + yi * w.x;
                yr = tr; yi = ti;
            }
            const int i = US8_PHYS(base + q * span);
            zr[i] = yr; zi[i] = yi;
        }
        __syncthreads();
    }
    // power above the split bin (filter.go:48-63) from the spectrum of the real frame
    auto rev = [](int k) { return ((k & 7) << 9) | (((k >> 3) & 7) << 6) | (((k >> 6) & 7) << 3) | ((k >> 9) & 7); };
    const int nyq = US8_N2;                                          // bin index of the Nyquist frequency (fft / 2)
    double pw = 0.0;
    for (int b = split_bin + tid; b <= nyq; b += 512) {
        double xr, xi;
        if (b == nyq) { const int i0 = US8_PHYS(0); xr = zr[i0] - zi[i0]; xi = 0.0; }
        else {
            const int ia = US8_PHYS(rev(b)), ib = US8_PHYS(rev((US8_N2 - b) & (US8_N2 - 1)));
            const double ar = zr[ia], ai = zi[ia], br = zr[ib], bi = -zi[ib];      // Z[b], conj(Z[N2 - b])
            const double er = 0.5 * (ar + br), ei = 0.5 * (ai + bi);
            const double dr = ar - br, di = ai - bi;
            const double orr = 0.5 * di, oi = -0.5 * dr;                             // O = -i/2 (Z[b] - conj(Z[N2 - b]))
            const double2 w = tw[b];                                                 // W_8192^b
            xr = er + (orr * w.x - oi * w.y); xi = ei + (orr * w.y + oi * w.x);
        }
        double q = xr * xr + xi * xi;
        if (b > 0 && b < nyq) q *= 2.0;
        pw += q;
    }
    for (int o = 32; o > 0; o >>= 1) pw += __shfl_down(pw, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = pw;
    __syncthreads();
    if (tid == 0) {
        double sum = 0.0;
        for (int w = 0; w < 8; w++) sum += red[w];
        powers[(size_t)clip * frames + frame] = sum;
    }
}

std::vector<double> us_twiddle_table(int fft_size) {
    std::vector<double> t((size_t)2 * fft_size);             // fft/2 (cos, -sin) pairs, then the fft window coefficients
    for (int j = 0; j < fft_size / 2; j++) {
        const double a = 6.283185307179586476925286766559 * (double)j / (double)fft_size;
        t[2 * j] = std::cos(a); t[2 * j + 1] = -std::sin(a);
    }
    const double hw = 6.283185307179586476925286766559 / (double)(fft_size - 1);
    for (int i = 0; i < fft_size; i++) t[(size_t)fft_size + i] = 0.5 * (1.0 - std::cos(hw * (double)i));   // filter.go:139-145
    return t;
}
void launch_us_frame_power(const void* samples, int pcm16, int n_clips, int n, int fft_size, int hop, int frames, int split_bin,
                           const double* d_tw, double* powers, hipStream_t s) {
    const double2* tw = reinterpret_cast<const double2*>(d_tw);
    static const bool no8 = getenv("BNHIP_US_RADIX2") != nullptr;
    if (fft_size == 8192 && !no8) {
        const size_t lds8 = (size_t)2 * US8_PHYS(US8_N2) * sizeof(double);
        if (pcm16) {
            lds_limit_once<&k_us_frame_power8<int16_t>>(80 * 1024);
            hipLaunchKernelGGL(k_us_frame_power8<int16_t>, dim3(frames, n_clips), dim3(512), lds8, s, static_cast<const int16_t*>(samples), n, hop,
                               frames, split_bin, tw, powers);
        } else {
            lds_limit_once<&k_us_frame_power8<double>>(80 * 1024);
            hipLaunchKernelGGL(k_us_frame_power8<double>, dim3(frames, n_clips), dim3(512), lds8, s, static_cast<const double*>(samples), n, hop,
                               frames, split_bin, tw, powers);
        }
        return;
    }
    int log2n = 0; while ((1 << log2n) < fft_size) log2n++;
    size_t lds = (size_t)fft_size * 2 * sizeof(double);
    int threads = fft_size / 2 < 1024 ? (fft_size / 2 < 64 ? 64 : fft_size / 2) : 1024;
    if (pcm16) {
        lds_limit_once<&k_us_frame_power<int16_t>>(160 * 1024 - 256);
        hipLaunchKernelGGL(k_us_frame_power<int16_t>, dim3(frames, n_clips), dim3(threads), lds, s, static_cast<const int16_t*>(samples), n,
                           fft_size, hop, frames, split_bin, log2n, tw, powers);
    } else {
        lds_limit_once<&k_us_frame_power<double>>(160 * 1024 - 256);
        hipLaunchKernelGGL(k_us_frame_power<double>, dim3(frames, n_clips), dim3(threads), lds, s, static_cast<const double*>(samples), n,
                           fft_size, hop, frames, split_bin, log2n, tw, powers);
    }
}
// filter.go:76-97, sequential like the Go loop
__global__ void k_us_cv(const double* __restrict__ powers, int frames, double* __restrict__ cv) {
    if (threadIdx.x != 0) return;
    const double* p = powers + (size_t)blockIdx.x * frames;
    double n = (double)frames, sum = 0.0;
    if (frames < 2) { cv[blockIdx.x] = 0.0; return; }
    for (int i = 0; i < frames; i++) sum += p[i];
    double mean = sum / n;
    if (mean <= 0.0) { cv[blockIdx.x] = 0.0; return; }
    double sq = 0.0;
    for (int i = 0; i < frames; i++) { double d = p[i] - mean; sq += d * d; }
    cv[blockIdx.x] = sqrt(sq / n) / mean;
}
void launch_us_cv(const double* powers, int n_clips, int frames, double* cv, hipStream_t s) {
    hipLaunchKernelGGL(k_us_cv, dim3(n_clips), dim3(64), 0, s, powers, frames, cv);
}

}  // namespace bnhip
