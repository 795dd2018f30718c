// FFT-based mel front-end for gfx950: normalise -> frame -> window -> real FFT (fp64, like TFLite's RFFT2D which runs
// Ooura fft2d on doubles) -> needed bins as fp32 (real part for the CAST graph, magnitude for the COMPLEX_ABS graph)
// -> [k_pw_gemm with the mel matrix] -> pow/pow -> NHWC store.
//
// Reference path being replaced: the in-graph front-end of the v2.4 .tflite the reference feeds its interpreter
// (`internal/inference/tflite/classifier.go:95-119`); op semantics restated from TFLite 2.17.1 (rfft2d.cc, cast.cc,
// complex_support.cc).  Unlike the folded-GEMM kernel (k_frontend) this path is not restricted to the real-part graph:
// the magnitude is non-linear, so the DFT itself has to be produced.
//
// One wave transforms one frame at a time.  A length-N real frame is packed into N/2 complex points z[n] = x[2n] +
// i x[2n+1]; the N/2-point complex FFT is a four-step decomposition N/2 = P x 64 (P = 16 / 8 / 4 for N = 2048 / 1024 / 512):
//   1. lane n2 holds z[64 n1 + n2], n1 = 0..P-1: P-point FFT in registers             -> Y[k1][n2]
//   2. twiddle W_{N/2}^{k1 n2} (LDS table), transpose through the wave's LDS slice
//   3. 64-point FFTs over n2 for each k1, themselves P x Q (Q = 64/P): lane (k1, q) runs a P-point FFT in registers
//      over n2 = Q j + q, twiddles W_64^{q j'}, exchanges through LDS, then Q-point FFTs in registers
//   4. Z[k1 + P (j' + P q')] back to LDS, and the real-input split X[k] = E + W_N^k O on the bins the mel matrix uses.
// All LDS traffic of a frame stays inside the wave's private 16 KB slice, so no block barriers are needed in the loop.
#include "kernels.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>
#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace bnhip {

// ------------------------------------------------------------------------------------------ normalise
// ((x - min) / (range + eps) - sub) * mul in the graph's op order, once per clip (the FFT kernel re-reads every sample
// ~7 times through L2 because frames overlap; normalising on the fly would repeat the division each time)
__global__ void k_normalize(const float* __restrict__ x, const float2* __restrict__ mm, float* __restrict__ out,
                            int n_samples, float norm_sub, float norm_mul) {
    const int b = blockIdx.y;
    const float2 m = mm[b];
    const float* xc = x + (size_t)b * n_samples;
    float* oc = out + (size_t)b * n_samples;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_samples; i += gridDim.x * blockDim.x) {
        float t = xc[i] - m.x;
        t = t / m.y;
        t = t - norm_sub;
        oc[i] = t * norm_mul;
    }
}
void launch_normalize(const float* x, const float2* mm, float* out, int n_clips, int n_samples, float norm_sub,
                      float norm_mul, hipStream_t s) {
    dim3 grid((n_samples + 256 * 8 - 1) / (256 * 8), n_clips);
    hipLaunchKernelGGL(k_normalize, grid, dim3(256), 0, s, x, mm, out, n_samples, norm_sub, norm_mul);
}

// ------------------------------------------------------------------------------------------ register FFTs
// In-place radix-2 DIT on P complex doubles held in registers; every index is a compile-time constant after unrolling.
template <int P, typename R>
__device__ __forceinline__ void fft_regs(R (&re)[P], R (&im)[P]) {
    // cos / sin of 2 pi t / 16, t = 0..7 (P <= 16 uses a stride into this table)
    constexpr double C16[8] = {1.0, 0.92387953251128673848, 0.70710678118654752440, 0.38268343236508977173,
                               0.0, -0.38268343236508977173, -0.70710678118654752440, -0.92387953251128673848};
    constexpr double S16[8] = {0.0, 0.38268343236508977173, 0.70710678118654752440, 0.92387953251128673848,
                               1.0, 0.92387953251128673848, 0.70710678118654752440, 0.38268343236508977173};
    constexpr int LOG = P == 16 ? 4 : (P == 8 ? 3 : (P == 4 ? 2 : 1));
#pragma unroll
    for (int i = 0; i < P; i++) {
        int j = 0;
#pragma unroll
        for (int bit = 0; bit < LOG; bit++) j |= ((i >> bit) & 1) << (LOG - 1 - bit);
        if (i < j) { R t = re[i]; re[i] = re[j]; re[j] = t; t = im[i]; im[i] = im[j]; im[j] = t; }
    }
#pragma unroll
    for (int len = 2; len <= P; len <<= 1) {
        const int half = len >> 1, step = 16 / len;
#pragma unroll
        for (int i = 0; i < P; i += len) {
#pragma unroll
            for (int k = 0; k < half; k++) {
                const R wr = (R)C16[k * step], wi = (R)-S16[k * step];      // e^{-2 pi i k / len}
                const R xr = re[i + k + half], xi = im[i + k + half];
                const R ur = re[i + k], ui = im[i + k];
                if (k == 0) {                                               // w = 1
                    re[i + k] = ur + xr; im[i + k] = ui + xi; re[i + k + half] = ur - xr; im[i + k + half] = ui - xi;
                } else if (2 * k == half) {                                 // w = -i
                    re[i + k] = ur + xi; im[i + k] = ui - xr; re[i + k + half] = ur - xi; im[i + k + half] = ui + xr;
                } else {                                                    // explicit FMAs (contraction is off file-wide)
                    re[i + k] = fma(xr, wr, fma(-xi, wi, ur));
                    im[i + k] = fma(xr, wi, fma(xi, wr, ui));
                    re[i + k + half] = fma(-xr, wr, fma(xi, wi, ur));
                    im[i + k + half] = fma(-xr, wi, fma(-xi, wr, ui));
                }
            }
        }
    }
}

__device__ __forceinline__ float mel_pow(float v, float p1, float p2);
__device__ __forceinline__ float mel_log(float v, float lfloor, float lscale);

// ------------------------------------------------------------------------------------------ STFT -> needed bins
// W = waves per block.  The 2048-point kernel (P = 16: 214 VGPRs, 16.6 KB of LDS per wave) holds two waves per SIMD
// either way; the 1024-point one (146 VGPRs, 8.4 KB) fits three when the blocks are 4 waves (three blocks per CU).
// MEL: fused epilogue - the frame's bins go to the wave's LDS slice instead of HBM, each lane sums (at most) two bands of the
// banded mel matrix over them (a narrow and a wide one: lane l takes bands l and n_mels - 1 - l, so every lane walks about
// the same number of quads), applies the compression and stores the values into the spectrogram image.  Same products in
// the same order as k_mel_banded (aligned quads of four bins, ascending): bit-identical, minus a kernel launch and the
// bins' round trip through HBM (0.9 MB per clip).
// R: the transform's arithmetic.  fp64 is what TFLite's RFFT2D computes in and what fp32 parity on near-empty bins needs (see the
// file header); float serves the "precision":"bf16" engines, whose every later tensor is rounded to 8 significant bits anyway -
// half the registers and LDS per wave, twice the VALU rate.
template <int P, int W, bool MEL = false, typename R = double>
__global__ __launch_bounds__(64 * W) void k_stft_bins(StftParams p) {
    constexpr int N2 = 64 * P;            // complex points
    constexpr int N = 2 * N2;             // real frame length (= fft length)
    constexpr int Q = 64 / P;
    constexpr int RS = P == 8 ? 66 : 65;  // padded row stride of the transposed Y[k1][n2]: the (k1, q) read pattern of
                                          // stage 3 then spreads evenly over the bank pairs (65 left P = 8 at 2x the minimum)
    constexpr int WSZ = P * RS;           // doubles per component in a wave's slice (>= N2)
    constexpr int NPAIR = (P * P + 63) / 64;   // (k1, j') pairs per lane in the last stage (P = 4: only 16 lanes hold one)
    extern __shared__ __attribute__((aligned(16))) unsigned char stft_sm[];
    R* sm = reinterpret_cast<R*>(stft_sm);
    R* twr = sm;                     // [P][64] Re W_{N2}^{k1 n2}
    R* twi = twr + P * 64;
    R* t2r = twi + P * 64;           // [Q][P]  W_64^{q j'}
    R* t2i = t2r + Q * P;
    R* work = t2i + Q * P;           // [W][2][WSZ]
    float4* melw4 = reinterpret_cast<float4*>(work + (size_t)W * 2 * WSZ);     // MEL: band weights, p.mel_quads float4s ...
    const int* binq = reinterpret_cast<const int*>(melw4 + p.mel_quads);        // ... and the bins quad each of them multiplies
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;

    // ---- tables (once per block): copied from the plan-time image (stft_build_tables); computing them here cost ten
    // fp64 sincos per thread and block - a sixth of the kernel at 16 frames per wave
    constexpr int NTAB = 2 * P * 64 + 2 * Q * P;
    for (int i = tid; i < NTAB; i += blockDim.x) sm[i] = (R)p.tw[i];
    int mA = 0, loA = 0, nA = 0, wA = 0, mB = 0, loB = 0, nB = 0, wB = 0;
    if (MEL) {
        const float4* src = reinterpret_cast<const float4*>(p.mel + 64 * 8);
        // (weights and the quad index list: mel_quads float4s + mel_quads ints rounded up to whole float4s)
        for (int i = tid; i < p.mel_quads + (p.mel_quads + 3) / 4; i += blockDim.x) melw4[i] = src[i];
        const int4* lt = reinterpret_cast<const int4*>(p.mel) + 2 * lane;
        const int4 ta = lt[0], tb = lt[1];
        mA = ta.x; loA = ta.y; nA = ta.z; wA = ta.w; mB = tb.x; loB = tb.y; nB = tb.z; wB = tb.w;
    }
    __syncthreads();

    R* wre = work + (size_t)wave * 2 * WSZ;
    R* wim = wre + WSZ;
    const float* xc = p.xn + (size_t)b * p.n_samples;
    // this lane's window taps: samples 128 n1 + 2 n2 (+1)
    float w0[P], w1[P];
#pragma unroll
    for (int n1 = 0; n1 < P; n1++) {
        int n = 128 * n1 + 2 * lane;
        w0[n1] = n < p.L ? p.window[n] : 0.f;
        w1[n1] = n + 1 < p.L ? p.window[n + 1] : 0.f;
    }
    const int k1b = lane % P, qb = lane / P;
    // this lane's output bins (constant across frames): LDS indices of Z[k], Z[N2-k]
    constexpr int NBL = 8;                               // bins per lane held in registers (nb <= 512)
    int ka_[NBL], kb_[NBL];
    R trr[NBL], tri[NBL];                           // W_N^k of those bins (registers: keeps the LDS for a second block)
#pragma unroll
    for (int t = 0; t < NBL; t++) {
        int idx = lane + 64 * t;
        int k = idx < p.nb ? p.bins[idx] : 0;
        ka_[t] = k % N2; kb_[t] = (N2 - k % N2) % N2;
        const bool in = idx < p.nb_cap;
        trr[t] = in ? (R)p.tw[NTAB + idx] : (R)1.0;
        tri[t] = in ? (R)p.tw[NTAB + p.nb_cap + idx] : (R)0.0;
    }

    const int f_begin = (blockIdx.x * W + wave) * p.fpw;
    const int f_end = min(f_begin + p.fpw, p.F);
    // the next frame's samples are requested before the current frame is transformed (a frame's 16 loads would
    // otherwise be an exposed L2 round trip per frame with two waves per SIMD)
    float2 nx[P];
    auto fetch = [&](int f) {
        // chunks of 128 samples that lie wholly beyond the frame (frame shorter than the transform) are not loaded: their
        // window taps are zero
        const int first = f * p.hop - p.pad_left;
        const int s0 = first + 2 * lane;
        if (first >= 0 && first + N <= p.n_samples && (first & 1) == 0) {        // whole transform window in range, 8-byte aligned
#pragma unroll
            for (int n1 = 0; n1 < P; n1++)
                nx[n1] = 128 * n1 < p.L ? *reinterpret_cast<const float2*>(xc + s0 + 128 * n1) : make_float2(0.f, 0.f);
        } else {
#pragma unroll
            for (int n1 = 0; n1 < P; n1++) {
                int s = s0 + 128 * n1;
                float2 v = make_float2(0.f, 0.f);
                if (128 * n1 < p.L) {
                    if (s >= 0 && s < p.n_samples) v.x = xc[s];
                    if (s + 1 >= 0 && s + 1 < p.n_samples) v.y = xc[s + 1];
                }
                nx[n1] = v;
            }
        }
    };
    if (f_begin < f_end) fetch(f_begin);
    for (int f = f_begin; f < f_end; f++) {
        R re[P], im[P];
        // ---- 1. window (fp32 product, as the graph's MUL) + P-point FFT over n1
#pragma unroll
        for (int n1 = 0; n1 < P; n1++) {
            re[n1] = (R)(nx[n1].x * w0[n1]);
            im[n1] = (R)(nx[n1].y * w1[n1]);
        }
        if (f + 1 < f_end) fetch(f + 1);
        fft_regs<P, R>(re, im);
        // ---- 2. twiddle + transpose
#pragma unroll
        for (int k1 = 0; k1 < P; k1++) {
            const R c = twr[k1 * 64 + lane], s = twi[k1 * 64 + lane];
            const R r = fma(re[k1], c, -(im[k1] * s)), i2 = fma(re[k1], s, im[k1] * c);
            wre[k1 * RS + lane] = r; wim[k1 * RS + lane] = i2;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < P; j++) { re[j] = wre[k1b * RS + Q * j + qb]; im[j] = wim[k1b * RS + Q * j + qb]; }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        // ---- 3. P-point FFT over j, twiddle W_64^{q j'}, exchange, Q-point FFTs
        fft_regs<P, R>(re, im);
#pragma unroll
        for (int j = 0; j < P; j++) {
            const R c = t2r[qb * P + j], s = t2i[qb * P + j];
            const R r = fma(re[j], c, -(im[j] * s)), i2 = fma(re[j], s, im[j] * c);
            const int slot = j * P + k1b;                 // [q][slot] layout: both sides of the exchange are conflict-free
            wre[qb * (P * P) + slot] = r; wim[qb * (P * P) + slot] = i2;   // ([slot][q] made the reads below 16-way bank conflicts)
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        R ar[NPAIR][Q], ai[NPAIR][Q];
#pragma unroll
        for (int t = 0; t < NPAIR; t++) {
            const int slot = min(lane + 64 * t, P * P - 1);
#pragma unroll
            for (int q = 0; q < Q; q++) { ar[t][q] = wre[q * (P * P) + slot]; ai[t][q] = wim[q * (P * P) + slot]; }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < NPAIR; t++) {
            const int slot = lane + 64 * t;              // = j' P + k1  ->  k = k1 + P j' + P^2 q'
            const int kbase = (slot % P) + P * (slot / P);
            // Output pruning (2048-point transform, Q = 4): the mel bank reads a fraction of the bins (v2.4: 129 of 1 025), so
            // most Z[k1 + P j' + P^2 q'] are never read - pass t of this stage only computes and stores the q' some needed bin
            // reads (p.zmask, wave-uniform, from the plan).  Each kept output is formed by the same additions in the same order
            // as the full radix-2 transform below: same bits.
            const unsigned zm = Q == 4 ? (p.zmask >> (4 * t)) & 15u : 15u;
            if (Q == 4 && zm != 15u) {
                if (slot < P * P) {
                    if (zm & 5u) {                       // q' = 0, 2: (x0 + x2) +- (x1 + x3)
                        const R a_r = ar[t][0] + ar[t][2], a_i = ai[t][0] + ai[t][2], c_r = ar[t][1] + ar[t][3], c_i = ai[t][1] + ai[t][3];
                        if (zm & 1u) { wre[kbase] = a_r + c_r; wim[kbase] = a_i + c_i; }
                        if (zm & 4u) { wre[kbase + 2 * P * P] = a_r - c_r; wim[kbase + 2 * P * P] = a_i - c_i; }
                    }
                    if (zm & 10u) {                      // q' = 1, 3: (x0 - x2) -+ i (x1 - x3)
                        const R b_r = ar[t][0] - ar[t][2], b_i = ai[t][0] - ai[t][2], d_r = ar[t][1] - ar[t][3], d_i = ai[t][1] - ai[t][3];
                        if (zm & 2u) { wre[kbase + P * P] = b_r + d_i; wim[kbase + P * P] = b_i - d_r; }
                        if (zm & 8u) { wre[kbase + 3 * P * P] = b_r - d_i; wim[kbase + 3 * P * P] = b_i + d_r; }
                    }
                }
                continue;
            }
            fft_regs<Q, R>(ar[t], ai[t]);
            if (slot < P * P) {
#pragma unroll
                for (int q = 0; q < Q; q++) { wre[kbase + P * P * q] = ar[t][q]; wim[kbase + P * P * q] = ai[t][q]; }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        // ---- 4. real-input split on the needed bins
        float* orow = MEL ? nullptr : p.out + ((size_t)b * p.F + f) * p.nbp;
        float ov[NBL];
#pragma unroll
        for (int t = 0; t < NBL; t++) {
            const int idx = lane + 64 * t;
            ov[t] = 0.f;
            if (idx >= p.nbp) break;
            float o = 0.f;
            if (idx < p.nb) {
                const int ka = ka_[t], kb = kb_[t];
                const R zr = wre[ka], zi = wim[ka], mr = wre[kb], mi = -wim[kb];       // Z[k], conj(Z[N2-k])
                const R er = (R)0.5 * (zr + mr), ei = (R)0.5 * (zi + mi);
                const R dr = zr - mr, di = zi - mi;
                const R or_ = (R)0.5 * di, oi = (R)-0.5 * dr;                                 // -i/2 (Z[k] - conj(Z[N2-k]))
                const R c = trr[t], s = tri[t];
                const R xr = er + fma(or_, c, -(oi * s));
                if (p.mode == 0) o = (float)xr;
                else {
                    const R xi = ei + fma(or_, s, oi * c);
                    o = hypotf((float)xr, (float)xi);        // COMPLEX_ABS on complex64
                }
            }
            if (MEL) ov[t] = o; else orow[idx] = o;
        }
        if (MEL) {
            // every lane has read what it needs of Z: the slice now holds the frame's bins as one float row
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            float* brow = reinterpret_cast<float*>(wre);
#pragma unroll
            for (int t = 0; t < NBL; t++) { const int idx = lane + 64 * t; if (idx < p.nbp) brow[idx] = ov[t]; }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            const float4* b4 = reinterpret_cast<const float4*>(brow);
            // phase a: the (band, quad) products spread evenly over the wave - ~2 K / 4 + n_mels quads per frame, 2-4 per lane
            // (a band per lane would make every frame wait for the widest band: measured +53 % on the 1024-point kernel)
            float* part = brow + 512;
            for (int e = lane; e < p.mel_quads; e += 64) {
                const float4 x4 = b4[binq[e]], w4 = melw4[e];
                part[e] = fmaf(x4.w, w4.w, fmaf(x4.z, w4.z, fmaf(x4.y, w4.y, x4.x * w4.x)));
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            // phase b: each band's quad sums added up in ascending order by the band's lane, four reads in flight
            (void)loA; (void)loB;
#pragma unroll 1
            for (int which = 0; which < 2; which++) {
                const int mm = which ? mB : mA, n4 = which ? nB : nA, wo = which ? wB : wA;
                if (n4 < 0) continue;                          // (no band in this slot; an EMPTY band still writes its pixel)
                float acc = 0.f;
                for (int g = 0; g < n4; g += 4) {
                    const float t0 = part[wo + g], t1 = g + 1 < n4 ? part[wo + g + 1] : 0.f, t2 = g + 2 < n4 ? part[wo + g + 2] : 0.f,
                                t3 = g + 3 < n4 ? part[wo + g + 3] : 0.f;
                    acc = (((acc + t0) + t1) + t2) + t3;
                }
                const float v = p.logc ? mel_log(acc, p.lfloor, p.lscale) : mel_pow(acc, p.p1, p.p2);
                const size_t o = p.time_major ? (((size_t)b * p.F + f) * p.n_mels + mm) * p.Ctot + p.c0
                                              : (((size_t)b * p.n_mels + mm) * p.F + f) * p.Ctot + p.c0;
                p.img[o] = v;
            }
        }
        // more than 512 needed bins (a 2048-point transform under a wide mel bank): the rest take their bin index and twiddle
        // from the plan-time image each frame instead of from registers
        for (int idx = lane + 64 * NBL; !MEL && idx < p.nbp; idx += 64) {
            float o = 0.f;
            if (idx < p.nb) {
                const int k = p.bins[idx];
                const int ka = k % N2, kb = (N2 - k % N2) % N2;
                const R zr = wre[ka], zi = wim[ka], mr = wre[kb], mi = -wim[kb];
                const R er = (R)0.5 * (zr + mr), ei = (R)0.5 * (zi + mi);
                const R dr = zr - mr, di = zi - mi;
                const R or_ = (R)0.5 * di, oi = (R)-0.5 * dr;
                const R c = (R)p.tw[NTAB + idx], s = (R)p.tw[NTAB + p.nb_cap + idx];
                const R xr = er + fma(or_, c, -(oi * s));
                if (p.mode == 0) o = (float)xr;
                else {
                    const R xi = ei + fma(or_, s, oi * c);
                    o = hypotf((float)xr, (float)xi);
                }
            }
            orow[idx] = o;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
}

// Plan-time twiddle image of one STFT step: [Re W_{N/2}^{k1 n2} (P x 64)][Im ..][Re W_64^{q j} (Q x P)][Im ..]
// [Re W_N^k of the needed bins, padded to a multiple of 64 with W^0][Im ..] - the kernel's LDS table layout followed by the
// per-lane bin twiddles.
std::vector<double> stft_build_tables(int Lfft, const int* bins, int nb) {
    const int P = Lfft / 128, Q = 64 / P, N2 = 64 * P, cap = (nb + 63) / 64 * 64;
    const double tau = -6.283185307179586476925286766559;
    std::vector<double> t((size_t)2 * P * 64 + 2 * Q * P + 2 * cap);
    double* twr = t.data();
    double* twi = twr + P * 64;
    double* t2r = twi + P * 64;
    double* t2i = t2r + Q * P;
    double* trr = t2i + Q * P;
    double* tri = trr + cap;
    for (int i = 0; i < P * 64; i++) {
        const double a = tau * (double)(((i >> 6) * (i & 63)) % N2) / (double)N2;
        twr[i] = std::cos(a); twi[i] = std::sin(a);
    }
    for (int i = 0; i < Q * P; i++) {
        const double a = tau * (double)(((i / P) * (i % P)) % 64) / 64.0;
        t2r[i] = std::cos(a); t2i[i] = std::sin(a);
    }
    for (int i = 0; i < cap; i++) {
        const double a = tau * (double)(i < nb ? bins[i] : 0) / (double)Lfft;
        trr[i] = std::cos(a); tri[i] = std::sin(a);
    }
    return t;
}
// Which outputs of the closing Q-point transforms does the real-input split of the needed bins read?  Bin k reads Z[k mod N2] and
// Z[(N2 - k) mod N2]; Z[k1 + P j' + P^2 q'] is produced in pass t = (j' P + k1) / 64 as output q'.  Four bits per pass; only the
// Q = 4 kernel (2048-point frames) prunes, everything else gets all ones.  BNHIP_STFT_PRUNE=0: all ones (A/B runs).
unsigned stft_zmask(int Lfft, const int* bins, int nb) {
    const char* e = getenv("BNHIP_STFT_PRUNE");           // (plan time only - never on the launch path)
    const bool off = e && atoi(e) == 0;
    const int P = Lfft / 128;
    if (P != 16 || off || nb <= 0) return 0xffffffffu;
    const int N2 = 64 * P;
    unsigned m = 0;
    auto need = [&](int z) { const int qp = z / (P * P), slot = ((z % (P * P)) / P) * P + z % P; m |= 1u << (4 * (slot / 64) + qp); };
    for (int i = 0; i < nb; i++) { const int k = bins[i] % N2; need(k); need((N2 - k) % N2); }
    return m | 0xffff0000u;
}
static int stft_waves(int P) { return P == 8 ? 4 : 8; }
size_t stft_lds_bytes(int P, int nb_cap) {
    (void)nb_cap;
    return (size_t)(2 * P * 64 + 2 * (64 / P) * P + stft_waves(P) * 2 * P * (P == 8 ? 66 : 65)) * sizeof(double);
}
bool stft_supported(int Lfft, int nb) {
    if (Lfft != 2048 && Lfft != 1024 && Lfft != 512) return false;
    int P = Lfft / 128, cap = (nb + 63) / 64 * 64;
    return nb <= Lfft / 2 + 1 && stft_lds_bytes(P, cap) <= 160 * 1024;      // (the first 512 bins' twiddles live in registers)
}
std::vector<float> stft_mel_table(int Lfft, const float* melw, const int* span, int n_mels, int nb, int nbp, int* quads) {
    *quads = 0;
    if (n_mels < 1 || n_mels > 128 || nbp > 512 || nb > nbp || (nbp & 3)) return {};
    std::vector<int> lo4(n_mels), n4(n_mels), wo(n_mels);
    int tot = 0;
    for (int m = 0; m < n_mels; m++) {
        const int lo = span[2 * m], hi = span[2 * m + 1];
        lo4[m] = lo >> 2; n4[m] = hi > lo ? ((hi + 3) >> 2) - (lo >> 2) : 0; wo[m] = tot;
        tot += n4[m];
    }
    const int P = Lfft / 128;
    if (stft_lds_bytes(P, 0) + (size_t)tot * 16 + (size_t)(tot + 3) / 4 * 16 > 160 * 1024) return {};
    if (nbp > 512 || tot > 256) return {};                       // (bins row and quad sums share the wave's LDS slice; four quads per lane)
    std::vector<float> t((size_t)64 * 8 + (size_t)tot * 4 + (size_t)(tot + 3) / 4 * 4, 0.f);
    int* lt = reinterpret_cast<int*>(t.data());
    const int half = (n_mels + 1) / 2;
    for (int l = 0; l < 64; l++) {
        int* e = lt + 8 * l;
        e[2] = e[6] = -1;                                      // quads < 0: no band in this slot
        if (l >= half) continue;
        const int a = l, bnd = n_mels - 1 - l;
        e[0] = a; e[1] = lo4[a]; e[2] = n4[a]; e[3] = wo[a];
        if (bnd != a) { e[4] = bnd; e[5] = lo4[bnd]; e[6] = n4[bnd]; e[7] = wo[bnd]; }
    }
    for (int m = 0; m < n_mels; m++)
        for (int g = 0; g < n4[m]; g++)
            for (int q = 0; q < 4; q++) {
                const int k = 4 * (lo4[m] + g) + q;
                t[(size_t)64 * 8 + 4 * ((size_t)wo[m] + g) + q] = k < nbp ? melw[(size_t)m * nbp + k] : 0.f;
            }
    int* bq = reinterpret_cast<int*>(t.data() + (size_t)64 * 8 + (size_t)tot * 4);
    for (int m = 0; m < n_mels; m++)
        for (int g = 0; g < n4[m]; g++) bq[wo[m] + g] = lo4[m] + g;
    *quads = tot;
    return t;
}
void launch_stft_bins(const StftParams& p0, hipStream_t s) {
    StftParams p = p0;
    p.nb_cap = (p.nb + 63) / 64 * 64;
    // frames per wave: 16 amortises the per-block table copy at batch size; small batches trade that for parallelism
    // (one clip with 16 frames per wave would occupy 4 of the 256 CUs)
    static int fpw_env = getenv("BNHIP_STFT_FPW") ? atoi(getenv("BNHIP_STFT_FPW")) : 0;
    long waves_wanted = 2048;
    long fpw = fpw_env > 0 ? fpw_env : ((long)p.F * p.n_clips + waves_wanted - 1) / waves_wanted;
    p.fpw = (int)std::min<long>(std::max<long>(fpw, 1), 16);
    const int P = p.Lfft / 128, W = stft_waves(P);
    size_t lds = stft_lds_bytes(P, p.nb_cap);
    dim3 grid((p.F + W * p.fpw - 1) / (W * p.fpw), p.n_clips);
    // (the dynamic-LDS limit is an attribute of the function ON THE CURRENT DEVICE: set per launch - the process-wide once-flags that stood here
    // left every further device of a multi-device handle at the 64 KB default, and were plain bools written by its worker threads)
    if (p.mel) {
        lds += (size_t)p.mel_quads * 16 + (size_t)(p.mel_quads + 3) / 4 * 16;
        if (P == 16) {
            lds_limit_once<&k_stft_bins<16, 8, true>>(160 * 1024);
            hipLaunchKernelGGL((k_stft_bins<16, 8, true>), grid, dim3(64 * W), lds, s, p);
        } else if (P == 8) {
            lds_limit_once<&k_stft_bins<8, 4, true>>(160 * 1024);
            hipLaunchKernelGGL((k_stft_bins<8, 4, true>), grid, dim3(64 * W), lds, s, p);
        } else {
            lds_limit_once<&k_stft_bins<4, 8, true>>(160 * 1024);
            hipLaunchKernelGGL((k_stft_bins<4, 8, true>), grid, dim3(64 * W), lds, s, p);
        }
        return;
    }
    if (p.f32) {                                          // "precision":"bf16" engines: the transform in fp32 (half the LDS)
        lds /= 2;
        if (P == 4) hipLaunchKernelGGL((k_stft_bins<4, 8, false, float>), grid, dim3(64 * W), lds, s, p);
        else if (P == 16) {
            lds_limit_once<&k_stft_bins<16, 8, false, float>>(160 * 1024);
            hipLaunchKernelGGL((k_stft_bins<16, 8, false, float>), grid, dim3(64 * W), lds, s, p);
        } else hipLaunchKernelGGL((k_stft_bins<8, 4, false, float>), grid, dim3(64 * W), lds, s, p);
        return;
    }
    if (P == 4) {
        hipLaunchKernelGGL((k_stft_bins<4, 8>), grid, dim3(64 * W), lds, s, p);      // < 64 KB of LDS
    } else if (P == 16) {
        lds_limit_once<&k_stft_bins<16, 8>>(160 * 1024);
        hipLaunchKernelGGL((k_stft_bins<16, 8>), grid, dim3(64 * W), lds, s, p);
    } else {
        hipLaunchKernelGGL((k_stft_bins<8, 4>), grid, dim3(64 * W), lds, s, p);      // 42 KB of LDS
    }
}

// ------------------------------------------------------------------------------------------ pow + NHWC store
// T_c[b][f][m] (mel GEMM outputs, one per channel) -> out[b][m][f][c] = pow(pow(v, p1), p2); a 32 x 32 LDS transpose keeps
// both sides coalesced and all channels of a pixel are written together.  The second power runs on the hardware
// log2/exp2 units (1 ulp each): with p1 = 2 the operand is a square, so there is no sign to carry.
__device__ __forceinline__ float mel_pow(float v, float p1, float p2) {
    float y = (p1 == 2.0f) ? v * v : powf(v, p1);
    if (p2 == 1.0f) return y;
    if (y >= 0.0f) return y == 0.0f ? 0.0f : __builtin_amdgcn_exp2f(p2 * __builtin_amdgcn_logf(y));
    return powf(y, p2);
}
// log compression of the Perch-style front-end: MAXIMUM, LOG, MUL as the graph orders them (v_log_f32 is log2, 1 ulp)
__device__ __forceinline__ float mel_log(float v, float lfloor, float lscale) {
    return lscale * (0.6931471805599453f * __builtin_amdgcn_logf(fmaxf(v, lfloor)));
}
template <int C>
__global__ __launch_bounds__(256) void k_mel_finish(MelFinParams p) {
    __shared__ float tile[C][32][33];
    const int b = blockIdx.z, f0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int c = 0; c < C; c++)
        for (int r = ty; r < 32; r += 8) {
            int f = f0 + r, m = m0 + tx;
            float v = 0.f;
            if (f < p.F && m < p.n_mels) {
                const float t = p.T[c][((size_t)b * p.F + f) * p.ldt + m];
                v = p.log ? mel_log(t, p.lfloor, p.lscale) : mel_pow(t, p.p1[c], p.p2[c]);
            }
            tile[c][r][tx] = v;
        }
    __syncthreads();
    if (p.time_major) {                                  // [B, F, n_mels, Ctot]: no transpose, mel index fastest
        for (int r = ty; r < 32; r += 8) {
            int f = f0 + r, m = m0 + tx;
            if (f < p.F && m < p.n_mels) {
                float* o = p.out + (((size_t)b * p.F + f) * p.n_mels + m) * p.Ctot + p.c0;
                if (C == 2) *reinterpret_cast<float2*>(o) = make_float2(tile[0][r][tx], tile[1][r][tx]);
                else o[0] = tile[0][r][tx];
            }
        }
        return;
    }
    for (int r = ty; r < 32; r += 8) {
        int m = m0 + r, f = f0 + tx;
        if (f < p.F && m < p.n_mels) {
            float* o = p.out + (((size_t)b * p.n_mels + m) * p.F + f) * p.Ctot + p.c0;
            if (C == 2) *reinterpret_cast<float2*>(o) = make_float2(tile[0][tx][r], tile[1][tx][r]);
            else o[0] = tile[0][tx][r];
        }
    }
}
// ------------------------------------------------------------------------------------------ banded mel + pow + NHWC store
// The mel matrix is a bank of triangular filters: every band touches a short contiguous run of DFT bins (2 K nonzeros in
// total, not K x n_mels).  As a dense GEMM it was 0.13 ms of MFMA time per batch for 98 % zero products; as a banded sum
// it is a bandwidth-class kernel, and with both channels in one launch the mel values never go to HBM either:
//   out[b][m][f][c] = pow(pow(sum_{k in [lo_m, hi_m)} bins_c[b][f][k] * w_c[m][k], p1_c), p2_c).
// Block = MSP_F = 8 consecutive frames of one clip (their bin rows are one contiguous chunk, staged in LDS), 192 threads = 96
// bands x 2 frame groups of 4 (16 frames per block measured 116 us, 8: 104 us, 4: 132 us: the kernel is three serial phases -
// stage, sum, store - and waits 70 % of its life, so smaller blocks = more of them per CU in different phases win until the
// weight rows are re-read too often).  A thread walks its band in aligned groups of four columns: one float4 of the (dense, zero
// outside the band) weight row from L1/L2 and one ds_read_b128 per frame - LDS instruction issue is what bounds the
// kernel (PMC/A-B: scalar reads with the weights in LDS were 10 % slower than with the weights in global memory), so
// the wide read is the lever.  Products are added in ascending k with fmaf (the GEMM summed them in its slab order).
// Bands are dealt to waves in order (wave w: bands 32 w ..), so the wide top bands do not set the trip count of every wave.
#define MSP_F 8
template <int C, int VAR = 0>
__global__ __launch_bounds__(256) void k_mel_banded(MelBandParams p) {
    extern __shared__ __attribute__((aligned(16))) float msm[];
    float* rows[2];
    rows[0] = msm;
    rows[1] = msm + (size_t)MSP_F * p.nbp[0];
    float* tile = rows[C - 1] + (size_t)MSP_F * p.nbp[C - 1];          // [C][MSP_F][n_mels + 1]
    const int TS = p.n_mels + 1;
    const int b = blockIdx.y, f0 = blockIdx.x * MSP_F, nf = min(MSP_F, p.F - f0), tid = threadIdx.x, nthr = blockDim.x;
#pragma unroll
    for (int c = 0; c < C; c++) {
        const float4* src = reinterpret_cast<const float4*>(p.bins[c] + ((size_t)b * p.F + f0) * p.nbp[c]);
        float4* dst = reinterpret_cast<float4*>(rows[c]);
        const int n4 = nf * p.nbp[c] / 4, n4all = MSP_F * p.nbp[c] / 4;  // nbp is a multiple of 4
        for (int base = 0; base < n4all; base += 4 * nthr) {             // four loads in flight per thread
            float4 v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int i = base + tid + nthr * j;
                v[j] = i < n4 ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int i = base + tid + nthr * j;
                if (i < n4all) dst[i] = v[j];
            }
        }
    }
    __syncthreads();
    const int m = tid >> 1, g = tid & 1;
    if (m < p.n_mels) {
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int lo = p.span[c][2 * m], hi = p.span[c][2 * m + 1], ld = p.nbp[c];
            const float4* wr = reinterpret_cast<const float4*>(p.w[c] + (size_t)m * ld);
            const float* L = rows[c] + (size_t)(g * (MSP_F / 2)) * ld;
            float acc[MSP_F / 2];
#pragma unroll
            for (int i = 0; i < MSP_F / 2; i++) acc[i] = 0.f;
            const int k4e = (hi + 3) >> 2, k4max = ld / 4 - 1;
            for (int k4 = lo >> 2; k4 < k4e; k4 += 4) {                  // four weight quads requested at once
                float4 w4[4];
#pragma unroll
                for (int j = 0; j < 4; j++) w4[j] = wr[min(k4 + j, k4max)];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (k4 + j < k4e) {
#pragma unroll
                        for (int i = 0; i < MSP_F / 2; i++) {
                            const float4 x4 = *reinterpret_cast<const float4*>(L + (size_t)i * ld + 4 * (k4 + j));
                            if (VAR == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            if (VAR == 1) {
                                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i]) : "v"(x4.x), "v"(w4[j].x));
                                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i]) : "v"(x4.y), "v"(w4[j].y));
                                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i]) : "v"(x4.z), "v"(w4[j].z));
                                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i]) : "v"(x4.w), "v"(w4[j].w));
                            } else
                            acc[i] = fmaf(x4.w, w4[j].w, fmaf(x4.z, w4[j].z, fmaf(x4.y, w4[j].y, fmaf(x4.x, w4[j].x, acc[i]))));
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < MSP_F / 2; i++)
                tile[(c * MSP_F + g * (MSP_F / 2) + i) * TS + m] = p.log ? mel_log(acc[i], p.lfloor, p.lscale) : mel_pow(acc[i], p.p1[c], p.p2[c]);
        }
    }
    __syncthreads();
    if (p.time_major) {                                  // [B, F, n_mels, Ctot]: mel index fastest
        for (int it = tid; it < p.n_mels * MSP_F; it += nthr) {
            const int f = it / p.n_mels, mm = it - f * p.n_mels;
            if (f < nf) {
                float* o = p.out + (((size_t)b * p.F + f0 + f) * p.n_mels + mm) * p.Ctot + p.c0;
                if (C == 2) *reinterpret_cast<float2*>(o) = make_float2(tile[f * TS + mm], tile[(MSP_F + f) * TS + mm]);
                else o[0] = tile[f * TS + mm];
            }
        }
        return;
    }
    for (int it = tid; it < p.n_mels * MSP_F; it += nthr) {
        const int mm = it / MSP_F, f = it % MSP_F;
        if (f < nf) {
            float* o = p.out + (((size_t)b * p.n_mels + mm) * p.F + f0 + f) * p.Ctot + p.c0;
            if (C == 2) *reinterpret_cast<float2*>(o) = make_float2(tile[f * TS + mm], tile[(MSP_F + f) * TS + mm]);
            else o[0] = tile[f * TS + mm];
        }
    }
}
bool mel_banded_supported(int n_mels, int nbp0, int nbp1) {
    size_t lds = (size_t)MSP_F * (nbp0 + nbp1) * 4 + (size_t)2 * MSP_F * (n_mels + 1) * 4;
    return n_mels <= 128 && (nbp0 & 3) == 0 && (nbp1 & 3) == 0 && lds <= 64 * 1024;
}
void launch_mel_banded(const MelBandParams& p, int nch, int n_clips, hipStream_t s) {
    size_t lds = (size_t)MSP_F * (p.nbp[0] + (nch == 2 ? p.nbp[1] : 0)) * 4 + (size_t)nch * MSP_F * (p.n_mels + 1) * 4;
    dim3 grid((p.F + MSP_F - 1) / MSP_F, n_clips);
    const dim3 block(p.n_mels <= 96 ? 192 : 256);        // two frame groups of 8 per band
    static const int var = getenv("BNHIP_MEL_VARIANT") ? atoi(getenv("BNHIP_MEL_VARIANT")) : 0;
    if (nch == 2 && var == 1) hipLaunchKernelGGL((k_mel_banded<2, 1>), grid, block, lds, s, p);
    else if (nch == 2 && var == 2) hipLaunchKernelGGL((k_mel_banded<2, 2>), grid, block, lds, s, p);
    else if (nch == 2) hipLaunchKernelGGL((k_mel_banded<2>), grid, block, lds, s, p);
    else hipLaunchKernelGGL((k_mel_banded<1>), grid, block, lds, s, p);
}

void launch_mel_finish(const MelFinParams& p, int nch, int n_clips, hipStream_t s) {
    dim3 grid((p.F + 31) / 32, (p.n_mels + 31) / 32, n_clips);
    if (nch == 2) hipLaunchKernelGGL((k_mel_finish<2>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((k_mel_finish<1>), grid, dim3(256), 0, s, p);
}

}  // namespace bnhip
