// Generic fallback kernels: every float op of a TFLite graph that the fused BirdNET plan does not absorb still has to
// execute (the reference call being replaced accepts any float graph: internal/inference/tflite/classifier.go:38-92).
// These are plain bandwidth-bound gfx950 kernels - one thread per output element, coalesced along the innermost
// (channel) dimension, 64-bit-safe indexing - with libm-accurate math: they exist for coverage, the fused kernels in
// kernels.hip are the fast path.  Activations are [clip][per-clip elements]; views are per-clip 4-D (dims + strides in
// elements) plus a clip stride (0 for constants).
#include "kernels.h"

#include <cmath>

namespace bnhip {

__device__ __forceinline__ float g_act(float v, int act) {
    switch (act) {
        case ACT_RELU: return fmaxf(v, 0.0f);
        case ACT_RELU6: return fminf(fmaxf(v, 0.0f), 6.0f);
        case ACT_RELU_N1_TO_1: return fminf(fmaxf(v, -1.0f), 1.0f);
        case ACT_TANH: return tanhf(v);
        default: return v;
    }
}

// ------------------------------------------------------------------------------------------ unary
__global__ void k_unary_op(const float* __restrict__ in, float* __restrict__ out, size_t n, int op, float alpha) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const float x = in[i];
        float y;
        switch (op) {
            case U_ABS: y = fabsf(x); break;
            case U_SQRT: y = sqrtf(x); break;
            case U_RSQRT: y = 1.0f / sqrtf(x); break;
            case U_LOG: y = logf(x); break;
            case U_EXP: y = expf(x); break;
            case U_NEG: y = -x; break;
            case U_SQUARE: y = x * x; break;
            case U_TANH: y = tanhf(x); break;
            case U_LEAKY_RELU: y = x > 0.0f ? x : x * alpha; break;
            case U_ELU: y = x > 0.0f ? x : expm1f(x); break;
            case U_SIN: y = sinf(x); break;
            case U_COS: y = cosf(x); break;
            case U_FLOOR: y = floorf(x); break;
            case U_CEIL: y = ceilf(x); break;
            case U_ROUND: y = rintf(x); break;                                  // TFLite ROUND: half to even
            case U_RELU_N1_TO_1: y = fminf(fmaxf(x, -1.0f), 1.0f); break;
            case U_LOGISTIC: y = 1.0f / (1.0f + expf(-x)); break;
            case U_RELU: y = fmaxf(x, 0.0f); break;
            case U_RELU6: y = fminf(fmaxf(x, 0.0f), 6.0f); break;
            case U_HARD_SWISH: y = x * fminf(fmaxf(x + 3.0f, 0.0f), 6.0f) / 6.0f; break;
            case U_GELU: y = 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); break;
            case U_GELU_TANH: y = 0.5f * x * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x))); break;
            default: y = x; break;
        }
        out[i] = y;
    }
}
void launch_unary_op(const float* in, float* out, size_t n, int op, float alpha, hipStream_t s) {
    size_t blocks = (n + 255) / 256; if (blocks > 16384) blocks = 16384; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_unary_op, dim3((unsigned)blocks), dim3(256), 0, s, in, out, n, op, alpha);
}

// ------------------------------------------------------------------------------------------ broadcasting binary
__global__ void k_binary_bcast(BcastParams p, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t per = (size_t)p.d[0] * p.d[1] * p.d[2] * p.d[3];
    for (; i < total; i += stride) {
        const size_t clip = i / per;
        size_t r = i - clip * per;
        const int i3 = (int)(r % p.d[3]); r /= p.d[3];
        const int i2 = (int)(r % p.d[2]); r /= p.d[2];
        const int i1 = (int)(r % p.d[1]);
        const int i0 = (int)(r / p.d[1]);
        const float x = p.a[clip * p.bsa + i0 * p.sa[0] + i1 * p.sa[1] + i2 * p.sa[2] + i3 * p.sa[3]];
        const float y = p.b[clip * p.bsb + i0 * p.sb[0] + i1 * p.sb[1] + i2 * p.sb[2] + i3 * p.sb[3]];
        float v;
        switch (p.op) {
            case B_ADD: v = x + y; break;
            case B_SUB: v = x - y; break;
            case B_MUL: v = x * y; break;
            case B_DIV: v = x / y; break;
            case B_POW: v = powf(x, y); break;
            case B_MAX: v = fmaxf(x, y); break;
            case B_MIN: v = fminf(x, y); break;
            case B_SQDIFF: v = (x - y) * (x - y); break;
            default: v = x; break;
        }
        p.out[i] = g_act(v, p.act);
    }
}
void launch_binary_bcast(const BcastParams& p, int n_clips, hipStream_t s) {
    const size_t total = (size_t)n_clips * p.d[0] * p.d[1] * p.d[2] * p.d[3];
    size_t blocks = (total + 255) / 256; if (blocks > 32768) blocks = 32768; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_binary_bcast, dim3((unsigned)blocks), dim3(256), 0, s, p, total);
}

// ------------------------------------------------------------------------------------------ strided view copy
// out[clip][off_o + sum i_k * so_k] = in[clip][off_i + sum i_k * si_k]: CONCATENATION (one launch per input), STRIDED_SLICE /
// SLICE, TRANSPOSE, REVERSE_V2 (negative strides), PAD (interior after a zero fill), SPLIT.
__global__ void k_copy_view(CopyParams p, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t per = (size_t)p.d[0] * p.d[1] * p.d[2] * p.d[3];
    for (; i < total; i += stride) {
        const size_t clip = i / per;
        size_t r = i - clip * per;
        const int i3 = (int)(r % p.d[3]); r /= p.d[3];
        const int i2 = (int)(r % p.d[2]); r /= p.d[2];
        const int i1 = (int)(r % p.d[1]);
        const int i0 = (int)(r / p.d[1]);
        const long src = (long)clip * p.bsi + p.offi + i0 * p.si[0] + i1 * p.si[1] + i2 * p.si[2] + i3 * p.si[3];
        const long dst = (long)clip * p.bso + p.offo + i0 * p.so[0] + i1 * p.so[1] + i2 * p.so[2] + i3 * p.so[3];
        p.out[dst] = p.in[src];
    }
}
void launch_copy_view(const CopyParams& p, int n_clips, hipStream_t s) {
    const size_t total = (size_t)n_clips * p.d[0] * p.d[1] * p.d[2] * p.d[3];
    if (!total) return;
    size_t blocks = (total + 255) / 256; if (blocks > 32768) blocks = 32768;
    hipLaunchKernelGGL(k_copy_view, dim3((unsigned)blocks), dim3(256), 0, s, p, total);
}
__global__ void k_fill(float* __restrict__ out, size_t n, float v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = v;
}
void launch_fill(float* out, size_t n, float v, hipStream_t s) {
    if (!n) return;
    size_t blocks = (n + 255) / 256; if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(k_fill, dim3((unsigned)blocks), dim3(256), 0, s, out, n, v);
}

// ------------------------------------------------------------------------------------------ pooling (NHWC)
// TFLite AVERAGE_POOL_2D divides by the number of in-image taps (padding is excluded from the count); MAX_POOL_2D ignores
// padding.
__global__ void k_pool2d(PoolParams p, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int c = (int)(i % p.C);
        size_t r = i / p.C;
        const int wo = (int)(r % p.Wo); r /= p.Wo;
        const int ho = (int)(r % p.Ho);
        const size_t b = r / p.Ho;
        const int h0 = ho * p.sh - p.pt, w0 = wo * p.sw - p.pl;
        float acc = p.mode == 0 ? 0.0f : -INFINITY;
        int cnt = 0;
        for (int a = 0; a < p.kh; a++) {
            const int h = h0 + a;
            if (h < 0 || h >= p.H) continue;
            for (int q = 0; q < p.kw; q++) {
                const int w = w0 + q;
                if (w < 0 || w >= p.W) continue;
                const float v = p.in[((b * p.H + h) * p.W + w) * p.C + c];
                if (p.mode == 0) acc += v; else acc = fmaxf(acc, v);
                cnt++;
            }
        }
        if (p.mode == 0) acc = cnt ? acc / (float)cnt : 0.0f;
        p.out[i] = g_act(acc, p.act);
    }
}
void launch_pool2d(const PoolParams& p, hipStream_t s) {
    const size_t total = (size_t)p.B * p.Ho * p.Wo * p.C;
    if (!total) return;
    size_t blocks = (total + 255) / 256; if (blocks > 32768) blocks = 32768;
    hipLaunchKernelGGL(k_pool2d, dim3((unsigned)blocks), dim3(256), 0, s, p, total);
}

// ------------------------------------------------------------------------------------------ in-graph SOFTMAX (last dim)
// TFLite reference softmax: exp((x - max) * beta) / sum, float32.  One block per row, tree reductions.
__global__ __launch_bounds__(256) void k_softmax_rows(const float* __restrict__ x, float* __restrict__ out, int n, float beta) {
    __shared__ float red[4];
    const float* xr = x + (size_t)blockIdx.x * n;
    float* orow = out + (size_t)blockIdx.x * n;
    float m = -INFINITY;
    for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, xr[i]);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.0f;
    for (int i = threadIdx.x; i < n; i += 256) { float e = expf((xr[i] - m) * beta); orow[i] = e; sum += e; }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    sum = (red[0] + red[1]) + (red[2] + red[3]);
    for (int i = threadIdx.x; i < n; i += 256) orow[i] = orow[i] / sum;
}
void launch_softmax_rows(const float* in, float* out, size_t rows, int n, float beta, hipStream_t s) {
    if (!rows) return;
    hipLaunchKernelGGL(k_softmax_rows, dim3((unsigned)rows), dim3(256), 0, s, in, out, n, beta);
}

// ------------------------------------------------------------------------------------------ generic reduction
// One thread per output element; reduced dimensions are walked serially in index order (float32 accumulation like the
// TFLite reference reducers).
__global__ void k_reduce(ReduceParams p, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    int od[4];
    for (int k = 0; k < 4; k++) od[k] = ((p.mask >> k) & 1) ? 1 : p.d[k];
    const size_t per_out = (size_t)od[0] * od[1] * od[2] * od[3];
    const size_t per_in = (size_t)p.d[0] * p.d[1] * p.d[2] * p.d[3];
    const long s3 = 1, s2 = p.d[3], s1 = (long)p.d[2] * p.d[3], s0 = (long)p.d[1] * p.d[2] * p.d[3];
    int cnt = 1;
    for (int k = 0; k < 4; k++) if ((p.mask >> k) & 1) cnt *= p.d[k];
    for (; i < total; i += stride) {
        const size_t clip = i / per_out;
        size_t r = i - clip * per_out;
        const int o3 = (int)(r % od[3]); r /= od[3];
        const int o2 = (int)(r % od[2]); r /= od[2];
        const int o1 = (int)(r % od[1]);
        const int o0 = (int)(r / od[1]);
        const float* base = p.in + clip * per_in + o0 * s0 + o1 * s1 + o2 * s2 + o3 * s3;
        const int n0 = (p.mask & 1) ? p.d[0] : 1, n1 = (p.mask & 2) ? p.d[1] : 1, n2 = (p.mask & 4) ? p.d[2] : 1, n3 = (p.mask & 8) ? p.d[3] : 1;
        float acc = p.op == R_MAX ? -INFINITY : p.op == R_MIN ? INFINITY : p.op == R_PROD ? 1.0f : 0.0f;
        for (int a = 0; a < n0; a++)
            for (int b = 0; b < n1; b++)
                for (int c = 0; c < n2; c++)
                    for (int d = 0; d < n3; d++) {
                        const float v = base[a * s0 + b * s1 + c * s2 + d * s3];
                        switch (p.op) {
                            case R_MAX: acc = fmaxf(acc, v); break;
                            case R_MIN: acc = fminf(acc, v); break;
                            case R_PROD: acc *= v; break;
                            default: acc += v; break;
                        }
                    }
        if (p.op == R_MEAN) acc = acc / (float)cnt;
        p.out[i] = acc;
    }
}
void launch_reduce(const ReduceParams& p, int n_clips, hipStream_t s) {
    size_t per_out = 1;
    for (int k = 0; k < 4; k++) per_out *= ((p.mask >> k) & 1) ? 1 : (size_t)p.d[k];
    const size_t total = per_out * (size_t)n_clips;
    if (!total) return;
    size_t blocks = (total + 255) / 256; if (blocks > 32768) blocks = 32768;
    hipLaunchKernelGGL(k_reduce, dim3((unsigned)blocks), dim3(256), 0, s, p, total);
}

// ------------------------------------------------------------------------------------------ generic convolutions
// CONV_2D with any kernel / stride / dilation / padding / channel counts: thread per output element, weights OHWI as in the
// file.  (The planner prefers k_pw_gemm for 1x1 and k_conv_direct for Cout % 4 == 0 without dilation.)
__global__ void k_conv_generic(GenConvParams p, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int oc = (int)(i % p.Cout);
        size_t r = i / p.Cout;
        const int wo = (int)(r % p.Wo); r /= p.Wo;
        const int ho = (int)(r % p.Ho);
        const size_t b = r / p.Ho;
        float acc = 0.0f;
        if (p.depthwise) {
            const int ic = oc / p.mult;
            for (int a = 0; a < p.kh; a++) {
                const int h = ho * p.sh - p.pt + a * p.dh;
                if (h < 0 || h >= p.H) continue;
                for (int q = 0; q < p.kw; q++) {
                    const int w = wo * p.sw - p.pl + q * p.dw;
                    if (w < 0 || w >= p.W) continue;
                    acc = fmaf(p.in[((b * p.H + h) * p.W + w) * p.Cin + ic], p.w[(size_t)(a * p.kw + q) * p.Cout + oc], acc);
                }
            }
        } else {
            for (int a = 0; a < p.kh; a++) {
                const int h = ho * p.sh - p.pt + a * p.dh;
                if (h < 0 || h >= p.H) continue;
                for (int q = 0; q < p.kw; q++) {
                    const int w = wo * p.sw - p.pl + q * p.dw;
                    if (w < 0 || w >= p.W) continue;
                    const float* ip = p.in + ((b * p.H + h) * p.W + w) * p.Cin;
                    const float* wp = p.w + (((size_t)oc * p.kh + a) * p.kw + q) * p.Cin;
                    for (int ci = 0; ci < p.Cin; ci++) acc = fmaf(ip[ci], wp[ci], acc);
                }
            }
        }
        if (p.bias) acc += p.bias[oc];
        float y;
        switch (p.act) {
            case ACT_SWISH: y = acc * (1.0f / (1.0f + expf(-acc))); break;     // LOGISTIC then MUL, as the graph does
            case ACT_SIGMOID: y = 1.0f / (1.0f + expf(-acc)); break;
            case ACT_HARD_SWISH: y = acc * fminf(fmaxf(acc + 3.0f, 0.0f), 6.0f) / 6.0f; break;
            default: y = g_act(acc, p.act); break;
        }
        p.out[i] = y;
    }
}
void launch_conv_generic(const GenConvParams& p, hipStream_t s) {
    const size_t total = (size_t)p.B * p.Ho * p.Wo * p.Cout;
    if (!total) return;
    size_t blocks = (total + 255) / 256; if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_conv_generic, dim3((unsigned)blocks), dim3(256), 0, s, p, total);
}

}  // namespace bnhip
