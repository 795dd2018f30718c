// Micro-benchmark: can ONE wave per SIMD keep the f32 MFMA pipe busy in a GEMM-shaped loop?  One 256-thread block per CU
// (LDS use forces it), block tile 192 x 288, waves 2 x 2, each wave 96 x 144 = 6 x 9 MFMA tiles (216 accumulator VGPRs).
// Per 16-wide K half: 15 ds_read_b128 fragments feed 216 MFMAs.  MODE 0: LDS reads only (no refill);
// MODE 1: + per-slab refill of the operand tiles from registers with two barriers; MODE 2: + the global loads behind it.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LS 40
#define TM 192
#define TN 288
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k(const float* __restrict__ g, float* out, int slabs) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Xs = lds;
    float* Ws = lds + TM * LS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;
    for (int i = tid; i < (TM + TN) * LS; i += 256) lds[i] = (float)(i % 97) * 1e-3f;
    __syncthreads();
    f32x4 acc[9][6];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int m = 0; m < 6; m++) acc[t][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int NQ = (TM + TN) * 8 / 256;   // float4 per thread per slab = 15
    float4 st[NQ];
    const float* gp = g + (size_t)blockIdx.x * (TM + TN) * 32;
    if (MODE >= 1) {
#pragma unroll
        for (int q = 0; q < NQ; q++) st[q] = MODE == 2 ? *reinterpret_cast<const float4*>(gp + 4 * (tid + 256 * q)) : make_float4(1e-3f * q, 0.f, 1.f, 2.f);
    }
    for (int sl = 0; sl < slabs; sl++) {
        f32x4 xf[2][6], wf[2][9];
        auto lread = [&](int h) {
#pragma unroll
            for (int m = 0; m < 6; m++) xf[h][m] = *reinterpret_cast<const f32x4*>(&Xs[(96 * wr + 16 * m + li) * LS + 16 * h + 4 * kq]);
#pragma unroll
            for (int t = 0; t < 9; t++) wf[h][t] = *reinterpret_cast<const f32x4*>(&Ws[(144 * wc + 16 * t + li) * LS + 16 * h + 4 * kq]);
        };
        lread(0);
        lread(1);
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int t = 0; t < 9; t++)
#pragma unroll
                    for (int m = 0; m < 6; m++)
                        acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[h][t][s], xf[h][m][s], acc[t][m], 0, 0, 0);
        if (MODE >= 1) {
            __syncthreads();
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                int idx = tid + 256 * q;
                *reinterpret_cast<float4*>(&lds[(idx >> 3) * LS + 4 * (idx & 7)]) = st[q];
            }
            if (MODE == 2) {
                const float* gn = gp + (size_t)((sl + 1) & 7) * 256 * (TM + TN) * 32;
#pragma unroll
                for (int q = 0; q < NQ; q++) st[q] = *reinterpret_cast<const float4*>(gn + 4 * (tid + 256 * q));
            }
            __syncthreads();
        }
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int m = 0; m < 6; m++) s += acc[t][m][0] + acc[t][m][1] + acc[t][m][2] + acc[t][m][3];
    out[blockIdx.x * 256 + tid] = s;
}
template <int MODE>
void run(const float* g, float* out) {
    const int slabs = 200;
    size_t lds = (size_t)(TM + TN) * LS * 4 + 70 * 1024;     // pad so that only one block fits a CU
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), lds, 0, g, out, slabs);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), lds, 0, g, out, slabs);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double flops = 256.0 * slabs * 2.0 * TM * TN * 32;
    printf("mode %d: %.3f ms  %.1f TF  (%s)\n", MODE, ms, flops / (ms * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
}
int main() {
    float *g, *out;
    (void)hipMalloc(&g, (size_t)8 * 256 * (TM + TN) * 32 * 4 + 4096);
    (void)hipMemset(g, 0, (size_t)8 * 256 * (TM + TN) * 32 * 4 + 4096);
    (void)hipMalloc(&out, 256 * 256 * 4);
    run<0>(g, out); run<1>(g, out); run<2>(g, out);
    return 0;
}
