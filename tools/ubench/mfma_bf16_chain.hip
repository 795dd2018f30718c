// Micro-benchmark: issue rate of v_mfma_f32_16x16x32_bf16 as a function of how many INDEPENDENT accumulator chains a wave
// interleaves (1 = every MFMA reads the previous one's result as its C operand, the shape of the six-product loop of
// k_pw_bx3 / k_pw_b16 with one 16-row tile per wave).  Prints cycles per MFMA per wave at 1 and 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_bf16_chain tools/ubench/mfma_bf16_chain.hip && ./mfma_bf16_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));
template <int CH>
__global__ void k(float* out, int iters, long long* cyc) {
    f32x4 acc[CH];
    for (int i = 0; i < CH; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    b16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (__bf16)(threadIdx.x * 1e-3f + i); b[i] = (__bf16)(1.0f + i); }
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int i = 0; i < CH; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    const long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < CH; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int CH>
void run(float* out, long long* dcyc, int wps) {
    const int iters = 4000;
    hipLaunchKernelGGL(k<CH>, dim3(256), dim3(64 * 4 * wps), 0, 0, out, iters, dcyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<CH>, dim3(256), dim3(64 * 4 * wps), 0, 0, out, iters, dcyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 6 * CH;
    const double tf = 256.0 * 4 * wps * n * 16384 / (ms * 1e-3) / 1e12;
    printf("chains %d, %d wave(s)/SIMD: %.1f clock64 ticks per MFMA per wave, %.0f TFLOP/s\n", CH, wps, (double)c / n, tf);
}
int main() {
    float* out; long long* dcyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&dcyc, 8);
    for (int wps = 1; wps <= 2; wps++) { run<1>(out, dcyc, wps); run<2>(out, dcyc, wps); run<3>(out, dcyc, wps); run<4>(out, dcyc, wps); run<8>(out, dcyc, wps); }
    return 0;
}
