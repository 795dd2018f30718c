// Micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 on gfx950 as a function of independent accumulator chains
// per wave (NACC) and waves per SIMD.  hipcc --offload-arch=gfx950 -O3 -o mfma_f64 mfma_f64.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k(double* out, int iters, long long* cyc) {
    f64x4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = (f64x4){0., 0., 0., 0.};
    double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3;
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC>
void run(int waves_per_simd, double* out, long long* cyc) {
    int iters = 4000;
    int threads = 64 * 4 * waves_per_simd;     // one block per CU, 4 SIMDs
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    double n_per_simd = (double)iters * NACC * waves_per_simd;
    double tf = 256.0 * 4 * n_per_simd * 2048 / (ms * 1e-3) / 1e12;
    printf("NACC=%d waves/SIMD=%d: %.3f ms, %.1f clk/MFMA/SIMD (s_memtime), %.1f TF\n", NACC, waves_per_simd, ms,
           (double)c / n_per_simd, tf);
}
int main() {
    double* out; long long* cyc;
    hipMalloc(&out, 256 * 1024 * 8); hipMalloc(&cyc, 8);
    for (int w = 1; w <= 4; w *= 2) { run<1>(w, out, cyc); run<2>(w, out, cyc); run<3>(w, out, cyc); run<4>(w, out, cyc); run<6>(w, out, cyc); }
    return 0;
}
