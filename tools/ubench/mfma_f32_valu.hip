// Micro-benchmark: does VALU work co-execute with v_mfma_f32_16x16x4_f32 on gfx950?  Each wave runs 3 independent
// accumulator chains; per MFMA it additionally executes NV VALU ops of a chosen kind.  2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NV, int KIND>
__global__ void k(double* out, int iters, float seed) {
    f32x4 acc[3];
    for (int i = 0; i < 3; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    float f[8]; double d[8];
    for (int i = 0; i < 8; i++) { f[i] = seed + i + threadIdx.x; d[i] = seed * i; }
    __shared__ float sh[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) sh[i] = i * seed;
    __syncthreads();
    int idx = threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; v++) {
                if (KIND == 0) f[v & 7] = f[v & 7] * 1.0001f;                       // v_mul_f32
                if (KIND == 1) d[v & 7] = d[v & 7] + 1.0;                           // v_add_f64
                if (KIND == 2) d[v & 7] = (double)f[v & 7] + d[(v + 1) & 7];        // cvt + add_f64
                if (KIND == 3) { idx = (idx + 67) & 4095; f[v & 7] += sh[idx]; }    // ds_read_b32 + add
                if (KIND == 4) idx = (idx * 3 + 1) & 4095;                          // int ops
                if (KIND == 5) f[v & 7] = __builtin_amdgcn_rcpf(1.0f + __expf(f[v & 7]));  // 2 transcendental + 2
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 3; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; i++) s += f[i] + d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + idx;
}
template <int NV, int KIND>
void run(double* out) {
    int iters = 6000, wps = 2;
    int threads = 64 * 4 * wps;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, KIND>), dim3(256), dim3(threads), 0, 0, out, iters, 1.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, KIND>), dim3(256), dim3(threads), 0, 0, out, iters, 1.5f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double n_per_simd = (double)iters * 3 * wps;
    double tf = 256.0 * 4 * n_per_simd * 2048 / (ms * 1e-3) / 1e12;
    const char* names[] = {"v_mul_f32", "v_add_f64", "cvt+add_f64", "ds_read+add", "int mad", "exp+rcp"};
    printf("%-12s x%d per MFMA: %.3f ms  %.1f TF\n", names[KIND], NV, ms, tf);
}
int main() {
    double* out; (void)hipMalloc(&out, 256 * 1024 * 8);
    run<0, 0>(out);
    run<2, 0>(out); run<4, 0>(out); run<8, 0>(out);
    run<2, 1>(out); run<4, 1>(out); run<8, 1>(out);
    run<2, 2>(out); run<4, 2>(out);
    run<1, 3>(out); run<2, 3>(out); run<4, 3>(out);
    run<2, 4>(out); run<4, 4>(out); run<8, 4>(out); run<2, 5>(out); run<4, 5>(out);
    return 0;
}
