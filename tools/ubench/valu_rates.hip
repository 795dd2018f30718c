// Micro-benchmark (round 6): issue cost of the VALU instruction kinds the fused expand + depthwise kernels are made of, alone on a SIMD
// (no MFMA beside them): W waves per SIMD, each a long unrolled stream of 8 independent chains of one instruction kind.
//   v_mul_f32 / v_fma_f32 / v_pk_mul_f32 / v_exp_f32 / v_rcp_f32 / the whole swish (mul, exp2, add, rcp, mul)
// Prints cycles per wave-instruction per SIMD (shader clock from s_memtime deltas), so "is a transcendental a quarter-rate op" has a
// number: DESIGN.md 5.3 prices the swish with it.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ void k(float* out, unsigned long long* cyc, int iters, float seed) {
    float f[8];
    f32x2 g[8];
    for (int i = 0; i < 8; i++) { f[i] = seed + 0.01f * i + 1e-4f * threadIdx.x; g[i] = (f32x2){f[i], f[i] * 0.5f}; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (KIND == 0) f[i] = f[i] * 1.0001f;
                if (KIND == 1) f[i] = __builtin_fmaf(f[i], 1.0001f, 1e-7f);
                if (KIND == 2) g[i] = g[i] * (f32x2){1.0001f, 0.9999f};
                if (KIND == 3) f[i] = __builtin_amdgcn_exp2f(f[i]) * 0.0f + f[i];         // exp + fma (the fma keeps the chain finite)
                if (KIND == 4) f[i] = __builtin_amdgcn_rcpf(f[i]) * 0.0f + f[i];          // rcp + fma
                if (KIND == 6) {                                   // v_dot2c_f32_bf16: two bf16 products accumulated into fp32 (no builtin selects it: inline asm)
                    const unsigned a = __float_as_uint(g[i][0]), b = 0x3f803f80u;
                    asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(f[i]) : "v"(a), "v"(b));
                }
                if (KIND == 7) {                                   // v_dot2_f32_f16
                    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                    const h2 a = __builtin_bit_cast(h2, __float_as_uint(g[i][0])), b = __builtin_bit_cast(h2, 0x3c003c00u);
                    f[i] = __builtin_amdgcn_fdot2(a, b, f[i], false);
                }
                if (KIND == 5) { const float t = f[i] * -1.4426950408889634f; const float e = __builtin_amdgcn_exp2f(t) + 1.0f; f[i] = f[i] * __builtin_amdgcn_rcpf(e) + 1.0f; }
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; i++) s += f[i] + g[i][0] + g[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}
template <int KIND>
void run(float* out, unsigned long long* cyc, int wps, int per_iter, const char* name) {
    const int iters = 2000, threads = 64 * 4 * wps, blocks = 256;
    hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.25f);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.25f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[64]; (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double c = 0; for (int i = 0; i < 4 * wps; i++) c += (double)h[i]; c /= 4 * wps;
    // s_memtime ticks at 100 MHz on gfx950: cycles from the wall time of the launch at the clock the kernel ran at is not known here,
    // so report both: ticks per instruction per wave (x wps = per SIMD) and ns per instruction per SIMD from the event time
    const double insts = (double)iters * 32 * per_iter;
    printf("%-26s %d waves/SIMD: %7.3f ns per wave-instruction per SIMD (event time)   %6.2f memtime ticks per instruction per wave\n", name, wps,
           ms * 1e6 / (insts * wps), c / insts);
}
int main() {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&cyc, 4096 * 8);
    for (int wps : {1, 2, 4}) {
        run<0>(out, cyc, wps, 1, "v_mul_f32");
        run<1>(out, cyc, wps, 1, "v_fma_f32");
        run<2>(out, cyc, wps, 1, "v_pk_mul_f32");
        run<3>(out, cyc, wps, 2, "v_exp_f32 + v_fma_f32");
        run<4>(out, cyc, wps, 2, "v_rcp_f32 + v_fma_f32");
        run<5>(out, cyc, wps, 5, "swish (mul exp add rcp fma)");
        run<6>(out, cyc, wps, 1, "v_dot2c_f32_bf16");
        run<7>(out, cyc, wps, 1, "v_dot2_f32_f16");
    }
    return 0;
}
