// gfx950 hazard reproducer, reduced from the two-context corruption of k_mel_banded ("cross-context"):
// packed-fp32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) return wrong values in some lanes while ANOTHER
// wave on the same CU executes v_mfma_f32_16x16x32_bf16.  Nothing is shared between the two kernels but the CU.
//
// Self-checking: every lane runs the same dependent chain twice, once with the packed instruction and once with the two
// scalar instructions it stands for (both are IEEE fma / mul / add, so the results must be bit-identical), and counts the
// lanes / halves that differ.  The operands live in registers only: no LDS, no memory traffic inside the loop.
//
//   hipcc -x hip --offload-arch=gfx950 -O2 tools/ubench/pkf32_vs_bf16mfma.hip -o /tmp/pk && /tmp/pk [launches]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void co_mfma_bf16(float* sink, int iters) {
    f32x4 acc[4] = {};
    bf16x8_t a, b;
    const float f = threadIdx.x * 0.001f + 0.5f;
    for (int j = 0; j < 8; j++) { a[j] = (__bf16)(f + j); b[j] = (__bf16)(1.0f + 0.01f * j); }
    for (int i = 0; i < iters; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
    if (acc[0][0] == 12345.f) sink[threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}
__global__ __launch_bounds__(256) void co_mfma_f32(float* sink, int iters) {
    f32x4 acc[4] = {};
    const float a = threadIdx.x * 0.001f, b = 1.0001f;
    for (int i = 0; i < iters; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
    if (acc[0][0] == 12345.f) sink[threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}
__global__ __launch_bounds__(256) void co_mfma_bf16_32(float* sink, int iters) {       // the older 16x16x16 bf16 (gfx90a+ "_1k") form
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    f32x4 acc[4] = {};
    s16x4 a = {0x3f80, 0x3f81, 0x3f82, 0x3f83}, b = {0x3f80, 0x3f80, 0x3f80, 0x3f80};
    a[0] += (short)threadIdx.x;
    for (int i = 0; i < iters; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc[j], 0, 0, 0);
    if (acc[0][0] == 12345.f) sink[threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

// MODE 0: v_pk_fma_f32 d, x, w, d          1: ... op_sel_hi:[1,0,1] (src1 low half for both)      2: ... op_sel:[0,1,0] (src1 high half for both)
//      3: v_pk_mul_f32 + v_pk_add_f32 (unfused pair)   4-8: the other ways a HIGH source half can feed the LOW result (see below)
// cnt[0..3]: mismatching LOW halves per lane quarter (lanes 0-15, 16-31, 32-47, 48-63); cnt[4..7]: HIGH halves
template <int MODE>
__global__ __launch_bounds__(256) void k_pk(unsigned* cnt, int iters) {
    const int lane = threadIdx.x & 63;
    f32x2 x = {0.37f + 0.001f * lane, -0.21f + 0.002f * lane}, w = {0.9991f, 1.0007f}, accp = {0.f, 0.f};
    float s0 = 0.f, s1 = 0.f;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(accp) : "v"(x), "v"(w));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s0) : "v"(x[0]), "v"(w[0]));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s1) : "v"(x[1]), "v"(w[1]));
        } else if (MODE == 1) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(accp) : "v"(x), "v"(w));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s0) : "v"(x[0]), "v"(w[0]));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s1) : "v"(x[1]), "v"(w[0]));
        } else if (MODE == 2) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(accp) : "v"(x), "v"(w));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s0) : "v"(x[0]), "v"(w[1]));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s1) : "v"(x[1]), "v"(w[1]));
        } else if (MODE == 4) {                  // src0's high half feeds the low result
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(accp) : "v"(x), "v"(w));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s0) : "v"(x[1]), "v"(w[0]));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s1) : "v"(x[1]), "v"(w[1]));
        } else if (MODE == 5) {                  // halves of src1 swapped
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "+v"(accp) : "v"(x), "v"(w));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s0) : "v"(x[0]), "v"(w[1]));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s1) : "v"(x[1]), "v"(w[0]));
        } else if (MODE == 6) {                  // packed multiply, src1 high half for both, then packed add
            f32x2 t; float t0, t1;
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(t) : "v"(x), "v"(w));
            asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(accp) : "v"(t));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t0) : "v"(x[0]), "v"(w[1]));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t1) : "v"(x[1]), "v"(w[1]));
            asm volatile("v_add_f32 %0, %1, %0" : "+v"(s0) : "v"(t0));
            asm volatile("v_add_f32 %0, %1, %0" : "+v"(s1) : "v"(t1));
        } else if (MODE == 7) {                  // packed add with src0's high half for both
            f32x2 t; float t0, t1;
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(x), "v"(w));
            asm volatile("v_pk_add_f32 %0, %1, %0 op_sel:[1,0]" : "+v"(accp) : "v"(t));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t0) : "v"(x[0]), "v"(w[0]));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t1) : "v"(x[1]), "v"(w[1]));
            asm volatile("v_add_f32 %0, %1, %0" : "+v"(s0) : "v"(t1));
            asm volatile("v_add_f32 %0, %1, %0" : "+v"(s1) : "v"(t1));
        } else if (MODE == 8) {                  // accumulator (src2) high half feeds the low result
            const float a1 = accp[1];
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,1]" : "+v"(accp) : "v"(x), "v"(w));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s0) : "v"(x[0]), "v"(w[0]), "v"(a1));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s1) : "v"(x[1]), "v"(w[1]));
        } else {
            f32x2 t; float t0, t1;
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(x), "v"(w));
            asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(accp) : "v"(t));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t0) : "v"(x[0]), "v"(w[0]));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t1) : "v"(x[1]), "v"(w[1]));
            asm volatile("v_add_f32 %0, %1, %0" : "+v"(s0) : "v"(t0));
            asm volatile("v_add_f32 %0, %1, %0" : "+v"(s1) : "v"(t1));
        }
        // keep the chains bounded and data-dependent: fold the accumulators back into x every 16 steps
        if ((i & 15) == 15) {
            x[0] = 0.5f * x[0] + 1e-3f * s0; x[1] = 0.5f * x[1] - 1e-3f * s1;
            const bool lo_bad = __float_as_uint(accp[0]) != __float_as_uint(s0), hi_bad = __float_as_uint(accp[1]) != __float_as_uint(s1);
            if (lo_bad) atomicAdd(&cnt[lane >> 4], 1u);
            if (hi_bad) atomicAdd(&cnt[4 + (lane >> 4)], 1u);
            accp[0] = s0 = 0.25f * s0; accp[1] = s1 = 0.25f * s1;
        }
    }
}

template <int MODE>
static void run(const char* what, int co, int launches, unsigned* cnt, float* sink, hipStream_t sa, hipStream_t sb) {
    (void)hipMemset(cnt, 0, 32);
    (void)hipDeviceSynchronize();
    for (int i = 0; i < launches; i++) {
        if (co == 1) hipLaunchKernelGGL(co_mfma_bf16, dim3(1024), dim3(256), 0, sb, sink, 3000);
        if (co == 2) hipLaunchKernelGGL(co_mfma_f32, dim3(1024), dim3(256), 0, sb, sink, 1500);
        if (co == 3) hipLaunchKernelGGL(co_mfma_bf16_32, dim3(1024), dim3(256), 0, sb, sink, 3000);
        hipLaunchKernelGGL(k_pk<MODE>, dim3(2048), dim3(256), 0, sa, cnt, 4096);
    }
    (void)hipDeviceSynchronize();
    unsigned h[8];
    (void)hipMemcpy(h, cnt, 32, hipMemcpyDeviceToHost);
    const char* con[] = {"alone", "beside v_mfma_f32_16x16x32_bf16", "beside v_mfma_f32_16x16x4_f32", "beside v_mfma_f32_16x16x16_bf16"};
    printf("%-44s %-34s: low-half mismatches by lane quarter %u %u %u %u | high-half %u %u %u %u\n", what, con[co], h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
}

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 20;
    unsigned* cnt; float* sink;
    (void)hipMalloc(&cnt, 32); (void)hipMalloc(&sink, 1 << 20);
    hipStream_t sa, sb;
    (void)hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    for (int co = 0; co < 2; co++) {
        run<0>("v_pk_fma_f32", co, launches, cnt, sink, sa, sb);
        run<1>("v_pk_fma_f32 op_sel_hi:[1,0,1]", co, launches, cnt, sink, sa, sb);
        run<2>("v_pk_fma_f32 op_sel:[0,1,0]", co, launches, cnt, sink, sa, sb);
        run<3>("v_pk_mul_f32 + v_pk_add_f32", co, launches, cnt, sink, sa, sb);
        run<4>("v_pk_fma_f32 op_sel:[1,0,0]", co, launches, cnt, sink, sa, sb);
        run<5>("v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,0,1]", co, launches, cnt, sink, sa, sb);
        run<6>("v_pk_mul_f32 op_sel:[0,1] + v_pk_add_f32", co, launches, cnt, sink, sa, sb);
        run<7>("v_pk_mul_f32 + v_pk_add_f32 op_sel:[1,0]", co, launches, cnt, sink, sa, sb);
        run<8>("v_pk_fma_f32 op_sel:[0,0,1]", co, launches, cnt, sink, sa, sb);
    }
    return 0;
}
