// Reproducer hunt for the two-context corruption (DESIGN.md "cross-context"): does k_mel_banded<2> - compiled WITH the SLP
// vectoriser, i.e. with the dependent v_pk_fma_f32 chains - return different values when OTHER kernels share the CU?
// Nothing is shared with the co-runner but the CU: own buffers, own streams, no engine, no arena.
//
//   stream A: k_mel_banded<2>(binsA) -> outA, compared on the device with the result of the same launch made alone
//   stream B: one of the co-runners below, in a loop, on its own buffers
//
// build (both ways) and run:
//   hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form \
//         -I birdnet-go_amd/csrc tools/ubench/melband_xctx.hip -o /tmp/mx_slp
//   hipcc ... -fno-slp-vectorize ... -o /tmp/mx_noslp
//   /tmp/mx_slp [iters]
#include "../../birdnet-go_amd/csrc/stft.hip"

#include <cstring>
#include <random>

using namespace bnhip;

// ---- co-runners -------------------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void co_mfma_f32(float* sink, int iters) {
    f32x4 acc[4] = {};
    float a = threadIdx.x * 0.001f, b = 1.0001f;
    for (int i = 0; i < iters; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
    if (acc[0][0] == 12345.f) sink[threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}
__global__ __launch_bounds__(256) void co_mfma_f64(double* sink, int iters) {
    f64x4 acc[2] = {};
    double a = threadIdx.x * 0.001, b = 1.0001;
    for (int i = 0; i < iters; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
    if (acc[0][0] == 12345.0) sink[threadIdx.x] = acc[0][0] + acc[1][1];
}
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
// the bf16 matrix pipe as the engine's split-bf16 GEMM (k_pw_bx3) drives it: v_mfma_f32_16x16x32_bf16, VGPR form, with the
// cvt / shift VALU work of the operand split in between
__global__ __launch_bounds__(256) void co_mfma_bf16(float* sink, int iters, int with_valu) {
    f32x4 acc[4] = {};
    bf16x8_t a, b;
    float f = threadIdx.x * 0.001f + 0.5f;
    for (int j = 0; j < 8; j++) { a[j] = (__bf16)(f + j); b[j] = (__bf16)(1.0f + 0.01f * j); }
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
        if (with_valu) {
            f = f * 1.0001f + 0.25f;
            unsigned u = __float_as_uint(f) & 0xffff0000u;
            float r = f - __uint_as_float(u);
            a[i & 7] = (__bf16)r;
        }
    }
    if (acc[0][0] == 12345.f) sink[threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}
// every VGPR a wave can get (512 / 1 wave per SIMD is not reachable with 256 threads; 128 arch VGPRs at 4 waves) written with a NaN
// pattern, and 64 KB of LDS too: whoever reads a register or an LDS word it never wrote sees this
__global__ __launch_bounds__(256) void co_poison(float* sink, int iters) {
    extern __shared__ float lds[];
    const float nanv = __int_as_float(0x7fc0dead);
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = nanv;
    float r[96];
#pragma unroll
    for (int j = 0; j < 96; j++) r[j] = nanv;
    for (int i = 0; i < iters; i++)
#pragma unroll
        for (int j = 0; j < 96; j++) asm volatile("v_mov_b32 %0, %1" : "+v"(r[j]) : "v"(nanv));
    float s = 0;
#pragma unroll
    for (int j = 0; j < 96; j++) s += r[j];
    if (s == 12345.f) sink[threadIdx.x] = s + lds[threadIdx.x];
}
__global__ __launch_bounds__(256) void co_trans(float* sink, int iters) {
    float x = threadIdx.x * 0.01f + 0.5f, y = x + 0.25f;
    for (int i = 0; i < iters; i++) { x = __expf(-x) + 0.5f; y = __builtin_amdgcn_rcpf(y) + 0.5f; }
    if (x + y == 12345.f) sink[threadIdx.x] = x + y;
}
__global__ __launch_bounds__(256) void co_fp64(double* sink, int iters) {
    double x = threadIdx.x * 0.01 + 0.5, y = 1.000001;
    for (int i = 0; i < iters; i++) x = fma(x, y, 1e-9);
    if (x == 12345.0) sink[threadIdx.x] = x;
}
__global__ void co_copy(const float4* a, float4* b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) b[i] = a[i];
}
__global__ void k_cmp(const unsigned* a, const unsigned* b, size_t n, unsigned* cnt, unsigned* first) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st)
        if (a[i] != b[i]) { unsigned k = atomicAdd(cnt, 1u); if (k < 64) first[k] = (unsigned)i; }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 60;
    const int B = 256, F = 511, M = 96, nb[2] = {132, 312};
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    // banded triangular weights, like the mel matrix the planner hands over
    std::vector<float> w[2]; std::vector<int> span[2];
    for (int c = 0; c < 2; c++) {
        w[c].assign((size_t)M * nb[c], 0.f); span[c].resize(2 * M);
        for (int m = 0; m < M; m++) {
            const float ctr = (m + 1.f) * nb[c] / (M + 1.f), half = 1.5f + 0.04f * m * nb[c] / 132.f;
            int lo = std::max(0, (int)std::floor(ctr - half)), hi = std::min(nb[c], (int)std::ceil(ctr + half) + 1);
            for (int k = lo; k < hi; k++) w[c][(size_t)m * nb[c] + k] = std::max(0.f, 1.f - std::fabs(k - ctr) / half) * 0.7f + 1e-3f;
            span[c][2 * m] = lo; span[c][2 * m + 1] = hi;
        }
    }
    auto dev = [](const void* h, size_t bytes) { void* d; hipMalloc(&d, bytes); if (h) hipMemcpy(d, h, bytes, hipMemcpyHostToDevice); return d; };
    MelBandParams pA{}, pB{};
    std::vector<float> hb;
    for (int which = 0; which < 2; which++) {
        MelBandParams& p = which ? pB : pA;
        for (int c = 0; c < 2; c++) {
            hb.resize((size_t)B * F * nb[c]);
            for (auto& v : hb) { float u = U(rng); v = which ? 40.f * u : u * u * u * 3.f; }      // context B: other data, other scale
            p.bins[c] = (const float*)dev(hb.data(), hb.size() * 4);
            p.w[c] = (const float*)dev(w[c].data(), w[c].size() * 4);
            p.span[c] = (const int*)dev(span[c].data(), span[c].size() * 4);
            p.nbp[c] = nb[c]; p.p1[c] = 2.f; p.p2[c] = 0.45f;
        }
        p.out = (float*)dev(nullptr, (size_t)B * M * F * 2 * 4);
        p.F = F; p.n_mels = M; p.Ctot = 2; p.c0 = 0;
    }
    const size_t n_out = (size_t)B * M * F * 2;
    float* ref = (float*)dev(nullptr, n_out * 4);
    unsigned *cnt, *first; hipMalloc(&cnt, 4); hipMalloc(&first, 256);
    float* sink = (float*)dev(nullptr, 1 << 20);
    float4 *ca = (float4*)dev(nullptr, (size_t)256 << 20), *cb = (float4*)dev(nullptr, (size_t)256 << 20);
    hipMemset(ca, 0, (size_t)256 << 20);
    hipStream_t sa, sb;
    hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&co_poison), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    // the serial truth, and that it repeats
    launch_mel_banded(pA, 2, B, sa);
    hipStreamSynchronize(sa);
    hipMemcpy(ref, pA.out, n_out * 4, hipMemcpyDeviceToDevice);
    const char* names[] = {"alone", "mel_banded(other data)", "f32 MFMA", "f64 MFMA", "VGPR+LDS poison", "transcendental VALU", "fp64 VALU", "HBM copy",
                           "mel_banded(other data) + f32 MFMA", "bf16 MFMA 16x16x32", "bf16 MFMA + split VALU", "bf16 MFMA, 512-thread grid tail"};
    const int NCO = 12;
    int total_bad = 0;
    for (int co = 0; co < NCO; co++) {
        hipMemset(cnt, 0, 4);
        hipDeviceSynchronize();
        unsigned bad_iters = 0, last = 0;
        for (int i = 0; i < iters; i++) {
            // the co-runner first, sized to outlive the kernel under test (~100 us)
            switch (co) {
                case 1: launch_mel_banded(pB, 2, B, sb); break;
                case 2: hipLaunchKernelGGL(co_mfma_f32, dim3(1024), dim3(256), 0, sb, sink, 2000); break;
                case 3: hipLaunchKernelGGL(co_mfma_f64, dim3(1024), dim3(256), 0, sb, (double*)sink, 1000); break;
                case 4: hipLaunchKernelGGL(co_poison, dim3(1024), dim3(256), 64 * 1024, sb, sink, 200); break;
                case 5: hipLaunchKernelGGL(co_trans, dim3(2048), dim3(256), 0, sb, sink, 3000); break;
                case 6: hipLaunchKernelGGL(co_fp64, dim3(2048), dim3(256), 0, sb, (double*)sink, 4000); break;
                case 7: hipLaunchKernelGGL(co_copy, dim3(4096), dim3(256), 0, sb, (const float4*)ca, cb, ((size_t)256 << 20) / 16); break;
                case 8: launch_mel_banded(pB, 2, B, sb); hipLaunchKernelGGL(co_mfma_f32, dim3(512), dim3(256), 0, sb, sink, 1000); break;
                case 9: hipLaunchKernelGGL(co_mfma_bf16, dim3(1024), dim3(256), 0, sb, sink, 4000, 0); break;
                case 10: hipLaunchKernelGGL(co_mfma_bf16, dim3(1024), dim3(256), 0, sb, sink, 3000, 1); break;
                case 11: for (int r = 0; r < 6; r++) hipLaunchKernelGGL(co_mfma_bf16, dim3(700), dim3(256), 0, sb, sink, 600, 1); break;
                default: break;
            }
            hipMemsetAsync(pA.out, 0xff, n_out * 4, sa);
            launch_mel_banded(pA, 2, B, sa);
            hipLaunchKernelGGL(k_cmp, dim3(1024), dim3(256), 0, sa, (const unsigned*)pA.out, (const unsigned*)ref, n_out, cnt, first);
            if ((i & 7) == 7) {
                hipStreamSynchronize(sa);
                unsigned h = 0; hipMemcpy(&h, cnt, 4, hipMemcpyDeviceToHost);
                if (h != last) { bad_iters++; last = h; }
            }
        }
        hipDeviceSynchronize();
        unsigned h = 0, hf[64]; hipMemcpy(&h, cnt, 4, hipMemcpyDeviceToHost); hipMemcpy(hf, first, 256, hipMemcpyDeviceToHost);
        printf("co-runner %-36s: %d launches, %u mismatching output words (in >= %u of the 8-launch groups)\n", names[co], iters, h, bad_iters);
        if (h) {
            printf("   first mismatches (flat index -> clip, mel, frame, channel):");
            for (unsigned k = 0; k < std::min(h, 12u); k++) {
                unsigned i = hf[k]; printf(" %u->(%u,%u,%u,%u)", i, i / (M * F * 2), (i / (F * 2)) % M, (i / 2) % F, i & 1);
            }
            printf("\n");
        }
        total_bad += h != 0;
    }
    printf("RESULT: %d of %d co-runner settings produced mismatches\n", total_bad, NCO);
    return 0;
}
