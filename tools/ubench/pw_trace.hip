// Where does a k_pw_gemm wave spend its time?  Builds the library kernel with PW_TRACE (shader-clock stamps at the phase
// boundaries of every K slab, first 64 logical blocks) and runs it on the late-layer shapes at batch 256.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -DPW_TRACE -I birdnet-go_amd/csrc \
//         -o tools/ubench/pw_trace.bin tools/ubench/pw_trace.hip
#include "../../birdnet-go_amd/csrc/kernels.hip"

#include <vector>
using namespace bnhip;

static void run(const char* name, int M, int N, int K, int nt, int wm) {
    float *A, *W, *bias, *out;
    (void)hipMalloc(&A, (size_t)M * K * 4); (void)hipMalloc(&W, (size_t)N * K * 4);
    (void)hipMalloc(&bias, (size_t)N * 4); (void)hipMalloc(&out, (size_t)M * N * 4);
    (void)hipMemset(A, 0, (size_t)M * K * 4); (void)hipMemset(W, 0, (size_t)N * K * 4); (void)hipMemset(bias, 0, (size_t)N * 4);
    long long* tr;
    const size_t nt_ = (size_t)64 * 4 * PW_TRACE_SLOTS;
    (void)hipMalloc(&tr, nt_ * 8); (void)hipMemset(tr, 0, nt_ * 8);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pw_trace), &tr, sizeof(tr));
    PwParams p{A, W, bias, nullptr, nullptr, out, M, N, K, 48, ACT_NONE, nt, wm};
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch_pw_gemm(p, 0);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    launch_pw_gemm(p, 0);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(nt_);
    (void)hipMemcpy(h.data(), tr, nt_ * 8, hipMemcpyDeviceToHost);
    const int nslab = (K + 31) / 32;
    double pro = 0, mfma = 0, bar1 = 0, store = 0, bar2 = 0, epi = 0, total = 0, vm = 0, lw = 0, gl = 0; int nw = 0;
    for (int w = 0; w < 64 * 4; w++) {
        const long long* t = &h[(size_t)w * PW_TRACE_SLOTS];
        if (!t[0] || !t[3]) continue;
        nw++;
        pro += t[1] - t[0]; epi += t[3] - t[2]; total += t[3] - t[0];
        for (int sl = 0; sl < nslab && sl < 14; sl++) {
            mfma += t[5 + 4 * sl] - t[4 + 4 * sl];
            if (sl + 1 < nslab) {
                bar1 += t[6 + 4 * sl] - t[5 + 4 * sl];
                store += t[7 + 4 * sl] - t[6 + 4 * sl];
                { vm += t[64 + 2 * sl] - t[6 + 4 * sl]; lw += t[65 + 2 * sl] - t[64 + 2 * sl]; gl += t[7 + 4 * sl] - t[65 + 2 * sl]; }
                bar2 += t[4 + 4 * (sl + 1)] - t[7 + 4 * sl];
            }
        }
    }
    printf("%-14s M=%d N=%d K=%d nt=%d wm=%d: %.1f us %.1f TF | per wave (clock64 ticks, %d waves): total %.0f = prologue %.0f + "
           "slabs[issue reads+MFMA %.0f, wait barrier1 %.0f, lstore+gload %.0f (vmcnt wait %.0f, LDS stores %.0f, gload issue %.0f), wait barrier2 %.0f] + epilogue %.0f  (%d slabs)\n",
           name, M, N, K, nt, wm, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12, nw, total / nw, pro / nw, mfma / nw, bar1 / nw,
           store / nw, vm / nw, lw / nw, gl / nw, bar2 / nw, epi / nw, nslab);
    hipFree(A); hipFree(W); hipFree(bias); hipFree(out); hipFree(tr);
}
int main() {
    run("b13/expand", 12288, 1152, 192, 4, 2);
    run("b13/expand", 12288, 1152, 192, 2, 1);
    run("b13/project", 12288, 192, 1152, 3, 1);
    run("b10/project", 49152, 112, 672, 4, 2);
    run("top", 12288, 1024, 320, 4, 1);
    return 0;
}
