// Where does a k_pw_b16 wave spend its time?  Builds the library kernel with B16_TRACE (shader-clock stamps at the phase
// boundaries of every K slab, first 64 logical blocks) and runs its six-product form on late-layer shapes of the v2.4 stack at
// batch 256 (operand values are irrelevant to timing: zeros).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form -DB16_TRACE \
//         -I birdnet-go_amd/csrc -o tools/ubench/bin/b16_trace tools/ubench/b16_trace.hip
#include "../../birdnet-go_amd/csrc/pw_b16.hip"

#include <vector>
using namespace bnhip;

static void run(const char* name, int M, int N, int K, int nt, int wm, int act, int hw = 48, bool scale = false) {
    const int spb = 1;
    float *A, *bias, *out; uint16_t* Wimg;
    const int Npad = (N + 15) / 16 * 16, nslab = (K + 31) / 32;
    (void)hipMalloc(&A, (size_t)M * K * 4); (void)hipMalloc(&Wimg, (size_t)nslab * 12 * Npad * 16);
    (void)hipMalloc(&bias, (size_t)N * 4); (void)hipMalloc(&out, (size_t)M * N * 4);
    (void)hipMemset(A, 0, (size_t)M * K * 4); (void)hipMemset(Wimg, 0, (size_t)nslab * 12 * Npad * 16); (void)hipMemset(bias, 0, (size_t)N * 4);
    long long* tr;
    const size_t nt_ = (size_t)64 * 4 * B16_TRACE_SLOTS;
    (void)hipMalloc(&tr, nt_ * 8); (void)hipMemset(tr, 0, nt_ * 8);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_b16_trace), &tr, sizeof(tr));
    float* sc = nullptr;
    if (scale) { (void)hipMalloc(&sc, (size_t)(M / hw + 1) * K * 4); (void)hipMemset(sc, 0, (size_t)(M / hw + 1) * K * 4); }
    PwParams p{A, nullptr, bias, sc, nullptr, out, M, N, K, hw, act, nt, wm == 1 ? 10 : 9};
    p.prec = 0;
    const int bm = 64 * wm, nblk_n = (N + nt * 16 - 1) / (nt * 16);
    const unsigned nblk = (unsigned)((M + bm - 1) / bm) * nblk_n;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch_pw_b16(p, Wimg, nt, wm, Npad, nblk_n, nblk, 0);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    launch_pw_b16(p, Wimg, nt, wm, Npad, nblk_n, nblk, 0);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(nt_);
    (void)hipMemcpy(h.data(), tr, nt_ * 8, hipMemcpyDeviceToHost);
    double pro = 0, bar0 = 0, wst = 0, work = 0, bar = 0, epi = 0, total = 0; int nw = 0;
    for (int w = 0; w < 64 * 4; w++) {
        const long long* t = &h[(size_t)w * B16_TRACE_SLOTS];
        if (!t[0] || !t[4]) continue;
        nw++;
        pro += t[1] - t[0]; bar0 += t[2] - t[1]; epi += t[4] - t[3]; total += t[4] - t[0];
        for (int sl = 0; sl < (nslab + spb - 1) / spb && 11 + 4 * sl < B16_TRACE_SLOTS; sl++) {
            wst += t[9 + 4 * sl] - t[8 + 4 * sl];
            work += t[10 + 4 * sl] - t[9 + 4 * sl];
            bar += t[11 + 4 * sl] - t[10 + 4 * sl];
        }
    }
    printf("%-12s spb=%d M=%d N=%d K=%d nt=%d wm=%d blocks=%u: %.1f us %.1f TF | per wave (clock64 ticks, %d waves): total %.0f = first loads + store %.0f + barrier %.0f "
           "+ %d slabs [weight store (waits for its loads) %.0f, split + reads + MFMA issue %.0f, barrier %.0f] + epilogue %.0f\n",
           name, spb, M, N, K, nt, wm, nblk, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12, nw, total / nw, pro / nw, bar0 / nw, nslab, wst / nw, work / nw,
           bar / nw, epi / nw);
    hipFree(A); hipFree(Wimg); hipFree(bias); hipFree(out); hipFree(tr);
}
int main() {
    run("b13/expand", 12288, 1152, 192, 4, 2, ACT_SWISH);
    run("b13/expand", 12288, 1152, 192, 4, 1, ACT_SWISH);
    run("b13/project", 12288, 192, 1152, 3, 1, ACT_NONE, 48, true);
    run("b10/project", 49152, 112, 672, 4, 2, ACT_NONE, 192, true);
    run("b8/project", 49152, 80, 480, 5, 1, ACT_NONE, 192, true);
    run("top", 12288, 1024, 320, 4, 2, ACT_SWISH);
    return 0;
}
