// Where does a k_pw_ws wave spend its time?  Builds the library kernel with WS_TRACE (shader-clock stamps at the phase boundaries
// of every wave, plus the 100 MHz wall clock at both ends to read the shader clock) and runs it on the late 6x expand of the v2.4
// stack at batch 256.  Operands are random (values do not change the instruction stream, but all-zero data lets the chip clock
// higher).  NOT linked against the built library: an executable that defines kernels of the same name as a shared library it loads
// interposes the library's host stubs, and the runtime may then launch the library's (untraced) code object for them.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form -DWS_TRACE \
//         -I birdnet-go_amd/csrc -o tools/ubench/bin/ws_trace tools/ubench/ws_trace.hip
#include "../../birdnet-go_amd/csrc/pw_ws.hip"

#include <cstdio>
#include <vector>
using namespace bnhip;

static void run(const char* name, int M, int N, int K, int nt, int act) {
    // a weight image of random small bf16 values (three planes of decreasing magnitude, as a real split has)
    const int Npad_ = (N + 15) / 16 * 16, nslab = (K + 31) / 32;
    std::vector<uint16_t> img((size_t)nslab * 12 * Npad_ * 8);
    unsigned rs = 12345u;
    for (size_t i = 0; i < img.size(); i++) {
        rs = rs * 1664525u + 1013904223u;
        const int plane = (int)((i / ((size_t)4 * Npad_ * 8)) % 3);
        const unsigned e = 0x3c00u - 0x0400u * (unsigned)plane;                 // exponents 2^-7, 2^-15, 2^-23
        img[i] = (uint16_t)(((rs >> 16) & 0x807fu) | e);
    }
    uint16_t* dimg; (void)hipMalloc(&dimg, img.size() * 2); (void)hipMemcpy(dimg, img.data(), img.size() * 2, hipMemcpyHostToDevice);
    float *A, *bias, *out;
    (void)hipMalloc(&A, (size_t)M * K * 4); (void)hipMalloc(&bias, (size_t)N * 4); (void)hipMalloc(&out, (size_t)M * N * 4);
    {
        std::vector<float> hA((size_t)M * K);
        for (auto& v : hA) { rs = rs * 1664525u + 1013904223u; v = ((int)(rs >> 8) - (1 << 23)) * (1.0f / (1 << 23)); }
        (void)hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    }
    (void)hipMemset(bias, 0, (size_t)N * 4);
    const size_t nw = 256 * 8, ns = 32;
    long long* tr; (void)hipMalloc(&tr, nw * ns * 8); (void)hipMemset(tr, 0, nw * ns * 8);
    PwParams p{A, nullptr, bias, nullptr, reinterpret_cast<const float*>(tr), out, M, N, K, 48, act, nt, 12};
    p.prec = 0;
    if (!pw_ws_ok(p)) { printf("%s: not a k_pw_ws layer\n", name); return; }
    const int Npad = (N + 15) / 16 * 16;
    launch_pw_ws(p, dimg, Npad, 0); (void)hipDeviceSynchronize();
    (void)hipMemset(tr, 0, nw * ns * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0); launch_pw_ws(p, dimg, Npad, 0); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(nw * ns);
    (void)hipMemcpy(h.data(), tr, nw * ns * 8, hipMemcpyDeviceToHost);
    double wl = 0, pro = 0, tot = 0, mf[8] = {0}, ep[8] = {0}, ticks = 0, wall = 0; int cnt[8] = {0}, n = 0;
    for (size_t w = 0; w < nw; w++) {
        const long long* t = &h[w * ns];
        if (!t[0] || !t[2]) continue;
        n++; wl += t[1] - t[0]; pro += t[2] - t[1];
        long long last = t[2];
        for (int k = 0; k < 8 && t[3 + 2 * k] && t[4 + 2 * k]; k++) { mf[k] += t[3 + 2 * k] - (k ? t[2 + 2 * k] : t[2]); ep[k] += t[4 + 2 * k] - t[3 + 2 * k]; cnt[k]++; last = t[4 + 2 * k]; }
        tot += last - t[0];
        if (t[31] > t[30]) { ticks += (double)(t[29] - t[0]); wall += (double)(t[31] - t[30]); }
    }
    printf("%-10s M=%d N=%d K=%d nt=%d: %.1f us | per wave (shader-clock ticks, %d waves, %.0f MHz): total %.0f = weight columns + barrier %.0f + first loads + split %.0f", name, M, N, K, nt,
           ms * 1e3, n, wall > 0 ? ticks / wall * 100 : 0.0, tot / n, wl / n, pro / n);
    for (int k = 0; k < 8 && cnt[k]; k++) printf(" | item %d (%d waves): slabs %.0f, epilogue %.0f", k, cnt[k], mf[k] / cnt[k], ep[k] / cnt[k]);
    printf("\n");
    hipFree(dimg); hipFree(A); hipFree(bias); hipFree(out); hipFree(tr);
}
int main() {
    run("b13/expand", 12288, 1152, 192, 6, ACT_SWISH);
    run("b13/expand", 12288, 1152, 192, 8, ACT_SWISH);
    run("top", 12288, 1024, 320, 4, ACT_SWISH);
    return 0;
}
