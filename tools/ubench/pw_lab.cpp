// Late-layer GEMM lab: every split-bf16 kernel form of the library on the v2.4 stack's pointwise shapes at batch 256, through
// the library's own dispatcher (launch_pw_bx3) - bitwise comparison against the tiled k_pw_bx3 and two clocks per candidate
// (ten launches back to back, best of five isolated launches).  Links the built library; no model, no Python.
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -I birdnet-go_amd/csrc -o tools/ubench/bin/pw_lab tools/ubench/pw_lab.cpp \
//         -L birdnet-go_amd/lib -lbnhip -Wl,-rpath,'$ORIGIN/../../../birdnet-go_amd/lib'
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "kernels.h"
using namespace bnhip;

struct Shape { const char* name; int M, N, K, HW, act; bool scale, res; bool f32 = false; };
struct Cand { int wm, nt; };

static float* dev_f(const std::vector<float>& h) {
    float* d; (void)hipMalloc(&d, h.size() * 4); (void)hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice); return d;
}

int main(int argc, char** argv) {
    const char* only = argc > 1 && strcmp(argv[1], "--fuzz") ? argv[1] : nullptr;
    const bool fuzz_mode = argc > 1 && !strcmp(argv[1], "--fuzz");
    const int only_wm = argc > 3 && !fuzz_mode ? atoi(argv[2]) : 0, only_nt = argc > 3 && !fuzz_mode ? atoi(argv[3]) : 0;      // pw_lab <shape> <wm> <nt>: one candidate
    const Shape shapes[] = {
        {"b13/expand", 12288, 1152, 192, 48, ACT_SWISH, false, false},
        {"top", 12288, 1024, 320, 48, ACT_SWISH, false, false},
        {"b13/project", 12288, 192, 1152, 48, ACT_NONE, true, true},
        {"b16/project", 12288, 320, 1152, 48, ACT_NONE, true, false},
        {"b12/project", 12288, 192, 672, 48, ACT_NONE, true, false},
        {"b10/project", 49152, 112, 672, 192, ACT_NONE, true, true},
        {"b9/project", 49152, 112, 480, 192, ACT_NONE, true, false},
        {"b8/project", 49152, 80, 480, 192, ACT_NONE, true, true},
        {"b6/project", 49152, 80, 240, 192, ACT_NONE, true, false},
        {"b5/project", 196608, 40, 240, 768, ACT_NONE, true, true},
        {"b4/project", 196608, 40, 144, 768, ACT_NONE, true, false},
        {"dense", 256, 6522, 1024, 1, ACT_NONE, false, false},
        // small calls (one and eight clips): wm = 14 is k_pw_lat (what launch_pw_bx3 takes for them), the others the tiled kernels with shrunk grids
        {"b13/project@1", 48, 192, 1152, 48, ACT_NONE, true, true},
        {"b13/project@8", 384, 192, 1152, 48, ACT_NONE, true, true},
        {"b16/project@1", 48, 320, 1152, 48, ACT_NONE, true, false},
        {"b10/project@1", 192, 112, 672, 192, ACT_NONE, true, true},
        {"b10/project@8", 1536, 112, 672, 192, ACT_NONE, true, true},
        {"b8/project@1", 192, 80, 480, 192, ACT_NONE, true, true},
        {"b13/project@16", 768, 192, 1152, 48, ACT_NONE, true, true},
        {"b13/project@32", 1536, 192, 1152, 48, ACT_NONE, true, true},
        {"b13/project@64", 3072, 192, 1152, 48, ACT_NONE, true, true},
        {"b13/project@128", 6144, 192, 1152, 48, ACT_NONE, true, true},
        {"b10/project@16", 3072, 112, 672, 192, ACT_NONE, true, true},
        {"b10/project@32", 6144, 112, 672, 192, ACT_NONE, true, true},
        {"b10/project@64", 12288, 112, 672, 192, ACT_NONE, true, true},
        {"dense@64", 64, 6522, 1024, 1, ACT_NONE, false, false},
        {"dense@1", 1, 6522, 1024, 1, ACT_NONE, false, false},
        {"dense@8", 8, 6522, 1024, 1, ACT_NONE, false, false},
        // the f32-MFMA family (HBM-bound early projections): wm 1 / 2 k_pw_gemm 64- / 128-row tiles, 3 / 4 k_pw_pipe
        {"b1/project", 3145728, 16, 32, 12288, ACT_NONE, true, false, true},
        {"b2/project", 786432, 24, 96, 3072, ACT_NONE, true, false, true},
        {"b3/project", 786432, 24, 144, 3072, ACT_NONE, true, true, true},
    };
    const Cand cands32[] = {{1, 1}, {1, 2}, {2, 1}, {2, 2}, {3, 2}, {4, 2}};
    const Cand cands[] = {{6, 4}, {6, 3}, {6, 6}, {6, 8}, {5, 3}, {5, 5}, {5, 7}, {8, 4}, {9, 4}, {9, 6}, {10, 3}, {10, 5}, {10, 7},
                          {12, 4}, {12, 6}, {12, 8}, {14, 0}};
    std::mt19937 rng(1234);
    std::normal_distribution<float> nd(0.f, 1.f);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    size_t fz_cmp = 0, fz_bad = 0, fz_ws = 0, fz_lat = 0;
    auto run = [&](const Shape& sh, bool fuzz) {
        const int M = sh.M, N = sh.N, K = sh.K, B = (M + sh.HW - 1) / sh.HW;
        std::vector<float> A((size_t)M * K), W((size_t)N * K), bias(N), sc((size_t)B * K), res((size_t)M * N);
        for (auto& v : A) v = nd(rng);
        for (auto& v : W) v = nd(rng) * 0.05f;
        for (auto& v : bias) v = nd(rng) * 0.1f;
        for (auto& v : sc) v = 0.5f + 0.5f * std::fabs(nd(rng)) * 0.3f;
        for (auto& v : res) v = nd(rng);
        const std::vector<uint16_t> img = pw_bx3_image(W.data(), N, K);
        uint16_t* dimg; (void)hipMalloc(&dimg, img.size() * 2); (void)hipMemcpy(dimg, img.data(), img.size() * 2, hipMemcpyHostToDevice);
        float *dA = dev_f(A), *dW = dev_f(W), *db = dev_f(bias), *ds = sh.scale ? dev_f(sc) : nullptr, *dr = sh.res ? dev_f(res) : nullptr;
        float *dref, *dout;
        (void)hipMalloc(&dref, (size_t)M * N * 4); (void)hipMalloc(&dout, (size_t)M * N * 4);
        auto params = [&](int wm, int nt, float* out) {
            PwParams p{dA, dW, db, ds, dr, out, M, N, K, sh.HW, sh.act, nt, wm == 14 ? 6 : wm};
            p.prec = 0;
            p.sw = wm == 14 ? 0 : PW_SW_LAT_OFF;          // (the tiled candidates and the reference must not be routed to k_pw_lat)
            return p;
        };
        auto launch = [&](const PwParams& p) { if (sh.f32) launch_pw_gemm(p, 0); else launch_pw_bx3(p, dimg, 0); };
        { PwParams p = sh.f32 ? params(1, 2, dref) : params(6, 4, dref); launch(p); (void)hipDeviceSynchronize(); }
        std::vector<float> href((size_t)M * N), hout((size_t)M * N);
        (void)hipMemcpy(href.data(), dref, href.size() * 4, hipMemcpyDeviceToHost);
        // sanity of the reference itself against a double-precision dot product on a few outputs
        double worst = 0;
        for (int t = 0; t < 64; t++) {
            const int m = (int)(rng() % M), n = (int)(rng() % N);
            double acc = 0;
            for (int k = 0; k < K; k++) acc += (double)(A[(size_t)m * K + k] * (sh.scale ? sc[(size_t)(m / sh.HW) * K + k] : 1.f)) * W[(size_t)n * K + k];
            acc += bias[n];
            if (sh.act == ACT_SWISH) acc = acc / (1.0 + std::exp(-acc));
            if (sh.res) acc += res[(size_t)m * N + n];
            worst = std::max(worst, std::fabs(acc - href[(size_t)m * N + n]));
        }
        if (!fuzz || worst > 1e-3)
            printf("== %-12s M=%d N=%d K=%d HW=%d scale=%d res=%d  (reference tiled kernel vs fp64 on 64 outputs: max |d| %.2e)\n", sh.name, M, N, K, sh.HW,
                   (int)sh.scale, (int)sh.res, worst);
        if (fuzz && worst > 1e-3) fz_bad++;
        std::vector<Cand> cl;
        if (sh.f32) cl.assign(std::begin(cands32), std::end(cands32)); else cl.assign(std::begin(cands), std::end(cands));
        for (const Cand& c : cl) {
            if (only_wm && (c.wm != only_wm || c.nt != only_nt)) continue;
            PwParams p = params(c.wm, c.nt, dout);
            if (sh.f32 && c.nt <= 8 && c.nt * 16 > (N + 15) / 16 * 16) continue;
            if (sh.f32 && c.wm > 2 && !pw_pipe_ok(c.nt, c.wm - 2, K)) continue;
            if (fuzz && c.wm == 12) p.sw |= PW_SW_WS_FORCE;          // (whatever the size: pw_ws_fills is a speed rule)
            if (!sh.f32 && c.wm == 12 && !pw_ws_ok(p)) continue;
            if (c.wm == 14 && (sh.f32 || !pw_lat_ok(p))) continue;
            if (!sh.f32 && c.wm == 8 && !pw_bx3p_ok(c.nt, 2, K)) continue;
            if (!fuzz && !sh.f32 && c.wm != 12 && c.wm != 14 && c.nt * 16 > (N + 15) / 16 * 16 * 13 / 10 && c.nt > 1) continue;
            if (fuzz && (c.wm == 6 || c.wm == 8) && c.nt != 3) continue;      // (a sample of the tiled forms is enough)
            (void)hipMemset(dout, 0xff, (size_t)M * N * 4);
            launch(p);
            hipError_t err = hipDeviceSynchronize();
            if (err != hipSuccess) { printf("   wm=%2d nt=%d: LAUNCH FAILED %s\n", c.wm, c.nt, hipGetErrorString(err)); (void)hipGetLastError(); continue; }
            (void)hipMemcpy(hout.data(), dout, hout.size() * 4, hipMemcpyDeviceToHost);
            size_t bad = 0, first = 0;
            for (size_t i = 0; i < hout.size(); i++)
                if (memcmp(&hout[i], &href[i], 4)) { if (!bad) first = i; bad++; }
            if (fuzz) {
                fz_cmp++; fz_ws += c.wm == 12; fz_lat += c.wm == 14;
                if (bad) { fz_bad++; printf("   MISMATCH M=%d N=%d K=%d HW=%d scale=%d res=%d act=%d wm=%d nt=%d: %zu of %zu, first at row %zu col %zu: %g vs %g\n", M, N, K, sh.HW,
                                            (int)sh.scale, (int)sh.res, sh.act, c.wm, c.nt, bad, hout.size(), first / N, first % N, hout[first], href[first]); }
                continue;
            }
            launch(p);
            (void)hipEventRecord(e0, 0);
            for (int r = 0; r < 10; r++) launch(p);
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float b2b; (void)hipEventElapsedTime(&b2b, e0, e1);
            float iso = 1e30f;
            for (int r = 0; r < 5; r++) {
                (void)hipEventRecord(e0, 0); launch(p); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
                float t; (void)hipEventElapsedTime(&t, e0, e1); iso = std::min(iso, t);
            }
            printf("   wm=%2d nt=%d: %7.1f us b2b  %7.1f us iso  %6.1f TF   %s", c.wm, c.nt, b2b * 100, iso * 1e3, 2.0 * M * N * K / (b2b * 1e-4) / 1e12,
                   bad ? "MISMATCH" : "bit-identical");
            if (bad) printf(" (%zu of %zu, first at row %zu col %zu: %g vs %g)", bad, hout.size(), first / N, first % N, hout[first], href[first]);
            printf("\n");
            fflush(stdout);
        }
        hipFree(dimg); hipFree(dA); hipFree(dW); hipFree(db); if (ds) hipFree(ds); if (dr) hipFree(dr); hipFree(dref); hipFree(dout);
    };
    if (argc > 1 && !strcmp(argv[1], "--fuzz")) {
        // pw_lab --fuzz <seed> <count>: random shapes (K tails, ragged N, N % 4 != 0, one row, scale / residual / swish on and off) through
        // every kernel form of the split-bf16 family - k_pw_ws and k_pw_lat wherever they accept the layer, whatever its size - bitwise
        // against the tiled k_pw_bx3.  Prints mismatches and one summary line (tests/test_pw_lab.py asserts on it).
        std::mt19937 fr((unsigned)(argc > 2 ? atoi(argv[2]) : 1));
        const int count = argc > 3 ? atoi(argv[3]) : 24;
        for (int i = 0; i < count; i++) {
            Shape sh{"fuzz", 0, 0, 0, 0, 0, false, false};
            const int kind = (int)(fr() % 4);
            sh.HW = 16 * (1 + (int)(fr() % 12)) + (fr() % 3 == 0 ? 7 : 0);
            sh.M = kind == 0 ? 1 + (int)(fr() % 40) : sh.HW * (1 + (int)(fr() % 12)) + (fr() % 4 == 0 ? 5 : 0);
            sh.N = kind == 3 ? 2 * (3 + (int)(fr() % 300)) + 1 * (int)(fr() % 2) : 4 * (2 + (int)(fr() % 90));
            sh.K = 4 * (4 + (int)(fr() % (kind == 1 ? 60 : 320)));
            sh.act = fr() % 2 ? ACT_SWISH : ACT_NONE;
            sh.scale = fr() % 2 == 0;
            sh.res = fr() % 3 == 0;
            if (kind == 1) {                                 // k_pw_ws territory: 3-6, 8 or 10 slabs (with and without a K tail), wide N, no scale
                static const int ks[] = {96, 100, 128, 136, 160, 172, 192, 232, 256, 300, 320};
                sh.K = ks[fr() % 11]; sh.N = 4 * (16 + (int)(fr() % 80)); sh.scale = false;
                sh.M = 16 * (17 + (int)(fr() % 100)) + (fr() % 3 == 0 ? 9 : 0);
            }
            run(sh, true);
        }
        printf("fuzz: %d shapes, %zu comparisons (%zu on k_pw_ws, %zu on k_pw_lat), %zu mismatches\n", count, fz_cmp, fz_ws, fz_lat, fz_bad);
        return fz_bad ? 1 : 0;
    }
    for (const Shape& sh : shapes) {
        if (only && !strstr(sh.name, only)) continue;
        run(sh, false);
    }
    return 0;
}
