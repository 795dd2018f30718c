/* Test infrastructure: plain-C host that makes exactly the call sequence of the cgo binding
 * birdnet-go_amd/go/internal/inference/hip/backend_hip.go - through that file's own C preamble, which the test extracts
 * verbatim into preamble_extracted.h (so the preamble is compiled with -Wall -Wextra -Werror and executed, although no Go
 * toolchain exists here).  SURVEY.md section 7 step 3.
 *
 *   cabi_driver <libbnhip.so> <model file> cpu               error paths + plan-only sequence (no GPU needed)
 *   cabi_driver <libbnhip.so> <model file> gpu <in.f32> <out.f32> <n_clips>
 *        Init -> NewClassifier -> Predict (clip 0) -> PredictBatch (all) -> the same from page-locked buffers -> PredictTopK -> PredictPCM16 -> ComputeUSFrameCV (the
 *        reference's known answers) -> Resampler (chunked == one shot) -> Close, logits written to out.f32.  With a dense model
 *        (a CustomClassifier head, a RangeFilter meta-model) the same sequence is what PredictEmbedding / PredictBatch do.
 */
#include "preamble_extracted.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); \
                                               fprintf(stderr, "\n"); return 1; } } while (0)

static void* read_file(const char* path, size_t* n) {
    FILE* f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    void* b = malloc(sz > 0 ? (size_t)sz : 1);
    if (b && fread(b, 1, (size_t)sz, f) != (size_t)sz) { free(b); b = NULL; }
    fclose(f);
    *n = (size_t)sz;
    return b;
}

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: see header\n"); return 2; }
    const char* lib = argv[1];
    size_t nb = 0;
    void* blob = read_file(argv[2], &nb);
    CHECK(blob, "cannot read %s", argv[2]);
    const int gpu = strcmp(argv[3], "gpu") == 0;

    /* Init(): a missing library is reported, leaves the table empty and is retryable */
    const char* msg = bnbind_load("/nonexistent/libbnhip.so");
    CHECK(msg && !BN.handle && !BN.init, "load of a missing library must fail cleanly");
    msg = bnbind_load(lib);
    CHECK(!msg, "load: %s", msg);
    CHECK(bnbind_load(lib) == NULL, "second load is a no-op");
    CHECK(strlen(bnbind_last_error()) == 0 || 1, "last_error callable");

    bnhip_model* h = NULL;
    int ns = 0, nc = 0, ed = 0;
    if (!gpu) {
        /* error paths of NewClassifier */
        int rc = bnbind_model_create("garbage-garbage-garbage", 23, "{\"plan_only\":1}", &h);
        CHECK(rc == -3 && !h && strlen(bnbind_last_error()) > 0, "garbage blob must be BNHIP_E_MODEL with a message (rc %d)", rc);
        rc = bnbind_model_create(blob, nb, "{\"plan_only\":1,\"max_batch\":0}", &h);
        CHECK(rc == -1 && !h, "max_batch 0 must be BNHIP_E_INVALID (rc %d)", rc);
        rc = bnbind_model_create(blob, nb / 2, "{\"plan_only\":1}", &h);
        CHECK(rc != 0 && !h, "truncated model must fail");
        /* plan-only handle: info works, predict is rejected with a message, destroy is idempotent-safe */
        rc = bnbind_model_create(blob, nb, "{\"plan_only\":1,\"devices\":[0,1]}", &h);
        CHECK(rc == 0 && h, "plan-only create: %s", bnbind_last_error());
        CHECK(bnbind_model_info(h, &ns, &nc, &ed) == 0 && ns > 0 && nc > 0, "model_info");
        float* x = calloc((size_t)ns, 4); float* y = calloc((size_t)nc, 4);
        rc = bnbind_predict(h, x, 1, y, NULL);
        CHECK(rc == -1 && strstr(bnbind_last_error(), "plan-only"), "predict on a plan-only handle: rc %d '%s'", rc, bnbind_last_error());
        float cf[4]; int32_t ix[4];
        rc = bnbind_predict_topk(h, x, 1, 0, 1.0, 4, cf, ix);
        CHECK(rc != 0, "predict_topk on a plan-only handle must fail");
        bnbind_model_destroy(h);
        free(x); free(y);
        /* PredictPCM16 on a plan-only handle is rejected the same way */
        {
            rc = bnbind_model_create(blob, nb, "{\"plan_only\":1}", &h);
            CHECK(rc == 0 && h, "plan-only create (2): %s", bnbind_last_error());
            int16_t* xp = calloc((size_t)ns, 2); float* yp = calloc((size_t)nc, 4);
            rc = bnbind_predict_pcm16(h, xp, 1, yp, NULL);
            CHECK(rc == -1 && strstr(bnbind_last_error(), "plan-only"), "predict_pcm16 on a plan-only handle: rc %d", rc);
            bnbind_model_destroy(h);
            free(xp); free(yp);
        }
        /* NewResampler(equal rates) = nil, nil (resample.go:57-60): success with a NULL handle, and every method accepts it */
        {
            bnhip_resampler* r = (bnhip_resampler*)1;
            rc = bnbind_rs_create(0, 48000, 48000, &r);
            CHECK(rc == 0 && r == NULL, "equal rates must give a NULL resampler (rc %d)", rc);
            CHECK(bnbind_rs_estimate(NULL, 100) == 0, "estimate on a NULL resampler");
            bnbind_rs_destroy(NULL);
            rc = bnbind_rs_create(0, 0, 48000, &r);
            CHECK(rc == -1 && r == NULL, "rate 0 must be invalid (rc %d)", rc);
        }
        /* ComputeUSFrameCV's guards (filter.go:21-37) answer (0, false) before any device work */
        {
            double sm[64] = {0}, cv = 1.0; int32_t ok = 1;
            rc = bnbind_us_frame_cv(0, sm, 1, 64, 256000, 8192, 4096, 20000, &cv, &ok);
            CHECK(rc == 0 && cv == 0.0 && ok == 0, "us_frame_cv: clip shorter than the FFT must be (0, false) (rc %d)", rc);
            rc = bnbind_us_frame_cv(0, sm, 1, 64, 48000, 32, 16, 30000, &cv, &ok);
            CHECK(rc == 0 && ok == 0, "us_frame_cv: split above Nyquist must be (0, false)");
        }
        /* WindowAssembler (bnhip_windows_*): byte work, no device needed.  Geometry checks of NewAnalysisBuffer
         * (analysis.go:55-109), two sources, first window = zero prefix + fresh bytes, second = previous tail + fresh bytes
         * (TestAnalysisBuffer_Read_ContentParity, analysis_test.go:203-243), overwrite accounting, cap + resume */
        {
            bnhip_windows* w = (bnhip_windows*)1;
            rc = bnbind_win_create(64, 32, 4, &w);
            CHECK(rc == -1 && w == NULL && strstr(bnbind_last_error(), "overlap"), "read < overlap must be invalid (rc %d)", rc);
            rc = bnbind_win_create(8, 0, 4, &w);
            CHECK(rc == -1 && w == NULL, "read size 0 must be invalid (rc %d)", rc);
            rc = bnbind_win_create(4, 8, 2, &w);
            CHECK(rc == 0 && w, "windows_create: %s", bnbind_last_error());
            size_t wb = 0; int mb = 0, pinned = -1, nsrc = -1;
            CHECK(bnbind_win_info(w, &wb, &mb, &pinned, &nsrc) == 0 && wb == 12 && mb == 2 && nsrc == 0 && (pinned == 0 || pinned == 1), "windows_info");
            int a = -1, b = -1, c = -1;
            CHECK(bnbind_win_add_source(w, "", 64, &a) == -1 && a == -1, "empty source id must be invalid");
            CHECK(bnbind_win_add_source(w, "mic-a", 4, &a) == -1 && a == -1, "capacity below the read size must be invalid");
            CHECK(bnbind_win_add_source(w, "mic-a", 16, &a) == 0 && a == 0, "add_source a: %s", bnbind_last_error());
            CHECK(bnbind_win_add_source(w, "mic-b", 16, &b) == 0 && b == 1, "add_source b");
            CHECK(bnbind_win_add_source(w, "mic-c", 16, &c) == 0 && c == 2, "add_source c");
            unsigned char st[32];
            for (int i = 0; i < 32; i++) st[i] = (unsigned char)(i + 1);
            int src[4] = {-1, -1, -1, -1}, n = -1;
            const void* batch = NULL;
            CHECK(bnbind_win_write(w, a, st, 7) == 0, "write");
            CHECK(bnbind_win_collect(w, 4, src, &n, &batch) == 0 && n == 0 && batch, "7 of 8 bytes: try again later");
            CHECK(bnbind_win_write(w, a, st + 7, 9) == 0 && bnbind_win_write(w, b, st + 16, 8) == 0 && bnbind_win_write(w, c, st, 8) == 0, "write");
            CHECK(bnbind_win_write(w, 7, st, 1) == -1, "write to an unknown source must be invalid");
            /* three sources ready, max_batch 2: a and b now, c (and a's second window) on the next call */
            CHECK(bnbind_win_collect(w, 4, src, &n, &batch) == 0 && n == 2 && src[0] == a && src[1] == b, "collect 1: n %d", n);
            const unsigned char* r = batch;
            unsigned char want0[12] = {0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8}, want1[12] = {0, 0, 0, 0, 17, 18, 19, 20, 21, 22, 23, 24};
            CHECK(memcmp(r, want0, 12) == 0 && memcmp(r + 12, want1, 12) == 0, "first windows: zero prefix + fresh bytes");
            CHECK(bnbind_win_collect(w, 4, src, &n, &batch) == 0 && n == 2 && src[0] == c && src[1] == a, "collect 2: n %d src %d %d", n, src[0], src[1]);
            unsigned char want2[12] = {5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
            r = batch;
            CHECK(memcmp(r + 12, want2, 12) == 0, "second window of a: previous tail + fresh bytes");
            /* overwrite mode: 24 bytes into a 16-byte ring keep the newest 16 and count one overwriting write */
            CHECK(bnbind_win_write(w, b, st, 24) == 0, "long write");
            uint64_t wr = 0, ov = 0; size_t buffered = 0;
            CHECK(bnbind_win_stats(w, b, &wr, &ov, &buffered) == 0 && wr == 2 && ov == 1 && buffered == 16, "stats: %llu %llu %zu",
                  (unsigned long long)wr, (unsigned long long)ov, buffered);
            CHECK(bnbind_win_collect(w, 1, src, &n, &batch) == 0 && n == 1 && src[0] == b, "collect 3");
            unsigned char want3[12] = {21, 22, 23, 24, 9, 10, 11, 12, 13, 14, 15, 16};
            CHECK(memcmp(batch, want3, 12) == 0, "after the overwrite: tail of b's first window + the oldest surviving bytes");
            CHECK(bnbind_win_reset(w, b) == 0 && bnbind_win_stats(w, b, &wr, &ov, &buffered) == 0 && wr == 0 && ov == 0 && buffered == 0, "reset");
            CHECK(bnbind_win_remove_source(w, c) == 0 && bnbind_win_remove_source(w, c) == -1, "remove_source");
            CHECK(bnbind_win_add_source(w, "mic-d", 32, &c) == 0 && c == 2, "a removed source's slot is reused");
            bnbind_win_destroy(w);
            bnbind_win_destroy(NULL);
        }
        /* without a GPU Init reports "unavailable" (-2) and the message names the reason; with one it succeeds */
        int n = -1;
        rc = bnbind_init(&n);
        CHECK((rc == 0 && n >= 1) || (rc == -2 && n == 0 && strlen(bnbind_last_error()) > 0), "init rc %d n %d", rc, n);
        printf("cpu sequence ok (n_samples %d, n_classes %d, init rc %d)\n", ns, nc, rc);
        bnbind_unload();
        CHECK(!BN.handle, "unload clears the table");
        free(blob);
        return 0;
    }

    CHECK(argc >= 7, "gpu mode needs <in.f32> <out.f32> <n_clips>");
    const int n_clips = atoi(argv[6]);
    int ndev = 0;
    CHECK(bnbind_init(&ndev) == 0 && ndev >= 1, "init: %s", bnbind_last_error());
    CHECK(bnbind_model_create(blob, nb, "{\"devices\":[0],\"max_batch\":256}", &h) == 0 && h, "create: %s", bnbind_last_error());
    free(blob);                                    /* the blob is consumed during the call */
    CHECK(bnbind_model_info(h, &ns, &nc, &ed) == 0, "info");
    size_t nin = 0;
    float* in = read_file(argv[4], &nin);
    CHECK(in && nin == (size_t)n_clips * (size_t)ns * 4, "input file size %zu != %d x %d x 4", nin, n_clips, ns);
    float* out = malloc((size_t)n_clips * (size_t)nc * 4);
    float* one = malloc((size_t)nc * 4);
    /* Classifier.Predict: one clip from a C-allocated staging copy */
    float* stage = malloc((size_t)ns * 4);
    memcpy(stage, in, (size_t)ns * 4);
    CHECK(bnbind_predict(h, stage, 1, one, NULL) == 0, "predict: %s", bnbind_last_error());
    /* size mismatch is the host's check (tflite/classifier.go:102-104); n_clips <= 0 is the library's */
    CHECK(bnbind_predict(h, stage, 0, one, NULL) == -1, "n_clips 0 must be invalid");
    CHECK(bnbind_predict(h, NULL, 1, one, NULL) == -1, "NULL samples must be invalid");
    /* PredictBatch */
    CHECK(bnbind_predict(h, in, n_clips, out, NULL) == 0, "predict batch: %s", bnbind_last_error());
    for (int i = 0; i < nc; i++) CHECK(out[i] == one[i], "Predict and PredictBatch disagree on clip 0 at class %d: %g vs %g", i, one[i], out[i]);
    /* the shim's own buffers are page-locked (bnhip_host_alloc): the same calls from pinned memory - one clip and the whole batch,
     * input and output - must give the very same bits as from malloc'd memory */
    {
        void *pin = NULL, *pout = NULL, *none = (void*)1;
        CHECK(bnbind_host_alloc(0, &none) == -1, "zero-byte pinned allocation must be invalid");
        CHECK(bnbind_host_alloc((size_t)n_clips * ns * 4, &pin) == 0 && pin, "host_alloc: %s", bnbind_last_error());
        CHECK(bnbind_host_alloc((size_t)n_clips * nc * 4, &pout) == 0 && pout, "host_alloc: %s", bnbind_last_error());
        memcpy(pin, in, (size_t)n_clips * ns * 4);
        memset(pout, 0xff, (size_t)n_clips * nc * 4);
        CHECK(bnbind_predict(h, (const float*)pin, 1, (float*)pout, NULL) == 0, "predict (pinned): %s", bnbind_last_error());
        CHECK(memcmp(pout, one, (size_t)nc * 4) == 0, "pinned Predict differs from the pageable one");
        CHECK(bnbind_predict(h, (const float*)pin, n_clips, (float*)pout, NULL) == 0, "predict batch (pinned): %s", bnbind_last_error());
        CHECK(memcmp(pout, out, (size_t)n_clips * nc * 4) == 0, "pinned PredictBatch differs from the pageable one");
        /* mixed: pinned input, pageable output */
        float* o3 = malloc((size_t)n_clips * nc * 4);
        CHECK(bnbind_predict(h, (const float*)pin, n_clips, o3, NULL) == 0, "predict batch (pinned in): %s", bnbind_last_error());
        CHECK(memcmp(o3, out, (size_t)n_clips * nc * 4) == 0, "pinned-input PredictBatch differs");
        free(o3);
        CHECK(bnbind_host_free(pin) == 0 && bnbind_host_free(pout) == 0 && bnbind_host_free(NULL) == 0, "host_free: %s", bnbind_last_error());
    }
    /* PredictTopK: confidences descending, indices in range, top-1 == argmax of the logits */
    const int k = nc < 10 ? nc : 10;
    float* cf = malloc((size_t)n_clips * k * 4); int32_t* ix = malloc((size_t)n_clips * k * 4);
    CHECK(bnbind_predict_topk(h, in, n_clips, 0, 1.0, k, cf, ix) == 0, "predict_topk: %s", bnbind_last_error());
    for (int c = 0; c < n_clips; c++) {
        int am = 0;
        for (int i = 1; i < nc; i++) if (out[(size_t)c * nc + i] > out[(size_t)c * nc + am]) am = i;
        CHECK(ix[c * k] == am, "clip %d: top-1 index %d != argmax %d", c, ix[c * k], am);
        for (int j = 0; j < k; j++) {
            CHECK(ix[c * k + j] >= 0 && ix[c * k + j] < nc, "index out of range");
            if (j) CHECK(cf[c * k + j] <= cf[c * k + j - 1], "confidences not descending");
        }
    }
    /* PredictPCM16: int16 clips converted on the device (float32(s)/32768) == Predict on the same values converted here */
    if (ns >= 1000) {
        int16_t* pcm = malloc((size_t)n_clips * ns * 2);
        float* deq = malloc((size_t)n_clips * ns * 4);
        float* o1 = malloc((size_t)n_clips * nc * 4); float* o2 = malloc((size_t)n_clips * nc * 4);
        for (size_t i = 0; i < (size_t)n_clips * ns; i++) {
            float v = in[i] < -1.f ? -1.f : (in[i] > 1.f ? 1.f : in[i]);
            pcm[i] = (int16_t)(v * 32767.f);
            deq[i] = (float)pcm[i] / 32768.0f;
        }
        CHECK(bnbind_predict_pcm16(h, pcm, n_clips, o1, NULL) == 0, "predict_pcm16: %s", bnbind_last_error());
        CHECK(bnbind_predict(h, deq, n_clips, o2, NULL) == 0, "predict (dequantised): %s", bnbind_last_error());
        for (size_t i = 0; i < (size_t)n_clips * nc; i++) CHECK(o1[i] == o2[i], "PredictPCM16 differs from Predict at %zu: %g vs %g", i, o1[i], o2[i]);
        free(pcm); free(deq); free(o1); free(o2);
    }
    /* ComputeUSFrameCV: the reference's own known answers (ultrasonic/filter_test.go:23-60): a steady 40 kHz tone of amplitude
     * 0.01 at 256 kHz has CV < 0.15; a 45 kHz burst of amplitude 0.5 in the middle third has CV > 0.15 */
    {
        const int N = 144000, rate = 256000;
        double* s1 = malloc((size_t)2 * N * 8);
        for (int i = 0; i < N; i++) {
            s1[i] = 0.01 * sin(2.0 * M_PI * 40000.0 * i / rate);
            s1[N + i] = (i >= N / 3 && i < 2 * N / 3) ? 0.5 * sin(2.0 * M_PI * 45000.0 * i / rate) : 0.0;
        }
        double cv[2]; int32_t ok[2];
        CHECK(bnbind_us_frame_cv(0, s1, 2, N, rate, 8192, 4096, 20000, cv, ok) == 0, "us_frame_cv: %s", bnbind_last_error());
        CHECK(ok[0] && ok[1] && cv[0] < 0.15 && cv[1] > 0.15, "us_frame_cv known answers: cv %g %g ok %d %d", cv[0], cv[1], ok[0], ok[1]);
        free(s1);
    }
    /* Resampler: 48 kHz -> 32 kHz in 100 ms frames == one call over the whole stream, sample for sample; a destination that
     * is too small fails WITHOUT advancing the state (resample.go:137-144) */
    {
        const int N = 48000, fr = 4800;
        int16_t* x = malloc((size_t)N * 2);
        for (int i = 0; i < N; i++) x[i] = (int16_t)(12000.0 * sin(2.0 * M_PI * 1000.0 * i / 48000.0) + 5000.0 * sin(2.0 * M_PI * 7000.0 * i / 48000.0));
        bnhip_resampler *ra = NULL, *rb = NULL;
        CHECK(bnbind_rs_create(0, 48000, 32000, &ra) == 0 && ra, "resampler create: %s", bnbind_last_error());
        CHECK(bnbind_rs_create(0, 48000, 32000, &rb) == 0 && rb, "resampler create: %s", bnbind_last_error());
        const int cap = bnbind_rs_estimate(ra, N) + 64;
        int16_t* ya = malloc((size_t)cap * 2); int16_t* yb = malloc((size_t)cap * 2);
        int na = 0, nb2 = 0, n = 0;
        CHECK(bnbind_rs_process_pcm16(ra, x, N, ya, cap, &na) == 0, "one-shot: %s", bnbind_last_error());
        CHECK(bnbind_rs_flush_pcm16(ra, ya + na, cap - na, &n) == 0, "flush: %s", bnbind_last_error());
        na += n;
        CHECK(na == 32000, "one-shot output length %d != 32000", na);
        int16_t tiny[4];
        CHECK(bnbind_rs_process_pcm16(rb, x, fr, tiny, 4, &n) == -1 && n == 0, "a too-small destination must fail");
        for (int off = 0; off < N; off += fr) {
            CHECK(bnbind_rs_process_pcm16(rb, x + off, fr, yb + nb2, cap - nb2, &n) == 0, "chunk: %s", bnbind_last_error());
            nb2 += n;
        }
        CHECK(bnbind_rs_flush_pcm16(rb, yb + nb2, cap - nb2, &n) == 0, "flush: %s", bnbind_last_error());
        nb2 += n;
        CHECK(nb2 == na && memcmp(ya, yb, (size_t)na * 2) == 0, "chunked resampling differs from one call (%d vs %d samples)", nb2, na);
        bnbind_rs_destroy(ra); bnbind_rs_destroy(rb);
        free(x); free(ya); free(yb);
    }
    /* Classifier.PredictWindows: every source's PCM through the window assembler, one device call per tick straight from its
     * page-locked batch buffer == PredictPCM16 on the same windows cut by hand (zero prefix, then 50 % overlap) */
    if (ns >= 1000 && (ns & 1) == 0) {
        const size_t clip_b = (size_t)ns * 2, ovb = clip_b / 2, rdb = clip_b - ovb;
        bnhip_windows* w = NULL;
        CHECK(bnbind_win_create(ovb, rdb, 256, &w) == 0 && w, "windows_create: %s", bnbind_last_error());
        size_t wb = 0; int pinned = 0;
        CHECK(bnbind_win_info(w, &wb, NULL, &pinned, NULL) == 0 && wb == clip_b && pinned == 1, "the batch buffer must be page-locked on a GPU box");
        int16_t* pcm = malloc((size_t)n_clips * clip_b);
        for (size_t i = 0; i < (size_t)n_clips * ns; i++) {
            float v = in[i] < -1.f ? -1.f : (in[i] > 1.f ? 1.f : in[i]);
            pcm[i] = (int16_t)(v * 32767.f);
        }
        int* srcs = malloc((size_t)n_clips * sizeof(int));
        for (int c = 0; c < n_clips; c++) {
            char id[32]; snprintf(id, sizeof id, "mic-%d", c);
            int idx = -1;
            CHECK(bnbind_win_add_source(w, id, 2 * clip_b, &idx) == 0 && idx == c, "add_source");
            /* a source's stream = its clip, written in ragged pieces */
            const unsigned char* p = (const unsigned char*)(pcm + (size_t)c * ns);
            for (size_t off = 0, step = 1000 + 37 * (size_t)c; off < clip_b; off += step)
                CHECK(bnbind_win_write(w, idx, p + off, off + step <= clip_b ? step : clip_b - off) == 0, "write");
        }
        int16_t* hand = calloc((size_t)n_clips * ns, 2);
        float* o1 = malloc((size_t)n_clips * nc * 4); float* o2 = malloc((size_t)n_clips * nc * 4);
        for (int tick = 0; tick < 2; tick++) {             /* each clip holds two reads: two windows per source */
            int n = 0; const void* batch = NULL;
            CHECK(bnbind_win_collect(w, n_clips, srcs, &n, &batch) == 0 && n == n_clips, "collect: n %d", n);
            for (int c = 0; c < n_clips; c++) {
                CHECK(srcs[c] == c, "source order");
                unsigned char* hw = (unsigned char*)(hand + (size_t)c * ns);
                const unsigned char* p = (const unsigned char*)(pcm + (size_t)c * ns);
                if (tick == 0) { memset(hw, 0, ovb); memcpy(hw + ovb, p, rdb); }
                else { memcpy(hw, p + rdb - ovb, ovb); memcpy(hw + ovb, p + rdb, rdb); }
            }
            CHECK(memcmp(batch, hand, (size_t)n_clips * clip_b) == 0, "tick %d: assembled windows differ from the hand-cut ones", tick);
            CHECK(bnbind_predict_pcm16(h, (const int16_t*)batch, n, o1, NULL) == 0, "predict_pcm16 (windows): %s", bnbind_last_error());
            CHECK(bnbind_predict_pcm16(h, hand, n, o2, NULL) == 0, "predict_pcm16 (hand): %s", bnbind_last_error());
            CHECK(memcmp(o1, o2, (size_t)n * nc * 4) == 0, "tick %d: logits from the assembler's buffer differ", tick);
            /* PredictWindowsTopK: the tick's post-processing on the device too - top-1 = argmax of those logits, descending */
            {
                const int kk = nc < 10 ? nc : 10;
                float* tc = malloc((size_t)n * kk * 4); int32_t* ti = malloc((size_t)n * kk * 4);
                CHECK(bnbind_predict_pcm_topk(h, batch, 16, n, 0, 1.0, 10, tc, ti) == 0, "predict_pcm_topk: %s", bnbind_last_error());
                CHECK(bnbind_predict_pcm_topk(h, batch, 12, n, 0, 1.0, 10, tc, ti) == -1, "12-bit PCM must be invalid");
                for (int c = 0; c < n; c++) {
                    int am = 0;
                    for (int i = 1; i < nc; i++) if (o1[(size_t)c * nc + i] > o1[(size_t)c * nc + am]) am = i;
                    CHECK(ti[c * kk] == am, "window %d: top-1 %d != argmax %d", c, ti[c * kk], am);
                    const double want = 1.0 / (1.0 + exp(-(double)o1[(size_t)c * nc + am]));
                    CHECK(fabs((double)tc[c * kk] - want) < 1e-6, "window %d: confidence %g vs %g", c, tc[c * kk], want);
                    for (int j = 1; j < kk; j++) CHECK(tc[c * kk + j] <= tc[c * kk + j - 1], "confidences not descending");
                }
                free(tc); free(ti);
            }
        }
        int n = -1; const void* batch = NULL;
        CHECK(bnbind_win_collect(w, n_clips, srcs, &n, &batch) == 0 && n == 0, "nothing left: try again later");
        /* the whole tick in one call (bnhip_windows_predict_topk): same streams again, top-k == collect + predict_pcm_topk */
        {
            const int kk = nc < 10 ? nc : 10;
            float* ta = malloc((size_t)256 * kk * 4); int32_t* ia = malloc((size_t)256 * kk * 4);
            float* tb = malloc((size_t)256 * kk * 4); int32_t* ib = malloc((size_t)256 * kk * 4);
            int* s2 = malloc(256 * sizeof(int));
            CHECK(bnbind_win_predict_topk(w, h, 16, 0, 1.0, 10, s2, &n, ta, ia, &batch) == 0 && n == 0, "tick with nothing ready: rc/n %d", n);
            CHECK(bnbind_win_predict_topk(w, h, 24, 0, 1.0, 10, s2, &n, ta, ia, &batch) == -1, "a bit depth that does not match the window size must be invalid");
            for (int c = 0; c < n_clips; c++) {
                CHECK(bnbind_win_reset(w, c) == 0, "reset");   /* back to the first-window state (zero prefix) */
                CHECK(bnbind_win_write(w, c, pcm + (size_t)c * ns, clip_b) == 0, "write");
            }
            for (int tick = 0; tick < 2; tick++) {
                CHECK(bnbind_win_predict_topk(w, h, 16, 0, 1.0, 10, s2, &n, ta, ia, &batch) == 0 && n == n_clips, "tick: %s (n %d)", bnbind_last_error(), n);
                for (int c = 0; c < n_clips; c++) {
                    CHECK(s2[c] == c, "source order");
                    unsigned char* hw = (unsigned char*)(hand + (size_t)c * ns);
                    const unsigned char* p = (const unsigned char*)(pcm + (size_t)c * ns);
                    if (tick == 0) { memset(hw, 0, ovb); memcpy(hw + ovb, p, rdb); }
                    else { memcpy(hw, p + rdb - ovb, ovb); memcpy(hw + ovb, p + rdb, rdb); }
                }
                CHECK(memcmp(batch, hand, (size_t)n_clips * clip_b) == 0, "tick %d: rows differ from the hand-cut windows", tick);
                CHECK(bnbind_predict_pcm_topk(h, hand, 16, n_clips, 0, 1.0, 10, tb, ib) == 0, "predict_pcm_topk: %s", bnbind_last_error());
                CHECK(memcmp(ta, tb, (size_t)n_clips * kk * 4) == 0 && memcmp(ia, ib, (size_t)n_clips * kk * 4) == 0, "tick %d: top-k differs", tick);
            }
            free(ta); free(ia); free(tb); free(ib); free(s2);
        }
        bnbind_win_destroy(w);
        free(pcm); free(srcs); free(hand); free(o1); free(o2);
    }
    FILE* fo = fopen(argv[5], "wb");
    CHECK(fo && fwrite(out, 4, (size_t)n_clips * nc, fo) == (size_t)n_clips * nc, "write %s", argv[5]);
    fclose(fo);
    bnbind_model_destroy(h);                       /* Close */
    bnbind_unload();
    printf("gpu sequence ok (%d clips, %d classes)\n", n_clips, nc);
    free(in); free(out); free(one); free(stage); free(cf); free(ix);
    return 0;
}
