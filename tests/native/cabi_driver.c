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
    FILE* fo = fopen(argv[5], "wb");
    CHECK(fo && fwrite(out, 4, (size_t)n_clips * nc, fo) == (size_t)n_clips * nc, "write %s", argv[5]);
    fclose(fo);
    bnbind_model_destroy(h);                       /* Close */
    bnbind_unload();
    printf("gpu sequence ok (%d clips, %d classes)\n", n_clips, nc);
    free(in); free(out); free(one); free(stage); free(cf); free(ix);
    return 0;
}
