/* Test infrastructure: plain-C host that makes exactly the call sequence of the cgo binding
 * birdnet-go_amd/go/internal/inference/hip/backend_hip.go - through that file's own C preamble, which the test extracts
 * verbatim into preamble_extracted.h (so the preamble is compiled with -Wall -Wextra -Werror and executed, although no Go
 * toolchain exists here).  SURVEY.md section 7 step 3.
 *
 *   cabi_driver <libbnhip.so> <model file> cpu               error paths + plan-only sequence (no GPU needed)
 *   cabi_driver <libbnhip.so> <model file> gpu <in.f32> <out.f32> <n_clips>
 *        Init -> NewClassifier -> Predict (clip 0) -> PredictBatch (all) -> PredictTopK -> Close, logits written to out.f32
 */
#include "preamble_extracted.h"

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
    FILE* fo = fopen(argv[5], "wb");
    CHECK(fo && fwrite(out, 4, (size_t)n_clips * nc, fo) == (size_t)n_clips * nc, "write %s", argv[5]);
    fclose(fo);
    bnbind_model_destroy(h);                       /* Close */
    bnbind_unload();
    printf("gpu sequence ok (%d clips, %d classes)\n", n_clips, nc);
    free(in); free(out); free(one); free(stage); free(cf); free(ix);
    return 0;
}
