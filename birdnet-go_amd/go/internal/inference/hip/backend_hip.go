//go:build hip

// Package hip binds libbnhip.so (the MI355X-native BirdNET engine) behind the reference's backend seam
// inference.Classifier / inference.EmbeddingExtractor (internal/inference/backend.go:8-29).
//
// Shape follows the reference's own native-accelerator precedent, the OpenVINO cgo shim
// (internal/inference/openvino/backend_openvino.go): dlopen'd library, process-global init under a
// mutex, one native handle per classifier, C-allocated input staging, sentinel "unavailable" error so
// callers fall back (internal/classifier/birdnet.go:321-335), and - because the native error text is
// thread-local - runtime.LockOSThread around every native call plus its error fetch
// (backend_openvino.go:480,581,729,805).
//
// NOTE: no Go toolchain exists in this repository's build environment.  The C preamble below is
// nevertheless compiled and executed: tests/test_cabi.py extracts it verbatim, builds it with
// `gcc -Wall -Wextra -Werror` and drives exactly the call sequence of this file through it
// (tests/native/cabi_driver.c), on the CPU for the error paths and on the GPU for the full sequence.
package hip

/*
#cgo LDFLAGS: -ldl
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct bnhip_model bnhip_model;
typedef int  (*fn_init)(int*);
typedef void (*fn_shutdown)(void);
typedef int  (*fn_model_create)(const void*, size_t, const char*, bnhip_model**);
typedef int  (*fn_model_info)(const bnhip_model*, int*, int*, int*);
typedef int  (*fn_predict)(bnhip_model*, const float*, int, float*, float*);
typedef int  (*fn_predict_topk)(bnhip_model*, const float*, int, int, double, int, float*, int32_t*);
typedef void (*fn_model_destroy)(bnhip_model*);
typedef const char* (*fn_last_error)(void);
typedef int  (*fn_predict_pcm16)(bnhip_model*, const int16_t*, int, float*, float*);
typedef int  (*fn_us_frame_cv)(int, const double*, int, int, int, int, int, int, double*, int32_t*);
typedef struct bnhip_resampler bnhip_resampler;
typedef int  (*fn_rs_create)(int, int, int, bnhip_resampler**);
typedef int  (*fn_rs_estimate)(const bnhip_resampler*, int);
typedef int  (*fn_rs_process_pcm16)(bnhip_resampler*, const int16_t*, int, int16_t*, int, int*);
typedef int  (*fn_rs_flush_pcm16)(bnhip_resampler*, int16_t*, int, int*);
typedef void (*fn_rs_destroy)(bnhip_resampler*);
typedef int  (*fn_host_alloc)(size_t, void**);
typedef int  (*fn_host_free)(void*);
typedef int  (*fn_predict_pcm_topk)(bnhip_model*, const void*, int, int, int, double, int, float*, int32_t*);
typedef struct bnhip_windows bnhip_windows;
typedef int  (*fn_win_create)(size_t, size_t, int, bnhip_windows**);
typedef int  (*fn_win_info)(const bnhip_windows*, size_t*, int*, int*, int*);
typedef int  (*fn_win_add_source)(bnhip_windows*, const char*, size_t, int*);
typedef int  (*fn_win_remove_source)(bnhip_windows*, int);
typedef int  (*fn_win_write)(bnhip_windows*, int, const void*, size_t);
typedef int  (*fn_win_collect)(bnhip_windows*, int, int*, int*, const void**);
typedef int  (*fn_win_stats)(const bnhip_windows*, int, uint64_t*, uint64_t*, size_t*);
typedef int  (*fn_win_reset)(bnhip_windows*, int);
typedef void (*fn_win_destroy)(bnhip_windows*);
typedef int  (*fn_win_predict_topk)(bnhip_windows*, bnhip_model*, int, int, double, int, int*, int*, float*, int32_t*, const void**);

typedef struct {
    void* handle;
    fn_init init; fn_shutdown shutdown; fn_model_create model_create; fn_model_info model_info;
    fn_predict predict; fn_predict_topk predict_topk; fn_model_destroy model_destroy; fn_last_error last_error;
    fn_predict_pcm16 predict_pcm16; fn_us_frame_cv us_frame_cv;
    fn_rs_create rs_create; fn_rs_estimate rs_estimate; fn_rs_process_pcm16 rs_process_pcm16; fn_rs_flush_pcm16 rs_flush_pcm16;
    fn_rs_destroy rs_destroy;
    fn_host_alloc host_alloc; fn_host_free host_free;
    fn_win_create win_create; fn_win_info win_info; fn_win_add_source win_add_source; fn_win_remove_source win_remove_source;
    fn_win_write win_write; fn_win_collect win_collect; fn_win_stats win_stats; fn_win_reset win_reset; fn_win_destroy win_destroy;
    fn_predict_pcm_topk predict_pcm_topk; fn_win_predict_topk win_predict_topk;
} bnbind_t;
static bnbind_t BN;
static char bnbind_errbuf[256];

// A failed resolve closes the library and clears the table: the next bnbind_load starts from scratch
// ("idempotent and retryable", backend_openvino.go:477-506) instead of returning success with NULL pointers.
static const char* bnbind_fail(const char* what, const char* detail) {
    snprintf(bnbind_errbuf, sizeof bnbind_errbuf, "%s%s", what, detail ? detail : "");
    if (BN.handle) dlclose(BN.handle);
    memset(&BN, 0, sizeof BN);
    return bnbind_errbuf;
}
#define BN_RESOLVE(field, sym) do { *(void**)(&BN.field) = dlsym(BN.handle, sym); \
    if (!BN.field) return bnbind_fail("missing symbol ", sym); } while (0)

static const char* bnbind_load(const char* path) {
    if (BN.handle) return NULL;
    BN.handle = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!BN.handle) return bnbind_fail("", dlerror());
    BN_RESOLVE(init, "bnhip_init"); BN_RESOLVE(shutdown, "bnhip_shutdown");
    BN_RESOLVE(model_create, "bnhip_model_create"); BN_RESOLVE(model_info, "bnhip_model_info");
    BN_RESOLVE(predict, "bnhip_predict"); BN_RESOLVE(predict_topk, "bnhip_predict_topk");
    BN_RESOLVE(model_destroy, "bnhip_model_destroy"); BN_RESOLVE(last_error, "bnhip_last_error");
    BN_RESOLVE(predict_pcm16, "bnhip_predict_pcm16"); BN_RESOLVE(us_frame_cv, "bnhip_us_frame_cv");
    BN_RESOLVE(rs_create, "bnhip_resampler_create"); BN_RESOLVE(rs_estimate, "bnhip_resampler_estimate");
    BN_RESOLVE(rs_process_pcm16, "bnhip_resampler_process_pcm16"); BN_RESOLVE(rs_flush_pcm16, "bnhip_resampler_flush_pcm16");
    BN_RESOLVE(rs_destroy, "bnhip_resampler_destroy");
    BN_RESOLVE(host_alloc, "bnhip_host_alloc"); BN_RESOLVE(host_free, "bnhip_host_free");
    BN_RESOLVE(win_create, "bnhip_windows_create"); BN_RESOLVE(win_info, "bnhip_windows_info");
    BN_RESOLVE(win_add_source, "bnhip_windows_add_source"); BN_RESOLVE(win_remove_source, "bnhip_windows_remove_source");
    BN_RESOLVE(win_write, "bnhip_windows_write"); BN_RESOLVE(win_collect, "bnhip_windows_collect");
    BN_RESOLVE(win_stats, "bnhip_windows_stats"); BN_RESOLVE(win_reset, "bnhip_windows_reset");
    BN_RESOLVE(win_destroy, "bnhip_windows_destroy");
    BN_RESOLVE(predict_pcm_topk, "bnhip_predict_pcm_topk"); BN_RESOLVE(win_predict_topk, "bnhip_windows_predict_topk");
    return NULL;
}
static void bnbind_unload(void) {
    if (BN.shutdown) BN.shutdown();
    if (BN.handle) dlclose(BN.handle);
    memset(&BN, 0, sizeof BN);
}
static int bnbind_win_predict_topk(bnhip_windows* w, bnhip_model* m, int bits, int act, double sens, int k, int* src, int* n,
                                   float* c, int32_t* i, const void** batch) {
    return BN.win_predict_topk(w, m, bits, act, sens, k, src, n, c, i, batch);
}
// fixed-arity wrappers (cgo cannot call function pointers directly)
static int bnbind_init(int* n) { return BN.init(n); }
static int bnbind_model_create(const void* b, size_t n, const char* o, bnhip_model** m) { return BN.model_create(b, n, o, m); }
static int bnbind_model_info(const bnhip_model* m, int* a, int* b, int* c) { return BN.model_info(m, a, b, c); }
static int bnbind_predict(bnhip_model* m, const float* s, int n, float* l, float* e) { return BN.predict(m, s, n, l, e); }
static int bnbind_predict_topk(bnhip_model* m, const float* s, int n, int act, double sens, int k, float* c, int32_t* i) {
    return BN.predict_topk(m, s, n, act, sens, k, c, i);
}
static void bnbind_model_destroy(bnhip_model* m) { BN.model_destroy(m); }
static const char* bnbind_last_error(void) { return BN.last_error ? BN.last_error() : ""; }
static int bnbind_predict_pcm16(bnhip_model* m, const int16_t* s, int n, float* l, float* e) { return BN.predict_pcm16(m, s, n, l, e); }
static int bnbind_us_frame_cv(int dev, const double* s, int n_clips, int n, int rate, int fft, int hop, int split, double* cv, int32_t* ok) {
    return BN.us_frame_cv(dev, s, n_clips, n, rate, fft, hop, split, cv, ok);
}
static int bnbind_rs_create(int dev, int from, int to, bnhip_resampler** r) { return BN.rs_create(dev, from, to, r); }
static int bnbind_rs_estimate(const bnhip_resampler* r, int n) { return BN.rs_estimate(r, n); }
static int bnbind_rs_process_pcm16(bnhip_resampler* r, const int16_t* in, int n, int16_t* out, int cap, int* n_out) {
    return BN.rs_process_pcm16(r, in, n, out, cap, n_out);
}
static int bnbind_rs_flush_pcm16(bnhip_resampler* r, int16_t* out, int cap, int* n_out) { return BN.rs_flush_pcm16(r, out, cap, n_out); }
static void bnbind_rs_destroy(bnhip_resampler* r) { BN.rs_destroy(r); }
static int bnbind_host_alloc(size_t n, void** p) { return BN.host_alloc(n, p); }
static int bnbind_host_free(void* p) { return BN.host_free ? BN.host_free(p) : 0; }
static int bnbind_win_create(size_t ov, size_t rd, int mb, bnhip_windows** w) { return BN.win_create(ov, rd, mb, w); }
static int bnbind_win_info(const bnhip_windows* w, size_t* wb, int* mb, int* pinned, int* ns) { return BN.win_info(w, wb, mb, pinned, ns); }
static int bnbind_win_add_source(bnhip_windows* w, const char* id, size_t cap, int* out) { return BN.win_add_source(w, id, cap, out); }
static int bnbind_win_remove_source(bnhip_windows* w, int s) { return BN.win_remove_source(w, s); }
static int bnbind_win_write(bnhip_windows* w, int s, const void* d, size_t n) { return BN.win_write(w, s, d, n); }
static int bnbind_win_collect(bnhip_windows* w, int cap, int* src, int* n, const void** batch) { return BN.win_collect(w, cap, src, n, batch); }
static int bnbind_win_stats(const bnhip_windows* w, int s, uint64_t* wr, uint64_t* ov, size_t* buffered) { return BN.win_stats(w, s, wr, ov, buffered); }
static int bnbind_win_reset(bnhip_windows* w, int s) { return BN.win_reset(w, s); }
static void bnbind_win_destroy(bnhip_windows* w) { if (BN.win_destroy) BN.win_destroy(w); }
static int bnbind_predict_pcm_topk(bnhip_model* m, const void* pcm, int bits, int n, int act, double sens, int k, float* c, int32_t* i) {
    return BN.predict_pcm_topk(m, pcm, bits, n, act, sens, k, c, i);
}
*/
import "C"

import (
	"errors"
	"fmt"
	"math"
	"runtime"
	"strings"
	"sync"
	"unsafe"
)

// Supported reports whether the HIP backend is compiled in (mirrors openvino.Supported).
const Supported = true

// ErrHIPUnavailable: library missing or no gfx950 device. Callers treat it as "fall back".
var ErrHIPUnavailable = errors.New("hip: backend unavailable")

var (
	initMu   sync.Mutex
	initDone bool
)

// lastError must run on the OS thread that made the failing call (the text is thread-local in the
// library); every caller below holds runtime.LockOSThread across call + fetch.
func lastError() string { return C.GoString(C.bnbind_last_error()) }

// Init loads libbnhip.so and initialises the HIP runtime. Idempotent and retryable.
func Init(libraryPath string) error {
	initMu.Lock()
	defer initMu.Unlock()
	if initDone {
		return nil
	}
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	cpath := C.CString(libraryPath)
	defer C.free(unsafe.Pointer(cpath))
	if msg := C.bnbind_load(cpath); msg != nil {
		return fmt.Errorf("%w: %s", ErrHIPUnavailable, C.GoString(msg))
	}
	var n C.int
	if rc := C.bnbind_init(&n); rc != 0 {
		err := fmt.Errorf("%w: %s", ErrHIPUnavailable, lastError())
		C.bnbind_unload() // retryable: the next Init reloads
		return err
	}
	initDone = true
	return nil
}

// Classifier implements inference.Classifier and inference.EmbeddingExtractor.
// NOT goroutine-safe (backend.go:7); BirdNET.mu serialises the whole native call.
type Classifier struct {
	h        *C.bnhip_model
	nSamples int
	nClasses int
	embDim   int
	// C-side staging, as the OpenVINO shim keeps it (backend_openvino.go:673-680): the Go slice returns to a pool right after
	// Predict (process.go:280-291) and the GC may move Go memory.  Both buffers are PAGE-LOCKED (bnhip_host_alloc), so the
	// library DMAs one clip in and the logits out without its own staging copy.
	in  *C.float // [nSamples]
	out *C.float // [nClasses + embDim]
}

// NewClassifier builds a classifier from the same in-memory model bytes the TFLite backend takes
// (tflite.NewTFLiteClassifier(modelData []byte, ...), internal/inference/tflite/classifier.go:38).
// devices: one ordinal = one GPU; several = one handle sharding every batch over them.
func NewClassifier(modelData []byte, devices ...int) (*Classifier, error) {
	return NewClassifierWithOptions(modelData, Options{Devices: devices})
}

// Options are the creation options of include/bnhip.h a host may want to set; zero values keep the library defaults.
type Options struct {
	Devices  []int // GPU ordinals (default: device 0)
	MaxBatch int   // largest PredictBatch the handle accepts (default 256)
	// Precision "bf16" rounds the MFMA operands to bf16 (fp32 accumulation): only for models that tolerate it
	// (Perch v2; never BirdNET v2.4, internal/classifier/model_openvino.go:99-103). Default "f32".
	Precision string
	// LogitsOutput / EmbeddingOutput name graph outputs explicitly (1-based here so that the zero value means "the
	// reference's per-family rule", internal/inference/onnx/detection.go:52-112).
	LogitsOutput, EmbeddingOutput int
	// StrictF32 keeps every contraction on the f32-input MFMA ("bf16x3":0) instead of letting compute-bound layers run as
	// six exact bf16 products per fp32 product (fp32-equivalent to 2^-23, include/bnhip.h): for hosts that want one kernel family.
	StrictF32 bool
	// TuneDir is a directory of recorded create-time tunings (include/bnhip.h "tune_dir": files <plan key>.tune as shipped under
	// birdnet-go_amd/tune/): an engine whose plan matches a file adopts it instead of timing its kernel candidates, which makes
	// the plan - and a clip's last bits - reproducible from process to process. Empty: the BNHIP_TUNE_DIR environment, else none.
	TuneDir string
}

// NewClassifierWithOptions is NewClassifier with explicit creation options (e.g. Perch v2 on bf16 operands).
func NewClassifierWithOptions(modelData []byte, o Options) (*Classifier, error) {
	if len(modelData) == 0 {
		return nil, errors.New("hip: empty model data")
	}
	devices := o.Devices
	if len(devices) == 0 {
		devices = []int{0}
	}
	list := ""
	for i, d := range devices {
		if i > 0 {
			list += ","
		}
		list += fmt.Sprint(d)
	}
	maxBatch := o.MaxBatch
	if maxBatch <= 0 {
		maxBatch = 256
	}
	js := fmt.Sprintf(`{"devices":[%s],"max_batch":%d`, list, maxBatch)
	if o.Precision == "bf16" || o.Precision == "f32" {
		js += fmt.Sprintf(`,"precision":"%s"`, o.Precision)
	} else if o.Precision != "" {
		return nil, fmt.Errorf("hip: unknown precision %q", o.Precision)
	}
	if o.StrictF32 {
		js += `,"bf16x3":0`
	}
	if o.LogitsOutput > 0 {
		js += fmt.Sprintf(`,"logits_output":%d`, o.LogitsOutput-1)
	}
	if o.EmbeddingOutput > 0 {
		js += fmt.Sprintf(`,"embedding_output":%d`, o.EmbeddingOutput-1)
	}
	if o.TuneDir != "" {
		if strings.ContainsAny(o.TuneDir, "\"\\") {
			return nil, fmt.Errorf("hip: tune directory %q cannot be passed (quote or backslash in the path)", o.TuneDir)
		}
		js += fmt.Sprintf(`,"tune_dir":"%s"`, o.TuneDir)
	}
	opts := C.CString(js + "}")
	defer C.free(unsafe.Pointer(opts))
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	var h *C.bnhip_model
	if rc := C.bnbind_model_create(unsafe.Pointer(&modelData[0]), C.size_t(len(modelData)), opts, &h); rc != 0 {
		msg := lastError()
		if rc == -2 {
			return nil, fmt.Errorf("%w: %s", ErrHIPUnavailable, msg)
		}
		return nil, fmt.Errorf("hip: model create failed (%d): %s", int(rc), msg)
	}
	var ns, nc, ed C.int
	C.bnbind_model_info(h, &ns, &nc, &ed)
	c := &Classifier{h: h, nSamples: int(ns), nClasses: int(nc), embDim: int(ed)}
	var pin, pout unsafe.Pointer
	if rc := C.bnbind_host_alloc(C.size_t(c.nSamples)*4, &pin); rc != 0 {
		msg := lastError()
		C.bnbind_model_destroy(h)
		return nil, fmt.Errorf("hip: pinned input allocation failed (%d): %s", int(rc), msg)
	}
	if rc := C.bnbind_host_alloc(C.size_t(c.nClasses+c.embDim)*4, &pout); rc != 0 {
		msg := lastError()
		C.bnbind_host_free(pin)
		C.bnbind_model_destroy(h)
		return nil, fmt.Errorf("hip: pinned output allocation failed (%d): %s", int(rc), msg)
	}
	c.in, c.out = (*C.float)(pin), (*C.float)(pout)
	return c, nil
}

// Predict returns raw logits, one per label, in a freshly allocated slice the caller owns.
func (c *Classifier) Predict(samples []float32) ([]float32, error) {
	logits, _, err := c.predict(samples, false)
	return logits, err
}

// PredictWithEmbeddings implements inference.EmbeddingExtractor.
func (c *Classifier) PredictWithEmbeddings(samples []float32) (logits, embeddings []float32, err error) {
	return c.predict(samples, c.embDim > 0)
}

func (c *Classifier) predict(samples []float32, wantEmb bool) ([]float32, []float32, error) {
	if c.h == nil {
		return nil, nil, errors.New("hip: classifier is closed")
	}
	if len(samples) != c.nSamples {
		return nil, nil, fmt.Errorf("input size mismatch: expected %d samples, got %d", c.nSamples, len(samples))
	}
	C.memcpy(unsafe.Pointer(c.in), unsafe.Pointer(&samples[0]), C.size_t(c.nSamples)*4)
	var ep *C.float
	if wantEmb {
		ep = (*C.float)(unsafe.Add(unsafe.Pointer(c.out), c.nClasses*4))
	}
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	if rc := C.bnbind_predict(c.h, c.in, 1, c.out, ep); rc != 0 {
		return nil, nil, fmt.Errorf("hip: predict failed (%d): %s", int(rc), lastError())
	}
	// a freshly allocated slice the caller owns (tflite/classifier.go:115-116)
	logits := make([]float32, c.nClasses)
	copy(logits, unsafe.Slice((*float32)(unsafe.Pointer(c.out)), c.nClasses))
	var emb []float32
	if wantEmb {
		emb = make([]float32, c.embDim)
		copy(emb, unsafe.Slice((*float32)(unsafe.Pointer(ep)), c.embDim))
	}
	return logits, emb, nil
}

// PredictBatch mirrors onnx.Classifier.PredictBatch (internal/inference/onnx/classifier.go:372-430):
// flat [batchSize*nSamples] in, flat [batchSize*nClasses] out.
func (c *Classifier) PredictBatch(flat []float32, batchSize int) ([]float32, error) {
	if c.h == nil {
		return nil, errors.New("hip: classifier is closed")
	}
	if batchSize <= 0 || len(flat) != batchSize*c.nSamples {
		return nil, fmt.Errorf("input size mismatch: expected %d samples, got %d", batchSize*c.nSamples, len(flat))
	}
	out := make([]float32, batchSize*c.nClasses)
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	if rc := C.bnbind_predict(c.h, (*C.float)(unsafe.Pointer(&flat[0])), C.int(batchSize),
		(*C.float)(unsafe.Pointer(&out[0])), nil); rc != 0 {
		return nil, fmt.Errorf("hip: predict failed (%d): %s", int(rc), lastError())
	}
	return out, nil
}

// PredictTopK runs predict + sigmoid(sensitivity) + top-k on the device ((*BirdNET).Predict's
// post-processing, classifier/analyze.go:113-115,197-253): confidences and label indices, descending.
func (c *Classifier) PredictTopK(flat []float32, batchSize, k int, sensitivity float64) ([]float32, []int32, error) {
	return c.predictTopK(flat, batchSize, k, 0, sensitivity)
}

// PredictTopKSoftmax is the Perch v2 form: softmax over the logits (perchSoftmax, classifier/perch_onnx.go:315-335:
// max-subtract, exp in float64, float32 running sum) + top-k on the device.
func (c *Classifier) PredictTopKSoftmax(flat []float32, batchSize, k int) ([]float32, []int32, error) {
	return c.predictTopK(flat, batchSize, k, 1, 1.0)
}

func (c *Classifier) predictTopK(flat []float32, batchSize, k, activation int, sensitivity float64) ([]float32, []int32, error) {
	if c.h == nil {
		return nil, nil, errors.New("hip: classifier is closed")
	}
	if batchSize <= 0 || k <= 0 || len(flat) != batchSize*c.nSamples {
		return nil, nil, fmt.Errorf("input size mismatch: expected %d samples, got %d", batchSize*c.nSamples, len(flat))
	}
	if k > c.nClasses {
		k = c.nClasses
	}
	conf := make([]float32, batchSize*k)
	idx := make([]int32, batchSize*k)
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	if rc := C.bnbind_predict_topk(c.h, (*C.float)(unsafe.Pointer(&flat[0])), C.int(batchSize), C.int(activation), C.double(sensitivity),
		C.int(k), (*C.float)(unsafe.Pointer(&conf[0])), (*C.int32_t)(unsafe.Pointer(&idx[0]))); rc != 0 {
		return nil, nil, fmt.Errorf("hip: predict_topk failed (%d): %s", int(rc), lastError())
	}
	return conf, idx, nil
}

// NumSpecies comes from the model output, not the label list (inference/openvino.go:72-81).
func (c *Classifier) NumSpecies() int { return c.nClasses }

// Close is idempotent and frees device memory now (BirdNET.Delete, classifier/birdnet.go:972-984).
func (c *Classifier) Close() {
	if c.h != nil {
		C.bnbind_model_destroy(c.h)
		c.h = nil
	}
	if c.in != nil {
		C.bnbind_host_free(unsafe.Pointer(c.in))
		c.in = nil
	}
	if c.out != nil {
		C.bnbind_host_free(unsafe.Pointer(c.out))
		c.out = nil
	}
}

// PinnedF32 is a float32 slice over page-locked memory (bnhip_host_alloc) for batch callers: fill Data, pass it to PredictBatch /
// PredictTopK - the library recognises pinned memory per call and lets the copy engines read it directly instead of staging it.
// The memory is C-owned: Free it, do not let the slice (or a sub-slice of it) outlive Free, and do not Free it while a Predict
// call that was handed the slice is still running - nothing on the Go side tracks either.
type PinnedF32 struct {
	Data []float32
	p    unsafe.Pointer
}

// AllocPinnedF32 returns n page-locked floats (Init must have succeeded).
func AllocPinnedF32(n int) (*PinnedF32, error) {
	if n <= 0 {
		return nil, errors.New("hip: pinned allocation of zero elements")
	}
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	var p unsafe.Pointer
	if rc := C.bnbind_host_alloc(C.size_t(n)*4, &p); rc != 0 {
		return nil, fmt.Errorf("hip: pinned allocation failed (%d): %s", int(rc), lastError())
	}
	return &PinnedF32{Data: unsafe.Slice((*float32)(p), n), p: p}, nil
}

// Free releases the buffer; idempotent.
func (b *PinnedF32) Free() {
	if b.p != nil {
		C.bnbind_host_free(b.p)
		b.p, b.Data = nil, nil
	}
}

// PredictPCM16 is Predict for a clip that is still 16-bit little-endian PCM: the conversion float32(s)/32768
// (internal/analysis/process.go:479-497, internal/audiocore/convert/pcm.go:226-237) runs on the device and half the bytes
// cross PCIe.  pcm holds batchSize clips of nSamples samples (2 bytes each); returns flat [batchSize*nClasses] logits.
func (c *Classifier) PredictPCM16(pcm []byte, batchSize int) ([]float32, error) {
	if c.h == nil {
		return nil, errors.New("hip: classifier is closed")
	}
	if batchSize <= 0 || len(pcm) != batchSize*c.nSamples*2 {
		return nil, fmt.Errorf("input size mismatch: expected %d bytes, got %d", batchSize*c.nSamples*2, len(pcm))
	}
	out := make([]float32, batchSize*c.nClasses)
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	if rc := C.bnbind_predict_pcm16(c.h, (*C.int16_t)(unsafe.Pointer(&pcm[0])), C.int(batchSize),
		(*C.float)(unsafe.Pointer(&out[0])), nil); rc != 0 {
		return nil, fmt.Errorf("hip: predict_pcm16 failed (%d): %s", int(rc), lastError())
	}
	return out, nil
}

// WindowAssembler holds the analysis buffers of every audio source of ONE model in the library (bnhip_windows): per source the
// ring in overwrite mode and the overlap tail of buffer.AnalysisBuffer (internal/audiocore/buffer/analysis.go:30-276), per tick
// the Read() of all of that model's poll loops (internal/analysis/buffer_manager.go:388-496) in one pass - every source with a
// window ready lands in a row of one page-locked batch buffer, which PredictWindows hands to the device as it is.  Where the
// reference queues one batch-1 Predict per window behind Orchestrator.inferenceMu (internal/classifier/orchestrator.go:531),
// a tick is one device call.  Write may be called from any capture goroutine; Collect / PredictWindows from one at a time.
//
// Lifetimes: every `windows` slice handed out below is a VIEW of the library's batch buffer (C memory): the next tick
// overwrites it and Close frees it.  Copy what must outlive the tick (the reference copies the PCM into its Results message
// too, process.go:364-372).  Close waits for a tick or a Write in flight (life) and ticks exclude each other (tick), so a
// Close racing a poll loop can neither free the buffer under a device call nor hand the C side a dead handle.
type WindowAssembler struct {
	h           *C.bnhip_windows
	windowBytes int
	maxBatch    int
	pinned      bool
	sources     []C.int      // scratch of Collect
	tick        sync.Mutex   // one Collect / PredictWindows* at a time
	life        sync.RWMutex // readers: every call that uses h; writer: Close
}

// NewWindowAssembler: overlapBytes + readBytes = the model's clip in bytes (ModelSpec.BufferDimensions, model.go:33-56).
func NewWindowAssembler(overlapBytes, readBytes, maxBatch int) (*WindowAssembler, error) {
	if overlapBytes < 0 || readBytes <= 0 || maxBatch <= 0 {
		return nil, fmt.Errorf("hip: invalid window geometry: overlap %d, read %d, max batch %d", overlapBytes, readBytes, maxBatch)
	}
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	var h *C.bnhip_windows
	if rc := C.bnbind_win_create(C.size_t(overlapBytes), C.size_t(readBytes), C.int(maxBatch), &h); rc != 0 {
		return nil, fmt.Errorf("hip: windows_create failed (%d): %s", int(rc), lastError())
	}
	var wb C.size_t
	var mb, pin C.int
	C.bnbind_win_info(h, &wb, &mb, &pin, nil)
	return &WindowAssembler{h: h, windowBytes: int(wb), maxBatch: int(mb), pinned: pin != 0, sources: make([]C.int, int(mb))}, nil
}

// AddSource = NewAnalysisBuffer(capacity, overlap, read, sourceID) for one more source; the index names it from then on.
func (w *WindowAssembler) AddSource(sourceID string, capacity int) (int, error) {
	w.life.RLock()
	defer w.life.RUnlock()
	if w.h == nil {
		return -1, errors.New("hip: window assembler is closed")
	}
	id := C.CString(sourceID)
	defer C.free(unsafe.Pointer(id))
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	var idx C.int
	if capacity < 0 {
		capacity = 0
	}
	if rc := C.bnbind_win_add_source(w.h, id, C.size_t(capacity), &idx); rc != 0 {
		return -1, fmt.Errorf("hip: windows_add_source failed (%d): %s", int(rc), lastError())
	}
	return int(idx), nil
}

func (w *WindowAssembler) RemoveSource(source int) error {
	w.life.RLock()
	defer w.life.RUnlock()
	if w.h == nil {
		return nil
	}
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	if rc := C.bnbind_win_remove_source(w.h, C.int(source)); rc != 0 {
		return fmt.Errorf("hip: windows_remove_source failed (%d): %s", int(rc), lastError())
	}
	return nil
}

// Write = AnalysisBuffer.Write (analysis.go:152-175): never blocks on the consumer, the oldest unread bytes go when the ring
// is full.  The bytes are copied before it returns.
func (w *WindowAssembler) Write(source int, data []byte) error {
	w.life.RLock()
	defer w.life.RUnlock()
	if w.h == nil {
		return errors.New("hip: window assembler is closed")
	}
	if len(data) == 0 {
		return nil
	}
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	if rc := C.bnbind_win_write(w.h, C.int(source), unsafe.Pointer(&data[0]), C.size_t(len(data))); rc != 0 {
		return fmt.Errorf("hip: windows_write failed (%d): %s", int(rc), lastError())
	}
	return nil
}

// Collect reads every source that has a window ready (at most maxBatch; the next call resumes behind the last source looked
// at).  windows is a view of the library's batch buffer - row k belongs to sources[k] - valid until the next Collect.
func (w *WindowAssembler) Collect() (sources []int, windows []byte, err error) {
	w.tick.Lock()
	defer w.tick.Unlock()
	w.life.RLock()
	defer w.life.RUnlock()
	return w.collectLocked()
}

// collectLocked: the caller holds tick and life.
func (w *WindowAssembler) collectLocked() (sources []int, windows []byte, err error) {
	if w.h == nil {
		return nil, nil, errors.New("hip: window assembler is closed")
	}
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	var n C.int
	var batch unsafe.Pointer
	if rc := C.bnbind_win_collect(w.h, C.int(w.maxBatch), &w.sources[0], &n, &batch); rc != 0 {
		return nil, nil, fmt.Errorf("hip: windows_collect failed (%d): %s", int(rc), lastError())
	}
	if n == 0 {
		return nil, nil, nil
	}
	sources = make([]int, int(n))
	for i := range sources {
		sources[i] = int(w.sources[i])
	}
	return sources, unsafe.Slice((*byte)(batch), int(n)*w.windowBytes), nil
}

// OverwriteStats: the OverwriteTracker's inputs for one source (buffer/overwrite.go) - writes and overwriting writes since
// creation or Reset.  The rate window and the notification policy stay with the caller.
func (w *WindowAssembler) OverwriteStats(source int) (writes, overwrites uint64, err error) {
	w.life.RLock()
	defer w.life.RUnlock()
	if w.h == nil {
		return 0, 0, errors.New("hip: window assembler is closed")
	}
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	var wr, ov C.uint64_t
	if rc := C.bnbind_win_stats(w.h, C.int(source), &wr, &ov, nil); rc != 0 {
		return 0, 0, fmt.Errorf("hip: windows_stats failed (%d): %s", int(rc), lastError())
	}
	return uint64(wr), uint64(ov), nil
}

// Reset = AnalysisBuffer.Reset (analysis.go:270-276) for one source.
func (w *WindowAssembler) Reset(source int) error {
	w.life.RLock()
	defer w.life.RUnlock()
	if w.h == nil {
		return nil
	}
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	if rc := C.bnbind_win_reset(w.h, C.int(source)); rc != 0 {
		return fmt.Errorf("hip: windows_reset failed (%d): %s", int(rc), lastError())
	}
	return nil
}

func (w *WindowAssembler) WindowBytes() int { return w.windowBytes }
func (w *WindowAssembler) Pinned() bool     { return w.pinned }

// Close frees the rings and the batch buffer; every slice a tick handed out dangles from here on.  Waits for calls in flight.
func (w *WindowAssembler) Close() {
	w.tick.Lock()
	defer w.tick.Unlock()
	w.life.Lock()
	defer w.life.Unlock()
	if w.h != nil {
		C.bnbind_win_destroy(w.h)
		w.h = nil
	}
}

// PredictWindows is one tick of the real-time path for this classifier's model: Collect, then one device call over all ready
// windows straight from the assembler's batch buffer (16-bit capture, conf.BytesPerSample; the /32768 conversion of
// process.go:479-497 runs on the device).  Returns the source of each row and flat [len(sources)*nClasses] logits; nothing
// ready = (nil, nil, nil), the reference's "try again later".  The caller builds one Results message per row, as ProcessData
// does per window (process.go:327-420); windows is the PCM it must copy into the message before the next tick (a view of C
// memory: the next tick overwrites it, Close frees it).  On a failed device call sources still lists who gave up a window.
func (c *Classifier) PredictWindows(w *WindowAssembler) (sources []int, windows []byte, logits []float32, err error) {
	if c.h == nil {
		return nil, nil, nil, errors.New("hip: classifier is closed")
	}
	if w.windowBytes != c.nSamples*2 {
		return nil, nil, nil, fmt.Errorf("window size mismatch: assembler %d bytes, model clip %d bytes", w.windowBytes, c.nSamples*2)
	}
	w.tick.Lock()
	defer w.tick.Unlock()
	w.life.RLock()
	defer w.life.RUnlock()
	sources, windows, err = w.collectLocked()
	if err != nil || len(sources) == 0 {
		return nil, nil, nil, err
	}
	logits = make([]float32, len(sources)*c.nClasses)
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	if rc := C.bnbind_predict_pcm16(c.h, (*C.int16_t)(unsafe.Pointer(&windows[0])), C.int(len(sources)),
		(*C.float)(unsafe.Pointer(&logits[0])), nil); rc != 0 {
		// the listed sources have given up their window all the same (their overlap tails advanced): the caller gets the list
		// with the error, as the reference's monitor has consumed its window before ProcessData fails (buffer_manager.go:494-499)
		return sources, nil, nil, fmt.Errorf("hip: predict_pcm16 failed (%d): %s", int(rc), lastError())
	}
	return sources, windows, logits, nil
}

// PredictWindowsTopK is PredictWindows with (*BirdNET).Predict's post-processing on the device as well: per ready window the
// k best confidences float32(1/(1+exp(-sensitivity*float64(x)))) and their label indices, descending (analyze.go:113-115,
// 197-208, 220-301) - the logits never reach the host.  conf / idx are flat [len(sources)*min(k, nClasses)].  On a failed device
// call sources still lists who gave up a window (everything else nil); windows is a view valid until the next tick or Close.
func (c *Classifier) PredictWindowsTopK(w *WindowAssembler, k int, sensitivity float64) (sources []int, windows []byte, conf []float32, idx []int32, err error) {
	if c.h == nil {
		return nil, nil, nil, nil, errors.New("hip: classifier is closed")
	}
	if k <= 0 {
		return nil, nil, nil, nil, errors.New("hip: k must be positive")
	}
	w.tick.Lock()
	defer w.tick.Unlock()
	w.life.RLock()
	defer w.life.RUnlock()
	if w.h == nil {
		return nil, nil, nil, nil, errors.New("hip: window assembler is closed")
	}
	kk := k
	if kk > c.nClasses {
		kk = c.nClasses
	}
	// one call: readiness pass, then the rows of chunk i+1 are assembled while chunk i is on the device
	cbuf := make([]float32, w.maxBatch*kk)
	ibuf := make([]int32, w.maxBatch*kk)
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	var n C.int
	var batch unsafe.Pointer
	if rc := C.bnbind_win_predict_topk(w.h, c.h, 16, 0, C.double(sensitivity), C.int(k), &w.sources[0], &n,
		(*C.float)(unsafe.Pointer(&cbuf[0])), (*C.int32_t)(unsafe.Pointer(&ibuf[0])), &batch); rc != 0 {
		// the n listed sources have given up their window all the same (buffer_manager.go:494-499; the C call reports them
		// on failure too): the caller gets the list with the error and knows which sources lost a window
		err = fmt.Errorf("hip: windows_predict_topk failed (%d): %s", int(rc), lastError())
		if n > 0 {
			sources = make([]int, int(n))
			for i := range sources {
				sources[i] = int(w.sources[i])
			}
		}
		return sources, nil, nil, nil, err
	}
	if n == 0 {
		return nil, nil, nil, nil, nil
	}
	sources = make([]int, int(n))
	for i := range sources {
		sources[i] = int(w.sources[i]) // -1: the source was reset under the tick, skip the row
	}
	return sources, unsafe.Slice((*byte)(batch), int(n)*w.windowBytes), cbuf[:int(n)*kk], ibuf[:int(n)*kk], nil
}

// CustomClassifier implements inference.CustomClassifier (internal/inference/backend.go:31-53) for a dense head file - the
// BattyBirdNET regional heads the reference loads next to the shared embeddings backbone (internal/classifier/bat_onnx.go:282).
// PredictEmbedding returns sigmoid-applied scores with the reference's arithmetic: 1/(1+float32(exp(float64(-x)))), the
// division in float32 (internal/inference/onnx/postprocess.go:8-10, custom_classifier.go:148-174).
type CustomClassifier struct {
	c      *Classifier
	labels []string
}

// NewCustomClassifier loads the head (ONNX or TFLite bytes) and checks the label count against the model's class count
// (the reference builder does the same, custom_classifier.go:74-136).
func NewCustomClassifier(modelData []byte, labels []string, devices ...int) (*CustomClassifier, error) {
	c, err := NewClassifierWithOptions(modelData, Options{Devices: devices, MaxBatch: 64})
	if err != nil {
		return nil, err
	}
	if len(labels) != c.nClasses {
		c.Close()
		return nil, fmt.Errorf("hip: label count %d does not match the model's %d classes", len(labels), c.nClasses)
	}
	return &CustomClassifier{c: c, labels: append([]string(nil), labels...)}, nil
}

// PredictEmbedding implements inference.CustomClassifier.
func (cc *CustomClassifier) PredictEmbedding(embeddings []float32) ([]float32, error) {
	if cc.c == nil {
		return nil, errors.New("hip: classifier is closed")
	}
	if len(embeddings) != cc.c.nSamples {
		return nil, fmt.Errorf("embedding size mismatch: expected %d, got %d", cc.c.nSamples, len(embeddings))
	}
	logits, err := cc.c.Predict(embeddings)
	if err != nil {
		return nil, err
	}
	for i, x := range logits {
		logits[i] = 1.0 / (1.0 + float32(math.Exp(float64(-x))))
	}
	return logits, nil
}

// NumClasses, InputDim, Labels, Close implement the rest of inference.CustomClassifier.
func (cc *CustomClassifier) NumClasses() int { return cc.c.nClasses }
func (cc *CustomClassifier) InputDim() int   { return cc.c.nSamples }
func (cc *CustomClassifier) Labels() []string { return cc.labels }
func (cc *CustomClassifier) Close() {
	if cc.c != nil {
		cc.c.Close()
		cc.c = nil
	}
}

// RangeFilter implements inference.RangeFilter and inference.BatchRangeFilter (internal/inference/backend.go:55-76) for
// the [lat, lon, week] -> per-species occurrence meta-model: the one inference the product already batches (heat-map grids,
// internal/classifier/heatmap_service.go:17-36; internal/inference/onnx/rangefilter.go:106-153).  Scores are whatever the
// graph produces (the reference's models end in an in-graph sigmoid).
type RangeFilter struct{ c *Classifier }

// NewRangeFilter loads the meta-model bytes (TFLite - fp16 constants behind DEQUANTIZE are widened at load - or ONNX).
func NewRangeFilter(modelData []byte, devices ...int) (*RangeFilter, error) {
	c, err := NewClassifierWithOptions(modelData, Options{Devices: devices, MaxBatch: 4096})
	if err != nil {
		return nil, err
	}
	if c.nSamples != 3 {
		c.Close()
		return nil, fmt.Errorf("hip: range filter expects 3 inputs (lat, lon, week), the model takes %d", c.nSamples)
	}
	return &RangeFilter{c: c}, nil
}

// Predict implements inference.RangeFilter.
func (r *RangeFilter) Predict(latitude, longitude, week float32) ([]float32, error) {
	return r.PredictBatch([]float32{latitude, longitude, week}, 1)
}

// PredictBatch implements inference.BatchRangeFilter: len(inputs) must equal batchSize * 3; row-major [batchSize*numSpecies] out.
func (r *RangeFilter) PredictBatch(inputs []float32, batchSize int) ([]float32, error) {
	if r.c == nil {
		return nil, errors.New("hip: range filter is closed")
	}
	if batchSize <= 0 || len(inputs) != batchSize*3 {
		return nil, fmt.Errorf("input size mismatch: expected %d values, got %d", batchSize*3, len(inputs))
	}
	return r.c.PredictBatch(inputs, batchSize)
}

func (r *RangeFilter) NumSpecies() int { return r.c.nClasses }
func (r *RangeFilter) Close() {
	if r.c != nil {
		r.c.Close()
		r.c = nil
	}
}

// USFilterConfig carries the three geometry fields of conf.UltrasonicFilterConfig that ComputeUSFrameCV reads
// (internal/conf/config.go:1389-1395).
type USFilterConfig struct {
	FFTSize, HopSize, FrequencySplitHz int
}

// ComputeUSFrameCV is internal/audiocore/ultrasonic/filter.go:20-66 on the GPU: frame-to-frame coefficient of variation of
// the ultrasonic band power of one chunk (samples = int16/32768 as float64, convert/pcm.go:108-113); (0, false) when the
// filter's guards reject the geometry - decided before any device work, exactly as the Go code does.  Call site:
// internal/analysis/processor/processor.go:893-935.
func ComputeUSFrameCV(samples []float64, sampleRate int, cfg USFilterConfig, device int) (float64, bool, error) {
	if len(samples) == 0 {
		return 0, false, nil
	}
	var cv C.double
	var ok C.int32_t
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	if rc := C.bnbind_us_frame_cv(C.int(device), (*C.double)(unsafe.Pointer(&samples[0])), 1, C.int(len(samples)), C.int(sampleRate),
		C.int(cfg.FFTSize), C.int(cfg.HopSize), C.int(cfg.FrequencySplitHz), &cv, &ok); rc != 0 {
		return 0, false, fmt.Errorf("hip: us_frame_cv failed (%d): %s", int(rc), lastError())
	}
	return float64(cv), ok != 0, nil
}

// Resampler mirrors internal/audiocore/resample.Resampler (resample.go:44-224) method for method: a stateful PCM16 resampler
// whose filter history lives on the device between calls, so ~100 ms frames (analysis/buffer_consumer.go:118,192) resample to
// exactly the samples one call over the whole stream would give.  Not safe for concurrent use (resample.go:43).
type Resampler struct {
	h        *C.bnhip_resampler
	fromRate int
	toRate   int
	outBuf   []byte
}

const bytesPerSample = 2

// NewResampler returns nil, nil when fromRate == toRate - no resampling is required (resample.go:57-60).
func NewResampler(fromRate, toRate int, device int) (*Resampler, error) {
	if fromRate == toRate {
		return nil, nil //nolint:nilnil // as the reference: nil means "no resampling needed"
	}
	var h *C.bnhip_resampler
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	if rc := C.bnbind_rs_create(C.int(device), C.int(fromRate), C.int(toRate), &h); rc != 0 || h == nil {
		return nil, fmt.Errorf("failed to create resampler from %d Hz to %d Hz: %s", fromRate, toRate, lastError())
	}
	return &Resampler{h: h, fromRate: fromRate, toRate: toRate}, nil
}

// EstimateOutputBytes: the maximum number of bytes ResampleTo may write for the given input length (resample.go:83-88).
func (r *Resampler) EstimateOutputBytes(inputBytes int) int {
	if inputBytes <= 0 {
		return 0
	}
	return int(C.bnbind_rs_estimate(r.h, C.int(inputBytes/bytesPerSample))) * bytesPerSample
}

// ResampleTo resamples raw 16-bit PCM into dst and returns the bytes written; a dst shorter than EstimateOutputBytes is an
// error that leaves the resampler state untouched; empty input writes nothing (resample.go:99-172).
func (r *Resampler) ResampleTo(input, dst []byte) (int, error) {
	if len(input) == 0 {
		return 0, nil
	}
	if len(input)%bytesPerSample != 0 {
		return 0, fmt.Errorf("input length %d is not a multiple of %d", len(input), bytesPerSample)
	}
	if need := r.EstimateOutputBytes(len(input)); len(dst) < need {
		return 0, fmt.Errorf("destination buffer too small: need %d bytes, have %d", need, len(dst))
	}
	var n C.int
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	if rc := C.bnbind_rs_process_pcm16(r.h, (*C.int16_t)(unsafe.Pointer(&input[0])), C.int(len(input)/bytesPerSample),
		(*C.int16_t)(unsafe.Pointer(&dst[0])), C.int(len(dst)/bytesPerSample), &n); rc != 0 {
		return 0, fmt.Errorf("hip: resample failed (%d): %s", int(rc), lastError())
	}
	return int(n) * bytesPerSample, nil
}

// ResampleInto resamples into the resampler's own output buffer; the returned slice is valid until the next call
// (resample.go:179-196).
func (r *Resampler) ResampleInto(input []byte) ([]byte, error) {
	need := r.EstimateOutputBytes(len(input))
	if cap(r.outBuf) < need {
		r.outBuf = make([]byte, need)
	}
	r.outBuf = r.outBuf[:cap(r.outBuf)]
	n, err := r.ResampleTo(input, r.outBuf)
	if err != nil {
		return nil, err
	}
	return r.outBuf[:n], nil
}

// Flush emits the tail that needs zero-padded future input and resets the state for a new stream (the go-audio-resampler
// engine's Flush, which the reference's one-shot ResampleBytes relies on, resample.go:228-262).
func (r *Resampler) Flush() ([]byte, error) {
	out := make([]byte, r.EstimateOutputBytes(2*bytesPerSample)+64*bytesPerSample)
	var n C.int
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	for {
		rc := C.bnbind_rs_flush_pcm16(r.h, (*C.int16_t)(unsafe.Pointer(&out[0])), C.int(len(out)/bytesPerSample), &n)
		if rc == 0 {
			return out[:int(n)*bytesPerSample], nil
		}
		if rc != -1 || len(out) > 1<<26 {
			return nil, fmt.Errorf("hip: resample flush failed (%d): %s", int(rc), lastError())
		}
		out = make([]byte, 2*len(out)) // destination too small: nothing was consumed, retry larger
	}
}

func (r *Resampler) FromRate() int { return r.fromRate }
func (r *Resampler) ToRate() int   { return r.toRate }

// Close releases the device state; idempotent (resample.go:206-217).
func (r *Resampler) Close() error {
	if r != nil && r.h != nil {
		C.bnbind_rs_destroy(r.h)
		r.h = nil
	}
	return nil
}

func (r *Resampler) String() string { return fmt.Sprintf("Resampler(%d Hz -> %d Hz, hip)", r.fromRate, r.toRate) }

// ResampleBytes is the one-shot helper (resample.go:228-262): equal rates return the input unchanged; otherwise a fresh
// resampler processes the input once and an independent copy of what it emitted is returned (like the reference, without a
// flush: the last ~10 output samples, which need future input, are not produced).
func ResampleBytes(pcm []byte, fromRate, toRate int, device int) ([]byte, error) {
	if fromRate == toRate {
		return pcm, nil
	}
	r, err := NewResampler(fromRate, toRate, device)
	if err != nil {
		return nil, err
	}
	defer func() { _ = r.Close() }()
	out, err := r.ResampleInto(pcm)
	if err != nil {
		return nil, err
	}
	result := make([]byte, len(out))
	copy(result, out)
	return result, nil
}
