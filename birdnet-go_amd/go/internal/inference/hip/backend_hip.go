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

typedef struct {
    void* handle;
    fn_init init; fn_shutdown shutdown; fn_model_create model_create; fn_model_info model_info;
    fn_predict predict; fn_predict_topk predict_topk; fn_model_destroy model_destroy; fn_last_error last_error;
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
    return NULL;
}
static void bnbind_unload(void) {
    if (BN.shutdown) BN.shutdown();
    if (BN.handle) dlclose(BN.handle);
    memset(&BN, 0, sizeof BN);
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
*/
import "C"

import (
	"errors"
	"fmt"
	"runtime"
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
	in       *C.float // C-allocated staging: the Go slice returns to a pool right after Predict (process.go:280-291)
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
	if o.LogitsOutput > 0 {
		js += fmt.Sprintf(`,"logits_output":%d`, o.LogitsOutput-1)
	}
	if o.EmbeddingOutput > 0 {
		js += fmt.Sprintf(`,"embedding_output":%d`, o.EmbeddingOutput-1)
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
	c.in = (*C.float)(C.malloc(C.size_t(c.nSamples) * 4))
	if c.in == nil {
		C.bnbind_model_destroy(h)
		return nil, errors.New("hip: out of memory")
	}
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
	logits := make([]float32, c.nClasses)
	var emb []float32
	var ep *C.float
	if wantEmb {
		emb = make([]float32, c.embDim)
		ep = (*C.float)(unsafe.Pointer(&emb[0]))
	}
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	if rc := C.bnbind_predict(c.h, c.in, 1, (*C.float)(unsafe.Pointer(&logits[0])), ep); rc != 0 {
		return nil, nil, fmt.Errorf("hip: predict failed (%d): %s", int(rc), lastError())
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
		C.free(unsafe.Pointer(c.in))
		c.in = nil
	}
}
