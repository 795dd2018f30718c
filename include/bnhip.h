/*
 * bnhip.h — C ABI of libbnhip.so, the MI355X-native (gfx950) BirdNET inference engine.
 *
 * This is the drop-in boundary for birdnet-go's classifier backend seam
 *   internal/inference/backend.go:8-29   (inference.Classifier / EmbeddingExtractor)
 * i.e. exactly what a cgo backend `internal/inference/hip/backend_hip.go` (build tag `hip`) binds,
 * shaped after the reference's own native-accelerator precedent, the OpenVINO cgo shim
 *   internal/inference/openvino/backend_openvino.go:16-413 (C preamble), :443-832 (Go side).
 * Plain pointers and sizes only; no C++/torch types.  All functions return 0 on success or a
 * negative BNHIP_E_* code; bnhip_last_error() returns a thread-local message (the OpenVINO shim
 * keeps thread-local error strings too: backend_openvino.go:100,325).
 *
 * Threading contract (same as the reference's backends, backend.go:7 "NOT goroutine-safe; callers
 * must synchronize"): a bnhip_model may be used by one thread at a time.  Every entry point calls
 * hipSetDevice for the model's device first, so results never depend on the calling thread.  The error TEXT does:
 * bnhip_last_error() is thread-local, so a cgo caller must fetch it on the OS thread that made the failing call -
 * runtime.LockOSThread around call + fetch, exactly as the OpenVINO shim does (backend_openvino.go:480,581,729,805);
 * the Go binding in birdnet-go_amd/go does so.
 * No entry point lets a C++ exception escape: allocation failure is BNHIP_E_NOMEM, anything else BNHIP_E_RUNTIME
 * ("never panic; any failure => fall back", internal/classifier/model_openvino.go:227-230).
 */
#ifndef BNHIP_H
#define BNHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bnhip_model bnhip_model;

enum {
    BNHIP_OK = 0,
    BNHIP_E_INVALID = -1,      /* bad argument (NULL, size mismatch) — CategoryValidation */
    BNHIP_E_NO_DEVICE = -2,    /* no usable gfx950 device / HIP runtime — maps to ErrHIPUnavailable */
    BNHIP_E_MODEL = -3,        /* model bytes are not a TFLite flatbuffer / corrupt */
    BNHIP_E_UNSUPPORTED = -4,  /* graph uses an op/pattern the engine does not implement: caller falls back */
    BNHIP_E_RUNTIME = -5,      /* HIP runtime error during alloc/launch/copy */
    BNHIP_E_NOMEM = -6
};

/* Process-global runtime init; idempotent, retryable after failure
 * (replaces InitOV, backend_openvino.go:477-506).  Returns the number of usable devices in
 * *n_devices (nullable). */
int bnhip_init(int* n_devices);

/* Releases process-global state (replaces DestroyOV, backend_openvino.go:512-536). */
void bnhip_shutdown(void);

/* Build a classifier from in-memory model bytes - the same byte slice the reference hands to
 * NewTFLiteClassifier(modelData []byte, ...) (internal/inference/tflite/classifier.go:38), or the bytes of an ONNX file
 * (the reference's ONNX backend takes a path, internal/inference/onnx/classifier.go:268-289; dense heads such as the
 * CustomClassifier / BattyBirdNET regional heads, onnx/custom_classifier.go:148-174, classifier/bat_onnx.go:252-282).  The
 * container is sniffed: "TFL3" at byte 4 = TFLite flatbuffer, otherwise ONNX ModelProto.  The blob is consumed during the
 * call and may be freed afterwards (classifier.go:37).
 * opts_json (nullable): {"device":0,"devices":[0,1,..],"replicate":"auto","max_batch":256,"plan_only":0,"debug_no_reuse":0,
 *                        "autotune":1,"graphs":0,"lanes":2,"frontend_fft":-1,"depth":1,"host_depth":2,"bf16x3":1,
 *                        "precision":"f32","logits_output":0,"embedding_output":1,"tune_dir":"/path"}
 * "devices": one handle over several GPUs: one engine per listed device, the clips of every host-
 *          pointer call are sharded index-contiguously over them and run concurrently (one worker thread per device, own
 *          streams and pinned-order staging per device).  The frozen weights are uploaded to the first device only and
 *          replicated device-to-device: "replicate":"auto" (default) = RCCL ncclBroadcast over xGMI when librccl is loadable
 *          and the devices are distinct, hipMemcpyPeer otherwise; "rccl" / "peer" force one (the same device may be listed
 *          twice with "peer": two shards on one GPU, used by the 1-GPU test of the sharding code).  The device-pointer entry
 *          bnhip_predict_device needs a single-device handle.
 * "tune_dir": directory of recorded create-time tunings (env BNHIP_TUNE_DIR; files <plan key>.tune).  The
 *          create-time tuners choose tiles by timing, so two processes need not agree on every tile; a recorded tuning makes the
 *          plan - and with it a clip's last bits - reproducible.  Inside one process every engine of the same plan (the shards
 *          of a "devices" handle, a second handle on the same model) adopts the first one's decisions whatever this option says.
 *          A file that describes another plan is ignored; BNHIP_TUNE_RECORD=1 writes a missing one after a timed tuning.
 * "plan_only": builds the kernel plan on the CPU without touching a device (info/describe work, predict is rejected).
 * "lanes": batches of >= 32 clips are split over this many concurrent streams inside one call (default 2).
 * "depth": > 1 lets successive bnhip_predict_device calls overlap on alternating contexts (own stream and activation
 *          arena each); their outputs are complete after bnhip_synchronize, not merely in the caller's stream order.
 * "host_depth": 2 (default) runs a host-pointer call of >= 128 clips as a pipeline of chunks over two contexts fed from pinned
 *          staging (csrc/hostpipe.cpp; env BNHIP_HOST_DEPTH): still blocking, outputs complete on return, results
 *          bit-identical to "host_depth":1 (one chunk at a time, round 2's behaviour).  Costs a second activation arena.
 * "frontend_fft": 0 selects the folded-GEMM mel front-end for real-part graphs instead of the FFT path.
 * "bf16x3": pointwise / dense layers on the split-bf16 MFMA path (three exact bf16 pieces per fp32 operand, six
 *          v_mfma_f32_16x16x32_bf16 products per k, fp32 accumulation: every product is reproduced to within 2^-23, see
 *          DESIGN.md): 1 (default) = the layers whose arithmetic intensity at max_batch is >= 12 flop/B (a shape rule, so the
 *          arithmetic never depends on create-time timing; the autotuner only picks tiles), 0 = f32 MFMA only (the Go shim's
 *          Options.StrictF32), 2 = every eligible layer (K >= 16, K a multiple of 4) including the fused expand + depthwise.
 * "logits_output" / "embedding_output": indices of the graph outputs returned as logits / embedding.  Default: the reference's
 *          per-family rule (internal/inference/onnx/detection.go:24-112): output 0 (+ 1 as embedding); 160000-sample graphs
 *          with 4 outputs (Perch v2) logits 3 / embedding 0; with 2 outputs (BirdNET v3.0) the 1280-wide one is the embedding.
 *          Other outputs are not computed.  "embedding_output": -1 = none.
 * "precision": "f32" (default) keeps every product fp32 (f32 MFMA or the six-product split above).  "bf16" rounds the MFMA
 *          operands of the pointwise / dense / fused-expand layers to bf16 (one product per k, fp32 accumulation) and keeps
 *          the expanded tensors between expand, depthwise and projection as bf16 in HBM (everything else fp32 storage;
 *          depthwise, squeeze-excite, front-end and head bias arithmetic stay fp32): the reduced-precision deployment the
 *          reference runs Perch v2 in (openvino f16 drift ~0.08 accepted, openvino_parity_functional_test.go:156-158;
 *          BASELINE configs[4]).  Never a default: BirdNET v2.4 is known not to survive f16 (model_openvino.go:99-103).  */
int bnhip_model_create(const void* blob, size_t n_bytes, const char* opts_json, bnhip_model** out);

/* n_samples: exact input length per clip (tflite/classifier.go:100-104); n_classes: size of the logits
 * output read from the model, not the label list (inference/openvino.go:72-81); emb_dim: 0 when the
 * graph exposes no embedding output (EmbeddingExtractor, backend.go:21-29). */
int bnhip_model_info(const bnhip_model* m, int* n_samples, int* n_classes, int* emb_dim);

/* Classifier.Predict / PredictWithEmbeddings / onnx PredictBatch (onnx/classifier.go:372-430).
 * samples: host float32 [n_clips * n_samples], copied before return (process.go:280-291 contract).
 * logits:  host float32 [n_clips * n_classes] raw pre-activation logits in label order.
 * emb:     nullable host float32 [n_clips * emb_dim].
 * Blocking. */
int bnhip_predict(bnhip_model* m, const float* samples, int n_clips, float* logits, float* emb);

/* Same, with 16-bit little-endian PCM input converted on device: float32(s)/32768
 * (internal/analysis/process.go:479-497, audiocore/convert/pcm.go:226-237). */
int bnhip_predict_pcm16(bnhip_model* m, const int16_t* pcm, int n_clips, float* logits, float* emb);

/* Same for the three bit depths of ConvertToFloat32 (internal/audiocore/convert/pcm.go:206-268): 16-bit /32768,
 * 24-bit packed little-endian with sign extension /8388608, 32-bit /2147483648.  pcm: n_clips * n_samples samples of
 * bits_per_sample / 8 bytes each.  Any other depth is BNHIP_E_INVALID (pcm.go:215-222 "supported_bit_depths 16,24,32"). */
int bnhip_predict_pcm(bnhip_model* m, const void* pcm, int bits_per_sample, int n_clips, float* logits, float* emb);

/* Window assembler: the real-time path's analysis buffers, one per audio source, read in one pass.
 * Replaces, per source, buffer.AnalysisBuffer (internal/audiocore/buffer/analysis.go:30-276: NewAnalysisBuffer :55-145, Write
 * :152-175, Read :187-252, Reset :270-276) and, per tick, the Read() of every (source, model) poll loop
 * (internal/analysis/buffer_manager.go:388-496) - the reference then makes one batch-1 Predict per window behind
 * Orchestrator.inferenceMu (internal/classifier/orchestrator.go:531); here all windows that are ready land in consecutive rows of
 * ONE batch buffer, which is what bnhip_predict_pcm takes (bits_per_sample as captured, n_clips = *n_windows).  The buffer is
 * page-locked when a device is present (*pinned), so the copy engines read the rows in place: a window's bytes move once
 * between the capture callback and the device.
 *   geometry  one assembler per model: overlap_bytes + read_bytes = the model's clip in bytes, overlap = clip / 2
 *             (internal/classifier/model.go:33-56); read_bytes >= overlap_bytes >= 0, read_bytes > 0 (analysis.go:65-90)
 *   write     any thread, any chunk size; overwrite mode - the oldest unread bytes are dropped when the data does not fit
 *             (analysis.go:119 SetOverwrite(true)), the write is counted as an overwrite when len(data) > free bytes (:154)
 *   collect   one thread at a time (writers may run beside it): every source with >= read_bytes buffered yields
 *             `previous tail (zeros the first time) || read_bytes fresh bytes`; at most min(cap, max_batch) windows per call,
 *             the next call resumes behind the last source looked at; sources[k] = source of row k; *batch = the rows, valid
 *             until the next collect / destroy.  A model that is inactive still collects (the audio is consumed, not analysed:
 *             buffer_manager.go:478-481) and simply skips the predict.
 * Needs no device and no bnhip_model.  bnhip_windows_destroy must not run beside any other call on the same assembler (stop the
 * capture callbacks first, as the reference stops its monitors first, RemoveMonitor / RemoveAllMonitors buffer_manager.go:255-292). */
typedef struct bnhip_windows bnhip_windows;
int bnhip_windows_create(size_t overlap_bytes, size_t read_bytes, int max_batch, bnhip_windows** out);
int bnhip_windows_info(const bnhip_windows* w, size_t* window_bytes, int* max_batch, int* pinned, int* n_sources);
/* capacity_bytes >= read_bytes (analysis.go:56-64,91-100); source_id non-empty (:101-109).  Slots of removed sources are reused. */
int bnhip_windows_add_source(bnhip_windows* w, const char* source_id, size_t capacity_bytes, int* out_source);
int bnhip_windows_remove_source(bnhip_windows* w, int source);
int bnhip_windows_write(bnhip_windows* w, int source, const void* data, size_t n_bytes);
int bnhip_windows_collect(bnhip_windows* w, int cap, int* sources, int* n_windows, const void** batch);
int bnhip_windows_ready(const bnhip_windows* w, int* n_ready);
/* writes / overwrites since creation or reset (the OverwriteTracker's inputs, buffer/overwrite.go; the rate window and the
 * notification policy stay with the host), bytes currently buffered.  Any output may be NULL. */
int bnhip_windows_stats(const bnhip_windows* w, int source, uint64_t* writes, uint64_t* overwrites, size_t* buffered_bytes);
int bnhip_windows_reset(bnhip_windows* w, int source);
/* One tick of the real-time path in one call: bnhip_windows_collect + bnhip_predict_pcm_topk, with the rows of chunk c + 1
 * assembled while chunk c is on the device (the host pipeline asks for them where it would otherwise stage caller memory).
 * sources: at least max_batch ints; *n_windows rows were taken (0 = "try again later", nothing is run); out_conf / out_idx:
 * at least max_batch * min(k, n_classes).  overlap_bytes + read_bytes must equal the model's clip at bits_per_sample.  A source
 * that was reset between the readiness pass and its row has sources[r] = -1 and a row of zeros: skip it.  On a device error
 * the listed sources have still given up their window, as the reference's monitor has consumed its window by the time
 * ProcessData fails (buffer_manager.go:494-499). */
int bnhip_windows_predict_topk(bnhip_windows* w, bnhip_model* m, int bits_per_sample, int activation, double sensitivity,
                               int k, int* sources, int* n_windows, float* out_conf, int32_t* out_idx, const void** batch);
void bnhip_windows_destroy(bnhip_windows* w);

/* Page-locked host buffers for the host-pointer entries above.  The reference's accelerator shim keeps a C-allocated input
 * buffer per classifier so that the native side reads memory the Go GC cannot move (backend_openvino.go:673-680); here the same
 * buffer is page-locked as well: when `samples` / `pcm` (and `logits`, `emb`) of a bnhip_predict* call lie in memory from
 * bnhip_host_alloc - detected per call, nothing to flag - the copy engines read and write the caller's buffers directly and the
 * staging pass through the library's own pinned slots is skipped.  Results are bit-identical either way.  Needs bnhip_init. */
int bnhip_host_alloc(size_t n_bytes, void** out);
int bnhip_host_free(void* p);

/* Device-resident variant: all pointers are device memory on the model's device; work is enqueued on
 * the model's stream and NOT synchronised (call bnhip_synchronize). Used by the throughput harness so
 * timing starts with inputs already in HBM.  With "depth" > 1 successive calls run on alternating contexts and may
 * overlap; the caller must not reuse an output (or overwrite an input) of an in-flight call before bnhip_synchronize. */
int bnhip_predict_device(bnhip_model* m, const float* d_samples, int n_clips, float* d_logits, float* d_emb);

/* Post-processing on device for a batch of logits already on the host:
 * conf = float32(1/(1+exp(-sensitivity*float64(x))))  (classifier/analyze.go:113-115,197-208), then
 * top-k by confidence, descending (analyze.go:220-253).  activation: 0 = sigmoid(sensitivity),
 * 1 = softmax (perch_onnx.go:315-335), 2 = plain float32-division sigmoid (onnx/postprocess.go:8-10).
 * out_conf/out_idx: [n_clips * k]. */
int bnhip_postprocess_topk(bnhip_model* m, const float* logits, int n_clips, int n_classes, int activation,
                           double sensitivity, int k, float* out_conf, int32_t* out_idx);

/* Fused convenience: predict + activation + top-k without the logits leaving the device. */
int bnhip_predict_topk(bnhip_model* m, const float* samples, int n_clips, int activation, double sensitivity,
                       int k, float* out_conf, int32_t* out_idx);

/* The same from PCM bytes as captured (bits_per_sample 16 / 24 / 32 as in bnhip_predict_pcm): what (*BirdNET).Predict does for
 * one analysis window - convert (internal/analysis/process.go:479-497) -> classifier -> sigmoid(sensitivity) -> top-10
 * (internal/classifier/analyze.go:25-110) - for n_clips windows in one call; with the rows of bnhip_windows_collect as `pcm`
 * this is one tick of the real-time path.  Neither the float samples nor the logits exist on the host. */
int bnhip_predict_pcm_topk(bnhip_model* m, const void* pcm, int bits_per_sample, int n_clips, int activation,
                           double sensitivity, int k, float* out_conf, int32_t* out_idx);

/* Ultrasonic frame-CV filter (internal/audiocore/ultrasonic/filter.go:20-66), float64 throughout.
 * samples: host float64 [n_clips * n] (int16/32768 as float64, convert/pcm.go:108-113).
 * cv/ok: [n_clips]. device: HIP device ordinal. */
int bnhip_us_frame_cv(int device, const double* samples, int n_clips, int n, int sample_rate, int fft_size,
                      int hop, int split_hz, double* cv, int32_t* ok);

/* Device-resident form of the same filter for batched pipelines (BASELINE config 4: 256 kHz bat material): d_samples is
 * float64 [n_clips * n] or, with pcm16 != 0, raw int16 PCM converted in the kernel as int16 / 32768 in float64
 * (convert/pcm.go:108-113); d_scratch is float64 [n_clips * frames] (frames = 1 + (n - fft_size) / hop), d_cv float64
 * [n_clips].  Enqueued on hip_stream (NULL = default stream), not synchronised.  Returns the frame count (> 0) or a
 * negative error; geometries the filter's guards reject (filter.go:21-37) are BNHIP_E_INVALID here. */
int bnhip_us_frame_cv_device(int device, const void* d_samples, int pcm16, int n_clips, int n, int sample_rate, int fft_size,
                             int hop, int split_hz, double* d_scratch, double* d_cv, void* hip_stream);

/* Polyphase resampler for the step upstream of the classifier (Resampler.ResampleTo, internal/audiocore/resample/
 * resample.go:99-172).  Stateless per clip; n_out = ceil(n_in * rate_out / rate_in) (bnhip_resample_length); equal rates
 * pass through (NewResampler returns nil, :58-60); a too-small destination is an error before any work (:137-144).
 * The filter arithmetic of the reference lives in go-audio-resampler v1.7.0 (not in its tree, values unpinned), so the
 * filter is this project's own spec = scipy.signal.resample_poly's default Kaiser design; the _pcm16 entry keeps the
 * reference's edges: float32(int16)/32768 in, clamp +-1 and int16(f*32767) truncation out (:120-124,161-169). */
int bnhip_resample_length(int n_in, int rate_in, int rate_out);
int bnhip_resample_f32(int device, const float* in, int n_clips, int n_in, int rate_in, int rate_out, float* out,
                       int n_out_cap, int* n_out);
int bnhip_resample_pcm16(int device, const int16_t* in, int n_clips, int n_in, int rate_in, int rate_out, int16_t* out,
                         int n_out_cap, int* n_out);

/* Streaming form = the reference's stateful Resampler (internal/audiocore/resample/resample.go:44-172; call sites feed it
 * ~100 ms frames: analysis/buffer_consumer.go:118,192).  The filter history lives on the device between calls, so ANY
 * chunking of a stream yields, concatenated, exactly the samples one bnhip_resample_* call over the whole stream
 * produces (bit for bit; the flush emits the tail that needs zero-padded future input).
 *   create:   equal rates -> *out = NULL and BNHIP_OK (NewResampler returns nil, nil: resample.go:58-60).
 *   estimate: upper bound of samples one process call of n_in samples may emit (EstimateOutputBytes, :83-88).
 *   process:  empty input writes nothing (:100-102); a destination smaller than the estimate fails BEFORE the state
 *             advances (:137-144); *n_out = samples written.  PCM16 edges as in the one-shot entry (:120-124,161-169).
 *   flush:    end of stream: emits the remaining ceil(N*L/M) - emitted samples and resets the state for a new stream.
 *   destroy:  Close (:212-224); NULL is accepted. */
typedef struct bnhip_resampler bnhip_resampler;
int bnhip_resampler_create(int device, int rate_in, int rate_out, bnhip_resampler** out);
int bnhip_resampler_estimate(const bnhip_resampler* r, int n_in);
int bnhip_resampler_process_pcm16(bnhip_resampler* r, const int16_t* in, int n_in, int16_t* out, int out_cap, int* n_out);
int bnhip_resampler_process_f32(bnhip_resampler* r, const float* in, int n_in, float* out, int out_cap, int* n_out);
int bnhip_resampler_flush_pcm16(bnhip_resampler* r, int16_t* out, int out_cap, int* n_out);
int bnhip_resampler_flush_f32(bnhip_resampler* r, float* out, int out_cap, int* n_out);
void bnhip_resampler_destroy(bnhip_resampler* r);

/* Stream plumbing for hosts that own a HIP stream (bench harness: torch's current stream). */
int bnhip_set_stream(bnhip_model* m, void* hip_stream);
int bnhip_synchronize(bnhip_model* m);

/* Per-kernel timing (the reference has no per-operator profile wired: doc/PROFILING.md:491-540).
 * When enabled every launch is bracketed by HIP events on the model stream; bnhip_profile_read
 * synchronises and writes a JSON array [{"kernel":..,"launches":..,"ms":..,"flops":..,"bytes":..},..]
 * into buf (NUL-terminated, truncated to cap) and resets the counters. Returns bytes needed. */
int bnhip_profile_enable(bnhip_model* m, int on);
/* Restrict the event bracketing to one kernel class (e.g. "expand_dw"; NULL/"" = all): bracketing every launch costs
 * ~7 % of a step (each event is a kernel boundary), bracketing only the dominant class ~1 %. */
int bnhip_profile_filter(bnhip_model* m, const char* kernel_class);
int bnhip_profile_read(bnhip_model* m, char* buf, size_t cap);

/* Whole-call timing: when enabled, every call's plan is bracketed by ONE event pair on the stream it runs on (two kernel
 * boundaries per call, cheap enough for a timed region).  bnhip_profile_steps_read synchronises, writes up to cap calls'
 * start / end times in milliseconds relative to the first call's start (either array may be NULL), clears the record and
 * returns the number of calls recorded.  Source of the per-batch median / p95 the reference's benchmarks report
 * (cmd/perch-benchmark/main.go:31-32,354-391).  Single-device handles (engine 0 of a multi-device one). */
int bnhip_profile_steps(bnhip_model* m, int on);
int bnhip_profile_steps_read(bnhip_model* m, double* start_ms, double* end_ms, int cap);

/* Plan description (JSON) for diagnostics/DESIGN tables: one entry per launch with shapes,
 * algorithmic flops and bytes. Returns bytes needed. */
int bnhip_model_describe(const bnhip_model* m, char* buf, size_t cap);

/* Diagnostics: copy the activation produced for TFLite tensor `tensor_index` by the LAST run of
 * n_clips clips to host. Only meaningful for models created with {"debug_no_reuse":1} (otherwise the
 * arena slot may already have been recycled). Returns floats per clip, or a negative error. */
int bnhip_debug_fetch(bnhip_model* m, int tensor_index, int n_clips, float* out, size_t cap_floats);

/* Devices of a handle: returns their count and writes up to cap ordinals (1 for a plain "device" handle). */
int bnhip_model_devices(const bnhip_model* m, int* devices, int cap);

/* Idempotent; frees device memory now (BirdNET.Delete, classifier/birdnet.go:972-984). */
void bnhip_model_destroy(bnhip_model* m);

const char* bnhip_last_error(void);
/* Same text copied into the caller's buffer (NUL-terminated, truncated to cap); returns the bytes needed.  For hosts that
 * prefer an out-buffer to a thread-local pointer. */
int bnhip_last_error_copy(char* buf, size_t cap);
const char* bnhip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* BNHIP_H */
