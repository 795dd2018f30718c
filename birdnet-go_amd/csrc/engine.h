// Graph planner + executor: TFLite op graph -> fused gfx950 kernel plan.
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <string>
#include <vector>

#include "kernels.h"
#include "tflite_model.h"

namespace bnhip {

enum StepKind { S_MINMAX, S_FRONTEND, S_NORMALIZE, S_STFT, S_MELFIN, S_MELBAND, S_CONV_DIRECT, S_PW, S_DW, S_EXPAND_DW, S_MEAN_PARTIAL, S_MEAN_FINISH, S_SE, S_UNARY, S_BINARY,
                // generic tier (generic.hip): any float op the fused plan does not absorb
                S_EW_UNARY, S_EW_BINARY, S_POOL, S_COPY, S_SOFTMAX, S_REDUCE, S_CONV_GENERIC,
                S_CONV_IGEMM };   // general convolution as an implicit GEMM on the f32 MFMA (k_pw_gemm<IM>)

struct GenGeom {             // per-clip 4-D view geometry of a generic step (see kernels.h BcastParams / CopyParams)
    int d[4] = {1, 1, 1, 1};
    long sa[4] = {0, 0, 0, 0}, sb[4] = {0, 0, 0, 0}, so[4] = {0, 0, 0, 0};
    long offa = 0, offo = 0;
    bool a_const = false, b_const = false;     // operand lives in the weight arena (no clip stride)
    int mask = 0;                              // S_REDUCE: reduced dims
    float alpha = 0.f;                         // S_EW_UNARY: LEAKY_RELU slope; S_COPY with fill: pad value; S_SOFTMAX: beta
    bool fill = false;                         // S_COPY: fill the output with alpha first (PAD)
    int dh = 1, dw = 1, depthwise = 0, mult = 1;   // S_CONV_GENERIC
};

struct Step {
    StepKind kind;
    std::string name;        // graph op name (for describe/profile)
    const char* kclass;      // kernel class for profile aggregation
    // value ids (indices into Engine::vals) ; -1 = none
    int in0 = -1, in1 = -1, in2 = -1, out = -1, out2 = -1;
    // weights (device pointers into the weight arena)
    const float *w0 = nullptr, *w1 = nullptr, *w2 = nullptr, *w3 = nullptr;
    const uint16_t* wbx = nullptr;   // S_PW / S_EXPAND_DW: split-bf16 weight image (pw_bx3_image / expdw_bx_image) of a bf16x3 engine
    int dwl = 0;                     // S_DW: 1 = LDS-staged kernel (launch_dwconv_lds, tile shape in `shape`), chosen by the autotuner
    int bx = 0;                      // S_EXPAND_DW: 1 = phase 1 on the split-bf16 MFMA (autotuned per layer; bf16x3 = 2 forces it)
    // geometry per clip
    int H = 0, W = 0, C = 0, Ho = 0, Wo = 0, Co = 0, kh = 1, kw = 1, sh = 1, sw = 1, pt = 0, pl = 0;
    int act = 0, act2 = 0, op = 0, mode = 0, S = 1, Cr = 0;
    int nt = 0, wm = 0;      // pw_gemm tile shape chosen by the create-time autotuner (0 = heuristic), for a lane's batch
    int H2 = 0, W2 = 0, pt2 = 0, pl2 = 0;   // fused stem + depthwise (mode == 1 on an S_EXPAND_DW step): raw image size, stem padding
    int shape = -1;          // expand_dw tile shape (index into the kernel's table) chosen by the autotuner; -1 = cost model
    int nt_full = 0, wm_full = 0;   // same, tuned at max_batch (calls that run unsplit: pipelined contexts, profiling)
    // front-end
    int spec = -1;
    GenGeom g;
    // accounting per clip
    double flops = 0, bytes = 0, wbytes = 0;   // wbytes: weight bytes per launch
};

struct Value {               // an activation tensor (per clip geometry)
    int tfl = -1;            // tflite tensor id (or -1 for internal scratch)
    size_t elems = 0;        // floats per clip
    size_t offset = 0;       // byte offset in the activation arena (for max_batch)
    size_t offset_lane = 0;  // byte offset inside one lane's region (layout for lane_cap clips)
    int first = -1, last = -1;   // step liveness
    bool external = false;   // bound at run time (graph input / outputs)
    bool half = false;       // stored as bf16 (first half of its block): "precision":"bf16" engines, Engine::mark_bf16_storage
};

struct FrontSpec {
    int L, Kp, Lfft, hop, F, n_mels, NTP, c;
    float p1, p2, eps, norm_sub, norm_mul;
    const double* G = nullptr;   // device, fp64
    const float* window = nullptr;
    // FFT path (stft.hip)
    bool fft = false;
    int nb = 0, nbp = 0, mode = 0;   // needed bins, padded to 4; 0 = real part, 1 = magnitude
    const float* window_full = nullptr;
    const double* stft_tw = nullptr;   // twiddle image of the STFT step (stft_build_tables)
    unsigned stft_zmask = 0xffffffffu; // which outputs of the closing FFT stage the needed bins read (stft_zmask)
    const int* bins = nullptr;
    // front-end variants beyond the v2.4 MelSpec layer (Perch-style log-mel): see FrontendMatch
    bool normalize = true, log_compress = false, time_major = false;
    int pad_left = 0;
    float log_floor = 0.f, log_scale = 1.f;
};

struct ProfEntry { hipEvent_t a, b; int step; int n; };

class Engine {
  public:
    ~Engine();
    // returns false and sets err (+ code: BNHIP_E_*) on failure
    bool build(TflModel m, int device, int max_batch, bool plan_only, std::string* err, int* code);
    bool run(const float* d_in, int n, float* d_logits, float* d_emb, std::string* err);

    int device = 0, max_batch = 256;
    bool no_reuse = false;              // diagnostics: every activation keeps its own buffer
    bool autotune = true;               // time pw_gemm tile widths per layer at create time (a few ms)
    // multi-device handles: every engine plans the same weight image; only the first uploads it from the host, the others
    // allocate their weight arena and receive the bytes device-to-device (RCCL broadcast / peer copy, see api.cpp) before
    // finish_deferred() runs the create-time autotune
    bool defer_weights = false;
    void finish_deferred();
    char* weights_ptr() const { return w_arena; }
    size_t weights_bytes() const { return w_bytes; }
    int frontend_fft = -1;              // -1 / 1: FFT path where the frame length is supported (512/1024/2048), 0: folded-GEMM kernel for real-part graphs
    int logits_output = -1, embedding_output = -2;   // graph output indices (options; -1 / -2 = the reference's per-family rule, see build())
    int precision = 0;                  // 0: fp32 products everywhere (f32 MFMA or the six-product split); 1 ("precision":"bf16"): the MFMA
                                        // layers round their operands to bf16 (one product, fp32 accumulate) - Perch-style deployments
    int bf16x3 = 0;                     // split-bf16 MFMA path for pointwise / dense layers (k_pw_bx3): 0 off, 1 per layer where the
                                        // create-time autotuner measures it faster, 2 every eligible layer (parity tests)
    int pw_sw = 0;                      // PW_SW_* switch bits of the split-bf16 GEMM family: the environment read ONCE in build(), carried in every PwParams
    bool use_graphs = false;            // opt-in: replay the plan as a hipGraph once a (pointers, n) combination repeats (measured: no gain on ROCm 7.2)
    void drop_graphs();
    void autotune_expdw();
    void mark_bf16_storage();           // "precision":"bf16": which activation values are kept as bf16 in HBM
    void autotune_dw();                 // S_DW: register-tiled k_dwconv_t vs the LDS-staged form, per layer
    // Pipelining across calls ("depth" option, bnhip_predict_device only): call i runs on context i % depth (own stream,
    // own activation arena), so the tail of one batch overlaps the head of the next.  Completion is then signalled by
    // synchronize(), not by the caller's stream.
    static constexpr int kMaxDepth = 3;
    int depth = 1;
    hipStream_t ctx_stream[kMaxDepth] = {nullptr, nullptr, nullptr};
    char* ctx_arena[kMaxDepth] = {nullptr, nullptr, nullptr};
    static constexpr int kMinMaxScratch = 16 * (2 * kMinMaxParts + 2);   // floats: 16 clips x (parts + counter) of k_clip_minmax_parts
    float* mm_scratch = nullptr;        // [context][lane][kMinMaxScratch]: outside the arenas (their layouts overlap), zeroed once
    bool mm_dirty = false;              // a call failed: the arrival counters in mm_scratch may be non-zero - re-zeroed before the next call
    hipEvent_t ev_ctx_fork = nullptr, ev_ctx_done[kMaxDepth] = {nullptr, nullptr, nullptr};
    unsigned call_idx = 0;
    bool run_pipelined(const float* d_in, int n, float* d_logits, float* d_emb, std::string* err);
    void sync_contexts();
    // Host-pointer pipeline (hostpipe.cpp): the blocking bnhip_predict* entries run calls of >= 128 clips as chunks on
    // alternating contexts ("host_depth", default 2), fed from a ring of pinned staging slots, so that chunk i+1's copy
    // and front half overlap chunk i's back half exactly as successive bnhip_predict_device calls do.  The contexts are
    // created on first use (ensure_contexts) when the engine was built with depth 1.
    int host_depth = 2;
    bool ensure_contexts(int d, std::string* err, bool with_streams = true);   // host pipeline: arenas only (it runs on kstream[0..1])
    bool run_on_context(int c, hipStream_t st, const float* d_in, int n, float* d_logits, float* d_emb, std::string* err);
    // Two-phase host calls (hostpipe.cpp, calls that fit one batch): the plan is cut at `split_step`, a launch boundary that exactly
    // one activation crosses (`v_hand`).  Steps [0, split_step) run per chunk as the chunks arrive and leave that value in hand-off
    // memory at the chunk's clip offset; steps [split_step, end) run over groups of chunks, where the late layers have the rows to
    // fill the chip.  Same kernels, same per-clip arithmetic: a clip's outputs do not depend on how the call was cut.
    int split_step = -1, v_hand = -1;   // chosen at plan time (pick_split); -1: no such cut / disabled
    float* d_hand = nullptr;            // [max_batch][hand_clip_bytes()], allocated on first use
    std::vector<int> split_candidates;  // every step index with a single crossing value (diagnostics, describe)
    void pick_split();
    size_t hand_clip_bytes() const { return v_hand < 0 ? 0 : vals[v_hand].elems * (vals[v_hand].half ? 2 : 4); }
    bool ensure_hand(std::string* err);
    // steps [s0, s1) of the plan for n clips in context c's arena on stream st; `hand` = the hand-off value's rows for these clips
    bool run_part(int c, hipStream_t st, int s0, int s1, const float* d_in, int n, float* hand, float* d_logits, float* d_emb, std::string* err);
    struct HostPipe* hostpipe = nullptr;
    static constexpr int kMaxLanes = 4;
    int n_lanes = 2;                    // batches of >= dual_lane_min clips are split over this many streams (see run_eager)
    int dual_lane_min = 32;
    hipStream_t lane_stream[kMaxLanes - 1] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[kMaxLanes - 1] = {nullptr, nullptr, nullptr};
    void autotune_pw();
    void tune_or_load();                            // the three create-time tuners, or their recorded result (BNHIP_TUNE_FILE)
    bool save_tuning(const char* path) const;      // BNHIP_TUNE_FILE: the create-time tuners' decisions, one line per step
    bool load_tuning(const char* path);             // false (and nothing changed) unless the file describes exactly this plan
    std::string tuning_text() const;                // the same decisions as text (what the file holds)
    bool apply_tuning_text(const std::string& text);
    std::string tune_key() const;                   // plan hash + batch / depth / precision / switches: cache key and file name of a recorded tuning
    std::string tune_dir;                           // "tune_dir" option / BNHIP_TUNE_DIR: directory of recorded tunings (<tune_key>.tune)
    std::string tune_source;                        // where this engine's tuning came from: file:… | process-cache | dir:… | self-tuned | (empty: autotune off)
    std::map<int, int> tensor_value;    // tflite tensor index -> value id (diagnostics)
    const float* value_ptr(int v) const { return reinterpret_cast<const float*>(act_arena + vals[v].offset); }
    int n_samples = 0, n_classes = 0, emb_dim = 0, C_spec = 0;
    hipStream_t stream = nullptr;       // main stream: kstream[0] until bnhip_set_stream hands in the caller's
    bool own_stream = false;
    // Kernel streams of the engine.  HIP maps the streams of one priority onto at most four hardware queues
    // (GPU_MAX_HW_QUEUES) and two streams that share one serialise each other's kernels - so lanes and contexts, which never run at
    // the same time, draw from ONE small pool: lane i = kstream[i] (lane 0 = the main stream), context c = kstream[1 + c] (never the
    // main stream: a pipelined call forks from whatever the main stream has queued, which must not include the previous call).
    static constexpr int kMaxKStreams = 4;
    hipStream_t kstream[kMaxKStreams] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t kernel_stream(int i);   // created on first use; nullptr on failure.  Shared by every engine on the device.
    hipStream_t copy_stream();          // the device's one copy stream (host pipeline: both directions)
    hipStream_t xfer_stream = nullptr;
    bool ds_ref = false;                // this engine holds a reference on its device's stream pool
    void release_streams();
    bool profiling = false;
    std::string profile_filter;      // non-empty: only launches of this kernel class are bracketed by events

    std::vector<Step> steps;
    std::vector<Value> vals;
    std::vector<FrontSpec> specs;
    int v_input = -1, v_logits = -1, v_emb = -1, v_mm = -1;

    // staging for host-pointer API
    float* d_stage_in = nullptr;      // [max_batch, n_samples]
    float* d_stage_logits = nullptr;  // [max_batch, n_classes]
    float* d_stage_emb = nullptr;     // [max_batch, emb_dim]
    int16_t* d_stage_pcm = nullptr;      // PCM staging of the bnhip_predict_pcm* entries (16/24/32-bit: sized in bytes)
    size_t stage_pcm_bytes = 0;
    float* d_post_conf = nullptr;     // [max_batch, n_classes]
    float* d_topk_conf = nullptr;
    int32_t* d_topk_idx = nullptr;
    int topk_cap = 0;

    std::string describe() const;
    std::string profile_read();
    // Whole-call timing (bnhip_profile_steps): one event pair around the plan of every call, on the stream the call runs on -
    // two kernel boundaries per call instead of 2 x 61, so it can stay on inside a timed region.  The per-step distribution
    // the reference's benchmarks report (median / p95 over >= 30 batches, cmd/perch-benchmark/main.go:31-32,354-391) comes
    // from here.
    bool step_timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> step_ev;
    int steps_read(double* start_ms, double* end_ms, int cap);     // relative to the first call's start; clears

  private:
    struct GraphEntry { const float* in; float* logits; float* emb; int n; int seen; hipGraphExec_t exec; };
    std::vector<GraphEntry> graphs;     // tiny cache: the host path always presents the same staging pointers
    bool run_eager(const float* d_in, int n, float* d_logits, float* d_emb, std::string* err);
    char* act_arena = nullptr;
    char* cur_arena = nullptr;          // arena the launches of the current call address (act_arena or a context's)
    hipStream_t cur_stream = nullptr;   // main stream of the current call (stream or a context's)
    int cur_ctx = -1;                   // context index of the current pipelined call (diagnostics)
    int part_s0 = 0, part_s1 = -1;      // step range of the current call (run_part); -1 = to the end
    float* part_hand = nullptr;         // where v_hand lives for the current call (run_part)
    size_t act_bytes = 0;
    char* w_arena = nullptr;
    size_t w_bytes = 0;
    std::vector<ProfEntry> prof;
    std::vector<hipEvent_t> ev_pool;
    float* vptr(int v, const float* d_in, float* d_logits, float* d_emb, int lane = -1) const;
    int lane_cap = 0;                   // clips one lane can hold (ceil(max_batch / n_lanes))
    size_t lane_bytes = 0;              // size of one lane's arena region
    hipEvent_t get_event();
};

}  // namespace bnhip
