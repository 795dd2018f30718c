// C ABI of libbnhip.so (see include/bnhip.h for the reference interfaces each entry point replaces).
#include "../../include/bnhip.h"

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>
#include <utility>

#include "engine.h"
#include "tflite_model.h"

using namespace bnhip;

struct bnhip_model {
    Engine eng;
};

namespace {

thread_local std::string g_err;
std::mutex g_init_mu;
int g_devices = -1;     // -1 = not initialised

int set_err(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

// tiny extractor for {"key": <int>} options; absent -> def
long json_int(const char* js, const char* key, long def) {
    if (!js) return def;
    std::string pat = std::string("\"") + key + "\"";
    const char* p = strstr(js, pat.c_str());
    if (!p) return def;
    p += pat.size();
    while (*p == ' ' || *p == ':' || *p == '\t') p++;
    char* end = nullptr;
    long v = strtol(p, &end, 10);
    return end == p ? def : v;
}

bool is_gfx950(int dev) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
    return strncmp(prop.gcnArchName, "gfx950", 6) == 0;
}

}  // namespace

namespace bnhip {
void resample_design(int L, int M, double beta, int half_factor, std::vector<float>* table, int* T_out, int* half_out);
int launch_resample(const void* d_in, void* d_out, const float* d_table, bool pcm16, int n_clips, int n_in, int n_out, int L,
                    int M, int T, int half, hipStream_t s);
}

extern "C" {

const char* bnhip_version(void) { return "bnhip 0.1 (gfx950)"; }
const char* bnhip_last_error(void) { return g_err.c_str(); }

int bnhip_init(int* n_devices) {
    std::lock_guard<std::mutex> lk(g_init_mu);
    if (g_devices < 0) {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || n <= 0) {
            (void)hipGetLastError();
            if (n_devices) *n_devices = 0;
            return set_err(BNHIP_E_NO_DEVICE, std::string("no HIP device available: ") + hipGetErrorString(e));
        }
        int usable = 0;
        for (int d = 0; d < n; d++) if (is_gfx950(d)) usable++;
        if (!usable) {
            if (n_devices) *n_devices = 0;
            return set_err(BNHIP_E_NO_DEVICE, "no gfx950 (MI355X) device found; this library ships gfx950 code objects only");
        }
        g_devices = n;
    }
    if (n_devices) *n_devices = g_devices;
    return BNHIP_OK;
}

void bnhip_shutdown(void) {
    std::lock_guard<std::mutex> lk(g_init_mu);
    g_devices = -1;
}

int bnhip_model_create(const void* blob, size_t n_bytes, const char* opts_json, bnhip_model** out) {
    if (!out) return set_err(BNHIP_E_INVALID, "out is NULL");
    *out = nullptr;
    if (!blob || n_bytes == 0) return set_err(BNHIP_E_INVALID, "empty model blob");
    // "plan_only": parse + plan on the CPU, no device touched (diagnostics / CPU-side tests); such a
    // model answers info/describe and rejects predict calls.
    const bool plan_only = json_int(opts_json, "plan_only", 0) != 0;
    int device = (int)json_int(opts_json, "device", 0);
    int max_batch = (int)json_int(opts_json, "max_batch", 256);
    if (max_batch < 1 || max_batch > 4096) return set_err(BNHIP_E_INVALID, "max_batch must be in [1, 4096]");
    if (!plan_only) {
        int rc = bnhip_init(nullptr);
        if (rc != BNHIP_OK) return rc;
        if (device < 0 || device >= g_devices) return set_err(BNHIP_E_INVALID, "device ordinal out of range");
        if (!is_gfx950(device)) return set_err(BNHIP_E_NO_DEVICE, "selected device is not gfx950");
    }

    TflModel tm;
    std::string err;
    if (!parse_tflite(blob, n_bytes, &tm, &err)) return set_err(BNHIP_E_MODEL, err);
    bnhip_model* m = new (std::nothrow) bnhip_model();
    if (!m) return set_err(BNHIP_E_NOMEM, "out of host memory");
    int code = BNHIP_E_UNSUPPORTED;
    m->eng.no_reuse = json_int(opts_json, "debug_no_reuse", 0) != 0;
    m->eng.autotune = json_int(opts_json, "autotune", 1) != 0;
    const char* lenv = getenv("BNHIP_LANES");            // experiment switch; the option wins when given
    m->eng.n_lanes = json_int(opts_json, "lanes", lenv ? atoi(lenv) : 2);
    const char* denv = getenv("BNHIP_DEPTH");            // experiment switch; the option wins when given
    m->eng.depth = json_int(opts_json, "depth", denv ? atoi(denv) : 1);
    const char* fenv = getenv("BNHIP_FE_FFT");           // experiment switch; the option wins when given
    m->eng.frontend_fft = json_int(opts_json, "frontend_fft", fenv ? atoi(fenv) : -1);
    const char* genv = getenv("BNHIP_GRAPHS");           // experiment switch; the option wins when given
    m->eng.use_graphs = json_int(opts_json, "graphs", genv ? atoi(genv) : 0) != 0;
    if (!m->eng.build(std::move(tm), device, max_batch, plan_only, &err, &code)) {
        delete m;
        return set_err(code == BNHIP_OK ? BNHIP_E_UNSUPPORTED : code, err);
    }
    *out = m;
    return BNHIP_OK;
}

int bnhip_model_info(const bnhip_model* m, int* n_samples, int* n_classes, int* emb_dim) {
    if (!m) return set_err(BNHIP_E_INVALID, "model is NULL");
    if (n_samples) *n_samples = m->eng.n_samples;
    if (n_classes) *n_classes = m->eng.n_classes;
    if (emb_dim) *emb_dim = m->eng.emb_dim;
    return BNHIP_OK;
}

void bnhip_model_destroy(bnhip_model* m) {
    if (!m) return;
    if (m->eng.device >= 0) hipSetDevice(m->eng.device);
    delete m;
}

int bnhip_set_stream(bnhip_model* m, void* hip_stream) {
    if (!m || m->eng.device < 0) return set_err(BNHIP_E_INVALID, "model is NULL or plan-only");
    Engine& e = m->eng;
    hipSetDevice(e.device);
    e.drop_graphs();                                    // captured on the old stream
    e.sync_contexts();
    if (e.own_stream && e.stream) { hipStreamSynchronize(e.stream); hipStreamDestroy(e.stream); }
    e.stream = reinterpret_cast<hipStream_t>(hip_stream);
    e.own_stream = false;
    return BNHIP_OK;
}

int bnhip_synchronize(bnhip_model* m) {
    if (!m || m->eng.device < 0) return set_err(BNHIP_E_INVALID, "model is NULL or plan-only");
    hipSetDevice(m->eng.device);
    m->eng.sync_contexts();
    hipError_t e = hipStreamSynchronize(m->eng.stream);
    if (e != hipSuccess) return set_err(BNHIP_E_RUNTIME, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
    return BNHIP_OK;
}

int bnhip_predict_device(bnhip_model* m, const float* d_samples, int n_clips, float* d_logits, float* d_emb) {
    if (!m || !d_samples || !d_logits) return set_err(BNHIP_E_INVALID, "NULL argument");
    if (n_clips <= 0) return set_err(BNHIP_E_INVALID, "n_clips must be positive");
    Engine& e = m->eng;
    if (e.device < 0) return set_err(BNHIP_E_INVALID, "plan-only model cannot run");
    if (hipSetDevice(e.device) != hipSuccess) return set_err(BNHIP_E_RUNTIME, "hipSetDevice failed");
    std::string err;
    for (int off = 0; off < n_clips; off += e.max_batch) {
        int n = std::min(e.max_batch, n_clips - off);
        if (!e.run_pipelined(d_samples + (size_t)off * e.n_samples, n, d_logits + (size_t)off * e.n_classes,
                             d_emb ? d_emb + (size_t)off * e.emb_dim : nullptr, &err))
            return set_err(BNHIP_E_RUNTIME, err);
    }
    return BNHIP_OK;
}

// pcm_bits: 0 = float32 samples, 16 / 24 / 32 = little-endian PCM converted on the device
static int predict_host(bnhip_model* m, const void* src, int pcm_bits, int n_clips, float* logits, float* emb) {
    if (!m || !src || !logits) return set_err(BNHIP_E_INVALID, "NULL argument");
    if (n_clips <= 0) return set_err(BNHIP_E_INVALID, "n_clips must be positive");
    Engine& e = m->eng;
    if (e.device < 0) return set_err(BNHIP_E_INVALID, "plan-only model cannot run");
    if (emb && !e.emb_dim) return set_err(BNHIP_E_INVALID, "model has no embedding output");
    if (hipSetDevice(e.device) != hipSuccess) return set_err(BNHIP_E_RUNTIME, "hipSetDevice failed");
    std::string err;
    const bool pcm = pcm_bits != 0;
    const size_t bps = (size_t)pcm_bits / 8;
    // chunk = max_batch for calls larger than it; a single large batch (>= 128 clips) is split too, so that the pageable
    // H2D copy of its second part overlaps the compute of the first (PCIe-inclusive rate of a 256-clip call: +25 %)
    static const int split_env = getenv("BNHIP_HOST_SPLIT") ? atoi(getenv("BNHIP_HOST_SPLIT")) : 2;
    int ck = e.max_batch;
    if (n_clips <= e.max_batch && n_clips >= 128 && split_env > 1) ck = (n_clips + split_env - 1) / split_env;
    const int nchunks = (n_clips + ck - 1) / ck;
    const bool pipelined = nchunks > 1;
    // PCM staging: one buffer, or two (one per in-flight chunk) when the call is pipelined
    const size_t pcm_half = (size_t)e.max_batch * e.n_samples * bps, pcm_need = pcm_half * (pipelined ? 2 : 1);
    if (pcm && e.stage_pcm_bytes < pcm_need) {
        if (e.d_stage_pcm) { hipStreamSynchronize(e.stream); hipFree(e.d_stage_pcm); e.d_stage_pcm = nullptr; e.stage_pcm_bytes = 0; }
        if (hipMalloc((void**)&e.d_stage_pcm, pcm_need) != hipSuccess)
            return set_err(BNHIP_E_NOMEM, "device allocation failed (pcm staging)");
        e.stage_pcm_bytes = pcm_need;
    }
    if (pipelined && !e.d_stage_in2) {       // second staging set, created on first use
        hipError_t he = hipMalloc((void**)&e.d_stage_in2, (size_t)e.max_batch * e.n_samples * 4);
        if (he == hipSuccess) he = hipMalloc((void**)&e.d_stage_logits2, (size_t)e.max_batch * e.n_classes * 4);
        if (he == hipSuccess && e.emb_dim) he = hipMalloc((void**)&e.d_stage_emb2, (size_t)e.max_batch * e.emb_dim * 4);
        if (he == hipSuccess) he = hipStreamCreateWithFlags(&e.copy_stream, hipStreamNonBlocking);
        for (int i = 0; i < 2 && he == hipSuccess; i++) {
            he = hipEventCreateWithFlags(&e.ev_copied[i], hipEventDisableTiming);
            if (he == hipSuccess) he = hipEventCreateWithFlags(&e.ev_done[i], hipEventDisableTiming);
        }
        if (he != hipSuccess) return set_err(BNHIP_E_NOMEM, std::string("staging allocation failed: ") + hipGetErrorString(he));
    }
    if (!pipelined) {
        for (int off = 0; off < n_clips; off += e.max_batch) {
            int n = std::min(e.max_batch, n_clips - off);
            size_t cnt = (size_t)n * e.n_samples;
            hipError_t he;
            if (pcm) {
                he = hipMemcpyAsync(e.d_stage_pcm, (const char*)src + (size_t)off * e.n_samples * bps, cnt * bps,
                                    hipMemcpyHostToDevice, e.stream);
                if (he == hipSuccess) launch_pcm_to_f32(e.d_stage_pcm, pcm_bits, e.d_stage_in, cnt, e.stream);
            } else {
                he = hipMemcpyAsync(e.d_stage_in, (const float*)src + (size_t)off * e.n_samples, cnt * 4,
                                    hipMemcpyHostToDevice, e.stream);
            }
            if (he != hipSuccess) return set_err(BNHIP_E_RUNTIME, std::string("H2D copy: ") + hipGetErrorString(he));
            if (!e.run(e.d_stage_in, n, e.d_stage_logits, emb ? e.d_stage_emb : nullptr, &err))
                return set_err(BNHIP_E_RUNTIME, err);
            he = hipMemcpyAsync(logits + (size_t)off * e.n_classes, e.d_stage_logits, (size_t)n * e.n_classes * 4,
                                hipMemcpyDeviceToHost, e.stream);
            if (he == hipSuccess && emb)
                he = hipMemcpyAsync(emb + (size_t)off * e.emb_dim, e.d_stage_emb, (size_t)n * e.emb_dim * 4,
                                    hipMemcpyDeviceToHost, e.stream);
            if (he == hipSuccess) he = hipStreamSynchronize(e.stream);
            if (he != hipSuccess) return set_err(BNHIP_E_RUNTIME, std::string("D2H copy/sync: ") + hipGetErrorString(he));
        }
        return BNHIP_OK;
    }
    // ---- pipelined: the (host-blocking) pageable H2D of chunk i+1 runs on the copy stream while the compute stream is
    // busy with chunk i; the D2H of chunk i-1 is issued after chunk i's kernels are queued.
    float* din[2] = {e.d_stage_in, e.d_stage_in2};
    float* dlog[2] = {e.d_stage_logits, e.d_stage_logits2};
    float* demb[2] = {e.d_stage_emb, e.d_stage_emb2};
    auto fail = [&](const std::string& what, hipError_t he) {
        hipStreamSynchronize(e.stream); hipStreamSynchronize(e.copy_stream);
        return set_err(BNHIP_E_RUNTIME, what + ": " + hipGetErrorString(he));
    };
    auto drain = [&](int c) -> hipError_t {      // copy chunk c's results to the caller (waits for its compute)
        int off = c * ck, n = std::min(ck, n_clips - off), b = c & 1;
        hipError_t he = hipStreamWaitEvent(e.copy_stream, e.ev_done[b], 0);
        if (he == hipSuccess) he = hipMemcpyAsync(logits + (size_t)off * e.n_classes, dlog[b], (size_t)n * e.n_classes * 4,
                                                  hipMemcpyDeviceToHost, e.copy_stream);
        if (he == hipSuccess && emb)
            he = hipMemcpyAsync(emb + (size_t)off * e.emb_dim, demb[b], (size_t)n * e.emb_dim * 4, hipMemcpyDeviceToHost,
                                e.copy_stream);
        if (he == hipSuccess) he = hipStreamSynchronize(e.copy_stream);     // buffer b is free again afterwards
        return he;
    };
    for (int c = 0; c < nchunks; c++) {
        int off = c * ck, n = std::min(ck, n_clips - off), b = c & 1;
        hipError_t he;
        if (pcm) {      // raw PCM over PCIe (a half or a quarter of the float bytes), converted on the copy stream
            char* dp = reinterpret_cast<char*>(e.d_stage_pcm) + (size_t)b * pcm_half;
            he = hipMemcpyAsync(dp, (const char*)src + (size_t)off * e.n_samples * bps, (size_t)n * e.n_samples * bps,
                                hipMemcpyHostToDevice, e.copy_stream);
            if (he == hipSuccess) launch_pcm_to_f32(dp, pcm_bits, din[b], (size_t)n * e.n_samples, e.copy_stream);
        } else {
            he = hipMemcpyAsync(din[b], (const float*)src + (size_t)off * e.n_samples, (size_t)n * e.n_samples * 4,
                                hipMemcpyHostToDevice, e.copy_stream);
        }
        if (he == hipSuccess) he = hipEventRecord(e.ev_copied[b], e.copy_stream);
        if (he == hipSuccess) he = hipStreamWaitEvent(e.stream, e.ev_copied[b], 0);
        if (he != hipSuccess) return fail("H2D copy", he);
        if (!e.run(din[b], n, dlog[b], emb ? demb[b] : nullptr, &err)) { hipStreamSynchronize(e.stream); return set_err(BNHIP_E_RUNTIME, err); }
        he = hipEventRecord(e.ev_done[b], e.stream);
        if (he != hipSuccess) return fail("event record", he);
        if (c >= 1) { he = drain(c - 1); if (he != hipSuccess) return fail("D2H copy", he); }
    }
    hipError_t he = drain(nchunks - 1);
    if (he == hipSuccess) he = hipStreamSynchronize(e.stream);
    if (he != hipSuccess) return fail("D2H copy/sync", he);
    return BNHIP_OK;
}

int bnhip_predict(bnhip_model* m, const float* samples, int n_clips, float* logits, float* emb) {
    return predict_host(m, samples, 0, n_clips, logits, emb);
}

int bnhip_predict_pcm16(bnhip_model* m, const int16_t* pcm, int n_clips, float* logits, float* emb) {
    return predict_host(m, pcm, 16, n_clips, logits, emb);
}

int bnhip_predict_pcm(bnhip_model* m, const void* pcm, int bits_per_sample, int n_clips, float* logits, float* emb) {
    if (bits_per_sample != 16 && bits_per_sample != 24 && bits_per_sample != 32)
        return set_err(BNHIP_E_INVALID, "unsupported bit depth (supported: 16, 24, 32)");
    return predict_host(m, pcm, bits_per_sample, n_clips, logits, emb);
}

static int ensure_topk(Engine& e, int k) {
    if (k <= e.topk_cap) return BNHIP_OK;
    if (e.d_topk_conf) hipFree(e.d_topk_conf);
    if (e.d_topk_idx) hipFree(e.d_topk_idx);
    e.d_topk_conf = nullptr; e.d_topk_idx = nullptr; e.topk_cap = 0;
    if (hipMalloc((void**)&e.d_topk_conf, (size_t)e.max_batch * k * 4) != hipSuccess ||
        hipMalloc((void**)&e.d_topk_idx, (size_t)e.max_batch * k * 4) != hipSuccess)
        return set_err(BNHIP_E_NOMEM, "device allocation failed (top-k)");
    e.topk_cap = k;
    return BNHIP_OK;
}

int bnhip_postprocess_topk(bnhip_model* m, const float* logits, int n_clips, int n_classes, int activation,
                           double sensitivity, int k, float* out_conf, int32_t* out_idx) {
    if (!m || !logits || !out_conf || !out_idx) return set_err(BNHIP_E_INVALID, "NULL argument");
    Engine& e = m->eng;
    if (n_clips <= 0 || k <= 0) return set_err(BNHIP_E_INVALID, "n_clips and k must be positive");
    if (e.device < 0) return set_err(BNHIP_E_INVALID, "plan-only model cannot run");
    if (n_classes != e.n_classes) return set_err(BNHIP_E_INVALID, "n_classes does not match the model");
    if (activation < 0 || activation > 2) return set_err(BNHIP_E_INVALID, "unknown activation");
    if (n_classes * 4 > 150 * 1024) return set_err(BNHIP_E_UNSUPPORTED, "too many classes for the LDS top-k");
    hipSetDevice(e.device);
    int kk = std::min(k, n_classes);
    int rc = ensure_topk(e, kk);
    if (rc) return rc;
    for (int off = 0; off < n_clips; off += e.max_batch) {
        int n = std::min(e.max_batch, n_clips - off);
        hipError_t he = hipMemcpyAsync(e.d_stage_logits, logits + (size_t)off * n_classes, (size_t)n * n_classes * 4,
                                       hipMemcpyHostToDevice, e.stream);
        if (he != hipSuccess) return set_err(BNHIP_E_RUNTIME, std::string("H2D copy: ") + hipGetErrorString(he));
        launch_activation(e.d_stage_logits, e.d_post_conf, n, n_classes, activation, sensitivity, e.stream);
        launch_topk(e.d_post_conf, n, n_classes, kk, e.d_topk_conf, e.d_topk_idx, e.stream);
        hipMemcpyAsync(out_conf + (size_t)off * kk, e.d_topk_conf, (size_t)n * kk * 4, hipMemcpyDeviceToHost, e.stream);
        hipMemcpyAsync(out_idx + (size_t)off * kk, e.d_topk_idx, (size_t)n * kk * 4, hipMemcpyDeviceToHost, e.stream);
        he = hipStreamSynchronize(e.stream);
        if (he != hipSuccess) return set_err(BNHIP_E_RUNTIME, std::string("postprocess: ") + hipGetErrorString(he));
    }
    return BNHIP_OK;
}

int bnhip_predict_topk(bnhip_model* m, const float* samples, int n_clips, int activation, double sensitivity, int k,
                       float* out_conf, int32_t* out_idx) {
    if (!m || !samples || !out_conf || !out_idx) return set_err(BNHIP_E_INVALID, "NULL argument");
    Engine& e = m->eng;
    if (n_clips <= 0 || k <= 0) return set_err(BNHIP_E_INVALID, "n_clips and k must be positive");
    if (activation < 0 || activation > 2) return set_err(BNHIP_E_INVALID, "unknown activation");
    if (e.n_classes * 4 > 150 * 1024) return set_err(BNHIP_E_UNSUPPORTED, "too many classes for the LDS top-k");
    if (e.device < 0) return set_err(BNHIP_E_INVALID, "plan-only model cannot run");
    hipSetDevice(e.device);
    int kk = std::min(k, e.n_classes);
    int rc = ensure_topk(e, kk);
    if (rc) return rc;
    std::string err;
    for (int off = 0; off < n_clips; off += e.max_batch) {
        int n = std::min(e.max_batch, n_clips - off);
        hipError_t he = hipMemcpyAsync(e.d_stage_in, samples + (size_t)off * e.n_samples, (size_t)n * e.n_samples * 4,
                                       hipMemcpyHostToDevice, e.stream);
        if (he != hipSuccess) return set_err(BNHIP_E_RUNTIME, std::string("H2D copy: ") + hipGetErrorString(he));
        if (!e.run(e.d_stage_in, n, e.d_stage_logits, nullptr, &err)) return set_err(BNHIP_E_RUNTIME, err);
        launch_activation(e.d_stage_logits, e.d_post_conf, n, e.n_classes, activation, sensitivity, e.stream);
        launch_topk(e.d_post_conf, n, e.n_classes, kk, e.d_topk_conf, e.d_topk_idx, e.stream);
        hipMemcpyAsync(out_conf + (size_t)off * kk, e.d_topk_conf, (size_t)n * kk * 4, hipMemcpyDeviceToHost, e.stream);
        hipMemcpyAsync(out_idx + (size_t)off * kk, e.d_topk_idx, (size_t)n * kk * 4, hipMemcpyDeviceToHost, e.stream);
        he = hipStreamSynchronize(e.stream);
        if (he != hipSuccess) return set_err(BNHIP_E_RUNTIME, std::string("predict_topk: ") + hipGetErrorString(he));
    }
    return BNHIP_OK;
}

int bnhip_us_frame_cv(int device, const double* samples, int n_clips, int n, int sample_rate, int fft_size, int hop,
                      int split_hz, double* cv, int32_t* ok) {
    if (!samples || !cv || !ok || n_clips <= 0) return set_err(BNHIP_E_INVALID, "NULL/empty argument");
    // guards: internal/audiocore/ultrasonic/filter.go:21-37
    bool valid = !(n < fft_size || sample_rate <= 0 || fft_size < 2 || hop <= 0) && (fft_size & (fft_size - 1)) == 0 &&
                 !(split_hz < 0 || split_hz >= sample_rate / 2);
    int frames = valid ? 1 + (n - fft_size) / hop : 0;
    if (!valid || frames < 2) {
        for (int i = 0; i < n_clips; i++) { cv[i] = 0.0; ok[i] = 0; }
        return BNHIP_OK;
    }
    if ((size_t)fft_size * 16 > 160 * 1024 - 256) return set_err(BNHIP_E_UNSUPPORTED, "FFT size exceeds the LDS-resident limit (8192)");
    int rc = bnhip_init(nullptr);
    if (rc) return rc;
    if (device < 0 || device >= g_devices) return set_err(BNHIP_E_INVALID, "device ordinal out of range");
    hipSetDevice(device);
    double bin_width = (double)sample_rate / (double)fft_size;
    int split_bin = (int)((double)split_hz / bin_width);
    double *d_s = nullptr, *d_p = nullptr, *d_cv = nullptr;
    hipError_t he = hipMalloc((void**)&d_s, (size_t)n_clips * n * 8);
    if (he == hipSuccess) he = hipMalloc((void**)&d_p, (size_t)n_clips * frames * 8);
    if (he == hipSuccess) he = hipMalloc((void**)&d_cv, (size_t)n_clips * 8);
    if (he == hipSuccess) he = hipMemcpy(d_s, samples, (size_t)n_clips * n * 8, hipMemcpyHostToDevice);
    if (he == hipSuccess) {
        launch_us_frame_power(d_s, n_clips, n, fft_size, hop, frames, split_bin, d_p, nullptr);
        launch_us_cv(d_p, n_clips, frames, d_cv, nullptr);
        he = hipMemcpy(cv, d_cv, (size_t)n_clips * 8, hipMemcpyDeviceToHost);
    }
    if (d_s) hipFree(d_s);
    if (d_p) hipFree(d_p);
    if (d_cv) hipFree(d_cv);
    if (he != hipSuccess) return set_err(BNHIP_E_RUNTIME, std::string("us_frame_cv: ") + hipGetErrorString(he));
    for (int i = 0; i < n_clips; i++) ok[i] = 1;
    return BNHIP_OK;
}

int bnhip_debug_fetch(bnhip_model* m, int tensor_index, int n_clips, float* out, size_t cap_floats) {
    if (!m || !out || m->eng.device < 0) return set_err(BNHIP_E_INVALID, "NULL argument or plan-only model");
    Engine& e = m->eng;
    auto it = e.tensor_value.find(tensor_index);
    if (it == e.tensor_value.end()) return set_err(BNHIP_E_INVALID, "tensor is not materialised by the plan (fused away)");
    const Value& v = e.vals[it->second];
    if (v.external) return set_err(BNHIP_E_INVALID, "tensor is bound externally (graph input/logits)");
    size_t n = v.elems * (size_t)n_clips;
    if (n > cap_floats || n_clips > e.max_batch) return set_err(BNHIP_E_INVALID, "buffer too small");
    hipSetDevice(e.device);
    hipStreamSynchronize(e.stream);
    if (hipMemcpy(out, e.value_ptr(it->second), n * 4, hipMemcpyDeviceToHost) != hipSuccess)
        return set_err(BNHIP_E_RUNTIME, "debug fetch copy failed");
    return (int)v.elems;
}

static int igcd(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

int bnhip_resample_length(int n_in, int rate_in, int rate_out) {
    if (n_in <= 0 || rate_in <= 0 || rate_out <= 0) return 0;
    int g = igcd(rate_in, rate_out);
    long long L = rate_out / g, M = rate_in / g;
    return (int)(((long long)n_in * L + M - 1) / M);
}

static int resample_impl(int device, const void* in, bool pcm16, int n_clips, int n_in, int rate_in, int rate_out, void* out,
                         int n_out_cap, int* n_out) {
    if (!in || !out || n_clips <= 0 || n_in <= 0 || rate_in <= 0 || rate_out <= 0)
        return set_err(BNHIP_E_INVALID, "bad resample arguments");
    const int no = bnhip_resample_length(n_in, rate_in, rate_out);
    if (n_out) *n_out = no;
    if (no > n_out_cap) return set_err(BNHIP_E_INVALID, "destination buffer too small");     // resample.go:137-144
    const size_t esz = pcm16 ? 2 : 4;
    if (rate_in == rate_out) {                                                              // NewResampler returns nil: passthrough
        memcpy(out, in, (size_t)n_clips * n_in * esz);
        return BNHIP_OK;
    }
    int rc = bnhip_init(nullptr);
    if (rc) return rc;
    if (device < 0 || device >= g_devices) return set_err(BNHIP_E_INVALID, "device ordinal out of range");
    hipSetDevice(device);
    int g = igcd(rate_in, rate_out), L = rate_out / g, M = rate_in / g, T = 0, half = 0;
    std::vector<float> table;
    resample_design(L, M, 5.0, 10, &table, &T, &half);
    void *d_in = nullptr, *d_out = nullptr; float* d_tab = nullptr;
    hipError_t he = hipMalloc(&d_in, (size_t)n_clips * n_in * esz);
    if (he == hipSuccess) he = hipMalloc(&d_out, (size_t)n_clips * no * esz);
    if (he == hipSuccess) he = hipMalloc((void**)&d_tab, table.size() * 4);
    if (he == hipSuccess) he = hipMemcpy(d_in, in, (size_t)n_clips * n_in * esz, hipMemcpyHostToDevice);
    if (he == hipSuccess) he = hipMemcpy(d_tab, table.data(), table.size() * 4, hipMemcpyHostToDevice);
    int lrc = 0;
    if (he == hipSuccess) {
        lrc = launch_resample(d_in, d_out, d_tab, pcm16, n_clips, n_in, no, L, M, T, half, nullptr);
        if (lrc == 0) he = hipMemcpy(out, d_out, (size_t)n_clips * no * esz, hipMemcpyDeviceToHost);
    }
    if (d_in) hipFree(d_in);
    if (d_out) hipFree(d_out);
    if (d_tab) hipFree(d_tab);
    if (lrc) return set_err(BNHIP_E_UNSUPPORTED, "resample ratio needs a phase table larger than LDS");
    if (he != hipSuccess) return set_err(BNHIP_E_RUNTIME, std::string("resample: ") + hipGetErrorString(he));
    return BNHIP_OK;
}

int bnhip_resample_f32(int device, const float* in, int n_clips, int n_in, int rate_in, int rate_out, float* out, int n_out_cap,
                       int* n_out) {
    return resample_impl(device, in, false, n_clips, n_in, rate_in, rate_out, out, n_out_cap, n_out);
}

int bnhip_resample_pcm16(int device, const int16_t* in, int n_clips, int n_in, int rate_in, int rate_out, int16_t* out,
                         int n_out_cap, int* n_out) {
    return resample_impl(device, in, true, n_clips, n_in, rate_in, rate_out, out, n_out_cap, n_out);
}

int bnhip_profile_enable(bnhip_model* m, int on) {
    if (!m) return set_err(BNHIP_E_INVALID, "model is NULL");
    m->eng.profiling = on != 0;
    return BNHIP_OK;
}

int bnhip_profile_filter(bnhip_model* m, const char* kernel_class) {
    if (!m) return set_err(BNHIP_E_INVALID, "model is NULL");
    m->eng.profile_filter = kernel_class ? kernel_class : "";
    return BNHIP_OK;
}

static int copy_out(const std::string& s, char* buf, size_t cap) {
    if (buf && cap) {
        size_t n = std::min(cap - 1, s.size());
        memcpy(buf, s.data(), n);
        buf[n] = 0;
    }
    return (int)s.size() + 1;
}

int bnhip_profile_read(bnhip_model* m, char* buf, size_t cap) {
    if (!m || m->eng.device < 0) return set_err(BNHIP_E_INVALID, "model is NULL or plan-only");
    hipSetDevice(m->eng.device);
    return copy_out(m->eng.profile_read(), buf, cap);
}

int bnhip_model_describe(const bnhip_model* m, char* buf, size_t cap) {
    if (!m) return set_err(BNHIP_E_INVALID, "model is NULL");
    return copy_out(m->eng.describe(), buf, cap);
}

}  // extern "C"
