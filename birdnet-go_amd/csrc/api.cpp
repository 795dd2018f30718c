// C ABI of libbnhip.so (see include/bnhip.h for the reference interfaces each entry point replaces).
//
// Every extern "C" entry point is exception-tight: the engine is C++ (std::vector / std::map / std::string), and an
// exception unwinding through a cgo frame aborts the host process, which would break the reference's rule for native
// backends - "never panic; any failure => fall back" (internal/classifier/model_openvino.go:227-230).  BN_GUARD turns
// std::bad_alloc into BNHIP_E_NOMEM and anything else into BNHIP_E_RUNTIME.
#include "../../include/bnhip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "engine.h"
#include "hostpipe.h"
#include "numa.h"
#include "model_onnx.h"
#include "tflite_model.h"
#include "windows.h"

using namespace bnhip;

namespace {

thread_local std::string g_err;
std::mutex g_init_mu;
int g_devices = -1;     // -1 = not initialised

int set_err(int code, const std::string& msg) noexcept {
    try { g_err = msg; } catch (...) { g_err.clear(); }
    return code;
}

#define BN_GUARD_BEGIN try {
#define BN_GUARD_END(fallback_stmt)                                                                   \
    }                                                                                                 \
    catch (const std::bad_alloc&) { fallback_stmt; return set_err(BNHIP_E_NOMEM, "out of host memory"); }                 \
    catch (const std::exception& ex_) { fallback_stmt; return set_err(BNHIP_E_RUNTIME, std::string("internal error: ") + ex_.what()); } \
    catch (...) { fallback_stmt; return set_err(BNHIP_E_RUNTIME, "internal error: unknown exception"); }

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
// {"key": [i, j, ...]} -> values; absent or malformed -> empty
std::vector<int> json_int_array(const char* js, const char* key) {
    std::vector<int> v;
    if (!js) return v;
    std::string pat = std::string("\"") + key + "\"";
    const char* p = strstr(js, pat.c_str());
    if (!p) return v;
    p += pat.size();
    while (*p == ' ' || *p == ':' || *p == '\t') p++;
    if (*p != '[') return v;
    p++;
    while (*p && *p != ']') {
        char* end = nullptr;
        long x = strtol(p, &end, 10);
        if (end == p) { v.clear(); return v; }
        v.push_back((int)x);
        p = end;
        while (*p == ' ' || *p == ',' || *p == '\t') p++;
    }
    return v;
}
// {"key": "text"} -> text; absent -> def
std::string json_str(const char* js, const char* key, const char* def) {
    if (!js) return def;
    std::string pat = std::string("\"") + key + "\"";
    const char* p = strstr(js, pat.c_str());
    if (!p) return def;
    p += pat.size();
    while (*p == ' ' || *p == ':' || *p == '\t') p++;
    if (*p != '"') return def;
    const char* q = strchr(p + 1, '"');
    return q ? std::string(p + 1, q) : std::string(def);
}

bool is_gfx950(int dev) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
    return strncmp(prop.gcnArchName, "gfx950", 6) == 0;
}

// One worker thread per engine of a multi-device handle: the thread owns its device's HIP context binding
// (hipSetDevice is thread-local), runs one job at a time, never lets an exception escape.
struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int(std::string&)> job;
    bool pending = false, stop = false;
    int rc = 0;
    std::string err;

    void start(int device) {
        th = std::thread([this, device] {
            hipSetDevice(device);
            std::unique_lock<std::mutex> lk(mu);
            for (;;) {
                cv.wait(lk, [this] { return pending || stop; });
                if (stop) return;
                std::function<int(std::string&)> j = std::move(job);
                lk.unlock();
                int r; std::string e;
                try { r = j(e); }
                catch (const std::bad_alloc&) { r = BNHIP_E_NOMEM; e = "out of host memory"; }
                catch (const std::exception& ex) { r = BNHIP_E_RUNTIME; try { e = std::string("internal error: ") + ex.what(); } catch (...) {} }
                catch (...) { r = BNHIP_E_RUNTIME; }
                lk.lock();
                rc = r; err.swap(e); pending = false;
                cv.notify_all();
            }
        });
    }
    void submit(std::function<int(std::string&)> j) {
        std::lock_guard<std::mutex> lk(mu);
        job = std::move(j); pending = true; rc = 0; err.clear();
        cv.notify_all();
    }
    int wait(std::string* e) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [this] { return !pending; });
        if (rc && e && e->empty()) *e = err;
        return rc;
    }
    ~Worker() {
        if (th.joinable()) {
            { std::lock_guard<std::mutex> lk(mu); stop = true; cv.notify_all(); }
            th.join();
        }
    }
};

}  // namespace

// A handle owns one engine per device of its "devices" list (one for the plain "device" form).  Clips of a call are
// sharded index-contiguously over the engines (independent clips, no exchange step).
struct bnhip_model {
    std::vector<std::unique_ptr<Engine>> engs;
    std::vector<std::unique_ptr<Worker>> workers;      // parallel to engs when engs.size() > 1
    std::string replication = "host-upload";           // how engines 1.. got their weights: "rccl-broadcast" | "peer-copy"
    Engine& eng() { return *engs[0]; }
    const Engine& eng() const { return *engs[0]; }
};

namespace bnhip {
void resample_design(int L, int M, double beta, int half_factor, std::vector<float>* table, int* T_out, int* half_out);
int launch_resample(const void* d_in, void* d_out, const float* d_table, int in_pcm16, int out_pcm16, int n_clips, int n_in,
                    int n_out, int L, int M, int T, int half, long long i_base, long long n_base, hipStream_t s);
}

namespace {

// twiddle tables of the ultrasonic FFT, one per (device, fft size), uploaded on first use and kept for the process
std::mutex g_tw_mu;
std::vector<std::pair<std::pair<int, int>, double*>> g_tw;
const double* us_twiddles(int device, int fft_size) {
    std::lock_guard<std::mutex> lk(g_tw_mu);
    for (auto& e : g_tw) if (e.first.first == device && e.first.second == fft_size) return e.second;
    std::vector<double> t = us_twiddle_table(fft_size);
    double* d = nullptr;
    if (hipMalloc((void**)&d, t.size() * 8) != hipSuccess) return nullptr;
    if (hipMemcpy(d, t.data(), t.size() * 8, hipMemcpyHostToDevice) != hipSuccess) { hipFree(d); return nullptr; }
    g_tw.push_back({{device, fft_size}, d});
    return d;
}

// ---------------------------------------------------------------------------------------------- RCCL (optional, dlopen'd)
// Weights of a multi-device handle are uploaded to the first device only and replicated device-to-device: RCCL
// ncclBroadcast over xGMI when librccl is loadable and the devices are distinct, hipMemcpyPeer otherwise.  The library is
// resolved at run time so libbnhip.so itself links against nothing but the HIP runtime.
struct Rccl {
    void* h = nullptr;
    int (*CommInitAll)(void**, int, const int*) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool load() {
        if (h) return true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
        if (!h) return false;
        *(void**)&CommInitAll = dlsym(h, "ncclCommInitAll");
        *(void**)&CommDestroy = dlsym(h, "ncclCommDestroy");
        *(void**)&GroupStart = dlsym(h, "ncclGroupStart");
        *(void**)&GroupEnd = dlsym(h, "ncclGroupEnd");
        *(void**)&Broadcast = dlsym(h, "ncclBroadcast");
        *(void**)&GetErrorString = dlsym(h, "ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Broadcast) { dlclose(h); h = nullptr; return false; }
        return true;
    }
};
Rccl g_rccl;
std::mutex g_rccl_mu;

// returns "" on success (and sets *how), else an error text
std::string replicate_weights(bnhip_model* m, const std::string& mode, std::string* how) {
    const int n = (int)m->engs.size();
    Engine& root = *m->engs[0];
    const size_t bytes = root.weights_bytes();
    bool distinct = true;
    for (int i = 0; i < n; i++)
        for (int j = i + 1; j < n; j++) if (m->engs[i]->device == m->engs[j]->device) distinct = false;
    bool want_rccl = mode == "rccl" || (mode == "auto" && distinct && n > 1);
    if (want_rccl && !distinct) return "replicate=rccl needs distinct devices";
    if (want_rccl) {
        std::lock_guard<std::mutex> lk(g_rccl_mu);
        if (!g_rccl.load()) {
            if (mode == "rccl") return "replicate=rccl requested but librccl could not be loaded";
            want_rccl = false;
        } else {
            std::vector<int> devs(n);
            for (int i = 0; i < n; i++) devs[i] = m->engs[i]->device;
            std::vector<void*> comms(n, nullptr);
            int rc = g_rccl.CommInitAll(comms.data(), n, devs.data());
            if (rc == 0) {
                rc = g_rccl.GroupStart();
                for (int i = 0; i < n && rc == 0; i++) {
                    hipSetDevice(devs[i]);
                    // count in 4-byte words (ncclFloat32 == 7); root sends in place
                    rc = g_rccl.Broadcast(root.weights_ptr(), m->engs[i]->weights_ptr(), (bytes + 3) / 4, 7, 0, comms[i],
                                          m->engs[i]->stream);
                }
                int rc2 = g_rccl.GroupEnd();
                if (rc == 0) rc = rc2;
                for (int i = 0; i < n; i++) { hipSetDevice(devs[i]); hipStreamSynchronize(m->engs[i]->stream); }
                for (int i = 0; i < n; i++) if (comms[i]) g_rccl.CommDestroy(comms[i]);
            }
            hipSetDevice(devs[0]);
            if (rc == 0) { *how = "rccl-broadcast"; return ""; }
            if (mode == "rccl")
                return std::string("RCCL broadcast failed: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
            want_rccl = false;          // auto: fall through to peer copies
        }
    }
    for (int i = 1; i < n; i++) {
        Engine& e = *m->engs[i];
        hipError_t he;
        if (e.device == root.device) he = hipMemcpy(e.weights_ptr(), root.weights_ptr(), bytes, hipMemcpyDeviceToDevice);
        else he = hipMemcpyPeer(e.weights_ptr(), e.device, root.weights_ptr(), root.device, bytes);
        if (he != hipSuccess) return std::string("weight peer copy failed: ") + hipGetErrorString(he);
    }
    hipSetDevice(root.device);
    *how = n > 1 ? "peer-copy" : "host-upload";
    return "";
}

// runs f(engine, first_clip, clip_count, err) -> rc for every shard of [0, n_clips); multi-device handles run the shards
// concurrently on their worker threads
template <class F>
int shard_run(bnhip_model* m, int n_clips, F f) {
    const int n = (int)m->engs.size();
    std::string err;
    if (n == 1) {
        int rc = f(*m->engs[0], 0, n_clips, err);
        return rc ? set_err(rc, err) : BNHIP_OK;
    }
    std::vector<int> used;
    for (int g = 0, off = 0; g < n; g++) {
        int cnt = n_clips / n + (g < n_clips % n ? 1 : 0);
        if (cnt > 0) {
            Engine* e = m->engs[g].get();
            const int o = off;
            m->workers[g]->submit([f, e, o, cnt](std::string& er) { return f(*e, o, cnt, er); });
            used.push_back(g);
        }
        off += cnt;
    }
    int rc = 0;
    for (int g : used) { int r = m->workers[g]->wait(&err); if (r && !rc) rc = r; }
    return rc ? set_err(rc, err) : BNHIP_OK;
}

// ---------------------------------------------------------------------------------------------- one engine, one shard
// pcm_bits: 0 = float32 samples, 16 / 24 / 32 = little-endian PCM converted on the device.  The work itself - small calls
// straight through the engine, calls of >= 128 clips as chunks on alternating contexts fed from pinned staging - is
// hostpipe.cpp's host_run.
int predict_host(bnhip_model* m, const void* src, int pcm_bits, int n_clips, float* logits, float* emb) {
    if (!m || !src || !logits) return set_err(BNHIP_E_INVALID, "NULL argument");
    if (n_clips <= 0) return set_err(BNHIP_E_INVALID, "n_clips must be positive");
    Engine& e0 = m->eng();
    if (e0.device < 0) return set_err(BNHIP_E_INVALID, "plan-only model cannot run");
    if (emb && !e0.emb_dim) return set_err(BNHIP_E_INVALID, "model has no embedding output");
    const size_t in_stride = (size_t)e0.n_samples * (pcm_bits ? (size_t)pcm_bits / 8 : 4);
    const int nc = e0.n_classes, ed = e0.emb_dim;
    return shard_run(m, n_clips, [=](Engine& e, int off, int cnt, std::string& err) {
        HostJob j;
        j.src = (const char*)src + (size_t)off * in_stride; j.pcm_bits = pcm_bits; j.n_clips = cnt;
        j.logits = logits + (size_t)off * nc; j.emb = emb ? emb + (size_t)off * ed : nullptr;
        return host_run(e, j, err);
    });
}

int ensure_topk(Engine& e, int k, std::string& err) {
    if (k <= e.topk_cap) return BNHIP_OK;
    if (e.d_topk_conf) hipFree(e.d_topk_conf);
    if (e.d_topk_idx) hipFree(e.d_topk_idx);
    e.d_topk_conf = nullptr; e.d_topk_idx = nullptr; e.topk_cap = 0;
    if (hipMalloc((void**)&e.d_topk_conf, (size_t)e.max_batch * k * 4) != hipSuccess ||
        hipMalloc((void**)&e.d_topk_idx, (size_t)e.max_batch * k * 4) != hipSuccess) {
        if (e.d_topk_conf) { hipFree(e.d_topk_conf); e.d_topk_conf = nullptr; }
        err = "device allocation failed (top-k)";
        return BNHIP_E_NOMEM;
    }
    e.topk_cap = k;
    return BNHIP_OK;
}

// activation + top-k of logits that are already on the host (bnhip_postprocess_topk)
int post_topk_one(Engine& e, const float* logits, int n_clips, int activation, double sensitivity, int k, float* out_conf,
                  int32_t* out_idx, std::string& err) {
    if (hipSetDevice(e.device) != hipSuccess) { err = "hipSetDevice failed"; return BNHIP_E_RUNTIME; }
    const int n_classes = e.n_classes;
    int kk = std::min(k, n_classes);
    int rc = ensure_topk(e, kk, err);
    if (rc) return rc;
    for (int off = 0; off < n_clips; off += e.max_batch) {
        int n = std::min(e.max_batch, n_clips - off);
        hipError_t he = hipMemcpyAsync(e.d_stage_logits, logits + (size_t)off * n_classes, (size_t)n * n_classes * 4,
                                       hipMemcpyHostToDevice, e.stream);
        if (he != hipSuccess) { err = std::string("H2D copy: ") + hipGetErrorString(he); return BNHIP_E_RUNTIME; }
        launch_activation(e.d_stage_logits, e.d_post_conf, n, n_classes, activation, sensitivity, e.stream);
        launch_topk(e.d_post_conf, n, n_classes, kk, e.d_topk_conf, e.d_topk_idx, e.stream);
        hipMemcpyAsync(out_conf + (size_t)off * kk, e.d_topk_conf, (size_t)n * kk * 4, hipMemcpyDeviceToHost, e.stream);
        hipMemcpyAsync(out_idx + (size_t)off * kk, e.d_topk_idx, (size_t)n * kk * 4, hipMemcpyDeviceToHost, e.stream);
        he = hipStreamSynchronize(e.stream);
        if (he != hipSuccess) { err = std::string("top-k: ") + hipGetErrorString(he); return BNHIP_E_RUNTIME; }
    }
    return BNHIP_OK;
}

int igcd(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

int copy_out(const std::string& s, char* buf, size_t cap) {
    if (buf && cap) {
        size_t n = std::min(cap - 1, s.size());
        memcpy(buf, s.data(), n);
        buf[n] = 0;
    }
    return (int)s.size() + 1;
}

}  // namespace

// Streaming resampler state (Resampler, internal/audiocore/resample/resample.go:44-52): the polyphase filter's input
// history lives on the device between calls so that any chunking of a stream produces the samples of one call over the
// whole stream, bit for bit.
struct bnhip_resampler {
    int device = 0, rate_in = 0, rate_out = 0, L = 1, M = 1, T = 0, half = 0;
    float* d_table = nullptr;
    float* d_work = nullptr;      // [hist | new chunk] as float32
    size_t work_cap = 0;          // floats
    void* d_in = nullptr;  size_t in_cap = 0;     // raw input staging (bytes)
    void* d_out = nullptr; size_t out_cap = 0;    // output staging (bytes)
    long long n_total = 0;        // input samples consumed so far
    long long i_next = 0;         // next output index
    long long n_base = 0;         // stream index of d_work[0]
    int n_hist = 0;               // valid history samples at the front of d_work
    hipStream_t stream = nullptr;
};

extern "C" {

const char* bnhip_version(void) { return "bnhip 0.2 (gfx950)"; }
const char* bnhip_last_error(void) { return g_err.c_str(); }

int bnhip_last_error_copy(char* buf, size_t cap) {
    if (!buf || !cap) return (int)g_err.size() + 1;
    size_t n = std::min(cap - 1, g_err.size());
    memcpy(buf, g_err.data(), n);
    buf[n] = 0;
    return (int)g_err.size() + 1;
}

int bnhip_init(int* n_devices) {
    BN_GUARD_BEGIN
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
    BN_GUARD_END((void)0)
}

int bnhip_host_alloc(size_t n_bytes, void** out) {
    BN_GUARD_BEGIN
    if (!out || !n_bytes) return set_err(BNHIP_E_INVALID, "bnhip_host_alloc: null output pointer or zero size");
    *out = nullptr;
    int rc = bnhip_init(nullptr);
    if (rc != BNHIP_OK) return rc;
    void* p = nullptr;
    // (round 6: from the NUMA node of the calling thread's current device when it has room - the buffer a Go classifier keeps per
    // model is read by that device's copy engines on every call; portable: pinned for every device of a multi-GPU handle)
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); cur = -1; }
    bnhip::NumaPrefer near_gpu(bnhip::device_numa_node(cur));
    hipError_t e = hipHostMalloc(&p, n_bytes, hipHostMallocPortable);
    if (e != hipSuccess) { (void)hipGetLastError(); return set_err(e == hipErrorOutOfMemory ? BNHIP_E_NOMEM : BNHIP_E_RUNTIME, std::string("pinned host allocation failed: ") + hipGetErrorString(e)); }
    *out = p;
    return BNHIP_OK;
    BN_GUARD_END((void)0)
}

int bnhip_host_free(void* p) {
    BN_GUARD_BEGIN
    if (!p) return BNHIP_OK;
    hipError_t e = hipHostFree(p);
    if (e != hipSuccess) { (void)hipGetLastError(); return set_err(BNHIP_E_INVALID, std::string("bnhip_host_free: not a bnhip_host_alloc pointer: ") + hipGetErrorString(e)); }
    return BNHIP_OK;
    BN_GUARD_END((void)0)
}

// ---------------------------------------------------------------------------------------------- window assembler (row a3)
struct bnhip_windows {
    std::unique_ptr<bnhip::WindowAssembler> a;
    uint8_t* batch = nullptr;
    bool pinned = false;
};

static void windows_free(bnhip_windows* w) {
    if (!w) return;
    if (w->batch) {
        if (w->pinned) { if (hipHostFree(w->batch) != hipSuccess) (void)hipGetLastError(); }
        else std::free(w->batch);
    }
    delete w;
}

int bnhip_windows_create(size_t overlap_bytes, size_t read_bytes, int max_batch, bnhip_windows** out) {
    if (!out) return set_err(BNHIP_E_INVALID, "out is NULL");
    *out = nullptr;
    // NewAnalysisBuffer's checks (analysis.go:65-90); the capacity ones are per source (bnhip_windows_add_source)
    if (read_bytes == 0) return set_err(BNHIP_E_INVALID, "invalid read size: 0, must be greater than 0");
    if (read_bytes < overlap_bytes) return set_err(BNHIP_E_INVALID, "read size must be >= overlap size");
    if (max_batch < 1) return set_err(BNHIP_E_INVALID, "max_batch must be positive");
    const size_t wb = overlap_bytes + read_bytes;
    if (wb < read_bytes || wb > ((size_t)1 << 40) / (size_t)max_batch) return set_err(BNHIP_E_INVALID, "window batch too large");
    bnhip_windows* w = nullptr;
    BN_GUARD_BEGIN
    w = new bnhip_windows();
    w->a = std::make_unique<bnhip::WindowAssembler>(overlap_bytes, read_bytes, max_batch);
    const size_t bytes = wb * (size_t)max_batch;
    // page-locked when there is a device to read it (the host pipeline then copies straight out of it); plain memory
    // otherwise, so that the byte work can be used and tested on a box without one
    int ndev = 0;
    if (bnhip_init(&ndev) == BNHIP_OK && ndev > 0) {
        void* p = nullptr;
        if (hipHostMalloc(&p, bytes, hipHostMallocPortable) == hipSuccess) { w->batch = static_cast<uint8_t*>(p); w->pinned = true; }
        else (void)hipGetLastError();
    }
    if (!w->batch) {
        w->batch = static_cast<uint8_t*>(std::malloc(bytes));
        if (!w->batch) { windows_free(w); w = nullptr; return set_err(BNHIP_E_NOMEM, "out of host memory"); }
    }
    *out = w;
    return BNHIP_OK;
    BN_GUARD_END(windows_free(w))
}

int bnhip_windows_info(const bnhip_windows* w, size_t* window_bytes, int* max_batch, int* pinned, int* n_sources) {
    if (!w) return set_err(BNHIP_E_INVALID, "NULL argument");
    BN_GUARD_BEGIN
    if (window_bytes) *window_bytes = w->a->window_bytes();
    if (max_batch) *max_batch = w->a->max_batch();
    if (pinned) *pinned = w->pinned ? 1 : 0;
    if (n_sources) *n_sources = w->a->n_sources();
    return BNHIP_OK;
    BN_GUARD_END((void)0)
}

int bnhip_windows_add_source(bnhip_windows* w, const char* source_id, size_t capacity_bytes, int* out_source) {
    if (!w || !out_source) return set_err(BNHIP_E_INVALID, "NULL argument");
    *out_source = -1;
    if (!source_id || !*source_id) return set_err(BNHIP_E_INVALID, "source ID must not be empty");
    if (capacity_bytes == 0) return set_err(BNHIP_E_INVALID, "invalid analysis buffer capacity: 0, must be greater than 0");
    if (capacity_bytes < w->a->read_bytes()) return set_err(BNHIP_E_INVALID, "capacity must be >= read size");
    if (capacity_bytes > ((size_t)1 << 40)) return set_err(BNHIP_E_INVALID, "capacity too large");
    BN_GUARD_BEGIN
    const int idx = w->a->add_source(source_id, capacity_bytes);
    if (idx < 0) return set_err(BNHIP_E_INVALID, "capacity must be >= read size");
    *out_source = idx;
    return BNHIP_OK;
    BN_GUARD_END((void)0)
}

int bnhip_windows_remove_source(bnhip_windows* w, int source) {
    if (!w) return set_err(BNHIP_E_INVALID, "NULL argument");
    BN_GUARD_BEGIN
    return w->a->remove_source(source) ? BNHIP_OK : set_err(BNHIP_E_INVALID, "no such source");
    BN_GUARD_END((void)0)
}

int bnhip_windows_write(bnhip_windows* w, int source, const void* data, size_t n_bytes) {
    if (!w || (!data && n_bytes)) return set_err(BNHIP_E_INVALID, "NULL argument");
    BN_GUARD_BEGIN
    return w->a->write(source, data, n_bytes) ? BNHIP_OK : set_err(BNHIP_E_INVALID, "no such source");
    BN_GUARD_END((void)0)
}

int bnhip_windows_collect(bnhip_windows* w, int cap, int* sources, int* n_windows, const void** batch) {
    if (!w || !sources || !n_windows) return set_err(BNHIP_E_INVALID, "NULL argument");
    *n_windows = 0;
    if (batch) *batch = w->batch;
    if (cap < 0) return set_err(BNHIP_E_INVALID, "cap must not be negative");
    BN_GUARD_BEGIN
    *n_windows = w->a->collect(w->batch, cap, sources);
    return BNHIP_OK;
    BN_GUARD_END((void)0)
}

int bnhip_windows_ready(const bnhip_windows* w, int* n_ready) {
    if (!w || !n_ready) return set_err(BNHIP_E_INVALID, "NULL argument");
    BN_GUARD_BEGIN
    *n_ready = w->a->ready();
    return BNHIP_OK;
    BN_GUARD_END((void)0)
}

int bnhip_windows_stats(const bnhip_windows* w, int source, uint64_t* writes, uint64_t* overwrites, size_t* buffered_bytes) {
    if (!w) return set_err(BNHIP_E_INVALID, "NULL argument");
    BN_GUARD_BEGIN
    return w->a->stats(source, writes, overwrites, buffered_bytes) ? BNHIP_OK : set_err(BNHIP_E_INVALID, "no such source");
    BN_GUARD_END((void)0)
}

int bnhip_windows_reset(bnhip_windows* w, int source) {
    if (!w) return set_err(BNHIP_E_INVALID, "NULL argument");
    BN_GUARD_BEGIN
    return w->a->reset(source) ? BNHIP_OK : set_err(BNHIP_E_INVALID, "no such source");
    BN_GUARD_END((void)0)
}

void bnhip_windows_destroy(bnhip_windows* w) {
    try { windows_free(w); } catch (...) {}
}

void bnhip_shutdown(void) {
    try {
        std::lock_guard<std::mutex> lk(g_init_mu);
        g_devices = -1;
    } catch (...) {}
}

int bnhip_model_create(const void* blob, size_t n_bytes, const char* opts_json, bnhip_model** out) {
    if (!out) return set_err(BNHIP_E_INVALID, "out is NULL");
    *out = nullptr;
    bnhip_model* m = nullptr;
    BN_GUARD_BEGIN
    if (!blob || n_bytes == 0) return set_err(BNHIP_E_INVALID, "empty model blob");
    // "plan_only": parse + plan on the CPU, no device touched (diagnostics / CPU-side tests); such a
    // model answers info/describe and rejects predict calls.
    const bool plan_only = json_int(opts_json, "plan_only", 0) != 0;
    std::vector<int> devices = json_int_array(opts_json, "devices");
    if (devices.empty()) devices.push_back((int)json_int(opts_json, "device", 0));
    if (devices.size() > 64) return set_err(BNHIP_E_INVALID, "too many devices");
    int max_batch = (int)json_int(opts_json, "max_batch", 256);
    if (max_batch < 1 || max_batch > 4096) return set_err(BNHIP_E_INVALID, "max_batch must be in [1, 4096]");
    if (!plan_only) {
        int rc = bnhip_init(nullptr);
        if (rc != BNHIP_OK) return rc;
        for (int device : devices) {
            if (device < 0 || device >= g_devices) return set_err(BNHIP_E_INVALID, "device ordinal out of range");
            if (!is_gfx950(device)) return set_err(BNHIP_E_NO_DEVICE, "selected device is not gfx950");
        }
    }

    // container: TFLite flatbuffer ("TFL3" at byte 4) or ONNX protobuf (internal/inference/onnx/classifier.go:268-430)
    TflModel tm;
    std::string err;
    const bool is_tfl = n_bytes >= 8 && memcmp((const char*)blob + 4, "TFL3", 4) == 0;
    if (is_tfl) {
        if (!parse_tflite(blob, n_bytes, &tm, &err)) return set_err(BNHIP_E_MODEL, err);
    } else {
        int ocode = BNHIP_E_MODEL;
        if (!parse_onnx(blob, n_bytes, &tm, &err, &ocode)) return set_err(ocode, err);
    }
    if (getenv("BNHIP_DUMP_IR")) {                      // diagnostics: the operator list the planner will see
        for (size_t oi = 0; oi < tm.ops.size(); oi++) {
            const TflOp& o = tm.ops[oi];
            fprintf(stderr, "[bnhip] ir %3zu %-18s", oi, op_name(o.code));
            for (int t : o.inputs) {
                if (t < 0) { fprintf(stderr, " -"); continue; }
                fprintf(stderr, " %s%d[", tm.tensors[t].data ? "c" : "t", t);
                for (size_t k = 0; k < tm.tensors[t].shape.size(); k++) fprintf(stderr, "%s%d", k ? "," : "", tm.tensors[t].shape[k]);
                fprintf(stderr, "]");
            }
            fprintf(stderr, " ->");
            for (int t : o.outputs) {
                fprintf(stderr, " t%d[", t);
                for (size_t k = 0; k < tm.tensors[t].shape.size(); k++) fprintf(stderr, "%s%d", k ? "," : "", tm.tensors[t].shape[k]);
                fprintf(stderr, "]");
            }
            fprintf(stderr, "\n");
        }
    }
    if (!validate_graph(tm, &err)) return set_err(BNHIP_E_MODEL, err);

    m = new bnhip_model();
    const char* lenv = getenv("BNHIP_LANES");            // experiment switches; the option wins when given
    const char* denv = getenv("BNHIP_DEPTH");
    const char* fenv = getenv("BNHIP_FE_FFT");
    const char* genv = getenv("BNHIP_GRAPHS");
    const char* benv = getenv("BNHIP_BF16X3");
    const int n_eng = (int)devices.size();
    for (int i = 0; i < n_eng; i++) {
        std::unique_ptr<Engine> e(new Engine());
        e->no_reuse = json_int(opts_json, "debug_no_reuse", 0) != 0;
        e->autotune = json_int(opts_json, "autotune", 1) != 0;
        {
            const char* tenv = getenv("BNHIP_TUNE_DIR");
            e->tune_dir = json_str(opts_json, "tune_dir", tenv ? tenv : "");
        }
        e->n_lanes = (int)json_int(opts_json, "lanes", lenv ? atoi(lenv) : 2);
        e->depth = (int)json_int(opts_json, "depth", denv ? atoi(denv) : 1);
        {
            const char* henv = getenv("BNHIP_HOST_DEPTH");
            e->host_depth = (int)json_int(opts_json, "host_depth", henv ? atoi(henv) : 2);
        }
        e->frontend_fft = (int)json_int(opts_json, "frontend_fft", fenv ? atoi(fenv) : -1);
        e->use_graphs = json_int(opts_json, "graphs", genv ? atoi(genv) : 0) != 0;
        // default 1: per layer where the create-time autotuner measures the split-bf16 kernel faster (fp32-equivalent
        // products; tests/test_bf16x3.py holds the error comparison against the float64 arbiter that decided the default)
        e->bf16x3 = (int)json_int(opts_json, "bf16x3", benv ? atoi(benv) : 1);
        e->logits_output = (int)json_int(opts_json, "logits_output", -1);
        e->embedding_output = (int)json_int(opts_json, "embedding_output", -2);       // -1: no embedding; -2: the family rule
        {
            // "precision": "f32" (default) | "bf16": MFMA operands rounded to bf16, fp32 accumulate and storage (BASELINE
            // configs[4] asks for this on Perch; never the default: v2.4 in reduced precision is known to fail, model_openvino.go:99-103)
            const char* penv = getenv("BNHIP_PRECISION");
            std::string prec = json_str(opts_json, "precision", penv ? penv : "f32");
            if (prec != "f32" && prec != "bf16") { delete m; return set_err(BNHIP_E_INVALID, "precision must be \"f32\" or \"bf16\""); }
            e->precision = prec == "bf16" ? 1 : 0;
            if (e->precision && !e->bf16x3) e->bf16x3 = 1;      // the bf16 kernels read the split weight images' first plane
        }
        e->defer_weights = i > 0 && !plan_only;
        int code = BNHIP_E_UNSUPPORTED;
        TflModel copy = tm;                               // tensors point into the caller's blob / tm-owned storage: cheap
        if (!e->build(std::move(copy), devices[i], max_batch, plan_only, &err, &code)) {
            delete m;
            return set_err(code == BNHIP_OK ? BNHIP_E_UNSUPPORTED : code, err);
        }
        m->engs.push_back(std::move(e));
    }
    if (n_eng > 1 && !plan_only) {
        std::string how;
        std::string rerr = replicate_weights(m, json_str(opts_json, "replicate", "auto"), &how);
        if (!rerr.empty()) { delete m; return set_err(BNHIP_E_RUNTIME, rerr); }
        m->replication = how;
        for (int i = 0; i < n_eng; i++) {
            m->workers.emplace_back(new Worker());
            m->workers.back()->start(devices[i]);
        }
        // create-time autotune of the deferred engines, concurrently on their own devices
        for (int i = 1; i < n_eng; i++) {
            Engine* e = m->engs[i].get();
            m->workers[i]->submit([e](std::string&) { e->finish_deferred(); return 0; });
        }
        for (int i = 1; i < n_eng; i++) m->workers[i]->wait(nullptr);
        hipSetDevice(devices[0]);
    } else if (n_eng == 1 && !plan_only && json_str(opts_json, "replicate", "") == "rccl") {
        // single device, RCCL explicitly requested: run the broadcast code path with one rank (in place) so that the
        // library load and call sequence are exercised on a one-GPU box
        std::string how;
        std::string rerr = replicate_weights(m, "rccl", &how);
        if (!rerr.empty()) { delete m; return set_err(BNHIP_E_RUNTIME, rerr); }
        m->replication = how;
    }
    *out = m;
    return BNHIP_OK;
    BN_GUARD_END(delete m)
}

int bnhip_model_info(const bnhip_model* m, int* n_samples, int* n_classes, int* emb_dim) {
    if (!m) return set_err(BNHIP_E_INVALID, "model is NULL");
    if (n_samples) *n_samples = m->eng().n_samples;
    if (n_classes) *n_classes = m->eng().n_classes;
    if (emb_dim) *emb_dim = m->eng().emb_dim;
    return BNHIP_OK;
}

int bnhip_model_devices(const bnhip_model* m, int* devices, int cap) {
    if (!m) return set_err(BNHIP_E_INVALID, "model is NULL");
    const int n = (int)m->engs.size();
    for (int i = 0; i < n && i < cap && devices; i++) devices[i] = m->engs[i]->device;
    return n;
}

void bnhip_model_destroy(bnhip_model* m) {
    if (!m) return;
    try {
        m->workers.clear();                             // joins the worker threads first
        for (auto& e : m->engs) {
            if (e && e->device >= 0) hipSetDevice(e->device);
            e.reset();
        }
        delete m;
    } catch (...) {}
}

int bnhip_set_stream(bnhip_model* m, void* hip_stream) {
    if (!m || m->eng().device < 0) return set_err(BNHIP_E_INVALID, "model is NULL or plan-only");
    if (m->engs.size() > 1) return set_err(BNHIP_E_INVALID, "bnhip_set_stream: multi-device handles own their streams");
    BN_GUARD_BEGIN
    Engine& e = m->eng();
    hipSetDevice(e.device);
    e.drop_graphs();                                    // captured on the old stream
    e.sync_contexts();
    if (e.stream) hipStreamSynchronize(e.stream);       // (the engine keeps its own streams: contexts and lanes run on them)
    e.stream = reinterpret_cast<hipStream_t>(hip_stream);
    e.own_stream = false;
    return BNHIP_OK;
    BN_GUARD_END((void)0)
}

int bnhip_synchronize(bnhip_model* m) {
    if (!m || m->eng().device < 0) return set_err(BNHIP_E_INVALID, "model is NULL or plan-only");
    BN_GUARD_BEGIN
    for (auto& ep : m->engs) {
        Engine& e = *ep;
        hipSetDevice(e.device);
        e.sync_contexts();
        hipError_t he = hipStreamSynchronize(e.stream);
        if (he != hipSuccess) return set_err(BNHIP_E_RUNTIME, std::string("hipStreamSynchronize: ") + hipGetErrorString(he));
    }
    hipSetDevice(m->eng().device);
    return BNHIP_OK;
    BN_GUARD_END((void)0)
}

int bnhip_predict_device(bnhip_model* m, const float* d_samples, int n_clips, float* d_logits, float* d_emb) {
    if (!m || !d_samples || !d_logits) return set_err(BNHIP_E_INVALID, "NULL argument");
    if (n_clips <= 0) return set_err(BNHIP_E_INVALID, "n_clips must be positive");
    if (m->engs.size() > 1) return set_err(BNHIP_E_INVALID, "bnhip_predict_device: device pointers belong to one device; use a single-device handle");
    BN_GUARD_BEGIN
    Engine& e = m->eng();
    if (e.device < 0) return set_err(BNHIP_E_INVALID, "plan-only model cannot run");
    if (d_emb && !e.emb_dim) return set_err(BNHIP_E_INVALID, "model has no embedding output");
    if (hipSetDevice(e.device) != hipSuccess) return set_err(BNHIP_E_RUNTIME, "hipSetDevice failed");
    std::string err;
    for (int off = 0; off < n_clips; off += e.max_batch) {
        int n = std::min(e.max_batch, n_clips - off);
        if (!e.run_pipelined(d_samples + (size_t)off * e.n_samples, n, d_logits + (size_t)off * e.n_classes,
                             d_emb ? d_emb + (size_t)off * e.emb_dim : nullptr, &err))
            return set_err(BNHIP_E_RUNTIME, err);
    }
    return BNHIP_OK;
    BN_GUARD_END((void)0)
}

int bnhip_predict(bnhip_model* m, const float* samples, int n_clips, float* logits, float* emb) {
    BN_GUARD_BEGIN
    return predict_host(m, samples, 0, n_clips, logits, emb);
    BN_GUARD_END((void)0)
}

int bnhip_predict_pcm16(bnhip_model* m, const int16_t* pcm, int n_clips, float* logits, float* emb) {
    BN_GUARD_BEGIN
    return predict_host(m, pcm, 16, n_clips, logits, emb);
    BN_GUARD_END((void)0)
}

int bnhip_predict_pcm(bnhip_model* m, const void* pcm, int bits_per_sample, int n_clips, float* logits, float* emb) {
    if (bits_per_sample != 16 && bits_per_sample != 24 && bits_per_sample != 32)
        return set_err(BNHIP_E_INVALID, "unsupported bit depth (supported: 16, 24, 32)");
    BN_GUARD_BEGIN
    return predict_host(m, pcm, bits_per_sample, n_clips, logits, emb);
    BN_GUARD_END((void)0)
}

int bnhip_postprocess_topk(bnhip_model* m, const float* logits, int n_clips, int n_classes, int activation,
                           double sensitivity, int k, float* out_conf, int32_t* out_idx) {
    if (!m || !logits || !out_conf || !out_idx) return set_err(BNHIP_E_INVALID, "NULL argument");
    BN_GUARD_BEGIN
    Engine& e = m->eng();
    if (n_clips <= 0 || k <= 0) return set_err(BNHIP_E_INVALID, "n_clips and k must be positive");
    if (e.device < 0) return set_err(BNHIP_E_INVALID, "plan-only model cannot run");
    if (n_classes != e.n_classes) return set_err(BNHIP_E_INVALID, "n_classes does not match the model");
    if (activation < 0 || activation > 2) return set_err(BNHIP_E_INVALID, "unknown activation");
    if ((size_t)n_classes * 4 > 150 * 1024) return set_err(BNHIP_E_UNSUPPORTED, "too many classes for the LDS top-k");
    const int kk = std::min(k, n_classes);
    return shard_run(m, n_clips, [=](Engine& en, int off, int cnt, std::string& err) {
        return post_topk_one(en, logits + (size_t)off * n_classes, cnt, activation, sensitivity, k,
                             out_conf + (size_t)off * kk, out_idx + (size_t)off * kk, err);
    });
    BN_GUARD_END((void)0)
}

// pcm_bits as in predict_host
static int predict_topk_host(bnhip_model* m, const void* src, int pcm_bits, int n_clips, int activation, double sensitivity, int k,
                             float* out_conf, int32_t* out_idx) {
    if (!m || !src || !out_conf || !out_idx) return set_err(BNHIP_E_INVALID, "NULL argument");
    Engine& e = m->eng();
    if (n_clips <= 0 || k <= 0) return set_err(BNHIP_E_INVALID, "n_clips and k must be positive");
    if (activation < 0 || activation > 2) return set_err(BNHIP_E_INVALID, "unknown activation");
    if ((size_t)e.n_classes * 4 > 150 * 1024) return set_err(BNHIP_E_UNSUPPORTED, "too many classes for the LDS top-k");
    if (e.device < 0) return set_err(BNHIP_E_INVALID, "plan-only model cannot run");
    const int kk = std::min(k, e.n_classes);
    const size_t in_stride = (size_t)e.n_samples * (pcm_bits ? (size_t)pcm_bits / 8 : 4);
    return shard_run(m, n_clips, [=](Engine& en, int off, int cnt, std::string& err) {
        HostJob j;
        j.src = (const char*)src + (size_t)off * in_stride; j.pcm_bits = pcm_bits; j.n_clips = cnt;
        j.topk = k; j.activation = activation; j.sensitivity = sensitivity;
        j.out_conf = out_conf + (size_t)off * kk; j.out_idx = out_idx + (size_t)off * kk;
        return host_run(en, j, err);
    });
}

int bnhip_predict_topk(bnhip_model* m, const float* samples, int n_clips, int activation, double sensitivity, int k,
                       float* out_conf, int32_t* out_idx) {
    BN_GUARD_BEGIN
    return predict_topk_host(m, samples, 0, n_clips, activation, sensitivity, k, out_conf, out_idx);
    BN_GUARD_END((void)0)
}

int bnhip_predict_pcm_topk(bnhip_model* m, const void* pcm, int bits_per_sample, int n_clips, int activation, double sensitivity,
                           int k, float* out_conf, int32_t* out_idx) {
    BN_GUARD_BEGIN
    if (bits_per_sample != 16 && bits_per_sample != 24 && bits_per_sample != 32)
        return set_err(BNHIP_E_INVALID, "unsupported bit depth: " + std::to_string(bits_per_sample) + " (supported: 16, 24, 32)");
    return predict_topk_host(m, pcm, bits_per_sample, n_clips, activation, sensitivity, k, out_conf, out_idx);
    BN_GUARD_END((void)0)
}

// One tick: who is ready -> rows filled chunk by chunk under the device's work on the previous chunks -> top-k.
int bnhip_windows_predict_topk(bnhip_windows* w, bnhip_model* m, int bits_per_sample, int activation, double sensitivity, int k,
                               int* sources, int* n_windows, float* out_conf, int32_t* out_idx, const void** batch) {
    if (!w || !m || !sources || !n_windows || !out_conf || !out_idx) return set_err(BNHIP_E_INVALID, "NULL argument");
    *n_windows = 0;
    if (batch) *batch = w->batch;
    if (bits_per_sample != 16 && bits_per_sample != 24 && bits_per_sample != 32)
        return set_err(BNHIP_E_INVALID, "unsupported bit depth: " + std::to_string(bits_per_sample) + " (supported: 16, 24, 32)");
    if (k <= 0) return set_err(BNHIP_E_INVALID, "n_clips and k must be positive");
    if (activation < 0 || activation > 2) return set_err(BNHIP_E_INVALID, "unknown activation");
    bool begun = false;
    BN_GUARD_BEGIN
    Engine& e = m->eng();
    if (e.device < 0) return set_err(BNHIP_E_INVALID, "plan-only model cannot run");
    if ((size_t)e.n_classes * 4 > 150 * 1024) return set_err(BNHIP_E_UNSUPPORTED, "too many classes for the LDS top-k");
    const size_t clip_bytes = (size_t)e.n_samples * (size_t)(bits_per_sample / 8);
    if (w->a->window_bytes() != clip_bytes)
        return set_err(BNHIP_E_INVALID, "window size mismatch: assembler " + std::to_string(w->a->window_bytes()) + " bytes, model clip " +
                                        std::to_string(clip_bytes) + " bytes");
    bnhip::WindowAssembler& a = *w->a;
    const int n = a.collect_begin(a.max_batch(), sources);
    begun = true;
    if (n == 0) { a.collect_end(); return BNHIP_OK; }       // "try again later"
    // every listed source gives up exactly one read, whatever happens to the device call: ranges the pipeline did not get to
    // (an error on the way) are consumed afterwards, as the reference's monitor has consumed its window before ProcessData fails
    std::vector<char> filled((size_t)n, 0);
    uint8_t* rows = w->batch;
    auto fill = [&a, &filled, rows, sources](int first, int cnt) {
        a.collect_rows(rows, sources, first, cnt);
        for (int r = first; r < first + cnt; r++) filled[(size_t)r] = 1;
    };
    const int kk = std::min(k, e.n_classes);
    int rc = shard_run(m, n, [=](Engine& en, int off, int cnt, std::string& err) {
        HostJob j;
        j.src = rows + (size_t)off * clip_bytes; j.pcm_bits = bits_per_sample; j.n_clips = cnt;
        j.topk = k; j.activation = activation; j.sensitivity = sensitivity;
        j.out_conf = out_conf + (size_t)off * kk; j.out_idx = out_idx + (size_t)off * kk;
        j.prepare = [=](int first, int c) { fill(off + first, c); };
        return host_run(en, j, err);
    });
    for (int r = 0; r < n;) {
        if (filled[(size_t)r]) { r++; continue; }
        int q = r;
        while (q < n && !filled[(size_t)q]) q++;
        fill(r, q - r);
        r = q;
    }
    a.collect_end();
    begun = false;
    *n_windows = n;
    return rc;
    BN_GUARD_END(if (begun) w->a->collect_end())
}

int bnhip_us_frame_cv(int device, const double* samples, int n_clips, int n, int sample_rate, int fft_size, int hop,
                      int split_hz, double* cv, int32_t* ok) {
    if (!samples || !cv || !ok || n_clips <= 0) return set_err(BNHIP_E_INVALID, "NULL/empty argument");
    BN_GUARD_BEGIN
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
        const double* d_tw = us_twiddles(device, fft_size);
        if (!d_tw) he = hipErrorOutOfMemory;
        else {
            launch_us_frame_power(d_s, 0, n_clips, n, fft_size, hop, frames, split_bin, d_tw, d_p, nullptr);
            launch_us_cv(d_p, n_clips, frames, d_cv, nullptr);
            he = hipGetLastError();
            if (he == hipSuccess) he = hipMemcpy(cv, d_cv, (size_t)n_clips * 8, hipMemcpyDeviceToHost);
        }
    }
    if (d_s) hipFree(d_s);
    if (d_p) hipFree(d_p);
    if (d_cv) hipFree(d_cv);
    if (he != hipSuccess) return set_err(BNHIP_E_RUNTIME, std::string("us_frame_cv: ") + hipGetErrorString(he));
    for (int i = 0; i < n_clips; i++) ok[i] = 1;
    return BNHIP_OK;
    BN_GUARD_END((void)0)
}

// Device-resident form: samples (float64, or raw int16 PCM) and results stay in HBM, work is enqueued on `hip_stream`
// (NULL = the default stream) and not synchronised.  d_scratch holds n_clips * frames float64 frame powers.
int bnhip_us_frame_cv_device(int device, const void* d_samples, int pcm16, int n_clips, int n, int sample_rate, int fft_size, int hop,
                             int split_hz, double* d_scratch, double* d_cv, void* hip_stream) {
    if (!d_samples || !d_cv || !d_scratch || n_clips <= 0) return set_err(BNHIP_E_INVALID, "NULL/empty argument");
    BN_GUARD_BEGIN
    bool valid = !(n < fft_size || sample_rate <= 0 || fft_size < 2 || hop <= 0) && (fft_size & (fft_size - 1)) == 0 &&
                 !(split_hz < 0 || split_hz >= sample_rate / 2);
    int frames = valid ? 1 + (n - fft_size) / hop : 0;
    if (!valid || frames < 2) return set_err(BNHIP_E_INVALID, "geometry rejected by the filter's guards (filter.go:21-37): use the host entry for the (0, false) answer");
    if ((size_t)fft_size * 16 > 160 * 1024 - 256) return set_err(BNHIP_E_UNSUPPORTED, "FFT size exceeds the LDS-resident limit (8192)");
    int rc = bnhip_init(nullptr);
    if (rc) return rc;
    if (device < 0 || device >= g_devices) return set_err(BNHIP_E_INVALID, "device ordinal out of range");
    hipSetDevice(device);
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    const int split_bin = (int)((double)split_hz / ((double)sample_rate / (double)fft_size));
    const double* d_tw = us_twiddles(device, fft_size);
    if (!d_tw) return set_err(BNHIP_E_NOMEM, "device allocation failed (FFT twiddle table)");
    launch_us_frame_power(d_samples, pcm16 != 0, n_clips, n, fft_size, hop, frames, split_bin, d_tw, d_scratch, st);
    launch_us_cv(d_scratch, n_clips, frames, d_cv, st);
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) return set_err(BNHIP_E_RUNTIME, std::string("us_frame_cv_device: ") + hipGetErrorString(he));
    return frames;
    BN_GUARD_END((void)0)
}

int bnhip_debug_fetch(bnhip_model* m, int tensor_index, int n_clips, float* out, size_t cap_floats) {
    if (!m || !out || m->eng().device < 0) return set_err(BNHIP_E_INVALID, "NULL argument or plan-only model");
    BN_GUARD_BEGIN
    Engine& e = m->eng();
    // tensor_index <= -2 names a plan value directly (value id = -tensor_index - 2: the "out_v" / "out2_v" of a describe()
    // step, which also covers internal scratch such as the squeeze-excite partial sums)
    int vid = -1;
    if (tensor_index <= -2) {
        vid = -tensor_index - 2;
        if (vid >= (int)e.vals.size()) return set_err(BNHIP_E_INVALID, "value id out of range");
    } else {
        auto it = e.tensor_value.find(tensor_index);
        if (it == e.tensor_value.end()) return set_err(BNHIP_E_INVALID, "tensor is not materialised by the plan (fused away)");
        vid = it->second;
    }
    const Value& v = e.vals[vid];
    if (v.external) return set_err(BNHIP_E_INVALID, "tensor is bound externally (graph input/logits)");
    size_t n = v.elems * (size_t)n_clips;
    if (n > cap_floats || n_clips > e.max_batch) return set_err(BNHIP_E_INVALID, "buffer too small");
    hipSetDevice(e.device);
    hipStreamSynchronize(e.stream);
    if (hipMemcpy(out, e.value_ptr(vid), n * 4, hipMemcpyDeviceToHost) != hipSuccess)
        return set_err(BNHIP_E_RUNTIME, "debug fetch copy failed");
    return (int)v.elems;
    BN_GUARD_END((void)0)
}

// ------------------------------------------------------------------------------------------------ resampler
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
        lrc = launch_resample(d_in, d_out, d_tab, pcm16, pcm16, n_clips, n_in, no, L, M, T, half, 0, 0, nullptr);
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
    BN_GUARD_BEGIN
    return resample_impl(device, in, false, n_clips, n_in, rate_in, rate_out, out, n_out_cap, n_out);
    BN_GUARD_END((void)0)
}

int bnhip_resample_pcm16(int device, const int16_t* in, int n_clips, int n_in, int rate_in, int rate_out, int16_t* out,
                         int n_out_cap, int* n_out) {
    BN_GUARD_BEGIN
    return resample_impl(device, in, true, n_clips, n_in, rate_in, rate_out, out, n_out_cap, n_out);
    BN_GUARD_END((void)0)
}

// ---- streaming form
static void resampler_free(bnhip_resampler* r) {
    if (!r) return;
    hipSetDevice(r->device);
    if (r->stream) { hipStreamSynchronize(r->stream); hipStreamDestroy(r->stream); }
    for (void* p : {(void*)r->d_table, (void*)r->d_work, r->d_in, r->d_out}) if (p) hipFree(p);
    delete r;
}

int bnhip_resampler_create(int device, int rate_in, int rate_out, bnhip_resampler** out) {
    if (!out) return set_err(BNHIP_E_INVALID, "out is NULL");
    *out = nullptr;
    if (rate_in <= 0 || rate_out <= 0) return set_err(BNHIP_E_INVALID, "sample rates must be positive");
    if (rate_in == rate_out) return BNHIP_OK;            // NewResampler returns nil, nil: no resampling required (resample.go:58-60)
    bnhip_resampler* r = nullptr;
    BN_GUARD_BEGIN
    int rc = bnhip_init(nullptr);
    if (rc) return rc;
    if (device < 0 || device >= g_devices) return set_err(BNHIP_E_INVALID, "device ordinal out of range");
    hipSetDevice(device);
    r = new bnhip_resampler();
    r->device = device; r->rate_in = rate_in; r->rate_out = rate_out;
    int g = igcd(rate_in, rate_out);
    r->L = rate_out / g; r->M = rate_in / g;
    std::vector<float> table;
    resample_design(r->L, r->M, 5.0, 10, &table, &r->T, &r->half);
    // same LDS bound as the kernel launch (phase table + worst-case span of 256 outputs)
    if (((size_t)r->L * r->T + (size_t)(255LL * r->M / r->L + r->T + 2)) * 4 > 150 * 1024) {
        delete r; r = nullptr;
        return set_err(BNHIP_E_UNSUPPORTED, "resample ratio needs a phase table larger than LDS");
    }
    hipError_t he = hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking);
    if (he == hipSuccess) he = hipMalloc((void**)&r->d_table, table.size() * 4);
    if (he == hipSuccess) he = hipMemcpy(r->d_table, table.data(), table.size() * 4, hipMemcpyHostToDevice);
    if (he != hipSuccess) { resampler_free(r); r = nullptr; return set_err(BNHIP_E_RUNTIME, std::string("resampler create: ") + hipGetErrorString(he)); }
    *out = r;
    return BNHIP_OK;
    BN_GUARD_END(resampler_free(r))
}

// outputs computable once n_total inputs are known: every i whose newest tap n0(i) = floor((i*M + half)/L) < n_total
static long long resampler_ready(const bnhip_resampler* r, long long n_total) {
    long long num = n_total * r->L - r->half;
    if (num <= 0) return 0;
    return (num + r->M - 1) / r->M;
}

int bnhip_resampler_estimate(const bnhip_resampler* r, int n_in) {
    if (!r || n_in <= 0) return 0;
    // EstimateOutput analogue (resample.go:83-88): an upper bound for any call, whatever the state
    return (int)(((long long)n_in * r->L + r->M - 1) / r->M) + 1;
}

// flush: 0 = emit what the inputs so far determine; 1 = end of stream (future inputs are zeros), then reset
static int resampler_run(bnhip_resampler* r, const void* in, bool pcm16, int n_in, void* out, int out_cap, int* n_out, int flush) {
    if (!r) return set_err(BNHIP_E_INVALID, "resampler is NULL");
    if (n_out) *n_out = 0;
    if (n_in < 0 || (n_in > 0 && !in) || !out) return set_err(BNHIP_E_INVALID, "bad resampler arguments");
    if (n_in == 0 && !flush) return BNHIP_OK;            // empty input: nothing written (resample.go:100-102)
    const long long n_after = r->n_total + n_in;
    const long long i_end = flush ? (n_after * r->L + r->M - 1) / r->M : resampler_ready(r, n_after);
    const long long cnt = i_end - r->i_next;
    // a too-small destination fails before the state advances (resample.go:137-144)
    if (cnt > out_cap || (!flush && bnhip_resampler_estimate(r, n_in) > out_cap))
        return set_err(BNHIP_E_INVALID, "destination buffer too small");
    hipSetDevice(r->device);
    const size_t esz = pcm16 ? 2 : 4;
    const size_t need = (size_t)r->n_hist + (size_t)n_in;
    const int n_work = r->n_hist + n_in;
    // what the next call still needs: the inputs from n0(i_end) - (T-1) on.  Computed up front so that every allocation
    // (including the staging the history compaction moves through) happens BEFORE any work is queued: a failure below
    // leaves n_total / i_next / n_hist / n_base exactly as they were ("fails before the state advances", resample.go:137-144).
    long long keep_from = (i_end * r->M + r->half) / r->L - (r->T - 1);
    if (keep_from < r->n_base) keep_from = r->n_base;
    if (keep_from > n_after) keep_from = n_after;
    const int drop = flush ? 0 : (int)(keep_from - r->n_base), keep = flush ? 0 : n_work - drop;
    if (need > r->work_cap) {
        size_t cap = std::max<size_t>(need * 2, 4096);
        float* nw = nullptr;
        if (hipMalloc((void**)&nw, cap * 4) != hipSuccess) { (void)hipGetLastError(); return set_err(BNHIP_E_NOMEM, "device allocation failed (resampler work buffer)"); }
        hipError_t hc = hipSuccess;
        if (r->n_hist) hc = hipMemcpyAsync(nw, r->d_work, (size_t)r->n_hist * 4, hipMemcpyDeviceToDevice, r->stream);
        if (hc == hipSuccess) hc = hipStreamSynchronize(r->stream);
        if (hc != hipSuccess) { hipFree(nw); return set_err(BNHIP_E_RUNTIME, std::string("resampler: ") + hipGetErrorString(hc)); }
        if (r->d_work) hipFree(r->d_work);
        r->d_work = nw; r->work_cap = cap;
    }
    {   // input staging; doubles as the bounce buffer of the (overlapping) history move, so it is sized for both
        const size_t in_need = std::max((size_t)n_in * esz, drop > 0 && keep > 0 ? (size_t)keep * 4 : (size_t)0);
        if (in_need > r->in_cap) {
            size_t cap = std::max<size_t>(in_need * 2, 8192);
            void* ni = nullptr;
            if (hipMalloc(&ni, cap) != hipSuccess) { (void)hipGetLastError(); return set_err(BNHIP_E_NOMEM, "device allocation failed (resampler input)"); }
            if (r->d_in) hipFree(r->d_in);
            r->d_in = ni; r->in_cap = cap;
        }
    }
    if (cnt > 0 && (size_t)cnt * esz > r->out_cap) {
        size_t cap = std::max<size_t>((size_t)cnt * esz * 2, 8192);
        void* no = nullptr;
        if (hipMalloc(&no, cap) != hipSuccess) { (void)hipGetLastError(); return set_err(BNHIP_E_NOMEM, "device allocation failed (resampler output)"); }
        if (r->d_out) hipFree(r->d_out);
        r->d_out = no; r->out_cap = cap;
    }
    hipError_t he = hipSuccess;
    if (n_in > 0) {
        if (pcm16) {
            he = hipMemcpyAsync(r->d_in, in, (size_t)n_in * 2, hipMemcpyHostToDevice, r->stream);
            if (he == hipSuccess) launch_pcm_to_f32(r->d_in, 16, r->d_work + r->n_hist, (size_t)n_in, r->stream);   // float32(int16)/32768, resample.go:120-124
        } else {
            he = hipMemcpyAsync(r->d_work + r->n_hist, in, (size_t)n_in * 4, hipMemcpyHostToDevice, r->stream);
        }
    }
    if (he == hipSuccess && cnt > 0) {
        int lrc = launch_resample(r->d_work, r->d_out, r->d_table, 0, pcm16 ? 1 : 0, 1, n_work, (int)cnt, r->L, r->M, r->T, r->half,
                                  r->i_next, r->n_base, r->stream);
        if (lrc) { hipStreamSynchronize(r->stream); return set_err(BNHIP_E_UNSUPPORTED, "resample ratio needs a phase table larger than LDS"); }
        he = hipMemcpyAsync(out, r->d_out, (size_t)cnt * esz, hipMemcpyDeviceToHost, r->stream);
    }
    // history compaction (an overlapping move inside one buffer, bounced through the now idle input staging), queued behind
    // the resample kernel that still reads the old layout
    if (he == hipSuccess && drop > 0 && keep > 0) {
        he = hipMemcpyAsync(r->d_in, r->d_work + drop, (size_t)keep * 4, hipMemcpyDeviceToDevice, r->stream);
        if (he == hipSuccess) he = hipMemcpyAsync(r->d_work, r->d_in, (size_t)keep * 4, hipMemcpyDeviceToDevice, r->stream);
    }
    if (he == hipSuccess) he = hipStreamSynchronize(r->stream);
    if (he != hipSuccess) { (void)hipGetLastError(); return set_err(BNHIP_E_RUNTIME, std::string("resampler: ") + hipGetErrorString(he)); }
    // ---- commit: everything above succeeded
    if (n_out) *n_out = (int)cnt;
    if (flush) {                                          // back to the initial state: the next call starts a new stream
        r->n_total = 0; r->i_next = 0; r->n_base = 0; r->n_hist = 0;
        return BNHIP_OK;
    }
    r->n_total = n_after; r->i_next = i_end;
    r->n_hist = keep > 0 ? keep : 0;
    r->n_base = keep_from;
    return BNHIP_OK;
}

int bnhip_resampler_process_pcm16(bnhip_resampler* r, const int16_t* in, int n_in, int16_t* out, int out_cap, int* n_out) {
    BN_GUARD_BEGIN
    return resampler_run(r, in, true, n_in, out, out_cap, n_out, 0);
    BN_GUARD_END((void)0)
}
int bnhip_resampler_process_f32(bnhip_resampler* r, const float* in, int n_in, float* out, int out_cap, int* n_out) {
    BN_GUARD_BEGIN
    return resampler_run(r, in, false, n_in, out, out_cap, n_out, 0);
    BN_GUARD_END((void)0)
}
int bnhip_resampler_flush_pcm16(bnhip_resampler* r, int16_t* out, int out_cap, int* n_out) {
    BN_GUARD_BEGIN
    return resampler_run(r, nullptr, true, 0, out, out_cap, n_out, 1);
    BN_GUARD_END((void)0)
}
int bnhip_resampler_flush_f32(bnhip_resampler* r, float* out, int out_cap, int* n_out) {
    BN_GUARD_BEGIN
    return resampler_run(r, nullptr, false, 0, out, out_cap, n_out, 1);
    BN_GUARD_END((void)0)
}
void bnhip_resampler_destroy(bnhip_resampler* r) {
    try { resampler_free(r); } catch (...) {}
}

// ------------------------------------------------------------------------------------------------ diagnostics
int bnhip_profile_enable(bnhip_model* m, int on) {
    if (!m) return set_err(BNHIP_E_INVALID, "model is NULL");
    m->eng().profiling = on != 0;
    return BNHIP_OK;
}

int bnhip_profile_filter(bnhip_model* m, const char* kernel_class) {
    if (!m) return set_err(BNHIP_E_INVALID, "model is NULL");
    BN_GUARD_BEGIN
    m->eng().profile_filter = kernel_class ? kernel_class : "";
    return BNHIP_OK;
    BN_GUARD_END((void)0)
}

int bnhip_profile_read(bnhip_model* m, char* buf, size_t cap) {
    if (!m || m->eng().device < 0) return set_err(BNHIP_E_INVALID, "model is NULL or plan-only");
    BN_GUARD_BEGIN
    hipSetDevice(m->eng().device);
    return copy_out(m->eng().profile_read(), buf, cap);
    BN_GUARD_END((void)0)
}

int bnhip_profile_steps(bnhip_model* m, int on) {
    if (!m) return set_err(BNHIP_E_INVALID, "model is NULL");
    m->eng().step_timing = on != 0;
    return BNHIP_OK;
}

int bnhip_profile_steps_read(bnhip_model* m, double* start_ms, double* end_ms, int cap) {
    if (!m || m->eng().device < 0) return set_err(BNHIP_E_INVALID, "model is NULL or plan-only");
    if (cap < 0) return set_err(BNHIP_E_INVALID, "negative capacity");
    BN_GUARD_BEGIN
    hipSetDevice(m->eng().device);
    return m->eng().steps_read(start_ms, end_ms, cap);
    BN_GUARD_END((void)0)
}

int bnhip_model_describe(const bnhip_model* m, char* buf, size_t cap) {
    if (!m) return set_err(BNHIP_E_INVALID, "model is NULL");
    BN_GUARD_BEGIN
    std::string d = m->eng().describe();
    // splice the handle-level facts in front of the engine's description
    std::string devs = "[";
    for (size_t i = 0; i < m->engs.size(); i++) devs += (i ? "," : "") + std::to_string(m->engs[i]->device);
    devs += "]";
    // (what every engine of the handle runs: one tuning adopted by all of them, or their own - "tune_sources"; "plans_identical":
    // every engine picked the same tile / kernel form for every step, so a clip's bits do not depend on the shard it lands on)
    std::string srcs = "[";
    bool same = true;
    for (size_t i = 0; i < m->engs.size(); i++) {
        srcs += std::string(i ? "," : "") + "\"" + m->engs[i]->tune_source + "\"";
        same = same && m->engs[i]->tuning_text() == m->engs[0]->tuning_text();
    }
    srcs += "]";
    std::string head = "{\"devices\":" + devs + ",\"weight_replication\":\"" + m->replication + "\",\"tune_sources\":" + srcs +
                       ",\"plans_identical\":" + (same ? "true" : "false") + ",";
    if (!d.empty() && d[0] == '{') d = head + d.substr(1);
    return copy_out(d, buf, cap);
    BN_GUARD_END((void)0)
}

}  // extern "C"
