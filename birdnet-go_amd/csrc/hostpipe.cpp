// Host-pointer pipeline: see hostpipe.h.  Replaces the reference's blocking backend call for batches
// (internal/inference/onnx/classifier.go:372-430 PredictBatch; internal/analysis/process.go:280-295 for the ownership rule:
// the caller's slice is only read before the entry returns).
#include "hostpipe.h"

#include <cstdint>

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/bnhip.h"
#include "engine.h"
#include "numa.h"

namespace bnhip {

// ------------------------------------------------------------------------------------------------ copy pool
// Pageable caller memory -> pinned staging.  One thread moves ~10 GB/s; a 256-clip int16 chunk is 74 MB and has to be
// staged in well under the 3.6 ms the GPU needs for it, so the copy is cut into 2 MB pieces served by a few threads (the
// calling thread helps).  One pool per NUMA node that has a GPU (round 6): its threads are bound to that node's CPUs, so the
// staging pass of a GPU's chunks reads the caller's pages and writes the pinned slot from the socket the GPU hangs off; engines
// on GPUs of one node (and every worker thread of a multi-device handle that serves them) share the node's pool.  A box
// without NUMA information has one unbound pool, as before.
namespace {

// `remaining` is only touched under `mu`: the waiter may own the Ticket on its stack (parallel_copy), so the last worker must
// be done with the mutex and the condition variable before the waiter can observe zero and pop the frame.
struct Ticket {
    int remaining = 0;
    std::mutex mu;
    std::condition_variable cv;
};

class CopyPool {
  public:
    struct Task { char* dst; const char* src; size_t n; Ticket* t; };
    explicit CopyPool(const std::vector<int>& cpus = {}) : cpus_(cpus) {
        int n = 0;
        if (const char* e = getenv("BNHIP_COPY_THREADS")) n = atoi(e);
        else {
            unsigned hw = std::thread::hardware_concurrency();
            n = (int)std::min(8u, std::max(1u, hw / 4));
        }
        n = std::max(0, std::min(n, 64));
        if (!cpus_.empty()) n = std::min<int>(n, (int)cpus_.size());       // (never more threads than CPUs they may run on)
        for (int i = 0; i < n; i++) th_.emplace_back([this] { bound_ += bind_this_thread(cpus_) ? 1 : 0; loop(); });
    }
    ~CopyPool() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto& t : th_) if (t.joinable()) t.join();
    }
    int threads() const { return (int)th_.size(); }
    int bound_threads() const { return bound_.load(); }
    int cpus() const { return (int)cpus_.size(); }
    // enqueue the copy; the ticket reaches zero when every piece has landed
    void submit(void* dst, const void* src, size_t bytes, Ticket* t) {
        const size_t piece = 2u << 20;
        const int np = (int)std::max<size_t>(1, (bytes + piece - 1) / piece);
        { std::lock_guard<std::mutex> lk(t->mu); t->remaining = np; }
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (int i = 0; i < np; i++) {
                size_t off = (size_t)i * piece, n = std::min(piece, bytes - off);
                q_.push_back(Task{(char*)dst + off, (const char*)src + off, n, t});
            }
        }
        cv_.notify_all();
    }
    // the caller works on queued pieces (its own or anybody's) until its ticket is done
    void wait(Ticket* t) {
        for (;;) {
            { std::lock_guard<std::mutex> lk(t->mu); if (t->remaining == 0) return; }
            Task k;
            bool got = false;
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (!q_.empty()) { k = q_.front(); q_.pop_front(); got = true; }
            }
            if (got) { run(k); continue; }
            std::unique_lock<std::mutex> lk(t->mu);
            t->cv.wait(lk, [t] { return t->remaining == 0; });
            return;
        }
    }

  private:
    static void run(const Task& k) {
        if (k.n) memcpy(k.dst, k.src, k.n);
        std::lock_guard<std::mutex> lk(k.t->mu);
        if (--k.t->remaining == 0) k.t->cv.notify_all();
    }
    void loop() {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            cv_.wait(lk, [this] { return stop_ || !q_.empty(); });
            if (stop_) return;
            Task k = q_.front(); q_.pop_front();
            lk.unlock();
            run(k);
            lk.lock();
        }
    }
    const std::vector<int> cpus_;
    std::atomic<int> bound_{0};
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Task> q_;
    bool stop_ = false;
};

// BNHIP_NUMA=0 turns the placement off (one unbound pool, default page placement): the A/B switch of the records under profiles/
bool numa_on() {
    static const bool on = !(getenv("BNHIP_NUMA") && atoi(getenv("BNHIP_NUMA")) == 0);
    return on;
}

std::mutex g_pool_mu;
std::map<int, CopyPool*> g_pools;            // by NUMA node; -1 = unbound.  Never destroyed: a Go host exits without static
                                             // destructors anyway, and a destructor joining threads at dlclose time can deadlock
                                             // under a loader lock
CopyPool& pool_for_node(int node) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto it = g_pools.find(node);
    if (it != g_pools.end()) return *it->second;          // (every later call: no sysfs, no allocation)
    std::vector<int> cpus;
    if (node >= 0) cpus = numa_usable_cpus(node);
    if (node >= 0 && cpus.empty()) {                      // the node's CPUs are outside this process' cpuset: the unbound pool serves it
        auto un = g_pools.find(-1);
        CopyPool* p = un != g_pools.end() ? un->second : (g_pools[-1] = new CopyPool());
        g_pools[node] = p;
        return *p;
    }
    CopyPool* p = new CopyPool(cpus);
    g_pools[node] = p;
    return *p;
}

std::mutex g_devnode_mu;
std::map<int, int> g_devnode;
}  // namespace

// NUMA node of a HIP device (from its PCI address), -1 when unknown or placement is off; cached
int device_numa_node(int device) {
    if (device < 0 || !numa_on()) return -1;
    std::lock_guard<std::mutex> lk(g_devnode_mu);
    auto it = g_devnode.find(device);
    if (it != g_devnode.end()) return it->second;
    char bdf[64] = {0};
    int node = -1;
    if (hipDeviceGetPCIBusId(bdf, sizeof bdf, device) == hipSuccess) node = pci_numa_node(bdf);
    else (void)hipGetLastError();
    g_devnode[device] = node;
    return node;
}

namespace {
CopyPool& pool_for_device(int device) { return pool_for_node(device_numa_node(device)); }
}  // namespace

void parallel_copy(void* dst, const void* src, size_t bytes, int device) {
    if (bytes < (4u << 20)) { if (bytes) memcpy(dst, src, bytes); return; }
    Ticket t;
    CopyPool& p = pool_for_device(device);
    p.submit(dst, src, bytes, &t);
    p.wait(&t);
}
int copy_pool_threads(int device) { return pool_for_device(device).threads(); }
void copy_pool_info(int device, int* node, int* threads, int* bound, int* cpus) {
    CopyPool& p = pool_for_device(device);
    if (node) *node = device_numa_node(device);
    if (threads) *threads = p.threads();
    if (bound) *bound = p.bound_threads();
    if (cpus) *cpus = p.cpus();
}

// ------------------------------------------------------------------------------------------------ staging ring
struct HostPipe {
    static constexpr int K = 4;              // slots: chunk c may be staged while c-1, c-2 are in flight and c-3 drains
    struct Slot {
        char* h_in = nullptr; size_t h_in_cap = 0;         // pinned, raw caller bytes of one chunk
        char* d_raw = nullptr;                             // device PCM bytes (only for the PCM entries)
        float* d_in = nullptr;                             // device float32 [max_batch, n_samples]
        float* d_logits = nullptr; float* h_logits = nullptr;
        float* d_emb = nullptr; float* h_emb = nullptr;
        float* d_conf = nullptr;                           // [max_batch, n_classes] activation output (top-k jobs)
        float* d_tkc = nullptr; int32_t* d_tki = nullptr; float* h_tkc = nullptr; int32_t* h_tki = nullptr; int tk_cap = 0;
        hipEvent_t ev_h2d = nullptr, ev_comp = nullptr, ev_done = nullptr, ev_out = nullptr;   // input landed / kernels done / results in pinned memory
        Ticket fill;
        int chunk = -1;                                    // chunk index in flight, -1 = idle
    };
    Slot s[K];
    hipStream_t xfer = nullptr;              // the one copy stream, both directions (see ensure_pipe)
    int numa_node = -1;                      // the device's NUMA node: where the pinned slots live and the copy threads run
    CopyPool* cp = nullptr;                  // that node's pool
};

void hostpipe_free(HostPipe* hp) {
    if (!hp) return;
    for (auto& s : hp->s) {
        for (void* p : {(void*)s.d_raw, (void*)s.d_in, (void*)s.d_logits, (void*)s.d_emb, (void*)s.d_conf, (void*)s.d_tkc, (void*)s.d_tki})
            if (p) hipFree(p);
        for (void* p : {(void*)s.h_in, (void*)s.h_logits, (void*)s.h_emb, (void*)s.h_tkc, (void*)s.h_tki})
            if (p) hipHostFree(p);
        if (s.ev_h2d) hipEventDestroy(s.ev_h2d);
        if (s.ev_done) hipEventDestroy(s.ev_done);
        if (s.ev_comp) hipEventDestroy(s.ev_comp);
        if (s.ev_out) hipEventDestroy(s.ev_out);
    }
    if (hp->xfer) hipStreamSynchronize(hp->xfer);       // (owned by the device's stream pool)
    delete hp;
}

namespace {

#define HP_TRY(call, what)                                                                 \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            (void)hipGetLastError();                                                       \
            err = std::string(what) + ": " + hipGetErrorString(e_);                        \
            return e_ == hipErrorOutOfMemory ? BNHIP_E_NOMEM : BNHIP_E_RUNTIME;            \
        }                                                                                  \
    } while (0)

// everything a job of this shape needs in every slot; idempotent, grows only
int ensure_pipe(Engine& e, const HostJob& j, size_t chunk_bytes, std::string& err) {
    if (!e.hostpipe) e.hostpipe = new HostPipe();
    HostPipe& hp = *e.hostpipe;
    // Streams are a scarce resource: HIP maps ALL streams of a process on a device onto at most four hardware queues
    // (GPU_MAX_HW_QUEUES; measured on ROCm 7.2: a fifth active stream lands on a queue another one uses, whatever its
    // priority), and two streams on one queue serialise - a copy marker behind a context's kernels stalls the next chunk,
    // two contexts on one queue stop overlapping altogether (both seen in BNHIP_HOST_TRACE timelines).  So the pipeline
    // runs on exactly three, all owned by the DEVICE's pool (engine.cpp: engines on one GPU share them): the two kernel
    // streams (main + lane stream, idle during a pipelined call) as the two contexts, and ONE copy stream that carries both
    // directions in an order that never blocks a prefetch: ... H2D(c+1), D2H(c-2), H2D(c+2), D2H(c-1) ... (a copy-out waits
    // for its chunk's kernels, which are done long before the input issued behind it is needed).
    hp.xfer = e.copy_stream();
    if (!hp.xfer) { err = "copy stream creation failed"; return BNHIP_E_RUNTIME; }
    if (!hp.cp) { hp.numa_node = device_numa_node(e.device); hp.cp = &pool_for_node(hp.numa_node); }
    // pinned slots come from the GPU's own NUMA node when it has room (the preference lasts for the allocation only: a steady-state
    // call allocates nothing and touches no policy)
    const int node = hp.numa_node;
    auto pinned = [node](void** p, size_t bytes) { NumaPrefer near_gpu(node); return hipHostMalloc(p, bytes, hipHostMallocDefault); };
    const size_t mb = (size_t)e.max_batch;
    for (auto& s : hp.s) {
        if (!s.ev_h2d) HP_TRY(hipEventCreateWithFlags(&s.ev_h2d, hipEventDisableTiming), "event");
        if (!s.ev_done) HP_TRY(hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming), "event");
        if (!s.ev_comp) HP_TRY(hipEventCreateWithFlags(&s.ev_comp, hipEventDisableTiming), "event");
        if (!s.ev_out) HP_TRY(hipEventCreateWithFlags(&s.ev_out, hipEventDisableTiming), "event");
        if (s.h_in_cap < chunk_bytes) {
            if (s.h_in) { hipHostFree(s.h_in); s.h_in = nullptr; s.h_in_cap = 0; }
            HP_TRY(pinned((void**)&s.h_in, chunk_bytes), "pinned staging allocation");
            s.h_in_cap = chunk_bytes;
        }
        if (!s.d_in) HP_TRY(hipMalloc((void**)&s.d_in, mb * e.n_samples * 4), "device staging allocation");
        if (j.pcm_bits && !s.d_raw) HP_TRY(hipMalloc((void**)&s.d_raw, mb * e.n_samples * 4), "device PCM staging allocation");
        if (!s.d_logits) HP_TRY(hipMalloc((void**)&s.d_logits, mb * e.n_classes * 4), "device logits staging allocation");
        if (j.logits && !s.h_logits) HP_TRY(pinned((void**)&s.h_logits, mb * e.n_classes * 4), "pinned logits staging allocation");
        if (j.emb && !s.d_emb) HP_TRY(hipMalloc((void**)&s.d_emb, mb * e.emb_dim * 4), "device embedding staging allocation");
        if (j.emb && !s.h_emb) HP_TRY(pinned((void**)&s.h_emb, mb * e.emb_dim * 4), "pinned embedding staging allocation");
        if (j.topk > 0) {
            if (!s.d_conf) HP_TRY(hipMalloc((void**)&s.d_conf, mb * e.n_classes * 4), "device confidence allocation");
            if (s.tk_cap < j.topk) {
                for (void* p : {(void*)s.d_tkc, (void*)s.d_tki}) if (p) hipFree(p);
                for (void* p : {(void*)s.h_tkc, (void*)s.h_tki}) if (p) hipHostFree(p);
                s.d_tkc = nullptr; s.d_tki = nullptr; s.h_tkc = nullptr; s.h_tki = nullptr; s.tk_cap = 0;
                HP_TRY(hipMalloc((void**)&s.d_tkc, mb * j.topk * 4), "device top-k allocation");
                HP_TRY(hipMalloc((void**)&s.d_tki, mb * j.topk * 4), "device top-k allocation");
                HP_TRY(pinned((void**)&s.h_tkc, mb * j.topk * 4), "pinned top-k allocation");
                HP_TRY(pinned((void**)&s.h_tki, mb * j.topk * 4), "pinned top-k allocation");
                s.tk_cap = j.topk;
            }
        }
    }
    return BNHIP_OK;
}

// Is [p, p + bytes) page-locked host memory the runtime knows (bnhip_host_alloc / hipHostMalloc / hipHostRegister)?  Then the
// copy engines can read / write it directly and the staging pass through the pinned slots is skipped: the reference's own
// accelerator shim keeps a C-allocated input buffer for the same reason (backend_openvino.go:673-680).
static std::atomic<long> g_pinned_inputs{0};      // diagnostics: pipelined calls whose input was recognised as page-locked (a test asserts the path is taken)
// The whole range must lie inside ONE allocation: two pinned allocations with pageable pages between them would pass a probe of
// the first and last byte.  hipMemGetAddressRange gives the allocation's base and size where the runtime answers it
// for host allocations; where it does not, the two probes must at least map to device addresses exactly bytes - 1 apart with the
// same owner and flags - anything else takes the staging path, which is always correct.
bool is_pinned(const void* p, size_t bytes) {
    if (!p || !bytes) return false;
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (a.type != hipMemoryTypeHost) return false;
    void* base = nullptr; size_t size = 0;
    if (hipMemGetAddressRange(reinterpret_cast<hipDeviceptr_t*>(&base), &size, const_cast<void*>(p)) == hipSuccess && base && size) {
        const uintptr_t lo = reinterpret_cast<uintptr_t>(base), q = reinterpret_cast<uintptr_t>(p);
        return q >= lo && bytes <= size && q - lo <= size - bytes;
    }
    (void)hipGetLastError();
    hipPointerAttribute_t b{};
    if (hipPointerGetAttributes(&b, (const char*)p + bytes - 1) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (b.type != hipMemoryTypeHost || b.device != a.device || b.allocationFlags != a.allocationFlags) return false;
    if (!a.devicePointer || !b.devicePointer) return false;
    return reinterpret_cast<uintptr_t>(b.devicePointer) - reinterpret_cast<uintptr_t>(a.devicePointer) == bytes - 1;
}

int ensure_small_topk(Engine& e, int k, std::string& err) {
    if (k <= e.topk_cap) return BNHIP_OK;
    if (e.d_topk_conf) hipFree(e.d_topk_conf);
    if (e.d_topk_idx) hipFree(e.d_topk_idx);
    e.d_topk_conf = nullptr; e.d_topk_idx = nullptr; e.topk_cap = 0;
    HP_TRY(hipMalloc((void**)&e.d_topk_conf, (size_t)e.max_batch * k * 4), "device top-k allocation");
    HP_TRY(hipMalloc((void**)&e.d_topk_idx, (size_t)e.max_batch * k * 4), "device top-k allocation");
    e.topk_cap = k;
    return BNHIP_OK;
}

// Calls below the pipelining threshold (the product's own pattern is ONE clip per Predict, analyze.go:60): straight
// through the engine's staging buffers on its main stream; the plan's two lanes split batches of >= 32 clips.
int small_run(Engine& e, const HostJob& j, std::string& err) {
    const size_t bps = j.pcm_bits ? (size_t)j.pcm_bits / 8 : 4;
    const int kk = j.topk > 0 ? std::min(j.topk, e.n_classes) : 0;
    if (kk) { int rc = ensure_small_topk(e, kk, err); if (rc) return rc; }
    if (j.pcm_bits) {
        const size_t need = (size_t)e.max_batch * e.n_samples * bps;
        if (e.stage_pcm_bytes < need) {
            if (e.d_stage_pcm) { hipStreamSynchronize(e.stream); hipFree(e.d_stage_pcm); e.d_stage_pcm = nullptr; e.stage_pcm_bytes = 0; }
            HP_TRY(hipMalloc((void**)&e.d_stage_pcm, need), "device PCM staging allocation");
            e.stage_pcm_bytes = need;
        }
    }
    for (int off = 0; off < j.n_clips; off += e.max_batch) {
        const int n = std::min(e.max_batch, j.n_clips - off);
        const size_t cnt = (size_t)n * e.n_samples;
        const char* src = (const char*)j.src + (size_t)off * e.n_samples * bps;
        if (j.prepare) j.prepare(off, n);
        if (j.pcm_bits) {
            HP_TRY(hipMemcpyAsync(e.d_stage_pcm, src, cnt * bps, hipMemcpyHostToDevice, e.stream), "H2D copy");
            launch_pcm_to_f32(e.d_stage_pcm, j.pcm_bits, e.d_stage_in, cnt, e.stream);
        } else {
            HP_TRY(hipMemcpyAsync(e.d_stage_in, src, cnt * 4, hipMemcpyHostToDevice, e.stream), "H2D copy");
        }
        if (!e.run(e.d_stage_in, n, e.d_stage_logits, j.emb ? e.d_stage_emb : nullptr, &err)) { hipStreamSynchronize(e.stream); return BNHIP_E_RUNTIME; }
        if (kk) {
            launch_activation(e.d_stage_logits, e.d_post_conf, n, e.n_classes, j.activation, j.sensitivity, e.stream);
            launch_topk(e.d_post_conf, n, e.n_classes, kk, e.d_topk_conf, e.d_topk_idx, e.stream);
            HP_TRY(hipMemcpyAsync(j.out_conf + (size_t)off * kk, e.d_topk_conf, (size_t)n * kk * 4, hipMemcpyDeviceToHost, e.stream), "D2H copy");
            HP_TRY(hipMemcpyAsync(j.out_idx + (size_t)off * kk, e.d_topk_idx, (size_t)n * kk * 4, hipMemcpyDeviceToHost, e.stream), "D2H copy");
        }
        if (j.logits)
            HP_TRY(hipMemcpyAsync(j.logits + (size_t)off * e.n_classes, e.d_stage_logits, (size_t)n * e.n_classes * 4, hipMemcpyDeviceToHost, e.stream), "D2H copy");
        if (j.emb)
            HP_TRY(hipMemcpyAsync(j.emb + (size_t)off * e.emb_dim, e.d_stage_emb, (size_t)n * e.emb_dim * 4, hipMemcpyDeviceToHost, e.stream), "D2H copy");
        HP_TRY(hipStreamSynchronize(e.stream), "synchronize");
    }
    return BNHIP_OK;
}


// Diagnostic switches of the host pipeline (BNHIP_HOST_SERIAL / _NOSPLIT / _PLAN / _CHUNKS / _SCHED) are read per CALL so that one
// process can compare schedules - but only in a process that started with BNHIP_HOST_DIAG set: a production call never touches the
// environment (getenv beside another thread's setenv is a data race; multi-device handles run calls on worker threads).
static const char* diag_env(const char* name) {
    static const bool on = getenv("BNHIP_HOST_DIAG") != nullptr;
    // (a diagnostic switch set WITHOUT BNHIP_HOST_DIAG used to be ignored silently - an A/B script then measured the
    // default path twice.  Say so once, at the first call, from the state of the environment at process start.)
    static const bool warned = [] {
        if (on) return false;
        bool any = false;
        for (const char* n : {"BNHIP_HOST_SERIAL", "BNHIP_HOST_NOSPLIT", "BNHIP_HOST_PLAN", "BNHIP_HOST_CHUNKS", "BNHIP_HOST_SCHED"})
            if (getenv(n)) { fprintf(stderr, "[bnhip] %s is set but BNHIP_HOST_DIAG is not: the host pipeline's diagnostic switches are ignored\n", n); any = true; }
        return any;
    }();
    (void)warned;
    return on ? getenv(name) : nullptr;
}

// BNHIP_HOST_TRACE events: created checked, destroyed on every way out of the call (leaked on the early returns)
struct TraceEvents {
    std::vector<hipEvent_t> ev;
    bool create(size_t n) {
        ev.assign(n, nullptr);
        for (auto& e : ev) if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); e = nullptr; destroy(); return false; }
        return true;
    }
    void destroy() { for (auto& e : ev) if (e) hipEventDestroy(e); ev.clear(); }
    hipEvent_t& operator[](size_t i) { return ev[i]; }
    bool on() const { return !ev.empty(); }
    ~TraceEvents() { destroy(); }
};

// ------------------------------------------------------------------------------------------------ two-phase call
// A blocking call that fits one batch starts on an idle GPU and ends on one: cut into whole-plan chunks, its first chunk runs alone,
// its last one runs alone, and every chunk runs the late layers of the stack with a quarter of the rows they need to fill the chip
// (a plan tuned for the chunk size does not change that: measured).  Here the PLAN is cut as well (Engine::pick_split):
// the front of the plan (front-end, early blocks: plenty of rows per clip) runs chunk by chunk on stream A as the copies land and
// leaves the one activation that crosses the cut in hand-off memory; the back of the plan runs on stream B over GROUPS of chunks -
// the first group under the later chunks' copies and front halves, the last one alone but over half the call's rows.
// Same kernels and per-clip arithmetic as any other cut: results are bit-identical to the serial path.
static std::atomic<long> g_split_calls{0};
int host_run_split(Engine& e, const HostJob& j, const std::vector<int>& csize, std::string& err) {
    const int nch = (int)csize.size();
    constexpr int K = HostPipe::K;
    std::vector<int> cfirst(nch + 1, 0);
    for (int c = 0; c < nch; c++) cfirst[c + 1] = cfirst[c] + csize[c];
    const size_t bps = j.pcm_bits ? (size_t)j.pcm_bits / 8 : 4;
    const size_t clip_bytes = (size_t)e.n_samples * bps;
    const int kk = j.topk > 0 ? std::min(j.topk, e.n_classes) : 0;
    HostJob jj = j; jj.topk = kk;
    int rc = ensure_pipe(e, jj, (size_t)e.max_batch * clip_bytes, err);
    if (rc) return rc;
    if (!e.ensure_contexts(2, &err, false) || !e.ensure_hand(&err)) {
        if (err.empty()) err = "two-phase host call: context arena / hand-off buffer allocation failed";
        return BNHIP_E_NOMEM;
    }
    HostPipe& hp = *e.hostpipe;
    hipStream_t sa = e.kernel_stream(0), sb = e.kernel_stream(1);
    if (!sa || !sb) { err = "hipStreamCreate failed"; return BNHIP_E_RUNTIME; }
    e.sync_contexts();
    if (e.stream) hipStreamSynchronize(e.stream);
    for (int c = 0; c < Engine::kMaxDepth; c++) if (e.ctx_stream[c]) hipStreamSynchronize(e.ctx_stream[c]);
    g_split_calls.fetch_add(1, std::memory_order_relaxed);

    const bool src_pinned = is_pinned(j.src, (size_t)j.n_clips * clip_bytes);
    if (src_pinned) g_pinned_inputs.fetch_add(1, std::memory_order_relaxed);
    const bool logits_pinned = j.logits && is_pinned(j.logits, (size_t)j.n_clips * e.n_classes * 4);
    const bool emb_pinned = j.emb && is_pinned(j.emb, (size_t)j.n_clips * e.emb_dim * 4);
    auto abort_all = [&]() {
        for (auto& s : hp.s) hp.cp->wait(&s.fill);
        hipStreamSynchronize(hp.xfer);
        for (int q = 0; q < 3; q++) if (e.kstream[q]) hipStreamSynchronize(e.kstream[q]);
        hipStreamSynchronize(hp.xfer);
        (void)hipGetLastError();
    };
#define HP_PIPE(call, what)                                                                \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            abort_all();                                                                   \
            err = std::string(what) + ": " + hipGetErrorString(e_);                        \
            return BNHIP_E_RUNTIME;                                                        \
        }                                                                                  \
    } while (0)
    // The schedule: fronts ('f') in chunk order and backs ('b'), each with the kernel stream it runs on (= the context arena it
    // uses: launches that may overlap never share an arena); a back covers the chunks whose fronts were queued since the previous
    // back and waits for exactly those.
    struct Op { char kind; int st; };
    std::vector<Op> ops;
    {
        std::string plan;
        if (const char* pe = diag_env("BNHIP_HOST_PLAN")) plan = pe;      // experiments: "f0 f0 b1 f0 f0 b1" (read per call)
        int nf = 0; bool ok = !plan.empty(), open = false;
        for (size_t i = 0; ok && i < plan.size(); i++) {
            if (plan[i] == ' ' || plan[i] == ',') continue;
            if ((plan[i] != 'f' && plan[i] != 'b') || i + 1 >= plan.size() || plan[i + 1] < '0' || plan[i + 1] > '2') { ok = false; break; }
            if (plan[i] == 'f') { nf++; open = true; } else { if (!open) ok = false; open = false; }
            ops.push_back(Op{plan[i], plan[i + 1] - '0'});
            i++;
        }
        int nb = 0; for (const Op& o : ops) nb += o.kind == 'b';
        if (!ok || nf != nch || open || nb > K) {
            // default: every front on stream 0; the back of all chunks but the last on stream 1 as soon as their fronts are queued, the
            // last chunk's back behind its front on stream 0 - the call ends with two backs overlapping instead of one running alone
            // (measured against even halves, three groups, fronts alternating over two streams and uneven chunks: profiles/r05_host_split.txt)
            ops.clear();
            for (int c = 0; c < nch; c++) {
                ops.push_back(Op{'f', 0});
                if (c + 2 == nch) ops.push_back(Op{'b', 1});
                if (c + 1 == nch) ops.push_back(Op{'b', 0});
            }
        }
    }
    int n_streams = 1;
    for (const Op& o : ops) n_streams = std::max(n_streams, o.st + 1);
    if (!e.ensure_contexts(n_streams, &err, false)) { if (err.empty()) err = "two-phase host call: context arena allocation failed"; return BNHIP_E_NOMEM; }
    hipStream_t ks[3] = {sa, sb, n_streams > 2 ? e.kernel_stream(2) : nullptr};
    if (n_streams > 2 && !ks[2]) { err = "hipStreamCreate failed"; return BNHIP_E_RUNTIME; }
    int ng = 0; for (const Op& o : ops) ng += o.kind == 'b';
    std::vector<int> gfirst, gcount;                      // per group: first clip, clips
    // A producer behind page-locked memory (bnhip_windows_predict_topk: rows assembled on demand): the call's FIRST chunk is on the
    // critical path - nothing runs on the GPU until it has landed - so it is produced and copied in quarters, the DMA of one quarter
    // under the assembly of the next (round 6: first kernel at 0.65 ms instead of 0.95).  Later chunks are produced whole, under the
    // previous chunk's kernels.
    const bool piecewise = j.prepare && src_pinned;
    auto start_fill = [&](int c) {
        HostPipe::Slot& s = hp.s[c % K];
        if (j.prepare && !(piecewise && c == 0)) j.prepare(cfirst[c], csize[c]);
        if (!src_pinned) hp.cp->submit(s.h_in, (const char*)j.src + (size_t)cfirst[c] * clip_bytes, (size_t)csize[c] * clip_bytes, &s.fill);
    };
    static const bool trace_env = getenv("BNHIP_HOST_TRACE") != nullptr;
    TraceEvents tev;                                      // trace: [base][per chunk: h2d done, front start, front done][per group: back start, back done]
    const bool trace = trace_env && tev.create(1 + 3 * (size_t)nch + 2 * (size_t)ng);
    if (trace) hipEventRecord(tev[0], hp.xfer);
    char* hand = reinterpret_cast<char*>(e.d_hand);
    const size_t hcb = e.hand_clip_bytes();
    const int ns = (int)e.steps.size();
    start_fill(0);
    int c = 0, g = 0, gopen = 0;                          // next chunk, next group, first chunk of the open group
    for (const Op& op : ops) {
        hipStream_t st = ks[op.st];
        if (op.kind == 'f') {
            HostPipe::Slot& s = hp.s[c % K];
            const int n = csize[c];
            const size_t cnt = (size_t)n * e.n_samples;
            if (!src_pinned) hp.cp->wait(&s.fill);
            const void* h_src = src_pinned ? (const void*)((const char*)j.src + (size_t)cfirst[c] * clip_bytes) : (const void*)s.h_in;
            char* d_dst = j.pcm_bits ? (char*)s.d_raw : (char*)s.d_in;
            if (piecewise && c == 0 && n >= 16) {
                const int pc = (n + 3) / 4;
                for (int q = 0; q < n; q += pc) {
                    const int m = std::min(pc, n - q);
                    j.prepare(cfirst[0] + q, m);
                    HP_PIPE(hipMemcpyAsync(d_dst + (size_t)q * clip_bytes, (const char*)h_src + (size_t)q * clip_bytes, (size_t)m * clip_bytes, hipMemcpyHostToDevice, hp.xfer), "H2D copy");
                }
            } else {
                if (piecewise && c == 0) j.prepare(cfirst[0], n);
                HP_PIPE(hipMemcpyAsync(d_dst, h_src, cnt * bps, hipMemcpyHostToDevice, hp.xfer), "H2D copy");
            }
            HP_PIPE(hipEventRecord(s.ev_h2d, hp.xfer), "event record");
            if (trace) hipEventRecord(tev[1 + 3 * c], hp.xfer);
            HP_PIPE(hipStreamWaitEvent(st, s.ev_h2d, 0), "stream wait");
            if (trace) hipEventRecord(tev[2 + 3 * c], st);
            if (j.pcm_bits) launch_pcm_to_f32(s.d_raw, j.pcm_bits, s.d_in, cnt, st);
            if (!e.run_part(op.st, st, 0, e.split_step, s.d_in, n, reinterpret_cast<float*>(hand + (size_t)cfirst[c] * hcb), s.d_logits, nullptr, &err)) { abort_all(); return BNHIP_E_RUNTIME; }
            HP_PIPE(hipEventRecord(s.ev_comp, st), "event record");      // (front of chunk c done)
            if (trace) hipEventRecord(tev[3 + 3 * c], st);
            c++;
            if (c < nch) start_fill(c);
        } else {
            HostPipe::Slot& o = hp.s[g];                              // (output side of slot g: logits / top-k / embedding staging)
            const int first = cfirst[gopen], gn = cfirst[c] - first;
            for (int q = gopen; q < c; q++) HP_PIPE(hipStreamWaitEvent(st, hp.s[q % K].ev_comp, 0), "stream wait");
            if (trace) hipEventRecord(tev[1 + 3 * nch + 2 * g], st);
            if (!e.run_part(op.st, st, e.split_step, ns, o.d_in, gn, reinterpret_cast<float*>(hand + (size_t)first * hcb), o.d_logits, j.emb ? o.d_emb : nullptr, &err)) { abort_all(); return BNHIP_E_RUNTIME; }
            if (kk) {
                launch_activation(o.d_logits, o.d_conf, gn, e.n_classes, j.activation, j.sensitivity, st);
                launch_topk(o.d_conf, gn, e.n_classes, kk, o.d_tkc, o.d_tki, st);
            }
            HP_PIPE(hipEventRecord(o.ev_done, st), "event record");   // (back phase of group g done)
            if (trace) hipEventRecord(tev[2 + 3 * nch + 2 * g], st);
            gfirst.push_back(first); gcount.push_back(gn);
            gopen = c; g++;
        }
    }
    // copy-out, behind every H2D on the copy stream
    for (int q = 0; q < ng; q++) {
        HostPipe::Slot& o = hp.s[q];
        const size_t off = (size_t)gfirst[q], gn = (size_t)gcount[q];
        HP_PIPE(hipStreamWaitEvent(hp.xfer, o.ev_done, 0), "stream wait");
        if (kk) {
            HP_PIPE(hipMemcpyAsync(o.h_tkc, o.d_tkc, gn * kk * 4, hipMemcpyDeviceToHost, hp.xfer), "D2H copy");
            HP_PIPE(hipMemcpyAsync(o.h_tki, o.d_tki, gn * kk * 4, hipMemcpyDeviceToHost, hp.xfer), "D2H copy");
        }
        if (j.logits) HP_PIPE(hipMemcpyAsync(logits_pinned ? (void*)(j.logits + off * e.n_classes) : (void*)o.h_logits, o.d_logits, gn * e.n_classes * 4, hipMemcpyDeviceToHost, hp.xfer), "D2H copy");
        if (j.emb) HP_PIPE(hipMemcpyAsync(emb_pinned ? (void*)(j.emb + off * e.emb_dim) : (void*)o.h_emb, o.d_emb, gn * e.emb_dim * 4, hipMemcpyDeviceToHost, hp.xfer), "D2H copy");
        HP_PIPE(hipEventRecord(o.ev_out, hp.xfer), "event record");
    }
    for (int q = 0; q < ng; q++) {
        HostPipe::Slot& o = hp.s[q];
        const size_t off = (size_t)gfirst[q], gn = (size_t)gcount[q];
        HP_PIPE(hipEventSynchronize(o.ev_out), "D2H copy/sync");
        if (j.logits && !logits_pinned) parallel_copy(j.logits + off * e.n_classes, o.h_logits, gn * e.n_classes * 4, e.device);
        if (j.emb && !emb_pinned) memcpy(j.emb + off * e.emb_dim, o.h_emb, gn * e.emb_dim * 4);
        if (kk) {
            memcpy(j.out_conf + off * kk, o.h_tkc, gn * kk * 4);
            memcpy(j.out_idx + off * kk, o.h_tki, gn * kk * 4);
        }
    }
    if (trace) {
        for (int c = 0; c < nch; c++) {
            float a = 0, b = 0, d = 0;
            hipEventElapsedTime(&a, tev[0], tev[1 + 3 * c]); hipEventElapsedTime(&b, tev[0], tev[2 + 3 * c]); hipEventElapsedTime(&d, tev[0], tev[3 + 3 * c]);
            fprintf(stderr, "[bnhip] host trace: chunk %d (%d clips) GPU: h2d done %.3f, front start %.3f, done %.3f ms\n", c, csize[c], a, b, d);
        }
        for (int q = 0; q < ng; q++) {
            float a = 0, b = 0;
            hipEventElapsedTime(&a, tev[0], tev[1 + 3 * nch + 2 * q]); hipEventElapsedTime(&b, tev[0], tev[2 + 3 * nch + 2 * q]);
            fprintf(stderr, "[bnhip] host trace: group %d (%d clips from %d) GPU: back start %.3f, done %.3f ms\n", q, gcount[q], gfirst[q], a, b);
        }
    }
#undef HP_PIPE
    return BNHIP_OK;
}

}  // namespace

int host_run(Engine& e, const HostJob& j, std::string& err) {
    if (hipSetDevice(e.device) != hipSuccess) { err = "hipSetDevice failed"; return BNHIP_E_RUNTIME; }
    static const int min_pipe = getenv("BNHIP_HOST_PIPE_MIN") ? atoi(getenv("BNHIP_HOST_PIPE_MIN")) : 128;
    const int D = std::max(1, std::min(e.host_depth, Engine::kMaxDepth));
    if (j.n_clips < std::max(2, min_pipe) || D < 1) return small_run(e, j, err);

    // Chunk schedule.  The call is blocking, so its first chunk starts on an idle GPU only after its own staging + H2D, and
    // its last chunk ends alone on one context: both cost the more the bigger those chunks are.  Hence a ramp: u, 2u, then
    // chunks of up to max_batch, then 5u/2, u/2 (u = max_batch / 4; measured within 1.5 % of each other: 64/128..128/64,
    // 64/128..160/32, 64/160..160/48 - the schedule stopped being the lever once the streams stopped sharing queues).  Calls too small for the full ramp are cut into u-sized
    // chunks (a call that fits one batch still overlaps its later parts' copies with the earlier parts' compute).
    static const int ramp_env = getenv("BNHIP_HOST_RAMP") ? atoi(getenv("BNHIP_HOST_RAMP")) : -1;
    const int unit = ramp_env >= 0 ? ramp_env : std::max(8, e.max_batch / 4);
    std::vector<int> csize;
    if (unit <= 0 || unit * 2 > e.max_batch) {
        int nc = (j.n_clips + e.max_batch - 1) / e.max_batch;
        if (nc == 1) nc = 2;
        const int ck = (j.n_clips + nc - 1) / nc;
        for (int left = j.n_clips; left > 0; left -= ck) csize.push_back(std::min(ck, left));
    } else if (j.n_clips < 6 * unit + e.max_batch) {
        int nc = std::max(2, (j.n_clips + unit - 1) / unit);
        // (a two-phase call below three units: four equal chunks - 128 clips fp32 3.75 -> 3.36 ms, 144 / 160 clips 3 % faster, from
        // 192 clips on the unit-sized chunks win; tools/debug/host_quarters.py)
        if (e.split_step > 0 && D >= 2 && j.n_clips <= e.max_batch && j.n_clips < 3 * unit && j.n_clips >= 4) nc = 4;
        const int ck = (j.n_clips + nc - 1) / nc;
        for (int left = j.n_clips; left > 0; left -= ck) csize.push_back(std::min(ck, left));
    } else {
        int h1 = unit, h2 = 2 * unit, t1 = 5 * unit / 2, t2 = std::max(1, unit / 2);   // (the context that frees first takes the larger last piece)
        if (const char* sc = diag_env("BNHIP_HOST_SCHED")) sscanf(sc, "%d,%d,%d,%d", &h1, &h2, &t1, &t2);   // experiments
        h1 = std::max(1, std::min(h1, e.max_batch)); h2 = std::max(1, std::min(h2, e.max_batch));
        t1 = std::max(1, std::min(t1, e.max_batch)); t2 = std::max(1, std::min(t2, e.max_batch));
        if (h1 + h2 + t1 + t2 > j.n_clips) { h1 = unit; h2 = 2 * unit; t1 = 5 * unit / 2; t2 = std::max(1, unit / 2); }   // (a BNHIP_HOST_SCHED that does not fit the call)
        const int mid = j.n_clips - (h1 + h2 + t1 + t2), nm = (mid + e.max_batch - 1) / e.max_batch;
        csize.push_back(h1); csize.push_back(h2);
        for (int k = 0, left = mid; k < nm; k++) { const int c = (left + (nm - k) - 1) / (nm - k); csize.push_back(c); left -= c; }
        csize.push_back(t1); csize.push_back(t2);
    }
    if (const char* cc = diag_env("BNHIP_HOST_CHUNKS")) {    // experiments: an explicit schedule "32,64,96,64" (read per call; must add up)
        std::vector<int> want;
        long sum = 0;
        for (const char* q = cc; *q;) {
            char* end = nullptr;
            const long v = strtol(q, &end, 10);
            if (end == q || v < 1 || v > e.max_batch) { want.clear(); break; }
            want.push_back((int)v); sum += v;
            q = *end == ',' ? end + 1 : end;
            if (*end && *end != ',') { want.clear(); break; }
        }
        if (!want.empty() && sum == j.n_clips) csize = want;
    }
    const int nch = (int)csize.size();
    // a call that fits one batch: cut the plan as well as the batch (host_run_split)
    if (e.split_step > 0 && D >= 2 && j.n_clips <= e.max_batch && nch >= 2 && nch <= HostPipe::K && !diag_env("BNHIP_HOST_SERIAL") && !diag_env("BNHIP_HOST_NOSPLIT"))      // (diagnostics, read per call)
        return host_run_split(e, j, csize, err);
    std::vector<int> cfirst(nch + 1, 0);
    for (int c = 0; c < nch; c++) cfirst[c + 1] = cfirst[c] + csize[c];
    const size_t bps = j.pcm_bits ? (size_t)j.pcm_bits / 8 : 4;
    const size_t clip_bytes = (size_t)e.n_samples * bps;
    const int kk = j.topk > 0 ? std::min(j.topk, e.n_classes) : 0;
    HostJob jj = j; jj.topk = kk;
    int rc = ensure_pipe(e, jj, (size_t)e.max_batch * clip_bytes, err);
    if (rc) return rc;
    if (!e.ensure_contexts(D, &err, false)) { if (err.empty()) err = "host pipeline: context arena allocation failed"; return BNHIP_E_NOMEM; }
    HostPipe& hp = *e.hostpipe;
    constexpr int K = HostPipe::K;
    // The chunks run in the context arenas on the device's kernel streams; context 0's arena is the engine's own.  Anything an
    // earlier asynchronous bnhip_predict_device call (or a caller-owned stream, bnhip_set_stream) still has queued on them must
    // be finished first - the old one-batch path got that ordering from Engine::run().  Idle streams return immediately.
    e.sync_contexts();
    if (e.stream) hipStreamSynchronize(e.stream);
    for (int c = 0; c < Engine::kMaxDepth; c++) if (e.ctx_stream[c]) hipStreamSynchronize(e.ctx_stream[c]);

    const bool src_pinned = is_pinned(j.src, (size_t)j.n_clips * clip_bytes);
    if (src_pinned) g_pinned_inputs.fetch_add(1, std::memory_order_relaxed);
    const bool logits_pinned = j.logits && is_pinned(j.logits, (size_t)j.n_clips * e.n_classes * 4);
    const bool emb_pinned = j.emb && is_pinned(j.emb, (size_t)j.n_clips * e.emb_dim * 4);
    auto chunk_n = [&](int c) { return csize[c]; };
    auto abort_all = [&]() {
        for (auto& s : hp.s) { hp.cp->wait(&s.fill); s.chunk = -1; }
        hipStreamSynchronize(hp.xfer);
        for (int c = 0; c < D; c++) if (e.kstream[c]) hipStreamSynchronize(e.kstream[c]);
        hipStreamSynchronize(hp.xfer);
        (void)hipGetLastError();
    };
    // results of the chunk a slot holds -> the caller's buffers (blocks until the chunk is done on the GPU)
    auto finish = [&](HostPipe::Slot& s) -> hipError_t {
        if (s.chunk < 0) return hipSuccess;
        hipError_t he = hipEventSynchronize(s.ev_done);
        if (he != hipSuccess) return he;
        const size_t off = (size_t)cfirst[s.chunk], n = (size_t)chunk_n(s.chunk);
        if (j.logits && !logits_pinned) parallel_copy(j.logits + off * e.n_classes, s.h_logits, n * e.n_classes * 4, e.device);
        if (j.emb && !emb_pinned) memcpy(j.emb + off * e.emb_dim, s.h_emb, n * e.emb_dim * 4);
        if (kk) {
            memcpy(j.out_conf + off * kk, s.h_tkc, n * kk * 4);
            memcpy(j.out_idx + off * kk, s.h_tki, n * kk * 4);
        }
        s.chunk = -1;
        return hipSuccess;
    };
    auto start_fill = [&](int c) -> hipError_t {
        HostPipe::Slot& s = hp.s[c % K];
        hipError_t he = finish(s);                       // the slot's previous chunk (c - K) must have left it
        if (he != hipSuccess) return he;
        s.chunk = c;
        if (j.prepare) j.prepare(cfirst[c], chunk_n(c));   // (the producer's own threads; returns with the clips in place)
        if (!src_pinned) hp.cp->submit(s.h_in, (const char*)j.src + (size_t)cfirst[c] * clip_bytes, (size_t)chunk_n(c) * clip_bytes, &s.fill);
        return hipSuccess;
    };
#define HP_PIPE(call, what)                                                                \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            abort_all();                                                                   \
            err = std::string(what) + ": " + hipGetErrorString(e_);                        \
            return BNHIP_E_RUNTIME;                                                        \
        }                                                                                  \
    } while (0)

    static const bool trace_env = getenv("BNHIP_HOST_TRACE") != nullptr;
    TraceEvents tev;                                      // trace: [base][per chunk: h2d done, compute start, done]
    const bool trace = trace_env && tev.create(1 + 3 * (size_t)nch);
    if (trace) hipEventRecord(tev[0], hp.xfer);
    const bool serial = diag_env("BNHIP_HOST_SERIAL") != nullptr;      // diagnostics: one chunk at a time (read per call)
    auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_call = now_ms();
    std::vector<double> tr;
    // copy-out of chunk c on the copy stream, behind its kernels
    auto issue_d2h = [&](int c) -> hipError_t {
        HostPipe::Slot& s = hp.s[c % K];
        const int n = chunk_n(c);
        hipError_t he = hipStreamWaitEvent(hp.xfer, s.ev_comp, 0);
        if (he == hipSuccess && kk) he = hipMemcpyAsync(s.h_tkc, s.d_tkc, (size_t)n * kk * 4, hipMemcpyDeviceToHost, hp.xfer);
        if (he == hipSuccess && kk) he = hipMemcpyAsync(s.h_tki, s.d_tki, (size_t)n * kk * 4, hipMemcpyDeviceToHost, hp.xfer);
        const size_t off = (size_t)cfirst[c];
        if (he == hipSuccess && j.logits) he = hipMemcpyAsync(logits_pinned ? (void*)(j.logits + off * e.n_classes) : (void*)s.h_logits, s.d_logits, (size_t)n * e.n_classes * 4, hipMemcpyDeviceToHost, hp.xfer);
        if (he == hipSuccess && j.emb) he = hipMemcpyAsync(emb_pinned ? (void*)(j.emb + off * e.emb_dim) : (void*)s.h_emb, s.d_emb, (size_t)n * e.emb_dim * 4, hipMemcpyDeviceToHost, hp.xfer);
        if (he == hipSuccess) he = hipEventRecord(s.ev_done, hp.xfer);
        return he;
    };
    constexpr int LAG = 2;                                 // D2H(c) is issued after H2D(c + LAG): LAG < K - 1
    HP_PIPE(start_fill(0), "staging");
    for (int c = 0; c < nch; c++) {
        if (trace) tr.push_back(now_ms() - t_call);
        HostPipe::Slot& s = hp.s[c % K];
        const int n = chunk_n(c), ctx = c % D;
        const size_t cnt = (size_t)n * e.n_samples;
        hipStream_t cs = e.kernel_stream(ctx);
        if (!cs) { abort_all(); err = "hipStreamCreate failed"; return BNHIP_E_RUNTIME; }
        if (!src_pinned) hp.cp->wait(&s.fill);
        if (trace) tr.push_back(now_ms() - t_call);
        const void* h_src = src_pinned ? (const void*)((const char*)j.src + (size_t)cfirst[c] * clip_bytes) : (const void*)s.h_in;
        HP_PIPE(hipMemcpyAsync(j.pcm_bits ? (void*)s.d_raw : (void*)s.d_in, h_src, cnt * bps, hipMemcpyHostToDevice, hp.xfer), "H2D copy");
        HP_PIPE(hipEventRecord(s.ev_h2d, hp.xfer), "event record");
        if (trace) hipEventRecord(tev[1 + 3 * c], hp.xfer);
        if (c >= LAG) HP_PIPE(issue_d2h(c - LAG), "D2H copy");
        HP_PIPE(hipStreamWaitEvent(cs, s.ev_h2d, 0), "stream wait");
        if (trace) hipEventRecord(tev[2 + 3 * c], cs);
        if (j.pcm_bits) launch_pcm_to_f32(s.d_raw, j.pcm_bits, s.d_in, cnt, cs);      // a1: PCM -> float32 on the device
        if (!e.run_on_context(ctx, cs, s.d_in, n, s.d_logits, j.emb ? s.d_emb : nullptr, &err)) { abort_all(); return BNHIP_E_RUNTIME; }
        if (kk) {
            launch_activation(s.d_logits, s.d_conf, n, e.n_classes, j.activation, j.sensitivity, cs);
            launch_topk(s.d_conf, n, e.n_classes, kk, s.d_tkc, s.d_tki, cs);
        }
        HP_PIPE(hipEventRecord(s.ev_comp, cs), "event record");
        if (trace) hipEventRecord(tev[3 + 3 * c], cs);
        if (serial) HP_PIPE(hipStreamSynchronize(cs), "synchronize");
        if (trace) tr.push_back(now_ms() - t_call);
        // stage the next chunk while this one and its predecessor are on the GPU
        if (c + 1 < nch) HP_PIPE(start_fill(c + 1), "staging");
        if (trace) tr.push_back(now_ms() - t_call);
    }
    for (int c = std::max(0, nch - LAG); c < nch; c++) HP_PIPE(issue_d2h(c), "D2H copy");
    for (int c = std::max(0, nch - K); c < nch; c++) {
        HP_PIPE(finish(hp.s[c % K]), "D2H copy/sync");
        if (trace) fprintf(stderr, "[bnhip] host trace: chunk %d results delivered at %.3f ms\n", c, now_ms() - t_call);
    }
    if (trace) {
        for (int c = 0; c < nch; c++) {
            float a = 0, b = 0, d = 0;
            hipEventElapsedTime(&a, tev[0], tev[1 + 3 * c]); hipEventElapsedTime(&b, tev[0], tev[2 + 3 * c]); hipEventElapsedTime(&d, tev[0], tev[3 + 3 * c]);
            fprintf(stderr, "[bnhip] host trace: chunk %d (%d clips, ctx %d) GPU: h2d done %.3f, compute start %.3f, done %.3f ms\n", c, csize[c], c % D, a, b, d);
        }
    }
    if (trace)
        for (int c = 0; c < nch; c++)
            fprintf(stderr, "[bnhip] host trace: chunk %d: loop %.3f, filled %.3f, enqueued %.3f, next fill started %.3f ms\n", c,
                    tr[4 * c], tr[4 * c + 1], tr[4 * c + 2], tr[4 * c + 3]);
#undef HP_PIPE
    return BNHIP_OK;
}

}  // namespace bnhip

// Diagnostics of the NUMA placement (numa.h; not part of the boundary): what a device's copy pool looks like, and the sysfs parsing
// on a tree of the caller's choosing (CPU tests build a fake /sys).
extern "C" int bnhip_debug_copy_pool(int device, int* node, int* threads, int* bound, int* cpus) {
    try { bnhip::copy_pool_info(device, node, threads, bound, cpus); return 0; } catch (...) { return -1; }
}
extern "C" int bnhip_debug_numa_probe(const char* sysroot, const char* bdf, int* node, int* cpus, int cap) {
    try {
        const std::string root = sysroot ? sysroot : "/sys";
        const int nd = bnhip::pci_numa_node(bdf ? bdf : "", root);
        if (node) *node = nd;
        const std::vector<int> c = bnhip::numa_node_cpus(nd, root);
        for (size_t i = 0; i < c.size() && (int)i < cap && cpus; i++) cpus[i] = c[i];
        return (int)c.size();
    } catch (...) { return -1; }
}
extern "C" int bnhip_debug_parse_cpulist(const char* text, int* cpus, int cap) {
    try {
        const std::vector<int> c = bnhip::parse_cpulist(text ? text : "");
        for (size_t i = 0; i < c.size() && (int)i < cap && cpus; i++) cpus[i] = c[i];
        return (int)c.size();
    } catch (...) { return -1; }
}
extern "C" long bnhip_debug_split_calls(void) { return bnhip::g_split_calls.load(std::memory_order_relaxed); }
extern "C" long bnhip_debug_pinned_inputs(void) { return bnhip::g_pinned_inputs.load(std::memory_order_relaxed); }
