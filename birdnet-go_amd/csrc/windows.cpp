// Window assembler (windows.h): per-source rings + overlap tails, all ready windows into one batch buffer.
#include "windows.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>

#if defined(__x86_64__) || defined(_M_X64)
#include <emmintrin.h>
#endif

namespace bnhip {

// Row bytes go into the (page-locked) batch buffer the copy engines read next and the CPU never reads again: streaming stores keep
// them out of the caches (no read-for-ownership of the destination lines, and the DMA does not have to snoop dirty lines out of
// 256 rows' worth of L2 - measured round 6: H2D of a freshly assembled 64-row chunk 0.43 ms against 0.35 ms from memory the CPU
// had not just written).  x86-64 only; elsewhere (and for the unaligned head / tail) plain memcpy.
static inline void copy_to_row(uint8_t* dst, const uint8_t* src, size_t n) {
#if defined(__x86_64__) || defined(_M_X64)
    if (n >= 4096) {
        const size_t head = (16 - (reinterpret_cast<uintptr_t>(dst) & 15)) & 15;
        if (head) { std::memcpy(dst, src, head); dst += head; src += head; n -= head; }
        size_t i = 0;
        for (; i + 64 <= n; i += 64) {
            const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i));
            const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i + 16));
            const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i + 32));
            const __m128i d = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i + 48));
            _mm_stream_si128(reinterpret_cast<__m128i*>(dst + i), a);
            _mm_stream_si128(reinterpret_cast<__m128i*>(dst + i + 16), b);
            _mm_stream_si128(reinterpret_cast<__m128i*>(dst + i + 32), c);
            _mm_stream_si128(reinterpret_cast<__m128i*>(dst + i + 48), d);
        }
        _mm_sfence();
        if (i < n) std::memcpy(dst + i, src + i, n - i);
        return;
    }
#endif
    std::memcpy(dst, src, n);
}

WindowAssembler::WindowAssembler(size_t overlap_bytes, size_t read_bytes, int max_batch)
    : overlap_(overlap_bytes), read_(read_bytes), max_batch_(std::max(1, max_batch)) {}

int WindowAssembler::add_source(const std::string& id, size_t capacity) {
    if (capacity < read_ || capacity == 0) return -1;       // analysis.go:91-100 (capacity must hold one read)
    auto s = std::make_shared<Source>();
    s->id = id;
    s->ring.assign(capacity, 0);
    s->prev.assign(overlap_, 0);
    std::unique_lock<std::shared_mutex> lk(table_mu_);
    for (size_t i = 0; i < src_.size(); i++)
        if (!src_[i]) { src_[i] = std::move(s); return (int)i; }
    src_.push_back(std::move(s));
    return (int)src_.size() - 1;
}

bool WindowAssembler::remove_source(int source) {
    std::unique_lock<std::shared_mutex> lk(table_mu_);
    if (source < 0 || (size_t)source >= src_.size() || !src_[source]) return false;
    { std::lock_guard<std::mutex> g(src_[source]->mu); src_[source]->removed = true; }     // (a collect in flight may still hold it)
    src_[source].reset();
    return true;
}

int WindowAssembler::n_sources() const {
    std::shared_lock<std::shared_mutex> lk(table_mu_);
    int n = 0;
    for (const auto& s : src_) n += s != nullptr;
    return n;
}

bool WindowAssembler::write(int source, const void* data, size_t n) {
    std::shared_lock<std::shared_mutex> lk(table_mu_);
    if (source < 0 || (size_t)source >= src_.size() || !src_[source]) return false;
    Source& s = *src_[source];
    const uint8_t* d = static_cast<const uint8_t*>(data);
    std::lock_guard<std::mutex> g(s.mu);
    const size_t cap = s.ring.size();
    s.writes++;
    if (n > cap - s.n) s.overwrites++;                      // willOverwrite := len(data) > ring.Free()
    if (n > cap) {                                          // only the newest `cap` bytes can survive; everything unread is older
        d += n - cap; n = cap;
        s.r = (s.r + s.n) % cap; s.n = 0;
    }
    const size_t free_ = cap - s.n;
    if (n > free_) {                                        // the read position moves past the bytes being overwritten
        const size_t over = n - free_;
        s.r = (s.r + over) % cap; s.n -= over;
    }
    const size_t w = (s.r + s.n) % cap, first = std::min(n, cap - w);
    if (first) std::memcpy(s.ring.data() + w, d, first);
    if (n > first) std::memcpy(s.ring.data(), d + first, n - first);
    s.n += n;
    return true;
}

// analysis.go:187-252 with n == readSize (the caller checked Length() >= readSize under the same lock): prefix = the previous
// window's last `overlap` bytes (zeros the first time), then `read` fresh bytes; the new tail is the window's last `overlap` bytes.
bool WindowAssembler::read_window(Source& s, uint8_t* win) {
    std::lock_guard<std::mutex> g(s.mu);
    if (s.removed || s.n < read_) return false;
    if (overlap_) {
        if (s.have_prev) copy_to_row(win, s.prev.data(), overlap_);
        else std::memset(win, 0, overlap_);
    }
    const size_t cap = s.ring.size(), first = std::min(read_, cap - s.r);
    copy_to_row(win + overlap_, s.ring.data() + s.r, first);
    if (read_ > first) copy_to_row(win + overlap_ + first, s.ring.data(), read_ - first);
    if (overlap_) {
        // the new tail = the window's last `overlap` bytes, taken from where they still sit in the CPU's caches (the ring; the old
        // tail), not read back from the row that was just streamed past them
        // (read >= overlap is part of the geometry's validation; a shorter read keeps the end of the old tail as well)
        if (read_ >= overlap_) {
            const size_t t0 = (s.r + read_ - overlap_) % cap, f2 = std::min(overlap_, cap - t0);
            std::memcpy(s.prev.data(), s.ring.data() + t0, f2);
            if (overlap_ > f2) std::memcpy(s.prev.data() + f2, s.ring.data(), overlap_ - f2);
        } else {
            std::memmove(s.prev.data(), s.prev.data() + read_, overlap_ - read_);
            const size_t f2 = std::min(read_, cap - s.r);
            std::memcpy(s.prev.data() + overlap_ - read_, s.ring.data() + s.r, f2);
            if (read_ > f2) std::memcpy(s.prev.data() + overlap_ - read_ + f2, s.ring.data(), read_ - f2);
        }
        s.have_prev = true;
    }
    s.r = (s.r + read_) % cap; s.n -= read_;
    return true;
}

// A window is three copies of half a clip each (tail -> row, ring -> row, row -> tail): 256 windows of BirdNET v2.4 are 110 MB of
// memcpy, 3.4 ms on one thread - as long as the device needs for the batch.  The rows are independent, so a few helper threads
// take them (process-wide, started on first use, never joined: a destructor joining threads at dlclose time can deadlock
// under the loader lock, and a Go host exits without static destructors anyway).  BNHIP_COPY_THREADS as for the host pipeline.
namespace {

class RowPool {
  public:
    RowPool() {
        int n = 0;
        if (const char* e = getenv("BNHIP_COPY_THREADS")) n = atoi(e);
        else n = (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency() / 4));
        n = std::max(0, std::min(n, 64));
        for (int i = 0; i < n; i++) std::thread([this] { loop(); }).detach();
    }
    // fn(i) for i in [0, n): the caller works too; returns when all are done
    void run(int n, const std::function<void(int)>& fn) {
        if (n <= 0) return;
        Job job{&fn, n};
        {
            std::lock_guard<std::mutex> lk(mu_);
            jobs_.push_back(&job);
        }
        cv_.notify_all();
        work(job);
        std::unique_lock<std::mutex> lk(mu_);
        jobs_.erase(std::find(jobs_.begin(), jobs_.end(), &job));         // no new helper may pick it up ...
        done_.wait(lk, [&] { return job.active == 0; });                  // ... and the ones inside have left
    }

  private:
    struct Job {
        const std::function<void(int)>* fn;
        int n;
        std::atomic<int> next{0};
        int active = 0;                                      // helpers inside work(); under mu_
    };
    static void work(Job& j) {
        for (int i; (i = j.next.fetch_add(1, std::memory_order_relaxed)) < j.n;) (*j.fn)(i);
    }
    void loop() {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            Job* j = nullptr;
            cv_.wait(lk, [&] {
                for (Job* c : jobs_) if (c->next.load(std::memory_order_relaxed) < c->n) { j = c; return true; }
                return false;
            });
            j->active++;
            lk.unlock();
            work(*j);
            lk.lock();
            if (--j->active == 0) done_.notify_all();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_, done_;
    std::vector<Job*> jobs_;
};

RowPool& row_pool() {
    static RowPool* p = new RowPool();
    return *p;
}

}  // namespace

int WindowAssembler::collect_begin(int cap, int* sources) {
    collect_mu_.lock();
    held_.clear();
    std::shared_lock<std::shared_mutex> lk(table_mu_);      // released on return: the rows' sources are held by reference
    const size_t ns = src_.size();
    if (!ns || cap <= 0) return 0;
    cap = std::min(cap, max_batch_);
    // who is ready (only the collecting thread consumes, so a ready source stays ready unless it is reset or removed in between)
    int k = 0;
    size_t i = 0;
    const size_t start = next_ % ns;
    for (; i < ns && k < cap; i++) {
        const size_t idx = (start + i) % ns;
        if (!src_[idx]) continue;
        std::lock_guard<std::mutex> g(src_[idx]->mu);
        if (src_[idx]->n >= read_) { sources[k++] = (int)idx; held_.push_back(src_[idx]); }
    }
    next_ = (start + i) % ns;                                // behind the last source looked at
    return k;
}

void WindowAssembler::collect_rows(uint8_t* batch, int* sources, int first, int n) {
    const size_t wb = window_bytes();
    auto fill = [&](int q) {
        const int r = first + q;
        uint8_t* row = batch + (size_t)r * wb;
        if (sources[r] < 0 || (size_t)r >= held_.size() || !read_window(*held_[r], row)) { std::memset(row, 0, wb); sources[r] = -1; }
    };
    if ((size_t)n * wb >= ((size_t)4 << 20) && n > 1) row_pool().run(n, fill);
    else for (int q = 0; q < n; q++) fill(q);
}

void WindowAssembler::collect_end() {
    held_.clear();
    collect_mu_.unlock();
}

int WindowAssembler::collect(uint8_t* batch, int cap, int* sources) {
    const int k = collect_begin(cap, sources);
    collect_rows(batch, sources, 0, k);
    const size_t wb = window_bytes();
    int out = 0;                                             // (a source reset between the passes: close the gap)
    for (int r = 0; r < k; r++) {
        if (sources[r] < 0) continue;
        if (out != r) { std::memmove(batch + (size_t)out * wb, batch + (size_t)r * wb, wb); sources[out] = sources[r]; }
        out++;
    }
    collect_end();
    return out;
}

int WindowAssembler::ready() const {
    std::shared_lock<std::shared_mutex> lk(table_mu_);
    int n = 0;
    for (const auto& s : src_) {
        if (!s) continue;
        std::lock_guard<std::mutex> g(s->mu);
        n += s->n >= read_;
    }
    return n;
}

bool WindowAssembler::stats(int source, uint64_t* writes, uint64_t* overwrites, size_t* buffered) const {
    std::shared_lock<std::shared_mutex> lk(table_mu_);
    if (source < 0 || (size_t)source >= src_.size() || !src_[source]) return false;
    const Source& s = *src_[source];
    std::lock_guard<std::mutex> g(s.mu);
    if (writes) *writes = s.writes;
    if (overwrites) *overwrites = s.overwrites;
    if (buffered) *buffered = s.n;
    return true;
}

bool WindowAssembler::reset(int source) {
    std::shared_lock<std::shared_mutex> lk(table_mu_);
    if (source < 0 || (size_t)source >= src_.size() || !src_[source]) return false;
    Source& s = *src_[source];
    std::lock_guard<std::mutex> g(s.mu);
    s.r = s.n = 0; s.have_prev = false;
    s.writes = s.overwrites = 0;
    return true;
}

}  // namespace bnhip
