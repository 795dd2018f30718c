// Window assembler (windows.h): per-source rings + overlap tails, all ready windows into one batch buffer.
#include "windows.h"

#include <algorithm>
#include <cstring>

namespace bnhip {

WindowAssembler::WindowAssembler(size_t overlap_bytes, size_t read_bytes, int max_batch)
    : overlap_(overlap_bytes), read_(read_bytes), max_batch_(std::max(1, max_batch)) {}

int WindowAssembler::add_source(const std::string& id, size_t capacity) {
    if (capacity < read_ || capacity == 0) return -1;       // analysis.go:91-100 (capacity must hold one read)
    auto s = std::make_unique<Source>();
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
    if (s.n < read_) return false;
    if (overlap_) {
        if (s.have_prev) std::memcpy(win, s.prev.data(), overlap_);
        else std::memset(win, 0, overlap_);
    }
    const size_t cap = s.ring.size(), first = std::min(read_, cap - s.r);
    std::memcpy(win + overlap_, s.ring.data() + s.r, first);
    if (read_ > first) std::memcpy(win + overlap_ + first, s.ring.data(), read_ - first);
    s.r = (s.r + read_) % cap; s.n -= read_;
    if (overlap_) {
        // (read >= overlap is part of the geometry's validation; a shorter read would keep the end of the old tail as well)
        if (read_ >= overlap_) std::memcpy(s.prev.data(), win + overlap_ + read_ - overlap_, overlap_);
        else {
            std::memmove(s.prev.data(), s.prev.data() + read_, overlap_ - read_);
            std::memcpy(s.prev.data() + overlap_ - read_, win + overlap_, read_);
        }
        s.have_prev = true;
    }
    return true;
}

int WindowAssembler::collect(uint8_t* batch, int cap, int* sources) {
    std::lock_guard<std::mutex> cg(collect_mu_);
    std::shared_lock<std::shared_mutex> lk(table_mu_);
    const size_t ns = src_.size();
    if (!ns || cap <= 0) return 0;
    cap = std::min(cap, max_batch_);
    const size_t wb = window_bytes();
    int k = 0;
    size_t i = 0;
    const size_t start = next_ % ns;
    for (; i < ns && k < cap; i++) {
        const size_t idx = (start + i) % ns;
        if (!src_[idx]) continue;
        if (read_window(*src_[idx], batch + (size_t)k * wb)) sources[k++] = (int)idx;
    }
    next_ = (start + i) % ns;                                // behind the last source looked at
    return k;
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
