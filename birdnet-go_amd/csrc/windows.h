// Window assembler: the per-source analysis buffers of the reference's real-time path turned the way one
// device call per tick wants them.
//
//   internal/audiocore/buffer/analysis.go:30-276    AnalysisBuffer: ring in overwrite mode, Read() = `overlap ‖ fresh`,
//                                                   first prefix zeros, "not enough data" = try again later
//   internal/analysis/buffer_manager.go:388-496     one 100 ms poll loop per (source, model), each ending in a batch-1 Predict
//
// Every source keeps its own ring and its own overlap tail exactly as the reference does; collect() is all those poll loops'
// Read() calls in one pass, each ready window written straight into its row of ONE batch buffer (page-locked when a device is
// there: the copy engines read it in place, hostpipe.cpp is_pinned) - the bytes of a window are moved once between the capture
// callback and HBM.  Byte work only; the 16/24/32-bit conversion happens on the device (bnhip_predict_pcm).
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <vector>

namespace bnhip {

class WindowAssembler {
  public:
    // (validation of the geometry is the caller's: analysis.go:56-109)
    WindowAssembler(size_t overlap_bytes, size_t read_bytes, int max_batch);

    size_t window_bytes() const { return overlap_ + read_; }
    size_t overlap_bytes() const { return overlap_; }
    size_t read_bytes() const { return read_; }
    int max_batch() const { return max_batch_; }

    // -> source index (slots of removed sources are reused), -1 when `capacity` cannot hold one read
    int add_source(const std::string& id, size_t capacity);
    bool remove_source(int source);
    // analysis.go:152-175: never fails for a live source; the oldest unread bytes go when the data does not fit, of a write longer
    // than the ring the newest `capacity` bytes survive.  false = no such source.
    bool write(int source, const void* data, size_t n);
    // Every source with at least read_bytes buffered, starting behind the last source served (a tick that hits `cap` does not
    // starve the sources at the end of the table): window k goes to batch + k * window_bytes(), its source index to sources[k].
    int collect(uint8_t* batch, int cap, int* sources);
    // The same in steps, for a consumer that overlaps the copies with something else (api.cpp bnhip_windows_predict_topk: the
    // host pipeline fills chunk c + 1's rows while chunk c is on the device).  begin: who is ready (row r <- sources[r]); holds
    // the COLLECT lock until end, which the same thread must call - the table lock is released before begin returns (the ready
    // sources are pinned by reference for the duration), so add_source / remove_source / write never wait for a device call
    // (on a writer-preferring shared mutex a pending remove_source stalled every capture thread until the call
    // ended).  rows: fills rows [first, first + n) - callable from several threads on disjoint ranges; a source that was reset
    // or removed since begin yields a row of zeros and sources[r] = -1.
    int collect_begin(int cap, int* sources);
    void collect_rows(uint8_t* batch, int* sources, int first, int n);
    void collect_end();
    int ready() const;
    bool stats(int source, uint64_t* writes, uint64_t* overwrites, size_t* buffered) const;
    bool reset(int source);                                  // analysis.go:270-276
    int n_sources() const;

  private:
    struct Source {
        mutable std::mutex mu;
        std::string id;
        std::vector<uint8_t> ring, prev;
        size_t r = 0, n = 0;                                 // read position, unread bytes
        bool have_prev = false;
        bool removed = false;                                // remove_source() ran: a collect that still holds a reference reads nothing
        uint64_t writes = 0, overwrites = 0;
    };
    bool read_window(Source& s, uint8_t* win);

    const size_t overlap_, read_;
    const int max_batch_;
    mutable std::shared_mutex table_mu_;                     // the table; a source's bytes are under its own mutex
    std::vector<std::shared_ptr<Source>> src_;               // nullptr = free slot
    std::mutex collect_mu_;                                  // one collect() at a time (writers run beside it)
    std::vector<std::shared_ptr<Source>> held_;              // collect_begin .. collect_end: the sources of the rows, under collect_mu_
    size_t next_ = 0;                                        // where the next collect() starts
};

}  // namespace bnhip
