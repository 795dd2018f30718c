// Test infrastructure: csrc/windows.cpp under ThreadSanitizer (tests/test_stream.py builds it with g++ -fsanitize=thread).
// Writers on every source, two assemblers collecting at once through the shared row pool, resets and source churn beside
// them; every collected row is checked against the stream it was cut from.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "windows.h"

using bnhip::WindowAssembler;

static uint8_t byte_at(int src, size_t pos) { return (uint8_t)((pos * 131u + (size_t)src * 17u) >> 3); }

static int run_one(int seed, size_t ov, size_t rd, int nsrc, int ticks) {
    WindowAssembler w(ov, rd, nsrc);
    std::vector<int> ids;
    for (int k = 0; k < nsrc; k++) ids.push_back(w.add_source("s" + std::to_string(seed) + "-" + std::to_string(k), (size_t)ticks * rd + 16));
    const size_t total = (size_t)ticks * rd;
    std::atomic<int> writers_left{nsrc};
    std::vector<std::thread> th;
    for (int k = 0; k < nsrc; k++)
        th.emplace_back([&, k] {
            std::vector<uint8_t> buf(4096);
            size_t pos = 0;
            unsigned step = 977 + 13 * (unsigned)k;
            while (pos < total) {
                const size_t n = std::min<size_t>(std::min<size_t>(step % 4096 + 1, buf.size()), total - pos);
                for (size_t i = 0; i < n; i++) buf[i] = byte_at(seed * 100 + k, pos + i);
                w.write(ids[k], buf.data(), n);
                pos += n; step = step * 1103515245u + 12345u;
            }
            writers_left--;
        });
    // a churn thread: adds and removes a spare source, resets it, reads the stats of the live ones
    std::atomic<bool> stop{false};
    std::thread churn([&] {
        while (!stop) {
            const int s = w.add_source("spare", rd * 2);
            uint8_t junk[64] = {0};
            w.write(s, junk, sizeof junk);
            w.reset(s);
            uint64_t a, b; size_t c;
            for (int k : ids) w.stats(k, &a, &b, &c);
            w.ready();
            w.remove_source(s);
        }
    });
    std::vector<uint8_t> batch(((size_t)nsrc + 1) * (ov + rd));
    std::vector<int> src((size_t)nsrc + 1);
    std::vector<size_t> done((size_t)nsrc, 0);              // fresh bytes seen per source
    int bad = 0;
    size_t got_total = 0;
    while (got_total < (size_t)nsrc * ticks) {
        const int n = w.collect(batch.data(), nsrc + 1, src.data());
        for (int r = 0; r < n; r++) {
            int k = -1;
            for (int q = 0; q < nsrc; q++) if (ids[q] == src[r]) k = q;
            if (k < 0) continue;                             // the spare source never has a window ready
            const uint8_t* row = batch.data() + (size_t)r * (ov + rd);
            const size_t base = done[k];
            for (size_t i = 0; i < ov; i++) {
                const uint8_t want = base >= ov - i ? byte_at(seed * 100 + k, base - ov + i) : 0;
                if (base == 0 ? row[i] != 0 : row[i] != want) bad++;
            }
            for (size_t i = 0; i < rd; i++) if (row[ov + i] != byte_at(seed * 100 + k, base + i)) bad++;
            done[k] += rd; got_total++;
        }
        if (!n) std::this_thread::yield();
    }
    stop = true;
    churn.join();
    for (auto& t : th) t.join();
    for (int k = 0; k < nsrc; k++) {
        uint64_t wr = 0, ovw = 0; size_t buffered = 1;
        w.stats(ids[k], &wr, &ovw, &buffered);
        if (ovw != 0 || buffered != 0) bad++;               // the rings hold the whole stream: nothing may have been dropped
    }
    return bad;
}

// Rings barely larger than one read, writes of every size up to three rings' worth: the overwrite paths (read position pushed
// forward, writes longer than the ring) under the sanitizers.  What can be said about the content: a window's fresh half is a
// run of consecutive stream positions only when nothing was dropped inside it, so the check is on the accounting instead -
// buffered bytes never exceed the capacity, overwriting writes are counted, every row is fully written (no poison left).
static int run_overwrite(int seed) {
    const size_t ov = 24, rd = 40, cap = 47;
    const int nsrc = 6;
    WindowAssembler w(ov, rd, 4);
    std::vector<int> ids;
    for (int k = 0; k < nsrc; k++) ids.push_back(w.add_source("o" + std::to_string(k), cap + (size_t)k));
    std::atomic<bool> stop{false};
    std::vector<std::thread> th;
    for (int k = 0; k < nsrc; k++)
        th.emplace_back([&, k] {
            std::vector<uint8_t> buf(3 * (cap + 8), (uint8_t)(k + 1));
            unsigned step = (unsigned)seed * 7919u + (unsigned)k;
            while (!stop) {
                step = step * 1103515245u + 12345u;
                w.write(ids[k], buf.data(), (step >> 8) % buf.size());
            }
        });
    int bad = 0;
    std::vector<uint8_t> batch(4 * (ov + rd));
    int src[4];
    for (int it = 0; it < 20000; it++) {
        std::fill(batch.begin(), batch.end(), (uint8_t)0xEE);
        const int n = w.collect(batch.data(), 1 + it % 5, src);
        if (n < 0 || n > 4) bad++;
        for (int r = 0; r < n; r++) {
            const int k = (int)(std::find(ids.begin(), ids.end(), src[r]) - ids.begin());
            const uint8_t* row = batch.data() + (size_t)r * (ov + rd);
            for (size_t i = 0; i < ov + rd; i++) if (row[i] != (uint8_t)(k + 1) && !(i < ov && row[i] == 0)) bad++;   // own bytes, or the zero prefix
        }
        if (it % 997 == 0) w.reset(ids[(size_t)it % nsrc]);
    }
    // (a reset zeroes the counters: on a loaded box a writer may not have been scheduled since the last one - wait for every
    // writer to have overwritten at least once before the accounting below is read, instead of betting on the scheduler)
    for (int spin = 0; spin < 200000; spin++) {
        bool all = true;
        for (int k = 0; k < nsrc; k++) { uint64_t wr = 0, ovw = 0; size_t bf = 0; w.stats(ids[k], &wr, &ovw, &bf); all = all && ovw > 0; }
        if (all) break;
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    stop = true;
    for (auto& t : th) t.join();
    for (int k = 0; k < nsrc; k++) {
        uint64_t wr = 0, ovw = 0; size_t buffered = 0;
        w.stats(ids[k], &wr, &ovw, &buffered);
        if (buffered > cap + (size_t)k || ovw > wr || ovw == 0) bad++;
    }
    return bad;
}

int main() {
    std::atomic<int> bad{0};
    std::thread a([&] { bad += run_one(1, 256 * 1024, 256 * 1024, 12, 4); });     // 6 MB batches: the row pool
    std::thread b([&] { bad += run_one(2, 192 * 1024, 320 * 1024, 10, 3); });     // a second assembler beside it
    a.join(); b.join();
    const int c = run_one(3, 8, 24, 5, 50) + run_overwrite(4);                     // small rows: the serial path; overwrite mode
    printf("windows stress: %d mismatches\n", bad + c);
    return bad + c ? 1 : 0;
}
