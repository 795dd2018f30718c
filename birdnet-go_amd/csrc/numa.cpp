// NUMA placement helpers: see numa.h.
#include "numa.h"

#include <pthread.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>

namespace bnhip {

std::vector<int> parse_cpulist(const std::string& s) {
    std::vector<int> out;
    size_t i = 0;
    const size_t n = s.size();
    auto skip = [&] { while (i < n && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t')) i++; };
    auto number = [&](long* v) {
        skip();
        if (i >= n || !isdigit((unsigned char)s[i])) return false;
        long x = 0;
        while (i < n && isdigit((unsigned char)s[i])) { x = x * 10 + (s[i] - '0'); if (x > 1 << 20) return false; i++; }
        *v = x;
        return true;
    };
    skip();
    while (i < n) {
        long a = 0, b = 0;
        if (!number(&a)) return {};
        b = a;
        skip();
        if (i < n && s[i] == '-') { i++; if (!number(&b) || b < a) return {}; }
        for (long c = a; c <= b; c++) out.push_back((int)c);
        skip();
        if (i < n) { if (s[i] != ',') return {}; i++; skip(); }
    }
    std::sort(out.begin(), out.end());
    out.erase(std::unique(out.begin(), out.end()), out.end());
    return out;
}

static bool read_small(const std::string& path, std::string* out) {
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return false;
    char buf[4096];
    const size_t k = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    out->assign(buf, k);
    return true;
}

int pci_numa_node(const std::string& bdf, const std::string& sysroot) {
    if (bdf.empty()) return -1;
    std::string id;
    for (char c : bdf) id += (char)tolower((unsigned char)c);
    std::string text;
    if (!read_small(sysroot + "/bus/pci/devices/" + id + "/numa_node", &text)) return -1;
    char* end = nullptr;
    const long v = strtol(text.c_str(), &end, 10);
    if (end == text.c_str() || v < 0 || v > 4095) return -1;          // the kernel prints -1 on a box without NUMA information
    return (int)v;
}

std::vector<int> numa_node_cpus(int node, const std::string& sysroot) {
    if (node < 0) return {};
    std::string text;
    if (!read_small(sysroot + "/devices/system/node/node" + std::to_string(node) + "/cpulist", &text)) return {};
    return parse_cpulist(text);
}

std::vector<int> numa_usable_cpus(int node, const std::string& sysroot) {
    std::vector<int> cpus = numa_node_cpus(node, sysroot), out;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (cpus.empty() || sched_getaffinity(0, sizeof set, &set) != 0) return {};
    for (int c : cpus) if (c < CPU_SETSIZE && CPU_ISSET(c, &set)) out.push_back(c);
    return out;
}

bool bind_this_thread(const std::vector<int>& cpus) {
    cpu_set_t set;
    CPU_ZERO(&set);
    int n = 0;
    for (int c : cpus) if (c >= 0 && c < CPU_SETSIZE) { CPU_SET(c, &set); n++; }
    if (!n) return false;
    return pthread_setaffinity_np(pthread_self(), sizeof set, &set) == 0;
}

namespace {
// <linux/mempolicy.h> values; the syscalls are reached directly (glibc has no wrappers, libnuma is not in the image)
constexpr int kMpolPreferred = 1;
constexpr unsigned long kMaxNode = 1024;
long set_mempolicy_raw(int mode, const unsigned long* mask, unsigned long maxnode) {
#ifdef SYS_set_mempolicy
    return syscall(SYS_set_mempolicy, mode, mask, maxnode);
#else
    (void)mode; (void)mask; (void)maxnode;
    return -1;
#endif
}
long get_mempolicy_raw(int* mode, unsigned long* mask, unsigned long maxnode) {
#ifdef SYS_get_mempolicy
    return syscall(SYS_get_mempolicy, mode, mask, maxnode, nullptr, 0ul);
#else
    (void)mode; (void)mask; (void)maxnode;
    return -1;
#endif
}
}  // namespace

// The calling thread's own policy is SAVED and put back: a host started under `numactl --interleave` / `--membind` (or a Go
// runtime thread whose policy somebody set) keeps it - only the allocation inside the scope sees the preference.  If the old
// policy cannot be read, nothing is changed at all.
NumaPrefer::NumaPrefer(int node) {
    if (node < 0 || node >= (int)kMaxNode) return;
    saved_mask_.assign(kMaxNode / (8 * sizeof(unsigned long)) + 1, 0ul);
    if (get_mempolicy_raw(&saved_mode_, saved_mask_.data(), kMaxNode + 1) != 0) return;
    unsigned long mask[kMaxNode / (8 * sizeof(unsigned long)) + 1] = {0};
    mask[node / (8 * sizeof(unsigned long))] |= 1ul << (node % (8 * sizeof(unsigned long)));
    active_ = set_mempolicy_raw(kMpolPreferred, mask, kMaxNode + 1) == 0;
}

NumaPrefer::~NumaPrefer() {
    if (!active_) return;
    bool any = false;
    for (unsigned long w : saved_mask_) any = any || w != 0;
    set_mempolicy_raw(saved_mode_, any ? saved_mask_.data() : nullptr, any ? kMaxNode + 1 : 0);
}

}  // namespace bnhip
