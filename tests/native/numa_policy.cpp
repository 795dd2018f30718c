// Test infrastructure: csrc/numa.cpp's NumaPrefer scope compiled with g++ (tests/test_numa.py).  The scope must give the allocation
// inside it a preferred node and put the calling thread's OWN policy back afterwards - a host started under `numactl --interleave`
// keeps its policy - and must change nothing when the kernel refuses the call.
#include <sys/syscall.h>
#include <unistd.h>

#include <cstdio>

#include "numa.h"

static int policy(unsigned long* mask0) {
    int mode = -1;
    unsigned long mask[17] = {0};
    if (syscall(SYS_get_mempolicy, &mode, mask, 1025ul, nullptr, 0ul) != 0) return -1;
    *mask0 = mask[0];
    return mode;
}

int main() {
    unsigned long m0 = 0, m1 = 0, m2 = 0;
    const int before = policy(&m0);
    if (before < 0) { printf("SKIP get_mempolicy refused\n"); return 0; }
    int inside = -1;
    bool active = false;
    {
        bnhip::NumaPrefer p(0);
        active = p.active();
        inside = policy(&m1);
    }
    const int after = policy(&m2);
    if (!active) {                                   // refused (seccomp, no NUMA support): nothing may have changed
        printf("%s\n", (inside == before && after == before && m1 == m0 && m2 == m0) ? "SKIP set_mempolicy refused, policy untouched" : "FAIL policy changed although the scope is inactive");
        return 0;
    }
    // an interleave policy set by the host survives a scope
    unsigned long il[17] = {1ul};
    int ok_il = syscall(SYS_set_mempolicy, 3 /* MPOL_INTERLEAVE */, il, 1025ul) == 0;
    int il_after = -1; unsigned long m3 = 0;
    if (ok_il) { { bnhip::NumaPrefer q(0); } il_after = policy(&m3); syscall(SYS_set_mempolicy, 0, nullptr, 0ul); }
    { bnhip::NumaPrefer none(-1); if (none.active()) { printf("FAIL node -1 must be a no-op\n"); return 0; } }
    const bool good = inside == 1 /* MPOL_PREFERRED */ && (m1 & 1ul) && after == before && m2 == m0 && (!ok_il || (il_after == 3 && (m3 & 1ul)));
    printf("%s before %d inside %d after %d interleave kept %d\n", good ? "OK" : "FAIL", before, inside, after, ok_il ? il_after == 3 : -1);
    return 0;
}
