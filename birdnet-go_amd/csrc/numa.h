// NUMA placement of the host side of the ingest path (round 6; SURVEY §8e: "scaling limiter is host PCM staging bandwidth").
//
// One MI355X node is two CPU sockets with four GPUs behind each.  A clip travels caller memory -> pinned staging slot ->
// PCIe; when the staging slot or the thread that fills it sits on the other socket every byte crosses the socket
// interconnect twice.  So, per GPU: the copy threads that serve it run on the CPUs of the GPU's own NUMA node and its pinned
// slots are allocated while the allocating thread prefers that node.  Everything here is best effort - a container without
// /sys, a cpuset that excludes the node's CPUs or a seccomp profile that refuses set_mempolicy leaves the defaults in place -
// and nothing here touches HIP (the PCI address comes from the caller).
#pragma once
#include <string>
#include <vector>

namespace bnhip {

// "0-3,8,10-11" -> {0,1,2,3,8,10,11}; malformed input -> empty
std::vector<int> parse_cpulist(const std::string& s);
// <sysroot>/bus/pci/devices/<bdf>/numa_node (bdf as hipDeviceGetPCIBusId prints it, any case); -1 = unknown / not a NUMA box
int pci_numa_node(const std::string& bdf, const std::string& sysroot = "/sys");
// <sysroot>/devices/system/node/node<N>/cpulist
std::vector<int> numa_node_cpus(int node, const std::string& sysroot = "/sys");
// the node's CPUs this process may run on (intersection with sched_getaffinity); empty = leave threads unbound
std::vector<int> numa_usable_cpus(int node, const std::string& sysroot = "/sys");
// binds the CALLING thread to `cpus`; false (nothing changed) when the list is empty or the kernel refuses
bool bind_this_thread(const std::vector<int>& cpus);

// While alive, pages the calling thread faults in (or a driver pins on its behalf: hipHostMalloc) are taken from `node` when it
// has room (MPOL_PREFERRED through the raw syscall: no libnuma in the image).  node < 0 or a refusing kernel: a no-op.
class NumaPrefer {
  public:
    explicit NumaPrefer(int node);
    ~NumaPrefer();
    bool active() const { return active_; }
    NumaPrefer(const NumaPrefer&) = delete;
    NumaPrefer& operator=(const NumaPrefer&) = delete;

  private:
    bool active_ = false;
    int saved_mode_ = 0;                       // the thread's policy before the scope (put back by the destructor)
    std::vector<unsigned long> saved_mask_;
};

}  // namespace bnhip
