// Host-pointer pipeline behind the blocking C-ABI entries (bnhip_predict, _pcm16, _pcm, _topk).
//
// The reference's callers hand the backend pageable host memory and block until the logits are back
// (internal/analysis/process.go:280-295 copy-before-return; internal/inference/onnx/classifier.go:372-430 PredictBatch).
// A call of >= 128 clips is cut into chunks; chunk i runs the whole plan on context i % host_depth (own stream + activation
// arena), fed from a ring of pinned staging slots that a small copy pool fills from the caller's memory, so the copy and
// front half of chunk i+1 overlap the back half of chunk i - the same overlap successive bnhip_predict_device calls get
// with "depth":2 - while the completion contract stays blocking: every output is in the caller's buffers on return.
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <string>

namespace bnhip {

class Engine;

struct HostJob {
    const void* src = nullptr;   // host samples: float32, or little-endian PCM of pcm_bits (16 / 24 / 32)
    int pcm_bits = 0;
    int n_clips = 0;
    float* logits = nullptr;     // host [n_clips * n_classes] (nullable when topk > 0)
    float* emb = nullptr;        // host [n_clips * emb_dim], nullable
    // device post-processing (bnhip_predict_topk): activation + top-k without the logits leaving the device
    int topk = 0, activation = 0;
    double sensitivity = 1.0;
    float* out_conf = nullptr;
    int32_t* out_idx = nullptr;
    // Optional producer of the input: when set, clips [first, first + n) of `src` do not exist until prepare(first, n) has
    // returned.  The pipeline calls it once per chunk, in order, where it would otherwise start staging that chunk - i.e. while
    // the previous chunks are on the device (bnhip_windows_predict_topk: the window assembler fills the rows then).
    std::function<void(int first, int n)> prepare;
};

// Runs the job on one engine (one shard of a multi-device call).  Returns a BNHIP_* code; err carries the message.
int host_run(Engine& e, const HostJob& j, std::string& err);
void hostpipe_free(struct HostPipe* hp);

// memcpy spread over the copy pool of `device`'s NUMA node (BNHIP_COPY_THREADS per pool, default min(8, cores / 4), never more
// than the node's usable CPUs; device < 0 or no NUMA information: the unbound pool); callable concurrently.
void parallel_copy(void* dst, const void* src, size_t bytes, int device = -1);
int copy_pool_threads(int device = -1);
// NUMA node of a HIP device from its PCI address (-1: unknown, or BNHIP_NUMA=0), and what its pool looks like
int device_numa_node(int device);
void copy_pool_info(int device, int* node, int* threads, int* bound, int* cpus);

}  // namespace bnhip
