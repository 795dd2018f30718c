// Do back-to-back kernels in ONE stream see each other's global writes while ANOTHER stream keeps the GPU busy?
// stream A: fill(buf, i) -> check(buf, i) repeated; stream B (optional): a streaming copy loop.
// hipcc --offload-arch=gfx950 -O2 tools/ubench/xstream_visibility.hip -o /tmp/xsv && /tmp/xsv
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k_fill(float* buf, size_t n, float v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) buf[i] = v + (float)(i & 1023);
}
// a different block -> element map than the producer's, so consumer blocks read lines other CUs / XCDs wrote
__global__ void k_check(const float* buf, size_t n, float v, unsigned* err) {
    size_t i = (size_t)(gridDim.x - 1 - blockIdx.x) * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    unsigned bad = 0;
    for (; i < n; i += st) bad += buf[i] != v + (float)(i & 1023);
    if (bad) atomicAdd(err, bad);
}
__global__ void k_copy(const float4* a, float4* b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) b[i] = a[i];
}
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 400;
    const size_t n = (size_t)8 << 20;                       // 32 MB of floats
    float *buf, *ca, *cb; unsigned* err;
    hipMalloc(&buf, n * 4); hipMalloc(&ca, (size_t)256 << 20); hipMalloc(&cb, (size_t)256 << 20); hipMalloc(&err, 4);
    hipMemset(ca, 0, (size_t)256 << 20);
    hipStream_t sa, sb;
    hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    for (int busy = 0; busy < 2; busy++) {
        hipMemset(err, 0, 4);
        hipDeviceSynchronize();
        for (int i = 0; i < iters; i++) {
            hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, sa, buf, n, (float)i);
            hipLaunchKernelGGL(k_check, dim3(1536), dim3(256), 0, sa, buf, n, (float)i, err);
            if (busy) hipLaunchKernelGGL(k_copy, dim3(4096), dim3(256), 0, sb, (const float4*)ca, (float4*)cb, ((size_t)256 << 20) / 16);
        }
        hipDeviceSynchronize();
        unsigned h = 0; hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost);
        printf("other stream busy=%d: %d fill->check pairs, mismatching elements %u\n", busy, iters, h);
    }
    return 0;
}
