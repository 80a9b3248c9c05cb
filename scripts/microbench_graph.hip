// Micro-benchmark: a chain of 8 dependent small kernels (the shape of a frame's launch sequence) enqueued one by one against the same
// chain as a hipGraph with one node's parameters updated per launch -- host microseconds per chain, and the device-side time from
// the first kernel's start to the last one's end (i.e. the kernels plus the gaps between dependent launches).
// Build + run on the GPU box: /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o scratch/graph scripts/microbench_graph.hip && ./scratch/graph
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_step(float* data, int n, float add, int spin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = i < n ? data[i] : 0.0f;
    for (int k = 0; k < spin; k++) v = v * 1.0001f + add;
    if (i < n) data[i] = v;
}

int main() {
    const int n = 512 * 256, chain = 8, reps = 2000;
    float* d; CK(hipMalloc(&d, n * 4)); CK(hipMemset(d, 0, n * 4));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int spin : {0, 2000}) {
        // (a) one launch per kernel
        for (int w = 0; w < 50; w++) for (int k = 0; k < chain; k++) hipLaunchKernelGGL(k_step, dim3(512), dim3(256), 0, s, d, n, 1.0f, spin);
        CK(hipStreamSynchronize(s));
        auto t0 = std::chrono::steady_clock::now();
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < reps; r++) for (int k = 0; k < chain; k++) hipLaunchKernelGGL(k_step, dim3(512), dim3(256), 0, s, d, n, float(r), spin);
        auto t1 = std::chrono::steady_clock::now();
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        std::printf("spin %4d  launches: host %.1f us per chain, device %.1f us per chain\n", spin, std::chrono::duration<double, std::micro>(t1 - t0).count() / reps, ms * 1e3 / reps);
        // (b) the chain as a graph, the first node's parameters updated before every launch
        hipGraph_t g; CK(hipGraphCreate(&g, 0));
        std::vector<hipGraphNode_t> nodes(chain);
        float add = 1.0f; int nn = n, sp = spin; float* dd = d;
        void* args[] = {&dd, &nn, &add, &sp};
        hipKernelNodeParams p{}; p.func = reinterpret_cast<void*>(k_step); p.gridDim = dim3(512); p.blockDim = dim3(256); p.sharedMemBytes = 0; p.kernelParams = args; p.extra = nullptr;
        for (int k = 0; k < chain; k++) CK(hipGraphAddKernelNode(&nodes[k], g, k ? &nodes[k - 1] : nullptr, k ? 1 : 0, &p));
        hipGraphExec_t ge; CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 50; w++) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        t0 = std::chrono::steady_clock::now();
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < reps; r++) { add = float(r); CK(hipGraphExecKernelNodeSetParams(ge, nodes[0], &p)); CK(hipGraphLaunch(ge, s)); }
        t1 = std::chrono::steady_clock::now();
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::printf("spin %4d  graph:    host %.1f us per chain, device %.1f us per chain\n", spin, std::chrono::duration<double, std::micro>(t1 - t0).count() / reps, ms * 1e3 / reps);
        // (c) every node's parameters updated before every launch (what a frame's chain needs without a parameter block: pointer, pose,
        //     sequence number and table change from frame to frame)
        t0 = std::chrono::steady_clock::now();
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < reps; r++) { add = float(r); for (int k = 0; k < chain; k++) CK(hipGraphExecKernelNodeSetParams(ge, nodes[k], &p)); CK(hipGraphLaunch(ge, s)); }
        t1 = std::chrono::steady_clock::now();
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::printf("spin %4d  graph, all %d nodes updated: host %.1f us per chain, device %.1f us per chain\n", spin, chain, std::chrono::duration<double, std::micro>(t1 - t0).count() / reps, ms * 1e3 / reps);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
