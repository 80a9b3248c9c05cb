// Micro-benchmark: how fast does the chip start the workgroups of a full-occupancy launch?  Each workgroup stamps wall_clock64 (10 ns ticks) at its start.
// Build + run on the GPU box: /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o scratch/ramp scripts/microbench_ramp.hip && ./scratch/ramp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int kRegs>
__global__ void __launch_bounds__(256) k_stamp(unsigned long long* start, unsigned long long* stop, int spin, int sleep_ticks) {
    extern __shared__ unsigned char lds[];
    const unsigned long long t0 = wall_clock64();
    // staggered-start experiment: every wave idles for a fixed time first, so that the dispatcher places the whole grid while the VALUs are quiet
    if (sleep_ticks) while (wall_clock64() - t0 < (unsigned long long)sleep_ticks) __builtin_amdgcn_s_sleep(4);
    float acc[kRegs];
    for (int i = 0; i < kRegs; i++) acc[i] = float(threadIdx.x + i);
    for (int k = 0; k < spin; k++)
        for (int i = 0; i < kRegs; i++) acc[i] = acc[i] * 1.0001f + 0.5f;
    float s = 0; for (int i = 0; i < kRegs; i++) s += acc[i];
    if (s == 12345.678f) lds[threadIdx.x] = 1;
    if (threadIdx.x == 0) { start[blockIdx.x] = t0; stop[blockIdx.x] = wall_clock64(); }
}
template <int kRegs> void run(int wgs, size_t lds, int spin, const char* name, int sleep_ticks = 0) {
    unsigned long long *d0, *d1;
    hipMalloc(&d0, wgs * 8); hipMalloc(&d1, wgs * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_stamp<kRegs>, dim3(wgs), dim3(256), lds, 0, d0, d1, spin, sleep_ticks);
        hipEventRecord(e1); hipDeviceSynchronize();
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> a(wgs), b(wgs);
    hipMemcpy(a.data(), d0, wgs * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, wgs * 8, hipMemcpyDeviceToHost);
    unsigned long long first = *std::min_element(a.begin(), a.end()), last_end = *std::max_element(b.begin(), b.end());
    std::vector<unsigned long long> rel(wgs); for (int i = 0; i < wgs; i++) rel[i] = a[i] - first;
    std::sort(rel.begin(), rel.end());
    printf("%-28s wgs %5d lds %6zu: start p50 %6llu p90 %6llu max %6llu ticks, span %llu ticks, events %.1f us\n", name, wgs, lds, rel[wgs / 2], rel[wgs * 9 / 10], rel[wgs - 1],
           last_end - first, ms * 1e3);
    hipFree(d0); hipFree(d1);
}
int main() {
    // __builtin_readcyclecounter = s_memtime (100 MHz constant clock on gfx9): 1 tick = 10 ns
    run<16>(1280, 0, 0, "16 regs, no lds");
    run<16>(1280, 21952, 0, "16 regs, 21 KB lds");
    run<80>(1280, 21952, 0, "~96 regs, 21 KB lds");
    run<80>(1280, 21952, 40, "~96 regs, 21 KB, 20 us work");
    run<80>(640, 21952, 40, "640 wgs");
    run<16>(5120, 0, 0, "5120 wgs small");
    run<80>(1280, 21952, 40, "20 us work, sleep 1 us", 100);
    run<80>(1280, 21952, 40, "20 us work, sleep 2 us", 200);
    run<80>(1280, 21952, 40, "20 us work, sleep 3 us", 300);
    run<80>(1280, 21952, 40, "20 us work, sleep 5 us", 500);
    run<80>(1280, 7200, 40, "7 KB lds, 20 us work", 0);
    run<80>(1280, 7200, 40, "7 KB lds, sleep 2 us", 200);
    run<80>(2560, 0, 40, "2560 wgs (2 rounds), 20 us", 0);
    return 0;
}
