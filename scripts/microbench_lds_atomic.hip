// LDS atomic throughput on gfx950, the shapes the tiled update uses: one 64-bit add per lane into a 14^3-cell tile.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_atomic scripts/microbench_lds_atomic.hip && /tmp/lds_atomic
// Prints LDS cycles per wave-instruction per CU (all CUs busy, W waves per CU) for: u64 add, u32 add, two u32 adds, f32 add, u64 add with
// return; addresses random over the tile, a lane stride of one cell (conflict-free), or all lanes of a row (same-address groups).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kCells = 14 * 14 * 14;
constexpr int kIters = 2000;

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int kOp, int kPattern>
__global__ void k(unsigned long long* out, unsigned long long* cycles, int active_mod) {
    extern __shared__ unsigned long long tile[];
    for (int i = threadIdx.x; i < kCells * int(blockDim.x / 256 ? 1 : 1); i += blockDim.x) tile[i] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t s = mix(blockIdx.x * 1024u + threadIdx.x);
    const bool on = (lane % active_mod) == 0;
    unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long acc = 0;
    for (int it = 0; it < kIters; it++) {
        s = s * 1664525u + 1013904223u;
        uint32_t cell;
        if (kPattern == 0) cell = (s >> 8) % kCells;                       // random
        else if (kPattern == 1) cell = (lane + 64 * ((s >> 8) % 40)) % kCells;   // lane-strided: conflict-free
        else cell = (((s >> 8) % 2000) & ~7u) + (lane & 7u);               // groups of 8 lanes on 8 neighbouring cells (same-address pressure)
        if (on) {
            if (kOp == 0) __hip_atomic_fetch_add(&tile[cell], (1ull << 44) | (s & 0xffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (kOp == 1) __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(tile) + cell, s & 0xffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (kOp == 2) {
                __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(tile) + 2 * cell, s & 0xffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(tile) + 2 * cell + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else if (kOp == 3) __hip_atomic_fetch_add(reinterpret_cast<float*>(tile) + cell, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (kOp == 4) acc += __hip_atomic_fetch_add(&tile[cell], (1ull << 44) | (s & 0xffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (kOp == 5) acc += tile[cell];                          // plain 64-bit read
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cycles[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
    if (acc == 0x1234567) out[0] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[1 + (blockIdx.x & 7)] = tile[3];
}

template <int kOp, int kPattern>
void run(const char* name, int waves_per_cu, int wgs_per_cu, int active_mod, unsigned long long* d_out, unsigned long long* d_cyc) {
    const int threads = waves_per_cu / wgs_per_cu * 64, blocks = 256 * wgs_per_cu;
    const size_t lds = 21968;
    hipLaunchKernelGGL((k<kOp, kPattern>), dim3(blocks), dim3(threads), lds, 0, d_out, d_cyc, active_mod);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((k<kOp, kPattern>), dim3(blocks), dim3(threads), lds, 0, d_out, d_cyc, active_mod);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> c(size_t(blocks) * (threads / 64));
    hipMemcpy(c.data(), d_cyc, c.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : c) mean += double(v); mean /= double(c.size());
    // one CU issues waves_per_cu * kIters wave-instructions (x2 for kOp 2) during `mean` cycles of each wave's life (all concurrent)
    const double instr = double(waves_per_cu) * kIters * (kOp == 2 ? 2 : 1);
    std::printf("%-34s waves/CU %2d (%d WG) lanes 1/%d : %8.1f us, %9.0f cycles per wave, %6.2f cycles per wave-instruction per CU\n", name, waves_per_cu, wgs_per_cu, active_mod, ms * 1e3, mean, mean / instr);
}

int main() {
    unsigned long long *d_out, *d_cyc;
    hipMalloc(&d_out, 128); hipMalloc(&d_cyc, 8 * 8192 * 4);
    hipMemset(d_out, 0, 128);
    for (int w : {4, 8, 16, 20}) {
        const int wg = w >= 20 ? 5 : (w >= 8 ? w / 4 : 1);
        run<0, 0>("u64 add, random cells", w, wg, 1, d_out, d_cyc);
    }
    run<0, 1>("u64 add, lane-strided", 16, 4, 1, d_out, d_cyc);
    run<0, 2>("u64 add, 8-lane groups", 16, 4, 1, d_out, d_cyc);
    run<0, 0>("u64 add, random, half the lanes", 16, 4, 2, d_out, d_cyc);
    run<0, 0>("u64 add, random, quarter", 16, 4, 4, d_out, d_cyc);
    run<1, 0>("u32 add, random cells", 16, 4, 1, d_out, d_cyc);
    run<1, 1>("u32 add, lane-strided", 16, 4, 1, d_out, d_cyc);
    run<2, 0>("two u32 adds, random cells", 16, 4, 1, d_out, d_cyc);
    run<3, 0>("f32 add, random cells", 16, 4, 1, d_out, d_cyc);
    run<4, 0>("u64 add with return, random", 16, 4, 1, d_out, d_cyc);
    run<5, 0>("u64 read, random", 16, 4, 1, d_out, d_cyc);
    run<5, 1>("u64 read, lane-strided", 16, 4, 1, d_out, d_cyc);
    return 0;
}
