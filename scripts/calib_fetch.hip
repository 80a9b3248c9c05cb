// calib_fetch.hip -- what rocprofv3's FETCH_SIZE reports for the update kernel's own load patterns, on a known byte count.
// MI355X_MICROARCH.md ("HBM"): on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced stream (16 B per lane, consecutive
// lanes consecutive addresses) and "other access widths are uncalibrated: calibrate on a known byte count in your own access
// pattern".  k_update_tiled reads its points and normals as float4 with a lane stride of the item's chunk count (lane l of chunk c
// reads element l * C + c: tile_update.hpp, the walk), so that the points of one voxel end up in different chunks.  Three kernels
// over 256 MiB each, every element read exactly once:
//   k_coalesced        element = global thread index                                  (the guide's calibration case: expect 0.5)
//   k_lane_strided<C>  wave w owns 64 * C consecutive elements, lane l reads l * C + c  (the update kernel's pattern, C = 4, 8, 16)
// Run:  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/calib -- /tmp/calib_fetch
// and divide each kernel's FETCH_SIZE (KiB) by 262144.   hipcc --offload-arch=gfx950 -O3 -o /tmp/calib_fetch scripts/calib_fetch.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k_coalesced(const float4* __restrict__ in, float* __restrict__ out, size_t n) {
    float acc = 0;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) { const float4 v = in[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) out[0] = acc;
}
template <int C>
__global__ void k_lane_strided(const float4* __restrict__ in, float* __restrict__ out, size_t n) {
    const size_t lane = threadIdx.x & 63, wave = (size_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6, waves = (size_t(gridDim.x) * blockDim.x) >> 6;
    float acc = 0;
    for (size_t base = wave * 64 * C; base + 64 * C <= n; base += waves * 64 * C)
        for (int c = 0; c < C; c++) { const float4 v = in[base + lane * C + c]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) out[0] = acc;
}

int main() {
    const size_t n = (size_t(256) << 20) / 16;   // a buffer of its own per kernel, read once: nothing comes from the Infinity Cache
    float4* in[4]; float* out;
    for (auto& p : in) { if (hipMalloc(&p, n * 16) != hipSuccess) return 1; hipMemset(p, 0, n * 16); }
    if (hipMalloc(&out, 4) != hipSuccess) return 1;
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k_coalesced, dim3(2048), dim3(256), 0, 0, in[0], out, n);
    hipLaunchKernelGGL(k_lane_strided<4>, dim3(2048), dim3(256), 0, 0, in[1], out, n);
    hipLaunchKernelGGL(k_lane_strided<8>, dim3(2048), dim3(256), 0, 0, in[2], out, n);
    hipLaunchKernelGGL(k_lane_strided<16>, dim3(2048), dim3(256), 0, 0, in[3], out, n);
    hipDeviceSynchronize();
    std::printf("done: each kernel read %zu MiB\n", (n * 16) >> 20);
    return 0;
}
