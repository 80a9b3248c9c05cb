// What does growing a store cost the host?  hipMalloc / hipFree of N bytes against the virtual-memory route: physical chunks created once
// (hipMemCreate), an address range reserved once (hipMemAddressReserve), growth = hipMemMap + hipMemSetAccess of further chunks.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/microbench_vmm scripts/microbench_vmm.hip && /tmp/microbench_vmm
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    int dev = 0, vmm = 0;
    CHECK(hipGetDevice(&dev));
    (void)hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, dev);
    std::printf("virtual memory management supported: %d\n", vmm);
    for (size_t gb : {1, 4, 12}) {
        const size_t bytes = gb << 30;
        void* p = nullptr;
        double t0 = now_ms();
        CHECK(hipMalloc(&p, bytes));
        double t1 = now_ms();
        CHECK(hipMemset(p, 0, 4096));
        CHECK(hipDeviceSynchronize());
        double t2 = now_ms();
        CHECK(hipFree(p));
        double t3 = now_ms();
        std::printf("hipMalloc %2zu GB: %8.3f ms   hipFree: %8.3f ms\n", gb, t1 - t0, t3 - t2);
    }
    if (!vmm) return 0;
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    CHECK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    std::printf("granularity %zu KB\n", gran >> 10);
    const size_t chunk = size_t(256) << 20, n_chunks = 48;   // 12 GB in 256 MB chunks
    std::vector<hipMemGenericAllocationHandle_t> h(n_chunks);
    double t0 = now_ms();
    for (size_t k = 0; k < n_chunks; k++) CHECK(hipMemCreate(&h[k], chunk, &prop, 0));
    double t1 = now_ms();
    std::printf("hipMemCreate  %zu x 256 MB: %8.3f ms (%.3f per chunk)\n", n_chunks, t1 - t0, (t1 - t0) / n_chunks);
    void* va = nullptr;
    t0 = now_ms();
    CHECK(hipMemAddressReserve(&va, size_t(64) << 30, gran, nullptr, 0));
    t1 = now_ms();
    std::printf("hipMemAddressReserve 64 GB: %8.3f ms\n", t1 - t0);
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    double map_ms = 0, acc_ms = 0;
    for (size_t k = 0; k < n_chunks; k++) {
        double a = now_ms();
        CHECK(hipMemMap(static_cast<char*>(va) + k * chunk, chunk, 0, h[k], 0));
        double b = now_ms();
        CHECK(hipMemSetAccess(static_cast<char*>(va) + k * chunk, chunk, &acc, 1));
        double c = now_ms();
        map_ms += b - a; acc_ms += c - b;
    }
    std::printf("growth by one chunk: hipMemMap %.3f ms + hipMemSetAccess %.3f ms (means over %zu)\n", map_ms / n_chunks, acc_ms / n_chunks, n_chunks);
    CHECK(hipMemset(va, 1, n_chunks * chunk));
    CHECK(hipDeviceSynchronize());
    t0 = now_ms();
    CHECK(hipMemUnmap(va, n_chunks * chunk));
    for (auto& x : h) CHECK(hipMemRelease(x));
    CHECK(hipMemAddressFree(va, size_t(64) << 30));
    t1 = now_ms();
    std::printf("unmap + release + free: %8.3f ms\n", t1 - t0);
    return 0;
}
