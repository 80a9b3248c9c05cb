// microbench_h2d.hip -- what a 12 MB frame (1 M points x 12 B) costs on its way from host memory into HBM, by path:
// the ceiling (pinned -> device), a direct copy from pageable memory, a registered (page-locked in place) buffer, and the
// host-side copy pageable -> pinned with 1..8 threads.  Feeds bench.py's pcie_inclusive.frac_of_h2d_ceiling and the choice of
// chad_map_insert's staging strategy (chad_hip.hip).   hipcc -O2 -o /tmp/h2d scripts/microbench_h2d.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const size_t bytes = argc > 1 ? size_t(std::atol(argv[1])) : size_t(12) << 20;
    const int reps = 40;
    char* dev = nullptr; CK(hipMalloc(&dev, bytes * 4));
    char* pinned = nullptr; CK(hipHostMalloc(&pinned, bytes * 4, hipHostMallocDefault));
    // four distinct pageable source buffers, cycled (a caller's frames are not the same lines every time)
    std::vector<char*> pageable;
    for (int k = 0; k < 4; k++) { char* p = static_cast<char*>(std::aligned_alloc(4096, bytes)); std::memset(p, k + 1, bytes); pageable.push_back(p); }
    std::memset(pinned, 1, bytes * 4);
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto timed = [&](const char* name, auto&& body) {
        body(0); CK(hipStreamSynchronize(s));
        const double t0 = now_ms();
        for (int r = 0; r < reps; r++) body(r);
        CK(hipStreamSynchronize(s));
        const double ms = (now_ms() - t0) / reps;
        std::printf("%-58s %.3f ms per %.1f MB = %.1f GB/s\n", name, ms, bytes / 1e6, bytes / ms / 1e6);
        return 0;
    };
    timed("pinned -> device, one hipMemcpyAsync (the ceiling)", [&](int r) { (void)hipMemcpyAsync(dev + (r & 3) * bytes, pinned + (r & 3) * bytes, bytes, hipMemcpyHostToDevice, s); });
    timed("pinned -> device, 2 MiB chunks", [&](int r) { for (size_t at = 0; at < bytes; at += size_t(2) << 20) (void)hipMemcpyAsync(dev + (r & 3) * bytes + at, pinned + (r & 3) * bytes + at, std::min(size_t(2) << 20, bytes - at), hipMemcpyHostToDevice, s); });
    timed("pageable -> device, hipMemcpyAsync direct", [&](int r) { (void)hipMemcpyAsync(dev + (r & 3) * bytes, pageable[r & 3], bytes, hipMemcpyHostToDevice, s); });
    timed("pageable -> device, hipMemcpy (synchronous)", [&](int r) { (void)hipMemcpy(dev + (r & 3) * bytes, pageable[r & 3], bytes, hipMemcpyHostToDevice); });
    {   // register + copy + unregister per frame (a caller's buffer seen once), and registered once (a buffer seen again)
        timed("pageable: hipHostRegister + copy + unregister per frame", [&](int r) {
            char* p = pageable[r & 3];
            if (hipHostRegister(p, bytes, hipHostRegisterDefault) != hipSuccess) return;
            (void)hipMemcpyAsync(dev + (r & 3) * bytes, p, bytes, hipMemcpyHostToDevice, s);
            (void)hipStreamSynchronize(s);
            (void)hipHostUnregister(p);
        });
        for (char* p : pageable) CK(hipHostRegister(p, bytes, hipHostRegisterDefault));
        timed("pageable registered once -> device", [&](int r) { (void)hipMemcpyAsync(dev + (r & 3) * bytes, pageable[r & 3], bytes, hipMemcpyHostToDevice, s); });
        for (char* p : pageable) CK(hipHostUnregister(p));
    }
    for (int threads : {1, 2, 4, 8}) {
        char name[96]; std::snprintf(name, sizeof name, "host memcpy pageable -> pinned, %d thread(s), no DMA", threads);
        timed(name, [&](int r) {
            std::vector<std::thread> ts;
            const size_t share = (bytes / threads + 4095) & ~size_t(4095);
            for (int t = 0; t < threads; t++) ts.emplace_back([&, t] { const size_t at = t * share; if (at < bytes) std::memcpy(pinned + (r & 3) * bytes + at, pageable[r & 3] + at, std::min(share, bytes - at)); });
            for (auto& t : ts) t.join();
        });
    }
    for (int threads : {1, 2, 4, 8}) {   // staged: memcpy chunks into pinned memory, each chunk's DMA issued as soon as it is in place
        char name[96]; std::snprintf(name, sizeof name, "staged pageable -> pinned -> device, %d thread(s), 1 MiB chunks", threads);
        timed(name, [&](int r) {
            const size_t chunk = size_t(1) << 20, chunks = (bytes + chunk - 1) / chunk;
            std::vector<std::thread> ts;
            std::vector<char> done(chunks, 0);
            volatile char* flags = done.data();
            for (int t = 1; t < threads; t++) ts.emplace_back([&, t] { for (size_t c = t; c < chunks; c += threads) { std::memcpy(pinned + (r & 3) * bytes + c * chunk, pageable[r & 3] + c * chunk, std::min(chunk, bytes - c * chunk)); __atomic_store_n(&flags[c], 1, __ATOMIC_RELEASE); } });
            for (size_t c = 0; c < chunks; c++) {
                if (c % threads == 0) std::memcpy(pinned + (r & 3) * bytes + c * chunk, pageable[r & 3] + c * chunk, std::min(chunk, bytes - c * chunk));
                else while (!__atomic_load_n(&flags[c], __ATOMIC_ACQUIRE)) {}
                (void)hipMemcpyAsync(dev + (r & 3) * bytes + c * chunk, pinned + (r & 3) * bytes + c * chunk, std::min(chunk, bytes - c * chunk), hipMemcpyHostToDevice, s);
            }
            for (auto& t : ts) t.join();
        });
    }
    std::printf("host threads available: %u\n", std::thread::hardware_concurrency());
    return 0;
}
