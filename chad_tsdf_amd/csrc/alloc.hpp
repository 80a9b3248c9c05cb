// alloc.hpp -- every device allocation of the library goes through here: metered (how many hipMalloc / hipFree calls, how many bytes,
// how long the host sat in them -- chad_map_pipeline_times reports the numbers, so that a measurement can show that its timed region
// allocated nothing), and, for the DAG's level stores, carved out of ONE up-front reservation (DevPool).
//
// Why: a hipMalloc / hipFree of hundreds of MB to several GB costs between 2 and 200+ ms depending on the box (page-table set-up and
// clearing of VRAM that has not been handed out before; a hipFree also waits for the whole device).  The reference's counterpart is
// VirtualArray (virtual_array.hpp:12-104): 4 GiB of address space reserved per store, committed on first touch -- storage that grows
// without ever moving or stalling.  Here: chad_options.dag_pool_bytes reserves the DAG's memory when the map is created, and a level that
// grows takes its new store from the pool and gives the old one back (stream order makes the hand-over safe: every access to a level
// is enqueued on the fold's stream, the copy / rehash out of the old store included).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstddef>
#include <cstdint>
#include <iterator>
#include <map>

#include "range_pool.hpp"

namespace chadhip {

struct AllocMeter {   // per host thread (a map is driven by one thread at a time: include/chad_hip.h)
    uint64_t mallocs = 0, frees = 0, bytes = 0;
    double seconds = 0;
};
inline AllocMeter& alloc_meter() {
    static thread_local AllocMeter m;
    return m;
}

template <class T>
inline hipError_t chad_malloc(T** p, size_t bytes) {
    const auto t0 = std::chrono::steady_clock::now();
    const hipError_t e = hipMalloc(reinterpret_cast<void**>(p), bytes);
    AllocMeter& m = alloc_meter();
    m.seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    m.mallocs++;
    if (e == hipSuccess) m.bytes += bytes;
    return e;
}
inline hipError_t chad_free(void* p) {
    if (!p) return hipSuccess;
    const auto t0 = std::chrono::steady_clock::now();
    const hipError_t e = hipFree(p);
    AllocMeter& m = alloc_meter();
    m.seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    m.frees++;
    return e;
}

// One reservation, handed out in pieces: the bookkeeping is RangePool's (range_pool.hpp: plain C++, checked on the host under the
// sanitizers by tests/test_range_pool_host.py); this adds the device memory behind it.
struct DevPool : RangePool {
    char* base = nullptr;

    hipError_t reserve(size_t bytes) {
        release();
        if (!bytes) return hipSuccess;
        bytes = (bytes + kAlign - 1) & ~(kAlign - 1);
        const hipError_t e = chad_malloc(&base, bytes);
        if (e != hipSuccess) { base = nullptr; return e; }
        init(bytes);
        return hipSuccess;
    }
    void release() {
        if (base) (void)chad_free(base);
        base = nullptr;
        init(0);
    }
    bool owns(const void* p) const { return base && static_cast<const char*>(p) >= base && static_cast<const char*>(p) < base + cap; }
    void* take(size_t bytes) {   // nullptr: no range of that length left
        if (!base) return nullptr;
        const size_t at = take_range(bytes);
        return at == kNone ? nullptr : base + at;
    }
    // A range goes back -- and may be handed out again at once: the caller guarantees that every access to it was enqueued on ONE stream
    // and that the next user's accesses are enqueued on that stream too (DeviceDag::store_alloc waits for an event when a level is first
    // touched from another stream).  false: `p` is not a range this pool handed out -- nothing is changed (the caller reports it; a double
    // release or a foreign pointer would otherwise corrupt the free list silently).
    bool give(void* p, size_t bytes) { return owns(p) && give_range(size_t(static_cast<char*>(p) - base), bytes); }
};

}  // namespace chadhip
