// Test harness for the range bookkeeping of the DAG's device pool (chad_tsdf_amd/csrc/range_pool.hpp: no HIP in it): a C face so that the
// CPU suite -- and `make asan` -- can drive it against a model.
#include <cstdint>

#include "../../chad_tsdf_amd/csrc/range_pool.hpp"

using namespace chadhip;

extern "C" {
void* poolcheck_new(uint64_t bytes) { RangePool* p = new RangePool; p->init(size_t(bytes)); return p; }
void poolcheck_delete(void* p) { delete static_cast<RangePool*>(p); }
uint64_t poolcheck_take(void* p, uint64_t bytes) { return uint64_t(static_cast<RangePool*>(p)->take_range(size_t(bytes))); }   // ~0: none left
int poolcheck_give(void* p, uint64_t at, uint64_t bytes) { return static_cast<RangePool*>(p)->give_range(size_t(at), size_t(bytes)) ? 1 : 0; }
void poolcheck_reset(void* p) { static_cast<RangePool*>(p)->reset(); }
// out: used, peak, number of free ranges, then (offset, length) of the first `room` of them
void poolcheck_state(void* p, uint64_t* out, uint32_t room) {
    const RangePool& r = *static_cast<RangePool*>(p);
    out[0] = r.used; out[1] = r.peak; out[2] = r.free_ranges.size();
    uint32_t k = 0;
    for (auto it = r.free_ranges.begin(); it != r.free_ranges.end() && k < room; ++it, ++k) { out[3 + 2 * k] = it->first; out[4 + 2 * k] = it->second; }
}
}
