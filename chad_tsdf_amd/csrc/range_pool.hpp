// range_pool.hpp -- the bookkeeping of DevPool (alloc.hpp) without the device memory: offsets into one reservation, handed out in pieces.
// Best fit over a list of free ranges ordered by offset, neighbours merged on release.  A few hundred allocations over a map's life
// (twenty-one levels, each doubling a dozen times): the list stays short.  Plain C++ (no HIP): tests/cpp/pool_check.cpp drives it on the
// host, under AddressSanitizer / UBSan with `make asan`.
#pragma once
#include <algorithm>
#include <cstddef>
#include <iterator>
#include <map>

namespace chadhip {

struct RangePool {
    size_t cap = 0, used = 0, peak = 0;
    std::map<size_t, size_t> free_ranges;   // offset -> length
    static constexpr size_t kAlign = 4096;
    static constexpr size_t kNone = ~size_t(0);

    static size_t rounded(size_t bytes) { return (std::max<size_t>(bytes, 1) + kAlign - 1) & ~(kAlign - 1); }
    void init(size_t bytes) {   // a reservation of `bytes` (a multiple of kAlign), all of it free; 0: none
        cap = bytes; used = peak = 0;
        free_ranges.clear();
        if (bytes) free_ranges[0] = bytes;
    }
    size_t take_range(size_t bytes) {   // offset of a range of that length, kNone: none left
        bytes = rounded(bytes);
        auto best = free_ranges.end();   // best fit: small requests fill small holes, the large ranges stay whole for the levels that grow into gigabytes
        for (auto it = free_ranges.begin(); it != free_ranges.end(); ++it)
            if (it->second >= bytes && (best == free_ranges.end() || it->second < best->second)) best = it;
        if (best == free_ranges.end()) return kNone;
        const size_t at = best->first, len = best->second;
        free_ranges.erase(best);
        used += bytes; peak = std::max(peak, used);
        if (bytes >= cap / 32) {   // the few large stores are carved from the top of their range, the many small ones from the bottom: they meet last
            if (len > bytes) free_ranges[at] = len - bytes;
            return at + (len - bytes);
        }
        if (len > bytes) free_ranges[at + bytes] = len - bytes;
        return at;
    }
    // false -- and nothing changed -- unless [at, at + bytes) is aligned, inside the reservation and entirely handed out (a double release
    // or a length that reaches into a free neighbour would otherwise corrupt the list silently)
    bool give_range(size_t at, size_t bytes) {
        bytes = rounded(bytes);
        size_t len = bytes;
        if (at % kAlign || at >= cap || bytes > cap - at || bytes > used) return false;
        auto next = free_ranges.lower_bound(at);
        if (next != free_ranges.end() && next->first < at + bytes) return false;                                             // overlaps the free range behind it
        if (next != free_ranges.begin()) { auto prev = std::prev(next); if (prev->first + prev->second > at) return false; }   // ... or the one in front: already free
        used -= bytes;
        if (next != free_ranges.begin()) {
            auto prev = std::prev(next);
            if (prev->first + prev->second == at) { at = prev->first; len += prev->second; free_ranges.erase(prev); }
        }
        if (next != free_ranges.end() && at + len == next->first) { len += next->second; free_ranges.erase(next); }
        free_ranges[at] = len;
        return true;
    }
    void reset() {   // everything handed out is dead (the DAG was thrown away)
        free_ranges.clear();
        if (cap) free_ranges[0] = cap;
        used = 0;
    }
};

}  // namespace chadhip
