// exchange_plan.hpp -- the host arithmetic of the sparse multi-GPU exchange (exchange_kernels.hpp): which Morton key ranges the ranks own,
// and who sends how many bytes to whom.  Plain functions of words that every rank holds alike after an all-gather, so every rank
// derives the same picture without further communication.  No HIP in here: the stage entries chad_exchange_splitters /
// chad_exchange_layout (include/chad_hip.h) expose them to the CPU tests, which run them on two gloo ranks.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <cstddef>
#include <utility>
#include <vector>

namespace chadhip {

constexpr uint32_t kXSamples = 64;            // sample keys per rank the splitters are drawn from
constexpr uint64_t kXMaxRankBlocks = 1ull << 30;   // blocks one rank may announce for one range, and for all ranges together (a table of 2^30 blocks would be 1.4 TB)
constexpr uint32_t kXHeadWords = kXSamples + 1;   // a rank's head: [block count (low 32 bits) | payload width flag (bit 32)], then its samples

inline size_t xchg_pad8_host(size_t bytes) { return (bytes + 7) & ~size_t(7); }
// a slice source -> destination: keys u64[blocks] | masks u64[blocks] | first u32[blocks] (padded to 8 bytes) | payload u64[words * width]
inline size_t xchg_slice_bytes(size_t blocks, size_t words, uint32_t width) { return 16 * blocks + xchg_pad8_host(4 * blocks) + 8 * words * width; }
// a range on its way to rank 0: keys u64[U] | first u32[U] (padded) | masks u8[U] (padded) | tsdf words u64[W] | weight words u64[W] (optional)
inline size_t xchg_range_bytes(size_t blocks, size_t words, bool weights) { return 8 * blocks + xchg_pad8_host(4 * blocks) + xchg_pad8_host(blocks) + 8 * words * (weights ? 2 : 1); }

// The world - 1 splitters: weighted quantiles of all ranks' samples (sample i of rank r is key (i + 1) n_r / 65 of its sorted list and
// stands for n_r / 65 of its blocks).  Range d = keys in [split[d - 1], split[d]), split[-1] = 0, split[world - 1] = infinity.
// false: a head holds an implausible count.
inline bool xchg_splitters(const uint64_t* heads /* [world][kXHeadWords] */, uint32_t world, uint64_t* split /* [world - 1] */) {
    std::vector<std::pair<uint64_t, double>> samples;
    double total = 0;
    for (uint32_t r = 0; r < world; r++) {
        const uint32_t n_r = uint32_t(heads[size_t(r) * kXHeadWords]);
        if (n_r >= (1u << 30)) return false;
        if (!n_r) continue;
        for (uint32_t i = 0; i < kXSamples; i++) samples.emplace_back(heads[size_t(r) * kXHeadWords + 1 + i], double(n_r) / double(kXSamples + 1));
        total += double(n_r) * double(kXSamples) / double(kXSamples + 1);
    }
    std::stable_sort(samples.begin(), samples.end(), [](const std::pair<uint64_t, double>& a, const std::pair<uint64_t, double>& b) { return a.first < b.first; });
    double cum = 0;
    size_t at = 0;
    for (uint32_t j = 1; j < world; j++) {
        const double target = total * double(j) / double(world);
        while (at < samples.size() && cum + samples[at].second < target) cum += samples[at++].second;
        split[j - 1] = at < samples.size() ? samples[at].first : ~0ull;
    }
    return true;
}

// Who sends what to whom in one point-to-point step: byte ranges of this rank's send and receive buffers per peer -- and, for the
// callback transport, where this rank's slice lies in every PEER's send buffer and how long the longest send buffer of the job is
// (that transport gathers whole send buffers: a rehearsal path, not the product's).
struct ExchangePlan {
    static constexpr int kMax = 16;
    size_t send_off[kMax] = {}, send_bytes[kMax] = {}, recv_off[kMax] = {}, recv_bytes[kMax] = {}, peer_off[kMax] = {};
    size_t longest_send = 0;
    size_t send_total() const { size_t b = 0; for (int r = 0; r < kMax; r++) b = std::max(b, send_off[r] + send_bytes[r]); return b; }
    size_t recv_total() const { size_t b = 0; for (int r = 0; r < kMax; r++) b = std::max(b, recv_off[r] + recv_bytes[r]); return b; }
    size_t bytes_out(int rank, int world) const { size_t b = 0; for (int r = 0; r < world; r++) if (r != rank) b += send_bytes[r]; return b; }
};

// The slice exchange's plan for rank `me` from the G x G matrix (matrix[(a * world + b) * 2] = blocks, + 1 = payload words rank a holds
// for rank b's range) and the ranks' payload widths.  false: an implausible entry.
inline bool xchg_slice_plan(const uint64_t* matrix, const uint32_t* wide_of, uint32_t world, uint32_t me, ExchangePlan& plan) {
    auto slice = [&](uint32_t from, uint32_t to) { return xchg_slice_bytes(matrix[(size_t(from) * world + to) * 2], matrix[(size_t(from) * world + to) * 2 + 1], wide_of[from] ? 2u : 1u); };
    plan = ExchangePlan{};
    // every entry of the matrix, not only this rank's column: the other entries come from the peers and decide offsets (peer_off) and, for
    // the callback transport, a host allocation (longest_send).  A rank's blocks for all ranges together are the blocks of its table.
    for (uint32_t a = 0; a < world; a++) {
        uint64_t row_blocks = 0;
        for (uint32_t b = 0; b < world; b++) {
            const uint64_t blocks = matrix[(size_t(a) * world + b) * 2], words = matrix[(size_t(a) * world + b) * 2 + 1];
            if (blocks >= kXMaxRankBlocks || words > blocks * 64) return false;
            row_blocks += blocks;
        }
        if (row_blocks >= kXMaxRankBlocks) return false;
    }
    size_t send_total = 0, recv_total = 0;
    for (uint32_t r = 0; r < world; r++) {
        plan.send_off[r] = send_total; plan.send_bytes[r] = slice(me, r); send_total += plan.send_bytes[r];
        plan.recv_off[r] = recv_total; plan.recv_bytes[r] = slice(r, me); recv_total += plan.recv_bytes[r];
        size_t before = 0, whole = 0;
        for (uint32_t d = 0; d < world; d++) { if (d < me) before += slice(r, d); whole += slice(r, d); }   // (bounded: kXMaxRankBlocks * 1044 bytes per rank, far from wrapping)
        plan.peer_off[r] = before; plan.longest_send = std::max(plan.longest_send, whole);
    }
    return true;
}

// ... and for the ranges' way to rank 0 (sizes[2 r] = blocks, [2 r + 1] = words of rank r's range)
inline bool xchg_gather_plan(const uint64_t* sizes, uint32_t world, uint32_t me, bool weights, ExchangePlan& plan, uint64_t* total_blocks) {
    plan = ExchangePlan{};
    size_t recv_total = 0;
    *total_blocks = 0;
    for (uint32_t r = 0; r < world; r++) {
        const uint64_t u = sizes[2 * r], w = sizes[2 * r + 1];
        if (u >= (1ull << 30) || w > u * 8) return false;
        *total_blocks += u;
        if (me == 0) { plan.recv_off[r] = recv_total; plan.recv_bytes[r] = xchg_range_bytes(u, w, weights); recv_total += plan.recv_bytes[r]; }
        plan.longest_send = std::max(plan.longest_send, xchg_range_bytes(u, w, weights));
    }
    plan.send_bytes[0] = xchg_range_bytes(sizes[2 * me], sizes[2 * me + 1], weights);   // (everything goes to rank 0; a peer's slice for rank 0 starts its buffer: peer_off = 0)
    return true;
}

}  // namespace chadhip
