// Test harness for the host arithmetic of the sparse multi-GPU exchange (chad_tsdf_amd/csrc/exchange_plan.hpp: no HIP in it): a C face
// so that the CPU suite -- and `make asan` -- can drive it with random heads and matrices for any number of ranks.
#include "../../chad_tsdf_amd/csrc/exchange_plan.hpp"

using namespace chadhip;

extern "C" {
int plancheck_splitters(const uint64_t* heads, uint32_t world, uint64_t* split) { return xchg_splitters(heads, world, split) ? 0 : 1; }
// out[5 r .. 5 r + 4] = send offset, send bytes, receive offset, receive bytes, peer offset; out[5 world] = longest send buffer
int plancheck_slices(const uint64_t* matrix, const uint32_t* wide, uint32_t world, uint32_t me, uint64_t* out) {
    ExchangePlan p;
    if (!xchg_slice_plan(matrix, wide, world, me, p)) return 1;
    for (uint32_t r = 0; r < world; r++) { out[5 * r] = p.send_off[r]; out[5 * r + 1] = p.send_bytes[r]; out[5 * r + 2] = p.recv_off[r]; out[5 * r + 3] = p.recv_bytes[r]; out[5 * r + 4] = p.peer_off[r]; }
    out[5 * world] = p.longest_send;
    return 0;
}
int plancheck_gather(const uint64_t* sizes, uint32_t world, uint32_t me, int weights, uint64_t* out) {
    ExchangePlan p;
    uint64_t total = 0;
    if (!xchg_gather_plan(sizes, world, me, weights != 0, p, &total)) return 1;
    for (uint32_t r = 0; r < world; r++) { out[4 * r] = p.send_bytes[r]; out[4 * r + 1] = p.recv_off[r]; out[4 * r + 2] = p.recv_bytes[r]; out[4 * r + 3] = p.peer_off[r]; }
    out[4 * world] = p.longest_send; out[4 * world + 1] = total;
    return 0;
}
uint64_t plancheck_slice_bytes(uint64_t blocks, uint64_t words, uint32_t width) { return xchg_slice_bytes(blocks, words, width); }
uint64_t plancheck_range_bytes(uint64_t blocks, uint64_t words, int weights) { return xchg_range_bytes(blocks, words, weights != 0); }
}
