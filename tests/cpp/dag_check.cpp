// Test harness for the host half of finalize (chad_tsdf_amd/csrc/dag.hpp): a C face over chadhip::Dag so that the CPU
// suite can drive it with cluster streams made from the oracle's leaves and compare the levels with the oracle's.
#include <climits>
#include "../../chad_tsdf_amd/csrc/dag.hpp"

using namespace chadhip;

extern "C" {
void* dagcheck_create() { return new Dag(); }
void dagcheck_destroy(void* h) { delete static_cast<Dag*>(h); }
void dagcheck_add_submap(void* h, const uint64_t* keys, const uint64_t* tsdf, const uint64_t* weight, size_t blocks, uint32_t* roots) {
    dag_add_submap(*static_cast<Dag*>(h), keys, tsdf, weight, blocks, &roots[0], &roots[1]);
}
size_t dagcheck_level_words(void* h, uint32_t level) {
    Dag* d = static_cast<Dag*>(h);
    return level < kDagDepth ? d->nodes[level].raw.size() : d->clusters.raw.size();
}
void dagcheck_level_copy(void* h, uint32_t level, void* out) {
    Dag* d = static_cast<Dag*>(h);
    if (level < kDagDepth) std::memcpy(out, d->nodes[level].raw.data(), d->nodes[level].raw.size() * 4);
    else std::memcpy(out, d->clusters.raw.data(), d->clusters.raw.size() * 8);
}
void dagcheck_level_counts(void* h, uint32_t level, uint32_t* out) {
    Dag* d = static_cast<Dag*>(h);
    if (level < kDagDepth) { out[0] = d->nodes[level].uniques; out[1] = d->nodes[level].dupes; }
    else { out[0] = d->clusters.uniques; out[1] = d->clusters.dupes; }
}
size_t dagcheck_walk(void* h, uint32_t root, uint64_t* mc, uint8_t* bits, size_t capacity) {
    size_t k = 0;
    dag_walk(*static_cast<Dag*>(h), root, [&](uint64_t code, uint8_t b) {
        if (k < capacity) { mc[k] = code; bits[k] = b; }
        k++;
    });
    return k;
}
float dagcheck_sd_of_bits(uint8_t bits, float trunc) { return cluster_sd_of_bits(bits, trunc); }
// DagCursor (the iterator of tsdf.hpp:126-151): script[i] > 0: that many next(), < 0: that many prev(), 0: first(), INT32_MIN: last().
// After every script entry the current leaf (code, sd byte, weight byte; code ~0 when invalid) is appended to the outputs.
size_t dagcheck_cursor_script(void* h, uint32_t root_tsdf, uint32_t root_weight, const int32_t* script, size_t steps, uint64_t* mc, uint8_t* sd_bits, uint8_t* w_bits) {
    DagCursor c;
    c.open(*static_cast<Dag*>(h), root_tsdf, root_weight);
    for (size_t i = 0; i < steps; i++) {
        if (script[i] == 0) c.first();
        else if (script[i] == INT32_MIN) c.last();
        else for (int32_t k = 0; k < (script[i] < 0 ? -script[i] : script[i]) && c.valid; k++) { if (script[i] > 0) c.next(); else c.prev(); }
        mc[i] = c.valid ? c.code() : ~0ull;
        sd_bits[i] = c.valid ? c.sd_bits() : 0;
        w_bits[i] = c.valid ? c.weight_bits() : 0;
    }
    return steps;
}
}
