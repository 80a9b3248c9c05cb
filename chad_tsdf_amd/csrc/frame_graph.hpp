// frame_graph.hpp -- one submit for a frame's chain of preparation kernels.
//
// An insert enqueues bounds -> keys -> two or three sort passes -> gather -> normals on the frame slot's stream:
// six to eight launches (round 4's measurements below still had the sort's catch-all as a launch of its own) at about 3 us of host time each.  The idea: record the chain once per frame slot as a hipGraph -- a linear
// chain of kernel nodes -- and replay it with every node's arguments replaced (hipGraphExecKernelNodeSetParams: input pointer,
// pose, sequence number and table change from frame to frame; grids and kernels do not, as long as the frame size, the sort's digit
// width and its pass count stay what they were).
//
// MEASURED (scripts/graph_ab.py, MI355X, ROCm 7.2): no gain, so the replay is opt-in (CHAD_GRAPHS=1).  A chain of eight trivial kernels
// costs the host 24 us launched one by one, 9.3 us as a graph and 11.6 us with all eight nodes' arguments replaced
// (scripts/microbench_graph.hip) -- but with this chain's kernels (fifteen or sixteen arguments each, structs by value) the replay
// costs as much as the launches: 26.9 against 24.5 us of host work per 200 k-point insert, 20.6 against 25.1 at 20 k points; and the
// device side is the same either way (a dependent kernel boundary costs what it costs).  What a stream of small frames waits for is
// the device -- a frame's life on its stream divided by the frames in flight -- not the host: of the 47.5 us an insert call takes
// in the cfg3 stream, 23 are spent waiting for a frame slot (chad_map_host_times).
//
// FrameLauncher is the one call site of a kernel launch in the chain: in kEager mode it launches, in kBuild mode it appends a node,
// in kUpdate mode it replaces the arguments of the node recorded at that position -- and reports a mismatch (another kernel, grid
// or LDS size than recorded) instead of launching the wrong thing; the caller then runs the chain eagerly and records it afresh.
#pragma once
#include <hip/hip_runtime.h>

#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

namespace chadhip {

struct FrameLauncher {
    enum Mode { kEager, kBuild, kUpdate };
    Mode mode = kEager;
    hipStream_t stream = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    struct Node { hipGraphNode_t node; void* func; dim3 grid, block; uint32_t lds; };
    std::vector<Node> nodes;
    size_t at = 0;              // kUpdate: position in the recorded chain
    bool matches = true;        // kUpdate: the chain is still what was recorded
    hipError_t error = hipSuccess;

    template <class... P, class... A>
    void operator()(void (*kernel)(P...), dim3 grid, dim3 block, size_t lds, A&&... a) {
        static_assert(sizeof...(P) == sizeof...(A), "one argument per kernel parameter");
        if (mode == kEager) {
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, static_cast<P>(a)...);
            return;
        }
        std::tuple<std::remove_cv_t<P>...> values(static_cast<P>(a)...);   // (copied by the runtime inside the calls below)
        void* ptrs[sizeof...(P) + 1];
        fill(ptrs, values, std::index_sequence_for<P...>{});
        hipKernelNodeParams p{};
        p.func = reinterpret_cast<void*>(kernel);
        p.gridDim = grid; p.blockDim = block; p.sharedMemBytes = unsigned(lds);
        p.kernelParams = ptrs; p.extra = nullptr;
        if (mode == kBuild) {
            hipGraphNode_t n = nullptr;
            const hipError_t e = hipGraphAddKernelNode(&n, graph, nodes.empty() ? nullptr : &nodes.back().node, nodes.empty() ? 0 : 1, &p);
            if (e != hipSuccess && error == hipSuccess) error = e;
            nodes.push_back(Node{n, p.func, grid, block, uint32_t(lds)});
            return;
        }
        if (!matches) return;
        if (at >= nodes.size() || nodes[at].func != p.func || !same(nodes[at].grid, grid) || !same(nodes[at].block, block) || nodes[at].lds != uint32_t(lds)) { matches = false; return; }
        const hipError_t e = hipGraphExecKernelNodeSetParams(exec, nodes[at].node, &p);
        if (e != hipSuccess) { matches = false; (void)hipGetLastError(); return; }
        at++;
    }

    // kBuild: start / finish a recording
    hipError_t begin_build() {
        destroy();
        mode = kBuild; error = hipSuccess;
        return hipGraphCreate(&graph, 0);
    }
    hipError_t end_build() {
        mode = kEager;
        if (error != hipSuccess) { const hipError_t e = error; destroy(); return e; }
        const hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (e != hipSuccess) destroy();
        return e;
    }
    void begin_update() { mode = kUpdate; at = 0; matches = true; }
    // the recorded chain with the arguments just given, if every launch of this frame found its node
    bool end_update_and_launch(hipError_t* e) {
        mode = kEager;
        if (!matches || at != nodes.size()) return false;
        *e = hipGraphLaunch(exec, stream);
        return true;
    }
    bool recorded() const { return exec != nullptr; }
    void destroy() {
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        exec = nullptr; graph = nullptr; nodes.clear(); at = 0; mode = kEager;
    }

private:
    static bool same(dim3 a, dim3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
    template <class Tuple, size_t... I>
    static void fill(void** ptrs, Tuple& values, std::index_sequence<I...>) { ((ptrs[I] = static_cast<void*>(&std::get<I>(values))), ...); (void)ptrs; (void)values; }
};

}  // namespace chadhip
