// launcher.hpp -- the one call site of a kernel launch in a frame's chain of preparation kernels (bounds -> keys -> sort passes -> gather ->
// normals, on the frame slot's stream).  Until round 3 this was also where the chain could be recorded and replayed as a hipGraph
// (every node's arguments replaced per frame); measured on MI355X / ROCm 7.2 that cost the host as much as the launches it stood for
// (26.9 against 24.5 us of host work per 200 k-point insert), the device side is the same either way, and a stream of frames is paced by
// the device: removed in round 4.
#pragma once
#include <hip/hip_runtime.h>

namespace chadhip {

struct FrameLauncher {
    hipStream_t stream = nullptr;
    template <class... P, class... A>
    void operator()(void (*kernel)(P...), dim3 grid, dim3 block, size_t lds, A&&... a) {
        static_assert(sizeof...(P) == sizeof...(A), "one argument per kernel parameter");
        hipLaunchKernelGGL(kernel, grid, block, lds, stream, static_cast<P>(a)...);
    }
};

}  // namespace chadhip
