// Which v_rcp_f32 + FMA sequences equal the correctly rounded 1.0f / x for EVERY f32 x in a range of exponents?  All 2^32 bit patterns.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/rcp_exhaustive scripts/rcp_exhaustive.hip && /tmp/rcp_exhaustive
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ float cand_a(float x) { const float r = __builtin_amdgcn_rcpf(x); const float e = __builtin_fmaf(-x, r, 1.0f); return __builtin_fmaf(e, r, r); }
__device__ __forceinline__ float cand_b(float x) { float r = cand_a(x); const float e = __builtin_fmaf(-x, r, 1.0f); return __builtin_fmaf(e, r, r); }
// one Newton step, then the remainder decides between the neighbours: the correctly rounded quotient q satisfies |1 - x q| <= |1 - x q'| for both neighbours q'
__device__ __forceinline__ float cand_c(float x) {
    const float r = cand_a(x);
    const float up = __uint_as_float(__float_as_uint(r) + 1u), down = __uint_as_float(__float_as_uint(r) - 1u);
    const float e0 = fabsf(__builtin_fmaf(-x, r, 1.0f)), e1 = fabsf(__builtin_fmaf(-x, up, 1.0f)), e2 = fabsf(__builtin_fmaf(-x, down, 1.0f));
    return e1 < e0 ? up : (e2 < e0 ? down : r);
}

__global__ void k_check(unsigned long long* bad /* [3][256]: per candidate, per input exponent */) {
    const uint64_t stride = uint64_t(gridDim.x) * blockDim.x;
    for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const uint32_t bits = uint32_t(i), ex = (bits >> 23) & 0xffu;
        if (ex == 0u || ex == 0xffu) continue;   // zero, denormals, inf, nan: the product takes the IEEE sequence there
        const float x = __uint_as_float(bits);
        const float want = 1.0f / x;
        if (__float_as_uint(cand_a(x)) != __float_as_uint(want)) atomicAdd(&bad[ex], 1ull);
        if (__float_as_uint(cand_b(x)) != __float_as_uint(want)) atomicAdd(&bad[256 + ex], 1ull);
        if (__float_as_uint(cand_c(x)) != __float_as_uint(want)) atomicAdd(&bad[512 + ex], 1ull);
    }
}

int main() {
    unsigned long long* d = nullptr;
    hipMalloc(&d, 768 * 8);
    hipMemset(d, 0, 768 * 8);
    hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, d);
    unsigned long long h[768];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[3] = {"one Newton step", "two Newton steps", "one step + remainder test"};
    for (int c = 0; c < 3; c++) {
        unsigned long long total = 0, inner = 0;
        int first_bad = -1, last_bad = -1, first_clean = -1, last_clean = -1;
        for (int e = 1; e < 255; e++) {
            total += h[c * 256 + e];
            if (e >= 27 && e <= 227) inner += h[c * 256 + e];   // 2^-100 .. 2^100
            if (h[c * 256 + e]) { if (first_bad < 0) first_bad = e; last_bad = e; } else { if (first_clean < 0) first_clean = e; last_clean = e; }
        }
        std::printf("%-28s mismatches: all normal x %llu, exponents 2^-100..2^100 %llu; exponents with mismatches %d..%d, clean %d..%d\n", names[c], total, inner, first_bad, last_bad, first_clean, last_clean);
        std::printf("   per exponent (biased):");
        for (int e = 1; e < 255; e++) if (h[c * 256 + e]) std::printf(" %d:%llu", e, h[c * 256 + e]);
        std::printf("\n");
    }
    return 0;
}
