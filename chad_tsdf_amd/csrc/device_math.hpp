// device_math.hpp -- Morton coding and the truncated-ray voxel traversal for gfx950.
//
// All float arithmetic here must round exactly like the reference's strict-IEEE CPU build so that voxel
// sets are bit-identical: the library is compiled with -ffp-contract=off (no FMA contraction), HIP's '/'
// and sqrtf are correctly rounded by default, and f32 denormals are kept (hipcc's default mode).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace chadhip {

constexpr uint64_t kBlockMask = 0xffffffffffffffc0ull;  // octree.hpp:33-35 -- one key per 4x4x4-voxel block
constexpr uint32_t kWave = 64;
// Counters that many workgroups add to are striped over separate 64-byte lines: atomics on ONE address are worked off at
// about 25 ns each, whoever issues them.
constexpr uint32_t kCountStripes = 256;   // MapView::brick_count[k * 16]: every new brick adds one (9409 per frame in the bench)
constexpr uint32_t kUpdateStripes = 128;  // FrameHeader::n_updates[k * 16]: one add per wave of the update kernel

struct F3 { float x, y, z; };
struct I3 { int x, y, z; };

// ---- Morton code (morton.hpp:21-37 over libmorton's morton3D_64: x -> bit 0, y -> bit 1, z -> bit 2) ----------
// gfx950 has no pdep/pext; 21-bit spread by the classic five-step mask cascade (bit-identical result).
__device__ __forceinline__ uint64_t spread21(uint32_t a) {
    uint64_t v = a & 0x1fffffu;
    v = (v | (v << 32)) & 0x001f00000000ffffull;
    v = (v | (v << 16)) & 0x001f0000ff0000ffull;
    v = (v | (v << 8)) & 0x100f00f00f00f00full;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}
__device__ __forceinline__ uint32_t compact21(uint64_t v) {
    v &= 0x1249249249249249ull;
    v = (v | (v >> 2)) & 0x10c30c30c30c30c3ull;
    v = (v | (v >> 4)) & 0x100f00f00f00f00full;
    v = (v | (v >> 8)) & 0x001f0000ff0000ffull;
    v = (v | (v >> 16)) & 0x001f00000000ffffull;
    v = (v | (v >> 32)) & 0x1fffffull;
    return uint32_t(v);
}
__device__ __forceinline__ uint64_t morton_encode(int vx, int vy, int vz) {
    // morton.hpp:24-27: bias by 2^20 in uint32, keep the low 21 bits of each axis
    return spread21((1u << 20) + uint32_t(vx)) | (spread21((1u << 20) + uint32_t(vy)) << 1) | (spread21((1u << 20) + uint32_t(vz)) << 2);
}
__device__ __forceinline__ I3 morton_decode(uint64_t mc) {
    // morton.hpp:30-36
    return {int(compact21(mc) - (1u << 20)), int(compact21(mc >> 1) - (1u << 20)), int(compact21(mc >> 2) - (1u << 20))};
}

// ---- small float helpers in the reference's evaluation order (glm 1.0.1 generic vec3 code) -------------------
__device__ __forceinline__ float dot3(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ F3 sub3(F3 a, F3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
// ---- correctly rounded 1 / x without the IEEE division's ten-instruction expansion ------------------------------------------------
// hipcc expands `1.0f / x` into v_div_scale x 2, v_rcp, four FMAs, v_div_fmas, v_div_fixup (and two mode switches): what the quotient
// needs over the whole f32 range.  For a NORMAL x whose reciprocal is normal too -- biased exponent 1 .. 252, |x| in [2^-126, 2^126) --
// v_rcp_f32 followed by ONE Newton step in FMAs, r + r (1 - x r), already IS the correctly rounded quotient: checked by enumeration
// of all 2^32 bit patterns on gfx950 (scripts/rcp_exhaustive.hip; the library's own function is held against `/` the same way by
// chad_selftest_reciprocal, tests/test_gpu_parity.py::test_fast_reciprocal_equals_the_ieee_quotient_for_every_float -- zero
// mismatches inside the range, every input of exponent 253 / 254 outside it wrong: the range test below is exact).  Everything else
// -- zero, denormals, huge values, infinities, NaN -- takes the IEEE sequence, in a branch a wave skips when none of its lanes needs it.
// A ray pays for four reciprocals (octree.hpp:92-93: normalize, 1 / dir): 44 instructions of its set-up become 12 + the range tests.
__device__ __forceinline__ float rcp_newton(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
}
__device__ __forceinline__ bool rcp_newton_exact_for(float x) {   // biased exponent in [1, 252]; false for NaN
    const float a = fabsf(x);
    return a >= 1.17549435e-38f /* 2^-126 */ && a < 8.50705917e37f /* 2^126 */;
}
__device__ __forceinline__ float recip1(float x) {
    float r;
    if (__builtin_expect(rcp_newton_exact_for(x), 1)) r = rcp_newton(x);
    else { asm volatile("; IEEE division for an operand outside [2^-126, 2^126)" ::: "memory"); r = 1.0f / x; }   // (the asm keeps the branch a branch)
    return r;
}
__device__ __forceinline__ F3 recip3(F3 v) {   // 1 / v, component by component: one range test for the three
    const float lo = __builtin_fminf(__builtin_fminf(fabsf(v.x), fabsf(v.y)), fabsf(v.z)), hi = __builtin_fmaxf(__builtin_fmaxf(fabsf(v.x), fabsf(v.y)), fabsf(v.z));
    F3 r;
    if (__builtin_expect(lo >= 1.17549435e-38f && hi < 8.50705917e37f, 1)) r = F3{rcp_newton(v.x), rcp_newton(v.y), rcp_newton(v.z)};   // (a NaN component -- excluded upstream, point_usable -- comes out NaN on either path)
    else { asm volatile("; IEEE divisions for a direction with a zero, tiny or non-finite component" ::: "memory"); r = F3{1.0f / v.x, 1.0f / v.y, 1.0f / v.z}; }
    return r;
}
__device__ __forceinline__ F3 unit3(F3 v) {  // glm::normalize: v * (1 / sqrt(dot(v, v)))
    const float s = recip1(sqrtf(dot3(v, v)));
    return {v.x * s, v.y * s, v.z * s};
}
__device__ __forceinline__ bool finite3(F3 v) { return isfinite(v.x) && isfinite(v.y) && isfinite(v.z); }
__device__ __forceinline__ int sgn(int v) { return (v > 0) - (v < 0); }

// voxel of a point: ivec3(floor(p * recip))  (morton.hpp:71-73)
__device__ __forceinline__ I3 voxel_of(F3 p, float recip) { return {int(floorf(p.x * recip)), int(floorf(p.y * recip)), int(floorf(p.z * recip))}; }

// A point takes part in a frame iff its ray direction is finite and its voxel is far from the int range
// (the reference has undefined behaviour otherwise: octree.hpp:92-97 converts NaN to int).
__device__ __forceinline__ bool point_usable(F3 p, F3 pos, float recip) {
    // "the unit vector towards the point is finite", without computing it: v * (1 / sqrt(dot(v, v))) is finite exactly when
    // v is finite and dot(v, v) > 0 -- a zero dot (point at the sensor, or an underflow) gives 0 * inf or x * inf, an infinite
    // dot gives a finite 0 * x -- so the square root and the division are not needed (the frame kernels test every point twice)
    const F3 v = sub3(p, pos);
    const float lim = 1073741824.0f;  // 2^30 voxels
    return finite3(v) && dot3(v, v) > 0.0f && fabsf(p.x * recip) < lim && fabsf(p.y * recip) < lim && fabsf(p.z * recip) < lim;
}

// ---- truncated-ray traversal (octree.hpp:86-152) -------------------------------------------------------------
// State of one ray's walk over the voxels of [p - trunc*dir, p + trunc*dir].  Per-axis scalars instead of
// indexable vectors: runtime-indexed arrays would live in scratch memory on gfx950.
struct RayWalk {
    I3 cur, last, step;
    F3 tmax, delta;
    int budget;  // emitted voxels never exceed 1 + sum|last - first| (each step moves one axis toward `last`)

    __device__ __forceinline__ void begin(F3 p, F3 pos, float res, float recip, float trunc) {
        const F3 dir = unit3(sub3(p, pos));                                     // :92
        const F3 inv = recip3(dir);                                             // :93
        const F3 a{p.x - dir.x * trunc, p.y - dir.y * trunc, p.z - dir.z * trunc};  // :94
        const F3 b{p.x + dir.x * trunc, p.y + dir.y * trunc, p.z + dir.z * trunc};  // :95
        cur = voxel_of(a, recip);                                               // :96
        last = voxel_of(b, recip);                                              // :97
        step = {sgn(last.x - cur.x), sgn(last.y - cur.y), sgn(last.z - cur.z)};  // :100
        delta = {fabsf(res * inv.x), fabsf(res * inv.y), fabsf(res * inv.z)};   // :102
        tmax.x = first_crossing(step.x, a.x, inv.x, res, recip);                // :104-118
        tmax.y = first_crossing(step.y, a.y, inv.y, res, recip);
        tmax.z = first_crossing(step.z, a.z, inv.z, res, recip);
        const long long span = (long long)abs(last.x - cur.x) + abs(last.y - cur.y) + abs(last.z - cur.z);
        budget = int(span < 4096 ? span : 4096);
    }
    // the same traversal rule on an arbitrary segment from -> to with direction dir (the space-carving extension walks the
    // free part of a ray with it: map_kernels.hpp, k_carve)
    __device__ __forceinline__ void begin_segment(F3 from, F3 to, F3 dir, float res, float recip) {
        const F3 inv = recip3(dir);
        cur = voxel_of(from, recip);
        last = voxel_of(to, recip);
        step = {sgn(last.x - cur.x), sgn(last.y - cur.y), sgn(last.z - cur.z)};
        delta = {fabsf(res * inv.x), fabsf(res * inv.y), fabsf(res * inv.z)};
        tmax.x = first_crossing(step.x, from.x, inv.x, res, recip);
        tmax.y = first_crossing(step.y, from.y, inv.y, res, recip);
        tmax.z = first_crossing(step.z, from.z, inv.z, res, recip);
        const long long span = (long long)abs(last.x - cur.x) + abs(last.y - cur.y) + abs(last.z - cur.z);
        budget = int(span < (1 << 24) ? span : (1 << 24));
    }
    // advance to the next voxel; false when the walk is over (the stepped axis passed `last`, :130-148)
    __device__ __forceinline__ bool next() {
        const bool sx = (tmax.x < tmax.y) && (tmax.x < tmax.z);   // comparison tree of :126-150
        const bool sy = !(tmax.x < tmax.y) && (tmax.y < tmax.z);
        bool over;
        if (sx)      { cur.x += step.x; tmax.x += delta.x; over = cur.x == last.x + step.x; }
        else if (sy) { cur.y += step.y; tmax.y += delta.y; over = cur.y == last.y + step.y; }
        else         { cur.z += step.z; tmax.z += delta.z; over = cur.z == last.z + step.z; }
        return !over && (--budget >= 0);
    }

    static __device__ __forceinline__ float first_crossing(int s, float a, float inv, float res, float recip) {
        // selects, not branches: three divergent if/else chains per ray cost more than computing both roundings
        const float scaled = a * recip;
        const float grid = s < 0 ? floorf(scaled) : ceilf(scaled);
        const float edge = s == 0 ? 3.402823466e+38f : res * grid;  // FLT_MAX, :108
        return fabsf((edge - a) * inv);
    }
};

// signed distance of one observation (octree.hpp:157-159): voxel corner minus point, projected on the normal
__device__ __forceinline__ float observe(I3 v, F3 p, F3 n, float res, float trunc) {
    const F3 c{float(v.x) * res - p.x, float(v.y) * res - p.y, float(v.z) * res - p.z};
    const float d = dot3(n, c);
    return d < -trunc ? -trunc : (trunc < d ? trunc : d);  // std::clamp: a NaN distance stays NaN (fmaxf would hide it)
}

}  // namespace chadhip
