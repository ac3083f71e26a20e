// Shared device helpers for the STAGE hot-path kernels (gfx950 / CDNA4 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define STAGE_WAVE 64
#define STAGE_NEG (-1e10f)  // model/model_utils.py:14-15, model/context_query_attention.py:100

#define STAGE_LAUNCH_CHECK()                       \
    do {                                           \
        hipError_t e__ = hipGetLastError();        \
        if (e__ != hipSuccess) return (int)e__;    \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
// reduction inside aligned groups of `width` lanes (width = power of two <= 64)
__device__ __forceinline__ float group_sum(float v, int width) {
    for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float group_max(float v, int width) {
    for (int o = width >> 1; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// Counter-based dropout: one 64-bit SplitMix hash per group of 4 consecutive elements, 16 bits per element.
// keep(element) <=> its 16-bit field >= thresh16, thresh16 = round(p * 65536).  The same (seed, index) pair
// regenerates the identical mask in the backward kernels, so no mask tensor is ever stored.
__device__ __forceinline__ uint64_t mix64(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// multipliers (0 or 1/(1-p)) for the 4 elements starting at element index 4*idx4
__device__ __forceinline__ float4 drop4(uint64_t seed, uint64_t idx4, uint32_t thresh16, float inv_keep) {
    uint64_t h = mix64(seed, idx4);
    float4 m;
    m.x = ((uint32_t)(h) & 0xFFFFu) >= thresh16 ? inv_keep : 0.f;
    m.y = ((uint32_t)(h >> 16) & 0xFFFFu) >= thresh16 ? inv_keep : 0.f;
    m.z = ((uint32_t)(h >> 32) & 0xFFFFu) >= thresh16 ? inv_keep : 0.f;
    m.w = ((uint32_t)(h >> 48) & 0xFFFFu) >= thresh16 ? inv_keep : 0.f;
    return m;
}
// single element `e` (global element index): same stream as drop4(seed, e/4)[e%4]
__device__ __forceinline__ float drop1(uint64_t seed, uint64_t e, uint32_t thresh16, float inv_keep) {
    uint64_t h = mix64(seed, e >> 2);
    uint32_t f = (uint32_t)(h >> (16 * (e & 3))) & 0xFFFFu;
    return f >= thresh16 ? inv_keep : 0.f;
}
__host__ __device__ __forceinline__ uint32_t drop_thresh16(float p) {
    float t = p * 65536.0f + 0.5f;
    return t <= 0.f ? 0u : (t >= 65535.f ? 65535u : (uint32_t)t);
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4scale(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float f4hsum(float4 a) { return (a.x + a.y) + (a.z + a.w); }

static inline int stage_pow2_ceil(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}
static inline int stage_grid_for(long long work_items, int per_block, int cap) {
    long long g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}
