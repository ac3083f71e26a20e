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

// ---- cross-lane reductions without the LDS crossbar --------------------------------------------------------------
// __shfl_xor compiles to ds_bpermute_b32 (an LDS-pipe round trip, ~100+ cycles of dependent latency per step, plus the
// byte-address VALU op).  Reductions only need "every lane ends up with the group's total", which DPP row operations
// (inside 16 lanes, foldable into the consuming VALU op) and gfx950's v_permlane16_swap / v_permlane32_swap (across
// 16-lane rows / wave halves, one VALU op each) provide directly.
//   quad_perm [1,0,3,2] = 0xB1 (xor 1), [2,3,0,1] = 0x4E (xor 2), row_half_mirror = 0x141, row_mirror = 0x140
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xf, 0xf, true));
}
// v_permlane16_swap(a, b): odd 16-lane rows of a <-> even rows of b.  With a = b = v the two results are v and its
// xor-16 partner (which one is which depends on the row), so any commutative op of the pair is the pair reduction.
// Two different inputs give a transpose-reduce step: rows 0/2 get a[own] + a[row + 1], rows 1/3 get b[row - 1] + b[own].
__device__ __forceinline__ float xsum16(float a, float b) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// v_permlane32_swap(a, b): upper half of a <-> lower half of b: lanes 0-31 get a[own] + a[lane + 32], lanes 32-63 get
// b[lane - 32] + b[own]
__device__ __forceinline__ float xsum32(float a, float b) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xmax16(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xmax32(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// reduction inside aligned groups of `width` lanes (width = power of two <= 64, uniform, all lanes of the group
// active); every lane of the group returns the total.  Steps past the width are computed and discarded by a uniform
// select rather than branched over: a runtime width keeps the code straight-line (the branches cut the unrolled row
// loops of the LayerNorm kernels into blocks the scheduler could not interleave), a constant width folds them away.
__device__ __forceinline__ float group_sum(float v, int width) {
    float t;
    t = v + dpp_mov<0xB1>(v);  v = width > 1 ? t : v;
    t = v + dpp_mov<0x4E>(v);  v = width > 2 ? t : v;
    t = v + dpp_mov<0x141>(v); v = width > 4 ? t : v;   // quads are uniform by now: mirroring inside 8 pairs quad 0 with 1
    t = v + dpp_mov<0x140>(v); v = width > 8 ? t : v;
    t = xsum16(v, v);          v = width > 16 ? t : v;
    t = xsum32(v, v);          v = width > 32 ? t : v;
    return v;
}
__device__ __forceinline__ float group_max(float v, int width) {
    float t;
    t = fmaxf(v, dpp_mov<0xB1>(v));  v = width > 1 ? t : v;
    t = fmaxf(v, dpp_mov<0x4E>(v));  v = width > 2 ? t : v;
    t = fmaxf(v, dpp_mov<0x141>(v)); v = width > 4 ? t : v;
    t = fmaxf(v, dpp_mov<0x140>(v)); v = width > 8 ? t : v;
    t = xmax16(v);                   v = width > 16 ? t : v;
    t = xmax32(v);                   v = width > 32 ? t : v;
    return v;
}
// the same through the LDS crossbar (ds_bpermute_b32): only for cat3_ln_bwd_rep_kernel, whose register allocation grows
// from 254 to 342 VGPRs (one wave per SIMD instead of two, +14 % time) with the VALU version above
__device__ __forceinline__ float group_sum_bperm(float v, int width) {
    for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) { return group_sum(v, 64); }
__device__ __forceinline__ float wave_max(float v) { return group_max(v, 64); }
// sum / max over the 4 lanes that share (lane & 15): lanes l, l^16, l^32, l^48
__device__ __forceinline__ float cross_row_sum(float v) {
    v = xsum16(v, v);
    return xsum32(v, v);
}
__device__ __forceinline__ float cross_row_max(float v) { return xmax32(xmax16(v)); }

// Ticket counters for dynamically distributed kernels WITHOUT a per-launch reset: a device-resident ring of 64 words that
// only ever count up.  The host knows exactly how many tickets a launch draws (every processed item past the static rounds
// draws one), so it hands the kernel the value its word will have when the launch starts (`base`, kernels use drawn - base in
// unsigned arithmetic) and adds the draw count afterwards.  A stream-ordered 4-byte memset per launch cost ~9 us of the K1
// forward's event-timed duration (launch gap + a second tiny kernel).  The 64 slots of a ring guard against a handful of
// launches being in flight on different streams.
struct StageTicket { unsigned int* word; unsigned int base; int dev; int slot; };
#include <mutex>
#define STAGE_MAX_DEVICES 16
struct StageTicketRing {
    unsigned int* ring = nullptr;     // device resident, 64 words
    unsigned int bases[64];
    unsigned int next = 0;
};
static inline StageTicketRing* stage_ticket_rings() { static StageTicketRing rings[STAGE_MAX_DEVICES]; return rings; }
static inline std::mutex& stage_ticket_mutex() { static std::mutex m; return m; }
// One ring per device (launches may come from several host threads / devices: nn.DataParallel replicas, autograd's backward
// threads).  The ring is zeroed with a host-synchronous memset at first use on a device, so every later launch -- on any
// stream -- is ordered after it.  Call stage_ticket_abort() when the launch that was to draw the tickets failed.
static inline StageTicket stage_next_ticket(unsigned int draws) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= STAGE_MAX_DEVICES) return {nullptr, 0u, -1, 0};
    std::lock_guard<std::mutex> lock(stage_ticket_mutex());
    StageTicketRing& r = stage_ticket_rings()[dev];
    if (!r.ring) {
        unsigned int* p = nullptr;
        if (hipMalloc((void**)&p, 64 * sizeof(unsigned int)) != hipSuccess) return {nullptr, 0u, -1, 0};
        if (hipMemset(p, 0, 64 * sizeof(unsigned int)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
            (void)hipFree(p);
            return {nullptr, 0u, -1, 0};
        }
        for (int i = 0; i < 64; i++) r.bases[i] = 0u;
        r.ring = p;
    }
    const int sl = (int)(r.next++ & 63u);
    StageTicket t = {r.ring + sl, r.bases[sl], dev, sl};
    r.bases[sl] += draws;
    return t;
}
// the launch failed: its tickets were never drawn -- put the slot back into a known state (word = base = 0)
static inline void stage_ticket_abort(const StageTicket& t) {
    if (!t.word || t.dev < 0) return;
    std::lock_guard<std::mutex> lock(stage_ticket_mutex());
    (void)hipMemset(t.word, 0, sizeof(unsigned int));
    (void)hipDeviceSynchronize();
    stage_ticket_rings()[t.dev].bases[t.slot] = 0u;
}
#define STAGE_LAUNCH_CHECK_TICKET(tk)              \
    do {                                           \
        hipError_t e__ = hipGetLastError();        \
        if (e__ != hipSuccess) {                   \
            stage_ticket_abort(tk);                \
            return (int)e__;                       \
        }                                          \
    } while (0)

// Counter-based dropout: one 64-bit SplitMix hash per group of 4 consecutive elements, 16 bits per element.
// keep(element) <=> its 16-bit field >= thresh16, thresh16 = round(p * 65536).  The same (seed, index) pair
// regenerates the identical mask in the backward kernels, so no mask tensor is ever stored.
__device__ __forceinline__ uint64_t mix64(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// mix64 in two stages: the first is LINEAR in idx -- a kernel whose hashes of a tile sit at idx0 + (compile-time constant) pays one 64-bit
// multiply per tile (mix64_lin) and a 64-bit add of a constant per hash instead of a multiply each:
//     mix64(seed, idx0 + c) == mix64_fin(mix64_lin(seed, idx0) + c * MIX64_C0)          (arithmetic mod 2^64: the same bits)
#define MIX64_C0 0x9E3779B97F4A7C15ull
__device__ __forceinline__ uint64_t mix64_lin(uint64_t seed, uint64_t idx) { return seed + (idx + 1) * MIX64_C0; }
__device__ __forceinline__ uint64_t mix64_fin(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ float4 drop4_fin(uint64_t z_lin, uint32_t thresh16, float inv_keep) {
    const uint64_t h = mix64_fin(z_lin);
    float4 m;
    m.x = ((uint32_t)(h) & 0xFFFFu) >= thresh16 ? inv_keep : 0.f;
    m.y = ((uint32_t)(h >> 16) & 0xFFFFu) >= thresh16 ? inv_keep : 0.f;
    m.z = ((uint32_t)(h >> 32) & 0xFFFFu) >= thresh16 ? inv_keep : 0.f;
    m.w = ((uint32_t)(h >> 48) & 0xFFFFu) >= thresh16 ? inv_keep : 0.f;
    return m;
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
// keep bits (bit i set <=> element 4*idx4 + i is kept) of the same stream as drop4
__device__ __forceinline__ uint32_t drop4_bits(uint64_t seed, uint64_t idx4, uint32_t thresh16) {
    const uint64_t h = mix64(seed, idx4);
    return (((uint32_t)(h) & 0xFFFFu) >= thresh16 ? 1u : 0u) | (((uint32_t)(h >> 16) & 0xFFFFu) >= thresh16 ? 2u : 0u) |
           (((uint32_t)(h >> 32) & 0xFFFFu) >= thresh16 ? 4u : 0u) | (((uint32_t)(h >> 48) & 0xFFFFu) >= thresh16 ? 8u : 0u);
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
// streaming read (a tensor this kernel reads exactly once): non-temporal hint, -2..5 % on the cat3 LayerNorm kernels and
// the masked max (tools/bench_rowops.py); no effect on the plain LayerNorm
__device__ __forceinline__ float4 ld4s(const float* p) {
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// streaming (non-temporal) 16-byte store for large outputs that are written once
__device__ __forceinline__ void st4_stream(float* p, float4 v) {
    __builtin_nontemporal_store((f32x4){v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4*>(p));
}

// ---- storage-typed element access (float, or bf16 as raw 16-bit words; arithmetic is always fp32) ----
// The bf16 storage mode (BASELINE.json configs[4]) keeps activations as bf16 in HBM: 4 elements = one 8-byte access.
struct stage_bf16 { unsigned short bits; };
__device__ __forceinline__ unsigned stage_pk_bf16(float lo, float hi) {   // round to nearest even, lo in the low half
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float4 ldv4(const float* p) { return ld4(p); }
__device__ __forceinline__ float4 ldv4(const stage_bf16* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xFFFF0000u));
}
__device__ __forceinline__ float4 ldv4s(const float* p) { return ld4s(p); }
__device__ __forceinline__ float4 ldv4s(const stage_bf16* p) { return ldv4(p); }
__device__ __forceinline__ void stv4(float* p, float4 v) { st4(p, v); }
__device__ __forceinline__ void stv4(stage_bf16* p, float4 v) {
    *reinterpret_cast<uint2*>(p) = make_uint2(stage_pk_bf16(v.x, v.y), stage_pk_bf16(v.z, v.w));
}
__device__ __forceinline__ void stv4s(float* p, float4 v) { st4_stream(p, v); }     // written once, not re-read here
__device__ __forceinline__ void stv4s(stage_bf16* p, float4 v) { stv4(p, v); }
__device__ __forceinline__ float ldv1(const float* p) { return *p; }
__device__ __forceinline__ float ldv1(const stage_bf16* p) { return __uint_as_float((unsigned)p->bits << 16); }
__device__ __forceinline__ void stv1(float* p, float v) { *p = v; }
__device__ __forceinline__ void stv1(stage_bf16* p, float v) { p->bits = (unsigned short)(stage_pk_bf16(v, 0.f) & 0xFFFFu); }
// ---- two-way fp16 split of fp32 operands for v_mfma_f32_*_f16 (gemm_stream.hip, str_attn_fwd_reg.hip; DESIGN.md finding 20) ----
typedef _Float16 sf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned h_cvt_pk(float lo, float hi) {     // round to nearest even, lo in the low half
    unsigned r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float h_lo_f32(unsigned p) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(p & 0xFFFFu)); }
__device__ __forceinline__ float h_hi_f32(unsigned p) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(p >> 16)); }
// (a, b) * sc -> packed fp16 pairs hi, lo with a * sc == hi + lo up to 2^-22 (sc a power of two)
__device__ __forceinline__ void h_split2(float a, float b, float sc, unsigned& hi, unsigned& lo) {
    const float as = a * sc, bs = b * sc;
    hi = h_cvt_pk(as, bs);
    lo = h_cvt_pk(as - h_lo_f32(hi), bs - h_hi_f32(hi));
}
// h_cvt_pk is INLINE ASM: the compiler's hazard recogniser does not see a VALU write in it, so when its result goes straight into a
// matrix instruction as an A / B operand nothing guarantees the wait states between the two (round 4: one schedule of the K1 forward
// put the conversion one s_nop in front of the MFMA that read it -- the first product of a tile then used a stale low-half word, errors
// of 5e-4 .. 10 in one output column, in one template instantiation only).  Call this on every register of a freshly split operand
// before the first matrix instruction that reads it: all of them are then written, and five wait states follow.
__device__ __forceinline__ void h_operands_ready(unsigned& a, unsigned& b, unsigned& c, unsigned& d) {
    asm volatile("s_nop 4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
// biased fp32 exponent of a magnitude -> the exponent field of the power of two that maps it into [2^11, 2^12)
__device__ __forceinline__ int h_up_field(int eb) { return min(265 - eb, 254); }
// max(|a|, |b|, |c|) in ONE instruction (the compiler builds |x| as max(|x|, |x|) and then a tree of two-input maxima: 17
// instructions for 8 values instead of 4)
__device__ __forceinline__ float h_amax3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4scale(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float f4hsum(float4 a) { return (a.x + a.y) + (a.z + a.w); }

// out[c] = sum_b part[b*stride + c] for c < C, fixed summation order (deterministic).  64 columns x 16 row groups per
// workgroup: the nb partial rows are read 16-way parallel (independent loads in flight) and combined by an LDS tree.
// Optional remap (k > 0): column c = t*D + d of a depthwise-conv partial goes to dw[d*k + t] (t < k) or db[d] (t == k).
__global__ __launch_bounds__(1024) static void stage_colreduce_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                                      float* __restrict__ out2, int nb, long stride,
                                                                      int C, int D, int k) {
    __shared__ float sm[16][64];
    const int x = threadIdx.x & 63, r = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + x;
    float acc = 0.f;
    if (c < C) {
#pragma unroll 4
        for (int b = r; b < nb; b += 16) acc += part[(size_t)b * stride + c];
    }
    sm[r][x] = acc;
    __syncthreads();
    if (r == 0 && c < C) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 16; j++) s += sm[j][x];
        if (k > 0) {
            const int t = c / D, d = c % D;
            if (t < k) out[d * k + t] = s;
            else out2[d] = s;
        } else out[c] = s;
    }
}
static inline void stage_colreduce(const float* part, float* out, float* out2, int nb, long stride, int C, int D, int k,
                                   hipStream_t st) {
    hipLaunchKernelGGL(stage_colreduce_kernel, dim3((C + 63) / 64), dim3(1024), 0, st, part, out, out2, nb, stride, C, D, k);
}

// positions per work item of the sliding-window kernels: sequences up to 48 positions are one chunk (no halo rows,
// no short tail chunk), longer ones are cut into equal chunks of at most 48
static inline int stage_chunk_len(int L) {
    const int n = (L + 47) / 48;
    return (L + n - 1) / n;
}
// two independent column reductions in ONE launch (weight and bias gradient partials of a GEMM): workgroups
// [0, ceil(CA/64)) reduce segment A, the rest segment B.  Same 16-way parallel, fixed-order scheme as above.
__global__ __launch_bounds__(1024) static void stage_colreduce2_kernel(const float* __restrict__ partA, float* __restrict__ outA,
                                                                       long strideA, int CA, const float* __restrict__ partB,
                                                                       float* __restrict__ outB, long strideB, int CB, int nb) {
    __shared__ float sm[16][64];
    const int x = threadIdx.x & 63, r = threadIdx.x >> 6;
    const int blocksA = (CA + 63) / 64;
    const bool isA = (int)blockIdx.x < blocksA;
    const float* part = isA ? partA : partB;
    float* out = isA ? outA : outB;
    const long stride = isA ? strideA : strideB;
    const int C = isA ? CA : CB;
    const int c = (isA ? blockIdx.x : blockIdx.x - blocksA) * 64 + x;
    float acc = 0.f;
    if (c < C) {
#pragma unroll 4
        for (int b = r; b < nb; b += 16) acc += part[(size_t)b * stride + c];
    }
    sm[r][x] = acc;
    __syncthreads();
    if (r == 0 && c < C) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 16; j++) s += sm[j][x];
        out[c] = s;
    }
}
static inline void stage_colreduce2(const float* partA, float* outA, long strideA, int CA, const float* partB, float* outB,
                                    long strideB, int CB, int nb, hipStream_t st) {
    hipLaunchKernelGGL(stage_colreduce2_kernel, dim3((CA + 63) / 64 + (CB + 63) / 64), dim3(1024), 0, st, partA, outA, strideA,
                       CA, partB, outB, strideB, CB, nb);
}

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
