// K5 on the matrix cores -- the scaled-dot-product core of MultiHeadedAttention (model/self_attention.py:56-71) for head
// widths dk = D/nh that are a multiple of 8 (<= 64) and L <= 64; other shapes run the scalar kernels of mha.hip.
// Same contract and quirks as mha.hip: `mask.view(M,1,L,1) == 0` fills whole QUERY rows with -1e9 (uniform attention over
// all L keys, padded keys included), keys are never masked; dropout on the probabilities regenerated from (seed, index).
//
// One wave per (sequence m, head h); nothing goes through LDS and NO probability tensor is stored: the backward recomputes
// S, softmax and the dropout multipliers from q, k (the (M, nh, L, L) tensor was 614 MB for the classifier encoder of the
// full config, written once and read once -- more bytes than q, k, v together).
//   forward   S^T tile (keys x queries) = K . Q^T  on v_mfma_f32_16x16x4_f32 -> lane (c15, g) owns query c15 and keys
//             jt*16 + 4g + reg: softmax over the keys is a per-lane loop + one cross-lane-group reduction, and P^T is
//             already the B operand (k = keys) of  O^T (d x queries) = V^T . P^T  -> 16-byte stores of 4 consecutive d.
//   backward  in the same "transposed" layout: dP^T = V . dout^T, dS^T, dQ^T = K^T . dS^T;
//             in the "normal" layout (lane owns a KEY, S = Q . K^T recomputed, row statistics by DPP reductions over the
//             16 lanes of a row): dV^T = dout^T . P', dK^T = Q^T . dS.  Both layouts loop over 16-query tiles, the
//             dV / dK accumulators stay in registers across them.
// The k index of every product is permuted freely (both operands agree): a lane's operand elements are contiguous in
// memory (dk/4 floats of its row), loaded as float4 / float2.
#include "common.h"
#include "../../include/stage_hip.h"

// row fragment: lane (c15, g) <- X[row][g*KS .. g*KS + KS - 1] of the head slice (row clamped by the caller)
// bf16 storage (TS = stage_bf16): KS consecutive bf16 of the row, converted to fp32 as they are loaded
template <int KS>
__device__ __forceinline__ void mha_row_frag(float (&f)[KS], const stage_bf16* __restrict__ p, int g) {
    const stage_bf16* s = p + g * KS;
    if (KS == 2) {
        const unsigned u = *reinterpret_cast<const unsigned*>(s);
        f[0] = __uint_as_float(u << 16); f[1] = __uint_as_float(u & 0xFFFF0000u);
    } else {
#pragma unroll
        for (int c = 0; c < KS / 4; c++) { const float4 v = ldv4(s + 4 * c); f[4 * c] = v.x; f[4 * c + 1] = v.y; f[4 * c + 2] = v.z; f[4 * c + 3] = v.w; }
    }
}
template <int KS>
__device__ __forceinline__ void mha_row_frag(float (&f)[KS], const float* __restrict__ p, int g) {
    const float* s = p + g * KS;
    if (KS == 2) { const float2 v = *reinterpret_cast<const float2*>(s); f[0] = v.x; f[1] = v.y; }
    else {
#pragma unroll
        for (int c = 0; c < KS / 4; c++) { const float4 v = ld4(s + 4 * c); f[4 * c] = v.x; f[4 * c + 1] = v.y; f[4 * c + 2] = v.z; f[4 * c + 3] = v.w; }
    }
}
template <int KS>
__device__ __forceinline__ f32x4 mha_dot(const float (&a)[KS], const float (&b)[KS]) {
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; s++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc, 0, 0, 0);
    return acc;
}

// ---------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------
template <int T, int KS, typename TS = float>   // TS: storage type of q, k, v, out (and of dout, dq, dk, dv in the backward)
__global__ __launch_bounds__(256) void mha_fwd_mfma_kernel(const TS* __restrict__ q, const TS* __restrict__ k,
                                                           const TS* __restrict__ v, const float* __restrict__ mask,
                                                           TS* __restrict__ out, long items, int L, int D, int nh,
                                                           uint64_t seed, uint32_t th, float inv_keep, int ld) {
    constexpr int DK = 4 * KS, DT = (DK + 15) / 16;
    const int lane = threadIdx.x & 63, c15 = lane & 15, g = lane >> 4;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= items) return;
    const long m = item / nh;
    const int h = (int)(item % nh);
    // ld: row stride (elements) of q, k, v (and of dq, dk, dv): D for separate tensors, 3 D when they are the thirds of ONE fused
    // projection output (M, L, 3 D) -- stage_mha_core_qkv_*; out / dout always have row stride D
    const long base = m * L * (long)ld + (long)h * DK, obase = m * L * (long)D + (long)h * DK, pbase = (m * nh + h) * (long)L;
    const float rs = sqrtf((float)DK);
    float kf[T][KS];
#pragma unroll
    for (int jt = 0; jt < T; jt++) mha_row_frag<KS>(kf[jt], k + base + (long)min(jt * 16 + c15, L - 1) * ld, g);
#pragma unroll
    for (int it = 0; it < T; it++) {
        if (it * 16 >= L) break;
        const int qi = it * 16 + c15, qc = min(qi, L - 1);
        float qf[KS];
        mha_row_frag<KS>(qf, q + base + (long)qc * ld, g);
        const bool dead = mask[m * L + qc] == 0.f;
        f32x4 p[T];
        float mx = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < T; jt++) {
            p[jt] = mha_dot<KS>(kf[jt], qf);                 // S^T[j = jt*16 + 4g + reg][i = qi]
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float s = dead ? -1e9f : p[jt][r] / rs;
                p[jt][r] = s;
                if (jt * 16 + 4 * g + r < L) mx = fmaxf(mx, s);
            }
        }
        mx = cross_row_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int jt = 0; jt < T; jt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float e = (jt * 16 + 4 * g + r < L) ? expf(p[jt][r] - mx) : 0.f;
                p[jt][r] = e;
                sum += e;
            }
        sum = cross_row_sum(sum);
#pragma unroll
        for (int jt = 0; jt < T; jt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float pr = p[jt][r] / sum;
                const int j = jt * 16 + 4 * g + r;
                if (th && j < L) pr *= drop1(seed, (uint64_t)((pbase + qc) * L + j), th, inv_keep);
                p[jt][r] = pr;
            }
        // O^T (d x queries) = V^T . P^T
#pragma unroll
        for (int dt = 0; dt < DT; dt++) {
            f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int dcol = min(dt * 16 + c15, DK - 1);
#pragma unroll
            for (int jt = 0; jt < T; jt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float a = ldv1(v + base + (long)min(jt * 16 + 4 * g + r, L - 1) * ld + dcol);   // P = 0 for keys >= L
                    o = __builtin_amdgcn_mfma_f32_16x16x4f32(a, p[jt][r], o, 0, 0, 0);
                }
            const int d0 = dt * 16 + 4 * g;
            if (qi < L && d0 < DK) stv4(out + obase + (long)qi * D + d0, make_float4(o[0], o[1], o[2], o[3]));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------------------
// (up to three key tiles and head widths <= 32 the backward fits 256 registers without spills: two waves per SIMD instead of one --
// the cls-encoder call 2.75 -> 2.4 ms; four tiles or 64-wide heads would spill)
template <int T, int KS, typename TS = float>
__global__ __launch_bounds__(256, ((T <= 3 && KS <= 8) || T == 1 ? 2 : 1)) void mha_bwd_mfma_kernel(const TS* __restrict__ dout, const TS* __restrict__ q,
                                                           const TS* __restrict__ k, const TS* __restrict__ v,
                                                           const float* __restrict__ mask, TS* __restrict__ dq,
                                                           TS* __restrict__ dkk, TS* __restrict__ dv, long items, int L,
                                                           int D, int nh, uint64_t seed, uint32_t th, float inv_keep, int ld) {
    constexpr int DK = 4 * KS, DT = (DK + 15) / 16;
    const int lane = threadIdx.x & 63, c15 = lane & 15, g = lane >> 4;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= items) return;
    const long m = item / nh;
    const int h = (int)(item % nh);
    // ld: row stride (elements) of q, k, v (and of dq, dk, dv): D for separate tensors, 3 D when they are the thirds of ONE fused
    // projection output (M, L, 3 D) -- stage_mha_core_qkv_*; out / dout always have row stride D
    const long base = m * L * (long)ld + (long)h * DK, obase = m * L * (long)D + (long)h * DK, pbase = (m * nh + h) * (long)L;
    const float rs = sqrtf((float)DK);
    float kf[T][KS], vf[T][KS];                      // row fragments of K and V (rows jt*16 + c15)
#pragma unroll
    for (int jt = 0; jt < T; jt++) {
        const long ro = (long)min(jt * 16 + c15, L - 1) * ld;
        mha_row_frag<KS>(kf[jt], k + base + ro, g);
        mha_row_frag<KS>(vf[jt], v + base + ro, g);
    }
    f32x4 adv[DT][T], adk[DT][T];                    // dV^T, dK^T (d x keys), accumulated over the query tiles
#pragma unroll
    for (int dt = 0; dt < DT; dt++)
#pragma unroll
        for (int jt = 0; jt < T; jt++) adv[dt][jt] = adk[dt][jt] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int it = 0; it < T; it++) {
        if (it * 16 >= L) break;
        float qf[KS], gf[KS];                        // row fragments of Q and dout (rows it*16 + c15)
        {
            const int rr = min(it * 16 + c15, L - 1);
            mha_row_frag<KS>(qf, q + base + (long)rr * ld, g);
            mha_row_frag<KS>(gf, dout + obase + (long)rr * D, g);
        }
        // ---------------- transposed layout: lane = query it*16 + c15, registers = keys -> dQ ----------------
        {
            const int qi = it * 16 + c15, qc = min(qi, L - 1);
            const bool dead = mask[m * L + qc] == 0.f;
            f32x4 p[T], dp[T];
            float mx = -INFINITY;
#pragma unroll
            for (int jt = 0; jt < T; jt++) {
                p[jt] = mha_dot<KS>(kf[jt], qf);
                dp[jt] = mha_dot<KS>(vf[jt], gf);          // dP'^T[j][i] = <v_j, dout_i>
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float s = dead ? -1e9f : p[jt][r] / rs;
                    p[jt][r] = s;
                    if (jt * 16 + 4 * g + r < L) mx = fmaxf(mx, s);
                }
            }
            mx = cross_row_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int jt = 0; jt < T; jt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float e = (jt * 16 + 4 * g + r < L) ? expf(p[jt][r] - mx) : 0.f;
                    p[jt][r] = e;
                    sum += e;
                }
            sum = cross_row_sum(sum);
            float dot = 0.f;
#pragma unroll
            for (int jt = 0; jt < T; jt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int j = jt * 16 + 4 * g + r;
                    const float pr = p[jt][r] / sum;
                    const float mult = (th && j < L) ? drop1(seed, (uint64_t)((pbase + qc) * L + j), th, inv_keep) : 1.0f;
                    p[jt][r] = pr;
                    dp[jt][r] *= mult;                      // dP = dP' * mult
                    dot += pr * dp[jt][r];
                }
            dot = cross_row_sum(dot);
#pragma unroll
            for (int jt = 0; jt < T; jt++)
#pragma unroll
                for (int r = 0; r < 4; r++) p[jt][r] = dead ? 0.f : p[jt][r] * (dp[jt][r] - dot);   // dS^T
#pragma unroll
            for (int dt = 0; dt < DT; dt++) {
                f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
                const int dcol = min(dt * 16 + c15, DK - 1);
#pragma unroll
                for (int jt = 0; jt < T; jt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float a = ldv1(k + base + (long)min(jt * 16 + 4 * g + r, L - 1) * ld + dcol);   // dS = 0 for keys >= L
                        o = __builtin_amdgcn_mfma_f32_16x16x4f32(a, p[jt][r], o, 0, 0, 0);
                    }
                const int d0 = dt * 16 + 4 * g;
                if (qi < L && d0 < DK) stv4(dq + base + (long)qi * ld + d0, make_float4(o[0] / rs, o[1] / rs, o[2] / rs, o[3] / rs));
            }
        }
        // ---------------- normal layout: lane = key jt*16 + c15, registers = queries it*16 + 4g + reg -> dV, dK ----------------
        {
            f32x4 p[T], dp[T];
            float mx[4], sum[4], dot[4];
            bool dead[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                mx[r] = -INFINITY;
                dead[r] = mask[m * L + min(it * 16 + 4 * g + r, L - 1)] == 0.f;
            }
#pragma unroll
            for (int jt = 0; jt < T; jt++) {
                p[jt] = mha_dot<KS>(qf, kf[jt]);           // S[i = it*16 + 4g + reg][j = jt*16 + c15]
                dp[jt] = mha_dot<KS>(gf, vf[jt]);
                const bool jok = jt * 16 + c15 < L;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float s = dead[r] ? -1e9f : p[jt][r] / rs;
                    p[jt][r] = s;
                    if (jok) mx[r] = fmaxf(mx[r], s);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; r++) { mx[r] = group_max(mx[r], 16); sum[r] = 0.f; dot[r] = 0.f; }
#pragma unroll
            for (int jt = 0; jt < T; jt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float e = (jt * 16 + c15 < L) ? expf(p[jt][r] - mx[r]) : 0.f;
                    p[jt][r] = e;
                    sum[r] += e;
                }
#pragma unroll
            for (int r = 0; r < 4; r++) sum[r] = group_sum(sum[r], 16);
            f32x4 pd[T];                                   // P' = P * mult (B operand of dV)
#pragma unroll
            for (int jt = 0; jt < T; jt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int i = min(it * 16 + 4 * g + r, L - 1), j = jt * 16 + c15;
                    const float pr = p[jt][r] / sum[r];
                    const float mult = (th && j < L) ? drop1(seed, (uint64_t)((pbase + i) * L + j), th, inv_keep) : 1.0f;
                    const bool iok = it * 16 + 4 * g + r < L;
                    p[jt][r] = pr;
                    pd[jt][r] = iok ? pr * mult : 0.f;     // query rows >= L contribute nothing
                    dp[jt][r] *= mult;
                    dot[r] += pr * dp[jt][r];
                }
#pragma unroll
            for (int r = 0; r < 4; r++) dot[r] = group_sum(dot[r], 16);
#pragma unroll
            for (int jt = 0; jt < T; jt++)
#pragma unroll
                for (int r = 0; r < 4; r++)
                    p[jt][r] = (dead[r] || it * 16 + 4 * g + r >= L) ? 0.f : p[jt][r] * (dp[jt][r] - dot[r]);   // dS
#pragma unroll
            for (int dt = 0; dt < DT; dt++) {
                const int dcol = min(dt * 16 + c15, DK - 1);
                float ga[4], qa[4];                        // dout^T / Q^T operands: rows d = dt*16 + c15, k = queries 4g + r
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int rr = min(it * 16 + 4 * g + r, L - 1);
                    ga[r] = ldv1(dout + obase + (long)rr * D + dcol);
                    qa[r] = ldv1(q + base + (long)rr * ld + dcol);
                }
#pragma unroll
                for (int jt = 0; jt < T; jt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        adv[dt][jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[r], pd[jt][r], adv[dt][jt], 0, 0, 0);
                        adk[dt][jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[r], p[jt][r], adk[dt][jt], 0, 0, 0);
                    }
            }
        }
    }
    // dV^T / dK^T: lane (c15 = key, g) holds d = dt*16 + 4g .. +3
#pragma unroll
    for (int dt = 0; dt < DT; dt++)
#pragma unroll
        for (int jt = 0; jt < T; jt++) {
            const int j = jt * 16 + c15, d0 = dt * 16 + 4 * g;
            if (j < L && d0 < DK) {
                stv4(dv + base + (long)j * ld + d0, make_float4(adv[dt][jt][0], adv[dt][jt][1], adv[dt][jt][2], adv[dt][jt][3]));
                stv4(dkk + base + (long)j * ld + d0,
                    make_float4(adk[dt][jt][0] / rs, adk[dt][jt][1] / rs, adk[dt][jt][2] / rs, adk[dt][jt][3] / rs));
            }
        }
}

// 1 if the matrix-core kernels take this shape (then `probs` is neither written nor read)
extern "C" int stage_mha_core_recomputes(int L, int D, int nh) {
    if (L < 1 || L > 64 || nh < 1 || D % nh != 0) return 0;
    const int dk = D / nh;
    return (dk == 8 || dk == 16 || dk == 32 || dk == 64) ? 1 : 0;
}

#define MHA_DISPATCH(KERNEL, ...)                                                                                        \
    do {                                                                                                                 \
        const int T = (L + 15) / 16;                                                                                     \
        const dim3 grid((unsigned)((items + 3) / 4)), block(256);                                                        \
        switch (dk) {                                                                                                    \
            case 8:  MHA_T(KERNEL, 2, __VA_ARGS__); break;                                                               \
            case 16: MHA_T(KERNEL, 4, __VA_ARGS__); break;                                                               \
            case 32: MHA_T(KERNEL, 8, __VA_ARGS__); break;                                                               \
            default: MHA_T(KERNEL, 16, __VA_ARGS__); break;                                                              \
        }                                                                                                                \
    } while (0)
#define MHA_T(KERNEL, KSV, ...)                                                                                          \
    switch (T) {                                                                                                         \
        case 1: hipLaunchKernelGGL((KERNEL<1, KSV, TS>), grid, block, 0, st, __VA_ARGS__); break;                            \
        case 2: hipLaunchKernelGGL((KERNEL<2, KSV, TS>), grid, block, 0, st, __VA_ARGS__); break;                            \
        case 3: hipLaunchKernelGGL((KERNEL<3, KSV, TS>), grid, block, 0, st, __VA_ARGS__); break;                            \
        default: hipLaunchKernelGGL((KERNEL<4, KSV, TS>), grid, block, 0, st, __VA_ARGS__); break;                           \
    }

template <typename TS>
static int mha_fwd_t(const TS* q, const TS* k, const TS* v, const float* mask, TS* out, long long M, int L, int D, int nh,
                     float p_drop, unsigned long long seed, void* stream, int ld = 0) {
    hipStream_t st = (hipStream_t)stream;
    const int dk = D / nh;
    const long items = (long)M * nh;
    uint32_t th = p_drop > 0.f ? drop_thresh16(p_drop) : 0u;
    if (p_drop > 0.f && th == 0u) th = 1u;
    const float ik = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    MHA_DISPATCH(mha_fwd_mfma_kernel, q, k, v, mask, out, items, L, D, nh, (uint64_t)seed, th, ik, ld > 0 ? ld : D);
    STAGE_LAUNCH_CHECK();
    return 0;
}

template <typename TS>
static int mha_bwd_t(const TS* dout, const TS* q, const TS* k, const TS* v, const float* mask, TS* dq, TS* dk_out, TS* dv,
                     long long M, int L, int D, int nh, float p_drop, unsigned long long seed, void* stream, int ld = 0) {
    hipStream_t st = (hipStream_t)stream;
    const int dk = D / nh;
    const long items = (long)M * nh;
    uint32_t th = p_drop > 0.f ? drop_thresh16(p_drop) : 0u;
    if (p_drop > 0.f && th == 0u) th = 1u;
    const float ik = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    MHA_DISPATCH(mha_bwd_mfma_kernel, dout, q, k, v, mask, dq, dk_out, dv, items, L, D, nh, (uint64_t)seed, th, ik, ld > 0 ? ld : D);
    STAGE_LAUNCH_CHECK();
    return 0;
}

int stage_mha_fwd_mfma(const float* q, const float* k, const float* v, const float* mask, float* out, long long M, int L, int D,
                       int nh, float p_drop, unsigned long long seed, void* stream) {
    return mha_fwd_t<float>(q, k, v, mask, out, M, L, D, nh, p_drop, seed, stream);
}

int stage_mha_bwd_mfma(const float* dout, const float* q, const float* k, const float* v, const float* mask, float* dq, float* dk_out,
                       float* dv, long long M, int L, int D, int nh, float p_drop, unsigned long long seed, void* stream) {
    return mha_bwd_t<float>(dout, q, k, v, mask, dq, dk_out, dv, M, L, D, nh, p_drop, seed, stream);
}

// bf16 storage mode: q, k, v, out (dout, dq, dk, dv) are bf16; the mask, the scores, the softmax and every product stay fp32.
// Only the shapes of the matrix-core kernels (stage_mha_core_recomputes == 1); others return STAGE_ERR_SHAPE.
extern "C" int stage_mha_core_fwd_bf16(const void* q, const void* k, const void* v, const float* mask, void* out, long long M,
                                       int L, int D, int nh, float p_drop, unsigned long long seed, void* stream) {
    if (M <= 0) return 0;
    if (!stage_mha_core_recomputes(L, D, nh)) return STAGE_ERR_SHAPE;
    typedef stage_bf16 B;
    return mha_fwd_t<B>((const B*)q, (const B*)k, (const B*)v, mask, (B*)out, M, L, D, nh, p_drop, seed, stream);
}
extern "C" int stage_mha_core_bwd_bf16(const void* dout, const void* q, const void* k, const void* v, const float* mask, void* dq,
                                       void* dk, void* dv, long long M, int L, int D, int nh, float p_drop,
                                       unsigned long long seed, void* stream) {
    if (M <= 0) return 0;
    if (!stage_mha_core_recomputes(L, D, nh)) return STAGE_ERR_SHAPE;
    typedef stage_bf16 B;
    return mha_bwd_t<B>((const B*)dout, (const B*)q, (const B*)k, (const B*)v, mask, (B*)dq, (B*)dk, (B*)dv, M, L, D, nh, p_drop,
                        seed, stream);
}

// Fused projections: q, k, v are the three thirds of ONE (M, L, 3 D) tensor -- the output of a single Linear(D -> 3 D) whose weight is
// [W_q; W_k; W_v] (model/self_attention.py:35-44 applies three Linears to the same input) -- and dq, dk, dv the thirds of ONE gradient
// tensor of that shape, which IS the output gradient of that Linear: one forward GEMM, one dX GEMM and one weight-gradient GEMM
// instead of three each, and no summation of three input gradients.  fp32 and bf16 storage (is_bf16); matrix-core shapes only.
extern "C" int stage_mha_core_qkv_fwd(const void* qkv, const float* mask, void* out, long long M, int L, int D, int nh, float p_drop,
                                      unsigned long long seed, int is_bf16, void* stream) {
    if (M <= 0) return 0;
    if (!stage_mha_core_recomputes(L, D, nh)) return STAGE_ERR_SHAPE;
    if (is_bf16) {
        typedef stage_bf16 B;
        const B* p = (const B*)qkv;
        return mha_fwd_t<B>(p, p + D, p + 2 * D, mask, (B*)out, M, L, D, nh, p_drop, seed, stream, 3 * D);
    }
    const float* p = (const float*)qkv;
    return mha_fwd_t<float>(p, p + D, p + 2 * D, mask, (float*)out, M, L, D, nh, p_drop, seed, stream, 3 * D);
}
extern "C" int stage_mha_core_qkv_bwd(const void* dout, const void* qkv, const float* mask, void* dqkv, long long M, int L, int D, int nh,
                                      float p_drop, unsigned long long seed, int is_bf16, void* stream) {
    if (M <= 0) return 0;
    if (!stage_mha_core_recomputes(L, D, nh)) return STAGE_ERR_SHAPE;
    if (is_bf16) {
        typedef stage_bf16 B;
        const B* p = (const B*)qkv;
        B* g = (B*)dqkv;
        return mha_bwd_t<B>((const B*)dout, p, p + D, p + 2 * D, mask, g, g + D, g + 2 * D, M, L, D, nh, p_drop, seed, stream, 3 * D);
    }
    const float* p = (const float*)qkv;
    float* g = (float*)dqkv;
    return mha_bwd_t<float>((const float*)dout, p, p + D, p + 2 * D, mask, g, g + D, g + 2 * D, M, L, D, nh, p_drop, seed, stream, 3 * D);
}
