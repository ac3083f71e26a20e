// K1 for LONG region rows and for bf16 storage (BASELINE.json configs[4]: bf16 weights / activations with fp32 softmax
// accumulation, D = 256, 512 subtitle words per frame) -- StructuredAttention, model/context_query_attention.py:35-101.
// The specialised kernels (str_attn_fwd*.hip, str_attn_bwd_fused.hip) hold a frame's whole score row in registers / LDS
// (Lr <= 64).  Here the region axis is walked in 16-region blocks with a two-pass softmax, any Lr:
//   forward, one wave per (frame, 16-row context tile), Cn fragments resident in registers
//     pass 1  S^T block = Qn . Cn^T (v_mfma_f32_16x16x4_f32, fp32 accumulate), raw scores stored, running max / sum
//             per context row (online softmax; fully masked rows come out uniform, as in the reference)
//     pass 2  raw scores read back (L2-hot), S_ = exp(scale*S - max) / sum * mask stored, A^T += Qraw^T . S_^T
//   backward: the softmax backward needs <P, dP> over ALL regions of a row -- but <P, dP> = sum_r P_r (dA . Q_r) =
//     dA . A, a per-row dot product of two tensors that exist anyway.  With it every region block is independent:
//     kernel B1 (same wave mapping): dS block = scale * P * (Q . dA^T - <dA, A>) (+ external), stored; dCn slab accumulated
//     kernel B2 (one wave per (frame, region block, 64-wide d block)): dQraw = P^T . dA, dQn = dS^T . Cn over the context rows
// Storage type T = float or bf16 for Cn, Q, Qn, A, dA (scores, masks and all gradients leaving here are fp32); every
// product accumulates in fp32 on the fp32 matrix-core path, so the bf16 mode's only error is the rounding of its inputs /
// of A.  This is the functional path for the stress shapes, not a tuned one: operands are loaded straight from global
// memory in MFMA layout (one element per lane and k-step for the k = region products).
#include <hip/hip_bf16.h>
#include "common.h"
#include "../../include/stage_hip.h"

template <typename T> __device__ __forceinline__ float lng_ld(const T* p);
template <> __device__ __forceinline__ float lng_ld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float lng_ld<__hip_bfloat16>(const __hip_bfloat16* p) {
    return __uint_as_float((unsigned)(*reinterpret_cast<const unsigned short*>(p)) << 16);
}
template <typename T> __device__ __forceinline__ void lng_st(T* p, float v);
template <> __device__ __forceinline__ void lng_st<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void lng_st<__hip_bfloat16>(__hip_bfloat16* p, float v) { *p = __float2bfloat16(v); }

// lane (c15, g) <- X[row][g*DQ .. g*DQ + DQ - 1] as floats (row fragment: element s is the operand of k-step s)
template <typename T, int DQ>
__device__ __forceinline__ void lng_frag(float (&f)[DQ], const T* __restrict__ row, int g) {
#pragma unroll
    for (int s = 0; s < DQ; s++) f[s] = lng_ld<T>(row + g * DQ + s);
}

// ---------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int DQ>
__global__ __launch_bounds__(256) void str_attn_long_fwd_kernel(const T* __restrict__ Cn, const T* __restrict__ Q,
                                                                const T* __restrict__ Qn, const float* __restrict__ cmask,
                                                                const float* __restrict__ qmask, T* __restrict__ A,
                                                                float* __restrict__ S, float* __restrict__ Sn, int N, int NA,
                                                                int Li, int Lqa, int Lr, float scale) {
    constexpr int D = 4 * DQ, DT = (D + 15) / 16;
    const int lane = threadIdx.x & 63, c15 = lane & 15, g = lane >> 4;
    const int CR = NA * Lqa, CT = (CR + 15) >> 4;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= (long)N * Li * CT) return;
    const int ct = (int)(item % CT);
    const long frame = item / CT;
    const int n = (int)(frame / Li), i = (int)(frame % Li);
    const int c = ct * 16 + c15, cc = min(c, CR - 1);
    const bool cvalid = c < CR;
    const long orow = ((long)(n * NA + cc / Lqa) * Li + i) * Lqa + cc % Lqa;
    const float cm = cvalid ? cmask[(long)n * CR + cc] : 0.f;
    float cf[DQ];
    lng_frag<T, DQ>(cf, Cn + ((long)n * CR + cc) * D, g);
    const T* qn = Qn + frame * Lr * (long)D;
    const T* qr = Q + frame * Lr * (long)D;
    const float* qm = qmask + frame * Lr;
    const int nb = (Lr + 15) >> 4;
    // ---- pass 1: raw scores, online max / sum of exp(scale * raw) over the row ----
    float mx = -INFINITY, sum = 0.f;
    for (int rb = 0; rb < nb; rb++) {
        float qf[DQ];
        lng_frag<T, DQ>(qf, qn + (long)min(rb * 16 + c15, Lr - 1) * D, g);
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < DQ; s++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[s], cf[s], acc, 0, 0, 0);
        // acc[k] = <Qn[r = rb*16 + 4g + k], Cn[c]>
        float bmx = -INFINITY, xs[4];
        {
#pragma clang fp contract(off)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int r = rb * 16 + 4 * g + k;
                const float msk = (r < Lr) ? cm * qm[min(r, Lr - 1)] : 0.f;
                const float raw = acc[k] - 1e10f * (1.0f - msk);
                xs[k] = raw * scale;             // ONE rounded product for the max and the exponent (see str_attn.hip)
                if (r < Lr) {
                    bmx = fmaxf(bmx, xs[k]);
                    if (cvalid) S[orow * Lr + r] = raw;
                }
            }
        }
        bmx = cross_row_max(bmx);
        const float nmx = fmaxf(mx, bmx);
        float bs = 0.f;
#pragma unroll
        for (int k = 0; k < 4; k++) bs += (rb * 16 + 4 * g + k < Lr) ? expf(xs[k] - nmx) : 0.f;
        bs = cross_row_sum(bs);
        sum = sum * expf(mx - nmx) + bs;         // mx = -inf at the first block: exp(-inf) = 0
        mx = nmx;
    }
    // ---- pass 2: normalised scores, A^T (d x ctx) += Qraw^T . S_^T ----
    f32x4 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; dt++) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int rb = 0; rb < nb; rb++) {
        float p[4];
        {
#pragma clang fp contract(off)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int r = rb * 16 + 4 * g + k, rc = min(r, Lr - 1);
                const float msk = (r < Lr) ? cm * qm[rc] : 0.f;
                // the value pass 1 stored (padded context rows stored nothing: any finite value, they are discarded)
                const float raw = cvalid ? S[orow * Lr + rc] : -1e10f;
                const float x = raw * scale;
                p[k] = (r < Lr) ? expf(x - mx) / sum * msk : 0.f;
                if (cvalid && r < Lr) Sn[orow * Lr + r] = p[k];
            }
        }
#pragma unroll
        for (int dt = 0; dt < DT; dt++) {
            const int dcol = min(dt * 16 + c15, D - 1);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float a = lng_ld<T>(qr + (long)min(rb * 16 + 4 * g + k, Lr - 1) * D + dcol);   // S_ = 0 past Lr
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, p[k], o[dt], 0, 0, 0);
            }
        }
    }
    if (cvalid) {
        T* pa = A + orow * D;
#pragma unroll
        for (int dt = 0; dt < DT; dt++)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int d = dt * 16 + 4 * g + k;
                if (d < D) lng_st<T>(pa + d, o[dt][k]);
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward B1: dS (stored, fp32) and dCn slabs.  One wave per (n, context tile, frame chunk): walks its frames.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int DQ>
__global__ __launch_bounds__(256) void str_attn_long_bwd_ds_kernel(const T* __restrict__ dA, const T* __restrict__ A,
                                                                   const float* __restrict__ ext, const T* __restrict__ Q,
                                                                   const T* __restrict__ Qn, const float* __restrict__ Sn,
                                                                   float* __restrict__ dS, float* __restrict__ part, int N,
                                                                   int NA, int Li, int Lqa, int Lr, float scale, int nchunks) {
    constexpr int D = 4 * DQ, DT = (D + 15) / 16;
    const int lane = threadIdx.x & 63, c15 = lane & 15, g = lane >> 4;
    const int CR = NA * Lqa, CT = (CR + 15) >> 4;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= (long)N * CT * nchunks) return;
    const int chunk = (int)(item % nchunks);
    const int ct = (int)((item / nchunks) % CT);
    const int n = (int)(item / ((long)nchunks * CT));
    const int c = ct * 16 + c15, cc = min(c, CR - 1);
    const bool cvalid = c < CR;
    const int nb = (Lr + 15) >> 4;
    f32x4 dcn[DT];
#pragma unroll
    for (int dt = 0; dt < DT; dt++) dcn[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int i = chunk; i < Li; i += nchunks) {
        const long frame = (long)n * Li + i;
        const long orow = ((long)(n * NA + cc / Lqa) * Li + i) * Lqa + cc % Lqa;
        float gf[DQ];
        lng_frag<T, DQ>(gf, dA + orow * D, g);
        // <P, dP> over the whole row = <dA, A>: this lane's share of the row, then across the four lane groups
        float dot = 0.f;
        {
            const T* pa = A + orow * D + g * DQ;
#pragma unroll
            for (int s = 0; s < DQ; s++) dot += gf[s] * lng_ld<T>(pa + s);
        }
        dot = cross_row_sum(dot);
        const T* qr = Q + frame * Lr * (long)D;
        const T* qn = Qn + frame * Lr * (long)D;
        for (int rb = 0; rb < nb; rb++) {
            float qf[DQ];
            lng_frag<T, DQ>(qf, qr + (long)min(rb * 16 + c15, Lr - 1) * D, g);
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < DQ; s++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[s], gf[s], acc, 0, 0, 0);
            // acc[k] = dP[c][r = rb*16 + 4g + k]
            f32x4 G;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int r = rb * 16 + 4 * g + k;
                float v = 0.f;
                if (r < Lr && cvalid) {
                    v = scale * Sn[orow * Lr + r] * (acc[k] - dot);
                    if (ext) v += ext[orow * Lr + r];
                    dS[orow * Lr + r] = v;
                }
                G[k] = v;
            }
#pragma unroll
            for (int dt = 0; dt < DT; dt++) {
                const int dcol = min(dt * 16 + c15, D - 1);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const float a = lng_ld<T>(qn + (long)min(rb * 16 + 4 * g + k, Lr - 1) * D + dcol);   // G = 0 past Lr
                    dcn[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, G[k], dcn[dt], 0, 0, 0);
                }
            }
        }
    }
    if (cvalid) {
        float* dst = part + (((size_t)chunk * N + n) * CR + c) * D;
#pragma unroll
        for (int dt = 0; dt < DT; dt++)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int d = dt * 16 + 4 * g + k;
                if (d < D) dst[d] = dcn[dt][k];
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward B2: dQraw[r, :] = sum_c P[c, r] dA[c, :]   dQn[r, :] = sum_c dS[c, r] Cn[c, :]
// one wave per (frame, 16-region block, 16-wide d tile pair): k-steps of 4 context rows
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void str_attn_long_bwd_dq_kernel(const T* __restrict__ dA, const float* __restrict__ Sn,
                                                                   const float* __restrict__ dS, const T* __restrict__ Cn,
                                                                   float* __restrict__ dQraw, float* __restrict__ dQn, int N,
                                                                   int NA, int Li, int Lqa, int Lr, int D) {
    const int lane = threadIdx.x & 63, c15 = lane & 15, g = lane >> 4;
    const int CR = NA * Lqa, nb = (Lr + 15) >> 4, DT = (D + 15) >> 4;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= (long)N * Li * nb * DT) return;
    const int dt = (int)(item % DT);
    const int rb = (int)((item / DT) % nb);
    const long frame = item / ((long)DT * nb);
    const int n = (int)(frame / Li), i = (int)(frame % Li);
    const int r = rb * 16 + c15, rc = min(r, Lr - 1);
    const int dcol = min(dt * 16 + c15, D - 1);
    f32x4 ar = (f32x4){0.f, 0.f, 0.f, 0.f}, an = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < CR; c0 += 4) {
        const int c = c0 + g;
        const bool ok = c < CR;
        const int cc = ok ? c : CR - 1;
        const long orow = ((long)(n * NA + cc / Lqa) * Li + i) * Lqa + cc % Lqa;
        const float p = ok ? Sn[orow * Lr + rc] : 0.f;      // A operands: row = region c15, k = context row
        const float gs = ok ? dS[orow * Lr + rc] : 0.f;
        const float da = ok ? lng_ld<T>(dA + orow * D + dcol) : 0.f;   // B operands: k = context row, column = d
        const float cn = ok ? lng_ld<T>(Cn + ((long)n * CR + cc) * D + dcol) : 0.f;
        ar = __builtin_amdgcn_mfma_f32_16x16x4f32(p, da, ar, 0, 0, 0);
        an = __builtin_amdgcn_mfma_f32_16x16x4f32(gs, cn, an, 0, 0, 0);
    }
    // C layout: row 4g + k = region inside the block, column c15 = d inside the tile
    const int d = dt * 16 + c15;
    if (d < D) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int rr = rb * 16 + 4 * g + k;
            if (rr < Lr) {
                dQraw[(frame * Lr + rr) * D + d] = ar[k];
                dQn[(frame * Lr + rr) * D + d] = an[k];
            }
        }
    }
}

__global__ __launch_bounds__(256) void lng_slab_sum_kernel(const float* __restrict__ part, float* __restrict__ out, int nb, long C) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= C) return;
    float acc = 0.f;
    for (int b = 0; b < nb; b++) acc += part[(size_t)b * C + e];
    out[e] = acc;
}

#define LNG_CHUNKS 8

extern "C" size_t stage_str_attn_long_bwd_ws_bytes(int N, int NA, int Lqa, int D) {
    return (size_t)LNG_CHUNKS * N * NA * Lqa * D * sizeof(float);
}

template <typename T>
static int lng_fwd(const void* Cn, const void* Q, const void* Qn, const float* cm, const float* qm, void* A, float* S, float* Sn,
                   int N, int NA, int Li, int Lqa, int Lr, int D, float scale, hipStream_t st) {
    const int CT = (NA * Lqa + 15) / 16;
    const long items = (long)N * Li * CT;
    const dim3 grid((unsigned)((items + 3) / 4)), block(256);
#define LNG_F(DQV)                                                                                                         \
    hipLaunchKernelGGL((str_attn_long_fwd_kernel<T, DQV>), grid, block, 0, st, (const T*)Cn, (const T*)Q, (const T*)Qn, cm, qm, \
                       (T*)A, S, Sn, N, NA, Li, Lqa, Lr, scale)
    switch (D) {
        case 16: LNG_F(4); break;
        case 32: LNG_F(8); break;
        case 64: LNG_F(16); break;
        case 128: LNG_F(32); break;
        case 256: LNG_F(64); break;
        default: return STAGE_ERR_SHAPE;
    }
#undef LNG_F
    STAGE_LAUNCH_CHECK();
    return 0;
}

template <typename T>
static int lng_bwd(const void* dA, const void* A, const float* ext, const void* Cn, const void* Q, const void* Qn, const float* Sn,
                   float* dS, float* dQraw, float* dQn, float* dCn, int N, int NA, int Li, int Lqa, int Lr, int D, float scale,
                   float* part, hipStream_t st) {
    const int CR = NA * Lqa, CT = (CR + 15) / 16;
    int nchunks = LNG_CHUNKS;
    if (nchunks > Li) nchunks = Li;
    const long items = (long)N * CT * nchunks;
    const dim3 grid((unsigned)((items + 3) / 4)), block(256);
#define LNG_B(DQV)                                                                                                         \
    hipLaunchKernelGGL((str_attn_long_bwd_ds_kernel<T, DQV>), grid, block, 0, st, (const T*)dA, (const T*)A, ext, (const T*)Q, \
                       (const T*)Qn, Sn, dS, part, N, NA, Li, Lqa, Lr, scale, nchunks)
    switch (D) {
        case 16: LNG_B(4); break;
        case 32: LNG_B(8); break;
        case 64: LNG_B(16); break;
        case 128: LNG_B(32); break;
        case 256: LNG_B(64); break;
        default: return STAGE_ERR_SHAPE;
    }
#undef LNG_B
    STAGE_LAUNCH_CHECK();
    const long C = (long)N * CR * D;
    hipLaunchKernelGGL(lng_slab_sum_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, st, (const float*)part, dCn, nchunks, C);
    STAGE_LAUNCH_CHECK();
    const long items2 = (long)N * Li * ((Lr + 15) / 16) * ((D + 15) / 16);
    hipLaunchKernelGGL((str_attn_long_bwd_dq_kernel<T>), dim3((unsigned)((items2 + 3) / 4)), dim3(256), 0, st, (const T*)dA, Sn,
                       (const float*)dS, (const T*)Cn, dQraw, dQn, N, NA, Li, Lqa, Lr, D);
    STAGE_LAUNCH_CHECK();
    return 0;
}

extern "C" int stage_str_attn_long_fwd(const void* Cn, const void* Q, const void* Qn, const float* c_mask, const float* q_mask,
                                       void* A, float* S_raw, float* S_norm, int N, int NA, int Li, int Lqa, int Lr, int D,
                                       float scale, int storage_bf16, void* stream) {
    if (N <= 0 || Li <= 0) return 0;
    if (Lr < 1 || Lqa < 1 || NA < 1) return STAGE_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    return storage_bf16 ? lng_fwd<__hip_bfloat16>(Cn, Q, Qn, c_mask, q_mask, A, S_raw, S_norm, N, NA, Li, Lqa, Lr, D, scale, st)
                        : lng_fwd<float>(Cn, Q, Qn, c_mask, q_mask, A, S_raw, S_norm, N, NA, Li, Lqa, Lr, D, scale, st);
}

extern "C" int stage_str_attn_long_bwd(const void* dA, const void* A, const float* dS_raw_ext, const void* Cn, const void* Q,
                                       const void* Qn, const float* S_norm, float* dS_ws, float* dQraw, float* dQn, float* dCn,
                                       int N, int NA, int Li, int Lqa, int Lr, int D, float scale, int storage_bf16, void* ws,
                                       size_t ws_bytes, void* stream) {
    if (N <= 0 || Li <= 0) return 0;
    if (Lr < 1 || Lqa < 1 || NA < 1) return STAGE_ERR_SHAPE;
    if (ws_bytes < stage_str_attn_long_bwd_ws_bytes(N, NA, Lqa, D)) return STAGE_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    return storage_bf16 ? lng_bwd<__hip_bfloat16>(dA, A, dS_raw_ext, Cn, Q, Qn, S_norm, dS_ws, dQraw, dQn, dCn, N, NA, Li, Lqa, Lr, D,
                                                  scale, (float*)ws, st)
                        : lng_bwd<float>(dA, A, dS_raw_ext, Cn, Q, Qn, S_norm, dS_ws, dQraw, dQn, dCn, N, NA, Li, Lqa, Lr, D, scale,
                                         (float*)ws, st);
}
