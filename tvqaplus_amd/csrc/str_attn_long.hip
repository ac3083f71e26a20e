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

// NEL consecutive elements -> floats with the widest loads their alignment allows (p is NEL-element aligned)
template <int NEL> __device__ __forceinline__ void lng_ldn(float (&f)[NEL], const float* p) {
    if constexpr (NEL % 4 == 0) {
#pragma unroll
        for (int q = 0; q < NEL / 4; q++) { const float4 v = ld4(p + 4 * q); f[4 * q] = v.x; f[4 * q + 1] = v.y; f[4 * q + 2] = v.z; f[4 * q + 3] = v.w; }
    } else if constexpr (NEL == 2) { const float2 v = *reinterpret_cast<const float2*>(p); f[0] = v.x; f[1] = v.y; }
    else f[0] = p[0];
}
template <int NEL> __device__ __forceinline__ void lng_ldn(float (&f)[NEL], const __hip_bfloat16* p) {
    auto lo = [](unsigned u) { return __uint_as_float(u << 16); };
    auto hi = [](unsigned u) { return __uint_as_float(u & 0xFFFF0000u); };
    if constexpr (NEL % 8 == 0) {
#pragma unroll
        for (int q = 0; q < NEL / 8; q++) {
            const uint4 v = *reinterpret_cast<const uint4*>(p + 8 * q);
            f[8 * q] = lo(v.x); f[8 * q + 1] = hi(v.x); f[8 * q + 2] = lo(v.y); f[8 * q + 3] = hi(v.y);
            f[8 * q + 4] = lo(v.z); f[8 * q + 5] = hi(v.z); f[8 * q + 6] = lo(v.w); f[8 * q + 7] = hi(v.w);
        }
    } else if constexpr (NEL == 4) {
        const uint2 v = *reinterpret_cast<const uint2*>(p);
        f[0] = lo(v.x); f[1] = hi(v.x); f[2] = lo(v.y); f[3] = hi(v.y);
    } else if constexpr (NEL == 2) { const unsigned v = *reinterpret_cast<const unsigned*>(p); f[0] = lo(v); f[1] = hi(v); }
    else f[0] = lng_ld<__hip_bfloat16>(p);
}
// The products whose contraction runs over the REGIONS (A = S_ . Q, dCn = dS . Qn) or over the context rows (dQ) take their
// d-side operand one element per lane and k-step.  With the natural mapping (d tile dt = columns 16 dt .. 16 dt + 15) a lane
// needs column 16 dt + c15 of a row for every tile: DT two-byte gathers per row.  The OUTPUT index may be permuted freely,
// so tile dt is defined as the columns DT c15 + dt: a lane then needs DT CONSECUTIVE elements of the row -- one or two
// 16-byte loads feed all DT tiles -- and it ends up owning DT consecutive output columns per accumulator register.
// lane (c15, g) <- X[row][g*DQ .. g*DQ + DQ - 1] as floats (row fragment: element s is the operand of k-step s)
template <typename T, int DQ>
__device__ __forceinline__ void lng_frag(float (&f)[DQ], const T* __restrict__ row, int g) {
#pragma unroll
    for (int s = 0; s < DQ; s++) f[s] = lng_ld<T>(row + g * DQ + s);
}

// Products whose contraction runs over d take both operands as row fragments.  On bf16 storage those operands ARE bf16, so the
// product is exact on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16, fp32 accumulation): D / 32 instructions of ~17 cycles
// instead of D / 4 fp32 ones of 32.  lane (c15, g) <- 8 consecutive elements at 32 s + 8 g of its row, per k-step s.
typedef __bf16 lng_bf16x8 __attribute__((ext_vector_type(8)));
template <int D> struct LngRowB { uint4 v[D / 32 > 0 ? D / 32 : 1]; };
template <int D> __device__ __forceinline__ void lng_rowb(LngRowB<D>& f, const __hip_bfloat16* __restrict__ row, int g) {
#pragma unroll
    for (int s = 0; s < D / 32; s++) f.v[s] = *reinterpret_cast<const uint4*>(row + 32 * s + 8 * g);
}
template <int D> __device__ __forceinline__ f32x4 lng_dotb(const LngRowB<D>& a, const LngRowB<D>& b) {
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < D / 32; s++)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(lng_bf16x8, a.v[s]), __builtin_bit_cast(lng_bf16x8, b.v[s]), acc, 0, 0, 0);
    return acc;
}
template <typename T, int D> struct LngUseB { static constexpr bool value = false; };
template <int D> struct LngUseB<__hip_bfloat16, D> { static constexpr bool value = (D % 32 == 0); };

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
    constexpr bool USEB = LngUseB<T, D>::value;
    float cf[USEB ? 1 : DQ];
    LngRowB<D> cfb;
    if constexpr (USEB) lng_rowb<D>(cfb, Cn + ((long)n * CR + cc) * D, g);
    else lng_frag<T, DQ>(cf, Cn + ((long)n * CR + cc) * D, g);
    const T* qn = Qn + frame * Lr * (long)D;
    const T* qr = Q + frame * Lr * (long)D;
    const float* qm = qmask + frame * Lr;
    const int nb = (Lr + 15) >> 4;
    const bool vec4 = (Lr & 3) == 0;
    // Region blocks behind the frame's last valid region are not computed: their raw scores are cos - 1e10 = -1e10 exactly (|cos| <= 1
    // is below half an ulp of 1e10), their weights exp(-1e11 - max) * mask = 0, they add nothing to A -- the constants are stored.
    // (A row without any unmasked region comes out 0 * uniform = 0 either way; a frame without a valid region computes nothing.)
    int nv = 0;
    for (int r0 = 0; r0 < Lr; r0 += 64) {
        const unsigned long long bal = __ballot(r0 + lane < Lr && qm[min(r0 + lane, Lr - 1)] != 0.f);
        if (bal) nv = r0 + 64 - __builtin_clzll(bal);
    }
    const int nvb = (nv + 15) >> 4;                   // blocks to compute (wave-uniform)
    // ---- pass 1: raw scores, online max / sum of exp(scale * raw) over the row ----
    float mx = -INFINITY, sum = 0.f;
    LngRowB<D> qnext;                    // bf16 path: the next block's Qn fragments are requested before this block is scored
    if constexpr (USEB) lng_rowb<D>(qnext, qn + (long)min(c15, Lr - 1) * D, g);
    for (int rb = 0; rb < nvb; rb++) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (USEB) {
            const LngRowB<D> qfb = qnext;
            lng_rowb<D>(qnext, qn + (long)min((rb + 1 < nvb ? rb + 1 : rb) * 16 + c15, Lr - 1) * D, g);
            acc = lng_dotb<D>(qfb, cfb);
        } else {
            float qf[DQ];
            lng_frag<T, DQ>(qf, qn + (long)min(rb * 16 + c15, Lr - 1) * D, g);
#pragma unroll
            for (int s = 0; s < DQ; s++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[s], cf[s], acc, 0, 0, 0);
        }
        // acc[k] = <Qn[r = rb*16 + 4g + k], Cn[c]>
        float bmx = -INFINITY, xs[4], rw[4];
        {
#pragma clang fp contract(off)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int r = rb * 16 + 4 * g + k;
                const float msk = (r < Lr) ? cm * qm[min(r, Lr - 1)] : 0.f;
                const float raw = acc[k] - 1e10f * (1.0f - msk);
                xs[k] = raw * scale;             // ONE rounded product for the max and the exponent (see str_attn.hip)
                rw[k] = raw;
                if (r < Lr) {
                    bmx = fmaxf(bmx, xs[k]);
                    if (cvalid && !vec4) S[orow * Lr + r] = raw;
                }
            }
            // a lane's four regions are consecutive: one 16-byte access per score map and block when Lr % 4 == 0
            if (vec4 && cvalid && rb * 16 + 4 * g < Lr) st4(S + orow * Lr + rb * 16 + 4 * g, make_float4(rw[0], rw[1], rw[2], rw[3]));
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
    // the blocks that are not computed: raw score -1e10, weight 0 (pass 2 of the bf16 path may still read the raw scores of the odd
    // block behind nvb: the same lane wrote them here)
    if (cvalid) {
        for (int rb = nvb; rb < nb; rb++) {
            const int b0 = rb * 16 + 4 * g;
            if (vec4) {
                if (b0 < Lr) {
                    st4(S + orow * Lr + b0, make_float4(-1e10f, -1e10f, -1e10f, -1e10f));
                    if (rb >= ((nvb + 1) & ~1)) st4(Sn + orow * Lr + b0, f4zero());
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (b0 + k < Lr) {
                        S[orow * Lr + b0 + k] = -1e10f;
                        if (rb >= ((nvb + 1) & ~1)) Sn[orow * Lr + b0 + k] = 0.f;
                    }
            }
        }
    }
    // ---- pass 2: normalised scores, A^T (d x ctx) += Qraw^T . S_^T ----
    f32x4 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; dt++) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr bool PF = USEB && (DT % 8 == 0);        // bf16 storage, D = 128 / 256
    constexpr int NW4 = PF ? DT / 8 : 1;
    // PF: the region contraction on the bf16 matrix cores, 32 regions (two blocks) per step.  K-slot (g, e) of the 16x16x32 MFMA =
    // region (rb + (e >> 2)) * 16 + 4 g + (e & 3): exactly the eight weights lane (c15, g) holds after the softmax of two blocks, so
    // the B operand needs no shuffle; the weights go in as bf16 pairs hi + lo (error 2^-16 of a weight -- the fp32 softmax survives),
    // Q is bf16 in HBM, so the product is exact.  A operand of d tile dt: element DT c15 + dt of the lane's eight Q rows, picked out of
    // the rows' 16-byte words with one v_perm per row pair.  32 MFMAs of ~16 cycles per step instead of 128 fp32 ones of 32.
    if constexpr (PF) {
        for (int rb = 0; rb < nvb; rb += 2) {
            uint4 q8[8][NW4];      // (requesting the next step's rows ahead -- 64 more registers -- measured the same)
#pragma unroll
            for (int e = 0; e < 8; e++)
#pragma unroll
                for (int q = 0; q < NW4; q++)
                    q8[e][q] = *reinterpret_cast<const uint4*>(qr + (long)min((rb + (e >> 2)) * 16 + 4 * g + (e & 3), Lr - 1) * D + DT * c15 + 8 * q);
            float p[8];
#pragma unroll
            for (int hb = 0; hb < 2; hb++) {
                const int b0 = (rb + hb) * 16 + 4 * g;
                const bool blk = b0 < Lr;
                float4 rv = make_float4(-1e10f, -1e10f, -1e10f, -1e10f);
                if (vec4 && cvalid && blk) rv = ld4(S + orow * Lr + b0);
                {
#pragma clang fp contract(off)
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int r = b0 + k, rc = min(r, Lr - 1);
                        const float msk = (r < Lr) ? cm * qm[rc] : 0.f;
                        const float raw = vec4 ? (k == 0 ? rv.x : (k == 1 ? rv.y : (k == 2 ? rv.z : rv.w))) : (cvalid ? S[orow * Lr + rc] : -1e10f);
                        const float x = raw * scale;
                        p[4 * hb + k] = (r < Lr) ? expf(x - mx) / sum * msk : 0.f;
                        if (!vec4 && cvalid && r < Lr) Sn[orow * Lr + r] = p[4 * hb + k];
                    }
                }
                if (vec4 && cvalid && blk) st4(Sn + orow * Lr + b0, make_float4(p[4 * hb], p[4 * hb + 1], p[4 * hb + 2], p[4 * hb + 3]));
            }
            unsigned bh[4], bl[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                bh[j] = stage_pk_bf16(p[2 * j], p[2 * j + 1]);
                bl[j] = stage_pk_bf16(p[2 * j] - __uint_as_float(bh[j] << 16), p[2 * j + 1] - __uint_as_float(bh[j] & 0xFFFF0000u));
            }
            const lng_bf16x8 vbh = __builtin_bit_cast(lng_bf16x8, make_uint4(bh[0], bh[1], bh[2], bh[3]));
            const lng_bf16x8 vbl = __builtin_bit_cast(lng_bf16x8, make_uint4(bl[0], bl[1], bl[2], bl[3]));
#pragma unroll
            for (int dt = 0; dt < DT; dt++) {
                unsigned aw[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint4 w0 = q8[2 * j][dt / 8], w1 = q8[2 * j + 1][dt / 8];
                    const int wi = (dt % 8) / 2;
                    const unsigned x0 = wi == 0 ? w0.x : (wi == 1 ? w0.y : (wi == 2 ? w0.z : w0.w));
                    const unsigned x1 = wi == 0 ? w1.x : (wi == 1 ? w1.y : (wi == 2 ? w1.z : w1.w));
                    aw[j] = __builtin_amdgcn_perm(x1, x0, (dt & 1) ? 0x07060302u : 0x05040100u);
                }
                const lng_bf16x8 va = __builtin_bit_cast(lng_bf16x8, make_uint4(aw[0], aw[1], aw[2], aw[3]));
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, vbl, o[dt], 0, 0, 0);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, vbh, o[dt], 0, 0, 0);
            }
        }
    } else
    for (int rb = 0; rb < ((nvb + 1) & ~1) && rb < nb; rb++) {
        float p[4];
        const bool blk = rb * 16 + 4 * g < Lr;
        float4 rv = make_float4(-1e10f, -1e10f, -1e10f, -1e10f);
        if (vec4 && cvalid && blk) rv = ld4(S + orow * Lr + rb * 16 + 4 * g);
        {
#pragma clang fp contract(off)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int r = rb * 16 + 4 * g + k, rc = min(r, Lr - 1);
                const float msk = (r < Lr) ? cm * qm[rc] : 0.f;
                // the value pass 1 stored (padded context rows stored nothing: any finite value, they are discarded)
                const float raw = vec4 ? (k == 0 ? rv.x : (k == 1 ? rv.y : (k == 2 ? rv.z : rv.w))) : (cvalid ? S[orow * Lr + rc] : -1e10f);
                const float x = raw * scale;
                p[k] = (r < Lr) ? expf(x - mx) / sum * msk : 0.f;
                if (!vec4 && cvalid && r < Lr) Sn[orow * Lr + r] = p[k];
            }
            if (vec4 && cvalid && blk) st4(Sn + orow * Lr + rb * 16 + 4 * g, make_float4(p[0], p[1], p[2], p[3]));
        }
        {
            float qd[4][DT];             // Q[region rb*16 + 4g + k][DT c15 .. DT c15 + DT - 1]: the operands of all DT tiles
#pragma unroll
            for (int k = 0; k < 4; k++) lng_ldn<DT>(qd[k], qr + (long)min(rb * 16 + 4 * g + k, Lr - 1) * D + DT * c15);   // S_ = 0 past Lr
#pragma unroll
            for (int dt = 0; dt < DT; dt++)
#pragma unroll
                for (int k = 0; k < 4; k++) o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qd[k][dt], p[k], o[dt], 0, 0, 0);
        }
    }
    if (cvalid) {   // o[dt][k] = A[c][d = DT (4g + k) + dt]: DT consecutive columns per register index k
        T* pa = A + orow * D;
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int dt = 0; dt < DT; dt++) lng_st<T>(pa + DT * (4 * g + k) + dt, o[dt][k]);
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
                                                                   int NA, int Li, int Lqa, int Lr, float scale, int nchunks,
                                                                   const int* __restrict__ fnv) {
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
    const bool vec4 = (Lr & 3) == 0;
    f32x4 dcn[DT];
#pragma unroll
    for (int dt = 0; dt < DT; dt++) dcn[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int i = chunk; i < Li; i += nchunks) {
        const long frame = (long)n * Li + i;
        // fnv: regions of the frame that can carry a gradient (lng_frame_nv_kernel); the blocks behind them have P = 0 and no external
        // gradient: dS = 0 there, nothing is added to dCn, and B2 does not read them
        const int nvb = fnv ? (__builtin_amdgcn_readfirstlane(fnv[frame]) + 15) >> 4 : nb;
        if (nvb == 0) continue;
        const long orow = ((long)(n * NA + cc / Lqa) * Li + i) * Lqa + cc % Lqa;
        float gf[LngUseB<T, D>::value ? 1 : DQ];
        LngRowB<D> gfb;                                       // bf16 storage: the row as packed bf16 matrix-core fragments
        // <P, dP> over the whole row = <dA, A>: this lane's share of the row, then across the four lane groups
        float dot = 0.f;
        if constexpr (LngUseB<T, D>::value) {
            lng_rowb<D>(gfb, dA + orow * D, g);
            LngRowB<D> afb;                                   // the A row in the same (k-step, lane group) partition
            lng_rowb<D>(afb, A + orow * D, g);
            auto lo = [](unsigned u) { return __uint_as_float(u << 16); };
            auto hi = [](unsigned u) { return __uint_as_float(u & 0xFFFF0000u); };
#pragma unroll
            for (int s = 0; s < D / 32; s++) {
                const uint4 a = afb.v[s], b = gfb.v[s];
                dot += lo(a.x) * lo(b.x) + hi(a.x) * hi(b.x) + lo(a.y) * lo(b.y) + hi(a.y) * hi(b.y)
                     + lo(a.z) * lo(b.z) + hi(a.z) * hi(b.z) + lo(a.w) * lo(b.w) + hi(a.w) * hi(b.w);
            }
        } else {
            lng_frag<T, DQ>(gf, dA + orow * D, g);
            const T* pa = A + orow * D + g * DQ;
#pragma unroll
            for (int s = 0; s < DQ; s++) dot += gf[s] * lng_ld<T>(pa + s);
        }
        dot = cross_row_sum(dot);
        const T* qr = Q + frame * Lr * (long)D;
        const T* qn = Qn + frame * Lr * (long)D;
        auto ds_block = [&](int rb) -> f32x4 {      // dP, dS (stored) of one 16-region block; returns the lane's four dS values
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (LngUseB<T, D>::value) {
                LngRowB<D> qfb;
                lng_rowb<D>(qfb, qr + (long)min(rb * 16 + c15, Lr - 1) * D, g);
                acc = lng_dotb<D>(qfb, gfb);
            } else {
                float qf[DQ];
                lng_frag<T, DQ>(qf, qr + (long)min(rb * 16 + c15, Lr - 1) * D, g);
#pragma unroll
                for (int s = 0; s < DQ; s++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[s], gf[s], acc, 0, 0, 0);
            }
            // acc[k] = dP[c][r = rb*16 + 4g + k]
            f32x4 G;
            if (vec4) {      // a lane's four regions are consecutive: 16-byte accesses
                float4 v = f4zero();
                if (cvalid && rb * 16 + 4 * g < Lr) {
                    const long off = orow * Lr + rb * 16 + 4 * g;
                    const float4 pv = ld4(Sn + off);
                    v = make_float4(scale * pv.x * (acc[0] - dot), scale * pv.y * (acc[1] - dot), scale * pv.z * (acc[2] - dot),
                                    scale * pv.w * (acc[3] - dot));
                    if (ext) v = f4add(v, ld4(ext + off));
                    st4(dS + off, v);
                }
                G = (f32x4){v.x, v.y, v.z, v.w};
            } else {
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
            }
            return G;
        };
        if constexpr (LngUseB<T, D>::value && (DT % 8 == 0)) {
            // bf16 storage, D = 128 / 256: dCn^T (d x ctx) += Qn^T . dS^T on the bf16 matrix cores, 32 regions (two blocks) per step, exactly as
            // the forward's A^T += Q^T . S_^T: K-slot (g, e) of the 16x16x32 MFMA = region (rb + (e >> 2)) * 16 + 4 g + (e & 3) = the eight dS
            // values lane (c15, g) holds after two blocks; dS goes in as bf16 pairs hi + lo (2^-16 of a value), Qn is bf16 in HBM.  32 MFMAs of
            // ~16 cycles per step instead of 128 fp32 ones of 32 (round 4: this product was 1.6 of the kernel's 2.7 ms at the stress shape).
            constexpr int NW4 = DT / 8;
            for (int rb = 0; rb < nvb; rb += 2) {
                uint4 q8[8][NW4];
#pragma unroll
                for (int e = 0; e < 8; e++)
#pragma unroll
                    for (int q = 0; q < NW4; q++)
                        q8[e][q] = *reinterpret_cast<const uint4*>(qn + (long)min((rb + (e >> 2)) * 16 + 4 * g + (e & 3), Lr - 1) * D + DT * c15 + 8 * q);
                const f32x4 G0 = ds_block(rb);
                f32x4 G1 = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (rb + 1 < nvb) G1 = ds_block(rb + 1);               // (wave-uniform)
                const float p[8] = {G0[0], G0[1], G0[2], G0[3], G1[0], G1[1], G1[2], G1[3]};
                unsigned bh[4], bl[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    bh[j] = stage_pk_bf16(p[2 * j], p[2 * j + 1]);
                    bl[j] = stage_pk_bf16(p[2 * j] - __uint_as_float(bh[j] << 16), p[2 * j + 1] - __uint_as_float(bh[j] & 0xFFFF0000u));
                }
                const lng_bf16x8 vbh = __builtin_bit_cast(lng_bf16x8, make_uint4(bh[0], bh[1], bh[2], bh[3]));
                const lng_bf16x8 vbl = __builtin_bit_cast(lng_bf16x8, make_uint4(bl[0], bl[1], bl[2], bl[3]));
#pragma unroll
                for (int dt = 0; dt < DT; dt++) {
                    unsigned aw[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const uint4 w0 = q8[2 * j][dt / 8], w1 = q8[2 * j + 1][dt / 8];
                        const int wi = (dt % 8) / 2;
                        const unsigned x0 = wi == 0 ? w0.x : (wi == 1 ? w0.y : (wi == 2 ? w0.z : w0.w));
                        const unsigned x1 = wi == 0 ? w1.x : (wi == 1 ? w1.y : (wi == 2 ? w1.z : w1.w));
                        aw[j] = __builtin_amdgcn_perm(x1, x0, (dt & 1) ? 0x07060302u : 0x05040100u);
                    }
                    const lng_bf16x8 va = __builtin_bit_cast(lng_bf16x8, make_uint4(aw[0], aw[1], aw[2], aw[3]));
                    dcn[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, vbl, dcn[dt], 0, 0, 0);
                    dcn[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, vbh, dcn[dt], 0, 0, 0);
                }
            }
        } else {
            for (int rb = 0; rb < nvb; rb++) {
                const f32x4 G = ds_block(rb);
                float qd[4][DT];         // Qn[region][DT c15 .. + DT - 1] (d tiles permuted as in the forward)
#pragma unroll
                for (int k = 0; k < 4; k++) lng_ldn<DT>(qd[k], qn + (long)min(rb * 16 + 4 * g + k, Lr - 1) * D + DT * c15);   // G = 0 past Lr
#pragma unroll
                for (int dt = 0; dt < DT; dt++)
#pragma unroll
                    for (int k = 0; k < 4; k++) dcn[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qd[k][dt], G[k], dcn[dt], 0, 0, 0);
            }
        }
    }
    if (cvalid) {
        float* dst = part + (((size_t)chunk * N + n) * CR + c) * D;
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int dt = 0; dt < DT; dt++) dst[DT * (4 * g + k) + dt] = dcn[dt][k];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward B2: dQraw[r, :] = sum_c P[c, r] dA[c, :]   dQn[r, :] = sum_c dS[c, r] Cn[c, :]
// one wave per (frame, 16-region block), all DT d tiles (permuted: tile dt = columns DT c15 + dt, so a lane takes DT
// consecutive elements of a dA / Cn row with one or two 16-byte loads and owns DT consecutive output columns); k-steps of 4
// context rows
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int DT, int DTW>   // DTW <= 8 d tiles per wave: 2 x DTW accumulators fit two waves per SIMD at D = 256
__global__ __launch_bounds__(256) void str_attn_long_bwd_dq_kernel(const T* __restrict__ dA, const float* __restrict__ Sn,
                                                                   const float* __restrict__ dS, const T* __restrict__ Cn,
                                                                   float* __restrict__ dQraw, float* __restrict__ dQn, int N,
                                                                   int NA, int Li, int Lqa, int Lr, const int* __restrict__ fnv) {
    constexpr int D = 16 * DT;
    const int lane = threadIdx.x & 63, c15 = lane & 15, g = lane >> 4;
    const int CR = NA * Lqa, nb = (Lr + 15) >> 4;
    constexpr int NH = DT / DTW;                             // d-tile groups per region block (one wave each)
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= (long)N * Li * nb * NH) return;
    const int hh = (int)(item % NH);
    const int rb = (int)((item / NH) % nb);
    const long frame = item / ((long)NH * nb);
    const int n = (int)(frame / Li), i = (int)(frame % Li);
    const int r = rb * 16 + c15, rc = min(r, Lr - 1);
    const int dofs = DT * c15 + DTW * hh;                    // this lane's DTW consecutive columns
    if (fnv && rb * 16 >= __builtin_amdgcn_readfirstlane(fnv[frame])) {
        // a block behind the last region that can carry a gradient (P = 0, dS = 0): both gradients are exact zeros
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int rr = rb * 16 + 4 * g + k;
            if (rr < Lr) {
                float* o1 = dQraw + (frame * Lr + rr) * D + dofs;
                float* o2 = dQn + (frame * Lr + rr) * D + dofs;
#pragma unroll
                for (int dt = 0; dt < DTW; dt++) { o1[dt] = 0.f; o2[dt] = 0.f; }
            }
        }
        return;
    }
    f32x4 ar[DTW], an[DTW];
#pragma unroll
    for (int dt = 0; dt < DTW; dt++) ar[dt] = an[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // one k-step = 4 context rows; the operands of the NEXT step are requested before the current one is multiplied (the
    // loop is a chain of memory round trips otherwise)
    struct Step { float p, gs; float da[DTW], cn[DTW]; };
    auto fetch = [&](Step& s, int c0) {
        const int c = c0 + g;
        const bool ok = c < CR;
        const int cc = ok ? c : CR - 1;
        const long orow = ((long)(n * NA + cc / Lqa) * Li + i) * Lqa + cc % Lqa;
        s.p = ok ? Sn[orow * Lr + rc] : 0.f;                // A operands: row = region c15, k = context row
        s.gs = ok ? dS[orow * Lr + rc] : 0.f;
        lng_ldn<DTW>(s.da, dA + orow * D + dofs);           // B operands: k = context row, columns dofs .. + DTW - 1
        lng_ldn<DTW>(s.cn, Cn + ((long)n * CR + cc) * D + dofs);
    };
    Step cur, nxt;
    fetch(nxt, 0);
    for (int c0 = 0; c0 < CR; c0 += 4) {
        cur = nxt;
        if (c0 + 4 < CR) fetch(nxt, c0 + 4);
#pragma unroll
        for (int dt = 0; dt < DTW; dt++) {                  // rows past CR: the A operands are zero (the clamped B rows are finite)
            ar[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.p, cur.da[dt], ar[dt], 0, 0, 0);
            an[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.gs, cur.cn[dt], an[dt], 0, 0, 0);
        }
    }
    // C layout: row 4g + k = region inside the block, column c15 -> d = DT c15 + dt
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int rr = rb * 16 + 4 * g + k;
        if (rr < Lr) {
            float* o1 = dQraw + (frame * Lr + rr) * D + dofs;
            float* o2 = dQn + (frame * Lr + rr) * D + dofs;
#pragma unroll
            for (int dt = 0; dt < DTW; dt++) { o1[dt] = ar[dt][k]; o2[dt] = an[dt][k]; }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward B2 on bf16 storage with D = 256 (BASELINE configs[4]): the kernel above spends its time on the fp32 matrix cores
// (0.5 TFLOP per call on v_mfma_f32_16x16x4_f32 = 3.2 ms at the stress shape) and every 16-region block re-reads the frame's dA and
// Cn rows.  Here a wave owns 32 regions and ONE of the two products (dQraw = P^T dA or dQn = dS^T Cn: 8 accumulator tiles of
// v_mfma_f32_32x32x16_bf16), so each row of dA / Cn is read once per 32 regions and product.  The contraction runs over the context
// rows: k-slot (h, e) = row c0 + 8h + e.  The weights (P or dS, fp32) go in as bf16 pairs hi + lo (error 2^-16 of a weight); dA / Cn ARE
// bf16.  d tile dt = columns 8 l + dt (l = lane position): a lane loads the 16 bytes at columns 8l..8l+7 of its eight rows -- two full
// 512-byte rows per instruction -- and an 8 x 8 transpose of 16-bit values (32 v_perm_b32, as in gemm_bf16_oct.hip) gives the B
// operand of every tile; the lane ends up owning 8 consecutive output columns of each of its 16 regions (full-row stores).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void str_attn_long_bwd_dq32_kernel(const __hip_bfloat16* __restrict__ dA, const float* __restrict__ Sn,
                                                                     const float* __restrict__ dS, const __hip_bfloat16* __restrict__ Cn,
                                                                     float* __restrict__ dQraw, float* __restrict__ dQn, int N, int NA,
                                                                     int Li, int Lqa, int Lr, const int* __restrict__ fnv) {
    constexpr int D = 256;
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    const int CR = NA * Lqa, nb = (Lr + 31) >> 5;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= (long)N * Li * nb * 2) return;
    const int prod = (int)(item & 1);                        // wave-uniform: 0 = dQraw (P, dA), 1 = dQn (dS, Cn)
    const int rb = (int)((item >> 1) % nb);
    const long frame = (item >> 1) / nb;
    const int n = (int)(frame / Li), i = (int)(frame % Li);
    float* __restrict__ out = prod ? dQn : dQraw;
    const float* __restrict__ wsrc = prod ? dS : Sn;
    const int rc = min(rb * 32 + l31, Lr - 1);
    const int nvf = fnv ? __builtin_amdgcn_readfirstlane(fnv[frame]) : Lr;
    // B1 writes dS for the 16-region blocks that can carry a gradient only: a region behind them has weight 0 (not whatever the
    // workspace holds)
    const bool reg_live = rb * 32 + l31 < ((nvf + 15) & ~15);
    if (rb * 32 >= nvf) {
        // regions behind the last one that can carry a gradient (P = 0, dS = 0): exact zeros
#pragma unroll
        for (int rr = 0; rr < 16; rr++) {
            const int reg = rb * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h;
            if (reg < Lr) {
                float* o = out + (frame * Lr + reg) * D + 8 * l31;
                st4(o, f4zero());
                st4(o + 4, f4zero());
            }
        }
        return;
    }
    f32x16 acc[8];
#pragma unroll
    for (int dt = 0; dt < 8; dt++)
#pragma unroll
        for (int rr = 0; rr < 16; rr++) acc[dt][rr] = 0.f;
    struct Step { float w[8]; uint4 q[8]; };
    auto fetch = [&](Step& s, int c0) {
        int c = c0 + 8 * h;
        int qa = c / Lqa, ma = c - qa * Lqa;                 // (candidate, word) of the lane's first row; the next seven follow
#pragma unroll
        for (int e = 0; e < 8; e++, c++) {
            const bool ok = c < CR;
            const int qq = ok ? qa : NA - 1, mm = ok ? ma : Lqa - 1;
            const long orow = ((long)(n * NA + qq) * Li + i) * Lqa + mm;
            s.w[e] = (ok && reg_live) ? wsrc[orow * Lr + rc] : 0.f;   // rows past CR: zero weights (the clamped rows are finite)
            s.q[e] = prod ? *reinterpret_cast<const uint4*>(Cn + ((long)n * CR + qq * Lqa + mm) * D + 8 * l31)
                          : *reinterpret_cast<const uint4*>(dA + orow * D + 8 * l31);
            if (++ma == Lqa) { ma = 0; qa++; }
        }
    };
    auto mul = [&](const Step& s) {
        unsigned bh[4], bl[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            bh[j] = stage_pk_bf16(s.w[2 * j], s.w[2 * j + 1]);
            bl[j] = stage_pk_bf16(s.w[2 * j] - __uint_as_float(bh[j] << 16), s.w[2 * j + 1] - __uint_as_float(bh[j] & 0xFFFF0000u));
        }
        typedef __bf16 dq_bf16x8 __attribute__((ext_vector_type(8)));
        const dq_bf16x8 vah = __builtin_bit_cast(dq_bf16x8, make_uint4(bh[0], bh[1], bh[2], bh[3]));
        const dq_bf16x8 val = __builtin_bit_cast(dq_bf16x8, make_uint4(bl[0], bl[1], bl[2], bl[3]));
#pragma unroll
        for (int dt = 0; dt < 8; dt++) {
            unsigned aw[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint4 w0 = s.q[2 * j], w1 = s.q[2 * j + 1];
                const int wi = dt >> 1;
                const unsigned x0 = wi == 0 ? w0.x : (wi == 1 ? w0.y : (wi == 2 ? w0.z : w0.w));
                const unsigned x1 = wi == 0 ? w1.x : (wi == 1 ? w1.y : (wi == 2 ? w1.z : w1.w));
                aw[j] = __builtin_amdgcn_perm(x1, x0, (dt & 1) ? 0x07060302u : 0x05040100u);
            }
            const dq_bf16x8 vb = __builtin_bit_cast(dq_bf16x8, make_uint4(aw[0], aw[1], aw[2], aw[3]));
            acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(val, vb, acc[dt], 0, 0, 0);
            acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vah, vb, acc[dt], 0, 0, 0);
        }
    };
    Step s0, s1;
    fetch(s0, 0);
    for (int c0 = 0; c0 < CR; c0 += 32) {
        if (c0 + 16 < CR) fetch(s1, c0 + 16);
        mul(s0);
        if (c0 + 16 < CR) {
            if (c0 + 32 < CR) fetch(s0, c0 + 32);
            mul(s1);
        }
    }
    // C/D layout: row (rr & 3) + 8 (rr >> 2) + 4 h = region inside the block, column l31 -> d = 8 l31 + dt
#pragma unroll
    for (int rr = 0; rr < 16; rr++) {
        const int reg = rb * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h;
        if (reg < Lr) {
            float* o = out + (frame * Lr + reg) * D + 8 * l31;
            st4(o, make_float4(acc[0][rr], acc[1][rr], acc[2][rr], acc[3][rr]));
            st4(o + 4, make_float4(acc[4][rr], acc[5][rr], acc[6][rr], acc[7][rr]));
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

// fnv[frame] = number of leading regions of the frame that can carry a gradient: last valid region + 1, or Lr when the external
// gradient on the raw scores is non-zero somewhere behind the blocks that hold valid regions (then nothing may be skipped: the
// scores of padded regions are cos - 1e10, d/dcos = 1).  One workgroup per frame; wave w scans the tails of context rows w, w + 4, ...
__global__ __launch_bounds__(256) void lng_frame_nv_kernel(const float* __restrict__ qmask, const float* __restrict__ ext,
                                                           int* __restrict__ fnv, int NA, int Li, int Lqa, int Lr) {
    const long frame = blockIdx.x;
    const int n = (int)(frame / Li), i = (int)(frame % Li), tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, CR = NA * Lqa;
    __shared__ int nv_sh[4];
    int nv = 0;
    for (int r = tid; r < Lr; r += 256)
        if (qmask[frame * Lr + r] != 0.f) nv = r + 1;            // ascending r per thread: the last hit is the largest
    nv = (int)wave_max((float)nv);
    if (lane == 0) nv_sh[wave] = nv;
    __syncthreads();
    nv = max(max(nv_sh[0], nv_sh[1]), max(nv_sh[2], nv_sh[3]));
    const int thr = ((nv + 15) >> 4) << 4;
    int hit = 0;
    if (ext && thr < Lr) {
        for (int c = wave; c < CR; c += 4) {
            const float* er = ext + ((((long)n * NA + c / Lqa) * Li + i) * Lqa + c % Lqa) * Lr;
            for (int col = thr + lane; col < Lr; col += 64) hit |= er[col] != 0.f;
        }
    }
    hit = __syncthreads_or(hit);
    if (tid == 0) fnv[frame] = hit ? Lr : nv;
}

#define LNG_CHUNKS 8

extern "C" size_t stage_str_attn_long_bwd_ws_bytes(int N, int NA, int Lqa, int D) {
    return (size_t)LNG_CHUNKS * N * NA * Lqa * D * sizeof(float);
}
extern "C" size_t stage_str_attn_long_bwd_qm_ws_bytes(int N, int NA, int Li, int Lqa, int D) {
    return stage_str_attn_long_bwd_ws_bytes(N, NA, Lqa, D) + (((size_t)N * Li * sizeof(int) + 255) & ~(size_t)255);
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
                   float* part, hipStream_t st, const float* qmask = nullptr, int* fnv = nullptr) {
    const int CR = NA * Lqa, CT = (CR + 15) / 16;
    if (qmask && fnv) {
        hipLaunchKernelGGL(lng_frame_nv_kernel, dim3((unsigned)(N * Li)), dim3(256), 0, st, qmask, ext, fnv, NA, Li, Lqa, Lr);
        STAGE_LAUNCH_CHECK();
    } else {
        fnv = nullptr;
    }
    int nchunks = LNG_CHUNKS;
    if (nchunks > Li) nchunks = Li;
    const long items = (long)N * CT * nchunks;
    const dim3 grid((unsigned)((items + 3) / 4)), block(256);
#define LNG_B(DQV)                                                                                                         \
    hipLaunchKernelGGL((str_attn_long_bwd_ds_kernel<T, DQV>), grid, block, 0, st, (const T*)dA, (const T*)A, ext, (const T*)Q, \
                       (const T*)Qn, Sn, dS, part, N, NA, Li, Lqa, Lr, scale, nchunks, (const int*)fnv)
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
    if constexpr (sizeof(T) == 2) {
        static const bool dq16 = getenv("STAGE_LONG_DQ16") != nullptr;      // developer switch: the 16-region fp32-MFMA kernel
        if (D == 256 && !dq16) {
            const long items32 = (long)N * Li * ((Lr + 31) / 32) * 2;
            hipLaunchKernelGGL(str_attn_long_bwd_dq32_kernel, dim3((unsigned)((items32 + 3) / 4)), dim3(256), 0, st, (const __hip_bfloat16*)dA, Sn,
                               (const float*)dS, (const __hip_bfloat16*)Cn, dQraw, dQn, N, NA, Li, Lqa, Lr, (const int*)fnv);
            STAGE_LAUNCH_CHECK();
            return 0;
        }
    }
    const bool dq_split = D > 128 && !(sizeof(T) == 2 && !getenv("STAGE_LONG_DQ_SPLIT"));
    const long items2 = (long)N * Li * ((Lr + 15) / 16) * (dq_split ? D / 128 : 1);
#define LNG_Q(DTV, DTWV)                                                                                                   \
    hipLaunchKernelGGL((str_attn_long_bwd_dq_kernel<T, DTV, DTWV>), dim3((unsigned)((items2 + 3) / 4)), dim3(256), 0, st, (const T*)dA, Sn, \
                       (const float*)dS, (const T*)Cn, dQraw, dQn, N, NA, Li, Lqa, Lr, (const int*)fnv)
    switch (D) {
        case 16: LNG_Q(1, 1); break;
        case 32: LNG_Q(2, 2); break;
        case 64: LNG_Q(4, 4); break;
        case 128: LNG_Q(8, 8); break;
        default:
            if (sizeof(T) == 2 && !getenv("STAGE_LONG_DQ_SPLIT")) LNG_Q(16, 16); else LNG_Q(16, 8);
            break;
    }
#undef LNG_Q
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

// The same with the region mask at hand: region blocks behind a frame's last valid region are skipped (exact zeros) unless the external
// gradient touches them (lng_frame_nv_kernel).  ws from stage_str_attn_long_bwd_qm_ws_bytes.
extern "C" int stage_str_attn_long_bwd_qm(const void* dA, const void* A, const float* dS_raw_ext, const void* Cn, const void* Q,
                                          const void* Qn, const float* S_norm, const float* q_mask, float* dS_ws, float* dQraw,
                                          float* dQn, float* dCn, int N, int NA, int Li, int Lqa, int Lr, int D, float scale,
                                          int storage_bf16, void* ws, size_t ws_bytes, void* stream) {
    if (N <= 0 || Li <= 0) return 0;
    if (Lr < 1 || Lqa < 1 || NA < 1 || !q_mask) return STAGE_ERR_SHAPE;
    if (ws_bytes < stage_str_attn_long_bwd_qm_ws_bytes(N, NA, Li, Lqa, D)) return STAGE_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    int* fnv = (int*)((char*)ws + stage_str_attn_long_bwd_ws_bytes(N, NA, Lqa, D));
    return storage_bf16 ? lng_bwd<__hip_bfloat16>(dA, A, dS_raw_ext, Cn, Q, Qn, S_norm, dS_ws, dQraw, dQn, dCn, N, NA, Li, Lqa, Lr, D,
                                                  scale, (float*)ws, st, q_mask, fnv)
                        : lng_bwd<float>(dA, A, dS_raw_ext, Cn, Q, Qn, S_norm, dS_ws, dQraw, dQn, dCn, N, NA, Li, Lqa, Lr, D, scale,
                                         (float*)ws, st, q_mask, fnv);
}
