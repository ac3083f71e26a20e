// K1 -- StructuredAttention (model/context_query_attention.py:35-101), the roofline kernel of this path.
//
//   Cn = drop(C / max(|C|,1e-12))  (pre-pass, rowops.hip)      C : (N, NA, Lqa, D)   broadcast over the Li frames
//   Qn = drop(Q / max(|Q|,1e-12))  (in-kernel, per frame)      Q : (N, Li, Lr, D)    broadcast over the NA answers
//   S  = Cn.Qn^T - 1e10*(1 - cm (x) qm)        raw_s           (N, NA, Li, Lqa, Lr)
//   S_ = softmax(scale*S, -1) * (cm (x) qm)    normalised      (N, NA, Li, Lqa, Lr)
//   A  = S_ . Q   (un-normalised Q)                            (N, NA, Li, Lqa, D)
//
// Forward work decomposition: one workgroup = (batch item n, a chunk of FPB frames).  Its NA*Lqa context rows are cut
// into 16-row tiles, TPW tiles per wave; each wave keeps its Cn fragments in registers for the whole chunk.  Per
// frame the Lr x D region tile is staged once in LDS (raw + normalised copy), then per context tile:
//   stage 1  S^T tile (regions x ctx) = Qn . Cn^T   on v_mfma_f32_16x16x4_f32 (A = Qn rows from LDS, B = Cn regs)
//            -> each lane owns one context column and 4 consecutive regions per region tile: the masked softmax is a
//               per-lane loop + two cross-lane-group shuffles, and the normalised weights are ALREADY in the B-operand
//               layout of stage 2 (any k-permutation is legal if A and B agree) -- no LDS round trip.
//   stage 2  A^T tile (d x ctx) = Qraw^T . S_^T     on the same MFMA -> each lane owns 4 consecutive d of one context
//            row: 16-byte stores of A.
// HBM traffic = algorithmic bytes: Q read once per frame, Cn re-read from L2 per chunk, A / S / S_ written once.
#include <stdlib.h>
#include "common.h"
#include "../../include/stage_hip.h"

#define TPW 2  // context tiles (16 rows each) per wave

__device__ __forceinline__ int dchunk(int g, int m, int nch) {
    // 4-float chunk index owned by lane group g at step m (bank-conflict-free ds_read_b128 for D = 128, see header)
    return m + nch * (g >> 1) + 2 * nch * (g & 1);
}

template <int RT, int MAXNCH>
__global__ __launch_bounds__(512) void str_attn_fwd_kernel(
    const float* __restrict__ Cn, const float* __restrict__ Q, const float* __restrict__ cmask,
    const float* __restrict__ qmask, float* __restrict__ A, float* __restrict__ S, float* __restrict__ Sn, int N,
    int NA, int Li, int Lqa, int Lr, int D, float scale, int FPB, uint64_t seed, uint32_t th, float inv_keep) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int LDQ = D + 4;                    // padded row stride (floats)
    float* Qr = lds;                          // [RT*16][LDQ] raw regions (pad rows stay zero)
    float* Qh = Qr + RT * 16 * LDQ;           // [RT*16][LDQ] normalised (+dropout)
    float* qm = Qh + RT * 16 * LDQ;           // [RT*16] region mask of the frame (0 for pad)
    volatile int* frame_any_p = (volatile int*)(qm + RT * 16);  // all LDS lives in the dynamic region (16-B base)
    const bool TRAIN = th != 0u;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int c15 = lane & 15, g = lane >> 4;
    const int CR = NA * Lqa;
    const int nch = D >> 4;                   // 4-float chunks per lane group
    const int chunks_per_n = (Li + FPB - 1) / FPB;
    const int n = blockIdx.x / chunks_per_n;
    const int f0 = (blockIdx.x % chunks_per_n) * FPB;
    const int f1 = min(Li, f0 + FPB);

    // zero both region tiles once (pad rows must be exact zeros: 0 * pad never becomes NaN)
    for (int i = tid; i < 2 * RT * 16 * LDQ; i += blockDim.x) lds[i] = 0.f;

    // this wave's context tiles: Cn fragments (B operand of stage 1), masks, output row bases
    float4 cf[TPW][MAXNCH];  // [tile][m < nch]
    float cmv[TPW];
    long orow_base[TPW];  // ((n*NA + a)*Li)*Lqa + w   (add i*Lqa per frame)
    bool cvalid[TPW];
#pragma unroll
    for (int t = 0; t < TPW; t++) {
        const int c = (wave * TPW + t) * 16 + c15;
        cvalid[t] = c < CR;
        const int a = cvalid[t] ? c / Lqa : 0, w = cvalid[t] ? c % Lqa : 0;
        orow_base[t] = ((long)(n * NA + a) * Li) * Lqa + w;
        cmv[t] = cvalid[t] ? cmask[(long)n * CR + c] : 0.f;
#pragma unroll
        for (int m = 0; m < MAXNCH; m++) {
            if (m < nch && cvalid[t]) cf[t][m] = ld4(Cn + ((long)n * CR + c) * D + 4 * dchunk(g, m, nch));
            else cf[t][m] = f4zero();
        }
    }
    __syncthreads();

    for (int i = f0; i < f1; i++) {
        const long frame = (long)n * Li + i;
        // ---- (a) stage the frame's regions: raw copy + mask ----
        const int D4 = D >> 2;
        for (int e = tid; e < Lr * D4; e += blockDim.x) {
            const int r = e / D4, q = e % D4;
            st4(&Qr[r * LDQ + 4 * q], ld4(Q + (frame * Lr + r) * D + 4 * q));
        }
        if (tid < RT * 16) qm[tid] = tid < Lr ? qmask[frame * Lr + tid] : 0.f;
        if (tid == 0) *frame_any_p = 0;
        __syncthreads();
        // ---- (b) row norms -> normalised copy ----
        for (int r = wave; r < Lr; r += nw) {
            float s = 0.f;
            for (int d = lane; d < D; d += 64) { const float v = Qr[r * LDQ + d]; s += v * v; }
            s = wave_sum(s);
            const float nrm = fmaxf(sqrtf(s), 1e-12f);
            for (int q = lane; q < D4; q += 64) {
                float4 v = ld4(&Qr[r * LDQ + 4 * q]);
                v = make_float4(v.x / nrm, v.y / nrm, v.z / nrm, v.w / nrm);
                if (TRAIN) v = f4mul(v, drop4(seed, (uint64_t)(frame * Lr + r) * D4 + q, th, inv_keep));
                st4(&Qh[r * LDQ + 4 * q], v);
            }
            if (lane == 0 && qm[r] != 0.f) *frame_any_p = 1;
        }
        __syncthreads();
        const bool any = *frame_any_p != 0;

        // ---- (c) stage 1: S^T tiles ----
        f32x4 acc[TPW][RT];
#pragma unroll
        for (int t = 0; t < TPW; t++)
#pragma unroll
            for (int rt = 0; rt < RT; rt++) acc[t][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (any) {
#pragma unroll
            for (int m = 0; m < MAXNCH; m++) {
                if (m < nch) {
                    float4 qv[RT];
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) qv[rt] = ld4(&Qh[(rt * 16 + c15) * LDQ + 4 * dchunk(g, m, nch)]);
#pragma unroll
                    for (int j = 0; j < 4; j++) {
#pragma unroll
                        for (int rt = 0; rt < RT; rt++) {
                            const float qa = j == 0 ? qv[rt].x : (j == 1 ? qv[rt].y : (j == 2 ? qv[rt].z : qv[rt].w));
#pragma unroll
                            for (int t = 0; t < TPW; t++) {
                                const float cb = j == 0 ? cf[t][m].x : (j == 1 ? cf[t][m].y : (j == 2 ? cf[t][m].z : cf[t][m].w));
                                acc[t][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa, cb, acc[t][rt], 0, 0, 0);
                            }
                        }
                    }
                }
            }
        }
        // ---- (d) mask, softmax over regions, store S / S_ ; acc becomes the stage-2 B operand ----
#pragma unroll
        for (int t = 0; t < TPW; t++) {
#pragma clang fp contract(off)  // scale*raw must be ONE rounded value for both the max and the exponent (see below)
            float mx = -INFINITY;
            float msk[RT][4], xs[RT][4];
#pragma unroll
            for (int rt = 0; rt < RT; rt++)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int R = rt * 16 + 4 * g + k;
                    msk[rt][k] = cmv[t] * qm[R];
                    const float raw = acc[t][rt][k] - 1e10f * (1.0f - msk[rt][k]);
                    acc[t][rt][k] = raw;
                    // a contracted fma(raw, scale, -mx) would see -1e11 exactly vs the rounded max: exp(-2048) = 0,
                    // 0/0 on fully masked rows.  Contraction is off in this block and the product is kept.
                    xs[rt][k] = raw * scale;
                    if (R < Lr) mx = fmaxf(mx, xs[rt][k]);
                }
            mx = cross_row_max(mx);
            float p[RT][4], sum = 0.f;
#pragma unroll
            for (int rt = 0; rt < RT; rt++)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int R = rt * 16 + 4 * g + k;
                    p[rt][k] = (R < Lr) ? expf(xs[rt][k] - mx) : 0.f;
                    sum += p[rt][k];
                }
            sum = cross_row_sum(sum);
            const long orow = orow_base[t] + (long)i * Lqa;
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                const int R0 = rt * 16 + 4 * g;
                float4 raw4 = make_float4(acc[t][rt][0], acc[t][rt][1], acc[t][rt][2], acc[t][rt][3]);
                float4 pn;
                pn.x = p[rt][0] / sum * msk[rt][0];
                pn.y = p[rt][1] / sum * msk[rt][1];
                pn.z = p[rt][2] / sum * msk[rt][2];
                pn.w = p[rt][3] / sum * msk[rt][3];
                acc[t][rt] = (f32x4){pn.x, pn.y, pn.z, pn.w};
                if (cvalid[t] && R0 < Lr) {
                    float* ps = S + orow * Lr + R0;
                    float* pp = Sn + orow * Lr + R0;
                    if ((Lr & 3) == 0) {
                        st4(ps, raw4);
                        st4(pp, pn);
                    } else {
                        const float rv[4] = {raw4.x, raw4.y, raw4.z, raw4.w}, pv[4] = {pn.x, pn.y, pn.z, pn.w};
#pragma unroll
                        for (int k = 0; k < 4; k++)
                            if (R0 + k < Lr) { ps[k] = rv[k]; pp[k] = pv[k]; }
                    }
                }
            }
        }
        // ---- (e) stage 2: A^T tiles, two 16-wide d tiles at a time ----
        for (int dt = 0; dt < (D >> 4); dt += 2) {
            const bool two = dt + 1 < (D >> 4);
            f32x4 o[TPW][2];
#pragma unroll
            for (int t = 0; t < TPW; t++) o[t][0] = o[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (any) {
#pragma unroll
                for (int rt = 0; rt < RT; rt++)
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const float* qrow = &Qr[(rt * 16 + 4 * g + k) * LDQ + dt * 16 + c15];
                        const float q0 = qrow[0];
                        const float q1 = two ? qrow[16] : 0.f;
#pragma unroll
                        for (int t = 0; t < TPW; t++) {
                            o[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(q0, acc[t][rt][k], o[t][0], 0, 0, 0);
                            o[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(q1, acc[t][rt][k], o[t][1], 0, 0, 0);
                        }
                    }
            }
#pragma unroll
            for (int t = 0; t < TPW; t++) {
                if (!cvalid[t]) continue;
                float* pa = A + (orow_base[t] + (long)i * Lqa) * D + dt * 16 + 4 * g;
                st4(pa, make_float4(o[t][0][0], o[t][0][1], o[t][0][2], o[t][0][3]));
                if (two) st4(pa + 16, make_float4(o[t][1][0], o[t][1][1], o[t][1][2], o[t][1][3]));
            }
        }
        __syncthreads();  // tiles are re-staged by the next frame
    }
}

static size_t fwd_lds_bytes(int RT, int D) { return ((size_t)2 * RT * 16 * (D + 4) + RT * 16 + 4) * sizeof(float); }

template <int RT>
static int launch_fwd(const float* Cn, const float* Q, const float* cm, const float* qm, float* A, float* S, float* Sn,
                      int N, int NA, int Li, int Lqa, int Lr, int D, float scale, float p_drop,
                      unsigned long long seed, hipStream_t st) {
    const int CR = NA * Lqa;
    const int CT = (CR + 15) / 16;
    const int nw = (CT + TPW - 1) / TPW;
    if (nw > 8) return STAGE_ERR_SHAPE;  // NA*Lqa <= 256 context rows (max_qa_l = 40 -> 200)
    // frames per workgroup: amortise the Cn fragment load but keep >= ~2048 workgroups in flight when possible
    int FPB = 8;
    while (FPB > 1 && (long)N * ((Li + FPB - 1) / FPB) < 2048) FPB >>= 1;
    const int grid = N * ((Li + FPB - 1) / FPB);
    const size_t lds = fwd_lds_bytes(RT, D);
    uint32_t th = p_drop > 0.f ? drop_thresh16(p_drop) : 0u;
    if (p_drop > 0.f && th == 0u) th = 1u;
    const float ik = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    if (lds > 64 * 1024) {  // above the default dynamic-LDS cap the limit must be raised per kernel (160 KiB/CU)
        (void)hipFuncSetAttribute((const void*)str_attn_fwd_kernel<RT, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)str_attn_fwd_kernel<RT, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    if (D <= 128)
        hipLaunchKernelGGL((str_attn_fwd_kernel<RT, 8>), dim3(grid), dim3(64 * nw), lds, st, Cn, Q, cm, qm, A, S, Sn, N,
                           NA, Li, Lqa, Lr, D, scale, FPB, (uint64_t)seed, th, ik);
    else
        hipLaunchKernelGGL((str_attn_fwd_kernel<RT, 16>), dim3(grid), dim3(64 * nw), lds, st, Cn, Q, cm, qm, A, S, Sn, N,
                           NA, Li, Lqa, Lr, D, scale, FPB, (uint64_t)seed, th, ik);
    STAGE_LAUNCH_CHECK();
    return 0;
}

extern "C" int stage_str_attn_fwd_v1(const float* Cn, const float* Q, const float* c_mask, const float* q_mask, float* A,
                                  float* S_raw, float* S_norm, int N, int NA, int Li, int Lqa, int Lr, int D,
                                  float scale, float p_drop, unsigned long long seed, void* stream) {
    if (N <= 0 || Li <= 0) return 0;
    if (D % 16 != 0 || D > 256 || Lr < 1 || Lr > 64 || Lqa < 1 || NA < 1) return STAGE_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int RT = (Lr + 15) / 16;
    switch (RT) {
        case 1: return launch_fwd<1>(Cn, Q, c_mask, q_mask, A, S_raw, S_norm, N, NA, Li, Lqa, Lr, D, scale, p_drop, seed, st);
        case 2: return launch_fwd<2>(Cn, Q, c_mask, q_mask, A, S_raw, S_norm, N, NA, Li, Lqa, Lr, D, scale, p_drop, seed, st);
        case 3: return launch_fwd<3>(Cn, Q, c_mask, q_mask, A, S_raw, S_norm, N, NA, Li, Lqa, Lr, D, scale, p_drop, seed, st);
        default: return launch_fwd<4>(Cn, Q, c_mask, q_mask, A, S_raw, S_norm, N, NA, Li, Lqa, Lr, D, scale, p_drop, seed, st);
    }
}

// ================================================================================================
// Backward.
//   B1 (MFMA, same skeleton as the forward):  dP^T = Qraw . dA^T ;  dZ = S_ * (dP - sum_r S_ dP) ;
//        dS = scale*dZ (+ external dS_raw)  -> written in the layout of S
//   B2 (per frame, fixed-order accumulation over the NA*Lqa rows):  dQraw = S_^T dA ,  dQn = dS^T Cn
//   B3 (per context row, fixed-order accumulation over frames x regions, frame-chunked slabs):  dCn = dS Qn
//   then rowops' l2norm backward turns dQn / dCn into gradients of the raw Q / C.
// ================================================================================================
template <int RT>
__global__ __launch_bounds__(512) void str_attn_bwd_ds_kernel(
    const float* __restrict__ dA, const float* __restrict__ Q, const float* __restrict__ Sn,
    const float* __restrict__ dS_ext, float* __restrict__ dS, int N, int NA, int Li, int Lqa, int Lr, int D,
    float scale) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int LDQ = D + 4;
    float* Qr = lds;  // [RT*16][LDQ]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c15 = lane & 15, g = lane >> 4;
    const int CR = NA * Lqa;
    const int nch = D >> 4;
    const long frame = blockIdx.x;  // n*Li + i
    const int n = (int)(frame / Li), i = (int)(frame % Li);
    const int D4 = D >> 2;
    for (int e = tid; e < RT * 16 * LDQ; e += blockDim.x) Qr[e] = 0.f;
    __syncthreads();
    for (int e = tid; e < Lr * D4; e += blockDim.x) {
        const int r = e / D4, q = e % D4;
        st4(&Qr[r * LDQ + 4 * q], ld4(Q + (frame * Lr + r) * D + 4 * q));
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < TPW; t++) {
        const int c = (wave * TPW + t) * 16 + c15;
        const bool cvalid = c < CR;
        const int a = cvalid ? c / Lqa : 0, w = cvalid ? c % Lqa : 0;
        const long orow = ((long)(n * NA + a) * Li + i) * Lqa + w;
        f32x4 acc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; rt++) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int m = 0; m < nch; m++) {
            const float4 gv = cvalid ? ld4(dA + orow * D + 4 * dchunk(g, m, nch)) : f4zero();
            const float gj[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                const float4 qv = ld4(&Qr[(rt * 16 + c15) * LDQ + 4 * dchunk(g, m, nch)]);
                const float qj[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
                for (int j = 0; j < 4; j++) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qj[j], gj[j], acc[rt], 0, 0, 0);
            }
        }
        // acc[rt][k] = dP[c][R = rt*16 + 4g + k]
        float p[RT][4], dot = 0.f;
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int R = rt * 16 + 4 * g + k;
                p[rt][k] = (cvalid && R < Lr) ? Sn[orow * Lr + R] : 0.f;
                dot += p[rt][k] * acc[rt][k];
            }
        dot = cross_row_sum(dot);
        if (cvalid) {
#pragma unroll
            for (int rt = 0; rt < RT; rt++)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int R = rt * 16 + 4 * g + k;
                    if (R < Lr) {
                        float v = scale * p[rt][k] * (acc[rt][k] - dot);
                        if (dS_ext) v += dS_ext[orow * Lr + R];
                        dS[orow * Lr + R] = v;
                    }
                }
        }
    }
}

// B1 for D = 128 and Lr % 4 == 0 (both streams of the published configs): same tiling, but compile-time trip counts, all
// global loads of a tile (8 dA fragments, the S_ quads, the external dS quads) issued up front from clamped addresses,
// 16-byte S_ loads / dS stores (register k of region tile rt holds region rt*16 + 4g + k: 4 consecutive floats).
template <int RT, bool HAS_EXT>
__global__ __launch_bounds__(512) void str_attn_bwd_ds_d128_kernel(
    const float* __restrict__ dA, const float* __restrict__ Q, const float* __restrict__ Sn,
    const float* __restrict__ dS_ext, float* __restrict__ dS, int N, int NA, int Li, int Lqa, int Lr, float scale) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int D = 128, LDQ = D + 4, NC = 8, D4 = 32;
    float* Qr = lds;  // [RT*16][LDQ]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c15 = lane & 15, g = lane >> 4;
    const int CR = NA * Lqa;
    const long frame = blockIdx.x;  // n*Li + i
    const int n = (int)(frame / Li), i = (int)(frame % Li);
    for (int e = tid; e < RT * 16 * D4; e += blockDim.x) {
        const int r = e / D4, q = e % D4;
        const float4 v = ld4(Q + (frame * Lr + min(r, Lr - 1)) * D + 4 * q);
        st4(&Qr[r * LDQ + 4 * q], r < Lr ? v : f4zero());
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < TPW; t++) {
        const int c = (wave * TPW + t) * 16 + c15;
        const bool cvalid = c < CR;
        const int cc = cvalid ? c : CR - 1;
        const long orow = ((long)(n * NA + cc / Lqa) * Li + i) * Lqa + cc % Lqa;
        float4 gv[NC];
        float2 pq[RT][2], eq[RT][2];   // 8-byte pieces: regions rt*16 + 4g + {0,1} and {2,3} (Lr even -> 8-byte aligned)
#pragma unroll
        for (int m = 0; m < NC; m++) gv[m] = ld4(dA + orow * D + 4 * dchunk(g, m, NC));
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                const int r0 = min(rt * 16 + 4 * g + 2 * hh, Lr - 2);
                pq[rt][hh] = *reinterpret_cast<const float2*>(Sn + orow * Lr + r0);
                if (HAS_EXT) eq[rt][hh] = *reinterpret_cast<const float2*>(dS_ext + orow * Lr + r0);
            }
        }
        f32x4 acc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; rt++) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < NC; m++) {
            const float gj[4] = {gv[m].x, gv[m].y, gv[m].z, gv[m].w};
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                const float4 qv = ld4(&Qr[(rt * 16 + c15) * LDQ + 4 * dchunk(g, m, NC)]);
                const float qj[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
                for (int j = 0; j < 4; j++) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qj[j], gj[j], acc[rt], 0, 0, 0);
            }
        }
        // acc[rt][k] = dP[c][R = rt*16 + 4g + k]
        float p[RT][4], dot = 0.f;
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            const bool rok0 = rt * 16 + 4 * g < Lr, rok1 = rt * 16 + 4 * g + 2 < Lr;
            p[rt][0] = rok0 ? pq[rt][0].x : 0.f;
            p[rt][1] = rok0 ? pq[rt][0].y : 0.f;
            p[rt][2] = rok1 ? pq[rt][1].x : 0.f;
            p[rt][3] = rok1 ? pq[rt][1].y : 0.f;
#pragma unroll
            for (int k = 0; k < 4; k++) dot += p[rt][k] * acc[rt][k];
        }
        dot = cross_row_sum(dot);
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                float2 o = make_float2(scale * p[rt][2 * hh] * (acc[rt][2 * hh] - dot),
                                       scale * p[rt][2 * hh + 1] * (acc[rt][2 * hh + 1] - dot));
                if (HAS_EXT) { o.x += eq[rt][hh].x; o.y += eq[rt][hh].y; }
                const int r0 = rt * 16 + 4 * g + 2 * hh;
                if (cvalid && r0 < Lr) *reinterpret_cast<float2*>(dS + orow * Lr + r0) = o;
            }
        }
    }
}

// B2: one workgroup per frame; thread (r, q) walks the NA*Lqa context rows in order.
__global__ __launch_bounds__(256) void str_attn_bwd_dq_kernel(const float* __restrict__ dA, const float* __restrict__ Sn,
                                                              const float* __restrict__ dS, const float* __restrict__ Cn,
                                                              float* __restrict__ dQraw, float* __restrict__ dQn, int N,
                                                              int NA, int Li, int Lqa, int Lr, int D) {
    const long frame = blockIdx.x;
    const int n = (int)(frame / Li), i = (int)(frame % Li);
    const int D4 = D >> 2, CR = NA * Lqa;
    for (int e = threadIdx.x; e < Lr * D4; e += blockDim.x) {
        const int r = e / D4, q = e % D4;
        float4 ar = f4zero(), an = f4zero();
        for (int c = 0; c < CR; c++) {
            const int a = c / Lqa, w = c % Lqa;
            const long orow = ((long)(n * NA + a) * Li + i) * Lqa + w;
            const float p = Sn[orow * Lr + r], ds = dS[orow * Lr + r];
            ar = f4add(ar, f4scale(ld4(dA + orow * D + 4 * q), p));
            an = f4add(an, f4scale(ld4(Cn + ((long)n * CR + c) * D + 4 * q), ds));
        }
        st4(dQraw + (frame * Lr + r) * D + 4 * q, ar);
        st4(dQn + (frame * Lr + r) * D + 4 * q, an);
    }
}

// B3: thread (context row, q) accumulates over a chunk of frames; slabs [chunk][N*CR][D] are then summed in order.
__global__ __launch_bounds__(256) void str_attn_bwd_dc_kernel(const float* __restrict__ dS, const float* __restrict__ Qn,
                                                              float* __restrict__ part, int N, int NA, int Li, int Lqa,
                                                              int Lr, int D, int frames_per_chunk) {
    const int D4 = D >> 2, CR = NA * Lqa;
    const long total = (long)N * CR * D4;
    const int chunk = blockIdx.y;
    const int fa = chunk * frames_per_chunk, fb = min(Li, fa + frames_per_chunk);
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int q = (int)(e % D4);
        const long cr = e / D4;  // n*CR + c
        const int n = (int)(cr / CR), c = (int)(cr % CR);
        const int a = c / Lqa, w = c % Lqa;
        float4 acc = f4zero();
        for (int i = fa; i < fb; i++) {
            const long orow = ((long)(n * NA + a) * Li + i) * Lqa + w;
            const float* pds = dS + orow * Lr;
            const float* pq = Qn + (((long)n * Li + i) * Lr) * D + 4 * q;
            for (int r = 0; r < Lr; r++) acc = f4add(acc, f4scale(ld4(pq + (long)r * D), pds[r]));
        }
        st4(part + ((size_t)chunk * N * CR * D4 + e) * 4, acc);
    }
}

__global__ void str_attn_slab_sum_kernel(const float* __restrict__ part, float* __restrict__ out, int nb, long C4) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= C4) return;
    float4 acc = f4zero();
    for (int b = 0; b < nb; b++) acc = f4add(acc, ld4(part + ((size_t)b * C4 + e) * 4));
    st4(out + e * 4, acc);
}

#define DC_CHUNKS 16

// MFMA versions of B2 / B3 for D % 64 == 0 (str_attn_bwd_mfma.hip)
int stage_str_attn_bwd_dq_mfma(const float* dA, const float* Sn, const float* dS, const float* Cn, float* dQraw,
                                          float* dQn, int N, int NA, int Li, int Lqa, int Lr, int D, void* stream);
int stage_str_attn_bwd_dc_mfma(const float* dS, const float* Qn, float* part, int N, int NA, int Li, int Lqa,
                                          int Lr, int D, int max_chunks, int* nchunks_out, void* stream);

extern "C" size_t stage_str_attn_bwd_ws_bytes(int N, int NA, int Lqa, int D) {
    return (size_t)DC_CHUNKS * N * NA * Lqa * D * sizeof(float);
}

// dS_out: (N,NA,Li,Lqa,Lr) scratch/out ; dQraw, dQn: (N,Li,Lr,D) ; dCn: (N,NA,Lqa,D)
extern "C" int stage_str_attn_bwd(const float* dA, const float* dS_raw_ext, const float* Cn, const float* Q,
                                  const float* Qn, const float* S_norm, float* dS_out, float* dQraw, float* dQn,
                                  float* dCn, int N, int NA, int Li, int Lqa, int Lr, int D, float scale, void* ws,
                                  size_t ws_bytes, void* stream) {
    if (N <= 0 || Li <= 0) return 0;
    if (D % 16 != 0 || D > 256 || Lr < 1 || Lr > 64 || Lqa < 1 || NA < 1) return STAGE_ERR_SHAPE;
    if (ws_bytes < stage_str_attn_bwd_ws_bytes(N, NA, Lqa, D)) return STAGE_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int CR = NA * Lqa, CT = (CR + 15) / 16, nw = (CT + TPW - 1) / TPW;
    if (nw > 8) return STAGE_ERR_SHAPE;
    const int RT = (Lr + 15) / 16;
    const size_t lds = (size_t)RT * 16 * (D + 4) * sizeof(float);
    const dim3 grid(N * Li), block(64 * nw);
    if (lds > 64 * 1024) {
        (void)hipFuncSetAttribute((const void*)str_attn_bwd_ds_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)str_attn_bwd_ds_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    static const bool ds_generic = getenv("STAGE_K1_DS_GENERIC") != nullptr;   // developer switch
    if (D == 128 && (Lr & 1) == 0 && Lr >= 2 && !ds_generic) {
#define LAUNCH_DS(RTV)                                                                                                  \
    do {                                                                                                                \
        if (lds > 64 * 1024) {                                                                                          \
            (void)hipFuncSetAttribute((const void*)str_attn_bwd_ds_d128_kernel<RTV, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
            (void)hipFuncSetAttribute((const void*)str_attn_bwd_ds_d128_kernel<RTV, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        }                                                                                                               \
        if (dS_raw_ext)                                                                                                 \
            hipLaunchKernelGGL((str_attn_bwd_ds_d128_kernel<RTV, true>), grid, block, lds, st, dA, Q, S_norm, dS_raw_ext, dS_out, N, NA, Li, Lqa, Lr, scale); \
        else                                                                                                            \
            hipLaunchKernelGGL((str_attn_bwd_ds_d128_kernel<RTV, false>), grid, block, lds, st, dA, Q, S_norm, dS_raw_ext, dS_out, N, NA, Li, Lqa, Lr, scale); \
    } while (0)
        switch (RT) {
            case 1: LAUNCH_DS(1); break;
            case 2: LAUNCH_DS(2); break;
            case 3: LAUNCH_DS(3); break;
            default: LAUNCH_DS(4); break;
        }
#undef LAUNCH_DS
    } else
    switch (RT) {
        case 1: hipLaunchKernelGGL((str_attn_bwd_ds_kernel<1>), grid, block, lds, st, dA, Q, S_norm, dS_raw_ext, dS_out, N, NA, Li, Lqa, Lr, D, scale); break;
        case 2: hipLaunchKernelGGL((str_attn_bwd_ds_kernel<2>), grid, block, lds, st, dA, Q, S_norm, dS_raw_ext, dS_out, N, NA, Li, Lqa, Lr, D, scale); break;
        case 3: hipLaunchKernelGGL((str_attn_bwd_ds_kernel<3>), grid, block, lds, st, dA, Q, S_norm, dS_raw_ext, dS_out, N, NA, Li, Lqa, Lr, D, scale); break;
        default: hipLaunchKernelGGL((str_attn_bwd_ds_kernel<4>), grid, block, lds, st, dA, Q, S_norm, dS_raw_ext, dS_out, N, NA, Li, Lqa, Lr, D, scale); break;
    }
    STAGE_LAUNCH_CHECK();
    static const bool scalar_bwd = getenv("STAGE_K1_BWD_SCALAR") != nullptr;   // developer switch: pre-MFMA B2/B3
    const bool mfma = (D % 64 == 0) && !scalar_bwd;
    const long total = (long)N * CR * (D / 4);
    int nchunks;
    if (mfma) {
        int rc = stage_str_attn_bwd_dq_mfma(dA, S_norm, dS_out, Cn, dQraw, dQn, N, NA, Li, Lqa, Lr, D, stream);
        if (rc) return rc;
        rc = stage_str_attn_bwd_dc_mfma(dS_out, Qn, (float*)ws, N, NA, Li, Lqa, Lr, D, DC_CHUNKS, &nchunks, stream);
        if (rc) return rc;
    } else {
        hipLaunchKernelGGL(str_attn_bwd_dq_kernel, dim3(N * Li), dim3(256), 0, st, dA, S_norm, dS_out, Cn, dQraw, dQn, N,
                           NA, Li, Lqa, Lr, D);
        STAGE_LAUNCH_CHECK();
        const int fpc = (Li + DC_CHUNKS - 1) / DC_CHUNKS;
        nchunks = (Li + fpc - 1) / fpc;
        hipLaunchKernelGGL(str_attn_bwd_dc_kernel, dim3((unsigned)((total + 255) / 256), nchunks), dim3(256), 0, st,
                           dS_out, Qn, (float*)ws, N, NA, Li, Lqa, Lr, D, fpc);
        STAGE_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(str_attn_slab_sum_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const float*)ws, dCn, nchunks, total);
    STAGE_LAUNCH_CHECK();
    return 0;
}
