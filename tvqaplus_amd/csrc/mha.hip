// K5 -- per-frame self-attention over object regions / words: the scaled-dot-product core of
// MultiHeadedAttention (model/self_attention.py:56-71).  The four D x D projections are stage_gemm_nt calls.
// Quirk kept from the reference (:37-38, 66-67): `mask.view(M,1,L,1) == 0` fills whole *query rows* with -1e9
// (uniform attention over all keys, padded keys included); keys are never masked.
// One workgroup per (sequence m, head h); L <= 64, so the whole L x L score tile lives in LDS and a softmax row is
// one wave64 shuffle reduction.
#include <stdlib.h>
#include "common.h"
#include "../../include/stage_hip.h"

__global__ __launch_bounds__(256) void mha_core_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                           const float* __restrict__ v, const float* __restrict__ mask,
                                                           float* __restrict__ out, float* __restrict__ probs, int L,
                                                           int D, int nh, uint64_t seed, uint32_t th, float inv_keep) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int dk = D / nh, ldk = dk + 1, ldp = L + 1;
    float* qs = sm;
    float* ks = qs + L * ldk;
    float* vs = ks + L * ldk;
    float* ps = vs + L * ldk;  // [L][L+1]
    const long m = blockIdx.x / nh;
    const int h = blockIdx.x % nh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    for (int e = tid; e < L * dk; e += blockDim.x) {
        const int i = e / dk, d = e % dk;
        const long gidx = (m * L + i) * D + h * dk + d;
        qs[i * ldk + d] = q[gidx];
        ks[i * ldk + d] = k[gidx];
        vs[i * ldk + d] = v[gidx];
    }
    __syncthreads();
    const float rs = sqrtf((float)dk);
    for (int e = tid; e < L * L; e += blockDim.x) {
        const int i = e / L, j = e % L;
        float s = 0.f;
        for (int d = 0; d < dk; d++) s += qs[i * ldk + d] * ks[j * ldk + d];
        s = s / rs;
        if (mask[m * L + i] == 0.f) s = -1e9f;
        ps[i * ldp + j] = s;
    }
    __syncthreads();
    for (int i = wave; i < L; i += nw) {
        const float s = lane < L ? ps[i * ldp + lane] : -INFINITY;
        const float mx = wave_max(s);
        const float e = lane < L ? expf(s - mx) : 0.f;
        const float sum = wave_sum(e);
        if (lane < L) {
            float p = e / sum;
            const long pidx = ((m * nh + h) * L + i) * L + lane;
            probs[pidx] = p;
            if (th) p *= drop1(seed, (uint64_t)pidx, th, inv_keep);
            ps[i * ldp + lane] = p;
        }
    }
    __syncthreads();
    for (int e = tid; e < L * dk; e += blockDim.x) {
        const int i = e / dk, d = e % dk;
        float o = 0.f;
        for (int j = 0; j < L; j++) o += ps[i * ldp + j] * vs[j * ldk + d];
        out[(m * L + i) * D + h * dk + d] = o;
    }
}

__global__ __launch_bounds__(256) void mha_core_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ q,
                                                           const float* __restrict__ k, const float* __restrict__ v,
                                                           const float* __restrict__ probs,
                                                           const float* __restrict__ mask, float* __restrict__ dq,
                                                           float* __restrict__ dkk, float* __restrict__ dv, int L, int D,
                                                           int nh, uint64_t seed, uint32_t th, float inv_keep) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int dk = D / nh, ldk = dk + 1, ldp = L + 1;
    float* qs = sm;
    float* ks = qs + L * ldk;
    float* vs = ks + L * ldk;
    float* gs = vs + L * ldk;   // dout head slice
    float* pd = gs + L * ldk;   // dropped probabilities P'   [L][L+1]
    float* ds = pd + L * ldp;   // dP -> dS                   [L][L+1]
    const long m = blockIdx.x / nh;
    const int h = blockIdx.x % nh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    for (int e = tid; e < L * dk; e += blockDim.x) {
        const int i = e / dk, d = e % dk;
        const long gidx = (m * L + i) * D + h * dk + d;
        qs[i * ldk + d] = q[gidx];
        ks[i * ldk + d] = k[gidx];
        vs[i * ldk + d] = v[gidx];
        gs[i * ldk + d] = dout[gidx];
    }
    __syncthreads();
    // dP'[i][j] = <dout_i, v_j> ; dP = dP' * dropmult ; P' = P * dropmult
    for (int e = tid; e < L * L; e += blockDim.x) {
        const int i = e / L, j = e % L;
        float s = 0.f;
        for (int d = 0; d < dk; d++) s += gs[i * ldk + d] * vs[j * ldk + d];
        const long pidx = ((m * nh + h) * L + i) * L + j;
        const float p = probs[pidx];
        const float mult = th ? drop1(seed, (uint64_t)pidx, th, inv_keep) : 1.0f;
        pd[i * ldp + j] = p * mult;
        ds[i * ldp + j] = s * mult;  // dP
    }
    __syncthreads();
    // dS = P * (dP - <P, dP>) ; zero for padded query rows (their scores were overwritten by a constant)
    for (int i = wave; i < L; i += nw) {
        const long pidx = ((m * nh + h) * L + i) * L + lane;
        const float p = lane < L ? probs[pidx] : 0.f;
        const float g = lane < L ? ds[i * ldp + lane] : 0.f;
        const float dot = wave_sum(p * g);
        const bool dead = mask[m * L + i] == 0.f;
        if (lane < L) ds[i * ldp + lane] = dead ? 0.f : p * (g - dot);
    }
    __syncthreads();
    const float rs = sqrtf((float)dk);
    for (int e = tid; e < L * dk; e += blockDim.x) {
        const int i = e / dk, d = e % dk;
        float aq = 0.f, ak = 0.f, av = 0.f;
        for (int j = 0; j < L; j++) {
            aq += ds[i * ldp + j] * ks[j * ldk + d];
            ak += ds[j * ldp + i] * qs[j * ldk + d];
            av += pd[j * ldp + i] * gs[j * ldk + d];
        }
        const long gidx = (m * L + i) * D + h * dk + d;
        dq[gidx] = aq / rs;
        dkk[gidx] = ak / rs;
        dv[gidx] = av;
    }
}

// matrix-core kernels (mha_mfma.hip): dk in {8, 16, 32, 64}; they recompute the probabilities in the backward
extern "C" int stage_mha_core_recomputes(int L, int D, int nh);
int stage_mha_fwd_mfma(const float* q, const float* k, const float* v, const float* mask, float* out, long long M, int L, int D,
                       int nh, float p_drop, unsigned long long seed, void* stream);
int stage_mha_bwd_mfma(const float* dout, const float* q, const float* k, const float* v, const float* mask, float* dq, float* dk_out,
                       float* dv, long long M, int L, int D, int nh, float p_drop, unsigned long long seed, void* stream);
static bool mha_use_mfma(int L, int D, int nh) {
    const bool scalar = getenv("STAGE_MHA_SCALAR") != nullptr;   // developer switch, read per call (cross-check in the tests)
    return !scalar && stage_mha_core_recomputes(L, D, nh) != 0;
}

extern "C" int stage_mha_core_fwd(const float* q, const float* k, const float* v, const float* mask, float* out,
                                  float* probs, long long M, int L, int D, int nh, float p_drop,
                                  unsigned long long seed, void* stream) {
    if (M <= 0) return 0;
    if (L < 1 || L > 64 || nh < 1 || D % nh != 0) return STAGE_ERR_SHAPE;
    if (mha_use_mfma(L, D, nh)) return stage_mha_fwd_mfma(q, k, v, mask, out, M, L, D, nh, p_drop, seed, stream);
    if (!probs) return STAGE_ERR_SHAPE;     // the scalar kernels keep the probabilities for the backward
    const int dk = D / nh;
    const size_t lds = ((size_t)3 * L * (dk + 1) + (size_t)L * (L + 1)) * sizeof(float);
    uint32_t th = p_drop > 0.f ? drop_thresh16(p_drop) : 0u;
    if (p_drop > 0.f && th == 0u) th = 1u;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)mha_core_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(mha_core_fwd_kernel, dim3((unsigned)(M * nh)), dim3(256), lds, (hipStream_t)stream, q, k, v, mask,
                       out, probs, L, D, nh, (uint64_t)seed, th, p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f);
    STAGE_LAUNCH_CHECK();
    return 0;
}

extern "C" int stage_mha_core_bwd(const float* dout, const float* q, const float* k, const float* v, const float* probs,
                                  const float* mask, float* dq, float* dk_out, float* dv, long long M, int L, int D,
                                  int nh, float p_drop, unsigned long long seed, void* stream) {
    if (M <= 0) return 0;
    if (L < 1 || L > 64 || nh < 1 || D % nh != 0) return STAGE_ERR_SHAPE;
    if (mha_use_mfma(L, D, nh)) return stage_mha_bwd_mfma(dout, q, k, v, mask, dq, dk_out, dv, M, L, D, nh, p_drop, seed, stream);
    if (!probs) return STAGE_ERR_SHAPE;
    const int dk = D / nh;
    const size_t lds = ((size_t)4 * L * (dk + 1) + (size_t)2 * L * (L + 1)) * sizeof(float);
    uint32_t th = p_drop > 0.f ? drop_thresh16(p_drop) : 0u;
    if (p_drop > 0.f && th == 0u) th = 1u;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)mha_core_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(mha_core_bwd_kernel, dim3((unsigned)(M * nh)), dim3(256), lds, (hipStream_t)stream, dout, q, k, v,
                       probs, mask, dq, dk_out, dv, L, D, nh, (uint64_t)seed, th,
                       p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f);
    STAGE_LAUNCH_CHECK();
    return 0;
}
