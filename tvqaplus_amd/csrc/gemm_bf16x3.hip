// fp32 GEMMs as split-bf16 products on the bf16 matrix cores (MI355X: bf16 MFMA runs at 16x the fp32-MFMA rate).
//
// An fp32 value is split EXACTLY into three bf16 terms (3 x 8 mantissa bits = fp32's 24):
//     x1 = bf16(x),  x2 = bf16(x - x1),  x3 = bf16(x - x1 - x2)        x == x1 + x2 + x3
// and a product keeps the six terms of order <= 2^-16:
//     x*w = x1w1 + (x1w2 + x2w1) + (x1w3 + x2w2 + x3w1)  [+ dropped x2w3 + x3w2 + x3w3 ~ 2^-24 |xw|]
// Every bf16 x bf16 product is exact in fp32 and v_mfma_f32_32x32x16_bf16 accumulates in fp32, so the result has
// fp32-level accuracy (same order as the f32 MFMA chain) while a 128x128x32 block chunk costs 1536 matrix-core cycles
// instead of 4096: every Linear / 1x1-conv of the STAGE path turns from MFMA-bound (~55 TFLOP/s fp32 in the step) into
// an HBM-streaming kernel.  NSPLIT = 2 (three products, ~1e-5 relative) is kept for experiments (STAGE_GEMM_SPLIT=2).
// Same entry-point semantics as gemm.hip (gate = fused ReLU backward, bias / ReLU / residual epilogue, split-M
// deterministic weight gradients); used by stage_gemm_nt / stage_gemm_tn unless STAGE_GEMM_F32 is set.
//
// LDS image: NSPLIT bf16 planes per operand, [128 rows][32 k] with an 80-byte row stride (5 x 16 B: the 16-lane service
// groups of ds_read_b128 hit distinct 16-B slots).  The TN form (weight gradient, contraction over rows) writes the
// planes TRANSPOSED ([column][m]) while staging, packing two consecutive rows per 32-bit LDS write.
#include <stdlib.h>
#include "common.h"
#include "../../include/stage_hip.h"

#define BM 128
#define BN 128
#define BK 32
#define XS 40               // bf16 per LDS row (32 + 8 pad)
#define PLANE (BM * XS)     // bf16 elements per plane

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// (a, b) -> NSPLIT packed bf16 pairs (a in the low half): successive exact residual splits
template <int NSPLIT>
__device__ __forceinline__ void splitn(float a, float b, unsigned (&out)[NSPLIT]) {
#pragma unroll
    for (int s = 0; s < NSPLIT; s++) {
        out[s] = cvt_pk_bf16(a, b);
        a -= __uint_as_float(out[s] << 16);
        b -= __uint_as_float(out[s] & 0xFFFF0000u);
    }
}
template <typename T>
__device__ __forceinline__ float4 load4_guard_b(const T* __restrict__ base, long row, long ld, int col, long nrows,
                                                int ncols, bool vec) {
    if (row >= nrows || col >= ncols) return f4zero();
    const T* p = base + row * ld + col;
    if (vec && col + 3 < ncols) return ldv4(p);
    float4 v = f4zero();
    v.x = ldv1(p);
    if (col + 1 < ncols) v.y = ldv1(p + 1);
    if (col + 2 < ncols) v.z = ldv1(p + 2);
    if (col + 3 < ncols) v.w = ldv1(p + 3);
    return v;
}
// branch-free variant for 16-B aligned operands with ncols % 4 == 0: clamp the address, select zero afterwards.  The
// guarded form above compiles to one exec-masked branch + s_waitcnt per load (loads serialised); this one lets the
// compiler issue a whole fetch back to back under a single wait.
template <typename T>
__device__ __forceinline__ float4 load4_clamp(const T* __restrict__ base, long row, long ld, int col, long nrows,
                                              int ncols) {
    const long r = row < nrows ? row : nrows - 1;
    const int c = col < ncols ? col : ncols - 4;
    float4 v = ldv4(base + r * ld + c);
    const bool ok = row < nrows && col < ncols;
    v.x = ok ? v.x : 0.f;
    v.y = ok ? v.y : 0.f;
    v.z = ok ? v.z : 0.f;
    v.w = ok ? v.w : 0.f;
    return v;
}
template <bool FAST, typename T>
__device__ __forceinline__ float4 load4_b(const T* __restrict__ base, long row, long ld, int col, long nrows, int ncols,
                                          bool vec) {
    if (FAST) return load4_clamp(base, row, ld, col, nrows, ncols);
    return load4_guard_b(base, row, ld, col, nrows, ncols, vec);
}
__device__ __forceinline__ float4 gate4_b(float4 v, float4 g) {
    return make_float4(g.x > 0.f ? v.x : 0.f, g.y > 0.f ? v.y : 0.f, g.z > 0.f ? v.z : 0.f, g.w > 0.f ? v.w : 0.f);
}
__device__ __forceinline__ bf16x8 lds_frag(const unsigned short* p) {
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(p));
}

// the MFMA block of one k-step: all kept cross terms, smallest first
template <int NSPLIT>
__device__ __forceinline__ void mma_terms(f32x16 (&acc)[2][2], const bf16x8 (&a)[NSPLIT][2], const bf16x8 (&b)[NSPLIT][2]) {
#pragma unroll
    for (int order = NSPLIT - 1; order >= 0; order--)   // order = sa + sb
#pragma unroll
        for (int sa = 0; sa <= order; sa++) {
            const int sb = order - sa;
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[sa][i], b[sb][j], acc[i][j], 0, 0, 0);
        }
}

// ------------------------------------------------------------------------------------------------
// NT:  Y = epi(Xg . W^T + bias)
// ------------------------------------------------------------------------------------------------
template <int NSPLIT, bool FAST, typename T = float>   // T: storage of X / gate / residual / Y (float, or bf16 with NSPLIT = 1)
__global__ __launch_bounds__(256, 2) void gemm_nt_split_kernel(const T* __restrict__ X, const T* __restrict__ G,
                                                               const float* __restrict__ W,
                                                               const float* __restrict__ bias,
                                                               const T* __restrict__ R, T* __restrict__ Y,
                                                               long M, int N, int K, int relu, int vecX, int vecW) {
    __shared__ __attribute__((aligned(16))) unsigned short Ap[NSPLIT * PLANE], Bp[NSPLIT * PLANE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    const long m0 = (long)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int lrow = tid >> 3, lcol = (tid & 7) * 4;  // staging: 32 rows x 8 float4 per pass, 4 passes

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    float4 pa[4], pb[4], pg[4];
    auto fetch = [&](int k0) {   // loads only (the gate is applied in stash): one uninterrupted burst, one wait
#pragma unroll
        for (int p = 0; p < 4; p++) {
            pa[p] = load4_b<FAST>(X, m0 + lrow + 32 * p, K, k0 + lcol, M, K, vecX);
            pb[p] = load4_b<FAST>(W, n0 + lrow + 32 * p, K, k0 + lcol, N, K, vecW);
        }
        if (G) {
#pragma unroll
            for (int p = 0; p < 4; p++) pg[p] = load4_b<FAST>(G, m0 + lrow + 32 * p, K, k0 + lcol, M, K, vecX);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int off = (lrow + 32 * p) * XS + lcol;
            unsigned s01[NSPLIT], s23[NSPLIT];
            const float4 av = G ? gate4_b(pa[p], pg[p]) : pa[p];
            splitn<NSPLIT>(av.x, av.y, s01);
            splitn<NSPLIT>(av.z, av.w, s23);
#pragma unroll
            for (int s = 0; s < NSPLIT; s++) *reinterpret_cast<uint2*>(&Ap[s * PLANE + off]) = make_uint2(s01[s], s23[s]);
            splitn<NSPLIT>(pb[p].x, pb[p].y, s01);
            splitn<NSPLIT>(pb[p].z, pb[p].w, s23);
#pragma unroll
            for (int s = 0; s < NSPLIT; s++) *reinterpret_cast<uint2*>(&Bp[s * PLANE + off]) = make_uint2(s01[s], s23[s]);
        }
    };

    fetch(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        __syncthreads();
        stash();
        __syncthreads();
        if (k0 + BK < K) fetch(k0 + BK);
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            const int koff = (2 * ks + h) * 8;
            bf16x8 a[NSPLIT][2], b[NSPLIT][2];
#pragma unroll
            for (int s = 0; s < NSPLIT; s++)
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    a[s][i] = lds_frag(&Ap[s * PLANE + (wm * 64 + i * 32 + l31) * XS + koff]);
                    b[s][i] = lds_frag(&Bp[s * PLANE + (wn * 64 + i * 32 + l31) * XS + koff]);
                }
            mma_terms<NSPLIT>(acc, a, b);
        }
    }
    // epilogue.  C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
        for (int ni = 0; ni < 2; ni++) {
            const int n = n0 + wn * 64 + ni * 32 + l31;
            if (n >= N) continue;
            const float bsv = bias ? bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const long m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < M) {
                    float v = acc[mi][ni][r] + bsv;
                    if (relu) v = fmaxf(v, 0.f);
                    if (R) v += ldv1(R + m * N + n);
                    stv1(Y + m * N + n, v);
                }
            }
        }
}

static int split_depth() {
    static int d = -1;
    if (d < 0) d = (getenv("STAGE_GEMM_SPLIT") && atoi(getenv("STAGE_GEMM_SPLIT")) == 2) ? 2 : 3;
    return d;
}

extern "C" int stage_gemm_nt_bf16x3(const float* X, const float* gate, const float* W, const float* bias,
                                    const float* residual, float* Y, long long M, int N, int K, int relu,
                                    void* stream) {
    if (M <= 0 || N <= 0) return 0;
    if (K <= 0) return STAGE_ERR_SHAPE;
    const int vecX = (K % 4 == 0) && (((uintptr_t)X & 15) == 0) && (!gate || ((uintptr_t)gate & 15) == 0);
    const int vecW = (K % 4 == 0) && (((uintptr_t)W & 15) == 0);
    dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((N + BN - 1) / BN));
    const bool fast = vecX && vecW && K >= 4;
#define LAUNCH_NT(NS, F)                                                                                              \
    hipLaunchKernelGGL((gemm_nt_split_kernel<NS, F>), grid, dim3(256), 0, (hipStream_t)stream, X, gate, W, bias, residual, \
                       Y, (long)M, N, K, relu, vecX, vecW)
    if (split_depth() == 2) { if (fast) LAUNCH_NT(2, true); else LAUNCH_NT(2, false); }
    else { if (fast) LAUNCH_NT(3, true); else LAUNCH_NT(3, false); }
#undef LAUNCH_NT
    STAGE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// TN:  partial[s][n][k] = sum_{m in slab s} Yg[m,n] X[m,k] ; partial_b[s][n] = sum_{m in slab s} Yg[m,n]
// ------------------------------------------------------------------------------------------------
#define TN_MAX_SPLIT 128

template <int NSPLIT, bool FAST, typename T = float>
__global__ __launch_bounds__(256, 2) void gemm_tn_split_kernel(const T* __restrict__ dY, const T* __restrict__ G,
                                                               const T* __restrict__ X, float* __restrict__ part,
                                                               float* __restrict__ part_b, long M, int N, int K,
                                                               long rows_per_split, int vecY, int vecX) {
    // transposed planes: [column][m]
    __shared__ __attribute__((aligned(16))) unsigned short Ap[NSPLIT * PLANE], Bp[NSPLIT * PLANE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * BM;  // output rows = n
    const int k0 = blockIdx.y * BN;  // output cols = k
    const int split = blockIdx.z;
    const long mbeg = (long)split * rows_per_split;
    const long mend = min(M, mbeg + rows_per_split);
    const int c4 = (tid & 31) * 4, mp = tid >> 5;  // staging: columns c4..c4+3, row pair 2*mp (+16 per pass)

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    float4 bsum = f4zero();

    float4 py[2][2], px[2][2], pg[2][2];  // [pass][row of the pair]
    auto fetch = [&](long mb) {   // loads only; the gate is applied in stash
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const long row = mb + 16 * p + 2 * mp + e;
                py[p][e] = load4_b<FAST>(dY, row, N, n0 + c4, mend, N, vecY);
                px[p][e] = load4_b<FAST>(X, row, K, k0 + c4, mend, K, vecX);
            }
        if (G) {
#pragma unroll
            for (int p = 0; p < 2; p++)
#pragma unroll
                for (int e = 0; e < 2; e++)
                    pg[p][e] = load4_b<FAST>(G, mb + 16 * p + 2 * mp + e, N, n0 + c4, mend, N, vecY);
        }
    };
    auto put = [&](unsigned short* P, int col, int m, float a, float b) {
        unsigned s[NSPLIT];
        splitn<NSPLIT>(a, b, s);
#pragma unroll
        for (int q = 0; q < NSPLIT; q++) *reinterpret_cast<unsigned*>(&P[q * PLANE + col * XS + m]) = s[q];
    };
    auto stash = [&]() {
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const int m = 16 * p + 2 * mp;
            if (G) {
                py[p][0] = gate4_b(py[p][0], pg[p][0]);
                py[p][1] = gate4_b(py[p][1], pg[p][1]);
            }
            put(Ap, c4 + 0, m, py[p][0].x, py[p][1].x);
            put(Ap, c4 + 1, m, py[p][0].y, py[p][1].y);
            put(Ap, c4 + 2, m, py[p][0].z, py[p][1].z);
            put(Ap, c4 + 3, m, py[p][0].w, py[p][1].w);
            put(Bp, c4 + 0, m, px[p][0].x, px[p][1].x);
            put(Bp, c4 + 1, m, px[p][0].y, px[p][1].y);
            put(Bp, c4 + 2, m, px[p][0].z, px[p][1].z);
            put(Bp, c4 + 3, m, px[p][0].w, px[p][1].w);
            bsum = f4add(bsum, f4add(py[p][0], py[p][1]));
        }
    };

    if (mbeg < mend) fetch(mbeg);
    for (long mb = mbeg; mb < mend; mb += BK) {
        __syncthreads();
        stash();
        __syncthreads();
        if (mb + BK < mend) fetch(mb + BK);
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            const int koff = (2 * ks + h) * 8;
            bf16x8 a[NSPLIT][2], b[NSPLIT][2];
#pragma unroll
            for (int s = 0; s < NSPLIT; s++)
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    a[s][i] = lds_frag(&Ap[s * PLANE + (wm * 64 + i * 32 + l31) * XS + koff]);
                    b[s][i] = lds_frag(&Bp[s * PLANE + (wn * 64 + i * 32 + l31) * XS + koff]);
                }
            mma_terms<NSPLIT>(acc, a, b);
        }
    }
    float* po = part + (size_t)split * N * K;
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
        for (int ni = 0; ni < 2; ni++) {
            const int k = k0 + wn * 64 + ni * 32 + l31;
            if (k >= K) continue;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int n = n0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (n < N) po[(size_t)n * K + k] = acc[mi][ni][r];
            }
        }
    // bias-gradient partial (exact fp32 column sums of the gated dY; only the k-tile 0 blocks emit it)
    if (part_b && blockIdx.y == 0) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(Ap);  // [8][BM] floats = 4 KB, planes are free now
        st4(&red[mp * BM + c4], bsum);
        __syncthreads();
        if (tid < BM) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 8; q++) s += red[q * BM + tid];
            if (n0 + tid < N) part_b[(size_t)split * N + n0 + tid] = s;
        }
    }
}

__global__ void slab_reduce_b_kernel(const float* __restrict__ part, float* __restrict__ out, int nb, long C) {
    long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float acc = 0.f;
#pragma unroll 8
    for (int b = 0; b < nb; b++) acc += part[(size_t)b * C + c];
    out[c] = acc;
}

static int tn_splits_b(long long M, int N, int K) {
    const long tiles = (long)((N + BM - 1) / BM) * ((K + BN - 1) / BN);
    long s = (1024 + tiles - 1) / tiles;
    const long max_by_rows = (M + 8 * BK - 1) / (8 * BK);
    if (s > max_by_rows) s = max_by_rows;
    if (s > TN_MAX_SPLIT) s = TN_MAX_SPLIT;
    if (s < 1) s = 1;
    return (int)s;
}

// workspace size is the same as stage_gemm_tn_ws_bytes (same split rule)
extern "C" int stage_gemm_tn_bf16x3(const float* dY, const float* gate, const float* X, float* dW, float* db, long long M,
                                    int N, int K, void* ws, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (N <= 0 || K <= 0) return 0;
    if (M <= 0) {
        (void)hipMemsetAsync(dW, 0, sizeof(float) * (size_t)N * K, st);
        if (db) (void)hipMemsetAsync(db, 0, sizeof(float) * N, st);
        return 0;
    }
    const int S = tn_splits_b(M, N, K);
    if (ws_bytes < (size_t)S * ((size_t)N * K + N) * sizeof(float)) return STAGE_ERR_WORKSPACE;
    long rps = (M + S - 1) / S;
    rps = (rps + BK - 1) / BK * BK;
    float* part = (float*)ws;
    float* part_b = part + (size_t)S * N * K;
    const int vecY = (N % 4 == 0) && (((uintptr_t)dY & 15) == 0) && (!gate || ((uintptr_t)gate & 15) == 0);
    const int vecX = (K % 4 == 0) && (((uintptr_t)X & 15) == 0);
    dim3 grid((N + BM - 1) / BM, (K + BN - 1) / BN, S);
    const bool fast = vecY && vecX && N >= 4 && K >= 4;
#define LAUNCH_TN(NS, F)                                                                                                 \
    hipLaunchKernelGGL((gemm_tn_split_kernel<NS, F>), grid, dim3(256), 0, st, dY, gate, X, part,                          \
                       db ? part_b : (float*)nullptr, (long)M, N, K, rps, vecY, vecX)
    if (split_depth() == 2) { if (fast) LAUNCH_TN(2, true); else LAUNCH_TN(2, false); }
    else { if (fast) LAUNCH_TN(3, true); else LAUNCH_TN(3, false); }
#undef LAUNCH_TN
    STAGE_LAUNCH_CHECK();
    const long C = (long)N * K;
    hipLaunchKernelGGL(slab_reduce_b_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, st, part, dW, S, C);
    if (db) hipLaunchKernelGGL(slab_reduce_b_kernel, dim3((N + 255) / 256), dim3(256), 0, st, part_b, db, S, (long)N);
    STAGE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// bf16 storage mode (BASELINE.json configs[4]: bf16 weights / activations): X, gate, residual and Y are bf16, the fp32
// master weight is rounded to bf16 while it is staged (NSPLIT = 1: ONE bf16 product per term, exact in fp32, fp32
// accumulation); bias, weight / bias gradients stay fp32.  8-byte alignment of the bf16 operands selects the vector path.
// ------------------------------------------------------------------------------------------------
int stage_gemm_nt_bf16_stream(const void* X, const void* gate, const float* W, const float* bias, void* Y, long long M, int N,
                              int K, int relu, void* stream);   // gemm_bf16_stream.hip; 1 = shape not handled there

extern "C" int stage_gemm_nt_bf16(const void* X, const void* gate, const float* W, const float* bias, const void* residual,
                                  void* Y, long long M, int N, int K, int relu, void* stream) {
    if (M <= 0 || N <= 0) return 0;
    if (K <= 0) return STAGE_ERR_SHAPE;
    if (!residual) {
        const int rc = stage_gemm_nt_bf16_stream(X, gate, W, bias, Y, M, N, K, relu, stream);
        if (rc != 1) return rc;
    }
    const int vecX = (K % 4 == 0) && (((uintptr_t)X & 7) == 0) && (!gate || ((uintptr_t)gate & 7) == 0);
    const int vecW = (K % 4 == 0) && (((uintptr_t)W & 15) == 0);
    dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((N + BN - 1) / BN));
    const bool fast = vecX && vecW && K >= 4;
    typedef stage_bf16 B;
    if (fast)
        hipLaunchKernelGGL((gemm_nt_split_kernel<1, true, B>), grid, dim3(256), 0, (hipStream_t)stream, (const B*)X, (const B*)gate, W,
                           bias, (const B*)residual, (B*)Y, (long)M, N, K, relu, vecX, vecW);
    else
        hipLaunchKernelGGL((gemm_nt_split_kernel<1, false, B>), grid, dim3(256), 0, (hipStream_t)stream, (const B*)X, (const B*)gate, W,
                           bias, (const B*)residual, (B*)Y, (long)M, N, K, relu, vecX, vecW);
    STAGE_LAUNCH_CHECK();
    return 0;
}

size_t stage_gemm_tn_bf16_stream_ws_bytes(long long M, int N, int K);                       // gemm_bf16_stream.hip
size_t stage_gemm_tn_bf16_oct_ws_bytes(long long M, int N, int K);                          // gemm_bf16_oct.hip
int stage_gemm_tn_bf16_oct(const void* dY, const void* gate, const void* X, float* dW, float* db, long long M, int N, int K, void* ws,
                           size_t ws_bytes, void* stream);
int stage_gemm_tn_bf16_stream(const void* dY, const void* gate, const void* X, float* part, float* part_b, long long M, int N,
                              int K, int* slabs, void* stream);                             // 1 = shape not handled there

extern "C" size_t stage_gemm_tn_bf16_ws_bytes(long long M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const size_t tiled = (size_t)tn_splits_b(M, N, K) * ((size_t)N * K + N) * sizeof(float);
    const size_t strm = stage_gemm_tn_bf16_stream_ws_bytes(M, N, K);
    const size_t oct = stage_gemm_tn_bf16_oct_ws_bytes(M, N, K);
    const size_t m = tiled > strm ? tiled : strm;
    return m > oct ? m : oct;
}

// workspace: stage_gemm_tn_bf16_ws_bytes(M, N, K)
extern "C" int stage_gemm_tn_bf16(const void* dY, const void* gate, const void* X, float* dW, float* db, long long M, int N,
                                  int K, void* ws, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (N <= 0 || K <= 0) return 0;
    if (M <= 0) {
        (void)hipMemsetAsync(dW, 0, sizeof(float) * (size_t)N * K, st);
        if (db) (void)hipMemsetAsync(db, 0, sizeof(float) * N, st);
        return 0;
    }
    {   // wide layers (128 < N <= 256): row-wise 16-byte loads, 256 x 256 tiles (gemm_bf16_oct.hip)
        const int rc = stage_gemm_tn_bf16_oct(dY, gate, X, dW, db, M, N, K, ws, ws_bytes, stream);
        if (rc != 1) return rc;
    }
    {   // streaming kernel (gemm_bf16_stream.hip) when the shape allows it and the workspace holds its partials
        const size_t need = stage_gemm_tn_bf16_stream_ws_bytes(M, N, K);
        if (need && ws_bytes >= need) {
            const size_t nslab = need / (((size_t)N * K + N) * sizeof(float));
            float* part = (float*)ws;
            float* part_b = part + nslab * (size_t)N * K;
            int slabs = 0;
            const int rc = stage_gemm_tn_bf16_stream(dY, gate, X, part, db ? part_b : (float*)nullptr, M, N, K, &slabs, stream);
            if (rc == 0) {
                const long C = (long)N * K;
                hipLaunchKernelGGL(slab_reduce_b_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, st, part, dW, slabs, C);
                if (db) hipLaunchKernelGGL(slab_reduce_b_kernel, dim3((N + 255) / 256), dim3(256), 0, st, part_b, db, slabs, (long)N);
                STAGE_LAUNCH_CHECK();
                return 0;
            }
            if (rc != 1) return rc;
        }
    }
    const int S = tn_splits_b(M, N, K);
    if (ws_bytes < (size_t)S * ((size_t)N * K + N) * sizeof(float)) return STAGE_ERR_WORKSPACE;
    long rps = (M + S - 1) / S;
    rps = (rps + BK - 1) / BK * BK;
    float* part = (float*)ws;
    float* part_b = part + (size_t)S * N * K;
    const int vecY = (N % 4 == 0) && (((uintptr_t)dY & 7) == 0) && (!gate || ((uintptr_t)gate & 7) == 0);
    const int vecX = (K % 4 == 0) && (((uintptr_t)X & 7) == 0);
    dim3 grid((N + BM - 1) / BM, (K + BN - 1) / BN, S);
    const bool fast = vecY && vecX && N >= 4 && K >= 4;
    typedef stage_bf16 B;
    if (fast)
        hipLaunchKernelGGL((gemm_tn_split_kernel<1, true, B>), grid, dim3(256), 0, st, (const B*)dY, (const B*)gate, (const B*)X, part,
                           db ? part_b : (float*)nullptr, (long)M, N, K, rps, vecY, vecX);
    else
        hipLaunchKernelGGL((gemm_tn_split_kernel<1, false, B>), grid, dim3(256), 0, st, (const B*)dY, (const B*)gate, (const B*)X, part,
                           db ? part_b : (float*)nullptr, (long)M, N, K, rps, vecY, vecX);
    STAGE_LAUNCH_CHECK();
    const long C = (long)N * K;
    hipLaunchKernelGGL(slab_reduce_b_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, st, part, dW, S, C);
    if (db) hipLaunchKernelGGL(slab_reduce_b_kernel, dim3((N + 255) / 256), dim3(256), 0, st, part_b, db, S, (long)N);
    STAGE_LAUNCH_CHECK();
    return 0;
}
