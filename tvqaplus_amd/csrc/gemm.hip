// fp32 GEMMs on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact f32, == an fmaf chain) for every dense
// projection of the STAGE hot path: nn.Linear in model/stage.py:85-120,133-138 + LinearWrapper :15-32, the 1x1
// pointwise Conv1d of model/cnn.py:27-28 and the four MHA projections of model/self_attention.py:32.
//
//   stage_gemm_nt : Y[M,N] = epi( (X (*) gate)[M,K] . W[N,K]^T + bias )      epi = ReLU / + residual
//                   (forward linear; also dX = (dY (*) relu-gate) . (W^T)^T with a pre-transposed weight)
//   stage_gemm_tn : dW[N,K] = sum_m (dY (*) gate)[m,n] X[m,k] ,  db[N] = sum_m (dY (*) gate)[m,n]
//                   split over M into deterministic partial slabs reduced in a fixed order.
//
// Tiling: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each 64x64 = 2x2 MFMA 32x32 tiles,
// 64 accumulator VGPRs), K/M chunks of 32 staged through LDS with a register prefetch of the next chunk.
// Lane half h = lane>>5 owns k = 16h + ks of a chunk (any k-permutation is legal as long as A and B agree), so the
// NT form reads 16 consecutive k per row with four ds_read_b128 (row stride 36 floats -> conflict-free).
#include <stdlib.h>
#include "common.h"
#include "../../include/stage_hip.h"

#define BM 128
#define BN 128
#define BK 32
#define LDS_STRIDE 36  // floats; 144 B rows keep ds_read_b128 16-B aligned and spread the 16 slots

__device__ __forceinline__ float4 load4_guard(const float* __restrict__ base, long row, long ld, int col, long nrows,
                                              int ncols, bool vec) {
    if (row >= nrows || col >= ncols) return f4zero();
    const float* p = base + row * ld + col;
    if (vec && col + 3 < ncols) return ld4(p);
    float4 v = f4zero();
    v.x = p[0];
    if (col + 1 < ncols) v.y = p[1];
    if (col + 2 < ncols) v.z = p[2];
    if (col + 3 < ncols) v.w = p[3];
    return v;
}
// branch-free variant (16-B aligned operands, ncols % 4 == 0): clamp the address, select zero afterwards -- the guarded
// form compiles to an exec-masked branch + s_waitcnt per load, which serialises the whole fetch
__device__ __forceinline__ float4 load4_clamp_f(const float* __restrict__ base, long row, long ld, int col, long nrows,
                                                int ncols) {
    const long r = row < nrows ? row : nrows - 1;
    const int c = col < ncols ? col : ncols - 4;
    float4 v = ld4(base + r * ld + c);
    const bool ok = row < nrows && col < ncols;
    v.x = ok ? v.x : 0.f;
    v.y = ok ? v.y : 0.f;
    v.z = ok ? v.z : 0.f;
    v.w = ok ? v.w : 0.f;
    return v;
}
template <bool FAST>
__device__ __forceinline__ float4 load4_f(const float* __restrict__ base, long row, long ld, int col, long nrows, int ncols,
                                          bool vec) {
    if (FAST) return load4_clamp_f(base, row, ld, col, nrows, ncols);
    return load4_guard(base, row, ld, col, nrows, ncols, vec);
}
__device__ __forceinline__ float4 gate4(float4 v, float4 g) {
    return make_float4(g.x > 0.f ? v.x : 0.f, g.y > 0.f ? v.y : 0.f, g.z > 0.f ? v.z : 0.f, g.w > 0.f ? v.w : 0.f);
}

// ------------------------------------------------------------------------------------------------
// NT:  Y = epi(Xg . W^T + bias)
// ------------------------------------------------------------------------------------------------
template <bool FAST>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const float* __restrict__ X, const float* __restrict__ G,
                                                         const float* __restrict__ W, const float* __restrict__ bias,
                                                         const float* __restrict__ R, float* __restrict__ Y, long M,
                                                         int N, int K, int relu, int vecX, int vecW) {
    __shared__ __attribute__((aligned(16))) float As[BM * LDS_STRIDE];
    __shared__ __attribute__((aligned(16))) float Bs[BN * LDS_STRIDE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    // 1-D grid, n-tile fastest: the workgroups that share a row block of X are dispatched back to back (L2 hits)
    const int n_tiles = (N + BN - 1) / BN;
    const long m0 = (long)(blockIdx.x / n_tiles) * BM;
    const int n0 = (int)(blockIdx.x % n_tiles) * BN;
    const int lrow = tid >> 3, lcol = (tid & 7) * 4;  // staging: 32 rows x 8 float4 per pass, 4 passes

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    float4 pa[4], pb[4], pg[4];
    auto fetch = [&](int k0) {   // loads only (gate applied in stash): one burst, one wait
#pragma unroll
        for (int p = 0; p < 4; p++) {
            pa[p] = load4_f<FAST>(X, m0 + lrow + 32 * p, K, k0 + lcol, M, K, vecX);
            pb[p] = load4_f<FAST>(W, n0 + lrow + 32 * p, K, k0 + lcol, N, K, vecW);
        }
        if (G) {
#pragma unroll
            for (int p = 0; p < 4; p++) pg[p] = load4_f<FAST>(G, m0 + lrow + 32 * p, K, k0 + lcol, M, K, vecX);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int p = 0; p < 4; p++) {
            if (G) pa[p] = gate4(pa[p], pg[p]);
            st4(&As[(lrow + 32 * p) * LDS_STRIDE + lcol], pa[p]);
            st4(&Bs[(lrow + 32 * p) * LDS_STRIDE + lcol], pb[p]);
        }
    };

    fetch(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        __syncthreads();  // previous chunk fully consumed
        stash();
        __syncthreads();
        if (k0 + BK < K) fetch(k0 + BK);
        const float* ap = &As[(wm * 64 + l31) * LDS_STRIDE + 16 * h];
        const float* bp = &Bs[(wn * 64 + l31) * LDS_STRIDE + 16 * h];
#pragma unroll
        for (int kg = 0; kg < 4; kg++) {
            float4 a0 = ld4(ap + 4 * kg), a1 = ld4(ap + 32 * LDS_STRIDE + 4 * kg);
            float4 b0 = ld4(bp + 4 * kg), b1 = ld4(bp + 32 * LDS_STRIDE + 4 * kg);
            const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
            const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[j], bv0[j], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[j], bv1[j], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[j], bv0[j], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[j], bv1[j], acc[1][1], 0, 0, 0);
            }
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
                    if (R) v += R[m * N + n];
                    Y[m * N + n] = v;
                }
            }
        }
}

// Shapes the streaming kernels (gemm_stream.hip: two-way fp16 split) do not take: forward-form GEMMs (NT) default to the
// tiled split-bf16 kernel with the exact 3-way split (gemm_bf16x3.hip: fp32-level accuracy, ~15-25 % faster than the f32 MFMA
// at these shapes); the weight-gradient form (TN) stays on the f32 MFMA, where the transposing split staging costs more than it saves.  STAGE_GEMM_F32=1 forces f32 everywhere,
// STAGE_GEMM_SPLIT_TN=1 also routes TN through the split kernel.
extern "C" int stage_gemm_nt_bf16x3(const float* X, const float* gate, const float* W, const float* bias,
                                    const float* residual, float* Y, long long M, int N, int K, int relu, void* stream);
extern "C" int stage_gemm_tn_bf16x3(const float* dY, const float* gate, const float* X, float* dW, float* db, long long M,
                                    int N, int K, void* ws, size_t ws_bytes, void* stream);
// gemm_stream.hip (gate_kind: 0 none, 1 fp32 tensor, 2 bit mask; return 1 = shape not handled there)
int stage_gemm_nt_stream(const float* X, const void* gate, int gate_kind, const float* W, const float* bias,
                         const float* residual, float* Y, unsigned* mask_out, long long M, int N, int K, int relu,
                         void* stream);
int stage_gemm_tn_stream(const float* dY, const void* gate, int gate_kind, const float* X, float* part, float* part_b,
                         long long M, int N, int K, int* S_io, long* rows_per_split_io, void* stream);
static bool gemm_exact_f32() {
    static int mode = -1;
    if (mode < 0) mode = getenv("STAGE_GEMM_F32") ? 1 : 0;
    return mode == 1;
}

extern "C" int stage_gemm_nt(const float* X, const float* gate, const float* W, const float* bias,
                             const float* residual, float* Y, long long M, int N, int K, int relu, void* stream) {
    if (M <= 0 || N <= 0) return 0;
    if (K <= 0) return STAGE_ERR_SHAPE;
    if (!gemm_exact_f32()) {
        static const bool tiled_only = getenv("STAGE_GEMM_TILED") != nullptr;   // developer switch
        if (!tiled_only) {
            const int rc = stage_gemm_nt_stream(X, gate, gate ? 1 : 0, W, bias, residual, Y, nullptr, M, N, K, relu, stream);
            if (rc <= 0) return rc;                      // 1 = shape not handled by the streaming kernel
        }
        return stage_gemm_nt_bf16x3(X, gate, W, bias, residual, Y, M, N, K, relu, stream);
    }
    if (K <= 0) return STAGE_ERR_SHAPE;
    const int vecX = (K % 4 == 0) && (((uintptr_t)X & 15) == 0) && (!gate || ((uintptr_t)gate & 15) == 0);
    const int vecW = (K % 4 == 0) && (((uintptr_t)W & 15) == 0);
    dim3 grid((unsigned)(((M + BM - 1) / BM) * ((N + BN - 1) / BN)));
    if (vecX && vecW && K >= 4)
        hipLaunchKernelGGL(gemm_nt_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, X, gate, W, bias, residual, Y,
                           (long)M, N, K, relu, vecX, vecW);
    else
        hipLaunchKernelGGL(gemm_nt_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, X, gate, W, bias, residual, Y,
                           (long)M, N, K, relu, vecX, vecW);
    STAGE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// TN:  partial[s][n][k] = sum_{m in slab s} Yg[m,n] X[m,k] ; partial_b[s][n] = sum_{m in slab s} Yg[m,n]
// ------------------------------------------------------------------------------------------------
#define TN_MAX_SPLIT 1024

template <bool FAST>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const float* __restrict__ dY, const float* __restrict__ G,
                                                         const float* __restrict__ X, float* __restrict__ part,
                                                         float* __restrict__ part_b, long M, int N, int K,
                                                         long rows_per_split, int vecY, int vecX) {
    __shared__ __attribute__((aligned(16))) float Ys[BK * BM];  // [m][n]
    __shared__ __attribute__((aligned(16))) float Xs[BK * BN];  // [m][k]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * BM;  // output rows = n
    const int k0 = blockIdx.y * BN;  // output cols = k
    const int split = blockIdx.z;
    const long mbeg = (long)split * rows_per_split;
    const long mend = min(M, mbeg + rows_per_split);
    const int lrow = tid >> 5, lcol = (tid & 31) * 4;  // staging: 8 rows x 32 float4 per pass, 4 passes

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    float4 bsum = f4zero();

    float4 py[4], px[4], pg[4];
    auto fetch = [&](long mb) {   // loads only (gate applied in stash): one burst, one wait
#pragma unroll
        for (int p = 0; p < 4; p++) {
            py[p] = load4_f<FAST>(dY, mb + lrow + 8 * p, N, n0 + lcol, mend, N, vecY);
            px[p] = load4_f<FAST>(X, mb + lrow + 8 * p, K, k0 + lcol, mend, K, vecX);
        }
        if (G) {
#pragma unroll
            for (int p = 0; p < 4; p++) pg[p] = load4_f<FAST>(G, mb + lrow + 8 * p, N, n0 + lcol, mend, N, vecY);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int p = 0; p < 4; p++) {
            if (G) py[p] = gate4(py[p], pg[p]);
            st4(&Ys[(lrow + 8 * p) * BM + lcol], py[p]);
            st4(&Xs[(lrow + 8 * p) * BN + lcol], px[p]);
            bsum = f4add(bsum, py[p]);
        }
    };

    if (mbeg < mend) fetch(mbeg);
    for (long mb = mbeg; mb < mend; mb += BK) {
        __syncthreads();
        stash();
        __syncthreads();
        if (mb + BK < mend) fetch(mb + BK);
        const float* ap = &Ys[(16 * h) * BM + wm * 64 + l31];
        const float* bp = &Xs[(16 * h) * BN + wn * 64 + l31];
#pragma unroll
        for (int ks = 0; ks < 16; ks++) {
            const float a0 = ap[ks * BM], a1 = ap[ks * BM + 32];
            const float b0 = bp[ks * BN], b1 = bp[ks * BN + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
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
    // bias-gradient partial: column sums of the staged dY tile (only the k-tile 0 blocks emit it)
    if (part_b && blockIdx.y == 0) {
        __syncthreads();
        float* red = Ys;  // reuse: [8][128]
        st4(&red[lrow * BM + lcol], bsum);
        __syncthreads();
        if (tid < BM) {
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < 8; g++) s += red[g * BM + tid];
            if (n0 + tid < N) part_b[(size_t)split * N + n0 + tid] = s;
        }
    }
}

static int tn_splits(long long M, int N, int K) {
    const long tiles = (long)((N + BM - 1) / BM) * ((K + BN - 1) / BN);
    // workgroups per launch: ~4 per CU; a single 128 x 128 tile runs best as ONE resident round (2 per CU): half the slab
    // partials to write and reduce, -6 % (M = 960000) ... -16 % (M = 240000)
    const long target = tiles == 1 ? 512 : 1024;
    long s = (target + tiles - 1) / tiles;
    const long max_by_rows = (M + 8 * BK - 1) / (8 * BK);
    if (s > max_by_rows) s = max_by_rows;
    if (s > TN_MAX_SPLIT) s = TN_MAX_SPLIT;
    if (s < 1) s = 1;
    return (int)s;
}

extern "C" size_t stage_gemm_tn_ws_bytes(long long M, int N, int K) {
    return (size_t)tn_splits(M, N, K) * ((size_t)N * K + N) * sizeof(float);
}

extern "C" int stage_gemm_tn(const float* dY, const float* gate, const float* X, float* dW, float* db, long long M,
                             int N, int K, void* ws, size_t ws_bytes, void* stream) {
    if (!gemm_exact_f32() && getenv("STAGE_GEMM_SPLIT_TN")) return stage_gemm_tn_bf16x3(dY, gate, X, dW, db, M, N, K, ws, ws_bytes, stream);
    hipStream_t st = (hipStream_t)stream;
    if (N <= 0 || K <= 0) return 0;
    if (M <= 0) {
        (void)hipMemsetAsync(dW, 0, sizeof(float) * (size_t)N * K, st);
        if (db) (void)hipMemsetAsync(db, 0, sizeof(float) * N, st);
        return 0;
    }
    if (ws_bytes < stage_gemm_tn_ws_bytes(M, N, K)) return STAGE_ERR_WORKSPACE;
    int S = tn_splits(M, N, K);
    long rps = (M + S - 1) / S;
    rps = (rps + BK - 1) / BK * BK;
    float* part = (float*)ws;
    float* part_b = part + (size_t)S * N * K;   // fixed by the workspace slab count, even if fewer slabs get used
    const int vecY = (N % 4 == 0) && (((uintptr_t)dY & 15) == 0) && (!gate || ((uintptr_t)gate & 15) == 0);
    const int vecX = (K % 4 == 0) && (((uintptr_t)X & 15) == 0);
    dim3 grid((N + BM - 1) / BM, (K + BN - 1) / BN, S);
    static const bool tn_tiled = getenv("STAGE_GEMM_TN_TILED") != nullptr;   // developer switch
    int handled = 1;
    if (!gemm_exact_f32() && !tn_tiled)
        handled = stage_gemm_tn_stream(dY, gate, gate ? 1 : 0, X, part, db ? part_b : (float*)nullptr, M, N, K, &S, &rps, stream);
    if (handled < 0 || handled > 1) return handled;
    if (handled == 0) {
    } else if (vecY && vecX && N >= 4 && K >= 4)
        hipLaunchKernelGGL(gemm_tn_kernel<true>, grid, dim3(256), 0, st, dY, gate, X, part, db ? part_b : (float*)nullptr,
                           (long)M, N, K, rps, vecY, vecX);
    else
        hipLaunchKernelGGL(gemm_tn_kernel<false>, grid, dim3(256), 0, st, dY, gate, X, part, db ? part_b : (float*)nullptr,
                           (long)M, N, K, rps, vecY, vecX);
    STAGE_LAUNCH_CHECK();
    const long C = (long)N * K;
    if (db) stage_colreduce2(part, dW, C, (int)C, part_b, db, (long)N, N, S, st);   // one launch for dW and db
    else stage_colreduce(part, dW, nullptr, S, C, (int)C, 1, 0, st);
    STAGE_LAUNCH_CHECK();
    return 0;
}


// ------------------------------------------------------------------------------------------------
// ReLU bit-mask variants (streaming kernels only).  The forward GEMM of a Linear+ReLU also emits mask[w][m] (uint32, word-major,
// w < ceil(N/32), bit b <=> Y[m][32w+b] > 0); both backward GEMMs take that mask in place of the fp32 gate tensor: 1/32 of
// the gate bytes, and the 15 MB mask of a (960000, 128) layer stays cache resident.
// ------------------------------------------------------------------------------------------------
extern "C" int stage_gemm_mask_supported(long long M, int N, int K) {
    if (gemm_exact_f32() || getenv("STAGE_GEMM_TILED") || getenv("STAGE_GEMM_TN_TILED") || getenv("STAGE_GEMM_NO_MASK")) return 0;
    // forward (M,K)->(M,N), dX (M,N)->(M,K), dW: all three must be taken by the streaming kernels
    const bool fwd = K % 4 == 0 && K >= 64 && M >= 4096 && M * (long long)K * 4 < (1ll << 31);
    const bool dx = N % 4 == 0 && N >= 64 && M * (long long)N * 4 < (1ll << 31);
    return fwd && dx ? 1 : 0;
}

// LayerNorm gain / bias gradients straight from the dX product of the Linear behind it (gemm_stream.hip: COLSUM): for a
// LayerNorm whose INPUT needs no gradient (the first layer of the input MLPs, model/stage.py:85-91, 98-104) the (M, N) gradient
// of its output is reduced over the rows inside the GEMM epilogue instead of being written and read back.
size_t stage_gemm_nt_lnparam_ws(long long M, int N);
int stage_gemm_nt_stream_lnparam(const float* dY, const unsigned* gate_mask, const float* Wt, const float* x, const float* mean,
                                 const float* rstd, const unsigned* keep_mask, float p_drop, float* dgamma, float* dbeta,
                                 long long M, int N, int K, void* ws, size_t ws_bytes, void* stream);
extern "C" int stage_gemm_nt_lnparam_supported(long long M, int N, int K) {
    if (gemm_exact_f32() || getenv("STAGE_GEMM_TILED") || getenv("STAGE_GEMM_NO_LNPARAM")) return 0;
    return (K % 4 == 0 && K >= 64 && N % 4 == 0 && M >= 4096 && M * (long long)K * 4 < (1ll << 31) &&
            M * (long long)N * 4 < (1ll << 31)) ? 1 : 0;
}
extern "C" size_t stage_gemm_nt_lnparam_ws_bytes(long long M, int N) { return stage_gemm_nt_lnparam_ws(M, N); }
extern "C" int stage_gemm_nt_lnparam(const float* dY, const unsigned* gate_mask, const float* Wt, const float* x, const float* mean,
                                     const float* rstd, const unsigned* keep_mask, float p_drop, float* dgamma, float* dbeta,
                                     long long M, int N, int K, void* ws, size_t ws_bytes, void* stream) {
    if (!stage_gemm_nt_lnparam_supported(M, N, K)) return STAGE_ERR_SHAPE;
    const int rc = stage_gemm_nt_stream_lnparam(dY, gate_mask, Wt, x, mean, rstd, keep_mask, p_drop, dgamma, dbeta, M, N, K, ws,
                                                ws_bytes, stream);
    return rc == 1 ? STAGE_ERR_SHAPE : rc;
}

extern "C" int stage_gemm_nt_mask(const float* X, const unsigned* gate_mask, const float* W, const float* bias, float* Y,
                                  unsigned* relu_mask_out, long long M, int N, int K, int relu, void* stream) {
    if (M <= 0 || N <= 0) return 0;
    if (K <= 0 || (relu_mask_out && !relu)) return STAGE_ERR_SHAPE;
    const int rc = stage_gemm_nt_stream(X, gate_mask, gate_mask ? 2 : 0, W, bias, nullptr, Y, relu_mask_out, M, N, K, relu,
                                        stream);
    return rc == 1 ? STAGE_ERR_SHAPE : rc;
}

extern "C" int stage_gemm_tn_mask(const float* dY, const unsigned* gate_mask, const float* X, float* dW, float* db,
                                  long long M, int N, int K, void* ws, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (N <= 0 || K <= 0) return 0;
    if (M <= 0) return STAGE_ERR_SHAPE;
    if (ws_bytes < stage_gemm_tn_ws_bytes(M, N, K)) return STAGE_ERR_WORKSPACE;
    int S = tn_splits(M, N, K);
    long rps = (M + S - 1) / S;
    rps = (rps + BK - 1) / BK * BK;
    float* part = (float*)ws;
    float* part_b = part + (size_t)S * N * K;
    const int rc = stage_gemm_tn_stream(dY, gate_mask, gate_mask ? 2 : 0, X, part, db ? part_b : (float*)nullptr, M, N, K, &S,
                                        &rps, stream);
    if (rc != 0) return rc == 1 ? STAGE_ERR_SHAPE : rc;
    const long C = (long)N * K;
    if (db) stage_colreduce2(part, dW, C, (int)C, part_b, db, (long)N, N, S, st);
    else stage_colreduce(part, dW, nullptr, S, C, (int)C, 1, 0, st);
    STAGE_LAUNCH_CHECK();
    return 0;
}
