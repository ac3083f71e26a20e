// K-group entry points: launch sequencing of STAGE's fused-op groups on the C side (last bullet:
// "one fwd and one bwd symbol per K-group").  Each group runs the same kernels, in the same order and with the same
// arguments, as the per-op path of tvqaplus_amd/ops.py -- but as ONE call from the host language: the Python thread sees
// ~25 calls per training step instead of ~360, so the step no longer depends on how fast the host interpreter can issue
// launches (the driver-timed step was host-bound).  fp32 storage; shapes a group does not take return
// STAGE_ERR_SHAPE before anything is launched (the caller then uses the per-op entry points).
//
// Memory protocol (all device memory belongs to the caller):
//   arena  what the forward keeps for the backward (normalised operands, GEMM outputs, ReLU bit masks, statistics), carved
//          deterministically from one caller buffer of stage_grp_*_arena_bytes(); the backward carves the same layout
//   tmp    backward-only scratch (gradients in flight, transposed weights, kernel workspaces), stage_grp_*_bwd_tmp_bytes()
//   flags  host ints written by the forward and handed back to the backward (which optional kernel paths were taken)
//   params / grads  host arrays of device pointers in the order each group documents; seeds: host array, one per dropout site
#include "common.h"
#include "../../include/stage_hip.h"

namespace {

struct Bump {
    char* base;
    size_t off;
    template <class T>
    T* take(size_t n) {
        off = (off + 255) & ~(size_t)255;
        T* p = base ? (T*)(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
};
inline size_t umax(size_t a, size_t b) { return a > b ? a : b; }
inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

#define TRY(call)                     \
    do {                              \
        const int rc__ = (call);      \
        if (rc__ != 0) return rc__;   \
    } while (0)

constexpr float EPS_LN = 1e-5f, EPS_L2 = 1e-12f;

__global__ void grp_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int N, int K) {
    // wt[k][n] = w[n][k]; weights are at most a few hundred KB: one element per thread, reads coalesced
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * K) return;
    const int n = (int)(i / K), k = (int)(i % K);
    wt[(size_t)k * N + n] = w[i];
}
int transpose(const float* w, float* wt, int N, int K, void* st) {
    const long total = (long)N * K;
    hipLaunchKernelGGL(grp_transpose_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)st, w, wt, N, K);
    STAGE_LAUNCH_CHECK();
    return 0;
}

// ---- Linear (+bias, +ReLU): tvqaplus_amd/ops.py::_Linear ----------------------------------------------------------
bool lin_wants_mask(const float* x, const float* W, long long M, int N, int K, int relu) {
    return relu && stage_gemm_mask_supported(M, N, K) && al16(x) && al16(W);
}
size_t lin_mask_words(long long M, int N) { return (size_t)((N + 31) / 32) * (size_t)M; }
// forward; *has_mask = 1 when the ReLU bit mask was written (mask may be NULL: never try)
int lin_fwd(const float* x, const float* W, const float* b, float* y, unsigned* mask, int* has_mask, long long M, int N, int K,
            int relu, void* st) {
    *has_mask = 0;
    if (mask && lin_wants_mask(x, W, M, N, K, relu)) {
        const int rc = stage_gemm_nt_mask(x, nullptr, W, b, y, mask, M, N, K, 1, st);
        if (rc == 0) { *has_mask = 1; return 0; }
        if (rc != STAGE_ERR_SHAPE) return rc;
    }
    return stage_gemm_nt(x, nullptr, W, b, nullptr, y, M, N, K, relu, st);
}
size_t lin_bwd_ws(long long M, int N, int K) { return umax(stage_gemm_tn_ws_bytes(M, N, K), 256); }
// backward: dx (M,K) [may be NULL], dW (N,K), db (N) [may be NULL]; wt = scratch for the (K,N) transposed weight
int lin_bwd(const float* dy, const float* x, const float* y, const unsigned* mask, int has_mask, int relu, const float* W, float* wt,
            float* dx, float* dW, float* db, long long M, int N, int K, void* ws, size_t wsb, void* st) {
    const float* gate = relu ? y : nullptr;
    const bool use_mask = has_mask && al16(dy);
    if (dx) {
        TRY(transpose(W, wt, N, K, st));
        bool done = false;
        if (use_mask) {
            const int rc = stage_gemm_nt_mask(dy, mask, wt, nullptr, dx, nullptr, M, K, N, 0, st);
            if (rc == 0) done = true;
            else if (rc != STAGE_ERR_SHAPE) return rc;
        }
        if (!done) TRY(stage_gemm_nt(dy, gate, wt, nullptr, nullptr, dx, M, K, N, 0, st));
    }
    // every kernel gets exactly the workspace size its own query names (some size their partial-sum grids by it: the per-op path
    // and this one then reduce in the same order and agree bit for bit)
    const size_t need = stage_gemm_tn_ws_bytes(M, N, K);
    if (wsb < need) return STAGE_ERR_WORKSPACE;
    if (use_mask) {
        const int rc = stage_gemm_tn_mask(dy, mask, x, dW, db, M, N, K, ws, need, st);
        if (rc == 0) return 0;
        if (rc != STAGE_ERR_SHAPE) return rc;
    }
    return stage_gemm_tn(dy, gate, x, dW, db, M, N, K, ws, need, st);
}

bool ln_dwconv_ok(int D, int k) {
    const int d4 = D / 4;
    return D % 4 == 0 && d4 >= 4 && d4 <= 64 && (d4 & (d4 - 1)) == 0 && k >= 1 && k <= 9 && (k % 2) == 1;
}
bool cat3_reduced_ok(int D, int rep, int inner) {
    const int d4 = D / 4;
    return rep > 1 && D % 4 == 0 && d4 >= 4 && d4 <= 64 && (d4 & (d4 - 1)) == 0 && inner <= 64;
}

// =====================================================================================================================
// G1  input MLP (model/stage.py:350-362 base_encoder up to the encoder; :85-91 / :98-104 bridge, :115-120 input_embedding;
//     l2 != 0: the F.normalize of :256 first):  [l2norm] -> LN(K0)+drop -> Linear(K0->H)+ReLU -> LN(H)+drop -> Linear(H->D)+ReLU -> LN(D)
//     params: g0 b0 W1 c1 g1 b1 W2 c2 g2 b2 ; seeds: [LN0 dropout, LN1 dropout] ; flags: [mask1, mask2]
// =====================================================================================================================
struct MlpArena {
    float *xn, *y0, *mean0, *rstd0, *h1, *y1, *mean1, *rstd1, *h2, *mean2, *rstd2;
    unsigned *mask1, *mask2;
    size_t bytes;
};
MlpArena mlp_layout(void* base, long long M, int K0, int H, int D, int l2) {
    Bump b{(char*)base, 0};
    MlpArena a;
    a.xn = l2 ? b.take<float>((size_t)M * K0) : nullptr;
    a.y0 = b.take<float>((size_t)M * K0);
    a.mean0 = b.take<float>((size_t)M);
    a.rstd0 = b.take<float>((size_t)M);
    a.h1 = b.take<float>((size_t)M * H);
    a.mask1 = b.take<unsigned>(lin_mask_words(M, H));
    a.y1 = b.take<float>((size_t)M * H);
    a.mean1 = b.take<float>((size_t)M);
    a.rstd1 = b.take<float>((size_t)M);
    a.h2 = b.take<float>((size_t)M * D);
    a.mask2 = b.take<unsigned>(lin_mask_words(M, D));
    a.mean2 = b.take<float>((size_t)M);
    a.rstd2 = b.take<float>((size_t)M);
    a.bytes = b.off;
    return a;
}

}  // namespace

extern "C" size_t stage_grp_input_mlp_arena_bytes(long long M, int K0, int H, int D, int l2) {
    return mlp_layout(nullptr, M, K0, H, D, l2).bytes;
}

namespace {
// gather != NULL (ragged context rows): the M rows are rows gather[0..M) of the padded feature tensor x
int input_mlp_fwd(const float* x, const int* gather, const float* const* P, float* out, void* arena, size_t arena_bytes, int* flags,
                  long long M, int K0, int H, int D, int l2, float p, const unsigned long long* seeds, void* st);
}
extern "C" int stage_grp_input_mlp_fwd(const float* x, const float* const* P, float* out, void* arena, size_t arena_bytes,
                                       int* flags, long long M, int K0, int H, int D, int l2, float p,
                                       const unsigned long long* seeds, void* st) {
    return input_mlp_fwd(x, nullptr, P, out, arena, arena_bytes, flags, M, K0, H, D, l2, p, seeds, st);
}
// ragged context rows (csrc/ragged.hip): x is the padded (frames * L, K0) feature tensor, src_rows (M) its live rows in order
extern "C" int stage_grp_input_mlp_rag_fwd(const float* x, const int* src_rows, const float* const* P, float* out, void* arena,
                                           size_t arena_bytes, int* flags, long long M, int K0, int H, int D, int l2, float p,
                                           const unsigned long long* seeds, void* st) {
    if (!src_rows) return STAGE_ERR_SHAPE;
    return input_mlp_fwd(x, src_rows, P, out, arena, arena_bytes, flags, M, K0, H, D, l2, p, seeds, st);
}
namespace {
int input_mlp_fwd(const float* x, const int* gather, const float* const* P, float* out, void* arena, size_t arena_bytes, int* flags,
                  long long M, int K0, int H, int D, int l2, float p, const unsigned long long* seeds, void* st) {
    if (M <= 0 || K0 % 4 || H % 4 || D % 4 || K0 > 1024 || H > 1024 || D > 1024) return STAGE_ERR_SHAPE;
    MlpArena a = mlp_layout(arena, M, K0, H, D, l2);
    if (arena_bytes < a.bytes) return STAGE_ERR_WORKSPACE;
    const float* xin = x;
    if (l2) {
        if (gather) TRY(stage_l2norm_gather_fwd(x, gather, a.xn, M, K0, EPS_L2, st));
        else TRY(stage_l2norm_fwd(x, a.xn, nullptr, M, K0, EPS_L2, 0.f, 0ull, st));
        xin = a.xn;
    }
    if (gather && !l2) TRY(stage_layernorm_gather_fwd(x, gather, P[0], P[1], a.y0, a.mean0, a.rstd0, M, K0, EPS_LN, p, seeds[0], st));
    else
    TRY(stage_layernorm_fwd(xin, nullptr, 0, nullptr, P[0], P[1], a.y0, a.mean0, a.rstd0, M, K0, EPS_LN, p, seeds[0], st));
    TRY(lin_fwd(a.y0, P[2], P[3], a.h1, a.mask1, &flags[0], M, H, K0, 1, st));
    TRY(stage_layernorm_fwd(a.h1, nullptr, 0, nullptr, P[4], P[5], a.y1, a.mean1, a.rstd1, M, H, EPS_LN, p, seeds[1], st));
    TRY(lin_fwd(a.y1, P[6], P[7], a.h2, a.mask2, &flags[1], M, D, H, 1, st));
    TRY(stage_layernorm_fwd(a.h2, nullptr, 0, nullptr, P[8], P[9], out, a.mean2, a.rstd2, M, D, EPS_LN, 0.f, 0ull, st));
    return 0;
}
}  // namespace

namespace {
struct MlpTmp { float *dh2, *dy1, *dh1, *dy0, *wt; void* ws; size_t wsb, bytes; };
MlpTmp mlp_tmp(void* base, long long M, int K0, int H, int D) {
    Bump b{(char*)base, 0};
    MlpTmp t;
    t.dh2 = b.take<float>((size_t)M * D);
    t.dy1 = b.take<float>((size_t)M * H);
    t.dh1 = b.take<float>((size_t)M * H);
    t.dy0 = b.take<float>((size_t)M * K0);
    t.wt = b.take<float>((size_t)umax((size_t)H * K0, (size_t)D * H));
    t.wsb = umax(umax(stage_ln_bwd_ws_bytes(K0), stage_ln_bwd_ws_bytes(H)), stage_ln_bwd_ws_bytes(D));
    t.wsb = umax(t.wsb, umax(lin_bwd_ws(M, H, K0), lin_bwd_ws(M, D, H)));
    t.ws = b.take<char>(t.wsb);
    t.bytes = b.off;
    return t;
}
}  // namespace

extern "C" size_t stage_grp_input_mlp_bwd_tmp_bytes(long long M, int K0, int H, int D) { return mlp_tmp(nullptr, M, K0, H, D).bytes; }

// grads: same order as params (dg0 db0 dW1 dc1 dg1 db1 dW2 dc2 dg2 db2).  The features need no gradient (they are data).
namespace {
int input_mlp_bwd(const float* dout, const float* x, const int* gather, const float* const* P, float* const* G, const void* arena,
                  size_t arena_bytes, const int* flags, void* tmp, size_t tmp_bytes, long long M, int K0, int H, int D, int l2, float p,
                  const unsigned long long* seeds, void* st);
}
extern "C" int stage_grp_input_mlp_bwd(const float* dout, const float* x, const float* const* P, float* const* G, const void* arena,
                                       size_t arena_bytes, const int* flags, void* tmp, size_t tmp_bytes, long long M, int K0,
                                       int H, int D, int l2, float p, const unsigned long long* seeds, void* st) {
    return input_mlp_bwd(dout, x, nullptr, P, G, arena, arena_bytes, flags, tmp, tmp_bytes, M, K0, H, D, l2, p, seeds, st);
}
extern "C" int stage_grp_input_mlp_rag_bwd(const float* dout, const float* x, const int* src_rows, const float* const* P,
                                           float* const* G, const void* arena, size_t arena_bytes, const int* flags, void* tmp,
                                           size_t tmp_bytes, long long M, int K0, int H, int D, int l2, float p,
                                           const unsigned long long* seeds, void* st) {
    if (!src_rows) return STAGE_ERR_SHAPE;
    return input_mlp_bwd(dout, x, src_rows, P, G, arena, arena_bytes, flags, tmp, tmp_bytes, M, K0, H, D, l2, p, seeds, st);
}
namespace {
int input_mlp_bwd(const float* dout, const float* x, const int* gather, const float* const* P, float* const* G, const void* arena,
                  size_t arena_bytes, const int* flags, void* tmp, size_t tmp_bytes, long long M, int K0, int H, int D, int l2, float p,
                  const unsigned long long* seeds, void* st) {
    MlpArena a = mlp_layout((void*)arena, M, K0, H, D, l2);
    MlpTmp t = mlp_tmp(tmp, M, K0, H, D);
    if (arena_bytes < a.bytes || tmp_bytes < t.bytes) return STAGE_ERR_WORKSPACE;
    const float* xin = l2 ? a.xn : x;
    TRY(stage_layernorm_bwd(dout, a.h2, a.mean2, a.rstd2, P[8], t.dh2, nullptr, G[8], G[9], M, D, 0.f, 0ull, t.ws, stage_ln_bwd_ws_bytes(D), st));
    TRY(lin_bwd(t.dh2, a.y1, a.h2, a.mask2, flags[1], 1, P[6], t.wt, t.dy1, G[6], G[7], M, D, H, t.ws, t.wsb, st));
    TRY(stage_layernorm_bwd(t.dy1, a.h1, a.mean1, a.rstd1, P[4], t.dh1, nullptr, G[4], G[5], M, H, p, seeds[1], t.ws, stage_ln_bwd_ws_bytes(H), st));
    TRY(lin_bwd(t.dh1, a.y0, a.h1, a.mask1, flags[0], 1, P[2], t.wt, t.dy0, G[2], G[3], M, H, K0, t.ws, t.wsb, st));
    if (gather && !l2)
        return stage_layernorm_gather_bwd(t.dy0, x, gather, a.mean0, a.rstd0, P[0], G[0], G[1], M, K0, p, seeds[0], t.ws, stage_ln_bwd_ws_bytes(K0), st);
    TRY(stage_layernorm_bwd(t.dy0, xin, a.mean0, a.rstd0, P[0], nullptr, nullptr, G[0], G[1], M, K0, p, seeds[0], t.ws, stage_ln_bwd_ws_bytes(K0), st));
    return 0;
}
}  // namespace

// =====================================================================================================================
// G2  encoder block without self-attention (model/encoder.py:29-52, model/cnn.py:37-47, model/position_encoding.py:38-43):
//     x + pe -> n_conv x [LN (+dropout on even i) -> depthwise conv -> 1x1 conv + ReLU -> + residual] -> final LN
//     pooled != 0: the caller only needs the masked max over L of the output (model/stage.py:503) -> out (M, D)
//     params: per conv i: ln_g ln_b dw_w dw_b pw_w pw_b ; then final_g final_b.  seeds: one per even conv index.
//     flags: [pw mask of conv i] for i < n_conv ; flags[n_conv] = 1 when LayerNorm + max ran fused
//     Residual adds are deferred into the next LayerNorm's prologue (exported sums), as in STAGE._encoder_block.
// =====================================================================================================================
namespace {
constexpr int ENC_MAX_CONV = 8;
struct EncArena {
    float *h[ENC_MAX_CONV], *s[ENC_MAX_CONV], *mean[ENC_MAX_CONV], *rstd[ENC_MAX_CONV], *g[ENC_MAX_CONV];
    unsigned* mask[ENC_MAX_CONV];
    float *sf, *meanf, *rstdf, *yf;
    int* idx;
    size_t bytes;
};
EncArena enc_layout(void* base, long long M, int L, int D, int n_conv, int pooled) {
    Bump b{(char*)base, 0};
    EncArena a;
    const size_t R = (size_t)M * L;
    for (int i = 0; i < n_conv; i++) {
        a.h[i] = b.take<float>(R * D);
        a.s[i] = b.take<float>(R * D);
        a.mean[i] = b.take<float>(R);
        a.rstd[i] = b.take<float>(R);
        a.g[i] = b.take<float>(R * D);
        a.mask[i] = b.take<unsigned>(lin_mask_words((long long)R, D));
    }
    a.sf = b.take<float>(R * D);
    a.meanf = b.take<float>(R);
    a.rstdf = b.take<float>(R);
    a.yf = pooled ? b.take<float>(R * D) : nullptr;          // only used when LayerNorm + max do not run fused
    a.idx = pooled ? b.take<int>((size_t)M * D) : nullptr;
    a.bytes = b.off;
    return a;
}
}  // namespace

extern "C" size_t stage_grp_encoder_arena_bytes(long long M, int L, int D, int n_conv, int pooled) {
    if (n_conv > ENC_MAX_CONV) return 0;
    return enc_layout(nullptr, M, L, D, n_conv, pooled).bytes;
}

extern "C" int stage_grp_encoder_fwd(const float* x, const float* pe, const float* pool_mask, const float* const* P, float* out,
                                     void* arena, size_t arena_bytes, int* flags, long long M, int L, int D, int n_conv, int k,
                                     float p, const unsigned long long* seeds, void* st) {
    if (M <= 0 || L <= 0 || n_conv < 0 || n_conv > ENC_MAX_CONV || !ln_dwconv_ok(D, k) || D > 1024) return STAGE_ERR_SHAPE;
    const int pooled = pool_mask != nullptr;
    EncArena a = enc_layout(arena, M, L, D, n_conv, pooled);
    if (arena_bytes < a.bytes) return STAGE_ERR_WORKSPACE;
    const long long R = M * L;
    const float* pending = x;
    const float* cur = pe;
    int period = L;
    for (int i = 0; i < n_conv; i++) {
        const float* const* Q = P + 6 * i;
        const bool drop = (i % 2) == 0;
        const unsigned long long seed = drop ? seeds[i / 2] : 0ull;
        TRY(stage_ln_dwconv_fwd(pending, cur, period, a.s[i], Q[0], Q[1], Q[2], Q[3], a.h[i], a.mean[i], a.rstd[i], M, L, D, k, EPS_LN,
                                drop ? p : 0.f, seed, st));
        cur = a.s[i];
        period = 0;
        TRY(lin_fwd(a.h[i], Q[4], Q[5], a.g[i], a.mask[i], &flags[i], R, D, D, 1, st));
        pending = a.g[i];
    }
    const float* const* F = P + 6 * n_conv;
    flags[n_conv] = 0;
    if (pooled && period == 0 && stage_ln_masked_max_supported(L, D)) {
        flags[n_conv] = 1;
        return stage_ln_masked_max_fwd(pending, cur, a.sf, F[0], F[1], pool_mask, out, a.idx, a.meanf, a.rstdf, M, L, D, EPS_LN, st);
    }
    float* y = pooled ? a.yf : out;
    TRY(stage_layernorm_fwd(pending, cur, period, a.sf, F[0], F[1], y, a.meanf, a.rstdf, R, D, EPS_LN, 0.f, 0ull, st));
    if (pooled) TRY(stage_masked_max_fwd(y, pool_mask, nullptr, out, a.idx, M, L, D, st));
    return 0;
}

namespace {
struct EncTmp { float *Ga, *Gb, *dh, *dyf, *wt; void* ws; size_t wsb, bytes; };
EncTmp enc_tmp(void* base, long long M, int L, int D, int k, int pooled) {
    Bump b{(char*)base, 0};
    EncTmp t;
    const size_t R = (size_t)M * L;
    t.Ga = b.take<float>(R * D);
    t.Gb = b.take<float>(R * D);
    t.dh = b.take<float>(R * D);
    t.dyf = pooled ? b.take<float>(R * D) : nullptr;
    t.wt = b.take<float>((size_t)D * D);
    t.wsb = umax(umax(stage_ln_bwd_ws_bytes(D), stage_ln_dwconv_bwd_ws_bytes(D, k)), lin_bwd_ws((long long)R, D, D));
    t.ws = b.take<char>(t.wsb);
    t.bytes = b.off;
    return t;
}
}  // namespace

extern "C" size_t stage_grp_encoder_bwd_tmp_bytes(long long M, int L, int D, int k, int pooled) {
    return enc_tmp(nullptr, M, L, D, k, pooled).bytes;
}

// dout: (M, L, D), or (M, D) when pooled.  dx (M, L, D) receives the gradient of x (may be NULL).  grads in params order.
extern "C" int stage_grp_encoder_bwd(const float* dout, const float* x, const float* pool_mask, const float* const* P,
                                     float* const* Gr, float* dx, const void* arena, size_t arena_bytes, const int* flags,
                                     void* tmp, size_t tmp_bytes, long long M, int L, int D, int n_conv, int k, float p,
                                     const unsigned long long* seeds, void* st) {
    (void)x;
    const int pooled = pool_mask != nullptr;
    if (n_conv < 0 || n_conv > ENC_MAX_CONV) return STAGE_ERR_SHAPE;
    EncArena a = enc_layout((void*)arena, M, L, D, n_conv, pooled);
    EncTmp t = enc_tmp(tmp, M, L, D, k, pooled);
    if (arena_bytes < a.bytes || tmp_bytes < t.bytes) return STAGE_ERR_WORKSPACE;
    const long long R = M * L;
    const float* const* F = P + 6 * n_conv;
    float* const* GF = Gr + 6 * n_conv;
    // G = gradient of the last exported sum (pending + cur): it feeds the last 1x1 conv's output AND the previous sum
    float* G = (n_conv == 0 && dx) ? dx : t.Ga;
    if (pooled && flags[n_conv]) {
        TRY(stage_ln_masked_max_bwd(dout, a.idx, pool_mask, a.sf, a.meanf, a.rstdf, F[0], G, GF[0], GF[1], M, L, D, t.ws, stage_ln_bwd_ws_bytes(D), st));
    } else {
        const float* dy = dout;
        if (pooled) {
            TRY(stage_masked_max_bwd(dout, a.idx, pool_mask, t.dyf, M, L, D, 0, st));
            dy = t.dyf;
        }
        TRY(stage_layernorm_bwd(dy, a.sf, a.meanf, a.rstdf, F[0], G, nullptr, GF[0], GF[1], R, D, 0.f, 0ull, t.ws, stage_ln_bwd_ws_bytes(D), st));
    }
    for (int i = n_conv - 1; i >= 0; i--) {
        const float* const* Q = P + 6 * i;
        float* const* GQ = Gr + 6 * i;
        const bool drop = (i % 2) == 0;
        const unsigned long long seed = drop ? seeds[i / 2] : 0ull;
        // 1x1 conv + ReLU: G is the gradient of its output; its input was the depthwise conv's output h[i]
        TRY(lin_bwd(G, a.h[i], a.g[i], a.mask[i], flags[i], 1, Q[4], t.wt, t.dh, GQ[4], GQ[5], R, D, D, t.ws, t.wsb, st));
        // LayerNorm -> depthwise conv: gradient of the sum s[i] = LN-path gradient + what reached the sum directly (G)
        float* Gn = (i == 0 && dx) ? dx : (G == t.Ga ? t.Gb : t.Ga);
        TRY(stage_ln_dwconv_bwd(t.dh, a.s[i], a.mean[i], a.rstd[i], Q[0], Q[1], Q[2], Gn, G, GQ[0], GQ[1], GQ[2], GQ[3], M, L, D, k,
                                drop ? p : 0.f, seed, t.ws, stage_ln_dwconv_bwd_ws_bytes(D, k), st));
        G = Gn;
    }
    return 0;
}

// =====================================================================================================================
// G3  QA <-> context attention + down-projection (model/stage.py:365-387, model/context_query_attention.py:35-101):
//     Cn = drop(normalize(qa)) ; (A, S_raw, S_norm) = StructuredAttention(Cn, ctx) ; z = drop(LN_3D([qa, A, qa*A])) ;
//     mixed = ReLU(Linear_3D->D(z)).  Fast attention kernels only (Lr <= 64; longer rows: STAGE_ERR_SHAPE -> per-op path).
//     params: ln_g ln_b W c ; seeds: [context-side dropout, region-side dropout, LayerNorm dropout] ; flags: [mask]
// =====================================================================================================================
namespace {
struct QaArena { float *Cn, *A, *z, *mean, *rstd; unsigned* mask; void* fwd_ws; size_t bytes; };
QaArena qa_layout(void* base, int N, int NA, int Li, int Lqa, int D) {
    Bump b{(char*)base, 0};
    QaArena a;
    const size_t U = (size_t)N * NA * Li * Lqa;
    a.Cn = b.take<float>((size_t)N * NA * Lqa * D);
    a.A = b.take<float>(U * D);
    a.z = b.take<float>(U * 3 * D);
    a.mean = b.take<float>(U);
    a.rstd = b.take<float>(U);
    a.mask = b.take<unsigned>(lin_mask_words((long long)U, D));
    a.fwd_ws = b.take<char>(stage_cat3_ln_gemm_fwd_ws_bytes());     // pre-split weight image of the fused forward (scratch)
    a.bytes = b.off;
    return a;
}
struct QaTmp { float *dz, *dA, *Qn, *dQn, *dCn, *dS, *da_full, *wt; void* ws; size_t wsb, bytes; };
QaTmp qa_tmp(void* base, int N, int NA, int Li, int Lqa, int Lr, int D) {
    Bump b{(char*)base, 0};
    QaTmp t;
    const size_t U = (size_t)N * NA * Li * Lqa, Qe = (size_t)N * Li * Lr * D;
    t.dz = b.take<float>(U * 3 * D);
    t.dA = b.take<float>(U * D);
    t.Qn = b.take<float>(Qe);
    t.dQn = b.take<float>(Qe);
    t.dCn = b.take<float>((size_t)N * NA * Lqa * D);
    t.wt = b.take<float>((size_t)3 * D * D);
    // the kernels that only run for shapes the fused ones reject share one region (they never run together with dz's consumers)
    t.dS = nullptr;
    t.da_full = nullptr;
    t.wsb = umax(lin_bwd_ws((long long)U, D, 3 * D), stage_ln_bwd_ws_bytes(3 * D));
    t.wsb = umax(t.wsb, stage_cat3_layernorm_bwd_reduced_ws_bytes((long long)U, D, Li, Lqa));
    t.wsb = umax(t.wsb, umax(stage_str_attn_bwd_fused_ws_bytes(N, NA, Li, Lqa, D), stage_str_attn_bwd_ws_bytes(N, NA, Lqa, D)));
    t.wsb = umax(t.wsb, stage_cat3_dx_ln_bwd_ws_bytes((long long)U, D, Li, Lqa));
    t.wsb = umax(t.wsb, stage_cat3_bwd_dw_ws_bytes((long long)U, D, Li, Lqa));
    t.ws = b.take<char>(t.wsb);
    t.bytes = b.off;
    return t;
}
}  // namespace

extern "C" size_t stage_grp_qa_ctx_arena_bytes(int N, int NA, int Li, int Lqa, int D) { return qa_layout(nullptr, N, NA, Li, Lqa, D).bytes; }

extern "C" int stage_grp_qa_ctx_fwd(const float* qa, const float* ctx, const float* qa_mask, const float* ctx_mask,
                                    const float* const* P, float* mixed, float* S_raw, float* S_norm, void* arena,
                                    size_t arena_bytes, int* flags, int N, int NA, int Li, int Lqa, int Lr, int D, float scale,
                                    float p, const unsigned long long* seeds, void* st) {
    if (N <= 0 || Lr > 64 || D % 16 || D > 256 || NA * Lqa > 256) return STAGE_ERR_SHAPE;
    QaArena a = qa_layout(arena, N, NA, Li, Lqa, D);
    if (arena_bytes < a.bytes) return STAGE_ERR_WORKSPACE;
    const long long U = (long long)N * NA * Li * Lqa;
    TRY(stage_l2norm_fwd(qa, a.Cn, nullptr, (long long)N * NA * Lqa, D, EPS_L2, p, seeds[0], st));
    TRY(stage_str_attn_fwd(a.Cn, ctx, qa_mask, ctx_mask, a.A, S_raw, S_norm, N, NA, Li, Lqa, Lr, D, scale, p, seeds[1], st));
    if (stage_cat3_ln_gemm_fwd_supported(U, D, Li, Lqa) && lin_wants_mask(a.z, P[2], U, D, 3 * D, 1)) {
        // LayerNorm + dropout + Linear + ReLU in one pass (csrc/cat3_fused.hip); fwd_ws: scratch for its pre-split weight image.
        // flags[0] = 2: z was NOT written -- the backward rebuilds it and forms the Linear's gradients itself (csrc/cat3_bwd_dw.hip)
        const bool dw = stage_cat3_bwd_dw_supported(U, D, Li, Lqa) != 0;
        const int rc = stage_cat3_ln_gemm_fwd(qa, a.A, P[0], P[1], P[2], P[3], dw ? nullptr : a.z, a.mean, a.rstd, mixed, a.mask, U, D, Li, Lqa,
                                              EPS_LN, p, seeds[2], a.fwd_ws, stage_cat3_ln_gemm_fwd_ws_bytes(), st);
        if (rc == 0) { flags[0] = dw ? 2 : 1; return 0; }
        if (rc != STAGE_ERR_SHAPE) return rc;
    }
    TRY(stage_cat3_layernorm_fwd(qa, a.A, P[0], P[1], a.z, a.mean, a.rstd, U, D, Li, Lqa, EPS_LN, p, seeds[2], st));
    return lin_fwd(a.z, P[2], P[3], mixed, a.mask, &flags[0], U, D, 3 * D, 1, st);
}

extern "C" size_t stage_grp_qa_ctx_bwd_tmp_bytes(int N, int NA, int Li, int Lqa, int Lr, int D) {
    size_t b = qa_tmp(nullptr, N, NA, Li, Lqa, Lr, D).bytes;
    // fallback kernels (shapes the fused ones reject): unreduced [a]-gradient rows / the materialised score gradient
    const size_t U = (size_t)N * NA * Li * Lqa;
    return b + 512 + umax(U * D, U * Lr) * sizeof(float);
}

// d_mixed (U, D); dS_ext: gradient on S_raw (supervised attention loss) or NULL.  Outputs: d_qa (N,NA,Lqa,D) = both uses of the QA
// embedding (attention context + the [a, b, a*b] operand), d_ctx (N,Li,Lr,D).  grads: dln_g dln_b dW dc.
extern "C" int stage_grp_qa_ctx_bwd(const float* d_mixed, const float* dS_ext, const float* qa, const float* ctx,
                                    const float* ctx_mask, const float* mixed, const float* S_norm, const float* const* P,
                                    float* const* G, float* d_qa, float* d_ctx, const void* arena, size_t arena_bytes,
                                    const int* flags, void* tmp, size_t tmp_bytes, int N, int NA, int Li, int Lqa, int Lr, int D,
                                    float scale, float p, const unsigned long long* seeds, void* st) {
    QaArena a = qa_layout((void*)arena, N, NA, Li, Lqa, D);
    QaTmp t = qa_tmp(tmp, N, NA, Li, Lqa, Lr, D);
    if (arena_bytes < a.bytes || tmp_bytes < stage_grp_qa_ctx_bwd_tmp_bytes(N, NA, Li, Lqa, Lr, D)) return STAGE_ERR_WORKSPACE;
    float* extra = (float*)((char*)tmp + ((t.bytes + 255) & ~(size_t)255));
    const long long U = (long long)N * NA * Li * Lqa, Crows = (long long)N * NA * Lqa, Qrows = (long long)N * Li * Lr;
    // Linear(3D -> D) + ReLU, LayerNorm over [a, b, a*b].  With the ReLU bit mask at hand the Linear's input gradient never
    // exists as a tensor (csrc/cat3_fused.hip); otherwise: dX GEMM, then the LayerNorm backward with the broadcast reduction
    bool reduced = false, fused = false;
    if (flags[0] == 2) {                                  // no saved z: everything in one kernel
        TRY(stage_cat3_bwd_dw(d_mixed, a.mask, P[2], qa, a.A, a.mean, a.rstd, P[0], P[1], d_qa, t.dA, G[0], G[1], G[2], G[3], U, D, Li, Lqa, p,
                              seeds[2], t.ws, stage_cat3_bwd_dw_ws_bytes(U, D, Li, Lqa), st));
        fused = reduced = true;
    } else if (flags[0] && al16(d_mixed) && stage_cat3_dx_ln_bwd_supported(U, D, Li, Lqa)) {
        TRY(lin_bwd(d_mixed, a.z, mixed, a.mask, flags[0], 1, P[2], t.wt, nullptr, G[2], G[3], U, D, 3 * D, t.ws, t.wsb, st));
        const int rc = stage_cat3_dx_ln_bwd(d_mixed, a.mask, P[2], qa, a.A, a.mean, a.rstd, P[0], d_qa, t.dA, G[0], G[1], U, D, Li, Lqa, p,
                                            seeds[2], t.ws, stage_cat3_dx_ln_bwd_ws_bytes(U, D, Li, Lqa), st);
        if (rc == 0) fused = reduced = true;
        else if (rc != STAGE_ERR_SHAPE) return rc;
    }
    if (!fused) TRY(lin_bwd(d_mixed, a.z, mixed, a.mask, flags[0], 1, P[2], t.wt, t.dz, G[2], G[3], U, D, 3 * D, t.ws, t.wsb, st));
    if (!fused && cat3_reduced_ok(D, Li, Lqa)) {
        const int rc = stage_cat3_layernorm_bwd_reduced(t.dz, qa, a.A, a.mean, a.rstd, P[0], d_qa, t.dA, G[0], G[1], U, D, Li, Lqa, p,
                                                        seeds[2], t.ws, stage_cat3_layernorm_bwd_reduced_ws_bytes(U, D, Li, Lqa), st);
        if (rc == 0) reduced = true;
        else if (rc != STAGE_ERR_SHAPE) return rc;
    }
    if (!reduced) {
        TRY(stage_cat3_layernorm_bwd(t.dz, qa, a.A, a.mean, a.rstd, P[0], extra, t.dA, G[0], G[1], U, D, Li, Lqa, p, seeds[2], t.ws,
                                     stage_ln_bwd_ws_bytes(3 * D), st));
        if (Li > 1) TRY(stage_reduce_rep(extra, d_qa, (long long)N * NA, Li, (long long)Lqa * D, st));
        else TRY((int)hipMemcpyAsync(d_qa, extra, (size_t)Crows * D * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)st));
    }
    // StructuredAttention backward
    TRY(stage_l2norm_fwd(ctx, t.Qn, nullptr, Qrows, D, EPS_L2, p, seeds[1], st));
    int rc = stage_str_attn_bwd_fused(t.dA, dS_ext, a.Cn, ctx, t.Qn, S_norm, ctx_mask, d_ctx, t.dQn, t.dCn, N, NA, Li, Lqa, Lr, D,
                                      scale, t.ws, stage_str_attn_bwd_fused_ws_bytes(N, NA, Li, Lqa, D), st);
    if (rc == STAGE_ERR_SHAPE)
        rc = stage_str_attn_bwd(t.dA, dS_ext, a.Cn, ctx, t.Qn, S_norm, extra, d_ctx, t.dQn, t.dCn, N, NA, Li, Lqa, Lr, D, scale, t.ws,
                                stage_str_attn_bwd_ws_bytes(N, NA, Lqa, D), st);
    if (rc != 0) return rc;
    // the two normalisations: gradients accumulate onto what is already in d_qa (the [a,b,a*b] path) / d_ctx (the value path)
    TRY(stage_l2norm_bwd(t.dCn, qa, d_qa, Crows, D, EPS_L2, p, seeds[0], 1, st));
    return stage_l2norm_bwd(t.dQn, ctx, d_ctx, Qrows, D, EPS_L2, p, seeds[1], 1, st);
}

// =====================================================================================================================
// G4  two-stream fusion concat_fc (model/stage.py:276-279, :106-113): LN_3D([s, v, s*v]) + drop -> Linear(3D->D) + ReLU -> LN(D)
//     params: ln3_g ln3_b W c ln_g ln_b ; seeds: [LayerNorm dropout] ; flags: [mask]
// =====================================================================================================================
namespace {
struct FcArena { float *z, *mean3, *rstd3, *h, *mean, *rstd; unsigned* mask; void* fwd_ws; size_t bytes; };
FcArena fc_layout(void* base, long long U, int D) {
    Bump b{(char*)base, 0};
    FcArena a;
    a.z = b.take<float>((size_t)U * 3 * D);
    a.mean3 = b.take<float>((size_t)U);
    a.rstd3 = b.take<float>((size_t)U);
    a.h = b.take<float>((size_t)U * D);
    a.mask = b.take<unsigned>(lin_mask_words(U, D));
    a.mean = b.take<float>((size_t)U);
    a.rstd = b.take<float>((size_t)U);
    a.fwd_ws = b.take<char>(stage_cat3_ln_gemm_fwd_ws_bytes());
    a.bytes = b.off;
    return a;
}
struct FcTmp { float *dh, *dz, *wt; void* ws; size_t wsb, bytes; };
FcTmp fc_tmp(void* base, long long U, int D) {
    Bump b{(char*)base, 0};
    FcTmp t;
    t.dh = b.take<float>((size_t)U * D);
    t.dz = b.take<float>((size_t)U * 3 * D);
    t.wt = b.take<float>((size_t)3 * D * D);
    t.wsb = umax(umax(lin_bwd_ws(U, D, 3 * D), stage_ln_bwd_ws_bytes(3 * D)), stage_ln_bwd_ws_bytes(D));
    t.wsb = umax(t.wsb, stage_cat3_dx_ln_bwd_ws_bytes(U, D, 1, 1));
    t.wsb = umax(t.wsb, stage_cat3_bwd_dw_ws_bytes(U, D, 1, 1));
    t.ws = b.take<char>(t.wsb);
    t.bytes = b.off;
    return t;
}
}  // namespace

extern "C" size_t stage_grp_concat_fc_arena_bytes(long long U, int D) { return fc_layout(nullptr, U, D).bytes; }
extern "C" size_t stage_grp_concat_fc_bwd_tmp_bytes(long long U, int D) { return fc_tmp(nullptr, U, D).bytes; }

extern "C" int stage_grp_concat_fc_fwd(const float* s, const float* v, const float* const* P, float* out, void* arena,
                                       size_t arena_bytes, int* flags, long long U, int D, float p,
                                       const unsigned long long* seeds, void* st) {
    if (U <= 0 || D % 4 || D > 256) return STAGE_ERR_SHAPE;
    FcArena a = fc_layout(arena, U, D);
    if (arena_bytes < a.bytes) return STAGE_ERR_WORKSPACE;
    bool fused = false;
    if (stage_cat3_ln_gemm_fwd_supported(U, D, 1, 1) && lin_wants_mask(a.z, P[2], U, D, 3 * D, 1)) {
        const bool dw = stage_cat3_bwd_dw_supported(U, D, 1, 1) != 0;          // flags[0] = 2: z not written (see G3)
        const int rc = stage_cat3_ln_gemm_fwd(s, v, P[0], P[1], P[2], P[3], dw ? nullptr : a.z, a.mean3, a.rstd3, a.h, a.mask, U, D, 1, 1, EPS_LN,
                                              p, seeds[0], a.fwd_ws, stage_cat3_ln_gemm_fwd_ws_bytes(), st);
        if (rc == 0) { flags[0] = dw ? 2 : 1; fused = true; }
        else if (rc != STAGE_ERR_SHAPE) return rc;
    }
    if (!fused) {
        TRY(stage_cat3_layernorm_fwd(s, v, P[0], P[1], a.z, a.mean3, a.rstd3, U, D, 1, 1, EPS_LN, p, seeds[0], st));
        TRY(lin_fwd(a.z, P[2], P[3], a.h, a.mask, &flags[0], U, D, 3 * D, 1, st));
    }
    return stage_layernorm_fwd(a.h, nullptr, 0, nullptr, P[4], P[5], out, a.mean, a.rstd, U, D, EPS_LN, 0.f, 0ull, st);
}

// ds, dv (U, D) ; grads: dln3_g dln3_b dW dc dln_g dln_b
extern "C" int stage_grp_concat_fc_bwd(const float* dout, const float* s, const float* v, const float* const* P, float* const* G,
                                       float* ds, float* dv, const void* arena, size_t arena_bytes, const int* flags, void* tmp,
                                       size_t tmp_bytes, long long U, int D, float p, const unsigned long long* seeds, void* st) {
    FcArena a = fc_layout((void*)arena, U, D);
    FcTmp t = fc_tmp(tmp, U, D);
    if (arena_bytes < a.bytes || tmp_bytes < t.bytes) return STAGE_ERR_WORKSPACE;
    TRY(stage_layernorm_bwd(dout, a.h, a.mean, a.rstd, P[4], t.dh, nullptr, G[4], G[5], U, D, 0.f, 0ull, t.ws, stage_ln_bwd_ws_bytes(D), st));
    if (flags[0] == 2)                                    // no saved z: the Linear's gradients come out of the same kernel
        return stage_cat3_bwd_dw(t.dh, a.mask, P[2], s, v, a.mean3, a.rstd3, P[0], P[1], ds, dv, G[0], G[1], G[2], G[3], U, D, 1, 1, p, seeds[0],
                                 t.ws, stage_cat3_bwd_dw_ws_bytes(U, D, 1, 1), st);
    if (flags[0] && stage_cat3_dx_ln_bwd_supported(U, D, 1, 1)) {      // no 3D-wide gradient tensor (csrc/cat3_fused.hip)
        TRY(lin_bwd(t.dh, a.z, a.h, a.mask, flags[0], 1, P[2], t.wt, nullptr, G[2], G[3], U, D, 3 * D, t.ws, t.wsb, st));
        const int rc = stage_cat3_dx_ln_bwd(t.dh, a.mask, P[2], s, v, a.mean3, a.rstd3, P[0], ds, dv, G[0], G[1], U, D, 1, 1, p, seeds[0],
                                            t.ws, stage_cat3_dx_ln_bwd_ws_bytes(U, D, 1, 1), st);
        if (rc != STAGE_ERR_SHAPE) return rc;
    }
    TRY(lin_bwd(t.dh, a.z, a.h, a.mask, flags[0], 1, P[2], t.wt, t.dz, G[2], G[3], U, D, 3 * D, t.ws, t.wsb, st));
    return stage_cat3_layernorm_bwd(t.dz, s, v, a.mean3, a.rstd3, P[0], ds, dv, G[0], G[1], U, D, 1, 1, p, seeds[0], t.ws,
                                    stage_ln_bwd_ws_bytes(3 * D), st);
}

// =====================================================================================================================
// G5  temporal head, layer 0 (model/stage.py:469-482 residual_temporal_predictor, LinearWrapper :15-32):
//     h = ReLU(Linear(drop(LN_p(enc)))) ; first = enc + h ; t_st = Linear_st(drop(LN_st(first))) ; t_ed = Linear_ed(drop(LN_ed(first)))
//     params: lnp_g lnp_b Wp cp lns_g lns_b Ws cs lne_g lne_b We ce ; seeds: [LN_p, LN_st, LN_ed dropout] ; flags: [mask_p]
//     outputs: first (R, D), t_st (R), t_ed (R)
// =====================================================================================================================
namespace {
struct ThArena { float *yp, *meanp, *rstdp, *h, *ys, *means, *rstds, *ye, *meane, *rstde; unsigned* mask; size_t bytes; };
ThArena th_layout(void* base, long long R, int D) {
    Bump b{(char*)base, 0};
    ThArena a;
    a.yp = b.take<float>((size_t)R * D);
    a.meanp = b.take<float>((size_t)R);
    a.rstdp = b.take<float>((size_t)R);
    a.h = b.take<float>((size_t)R * D);
    a.mask = b.take<unsigned>(lin_mask_words(R, D));
    a.ys = b.take<float>((size_t)R * D);
    a.means = b.take<float>((size_t)R);
    a.rstds = b.take<float>((size_t)R);
    a.ye = b.take<float>((size_t)R * D);
    a.meane = b.take<float>((size_t)R);
    a.rstde = b.take<float>((size_t)R);
    a.bytes = b.off;
    return a;
}
struct ThTmp { float *dye, *dfirst, *dys, *G, *dyp, *wt; void* ws; size_t wsb, bytes; };
ThTmp th_tmp(void* base, long long R, int D) {
    Bump b{(char*)base, 0};
    ThTmp t;
    t.dye = b.take<float>((size_t)R * D);
    t.dfirst = b.take<float>((size_t)R * D);
    t.dys = b.take<float>((size_t)R * D);
    t.G = b.take<float>((size_t)R * D);
    t.dyp = b.take<float>((size_t)R * D);
    t.wt = b.take<float>((size_t)D * D);
    t.wsb = umax(umax(lin_bwd_ws(R, D, D), lin_bwd_ws(R, 1, D)), stage_ln_bwd_ws_bytes(D));
    t.ws = b.take<char>(t.wsb);
    t.bytes = b.off;
    return t;
}
}  // namespace

extern "C" size_t stage_grp_temporal_head_arena_bytes(long long R, int D) { return th_layout(nullptr, R, D).bytes; }
extern "C" size_t stage_grp_temporal_head_bwd_tmp_bytes(long long R, int D) { return th_tmp(nullptr, R, D).bytes; }

extern "C" int stage_grp_temporal_head_fwd(const float* enc, const float* const* P, float* first, float* t_st, float* t_ed,
                                           void* arena, size_t arena_bytes, int* flags, long long R, int D, float p,
                                           const unsigned long long* seeds, void* st) {
    if (R <= 0 || D % 4 || D > 1024) return STAGE_ERR_SHAPE;
    ThArena a = th_layout(arena, R, D);
    if (arena_bytes < a.bytes) return STAGE_ERR_WORKSPACE;
    int none = 0;
    TRY(stage_layernorm_fwd(enc, nullptr, 0, nullptr, P[0], P[1], a.yp, a.meanp, a.rstdp, R, D, EPS_LN, p, seeds[0], st));
    TRY(lin_fwd(a.yp, P[2], P[3], a.h, a.mask, &flags[0], R, D, D, 1, st));
    TRY(stage_layernorm_fwd(a.h, enc, 0, first, P[4], P[5], a.ys, a.means, a.rstds, R, D, EPS_LN, p, seeds[1], st));
    TRY(lin_fwd(a.ys, P[6], P[7], t_st, nullptr, &none, R, 1, D, 0, st));
    TRY(stage_layernorm_fwd(first, nullptr, 0, nullptr, P[8], P[9], a.ye, a.meane, a.rstde, R, D, EPS_LN, p, seeds[2], st));
    return lin_fwd(a.ye, P[10], P[11], t_ed, nullptr, &none, R, 1, D, 0, st);
}

// d_first may be NULL (no gradient arrived on the exported sum).  d_enc (R, D).  grads in params order.
extern "C" int stage_grp_temporal_head_bwd(const float* d_first, const float* d_st, const float* d_ed, const float* enc,
                                           const float* first, const float* const* P, float* const* G, float* d_enc,
                                           const void* arena, size_t arena_bytes, const int* flags, void* tmp, size_t tmp_bytes,
                                           long long R, int D, float p, const unsigned long long* seeds, void* st) {
    ThArena a = th_layout((void*)arena, R, D);
    ThTmp t = th_tmp(tmp, R, D);
    if (arena_bytes < a.bytes || tmp_bytes < t.bytes) return STAGE_ERR_WORKSPACE;
    // end scorer: Linear(D -> 1), LayerNorm of `first`; its input gradient joins what arrived on `first` from the pooling path
    TRY(lin_bwd(d_ed, a.ye, nullptr, nullptr, 0, 0, P[10], t.wt, t.dye, G[10], G[11], R, 1, D, t.ws, t.wsb, st));
    TRY(stage_layernorm_bwd(t.dye, first, a.meane, a.rstde, P[8], t.dfirst, d_first, G[8], G[9], R, D, p, seeds[2], t.ws, stage_ln_bwd_ws_bytes(D), st));
    // start scorer: LayerNorm of (h + enc) with the exported sum `first`: total gradient of the sum = own path + t.dfirst
    TRY(lin_bwd(d_st, a.ys, nullptr, nullptr, 0, 0, P[6], t.wt, t.dys, G[6], G[7], R, 1, D, t.ws, t.wsb, st));
    TRY(stage_layernorm_bwd(t.dys, first, a.means, a.rstds, P[4], t.G, t.dfirst, G[4], G[5], R, D, p, seeds[1], t.ws, stage_ln_bwd_ws_bytes(D), st));
    // projection: t.G is the gradient of h (and, as the residual, of enc)
    TRY(lin_bwd(t.G, a.yp, a.h, a.mask, flags[0], 1, P[2], t.wt, t.dyp, G[2], G[3], R, D, D, t.ws, t.wsb, st));
    return stage_layernorm_bwd(t.dyp, enc, a.meanp, a.rstdp, P[0], d_enc, t.G, G[0], G[1], R, D, p, seeds[0], t.ws, stage_ln_bwd_ws_bytes(D), st);
}

// =====================================================================================================================
// RAGGED TOKEN ROWS (csrc/ragged.hip, include/stage_hip.h): the groups that touch the (N, 5, Li, Lqa, .) tensors, on live rows only.
//   U = live (compact) rows, Ucap >= U the row count the caller sized the arena for (a few distinct sizes per run instead of one per
//   batch), Fc = rows of the frame-compact attention output (incl. the dump slots), T = device int32 tables:
//     T[0] fmap   [N*Li] slot of every frame (< 0: dead) | [N] slots of the example | [N] first sequence of the example
//     T[1] gdesc  (N*NA, 4): first compact row, live words Lc, slots, first frame-compact sequence
//     T[2] seq    (S, 4): first compact row, length, group g, dense output row g*Li + i         (one per (group, live frame))
//     T[3] rowinfo (U, 4) from stage_rag_rowinfo
//     T[5] wtab   balanced work table of the fused [a, b, a*b] backward (stage_cat3_dx_ln_bwd_rag), or NULL
//     T[4] cq     (N*Li, 2) or NULL: the context stream itself is ragged -- frame f = rows cq[f].x .. + cq[f].y - 1 of ctx / d_ctx
//                 (Uc rows in all: its valid words / regions + the halo of the input encoder's convolutions); NULL: dense (N, Li, Lr, D)
// G3r  as G3; `mixed` is (U, D) compact, S_raw / S_norm stay dense.  The backward overwrites the arena's copy of the attention
//      output with its gradient (in place: a row is read before it is written): ONE backward per forward.
// G2r  as G2 pooled: x (U, D) compact, out (N*NA*Li, D) dense; n_conv >= 1; the word mask comes from qa_mask (N*NA, Lqa).
// =====================================================================================================================
namespace {
struct QaRagArena { float *Cn, *A, *z, *mean, *rstd; unsigned* mask; void* fwd_ws; size_t bytes; };
QaRagArena qa_rag_layout(void* base, int N, int NA, int Lqa, int D, long long Ucap, long long Fc) {
    Bump b{(char*)base, 0};
    QaRagArena a;
    a.Cn = b.take<float>((size_t)N * NA * Lqa * D);
    a.A = b.take<float>((size_t)Fc * D);
    a.z = b.take<float>((size_t)Ucap * 3 * D);
    a.mean = b.take<float>((size_t)Ucap);
    a.rstd = b.take<float>((size_t)Ucap);
    a.mask = b.take<unsigned>(lin_mask_words(Ucap, D));
    a.fwd_ws = b.take<char>(stage_cat3_ln_gemm_fwd_ws_bytes());
    a.bytes = b.off;
    return a;
}
struct QaRagTmp { float *Qn, *dQn, *dCn, *wt; void* ws; size_t wsb, bytes; };
QaRagTmp qa_rag_tmp(void* base, int N, int NA, int Li, int Lqa, int Lr, int D, long long Ucap, long long Uc) {
    Bump b{(char*)base, 0};
    QaRagTmp t;
    const size_t Qe = (size_t)Uc * D;
    t.Qn = b.take<float>(Qe);
    t.dQn = b.take<float>(Qe);
    t.dCn = b.take<float>((size_t)N * NA * Lqa * D);
    t.wt = b.take<float>((size_t)3 * D * D);
    t.wsb = umax(lin_bwd_ws(Ucap, D, 3 * D), stage_cat3_dx_ln_bwd_rag_ws_bytes(N * NA, Li, Lqa));
    t.wsb = umax(t.wsb, stage_cat3_bwd_dw_rag_ws_bytes(N * NA, Lqa));
    t.wsb = umax(t.wsb, stage_str_attn_bwd_fused_ws_bytes(N, NA, Li, Lqa, D));
    t.ws = b.take<char>(t.wsb);
    t.bytes = b.off;
    return t;
}
bool qa_rag_ok(int N, int NA, int Li, int Lqa, int Lr, int D, long long U, long long Fc) {
    return N > 0 && NA > 0 && Li > 0 && U > 0 && D == 128 && Lr >= 2 && Lr <= 64 && (Lr & 1) == 0 && Lqa >= 4 && Lqa <= 40 && NA * Lqa <= 256 &&
           (long long)NA * (Li + 1) * Lqa * D < (1ll << 29) && stage_cat3_ln_gemm_fwd_rag_supported(U, (long long)N * NA * Lqa, Fc, D) &&
           stage_cat3_dx_ln_bwd_rag_supported(U, Fc, D, N * NA, Li, Lqa) &&
           stage_gemm_mask_supported(U > 4096 ? U : 4096, D, 3 * D);   // (off in the exact-fp32 / no-mask developer modes; any row count)
}
}  // namespace

extern "C" int stage_grp_qa_ctx_rag_supported(int N, int NA, int Li, int Lqa, int Lr, int D, long long U, long long Fc) {
    return qa_rag_ok(N, NA, Li, Lqa, Lr, D, U, Fc) ? 1 : 0;
}
extern "C" size_t stage_grp_qa_ctx_rag_arena_bytes(int N, int NA, int Lqa, int D, long long Ucap, long long Fc) {
    return qa_rag_layout(nullptr, N, NA, Lqa, D, Ucap, Fc).bytes;
}
extern "C" size_t stage_grp_qa_ctx_rag_bwd_tmp_bytes(int N, int NA, int Li, int Lqa, int Lr, int D, long long Ucap, long long Uc) {
    return qa_rag_tmp(nullptr, N, NA, Li, Lqa, Lr, D, Ucap, Uc).bytes;
}

extern "C" int stage_grp_qa_ctx_rag_fwd(const float* qa, const float* ctx, const float* qa_mask, const float* ctx_mask,
                                        const float* const* P, float* mixed, float* S_raw, float* S_norm, const int* const* T,
                                        void* arena, size_t arena_bytes, int* flags, int N, int NA, int Li, int Lqa, int Lr, int D,
                                        long long U, long long Ucap, long long Fc, long long Uc, float scale, float p,
                                        const unsigned long long* seeds, void* st) {
    if (!qa_rag_ok(N, NA, Li, Lqa, Lr, D, U, Fc) || Ucap < U || !al16(P[2]) || Uc < 1) return STAGE_ERR_SHAPE;
    QaRagArena a = qa_rag_layout(arena, N, NA, Lqa, D, Ucap, Fc);
    if (arena_bytes < a.bytes) return STAGE_ERR_WORKSPACE;
    TRY(stage_l2norm_fwd(qa, a.Cn, nullptr, (long long)N * NA * Lqa, D, EPS_L2, p, seeds[0], st));
    TRY(stage_str_attn_fwd_fc(a.Cn, ctx, qa_mask, ctx_mask, a.A, S_raw, S_norm, T[0], T[4], N, NA, Li, Lqa, Lr, D, scale, p, seeds[1], st));
    // flags[0] = 2: z is not written -- the backward rebuilds it and forms the Linear's gradients itself (csrc/cat3_bwd_dw.hip; needs
    // the balanced work table T[5])
    const bool dw = T[5] != nullptr && stage_cat3_bwd_dw_rag_supported(U, Fc, D, N * NA, Li, Lqa) != 0;
    TRY(stage_cat3_ln_gemm_fwd_rag(qa, a.A, P[0], P[1], P[2], P[3], dw ? nullptr : a.z, a.mean, a.rstd, mixed, a.mask, T[3], U,
                                   (long long)N * NA * Lqa, Fc, D, EPS_LN, p, seeds[2], a.fwd_ws, stage_cat3_ln_gemm_fwd_ws_bytes(), st));
    flags[0] = dw ? 2 : 1;
    return 0;
}

// d_mixed (U, D) compact; dS_ext dense or NULL.  Outputs as stage_grp_qa_ctx_bwd.  The arena's attention output is overwritten.
extern "C" int stage_grp_qa_ctx_rag_bwd(const float* d_mixed, const float* dS_ext, const float* qa, const float* ctx,
                                        const float* ctx_mask, const float* mixed, const float* S_norm, const float* const* P,
                                        float* const* G, float* d_qa, float* d_ctx, const int* const* T, void* arena,
                                        size_t arena_bytes, const int* flags, void* tmp, size_t tmp_bytes, int N, int NA, int Li,
                                        int Lqa, int Lr, int D, long long U, long long Ucap, long long Fc, long long Uc, float scale,
                                        float p, const unsigned long long* seeds, void* st) {
    QaRagArena a = qa_rag_layout(arena, N, NA, Lqa, D, Ucap, Fc);
    QaRagTmp t = qa_rag_tmp(tmp, N, NA, Li, Lqa, Lr, D, Ucap, Uc);
    if (arena_bytes < a.bytes || tmp_bytes < t.bytes) return STAGE_ERR_WORKSPACE;
    if (!al16(d_mixed)) return STAGE_ERR_SHAPE;
    const long long Crows = (long long)N * NA * Lqa, Qrows = Uc;
    // weight / bias gradient of the Linear (contracts over the saved normalised concat), then its input gradient fused with the
    // LayerNorm backward: da accumulated over the live frames, db written over the attention output it came from
    if (flags[0] == 2) {
        TRY(stage_cat3_bwd_dw_rag(d_mixed, a.mask, P[2], qa, a.A, a.mean, a.rstd, P[0], P[1], d_qa, a.A, G[0], G[1], G[2], G[3], T[1], T[5], U, Fc,
                                  D, N * NA, Li, Lqa, p, seeds[2], t.ws, stage_cat3_bwd_dw_rag_ws_bytes(N * NA, Lqa), st));
    } else {
        TRY(lin_bwd(d_mixed, a.z, mixed, a.mask, 1, 1, P[2], t.wt, nullptr, G[2], G[3], U, D, 3 * D, t.ws, t.wsb, st));
        TRY(stage_cat3_dx_ln_bwd_rag(d_mixed, a.mask, P[2], qa, a.A, a.mean, a.rstd, P[0], d_qa, a.A, G[0], G[1], T[1], T[5], U, Fc, D, N * NA,
                                     Li, Lqa, p, seeds[2], t.ws, stage_cat3_dx_ln_bwd_rag_ws_bytes(N * NA, Li, Lqa), st));
    }
    TRY(stage_rag_zero_dump(a.A, T[0], N, NA, Li, Lqa, D, st));
    TRY(stage_l2norm_fwd(ctx, t.Qn, nullptr, Qrows, D, EPS_L2, p, seeds[1], st));
    TRY(stage_str_attn_bwd_fused_fc(a.A, dS_ext, a.Cn, ctx, t.Qn, S_norm, ctx_mask, d_ctx, t.dQn, t.dCn, T[0], T[4], N, NA, Li, Lqa, Lr, D, scale,
                                    t.ws, stage_str_attn_bwd_fused_ws_bytes(N, NA, Li, Lqa, D), st));
    TRY(stage_l2norm_bwd(t.dCn, qa, d_qa, Crows, D, EPS_L2, p, seeds[0], 1, st));
    return stage_l2norm_bwd(t.dQn, ctx, d_ctx, Qrows, D, EPS_L2, p, seeds[1], 1, st);
}

namespace {
EncArena enc_rag_layout(void* base, long long Ucap, long long Rd, int D, int n_conv) {
    Bump b{(char*)base, 0};
    EncArena a;
    const size_t R = (size_t)Ucap;
    for (int i = 0; i < n_conv; i++) {
        a.h[i] = b.take<float>(R * D);
        a.s[i] = b.take<float>(R * D);
        a.mean[i] = b.take<float>(R);
        a.rstd[i] = b.take<float>(R);
        a.g[i] = b.take<float>(R * D);
        a.mask[i] = b.take<unsigned>(lin_mask_words((long long)R, D));
    }
    a.sf = b.take<float>(R * D);
    a.meanf = b.take<float>(R);
    a.rstdf = b.take<float>(R);
    a.yf = nullptr;
    a.idx = b.take<int>((size_t)Rd * D);
    a.bytes = b.off;
    return a;
}
EncTmp enc_rag_tmp(void* base, long long Ucap, int D, int k) {
    Bump b{(char*)base, 0};
    EncTmp t;
    const size_t R = (size_t)Ucap;
    t.Ga = b.take<float>(R * D);
    t.Gb = b.take<float>(R * D);
    t.dh = b.take<float>(R * D);
    t.dyf = nullptr;
    t.wt = b.take<float>((size_t)D * D);
    t.wsb = umax(umax(stage_ln_bwd_ws_bytes(D), stage_ln_dwconv_bwd_ws_bytes(D, k)), lin_bwd_ws(Ucap, D, D));
    t.ws = b.take<char>(t.wsb);
    t.bytes = b.off;
    return t;
}
}  // namespace

extern "C" size_t stage_grp_encoder_rag_arena_bytes(long long Ucap, long long Rd, int D, int n_conv) {
    if (n_conv > ENC_MAX_CONV) return 0;
    return enc_rag_layout(nullptr, Ucap, Rd, D, n_conv).bytes;
}
extern "C" size_t stage_grp_encoder_rag_bwd_tmp_bytes(long long Ucap, int D, int k) { return enc_rag_tmp(nullptr, Ucap, D, k).bytes; }

// x (U, D) compact; pe (>= Lqa, D) position table; qa_mask (groups, Lqa); out (Rd, D) dense: the masked max over the words of every
// (group, frame).  T as above (seq has S entries).  qa_mask == NULL: no pooling -- out is (U, D), the block's output on the same
// ragged sequences (the input encoder over a ragged context stream: T[2] = one entry per live frame, Rd is ignored)
extern "C" int stage_grp_encoder_rag_fwd(const float* x, const float* pe, const float* qa_mask, const float* const* P, float* out,
                                         const int* const* T, void* arena, size_t arena_bytes, int* flags, long long U, long long Ucap,
                                         long long S, long long Rd, int Lqa, int D, int n_conv, int k, float p,
                                         const unsigned long long* seeds, void* st) {
    const bool pooled = qa_mask != nullptr;
    if (U <= 0 || S <= 0 || Ucap < U || n_conv < 1 || n_conv > ENC_MAX_CONV || !ln_dwconv_ok(D, k) || (pooled && D != 128) || Lqa < 1 || D > 1024)
        return STAGE_ERR_SHAPE;
    if (!pooled) Rd = 0;
    EncArena a = enc_rag_layout(arena, Ucap, Rd, D, n_conv);
    if (arena_bytes < a.bytes) return STAGE_ERR_WORKSPACE;
    const float* pending = x;
    const float* cur = nullptr;
    for (int i = 0; i < n_conv; i++) {
        const float* const* Q = P + 6 * i;
        const bool drop = (i % 2) == 0;
        const unsigned long long seed = drop ? seeds[i / 2] : 0ull;
        TRY(stage_ln_dwconv_rag_fwd(pending, cur, i == 0 ? pe : nullptr, a.s[i], Q[0], Q[1], Q[2], Q[3], a.h[i], a.mean[i], a.rstd[i], T[2], S,
                                    Lqa, D, k, EPS_LN, drop ? p : 0.f, seed, st));
        cur = a.s[i];
        TRY(lin_fwd(a.h[i], Q[4], Q[5], a.g[i], a.mask[i], &flags[i], U, D, D, 1, st));
        pending = a.g[i];
    }
    const float* const* F = P + 6 * n_conv;
    flags[n_conv] = pooled ? 1 : 0;
    if (!pooled) return stage_layernorm_fwd(pending, cur, 0, a.sf, F[0], F[1], out, a.meanf, a.rstdf, U, D, EPS_LN, 0.f, 0ull, st);
    TRY(stage_rag_fill_pooled(out, a.idx, Rd, D, st));
    return stage_ln_masked_max_rag_fwd(pending, cur, a.sf, F[0], F[1], qa_mask, out, a.idx, a.meanf, a.rstdf, T[2], S, Lqa, D, EPS_LN, st);
}

// dout (Rd, D) dense; dx (U, D) compact (may be NULL); grads in params order
extern "C" int stage_grp_encoder_rag_bwd(const float* dout, const float* qa_mask, const float* const* P, float* const* Gr, float* dx,
                                         const int* const* T, const void* arena, size_t arena_bytes, const int* flags, void* tmp,
                                         size_t tmp_bytes, long long U, long long Ucap, long long S, long long Rd, int Lqa, int D,
                                         int n_conv, int k, float p, const unsigned long long* seeds, void* st) {
    if (n_conv < 1 || n_conv > ENC_MAX_CONV) return STAGE_ERR_SHAPE;
    const bool pooled = qa_mask != nullptr;
    if (!pooled) Rd = 0;
    EncArena a = enc_rag_layout((void*)arena, Ucap, Rd, D, n_conv);
    EncTmp t = enc_rag_tmp(tmp, Ucap, D, k);
    if (arena_bytes < a.bytes || tmp_bytes < t.bytes) return STAGE_ERR_WORKSPACE;
    const float* const* F = P + 6 * n_conv;
    float* const* GF = Gr + 6 * n_conv;
    float* G = t.Ga;
    if (pooled) TRY(stage_ln_masked_max_rag_bwd(dout, a.idx, qa_mask, a.sf, a.meanf, a.rstdf, F[0], G, GF[0], GF[1], T[3], U, D, t.ws, stage_ln_bwd_ws_bytes(D), st));
    else TRY(stage_layernorm_bwd(dout, a.sf, a.meanf, a.rstdf, F[0], G, nullptr, GF[0], GF[1], U, D, 0.f, 0ull, t.ws, stage_ln_bwd_ws_bytes(D), st));
    for (int i = n_conv - 1; i >= 0; i--) {
        const float* const* Q = P + 6 * i;
        float* const* GQ = Gr + 6 * i;
        const bool drop = (i % 2) == 0;
        const unsigned long long seed = drop ? seeds[i / 2] : 0ull;
        TRY(lin_bwd(G, a.h[i], a.g[i], a.mask[i], flags[i], 1, Q[4], t.wt, t.dh, GQ[4], GQ[5], U, D, D, t.ws, t.wsb, st));
        float* Gn = (i == 0 && dx) ? dx : (G == t.Ga ? t.Gb : t.Ga);
        TRY(stage_ln_dwconv_rag_bwd(t.dh, a.s[i], a.mean[i], a.rstd[i], Q[0], Q[1], Q[2], Gn, G, GQ[0], GQ[1], GQ[2], GQ[3], T[2], S, Lqa, D, k,
                                    drop ? p : 0.f, seed, t.ws, stage_ln_dwconv_bwd_ws_bytes(D, k), st));
        G = Gn;
    }
    return 0;
}

// =====================================================================================================================
// Head glue: the small tensor algebra around the temporal scores, span proposals, pooling, classifier and the two auxiliary
// losses (model/stage.py:389-467, 484-555, 613-746).  In the per-op path this is ~100 tiny ATen / HIP launches issued right
// after the step's only host synchronisation (the proposal read-back), i.e. while the device has nothing else queued.
// =====================================================================================================================
namespace {

// t_scores[n, a, i, c] = t_c[(n*NA + a)*Li + i] * tm[n, i] + (1 - tm[n, i]) * (-1e10)     (model/stage.py:520-521, mask_logits)
__global__ void tscores_fwd_kernel(const float* __restrict__ t_st, const float* __restrict__ t_ed, const float* __restrict__ tm,
                                   float* __restrict__ out, long R, int NA, int Li) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const long n = r / ((long)NA * Li);
    const int i = (int)(r % Li);
    const float m = tm[n * Li + i], off = (1.0f - m) * STAGE_NEG;
    reinterpret_cast<float2*>(out)[r] = make_float2(t_st[r] * m + off, t_ed[r] * m + off);
}
__global__ void tscores_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ tm, float* __restrict__ d_st,
                                   float* __restrict__ d_ed, long R, int NA, int Li) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const long n = r / ((long)NA * Li);
    const float m = tm[n * Li + (int)(r % Li)];
    const float2 g = reinterpret_cast<const float2*>(dout)[r];
    d_st[r] = g.x * m;
    d_ed[r] = g.y * m;
}

// block-wide reductions in a fixed order (256 threads)
__device__ __forceinline__ float block_max256(float v, float* sh) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}
__device__ __forceinline__ float block_sum256(float v, float* sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// Span proposal of the ground-truth candidate (training; model/stage.py:408-418, model/model_utils.py:92-123): softmax over the
// frames of its start / end scores, arg max of the upper-triangular products p_st[i] * p_ed[j] (i <= j; the first maximal pair in
// row-major order).  One workgroup per example.  spans (6, N): st, ed, confidence, label start, label end, answer index.
#define SPAN_MAX_LI 2048
__global__ __launch_bounds__(256) void span_kernel(const float* __restrict__ t_scores, const long long* __restrict__ target,
                                                   const long long* __restrict__ lab_st, const long long* __restrict__ lab_ed,
                                                   float* __restrict__ spans, int N, int NA, int Li) {
    __shared__ float ps[SPAN_MAX_LI], pe[SPAN_MAX_LI], sh[4];
    __shared__ float bv[256];
    __shared__ int bi[256];
    const int n = blockIdx.x, tid = threadIdx.x;
    // (a target outside [0, NA) is an input error -- the reference's indexing raises; here the read is clamped and the temporal loss of
    // that example comes out as NaN, ts_loss_kernel below, so the step fails loudly instead of reading out of bounds)
    const float* x = t_scores + ((long)n * NA + min(max((int)target[n], 0), NA - 1)) * Li * 2;
    float m0 = -INFINITY, m1 = -INFINITY;
    for (int i = tid; i < Li; i += 256) {
        const float2 v = reinterpret_cast<const float2*>(x)[i];
        m0 = fmaxf(m0, v.x);
        m1 = fmaxf(m1, v.y);
    }
    m0 = block_max256(m0, sh);
    m1 = block_max256(m1, sh);
    float s0 = 0.f, s1 = 0.f;
    for (int i = tid; i < Li; i += 256) {
        const float2 v = reinterpret_cast<const float2*>(x)[i];
        const float e0 = expf(v.x - m0), e1 = expf(v.y - m1);
        ps[i] = e0;
        pe[i] = e1;
        s0 += e0;
        s1 += e1;
    }
    s0 = block_sum256(s0, sh);
    s1 = block_sum256(s1, sh);
    for (int i = tid; i < Li; i += 256) {
        ps[i] = ps[i] / s0;
        pe[i] = pe[i] / s1;
    }
    __syncthreads();
    // rows i = tid, tid + 256, ...: best j >= i
    float best = -1.f;
    int bidx = 0x7fffffff;
    for (int i = tid; i < Li; i += 256) {
        const float a = ps[i];
        for (int j = i; j < Li; j++) {
            const float v = a * pe[j];
            if (v > best) { best = v; bidx = i * Li + j; }       // ascending (i, j): strict > keeps the first maximum of this thread
        }
    }
    bv[tid] = best;
    bi[tid] = bidx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            const float v2 = bv[tid + o];
            const int i2 = bi[tid + o];
            if (v2 > bv[tid] || (v2 == bv[tid] && i2 < bi[tid])) { bv[tid] = v2; bi[tid] = i2; }
        }
        __syncthreads();
    }
    if (tid == 0) {
        const int idx = bi[0] == 0x7fffffff ? 0 : bi[0];
        spans[0 * N + n] = (float)(idx / Li);
        spans[1 * N + n] = (float)(idx % Li);
        spans[2 * N + n] = fmaxf(bv[0], 0.f);
        spans[3 * N + n] = (float)lab_st[n];
        spans[4 * N + n] = (float)lab_ed[n];
        spans[5 * N + n] = (float)target[n];
    }
}

// Local (windowed) + global pooling of the proposals (model/stage.py:420-467): output row r = (proposal p, candidate a) reads the
// frames of source row src[p]*NA + a.  pooled[r, 0:D] = masked max over the window, pooled[r, D:2D] = glob[source row].
__global__ __launch_bounds__(256) void pool_fwd_kernel(const float* __restrict__ x, const float* __restrict__ m,
                                                       const float* __restrict__ glob, const int* __restrict__ src,
                                                       const int* __restrict__ win, float* __restrict__ pooled,
                                                       int* __restrict__ idx, long Rn, int NA, int L, int D4) {
    const long total = Rn * D4;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long r = e / D4;
        const int q = (int)(e % D4);
        const long p = r / NA;
        const long sr = (long)src[p] * NA + (r % NA);
        const int st = max(0, win[2 * p]), ed = min(L, win[2 * p + 1]);
        float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        int4 bi = make_int4(-1, -1, -1, -1);
        const float* px = x + (sr * L) * (long)D4 * 4 + 4 * q;
        // 8 frames per step, all 16 loads issued before the first compare (a one-frame loop keeps a single dependent load in flight
        // per lane: 100 us for a few hundred rows); ascending order with strict >: the first maximum, as torch.max
        for (int l0 = st; l0 < ed; l0 += 8) {
            float4 v[8];
            float mk[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int l = min(l0 + u, ed - 1);
                v[u] = ld4(px + (long)l * D4 * 4);
                mk[u] = m[sr * L + l];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int l = l0 + u;
                if (l < ed) {
                    const float off = (1.0f - mk[u]) * STAGE_NEG;
                    const float4 w = make_float4(v[u].x * mk[u] + off, v[u].y * mk[u] + off, v[u].z * mk[u] + off, v[u].w * mk[u] + off);
                    if (w.x > best.x) { best.x = w.x; bi.x = l; }
                    if (w.y > best.y) { best.y = w.y; bi.y = l; }
                    if (w.z > best.z) { best.z = w.z; bi.z = l; }
                    if (w.w > best.w) { best.w = w.w; bi.w = l; }
                }
            }
        }
        float* po = pooled + r * (long)D4 * 8 + 4 * q;
        *reinterpret_cast<float4*>(po) = best;
        *reinterpret_cast<float4*>(po + D4 * 4) = ld4(glob + sr * (long)D4 * 4 + 4 * q);
        *reinterpret_cast<int4*>(idx + e * 4) = bi;
    }
}
// dx[source row, l, d] = m[l] * ( sum_j [idx[r_j, d] == l] dpooled[r_j, d]  +  [idx_g[d] == l] sum_j dpooled[r_j, D + d] ),
// r_j = the (at most two) proposals of the example: inv (N, 2), -1 = none.  The whole dx is written.
__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ dpooled, const int* __restrict__ idx,
                                                       const int* __restrict__ idx_g, const float* __restrict__ m,
                                                       const int* __restrict__ inv, float* __restrict__ dx, long Rs, int NA, int L,
                                                       int D4) {
    const long total = Rs * L * D4;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int q = (int)(e % D4);
        const long rl = e / D4;
        const int l = (int)(rl % L);
        const long sr = rl / L;
        const long n = sr / NA;
        const int a = (int)(sr % NA);
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f), gg = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int p = inv[2 * n + j];
            if (p < 0) continue;
            const long r = (long)p * NA + a;
            const int4 bi = *reinterpret_cast<const int4*>(idx + (r * D4 + q) * 4);
            const float4 g = ld4(dpooled + r * (long)D4 * 8 + 4 * q);
            o.x += bi.x == l ? g.x : 0.f;
            o.y += bi.y == l ? g.y : 0.f;
            o.z += bi.z == l ? g.z : 0.f;
            o.w += bi.w == l ? g.w : 0.f;
            gg = f4add(gg, ld4(dpooled + r * (long)D4 * 8 + D4 * 4 + 4 * q));
        }
        const int4 big = *reinterpret_cast<const int4*>(idx_g + (sr * D4 + q) * 4);
        const float mk = m[rl];
        o.x = (o.x + (big.x == l ? gg.x : 0.f)) * mk;
        o.y = (o.y + (big.y == l ? gg.y : 0.f)) * mk;
        o.z = (o.z + (big.z == l ? gg.z : 0.f)) * mk;
        o.w = (o.w + (big.w == l ? gg.w : 0.f)) * mk;
        *reinterpret_cast<float4*>(dx + e * 4) = o;
    }
}

// Temporal loss (model/stage.py:539-555): 0.5 * (CE_sum(start scores of the ground-truth candidate, st) + CE_sum(end scores, ed)),
// and its gradient w.r.t. t_scores in the same pass.  One workgroup per (example, local candidate); examples whose ground-truth
// candidate is not local (candidate-sharded batches, cand_offset) contribute nothing here.
__global__ __launch_bounds__(256) void ts_loss_kernel(const float* __restrict__ t, const long long* __restrict__ target,
                                                      const long long* __restrict__ lab_st, const long long* __restrict__ lab_ed,
                                                      float* __restrict__ grad, float* __restrict__ part, int NA, int Li,
                                                      int cand_offset, int na_total) {
    __shared__ float sh[4];
    const int n = blockIdx.x / NA, a = blockIdx.x % NA, tid = threadIdx.x;
    const float* x = t + (long)blockIdx.x * Li * 2;
    float* g = grad + (long)blockIdx.x * Li * 2;
    const int local = (int)target[n] - cand_offset;
    if (a != local) {
        for (int i = tid; i < Li; i += 256) reinterpret_cast<float2*>(g)[i] = make_float2(0.f, 0.f);
        // not among the local candidates: nothing here (candidate-sharded batches) -- unless no rank can hold it (a target outside
        // the na_total candidates of the model: NaN, what the reference's gather would fault on)
        if (tid == 0 && a == 0 && (local < 0 || local >= NA)) part[n] = (local + cand_offset < 0 || local + cand_offset >= na_total) ? NAN : 0.f;
        return;
    }
    float m0 = -INFINITY, m1 = -INFINITY;
    for (int i = tid; i < Li; i += 256) {
        const float2 v = reinterpret_cast<const float2*>(x)[i];
        m0 = fmaxf(m0, v.x);
        m1 = fmaxf(m1, v.y);
    }
    m0 = block_max256(m0, sh);
    m1 = block_max256(m1, sh);
    float s0 = 0.f, s1 = 0.f;
    for (int i = tid; i < Li; i += 256) {
        const float2 v = reinterpret_cast<const float2*>(x)[i];
        s0 += expf(v.x - m0);
        s1 += expf(v.y - m1);
    }
    s0 = block_sum256(s0, sh);
    s1 = block_sum256(s1, sh);
    const float l0 = logf(s0), l1 = logf(s1);
    const int st = (int)lab_st[n], ed = (int)lab_ed[n];
    const bool lab_ok = st >= 0 && st < Li && ed >= 0 && ed < Li;      // nn.CrossEntropyLoss raises on such a label: NaN here, no read
    for (int i = tid; i < Li; i += 256) {
        const float2 v = reinterpret_cast<const float2*>(x)[i];
        const float p0 = expf(v.x - m0 - l0), p1 = expf(v.y - m1 - l1);
        reinterpret_cast<float2*>(g)[i] = make_float2(0.5f * (p0 - (i == st ? 1.f : 0.f)), 0.5f * (p1 - (i == ed ? 1.f : 0.f)));
    }
    if (tid == 0) {
        const float lp0 = lab_ok ? x[2 * st] - m0 - l0 : NAN, lp1 = lab_ok ? x[2 * ed + 1] - m1 - l1 : NAN;
        part[n] = -0.5f * (lp0 + lp1);
    }
}
__global__ void sum_small_kernel(const float* __restrict__ in, int n, float* __restrict__ out) {
    // one wave, fixed order: lane-strided partial sums, then the wave reduction
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 64) s += in[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[0] = s;
}

// Supervised attention loss (model/stage.py:738-745) on gathered (positive, negative) score pairs: lse: log1p(exp(alpha (s_neg - s_pos)));
// hinge: max(0, margin + s_neg - s_pos).  coef[p] = d loss / d (s_neg - s_pos).  One workgroup, fixed summation order.
__global__ __launch_bounds__(256) void att_loss_kernel(const float* __restrict__ scores, const long long* __restrict__ flat, long M,
                                                       int hinge, float alpha, float margin, float* __restrict__ coef,
                                                       float* __restrict__ loss) {
    __shared__ float sh[4];
    float s = 0.f;
    for (long p = threadIdx.x; p < M; p += 256) {
        const float sp = scores[flat[p]], sn = scores[flat[M + p]];
        if (hinge) {
            const float v = margin + sn - sp;
            s += fmaxf(v, 0.f);
            coef[p] = v > 0.f ? 1.f : 0.f;
        } else {
            const float e = expf(alpha * (sn - sp));
            s += log1pf(e);
            coef[p] = isinf(e) ? alpha : alpha * (e / (1.f + e));
        }
    }
    s = block_sum256(s, sh);
    if (threadIdx.x == 0) loss[0] = s;
}
__global__ void att_scatter_kernel(const long long* __restrict__ flat, const float* __restrict__ coef, const float* __restrict__ gout,
                                   long M, float* __restrict__ dS) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= M) return;
    const float c = coef[p] * gout[0];
    atomicAdd(dS + flat[p], -c);
    atomicAdd(dS + flat[M + p], c);
}

inline unsigned blocks_for(long n, int bs) { return (unsigned)((n + bs - 1) / bs); }
inline unsigned capped_grid(long n, int bs) {
    long g = (n + bs - 1) / bs;
    return (unsigned)(g > 65536 * 4 ? 65536 * 4 : (g < 1 ? 1 : g));
}
}  // namespace

// t_st, t_ed (N*NA*Li) + frame mask (N, Li) -> masked scores (N, NA, Li, 2)   (model/stage.py:515-521)
extern "C" int stage_tscores_fwd(const float* t_st, const float* t_ed, const float* tm, float* out, int N, int NA, int Li, void* st) {
    const long R = (long)N * NA * Li;
    if (R <= 0) return 0;
    hipLaunchKernelGGL(tscores_fwd_kernel, dim3(blocks_for(R, 256)), dim3(256), 0, (hipStream_t)st, t_st, t_ed, tm, out, R, NA, Li);
    STAGE_LAUNCH_CHECK();
    return 0;
}
extern "C" int stage_tscores_bwd(const float* dout, const float* tm, float* d_st, float* d_ed, int N, int NA, int Li, void* st) {
    const long R = (long)N * NA * Li;
    if (R <= 0) return 0;
    hipLaunchKernelGGL(tscores_bwd_kernel, dim3(blocks_for(R, 256)), dim3(256), 0, (hipStream_t)st, dout, tm, d_st, d_ed, R, NA, Li);
    STAGE_LAUNCH_CHECK();
    return 0;
}
// spans (6, N) float: predicted start, end, confidence of the ground-truth candidate, the label's start / end, the answer index
extern "C" int stage_gt_spans(const float* t_scores, const long long* target, const long long* lab_st, const long long* lab_ed,
                              float* spans, int N, int NA, int Li, void* st) {
    if (N <= 0) return 0;
    if (Li > SPAN_MAX_LI || Li < 1) return STAGE_ERR_SHAPE;
    hipLaunchKernelGGL(span_kernel, dim3(N), dim3(256), 0, (hipStream_t)st, t_scores, target, lab_st, lab_ed, spans, N, NA, Li);
    STAGE_LAUNCH_CHECK();
    return 0;
}
// loss (1), grad (N, NA, Li, 2) = d loss / d t_scores; scratch: N floats
extern "C" int stage_ts_loss(const float* t_scores, const long long* target, const long long* lab_st, const long long* lab_ed,
                             float* loss, float* grad, float* scratch, int N, int NA, int Li, int cand_offset, int na_total, void* st) {
    if (N <= 0 || NA <= 0 || Li <= 0) return STAGE_ERR_SHAPE;
    hipLaunchKernelGGL(ts_loss_kernel, dim3(N * NA), dim3(256), 0, (hipStream_t)st, t_scores, target, lab_st, lab_ed, grad, scratch, NA,
                       Li, cand_offset, na_total > 0 ? na_total : cand_offset + NA);
    hipLaunchKernelGGL(sum_small_kernel, dim3(1), dim3(64), 0, (hipStream_t)st, scratch, N, loss);
    STAGE_LAUNCH_CHECK();
    return 0;
}
// The caller's loss line (main.py:55-60): loss = CE_sum(logits, targets) * scale + att_w * att_loss + ts_w * t_loss, and the gradient
// of the cross entropy w.r.t. the logits (scale * (softmax - onehot)) in the same pass -- one launch instead of the ~20 tiny ones of
// the eager expression and its backward, which arrive one host call at a time right behind the step's only read-back, while the
// device has nothing else queued (profiles/r06_step_idle_gaps.txt).  One workgroup: P = proposals (<= 2 N), C = candidates.
// scale: scale_dev[0] if given (multi-GPU: the global N / N_new as a device word), else scale_host.  A target outside [0, C) is
// ignored like F.cross_entropy's ignore_index (-100) when negative, NaN otherwise (the eager call raises).
__global__ __launch_bounds__(256) void train_loss_kernel(const float* __restrict__ logits, const long long* __restrict__ targets,
                                                         const float* __restrict__ att_loss, const float* __restrict__ t_loss,
                                                         const float* __restrict__ scale_dev, float scale_host, float att_w, float ts_w,
                                                         float* __restrict__ loss, float* __restrict__ dlogits, int P, int C) {
    __shared__ float sh[4];
    const int tid = threadIdx.x;
    const float scale = scale_dev ? scale_dev[0] : scale_host;
    float ce = 0.f;
    for (int r = tid; r < P; r += 256) {
        const float* x = logits + (long)r * C;
        float m = -INFINITY;
        for (int c = 0; c < C; c++) m = fmaxf(m, x[c]);
        float sum = 0.f;
        for (int c = 0; c < C; c++) sum += expf(x[c] - m);
        const float lse = m + logf(sum);
        const long long t = targets[r];
        const bool ign = t < 0;
        for (int c = 0; c < C; c++) dlogits[(long)r * C + c] = ign ? 0.f : scale * (expf(x[c] - lse) - (c == (int)t ? 1.f : 0.f));
        ce += ign ? 0.f : (t < C ? lse - x[(int)t] : NAN);
    }
    ce = block_sum256(ce, sh);
    if (tid == 0) loss[0] = ce * scale + (att_loss ? att_w * att_loss[0] : 0.f) + (t_loss ? ts_w * t_loss[0] : 0.f);
}
extern "C" int stage_train_loss(const float* logits, const long long* targets, const float* att_loss, const float* t_loss,
                                const float* scale_dev, float scale_host, float att_w, float ts_w, float* loss, float* dlogits, int P, int C,
                                void* st) {
    if (P <= 0 || C <= 0 || C > 4096) return STAGE_ERR_SHAPE;
    hipLaunchKernelGGL(train_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)st, logits, targets, att_loss, t_loss, scale_dev, scale_host,
                       att_w, ts_w, loss, dlogits, P, C);
    STAGE_LAUNCH_CHECK();
    return 0;
}
// flat: 2M int64 indices into `scores` (M positives, then M negatives); coef (M) is kept for the backward
extern "C" int stage_att_loss_fwd(const float* scores, const long long* flat, long long M, int hinge, float alpha, float margin,
                                  float* coef, float* loss, void* st) {
    if (M <= 0) return STAGE_ERR_SHAPE;
    hipLaunchKernelGGL(att_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)st, scores, flat, (long)M, hinge, alpha, margin, coef, loss);
    STAGE_LAUNCH_CHECK();
    return 0;
}
// dS (n_scores, zero-filled here) += gout[0] * coef scattered to the pair indices
extern "C" int stage_att_loss_bwd(const long long* flat, const float* coef, const float* gout, long long M, float* dS,
                                  long long n_scores, void* st) {
    TRY((int)hipMemsetAsync(dS, 0, (size_t)n_scores * sizeof(float), (hipStream_t)st));
    hipLaunchKernelGGL(att_scatter_kernel, dim3(blocks_for((long)M, 256)), dim3(256), 0, (hipStream_t)st, flat, coef, gout, (long)M, dS);
    STAGE_LAUNCH_CHECK();
    return 0;
}

// ---- G6  proposal pooling + answer classifier (model/stage.py:420-467, 526-536; LinearWrapper :15-32) -----------------------
//     first (N*NA, Li, D), mask (N*NA, Li), glob / idx_g = its global masked max (stage_masked_max_fwd, computed before the
//     proposal read-back), meta (int32, device): src[P] | win[2P] | inv[2N].  logits (P*NA).  params: ln_g ln_b W c (2D wide);
//     seeds[1]; no flags.  Backward: d_first (N*NA, Li, D) receives BOTH pooling paths (local windows and the global max).
namespace {
struct PcArena { float *pooled, *y, *mean, *rstd; int* idx; size_t bytes; };
PcArena pc_layout(void* base, long Rn, int D) {
    Bump b{(char*)base, 0};
    PcArena a;
    a.pooled = b.take<float>((size_t)Rn * 2 * D);
    a.idx = b.take<int>((size_t)Rn * D);
    a.y = b.take<float>((size_t)Rn * 2 * D);
    a.mean = b.take<float>((size_t)Rn);
    a.rstd = b.take<float>((size_t)Rn);
    a.bytes = b.off;
    return a;
}
struct PcTmp { float *dy, *dpooled, *wt; void* ws; size_t wsb, bytes; };
PcTmp pc_tmp(void* base, long Rn, int D) {
    Bump b{(char*)base, 0};
    PcTmp t;
    t.dy = b.take<float>((size_t)Rn * 2 * D);
    t.dpooled = b.take<float>((size_t)Rn * 2 * D);
    t.wt = b.take<float>((size_t)2 * D);
    t.wsb = umax(lin_bwd_ws(Rn, 1, 2 * D), stage_ln_bwd_ws_bytes(2 * D));
    t.ws = b.take<char>(t.wsb);
    t.bytes = b.off;
    return t;
}
}  // namespace

extern "C" size_t stage_grp_pool_cls_arena_bytes(long long P, int NA, int D) { return pc_layout(nullptr, (long)P * NA, D).bytes; }
extern "C" size_t stage_grp_pool_cls_bwd_tmp_bytes(long long P, int NA, int D) { return pc_tmp(nullptr, (long)P * NA, D).bytes; }

extern "C" int stage_grp_pool_cls_fwd(const float* first, const float* mask, const float* glob, const int* meta,
                                      const float* const* P, float* logits, void* arena, size_t arena_bytes, int N, int NA, int Li,
                                      int D, long long Pn, float p, const unsigned long long* seeds, void* st) {
    if (N <= 0 || Pn <= 0 || D % 4 || 2 * D > 1024) return STAGE_ERR_SHAPE;
    const long Rn = (long)Pn * NA;
    PcArena a = pc_layout(arena, Rn, D);
    if (arena_bytes < a.bytes) return STAGE_ERR_WORKSPACE;
    const int* src = meta;
    const int* win = meta + Pn;
    hipLaunchKernelGGL(pool_fwd_kernel, dim3(capped_grid(Rn * (D / 4), 256)), dim3(256), 0, (hipStream_t)st, first, mask, glob, src, win,
                       a.pooled, a.idx, Rn, NA, Li, D / 4);
    STAGE_LAUNCH_CHECK();
    TRY(stage_layernorm_fwd(a.pooled, nullptr, 0, nullptr, P[0], P[1], a.y, a.mean, a.rstd, Rn, 2 * D, EPS_LN, p, seeds[0], st));
    int none = 0;
    return lin_fwd(a.y, P[2], P[3], logits, nullptr, &none, Rn, 1, 2 * D, 0, st);
}

extern "C" int stage_grp_pool_cls_bwd(const float* d_logits, const float* mask, const int* idx_g, const int* meta,
                                      const float* const* P, float* const* G, float* d_first, const void* arena, size_t arena_bytes,
                                      void* tmp, size_t tmp_bytes, int N, int NA, int Li, int D, long long Pn, float p,
                                      const unsigned long long* seeds, void* st) {
    const long Rn = (long)Pn * NA;
    PcArena a = pc_layout((void*)arena, Rn, D);
    PcTmp t = pc_tmp(tmp, Rn, D);
    if (arena_bytes < a.bytes || tmp_bytes < t.bytes) return STAGE_ERR_WORKSPACE;
    TRY(lin_bwd(d_logits, a.y, nullptr, nullptr, 0, 0, P[2], t.wt, t.dy, G[2], G[3], Rn, 1, 2 * D, t.ws, t.wsb, st));
    TRY(stage_layernorm_bwd(t.dy, a.pooled, a.mean, a.rstd, P[0], t.dpooled, nullptr, G[0], G[1], Rn, 2 * D, p, seeds[0], t.ws,
                            stage_ln_bwd_ws_bytes(2 * D), st));
    const int* inv = meta + 3 * Pn;
    const long Rs = (long)N * NA;
    hipLaunchKernelGGL(pool_bwd_kernel, dim3(capped_grid(Rs * Li * (D / 4), 256)), dim3(256), 0, (hipStream_t)st, t.dpooled, a.idx, idx_g,
                       mask, inv, d_first, Rs, NA, Li, D / 4);
    STAGE_LAUNCH_CHECK();
    return 0;
}
