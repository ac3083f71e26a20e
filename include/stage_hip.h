/*
 * stage_hip.h -- C ABI of libstage_hip.so: the MI355X (gfx950) kernels behind model.stage.STAGE's forward/backward.
 *
 * The reference (jayleicn/TVQAplus) is pure Python/PyTorch and has no FFI; the boundary a maintainer binds is the
 * set of fused-op groups that STAGE.forward_main executes.  Every entry point below
 * names the reference lines (relative to the reference repo root) whose arithmetic it replaces.
 *
 * Conventions
 *   - all tensors are dense row-major fp32 device pointers (masks are fp32 0/1 like the reference's); `int*` where said
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every call is asynchronous on that stream
 *   - return value: 0 on success, a hipError_t (>0) from a failed launch, or a negative STAGE_ERR_* code
 *   - dropout is counter based: (seed, element index) -> keep/drop, so the backward entry points regenerate the
 *     forward mask from the same seed; p_drop = 0 disables it (eval mode)
 *   - `ws`/`ws_bytes`: caller-provided device scratch, size from the matching *_ws_bytes() query
 *   - reductions are two-stage with a fixed summation order: results are run-to-run deterministic
 */
#ifndef STAGE_HIP_H
#define STAGE_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STAGE_ERR_SHAPE (-1)      /* unsupported shape (e.g. width not a multiple of 4 / 16, Lr > 64) */
#define STAGE_ERR_WORKSPACE (-2)  /* workspace too small */

/* ---- library info -------------------------------------------------------------------------------------------- */
#define STAGE_HIP_ABI_VERSION 5   /* what stage_hip_abi_version() of a matching library returns; bumped whenever a symbol or a signature changes */
int stage_hip_abi_version(void);
const char* stage_hip_error_string(int code);
/* Measurement helpers (bench.py; no reference counterpart): events for hosts without a HIP binding, and a one-shot hook that makes the
 * next stage_str_attn_fwd call with `Lr` regions record (start, stop) on its stream around its kernel -- the K1 forward's duration
 * INSIDE a training step.  stage_k1_fwd_timer(NULL, NULL, 0) disarms.  stage_timer_elapsed_ms waits for `stop`; < 0 on error. */
void* stage_timer_create(void);
void stage_timer_destroy(void* event);
float stage_timer_elapsed_ms(void* start, void* stop);
void stage_k1_fwd_timer(void* start, void* stop, int Lr);

/* ---- K1: StructuredAttention (model/context_query_attention.py:35-101, called at model/stage.py:378) ----------
 * Cn      (N, NA, Lqa, D)  L2-normalised (+dropout) QA/context side, produced by stage_l2norm_fwd
 * Q       (N, Li, Lr, D)   raw region / subtitle-word side (normalised + dropped in-kernel for the similarity,
 *                          used raw for the weighted sum, line 81)
 * c_mask  (N, NA, Lqa), q_mask (N, Li, Lr)
 * A       (N, NA, Li, Lqa, D) ; S_raw, S_norm (N, NA, Li, Lqa, Lr)   (raw_s / s_normalized of model/stage.py:378-380)
 * limits: D % 16 == 0, D <= 256, Lr <= 64, NA*Lqa <= 256                                                        */
int stage_str_attn_fwd(const float* Cn, const float* Q, const float* c_mask, const float* q_mask, float* A,
                       float* S_raw, float* S_norm, int N, int NA, int Li, int Lqa, int Lr, int D, float scale,
                       float p_drop, unsigned long long seed, void* stream);
size_t stage_str_attn_bwd_ws_bytes(int N, int NA, int Lqa, int D);
/* dA (like A), dS_raw_ext (like S_raw, may be NULL: gradient arriving on raw_s from the attention loss),
 * Qn = stage_l2norm_fwd(Q) with the forward seed.  Outputs: dS_out (like S_raw; scratch + gradient wrt raw scores),
 * dQraw, dQn (N, Li, Lr, D) (value path / normalised path), dCn (N, NA, Lqa, D).                                 */
int stage_str_attn_bwd(const float* dA, const float* dS_raw_ext, const float* Cn, const float* Q, const float* Qn,
                       const float* S_norm, float* dS_out, float* dQraw, float* dQn, float* dCn, int N, int NA, int Li,
                       int Lqa, int Lr, int D, float scale, void* ws, size_t ws_bytes, void* stream);

/* Fused single-pass backward (D == 128, Lr even <= 64, Lqa >= 4, NA*Lqa <= 256; otherwise STAGE_ERR_SHAPE -> use
 * stage_str_attn_bwd): same mathematics (model/context_query_attention.py:58-61, 81, 95-101), but dA is read from HBM
 * once and dS never leaves the compute unit.  q_mask (N, Li, Lr) lets region tiles / frames without a valid region be
 * skipped: their gradient is exactly zero unless dS_raw_ext is non-zero in the skipped columns, which a scan of those
 * columns detects per frame (such frames are processed in full).  ws sized by stage_str_attn_bwd_fused_ws_bytes.      */
size_t stage_str_attn_bwd_fused_ws_bytes(int N, int NA, int Li, int Lqa, int D);
int stage_str_attn_bwd_fused(const float* dA, const float* dS_raw_ext, const float* Cn, const float* Q, const float* Qn,
                             const float* S_norm, const float* q_mask, float* dQraw, float* dQn, float* dCn, int N,
                             int NA, int Li, int Lqa, int Lr, int D, float scale, void* ws, size_t ws_bytes,
                             void* stream);

/* Long region rows / bf16 storage (BASELINE.json configs[4]: bf16 weights and activations with fp32 softmax accumulation,
 * D = 256, 512 subtitle words per frame): the same StructuredAttention for ANY Lr (16-region blocks, two-pass softmax) and
 * D in {16, 32, 64, 128, 256}.  storage_bf16 != 0: Cn, Q, Qn, A, dA are bf16 (masks, scores, S maps and the gradients
 * dQraw / dQn / dCn stay fp32; every product accumulates in fp32).  Qn = normalised (+dropped) Q in storage precision.
 * Backward: A is the forward output (the softmax backward's row term <P, dP> equals <dA, A>); dS_ws is an (N,NA,Li,Lqa,Lr)
 * fp32 scratch that receives the gradient wrt the raw scores; ws from stage_str_attn_long_bwd_ws_bytes.              */
int stage_str_attn_long_fwd(const void* Cn, const void* Q, const void* Qn, const float* c_mask, const float* q_mask, void* A,
                            float* S_raw, float* S_norm, int N, int NA, int Li, int Lqa, int Lr, int D, float scale,
                            int storage_bf16, void* stream);
size_t stage_str_attn_long_bwd_ws_bytes(int N, int NA, int Lqa, int D);
int stage_str_attn_long_bwd(const void* dA, const void* A, const float* dS_raw_ext, const void* Cn, const void* Q,
                            const void* Qn, const float* S_norm, float* dS_ws, float* dQraw, float* dQn, float* dCn, int N,
                            int NA, int Li, int Lqa, int Lr, int D, float scale, int storage_bf16, void* ws, size_t ws_bytes,
                            void* stream);

/* The backward with the region mask at hand (round 4): region blocks behind a frame's last valid region are skipped -- there S_ = 0,
 * so dS, their share of dCn and the dQ rows are exact zeros (model/context_query_attention.py:58-61: masked scores are cos - 1e10) --
 * unless dS_raw_ext is non-zero in them (then the frame is processed in full).  The forward skips the same blocks on its own (it has
 * q_mask): raw scores -1e10 and weights 0 are stored as constants.  ws from stage_str_attn_long_bwd_qm_ws_bytes.                   */
size_t stage_str_attn_long_bwd_qm_ws_bytes(int N, int NA, int Li, int Lqa, int D);
int stage_str_attn_long_bwd_qm(const void* dA, const void* A, const float* dS_raw_ext, const void* Cn, const void* Q, const void* Qn,
                               const float* S_norm, const float* q_mask, float* dS_ws, float* dQraw, float* dQn, float* dCn, int N,
                               int NA, int Li, int Lqa, int Lr, int D, float scale, int storage_bf16, void* ws, size_t ws_bytes,
                               void* stream);

/* ---- F.normalize(p=2, eps) (+dropout)  (model/stage.py:256, model/context_query_attention.py:95-96) ----------- */
int stage_l2norm_fwd(const float* x, float* y, float* norm_out /*may be NULL*/, long long rows, int K, float eps,
                     float p_drop, unsigned long long seed, void* stream);
int stage_l2norm_bwd(const float* dy, const float* x, float* dx, long long rows, int K, float eps, float p_drop,
                     unsigned long long seed, int accumulate, void* stream);

/* ---- nn.LayerNorm(K, eps) followed by nn.Dropout(p)  (model/stage.py:85-120,133-138, LinearWrapper :15-32,
 *      model/encoder.py:37-41,47,52).  K % 4 == 0, K <= 1024.  mean/rstd (rows) are saved for the backward.
 * Fused prologue: y = drop(LN(x + res)); res may be NULL.  res_period = 0: res is (rows, K) -- the residual adds of
 * model/encoder.py:44,50 and model/stage.py:478; res_period = L: res is the (>=L, K) position table and row r uses
 * res[r % L] (model/position_encoding.py:38-43).  sum_out (rows, K), if not NULL, receives x + res.
 * In the backward, dx may be NULL when the input needs no gradient (raw features).                            ---- */
int stage_layernorm_fwd(const float* x, const float* res, int res_period, float* sum_out, const float* gamma,
                        const float* beta, float* y, float* mean, float* rstd, long long rows, int K, float eps,
                        float p_drop, unsigned long long seed, void* stream);
size_t stage_ln_bwd_ws_bytes(int K);
/* dx = LN-backward(dy) + dx_add (dx_add may be NULL: gradient that reached the exported sum x + res directly) */
int stage_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                        float* dx, const float* dx_add, float* dgamma, float* dbeta, long long rows, int K,
                        float p_drop, unsigned long long seed, void* ws, size_t ws_bytes, void* stream);

/* ---- LayerNorm(3D) over the virtual row cat([a, b, a*b])  (model/stage.py:276-279 concat_fc, :381-385 c2q) -----
 * a may be broadcast over `rep` (the Li frames): a_row = (row / (rep*inner)) * inner + row % inner; rep = 1: none.
 * y (rows, 3D).  Backward returns da_full (rows, D) NOT yet reduced over rep (use stage_reduce_rep) and db (rows, D);
 * ws sized by stage_ln_bwd_ws_bytes(3*D).  D % 4 == 0, D <= 256.                                                  */
int stage_cat3_layernorm_fwd(const float* a, const float* b, const float* gamma, const float* beta, float* y,
                             float* mean, float* rstd, long long rows, int D, int rep, int inner, float eps,
                             float p_drop, unsigned long long seed, void* stream);
int stage_cat3_layernorm_bwd(const float* dy, const float* a, const float* b, const float* mean, const float* rstd,
                             const float* gamma, float* da_full, float* db, float* dgamma, float* dbeta, long long rows,
                             int D, int rep, int inner, float p_drop, unsigned long long seed, void* ws,
                             size_t ws_bytes, void* stream);
/* Same backward with the reduction over `rep` fused in (rep > 1, D/4 a power of two in [4,64], inner <= 64): da is
 * (rows/rep, D), already summed over the frames that share a row of `a` (the `.repeat` gradient of model/stage.py:381).
 * ws sized by stage_cat3_layernorm_bwd_reduced_ws_bytes(rows, D, rep, inner).                                        */
size_t stage_cat3_layernorm_bwd_reduced_ws_bytes(long long rows, int D, int rep, int inner);
int stage_cat3_layernorm_bwd_reduced(const float* dy, const float* a, const float* b, const float* mean,
                                     const float* rstd, const float* gamma, float* da, float* db, float* dgamma,
                                     float* dbeta, long long rows, int D, int rep, int inner, float p_drop,
                                     unsigned long long seed, void* ws, size_t ws_bytes, void* stream);
/* out[g, inner_elems] = sum_{r<rep} in[g, r, inner_elems]  (gradient of a `.repeat`/broadcast, model/stage.py:381) */
int stage_reduce_rep(const float* in, float* out, long long groups, int rep, long long inner_elems, void* stream);

/* ---- fused LayerNorm (+ residual, + dropout) -> depthwise Conv1d of the encoder blocks (model/encoder.py:37-44,
 * model/cnn.py:37-47): h = dwconv(drop(LN(x + res))); the LayerNorm output is never materialised.  x (M*L, D) rows,
 * sequences of L rows; res / res_period / sum_out as in stage_layernorm_fwd; w (D,1,k), k odd <= 9; D/4 a power of two
 * in [4, 64].  Backward: xin = the LayerNorm input (x or the exported sum), dx = d(xin) + dx_add, plus the parameter
 * gradients of both layers; ws sized by stage_ln_dwconv_bwd_ws_bytes(D, k).                                          */
int stage_ln_dwconv_fwd(const float* x, const float* res, int res_period, float* sum_out, const float* gamma,
                        const float* beta, const float* w, const float* bias, float* h, float* mean, float* rstd,
                        long long M, int L, int D, int k, float eps, float p_drop, unsigned long long seed, void* stream);
size_t stage_ln_dwconv_bwd_ws_bytes(int D, int k);
int stage_ln_dwconv_bwd(const float* dh, const float* xin, const float* mean, const float* rstd, const float* gamma,
                        const float* beta, const float* w, float* dx, const float* dx_add, float* dgamma, float* dbeta,
                        float* dw, float* db, long long M, int L, int D, int k, float p_drop, unsigned long long seed,
                        void* ws, size_t ws_bytes, void* stream);

/* ---- nn.Linear / 1x1 Conv1d on the matrix cores (fp32 in, fp32 out, fp32 accumulate) -----------------------------
 * Arithmetic: fp32-faithful products -- a two-way fp16 split with exact power-of-two scaling in the streaming kernels
 * (M >= 4096; error below an fp32 FMA chain's, DESIGN.md finding 20), the exact 3-way bf16 split / the f32 MFMA in the
 * tiled fallbacks.
 * Y[M,N] = epi((X .* [gate>0])[M,K] . W[N,K]^T + bias) ; epi: optional ReLU then optional + residual[M,N].
 * (model/stage.py:88,101,110,117,136; LinearWrapper :23; model/cnn.py:27-28,44-46; model/self_attention.py:32,46,54)
 * `gate` (M,K) fuses a ReLU backward on the input operand (dX = (dY .* [Y>0]) . W with W passed transposed).       */
int stage_gemm_nt(const float* X, const float* gate, const float* W, const float* bias, const float* residual, float* Y,
                  long long M, int N, int K, int relu, void* stream);
/* LayerNorm gain / bias gradients straight from the dX product of the Linear behind it, for a LayerNorm whose INPUT needs no
 * gradient (first layer of the input MLPs: LayerNorm -> Dropout -> Linear -> ReLU, model/stage.py:85-91, 98-104):
 *   dgamma[n] = sum_m G[m,n] x_hat[m,n] , dbeta[n] = sum_m G[m,n] ,  G = ((dY .* relu mask) . Wt^T) .* keep / (1 - p)
 * dY (M, K), gate_mask = the ReLU bit mask of the forward (stage_gemm_nt_mask) or NULL, Wt (N, K) = the weight transposed,
 * x / mean / rstd = the LayerNorm input and statistics, keep_mask = stage_dropout_keepmask of its dropout stream (NULL when
 * p_drop == 0).  The (M, N) gradient is reduced inside the GEMM epilogue and never written.                              */
int stage_gemm_nt_lnparam_supported(long long M, int N, int K);
size_t stage_gemm_nt_lnparam_ws_bytes(long long M, int N);
int stage_gemm_nt_lnparam(const float* dY, const unsigned* gate_mask, const float* Wt, const float* x, const float* mean,
                          const float* rstd, const unsigned* keep_mask, float p_drop, float* dgamma, float* dbeta, long long M,
                          int N, int K, void* ws, size_t ws_bytes, void* stream);
/* keep bits of the dropout stream stage_layernorm_fwd(p_drop, seed) applied to a (rows, K) output: [ceil(K/32)][rows] words */
int stage_dropout_keepmask(float p_drop, unsigned long long seed, unsigned* mask, long long rows, int K, void* stream);
/* dW[N,K] = sum_m (dY .* [gate>0])[m,n] X[m,k] ; db[N] = column sums (db may be NULL)                              */
size_t stage_gemm_tn_ws_bytes(long long M, int N, int K);
int stage_gemm_tn(const float* dY, const float* gate, const float* X, float* dW, float* db, long long M, int N, int K,
                  void* ws, size_t ws_bytes, void* stream);

/* ReLU bit masks (streaming kernels; check stage_gemm_mask_supported first).  The forward GEMM of a Linear + ReLU also
 * emits relu_mask_out[w][m] (uint32, word-major: w < ceil(N/32) rows of M words, bit b <=> Y[m][32w+b] > 0); the two backward GEMMs take that mask
 * instead of the fp32 gate (nn.ReLU backward, model/stage.py:88,101; model/cnn.py:46): 1/32 of the gate bytes.
 * stage_gemm_nt_mask: gate_mask refers to the columns of X (ceil(K/32) word rows of M words).  Returns STAGE_ERR_SHAPE when the
 * shape is not taken by the streaming kernel (fall back to stage_gemm_nt / stage_gemm_tn with the fp32 gate).        */
int stage_gemm_mask_supported(long long M, int N, int K);
int stage_gemm_nt_mask(const float* X, const unsigned* gate_mask, const float* W, const float* bias, float* Y,
                       unsigned* relu_mask_out, long long M, int N, int K, int relu, void* stream);
int stage_gemm_tn_mask(const float* dY, const unsigned* gate_mask, const float* X, float* dW, float* db, long long M,
                       int N, int K, void* ws, size_t ws_bytes, void* stream);

/* ---- encoder block pieces (model/encoder.py:35-44, model/position_encoding.py:38-43, model/cnn.py:23-26,44) ---- */
int stage_add_pe(const float* x, const float* pe /*(>=L, D)*/, float* y, long long M, int L, int D, void* stream);
int stage_dwconv_fwd(const float* in, const float* w /*(D,1,k)*/, const float* bias, float* out, long long M, int L,
                     int D, int k, void* stream);
size_t stage_dwconv_bwd_ws_bytes(int D, int k);
int stage_dwconv_bwd(const float* dout, const float* in, const float* w, float* din, float* dw, float* db, long long M,
                     int L, int D, int k, void* ws, size_t ws_bytes, void* stream);

/* ---- multi-head self-attention core (model/self_attention.py:56-71) with the query-row mask quirk --------------
 * q,k,v,out: (M, L, D) with heads interleaved along D (head h = columns [h*dk, (h+1)*dk)); mask (M, L);
 * probs (M, nh, L, L) saved for the backward (post-softmax, pre-dropout).  L <= 64.
 * stage_mha_core_recomputes(L, D, nh) == 1 (head width D/nh in {8, 16, 32, 64}): the matrix-core kernels run, which
 * recompute the probabilities in the backward -- `probs` is then neither written nor read and may be NULL.            */
int stage_mha_core_recomputes(int L, int D, int nh);
int stage_mha_core_fwd(const float* q, const float* k, const float* v, const float* mask, float* out, float* probs,
                       long long M, int L, int D, int nh, float p_drop, unsigned long long seed, void* stream);
int stage_mha_core_bwd(const float* dout, const float* q, const float* k, const float* v, const float* probs,
                       const float* mask, float* dq, float* dk, float* dv, long long M, int L, int D, int nh,
                       float p_drop, unsigned long long seed, void* stream);

/* ---- mask_logits + max over a sequence axis (model/stage.py:503, 425, 429-432, 456-461, 532-533) ---------------
 * x (R, L, D), mask (R, L), window (R, 2) int [st, ed) or NULL for the whole axis -> out (R, D), argmax (R, D) int */
int stage_masked_max_fwd(const float* x, const float* mask, const int* window, float* out, int* argmax, long long R,
                         int L, int D, void* stream);
int stage_masked_max_bwd(const float* dout, const int* argmax, const float* mask, float* dx, long long R, int L, int D,
                         int accumulate, void* stream);

/* ---- LayerNorm whose output only feeds a masked max over the sequence axis, in one pass ---------------------------
 * The classifier head: final_layer_norm of the cls_encoder (model/encoder.py:52) -> mask_logits + max over the Lqa words
 * (model/stage.py:503).  x, res (optional, added before the norm; sum_out = x + res is written for the backward) are
 * (R, L, K); mask (R, L); out (R, K), argmax (R, K) int; mean / rstd (R * L).  The normalised (R, L, K) tensor and, in the
 * backward, its dense gradient are never materialised.  `stage_ln_masked_max_supported`: K == 128.
 * Backward: xin = the saved sum (or x when res == NULL); dx (R, L, K) is the gradient of x and of res; workspace as
 * stage_ln_bwd_ws_bytes(K).                                                                                          */
int stage_ln_masked_max_supported(int L, int K);
int stage_ln_masked_max_fwd(const float* x, const float* res, float* sum_out, const float* gamma, const float* beta,
                            const float* mask, float* out, int* argmax, float* mean, float* rstd, long long R, int L, int K,
                            float eps, void* stream);
int stage_ln_masked_max_bwd(const float* dout, const int* argmax, const float* mask, const float* xin, const float* mean,
                            const float* rstd, const float* gamma, float* dx, float* dgamma, float* dbeta, long long R, int L,
                            int K, void* ws, size_t ws_bytes, void* stream);

/* ---- bf16 storage mode (BASELINE.json configs[4]: bf16 weights / activations, fp32 softmax and accumulation) ------
 * Same operations and argument meaning as the fp32 entry points of the same name; every pointer typed `void*` is a
 * tensor of bf16 (raw 16-bit words) instead of float.  Statistics, affine parameters, weights, biases, masks, arg-max
 * indices and all parameter gradients stay fp32; a weight is rounded to bf16 while the GEMM stages it.  The
 * attention kernels of this mode: stage_str_attn_fwd_bf16 / stage_str_attn_bwd_fused_bf16 (D == 128, rows <= 64) and
 * stage_str_attn_long_* with storage == 1 (any row length, D up to 256).  The row / conv kernels are the fp32 ones instantiated
 * on 16-bit elements; the GEMMs stream 16-bit operands with one bf16 matrix-core term (gemm_bf16_stream.hip), tiled for the
 * shapes streaming does not pay for.                                                                                   */
/* K1 fast kernels on bf16 Q / A / dA (D == 128, Lr <= 64; STAGE_ERR_SHAPE otherwise -> stage_str_attn_long_*): Cn, the
 * masks, both score maps and the three gradients leaving the backward stay fp32.                                    */
int stage_str_attn_fwd_bf16(const float* Cn, const void* Q, const float* c_mask, const float* q_mask, void* A,
                            float* S_raw, float* S_norm, int N, int NA, int Li, int Lqa, int Lr, int D, float scale,
                            float p_drop, unsigned long long seed, void* stream);
int stage_str_attn_bwd_fused_bf16(const void* dA, const float* dS_raw_ext, const float* Cn, const void* Q, const void* Qn,
                                  const float* S_norm, const float* q_mask, float* dQraw, float* dQn, float* dCn, int N,
                                  int NA, int Li, int Lqa, int Lr, int D, float scale, void* ws, size_t ws_bytes,
                                  void* stream);
int stage_layernorm_fwd_bf16(const void* x, const void* res, int res_period, void* sum_out, const float* gamma,
                             const float* beta, void* y, float* mean, float* rstd, long long rows, int K, float eps,
                             float p_drop, unsigned long long seed, void* stream);
int stage_layernorm_bwd_bf16(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                             void* dx, const void* dx_add, float* dgamma, float* dbeta, long long rows, int K,
                             float p_drop, unsigned long long seed, void* ws, size_t ws_bytes, void* stream);
int stage_cat3_layernorm_fwd_bf16(const void* a, const void* b, const float* gamma, const float* beta, void* y,
                                  float* mean, float* rstd, long long rows, int D, int rep, int inner, float eps,
                                  float p_drop, unsigned long long seed, void* stream);
/* da_full (rows, D) stays fp32 (reduce it over `rep` with stage_reduce_rep); db (rows, D) is bf16 */
int stage_cat3_layernorm_bwd_bf16(const void* dy, const void* a, const void* b, const float* mean, const float* rstd,
                                  const float* gamma, float* da_full, void* db, float* dgamma, float* dbeta,
                                  long long rows, int D, int rep, int inner, float p_drop, unsigned long long seed,
                                  void* ws, size_t ws_bytes, void* stream);
/* da (rows / rep, D) fp32 (a sum over the `rep` frames), db (rows, D) bf16; ws as for the fp32 entry point */
int stage_cat3_layernorm_bwd_reduced_bf16(const void* dy, const void* a, const void* b, const float* mean,
                                          const float* rstd, const float* gamma, float* da, void* db, float* dgamma,
                                          float* dbeta, long long rows, int D, int rep, int inner, float p_drop,
                                          unsigned long long seed, void* ws, size_t ws_bytes, void* stream);
int stage_l2norm_fwd_bf16(const void* x, void* y, float* norm_out, long long rows, int K, float eps, float p_drop,
                          unsigned long long seed, void* stream);
int stage_l2norm_bwd_bf16(const void* dy, const void* x, void* dx, long long rows, int K, float eps, float p_drop,
                          unsigned long long seed, int accumulate, void* stream);
/* dx (bf16) = add (fp32, may be NULL) + l2norm backward of (dy fp32, x bf16): folds the attention backward's two fp32
 * gradients (raw path, normalised path) into one bf16 gradient in a single pass                                      */
int stage_l2norm_bwd_mixed_bf16(const float* dy, const void* x, const float* add, void* dx, long long rows, int K, float eps,
                                float p_drop, unsigned long long seed, void* stream);
int stage_gemm_nt_bf16(const void* X, const void* gate, const float* W, const float* bias, const void* residual, void* Y,
                       long long M, int N, int K, int relu, void* stream);
size_t stage_gemm_tn_bf16_ws_bytes(long long M, int N, int K);
int stage_gemm_tn_bf16(const void* dY, const void* gate, const void* X, float* dW, float* db, long long M, int N, int K,
                       void* ws, size_t ws_bytes, void* stream);
int stage_dwconv_fwd_bf16(const void* in, const float* w, const float* bias, void* out, long long M, int L, int D, int k,
                          void* stream);
int stage_dwconv_bwd_bf16(const void* dout, const void* in, const float* w, void* din, float* dw, float* db, long long M,
                          int L, int D, int k, void* ws, size_t ws_bytes, void* stream);
int stage_ln_dwconv_fwd_bf16(const void* x, const void* res, int res_period, void* sum_out, const float* gamma,
                             const float* beta, const float* w, const float* bias, void* h, float* mean, float* rstd,
                             long long M, int L, int D, int k, float eps, float p_drop, unsigned long long seed,
                             void* stream);
int stage_ln_dwconv_bwd_bf16(const void* dh, const void* xin, const float* mean, const float* rstd, const float* gamma,
                             const float* beta, const float* w, void* dx, const void* dx_add, float* dgamma, float* dbeta,
                             float* dw, float* db, long long M, int L, int D, int k, float p_drop, unsigned long long seed,
                             void* ws, size_t ws_bytes, void* stream);
/* self-attention core on bf16 q / k / v / out: the matrix-core kernels only (stage_mha_core_recomputes == 1) */
int stage_mha_core_fwd_bf16(const void* q, const void* k, const void* v, const float* mask, void* out, long long M, int L,
                            int D, int nh, float p_drop, unsigned long long seed, void* stream);
int stage_mha_core_bwd_bf16(const void* dout, const void* q, const void* k, const void* v, const float* mask, void* dq,
                            void* dk, void* dv, long long M, int L, int D, int nh, float p_drop, unsigned long long seed,
                            void* stream);
/* self-attention core on FUSED projections (round 5): qkv (M, L, 3D) = the output of ONE Linear(D -> 3D) with the weight
 * [W_q; W_k; W_v] (model/self_attention.py:35-44: three Linears on the same input), thirds q | k | v of every row; dqkv (M, L, 3D)
 * receives dq | dk | dv and IS that Linear's output gradient.  out / dout (M, L, D).  is_bf16: storage type of all four.  Matrix-core
 * shapes only (stage_mha_core_recomputes == 1), same masking / dropout contract as stage_mha_core_fwd.                          */
int stage_mha_core_qkv_fwd(const void* qkv, const float* mask, void* out, long long M, int L, int D, int nh, float p_drop,
                           unsigned long long seed, int is_bf16, void* stream);
int stage_mha_core_qkv_bwd(const void* dout, const void* qkv, const float* mask, void* dqkv, long long M, int L, int D, int nh,
                           float p_drop, unsigned long long seed, int is_bf16, void* stream);
int stage_masked_max_fwd_bf16(const void* x, const float* mask, const int* window, void* out, int* argmax, long long R,
                              int L, int D, void* stream);
int stage_masked_max_bwd_bf16(const void* dout, const int* argmax, const float* mask, void* dx, long long R, int L, int D,
                              int accumulate, void* stream);
int stage_ln_masked_max_fwd_bf16(const void* x, const void* res, void* sum_out, const float* gamma, const float* beta,
                                 const float* mask, void* out, int* argmax, float* mean, float* rstd, long long R, int L,
                                 int K, float eps, void* stream);
int stage_ln_masked_max_bwd_bf16(const void* dout, const int* argmax, const float* mask, const void* xin, const float* mean,
                                 const float* rstd, const float* gamma, void* dx, float* dgamma, float* dbeta, long long R,
                                 int L, int K, void* ws, size_t ws_bytes, void* stream);

/* ---- backward of  y = ReLU(drop(LN_3D([a, b, a*b])) W^T + c)  without the 3D-wide gradient (csrc/cat3_fused.hip) -----------------
 * (model/stage.py:381-385 c2q_down_projection, :276-279 concat_fc.)  The Linear's input gradient (dy .* relu') W, 1.47 GB at the
 * full configuration, is neither written nor read back: one workgroup owns complete 3D-wide rows, the product stays in matrix-core
 * accumulators and the LayerNorm backward runs on them.  dy (rows, D) = gradient of the Linear's output BEFORE the ReLU gate;
 * relu_mask = the forward's ReLU bit mask ([D/32][rows], stage_gemm_nt_mask); W (D, 3D); a, b, mean, rstd, gamma, p_drop, seed as
 * for stage_cat3_layernorm_bwd*.  Outputs: rep > 1: da (rows / rep, D) summed over the frames (as ..._bwd_reduced), rep == 1:
 * da (rows, D); db (rows, D); dgamma, dbeta (3D).  D == 128, rows >= 4096 (STAGE_ERR_SHAPE otherwise).                         */
int stage_cat3_dx_ln_bwd_supported(long long rows, int D, int rep, int inner);
size_t stage_cat3_dx_ln_bwd_ws_bytes(long long rows, int D, int rep, int inner);
int stage_cat3_dx_ln_bwd(const float* dy, const unsigned* relu_mask, const float* W, const float* a, const float* b,
                         const float* mean, const float* rstd, const float* gamma, float* da, float* db, float* dgamma,
                         float* dbeta, long long rows, int D, int rep, int inner, float p_drop, unsigned long long seed, void* ws,
                         size_t ws_bytes, void* stream);
/* Forward twin: z = drop(LN_3D([a, b, a*b])) and y = ReLU(z W^T + bias) in one pass over a and b (z is written once for the
 * backward's weight-gradient GEMM, never read back here).  z, mean, rstd are bit-identical to stage_cat3_layernorm_fwd; y and
 * relu_mask_out ([D/32][rows]) as stage_gemm_nt_mask on that z.  D == 128, rows >= 4096; ws: stage_cat3_ln_gemm_fwd_ws_bytes(). */
int stage_cat3_ln_gemm_fwd_supported(long long rows, int D, int rep, int inner);
size_t stage_cat3_ln_gemm_fwd_ws_bytes(void);
int stage_cat3_ln_gemm_fwd(const float* a, const float* b, const float* gamma, const float* beta, const float* W, const float* bias,
                           float* z, float* mean, float* rstd, float* y, unsigned* relu_mask_out, long long rows, int D, int rep,
                           int inner, float eps, float p_drop, unsigned long long seed, void* ws, size_t ws_bytes, void* stream);

/* ---- the same backward WITH the Linear's own gradients, no saved z (csrc/cat3_bwd_dw.hip) -----------------------------------------
 * (model/stage.py:381-385, :276-279, :107-113, :133-138.)  One persistent 8-wave workgroup per compute unit owns complete rows and
 * the whole (D, 3D) weight gradient in matrix-core accumulators: dz = (dy .* relu') W never exists as a tensor (as above) AND
 * dW[n][k] = sum_rows (dy .* relu')[row][n] z[row][k], dc[n] = sum_rows (dy .* relu')[row][n] are formed from z REBUILT in registers
 * (a, b, the saved row statistics, gamma / beta, the dropout stream) -- the forward passes z = NULL to stage_cat3_ln_gemm_fwd* and
 * no weight-gradient GEMM runs.  Arguments as stage_cat3_dx_ln_bwd plus beta (3D), dW (D, 3D), dc (D).                            */
int stage_cat3_bwd_dw_supported(long long rows, int D, int rep, int inner);
size_t stage_cat3_bwd_dw_ws_bytes(long long rows, int D, int rep, int inner);
int stage_cat3_bwd_dw(const float* dy, const unsigned* relu_mask, const float* W, const float* a, const float* b, const float* mean,
                      const float* rstd, const float* gamma, const float* beta, float* da, float* db, float* dgamma, float* dbeta,
                      float* dW, float* dc, long long rows, int D, int rep, int inner, float p_drop, unsigned long long seed, void* ws,
                      size_t ws_bytes, void* stream);

/* ---- K-groups: launch sequencing on the C side (one forward and one backward symbol per fused-op group) --
 * Each group runs the kernels above in the order tvqaplus_amd/ops.py would, as ONE call (csrc/groups.hip); fp32 storage.
 * Memory protocol: `arena` (stage_grp_*_arena_bytes) = what the forward keeps for the backward, carved deterministically;
 * `tmp` (stage_grp_*_bwd_tmp_bytes) = backward-only scratch; `flags` = host ints written by the forward and handed back to the
 * backward (which optional kernel paths ran; 16 ints are enough for every group); `params` / `grads` = HOST arrays of device
 * pointers in the order each group documents; `seeds` = host array, one entry per dropout site in forward order.
 * STAGE_ERR_SHAPE is returned before anything is launched when a group does not take the shape (use the per-op entry points). */
/* G1 input MLP (model/stage.py:350-362, :85-91, :98-104, :115-120; l2 != 0: F.normalize of :256 first):
 * [l2norm] -> LN(K0)+drop -> Linear(K0->H)+ReLU -> LN(H)+drop -> Linear(H->D)+ReLU -> LN(D).  x (M, K0) -> out (M, D).
 * params / grads: g0 b0 W1 c1 g1 b1 W2 c2 g2 b2; seeds[2].  The features receive no gradient.                            */
size_t stage_grp_input_mlp_arena_bytes(long long M, int K0, int H, int D, int l2);
int stage_grp_input_mlp_fwd(const float* x, const float* const* params, float* out, void* arena, size_t arena_bytes, int* flags,
                            long long M, int K0, int H, int D, int l2, float p_drop, const unsigned long long* seeds,
                            void* stream);
size_t stage_grp_input_mlp_bwd_tmp_bytes(long long M, int K0, int H, int D);
int stage_grp_input_mlp_bwd(const float* dout, const float* x, const float* const* params, float* const* grads,
                            const void* arena, size_t arena_bytes, const int* flags, void* tmp, size_t tmp_bytes, long long M,
                            int K0, int H, int D, int l2, float p_drop, const unsigned long long* seeds, void* stream);
/* G2 encoder block without self-attention (model/encoder.py:29-52, model/cnn.py:37-47, model/position_encoding.py:38-43):
 * x (M, L, D), pe (>= L, D).  pool_mask (M, L) != NULL: out = masked max over L of the block's output (model/stage.py:503),
 * (M, D); otherwise out (M, L, D).  params / grads: per conv i (ln_g ln_b dw_w dw_b pw_w pw_b), then final_g final_b;
 * seeds: one per even conv index.  dx may be NULL.                                                                        */
size_t stage_grp_encoder_arena_bytes(long long M, int L, int D, int n_conv, int pooled);
int stage_grp_encoder_fwd(const float* x, const float* pe, const float* pool_mask, const float* const* params, float* out,
                          void* arena, size_t arena_bytes, int* flags, long long M, int L, int D, int n_conv, int k, float p_drop,
                          const unsigned long long* seeds, void* stream);
size_t stage_grp_encoder_bwd_tmp_bytes(long long M, int L, int D, int k, int pooled);
int stage_grp_encoder_bwd(const float* dout, const float* x, const float* pool_mask, const float* const* params,
                          float* const* grads, float* dx, const void* arena, size_t arena_bytes, const int* flags, void* tmp,
                          size_t tmp_bytes, long long M, int L, int D, int n_conv, int k, float p_drop,
                          const unsigned long long* seeds, void* stream);
/* G3 QA <-> context attention + down-projection (model/stage.py:365-387): qa (N,NA,Lqa,D), ctx (N,Li,Lr,D) ->
 * mixed (N,NA,Li,Lqa,D), S_raw / S_norm (N,NA,Li,Lqa,Lr).  params / grads: ln_g ln_b W c; seeds[3] = context-side, region-side,
 * LayerNorm dropout.  Backward: dS_ext = gradient on S_raw or NULL; d_qa sums both uses of the QA embedding.              */
size_t stage_grp_qa_ctx_arena_bytes(int N, int NA, int Li, int Lqa, int D);
int stage_grp_qa_ctx_fwd(const float* qa, const float* ctx, const float* qa_mask, const float* ctx_mask,
                         const float* const* params, float* mixed, float* S_raw, float* S_norm, void* arena, size_t arena_bytes,
                         int* flags, int N, int NA, int Li, int Lqa, int Lr, int D, float scale, float p_drop,
                         const unsigned long long* seeds, void* stream);
size_t stage_grp_qa_ctx_bwd_tmp_bytes(int N, int NA, int Li, int Lqa, int Lr, int D);
int stage_grp_qa_ctx_bwd(const float* d_mixed, const float* dS_ext, const float* qa, const float* ctx, const float* ctx_mask,
                         const float* mixed, const float* S_norm, const float* const* params, float* const* grads, float* d_qa,
                         float* d_ctx, const void* arena, size_t arena_bytes, const int* flags, void* tmp, size_t tmp_bytes, int N,
                         int NA, int Li, int Lqa, int Lr, int D, float scale, float p_drop, const unsigned long long* seeds,
                         void* stream);
/* G4 two-stream fusion (model/stage.py:276-279, :106-113): s, v (U, D) -> out (U, D).  params / grads: ln3_g ln3_b W c ln_g ln_b;
 * seeds[1].                                                                                                                */
size_t stage_grp_concat_fc_arena_bytes(long long U, int D);
int stage_grp_concat_fc_fwd(const float* s, const float* v, const float* const* params, float* out, void* arena,
                            size_t arena_bytes, int* flags, long long U, int D, float p_drop, const unsigned long long* seeds,
                            void* stream);
size_t stage_grp_concat_fc_bwd_tmp_bytes(long long U, int D);
int stage_grp_concat_fc_bwd(const float* dout, const float* s, const float* v, const float* const* params, float* const* grads,
                            float* ds, float* dv, const void* arena, size_t arena_bytes, const int* flags, void* tmp,
                            size_t tmp_bytes, long long U, int D, float p_drop, const unsigned long long* seeds, void* stream);
/* G5 temporal head, layer 0 (model/stage.py:469-482, LinearWrapper :15-32): enc (R, D) -> first = enc + h (R, D), t_st, t_ed (R).
 * params / grads: lnp_g lnp_b Wp cp lns_g lns_b Ws cs lne_g lne_b We ce; seeds[3].  d_first may be NULL.                  */
size_t stage_grp_temporal_head_arena_bytes(long long R, int D);
int stage_grp_temporal_head_fwd(const float* enc, const float* const* params, float* first, float* t_st, float* t_ed,
                                void* arena, size_t arena_bytes, int* flags, long long R, int D, float p_drop,
                                const unsigned long long* seeds, void* stream);
size_t stage_grp_temporal_head_bwd_tmp_bytes(long long R, int D);
int stage_grp_temporal_head_bwd(const float* d_first, const float* d_st, const float* d_ed, const float* enc, const float* first,
                                const float* const* params, float* const* grads, float* d_enc, const void* arena,
                                size_t arena_bytes, const int* flags, void* tmp, size_t tmp_bytes, long long R, int D,
                                float p_drop, const unsigned long long* seeds, void* stream);

/* ---- head glue (model/stage.py:389-467, 484-555, 613-746): the small tensor algebra around the temporal scores, the span
 * proposals, the pooling, the classifier and the two auxiliary losses as a handful of kernels (csrc/groups.hip) ----------- */
/* t_scores (N, NA, Li, 2) = mask_logits(cat(t_st, t_ed), frame mask (N, Li))   (model/stage.py:515-521) and its backward */
int stage_tscores_fwd(const float* t_st, const float* t_ed, const float* tm, float* out, int N, int NA, int Li, void* stream);
int stage_tscores_bwd(const float* dout, const float* tm, float* d_st, float* d_ed, int N, int NA, int Li, void* stream);
/* span proposal of the ground-truth candidate (training, model/stage.py:408-418, model/model_utils.py:92-123): softmax over the
 * frames, arg max of the upper-triangular products (first maximal pair).  target / labels: int64 (N).  spans (6, N) float:
 * predicted start, end, confidence, label start, label end, answer index (so that ONE device-to-host copy carries everything
 * the proposal bookkeeping needs).  Li <= 2048.                                                                               */
int stage_gt_spans(const float* t_scores, const long long* target, const long long* lab_st, const long long* lab_ed,
                   float* spans, int N, int NA, int Li, void* stream);
/* temporal loss (model/stage.py:539-555) and d loss / d t_scores in one pass; scratch: N floats; cand_offset: global index of
 * local candidate 0 (candidate-sharded batches: examples whose ground truth is not local contribute nothing); na_total: the
 * model's candidate count over all ranks (a target outside [0, na_total) makes the loss NaN; <= 0: cand_offset + NA)        */
int stage_ts_loss(const float* t_scores, const long long* target, const long long* lab_st, const long long* lab_ed, float* loss,
                  float* grad, float* scratch, int N, int NA, int Li, int cand_offset, int na_total, void* stream);
/* The caller's loss line (main.py:55-60) in one launch: loss[0] = CE_sum(logits (P, C), targets (P)) * scale + att_w * att_loss[0] +
 * ts_w * t_loss[0] (att_loss / t_loss may be NULL), dlogits (P, C) = scale * (softmax - onehot) = the gradient of the first term.
 * scale = scale_dev[0] if scale_dev != NULL (multi-GPU: a device word), else scale_host.  Negative targets are ignored (ignore_index). */
int stage_train_loss(const float* logits, const long long* targets, const float* att_loss, const float* t_loss, const float* scale_dev,
                     float scale_host, float att_w, float ts_w, float* loss, float* dlogits, int P, int C, void* stream);
/* supervised attention loss (model/stage.py:738-745) over M (positive, negative) pairs: flat = 2M int64 indices into scores
 * (positives, then negatives); hinge != 0: max(0, margin + s_neg - s_pos), else log1p(exp(alpha (s_neg - s_pos))).  coef (M) is
 * kept for the backward, which zero-fills dS (n_scores floats) and scatters gout[0] * coef into it.                          */
int stage_att_loss_fwd(const float* scores, const long long* flat, long long M, int hinge, float alpha, float margin, float* coef,
                       float* loss, void* stream);
int stage_att_loss_bwd(const long long* flat, const float* coef, const float* gout, long long M, float* dS, long long n_scores,
                       void* stream);
/* G6 proposal pooling + answer classifier (model/stage.py:420-467, 526-536): first (N*NA, Li, D), mask (N*NA, Li), glob / idx_g =
 * stage_masked_max_fwd(first, mask) computed ahead; meta (device int32) = src[P] | win[2P] | inv[2N] (proposal -> example, frame
 * window [st, ed), example -> its <= 2 proposals or -1).  logits (P*NA).  params / grads: ln_g ln_b W c (2D wide); seeds[1].
 * d_first (N*NA, Li, D) receives both pooling paths.                                                                          */
size_t stage_grp_pool_cls_arena_bytes(long long P, int NA, int D);
int stage_grp_pool_cls_fwd(const float* first, const float* mask, const float* glob, const int* meta, const float* const* params,
                           float* logits, void* arena, size_t arena_bytes, int N, int NA, int Li, int D, long long P, float p_drop,
                           const unsigned long long* seeds, void* stream);
size_t stage_grp_pool_cls_bwd_tmp_bytes(long long P, int NA, int D);
int stage_grp_pool_cls_bwd(const float* d_logits, const float* mask, const int* idx_g, const int* meta, const float* const* params,
                           float* const* grads, float* d_first, const void* arena, size_t arena_bytes, void* tmp, size_t tmp_bytes,
                           int N, int NA, int Li, int D, long long P, float p_drop, const unsigned long long* seeds, void* stream);


/* ==== Ragged token rows: live-range execution of the (N, 5, Li, Lqa, .) kernels ===================================================
 * The reference computes every padded row of the (N,5,Li,Lqa,D) tensors (model/stage.py:365-387 qa_ctx_attention, :276-279 concat_fc,
 * :484-505 classifier head; no masking inside LayerNorm / Linear / the convolutions of model/encoder.py:35-52, model/cnn.py:42-47).
 * Only this reaches an output or a gradient: frames whose statement mask is not all zero (the others pool to the constant -1e10 at
 * model/stage.py:503 and get the gradient dout * mask = 0), and of those the words below Lc = min(Lqa, last valid word + 1 + halo),
 * halo = (convolution layers of the classifier encoder) * (kernel_size / 2) -- the receptive field through which padded words leak
 * into valid ones.  The entry points below run on exactly those rows ("compact rows": [group g = (n, a)][live frame][word < Lc(g)]);
 * the attention output A and its gradient keep all Lqa words of a live frame ("frame-compact rows":
 * [sequence = first(n) + a * slots(n) + slot][word], slots(n) = live frames of example n + one dump slot that dead frames use).
 * Tables (device int32, built by tvqaplus_amd/ragged.py from the masks):
 *   fmap    [N*Li] slot of each frame (< 0 dead) | [N] slots(n) | [N] first(n)
 *   gdesc   (N*NA, 4)  first compact row, Lc, slots(n), first frame-compact sequence of the group
 *   seq     (S, 4)     first compact row, length, group, dense output row g*Li + i            one per (group, live frame)
 *   seqfc   (S)        frame-compact sequence of each entry of seq
 *   rowinfo (U, 4)     QA row g*Lqa + w, frame-compact row, dense output row, w                 (stage_rag_rowinfo)                 */
int stage_rag_rowinfo(const int* seq, const int* seqfc, long long S, int Lqa, int* rowinfo, void* stream);
/* Ragged CONTEXT rows (the (N, Li, Lw|Lr, .) streams in front of the attention: model/stage.py:235-270): frame f keeps its valid words /
 * regions plus the halo of the input encoder's convolutions, cq[f] = (first compact row, rows); src_rows[r] = row of the padded feature
 * tensor that compact row r reads (stage_rag_ctx_rows).  The *_gather entry points are LayerNorm / L2 normalisation reading those rows
 * in place (model/stage.py:85-91, 98-104, 256); everything behind them runs on the compact rows. */
int stage_rag_ctx_rows(const int* cq, long long frames, int L, int* src_rows, void* stream);
int stage_layernorm_gather_fwd(const float* x, const int* gather, const float* gamma, const float* beta, float* y, float* mean,
                               float* rstd, long long rows, int K, float eps, float p_drop, unsigned long long seed, void* stream);
int stage_layernorm_gather_bwd(const float* dy, const float* x, const int* gather, const float* mean, const float* rstd,
                               const float* gamma, float* dgamma, float* dbeta, long long rows, int K, float p_drop,
                               unsigned long long seed, void* ws, size_t ws_bytes, void* stream);
int stage_l2norm_gather_fwd(const float* x, const int* gather, float* y, long long rows, int K, float eps, void* stream);
int stage_grp_input_mlp_rag_fwd(const float* x, const int* src_rows, const float* const* P, float* out, void* arena,
                                size_t arena_bytes, int* flags, long long M, int K0, int H, int D, int l2, float p,
                                const unsigned long long* seeds, void* stream);
int stage_grp_input_mlp_rag_bwd(const float* dout, const float* x, const int* src_rows, const float* const* P, float* const* G,
                                const void* arena, size_t arena_bytes, const int* flags, void* tmp, size_t tmp_bytes, long long M,
                                int K0, int H, int D, int l2, float p, const unsigned long long* seeds, void* stream);
int stage_rag_fill_pooled(float* out, int* argmax, long long rows, int D, void* stream);      /* -1e10 / 0: model/stage.py:503 on an all-masked group */
int stage_rag_zero_dump(float* A_fc, const int* fmap, int N, int NA, int Li, int Lqa, int D, void* stream);
/* K1 with a frame-compact A / dA (model/context_query_attention.py:35-101; D == 128, the fast kernels only) */
/* cq (may be NULL): Q / Qn / dQraw / dQn hold compact context rows (above); q_mask stays dense (N, Li, Lr) */
int stage_str_attn_fwd_fc(const float* Cn, const float* Q, const float* c_mask, const float* q_mask, float* A_fc, float* S_raw,
                          float* S_norm, const int* fmap, const int* cq, int N, int NA, int Li, int Lqa, int Lr, int D, float scale,
                          float p_drop, unsigned long long seed, void* stream);
int stage_str_attn_bwd_fused_fc(const float* dA_fc, const float* dS_raw_ext, const float* Cn, const float* Q, const float* Qn,
                                const float* S_norm, const float* q_mask, float* dQraw, float* dQn, float* dCn, const int* fmap,
                                const int* cq, int N, int NA, int Li, int Lqa, int Lr, int D, float scale, void* ws, size_t ws_bytes,
                                void* stream);
/* [a, b, a*b] LayerNorm -> Linear -> ReLU (model/stage.py:381-385) on compact rows: a = QA rows, b = frame-compact rows */
int stage_cat3_ln_gemm_fwd_rag_supported(long long rows, long long a_rows, long long b_rows, int D);
int stage_cat3_ln_gemm_fwd_rag(const float* a, const float* b, const float* gamma, const float* beta, const float* W,
                               const float* bias, float* z, float* mean, float* rstd, float* y, unsigned* relu_mask_out,
                               const int* rowinfo, long long rows, long long a_rows, long long b_rows, int D, float eps,
                               float p_drop, unsigned long long seed, void* ws, size_t ws_bytes, void* stream);
int stage_cat3_dx_ln_bwd_rag_supported(long long rows, long long fc_rows, int D, int groups, int max_frames, int Lqa);
size_t stage_cat3_dx_ln_bwd_rag_ws_bytes(int groups, int max_frames, int Lqa);
/* wtab (may be NULL): balanced work table for stage_cat3_rag_work_groups() persistent workgroups (tvqaplus_amd/ragged.py:
 * RaggedTables.work_table -- [first segment of every workgroup, W + 1 ints padded to 4][(first slab, slabs) per group][segments
 * (group, first frame, end frame, workgroup)]): equal tile counts per workgroup; NULL: one workgroup per (group, chunk of max_frames) */
int stage_cat3_rag_work_groups(void);
int stage_cat3_dx_ln_bwd_rag(const float* dy, const unsigned* relu_mask, const float* W, const float* a, const float* b_fc,
                             const float* mean, const float* rstd, const float* gamma, float* da, float* db_fc, float* dgamma,
                             float* dbeta, const int* gdesc, const int* wtab, long long rows, long long fc_rows, int D, int groups,
                             int max_frames, int Lqa, float p_drop, unsigned long long seed, void* ws, size_t ws_bytes, void* stream);
/* the ragged backward with the Linear's gradients inside (csrc/cat3_bwd_dw.hip; see stage_cat3_bwd_dw): wtab is REQUIRED and built
 * for stage_cat3_bwd_dw_rag_work_groups() workgroups (= what stage_cat3_rag_work_groups() returns while this path is enabled) */
int stage_cat3_bwd_dw_rag_supported(long long rows, long long fc_rows, int D, int groups, int max_frames, int Lqa);
int stage_cat3_bwd_dw_rag_work_groups(void);
size_t stage_cat3_bwd_dw_rag_ws_bytes(int groups, int Lqa);
int stage_cat3_bwd_dw_rag(const float* dy, const unsigned* relu_mask, const float* W, const float* a, const float* b_fc,
                          const float* mean, const float* rstd, const float* gamma, const float* beta, float* da, float* db_fc,
                          float* dgamma, float* dbeta, float* dW, float* dc, const int* gdesc, const int* wtab, long long rows,
                          long long fc_rows, int D, int groups, int max_frames, int Lqa, float p_drop, unsigned long long seed,
                          void* ws, size_t ws_bytes, void* stream);
/* encoder block pieces on ragged sequences (model/encoder.py:35-52; model/stage.py:503 for the pooled LayerNorm) */
int stage_ln_dwconv_rag_fwd(const float* x, const float* res, const float* pe, float* sum_out, const float* gamma,
                            const float* beta, const float* w, const float* bias, float* h, float* mean, float* rstd,
                            const int* seq, long long S, int Lmax, int D, int k, float eps, float p_drop,
                            unsigned long long seed, void* stream);
int stage_ln_dwconv_rag_bwd(const float* dh, const float* xin, const float* mean, const float* rstd, const float* gamma,
                            const float* beta, const float* w, float* dx, const float* dx_add, float* dgamma, float* dbeta,
                            float* dw, float* db, const int* seq, long long S, int Lmax, int D, int k, float p_drop,
                            unsigned long long seed, void* ws, size_t ws_bytes, void* stream);
int stage_ln_masked_max_rag_fwd(const float* x, const float* res, float* sum_out, const float* gamma, const float* beta,
                                const float* qmask, float* out, int* argmax, float* mean, float* rstd, const int* seq,
                                long long S, int Lq, int K, float eps, void* stream);
int stage_ln_masked_max_rag_bwd(const float* dout, const int* argmax, const float* qmask, const float* xin, const float* mean,
                                const float* rstd, const float* gamma, float* dx, float* dgamma, float* dbeta,
                                const int* rowinfo, long long rows, int K, void* ws, size_t ws_bytes, void* stream);
/* K-groups on ragged rows (csrc/groups.hip "RAGGED TOKEN ROWS"): T = host array of the device tables fmap, gdesc, seq, rowinfo, cq
 * (cq NULL: dense context stream; Uc = rows of ctx / d_ctx) and -- the qa_ctx group, 6 entries -- wtab (the balanced work table of
 * stage_cat3_dx_ln_bwd_rag, or NULL).  The encoder group pools when qa_mask != NULL, else returns (U, D). */
int stage_grp_qa_ctx_rag_supported(int N, int NA, int Li, int Lqa, int Lr, int D, long long U, long long Fc);
size_t stage_grp_qa_ctx_rag_arena_bytes(int N, int NA, int Lqa, int D, long long Ucap, long long Fc);
size_t stage_grp_qa_ctx_rag_bwd_tmp_bytes(int N, int NA, int Li, int Lqa, int Lr, int D, long long Ucap, long long Uc);
int stage_grp_qa_ctx_rag_fwd(const float* qa, const float* ctx, const float* qa_mask, const float* ctx_mask,
                             const float* const* P, float* mixed, float* S_raw, float* S_norm, const int* const* T, void* arena,
                             size_t arena_bytes, int* flags, int N, int NA, int Li, int Lqa, int Lr, int D, long long U,
                             long long Ucap, long long Fc, long long Uc, float scale, float p, const unsigned long long* seeds,
                             void* stream);
int stage_grp_qa_ctx_rag_bwd(const float* d_mixed, const float* dS_ext, const float* qa, const float* ctx, const float* ctx_mask,
                             const float* mixed, const float* S_norm, const float* const* P, float* const* G, float* d_qa,
                             float* d_ctx, const int* const* T, void* arena, size_t arena_bytes, const int* flags, void* tmp,
                             size_t tmp_bytes, int N, int NA, int Li, int Lqa, int Lr, int D, long long U, long long Ucap,
                             long long Fc, long long Uc, float scale, float p, const unsigned long long* seeds, void* stream);
size_t stage_grp_encoder_rag_arena_bytes(long long Ucap, long long Rd, int D, int n_conv);
size_t stage_grp_encoder_rag_bwd_tmp_bytes(long long Ucap, int D, int k);
int stage_grp_encoder_rag_fwd(const float* x, const float* pe, const float* qa_mask, const float* const* P, float* out,
                              const int* const* T, void* arena, size_t arena_bytes, int* flags, long long U, long long Ucap,
                              long long S, long long Rd, int Lqa, int D, int n_conv, int k, float p,
                              const unsigned long long* seeds, void* stream);
int stage_grp_encoder_rag_bwd(const float* dout, const float* qa_mask, const float* const* P, float* const* Gr, float* dx,
                              const int* const* T, const void* arena, size_t arena_bytes, const int* flags, void* tmp,
                              size_t tmp_bytes, long long U, long long Ucap, long long S, long long Rd, int Lqa, int D, int n_conv,
                              int k, float p, const unsigned long long* seeds, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STAGE_HIP_H */
