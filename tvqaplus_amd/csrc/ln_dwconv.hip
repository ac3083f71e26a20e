// Fused  LayerNorm (+ residual add, + inverted dropout)  ->  depthwise Conv1d  of the encoder blocks
// (model/encoder.py:37-44: x = x + conv(drop(LN(x))) with conv = depthwise then pointwise; model/cnn.py:37-47).
// The LayerNorm output only ever feeds the depthwise conv, so it is never written: forward reads x (+ res) and writes the
// exported sum and the conv output; backward recomputes the normalised rows from the saved sum / mean / rstd.  Per conv
// layer of the large (960000 x 128) tensors this removes two full passes in the forward and three in the backward.
//
// Work decomposition = the sliding-window conv kernels (encoder.hip): a thread owns one float4 column of one chunk of <= 48
// positions of one sequence and walks it with the k taps in registers; the D/4 lanes of a row (a power of two <= 64, so a
// row group never straddles a wave) compute the row statistics with cross-lane sums, every row exactly as the
// LayerNorm kernels of rowops.hip do (same expressions, same summation order, same dropout counter).
#include <stdlib.h>
#include "common.h"
#include "../../include/stage_hip.h"

#define LD_GRID_CAP 16384
#define LD_PART_CAP 512

// (up to five taps the kernel fits 128 registers -- four waves per SIMD instead of three: 398 / 357 -> 371 / 341 us at the classifier
// encoder; with seven taps the cap spills 22-26 registers and doubles the run time)
// RAG: ragged sequences (include/stage_hip.h "ragged token rows"): sequence m = seq[m] = (first row, length <= L, ., .) of a compact row
// space; rows outside [0, length) are the conv's zero padding, exactly as the ends of a dense sequence are.  res_period > 0 then means
// "res is a (L, D) position table": row ll of a sequence takes res[ll].
template <int KT, bool DROP, typename T = float, bool RAG = false>
__global__ __launch_bounds__(256, (KT <= 5 ? 4 : 1)) void ln_dwconv_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                            int res_period, T* __restrict__ sum_out,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ w, const float* __restrict__ bias,
                                                            T* __restrict__ h, float* __restrict__ mean,
                                                            float* __restrict__ rstd, long M, int L, int D, float eps,
                                                            uint64_t seed, uint32_t th, float inv_keep, int clen,
                                                            const int4* __restrict__ seq = nullptr) {
    constexpr int pad = KT / 2;
    const int D4 = D >> 2, rpi = blockDim.x / D4;
    const int q = threadIdx.x % D4, rsub = threadIdx.x / D4;
    const float invK = 1.0f / (float)D;
    const float4 gm = ld4(gamma + 4 * q), bt = ld4(beta + 4 * q), bq = ld4(bias + 4 * q);
    float4 wt[KT];
#pragma unroll
    for (int t = 0; t < KT; t++)
        wt[t] = make_float4(w[(4 * q + 0) * KT + t], w[(4 * q + 1) * KT + t], w[(4 * q + 2) * KT + t], w[(4 * q + 3) * KT + t]);
    const int chunks = (L + clen - 1) / clen;
    const long items = M * chunks;
    for (long it = (long)blockIdx.x * rpi + rsub; it < items; it += (long)gridDim.x * rpi) {
        const long m = it / chunks;
        long sbase = m * L;                  // first row of the sequence
        int Ls = L;                          // its length
        if (RAG) {
            const int4 sq = seq[m];
            sbase = sq.x;
            Ls = sq.y;
        }
        const int l0 = (int)(it % chunks) * clen, l1 = min(Ls, l0 + clen);
        // normalised (and dropped) row ll of sequence m is 0 outside the sequence.  Rows of this chunk also export the
        // sum and the statistics (every row is normalised by exactly one chunk as its own row, halo rows are recomputed).
        // Rows enter the window in batches: all loads of a batch first (clamped addresses), then the statistics (two
        // cross-lane reductions per row, independent chains), then the stores of the chunk's own rows -- a store or a
        // guarded load in the middle of the chain makes the compiler wait for every load before it.
        auto load_row = [&](int ll, float4& v) {
            const int lc = min(max(ll, 0), max(Ls - 1, 0));
            const long row = sbase + lc;
            v = ldv4(x + row * D + 4 * q);
            if (res) v = f4add(v, ldv4(res + (res_period > 0 ? (RAG ? (long)lc : (long)((unsigned long)row % (unsigned)res_period)) : row) * D + 4 * q));
        };
        auto norm_row = [&](int ll, const float4& v, float& mu, float& rs) -> float4 {
            const bool inside = ll >= 0 && ll < Ls;
            const long row = sbase + min(max(ll, 0), max(Ls - 1, 0));
            mu = group_sum(f4hsum(v), D4) * invK;
            const float4 d = make_float4(v.x - mu, v.y - mu, v.z - mu, v.w - mu);
            rs = 1.0f / sqrtf(group_sum(f4hsum(f4mul(d, d)), D4) * invK + eps);
            float4 o;
            o.x = (v.x - mu) * rs * gm.x + bt.x;
            o.y = (v.y - mu) * rs * gm.y + bt.y;
            o.z = (v.z - mu) * rs * gm.z + bt.z;
            o.w = (v.w - mu) * rs * gm.w + bt.w;
            if (DROP) o = f4mul(o, drop4(seed, (uint64_t)row * D4 + q, th, inv_keep));
            return inside ? o : f4zero();
        };
        auto export_row = [&](int ll, const float4& v, float mu, float rs) {   // the chunk's own rows only
            if (ll >= l0 && ll < l1) {
                const long row = sbase + ll;
                if (sum_out) stv4(sum_out + row * D + 4 * q, v);
                if (q == 0) {
                    mean[row] = mu;
                    rstd[row] = rs;
                }
            }
        };
        float4 win[KT];   // win[t] = y[l + t - pad] for the current l
        {
            float4 pv[KT > 1 ? KT - 1 : 1];
            float pm[KT > 1 ? KT - 1 : 1], pr[KT > 1 ? KT - 1 : 1];
#pragma unroll
            for (int t = 0; t < KT - 1; t++) load_row(l0 + t - pad, pv[t]);
#pragma unroll
            for (int t = 0; t < KT - 1; t++) win[t + 1] = norm_row(l0 + t - pad, pv[t], pm[t], pr[t]);
#pragma unroll
            for (int t = 0; t < KT - 1; t++) export_row(l0 + t - pad, pv[t], pm[t], pr[t]);
        }
        for (int l = l0; l < l1; l += 4) {
            float4 nv[4], nr[4];
            float nm[4], ns[4];
#pragma unroll
            for (int u = 0; u < 4; u++) load_row(l + u + pad, nv[u]);
#pragma unroll
            for (int u = 0; u < 4; u++) nr[u] = norm_row(l + u + pad, nv[u], nm[u], ns[u]);
#pragma unroll
            for (int u = 0; u < 4; u++) export_row(l + u + pad, nv[u], nm[u], ns[u]);
#pragma unroll
            for (int u = 0; u < 4; u++) {
#pragma unroll
                for (int t = 0; t < KT - 1; t++) win[t] = win[t + 1];
                win[KT - 1] = nr[u];
                float4 acc = bq;
#pragma unroll
                for (int t = 0; t < KT; t++) acc = f4add(acc, f4mul(win[t], wt[t]));
                if (l + u < l1) stv4s(h + ((sbase + l + u) * D + 4 * q), acc);
            }
        }
    }
}

// dx = LN-backward(conv-backward(dh)) + dx_add ; partials of dw/db (conv) and dgamma/dbeta (LayerNorm) per workgroup.
// (register caps for more waves per SIMD were measured here too: 168 registers spill 80-100 and run 4x slower)
template <int KT, bool DROP, typename T = float, bool RAG = false>
__global__ __launch_bounds__(256) void ln_dwconv_bwd_kernel(const T* __restrict__ dh, const T* __restrict__ xin,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ w, T* __restrict__ dx,
                                                            const T* __restrict__ dx_add, float* __restrict__ part_conv,
                                                            float* __restrict__ part_ln, long M, int L, int D,
                                                            uint64_t seed, uint32_t th, float inv_keep, int clen,
                                                            const int4* __restrict__ seq = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // reduction scratch [rpi][KT + 1][D]
    constexpr int pad = KT / 2;
    const int D4 = D >> 2, rpi = blockDim.x / D4;
    const int q = threadIdx.x % D4, rsub = threadIdx.x / D4;
    const float invK = 1.0f / (float)D;
    const float4 gm = ld4(gamma + 4 * q), bt = ld4(beta + 4 * q);
    float4 wt[KT], aw[KT], abc = f4zero(), ag = f4zero(), abl = f4zero();
#pragma unroll
    for (int t = 0; t < KT; t++) {
        wt[t] = make_float4(w[(4 * q + 0) * KT + t], w[(4 * q + 1) * KT + t], w[(4 * q + 2) * KT + t], w[(4 * q + 3) * KT + t]);
        aw[t] = f4zero();
    }
    const int chunks = (L + clen - 1) / clen;
    const long items = M * chunks;
    for (long it = (long)blockIdx.x * rpi + rsub; it < items; it += (long)gridDim.x * rpi) {
        const long m = it / chunks;
        long sbase = m * L;
        int Ls = L;
        if (RAG) {
            const int4 sq = seq[m];
            sbase = sq.x;
            Ls = sq.y;
        }
        const int l0 = (int)(it % chunks) * clen, l1 = min(Ls, l0 + clen);
        const int Lm1 = max(Ls - 1, 0);
        // recomputed LayerNorm output (after dropout) of row ll, 0 outside the sequence
        auto y_row = [&](int ll) -> float4 {
            const bool inside = ll >= 0 && ll < Ls;
            const long row = sbase + min(max(ll, 0), Lm1);
            const float4 v = ldv4(xin + row * D + 4 * q);
            const float mu = mean[row], rs = rstd[row];
            float4 o;
            o.x = (v.x - mu) * rs * gm.x + bt.x;
            o.y = (v.y - mu) * rs * gm.y + bt.y;
            o.z = (v.z - mu) * rs * gm.z + bt.z;
            o.w = (v.w - mu) * rs * gm.w + bt.w;
            if (DROP) o = f4mul(o, drop4(seed, (uint64_t)row * D4 + q, th, inv_keep));
            return inside ? o : f4zero();
        };
        auto dh_row = [&](int ll) -> float4 {
            const bool inside = ll >= 0 && ll < Ls;
            const float4 v = ldv4(dh + (sbase + min(max(ll, 0), Lm1)) * D + 4 * q);
            return inside ? v : f4zero();
        };
        float4 wy[KT], wo[KT];   // wy[t] = y[l + t - pad], wo[t] = dh[l + t - pad]
#pragma unroll
        for (int t = 0; t < KT - 1; t++) {
            wy[t + 1] = y_row(l0 + t - pad);
            wo[t + 1] = dh_row(l0 + t - pad);
        }
        for (int l = l0; l < l1; l += 4) {
            // four positions per iteration: all loads of the iteration (entering rows of both windows, the centre rows
            // with their statistics and the incoming sum gradient) are issued up front
            float4 ny[4], no[4], cv[4], ca[4];
            float cmu[4], crs[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                ny[u] = y_row(l + u + pad);
                no[u] = dh_row(l + u + pad);
                const long row = sbase + min(l + u, Lm1);
                cv[u] = ldv4(xin + row * D + 4 * q);
                cmu[u] = mean[row];
                crs[u] = rstd[row];
                if (dx && dx_add) ca[u] = ldv4(dx_add + row * D + 4 * q);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
#pragma unroll
                for (int t = 0; t < KT - 1; t++) { wy[t] = wy[t + 1]; wo[t] = wo[t + 1]; }
                wy[KT - 1] = ny[u];
                wo[KT - 1] = no[u];
                const bool live = l + u < l1;            // positions past the chunk end contribute nothing
                // depthwise conv backward at position l + u
                const float4 go = live ? wo[pad] : f4zero();
                abc = f4add(abc, go);
                float4 din = f4zero();
#pragma unroll
                for (int t = 0; t < KT; t++) {
                    aw[t] = f4add(aw[t], f4mul(go, wy[t]));                  // dw[t] += dh[l] * y[l + t - pad]
                    din = f4add(din, f4mul(wo[KT - 1 - t], wt[t]));           // dy[l] += dh[l - t + pad] * w[t]
                }
                // LayerNorm backward of the row (same expressions as rowops.hip)
                const long row = sbase + min(l + u, Lm1);
                const float mu = cmu[u], rs = crs[u];
                float4 dd = live ? din : f4zero();
                if (DROP) dd = f4mul(dd, drop4(seed, (uint64_t)row * D4 + q, th, inv_keep));
                const float4 v = cv[u];
                const float4 xh = make_float4((v.x - mu) * rs, (v.y - mu) * rs, (v.z - mu) * rs, (v.w - mu) * rs);
                const float4 g = f4mul(dd, gm);
                const float s1 = group_sum(f4hsum(g), D4) * invK;
                const float s2 = group_sum(f4hsum(f4mul(g, xh)), D4) * invK;
                ag = f4add(ag, f4mul(dd, xh));
                abl = f4add(abl, dd);
                if (dx && live) {
                    float4 o;
                    o.x = rs * (g.x - s1 - xh.x * s2);
                    o.y = rs * (g.y - s1 - xh.y * s2);
                    o.z = rs * (g.z - s1 - xh.z * s2);
                    o.w = rs * (g.w - s1 - xh.w * s2);
                    if (dx_add) o = f4add(o, ca[u]);
                    stv4(dx + row * D + 4 * q, o);
                }
            }
        }
    }
    // block reductions (fixed order): conv partials [KT+1][D], then LayerNorm partials [2][D]
#pragma unroll
    for (int t = 0; t < KT; t++) st4(&sm[((size_t)rsub * (KT + 1) + t) * D + 4 * q], aw[t]);
    st4(&sm[((size_t)rsub * (KT + 1) + KT) * D + 4 * q], abc);
    __syncthreads();
    const int C = (KT + 1) * D;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.f;
        for (int r = 0; r < rpi; r++) s += sm[(size_t)r * C + c];
        part_conv[(size_t)blockIdx.x * C + c] = s;
    }
    __syncthreads();
    st4(&sm[((size_t)rsub * 2 + 0) * D + 4 * q], ag);
    st4(&sm[((size_t)rsub * 2 + 1) * D + 4 * q], abl);
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * D; c += blockDim.x) {
        float s = 0.f;
        for (int r = 0; r < rpi; r++) s += sm[(size_t)r * 2 * D + c];
        part_ln[(size_t)blockIdx.x * 2 * D + c] = s;
    }
}

static bool ld_shape_ok(int D, int k) {
    const int D4 = D / 4;
    return D % 4 == 0 && D4 >= 4 && D4 <= 64 && (D4 & (D4 - 1)) == 0 && k >= 1 && k <= 9 && (k & 1) == 1;
}

template <typename T>
static int ln_dwconv_fwd_t(const T* x, const T* res, int res_period, T* sum_out, const float* gamma, const float* beta,
                           const float* w, const float* bias, T* h, float* mean, float* rstd, long long M, int L, int D, int k,
                           float eps, float p_drop, unsigned long long seed, void* stream) {
    if (M <= 0 || L <= 0) return 0;
    if (!ld_shape_ok(D, k)) return STAGE_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int rpi = 256 / (D / 4);
    const int clen = stage_chunk_len(L);
    const long items = (long)M * ((L + clen - 1) / clen);
    const int grid = stage_grid_for(items, rpi, LD_GRID_CAP);
    const bool dr = p_drop > 0.f;
    const uint64_t sd = dr ? (uint64_t)seed : 0;
    const uint32_t th = dr ? drop_thresh16(p_drop) : 0u;
    const float ik = dr ? 1.0f / (1.0f - p_drop) : 1.0f;
#define LD_FWD(KV, DR)                                                                                                   \
    hipLaunchKernelGGL((ln_dwconv_fwd_kernel<KV, DR, T>), dim3(grid), dim3(256), 0, st, x, res, res_period, sum_out, gamma,  \
                       beta, w, bias, h, mean, rstd, (long)M, L, D, eps, sd, th, ik, clen)
    switch (k) {
        case 1: if (dr) LD_FWD(1, true); else LD_FWD(1, false); break;
        case 3: if (dr) LD_FWD(3, true); else LD_FWD(3, false); break;
        case 5: if (dr) LD_FWD(5, true); else LD_FWD(5, false); break;
        case 7: if (dr) LD_FWD(7, true); else LD_FWD(7, false); break;
        default: if (dr) LD_FWD(9, true); else LD_FWD(9, false); break;
    }
#undef LD_FWD
    STAGE_LAUNCH_CHECK();
    return 0;
}

extern "C" int stage_ln_dwconv_fwd(const float* x, const float* res, int res_period, float* sum_out, const float* gamma,
                                   const float* beta, const float* w, const float* bias, float* h, float* mean,
                                   float* rstd, long long M, int L, int D, int k, float eps, float p_drop,
                                   unsigned long long seed, void* stream) {
    return ln_dwconv_fwd_t<float>(x, res, res_period, sum_out, gamma, beta, w, bias, h, mean, rstd, M, L, D, k, eps, p_drop, seed,
                                  stream);
}
extern "C" int stage_ln_dwconv_fwd_bf16(const void* x, const void* res, int res_period, void* sum_out, const float* gamma,
                                        const float* beta, const float* w, const float* bias, void* h, float* mean,
                                        float* rstd, long long M, int L, int D, int k, float eps, float p_drop,
                                        unsigned long long seed, void* stream) {
    typedef stage_bf16 B;
    return ln_dwconv_fwd_t<B>((const B*)x, (const B*)res, res_period, (B*)sum_out, gamma, beta, w, bias, (B*)h, mean, rstd, M, L, D,
                              k, eps, p_drop, seed, stream);
}

// Ragged sequences: S sequences seq[s] = (first row, length <= Lmax, ., .) in a compact row space; pe != NULL: the (Lmax, D) position
// table added to row ll of every sequence (the encoder's first LayerNorm), else res (compact rows, may be NULL) is added row by row.
extern "C" int stage_ln_dwconv_rag_fwd(const float* x, const float* res, const float* pe, float* sum_out, const float* gamma,
                                       const float* beta, const float* w, const float* bias, float* h, float* mean, float* rstd,
                                       const int* seq, long long S, int Lmax, int D, int k, float eps, float p_drop,
                                       unsigned long long seed, void* stream) {
    if (S <= 0 || Lmax <= 0) return 0;
    if (!ld_shape_ok(D, k) || (pe && res)) return STAGE_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int rpi = 256 / (D / 4);
    const int clen = stage_chunk_len(Lmax);
    const long items = (long)S * ((Lmax + clen - 1) / clen);
    const int grid = stage_grid_for(items, rpi, LD_GRID_CAP);
    const bool dr = p_drop > 0.f;
    const uint64_t sd = dr ? (uint64_t)seed : 0;
    const uint32_t th = dr ? drop_thresh16(p_drop) : 0u;
    const float ik = dr ? 1.0f / (1.0f - p_drop) : 1.0f;
    const float* r = pe ? pe : res;
    const int period = pe ? Lmax : 0;
#define LD_FWD(KV, DR)                                                                                                       \
    hipLaunchKernelGGL((ln_dwconv_fwd_kernel<KV, DR, float, true>), dim3(grid), dim3(256), 0, st, x, r, period, sum_out, gamma,  \
                       beta, w, bias, h, mean, rstd, (long)S, Lmax, D, eps, sd, th, ik, clen, (const int4*)seq)
    switch (k) {
        case 1: if (dr) LD_FWD(1, true); else LD_FWD(1, false); break;
        case 3: if (dr) LD_FWD(3, true); else LD_FWD(3, false); break;
        case 5: if (dr) LD_FWD(5, true); else LD_FWD(5, false); break;
        case 7: if (dr) LD_FWD(7, true); else LD_FWD(7, false); break;
        default: if (dr) LD_FWD(9, true); else LD_FWD(9, false); break;
    }
#undef LD_FWD
    STAGE_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t stage_ln_dwconv_bwd_ws_bytes(int D, int k) { return (size_t)LD_PART_CAP * (k + 3) * D * sizeof(float); }

extern "C" int stage_ln_dwconv_rag_bwd(const float* dh, const float* xin, const float* mean, const float* rstd, const float* gamma,
                                       const float* beta, const float* w, float* dx, const float* dx_add, float* dgamma,
                                       float* dbeta, float* dw, float* db, const int* seq, long long S, int Lmax, int D, int k,
                                       float p_drop, unsigned long long seed, void* ws, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!ld_shape_ok(D, k)) return STAGE_ERR_SHAPE;
    if (ws_bytes < stage_ln_dwconv_bwd_ws_bytes(D, k)) return STAGE_ERR_WORKSPACE;
    if (S <= 0 || Lmax <= 0) {
        (void)hipMemsetAsync(dw, 0, sizeof(float) * D * k, st);
        (void)hipMemsetAsync(db, 0, sizeof(float) * D, st);
        (void)hipMemsetAsync(dgamma, 0, sizeof(float) * D, st);
        (void)hipMemsetAsync(dbeta, 0, sizeof(float) * D, st);
        return 0;
    }
    const int rpi = 256 / (D / 4);
    const int clen = stage_chunk_len(Lmax);
    const long items = (long)S * ((Lmax + clen - 1) / clen);
    const int grid = stage_grid_for(items, rpi, LD_PART_CAP);
    float* part_conv = (float*)ws;
    float* part_ln = part_conv + (size_t)LD_PART_CAP * (k + 1) * D;
    const size_t lds = (size_t)rpi * (k + 1) * D * sizeof(float);
    const bool dr = p_drop > 0.f;
    const uint64_t sd = dr ? (uint64_t)seed : 0;
    const uint32_t th = dr ? drop_thresh16(p_drop) : 0u;
    const float ik = dr ? 1.0f / (1.0f - p_drop) : 1.0f;
#define LD_BWD(KV, DR)                                                                                                              \
    hipLaunchKernelGGL((ln_dwconv_bwd_kernel<KV, DR, float, true>), dim3(grid), dim3(256), lds, st, dh, xin, mean, rstd, gamma, beta, w, \
                       dx, dx_add, part_conv, part_ln, (long)S, Lmax, D, sd, th, ik, clen, (const int4*)seq)
    switch (k) {
        case 1: if (dr) LD_BWD(1, true); else LD_BWD(1, false); break;
        case 3: if (dr) LD_BWD(3, true); else LD_BWD(3, false); break;
        case 5: if (dr) LD_BWD(5, true); else LD_BWD(5, false); break;
        case 7: if (dr) LD_BWD(7, true); else LD_BWD(7, false); break;
        default: if (dr) LD_BWD(9, true); else LD_BWD(9, false); break;
    }
#undef LD_BWD
    STAGE_LAUNCH_CHECK();
    stage_colreduce(part_conv, dw, db, grid, (long)(k + 1) * D, (k + 1) * D, D, k, st);
    stage_colreduce(part_ln, dgamma, dbeta, grid, (long)2 * D, 2 * D, D, 1, st);
    STAGE_LAUNCH_CHECK();
    return 0;
}

template <typename T>
static int ln_dwconv_bwd_t(const T* dh, const T* xin, const float* mean, const float* rstd, const float* gamma,
                           const float* beta, const float* w, T* dx, const T* dx_add, float* dgamma, float* dbeta, float* dw,
                           float* db, long long M, int L, int D, int k, float p_drop, unsigned long long seed, void* ws,
                           size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!ld_shape_ok(D, k)) return STAGE_ERR_SHAPE;
    if (ws_bytes < stage_ln_dwconv_bwd_ws_bytes(D, k)) return STAGE_ERR_WORKSPACE;
    if (M <= 0 || L <= 0) {
        (void)hipMemsetAsync(dw, 0, sizeof(float) * D * k, st);
        (void)hipMemsetAsync(db, 0, sizeof(float) * D, st);
        (void)hipMemsetAsync(dgamma, 0, sizeof(float) * D, st);
        (void)hipMemsetAsync(dbeta, 0, sizeof(float) * D, st);
        return 0;
    }
    const int rpi = 256 / (D / 4);
    const int clen = stage_chunk_len(L);
    const long items = (long)M * ((L + clen - 1) / clen);
    const int grid = stage_grid_for(items, rpi, LD_PART_CAP);
    float* part_conv = (float*)ws;
    float* part_ln = part_conv + (size_t)LD_PART_CAP * (k + 1) * D;
    const size_t lds = (size_t)rpi * (k + 1) * D * sizeof(float);
    const bool dr = p_drop > 0.f;
    const uint64_t sd = dr ? (uint64_t)seed : 0;
    const uint32_t th = dr ? drop_thresh16(p_drop) : 0u;
    const float ik = dr ? 1.0f / (1.0f - p_drop) : 1.0f;
#define LD_BWD(KV, DR)                                                                                                    \
    hipLaunchKernelGGL((ln_dwconv_bwd_kernel<KV, DR, T>), dim3(grid), dim3(256), lds, st, dh, xin, mean, rstd, gamma, beta, w, \
                       dx, dx_add, part_conv, part_ln, (long)M, L, D, sd, th, ik, clen)
    switch (k) {
        case 1: if (dr) LD_BWD(1, true); else LD_BWD(1, false); break;
        case 3: if (dr) LD_BWD(3, true); else LD_BWD(3, false); break;
        case 5: if (dr) LD_BWD(5, true); else LD_BWD(5, false); break;
        case 7: if (dr) LD_BWD(7, true); else LD_BWD(7, false); break;
        default: if (dr) LD_BWD(9, true); else LD_BWD(9, false); break;
    }
#undef LD_BWD
    STAGE_LAUNCH_CHECK();
    stage_colreduce(part_conv, dw, db, grid, (long)(k + 1) * D, (k + 1) * D, D, k, st);
    stage_colreduce(part_ln, dgamma, dbeta, grid, (long)2 * D, 2 * D, D, 1, st);
    STAGE_LAUNCH_CHECK();
    return 0;
}

extern "C" int stage_ln_dwconv_bwd(const float* dh, const float* xin, const float* mean, const float* rstd,
                                   const float* gamma, const float* beta, const float* w, float* dx, const float* dx_add,
                                   float* dgamma, float* dbeta, float* dw, float* db, long long M, int L, int D, int k,
                                   float p_drop, unsigned long long seed, void* ws, size_t ws_bytes, void* stream) {
    return ln_dwconv_bwd_t<float>(dh, xin, mean, rstd, gamma, beta, w, dx, dx_add, dgamma, dbeta, dw, db, M, L, D, k, p_drop, seed,
                                  ws, ws_bytes, stream);
}
// bf16 storage mode: activations (dh, xin, dx, dx_add) bf16; statistics, parameters and their gradients fp32
extern "C" int stage_ln_dwconv_bwd_bf16(const void* dh, const void* xin, const float* mean, const float* rstd,
                                        const float* gamma, const float* beta, const float* w, void* dx, const void* dx_add,
                                        float* dgamma, float* dbeta, float* dw, float* db, long long M, int L, int D, int k,
                                        float p_drop, unsigned long long seed, void* ws, size_t ws_bytes, void* stream) {
    typedef stage_bf16 B;
    return ln_dwconv_bwd_t<B>((const B*)dh, (const B*)xin, mean, rstd, gamma, beta, w, (B*)dx, (const B*)dx_add, dgamma, dbeta, dw,
                              db, M, L, D, k, p_drop, seed, ws, ws_bytes, stream);
}
