// Row-wise streaming kernels of the STAGE hot path (all HBM-bound; fp32; float4 per lane; wave64):
//   * LayerNorm (+ fused inverted dropout) forward / backward            nn.LayerNorm + nn.Dropout pairs,
//       model/stage.py:85-91,98-104,107-120,133-138, LinearWrapper :15-32, model/encoder.py:37-41
//   * the same over the virtual row [a, b, a*b] (never materialised)       model/stage.py:276-279, 381-385
//   * L2 row normalisation (+dropout) forward / backward                   F.normalize, model/stage.py:256,
//       model/context_query_attention.py:95-96
//   * masked max over a (ranged) sequence axis forward / backward          model/stage.py:503-505, 425-432, 532-533
//   * deterministic two-stage column reductions for the affine/bias gradients
// One "row group" of LPR lanes (power of two, 4..64) owns one row; a wave processes 64/LPR rows at a time.
#include <stdlib.h>
#include "common.h"
#include "../../include/stage_hip.h"

#define MAXV 4           // float4 per lane per row  -> K <= 4*4*64 = 1024
#define GRID_CAP 1024    // blocks of a grid-stride streaming launch (256 CU x 4)
#define PART_CAP 512     // blocks that emit column partials (bounds the workspace)

// ------------------------------------------------------------------------------------------------
// LayerNorm forward.  MODE 0: plain row of K floats.  MODE 1: virtual row [a, b, a*b] with K = 3*D.
// For MODE 1 the a-row may be broadcast: a_row = (row / (rep*inner)) * inner + row % inner.
// ------------------------------------------------------------------------------------------------
template <typename T> struct RowSrcT {
    const T* x;       // MODE 0: x ; MODE 1: a
    const T* b;       // MODE 1: b ; MODE 0: optional residual added before the norm (x + res), may be NULL
    int D;            // MODE 1: width of a / b
    int rep, inner;   // MODE 1: broadcast description of a (rep == 1 -> none)
                      // MODE 0: inner = residual period in rows (0: res row == row; L: res row = row % L -> pe table)
    T* sum_out;       // MODE 0: where to write x + res (NULL: not needed)
    const int* gather;  // MODE 0, fast kernels: row r reads x[gather[r]] (ragged context rows: the compact rows of a padded feature
                        // tensor, csrc/ragged.hip); everything else -- y, statistics, dropout counter -- is indexed by r.  NULL: x[r]
};
typedef RowSrcT<float> RowSrc;   // T = stage_bf16: bf16 storage (generic kernels only; statistics, affine parameters and
                                 // all arithmetic stay fp32)

__device__ __forceinline__ long a_row_of(long row, int rep, int inner) {
    if (rep == 1) return row;
    long g = row / ((long)rep * inner);
    return g * inner + row % inner;
}

template <int MODE, bool DROP, typename T = float>
__global__ __launch_bounds__(256) void ln_fwd_kernel(RowSrcT<T> src, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, long rows,
                                                     int K, float eps, int LPR, uint64_t seed, uint32_t th,
                                                     float inv_keep) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int RPW = 64 / LPR, sub = lane / LPR, sl = lane % LPR;
    const int K4 = K >> 2;
    const float invK = 1.0f / (float)K;
    for (long base = ((long)blockIdx.x * wpb + wave) * RPW; base < rows; base += (long)gridDim.x * wpb * RPW) {
        const long row = base + sub;
        const bool ok = row < rows;
        float4 v[MAXV];
        float s = 0.f;
        if (MODE == 0) {
#pragma unroll
            for (int t = 0; t < MAXV; t++) {
                int j = sl + t * LPR;
                if (ok && j < K4) {
                    v[t] = ldv4(src.x + row * K + 4 * j);
                    if (src.b) {
                        const long rr = src.inner > 0 ? row % src.inner : row;
                        v[t] = f4add(v[t], ldv4(src.b + rr * K + 4 * j));
                        if (src.sum_out) stv4(src.sum_out + row * K + 4 * j, v[t]);
                    }
                    s += f4hsum(v[t]);
                } else v[t] = f4zero();
            }
        } else {
            // lane owns d-quads q = sl (and sl + LPR); v[0..2] = a,b,a*b of quad 0 ; v[3] unused unless D4 > LPR
            // (MODE 1 supports D/4 <= LPR, i.e. one quad per lane: D <= 256)
            const int D4 = src.D >> 2;
            if (ok && sl < D4) {
                long ar = a_row_of(row, src.rep, src.inner);
                v[0] = ldv4(src.x + ar * src.D + 4 * sl);
                v[1] = ldv4(src.b + row * src.D + 4 * sl);
                v[2] = f4mul(v[0], v[1]);
                s = f4hsum(v[0]) + f4hsum(v[1]) + f4hsum(v[2]);
            } else { v[0] = v[1] = v[2] = f4zero(); }
            v[3] = f4zero();
        }
        s = group_sum(s, LPR);
        const float mu = s * invK;
        float q = 0.f;
        if (MODE == 0) {
#pragma unroll
            for (int t = 0; t < MAXV; t++) {
                int j = sl + t * LPR;
                if (j < K4) {
                    float4 d = make_float4(v[t].x - mu, v[t].y - mu, v[t].z - mu, v[t].w - mu);
                    q += f4hsum(f4mul(d, d));
                }
            }
        } else {
            if (sl < (src.D >> 2)) {
#pragma unroll
                for (int t = 0; t < 3; t++) {
                    float4 d = make_float4(v[t].x - mu, v[t].y - mu, v[t].z - mu, v[t].w - mu);
                    q += f4hsum(f4mul(d, d));
                }
            }
        }
        q = group_sum(q, LPR);
        const float rs = 1.0f / sqrtf(q * invK + eps);
        if (ok && sl == 0) {
            if (mean) mean[row] = mu;
            if (rstd) rstd[row] = rs;
        }
        if (!ok) continue;
        const int nv = (MODE == 0) ? MAXV : 3;
#pragma unroll
        for (int t = 0; t < nv; t++) {
            int j = (MODE == 0) ? (sl + t * LPR) : (t * (src.D >> 2) + sl);  // float4 column index inside the row
            bool live = (MODE == 0) ? (j < K4) : (sl < (src.D >> 2));
            if (live) {
                float4 g = ld4(gamma + 4 * j), bb = ld4(beta + 4 * j);
                float4 o;
                o.x = (v[t].x - mu) * rs * g.x + bb.x;
                o.y = (v[t].y - mu) * rs * g.y + bb.y;
                o.z = (v[t].z - mu) * rs * g.z + bb.z;
                o.w = (v[t].w - mu) * rs * g.w + bb.w;
                if (DROP) o = f4mul(o, drop4(seed, (uint64_t)row * K4 + j, th, inv_keep));
                stv4(y + row * K + 4 * j, o);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm backward: dx (MODE 0) or da_full/db (MODE 1) + per-block partial dgamma/dbeta.
// part layout: [gridDim.x][2][K]  (0: dgamma, 1: dbeta), reduced by colreduce_kernel.
// ------------------------------------------------------------------------------------------------
template <int MODE, bool DROP, typename T = float, typename TDX = T>   // TDX: dx (MODE 1: the unreduced da rows, kept fp32)
__global__ __launch_bounds__(256) void ln_bwd_kernel(RowSrcT<T> src, const T* __restrict__ dy,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, TDX* __restrict__ dx,
                                                     T* __restrict__ db_out, float* __restrict__ part, long rows,
                                                     int K, int LPR, uint64_t seed, uint32_t th, float inv_keep) {
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [wpb*RPW][2][K]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int RPW = 64 / LPR, sub = lane / LPR, sl = lane % LPR;
    const int K4 = K >> 2;
    const float invK = 1.0f / (float)K;
    float4 ag[MAXV], ab[MAXV];
#pragma unroll
    for (int t = 0; t < MAXV; t++) ag[t] = ab[t] = f4zero();
    const int D4 = (MODE == 1) ? (src.D >> 2) : 0;

    for (long base = ((long)blockIdx.x * wpb + wave) * RPW; base < rows; base += (long)gridDim.x * wpb * RPW) {
        const long row = base + sub;
        const bool ok = row < rows;
        float4 xh[MAXV], g[MAXV], av = f4zero(), bv = f4zero();
        float mu = 0.f, rs = 0.f;
        if (ok) { mu = mean[row]; rs = rstd[row]; }
        float s1 = 0.f, s2 = 0.f;
        const int nv = (MODE == 0) ? MAXV : 3;
        if (MODE == 1 && ok && sl < D4) {
            long ar = a_row_of(row, src.rep, src.inner);
            av = ldv4(src.x + ar * src.D + 4 * sl);
            bv = ldv4(src.b + row * src.D + 4 * sl);
        }
#pragma unroll
        for (int t = 0; t < nv; t++) {
            int j = (MODE == 0) ? (sl + t * LPR) : (t * D4 + sl);
            bool live = ok && ((MODE == 0) ? (j < K4) : (sl < D4));
            if (live) {
                float4 xv;
                if (MODE == 0) xv = ldv4(src.x + row * K + 4 * j);
                else xv = (t == 0) ? av : ((t == 1) ? bv : f4mul(av, bv));
                float4 d = ldv4(dy + row * K + 4 * j);
                if (DROP) d = f4mul(d, drop4(seed, (uint64_t)row * K4 + j, th, inv_keep));
                float4 gm = ld4(gamma + 4 * j);
                xh[t] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
                g[t] = f4mul(d, gm);
                s1 += f4hsum(g[t]);
                s2 += f4hsum(f4mul(g[t], xh[t]));
                ag[t] = f4add(ag[t], f4mul(d, xh[t]));
                ab[t] = f4add(ab[t], d);
            } else { xh[t] = g[t] = f4zero(); }
        }
        s1 = group_sum(s1, LPR) * invK;
        s2 = group_sum(s2, LPR) * invK;
        if (!ok) continue;
        float4 dz[3];
#pragma unroll
        for (int t = 0; t < nv; t++) {
            int j = (MODE == 0) ? (sl + t * LPR) : (t * D4 + sl);
            bool live = (MODE == 0) ? (j < K4) : (sl < D4);
            if (live) {
                float4 o;
                o.x = rs * (g[t].x - s1 - xh[t].x * s2);
                o.y = rs * (g[t].y - s1 - xh[t].y * s2);
                o.z = rs * (g[t].z - s1 - xh[t].z * s2);
                o.w = rs * (g[t].w - s1 - xh[t].w * s2);
                if (MODE == 0) {
                    if (dx) {
                        if (src.b) o = f4add(o, ldv4(src.b + row * K + 4 * j));  // + gradient of the exported sum
                        stv4(dx + row * K + 4 * j, o);
                    }
                }
                else dz[t] = o;
            }
        }
        if (MODE == 1 && sl < D4) {
            // z = [a, b, a*b]:  da = dz0 + dz2*b ; db = dz1 + dz2*a
            stv4(dx + row * src.D + 4 * sl, f4add(dz[0], f4mul(dz[2], bv)));
            stv4(db_out + row * src.D + 4 * sl, f4add(dz[1], f4mul(dz[2], av)));
        }
    }
    // block reduction of the per-lane column partials
    const int slot = wave * RPW + sub;
    float* sg = smem + (size_t)slot * 2 * K;
    const int nv2 = (MODE == 0) ? MAXV : 3;
#pragma unroll
    for (int t = 0; t < nv2; t++) {
        int j = (MODE == 0) ? (sl + t * LPR) : (t * D4 + sl);
        bool live = (MODE == 0) ? (j < K4) : (sl < D4);
        if (live) {
            st4(sg + 4 * j, ag[t]);
            st4(sg + K + 4 * j, ab[t]);
        }
    }
    __syncthreads();
    const int nslots = wpb * RPW;
    for (int c = threadIdx.x; c < 2 * K; c += blockDim.x) {
        float acc = 0.f;
        for (int s = 0; s < nslots; s++) acc += smem[(size_t)s * 2 * K + c];
        part[(size_t)blockIdx.x * 2 * K + c] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// Fast variants for the shapes the STAGE path actually runs (K = 4*LPR for MODE 0, D = 4*LPR for MODE 1: every lane owns
// the same column quad(s) of every row it touches).  Same arithmetic, same summation order as the generic kernels above;
// what changes is the memory schedule: UR row slots per wave iteration, all loads of an iteration issued back to back
// from clamped addresses (a guarded load compiles to an exec branch + s_waitcnt per load and serialises the row), the
// affine parameters live in registers for the whole kernel, stores are the only predicated memory operations.
// ------------------------------------------------------------------------------------------------
#define LN_UR 4
__device__ __forceinline__ long a_row_fast(long row, int rep, int inner) {
    if (rep == 1) return row;
    const unsigned r = (unsigned)row, ri = (unsigned)(rep * inner);   // rows < 2^31 on this path (checked by the launcher)
    const unsigned gq = r / ri;
    return (long)(gq * (unsigned)inner + r % (unsigned)inner);
}

template <int MODE, bool DROP, int NQ0, typename T = float, bool GATHER = false>
__global__ __launch_bounds__(256) void ln_fwd_fast_kernel(RowSrcT<T> src, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, T* __restrict__ y,
                                                          float* __restrict__ mean, float* __restrict__ rstd, long rows,
                                                          int K, float eps, int LPR, uint64_t seed, uint32_t th,
                                                          float inv_keep) {
    constexpr int NQ = (MODE == 0) ? NQ0 : 3;        // column quads per lane (MODE 0: quad sl + t*LPR, the last may be ragged)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int RPW = 64 / LPR, sub = lane / LPR, sl = lane % LPR;
    const int K4 = K >> 2, D = (MODE == 0) ? K : src.D, D4 = D >> 2;
    const float invK = 1.0f / (float)K;
    int jq[NQ], jc[NQ];
    bool jok[NQ];
    float4 gm[NQ], bt[NQ];
#pragma unroll
    for (int t = 0; t < NQ; t++) {
        jq[t] = (MODE == 0) ? sl + t * LPR : t * D4 + sl;
        jok[t] = jq[t] < K4;
        jc[t] = jok[t] ? jq[t] : K4 - 1;
        gm[t] = ld4(gamma + 4 * jc[t]);
        bt[t] = ld4(beta + 4 * jc[t]);
    }
    const long step = (long)gridDim.x * wpb * RPW * LN_UR;
    for (long base = ((long)blockIdx.x * wpb + wave) * RPW * LN_UR; base < rows; base += step) {
        float4 v[LN_UR][NQ], rv[LN_UR][(MODE == 0) ? NQ : 1];
        long row[LN_UR];
#pragma unroll
        for (int u = 0; u < LN_UR; u++) {
            row[u] = base + u * RPW + sub;
            const long rc = row[u] < rows ? row[u] : rows - 1;
            if (MODE == 0) {
                const long rr = src.b ? (src.inner > 0 ? (long)((unsigned)rc % (unsigned)src.inner) : rc) : 0;
                const long xr = GATHER ? (long)src.gather[rc] : rc;
#pragma unroll
                for (int t = 0; t < NQ; t++) {
                    v[u][t] = ldv4s(src.x + xr * K + 4 * jc[t]);
                    if (src.b) rv[u][t] = ldv4(src.b + rr * K + 4 * jc[t]);
                }
            } else {
                v[u][0] = ldv4(src.x + a_row_fast(rc, src.rep, src.inner) * D + 4 * sl);
                v[u][1] = ldv4s(src.b + rc * D + 4 * sl);
            }
        }
#pragma unroll
        for (int u = 0; u < LN_UR; u++) {
            const bool ok = row[u] < rows;
            float s = 0.f;
            if (MODE == 0) {
#pragma unroll
                for (int t = 0; t < NQ; t++) {
                    if (src.b) {
                        v[u][t] = f4add(v[u][t], rv[u][t]);
                        if (src.sum_out && ok && jok[t]) stv4(src.sum_out + row[u] * K + 4 * jq[t], v[u][t]);
                    }
                    if (!jok[t]) v[u][t] = f4zero();
                    s += f4hsum(v[u][t]);
                }
            } else {
                v[u][2] = f4mul(v[u][0], v[u][1]);
                s = f4hsum(v[u][0]) + f4hsum(v[u][1]) + f4hsum(v[u][2]);
            }
            s = group_sum(s, LPR);
            const float mu = s * invK;
            float q = 0.f;
#pragma unroll
            for (int t = 0; t < NQ; t++) {
                const float4 d = make_float4(v[u][t].x - mu, v[u][t].y - mu, v[u][t].z - mu, v[u][t].w - mu);
                if (jok[t]) q += f4hsum(f4mul(d, d));
            }
            q = group_sum(q, LPR);
            const float rs = 1.0f / sqrtf(q * invK + eps);
            if (ok && sl == 0) {
                if (mean) mean[row[u]] = mu;
                if (rstd) rstd[row[u]] = rs;
            }
#pragma unroll
            for (int t = 0; t < NQ; t++) {
                float4 o;
                o.x = (v[u][t].x - mu) * rs * gm[t].x + bt[t].x;
                o.y = (v[u][t].y - mu) * rs * gm[t].y + bt[t].y;
                o.z = (v[u][t].z - mu) * rs * gm[t].z + bt[t].z;
                o.w = (v[u][t].w - mu) * rs * gm[t].w + bt[t].w;
                if (DROP) o = f4mul(o, drop4(seed, (uint64_t)row[u] * K4 + jq[t], th, inv_keep));
                if (ok && jok[t]) stv4(y + row[u] * K + 4 * jq[t], o);
            }
        }
    }
}

// MM (MODE 0): the incoming gradient is that of a masked max over groups of mm_L consecutive rows (stage_ln_masked_max_*): `dy`
// is the (rows / mm_L, K) gradient of the maxima and the row gradient dy[row][d] = (mm_idx[grp][d] == l) ? dy[grp][d] * mask[row] : 0
// is formed while it is loaded -- the dense (rows, K) gradient tensor (491 MB at the classifier head) is never written or read.
template <int MODE, bool DROP, int NQ0, typename T = float, typename TDX = T, bool MM = false, bool GATHER = false>
__global__ __launch_bounds__(256) void ln_bwd_fast_kernel(RowSrcT<T> src, const T* __restrict__ dy,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma, TDX* __restrict__ dx,
                                                          T* __restrict__ db_out, float* __restrict__ part, long rows,
                                                          int K, int LPR, uint64_t seed, uint32_t th, float inv_keep,
                                                          const int* __restrict__ mm_idx = nullptr,
                                                          const float* __restrict__ mm_mask = nullptr, int mm_L = 1,
                                                          const int4* __restrict__ mm_rowinfo = nullptr) {
    // mm_rowinfo != NULL (ragged token rows): row r belongs to group mm_rowinfo[r].z at position .w and its mask is mm_mask[.x]
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [wpb*RPW][2][K]
    constexpr int NQ = (MODE == 0) ? NQ0 : 3;
    constexpr int NX = (MODE == 0) ? NQ0 : 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int RPW = 64 / LPR, sub = lane / LPR, sl = lane % LPR;
    const int K4 = K >> 2, D = (MODE == 0) ? K : src.D, D4 = D >> 2;
    const float invK = 1.0f / (float)K;
    int jq[NQ], jc[NQ];
    bool jok[NQ];
    float4 gm[NQ], ag[NQ], ab[NQ];
#pragma unroll
    for (int t = 0; t < NQ; t++) {
        jq[t] = (MODE == 0) ? sl + t * LPR : t * D4 + sl;
        jok[t] = jq[t] < K4;
        jc[t] = jok[t] ? jq[t] : K4 - 1;
        gm[t] = ld4(gamma + 4 * jc[t]);
        ag[t] = ab[t] = f4zero();
    }
    const long step = (long)gridDim.x * wpb * RPW * LN_UR;
    for (long base = ((long)blockIdx.x * wpb + wave) * RPW * LN_UR; base < rows; base += step) {
        float4 xv[LN_UR][NX], d[LN_UR][NQ], ra[LN_UR][(MODE == 0) ? NQ : 1];
        float mu[LN_UR], rs[LN_UR];
        long row[LN_UR];
#pragma unroll
        for (int u = 0; u < LN_UR; u++) {
            row[u] = base + u * RPW + sub;
            const long rc = row[u] < rows ? row[u] : rows - 1;
            mu[u] = mean[rc];
            rs[u] = rstd[rc];
            if (MODE == 0) {
                const long xr = GATHER ? (long)src.gather[rc] : rc;
#pragma unroll
                for (int t = 0; t < NQ; t++) {
                    xv[u][t] = ldv4s(src.x + xr * K + 4 * jc[t]);
                    if (dx && src.b) ra[u][t] = ldv4(src.b + rc * K + 4 * jc[t]);
                }
            } else {
                xv[u][0] = ldv4(src.x + a_row_fast(rc, src.rep, src.inner) * D + 4 * sl);
                xv[u][1] = ldv4s(src.b + rc * D + 4 * sl);
            }
            if (MM) {
                long grp = rc / mm_L;
                int l = (int)(rc - grp * mm_L);
                long mrow = rc;
                if (mm_rowinfo) {
                    const int4 ri = mm_rowinfo[rc];
                    grp = ri.z;
                    l = ri.w;
                    mrow = ri.x;
                }
                const float mk = mm_mask[mrow];
#pragma unroll
                for (int t = 0; t < NQ; t++) {
                    const int4 bi = *reinterpret_cast<const int4*>(mm_idx + grp * K + 4 * jc[t]);
                    const float4 gq = ldv4(dy + grp * K + 4 * jc[t]);
                    d[u][t] = make_float4(bi.x == l ? gq.x * mk : 0.f, bi.y == l ? gq.y * mk : 0.f, bi.z == l ? gq.z * mk : 0.f,
                                          bi.w == l ? gq.w * mk : 0.f);
                }
            } else {
#pragma unroll
                for (int t = 0; t < NQ; t++) d[u][t] = ldv4s(dy + rc * K + 4 * jc[t]);
            }
        }
#pragma unroll
        for (int u = 0; u < LN_UR; u++) {
            const bool ok = row[u] < rows;
            float4 xh[NQ], g[NQ];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int t = 0; t < NQ; t++) {
                const float4 x = (MODE == 0) ? xv[u][t] : ((t == 0) ? xv[u][0] : ((t == 1) ? xv[u][1] : f4mul(xv[u][0], xv[u][1])));
                float4 dd = d[u][t];
                if (DROP) dd = f4mul(dd, drop4(seed, (uint64_t)row[u] * K4 + jq[t], th, inv_keep));
                if (!ok || !jok[t]) dd = f4zero();   // rows / quads past the end were read from a clamped address
                xh[t] = make_float4((x.x - mu[u]) * rs[u], (x.y - mu[u]) * rs[u], (x.z - mu[u]) * rs[u], (x.w - mu[u]) * rs[u]);
                g[t] = f4mul(dd, gm[t]);
                s1 += f4hsum(g[t]);
                s2 += f4hsum(f4mul(g[t], xh[t]));
                ag[t] = f4add(ag[t], f4mul(dd, xh[t]));
                ab[t] = f4add(ab[t], dd);
            }
            s1 = group_sum(s1, LPR) * invK;
            s2 = group_sum(s2, LPR) * invK;
            float4 dz[NQ];
#pragma unroll
            for (int t = 0; t < NQ; t++) {
                dz[t].x = rs[u] * (g[t].x - s1 - xh[t].x * s2);
                dz[t].y = rs[u] * (g[t].y - s1 - xh[t].y * s2);
                dz[t].z = rs[u] * (g[t].z - s1 - xh[t].z * s2);
                dz[t].w = rs[u] * (g[t].w - s1 - xh[t].w * s2);
            }
            if (MODE == 0) {
                if (dx && ok) {
#pragma unroll
                    for (int t = 0; t < NQ; t++) {
                        if (src.b) dz[t] = f4add(dz[t], ra[u][t]);  // + gradient of the exported sum
                        if (jok[t]) stv4(dx + row[u] * K + 4 * jq[t], dz[t]);
                    }
                }
            } else if (ok) {
                // z = [a, b, a*b]:  da = dz0 + dz2*b ; db = dz1 + dz2*a
                stv4(dx + row[u] * D + 4 * sl, f4add(dz[0], f4mul(dz[NQ - 1], xv[u][NX - 1])));
                stv4(db_out + row[u] * D + 4 * sl, f4add(dz[NQ > 1 ? 1 : 0], f4mul(dz[NQ - 1], xv[u][0])));
            }
        }
    }
    // block reduction of the per-lane column partials (same slot layout and order as the generic kernel)
    const int slot = wave * RPW + sub;
    float* sg = smem + (size_t)slot * 2 * K;
#pragma unroll
    for (int t = 0; t < NQ; t++)
        if (jok[t]) {
            st4(sg + 4 * jq[t], ag[t]);
            st4(sg + K + 4 * jq[t], ab[t]);
        }
    __syncthreads();
    const int nslots = wpb * RPW;
    for (int c = threadIdx.x; c < 2 * K; c += blockDim.x) {
        float acc = 0.f;
        for (int s = 0; s < nslots; s++) acc += smem[(size_t)s * 2 * K + c];
        part[(size_t)blockIdx.x * 2 * K + c] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// cat3 LayerNorm backward with the broadcast reduction fused in (rep > 1: `a` is shared by the `rep` frames of a group).
// A workgroup owns one (group, chunk of frames); a lane keeps the SAME positions `in` of the group for every frame, so the
// gradient of the broadcast operand accumulates in registers in frame order (deterministic) and only one partial row per
// (group, chunk, in) is written -- instead of a full (rows, D) tensor that stage_reduce_rep reads back (2 x 491 MB at the
// full config).  `a` is loaded once per workgroup.  da_part: [G][CH][inner][D].
// ------------------------------------------------------------------------------------------------
template <bool DROP, int KI, typename T = float>
__global__ __launch_bounds__(256) void cat3_ln_bwd_rep_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                              const T* __restrict__ dy, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                              float* __restrict__ da_part, T* __restrict__ db,
                                                              float* __restrict__ part, int D, int rep, int inner, int CH,
                                                              int fpc, uint64_t seed, uint32_t th, float inv_keep) {
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [RB][2][K]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int D4 = D >> 2, LPR = D4, RPW = 64 / LPR, RB = wpb * RPW;
    const int sub = lane / LPR, sl = lane % LPR, slot = wave * RPW + sub;
    const int K = 3 * D, K4 = 3 * D4;
    const float invK = 1.0f / (float)K;
    const int g = blockIdx.x / CH, chunk = blockIdx.x % CH;
    const int f0 = chunk * fpc, f1 = min(rep, f0 + fpc);
    float4 gm[3], ag[3], ab[3], av[KI], dacc[KI];
#pragma unroll
    for (int t = 0; t < 3; t++) { gm[t] = ld4(gamma + 4 * (t * D4 + sl)); ag[t] = ab[t] = f4zero(); }
    int inx[KI];
    bool iok[KI];
#pragma unroll
    for (int k = 0; k < KI; k++) {
        const int in = slot + RB * k;
        iok[k] = in < inner;
        inx[k] = iok[k] ? in : inner - 1;
        av[k] = ldv4(a + ((long)g * inner + inx[k]) * D + 4 * sl);
        dacc[k] = f4zero();
    }
    for (int f = f0; f < f1; f++) {
        const long rbase = ((long)g * rep + f) * inner;
        float4 bv[KI], d[KI][3];
        float mu[KI], rs[KI];
#pragma unroll
        for (int k = 0; k < KI; k++) {
            const long row = rbase + inx[k];
            bv[k] = ldv4s(b + row * D + 4 * sl);
#pragma unroll
            for (int t = 0; t < 3; t++) d[k][t] = ldv4s(dy + row * K + 4 * (t * D4 + sl));
            mu[k] = mean[row];
            rs[k] = rstd[row];
        }
#pragma unroll
        for (int k = 0; k < KI; k++) {
            const long row = rbase + inx[k];
            float4 xh[3], gq[3];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int t = 0; t < 3; t++) {
                const float4 x = (t == 0) ? av[k] : ((t == 1) ? bv[k] : f4mul(av[k], bv[k]));
                float4 dd = d[k][t];
                if (DROP) dd = f4mul(dd, drop4(seed, (uint64_t)row * K4 + t * D4 + sl, th, inv_keep));
                if (!iok[k]) dd = f4zero();
                xh[t] = make_float4((x.x - mu[k]) * rs[k], (x.y - mu[k]) * rs[k], (x.z - mu[k]) * rs[k], (x.w - mu[k]) * rs[k]);
                gq[t] = f4mul(dd, gm[t]);
                s1 += f4hsum(gq[t]);
                s2 += f4hsum(f4mul(gq[t], xh[t]));
                ag[t] = f4add(ag[t], f4mul(dd, xh[t]));
                ab[t] = f4add(ab[t], dd);
            }
            s1 = group_sum_bperm(s1, LPR) * invK;
            s2 = group_sum_bperm(s2, LPR) * invK;
            float4 dz[3];
#pragma unroll
            for (int t = 0; t < 3; t++) {
                dz[t].x = rs[k] * (gq[t].x - s1 - xh[t].x * s2);
                dz[t].y = rs[k] * (gq[t].y - s1 - xh[t].y * s2);
                dz[t].z = rs[k] * (gq[t].z - s1 - xh[t].z * s2);
                dz[t].w = rs[k] * (gq[t].w - s1 - xh[t].w * s2);
            }
            // z = [a, b, a*b]:  da = dz0 + dz2*b ; db = dz1 + dz2*a
            if (iok[k]) {
                dacc[k] = f4add(dacc[k], f4add(dz[0], f4mul(dz[2], bv[k])));
                stv4(db + row * D + 4 * sl, f4add(dz[1], f4mul(dz[2], av[k])));
            }
        }
    }
#pragma unroll
    for (int k = 0; k < KI; k++)
        if (iok[k]) st4(da_part + (((long)g * CH + chunk) * inner + inx[k]) * D + 4 * sl, dacc[k]);
    // block reduction of the per-lane column partials
    float* sg = smem + (size_t)slot * 2 * K;
#pragma unroll
    for (int t = 0; t < 3; t++) {
        st4(sg + 4 * (t * D4 + sl), ag[t]);
        st4(sg + K + 4 * (t * D4 + sl), ab[t]);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * K; c += blockDim.x) {
        float acc = 0.f;
        for (int r = 0; r < RB; r++) acc += smem[(size_t)r * 2 * K + c];
        part[(size_t)blockIdx.x * 2 * K + c] = acc;
    }
}

// out[c] = sum_b part[b*stride + c]   (fixed order -> deterministic)
__global__ void colreduce_kernel(const float* __restrict__ part, float* __restrict__ out, int nb, long stride, int C) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float acc = 0.f;
    for (int b = 0; b < nb; b++) acc += part[(size_t)b * stride + c];
    out[c] = acc;
}

static int ln_lpr(int K4) {
    int l = stage_pow2_ceil(K4);
    if (l < 4) l = 4;
    if (l > 64) l = 64;
    return l;
}

extern "C" size_t stage_ln_bwd_ws_bytes(int K) { return (size_t)PART_CAP * 2 * (size_t)K * sizeof(float); }

template <int MODE, typename T = float>
static int ln_fwd_launch(RowSrcT<T> src, const float* gamma, const float* beta, T* y, float* mean, float* rstd,
                         long long rows, int K, int LPR, float eps, float p_drop, unsigned long long seed,
                         hipStream_t st) {
    const int rows_per_block = 4 * (64 / LPR);
    const int q4 = (MODE == 0) ? K / 4 : src.D / 4;
    const int nq = (MODE == 0) ? (q4 + LPR - 1) / LPR : (q4 == LPR ? 1 : 0);   // quads per lane; MODE 1 needs D/4 == LPR
    if (nq >= 1 && nq <= 4 && rows < (1ll << 31) && !getenv("STAGE_LN_GENERIC")) {   // fixed column quads per lane
        const int gridf = stage_grid_for(rows, rows_per_block * LN_UR, GRID_CAP * 2);
        const bool dr = p_drop > 0.f;
        const uint64_t sd = dr ? (uint64_t)seed : 0;
        const uint32_t th = dr ? drop_thresh16(p_drop) : 0u;
        const float ik = dr ? 1.0f / (1.0f - p_drop) : 1.0f;
#define LN_FWD_FAST(DR, NQV)                                                                                           \
    do {                                                                                                                \
        if (MODE == 0 && src.gather)                                                                                    \
            hipLaunchKernelGGL((ln_fwd_fast_kernel<MODE, DR, NQV, T, MODE == 0>), dim3(gridf), dim3(256), 0, st, src, gamma, beta, y, mean, \
                               rstd, (long)rows, K, eps, LPR, sd, th, ik);                                              \
        else                                                                                                            \
            hipLaunchKernelGGL((ln_fwd_fast_kernel<MODE, DR, NQV, T>), dim3(gridf), dim3(256), 0, st, src, gamma, beta, y, mean, rstd, \
                               (long)rows, K, eps, LPR, sd, th, ik);                                                    \
    } while (0)
        switch (MODE == 0 ? nq : 1) {
            case 1: if (dr) LN_FWD_FAST(true, 1); else LN_FWD_FAST(false, 1); break;
            case 2: if (dr) LN_FWD_FAST(true, 2); else LN_FWD_FAST(false, 2); break;
            case 3: if (dr) LN_FWD_FAST(true, 3); else LN_FWD_FAST(false, 3); break;
            default: if (dr) LN_FWD_FAST(true, 4); else LN_FWD_FAST(false, 4); break;
        }
#undef LN_FWD_FAST
        STAGE_LAUNCH_CHECK();
        return 0;
    }
    if (src.gather) return STAGE_ERR_SHAPE;      // gathered rows: fast kernels only
    const int grid = stage_grid_for(rows, rows_per_block, GRID_CAP * 4);
    if (p_drop > 0.f)
        hipLaunchKernelGGL((ln_fwd_kernel<MODE, true, T>), dim3(grid), dim3(256), 0, st, src, gamma, beta, y, mean, rstd,
                           (long)rows, K, eps, LPR, (uint64_t)seed, drop_thresh16(p_drop), 1.0f / (1.0f - p_drop));
    else
        hipLaunchKernelGGL((ln_fwd_kernel<MODE, false, T>), dim3(grid), dim3(256), 0, st, src, gamma, beta, y, mean, rstd,
                           (long)rows, K, eps, LPR, (uint64_t)0, 0u, 1.0f);
    STAGE_LAUNCH_CHECK();
    return 0;
}

template <int MODE, typename T = float, typename TDX = T>
static int ln_bwd_launch(RowSrcT<T> src, const T* dy, const float* mean, const float* rstd, const float* gamma,
                         TDX* dx, T* db_out, float* dgamma, float* dbeta, long long rows, int K, int LPR,
                         float p_drop, unsigned long long seed, void* ws, size_t ws_bytes, hipStream_t st) {
    if (ws_bytes < stage_ln_bwd_ws_bytes(K)) return STAGE_ERR_WORKSPACE;
    if (rows <= 0) {
        (void)hipMemsetAsync(dgamma, 0, sizeof(float) * K, st);
        (void)hipMemsetAsync(dbeta, 0, sizeof(float) * K, st);
        return 0;
    }
    const int rows_per_block = 4 * (64 / LPR);
    const int grid = stage_grid_for(rows, rows_per_block * 8, PART_CAP);
    const size_t lds = (size_t)rows_per_block * 2 * K * sizeof(float);
    float* part = (float*)ws;
    const int q4 = (MODE == 0) ? K / 4 : src.D / 4;
    const int nq = (MODE == 0) ? (q4 + LPR - 1) / LPR : (q4 == LPR ? 1 : 0);
    const bool fast = nq >= 1 && nq <= 4 && rows < (1ll << 31) && !getenv("STAGE_LN_GENERIC");
    if (fast) {
        const bool dr = p_drop > 0.f;
        const uint64_t sd = dr ? (uint64_t)seed : 0;
        const uint32_t th = dr ? drop_thresh16(p_drop) : 0u;
        const float ik = dr ? 1.0f / (1.0f - p_drop) : 1.0f;
#define LN_BWD_FAST(DR, NQV)                                                                                            \
    do {                                                                                                                 \
        if (MODE == 0 && src.gather)                                                                                     \
            hipLaunchKernelGGL((ln_bwd_fast_kernel<MODE, DR, NQV, T, TDX, false, MODE == 0>), dim3(grid), dim3(256), lds, st, src, dy, mean, \
                               rstd, gamma, dx, db_out, part, (long)rows, K, LPR, sd, th, ik);                           \
        else                                                                                                             \
            hipLaunchKernelGGL((ln_bwd_fast_kernel<MODE, DR, NQV, T, TDX>), dim3(grid), dim3(256), lds, st, src, dy, mean, rstd, gamma, dx, \
                               db_out, part, (long)rows, K, LPR, sd, th, ik);                                            \
    } while (0)
        switch (MODE == 0 ? nq : 1) {
            case 1: if (dr) LN_BWD_FAST(true, 1); else LN_BWD_FAST(false, 1); break;
            case 2: if (dr) LN_BWD_FAST(true, 2); else LN_BWD_FAST(false, 2); break;
            case 3: if (dr) LN_BWD_FAST(true, 3); else LN_BWD_FAST(false, 3); break;
            default: if (dr) LN_BWD_FAST(true, 4); else LN_BWD_FAST(false, 4); break;
        }
#undef LN_BWD_FAST
    }
    else if (src.gather) return STAGE_ERR_SHAPE;
    else if (p_drop > 0.f)
        hipLaunchKernelGGL((ln_bwd_kernel<MODE, true, T, TDX>), dim3(grid), dim3(256), lds, st, src, dy, mean, rstd, gamma, dx,
                           db_out, part, (long)rows, K, LPR, (uint64_t)seed, drop_thresh16(p_drop),
                           1.0f / (1.0f - p_drop));
    else
        hipLaunchKernelGGL((ln_bwd_kernel<MODE, false, T, TDX>), dim3(grid), dim3(256), lds, st, src, dy, mean, rstd, gamma, dx,
                           db_out, part, (long)rows, K, LPR, (uint64_t)0, 0u, 1.0f);
    STAGE_LAUNCH_CHECK();
    // one launch for both: column c = t*K + d of the [2][K] partial rows goes to dgamma[d] (t = 0) or dbeta[d] (t = 1)
    stage_colreduce(part, dgamma, dbeta, grid, (long)2 * K, 2 * K, K, 1, st);
    STAGE_LAUNCH_CHECK();
    return 0;
}

extern "C" int stage_layernorm_fwd(const float* x, const float* res, int res_period, float* sum_out, const float* gamma,
                                   const float* beta, float* y, float* mean, float* rstd, long long rows, int K,
                                   float eps, float p_drop, unsigned long long seed, void* stream) {
    if (rows <= 0) return 0;
    if (K % 4 != 0 || K > 4 * MAXV * 64) return STAGE_ERR_SHAPE;
    RowSrc src{x, res, 0, 1, res_period, sum_out};
    return ln_fwd_launch<0, float>(src, gamma, beta, y, mean, rstd, rows, K, ln_lpr(K / 4), eps, p_drop, seed,
                            (hipStream_t)stream);
}

extern "C" int stage_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd,
                                   const float* gamma, float* dx, const float* dx_add, float* dgamma, float* dbeta,
                                   long long rows, int K, float p_drop, unsigned long long seed, void* ws,
                                   size_t ws_bytes, void* stream) {
    if (K % 4 != 0 || K > 4 * MAXV * 64) return STAGE_ERR_SHAPE;
    RowSrc src{x, dx_add, 0, 1, 0, nullptr};
    return ln_bwd_launch<0, float, float>(src, dy, mean, rstd, gamma, dx, (float*)nullptr, dgamma, dbeta, rows, K, ln_lpr(K / 4), p_drop,
                            seed, ws, ws_bytes, (hipStream_t)stream);
}

// The same with GATHERED input rows: row r of y / mean / rstd / the dropout stream reads x[gather[r]] (ragged context rows: the first
// LayerNorm of the input MLPs reads the live rows of the padded feature tensor in place).  The backward gives no input gradient
// (features are data): d gamma / d beta only.
extern "C" int stage_layernorm_gather_fwd(const float* x, const int* gather, const float* gamma, const float* beta, float* y,
                                          float* mean, float* rstd, long long rows, int K, float eps, float p_drop,
                                          unsigned long long seed, void* stream) {
    if (rows <= 0) return 0;
    if (K % 4 != 0 || K > 4 * MAXV * 64 || !gather) return STAGE_ERR_SHAPE;
    RowSrc src{x, nullptr, 0, 1, 0, nullptr, gather};
    return ln_fwd_launch<0, float>(src, gamma, beta, y, mean, rstd, rows, K, ln_lpr(K / 4), eps, p_drop, seed, (hipStream_t)stream);
}
extern "C" int stage_layernorm_gather_bwd(const float* dy, const float* x, const int* gather, const float* mean, const float* rstd,
                                          const float* gamma, float* dgamma, float* dbeta, long long rows, int K, float p_drop,
                                          unsigned long long seed, void* ws, size_t ws_bytes, void* stream) {
    if (K % 4 != 0 || K > 4 * MAXV * 64 || !gather) return STAGE_ERR_SHAPE;
    RowSrc src{x, nullptr, 0, 1, 0, nullptr, gather};
    return ln_bwd_launch<0, float, float>(src, dy, mean, rstd, gamma, (float*)nullptr, (float*)nullptr, dgamma, dbeta, rows, K, ln_lpr(K / 4),
                                          p_drop, seed, ws, ws_bytes, (hipStream_t)stream);
}

// y[rows, 3D] = drop(LN([a, b, a*b]))     a: [rows/(rep) , D] broadcast over `rep` (see a_row_of), b: [rows, D]
extern "C" int stage_cat3_layernorm_fwd(const float* a, const float* b, const float* gamma, const float* beta,
                                        float* y, float* mean, float* rstd, long long rows, int D, int rep, int inner,
                                        float eps, float p_drop, unsigned long long seed, void* stream) {
    if (rows <= 0) return 0;
    if (D % 4 != 0 || D > 256 || rep < 1 || inner < 1) return STAGE_ERR_SHAPE;
    RowSrc src{a, b, D, rep, inner, nullptr};
    return ln_fwd_launch<1, float>(src, gamma, beta, y, mean, rstd, rows, 3 * D, ln_lpr(D / 4), eps, p_drop, seed,
                            (hipStream_t)stream);
}

// da_full[rows, D] (NOT yet reduced over `rep`), db[rows, D], dgamma/dbeta[3D]
extern "C" int stage_cat3_layernorm_bwd(const float* dy, const float* a, const float* b, const float* mean,
                                        const float* rstd, const float* gamma, float* da_full, float* db,
                                        float* dgamma, float* dbeta, long long rows, int D, int rep, int inner,
                                        float p_drop, unsigned long long seed, void* ws, size_t ws_bytes,
                                        void* stream) {
    if (D % 4 != 0 || D > 256 || rep < 1 || inner < 1) return STAGE_ERR_SHAPE;
    RowSrc src{a, b, D, rep, inner, nullptr};
    return ln_bwd_launch<1, float, float>(src, dy, mean, rstd, gamma, da_full, db, dgamma, dbeta, rows, 3 * D, ln_lpr(D / 4),
                            p_drop, seed, ws, ws_bytes, (hipStream_t)stream);
}

// frames per chunk / chunks for the fused broadcast reduction: ~1500 workgroups
static void cat3_rep_chunks(long long groups, int rep, int* CH, int* fpc) {
    long want = (1536 + groups - 1) / groups;
    if (want < 1) want = 1;
    if (want > rep) want = rep;
    *fpc = (int)((rep + want - 1) / want);
    *CH = (rep + *fpc - 1) / *fpc;
}
extern "C" size_t stage_cat3_layernorm_bwd_reduced_ws_bytes(long long rows, int D, int rep, int inner) {
    if (rep < 1 || inner < 1 || rows <= 0) return 0;
    const long long groups = rows / ((long long)rep * inner);
    int CH, fpc;
    cat3_rep_chunks(groups, rep, &CH, &fpc);
    return ((size_t)groups * CH * 2 * 3 * D + (size_t)groups * CH * inner * D) * sizeof(float);
}
// da[rows/rep, D] already reduced over `rep`; requires rep > 1, D/4 a power of two in [4, 64], inner <= 64
template <typename T>
static int cat3_bwd_reduced_t(const T* dy, const T* a, const T* b, const float* mean, const float* rstd, const float* gamma,
                              float* da, T* db, float* dgamma, float* dbeta, long long rows, int D, int rep, int inner,
                              float p_drop, unsigned long long seed, void* ws, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int D4 = D / 4;
    if (D % 4 != 0 || D4 < 4 || D4 > 64 || (D4 & (D4 - 1)) != 0 || rep < 2 || inner < 1 || inner > 64 ||
        rows % ((long long)rep * inner) != 0)
        return STAGE_ERR_SHAPE;
    if (rows <= 0) return 0;
    if (ws_bytes < stage_cat3_layernorm_bwd_reduced_ws_bytes(rows, D, rep, inner)) return STAGE_ERR_WORKSPACE;
    const long long groups = rows / ((long long)rep * inner);
    int CH, fpc;
    cat3_rep_chunks(groups, rep, &CH, &fpc);
    const int K = 3 * D, RB = 4 * (64 / D4), KI = (inner + RB - 1) / RB;
    float* part = (float*)ws;
    float* da_part = part + (size_t)groups * CH * 2 * K;
    const int grid = (int)(groups * CH);
    const size_t lds = (size_t)RB * 2 * K * sizeof(float);
    const bool dr = p_drop > 0.f;
    const uint64_t sd = dr ? (uint64_t)seed : 0;
    const uint32_t th = dr ? drop_thresh16(p_drop) : 0u;
    const float ik = dr ? 1.0f / (1.0f - p_drop) : 1.0f;
#define CAT3_REP(DR, KIV)                                                                                               \
    hipLaunchKernelGGL((cat3_ln_bwd_rep_kernel<DR, KIV, T>), dim3(grid), dim3(256), lds, st, a, b, dy, mean, rstd, gamma,    \
                       da_part, db, part, D, rep, inner, CH, fpc, sd, th, ik)
    if (KI > 8) return STAGE_ERR_SHAPE;
    switch (KI) {
        case 1: if (dr) CAT3_REP(true, 1); else CAT3_REP(false, 1); break;
        case 2: if (dr) CAT3_REP(true, 2); else CAT3_REP(false, 2); break;
        case 3: if (dr) CAT3_REP(true, 3); else CAT3_REP(false, 3); break;
        case 4: if (dr) CAT3_REP(true, 4); else CAT3_REP(false, 4); break;
        case 5: if (dr) CAT3_REP(true, 5); else CAT3_REP(false, 5); break;
        case 6: if (dr) CAT3_REP(true, 6); else CAT3_REP(false, 6); break;
        case 7: if (dr) CAT3_REP(true, 7); else CAT3_REP(false, 7); break;
        default: if (dr) CAT3_REP(true, 8); else CAT3_REP(false, 8); break;
    }
#undef CAT3_REP
    STAGE_LAUNCH_CHECK();
    stage_colreduce(part, dgamma, dbeta, grid, (long)2 * K, 2 * K, K, 1, st);
    STAGE_LAUNCH_CHECK();
    return stage_reduce_rep(da_part, da, groups, CH, (long long)inner * D, stream);
}

extern "C" int stage_cat3_layernorm_bwd_reduced(const float* dy, const float* a, const float* b, const float* mean,
                                                const float* rstd, const float* gamma, float* da, float* db,
                                                float* dgamma, float* dbeta, long long rows, int D, int rep, int inner,
                                                float p_drop, unsigned long long seed, void* ws, size_t ws_bytes,
                                                void* stream) {
    return cat3_bwd_reduced_t<float>(dy, a, b, mean, rstd, gamma, da, db, dgamma, dbeta, rows, D, rep, inner, p_drop, seed, ws,
                                     ws_bytes, stream);
}

// ------------------------------------------------------------------------------------------------
// out[g, inner, D] = sum_{r < rep} in[g, r, inner, D]      (undoes a broadcast over `rep`; deterministic)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void reduce_rep_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                         long groups, int rep, long inner4) {
    const long total = groups * inner4;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long g = e / inner4, q = e % inner4;
        const float* p = in + (g * rep * inner4 + q) * 4;
        float4 acc = f4zero();
        for (int r = 0; r < rep; r++) acc = f4add(acc, ld4(p + (long)r * inner4 * 4));
        st4(out + e * 4, acc);
    }
}

extern "C" int stage_reduce_rep(const float* in, float* out, long long groups, int rep, long long inner_elems,
                                void* stream) {
    if (groups <= 0 || inner_elems <= 0) return 0;
    if (inner_elems % 4 != 0) return STAGE_ERR_SHAPE;
    const long inner4 = inner_elems / 4;
    const int grid = stage_grid_for(groups * inner4, 256, GRID_CAP * 4);
    hipLaunchKernelGGL(reduce_rep_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, out, (long)groups, rep,
                       inner4);
    STAGE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// L2 row normalisation  y = drop(x / max(||x||, eps))   (F.normalize p=2, eps 1e-12) and its backward.
// ------------------------------------------------------------------------------------------------
template <bool DROP, typename T = float>
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                         float* __restrict__ nrm, long rows, int K, float eps, int LPR,
                                                         uint64_t seed, uint32_t th, float inv_keep,
                                                         const int* __restrict__ gather = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int RPW = 64 / LPR, sub = lane / LPR, sl = lane % LPR;
    const int K4 = K >> 2;
    for (long base = ((long)blockIdx.x * wpb + wave) * RPW; base < rows; base += (long)gridDim.x * wpb * RPW) {
        const long row = base + sub;
        const bool ok = row < rows;
        const long xr = (gather && ok) ? (long)gather[row] : row;     // (gathered rows: ragged context rows, see RowSrcT)
        float4 v[MAXV];
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < MAXV; t++) {
            int j = sl + t * LPR;
            if (ok && j < K4) {
                v[t] = ldv4(x + xr * K + 4 * j);
                s += f4hsum(f4mul(v[t], v[t]));
            } else v[t] = f4zero();
        }
        s = group_sum(s, LPR);
        const float n = fmaxf(sqrtf(s), eps);
        if (!ok) continue;
        if (nrm && sl == 0) nrm[row] = n;
#pragma unroll
        for (int t = 0; t < MAXV; t++) {
            int j = sl + t * LPR;
            if (j < K4) {
                float4 o = make_float4(v[t].x / n, v[t].y / n, v[t].z / n, v[t].w / n);
                if (DROP) o = f4mul(o, drop4(seed, (uint64_t)row * K4 + j, th, inv_keep));
                stv4(y + row * K + 4 * j, o);
            }
        }
    }
}

// dx = (g - xh * <xh, g>) / n   with g = dy * dropmask, xh = x / n, n = max(||x||, eps); if ||x|| <= eps the clamp is
// a constant and dx = g / eps (what autograd gives for clamp_min).
// TY: type of dy and of the optional addend `add` (dx = add + ...); TX: type of x; T: type of dx (and of the in-place accumulate)
template <bool DROP, typename T = float, typename TX = T, typename TY = T>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const TY* __restrict__ dy, const TX* __restrict__ x,
                                                         T* __restrict__ dx, long rows, int K, float eps, int LPR,
                                                         uint64_t seed, uint32_t th, float inv_keep, int accumulate,
                                                         const TY* __restrict__ add = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int RPW = 64 / LPR, sub = lane / LPR, sl = lane % LPR;
    const int K4 = K >> 2;
    for (long base = ((long)blockIdx.x * wpb + wave) * RPW; base < rows; base += (long)gridDim.x * wpb * RPW) {
        const long row = base + sub;
        const bool ok = row < rows;
        float4 v[MAXV], g[MAXV];
        float s = 0.f, dot = 0.f;
#pragma unroll
        for (int t = 0; t < MAXV; t++) {
            int j = sl + t * LPR;
            if (ok && j < K4) {
                v[t] = ldv4(x + row * K + 4 * j);
                g[t] = ldv4(dy + row * K + 4 * j);
                if (DROP) g[t] = f4mul(g[t], drop4(seed, (uint64_t)row * K4 + j, th, inv_keep));
                s += f4hsum(f4mul(v[t], v[t]));
                dot += f4hsum(f4mul(v[t], g[t]));
            } else v[t] = g[t] = f4zero();
        }
        s = group_sum(s, LPR);
        dot = group_sum(dot, LPR);
        if (!ok) continue;
        const float nr = sqrtf(s);
        const bool clamped = !(nr > eps);
        const float n = clamped ? eps : nr;
        const float c = clamped ? 0.f : dot / (n * n * n);  // <x,g>/n^3
#pragma unroll
        for (int t = 0; t < MAXV; t++) {
            int j = sl + t * LPR;
            if (j < K4) {
                float4 o = make_float4(g[t].x / n - v[t].x * c, g[t].y / n - v[t].y * c, g[t].z / n - v[t].z * c,
                                       g[t].w / n - v[t].w * c);
                T* p = dx + row * K + 4 * j;
                if (accumulate) o = f4add(o, ldv4(p));
                if (add) o = f4add(o, ldv4(add + row * K + 4 * j));
                stv4(p, o);
            }
        }
    }
}

extern "C" int stage_l2norm_fwd(const float* x, float* y, float* norm_out, long long rows, int K, float eps,
                                float p_drop, unsigned long long seed, void* stream) {
    if (rows <= 0) return 0;
    if (K % 4 != 0 || K > 4 * MAXV * 64) return STAGE_ERR_SHAPE;
    const int LPR = ln_lpr(K / 4);
    const int grid = stage_grid_for(rows, 4 * (64 / LPR), GRID_CAP * 4);
    hipStream_t st = (hipStream_t)stream;
    if (p_drop > 0.f)
        hipLaunchKernelGGL((l2norm_fwd_kernel<true>), dim3(grid), dim3(256), 0, st, x, y, norm_out, (long)rows, K, eps,
                           LPR, (uint64_t)seed, drop_thresh16(p_drop), 1.0f / (1.0f - p_drop));
    else
        hipLaunchKernelGGL((l2norm_fwd_kernel<false>), dim3(grid), dim3(256), 0, st, x, y, norm_out, (long)rows, K,
                           eps, LPR, (uint64_t)0, 0u, 1.0f);
    STAGE_LAUNCH_CHECK();
    return 0;
}

extern "C" int stage_l2norm_gather_fwd(const float* x, const int* gather, float* y, long long rows, int K, float eps, void* stream) {
    if (rows <= 0) return 0;
    if (K % 4 != 0 || K > 4 * MAXV * 64 || !gather) return STAGE_ERR_SHAPE;
    const int LPR = ln_lpr(K / 4);
    const int grid = stage_grid_for(rows, 4 * (64 / LPR), GRID_CAP * 4);
    hipLaunchKernelGGL((l2norm_fwd_kernel<false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, y, (float*)nullptr, (long)rows, K, eps,
                       LPR, (uint64_t)0, 0u, 1.0f, gather);
    STAGE_LAUNCH_CHECK();
    return 0;
}

extern "C" int stage_l2norm_bwd(const float* dy, const float* x, float* dx, long long rows, int K, float eps,
                                float p_drop, unsigned long long seed, int accumulate, void* stream) {
    if (rows <= 0) return 0;
    if (K % 4 != 0 || K > 4 * MAXV * 64) return STAGE_ERR_SHAPE;
    const int LPR = ln_lpr(K / 4);
    const int grid = stage_grid_for(rows, 4 * (64 / LPR), GRID_CAP * 4);
    hipStream_t st = (hipStream_t)stream;
    if (p_drop > 0.f)
        hipLaunchKernelGGL((l2norm_bwd_kernel<true>), dim3(grid), dim3(256), 0, st, dy, x, dx, (long)rows, K, eps, LPR,
                           (uint64_t)seed, drop_thresh16(p_drop), 1.0f / (1.0f - p_drop), accumulate);
    else
        hipLaunchKernelGGL((l2norm_bwd_kernel<false>), dim3(grid), dim3(256), 0, st, dy, x, dx, (long)rows, K, eps,
                           LPR, (uint64_t)0, 0u, 1.0f, accumulate);
    STAGE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Masked max over the sequence axis with an optional per-row window:
//   out[r, d] = max_{l in [st_r, ed_r)} ( x[r,l,d]*m[r,l] + (1-m[r,l])*(-1e10) ),  idx = first arg max
// (mask_logits + torch.max: model/stage.py:503, 425, 429-432, 532-533).  An empty window is an error in the
// reference (torch.max of an empty tensor) and yields -inf / idx -1 here.
// Backward: dx[r, idx, d] = dout[r,d] * m[r, idx]; zero elsewhere (the whole dx is written).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void masked_max_fwd_kernel(const T* __restrict__ x, const float* __restrict__ m,
                                                             const int* __restrict__ win, T* __restrict__ out,
                                                             int* __restrict__ idx, long R, int L, int D4) {
    const long total = R * D4;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long r = e / D4;
        const int q = (int)(e % D4);
        int st = 0, ed = L;
        if (win) { st = max(0, win[2 * r]); ed = min(L, win[2 * r + 1]); }
        float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        int4 bi = make_int4(-1, -1, -1, -1);
        const T* px = x + (r * L) * (long)D4 * 4 + 4 * q;
        // 8 positions per step, all 16 loads issued before the first compare (clamped addresses; a one-position loop keeps
        // a single load in flight per lane); strict > in ascending order keeps the first maximum, as torch.max does
        for (int l0 = st; l0 < ed; l0 += 8) {
            float4 v[8];
            float mk[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int l = min(l0 + u, ed - 1);
                v[u] = ldv4s(px + (long)l * D4 * 4);
                mk[u] = m[r * L + l];
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
        stv4(out + e * 4, best);
        *reinterpret_cast<int4*>(idx + e * 4) = bi;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void masked_max_bwd_kernel(const T* __restrict__ dout, const int* __restrict__ idx,
                                                             const float* __restrict__ m, T* __restrict__ dx,
                                                             long R, int L, int D4, int accumulate) {
    const long total = R * L * D4;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int q = (int)(e % D4);
        const long rl = e / D4;
        const int l = (int)(rl % L);
        const long r = rl / L;
        const int4 bi = *reinterpret_cast<const int4*>(idx + (r * D4 + q) * 4);
        const float4 g = ldv4(dout + (r * D4 + q) * 4);
        const float mk = m[rl];
        float4 o = make_float4(bi.x == l ? g.x * mk : 0.f, bi.y == l ? g.y * mk : 0.f, bi.z == l ? g.z * mk : 0.f,
                               bi.w == l ? g.w * mk : 0.f);
        if (accumulate) o = f4add(o, ldv4(dx + e * 4));
        stv4(dx + e * 4, o);
    }
}

extern "C" int stage_masked_max_fwd(const float* x, const float* mask, const int* window, float* out, int* argmax,
                                    long long R, int L, int D, void* stream) {
    if (R <= 0) return 0;
    if (D % 4 != 0 || L < 1) return STAGE_ERR_SHAPE;
    const int grid = stage_grid_for(R * (D / 4), 256, GRID_CAP * 8);
    hipLaunchKernelGGL(masked_max_fwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, mask, window, out,
                       argmax, (long)R, L, D / 4);
    STAGE_LAUNCH_CHECK();
    return 0;
}

extern "C" int stage_masked_max_bwd(const float* dout, const int* argmax, const float* mask, float* dx, long long R,
                                    int L, int D, int accumulate, void* stream) {
    if (R <= 0) return 0;
    if (D % 4 != 0 || L < 1) return STAGE_ERR_SHAPE;
    const int grid = stage_grid_for(R * L * (D / 4), 256, GRID_CAP * 8);
    hipLaunchKernelGGL(masked_max_bwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, dout, argmax, mask, dx,
                       (long)R, L, D / 4, accumulate);
    STAGE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// LayerNorm whose output only feeds a masked max over the sequence axis (the classifier head: final_layer_norm of the
// cls_encoder -> mask_logits + max over the Lqa words, model/stage.py:503, model/encoder.py:52): one pass, the normalised
// (R, L, K) tensor is never written.  One wave per group of L rows; 32 lanes per row (K = 128), two rows per step, eight rows
// in flight.  Same arithmetic as ln_fwd_fast_kernel / masked_max_fwd_kernel (first maximum wins).
//     v = x + res ; sum_out = v ; y = (v - mean) * rstd * gamma + beta ; out[g, d] = max_l ( y * m + (1 - m) * NEG )
// Backward: ln_bwd_fast_kernel<..., MM = true> (the row gradient is gathered from (dout, argmax) on the fly).
// ------------------------------------------------------------------------------------------------
// RAG (ragged token rows): group s = seq[s] = (first compact row, length, mask row g, output row): the Lseq rows of one (example, candidate,
// live frame), mask = qmask[g * Lq + l] (the QA word mask), result written to the DENSE row seq[s].w of out / idx (the caller pre-fills
// the rows of dead frames with -1e10 / 0, which is what the dense kernel computes for an all-masked group)
template <typename T, bool RAG = false>
__global__ __launch_bounds__(256) void ln_mm_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                        T* __restrict__ sum_out, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ mask,
                                                        T* __restrict__ out, int* __restrict__ idx,
                                                        float* __restrict__ mean, float* __restrict__ rstd, long R, int L_in,
                                                        float eps, const int4* __restrict__ seq = nullptr, int Lq = 0) {
    constexpr int K = 128;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int sub = lane >> 5, sl = lane & 31;
    const float4 gm = ld4(gamma + 4 * sl), bt = ld4(beta + 4 * sl);
    for (long grp0 = (long)blockIdx.x * wpb + wave; grp0 < R; grp0 += (long)gridDim.x * wpb) {
        float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        int4 bi = make_int4(-1, -1, -1, -1);
        long row0 = grp0 * L_in, grp = grp0;
        int L = L_in;
        const float* mrow = mask + row0;
        if (RAG) {
            const int4 sq = seq[grp0];
            row0 = sq.x;
            L = sq.y;
            mrow = mask + (long)sq.z * Lq;
            grp = sq.w;
        }
        for (int l0 = 0; l0 < L; l0 += 8) {
            float4 v[4], rv[4];
            float mk[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int l = min(l0 + 2 * u + sub, L - 1);
                v[u] = ldv4s(x + (row0 + l) * K + 4 * sl);
                if (res) rv[u] = ldv4s(res + (row0 + l) * K + 4 * sl);
                mk[u] = mrow[l];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int l = l0 + 2 * u + sub;
                const bool ok = l < L;
                if (res) {
                    v[u] = f4add(v[u], rv[u]);
                    if (ok) stv4(sum_out + (row0 + l) * K + 4 * sl, v[u]);
                }
                const float mu = group_sum(f4hsum(v[u]), 32) * (1.0f / K);   // (16-bit storage: statistics of the unrounded sum, as in ln_fwd_fast_kernel)
                const float4 dd = make_float4(v[u].x - mu, v[u].y - mu, v[u].z - mu, v[u].w - mu);
                const float q = group_sum(f4hsum(f4mul(dd, dd)), 32);
                const float rs = 1.0f / sqrtf(q * (1.0f / K) + eps);
                if (ok && sl == 0) {
                    mean[row0 + l] = mu;
                    rstd[row0 + l] = rs;
                }
                if (ok) {
                    const float off = (1.0f - mk[u]) * STAGE_NEG;
                    const float4 w = make_float4(((v[u].x - mu) * rs * gm.x + bt.x) * mk[u] + off, ((v[u].y - mu) * rs * gm.y + bt.y) * mk[u] + off,
                                                 ((v[u].z - mu) * rs * gm.z + bt.z) * mk[u] + off, ((v[u].w - mu) * rs * gm.w + bt.w) * mk[u] + off);
                    if (w.x > best.x) { best.x = w.x; bi.x = l; }
                    if (w.y > best.y) { best.y = w.y; bi.y = l; }
                    if (w.z > best.z) { best.z = w.z; bi.z = l; }
                    if (w.w > best.w) { best.w = w.w; bi.w = l; }
                }
            }
        }
        // the two row parities: larger value wins, on a tie the smaller row (= the first maximum in row order)
#define LN_MM_MERGE(C)                                                                                                  \
    {                                                                                                                   \
        const float ov = __shfl_xor(best.C, 32);                                                                        \
        const int oi = __shfl_xor(bi.C, 32);                                                                            \
        if (oi >= 0 && (bi.C < 0 || ov > best.C || (ov == best.C && oi < bi.C))) { best.C = ov; bi.C = oi; }            \
    }
        LN_MM_MERGE(x) LN_MM_MERGE(y) LN_MM_MERGE(z) LN_MM_MERGE(w)
#undef LN_MM_MERGE
        if (sub == 0) {
            stv4(out + grp * K + 4 * sl, best);
            *reinterpret_cast<int4*>(idx + grp * K + 4 * sl) = bi;
        }
    }
}

// keep bits of the dropout stream that stage_layernorm_fwd (p_drop, seed) applied to a (rows, K) output: word-major
// [ceil(K/32)][rows], bit b of word w <=> element 32 w + b kept (the format of the ReLU masks of the GEMMs)
__global__ __launch_bounds__(256) void dropout_keepmask_kernel(unsigned* __restrict__ mask, long rows, int K, uint64_t seed, uint32_t th) {
    const int K4 = K >> 2, NW = (K + 31) >> 5;
    const long total = rows * NW;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int w = (int)(e / rows);
        const long row = e - (long)w * rows;
        unsigned bits = 0u;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int jq = 8 * w + q;
            if (jq < K4) bits |= drop4_bits(seed, (uint64_t)row * K4 + jq, th) << (4 * q);
        }
        mask[e] = bits;
    }
}
extern "C" int stage_dropout_keepmask(float p_drop, unsigned long long seed, unsigned* mask, long long rows, int K, void* stream) {
    if (rows <= 0) return 0;
    if (K % 4 != 0 || !(p_drop > 0.f)) return STAGE_ERR_SHAPE;
    const long total = rows * ((K + 31) / 32);
    const int grid = stage_grid_for(total, 256, GRID_CAP * 8);
    hipLaunchKernelGGL(dropout_keepmask_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, mask, (long)rows, K, (uint64_t)seed,
                       drop_thresh16(p_drop));
    STAGE_LAUNCH_CHECK();
    return 0;
}

extern "C" int stage_ln_masked_max_supported(int L, int K) { return (K == 128 && L >= 1) ? 1 : 0; }

// Ragged token rows: S groups seq[s] = (first row, length, mask row, output row) over compact rows; qmask (G, Lq); out / argmax are
// DENSE (the caller pre-fills them for the groups that do not exist: stage_rag_fill_pooled).  K == 128.
extern "C" int stage_ln_masked_max_rag_fwd(const float* x, const float* res, float* sum_out, const float* gamma, const float* beta,
                                           const float* qmask, float* out, int* argmax, float* mean, float* rstd, const int* seq,
                                           long long S, int Lq, int K, float eps, void* stream) {
    if (S <= 0) return 0;
    if (K != 128 || Lq < 1 || (res && !sum_out)) return STAGE_ERR_SHAPE;
    const int grid = stage_grid_for(S, 4, GRID_CAP * 8);
    hipLaunchKernelGGL((ln_mm_fwd_kernel<float, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, res, sum_out, gamma, beta, qmask,
                       out, argmax, mean, rstd, (long)S, 0, eps, (const int4*)seq, Lq);
    STAGE_LAUNCH_CHECK();
    return 0;
}
// dout / argmax dense (as written by the forward), xin / mean / rstd / dx compact (rows), rowinfo (rows, 4) from stage_rag_rowinfo
extern "C" int stage_ln_masked_max_rag_bwd(const float* dout, const int* argmax, const float* qmask, const float* xin, const float* mean,
                                           const float* rstd, const float* gamma, float* dx, float* dgamma, float* dbeta,
                                           const int* rowinfo, long long rows, int K, void* ws, size_t ws_bytes, void* stream) {
    if (K != 128) return STAGE_ERR_SHAPE;
    if (ws_bytes < stage_ln_bwd_ws_bytes(K)) return STAGE_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (rows <= 0) {
        (void)hipMemsetAsync(dgamma, 0, sizeof(float) * K, st);
        (void)hipMemsetAsync(dbeta, 0, sizeof(float) * K, st);
        return 0;
    }
    if (rows >= (1ll << 31)) return STAGE_ERR_SHAPE;
    const int LPR = ln_lpr(K / 4);
    const int rows_per_block = 4 * (64 / LPR);
    const int grid = stage_grid_for(rows, rows_per_block * 8, PART_CAP);
    const size_t lds = (size_t)rows_per_block * 2 * K * sizeof(float);
    float* part = (float*)ws;
    RowSrcT<float> src{xin, nullptr, 0, 1, 0, nullptr};
    if ((K / 4 + LPR - 1) / LPR != 1) return STAGE_ERR_SHAPE;
    hipLaunchKernelGGL((ln_bwd_fast_kernel<0, false, 1, float, float, true>), dim3(grid), dim3(256), lds, st, src, dout, mean, rstd,
                       gamma, dx, (float*)nullptr, part, (long)rows, K, LPR, (uint64_t)0, 0u, 1.0f, argmax, qmask, 1, (const int4*)rowinfo);
    STAGE_LAUNCH_CHECK();
    stage_colreduce(part, dgamma, dbeta, grid, (long)2 * K, 2 * K, K, 1, st);
    STAGE_LAUNCH_CHECK();
    return 0;
}

extern "C" int stage_ln_masked_max_fwd(const float* x, const float* res, float* sum_out, const float* gamma, const float* beta,
                                       const float* mask, float* out, int* argmax, float* mean, float* rstd, long long R, int L,
                                       int K, float eps, void* stream) {
    if (R <= 0) return 0;
    if (!stage_ln_masked_max_supported(L, K) || (res && !sum_out)) return STAGE_ERR_SHAPE;
    const int grid = stage_grid_for(R, 4, GRID_CAP * 8);
    hipLaunchKernelGGL(ln_mm_fwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, res, sum_out, gamma, beta, mask, out,
                       argmax, mean, rstd, (long)R, L, eps);
    STAGE_LAUNCH_CHECK();
    return 0;
}

// xin = x + res as saved by the forward; dx is the gradient of both x and res
template <typename T>
static int ln_masked_max_bwd_t(const T* dout, const int* argmax, const float* mask, const T* xin, const float* mean,
                               const float* rstd, const float* gamma, T* dx, float* dgamma, float* dbeta, long long R, int L, int K,
                               void* ws, size_t ws_bytes, void* stream) {
    if (!stage_ln_masked_max_supported(L, K)) return STAGE_ERR_SHAPE;
    if (ws_bytes < stage_ln_bwd_ws_bytes(K)) return STAGE_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const long long rows = R * L;
    if (rows <= 0) {
        (void)hipMemsetAsync(dgamma, 0, sizeof(float) * K, st);
        (void)hipMemsetAsync(dbeta, 0, sizeof(float) * K, st);
        return 0;
    }
    if (rows >= (1ll << 31)) return STAGE_ERR_SHAPE;
    const int LPR = ln_lpr(K / 4);
    const int rows_per_block = 4 * (64 / LPR);
    const int grid = stage_grid_for(rows, rows_per_block * 8, PART_CAP);
    const size_t lds = (size_t)rows_per_block * 2 * K * sizeof(float);
    float* part = (float*)ws;
    RowSrcT<T> src{xin, nullptr, 0, 1, 0, nullptr};
    if ((K / 4 + LPR - 1) / LPR != 1) return STAGE_ERR_SHAPE;
    hipLaunchKernelGGL((ln_bwd_fast_kernel<0, false, 1, T, T, true>), dim3(grid), dim3(256), lds, st, src, dout, mean, rstd,
                       gamma, dx, (T*)nullptr, part, (long)rows, K, LPR, (uint64_t)0, 0u, 1.0f, argmax, mask, L);
    STAGE_LAUNCH_CHECK();
    stage_colreduce(part, dgamma, dbeta, grid, (long)2 * K, 2 * K, K, 1, st);
    STAGE_LAUNCH_CHECK();
    return 0;
}
extern "C" int stage_ln_masked_max_bwd(const float* dout, const int* argmax, const float* mask, const float* xin,
                                       const float* mean, const float* rstd, const float* gamma, float* dx, float* dgamma,
                                       float* dbeta, long long R, int L, int K, void* ws, size_t ws_bytes, void* stream) {
    return ln_masked_max_bwd_t<float>(dout, argmax, mask, xin, mean, rstd, gamma, dx, dgamma, dbeta, R, L, K, ws, ws_bytes, stream);
}

// ------------------------------------------------------------------------------------------------
// bf16 storage mode (BASELINE.json configs[4]): the same generic kernels instantiated on 16-bit activations.  Statistics,
// affine parameters, masks, arg-max indices, parameter gradients and every intermediate stay fp32; only the tensors that
// stream through HBM between kernels are bf16.  Pointers to bf16 data cross the C ABI as void*.
// ------------------------------------------------------------------------------------------------
typedef stage_bf16 B16;

extern "C" int stage_layernorm_fwd_bf16(const void* x, const void* res, int res_period, void* sum_out, const float* gamma,
                                        const float* beta, void* y, float* mean, float* rstd, long long rows, int K,
                                        float eps, float p_drop, unsigned long long seed, void* stream) {
    if (rows <= 0) return 0;
    if (K % 4 != 0 || K > 4 * MAXV * 64) return STAGE_ERR_SHAPE;
    RowSrcT<B16> src{(const B16*)x, (const B16*)res, 0, 1, res_period, (B16*)sum_out};
    return ln_fwd_launch<0, B16>(src, gamma, beta, (B16*)y, mean, rstd, rows, K, ln_lpr(K / 4), eps, p_drop, seed, (hipStream_t)stream);
}

extern "C" int stage_layernorm_bwd_bf16(const void* dy, const void* x, const float* mean, const float* rstd,
                                        const float* gamma, void* dx, const void* dx_add, float* dgamma, float* dbeta,
                                        long long rows, int K, float p_drop, unsigned long long seed, void* ws,
                                        size_t ws_bytes, void* stream) {
    if (K % 4 != 0 || K > 4 * MAXV * 64) return STAGE_ERR_SHAPE;
    RowSrcT<B16> src{(const B16*)x, (const B16*)dx_add, 0, 1, 0, nullptr};
    return ln_bwd_launch<0, B16, B16>(src, (const B16*)dy, mean, rstd, gamma, (B16*)dx, (B16*)nullptr, dgamma, dbeta, rows, K,
                                   ln_lpr(K / 4), p_drop, seed, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int stage_cat3_layernorm_fwd_bf16(const void* a, const void* b, const float* gamma, const float* beta, void* y,
                                             float* mean, float* rstd, long long rows, int D, int rep, int inner, float eps,
                                             float p_drop, unsigned long long seed, void* stream) {
    if (rows <= 0) return 0;
    if (D % 4 != 0 || D > 256 || rep < 1 || inner < 1) return STAGE_ERR_SHAPE;
    RowSrcT<B16> src{(const B16*)a, (const B16*)b, D, rep, inner, nullptr};
    return ln_fwd_launch<1, B16>(src, gamma, beta, (B16*)y, mean, rstd, rows, 3 * D, ln_lpr(D / 4), eps, p_drop, seed,
                              (hipStream_t)stream);
}

// da_full[rows, D] stays fp32 (scratch, reduced over `rep` by stage_reduce_rep), db[rows, D] is bf16
extern "C" int stage_cat3_layernorm_bwd_bf16(const void* dy, const void* a, const void* b, const float* mean,
                                             const float* rstd, const float* gamma, float* da_full, void* db, float* dgamma,
                                             float* dbeta, long long rows, int D, int rep, int inner, float p_drop,
                                             unsigned long long seed, void* ws, size_t ws_bytes, void* stream) {
    if (D % 4 != 0 || D > 256 || rep < 1 || inner < 1) return STAGE_ERR_SHAPE;
    RowSrcT<B16> src{(const B16*)a, (const B16*)b, D, rep, inner, nullptr};
    return ln_bwd_launch<1, B16, float>(src, (const B16*)dy, mean, rstd, gamma, da_full, (B16*)db, dgamma, dbeta, rows, 3 * D,
                                     ln_lpr(D / 4), p_drop, seed, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int stage_l2norm_fwd_bf16(const void* x, void* y, float* norm_out, long long rows, int K, float eps, float p_drop,
                                     unsigned long long seed, void* stream) {
    if (rows <= 0) return 0;
    if (K % 4 != 0 || K > 4 * MAXV * 64) return STAGE_ERR_SHAPE;
    const int LPR = ln_lpr(K / 4);
    const int grid = stage_grid_for(rows, 4 * (64 / LPR), GRID_CAP * 4);
    hipStream_t st = (hipStream_t)stream;
    if (p_drop > 0.f)
        hipLaunchKernelGGL((l2norm_fwd_kernel<true, B16>), dim3(grid), dim3(256), 0, st, (const B16*)x, (B16*)y, norm_out,
                           (long)rows, K, eps, LPR, (uint64_t)seed, drop_thresh16(p_drop), 1.0f / (1.0f - p_drop));
    else
        hipLaunchKernelGGL((l2norm_fwd_kernel<false, B16>), dim3(grid), dim3(256), 0, st, (const B16*)x, (B16*)y, norm_out,
                           (long)rows, K, eps, LPR, (uint64_t)0, 0u, 1.0f);
    STAGE_LAUNCH_CHECK();
    return 0;
}

extern "C" int stage_l2norm_bwd_bf16(const void* dy, const void* x, void* dx, long long rows, int K, float eps, float p_drop,
                                     unsigned long long seed, int accumulate, void* stream) {
    if (rows <= 0) return 0;
    if (K % 4 != 0 || K > 4 * MAXV * 64) return STAGE_ERR_SHAPE;
    const int LPR = ln_lpr(K / 4);
    const int grid = stage_grid_for(rows, 4 * (64 / LPR), GRID_CAP * 4);
    hipStream_t st = (hipStream_t)stream;
    if (p_drop > 0.f)
        hipLaunchKernelGGL((l2norm_bwd_kernel<true, B16>), dim3(grid), dim3(256), 0, st, (const B16*)dy, (const B16*)x, (B16*)dx,
                           (long)rows, K, eps, LPR, (uint64_t)seed, drop_thresh16(p_drop), 1.0f / (1.0f - p_drop), accumulate);
    else
        hipLaunchKernelGGL((l2norm_bwd_kernel<false, B16>), dim3(grid), dim3(256), 0, st, (const B16*)dy, (const B16*)x, (B16*)dx,
                           (long)rows, K, eps, LPR, (uint64_t)0, 0u, 1.0f, accumulate);
    STAGE_LAUNCH_CHECK();
    return 0;
}

extern "C" int stage_masked_max_fwd_bf16(const void* x, const float* mask, const int* window, void* out, int* argmax,
                                         long long R, int L, int D, void* stream) {
    if (R <= 0) return 0;
    if (D % 4 != 0 || L < 1) return STAGE_ERR_SHAPE;
    const int grid = stage_grid_for(R * (D / 4), 256, GRID_CAP * 8);
    hipLaunchKernelGGL(masked_max_fwd_kernel<B16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const B16*)x, mask, window,
                       (B16*)out, argmax, (long)R, L, D / 4);
    STAGE_LAUNCH_CHECK();
    return 0;
}

extern "C" int stage_masked_max_bwd_bf16(const void* dout, const int* argmax, const float* mask, void* dx, long long R, int L,
                                         int D, int accumulate, void* stream) {
    if (R <= 0) return 0;
    if (D % 4 != 0 || L < 1) return STAGE_ERR_SHAPE;
    const int grid = stage_grid_for(R * L * (D / 4), 256, GRID_CAP * 8);
    hipLaunchKernelGGL(masked_max_bwd_kernel<B16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const B16*)dout, argmax, mask,
                       (B16*)dx, (long)R, L, D / 4, accumulate);
    STAGE_LAUNCH_CHECK();
    return 0;
}

// da (rows / rep, D) comes back in fp32 (it is a sum over the `rep` frames), db (rows, D) in bf16
extern "C" int stage_cat3_layernorm_bwd_reduced_bf16(const void* dy, const void* a, const void* b, const float* mean,
                                                     const float* rstd, const float* gamma, float* da, void* db,
                                                     float* dgamma, float* dbeta, long long rows, int D, int rep, int inner,
                                                     float p_drop, unsigned long long seed, void* ws, size_t ws_bytes,
                                                     void* stream) {
    return cat3_bwd_reduced_t<B16>((const B16*)dy, (const B16*)a, (const B16*)b, mean, rstd, gamma, da, (B16*)db, dgamma, dbeta,
                                   rows, D, rep, inner, p_drop, seed, ws, ws_bytes, stream);
}

// dx (bf16) = add (fp32, may be NULL) + l2norm-backward(dy (fp32), x (bf16)): the attention backward leaves its two gradients
// w.r.t. the raw and the normalised rows in fp32; this folds them into ONE bf16 gradient without an fp32 copy of x, an fp32
// accumulate pass and a cast pass (rows x K each)
extern "C" int stage_l2norm_bwd_mixed_bf16(const float* dy, const void* x, const float* add, void* dx, long long rows, int K,
                                           float eps, float p_drop, unsigned long long seed, void* stream) {
    if (rows <= 0) return 0;
    if (K % 4 != 0 || K > 4 * MAXV * 64) return STAGE_ERR_SHAPE;
    const int LPR = ln_lpr(K / 4);
    const int grid = stage_grid_for(rows, 4 * (64 / LPR), GRID_CAP * 4);
    hipStream_t st = (hipStream_t)stream;
    if (p_drop > 0.f)
        hipLaunchKernelGGL((l2norm_bwd_kernel<true, B16, B16, float>), dim3(grid), dim3(256), 0, st, dy, (const B16*)x, (B16*)dx,
                           (long)rows, K, eps, LPR, (uint64_t)seed, drop_thresh16(p_drop), 1.0f / (1.0f - p_drop), 0, add);
    else
        hipLaunchKernelGGL((l2norm_bwd_kernel<false, B16, B16, float>), dim3(grid), dim3(256), 0, st, dy, (const B16*)x, (B16*)dx,
                           (long)rows, K, eps, LPR, (uint64_t)0, 0u, 1.0f, 0, add);
    STAGE_LAUNCH_CHECK();
    return 0;
}

extern "C" int stage_ln_masked_max_fwd_bf16(const void* x, const void* res, void* sum_out, const float* gamma, const float* beta,
                                            const float* mask, void* out, int* argmax, float* mean, float* rstd, long long R,
                                            int L, int K, float eps, void* stream) {
    if (R <= 0) return 0;
    if (!stage_ln_masked_max_supported(L, K) || (res && !sum_out)) return STAGE_ERR_SHAPE;
    const int grid = stage_grid_for(R, 4, GRID_CAP * 8);
    hipLaunchKernelGGL(ln_mm_fwd_kernel<B16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const B16*)x, (const B16*)res,
                       (B16*)sum_out, gamma, beta, mask, (B16*)out, argmax, mean, rstd, (long)R, L, eps);
    STAGE_LAUNCH_CHECK();
    return 0;
}
extern "C" int stage_ln_masked_max_bwd_bf16(const void* dout, const int* argmax, const float* mask, const void* xin,
                                            const float* mean, const float* rstd, const float* gamma, void* dx, float* dgamma,
                                            float* dbeta, long long R, int L, int K, void* ws, size_t ws_bytes, void* stream) {
    return ln_masked_max_bwd_t<B16>((const B16*)dout, argmax, mask, (const B16*)xin, mean, rstd, gamma, (B16*)dx, dgamma, dbeta, R, L,
                                    K, ws, ws_bytes, stream);
}
