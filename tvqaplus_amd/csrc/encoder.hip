// Encoder-block streaming pieces (model/encoder.py:29-52, model/cnn.py:37-47, model/position_encoding.py:33-43):
//   * x + pe[:L]                                     (position table broadcast over the leading axes)
//   * depthwise Conv1d(k, groups=D, zero padding k//2) along L of (M, L, D) tensors, forward and backward
// The 1x1 pointwise conv + ReLU + residual is stage_gemm_nt's epilogue; the LayerNorms are rowops.hip.
// Padded sequence positions are NOT masked anywhere here (bug-compatible with the reference).
#include <stdlib.h>
#include "common.h"
#include "../../include/stage_hip.h"

#define KMAX 9
#define GRID_CAP 4096
#define DW_PART_CAP 512

__global__ __launch_bounds__(256) void add_pe_kernel(const float* __restrict__ x, const float* __restrict__ pe,
                                                     float* __restrict__ y, long rows, int L, int D4) {
    const long total = rows * D4;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long row = e / D4;
        const int q = (int)(e % D4);
        const int l = (int)(row % L);
        st4(y + e * 4, f4add(ld4(x + e * 4), ld4(pe + ((long)l * D4 + q) * 4)));
    }
}

extern "C" int stage_add_pe(const float* x, const float* pe, float* y, long long M, int L, int D, void* stream) {
    if (M <= 0) return 0;
    if (D % 4 != 0) return STAGE_ERR_SHAPE;
    const int grid = stage_grid_for(M * L * (D / 4), 256, GRID_CAP);
    hipLaunchKernelGGL(add_pe_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, pe, y, (long)(M * L), L, D / 4);
    STAGE_LAUNCH_CHECK();
    return 0;
}

// out[m,l,:] = bias + sum_t w[:,t] * in[m, l+t-pad, :]           w: (D,1,k) as stored by nn.Conv1d
template <typename T>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const T* __restrict__ in, const float* __restrict__ w,
                                                         const float* __restrict__ bias, T* __restrict__ out,
                                                         long rows, int L, int D, int k) {
    extern __shared__ __attribute__((aligned(16))) float wT[];  // [k][D] + bias[D]
    for (int i = threadIdx.x; i < k * D; i += blockDim.x) {
        const int t = i / D, d = i % D;
        wT[i] = w[d * k + t];
    }
    for (int d = threadIdx.x; d < D; d += blockDim.x) wT[k * D + d] = bias[d];
    __syncthreads();
    const int D4 = D >> 2, pad = k >> 1;
    const long total = rows * D4;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long row = e / D4;
        const int q = (int)(e % D4);
        const int l = (int)(row % L);
        float4 acc = ld4(&wT[k * D + 4 * q]);
        for (int t = 0; t < k; t++) {
            const int ll = l + t - pad;
            if (ll >= 0 && ll < L) {
                const float4 v = ldv4(in + ((row + (t - pad)) * D4 + q) * 4);
                acc = f4add(acc, f4mul(v, ld4(&wT[t * D + 4 * q])));
            }
        }
        stv4(out + e * 4, acc);
    }
}

// din[m,l,:] = sum_t w[:,t] * dout[m, l-t+pad, :] ; partial dw[t][d], db[d] per block.
template <typename T>
__global__ __launch_bounds__(256) void dwconv_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ in,
                                                         const float* __restrict__ w, T* __restrict__ din,
                                                         float* __restrict__ part, long rows, int L, int D, int k) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // wT [k][D]  then reduction scratch
    for (int i = threadIdx.x; i < k * D; i += blockDim.x) {
        const int t = i / D, d = i % D;
        sm[i] = w[d * k + t];
    }
    __syncthreads();
    const int D4 = D >> 2, pad = k >> 1;
    const int rpi = blockDim.x / D4;  // rows per iteration of this block
    const int q = threadIdx.x % D4, rsub = threadIdx.x / D4;
    const bool active = rsub < rpi;
    float4 aw[KMAX], ab = f4zero();
#pragma unroll
    for (int t = 0; t < KMAX; t++) aw[t] = f4zero();
    if (active) {
        for (long row = (long)blockIdx.x * rpi + rsub; row < rows; row += (long)gridDim.x * rpi) {
            const int l = (int)(row % L);
            const float4 go = ldv4(dout + (row * D4 + q) * 4);
            ab = f4add(ab, go);
            float4 gi = f4zero();
#pragma unroll
            for (int t = 0; t < KMAX; t++) {
                if (t < k) {
                    const int lf = l + t - pad;  // forward tap: out[l] uses in[l+t-pad]
                    if (lf >= 0 && lf < L)
                        aw[t] = f4add(aw[t], f4mul(go, ldv4(in + ((row + (t - pad)) * D4 + q) * 4)));
                    const int lb = l - t + pad;  // din[l] collects dout[l-t+pad] * w[t]
                    if (lb >= 0 && lb < L)
                        gi = f4add(gi, f4mul(ldv4(dout + ((row - t + pad) * D4 + q) * 4), ld4(&sm[t * D + 4 * q])));
                }
            }
            stv4(din + (row * D4 + q) * 4, gi);
        }
    }
    __syncthreads();
    // block reduce: scratch [rpi][(k+1)][D]
    float* red = sm;
    if (active) {
#pragma unroll
        for (int t = 0; t < KMAX; t++)
            if (t < k) st4(&red[((size_t)rsub * (k + 1) + t) * D + 4 * q], aw[t]);
        st4(&red[((size_t)rsub * (k + 1) + k) * D + 4 * q], ab);
    }
    __syncthreads();
    const int C = (k + 1) * D;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.f;
        for (int r = 0; r < rpi; r++) s += red[(size_t)r * C + c];
        part[(size_t)blockIdx.x * C + c] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// Sliding-window versions (the ones that run): a thread owns one float4 column of one chunk of <= 48 positions of one
// sequence and walks it, keeping the k taps of `in` (and of `dout` in the backward) in registers: every element is
// loaded once per chunk (+ a halo of k-1 rows) by straight-line loads from clamped addresses, instead of 2k+1 guarded
// loads per output element.  Same arithmetic order over the taps as the kernels above.
// ------------------------------------------------------------------------------------------------

template <int KT, typename T = float>
__global__ __launch_bounds__(256) void dwconv_fwd_sw_kernel(const T* __restrict__ in, const float* __restrict__ w,
                                                            const float* __restrict__ bias, T* __restrict__ out,
                                                            long M, int L, int D, int clen) {
    constexpr int pad = KT / 2;
    const int D4 = D >> 2, rpi = blockDim.x / D4;
    const int q = threadIdx.x % D4, rsub = threadIdx.x / D4;
    if (rsub >= rpi) return;
    float4 wt[KT];
#pragma unroll
    for (int t = 0; t < KT; t++)
        wt[t] = make_float4(w[(4 * q + 0) * KT + t], w[(4 * q + 1) * KT + t], w[(4 * q + 2) * KT + t], w[(4 * q + 3) * KT + t]);
    const float4 bq = ld4(bias + 4 * q);
    const int chunks = (L + clen - 1) / clen;
    const long items = M * chunks;
    for (long it = (long)blockIdx.x * rpi + rsub; it < items; it += (long)gridDim.x * rpi) {
        const long m = it / chunks;
        const int l0 = (int)(it % chunks) * clen, l1 = min(L, l0 + clen);
        const T* base = in + (m * L) * D + 4 * q;
        float4 win[KT];   // win[t] = in[l + t - pad] for the current l
#pragma unroll
        for (int t = 0; t < KT - 1; t++) {
            const int ll = l0 + t - pad;
            const float4 v = ldv4(base + (long)min(max(ll, 0), L - 1) * D);
            win[t + 1] = (ll >= 0 && ll < L) ? v : f4zero();
        }
#pragma unroll 4
        for (int l = l0; l < l1; l++) {
#pragma unroll
            for (int t = 0; t < KT - 1; t++) win[t] = win[t + 1];
            const int ll = l + pad;
            const float4 v = ldv4(base + (long)min(ll, L - 1) * D);
            win[KT - 1] = ll < L ? v : f4zero();
            float4 acc = bq;
#pragma unroll
            for (int t = 0; t < KT; t++) acc = f4add(acc, f4mul(win[t], wt[t]));
            stv4s(out + ((m * L + l) * D + 4 * q), acc);
        }
    }
}

template <int KT, typename T = float>
__global__ __launch_bounds__(256) void dwconv_bwd_sw_kernel(const T* __restrict__ dout, const T* __restrict__ in,
                                                            const float* __restrict__ w, T* __restrict__ din,
                                                            float* __restrict__ part, long M, int L, int D, int clen) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // reduction scratch [rpi][KT + 1][D]
    constexpr int pad = KT / 2;
    const int D4 = D >> 2, rpi = blockDim.x / D4;
    const int q = threadIdx.x % D4, rsub = threadIdx.x / D4;
    const bool active = rsub < rpi;
    float4 aw[KT], ab = f4zero();
#pragma unroll
    for (int t = 0; t < KT; t++) aw[t] = f4zero();
    if (active) {
        float4 wt[KT];
#pragma unroll
        for (int t = 0; t < KT; t++)
            wt[t] = make_float4(w[(4 * q + 0) * KT + t], w[(4 * q + 1) * KT + t], w[(4 * q + 2) * KT + t], w[(4 * q + 3) * KT + t]);
        const int chunks = (L + clen - 1) / clen;
        const long items = M * chunks;
        for (long it = (long)blockIdx.x * rpi + rsub; it < items; it += (long)gridDim.x * rpi) {
            const long m = it / chunks;
            const int l0 = (int)(it % chunks) * clen, l1 = min(L, l0 + clen);
            const T* bi = in + (m * L) * D + 4 * q;
            const T* bo = dout + (m * L) * D + 4 * q;
            float4 wi[KT], wo[KT];   // wi[t] = in[l + t - pad], wo[t] = dout[l + t - pad]
#pragma unroll
            for (int t = 0; t < KT - 1; t++) {
                const int ll = l0 + t - pad;
                const long ro = (long)min(max(ll, 0), L - 1) * D;
                const float4 vi = ldv4(bi + ro), vo = ldv4(bo + ro);
                const bool ok = ll >= 0 && ll < L;
                wi[t + 1] = ok ? vi : f4zero();
                wo[t + 1] = ok ? vo : f4zero();
            }
#pragma unroll 2
            for (int l = l0; l < l1; l++) {
#pragma unroll
                for (int t = 0; t < KT - 1; t++) { wi[t] = wi[t + 1]; wo[t] = wo[t + 1]; }
                const int ll = l + pad;
                const long ro = (long)min(ll, L - 1) * D;
                const float4 vi = ldv4(bi + ro), vo = ldv4(bo + ro);
                wi[KT - 1] = ll < L ? vi : f4zero();
                wo[KT - 1] = ll < L ? vo : f4zero();
                const float4 go = wo[pad];
                ab = f4add(ab, go);
                float4 gi = f4zero();
#pragma unroll
                for (int t = 0; t < KT; t++) {
                    aw[t] = f4add(aw[t], f4mul(go, wi[t]));                 // dw[t] += dout[l] * in[l + t - pad]
                    gi = f4add(gi, f4mul(wo[KT - 1 - t], wt[t]));           // din[l] += dout[l - t + pad] * w[t]
                }
                stv4s(din + ((m * L + l) * D + 4 * q), gi);
            }
        }
    }
    // block reduce: scratch [rpi][(KT+1)][D]
    if (active) {
#pragma unroll
        for (int t = 0; t < KT; t++) st4(&sm[((size_t)rsub * (KT + 1) + t) * D + 4 * q], aw[t]);
        st4(&sm[((size_t)rsub * (KT + 1) + KT) * D + 4 * q], ab);
    }
    __syncthreads();
    const int C = (KT + 1) * D;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.f;
        for (int r = 0; r < rpi; r++) s += sm[(size_t)r * C + c];
        part[(size_t)blockIdx.x * C + c] = s;
    }
}

// dw[d*k + t] = sum_b part[b][t][d] ; db[d] = sum_b part[b][k][d]
__global__ void dwconv_final_kernel(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ db,
                                    int nb, int D, int k) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int C = (k + 1) * D;
    if (c >= C) return;
    float s = 0.f;
    for (int b = 0; b < nb; b++) s += part[(size_t)b * C + c];
    const int t = c / D, d = c % D;
    if (t < k) dw[d * k + t] = s;
    else db[d] = s;
}

template <typename T>
static int dwconv_fwd_t(const T* in, const float* w, const float* bias, T* out, long long M, int L, int D, int k, void* stream) {
    if (M <= 0) return 0;
    if (D % 4 != 0 || k < 1 || k > KMAX || (k & 1) == 0) return STAGE_ERR_SHAPE;
    if (D / 4 <= 256 && !getenv("STAGE_DWCONV_GENERIC")) {
        const int rpi = 256 / (D / 4);
        const int clen = stage_chunk_len(L);
        const long items = (long)M * ((L + clen - 1) / clen);
        const int gridw = stage_grid_for(items, rpi, GRID_CAP * 4);
        hipStream_t st = (hipStream_t)stream;
        switch (k) {
            case 1: hipLaunchKernelGGL((dwconv_fwd_sw_kernel<1, T>), dim3(gridw), dim3(256), 0, st, in, w, bias, out, (long)M, L, D, clen); break;
            case 3: hipLaunchKernelGGL((dwconv_fwd_sw_kernel<3, T>), dim3(gridw), dim3(256), 0, st, in, w, bias, out, (long)M, L, D, clen); break;
            case 5: hipLaunchKernelGGL((dwconv_fwd_sw_kernel<5, T>), dim3(gridw), dim3(256), 0, st, in, w, bias, out, (long)M, L, D, clen); break;
            case 7: hipLaunchKernelGGL((dwconv_fwd_sw_kernel<7, T>), dim3(gridw), dim3(256), 0, st, in, w, bias, out, (long)M, L, D, clen); break;
            default: hipLaunchKernelGGL((dwconv_fwd_sw_kernel<9, T>), dim3(gridw), dim3(256), 0, st, in, w, bias, out, (long)M, L, D, clen); break;
        }
        STAGE_LAUNCH_CHECK();
        return 0;
    }
    const int grid = stage_grid_for(M * L * (D / 4), 256, GRID_CAP);
    hipLaunchKernelGGL(dwconv_fwd_kernel<T>, dim3(grid), dim3(256), (size_t)(k + 1) * D * sizeof(float),
                       (hipStream_t)stream, in, w, bias, out, (long)(M * L), L, D, k);
    STAGE_LAUNCH_CHECK();
    return 0;
}

extern "C" int stage_dwconv_fwd(const float* in, const float* w, const float* bias, float* out, long long M, int L,
                                int D, int k, void* stream) {
    return dwconv_fwd_t<float>(in, w, bias, out, M, L, D, k, stream);
}

extern "C" size_t stage_dwconv_bwd_ws_bytes(int D, int k) { return (size_t)DW_PART_CAP * (k + 1) * D * sizeof(float); }

template <typename T>
static int dwconv_bwd_t(const T* dout, const T* in, const float* w, T* din, float* dw, float* db, long long M, int L, int D,
                        int k, void* ws, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (D % 4 != 0 || D / 4 > 256 || k < 1 || k > KMAX || (k & 1) == 0) return STAGE_ERR_SHAPE;
    if (ws_bytes < stage_dwconv_bwd_ws_bytes(D, k)) return STAGE_ERR_WORKSPACE;
    if (M <= 0) {
        (void)hipMemsetAsync(dw, 0, sizeof(float) * D * k, st);
        (void)hipMemsetAsync(db, 0, sizeof(float) * D, st);
        return 0;
    }
    const int rpi = 256 / (D / 4);
    int grid;
    size_t lds = (size_t)k * D;
    const size_t red = (size_t)rpi * (k + 1) * D;
    if (red > lds) lds = red;
    if (!getenv("STAGE_DWCONV_GENERIC")) {
        const int clen = stage_chunk_len(L);
        const long items = (long)M * ((L + clen - 1) / clen);
        grid = stage_grid_for(items, rpi, DW_PART_CAP);
        const size_t ldb = red * sizeof(float);
        switch (k) {
            case 1: hipLaunchKernelGGL((dwconv_bwd_sw_kernel<1, T>), dim3(grid), dim3(256), ldb, st, dout, in, w, din, (float*)ws, (long)M, L, D, clen); break;
            case 3: hipLaunchKernelGGL((dwconv_bwd_sw_kernel<3, T>), dim3(grid), dim3(256), ldb, st, dout, in, w, din, (float*)ws, (long)M, L, D, clen); break;
            case 5: hipLaunchKernelGGL((dwconv_bwd_sw_kernel<5, T>), dim3(grid), dim3(256), ldb, st, dout, in, w, din, (float*)ws, (long)M, L, D, clen); break;
            case 7: hipLaunchKernelGGL((dwconv_bwd_sw_kernel<7, T>), dim3(grid), dim3(256), ldb, st, dout, in, w, din, (float*)ws, (long)M, L, D, clen); break;
            default: hipLaunchKernelGGL((dwconv_bwd_sw_kernel<9, T>), dim3(grid), dim3(256), ldb, st, dout, in, w, din, (float*)ws, (long)M, L, D, clen); break;
        }
    } else {
        grid = stage_grid_for(M * L, rpi * 8, DW_PART_CAP);
        hipLaunchKernelGGL(dwconv_bwd_kernel<T>, dim3(grid), dim3(256), lds * sizeof(float), st, dout, in, w, din,
                           (float*)ws, (long)(M * L), L, D, k);
    }
    STAGE_LAUNCH_CHECK();
    stage_colreduce((const float*)ws, dw, db, grid, (long)(k + 1) * D, (k + 1) * D, D, k, st);
    STAGE_LAUNCH_CHECK();
    return 0;
}

extern "C" int stage_dwconv_bwd(const float* dout, const float* in, const float* w, float* din, float* dw, float* db,
                                long long M, int L, int D, int k, void* ws, size_t ws_bytes, void* stream) {
    return dwconv_bwd_t<float>(dout, in, w, din, dw, db, M, L, D, k, ws, ws_bytes, stream);
}

// bf16 storage mode (BASELINE.json configs[4]): the same kernels on 16-bit activations; weights, bias and their gradients
// stay fp32
extern "C" int stage_dwconv_fwd_bf16(const void* in, const float* w, const float* bias, void* out, long long M, int L, int D,
                                     int k, void* stream) {
    return dwconv_fwd_t<stage_bf16>((const stage_bf16*)in, w, bias, (stage_bf16*)out, M, L, D, k, stream);
}

extern "C" int stage_dwconv_bwd_bf16(const void* dout, const void* in, const float* w, void* din, float* dw, float* db,
                                     long long M, int L, int D, int k, void* ws, size_t ws_bytes, void* stream) {
    return dwconv_bwd_t<stage_bf16>((const stage_bf16*)dout, (const stage_bf16*)in, w, (stage_bf16*)din, dw, db, M, L, D, k, ws,
                                    ws_bytes, stream);
}
