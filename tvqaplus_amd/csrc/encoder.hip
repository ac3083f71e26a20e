// Encoder-block streaming pieces (model/encoder.py:29-52, model/cnn.py:37-47, model/position_encoding.py:33-43):
//   * x + pe[:L]                                     (position table broadcast over the leading axes)
//   * depthwise Conv1d(k, groups=D, zero padding k//2) along L of (M, L, D) tensors, forward and backward
// The 1x1 pointwise conv + ReLU + residual is stage_gemm_nt's epilogue; the LayerNorms are rowops.hip.
// Padded sequence positions are NOT masked anywhere here (bug-compatible with the reference).
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
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ out,
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
                const float4 v = ld4(in + ((row + (t - pad)) * D4 + q) * 4);
                acc = f4add(acc, f4mul(v, ld4(&wT[t * D + 4 * q])));
            }
        }
        st4(out + e * 4, acc);
    }
}

// din[m,l,:] = sum_t w[:,t] * dout[m, l-t+pad, :] ; partial dw[t][d], db[d] per block.
__global__ __launch_bounds__(256) void dwconv_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ in,
                                                         const float* __restrict__ w, float* __restrict__ din,
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
            const float4 go = ld4(dout + (row * D4 + q) * 4);
            ab = f4add(ab, go);
            float4 gi = f4zero();
#pragma unroll
            for (int t = 0; t < KMAX; t++) {
                if (t < k) {
                    const int lf = l + t - pad;  // forward tap: out[l] uses in[l+t-pad]
                    if (lf >= 0 && lf < L)
                        aw[t] = f4add(aw[t], f4mul(go, ld4(in + ((row + (t - pad)) * D4 + q) * 4)));
                    const int lb = l - t + pad;  // din[l] collects dout[l-t+pad] * w[t]
                    if (lb >= 0 && lb < L)
                        gi = f4add(gi, f4mul(ld4(dout + ((row - t + pad) * D4 + q) * 4), ld4(&sm[t * D + 4 * q])));
                }
            }
            st4(din + (row * D4 + q) * 4, gi);
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

extern "C" int stage_dwconv_fwd(const float* in, const float* w, const float* bias, float* out, long long M, int L,
                                int D, int k, void* stream) {
    if (M <= 0) return 0;
    if (D % 4 != 0 || k < 1 || k > KMAX || (k & 1) == 0) return STAGE_ERR_SHAPE;
    const int grid = stage_grid_for(M * L * (D / 4), 256, GRID_CAP);
    hipLaunchKernelGGL(dwconv_fwd_kernel, dim3(grid), dim3(256), (size_t)(k + 1) * D * sizeof(float),
                       (hipStream_t)stream, in, w, bias, out, (long)(M * L), L, D, k);
    STAGE_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t stage_dwconv_bwd_ws_bytes(int D, int k) { return (size_t)DW_PART_CAP * (k + 1) * D * sizeof(float); }

extern "C" int stage_dwconv_bwd(const float* dout, const float* in, const float* w, float* din, float* dw, float* db,
                                long long M, int L, int D, int k, void* ws, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (D % 4 != 0 || D / 4 > 256 || k < 1 || k > KMAX || (k & 1) == 0) return STAGE_ERR_SHAPE;
    if (ws_bytes < stage_dwconv_bwd_ws_bytes(D, k)) return STAGE_ERR_WORKSPACE;
    if (M <= 0) {
        (void)hipMemsetAsync(dw, 0, sizeof(float) * D * k, st);
        (void)hipMemsetAsync(db, 0, sizeof(float) * D, st);
        return 0;
    }
    const int rpi = 256 / (D / 4);
    const int grid = stage_grid_for(M * L, rpi * 8, DW_PART_CAP);
    size_t lds = (size_t)k * D;
    const size_t red = (size_t)rpi * (k + 1) * D;
    if (red > lds) lds = red;
    hipLaunchKernelGGL(dwconv_bwd_kernel, dim3(grid), dim3(256), lds * sizeof(float), st, dout, in, w, din,
                       (float*)ws, (long)(M * L), L, D, k);
    STAGE_LAUNCH_CHECK();
    stage_colreduce((const float*)ws, dw, db, grid, (long)(k + 1) * D, (k + 1) * D, D, k, st);
    STAGE_LAUNCH_CHECK();
    return 0;
}
