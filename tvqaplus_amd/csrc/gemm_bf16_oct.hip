// Weight gradients of the bf16 storage mode for wide layers (N > 128 output features, N and K multiples of 4): dW = dY^T . X with the
// contraction over the ROWS of two row-major bf16 tensors, streamed from HBM once (256-column blocks of either operand that share a
// slab of rows run side by side on one XCD and meet in its L2).
//
// The tiled kernel of gemm_bf16x3.hip (the only one that took these shapes: hsz = 256 has 16 / 48 patches of 64 x 64, the streaming
// kernel of gemm_bf16_stream.hip stops at 12) ran at 1.0-1.8 TB/s of algorithmic bytes (tools/experiments/tn_bf16_time.py: 2.7 ms for
// 1.4 M rows x 256^T x 768): 8-byte loads, a transpose through 4-byte LDS writes, dY re-read per 128-column tile of X.  This one:
//
//   * 16-byte loads along the rows.  What makes them usable without a transpose through memory: the output index of a GEMM may be
//     permuted freely, so the eight MFMA tiles of a 256-column group (an "octet") are INTERLEAVED -- tile t holds the columns
//     8 l + t (l = lane position 0..31).  A lane that loads the 16 bytes at columns 8l..8l+7 of eight rows holds, after an 8 x 8
//     transpose of 16-bit values in registers (32 v_perm_b32), one 8-row operand fragment of v_mfma_f32_32x32x16_bf16 for each
//     of the eight tiles, at its own lane position.  (The fp32 kernels use the same idea with four tiles: gemm_stream.hip, "quad".)
//   * a workgroup = 8 waves, output tile 256 (all of N) x 256 columns of X; a 64-row super-step is four 16-row MFMA steps: wave w
//     PRODUCES the fragments of operand w & 1 (0 = dY, 1 = X) for step w >> 1 -- eight 16-byte buffer loads per lane, issued one
//     super-step ahead -- and CONSUMES a 4 x 2 block of tiles (dY tiles 4 (w >> 2) .., X tiles 2 (w & 3) ..): per step six
//     conflict-free ds_read_b128 for eight MFMAs.  Fragments are double-buffered in LDS (128 KB): ONE barrier per super-step.
//   * per slab of rows one fp32 partial, written in the permuted order (coalesced), summed in slab order (deterministic, no atomics)
//     by a reduction that undoes the permutation.  Rows past the end of a slab fall outside the buffer descriptor and read as zero.
//   * the workgroups that share a slab (K > 256: one per 256-column block of X, each re-reading dY) are placed on the same XCD and
//     run at the same time: the re-read is served by that XCD's L2.
// Algorithmic bytes: M * (N + K) * 2; MFMA work at the HBM roofline is about a quarter of the bf16 matrix-core peak.
#include <stdlib.h>
#include <stdint.h>
#include "common.h"
#include "../../include/stage_hip.h"

typedef __bf16 go_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned go_u4 __attribute__((ext_vector_type(4)));

#define GO_LDS_U4 (2 * 4 * 2 * 8 * 64)             // [buffer][step][operand][tile][lane] uint4 = 128 KB

__device__ __forceinline__ unsigned go_gate1(unsigned v, unsigned gg) {
    // keep the bf16 halves of v whose gate half is > 0 (ReLU backward with the saved bf16 output as gate)
    const unsigned lo = ((int)(gg << 16) > 0) ? 0x0000FFFFu : 0u;
    const unsigned hi = ((int)(gg & 0xFFFF0000u) > 0) ? 0xFFFF0000u : 0u;
    return v & (lo | hi);
}

template <bool HAS_GATE>
__global__ __launch_bounds__(512) void gemm_tn_bf16_oct_kernel(const stage_bf16* __restrict__ dY, const stage_bf16* __restrict__ G,
                                                               const stage_bf16* __restrict__ X, float* __restrict__ part,
                                                               float* __restrict__ part_b, long M, int N, int K, int Kp, int KB,
                                                               int NB, int S, long rows_per_slab) {
    extern __shared__ __attribute__((aligned(16))) uint4 ex[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    // workgroup -> (slab, column block): the KB blocks of a slab sit on one XCD (workgroup ids go round the 8 XCDs)
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int NBK = NB * KB, blk = idx % NBK;
    const int slab = (idx / NBK) * 8 + xcd, kb = blk % KB, nb = blk / KB;
    if (slab >= S) return;
    const int k0 = kb * 256, n0 = nb * 256;
    const long mbeg = (long)slab * rows_per_slab, mend = min(M, mbeg + rows_per_slab);
    if (mbeg >= mend) return;
    // ---- producer role ----
    const bool is_y = (wave & 1) == 0;                  // wave-uniform
    const int sub = wave >> 1;                          // the 16-row step of a super-step this wave prepares
    const int ld = is_y ? N : K;
    const int col = is_y ? n0 + 8 * l31 : k0 + 8 * l31;
    // columns of this group inside the operand: 8, 0, or 4 (N or K = 8 q + 4: the last group is loaded four columns early -- a load
    // that runs past the end of the slab would be dropped as a whole -- and shifted down afterwards)
    const int vc = min(max(ld - col, 0), 8);
    const bool all_cols = __all(vc == 8);
    const unsigned slab_bytes = (unsigned)((mend - mbeg) * (long)ld * 2);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((is_y ? dY : X) + mbeg * ld), 0, (int)slab_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc((void*)((HAS_GATE ? G : dY) + mbeg * N), 0,
                                                                         (int)((unsigned)((mend - mbeg) * (long)N * 2)), 0x00020000);
    const int voff = ((16 * sub + 8 * h) * ld + (vc == 8 ? col : vc == 4 ? col - 4 : 0)) * 2;
    const bool want_b = part_b != nullptr && kb == 0 && is_y;
    float bs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    go_u4 va[8], vg[8];
    auto fetch = [&](long ms) {                         // ms: first row of the super-step, relative to the slab
#pragma unroll
        for (int r = 0; r < 8; r++) {
            // (the row offset is part of the VGPR offset: the range check of a buffer load does not see the SGPR offset)
            const int off = voff + (int)((ms + r) * ld * 2);
            va[r] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);                              // rows past the slab read 0
            if (HAS_GATE && is_y) vg[r] = __builtin_amdgcn_raw_buffer_load_b128(rsg, off, 0, 0);
        }
    };
    auto produce = [&](int buf) {
        if (HAS_GATE && is_y) {
#pragma unroll
            for (int r = 0; r < 8; r++)
#pragma unroll
                for (int d = 0; d < 4; d++) va[r][d] = go_gate1(va[r][d], vg[r][d]);
        }
        if (!all_cols) {                                // ragged last column block (one uniform branch)
#pragma unroll
            for (int r = 0; r < 8; r++)
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const unsigned shifted = d < 2 ? va[r][d + 2] : 0u;
                    va[r][d] = vc == 8 ? va[r][d] : vc == 4 ? shifted : 0u;
                }
        }
        if (want_b) {
#pragma unroll
            for (int r = 0; r < 8; r++)
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    bs[2 * d] += __uint_as_float(va[r][d] << 16);
                    bs[2 * d + 1] += __uint_as_float(va[r][d] & 0xFFFF0000u);
                }
        }
        uint4* dst = ex + (size_t)(((buf * 4 + sub) * 2 + (is_y ? 0 : 1)) * 8) * 64 + lane;
#pragma unroll
        for (int t = 0; t < 8; t++) {                   // tile t = column 8 l + t: dword q = rows (2q, 2q + 1)
            const unsigned sel = (t & 1) ? 0x07060302u : 0x05040100u;
            uint4 f;
            f.x = __builtin_amdgcn_perm(va[1][t >> 1], va[0][t >> 1], sel);
            f.y = __builtin_amdgcn_perm(va[3][t >> 1], va[2][t >> 1], sel);
            f.z = __builtin_amdgcn_perm(va[5][t >> 1], va[4][t >> 1], sel);
            f.w = __builtin_amdgcn_perm(va[7][t >> 1], va[6][t >> 1], sel);
            dst[t * 64] = f;
        }
    };
    // ---- consumer role: dY tiles 4 pn + i, X tiles 2 pk + j ----
    const int pn = wave >> 2, pk = wave & 3;
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    auto consume = [&](int buf) {
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const uint4* ya = ex + (size_t)(((buf * 4 + s) * 2 + 0) * 8 + 4 * pn) * 64 + lane;
            const uint4* xa = ex + (size_t)(((buf * 4 + s) * 2 + 1) * 8 + 2 * pk) * 64 + lane;
            go_bf16x8 a[4], b[2];
#pragma unroll
            for (int i = 0; i < 4; i++) a[i] = __builtin_bit_cast(go_bf16x8, ya[i * 64]);
#pragma unroll
            for (int j = 0; j < 2; j++) b[j] = __builtin_bit_cast(go_bf16x8, xa[j * 64]);
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };
    const long nsup = (mend - mbeg + 63) >> 6;          // workgroup-uniform: the barriers match
    fetch(0);
    produce(0);
    if (nsup > 1) fetch(64);
    __syncthreads();
    for (long s = 0; s < nsup; s++) {
        const int buf = (int)(s & 1);
        if (s + 1 < nsup) {
            produce(buf ^ 1);                           // its loads were issued a super-step ago
            if (s + 2 < nsup) fetch((s + 2) * 64);
        }
        consume(buf);
        __syncthreads();
    }
    // ---- partial of this slab, permuted: position k0 + 32 tx + j holds column k0 + 8 j + tx ----
    float* po = part + (size_t)slab * (size_t)N * Kp;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int pos = k0 + 32 * (2 * pk + j) + l31;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int n = n0 + 8 * ((r & 3) + 8 * (r >> 2) + 4 * h) + 4 * pn + i;
                if (n < N) po[(size_t)n * Kp + pos] = acc[i][j][r];
            }
        }
    // ---- bias-gradient partial: exact fp32 column sums of the gated dY (column block 0 only) ----
    if (part_b != nullptr && kb == 0) {
        float* red = reinterpret_cast<float*>(ex);      // [step 4][half 2][256 columns]; the fragment buffers are free (barrier above)
        if (is_y) {
#pragma unroll
            for (int t = 0; t < 8; t++) red[(sub * 2 + h) * 256 + 8 * l31 + t] = bs[t];
        }
        __syncthreads();
        if (threadIdx.x < 256 && n0 + (int)threadIdx.x < N) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 8; q++) s += red[q * 256 + threadIdx.x];
            part_b[(size_t)slab * N + n0 + threadIdx.x] = s;
        }
    }
}

// dW[n][k] = sum over slabs of the permuted partials (fixed order); four consecutive positions per thread
__global__ void oct_reduce_kernel(const float* __restrict__ part, float* __restrict__ dW, int S, int N, int K, int Kp) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const long C = (long)N * Kp;
    if (i >= C) return;
    const int n = (int)(i / Kp), p = (int)(i - (long)n * Kp);
    float4 s = f4zero();
#pragma unroll 8
    for (int q = 0; q < S; q++) s = f4add(s, ld4(part + (size_t)q * C + i));
    const float v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const int pe = p + e;
        const int k = (pe & ~255) + 8 * (pe & 31) + ((pe & 255) >> 5);
        if (k < K) dW[(size_t)n * K + k] = v[e];
    }
}
// db[n] = sum over slabs, 8 slab groups per column side by side, combined in a fixed order
__global__ __launch_bounds__(256) void oct_reduce_b_kernel(const float* __restrict__ part_b, float* __restrict__ db, int S, int N) {
    __shared__ float red[8][32];
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int n = blockIdx.x * 32 + c;
    float s = 0.f;
    if (n < N) {
#pragma unroll 4
        for (int q = g; q < S; q += 8) s += part_b[(size_t)q * N + n];
    }
    red[g][c] = s;
    __syncthreads();
    if (g == 0 && n < N) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 8; q++) t += red[q][c];
        db[n] = t;
    }
}

// slabs of the octet kernel for a shape (0: not handled here)
static int go_plan(long long M, int N, int K, int* KB, int* NB, int* Kp, long* rps) {
    static const bool off = getenv("STAGE_GEMM_BF16_NO_OCT") != nullptr;
    if (off || M < 8192 || N % 4 != 0 || K % 4 != 0 || N <= 128 || K < 64) return 0;
    *KB = (K + 255) / 256;
    *NB = (N + 255) / 256;
    *Kp = *KB * 256;
    if (*KB * *NB > 32) return 0;
    int S = 8 * (32 / (*KB * *NB));                      // one workgroup per CU, the blocks of a slab on one XCD
    long r = (long)((M + S - 1) / S);
    r = (r + 63) / 64 * 64;
    if (r < 1024) r = 1024;                              // (short slabs: the partials would outweigh the operands)
    if (r * (long)(N > K ? N : K) * 2 >= (1L << 31)) return 0;      // a slab must fit a buffer descriptor's 32-bit offsets
    *rps = r;
    return (int)((M + r - 1) / r);
}

size_t stage_gemm_tn_bf16_oct_ws_bytes(long long M, int N, int K) {
    int KB, NB, Kp; long rps;
    const int S = go_plan(M, N, K, &KB, &NB, &Kp, &rps);
    return (size_t)S * ((size_t)N * Kp + N) * sizeof(float);
}

// returns 1 if the shape / alignment is not handled here, 0 after writing dW (and db)
int stage_gemm_tn_bf16_oct(const void* dY, const void* gate, const void* X, float* dW, float* db, long long M, int N, int K, void* ws,
                           size_t ws_bytes, void* stream) {
    int KB, NB, Kp; long rps;
    const int S = go_plan(M, N, K, &KB, &NB, &Kp, &rps);
    if (S == 0 || ((uintptr_t)dY & 7) || ((uintptr_t)X & 7) || (gate && ((uintptr_t)gate & 7))) return 1;
    if (ws_bytes < (size_t)S * ((size_t)N * Kp + N) * sizeof(float)) return 1;
    hipStream_t st = (hipStream_t)stream;
    typedef stage_bf16 B;
    float* part = (float*)ws;
    float* part_b = part + (size_t)S * N * Kp;
    const int lds = GO_LDS_U4 * 16;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm_tn_bf16_oct_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void*)gemm_tn_bf16_oct_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr = true;
    }
    const unsigned grid = 8u * (unsigned)((S + 7) / 8) * (unsigned)(KB * NB);
    if (gate)
        hipLaunchKernelGGL(gemm_tn_bf16_oct_kernel<true>, dim3(grid), dim3(512), lds, st, (const B*)dY, (const B*)gate, (const B*)X, part,
                           db ? part_b : (float*)nullptr, (long)M, N, K, Kp, KB, NB, S, rps);
    else
        hipLaunchKernelGGL(gemm_tn_bf16_oct_kernel<false>, dim3(grid), dim3(512), lds, st, (const B*)dY, (const B*)gate, (const B*)X, part,
                           db ? part_b : (float*)nullptr, (long)M, N, K, Kp, KB, NB, S, rps);
    const long C = (long)N * Kp;
    hipLaunchKernelGGL(oct_reduce_kernel, dim3((unsigned)((C / 4 + 255) / 256)), dim3(256), 0, st, part, dW, S, N, K, Kp);
    if (db) hipLaunchKernelGGL(oct_reduce_b_kernel, dim3((N + 31) / 32), dim3(256), 0, st, part_b, db, S, N);
    STAGE_LAUNCH_CHECK();
    return 0;
}
