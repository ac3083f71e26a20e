// Streaming NT GEMM for the tall-skinny Linear / 1x1-conv shapes of the STAGE path (M ~ 1e5..1e6 rows, N, K ~ 128..768):
//     Y[M,N] = epi( (X (*) gate)[M,K] . W[N,K]^T + bias )           same contract as stage_gemm_nt (gemm.hip)
// fp32 accuracy through a two-way fp16 split with power-of-two scaling (below: STAGE_GEMM_NT_F16; the 3-way bf16 split of
// gemm_bf16x3.hip is the other compile-time option), organised for HBM streaming:
//   * the weight tile (128 output columns x 128 k) is split ONCE per workgroup into its two fp16 planes and stays in
//     LDS (70 KB); for K <= 128 it is never reloaded while the workgroup walks its row tiles;
//   * the X operand never touches LDS: every wave owns 32-row tiles and loads them straight from HBM in MFMA operand
//     layout (lane = (row, k-half), 16 B per lane).  The MFMA contraction order is free as long as both operands agree,
//     so k is permuted inside every 16-group such that a lane's two float4 loads (k = 8c + 4h + 0..3, c = 0,1) form its
//     8-element operand; the LDS image of W is written with the same permutation.  tools/ubench/copy_bw.hip measures
//     this access shape at 5.6 TB/s copy bandwidth, the same as fully row-contiguous loads;
//   * no workgroup barrier in the row loop (K <= 128): 8 waves per CU drift freely, so loads, the split VALU work, the
//     matrix cores and the stores of different waves overlap; the next 32-k line is prefetched while one is multiplied.
// K > 128 walks K in 128-wide chunks with a barrier pair per chunk (weight chunk reload), accumulators stay in registers.
#include <stdlib.h>
#include "common.h"
#include "../../include/stage_hip.h"

#ifndef GEMM_ABL
#define GEMM_ABL 0      // developer ablation bits (timing experiments only, results wrong; 64: no row-exponent tracking): 1 one MFMA per column tile instead
#endif                  // of 6, 2 no bf16 split (raw bits as operands), 4 X lines fetched once per wave, 8 no stores,
                        // 16 weight fragments read from LDS for one of the four column tiles only; TN share kernel: 1, 2, 4 alike,
                        // 32 no barrier
#ifndef STAGE_GEMM_TERMS
#define STAGE_GEMM_TERMS 3      // bf16 mode only (STAGE_GEMM_NT_F16 / _TN_F16 = 0): bf16 terms per fp32 operand: 3 = exact split (six products, 3e-7 vs fp64);
#endif                          // 2 = hi + mid only (three products kept: every product carries a relative error <= ~2^-17,
                                // measured ~1e-5 of the result's scale; half the matrix-core work) -- `make TERMS=2`
#define SBN 128                 // output columns per workgroup
#define SKC 128                 // k per resident weight chunk
#define SWS (SKC + 8)           // bf16 per LDS row of a plane (272 B: 16-lane ds_read_b128 groups hit distinct slots)
#define SPLANE (SBN * SWS)      // bf16 elements per plane
#define SWAVES 8

typedef __bf16 sbf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned s_cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// (a, b) -> three packed bf16 pairs (a low half): exact residual splits, a == a1 + a2 + a3
__device__ __forceinline__ void s_split3(float a, float b, unsigned (&out)[3]) {
#pragma unroll
    for (int s = 0; s < 3; s++) {
        if (s < STAGE_GEMM_TERMS) {
            out[s] = s_cvt_pk_bf16(a, b);
            if (s + 1 < STAGE_GEMM_TERMS) {
                a -= __uint_as_float(out[s] << 16);
                b -= __uint_as_float(out[s] & 0xFFFF0000u);
            }
        } else out[s] = 0u;
    }
}
// acc += sum of the kept cross terms of (a0 + a1 + a2) x (b0 + b1 + b2), smallest first
__device__ __forceinline__ f32x16 s_mfma_terms(const sbf16x8 (&a)[3], const sbf16x8 (&b)[3], f32x16 acc) {
    if (STAGE_GEMM_TERMS == 3) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    return acc;
}
// ---------------------------------------------------------------------------------------------------------------------
// NT kernel: fp32 products as a TWO-way fp16 split (STAGE_GEMM_NT_F16, default).  fp16 carries 11 significant bits, so
// x = hi + lo (both round-to-nearest) holds 22-23 bits of x and the three products hi*hi + hi*lo + lo*hi reproduce the fp32
// product to ~2^-22 -- against an fp64 product the result's error is BELOW that of a plain fp32 FMA chain (numpy emulation,
// K = 384, relative to the result's rms: max 1.2e-6 / rms 2.1e-7; torch / numpy fp32 matmul 3.8e-6 / 3.5e-7) -- with HALF the
// matrix-core instructions of the 3-way bf16 split and 5 instead of 11 VALU instructions per operand pair.  What fp16 lacks
// is range, so operands are scaled by powers of two (exact) on the way in and the accumulators on the way out:
//   * X: one exponent per ROW (= per lane, both lane halves agree through v_permlane32_swap): the row's largest magnitude
//     seen so far is mapped into [2^11, 2^12); a later line that would pass 2^15 raises the row's exponent and the
//     accumulators of the wave are rescaled (v_ldexp, a wave-uniform branch that real data takes once per tile at most);
//   * W: one exponent per workgroup from a pre-pass over its (<= 128 x K) weight tile.
// An element far below its row's maximum keeps an ABSOLUTE error of 2^-26 of that maximum (fp16 denormals) instead of
// fp32's relative 2^-24: irrelevant for a dot product, whose error scale is the largest terms.
#ifndef STAGE_GEMM_NT_F16
#define STAGE_GEMM_NT_F16 1
#endif
__device__ __forceinline__ f32x16 h_mfma_terms(const sf16x8 (&a)[2], const sf16x8 (&b)[2], f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], acc, 0, 0, 0);
    return acc;
}
// TN kernels (contraction over rows): the scale belongs to an operand COLUMN = a lane of the tile's producer.  Running
// exponent of the column (both lane halves agree); returns the change of the scale's exponent field at this step (<= 0),
// which the consumers of the tile apply to their accumulators before they use it.
#ifndef STAGE_GEMM_TN_F16
#define STAGE_GEMM_TN_F16 1
#endif
__device__ __forceinline__ float h_amax8(const float (&v)[8]) {
    return h_amax3(h_amax3(h_amax3(h_amax3(v[0], v[1], v[2]), v[3], v[4]), v[5], v[6]), v[7], v[7]);
}
__device__ __forceinline__ int h_track8(const float (&v)[8], int& eb) {
    const float m = xmax32(h_amax8(v));
    const int ec = (int)(__float_as_uint(m) >> 23) & 0xff;
    const int neb = ec > eb + 3 ? ec : eb;
    const int d = h_up_field(neb) - h_up_field(eb);
    eb = neb;
    return d;
}
__device__ __forceinline__ void h_split8(const float (&v)[8], int eb, uint4& hi, uint4& lo) {
    const float sc = __uint_as_float((unsigned)h_up_field(eb) << 23);
    h_split2(v[0], v[1], sc, hi.x, lo.x);
    h_split2(v[2], v[3], sc, hi.y, lo.y);
    h_split2(v[4], v[5], sc, hi.z, lo.z);
    h_split2(v[6], v[7], sc, hi.w, lo.w);
}
__device__ __forceinline__ int h_row_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }   // C/D row of register r

__device__ __forceinline__ float s_absmax4(float m, float4 v) { return h_amax3(h_amax3(m, v.x, v.y), v.z, v.w); }

__device__ __forceinline__ float4 s_load4(const float* __restrict__ base, long row, long ld, int col, long nrows, int ncols) {
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
// address-clamped load WITHOUT the zero select (a select at the load site forces an s_waitcnt right behind the load)
__device__ __forceinline__ float4 s_load4_raw(const float* __restrict__ base, long row, long ld, int col, long nrows, int ncols) {
    const long r = row < nrows ? row : nrows - 1;
    const int c = col < ncols ? col : ncols - 4;
    return ld4(base + r * ld + c);
}
__device__ __forceinline__ float4 s_keep4(float4 v, bool ok) {
    return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}
__device__ __forceinline__ float4 s_gate4(float4 v, float4 g) {
    return make_float4(g.x > 0.f ? v.x : 0.f, g.y > 0.f ? v.y : 0.f, g.z > 0.f ? v.z : 0.f, g.w > 0.f ? v.w : 0.f);
}

// GATE: 0 none, 1 fp32 tensor (x kept where gate > 0), 2 bit mask (uint32 [ceil(K/32)][M] word-major, bit b of word w <=>
// column 32w + b kept).  MASK_OUT: also emit the ReLU bit mask of Y (uint32 [ceil(N/32)][M]) for the two backward GEMMs -- they
// then read 1/32 of the bytes the fp32 gate costs.
// COLSUM: the product is the gradient of a LayerNorm OUTPUT whose input needs no gradient (the first layer of the input MLPs:
// LN(768) -> dropout -> Linear(768 -> 300), model/stage.py:85-91): only the LayerNorm's gain / bias gradients are wanted,
//     dgamma[n] = sum_m Y[m,n] * keep[m,n] / (1-p) * x_hat[m,n]      dbeta[n] = sum_m Y[m,n] * keep[m,n] / (1-p)
// so the tile is reduced over its rows in the epilogue and never stored (737 MB written and read back by a LayerNorm backward
// for 2 x 768 numbers, at the subtitle stream).  R = the LayerNorm input x (M, N), cs = {mean, rstd (M each), keep bits}; per wave and
// column the partial sums go to cs_part[(row group * 8 + wave)][2][N], finished by the column reduction of the LayerNorm backward.
struct StageColsum {
    const float* mean;
    const float* rstd;
    float* part;
    const unsigned* keep;   // dropout keep bits of the LayerNorm output, [ceil(N/32)][M] words (NULL: no dropout); hashing them
    float inv_keep;         // here (64-bit multiplies) spilled 600 registers, a 23 MB bit mask from a 30 us pre-pass does not
};
template <int GATE, bool HAS_RES, bool WPRE, bool MASK_OUT, bool COLSUM = false>
__global__ __launch_bounds__(64 * SWAVES, 2) void gemm_nt_stream_kernel(const float* __restrict__ X,
                                                                        const float* __restrict__ G,
                                                                        const float* __restrict__ W,
                                                                        const float* __restrict__ bias,
                                                                        const float* __restrict__ R, float* __restrict__ Y,
                                                                        unsigned* __restrict__ mask_out, long M, int N,
                                                                        int K, int relu, int xcd_gx, StageColsum cs = StageColsum()) {
    constexpr bool HAS_GATE = GATE == 1;
    extern __shared__ __attribute__((aligned(16))) unsigned short Wp[];   // [2 fp16 planes (3 bf16 planes)][SBN][SWS], k permuted per 16-group
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    // Workgroup -> (row group bx, column tile by).  Workgroups are dealt round-robin to the 8 XCDs (id % 8), each with its own
    // L2: with a plain 2-D grid the column tiles of one row range land on different XCDs and every one of them pulls the X
    // rows from HBM again (N = 384: 3 x 491 MB instead of 1 x).  xcd_gx > 0 (a multiple of 8 row groups): 1-D grid,
    // id = 8 * (n_tiles * q + by) + r with bx = 8 q + r -- all column tiles of a row group share id % 8, the re-reads hit L2.
    int bx = blockIdx.x, by = blockIdx.y, gx = gridDim.x;
    if (xcd_gx > 0) {
        const int n_tiles_g = (N + SBN - 1) / SBN;
        const int r = blockIdx.x & 7, tq = blockIdx.x >> 3;
        by = tq % n_tiles_g;
        bx = (tq / n_tiles_g) * 8 + r;
        gx = xcd_gx;
    }
    const int n0 = by * SBN;
    const int nkc = (K + SKC - 1) / SKC;
    const long MT = (M + 31) >> 5;                       // 32-row wave tiles
    const long n_bt = (MT + SWAVES - 1) / SWAVES;        // workgroup iterations

    // weight chunk kc -> LDS planes.  float4 group q of a row covers k = 4q..4q+3 = 16u + 8c + 4h' + e  (q = 4u + 2c + h')
    // and lands at plane position 16u + 8h' + 4c + e, i.e. lane-half h' finds its 8 operand values contiguous.
    // fp16 mode: power-of-two scale of this workgroup's weight tile (all K), from a pre-pass over its rows (L2 reads)
    int w_up = 127;                                      // exponent field of the scale; 127 = 1.0
    if (STAGE_GEMM_NT_F16) {
        float wm = 0.f;
        const int kq = K >> 2;
        for (int e = tid; e < SBN * kq; e += 64 * SWAVES) {
            const int n = e / kq, q = e - n * kq;
            wm = s_absmax4(wm, s_load4(W, n0 + n, K, 4 * q, N, K));
        }
        wm = wave_max(wm);
        float* red = reinterpret_cast<float*>(Wp);
        if (lane == 0) red[wave] = wm;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < SWAVES; i++) wm = fmaxf(wm, red[i]);
        __syncthreads();                                 // the planes are written next
        w_up = h_up_field((int)(__float_as_uint(wm) >> 23) & 0xff);
    }
    const float w_sc = __uint_as_float((unsigned)w_up << 23);
    auto put_w = [&](float4 v, int n, int q) {
        const int pos = 16 * (q >> 2) + 8 * (q & 1) + 4 * ((q >> 1) & 1);
        if (STAGE_GEMM_NT_F16) {
            unsigned h01, l01, h23, l23;
            h_split2(v.x, v.y, w_sc, h01, l01);
            h_split2(v.z, v.w, w_sc, h23, l23);
            *reinterpret_cast<uint2*>(&Wp[n * SWS + pos]) = make_uint2(h01, h23);
            *reinterpret_cast<uint2*>(&Wp[SPLANE + n * SWS + pos]) = make_uint2(l01, l23);
        } else {
            unsigned s01[3], s23[3];
            s_split3(v.x, v.y, s01);
            s_split3(v.z, v.w, s23);
#pragma unroll
            for (int s = 0; s < STAGE_GEMM_TERMS; s++) *reinterpret_cast<uint2*>(&Wp[s * SPLANE + n * SWS + pos]) = make_uint2(s01[s], s23[s]);
        }
    };
    auto load_w = [&](int kc) {
        for (int e = tid; e < SBN * (SKC / 4); e += 64 * SWAVES) {
            const int n = e >> 5, q = e & 31;
            put_w(s_load4(W, n0 + n, K, kc * SKC + 4 * q, N, K), n, q);
        }
    };

    // K > 128 with WPRE: the NEXT weight chunk is requested (untracked loads, 8 float4 per thread) before the current chunk
    // is multiplied and split into the LDS planes after the barrier that ends the chunk, so its L2 latency hides behind
    // 192 MFMAs per wave instead of stalling all 8 waves between two barriers.
    f32x4 wr[8];
    const bool multi = WPRE && nkc > 1;
    auto w_issue = [&](int kc) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int e = tid + j * 64 * SWAVES, n = e >> 5, q = e & 31;
            const float* pw = W + (long)min(n0 + n, N - 1) * K + min(kc * SKC + 4 * q, K - 4);
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(wr[j]) : "v"(pw) : "memory");
        }
    };
    auto w_stash = [&](int kc) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int e = tid + j * 64 * SWAVES, n = e >> 5, q = e & 31;
            const bool ok = n0 + n < N && kc * SKC + 4 * q < K;
            put_w(make_float4(ok ? wr[j][0] : 0.f, ok ? wr[j][1] : 0.f, ok ? wr[j][2] : 0.f, ok ? wr[j][3] : 0.f), n, q);
        }
    };
#define WAIT_W(n)                                                                                                      \
    asm volatile("s_waitcnt vmcnt(%8)"                                                                                 \
                 : "+v"(wr[0]), "+v"(wr[1]), "+v"(wr[2]), "+v"(wr[3]), "+v"(wr[4]), "+v"(wr[5]), "+v"(wr[6]), "+v"(wr[7]) \
                 : "n"(n)                                                                                              \
                 : "memory")

    // X (and gate) lines in flight: two 32-k line buffers, ping-ponged inside a tile.  vmcnt retires in issue order on
    // gfx9, so a load issued after the 64 epilogue stores of a tile cannot be consumed before those stores are
    // acknowledged: the first two lines of the NEXT tile are therefore requested BEFORE the stores of the current one.
    // The compiler's own s_waitcnt bookkeeping cannot express "older than 64 stores" (it falls back to vmcnt(0..2) at the
    // loop head, draining the stores), so the X/gate loads are issued as asm and awaited with hand-counted vmcnt values.
    // Rule for every count below: it must not exceed the number of VMEM instructions (of any kind) issued after the
    // awaited loads -- a smaller count only waits longer.
    constexpr int NL = GATE == 1 ? 8 : (GATE == 2 ? 5 : 4);   // loads per fetch
    f32x4 xa[2][4], ga[2][4];
    unsigned gw[2];                                          // GATE 2: the mask word of the line
    // buffer loads: resource + 32-bit lane offset + immediate.  A ragged last chunk (K % 128 != 0) reads past the end of
    // its row -- into the next row, or as 0 past the end of the tensor -- and is zeroed at use (mul_line).
    typedef int s_rsrc_t __attribute__((ext_vector_type(4)));
    auto make_rsrc = [&](const float* p) -> s_rsrc_t {
        const unsigned long long a = (unsigned long long)p;
        return (s_rsrc_t){(int)(unsigned)a, (int)(unsigned)((a >> 32) & 0xffffu), (int)(M * K * 4), 0x00020000};
    };
    const s_rsrc_t rsx = make_rsrc(X), rsg = make_rsrc(HAS_GATE ? G : X);
    const int NWK = (K + 31) >> 5;                        // mask words per row (GATE 2)
    bool abl_fetched[2] = {false, false};
    auto fetch = [&](int buf, long row, int k_line) {   // one offset per lane, immediate offsets for the 4 chunks
        if ((GEMM_ABL & 4) && abl_fetched[buf]) return;
        abl_fetched[buf] = true;
        const long rcl = row < M ? row : M - 1;
        const int off = (int)((rcl * K + k_line + 4 * h) * 4);
        if (GATE == 2) {
            const unsigned* pm = reinterpret_cast<const unsigned*>(G) + (long)min(k_line >> 5, NWK - 1) * M + rcl;   // [word][row]
            asm volatile("global_load_dword %0, %1, off" : "=v"(gw[buf]) : "v"(pm) : "memory");
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3"
                         : "=v"(xa[buf][c]) : "v"(off), "s"(rsx), "n"(32 * c) : "memory");
            if (HAS_GATE)
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3"
                             : "=v"(ga[buf][c]) : "v"(off), "s"(rsg), "n"(32 * c) : "memory");
        }
    };
#define WAIT_LINE(buf, n)                                                                                              \
    do {                                                                                                               \
        if constexpr (GATE == 2)                                                                                       \
            asm volatile("s_waitcnt vmcnt(%5)"                                                                         \
                         : "+v"(xa[buf][0]), "+v"(xa[buf][1]), "+v"(xa[buf][2]), "+v"(xa[buf][3]), "+v"(gw[buf])       \
                         : "n"(n)                                                                                      \
                         : "memory");                                                                                  \
        else if constexpr (HAS_GATE)                                                                                   \
            asm volatile("s_waitcnt vmcnt(%8)"                                                                         \
                         : "+v"(xa[buf][0]), "+v"(xa[buf][1]), "+v"(xa[buf][2]), "+v"(xa[buf][3]), "+v"(ga[buf][0]),   \
                           "+v"(ga[buf][1]), "+v"(ga[buf][2]), "+v"(ga[buf][3])                                        \
                         : "n"(n)                                                                                      \
                         : "memory");                                                                                  \
        else                                                                                                           \
            asm volatile("s_waitcnt vmcnt(%4)"                                                                         \
                         : "+v"(xa[buf][0]), "+v"(xa[buf][1]), "+v"(xa[buf][2]), "+v"(xa[buf][3])                      \
                         : "n"(n)                                                                                      \
                         : "memory");                                                                                  \
    } while (0)
    // bias of the 128 output columns goes to LDS once: a register loaded from global before the row loop makes the
    // compiler guard every epilogue with s_waitcnt vmcnt(0) (which would also drain the stores and the prefetch)
    float* bias_s = reinterpret_cast<float*>(Wp + 3 * SPLANE);
    if (tid < SBN) bias_s[tid] = (bias && n0 + tid < N) ? bias[n0 + tid] : 0.f;
    // Units of work of a wave: (tile, k-chunk).  Lines 0 and 1 of the NEXT unit are always requested right after line 3 of
    // the current one has been consumed (straight-line code: asm results defined inside a branch get merged by register
    // copies, and a copy of a register whose load is still in flight copies garbage).  When the next unit is a new tile,
    // that request therefore precedes the 64 stores of the current tile ("steady": the waits then use vmcnt(63)).
    const bool full_cols = !COLSUM && n0 + SBN <= N;      // this workgroup stores all 4 column tiles: 64 stores per tile (COLSUM: none)
    bool w_loaded = false, first = true, after_stores = false;
    if (multi) w_issue(0);
    fetch(0, ((long)bx * SWAVES + wave) * 32 + l31, 0);
    fetch(1, ((long)bx * SWAVES + wave) * 32 + l31, 32);
    // COLSUM: running column sums of this wave, kept in LDS (lane-private slots [wave][h][dgamma | dbeta][128 columns]: eight more
    // live registers spill next to the loads in flight)
    float* cs_acc = reinterpret_cast<float*>(Wp + 3 * SPLANE) + SBN + (wave * 2 + h) * 2 * SBN;
    const int NWN_cs = (N + 31) >> 5;
    const __amdgpu_buffer_rsrc_t cs_rx = __builtin_amdgcn_make_buffer_rsrc((void*)(COLSUM ? R : X), 0, (int)(M * (COLSUM ? N : K) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t cs_rm = __builtin_amdgcn_make_buffer_rsrc((void*)(COLSUM ? cs.mean : X), 0, (int)(M * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t cs_rr = __builtin_amdgcn_make_buffer_rsrc((void*)(COLSUM ? cs.rstd : X), 0, (int)(M * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t cs_rk = __builtin_amdgcn_make_buffer_rsrc((void*)((COLSUM && cs.keep) ? (const void*)cs.keep : (const void*)X), 0,
                                                                           (int)(M * NWN_cs * 4), 0x00020000);
    if (COLSUM) {
#pragma unroll
        for (int nt = 0; nt < 4; nt++) cs_acc[nt * 32 + l31] = cs_acc[SBN + nt * 32 + l31] = 0.f;
    }
    for (long bt = bx; bt < n_bt; bt += gx) {
        const long t = bt * SWAVES + wave;               // this wave's tile (may be past the end: then it only syncs)
        const bool live = t < MT;
        const long row = t * 32 + l31;
        f32x16 acc[4];
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[nt][r] = 0.f;
        int xeb = 0;                                     // fp16 mode: biased exponent of this lane's row maximum so far

        for (int kc = 0; kc < nkc; kc++) {
            const int kb = kc * SKC;
            if (multi) {
                __syncthreads();                         // everyone done with the previous chunk
                // younger than the weight request: lines 0/1 of this unit (2 NL; the lines requested while the previous
                // chunk was multiplied were awaited there) and, after a tile end, its 64 stores; nothing for a wave
                // without tiles.  Exact counts in untied waits, the tied wait carries the registers.
                if (!live) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (!after_stores) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NL) : "memory");
                WAIT_W(63);
                w_stash(kc);
                __syncthreads();
                w_issue(kc + 1 == nkc ? 0 : kc + 1);
            } else if (nkc > 1 || !w_loaded) {
                __syncthreads();                         // everyone done with the previous chunk
                load_w(kc);
                __syncthreads();
                w_loaded = true;
            }
            if (!live) continue;
            auto mul_line = [&](int buf, int L) {
                float4 vv[4];                             // the lane's 16 values of the line: [2 up + c]
#pragma unroll
                for (int up = 0; up < 2; up++) {          // two 16-k MFMA steps per line
                    float4 v0 = make_float4(xa[buf][2 * up][0], xa[buf][2 * up][1], xa[buf][2 * up][2], xa[buf][2 * up][3]);
                    float4 v1 = make_float4(xa[buf][2 * up + 1][0], xa[buf][2 * up + 1][1], xa[buf][2 * up + 1][2],
                                            xa[buf][2 * up + 1][3]);
                    if (K & (SKC - 1)) {   // ragged last chunk: columns past K hold the next row's data (or 0)
                        v0 = s_keep4(v0, kb + 32 * L + 16 * up + 4 * h < K);
                        v1 = s_keep4(v1, kb + 32 * L + 16 * up + 8 + 4 * h < K);
                    }
                    if (HAS_GATE) {
                        const f32x4 g0 = ga[buf][2 * up], g1 = ga[buf][2 * up + 1];
                        v0 = s_gate4(v0, make_float4(g0[0], g0[1], g0[2], g0[3]));
                        v1 = s_gate4(v1, make_float4(g1[0], g1[1], g1[2], g1[3]));
                    }
                    if (GATE == 2) {   // bit 8c + 4h + e of the line's word (c = 2up, 2up+1): 0 / -1 by v_bfe_i32, then and
                        const int wbits = (int)gw[buf], p0b = 16 * up + 4 * h;
                        v0.x = __int_as_float(__float_as_int(v0.x) & __builtin_amdgcn_sbfe(wbits, p0b + 0, 1));
                        v0.y = __int_as_float(__float_as_int(v0.y) & __builtin_amdgcn_sbfe(wbits, p0b + 1, 1));
                        v0.z = __int_as_float(__float_as_int(v0.z) & __builtin_amdgcn_sbfe(wbits, p0b + 2, 1));
                        v0.w = __int_as_float(__float_as_int(v0.w) & __builtin_amdgcn_sbfe(wbits, p0b + 3, 1));
                        v1.x = __int_as_float(__float_as_int(v1.x) & __builtin_amdgcn_sbfe(wbits, p0b + 8, 1));
                        v1.y = __int_as_float(__float_as_int(v1.y) & __builtin_amdgcn_sbfe(wbits, p0b + 9, 1));
                        v1.z = __int_as_float(__float_as_int(v1.z) & __builtin_amdgcn_sbfe(wbits, p0b + 10, 1));
                        v1.w = __int_as_float(__float_as_int(v1.w) & __builtin_amdgcn_sbfe(wbits, p0b + 11, 1));
                    }
                    vv[2 * up] = v0;
                    vv[2 * up + 1] = v1;
                }
                float x_sc = 1.f;
                if (STAGE_GEMM_NT_F16 && !(GEMM_ABL & 64)) {
                    // the row's exponent: largest magnitude of the line (both lane halves), raised only when a value would
                    // pass 2^15 after scaling; the first line of a tile sets it (the accumulators are zero)
                    const float m = xmax32(s_absmax4(s_absmax4(s_absmax4(s_absmax4(0.f, vv[0]), vv[1]), vv[2]), vv[3]));
                    const int ec = (int)(__float_as_uint(m) >> 23) & 0xff;
                    if (kc == 0 && L == 0) xeb = ec;
                    else if (__any(ec > xeb + 3)) {
                        const int neb = ec > xeb + 3 ? ec : xeb;
                        const int d = h_up_field(neb) - h_up_field(xeb);          // <= 0: this row's accumulators shrink by 2^d
#pragma unroll
                        for (int r = 0; r < 16; r++) {
                            const int dr = __builtin_amdgcn_ds_bpermute(4 * ((r & 3) + 8 * (r >> 2) + 4 * h), d);
#pragma unroll
                            for (int nt = 0; nt < 4; nt++) acc[nt][r] = __builtin_ldexpf(acc[nt][r], dr);
                        }
                        xeb = neb;
                    }
                    x_sc = __uint_as_float((unsigned)h_up_field(xeb) << 23);
                }
#pragma unroll
                for (int up = 0; up < 2; up++) {
                    const float4 v0 = vv[2 * up], v1 = vv[2 * up + 1];
                    const int koff = 16 * (2 * L + up) + 8 * h;
                    if (STAGE_GEMM_NT_F16) {
                        unsigned ph[4], pl[4];
                        sf16x8 a[2], b[2];
                        if (GEMM_ABL & 2) {
                            a[0] = __builtin_bit_cast(sf16x8, v0);
                            a[1] = __builtin_bit_cast(sf16x8, v1);
                        } else {
                            h_split2(v0.x, v0.y, x_sc, ph[0], pl[0]);
                            h_split2(v0.z, v0.w, x_sc, ph[1], pl[1]);
                            h_split2(v1.x, v1.y, x_sc, ph[2], pl[2]);
                            h_split2(v1.z, v1.w, x_sc, ph[3], pl[3]);
                            a[0] = __builtin_bit_cast(sf16x8, make_uint4(ph[0], ph[1], ph[2], ph[3]));
                            a[1] = __builtin_bit_cast(sf16x8, make_uint4(pl[0], pl[1], pl[2], pl[3]));
                        }
#pragma unroll
                        for (int nt = 0; nt < 4; nt++) {
#pragma unroll
                            for (int s2 = 0; s2 < 2; s2++)
                                if (!(GEMM_ABL & 16) || nt == 0)
                                    b[s2] = __builtin_bit_cast(sf16x8, *reinterpret_cast<const uint4*>(&Wp[s2 * SPLANE + (nt * 32 + l31) * SWS + koff]));
                            if (GEMM_ABL & 1) {
                                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0] + a[1], b[0] + b[1], acc[nt], 0, 0, 0);
                                continue;
                            }
                            acc[nt] = h_mfma_terms(a, b, acc[nt]);
                        }
                        continue;
                    }
                    unsigned p0[3], p1[3], p2[3], p3[3];
                    sbf16x8 a[3];
                    if (GEMM_ABL & 2) {
                        a[0] = __builtin_bit_cast(sbf16x8, v0);
                        a[1] = __builtin_bit_cast(sbf16x8, v1);
                        a[2] = __builtin_bit_cast(sbf16x8, make_float4(v0.x, v1.y, v0.z, v1.w));
                    } else {
                        s_split3(v0.x, v0.y, p0);
                        s_split3(v0.z, v0.w, p1);
                        s_split3(v1.x, v1.y, p2);
                        s_split3(v1.z, v1.w, p3);
#pragma unroll
                        for (int s = 0; s < STAGE_GEMM_TERMS; s++) a[s] = __builtin_bit_cast(sbf16x8, make_uint4(p0[s], p1[s], p2[s], p3[s]));
                    }
                    sbf16x8 b[3];
#pragma unroll
                    for (int nt = 0; nt < 4; nt++) {
#pragma unroll
                        for (int s = 0; s < STAGE_GEMM_TERMS; s++)
                            if (!(GEMM_ABL & 16) || nt == 0)   // ablation: weight fragments read for the first column tile only
                                b[s] = __builtin_bit_cast(sbf16x8, *reinterpret_cast<const uint4*>(&Wp[s * SPLANE + (nt * 32 + l31) * SWS + koff]));
                        // kept cross terms, smallest first
                        if (GEMM_ABL & 1) {
                            const sbf16x8 am = __builtin_bit_cast(sbf16x8, __builtin_bit_cast(uint4, a[0]) ^ __builtin_bit_cast(uint4, a[1]) ^ __builtin_bit_cast(uint4, a[2]));
                            const sbf16x8 bm = __builtin_bit_cast(sbf16x8, __builtin_bit_cast(uint4, b[0]) ^ __builtin_bit_cast(uint4, b[1]) ^ __builtin_bit_cast(uint4, b[2]));
                            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[nt], 0, 0, 0);
                            continue;
                        }
                        acc[nt] = s_mfma_terms(a, b, acc[nt]);
                    }
                }
            };
            // In flight (issue order): [line 0][line 1] {64 stores of the previous tile, if steady}.
            // Not steady: the exact counts are applied first by an untied wait inside the (uniform) branch; the tied
            // waits that carry the registers stay outside of any branch.
            const bool steady = kc == 0 && !first && full_cols;
            // (multi: the 8 weight loads of the next chunk were requested after lines 0/1 as well)
            if (!steady) {
                if (multi) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL + 8) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");          // newer than line 0: line 1
            }
            WAIT_LINE(0, 63);
            mul_line(0, 0);
            fetch(0, row, kb + 64);
            if (!steady) {
                if (multi) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL + 8) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");          // newer than line 1: line 2
            }
            WAIT_LINE(1, 63);                                                         // steady: 64 stores + line 2
            mul_line(1, 1);
            fetch(1, row, kb + 96);
            WAIT_LINE(0, NL);                                                         // newer than line 2: line 3
            mul_line(0, 2);
            WAIT_LINE(1, 0);
            mul_line(1, 3);
            // lines 0, 1 of the next unit (same tile / next chunk, or next tile / chunk 0; clamped past the end)
            const bool last_chunk = kc + 1 == nkc;
            const long nrow = last_chunk ? ((bt + gx) * SWAVES + wave) * 32 + l31 : row;
            const int nk = last_chunk ? 0 : kb + SKC;
            fetch(0, nrow, nk);
            fetch(1, nrow, nk + 32);
            first = false;
            after_stores = last_chunk && full_cols;   // this wave is live: its 64 epilogue stores follow
        }
        if (!live) continue;
        // epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
        // (Accumulating the tile transposed -- weight fragment as the A operand -- gives every lane 4 consecutive columns of
        // one row and 16-byte stores, 16 per tile instead of 64; measured 25 % SLOWER at K = 128: an instruction then covers 32
        // rows x 32 bytes instead of 2 rows x 128 bytes, and partial-line writes are what the memory system handles worst.)
        // Straight-line stores: a residual load or a per-row guard inside this loop makes the compiler put an
        // s_waitcnt vmcnt(0) in front of every store (each store then waits for the previous one to be acknowledged).
        const bool full = t * 32 + 32 <= M;
        if (STAGE_GEMM_NT_F16) {                          // back to true units: 2^-(row scale + weight scale), per output row
            const int dn = 254 - h_up_field(xeb) - w_up;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int dr = __builtin_amdgcn_ds_bpermute(4 * ((r & 3) + 8 * (r >> 2) + 4 * h), dn);
#pragma unroll
                for (int nt = 0; nt < 4; nt++) acc[nt][r] = __builtin_ldexpf(acc[nt][r], dr);
            }
        }
        if (COLSUM) {
            // rows of this lane: t*32 + 4h + dm(r); column n0 + nt*32 + l31 = bit l31 of keep word (n0 >> 5) + nt of the row
#pragma unroll
            for (int nt = 0; nt < 4; nt++) {
                if (n0 + nt * 32 >= N) continue;          // whole column tile past N (uniform)
                const int n = n0 + nt * 32 + l31;
                const bool nok = n < N;
                const int ncl = nok ? n : N - 1;
                float sg = 0.f, sb = 0.f;
                // buffer loads: descriptors in SGPRs, one 32-bit lane offset per group, the row step as a scalar offset (64-bit
                // per-load addresses cost two VGPRs each and spilled).  Rows past the end of a tensor read 0: rstd = 0 removes them.
#pragma unroll
                for (int rq = 0; rq < 4; rq++) {          // rows 8 rq + 4h + 0..3 of the tile
                    const int m0 = (int)(t * 32) + 4 * h + 8 * rq;
                    const int vx = (m0 * N + ncl) * 4, vm = m0 * 4;
                    const int vk = (((n0 >> 5) + nt) * (int)M + m0) * 4;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const float xv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(cs_rx, vx, j * N * 4, 0));
                        const float mu = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(cs_rm, vm, j * 4, 0));
                        const float rsd = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(cs_rr, vm, j * 4, 0));
                        const unsigned kw = cs.keep ? __builtin_amdgcn_raw_buffer_load_b32(cs_rk, vk, j * 4, 0) : 0xFFFFFFFFu;
                        const bool keep = nok && m0 + j < M && ((kw >> l31) & 1u);
                        const float gq = keep ? acc[nt][4 * rq + j] * cs.inv_keep : 0.f;
                        sb += gq;
                        sg += gq * ((xv - mu) * rsd);
                    }
                }
                cs_acc[nt * 32 + l31] += sg;
                cs_acc[SBN + nt * 32 + l31] += sb;
            }
            continue;
        }
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
            if (n0 + nt * 32 >= N) continue;              // whole column tile past N (uniform)
            const int n = n0 + nt * 32 + l31;
            const bool nok = n < N;
            const float bsv = bias_s[nt * 32 + l31];
            float* yp = Y + (t * 32 + 4 * h) * N + (nok ? n : N - 1);
            float rv[16];
            if (HAS_RES) {
                const float* rp = R + (t * 32 + 4 * h) * N + (nok ? n : N - 1);
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int dm = (r & 3) + 8 * (r >> 2);
                    rv[r] = (full || t * 32 + 4 * h + dm < M) ? rp[(long)dm * N] : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                float v = acc[nt][r] + bsv;
                if (relu) v = fmaxf(v, 0.f);
                if (HAS_RES) v += rv[r];
                acc[nt][r] = v;
            }
            if (MASK_OUT) {
                // ReLU bit mask: one ballot per accumulator register = the 32 columns of two rows (lane halves); lane
                // (l31 = r, h) keeps the word of its half, so after 16 ballots lanes l31 < 16 hold one row word each
                unsigned myw = 0u;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const unsigned long long bal = __ballot(nok && acc[nt][r] > 0.f);
                    const unsigned wsel = h ? (unsigned)(bal >> 32) : (unsigned)bal;
                    if (l31 == r) myw = wsel;
                }
                const long mrow = t * 32 + 4 * h + (l31 & 3) + 8 * ((l31 & 15) >> 2);
                if (l31 < 16 && mrow < M) mask_out[(long)((n0 >> 5) + nt) * M + mrow] = myw;   // [word][row]
            }
            if (full) {
#pragma unroll
                for (int r = 0; r < 16; r++)
                    if (nok && (!(GEMM_ABL & 8) || acc[nt][r] == 1.2345e30f))
                        __builtin_nontemporal_store(acc[nt][r], &yp[(long)((r & 3) + 8 * (r >> 2)) * N]);
            } else {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int dm = (r & 3) + 8 * (r >> 2);
                    if (nok && t * 32 + 4 * h + dm < M) yp[(long)dm * N] = acc[nt][r];
                }
            }
        }
    }
    if (COLSUM) {   // the two lane halves hold different rows of the same columns; one partial row per (row group, wave)
        float* prow = cs.part + ((long)bx * SWAVES + wave) * 2 * N;
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
            const float g1 = cs_acc[nt * 32 + l31], b1 = cs_acc[SBN + nt * 32 + l31];
            const float sg = xsum32(g1, g1), sb = xsum32(b1, b1);
            const int n = n0 + nt * 32 + l31;
            if (h == 0 && n < N) {
                prow[n] = sg;
                prow[N + n] = sb;
            }
        }
    }
}

// returns 1 if the shape/alignment is not handled here (caller falls back to the tiled kernel), 0 on launch, <0 on error
// gate_kind: 0 none, 1 fp32 tensor, 2 bit mask; mask_out (optional, relu only): ReLU bit mask of Y
int stage_gemm_nt_stream(const float* X, const void* gate, int gate_kind, const float* W, const float* bias,
                         const float* residual, float* Y, unsigned* mask_out, long long M, int N, int K, int relu,
                         void* stream) {
    if (!gate) gate_kind = 0;
    {   // developer bisect switch: STAGE_GEMM_STREAM_SKIP="n384" / "k300" / "g2" (gate kind) / "r" (residual) / "m" (mask out)
        static const char* skip = getenv("STAGE_GEMM_STREAM_SKIP");
        if (skip) {
            const int v = atoi(skip + 1);
            if ((skip[0] == 'n' && v == N) || (skip[0] == 'k' && v == K) || (skip[0] == 'g' && v == gate_kind) ||
                (skip[0] == 'r' && residual) || (skip[0] == 'm' && mask_out))
                return 1;
        }
    }
    const bool vec = (K % 4 == 0) && K >= 4 && (((uintptr_t)X & 15) == 0) && (((uintptr_t)W & 15) == 0) &&
                     (gate_kind != 1 || ((uintptr_t)gate & 15) == 0);
    // rows: from 4096; from 1024 for K <= 384, where a handful of workgroups with one 32-row tile per wave still beat the
    // tiled kernel's 128 x 128 tiles (M = 3200: 20 -> 12 us at K = 128, 35 -> 26 us at K = 384, but 64 -> 80 us at K = 768)
    static const long long min_m_env = getenv("STAGE_GEMM_STREAM_MIN_M") ? atoll(getenv("STAGE_GEMM_STREAM_MIN_M")) : 0;
    const long long min_m_nt = min_m_env ? min_m_env : (K <= 384 ? 1024 : 4096);
    if (!vec || M < min_m_nt || K < 64 || M * (long long)K * 4 >= (1ll << 31)) return 1;   // buffer addressing: < 2 GiB
    if ((gate_kind == 2 || mask_out) && residual) return 1;                              // combinations nobody needs
    if (mask_out && gate_kind != 0) return 1;
    const int lds = 3 * SPLANE * (int)sizeof(unsigned short) + SBN * (int)sizeof(float);
    const long MT = (M + 31) / 32, n_bt = (MT + SWAVES - 1) / SWAVES;
    const int n_tiles = (N + SBN - 1) / SBN;
    long gx = 256 / n_tiles;                              // one resident workgroup per CU; the column tiles of a row
    if (gx < 1) gx = 1;                                   // range run side by side
    if (gx > n_bt) gx = n_bt;
    // several column tiles and at least 8 row groups: XCD-aware 1-D grid (kernel comment), gx rounded down to a multiple of 8
    static const bool no_xcd = getenv("STAGE_GEMM_NO_XCD") != nullptr;
    int xcd_gx = 0;
    if (n_tiles > 1 && gx >= 8 && !no_xcd) {
        gx = gx / 8 * 8;
        xcd_gx = (int)gx;
    }
    dim3 grid(xcd_gx ? (unsigned)(gx * n_tiles) : (unsigned)gx, xcd_gx ? 1u : (unsigned)n_tiles), block(64 * SWAVES);
    const float* G = (const float*)gate;
#define LAUNCH_ST(GT, RS, MO)                                                                                          \
    do {                                                                                                               \
        static bool attr_done = false;                                                                                 \
        if (!attr_done) {                                                                                              \
            (void)hipFuncSetAttribute((const void*)gemm_nt_stream_kernel<GT, RS, !(GT == 1 && RS), MO>,                \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds);                                \
            attr_done = true;                                                                                          \
        }                                                                                                              \
        hipLaunchKernelGGL((gemm_nt_stream_kernel<GT, RS, !(GT == 1 && RS), MO>), grid, block, lds, (hipStream_t)stream, \
                           X, G, W, bias, residual, Y, mask_out, (long)M, N, K, relu, xcd_gx);                         \
    } while (0)
    if (mask_out) LAUNCH_ST(0, false, true);
    else if (gate_kind == 2) LAUNCH_ST(2, false, false);
    else if (gate_kind == 1) { if (residual) LAUNCH_ST(1, true, false); else LAUNCH_ST(1, false, false); }
    else { if (residual) LAUNCH_ST(0, true, false); else LAUNCH_ST(0, false, false); }
#undef LAUNCH_ST
    STAGE_LAUNCH_CHECK();
    return 0;
}

// LayerNorm gain / bias gradients straight from the dX product (kernel comment at StageColsum).  dY (M, K) [gated by the ReLU bit
// mask], Wt (N, K) = the Linear's weight transposed, x / mean / rstd = the LayerNorm's input and statistics, keep_mask = the keep
// bits of its dropout ([ceil(N/32)][M] words, stage_dropout_keepmask; NULL when p_drop == 0).  Workspace: stage_gemm_nt_lnparam_ws_bytes.  Returns 1 when the shape is not handled (caller: dX + LayerNorm backward).
static long colsum_rows(long long M, int N) {
    const long MT = (M + 31) / 32, n_bt = (MT + SWAVES - 1) / SWAVES;
    const int n_tiles = (N + SBN - 1) / SBN;
    long gx = 256 / n_tiles;
    if (gx < 1) gx = 1;
    if (gx > n_bt) gx = n_bt;
    if (n_tiles > 1 && gx >= 8) gx = gx / 8 * 8;
    return gx * SWAVES;
}
size_t stage_gemm_nt_lnparam_ws(long long M, int N) { return (size_t)colsum_rows(M, N) * 2 * (size_t)N * sizeof(float); }
int stage_gemm_nt_stream_lnparam(const float* dY, const unsigned* gate_mask, const float* Wt, const float* x, const float* mean,
                                 const float* rstd, const unsigned* keep_mask, float p_drop, float* dgamma, float* dbeta,
                                 long long M, int N, int K, void* ws, size_t ws_bytes, void* stream) {
    const bool vec = (K % 4 == 0) && K >= 4 && (((uintptr_t)dY & 15) == 0) && (((uintptr_t)Wt & 15) == 0);
    if (!STAGE_GEMM_NT_F16 || !vec || M < 4096 || K < 64 || N % 4 != 0 || M * (long long)K * 4 >= (1ll << 31) ||
        M * (long long)N * 4 >= (1ll << 31))
        return 1;
    if (ws_bytes < stage_gemm_nt_lnparam_ws(M, N)) return STAGE_ERR_WORKSPACE;
    const int lds = 3 * SPLANE * (int)sizeof(unsigned short) + SBN * (int)sizeof(float) + SWAVES * 2 * 2 * SBN * (int)sizeof(float);
    const long MT = (M + 31) / 32, n_bt = (MT + SWAVES - 1) / SWAVES;
    const int n_tiles = (N + SBN - 1) / SBN;
    long gx = 256 / n_tiles;
    if (gx < 1) gx = 1;
    if (gx > n_bt) gx = n_bt;
    int xcd_gx = 0;
    if (n_tiles > 1 && gx >= 8) {
        gx = gx / 8 * 8;
        xcd_gx = (int)gx;
    }
    dim3 grid(xcd_gx ? (unsigned)(gx * n_tiles) : (unsigned)gx, xcd_gx ? 1u : (unsigned)n_tiles), block(64 * SWAVES);
    StageColsum cs;
    cs.mean = mean;
    cs.rstd = rstd;
    cs.part = (float*)ws;
    const bool dr = p_drop > 0.f;
    if (dr && !keep_mask) return STAGE_ERR_SHAPE;
    cs.keep = dr ? keep_mask : nullptr;
    cs.inv_keep = dr ? 1.0f / (1.0f - p_drop) : 1.0f;
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_CS(GT)                                                                                                  \
    do {                                                                                                               \
        static bool attr_done = false;                                                                                 \
        if (!attr_done) {                                                                                              \
            (void)hipFuncSetAttribute((const void*)gemm_nt_stream_kernel<GT, false, false, false, true>,               \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds);                                \
            attr_done = true;                                                                                          \
        }                                                                                                              \
        hipLaunchKernelGGL((gemm_nt_stream_kernel<GT, false, false, false, true>), grid, block, lds, st, dY,            \
                           (const float*)gate_mask, Wt, (const float*)nullptr, x, (float*)nullptr, (unsigned*)nullptr, \
                           (long)M, N, K, 0, xcd_gx, cs);                                                              \
    } while (0)
    if (gate_mask) LAUNCH_CS(2);
    else LAUNCH_CS(0);
#undef LAUNCH_CS
    STAGE_LAUNCH_CHECK();
    // column c = t*N + n of the [2][N] partial rows goes to dgamma[n] (t = 0) or dbeta[n] (t = 1)
    stage_colreduce(cs.part, dgamma, dbeta, (int)colsum_rows(M, N), (long)2 * N, 2 * N, N, 1, st);
    STAGE_LAUNCH_CHECK();
    return 0;
}

// =====================================================================================================================
// Streaming TN GEMM (weight gradients):  part[s][n][k] = sum_{m in slab s} (dY (*) gate)[m,n] X[m,k] ,
//                                        part_b[s][n]  = sum_{m in slab s} (dY (*) gate)[m,n]
// The contraction runs over ROWS, so an MFMA operand lane (column c, k-half h) needs 8 consecutive rows of one column.
// Instead of transposing through LDS (the split-bf16 tiled TN kernel loses to the fp32 one on exactly that), every lane
// loads its 8 values with 8 dword loads: a load instruction then covers 2 rows x 128 contiguous bytes -- full cache
// lines, 4x more VMEM instructions than dwordx4 but no LDS, no barrier, no transpose.  The loads are buffer loads
// (resource + wave-uniform row offset in an SGPR + constant 32-bit lane offset): no 64-bit per-lane address arithmetic
// in the loop, and rows past the end of the tensor read as 0.  Each wave owns a 64 x 64 patch of the 128 x 128 output tile
// of its workgroup (2 x 2 MFMA tiles; this variant without operand sharing still runs the 3-way bf16 split) and walks the rows of its slab
// 16 at a time with the next step's 32 (48 with gate) loads in flight.
// =====================================================================================================================
__device__ __forceinline__ float s_buf_load(__amdgpu_buffer_rsrc_t r, int voff_bytes, int soff_bytes) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff_bytes, soff_bytes, 0));
}

template <int GATE>   // 0 none, 1 fp32 gate (same layout as dY), 2 bit mask (uint32 [ceil(N/32)][M], word-major)
__global__ __launch_bounds__(256, 2) void gemm_tn_stream_kernel(const float* __restrict__ dY, const float* __restrict__ G,
                                                                const float* __restrict__ X, float* __restrict__ part,
                                                                float* __restrict__ part_b, long M, int N, int K,
                                                                long rows_per_split) {
    // Two step buffers (loads one step ahead).  A third buffer / loads two steps ahead measured no faster (0.283 vs 0.284 ms
    // at 960000 x 128 x 128), i.e. this kernel is not bound by the latency of its dword loads.
    constexpr int NBUF = 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int pn = wave >> 1, pk = wave & 1;
    const int n_base = blockIdx.x * 128 + pn * 64, k_base = blockIdx.y * 128 + pk * 64;
    const int split = blockIdx.z;
    const long mbeg = (long)split * rows_per_split;
    const long mend = min(M, mbeg + rows_per_split);
    const int NWN = (N + 31) >> 5;                        // mask words per row
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)dY, 0, (int)(M * N * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)(M * K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(GATE != 0 ? G : dY), 0, GATE == 2 ? (int)(M * NWN * 4) : (int)(M * N * 4), 0x00020000);
    // this lane's two dY columns and two X columns (clamped for the address, zeroed by the flag) as byte offsets inside
    // a 16-row step
    int yoff[2], xoff[2], gbit[2];
    bool nok[2], kok[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int n = n_base + 32 * t + l31, k = k_base + 32 * t + l31;
        nok[t] = n < N;
        kok[t] = k < K;
        const int nc = nok[t] ? n : N - 1;
        yoff[t] = (8 * h * N + nc) * 4;
        xoff[t] = (8 * h * K + (kok[t] ? k : K - 1)) * 4;
        gbit[t] = nc & 31;
    }
    // GATE 2: byte offset of this lane's mask word column (word-major mask: [word][row]) for its row half
    int moff[2];
#pragma unroll
    for (int t = 0; t < 2; t++) moff[t] = (int)(((long)min((n_base >> 5) + t, NWN - 1) * M + 8 * h) * 4);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    float bsum[2] = {0.f, 0.f};

    float ya[NBUF][2][8], xa[NBUF][2][8], ga[GATE != 0 ? NBUF : 1][2][8];   // [buffer][tile][row]; GATE 2: mask words
    auto fetch = [&](int buf, long m0) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int sy = (int)((m0 + r) * N * 4), sx = (int)((m0 + r) * K * 4);   // wave-uniform row offsets (bytes)
#pragma unroll
            for (int t = 0; t < 2; t++) {
                ya[buf][t][r] = s_buf_load(ry, yoff[t], sy);
                xa[buf][t][r] = s_buf_load(rx, xoff[t], sx);
                if (GATE == 1) ga[buf][t][r] = s_buf_load(rg, yoff[t], sy);
                if (GATE == 2) ga[buf][t][r] = s_buf_load(rg, moff[t], (int)((m0 + r) * 4));   // rows past the end read 0
            }
        }
    };
    // The SIMD issues one instruction at a time and only ~5-7 of them hide behind each MFMA: at ~17 non-MFMA instructions
    // per MFMA this kernel is issue-bound, so the steady state (full 16-row step, all four column tiles inside the matrix)
    // drops the per-element validity selects, and only the waves that emit the bias gradient keep its running sums.
    const bool cols_in = nok[0] && nok[1] && kok[0] && kok[1];           // lane-level, but false only in edge tiles
    const bool want_b = part_b != nullptr && blockIdx.y == 0 && pk == 0;
    auto step = [&](int buf, long m0) {
        const bool full = m0 + 16 <= mend;
        const bool fast = full && __all(cols_in);
        sbf16x8 a[2][3], b[2][3];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            float yv[8], xv[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                float y = ya[buf][t][r];
                if (GATE == 1) y = ga[buf][t][r] > 0.f ? y : 0.f;
                if (GATE == 2) y = ((__float_as_uint(ga[buf][t][r]) >> gbit[t]) & 1u) ? y : 0.f;
                yv[r] = y;
                xv[r] = xa[buf][t][r];
            }
            if (!fast) {   // one uniform branch per tile, straight-line selects inside
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const bool rok = full || (m0 + 8 * h + r < mend);   // rows of the next slab / past the end contribute 0
                    yv[r] = (rok && nok[t]) ? yv[r] : 0.f;
                    xv[r] = (rok && kok[t]) ? xv[r] : 0.f;
                }
            }
            if (want_b) {
#pragma unroll
                for (int r = 0; r < 8; r++) bsum[t] += yv[r];
            }
            unsigned p[4][3], q[4][3];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                s_split3(yv[2 * i], yv[2 * i + 1], p[i]);
                s_split3(xv[2 * i], xv[2 * i + 1], q[i]);
            }
#pragma unroll
            for (int s = 0; s < STAGE_GEMM_TERMS; s++) {
                a[t][s] = __builtin_bit_cast(sbf16x8, make_uint4(p[0][s], p[1][s], p[2][s], p[3][s]));
                b[t][s] = __builtin_bit_cast(sbf16x8, make_uint4(q[0][s], q[1][s], q[2][s], q[3][s]));
            }
        }
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                // kept cross terms, smallest first
                acc[i][j] = s_mfma_terms(a[i], b[j], acc[i][j]);
            }
    };
    if (mbeg < mend) {
        if (NBUF == 2) {
            fetch(0, mbeg);
            for (long m0 = mbeg; m0 < mend; m0 += 32) {
                if (m0 + 16 < mend) fetch(1, m0 + 16);
                step(0, m0);
                if (m0 + 16 < mend) {
                    if (m0 + 32 < mend) fetch(0, m0 + 32);
                    step(1, m0 + 16);
                }
            }
        } else {
            // loads two steps ahead (rows past the slab are masked in step, rows past the tensor read 0)
            fetch(0, mbeg);
            fetch(1, mbeg + 16);
            for (long m0 = mbeg; m0 < mend; m0 += 48) {
                fetch(2 % NBUF, m0 + 32);
                step(0, m0);
                if (m0 + 16 < mend) {
                    fetch(0, m0 + 48);
                    step(1, m0 + 16);
                }
                if (m0 + 32 < mend) {
                    fetch(1, m0 + 64);
                    step(2 % NBUF, m0 + 32);
                }
            }
        }
    }
    // C/D layout of the 32x32 MFMA: col = lane & 31 (k), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (n)
    float* po = part + (size_t)split * N * K;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int k = k_base + 32 * j + l31;
            if (k >= K) continue;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int n = n_base + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (n < N) po[(size_t)n * K + k] = acc[i][j][r];
            }
        }
    if (want_b) {
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const float s = xsum32(bsum[t], bsum[t]);
            const int n = n_base + 32 * t + l31;
            if (h == 0 && n < N) part_b[(size_t)split * N + n] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Same patch layout, but every 32-column operand tile is loaded, gated and split by ONE wave and handed to the wave that
// shares it through LDS: wave (pn, pk) prepares dY tile pk of its row patch (shared with wave (pn, 1 - pk)) and X tile pn
// of its column patch (shared with wave (1 - pn, pk)).  Matrix-core and VALU time of a SIMD add (DESIGN.md finding 13),
// and this kernel spent ~480 VALU cycles per step next to 768 MFMA cycles on work that was done twice: per wave and step
// now 16 + 8 loads (was 32 + 16), 16 operand values to gate and split (was 32), plus 6 ds_write_b128 / 12 ds_read_b128 of
// finished fragments (identical lane mapping on both sides) and one barrier.  Two exchange buffers: a wave that runs ahead
// writes the buffer of step i + 1 only after everyone has passed the barrier of step i, i.e. finished reading step i - 1.
// ---------------------------------------------------------------------------------------------------------------------
template <int GATE>
__global__ __launch_bounds__(256, 2) void gemm_tn_share_kernel(const float* __restrict__ dY, const float* __restrict__ G,
                                                               const float* __restrict__ X, float* __restrict__ part,
                                                               float* __restrict__ part_b, long M, int N, int K,
                                                               long rows_per_split) {
    __shared__ uint4 ex[2][4][6][64];                     // [buffer][wave][3 dY planes, 3 X planes][lane]  (48 KB)
    __shared__ int exd[2][4][2][64];                      // fp16 mode: exponent change of the dY / X tile column at this step
    int eby = 0, ebx = 0;                                 // fp16 mode: running exponents of the columns this lane prepares
    int upA[2] = {254, 254}, upB[2] = {254, 254};         // and the scale fields in force for the tiles this wave consumes
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int pn = wave >> 1, pk = wave & 1;
    const int n_base = blockIdx.x * 128 + pn * 64, k_base = blockIdx.y * 128 + pk * 64;
    const int split = blockIdx.z;
    const long mbeg = (long)split * rows_per_split;
    const long mend = min(M, mbeg + rows_per_split);
    const int NWN = (N + 31) >> 5;                        // mask words per row
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)dY, 0, (int)(M * N * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)(M * K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(GATE != 0 ? G : dY), 0, GATE == 2 ? (int)(M * NWN * 4) : (int)(M * N * 4), 0x00020000);
    // the dY column (tile pk of the patch) and the X column (tile pn of the patch) this lane prepares
    const int n_mine = n_base + 32 * pk + l31, k_mine = k_base + 32 * pn + l31;
    const bool nok = n_mine < N, kok = k_mine < K;
    const int nc = nok ? n_mine : N - 1;
    const int yoff = (8 * h * N + nc) * 4, xoff = (8 * h * K + (kok ? k_mine : K - 1)) * 4, gbit = nc & 31;
    const int moff = (int)(((long)min((n_base >> 5) + pk, NWN - 1) * M + 8 * h) * 4);   // GATE 2: [word][row] mask
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    float bsum = 0.f;
    float ya[2][8], xa[2][8], ga[GATE != 0 ? 2 : 1][8];
    bool abl_f[2] = {false, false};
    auto fetch = [&](int buf, long m0) {
        if ((GEMM_ABL & 4) && abl_f[buf]) return;
        abl_f[buf] = true;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int sy = (int)((m0 + r) * N * 4), sx = (int)((m0 + r) * K * 4);   // wave-uniform row offsets (bytes)
            ya[buf][r] = s_buf_load(ry, yoff, sy);
            xa[buf][r] = s_buf_load(rx, xoff, sx);
            if (GATE == 1) ga[buf][r] = s_buf_load(rg, yoff, sy);
            if (GATE == 2) ga[buf][r] = s_buf_load(rg, moff, (int)((m0 + r) * 4));   // rows past the end read 0
        }
    };
    const bool cols_in = nok && kok;
    const bool want_b = part_b != nullptr && blockIdx.y == 0;   // every wave owns the bias sums of the dY tile it prepares
    auto step = [&](int buf, long m0, int xb) {
        const bool full = m0 + 16 <= mend;
        const bool fast = full && __all(cols_in);
        float yv[8], xv[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            float y = ya[buf][r];
            if (GATE == 1) y = ga[buf][r] > 0.f ? y : 0.f;
            if (GATE == 2) y = ((__float_as_uint(ga[buf][r]) >> gbit) & 1u) ? y : 0.f;
            yv[r] = y;
            xv[r] = xa[buf][r];
        }
        if (!fast) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const bool rok = full || (m0 + 8 * h + r < mend);   // rows of the next slab / past the end contribute 0
                yv[r] = (rok && nok) ? yv[r] : 0.f;
                xv[r] = (rok && kok) ? xv[r] : 0.f;
            }
        }
        if (want_b) {
#pragma unroll
            for (int r = 0; r < 8; r++) bsum += yv[r];
        }
        if (STAGE_GEMM_TN_F16) {
            const int dy_ = h_track8(yv, eby), dx_ = h_track8(xv, ebx);
            uint4 yh, yl, xh, xl;
            h_split8(yv, eby, yh, yl);
            h_split8(xv, ebx, xh, xl);
            ex[xb][wave][0][lane] = yh;
            ex[xb][wave][1][lane] = yl;
            ex[xb][wave][3][lane] = xh;
            ex[xb][wave][4][lane] = xl;
            exd[xb][wave][0][lane] = dy_;
            exd[xb][wave][1][lane] = dx_;
            __syncthreads();
            sf16x8 a[2][2], b[2][2];
            int dA[2], dB[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++) {
                    a[t][s2] = __builtin_bit_cast(sf16x8, ex[xb][pn * 2 + t][s2][lane]);
                    b[t][s2] = __builtin_bit_cast(sf16x8, ex[xb][t * 2 + pk][3 + s2][lane]);
                }
                dA[t] = exd[xb][pn * 2 + t][0][lane];
                dB[t] = exd[xb][t * 2 + pk][1][lane];
            }
            if (__any((dA[0] | dA[1] | dB[0] | dB[1]) != 0)) {   // a column's scale moved: bring the accumulators along
#pragma unroll
                for (int i = 0; i < 2; i++) {
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int dr = __builtin_amdgcn_ds_bpermute(4 * h_row_of(r, h), dA[i]);
#pragma unroll
                        for (int j = 0; j < 2; j++) acc[i][j][r] = __builtin_ldexpf(acc[i][j][r], dr + dB[j]);
                    }
                    upA[i] += dA[i];
                    upB[i] += dB[i];
                }
            }
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = h_mfma_terms(a[i], b[j], acc[i][j]);
            return;
        }
        unsigned p[4][3], q[4][3];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (GEMM_ABL & 2) {
                p[i][0] = __float_as_uint(yv[2 * i]); p[i][1] = __float_as_uint(yv[2 * i + 1]); p[i][2] = p[i][0] ^ p[i][1];
                q[i][0] = __float_as_uint(xv[2 * i]); q[i][1] = __float_as_uint(xv[2 * i + 1]); q[i][2] = q[i][0] ^ q[i][1];
                continue;
            }
            s_split3(yv[2 * i], yv[2 * i + 1], p[i]);
            s_split3(xv[2 * i], xv[2 * i + 1], q[i]);
        }
#pragma unroll
        for (int s = 0; s < STAGE_GEMM_TERMS; s++) {
            ex[xb][wave][s][lane] = make_uint4(p[0][s], p[1][s], p[2][s], p[3][s]);
            ex[xb][wave][3 + s][lane] = make_uint4(q[0][s], q[1][s], q[2][s], q[3][s]);
        }
        if (!(GEMM_ABL & 32)) __syncthreads();
        // dY tile t of this row patch was prepared by wave (pn, t), X tile t of this column patch by wave (t, pk); the own
        // fragments are read back as well: a register array indexed by the wave's pk / pn would live in scratch
        sbf16x8 a[2][3], b[2][3];
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int s = 0; s < STAGE_GEMM_TERMS; s++) {
                a[t][s] = __builtin_bit_cast(sbf16x8, ex[xb][pn * 2 + t][s][lane]);
                b[t][s] = __builtin_bit_cast(sbf16x8, ex[xb][t * 2 + pk][3 + s][lane]);
            }
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                if (GEMM_ABL & 1) {
                    const sbf16x8 am = __builtin_bit_cast(sbf16x8, __builtin_bit_cast(uint4, a[i][0]) ^ __builtin_bit_cast(uint4, a[i][1]) ^ __builtin_bit_cast(uint4, a[i][2]));
                    const sbf16x8 bm = __builtin_bit_cast(sbf16x8, __builtin_bit_cast(uint4, b[j][0]) ^ __builtin_bit_cast(uint4, b[j][1]) ^ __builtin_bit_cast(uint4, b[j][2]));
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[i][j], 0, 0, 0);
                    continue;
                }
                // kept cross terms, smallest first
                acc[i][j] = s_mfma_terms(a[i], b[j], acc[i][j]);
            }
    };
    // the loop bounds are workgroup-uniform (every wave of a workgroup walks the same slab), so the barriers match
    if (mbeg < mend) {
        fetch(0, mbeg);
        for (long m0 = mbeg; m0 < mend; m0 += 32) {
            if (m0 + 16 < mend) fetch(1, m0 + 16);
            step(0, m0, 0);
            if (m0 + 16 < mend) {
                if (m0 + 32 < mend) fetch(0, m0 + 32);
                step(1, m0 + 16, 1);
            }
        }
    }
    if (STAGE_GEMM_TN_F16) {                              // back to true units: 2^-(dY column scale + X column scale)
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int ua = __builtin_amdgcn_ds_bpermute(4 * h_row_of(r, h), upA[i]);
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j][r] = __builtin_ldexpf(acc[i][j][r], 254 - ua - upB[j]);
            }
    }
    // C/D layout of the 32x32 MFMA: col = lane & 31 (k), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (n)
    float* po = part + (size_t)split * N * K;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int k = k_base + 32 * j + l31;
            if (k >= K) continue;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int n = n_base + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (n < N) po[(size_t)n * K + k] = acc[i][j][r];
            }
        }
    if (want_b) {
        const float s = xsum32(bsum, bsum);
        if (h == 0 && nok) part_b[(size_t)split * N + n_mine] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Wide variant for K > 128: a workgroup of 8 waves (2 x 4) owns a 128 x 384 output tile, wave (pn, pk) the 64 x 96 patch
// (2 x 3 MFMA tiles, 96 accumulator registers).  The 16 operand tiles of a 16-row step (4 of dY, 12 of X) are prepared
// by the 8 waves, two each (wave w: units w and w + 8; units 0..3 = dY tiles), and exchanged through LDS as above.  Against
// three 128 x 128 workgroups: dY is read (and gated, split) once instead of three times, and every wave issues 36 MFMAs
// per 16 prepared operand values instead of 24.
// ---------------------------------------------------------------------------------------------------------------------
#define TW_NA 4         // dY tiles per workgroup
#define TW_NB 12        // X tiles per workgroup
#define TW_WAVES 8
template <int GATE>
__global__ __launch_bounds__(64 * TW_WAVES, 2) void gemm_tn_wide_kernel(const float* __restrict__ dY, const float* __restrict__ G,
                                                                        const float* __restrict__ X, float* __restrict__ part,
                                                                        float* __restrict__ part_b, long M, int N, int K,
                                                                        long rows_per_split) {
    extern __shared__ __attribute__((aligned(16))) uint4 exw[];   // [2 buffers][16 units][3 planes][64 lanes]  (96 KB)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // SGPR: no waterfall loops
    const int l31 = lane & 31, h = lane >> 5;
    const int pn = wave >> 2, pk = wave & 3;
    const int n0 = blockIdx.x * 128, k0 = blockIdx.y * 384;
    const int split = blockIdx.z;
    const long mbeg = (long)split * rows_per_split;
    const long mend = min(M, mbeg + rows_per_split);
    const int NWN = (N + 31) >> 5;                        // mask words per row
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)dY, 0, (int)(M * N * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)(M * K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(GATE != 0 ? G : dY), 0, GATE == 2 ? (int)(M * NWN * 4) : (int)(M * N * 4), 0x00020000);
    // the two operand tiles ("units") this wave prepares: unit 0 of waves 0..3 is a dY tile, everything else an X tile
    const bool u0_is_y = wave < TW_NA;                    // wave-uniform
    const int y_col = n0 + 32 * wave + l31;               // only meaningful when u0_is_y
    const bool y_ok = u0_is_y && y_col < N;
    const int ycl = y_col < N ? y_col : N - 1;
    const int yoff = (8 * h * N + ycl) * 4, gbit = ycl & 31;
    const int moff = (int)(((long)min((n0 >> 5) + wave, NWN - 1) * M + 8 * h) * 4);   // GATE 2: [word][row] mask
    int xcol[2];
    bool x_ok[2];
    int xoff[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int bt = wave + TW_WAVES * u - TW_NA;       // X tile index of unit u (negative: unit 0 of a dY wave)
        xcol[u] = k0 + 32 * (bt < 0 ? 0 : bt) + l31;
        x_ok[u] = bt >= 0 && xcol[u] < K;
        xoff[u] = (8 * h * K + (xcol[u] < K ? xcol[u] : K - 1)) * 4;
    }
    f32x16 acc[2][3];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    float bsum = 0.f;
    int ebu[2] = {0, 0};                                  // fp16 mode: running exponents of the two unit columns this lane prepares
    int upA[2] = {254, 254}, upB[3] = {254, 254, 254};    // and the scale fields in force for the tiles this wave consumes
    float va[2][2][8], ga[GATE != 0 ? 2 : 1][8];          // [buffer][unit][row]
    bool abl_fw[2] = {false, false};
    auto fetch = [&](int buf, long m0) {
        if ((GEMM_ABL & 4) && abl_fw[buf]) return;
        abl_fw[buf] = true;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int sy = (int)((m0 + r) * N * 4), sx = (int)((m0 + r) * K * 4);   // wave-uniform row offsets (bytes)
            // unit 0: one load through a wave-uniformly selected descriptor / offset (dY for waves 0..3, X otherwise); the
            // gate is fetched by every wave (valid clamped address, ignored by the X waves): a conditionally defined
            // register array ends up in scratch
            va[buf][0][r] = s_buf_load(u0_is_y ? ry : rx, u0_is_y ? yoff : xoff[0], u0_is_y ? sy : sx);
            if (GATE == 1) ga[buf][r] = s_buf_load(rg, yoff, sy);
            if (GATE == 2) ga[buf][r] = s_buf_load(rg, moff, (int)((m0 + r) * 4));   // rows past the end read 0
            va[buf][1][r] = s_buf_load(rx, xoff[1], sx);
        }
    };
    const bool want_b = part_b != nullptr && blockIdx.y == 0 && u0_is_y;
    const int u_mine[2] = {wave, wave + TW_WAVES};
    auto step = [&](int buf, long m0, int xb) {
        const bool full = m0 + 16 <= mend;
        uint4* exb = exw + (size_t)xb * (TW_NA + TW_NB) * 3 * 64;
#pragma unroll
        for (int u = 0; u < 2; u++) {
            float v[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                float y = va[buf][u][r];
                if (u == 0 && u0_is_y) {
                    if (GATE == 1) y = ga[buf][r] > 0.f ? y : 0.f;
                    if (GATE == 2) y = ((__float_as_uint(ga[buf][r]) >> gbit) & 1u) ? y : 0.f;
                }
                const bool rok = full || (m0 + 8 * h + r < mend);   // rows of the next slab / past the end contribute 0
                const bool cok = (u == 0 && u0_is_y) ? y_ok : x_ok[u];
                v[r] = (rok && cok) ? y : 0.f;
            }
            if (u == 0 && want_b) {
#pragma unroll
                for (int r = 0; r < 8; r++) bsum += v[r];
            }
            if (STAGE_GEMM_TN_F16) {   // planes 0 / 1 = hi / lo, the slot of plane 2 carries the exponent change of the column
                const int du = h_track8(v, ebu[u]);
                uint4 vh, vl;
                h_split8(v, ebu[u], vh, vl);
                exb[(u_mine[u] * 3 + 0) * 64 + lane] = vh;
                exb[(u_mine[u] * 3 + 1) * 64 + lane] = vl;
                reinterpret_cast<int*>(&exb[(u_mine[u] * 3 + 2) * 64])[lane] = du;
                continue;
            }
            unsigned p[4][3];
#pragma unroll
            for (int i = 0; i < 4; i++) s_split3(v[2 * i], v[2 * i + 1], p[i]);
#pragma unroll
            for (int s = 0; s < STAGE_GEMM_TERMS; s++) exb[(u_mine[u] * 3 + s) * 64 + lane] = make_uint4(p[0][s], p[1][s], p[2][s], p[3][s]);
        }
        __syncthreads();
        if (STAGE_GEMM_TN_F16) {
            sf16x8 a[2][2];
            int dA[2], dB[3];
#pragma unroll
            for (int t = 0; t < 2; t++) {
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++) a[t][s2] = __builtin_bit_cast(sf16x8, exb[((pn * 2 + t) * 3 + s2) * 64 + lane]);
                dA[t] = reinterpret_cast<const int*>(&exb[((pn * 2 + t) * 3 + 2) * 64])[lane];
            }
#pragma unroll
            for (int j = 0; j < 3; j++) dB[j] = reinterpret_cast<const int*>(&exb[((TW_NA + pk * 3 + j) * 3 + 2) * 64])[lane];
            if (__any((dA[0] | dA[1] | dB[0] | dB[1] | dB[2]) != 0)) {   // a column's scale moved: bring the accumulators along
#pragma unroll
                for (int i = 0; i < 2; i++) {
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int dr = __builtin_amdgcn_ds_bpermute(4 * h_row_of(r, h), dA[i]);
#pragma unroll
                        for (int j = 0; j < 3; j++) acc[i][j][r] = __builtin_ldexpf(acc[i][j][r], dr + dB[j]);
                    }
                    upA[i] += dA[i];
                }
#pragma unroll
                for (int j = 0; j < 3; j++) upB[j] += dB[j];
            }
#pragma unroll
            for (int j = 0; j < 3; j++) {
                sf16x8 b[2];
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++) b[s2] = __builtin_bit_cast(sf16x8, exb[((TW_NA + pk * 3 + j) * 3 + s2) * 64 + lane]);
#pragma unroll
                for (int i = 0; i < 2; i++) acc[i][j] = h_mfma_terms(a[i], b, acc[i][j]);
            }
            return;
        }
        sbf16x8 a[2][3];
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int s = 0; s < STAGE_GEMM_TERMS; s++) a[t][s] = __builtin_bit_cast(sbf16x8, exb[((pn * 2 + t) * 3 + s) * 64 + lane]);
#pragma unroll
        for (int j = 0; j < 3; j++) {
            sbf16x8 b[3];
#pragma unroll
            for (int s = 0; s < STAGE_GEMM_TERMS; s++) b[s] = __builtin_bit_cast(sbf16x8, exb[((TW_NA + pk * 3 + j) * 3 + s) * 64 + lane]);
#pragma unroll
            for (int i = 0; i < 2; i++) {
                // kept cross terms, smallest first
                acc[i][j] = s_mfma_terms(a[i], b, acc[i][j]);
            }
        }
    };
    if (mbeg < mend) {   // workgroup-uniform bounds: the barriers match
        fetch(0, mbeg);
        for (long m0 = mbeg; m0 < mend; m0 += 32) {
            if (m0 + 16 < mend) fetch(1, m0 + 16);
            step(0, m0, 0);
            if (m0 + 16 < mend) {
                if (m0 + 32 < mend) fetch(0, m0 + 32);
                step(1, m0 + 16, 1);
            }
        }
    }
    // C/D layout of the 32x32 MFMA: col = lane & 31 (k), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (n)
    if (STAGE_GEMM_TN_F16) {                              // back to true units: 2^-(dY column scale + X column scale)
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int ua = __builtin_amdgcn_ds_bpermute(4 * h_row_of(r, h), upA[i]);
#pragma unroll
                for (int j = 0; j < 3; j++) acc[i][j][r] = __builtin_ldexpf(acc[i][j][r], 254 - ua - upB[j]);
            }
    }
    float* po = part + (size_t)split * N * K;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int k = k0 + 32 * (pk * 3 + j) + l31;
            if (k >= K) continue;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int n = n0 + 32 * (pn * 2 + i) + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (n < N) po[(size_t)n * K + k] = acc[i][j][r];
            }
        }
    if (want_b) {
        const float s = xsum32(bsum, bsum);
        if (h == 0 && y_ok) part_b[(size_t)split * N + y_col] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// "Quad" variant of the wide kernel (K > 128, N % 4 == 0, K % 4 == 0, no fp32 gate): the same 128 x 384 output tile and
// 2 x 4 consumer waves, but the operands are fetched with 16-byte loads along the ROWS.  The dword-per-row loads of the
// kernels above cost 40 % of their run time (ablation: 697 -> 416 us at 960000 x 128^T x 384 with the operand loads
// removed).  What makes row-wise loads possible without a transpose: the output index of a GEMM may be permuted freely, so
// the four MFMA tiles of a 128-column group ("quad") are INTERLEAVED -- tile t holds the columns 4 l + t (l = lane position
// 0..31).  A lane that loads the float4 at columns 4l..4l+3 of eight rows then holds one 8-row operand fragment for each of
// the four tiles, at its own lane position.  Work split of a 32-row double step (two 16-row MFMA steps, ONE barrier): wave w
// prepares quad w >> 1 (0 = dY, 1..3 = X) for the column groups l = 16 (w & 1) + (lane & 15); its lane group g = lane >> 4
// takes MFMA step g >> 1, row half g & 1 -- eight dwordx4 loads per lane and double step instead of 32 dword loads, every
// load instruction four rows x 256 contiguous bytes.  One scale exponent per COLUMN, tracked by the producer as in the
// kernels above (a shared exponent for the four columns of a lane would save 30 VALU instructions per double step, but a
// weak column next to a strong one would lose its relative accuracy -- and Adam normalises every weight's gradient).
// ---------------------------------------------------------------------------------------------------------------------
// NXQ = X quads per workgroup: 3 (K > 128: 128 x 384 tile, 8 waves, wave patch 64 x 96) or 1 (K <= 128: 128 x 128 tile, 4 waves,
// wave patch 64 x 64, two workgroups per CU)
template <int GATE, int NXQ>     // GATE: 0 none, 2 bit mask
__global__ __launch_bounds__(128 * (1 + NXQ), 2) void gemm_tn_quad_kernel(const float* __restrict__ dY, const unsigned* __restrict__ G,
                                                                        const float* __restrict__ X, float* __restrict__ part,
                                                                        float* __restrict__ part_b, long M, int N, int K,
                                                                        long rows_per_split) {
    // [2 buffers][2 MFMA steps][TQ_TILES][2 planes][64 lanes] uint4 (128 / 64 KB), then [2 buffers][TQ_TILES][32] exponent changes
    constexpr int TQ_TILES = 4 * (1 + NXQ);           // operand tiles per workgroup: 4 of dY (one quad), 4 per X quad
    constexpr int TJ = NXQ == 3 ? 3 : 2;              // X tiles per consumer wave
    extern __shared__ __attribute__((aligned(16))) uint4 exq[];
    int* exd = reinterpret_cast<int*>(exq + 2 * 2 * TQ_TILES * 2 * 64);
    // the wave index decides descriptor and leading dimension of the loads: as an SGPR value (the compiler cannot prove
    // threadIdx.x >> 6 uniform and would wrap every buffer load into a waterfall loop)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int pn = NXQ == 3 ? wave >> 2 : wave >> 1, pk = NXQ == 3 ? wave & 3 : wave & 1;   // consumer role: rows pn (2 dY tiles), columns pk (TJ X tiles)
    const int n0 = blockIdx.x * 128, k0 = blockIdx.y * 128 * NXQ;
    const int split = blockIdx.z;
    const long mbeg = (long)split * rows_per_split;
    const long mend = min(M, mbeg + rows_per_split);
    const int NWN = (N + 31) >> 5;                        // mask words per row
    // producer role
    const int q = wave >> 1;                              // quad: 0 = dY, 1..3 = X
    const bool is_y = q == 0;                             // wave-uniform
    const int pl = 16 * (wave & 1) + (lane & 15);         // column group (lane position of the fragments it writes)
    const int pg = lane >> 4, ps = pg >> 1, ph = pg & 1;  // MFMA step and row half of its eight rows
    const int ld = is_y ? N : K;
    const int col = is_y ? n0 + 4 * pl : k0 + 128 * (q - 1) + 4 * pl;
    const bool col_ok = col < ld;                         // N, K are multiples of 4: a group is inside or outside as a whole
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(is_y ? dY : X), 0, (int)(M * ld * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)(GATE == 2 ? (const void*)G : (const void*)dY), 0,
                                                                        GATE == 2 ? (int)(M * NWN * 4) : (int)(M * N * 4), 0x00020000);
    const int voff = ((16 * ps + 8 * ph) * ld + (col_ok ? col : ld - 4)) * 4;
    const int gcol = min(n0 + 4 * pl, N - 4);
    const int gbit = gcol & 31;                           // bits gbit .. gbit + 3 of the mask word belong to this group
    const int moff = (int)(((long)(gcol >> 5) * M + 16 * ps + 8 * ph) * 4);   // [word][row] mask
    f32x16 acc[2][TJ];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < TJ; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    int eb[4] = {0, 0, 0, 0};                             // running exponents of the four columns this lane prepares
    int upA[2] = {254, 254}, upB[3] = {254, 254, 254};    // (TJ of upB used)    // scale fields in force for the tiles this wave consumes
    const bool want_b = part_b != nullptr && blockIdx.y == 0 && is_y;

    typedef unsigned tq_u4 __attribute__((ext_vector_type(4)));
    tq_u4 va[8];
    tq_u4 gw[2];                                          // GATE 2: the mask words of the lane's 8 rows ([word][row]: contiguous)
    auto fetch = [&](long m0) {
#pragma unroll
        for (int r = 0; r < 8; r++)
            va[r] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (int)((m0 + r) * ld * 4), 0);   // rows past the end read 0
        if (GATE == 2) {
            // word rows start at word * M: 16-byte aligned when M % 4 == 0 (the dense shapes), dword aligned otherwise (ragged row
            // counts; buffer loads take any dword alignment -- round 4: falling back to the dword-load kernels for M % 4 != 0
            // cost 0.4 ms per step).  Slabs start at multiples of 16
            gw[0] = __builtin_amdgcn_raw_buffer_load_b128(rg, moff, (int)(m0 * 4), 0);
            gw[1] = __builtin_amdgcn_raw_buffer_load_b128(rg, moff, (int)((m0 + 4) * 4), 0);
        }
    };
    auto produce = [&](long m0, int xb) {
        const bool full = m0 + 32 <= mend;
        float v[4][8];                                    // [tile][row]
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int t = 0; t < 4; t++) {
                float y = __uint_as_float(va[r][t]);
                if (GATE == 2 && is_y) y = ((gw[r >> 2][r & 3] >> (gbit + t)) & 1u) ? y : 0.f;
                v[t][r] = y;
            }
        if (!(full && __all(col_ok))) {                   // edge tiles / last double step of the slab (one uniform branch)
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const bool rok = (full || (m0 + 16 * ps + 8 * ph + r < mend)) && col_ok;   // rows of the next slab contribute 0
#pragma unroll
                for (int t = 0; t < 4; t++) v[t][r] = rok ? v[t][r] : 0.f;
            }
        }
        if (want_b) {
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int r = 0; r < 8; r++) bsum[t] += v[t][r];
        }
        // per column: largest magnitude of the lane's 8 rows, then of the four lanes (lane groups g = 0..3) that hold the
        // column's 32 rows
        uint4* exb = exq + (size_t)((xb * 2 + ps) * TQ_TILES + 4 * q) * 2 * 64 + (ph * 32 + pl);
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const float m = xmax32(xmax16(h_amax8(v[t])));
            const int ec = (int)(__float_as_uint(m) >> 23) & 0xff;
            const int neb = ec > eb[t] + 3 ? ec : eb[t];
            const int d = h_up_field(neb) - h_up_field(eb[t]);
            eb[t] = neb;
            uint4 vh, vl;
            h_split8(v[t], neb, vh, vl);
            exb[(t * 2 + 0) * 64] = vh;
            exb[(t * 2 + 1) * 64] = vl;
            if (pg == 0) exd[(xb * TQ_TILES + 4 * q + t) * 32 + pl] = d;
        }
    };
    auto consume = [&](int xb) {
        int dA[2], dB[3] = {0, 0, 0};
#pragma unroll
        for (int i = 0; i < 2; i++) dA[i] = exd[(xb * TQ_TILES + pn * 2 + i) * 32 + l31];
#pragma unroll
        for (int j = 0; j < TJ; j++) dB[j] = exd[(xb * TQ_TILES + 4 + pk * TJ + j) * 32 + l31];
        if (__any((dA[0] | dA[1] | dB[0] | dB[1] | dB[2]) != 0)) {   // (dB[2] stays 0 for TJ == 2)   // a column's scale moved: bring the accumulators along
#pragma unroll
            for (int i = 0; i < 2; i++) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int dr = __builtin_amdgcn_ds_bpermute(4 * h_row_of(r, h), dA[i]);
#pragma unroll
                    for (int j = 0; j < TJ; j++) acc[i][j][r] = __builtin_ldexpf(acc[i][j][r], dr + dB[j]);
                }
                upA[i] += dA[i];
            }
#pragma unroll
            for (int j = 0; j < TJ; j++) upB[j] += dB[j];
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
            const uint4* exb = exq + (size_t)((xb * 2 + s2) * TQ_TILES) * 2 * 64 + lane;
            sf16x8 a[2][2];
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int p2 = 0; p2 < 2; p2++) a[i][p2] = __builtin_bit_cast(sf16x8, exb[((pn * 2 + i) * 2 + p2) * 64]);
#pragma unroll
            for (int j = 0; j < TJ; j++) {
                sf16x8 b[2];
#pragma unroll
                for (int p2 = 0; p2 < 2; p2++) b[p2] = __builtin_bit_cast(sf16x8, exb[((4 + pk * TJ + j) * 2 + p2) * 64]);
#pragma unroll
                for (int i = 0; i < 2; i++) acc[i][j] = h_mfma_terms(a[i], b, acc[i][j]);
            }
        }
    };
    if (mbeg < mend) {   // workgroup-uniform bounds: the barriers match
        fetch(mbeg);
        int xb = 0;
        for (long m0 = mbeg; m0 < mend; m0 += 32, xb ^= 1) {
            produce(m0, xb);
            // the raw values are dead: the next double step loads behind the MFMAs (requesting it before the split instead --
            // both register sets live -- measured the same: 573 vs 573 us)
            if (m0 + 32 < mend && !((GEMM_ABL & 4) && m0 > mbeg)) fetch(m0 + 32);
            __syncthreads();
            consume(xb);
        }
    }
    // C/D layout of the 32x32 MFMA: col = lane & 31 -> k group, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) -> n group
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int ua = __builtin_amdgcn_ds_bpermute(4 * h_row_of(r, h), upA[i]);
#pragma unroll
            for (int j = 0; j < TJ; j++) acc[i][j][r] = __builtin_ldexpf(acc[i][j][r], 254 - ua - upB[j]);
        }
    float* po = part + (size_t)split * N * K;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < TJ; j++) {
            const int idx = pk * TJ + j;
            const int k = k0 + 128 * (idx >> 2) + 4 * l31 + (idx & 3);
            if (k >= K) continue;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int n = n0 + 4 * h_row_of(r, h) + pn * 2 + i;
                if (n < N) po[(size_t)n * K + k] = acc[i][j][r];
            }
        }
    if (want_b) {
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const float sb = xsum32(bsum[t], bsum[t]);
            const float sb2 = xsum16(sb, sb);
            if (pg == 0 && col_ok) part_b[(size_t)split * N + col + t] = sb2;
        }
    }
}

// same slab rule / workspace layout as stage_gemm_tn (gemm.hip); returns 1 if not handled here
int stage_gemm_tn_stream(const float* dY, const void* gate, int gate_kind, const float* X, float* part, float* part_b,
                         long long M, int N, int K, int* S_io, long* rows_per_split_io, void* stream) {
    // buffer addressing: every operand must be smaller than 2 GiB
    // rows: from 4096; from 1024 for K <= 128 (M = 3200: 27 -> 21 us; wider K loses to the tiled fp32 kernel there)
    static const long long min_m_env = getenv("STAGE_GEMM_STREAM_MIN_M") ? atoll(getenv("STAGE_GEMM_STREAM_MIN_M")) : 0;
    const long long min_m_tn = min_m_env ? min_m_env : (K <= 128 ? 1024 : 4096);
    if (M < min_m_tn || M * (long long)N * 4 >= (1ll << 31) || M * (long long)K * 4 >= (1ll << 31)) return 1;
    if (!gate) gate_kind = 0;
    const float* G = (const float*)gate;
    int S = *S_io;
    long rows_per_split = *rows_per_split_io;
    static const bool no_wide = getenv("STAGE_GEMM_TN_NOWIDE") != nullptr;
    static const bool no_share_w = getenv("STAGE_GEMM_TN_NOSHARE") != nullptr;
    if (K > 128 && !no_wide && !no_share_w) {
        // 128 x 384 tiles, one 8-wave workgroup per CU: as many slabs as give whole rounds of 256 workgroups (never more
        // than the workspace was sized for)
        const int wps = ((N + 127) / 128) * ((K + 383) / 384);
        int sw = 256 / wps;
        if (sw < 1) sw = 1;
        if (2 * sw <= S) sw *= 2;
        if (sw > S) sw = S;
        long rps = (M + sw - 1) / sw;
        rps = (rps + 15) / 16 * 16;
        sw = (int)((M + rps - 1) / rps);
        S = sw;
        rows_per_split = rps;
        *S_io = S;
        *rows_per_split_io = rps;
        dim3 gridw((N + 127) / 128, (K + 383) / 384, S);
        static const bool no_quad = getenv("STAGE_GEMM_TN_NOQUAD") != nullptr;
        if (STAGE_GEMM_TN_F16 && !no_quad && gate_kind != 1 && N % 4 == 0 && K % 4 == 0 && (((uintptr_t)dY | (uintptr_t)X) & 15) == 0 &&
            (gate_kind != 2 || (((uintptr_t)gate & 15) == 0 && rows_per_split % 16 == 0))) {
            const int ldsq = 2 * 2 * 16 * 2 * 64 * (int)sizeof(uint4) + 2 * 16 * 32 * (int)sizeof(int);
#define LAUNCH_TNQ(GT)                                                                                                 \
    do {                                                                                                               \
        static bool attr_done = false;                                                                                 \
        if (!attr_done) {                                                                                              \
            (void)hipFuncSetAttribute((const void*)gemm_tn_quad_kernel<GT, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsq); \
            attr_done = true;                                                                                          \
        }                                                                                                              \
        hipLaunchKernelGGL((gemm_tn_quad_kernel<GT, 3>), gridw, dim3(64 * TW_WAVES), ldsq, (hipStream_t)stream, dY,    \
                           (const unsigned*)gate, X, part, part_b, (long)M, N, K, rows_per_split);                     \
    } while (0)
            if (gate_kind == 2) LAUNCH_TNQ(2);
            else LAUNCH_TNQ(0);
#undef LAUNCH_TNQ
            STAGE_LAUNCH_CHECK();
            return 0;
        }
        const int lds = 2 * (TW_NA + TW_NB) * 3 * 64 * (int)sizeof(uint4);
#define LAUNCH_TNW(GT)                                                                                                 \
    do {                                                                                                               \
        static bool attr_done = false;                                                                                 \
        if (!attr_done) {                                                                                              \
            (void)hipFuncSetAttribute((const void*)gemm_tn_wide_kernel<GT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
            attr_done = true;                                                                                          \
        }                                                                                                              \
        hipLaunchKernelGGL(gemm_tn_wide_kernel<GT>, gridw, dim3(64 * TW_WAVES), lds, (hipStream_t)stream, dY, G, X, part, \
                           part_b, (long)M, N, K, rows_per_split);                                                     \
    } while (0)
        if (gate_kind == 2) LAUNCH_TNW(2);
        else if (gate_kind == 1) LAUNCH_TNW(1);
        else LAUNCH_TNW(0);
#undef LAUNCH_TNW
        STAGE_LAUNCH_CHECK();
        return 0;
    }
    dim3 grid((N + 127) / 128, (K + 127) / 128, S);
    {   // K <= 128: the quad kernel with ONE X quad (row-wise 16-byte loads; same conditions as above)
        static const bool no_quad1 = getenv("STAGE_GEMM_TN_NOQUAD") != nullptr;
        if (STAGE_GEMM_TN_F16 && !no_quad1 && gate_kind != 1 && N % 4 == 0 && K % 4 == 0 && (((uintptr_t)dY | (uintptr_t)X) & 15) == 0 &&
            (gate_kind != 2 || (((uintptr_t)gate & 15) == 0 && rows_per_split % 16 == 0))) {
            const int ldsq1 = 2 * 2 * 8 * 2 * 64 * (int)sizeof(uint4) + 2 * 8 * 32 * (int)sizeof(int);
#define LAUNCH_TNQ1(GT)                                                                                                \
    do {                                                                                                               \
        static bool attr_done = false;                                                                                 \
        if (!attr_done) {                                                                                              \
            (void)hipFuncSetAttribute((const void*)gemm_tn_quad_kernel<GT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsq1); \
            attr_done = true;                                                                                          \
        }                                                                                                              \
        hipLaunchKernelGGL((gemm_tn_quad_kernel<GT, 1>), grid, dim3(256), ldsq1, (hipStream_t)stream, dY,              \
                           (const unsigned*)gate, X, part, part_b, (long)M, N, K, rows_per_split);                     \
    } while (0)
            if (gate_kind == 2) LAUNCH_TNQ1(2);
            else LAUNCH_TNQ1(0);
#undef LAUNCH_TNQ1
            STAGE_LAUNCH_CHECK();
            return 0;
        }
    }
    static const bool no_share = getenv("STAGE_GEMM_TN_NOSHARE") != nullptr;
#define LAUNCH_TNS(GT)                                                                                                 \
    do {                                                                                                               \
        if (no_share)                                                                                                  \
            hipLaunchKernelGGL(gemm_tn_stream_kernel<GT>, grid, dim3(256), 0, (hipStream_t)stream, dY, G, X, part,      \
                               part_b, (long)M, N, K, rows_per_split);                                                 \
        else                                                                                                           \
            hipLaunchKernelGGL(gemm_tn_share_kernel<GT>, grid, dim3(256), 0, (hipStream_t)stream, dY, G, X, part,       \
                               part_b, (long)M, N, K, rows_per_split);                                                 \
    } while (0)
    if (gate_kind == 2) LAUNCH_TNS(2);
    else if (gate_kind == 1) LAUNCH_TNS(1);
    else LAUNCH_TNS(0);
#undef LAUNCH_TNS
    STAGE_LAUNCH_CHECK();
    return 0;
}
