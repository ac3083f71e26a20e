// bf16 storage mode (BASELINE.json configs[4]): the tall-skinny Linear / 1x1-conv forward and dX GEMMs as HBM-streaming
// kernels on 16-bit activations.
//
//   Y[M, N] = epi( (X .* [gate > 0]) [M, K] . W[N, K]^T + bias )        X, gate, Y: bf16   W, bias: fp32 master copies
//
// Same idea as gemm_stream.hip (the fp32 path), minus everything the 16-bit operands make unnecessary: the activations ARE
// bf16, so there is no split and ONE v_mfma_f32_32x32x16_bf16 per term; the weight tile (128 output columns x K <= 384) is
// rounded to bf16 once per workgroup and stays in LDS; every wave walks 32-row tiles of X, loading its rows straight from
// HBM in MFMA operand layout (lane (row, half) reads 128 contiguous bytes of a 128-element chunk: the contraction index is
// permuted the same way on both operands, which is free), the next chunk / next tile requested before the current one is
// multiplied.  The product is accumulated TRANSPOSED (weight fragment = A operand): a lane then owns four consecutive output
// columns of one row, packs them to 8 bytes, and the tile leaves through a per-wave LDS staging area as full 128-byte lines
// (a direct store would write 8-byte pieces of 32 different lines per instruction -- what the memory system handles worst).
// Shapes outside (K % 4 == 0, K <= 384, N % 128 == 0, M >= 4096, 8-byte aligned operands) take the 64-column variant below or the tiled kernel of
// gemm_bf16x3.hip.
#include <stdlib.h>
#include "common.h"
#include "../../include/stage_hip.h"

#define GB_BN 128                 // output columns per workgroup
#define GB_WAVES 8
#define GB_KC 64                  // k per chunk (4 MFMA steps): lane (row, half) reads 64 contiguous bytes
#define GB_STG (64 + 8)           // bf16 per row of the store staging area (64 columns + pad: 144-byte stride)

typedef __bf16 gb_bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned gb_gate1(unsigned v, unsigned gg) {
    // keep the bf16 halves of v whose gate half is > 0 (ReLU backward with the saved bf16 output as gate)
    const unsigned lo = ((int)(gg << 16) > 0) ? 0x0000FFFFu : 0u;
    const unsigned hi = ((int)(gg & 0xFFFF0000u) > 0) ? 0xFFFF0000u : 0u;
    return v & (lo | hi);
}
__device__ __forceinline__ uint4 gb_gate(uint4 v, uint4 g) {
    return make_uint4(gb_gate1(v.x, g.x), gb_gate1(v.y, g.y), gb_gate1(v.z, g.z), gb_gate1(v.w, g.w));
}

struct GbBuf { uint4 x[4], g[4]; };
// 16 bytes from an address that is only 8-byte aligned (rows of K % 8 == 4 elements): the hardware takes multi-dword global
// loads at dword alignment; the type only tells the compiler not to assume more
struct __attribute__((aligned(8))) GbU4 { unsigned x, y, z, w; };

template <bool HAS_GATE, int NKC>
__global__ __launch_bounds__(64 * GB_WAVES) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_nt_bf16_stream_kernel(
    const stage_bf16* __restrict__ X, const stage_bf16* __restrict__ G, const float* __restrict__ W,
    const float* __restrict__ bias, stage_bf16* __restrict__ Y, long M, int N, int K, int Kp, int relu, int gx, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    unsigned short* Wl = lds;                                            // [GB_BN][Kp] bf16, natural k order, zero padded
    unsigned short* stg = Wl + GB_BN * Kp;                               // [GB_WAVES][32][GB_STG]
    float* bias_s = reinterpret_cast<float*>(stg + GB_WAVES * 32 * GB_STG);   // [GB_BN]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    // 1-D grid, XCD-aware (as the 64-column kernel below): the n_tiles column tiles of a row group get consecutive slots of ONE XCD
    // (workgroup ids go round the 8 XCDs), so their re-reads of X (and the gate) meet in that XCD's L2
    const int r8 = blockIdx.x & 7, tq = blockIdx.x >> 3;
    const int by = tq % n_tiles, bx = (tq / n_tiles) * 8 + r8;
    const int n0 = by * GB_BN;
    // ---- weight tile -> LDS (rounded to bf16 once) ----
    const int K4p = Kp >> 2;
    for (int e = tid; e < GB_BN * K4p; e += 64 * GB_WAVES) {
        const int n = e / K4p, q = e - n * K4p;
        float4 v = f4zero();
        if (4 * q < K) v = ld4(W + (long)(n0 + n) * K + 4 * q);          // K % 4 == 0, n0 + n < N by the launcher
        *reinterpret_cast<uint2*>(&Wl[n * Kp + 4 * q]) = make_uint2(stage_pk_bf16(v.x, v.y), stage_pk_bf16(v.z, v.w));
    }
    if (tid < GB_BN) bias_s[tid] = bias ? bias[n0 + tid] : 0.f;
    __syncthreads();

    const long MT = (M + 31) >> 5;
    const long nw = (long)gx * GB_WAVES;
    unsigned short* my_stg = stg + wave * 32 * GB_STG;
    // chunk c (64 k = 128 bytes per row) of row tile `tile`, COALESCED: load j covers rows 8 j .. 8 j + 7, eight lanes per row, 16 bytes
    // each (8 cache lines per instruction).  Round 4: the former scheme -- every lane reading 64 contiguous bytes of ITS row, i.e. the
    // MFMA operand layout straight from memory -- touched 32 lines per instruction and ran at ~2.5 us per chunk and CU whatever the
    // prefetch depth, the column-tile placement or the HBM traffic (PMC: every byte fetched once, 72 % of the wave cycles waiting):
    // the vector memory pipeline's line rate was the limit.  The operand layout is now made by a pass through the wave's staging rows.
    auto fetch = [&](GbBuf& b, long tile, int c) {
        const int k = c * GB_KC + 8 * (lane & 7);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const long row = tile * 32 + 8 * j + (lane >> 3);
            const long ro = (row < M ? row : M - 1) * K;
            if (k + 8 <= K) {                            // (rows of K = 8 q + 4 elements are 8-byte aligned: GbU4)
                const GbU4 v = *reinterpret_cast<const GbU4*>(X + ro + k);
                b.x[j] = make_uint4(v.x, v.y, v.z, v.w);
                if (HAS_GATE) { const GbU4 g4 = *reinterpret_cast<const GbU4*>(G + ro + k); b.g[j] = make_uint4(g4.x, g4.y, g4.z, g4.w); }
            } else if (k < K) {                          // K % 8 == 4: half a group at the end of the row
                const uint2 v = *reinterpret_cast<const uint2*>(X + ro + k);
                b.x[j] = make_uint4(v.x, v.y, 0u, 0u);
                if (HAS_GATE) { const uint2 g2 = *reinterpret_cast<const uint2*>(G + ro + k); b.g[j] = make_uint4(g2.x, g2.y, 0u, 0u); }
            } else {
                b.x[j] = make_uint4(0u, 0u, 0u, 0u);
                if (HAS_GATE) b.g[j] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
    };
    long t = (long)bx * GB_WAVES + wave;
    // A WHOLE row tile ahead: chunk c of the next tile is requested the moment chunk c of this one has been multiplied (its buffer is
    // free), so DEPTH chunks per wave are always in flight.  (Round 4: with one chunk ahead a wave waited out one memory latency per
    // 16 MFMAs -- 72 % of the wave cycles waiting, 2.1 TB/s at 960000 x 256 -> 768 with every byte fetched from HBM once.)
    constexpr int DEPTH = NKC == 6 ? 3 : NKC;            // chunks in flight (divides NKC; six gated chunks would not fit the registers)
    GbBuf buf[DEPTH];
    if (t < MT) {
#pragma unroll
        for (int c = 0; c < DEPTH; c++) fetch(buf[c], t, c);
    }
    for (; t < MT; t += nw) {
        f32x16 acc[4];
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[nt][r] = 0.f;
#pragma unroll
        for (int c = 0; c < NKC; c++) {
            GbBuf& cur = buf[c % DEPTH];
            // gate, zero the K tail, and turn the coalesced pieces into MFMA operands through the wave's own staging rows (144-byte row
            // stride: both the 16-byte writes and the reads are conflict free; the LDS operations of one wave complete in order)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                uint4 xv = cur.x[j];                     // (zero past K: fetch)
                if (HAS_GATE) xv = gb_gate(xv, cur.g[j]);
                *reinterpret_cast<uint4*>(&my_stg[(8 * j + (lane >> 3)) * GB_STG + 8 * (lane & 7)]) = xv;
            }
            uint4 xf[4];
#pragma unroll
            for (int s = 0; s < 4; s++) xf[s] = *reinterpret_cast<const uint4*>(&my_stg[l31 * GB_STG + 32 * h + 8 * s]);
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const int k = c * GB_KC + 32 * h + 8 * s;
                const gb_bf16x8 b = __builtin_bit_cast(gb_bf16x8, xf[s]);
#pragma unroll
                for (int nt = 0; nt < 4; nt++) {
                    const gb_bf16x8 a = __builtin_bit_cast(gb_bf16x8, *reinterpret_cast<const uint4*>(&Wl[(nt * 32 + l31) * Kp + k]));
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[nt], 0, 0, 0);   // D[n][m]
                }
            }
            if (c + DEPTH < NKC) fetch(cur, t, c + DEPTH);
            else fetch(cur, t + nw, c + DEPTH - NKC);                   // ... of the next tile (clamped rows)
        }
        // ---- epilogue: D[n][m], lane = row m (l31), registers = columns 8 (r >> 2) + 4 h + (r & 3) of n-tile nt ----
#pragma unroll
        for (int p = 0; p < 2; p++) {
#pragma unroll
            for (int nt2 = 0; nt2 < 2; nt2++) {
                const int nt = 2 * p + nt2;
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const float4 bs = ld4(&bias_s[32 * nt + 8 * g + 4 * h]);
                    float4 v = make_float4(acc[nt][4 * g + 0] + bs.x, acc[nt][4 * g + 1] + bs.y, acc[nt][4 * g + 2] + bs.z,
                                           acc[nt][4 * g + 3] + bs.w);
                    if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                    *reinterpret_cast<uint2*>(&my_stg[l31 * GB_STG + 32 * nt2 + 8 * g + 4 * h]) =
                        make_uint2(stage_pk_bf16(v.x, v.y), stage_pk_bf16(v.z, v.w));
                }
            }
            // the wave's own staging rows: the LDS operations of one wave complete in order, no barrier
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int r = 8 * i + (lane >> 3), c8 = (lane & 7) * 8;
                const uint4 v = *reinterpret_cast<const uint4*>(&my_stg[r * GB_STG + c8]);
                const long m = t * 32 + r;
                if (m < M) *reinterpret_cast<uint4*>(Y + m * N + n0 + 64 * p + c8) = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same kernel for the shapes the 128-column one rejects: wide K (the 768 / 300-wide input layers: a 64-column weight tile
// of K <= 960 still fits the LDS), N not a multiple of 128 (300), rows that are only 8-byte aligned (K or N % 8 == 4).
// 64 output columns per workgroup, run-time chunk count, 8-byte loads / stores where 16 are not aligned, and the XCD-aware
// 1-D grid of gemm_stream.hip: the ceil(N / 64) column tiles of a row group sit on one XCD, so their re-reads of X hit its L2.
// ---------------------------------------------------------------------------------------------------------------------
template <bool HAS_GATE>
__global__ __launch_bounds__(64 * GB_WAVES) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_nt_bf16_stream64_kernel(
    const stage_bf16* __restrict__ X, const stage_bf16* __restrict__ G, const float* __restrict__ W,
    const float* __restrict__ bias, stage_bf16* __restrict__ Y, long M, int N, int K, int Kp, int relu, int gx, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    unsigned short* Wl = lds;                                            // [64][Kp]
    unsigned short* stg = Wl + 64 * Kp;                                  // [GB_WAVES][32][GB_STG]
    float* bias_s = reinterpret_cast<float*>(stg + GB_WAVES * 32 * GB_STG);   // [64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int r8 = blockIdx.x & 7, tq = blockIdx.x >> 3;
    const int by = tq % n_tiles, bx = (tq / n_tiles) * 8 + r8;           // id = 8 (n_tiles q + by) + r,  bx = 8 q + r
    const int n0 = by * 64;
    const int nkc = (K + GB_KC - 1) / GB_KC;
    const int K4p = Kp >> 2;
    for (int e = tid; e < 64 * K4p; e += 64 * GB_WAVES) {
        const int n = e / K4p, q = e - n * K4p;
        float4 v = f4zero();
        if (4 * q < K && n0 + n < N) v = ld4(W + (long)(n0 + n) * K + 4 * q);
        *reinterpret_cast<uint2*>(&Wl[n * Kp + 4 * q]) = make_uint2(stage_pk_bf16(v.x, v.y), stage_pk_bf16(v.z, v.w));
    }
    if (tid < 64) bias_s[tid] = (bias && n0 + tid < N) ? bias[n0 + tid] : 0.f;
    __syncthreads();

    const long MT = (M + 31) >> 5;
    const long nw = (long)gx * GB_WAVES;
    unsigned short* my_stg = stg + wave * 32 * GB_STG;
    auto ld8 = [&](const stage_bf16* p, long rowoff, int k) -> uint4 {       // 8 elements at k (k % 8 == 0), zero past K
        if (k + 8 <= K) {
            const GbU4 v = *reinterpret_cast<const GbU4*>(p + rowoff + k);
            return make_uint4(v.x, v.y, v.z, v.w);
        }
        if (k < K) {                                                         // K % 8 == 4: the last group is half a group
            const uint2 a = *reinterpret_cast<const uint2*>(p + rowoff + k);
            return make_uint4(a.x, a.y, 0u, 0u);
        }
        return make_uint4(0u, 0u, 0u, 0u);
    };
    // coalesced, as in the 128-column kernel: load j = rows 8 j .. 8 j + 7 of the tile, eight lanes per row
    auto fetch = [&](GbBuf& b, long tile, int c) {
        const int k = c * GB_KC + 8 * (lane & 7);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const long row = tile * 32 + 8 * j + (lane >> 3);
            const long ro = (row < M ? row : M - 1) * K;
            b.x[j] = ld8(X, ro, k);
            if (HAS_GATE) b.g[j] = ld8(G, ro, k);
        }
    };
    f32x16 acc[2];
    auto mul = [&](GbBuf& cur, int c) {
#pragma unroll
        for (int j = 0; j < 4; j++) {                    // gate, then through the wave's staging rows into the MFMA operand layout
            uint4 xv = cur.x[j];
            if (HAS_GATE) xv = gb_gate(xv, cur.g[j]);
            *reinterpret_cast<uint4*>(&my_stg[(8 * j + (lane >> 3)) * GB_STG + 8 * (lane & 7)]) = xv;
        }
        uint4 xf[4];
#pragma unroll
        for (int s = 0; s < 4; s++) xf[s] = *reinterpret_cast<const uint4*>(&my_stg[l31 * GB_STG + 32 * h + 8 * s]);
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int k = c * GB_KC + 32 * h + 8 * s;
            const gb_bf16x8 b = __builtin_bit_cast(gb_bf16x8, xf[s]);
#pragma unroll
            for (int nt = 0; nt < 2; nt++) {
                const gb_bf16x8 a = __builtin_bit_cast(gb_bf16x8, *reinterpret_cast<const uint4*>(&Wl[(nt * 32 + l31) * Kp + k]));
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[nt], 0, 0, 0);   // D[n][m]
            }
        }
    };
    long t = (long)bx * GB_WAVES + wave;
    GbBuf b0, b1, bn;
    if (t < MT) fetch(b0, t, 0);
    for (; t < MT; t += nw) {
        const long row = t;
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[nt][r] = 0.f;
#pragma unroll 1
        for (int c = 0; c < nkc; c += 2) {
            const bool has1 = c + 1 < nkc;
            if (has1) fetch(b1, row, c + 1);
            else fetch(bn, t + nw, 0);
            mul(b0, c);
            if (has1) {
                if (c + 2 < nkc) fetch(b0, row, c + 2);
                else fetch(bn, t + nw, 0);
                mul(b1, c + 1);
            }
        }
        // ---- epilogue (one 64-column pass): D[n][m], lane = row m, registers = columns 8 (r >> 2) + 4 h + (r & 3) of n-tile nt ----
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const float4 bs = ld4(&bias_s[32 * nt + 8 * g + 4 * h]);
                float4 v = make_float4(acc[nt][4 * g + 0] + bs.x, acc[nt][4 * g + 1] + bs.y, acc[nt][4 * g + 2] + bs.z,
                                       acc[nt][4 * g + 3] + bs.w);
                if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                *reinterpret_cast<uint2*>(&my_stg[l31 * GB_STG + 32 * nt + 8 * g + 4 * h]) =
                    make_uint2(stage_pk_bf16(v.x, v.y), stage_pk_bf16(v.z, v.w));
            }
#pragma unroll
        for (int i = 0; i < 8; i++) {                 // 8-byte pieces: rows are 8-byte aligned for any N % 4 == 0
            const int r = 4 * i + (lane >> 4), c4 = (lane & 15) * 4;
            const uint2 v = *reinterpret_cast<const uint2*>(&my_stg[r * GB_STG + c4]);
            const long m = t * 32 + r;
            if (m < M && n0 + c4 < N) *reinterpret_cast<uint2*>(Y + m * N + n0 + c4) = v;
        }
        b0 = bn;
    }
}

// returns 1 if the shape / alignment is not handled here (caller falls back to the tiled kernel), 0 on launch
static int gb_launch64(const void* X, const void* gate, const float* W, const float* bias, void* Y, long long M, int N, int K,
                       int relu, void* stream) {
    if (M < 4096 || K % 4 != 0 || K < 64 || N % 4 != 0 || N < 4) return 1;
    if (((uintptr_t)X & 7) || ((uintptr_t)Y & 7) || ((uintptr_t)W & 15) || (gate && ((uintptr_t)gate & 7))) return 1;
    const int nkc = (K + GB_KC - 1) / GB_KC;
    int Kp = nkc * GB_KC;
    Kp += ((Kp / 8) % 2 == 0) ? 8 : 16;
    const size_t lds = (size_t)64 * Kp * 2 + (size_t)GB_WAVES * 32 * GB_STG * 2 + 64 * sizeof(float);
    if (lds > 160 * 1024) return 1;
    // a gated operand in half-aligned rows with many column tiles: measured slower than the tiled kernel (2.46 M rows,
    // 300 -> 768: 12.7 vs 9.1 ms; 240 k rows: 1.04 vs 0.88 ms)
    static const bool tiled_gated = getenv("STAGE_GEMM_BF16_TILED_GATED") != nullptr;    // (developer switch: the rule of round 3)
    if (tiled_gated && gate && K % 8 != 0 && N > 256) return 1;
    const long MT = (M + 31) / 32;
    const int n_tiles = (N + 63) / 64;
    long gx = (256 / n_tiles) / 8 * 8;
    if (gx < 8) gx = 8;
    const long need = ((MT + GB_WAVES - 1) / GB_WAVES + 7) / 8 * 8;
    if (gx > need) gx = need;
    dim3 grid((unsigned)(gx * n_tiles)), block(64 * GB_WAVES);
    typedef stage_bf16 B;
#define GB64(GT)                                                                                                                \
    do {                                                                                                                        \
        (void)hipFuncSetAttribute((const void*)gemm_nt_bf16_stream64_kernel<GT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((gemm_nt_bf16_stream64_kernel<GT>), grid, block, lds, (hipStream_t)stream, (const B*)X, (const B*)gate, W, \
                           bias, (B*)Y, (long)M, N, K, Kp, relu, (int)gx, n_tiles);                                             \
    } while (0)
    if (gate) GB64(true); else GB64(false);
#undef GB64
    STAGE_LAUNCH_CHECK();
    return 0;
}

int stage_gemm_nt_bf16_stream(const void* X, const void* gate, const float* W, const float* bias, void* Y, long long M, int N,
                              int K, int relu, void* stream) {
    static const bool off = getenv("STAGE_GEMM_BF16_TILED") != nullptr;   // developer switch: tiled kernel everywhere
    if (off) return 1;
    if (M < 4096 || K % 4 != 0 || K < 64 || K > 384 || N % GB_BN != 0 ||
        ((uintptr_t)X & 7) || ((uintptr_t)Y & 15) || ((uintptr_t)W & 15) || (gate && ((uintptr_t)gate & 7)))
        return gb_launch64(X, gate, W, bias, Y, M, N, K, relu, stream);   // wide K / ragged N / 8-byte rows
    int nkc = (K + GB_KC - 1) / GB_KC;
    if (gate && nkc == 5) nkc = 6;                  // five gated chunks in flight spill; six run three deep (the sixth is all zeros: no loads)
    int Kp = nkc * GB_KC;                           // whole chunks, zero padded
    Kp += ((Kp / 8) % 2 == 0) ? 8 : 16;             // odd number of 16-byte slots per row: conflict-free ds_read_b128
    const size_t lds = (size_t)GB_BN * Kp * 2 + (size_t)GB_WAVES * 32 * GB_STG * 2 + GB_BN * sizeof(float);
    if (lds > 160 * 1024) return 1;
    const long MT = (M + 31) / 32;
    const int n_tiles = N / GB_BN;
    long gx = (256 / n_tiles) / 8 * 8;              // row groups: a multiple of 8 (one per XCD slot), one workgroup per CU
    if (gx < 8) gx = 8;
    const long need = ((MT + GB_WAVES - 1) / GB_WAVES + 7) / 8 * 8;
    if (gx > need) gx = need;
    dim3 grid((unsigned)(gx * n_tiles)), block(64 * GB_WAVES);
    typedef stage_bf16 B;
#define GB_GO(GT, NK)                                                                                                          \
    do {                                                                                                                       \
        (void)hipFuncSetAttribute((const void*)gemm_nt_bf16_stream_kernel<GT, NK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((gemm_nt_bf16_stream_kernel<GT, NK>), grid, block, lds, (hipStream_t)stream, (const B*)X, (const B*)gate, W, \
                           bias, (B*)Y, (long)M, N, K, Kp, relu, (int)gx, n_tiles);                                            \
    } while (0)
#define GB_NK(NK) do { if (gate) GB_GO(true, NK); else GB_GO(false, NK); } while (0)
    switch (nkc) {
        case 1: GB_NK(1); break;
        case 2: GB_NK(2); break;
        case 3: GB_NK(3); break;
        case 4: GB_NK(4); break;
        case 5: GB_NK(5); break;
        default: GB_NK(6); break;
    }
#undef GB_NK
#undef GB_GO
    STAGE_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient on 16-bit activations:  dW[N, K] = sum_m (dY .* [gate > 0])[m, n] X[m, k] ,  db[n] = sum_m (...)[m, n]
// The contraction runs over the ROWS, so an MFMA operand is 8 consecutive rows of one column.  A lane loads one dword (two
// adjacent bf16 columns) from each of its 8 rows -- 32 lanes x 4 bytes = one full 128-byte line per row and instruction, as in
// the fp32 kernel -- and repacks the 8 dwords into the operands of its even and its odd column with eight v_perm_b32.  A wave
// owns a 64 x 64 patch of dW (even / odd columns of 64 dY columns x even / odd of 64 X columns = 2 x 2 MFMA tiles) and
// walks its row slab 16 rows at a time; the waves of a workgroup cover the patches of a (<= 12 patches) group and, when there
// are fewer, interleave the 16-row steps of the slab.  Every (slab, interleave) pair writes one fp32 partial; the ordered slab
// sum of gemm_bf16x3.hip finishes (deterministic, no atomics).
// ---------------------------------------------------------------------------------------------------------------------
struct TbStep { unsigned y[8], g[8], x[8]; };

template <bool HAS_GATE>
__global__ __launch_bounds__(768) void gemm_tn_bf16_stream_kernel(const stage_bf16* __restrict__ dY, const stage_bf16* __restrict__ G,
                                                                    const stage_bf16* __restrict__ X, float* __restrict__ part,
                                                                    float* __restrict__ part_b, long M, int N, int K, long rows_per_slab,
                                                                    int P, int RS, int KPn) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
    const int pl = wave % P + blockIdx.y * P;               // patch index (k-major: pk = pl / NPn, pn = pl % NPn)
    const int rs = wave / P;
    const int NPn = (N + 63) >> 6;
    const bool idle = pl >= NPn * KPn;    // (only with several patch groups, where RS == 1 and no barrier follows)
    if (idle) return;
    // k-major: a workgroup group covers ALL dY column patches for a few X column patches, so with several groups X (the wider
    // operand) is still read once and only dY is re-read per group
    const int pk = pl / NPn, pn = pl - pk * NPn;
    const long mbeg = (long)blockIdx.x * rows_per_slab, mend = min(M, mbeg + rows_per_slab);
    // ragged last patch (N or K not a multiple of 64; both are even): columns past the end read column 0 and are zeroed
    const bool yok = 64 * pn + 2 * l31 < N, xok = 64 * pk + 2 * l31 < K;
    const unsigned* yb = reinterpret_cast<const unsigned*>(dY) + (yok ? pn * 32 + l31 : 0);      // dword = columns 64 pn + 2 l31, +1
    const unsigned* gb = reinterpret_cast<const unsigned*>(HAS_GATE ? G : dY) + (yok ? pn * 32 + l31 : 0);
    const unsigned* xb = reinterpret_cast<const unsigned*>(X) + (xok ? pk * 32 + l31 : 0);
    const long ldy = N >> 1, ldx = K >> 1;                   // row strides in dwords
    f32x16 acc[2][2];
#pragma unroll
    for (int e = 0; e < 2; e++)
#pragma unroll
        for (int f = 0; f < 2; f++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[e][f][r] = 0.f;
    float bs0 = 0.f, bs1 = 0.f;                              // column sums of the (gated) dY dwords of this lane (pk == 0 only)
    auto fetch = [&](TbStep& s, long m0) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const long r = min(m0 + 8 * h + j, M - 1);       // rows past the slab are zeroed at use
            s.y[j] = yb[r * ldy];
            if (HAS_GATE) s.g[j] = gb[r * ldy];
            s.x[j] = xb[r * ldx];
        }
    };
    auto step = [&](TbStep& s, long m0) {
        unsigned y[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            unsigned v = s.y[j];
            if (HAS_GATE) v = gb_gate1(v, s.g[j]);
            if (m0 + 8 * h + j >= mend || !yok) v = 0u;
            y[j] = v;
        }
        if (part_b && pk == 0) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                bs0 += __uint_as_float(y[j] << 16);
                bs1 += __uint_as_float(y[j] & 0xFFFF0000u);
            }
        }
        // operands: dword q of column c holds rows (2q, 2q+1): even column = low halves, odd column = high halves
        uint4 ae, ao, be, bo;
        ae.x = __builtin_amdgcn_perm(y[1], y[0], 0x05040100u); ao.x = __builtin_amdgcn_perm(y[1], y[0], 0x07060302u);
        ae.y = __builtin_amdgcn_perm(y[3], y[2], 0x05040100u); ao.y = __builtin_amdgcn_perm(y[3], y[2], 0x07060302u);
        ae.z = __builtin_amdgcn_perm(y[5], y[4], 0x05040100u); ao.z = __builtin_amdgcn_perm(y[5], y[4], 0x07060302u);
        ae.w = __builtin_amdgcn_perm(y[7], y[6], 0x05040100u); ao.w = __builtin_amdgcn_perm(y[7], y[6], 0x07060302u);
        if (!xok) {
#pragma unroll
            for (int j = 0; j < 8; j++) s.x[j] = 0u;
        }
        be.x = __builtin_amdgcn_perm(s.x[1], s.x[0], 0x05040100u); bo.x = __builtin_amdgcn_perm(s.x[1], s.x[0], 0x07060302u);
        be.y = __builtin_amdgcn_perm(s.x[3], s.x[2], 0x05040100u); bo.y = __builtin_amdgcn_perm(s.x[3], s.x[2], 0x07060302u);
        be.z = __builtin_amdgcn_perm(s.x[5], s.x[4], 0x05040100u); bo.z = __builtin_amdgcn_perm(s.x[5], s.x[4], 0x07060302u);
        be.w = __builtin_amdgcn_perm(s.x[7], s.x[6], 0x05040100u); bo.w = __builtin_amdgcn_perm(s.x[7], s.x[6], 0x07060302u);
        const gb_bf16x8 A0 = __builtin_bit_cast(gb_bf16x8, ae), A1 = __builtin_bit_cast(gb_bf16x8, ao);
        const gb_bf16x8 B0 = __builtin_bit_cast(gb_bf16x8, be), B1 = __builtin_bit_cast(gb_bf16x8, bo);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B1, acc[1][1], 0, 0, 0);
    };
    // X rows past the slab: the dY operand is zero there, but a NaN in a clamped X row would still poison 0 * NaN -- the
    // clamp reads row M - 1 at worst (a real row), and rows of OTHER slabs are finite wherever this tensor is finite at all
    const long stride = 16L * RS;
    long m0 = mbeg + 16L * rs;
    TbStep s0, s1;
    if (m0 < mend) fetch(s0, m0);
    for (; m0 < mend; m0 += 2 * stride) {
        const bool more1 = m0 + stride < mend;
        if (more1) fetch(s1, m0 + stride);
        step(s0, m0);
        if (m0 + 2 * stride < mend) fetch(s0, m0 + 2 * stride);
        if (more1) step(s1, m0 + stride);
    }
    // ---- the RS interleaves of a patch are summed through LDS (fixed order: deterministic), interleave 0 writes the slab ----
    extern __shared__ __attribute__((aligned(16))) float red[];         // [(RS - 1) * P waves][66][64]: 64 accumulators + 2 bias sums per lane
    if (RS > 1) {
        if (rs > 0) {
            float* mine = red + (size_t)((rs - 1) * P + wave % P) * 66 * 64 + lane;
#pragma unroll
            for (int e = 0; e < 2; e++)
#pragma unroll
                for (int f = 0; f < 2; f++)
#pragma unroll
                    for (int r = 0; r < 16; r++) mine[(size_t)((e * 2 + f) * 16 + r) * 64] = acc[e][f][r];
            mine[(size_t)64 * 64] = bs0;
            mine[(size_t)65 * 64] = bs1;
        }
        __syncthreads();
        if (rs > 0) return;
        for (int o = 1; o < RS; o++) {
            const float* theirs = red + (size_t)((o - 1) * P + wave % P) * 66 * 64 + lane;
#pragma unroll
            for (int e = 0; e < 2; e++)
#pragma unroll
                for (int f = 0; f < 2; f++)
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[e][f][r] += theirs[(size_t)((e * 2 + f) * 16 + r) * 64];
            bs0 += theirs[(size_t)64 * 64];
            bs1 += theirs[(size_t)65 * 64];
        }
    }
    // ---- partial of this slab: D[i][j] -> dW[n = 64 pn + 2 i + e][k = 64 pk + 2 j + f] ----
    float* po = part + (size_t)blockIdx.x * (size_t)N * K;
#pragma unroll
    for (int e = 0; e < 2; e++)
#pragma unroll
        for (int f = 0; f < 2; f++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
                const int n = 64 * pn + 2 * i + e, k = 64 * pk + 2 * l31 + f;
                if (n < N && k < K) po[(size_t)n * K + k] = acc[e][f][r];
            }
    if (part_b && pk == 0) {
        bs0 += __shfl_xor(bs0, 32);
        bs1 += __shfl_xor(bs1, 32);
        if (h == 0 && yok) {
            float* pb = part_b + (size_t)blockIdx.x * N + 64 * pn + 2 * l31;
            pb[0] = bs0;
            pb[1] = bs1;
        }
    }
}

// slabs x interleaves of the streaming TN kernel for a shape (0: not handled)
static int tb_plan(long long M, int N, int K, int* P, int* RS, int* KPn, int* GY, long* rps) {
    if (M < 4096 || N % 2 != 0 || K % 2 != 0 || N < 2 || K < 2) return 0;
    *KPn = (K + 63) / 64;
    const int total = ((N + 63) / 64) * *KPn;
    // more than 12 patches need several workgroup groups, each re-reading one operand: measured slower than the tiled kernel
    // in both group orders (2.46 M rows, 768 x 300 and 768 x 256: stress step 111.4 vs 108.9 ms) -- those shapes stay tiled
    // unless STAGE_GEMM_BF16_TN_GROUPS is set (developer switch)
    if (total > 12 && !getenv("STAGE_GEMM_BF16_TN_GROUPS")) return 0;
    *GY = (total + 11) / 12;                    // <= 12 waves per workgroup (768 threads: 168 registers per lane)
    *P = (total + *GY - 1) / *GY;
    *RS = *GY == 1 ? (12 / *P) : 1;
    if (*RS > 1 + 8 / *P) *RS = 1 + 8 / *P;    // LDS for the in-workgroup sum: (RS - 1) * P * 16.5 KB <= 132 KB
    if (*RS < 1) *RS = 1;
    // one resident workgroup per CU: 256 / GY slabs, at least 32 steps of 16 rows per interleave
    long S = 256 / *GY;
    const long max_by_rows = (M + 16L * 32 * *RS - 1) / (16L * 32 * *RS);
    if (S > max_by_rows) S = max_by_rows;
    if (S < 1) S = 1;
    long r = (M + S - 1) / S;
    r = (r + 16L * *RS - 1) / (16L * *RS) * (16L * *RS);
    *rps = r;
    return (int)((M + r - 1) / r);
}

size_t stage_gemm_tn_bf16_stream_ws_bytes(long long M, int N, int K) {
    int P, RS, KPn, GY; long rps;
    const int S = tb_plan(M, N, K, &P, &RS, &KPn, &GY, &rps);
    return (size_t)S * ((size_t)N * K + N) * sizeof(float);
}

// returns 1 if the shape / alignment is not handled here, 0 on launch; *slabs = number of partials written
int stage_gemm_tn_bf16_stream(const void* dY, const void* gate, const void* X, float* part, float* part_b, long long M, int N,
                              int K, int* slabs, void* stream) {
    static const bool off = getenv("STAGE_GEMM_BF16_TILED") != nullptr;
    int P, RS, KPn, GY; long rps;
    const int S = off ? 0 : tb_plan(M, N, K, &P, &RS, &KPn, &GY, &rps);
    if (S == 0 || ((uintptr_t)dY & 3) || ((uintptr_t)X & 3) || (gate && ((uintptr_t)gate & 3))) return 1;
    typedef stage_bf16 B;
    dim3 grid((unsigned)S, (unsigned)GY), block(64 * P * RS);
    const size_t lds = (size_t)(RS - 1) * P * 66 * 64 * sizeof(float);
    if (lds > 64 * 1024) {
        (void)hipFuncSetAttribute((const void*)gemm_tn_bf16_stream_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)gemm_tn_bf16_stream_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    if (gate)
        hipLaunchKernelGGL(gemm_tn_bf16_stream_kernel<true>, grid, block, lds, (hipStream_t)stream, (const B*)dY, (const B*)gate,
                           (const B*)X, part, part_b, (long)M, N, K, rps, P, RS, KPn);
    else
        hipLaunchKernelGGL(gemm_tn_bf16_stream_kernel<false>, grid, block, lds, (hipStream_t)stream, (const B*)dY, (const B*)gate,
                           (const B*)X, part, part_b, (long)M, N, K, rps, P, RS, KPn);
    STAGE_LAUNCH_CHECK();
    *slabs = S;
    return 0;
}
