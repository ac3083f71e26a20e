// Backward of  z = drop(LN_3D([a, b, a*b]));  y = ReLU(z W^T + c)   (model/stage.py:381-385 c2q_down_projection, :276-279 concat_fc)
// with EVERYTHING in one pass over dy, a and b: the input gradient of the Linear (dz = (dy .* relu') W, never a tensor -- as
// cat3_fused.hip: cf_bwd_kernel), the LayerNorm backward on it, AND the Linear's own weight / bias gradient
//     dW[n][k] = sum_rows (dy .* relu')[row][n] * z[row][k]          dc[n] = sum_rows (dy .* relu')[row][n]
// so that the forward does not have to write z (1536 B per row = 74 % of its stores) and no weight-gradient GEMM reads it back.
//
// One persistent workgroup of EIGHT waves per compute unit, two per SIMD with different jobs (the 128 x 384 weight gradient is 192
// accumulator registers per lane at four waves: next to the LayerNorm epilogue's own registers it fits no single wave, not even with
// the whole 512-entry file -- so the two halves of the register file get one job each, and the vector work is split between them: a
// lone wave issues at most every other cycle of its SIMD):
//   * E waves 0..3 (tile = four passes of 8 rows as in cf_bwd_kernel, wave w owns columns 32 w .. 32 w + 31 of each third): dX product
//     (v_mfma_f32_32x32x16_f16, two-way fp16 split / three products; A = the gated dy tile as row-major fp16 planes in LDS, B = weight
//     fragments from an L2-resident image), LayerNorm backward on the accumulators (row statistics exchanged through LDS), db / da out.
//   * W waves 4..7: everything else.  (1) the memory side: every byte a tile needs (dy rows, mask words, rows of b (and a), mean, rstd)
//     comes into LDS by buffer_load ... lds (no registers, bounds-checked) two tiles ahead; (2) staging of the next tile: ReLU gate, one
//     power-of-two scale per row, two fp16 planes, row-major; the dropout hashes of the next tile (handed to the E waves through LDS,
//     kept in registers for (3)); the bias gradient; (3) the dW product of the PREVIOUS tile: z is rebuilt from a, b, the saved row
//     statistics, gamma / beta and the dropout bits directly in the B-operand layout of a contraction over the tile's rows (lane = column,
//     k-slot (k-step s, lane half h, element e) <-> row 16 s + 8 (e >> 2) + (e & 3) + 4 h); the A operand (dy^T) comes out of the
//     row-major planes through ds_read_b64_tr_b16 under the same permutation; accumulators live for the whole launch.
//   The dy planes carry one scale per ROW (fp16 range), which a contraction over rows cannot factor out: 2^(E - up_row) goes into z,
//   E = the smallest scale field the workgroup has seen so far; when a tile lowers it the W waves rescale their accumulators (a
//   wave-uniform branch); the final store divides 2^(E - 127) and z's own column scale out again.
// Two barriers per tile (all eight waves): Ba(i) planes / row scales / dropout bits of tile i ready | Bb(i) row statistics of tile i
// ready, dW of tile i - 1 done (its planes are free), the LDS-DMA of tile i + 1 landed.
#include "common.h"
#include "../../include/stage_hip.h"

#ifndef CW_ABL
#define CW_ABL 0      // developer ablation bits (timing only, results wrong): 1 no dX MFMAs, 2 no dW MFMAs, 4 no dropout hashes, 8 no
#endif                // LayerNorm epilogue (both halves), 16 no staging arithmetic, 32 no z rebuild
namespace {
constexpr int CW_D = 128, CW_K3 = 384;
constexpr int CW_KS = CW_D / 16;                      // k-steps of the dX product
constexpr int CW_WFRAG = 2 * 12 * CW_KS * 64;         // uint4 fragments of the pre-split weight image (both planes)
constexpr int CW_PITCH = 320;                         // bytes per row of a row-major dy plane: the [4 rows][16 columns] blocks of the two
                                                      // 16-lane groups a ds_read_b64_tr_b16 serves together sit on 32 different bank pairs
constexpr int CW_RP = 2 * 32 * CW_PITCH;              // row-major planes of one tile [plane][row][pitch]
constexpr int CW_TILE = 32 * CW_D * 4;                // one 32-row tile of a (rows, D) fp32 tensor
constexpr int CW_OFF_RP = 0;                          // [2 buffers] dy planes
constexpr int CW_OFF_RAW = CW_OFF_RP + 2 * CW_RP;     // [32][128] float   dy rows in flight (LDS-DMA; W wave v: rows 8 v .. 8 v + 7)
constexpr int CW_OFF_BT = CW_OFF_RAW + CW_TILE;       // [3 buffers][32][128] float  rows of b
constexpr int CW_OFF_AT = CW_OFF_BT + 3 * CW_TILE;    // [3 buffers][32][128] float  rows of a (flat) / [40][128] the group's block (broadcast)
constexpr int CW_OFF_MK = CW_OFF_AT + 3 * CW_TILE;    // [4 W waves][4 words][8 rows] u32  ReLU mask words in flight
constexpr int CW_OFF_MS = CW_OFF_MK + 4 * 32 * 4;     // [3 buffers][mean | rstd][32] float
constexpr int CW_OFF_UP = CW_OFF_MS + 3 * 64 * 4;     // [3 buffers][32] int         row scale exponent fields
constexpr int CW_OFF_KB = CW_OFF_UP + 3 * 32 * 4;     // [4 E waves][2][64] u32      dropout bits of the tile (three nibbles per pass)
constexpr int CW_OFF_TE = CW_OFF_KB + 4 * 2 * 64 * 4; // [3] int (+ pad)             tile minimum of the scale fields
constexpr int CW_OFF_ST = CW_OFF_TE + 16;             // [4 E waves][32][2] float    partial row statistics
constexpr int CW_OFF_GZ = CW_OFF_ST + 4 * 32 * 8;     // [2][384] float              z's column scale folded into gamma / beta (W waves)
constexpr int CW_LDS = CW_OFF_GZ + 2 * CW_K3 * 4;
static_assert(CW_LDS <= 160 * 1024, "LDS budget");
#define CW_WTAB_GSEG(W) ((((W) + 1) + 3) & ~3)
#define CW_WTAB_SEG(W, G) (CW_WTAB_GSEG(W) + ((2 * (G) + 3) & ~3))

typedef unsigned cw_u4 __attribute__((ext_vector_type(4)));
typedef __fp16 cw_h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) void* cw_lds_ptr;

// Weight image (as cat3_fused.hip: cf_prep_w_kernel): img[plane][column tile ct < 12][k-step ks < 8][lane] = the 8 fp16 of B-operand
// lane (col = 32 ct + (lane & 31), k = 16 ks + 8 (lane >> 5) + e) of W[k][col]  (W = the Linear's (D, 3D) weight: dz = dy_gated . W)
__global__ __launch_bounds__(1024) void cw_prep_w_kernel(const float* __restrict__ W, uint4* __restrict__ img, int* __restrict__ w_up_out) {
    __shared__ float red[16];
    const int tid = threadIdx.x;
    float m = 0.f;
    for (int e = tid; e < CW_D * CW_K3; e += 1024) m = fmaxf(m, fabsf(W[e]));
    m = wave_max(m);
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = red[0];
#pragma unroll
    for (int i = 1; i < 16; i++) m = fmaxf(m, red[i]);
    const int w_up = h_up_field((int)(__float_as_uint(m) >> 23) & 0xff);
    const float sc = __uint_as_float((unsigned)w_up << 23);
    if (tid == 0 && blockIdx.x == 0) w_up_out[0] = w_up;
    for (int f = blockIdx.x * 1024 + tid; f < 12 * CW_KS * 64; f += gridDim.x * 1024) {
        const int lane = f & 63, ks = (f >> 6) % CW_KS, ct = f / (64 * CW_KS);
        const int col = 32 * ct + (lane & 31), k0 = 16 * ks + 8 * (lane >> 5);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = W[(long)(k0 + e) * CW_K3 + col];
        uint4 hi, lo;
        h_split2(v[0], v[1], sc, hi.x, lo.x);
        h_split2(v[2], v[3], sc, hi.y, lo.y);
        h_split2(v[4], v[5], sc, hi.z, lo.z);
        h_split2(v[6], v[7], sc, hi.w, lo.w);
        img[f] = hi;
        img[12 * CW_KS * 64 + f] = lo;
    }
}

__device__ __forceinline__ unsigned cw_quad_bcast(unsigned v, int sel) {
    switch (sel) {
        case 0: return __builtin_amdgcn_update_dpp(0u, v, 0x00, 0xf, 0xf, true);
        case 1: return __builtin_amdgcn_update_dpp(0u, v, 0x55, 0xf, 0xf, true);
        case 2: return __builtin_amdgcn_update_dpp(0u, v, 0xAA, 0xf, 0xf, true);
        default: return __builtin_amdgcn_update_dpp(0u, v, 0xFF, 0xf, 0xf, true);
    }
}
// Four consecutive ROWS of one column out of a row-major fp16 image (tools/ubench/tr16.hip): lane i of a 16-lane group passes the
// address of row (i >> 2), columns 4 (i & 3) .. + 3 of a [4][16] block and receives column i, rows 0 .. 3.
__device__ __forceinline__ uint2 cw_tr4(const unsigned char* p) {
    const cw_h4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) cw_h4*)p);
    return __builtin_bit_cast(uint2, v);
}
// Scale field (biased exponent of the power-of-two multiplier) of a z column: |z| <= (sqrt(3D - 1) |gamma| + |beta|) / keep -- a
// LayerNorm output over 384 elements cannot exceed sqrt(383) = 19.6 in magnitude -- mapped below 2^14
__device__ __forceinline__ int cw_zfield(float g, float b, float inv_keep) {
    const float bound = (19.6f * fabsf(g) + fabsf(b)) * inv_keep;
    const int ebb = ((int)(__float_as_uint(bound) >> 23) & 0xff) + 1;
    return max(1, min(268 - ebb, 254));
}

// LDS-DMA (buffer_load ... lds: 16 / 4 bytes per lane straight into LDS at M0 + 16 / 4 * lane, bounds-checked against the descriptor) as
// inline assembly: the compiler orders every LDS read behind a pending LDS-DMA it KNOWS of with s_waitcnt vmcnt(0) -- the loads would
// be drained at the first ds_read of the next phase instead of landing behind it.  Hidden from its counters they are waited for by
// hand (cw_dma_wait) before the barrier that publishes them.  M0 is compiler-reserved: saved and restored inside the statement.
typedef int cw_i4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ cw_i4 cw_rsrc(const void* p, long bytes) {
    const unsigned long long pa = (unsigned long long)p;
    return (cw_i4){(int)(unsigned)pa, (int)(unsigned)((pa >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
__device__ __forceinline__ void cw_dma16(cw_i4 rsrc, unsigned lds, int voff, int soff) {
    unsigned keep;
    lds = __builtin_amdgcn_readfirstlane(lds);        // (uniform by construction; the allocator does not always keep them scalar)
    soff = __builtin_amdgcn_readfirstlane(soff);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ void cw_dma4(cw_i4 rsrc, unsigned lds, int voff, int soff) {
    unsigned keep;
    lds = __builtin_amdgcn_readfirstlane(lds);        // (uniform by construction; the allocator does not always keep them scalar)
    soff = __builtin_amdgcn_readfirstlane(soff);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ void cw_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// the lane id, RECOMPUTED (two mbcnt instructions on an opaque all-ones mask) instead of kept: see dma_tile
__device__ __forceinline__ int cw_lane() {
    unsigned m = ~0u;
    asm volatile("" : "+s"(m));
    return (int)__builtin_amdgcn_mbcnt_hi(m, __builtin_amdgcn_mbcnt_lo(m, 0u));
}
// developer build -DCW_PROF: shader-clock cycles per phase of workgroup 0's first W wave (slots 0..) and first E wave (16..), read back
// through stage_cw_prof(); the product build compiles none of it
#ifdef CW_PROF
__device__ unsigned long long cw_prof_buf[32];
#define CW_PROF_DECL unsigned long long pf[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pt = __builtin_readcyclecounter()
#define CW_MARK(slot) do { const unsigned long long t_ = __builtin_readcyclecounter(); pf[slot] += t_ - pt; pt = t_; } while (0)
#define CW_PROF_OUT(base, who) do { if (blockIdx.x == 0 && tid == (who)) for (int i_ = 0; i_ < 10; i_++) cw_prof_buf[(base) + i_] = pf[i_]; } while (0)
#else
#define CW_PROF_DECL
#define CW_MARK(slot)
#define CW_PROF_OUT(base, who)
#endif

// MODE 0: rep == 1, tiles of 32 consecutive rows, grid-stride.  MODE 1: rep > 1, inner <= 32: one tile per frame.  MODE 2: rep > 1,
// inner == 40: per four frames four main tiles + one rest tile.  MODE 3: ragged token rows with the balanced work table `wtab`
// (cat3_fused.hip: cf_bwd_kernel).  MODE 1 / 2: the workgroup walks the (group, chunk of frames) items blockIdx.x, + gridDim.x, ...
template <bool DROP, int MODE>
__global__ __launch_bounds__(512, 2) void cw_bwd_kernel(const float* __restrict__ dy, const unsigned* __restrict__ rmask,
                                                        const uint4* __restrict__ wimg, const int* __restrict__ w_up_p,
                                                        const float* __restrict__ a, const float* __restrict__ b,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ da, float* __restrict__ db, float* __restrict__ part,
                                                        float* __restrict__ partW, float* __restrict__ partB, long M, int rep, int inner,
                                                        int CH, int frames_per_chunk, int n_items, uint64_t seed, uint32_t th,
                                                        float inv_keep, const int4* __restrict__ gdesc, long b_rows, long a_rows,
                                                        const int* __restrict__ wtab) {
    constexpr bool REP = MODE > 0;
    constexpr bool RAG = MODE == 3;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_w = wave >= 4;                          // (uniform) role of this wave
    const int wv = wave & 3;
    const int l31 = lane & 31, h = lane >> 5, h4 = 4 * h;
    const int c = 32 * wv + l31;                          // column inside each third (E: its dX / epilogue column, W: its dW column)
    constexpr int K3 = CW_K3;
    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, (int)(M * CW_D * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_mk = __builtin_amdgcn_make_buffer_rsrc((void*)rmask, 0, (int)(M * (CW_D / 32) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)b, 0, (int)((RAG ? b_rows : M) * CW_D * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, (int)((RAG ? a_rows : (REP ? M / rep : M)) * CW_D * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_mean = __builtin_amdgcn_make_buffer_rsrc((void*)mean, 0, (int)(M * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_rstd = __builtin_amdgcn_make_buffer_rsrc((void*)rstd, 0, (int)(M * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_db = __builtin_amdgcn_make_buffer_rsrc((void*)db, 0, (int)((RAG ? b_rows : M) * CW_D * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_da = __builtin_amdgcn_make_buffer_rsrc((void*)da, 0, REP ? 0 : (int)(M * CW_D * 4), 0x00020000);
    int* const row_up = reinterpret_cast<int*>(smem + CW_OFF_UP);
    int* const tmin = reinterpret_cast<int*>(smem + CW_OFF_TE);
    float* const st_part = reinterpret_cast<float*>(smem + CW_OFF_ST);
    if (tid < 3) tmin[tid] = 254;
    __syncthreads();
    if (is_w) {                                           // z's column scale folded into gamma / beta: read back per use (registers are for accW)
        float* gzl = reinterpret_cast<float*>(smem + CW_OFF_GZ);
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const float g_ = gamma[t * CW_D + c], b_ = beta[t * CW_D + c];
            const float zc = __uint_as_float((unsigned)cw_zfield(g_, b_, inv_keep) << 23) * inv_keep;
            gzl[t * CW_D + c] = g_ * zc;
            gzl[K3 + t * CW_D + c] = b_ * zc;
        }
    }
    __syncthreads();

    // ---- work of this workgroup (every wave walks the same tiles: the control flow below is uniform over the workgroup) ----
    int s_beg = blockIdx.x, s_end = REP ? n_items : (int)blockIdx.x + 1, s_step = REP ? (int)gridDim.x : 1;
    const int4* segs = nullptr;
    if (RAG) {
        s_beg = __builtin_amdgcn_readfirstlane(wtab[blockIdx.x]);
        s_end = __builtin_amdgcn_readfirstlane(wtab[blockIdx.x + 1]);
        s_step = 1;
        segs = reinterpret_cast<const int4*>(wtab + CW_WTAB_SEG(gridDim.x, (int)(a_rows / inner)));
    }
    const long GR = REP ? (long)rep * inner : 0;
    int grp = 0, f_beg = 0, f_end = 0;
    long n_tiles = 0;
    int rg_row0 = 0, rg_seq0 = 0, rg_lc = 0;
    bool rg_big = false;
    // geometry of tile `t_it` of the current segment: four passes of 8 rows; pass p covers compact rows pb[p] .. + nv[p] - 1 and the
    // rows pbB[p] .. of b / db (uniform values)
    auto geom = [&](long t_it, int (&pb)[4], int (&pbB)[4], int (&nv)[4], bool& rest) {
        rest = false;
        if (RAG) {
            if (!rg_big) {
                const int f = f_beg + (int)t_it;
#pragma unroll
                for (int p2 = 0; p2 < 4; p2++) {
                    pb[p2] = rg_row0 + f * rg_lc + 8 * p2;
                    pbB[p2] = (rg_seq0 + f) * inner + 8 * p2;
                    nv[p2] = max(0, min(8, rg_lc - 8 * p2));
                }
            } else {
                const int quad = (int)(t_it / 5), k = (int)(t_it - 5l * quad);
                rest = k == 4;
                const int f0 = f_beg + 4 * quad;
#pragma unroll
                for (int p2 = 0; p2 < 4; p2++) {
                    const int f = rest ? f0 + p2 : f0 + k;
                    pb[p2] = rg_row0 + f * rg_lc + (rest ? 32 : 8 * p2);
                    pbB[p2] = (rg_seq0 + f) * inner + (rest ? 32 : 8 * p2);
                    nv[p2] = f < f_end ? (rest ? rg_lc - 32 : 8) : 0;
                }
            }
        } else if (MODE == 0) {
            const long t0 = ((long)blockIdx.x + t_it * gridDim.x) * 32;
#pragma unroll
            for (int p2 = 0; p2 < 4; p2++) { pb[p2] = (int)(t0 + 8 * p2); nv[p2] = (int)max(0l, min(8l, M - (t0 + 8 * p2))); }
        } else if (MODE == 1) {
            const long t0 = (long)grp * GR + (long)(f_beg + t_it) * inner;
#pragma unroll
            for (int p2 = 0; p2 < 4; p2++) { pb[p2] = (int)(t0 + 8 * p2); nv[p2] = max(0, min(8, inner - 8 * p2)); }
        } else {
            const int quad = (int)(t_it / 5), k = (int)(t_it - 5l * quad);
            rest = k == 4;
            const int f0 = f_beg + 4 * quad;
#pragma unroll
            for (int p2 = 0; p2 < 4; p2++) {
                const int f = rest ? f0 + p2 : f0 + k;
                pb[p2] = (int)((long)grp * GR + (long)f * inner + (rest ? 32 : 8 * p2));
                nv[p2] = f < f_end ? 8 : 0;
            }
        }
        if (!RAG) {
#pragma unroll
            for (int p2 = 0; p2 < 4; p2++) pbB[p2] = pb[p2];
        }
#pragma unroll
        for (int p2 = 0; p2 < 4; p2++) {
            pb[p2] = __builtin_amdgcn_readfirstlane(pb[p2]);
            pbB[p2] = __builtin_amdgcn_readfirstlane(pbB[p2]);
            nv[p2] = __builtin_amdgcn_readfirstlane(nv[p2]);
        }
    };
    // (uniform) advance to the next segment; false: no more work
    int sg = s_beg - s_step;
    auto next_segment = [&]() -> bool {
        sg += s_step;
        if (sg >= s_end) return false;
        if (REP) {
            int frames = rep;
            if (RAG) {
                const int4 sd = segs[sg];
                grp = __builtin_amdgcn_readfirstlane(sd.x);
                f_beg = __builtin_amdgcn_readfirstlane(sd.y);
                f_end = __builtin_amdgcn_readfirstlane(sd.z);
                const int4 gd = gdesc[grp];
                rg_row0 = __builtin_amdgcn_readfirstlane(gd.x);
                rg_lc = __builtin_amdgcn_readfirstlane(gd.y);
                frames = rg_lc > 0 ? __builtin_amdgcn_readfirstlane(gd.z) - 1 : 0;
                rg_seq0 = __builtin_amdgcn_readfirstlane(gd.w);
                rg_big = rg_lc > 32;
            } else {
                grp = sg / CH;
                f_beg = (sg % CH) * frames_per_chunk;
                f_end = min(frames, f_beg + frames_per_chunk);
            }
            const int nf = max(f_end - f_beg, 0);
            n_tiles = (MODE == 1 || (RAG && !rg_big)) ? nf : 5l * ((nf + 3) / 4);
        } else {
            const long all = (M + 31) / 32;
            n_tiles = all > (long)blockIdx.x ? (all - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
        }
        return true;
    };


    // LDS-DMA of a tile: wave wv OF EITHER ROLE brings pass wv (8 rows) of dy, b (and a), its 32 mask words; wave 0 / 1 the means / rstds
    // (the W waves for the first tile of a segment, the E waves -- who have the idle cycles -- for every later one).
    // Rows past the end of a tensor read zeros (buffer bounds check); rows past the end of their pass are masked on use.
    const cw_i4 d_dy = cw_rsrc(dy, M * CW_D * 4), d_b = cw_rsrc(b, (RAG ? b_rows : M) * CW_D * 4),
                d_a = cw_rsrc(a, (RAG ? a_rows : (REP ? M / rep : M)) * CW_D * 4), d_mk = cw_rsrc(rmask, M * (CW_D / 32) * 4),
                d_mean = cw_rsrc(mean, M * 4), d_rstd = cw_rsrc(rstd, M * 4);
    const unsigned lds0 = (unsigned)(size_t)(cw_lds_ptr)smem;
    // (pieces: bit k < 4 = the row pairs 2 k, 2 k + 1 of the pass, bit 4 = the 4-byte words.  Spreading the pieces over the E waves'
    // epilogue -- so that the four of them do not queue 48 KB on the CU's address path at once -- cost more in spills than it saved)
    auto dma_tile = [&](const int (&pb)[4], const int (&pbB)[4], int nb3, int pieces) {
        const int pbw = wv == 0 ? pb[0] : (wv == 1 ? pb[1] : (wv == 2 ? pb[2] : pb[3]));
        const int pbBw = wv == 0 ? pbB[0] : (wv == 1 ? pbB[1] : (wv == 2 ? pbB[2] : pbB[3]));
        // (every per-lane constant below is re-derived from an OPAQUE copy of the lane id: hoisted out of the tile loop they are a
        // dozen loop invariants next to 192 accumulator registers -- the allocator parks them in scratch and every reload is a memory
        // round trip that nothing overlaps in a wave that is alone on its SIMD)
        const int lane_o = cw_lane();
        const int l31 = lane_o & 31, h = lane_o >> 5;
        const int vo = (h * CW_D + 4 * l31) * 4;      // row lane >> 5 of a pair of rows, float4 lane & 31
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (!((pieces >> k) & 1)) continue;
            cw_dma16(d_dy, lds0 + CW_OFF_RAW + (8 * wv + 2 * k) * (CW_D * 4), vo, (pbw + 2 * k) * (CW_D * 4));
            cw_dma16(d_b, lds0 + CW_OFF_BT + nb3 * CW_TILE + (8 * wv + 2 * k) * (CW_D * 4), vo, (pbBw + 2 * k) * (CW_D * 4));
            if (!REP) cw_dma16(d_a, lds0 + CW_OFF_AT + nb3 * CW_TILE + (8 * wv + 2 * k) * (CW_D * 4), vo, (pbBw + 2 * k) * (CW_D * 4));
        }
        // the 4-byte words: 32 per instruction, by the lower lane half only (an inactive lane writes nothing; an active lane whose
        // offset is out of range would write a ZERO to its slot -- the next array)
        if (h == 0 && (pieces & 16)) {
            // its own mask words: word lane >> 3 of row lane & 7
            cw_dma4(d_mk, lds0 + CW_OFF_MK + wv * 128, (int)(((long)(l31 >> 3) * M + (l31 & 7)) * 4), pbw * 4);
            // per-row statistics of the 32 rows: tile row `lane` = row (lane & 7) of pass lane >> 3
            const int pr = l31 >> 3;
            const int prow = (pr == 0 ? pb[0] : (pr == 1 ? pb[1] : (pr == 2 ? pb[2] : pb[3]))) + (l31 & 7);
            if (wv == 0) cw_dma4(d_mean, lds0 + CW_OFF_MS + nb3 * 256, prow * 4, 0);
            if (wv == 1) cw_dma4(d_rstd, lds0 + CW_OFF_MS + nb3 * 256 + 128, prow * 4, 0);
        }
    };
    if (is_w) {
        // =============================================================================================================
        // W waves: LDS-DMA two tiles ahead, staging + dropout hashes of the next tile, z and the dW product of the previous one
        // =============================================================================================================
        f32x16 accW[4][3];                                // dW rows 32 nt + (C/D row), column 128 t + c
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
#pragma unroll
            for (int t = 0; t < 3; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) accW[nt][t][r] = 0.f;
        int E_acc = 254;                                  // scale field the accumulators are held at = running minimum (uniform)
        float dbs[4] = {0.f, 0.f, 0.f, 0.f};
        unsigned kbA0 = 0xffffffu, kbA1 = 0xffffffu, kbN0 = 0xffffffu, kbN1 = 0xffffffu;   // dropout bits: tile whose dW is pending / newest
        // staging of a tile (this wave: its pass): gate by the ReLU bits, one power-of-two scale per row, two fp16 planes, row-major; the
        // bias gradient; the tile's smallest scale field
        auto stage_tile = [&](const int (&nv)[4], int pbuf, int nb3) {
            const int lane_o = cw_lane();
            const int l31 = lane_o & 31, h = lane_o >> 5;
            const int nvw = wv == 0 ? nv[0] : (wv == 1 ? nv[1] : (wv == 2 ? nv[2] : nv[3]));
            unsigned char* const Rp = smem + CW_OFF_RP + pbuf * CW_RP;
            const float* raw = reinterpret_cast<const float*>(smem + CW_OFF_RAW) + (8 * wv + h) * CW_D + 4 * l31;
            const unsigned* mk = reinterpret_cast<const unsigned*>(smem + CW_OFF_MK) + wv * 32 + (l31 >> 3) * 8 + h;
#pragma unroll 1                                           // (rolled, like the hashes below: these waves have the time, not the registers -- an
            for (int k = 0; k < ((CW_ABL & 16) ? 0 : 4); k++) {   //  unrolled body makes the allocator park all of accW in scratch around it)
                const int rl = 8 * wv + 2 * k + h;
                const bool ok = 2 * k + h < nvw;
                const unsigned wbits = mk[2 * k] >> (4 * (l31 & 7));      // the bits of this lane's 4 columns
                float4 v = *reinterpret_cast<const float4*>(raw + 2 * k * CW_D);
                v.x = (ok && (wbits & 1u)) ? v.x : 0.f;
                v.y = (ok && (wbits & 2u)) ? v.y : 0.f;
                v.z = (ok && (wbits & 4u)) ? v.z : 0.f;
                v.w = (ok && (wbits & 8u)) ? v.w : 0.f;
                dbs[0] += v.x; dbs[1] += v.y; dbs[2] += v.z; dbs[3] += v.w;
                float m = h_amax3(h_amax3(v.x, v.y, v.z), v.w, v.w);
                m = group_max(m, 32);
                const int up = h_up_field((int)(__float_as_uint(m) >> 23) & 0xff);
                const float sc = __uint_as_float((unsigned)up << 23);
                unsigned h01, l01, h23, l23;
                h_split2(v.x, v.y, sc, h01, l01);
                h_split2(v.z, v.w, sc, h23, l23);
                *reinterpret_cast<uint2*>(Rp + rl * CW_PITCH + 8 * l31) = make_uint2(h01, h23);
                *reinterpret_cast<uint2*>(Rp + 32 * CW_PITCH + rl * CW_PITCH + 8 * l31) = make_uint2(l01, l23);
                if (l31 == 0) {
                    row_up[nb3 * 32 + rl] = up;
                    atomicMin(&tmin[nb3], up);
                }
            }
        };
        // dropout stream of stage_cat3_layernorm_fwd: element row * 3D + t * D + c, one hash per 4 consecutive columns = the 4 lanes of
        // a quad: quad lane q hashes for the C/D registers 4 j + q (row pb[j] + q + 4 h); three keep nibbles per pass j, two passes per word
        auto hash_tile = [&](const int (&pb)[4]) {
            unsigned w0 = 0u, w1 = 0u;
            const int lane_o = cw_lane();
            const int l31 = lane_o & 31, h4 = 4 * (lane_o >> 5), c = 32 * wv + l31;
            if (DROP && !(CW_ABL & 4)) {
                const unsigned drop_lane = (unsigned)(((l31 & 3) + h4) * (K3 / 4) + (c >> 2));
                // mix64(seed, idx) = finish(seed + (idx + 1) * C0): the three thirds of a pass are idx, idx + D/4, idx + 2 D/4 -- one 64-bit
                // multiply per pass and two 64-bit adds of constants instead of three multiplies (same values: arithmetic mod 2^64)
                constexpr uint64_t C0 = 0x9E3779B97F4A7C15ull;
#pragma unroll 1
                for (int j = 0; j < 4; j++) {             // pass j (rolled: these waves have the time, not the registers)
                    const int pbj = j == 0 ? pb[0] : (j == 1 ? pb[1] : (j == 2 ? pb[2] : pb[3]));
                    const uint64_t zj = seed + ((uint64_t)pbj * (uint64_t)(K3 / 4) + (uint64_t)drop_lane + 1ull) * C0;
                    unsigned bj = 0u;
#pragma unroll
                    for (int t = 0; t < 3; t++) {
                        uint64_t z = zj + (uint64_t)(t * (CW_D / 4)) * C0;
                        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
                        const uint64_t hh = z ^ (z >> 31);
                        const unsigned lo = (unsigned)hh, hi = (unsigned)(hh >> 32);
                        const unsigned bits = ((lo & 0xFFFFu) >= th ? 1u : 0u) | ((lo >> 16) >= th ? 2u : 0u) | ((hi & 0xFFFFu) >= th ? 4u : 0u) |
                                              ((hi >> 16) >= th ? 8u : 0u);
                        bj |= bits << (4 * t);
                    }
                    bj <<= 12 * (j & 1);
                    if (j < 2) w0 |= bj; else w1 |= bj;
                }
            } else {
                w0 = w1 = 0xffffffu;
            }
            kbN0 = w0;
            kbN1 = w1;
        };
        auto publish_bits = [&]() {                       // the newest bits -> LDS for the E waves (behind Bb: they read a tile's bits right after Ba)
            unsigned* kbp = reinterpret_cast<unsigned*>(smem + CW_OFF_KB) + wv * 128 + cw_lane();
            kbp[0] = kbN0;
            kbp[64] = kbN1;
        };
        // z and the dW product of one tile: planes `pbuf`, operand buffers `nb3`, dropout bits kbA, pass lengths nvp
        auto dw_tile = [&](int pbuf, int nb3, const int (&nvp)[4], bool restp) {
            const int lane_o = cw_lane();
            const int l31 = lane_o & 31, h4 = 4 * (lane_o >> 5), c = 32 * wv + l31, q = l31 & 3;
            const int G = lane_o >> 4, i16 = lane_o & 15;
            const int tr_off = (4 * (G >> 1) + (i16 >> 2)) * CW_PITCH + (16 * (G & 1) + 4 * (i16 & 3)) * 2;
            const int Et = __builtin_amdgcn_readfirstlane(tmin[nb3]);
            if (Et < E_acc) {                             // (uniform, rare) a row larger than anything so far: bring the accumulators along
                const int d = Et - E_acc;
#pragma unroll
                for (int nt = 0; nt < 4; nt++)
#pragma unroll
                    for (int t = 0; t < 3; t++)
#pragma unroll
                        for (int r = 0; r < 16; r++) accW[nt][t][r] = __builtin_ldexpf(accW[nt][t][r], d);
                E_acc = Et;
            }
            const unsigned char* const trow = smem + CW_OFF_RP + pbuf * CW_RP + tr_off;
            const float* const bt_h = reinterpret_cast<const float*>(smem + CW_OFF_BT + nb3 * CW_TILE) + h4 * CW_D + c;
            const float* const at_h = reinterpret_cast<const float*>(smem + CW_OFF_AT + (REP ? 0 : nb3 * CW_TILE)) + h4 * CW_D + c;
            const int* const up_h = row_up + nb3 * 32 + h4;
            const float* const mu_h = reinterpret_cast<const float*>(smem + CW_OFF_MS + nb3 * 256) + h4;
            const float* const gz_c = reinterpret_cast<const float*>(smem + CW_OFF_GZ) + c;
            // (gamma' / beta' of this lane's three columns: read HERE and pinned -- left inside the keep-select below the compiler turns
            // every one of the 48 elements into a branch with two LDS reads and their full latency inside)
            float gzv[3], bzv[3];
#pragma unroll
            for (int t = 0; t < 3; t++) {
                gzv[t] = gz_c[t * CW_D];
                bzv[t] = gz_c[K3 + t * CW_D];
                asm volatile("" : "+v"(gzv[t]), "+v"(bzv[t]));
            }
#pragma unroll
            for (int s = 0; s < ((CW_ABL & 2) ? 0 : 2); s++) {
                // z of the 8 rows of k-step s in this lane's column of each third (element e = C/D register r = 8 s + e), split pair by
                // pair into the B operands
                uint4 zh[3], zl[3];
                float zprev[3] = {0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int r = 8 * s + e, rl = 8 * (r >> 2) + (r & 3);       // + 4 h
                    const bool valid = h4 < nvp[r >> 2] - (r & 3);
                    int upv = up_h[rl];
                    asm volatile("" : "+v"(upv));
                    const float f = valid ? __builtin_ldexpf(1.0f, E_acc - upv) : 0.f;
                    const float mu = mu_h[rl], rsv = mu_h[32 + rl];
                    const float rs = valid ? rsv : 0.f;
                    const float bvr = bt_h[rl * CW_D];
                    const float avr = at_h[((REP && MODE >= 2 && restp) ? 32 + (r & 3) : rl) * CW_D];
                    unsigned bw = 0xfffu;
                    if (DROP) bw = cw_quad_bcast(((r >> 3) ? kbA1 : kbA0) >> (12 * ((r >> 2) & 1)), r & 3) >> q;
#pragma unroll
                    for (int t = 0; t < 3; t++) {
                        const float x = t == 0 ? avr : (t == 1 ? bvr : avr * bvr);
                        const float xh = (x - mu) * rs;
                        float zfull = (xh * gzv[t] + bzv[t]) * f;
                        asm volatile("" : "+v"(zfull));
                        const float zv = (!(CW_ABL & 32) && ((bw >> (4 * t)) & 1u)) ? zfull : 0.f;
                        if (e & 1) {
                            unsigned hi, lo;
                            h_split2(zprev[t], zv, 1.0f, hi, lo);
                            if ((e >> 1) == 0) { zh[t].x = hi; zl[t].x = lo; }
                            else if ((e >> 1) == 1) { zh[t].y = hi; zl[t].y = lo; }
                            else if ((e >> 1) == 2) { zh[t].z = hi; zl[t].z = lo; }
                            else { zh[t].w = hi; zl[t].w = lo; }
                        } else zprev[t] = zv;
                    }
                    if (e & 1) __builtin_amdgcn_sched_barrier(0);     // a pair of rows at a time: the accumulators leave few registers
                }
                sf16x8 vzh[3], vzl[3];
#pragma unroll
                for (int t = 0; t < 3; t++) {
                    h_operands_ready(zh[t].x, zh[t].y, zh[t].z, zh[t].w);
                    h_operands_ready(zl[t].x, zl[t].y, zl[t].z, zl[t].w);
                    vzh[t] = __builtin_bit_cast(sf16x8, zh[t]);
                    vzl[t] = __builtin_bit_cast(sf16x8, zl[t]);
                }
#pragma unroll
                for (int nt = 0; nt < 4; nt++) {
                    // A operand lane (n = 32 nt + l31, h), element e <-> tile row 16 s + 8 (e >> 2) + (e & 3) + 4 h: two transposed reads
                    // (e < 4, e >= 4); 16-lane group G covers columns 32 nt + 16 (G & 1) .. + 15, rows .. + 4 (G >> 1)
                    const unsigned char* p = trow + (16 * s) * CW_PITCH + 64 * nt;
                    const uint2 h0 = cw_tr4(p), h1 = cw_tr4(p + 8 * CW_PITCH);
                    const uint2 l0 = cw_tr4(p + 32 * CW_PITCH), l1 = cw_tr4(p + 32 * CW_PITCH + 8 * CW_PITCH);
                    const sf16x8 ah = __builtin_bit_cast(sf16x8, make_uint4(h0.x, h0.y, h1.x, h1.y));
                    const sf16x8 al = __builtin_bit_cast(sf16x8, make_uint4(l0.x, l0.y, l1.x, l1.y));
#pragma unroll
                    for (int t = 0; t < 3; t++) {
                        accW[nt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, vzh[t], accW[nt][t], 0, 0, 0);
                        accW[nt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, vzl[t], accW[nt][t], 0, 0, 0);
                        accW[nt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, vzh[t], accW[nt][t], 0, 0, 0);
                    }
                }
            }
        };
        int tc = 0;                                       // tiles so far: tile tc uses planes tc & 1, operand buffers tc % 3
        CW_PROF_DECL;
        while (next_segment()) {
            if (REP) { __syncthreads(); __syncthreads(); }    // (the E waves copy the group's block of `a` between these two)
            if (n_tiles <= 0) continue;
            int pb[4], pbB[4], nv[4], nvp[4] = {0, 0, 0, 0};
            bool rest, restp = false;
            // prologue: tile 0 of the segment is brought in and staged, tile 1 requested
            geom(0, pb, pbB, nv, rest);
            dma_tile(pb, pbB, tc % 3, 31);
            cw_dma_wait();
            __syncthreads();                              // P1 (its mask words / rows are this wave's own; b / a / statistics: for the E waves)
            stage_tile(nv, tc & 1, tc % 3);
            hash_tile(pb);
            publish_bits();
            // (ONE call site of the product inside the loop -- with a second one the allocator parks all of accW in scratch between
            // them: the last tile of a segment is drained by an extra trip that runs the product only)
            CW_MARK(0);                                   // (segment prologue)
            for (long it = 0;; it++) {
                if (it < n_tiles) __syncthreads();        // Ba(it)
                CW_MARK(1);
                if (it > 0) dw_tile((tc - 1) & 1, (tc + 2) % 3, nvp, restp);       // tile tc - 1
                CW_MARK(2);
                if (it >= n_tiles) break;
                if (tid == 256) tmin[(tc + 1) % 3] = 254; // (the word of tile tc + 1 = of tile tc - 2: read in the previous slot, written behind Bb)
                kbA0 = kbN0; kbA1 = kbN1;                 // (the bits of tile tc: pending from here on)
#pragma unroll
                for (int p2 = 0; p2 < 4; p2++) nvp[p2] = nv[p2];
                restp = rest;
                // the dropout hashes of tile tc + 1 go HERE, next to the E waves' product (their slot behind Bb is the short one); the
                // bits stay in registers until Bb
                if (MODE == 0 || it + 1 < n_tiles) {      // (MODE 0: one segment -- the one wasted hash keeps the allocator's plan of the loop)
                    geom(it + 1, pb, pbB, nv, rest);
                    hash_tile(pb);
                }
                CW_MARK(6);
                __syncthreads();                          // Bb(it): the rows of tile tc + 1 have landed (requested and awaited by the E waves)
                CW_MARK(4);
                if (it + 1 < n_tiles) {
                    stage_tile(nv, (tc + 1) & 1, (tc + 1) % 3);
                    publish_bits();
                    CW_MARK(5);
                }
                tc++;
            }
        }
        CW_PROF_OUT(0, 256);
        // weight gradient in true units: the accumulators hold dW * 2^(zf_t - 127) * 2^(E_acc - 127)
        float* pw = partW + (size_t)blockIdx.x * CW_D * K3;
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const int e = 254 - E_acc - cw_zfield(gamma[t * CW_D + c], beta[t * CW_D + c], inv_keep);
#pragma unroll
            for (int nt = 0; nt < 4; nt++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    pw[(size_t)(32 * nt + 8 * (r >> 2) + (r & 3) + h4) * K3 + t * CW_D + c] = __builtin_ldexpf(accW[nt][t][r], e);
        }
        // bias gradient of the Linear: the 8 staging rows (4 waves x 2 lane halves) of a column meet in LDS
        __syncthreads();                                  // (every wave is past its last tile: the planes are reduction scratch)
        float* red = reinterpret_cast<float*>(smem + CW_OFF_RP);          // [8][128]
        *reinterpret_cast<float4*>(red + (2 * wv + h) * CW_D + 4 * l31) = make_float4(dbs[0], dbs[1], dbs[2], dbs[3]);
        __syncthreads();
        if (tid - 256 < CW_D) {
            float sm = 0.f;
#pragma unroll
            for (int r = 0; r < 8; r++) sm += red[r * CW_D + (tid - 256)];
            partB[(size_t)blockIdx.x * CW_D + (tid - 256)] = sm;
        }
        return;
    }

    // =================================================================================================================
    // E waves: dX product, LayerNorm backward
    // =================================================================================================================
    const int w_up = w_up_p[0];
    const float invK = 1.0f / (float)K3;
    float gm[3], ag[3] = {0.f, 0.f, 0.f}, ab[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 3; t++) gm[t] = gamma[t * CW_D + c];
    const int vo_row = (h4 * CW_D + c) * 4;               // byte offset of (row 4 h, column c) in a (rows, D) tensor
    // gradient of the broadcast operand: slot r of a main tile = position 8 (r >> 2) + (r & 3) + 4 h; the rest tile adds its registers
    // r, r + 4, r + 8, r + 12 (four frames) into slot r & 3 = position 32 + (r & 3) + 4 h
    float dacc[16], dacc_rest[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; r++) dacc[r] = 0.f;
    auto write_slab = [&](int slab) {                     // da slab [inner][D] of an item / segment
        float* dst = da + (size_t)slab * inner * CW_D;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int pos = 8 * (r >> 2) + (r & 3) + h4;
            if (pos < min(inner, 32)) dst[pos * CW_D + c] = dacc[r];
        }
        if (MODE >= 2) {                                  // (ragged: every one of the Lqa positions is written, zeros past the live words)
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (32 + r + h4 < inner) dst[(32 + r + h4) * CW_D + c] = dacc_rest[r];
        }
    };
    int tc = 0;
    CW_PROF_DECL;
    while (next_segment()) {
        if (REP) {
            // the group's block of the broadcast operand -> LDS (behind a barrier: a slower wave may still read the previous one)
            __syncthreads();
            float4* dst = reinterpret_cast<float4*>(smem + CW_OFF_AT);
            const float4* src = reinterpret_cast<const float4*>(a + (size_t)grp * inner * CW_D);
            for (int e = tid; e < 40 * (CW_D / 4); e += 256) dst[e] = e < inner * (CW_D / 4) ? src[e] : f4zero();   // (slots past the last
            __syncthreads();                              //  position read zeros: their gradients are masked, not their values)
        }
        if (n_tiles <= 0) continue;
        __syncthreads();                                  // P1
        CW_MARK(0);
        for (long it = 0; it < n_tiles; it++) {
            int pb[4], pbB[4], nv[4];
            bool rest;
            geom(it, pb, pbB, nv, rest);
            const int nb3 = tc % 3;
            const unsigned char* const Rp = smem + CW_OFF_RP + (tc & 1) * CW_RP;
            const float* const bt_b = reinterpret_cast<const float*>(smem + CW_OFF_BT + nb3 * CW_TILE);
            const float* const ms_b = reinterpret_cast<const float*>(smem + CW_OFF_MS + nb3 * 256);
            CW_MARK(1);
            __syncthreads();                              // Ba: planes, row scales, dropout bits of the tile
            CW_MARK(2);
            unsigned kb0 = 0xffffffu, kb1 = 0xffffffu;
            if (DROP) {
                const unsigned* kbp = reinterpret_cast<const unsigned*>(smem + CW_OFF_KB) + wv * 128 + lane;
                kb0 = kbp[0];
                kb1 = kbp[64];
            }
            // ---- dX product: acc[third] = dy tile (32 x 128) . W[:, columns 32 w .. 32 w + 31 of each third].  A operand lane (row l31,
            // h), k-step ks = 16 bytes at column 16 ks + 8 h of its row (a 4-way bank conflict on 16 reads per tile: the pitch serves the
            // transposed reads of the W waves) ----
            f32x16 acc[3];
#pragma unroll
            for (int t = 0; t < 3; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
            {
                const unsigned char* const arow = Rp + l31 * CW_PITCH + 16 * h;
                auto load_b = [&](sf16x8 (&bf)[3][2], int ks) {
#pragma unroll
                    for (int t = 0; t < 3; t++)
#pragma unroll
                        for (int p2 = 0; p2 < 2; p2++)
                            bf[t][p2] = __builtin_bit_cast(sf16x8, wimg[(size_t)p2 * 12 * CW_KS * 64 + ((4 * t + wv) * CW_KS + ks) * 64 + lane]);
                };
                auto mul_b = [&](const sf16x8 (&bf)[3][2], int ks) {
                    sf16x8 af[2];
#pragma unroll
                    for (int p2 = 0; p2 < 2; p2++) af[p2] = __builtin_bit_cast(sf16x8, *reinterpret_cast<const uint4*>(arow + p2 * 32 * CW_PITCH + 32 * ks));
#pragma unroll
                    for (int t = 0; t < 3; t++) {
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1], bf[t][0], acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0], bf[t][1], acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0], bf[t][0], acc[t], 0, 0, 0);
                    }
                };
                // weight fragments double-buffered by hand, two k-steps per trip of a loop the compiler must not unroll
                sf16x8 bfa[3][2], bfb[3][2];
                load_b(bfa, 0);
#pragma unroll 1
                for (int ks = 0; ks < ((CW_ABL & 1) ? 0 : CW_KS); ks += 2) {
                    load_b(bfb, ks + 1);
                    mul_b(bfa, ks);
                    load_b(bfa, ks + 2 < CW_KS ? ks + 2 : 0);  // (the last request is a harmless re-read of k-step 0)
                    mul_b(bfb, ks + 1);
                }
            }
            // the LDS-DMA of tile it + 1 (staged by the W waves behind Bb(it)): requested here, behind this tile's weight-fragment loads
            // (vmcnt retires in order: in front of them it would put an HBM round trip into the product loop), awaited before Bb
            if (it + 1 < n_tiles) {
                int pbn[4], pbBn[4], nvn[4];
                bool restn;
                geom(it + 1, pbn, pbBn, nvn, restn);
                dma_tile(pbn, pbBn, (tc + 1) % 3, 31);
            }
            CW_MARK(3);
            // ---- LayerNorm backward, first half: gradient of the LayerNorm output in true units, row statistics.  Slot r = tile row
            // 8 (r >> 2) + (r & 3) + 4 h, column c of each third; a / b values from the LDS tiles ----
            const float* const bt_h = bt_b + h4 * CW_D + c;
            const float* const at_h = reinterpret_cast<const float*>(smem + CW_OFF_AT + (REP ? 0 : nb3 * CW_TILE)) + h4 * CW_D + c;
            const int* const up_h = row_up + nb3 * 32 + h4;
            const float* const mu_h = ms_b + h4;
            float* const stw_h = st_part + (wv * 32 + h4) * 2;
            const float* const sta_h = st_part + h4 * 2;
            const int q = l31 & 3;
#pragma unroll
            for (int r = 0; r < ((CW_ABL & 8) ? 0 : 16); r++) {
                const int rl = 8 * (r >> 2) + (r & 3);    // + 4 h
                const bool valid = h4 < nv[r >> 2] - (r & 3);
                int upv = up_h[rl];
                asm volatile("" : "+v"(upv));             // (keeps the load out of the select: no branch per slot)
                float un = valid ? __builtin_ldexpf(1.0f, 254 - upv - w_up) : 0.f;
                unsigned bw = 0xfffu;
                if (DROP) {
                    bw = cw_quad_bcast(((r >> 3) ? kb1 : kb0) >> (12 * ((r >> 2) & 1)), r & 3) >> q;
                    un *= inv_keep;
                }
                const float mu = mu_h[rl], rsv = mu_h[32 + rl];
                const float rs = valid ? rsv : 0.f;
                const float bvr = bt_h[rl * CW_D];
                const float avr = at_h[((REP && MODE >= 2 && rest) ? 32 + (r & 3) : rl) * CW_D];
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int t = 0; t < 3; t++) {
                    const float dd = ((bw >> (4 * t)) & 1u) ? acc[t][r] * un : 0.f;
                    acc[t][r] = dd;
                    const float x = t == 0 ? avr : (t == 1 ? bvr : avr * bvr);
                    const float xh = (x - mu) * rs;
                    const float gq = dd * gm[t];
                    s1 += gq;
                    s2 += gq * xh;
                    ag[t] += dd * xh;
                    ab[t] += dd;
                }
                s1 = group_sum(s1, 32);
                s2 = group_sum(s2, 32);
                *reinterpret_cast<float2*>(stw_h + rl * 2) = make_float2(s1, s2);      // (all 32 lanes of the half: the same value)
                if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // four rows in flight at a time: bounded register pressure
            }
            cw_dma_wait();
            CW_MARK(4);
            __syncthreads();                              // Bb: the partial statistics of all four E waves; tile it + 1's rows in LDS
            CW_MARK(5);
            // ---- second half: dz and the gradients of a and b ----
#pragma unroll
            for (int r = 0; r < ((CW_ABL & 8) ? 0 : 16); r++) {
                const int rl = 8 * (r >> 2) + (r & 3);
                const bool valid = h4 < nv[r >> 2] - (r & 3);
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int w2 = 0; w2 < 4; w2++) {          // fixed order: identical totals in all four waves
                    const float2 p = *reinterpret_cast<const float2*>(sta_h + (w2 * 32 + rl) * 2);
                    s1 += p.x;
                    s2 += p.y;
                }
                s1 *= invK;
                s2 *= invK;
                const float mu = mu_h[rl], rsv = mu_h[32 + rl];
                const float rs = valid ? rsv : 0.f;
                const float bvr = bt_h[rl * CW_D];
                const float avr = at_h[((REP && MODE >= 2 && rest) ? 32 + (r & 3) : rl) * CW_D];
                float dz[3];
#pragma unroll
                for (int t = 0; t < 3; t++) {
                    const float x = t == 0 ? avr : (t == 1 ? bvr : avr * bvr);
                    const float xh = (x - mu) * rs;
                    dz[t] = rs * (acc[t][r] * gm[t] - s1 - xh * s2);
                }
                // z = [a, b, a*b]:  da = dz0 + dz2 * b ; db = dz1 + dz2 * a
                const float da_v = dz[0] + dz[2] * bvr, db_v = dz[1] + dz[2] * avr;
                const int so = pbB[r >> 2] * (CW_D * 4) + (r & 3) * (CW_D * 4);
                // an invalid slot (a row of the next frame / past the end) gets an out-of-range lane offset: dropped by the bounds check
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(db_v), rs_db, valid ? vo_row : 0x7ffffff0, so, 0);
                if (!REP) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(da_v), rs_da, valid ? vo_row : 0x7ffffff0, so, 0);
                else if (MODE >= 2 && rest) dacc_rest[r & 3] += valid ? da_v : 0.f;
                else dacc[r] += valid ? da_v : 0.f;
                if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            CW_MARK(6);
            tc++;
        }
        if (REP) {
            write_slab(sg);
#pragma unroll
            for (int r = 0; r < 16; r++) dacc[r] = 0.f;
#pragma unroll
            for (int r = 0; r < 4; r++) dacc_rest[r] = 0.f;
        }
    }
    CW_PROF_OUT(16, 0);
    // ---- what the E waves summed over their rows: column partials of d gamma / d beta (the two lane halves hold different rows of
    // the same columns) ----
    {
        float* prow = part + (size_t)blockIdx.x * 2 * K3;
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const float sg2 = xsum32(ag[t], ag[t]), sb = xsum32(ab[t], ab[t]);
            if (h == 0) {
                prow[t * CW_D + c] = sg2;
                prow[K3 + t * CW_D + c] = sb;
            }
        }
    }
    __syncthreads();                                      // (the W waves' bias-gradient reduction: two barriers)
    __syncthreads();
}

inline size_t cw_align(size_t v) { return (v + 255) & ~(size_t)255; }
inline int cw_mode(int rep, int inner) { return rep == 1 ? 0 : (inner <= 32 ? 1 : (inner == 40 ? 2 : -1)); }
inline int cw_grid() {                                    // one persistent workgroup per compute unit
    static int cus = 0;
    if (!cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
            cus = n > 512 ? 512 : n;
        else cus = 256;
        if (getenv("STAGE_CW_GRID")) cus = atoi(getenv("STAGE_CW_GRID"));
    }
    return cus;
}
// (group, chunk of frames) items of the dense broadcast modes: ~4 per workgroup; MODE 2 walks the frames four at a time
void cw_chunks(long long groups, int rep, int mode, int* CH, int* fpc) {
    const long target = 4l * cw_grid();
    int ch = (int)((target + groups - 1) / groups);
    if (ch > rep) ch = rep;
    if (ch < 1) ch = 1;
    int f = (rep + ch - 1) / ch;
    if (mode == 2) f = (f + 3) / 4 * 4;
    *fpc = f;
    *CH = (rep + f - 1) / f;
}
// ON by default since round 6 (STAGE_CAT3_DW=0 switches back to cf_bwd_kernel + the weight-gradient GEMM on a saved z).  Measured on
// MI355X at 960 000 rows (profiles/r06_cat3_dw_ab.txt): 1.43 ms (broadcast a) / 1.27 ms (flat) for this launch against 0.92 ms +
// 0.42-0.56 ms for the two it replaces, and the forward without the z store saves 0.13-0.15 ms more; in the training step (three
// instances, next to other branches' kernels) 0.4-0.85 ms per step in the round's bench runs.  EVERY change of this kernel goes
// through tools/experiments/step_repeat_small.py at >= 6000 steps (docs/findings.md, finding 62: two faster epilogue forms lost bit
// repeatability once in 200 / 2000 training steps and were withdrawn).  The per-phase cycle counters
// (-DCW_PROF) say where the rest is: one E and one W wave per SIMD are two dependency chains in lock step (two barriers per tile), ~13 us
// per tile: dX is bound by the L2 bandwidth of the weight image (192 KB per 32 rows), the epilogue, the z rebuild, staging and the
// hashes by the latency of a single wave with 64 working registers next to its 192 accumulators.  (Read on every call: the tests
// switch it inside one process.)
bool cw_enabled() {
    const char* e = getenv("STAGE_CAT3_DW");
    return !(e != nullptr && e[0] == '0') && getenv("STAGE_NO_CAT3_FUSED") == nullptr;
}
template <typename K>
void cw_set_lds(K kern) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, CW_LDS);
}
// workspace: [weight image | scale word | d gamma / d beta partials | dW partials | dc partials | da slabs]
struct CwWs { uint4* img; int* w_up; float *part, *partW, *partB, *slabs; size_t bytes; };
CwWs cw_ws(void* base, int grid, size_t slabs, int inner) {
    char* p = (char*)base;
    size_t off = 0;
    CwWs w;
    w.img = (uint4*)(p + off); off += cw_align((size_t)CW_WFRAG * sizeof(uint4));
    w.w_up = (int*)(p + off); off += 256;
    w.part = (float*)(p + off); off += cw_align((size_t)grid * 2 * CW_K3 * sizeof(float));
    w.partW = (float*)(p + off); off += cw_align((size_t)grid * CW_D * CW_K3 * sizeof(float));
    w.partB = (float*)(p + off); off += cw_align((size_t)grid * CW_D * sizeof(float));
    w.slabs = (float*)(p + off); off += cw_align(slabs * (size_t)inner * CW_D * sizeof(float));
    w.bytes = off;
    return w;
}
__global__ __launch_bounds__(256) void cw_reduce_seg_kernel(const float* __restrict__ in, float* __restrict__ out, const int2* __restrict__ gseg,
                                                            long groups, long inner4) {
    const long total = groups * inner4;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long gq = e / inner4, q = e % inner4;
        const int2 gs = gseg[gq];
        const float* p = in + ((long)gs.x * inner4 + q) * 4;
        float4 acc = f4zero();
        for (int r = 0; r < gs.y; r++) acc = f4add(acc, ld4(p + (long)r * inner4 * 4));
        st4(out + e * 4, acc);
    }
}
// the ordered sums of the workgroup partials, ONE launch for the four of them (dW: 50 MB of partials at one workgroup per CU -- the
// generic 4-byte column reduction took 45 us for it): a workgroup = 64 column quads x 16 interleaved partial groups, 16-byte loads,
// fixed order (partials b = g, g + 16, ... per group, then the 16 groups in order)
struct CwFin { const float* src[4]; float* dst[4]; long stride[4]; int quads[4]; int blk0[5]; };
__global__ __launch_bounds__(1024) void cw_finish_kernel(CwFin f, int nb) {
    __shared__ float4 sm[16][64];
    const int x = threadIdx.x & 63, g = threadIdx.x >> 6;
    int sidx = 0;
#pragma unroll
    for (int i = 1; i < 4; i++) sidx = (int)blockIdx.x >= f.blk0[i] ? i : sidx;
    const int qd = ((int)blockIdx.x - f.blk0[sidx]) * 64 + x;
    const bool ok = qd < f.quads[sidx];
    float4 acc = f4zero();
    if (ok) {
        const float* p = f.src[sidx] + (size_t)qd * 4;
#pragma unroll 4
        for (int b = g; b < nb; b += 16) acc = f4add(acc, ld4(p + (size_t)b * f.stride[sidx]));
    }
    sm[g][x] = acc;
    __syncthreads();
    if (g == 0 && ok) {
        float4 t = sm[0][x];
#pragma unroll
        for (int j = 1; j < 16; j++) t = f4add(t, sm[j][x]);
        st4(f.dst[sidx] + (size_t)qd * 4, t);
    }
}
int cw_finish(const CwWs& w, int grid, float* dgamma, float* dbeta, float* dW, float* dc, hipStream_t st) {
    CwFin f;
    f.src[0] = w.partW; f.dst[0] = dW; f.stride[0] = (long)CW_D * CW_K3; f.quads[0] = CW_D * CW_K3 / 4;
    f.src[1] = w.partB; f.dst[1] = dc; f.stride[1] = CW_D; f.quads[1] = CW_D / 4;
    f.src[2] = w.part; f.dst[2] = dgamma; f.stride[2] = 2 * CW_K3; f.quads[2] = CW_K3 / 4;
    f.src[3] = w.part + CW_K3; f.dst[3] = dbeta; f.stride[3] = 2 * CW_K3; f.quads[3] = CW_K3 / 4;
    f.blk0[0] = 0;
    for (int i = 0; i < 4; i++) f.blk0[i + 1] = f.blk0[i] + (f.quads[i] + 63) / 64;
    hipLaunchKernelGGL(cw_finish_kernel, dim3(f.blk0[4]), dim3(1024), 0, st, f, grid);
    STAGE_LAUNCH_CHECK();
    return 0;
}
}  // namespace

extern "C" int stage_cat3_bwd_dw_supported(long long rows, int D, int rep, int inner) {
    if (!cw_enabled()) return 0;
    return (D == CW_D && rows >= 4096 && rows * (long long)D * 4 < (1ll << 31) && rep >= 1 && inner >= 1 && cw_mode(rep, inner) >= 0 &&
            rows % ((long long)rep * inner) == 0) ? 1 : 0;
}
extern "C" size_t stage_cat3_bwd_dw_ws_bytes(long long rows, int D, int rep, int inner) {
    if (D != CW_D || rep < 1 || inner < 1 || cw_mode(rep, inner) < 0) return 0;
    size_t slabs = 0;
    if (rep > 1) {
        int CH, fpc;
        cw_chunks(rows / ((long long)rep * inner), rep, cw_mode(rep, inner), &CH, &fpc);
        slabs = (size_t)(rows / ((long long)rep * inner)) * CH;
    }
    return cw_ws(nullptr, cw_grid(), slabs, inner).bytes;
}
// As stage_cat3_dx_ln_bwd (cat3_fused.hip), plus the Linear's own gradients: dW (D, 3D) and dc (D) -- no saved z.  beta = the
// LayerNorm's bias (z is rebuilt from it).
extern "C" int stage_cat3_bwd_dw(const float* dy, const unsigned* relu_mask, const float* W, const float* a, const float* b,
                                 const float* mean, const float* rstd, const float* gamma, const float* beta, float* da, float* db,
                                 float* dgamma, float* dbeta, float* dW, float* dc, long long rows, int D, int rep, int inner,
                                 float p_drop, unsigned long long seed, void* ws, size_t ws_bytes, void* stream) {
    if (!stage_cat3_bwd_dw_supported(rows, D, rep, inner)) return STAGE_ERR_SHAPE;
    if (ws_bytes < stage_cat3_bwd_dw_ws_bytes(rows, D, rep, inner)) return STAGE_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int mode = cw_mode(rep, inner);
    const long long groups = rows / ((long long)rep * inner);
    int CH = 1, fpc = 0;
    size_t slabs = 0;
    if (rep > 1) {
        cw_chunks(groups, rep, mode, &CH, &fpc);
        slabs = (size_t)groups * CH;
    }
    int grid = cw_grid();
    if (rep == 1) {
        const long tiles = (long)((rows + 31) / 32);
        if (tiles < grid) grid = (int)tiles;
    } else if ((long long)grid > groups * CH) grid = (int)(groups * CH);
    const CwWs w = cw_ws(ws, cw_grid(), slabs, inner);
    hipLaunchKernelGGL(cw_prep_w_kernel, dim3(6), dim3(1024), 0, st, W, w.img, w.w_up);
    const uint32_t th = drop_thresh16(p_drop);
    const float inv_keep = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    const bool drop = p_drop > 0.f;
    float* da_out = rep > 1 ? w.slabs : da;
    static bool attr_done = false;
    if (!attr_done) {
        cw_set_lds(cw_bwd_kernel<true, 0>); cw_set_lds(cw_bwd_kernel<true, 1>); cw_set_lds(cw_bwd_kernel<true, 2>);
        cw_set_lds(cw_bwd_kernel<false, 0>); cw_set_lds(cw_bwd_kernel<false, 1>); cw_set_lds(cw_bwd_kernel<false, 2>);
        attr_done = true;
    }
#define CW_LAUNCH(DR, MD)                                                                                                            \
    hipLaunchKernelGGL((cw_bwd_kernel<DR, MD>), dim3(grid), dim3(512), CW_LDS, st, dy, relu_mask, w.img, w.w_up, a, b, mean, rstd, gamma, \
                       beta, da_out, db, w.part, w.partW, w.partB, (long)rows, rep, inner, CH, fpc, (int)(groups * CH), (uint64_t)seed,  \
                       th, inv_keep, (const int4*)nullptr, 0l, 0l, (const int*)nullptr)
    if (drop) { if (mode == 0) CW_LAUNCH(true, 0); else if (mode == 1) CW_LAUNCH(true, 1); else CW_LAUNCH(true, 2); }
    else { if (mode == 0) CW_LAUNCH(false, 0); else if (mode == 1) CW_LAUNCH(false, 1); else CW_LAUNCH(false, 2); }
#undef CW_LAUNCH
    STAGE_LAUNCH_CHECK();
    const int rc = cw_finish(w, grid, dgamma, dbeta, dW, dc, st);
    if (rc) return rc;
    if (rep > 1) return stage_reduce_rep(da_out, da, groups, CH, (long long)inner * CW_D, st);
    return 0;
}

// ---- ragged token rows ------------------------------------------------------------------------------------------------
extern "C" int stage_cat3_bwd_dw_rag_supported(long long rows, long long fc_rows, int D, int groups, int max_frames, int Lqa) {
    if (!cw_enabled()) return 0;
    return (D == CW_D && rows >= 1 && rows * (long long)D * 4 < (1ll << 31) && fc_rows * (long long)D * 4 < (1ll << 31) && groups >= 1 &&
            max_frames >= 1 && Lqa >= 1 && Lqa <= 40) ? 1 : 0;
}
// persistent workgroups of the ragged launch = the `n_wg` the work table must be built for (tvqaplus_amd/ragged.py)
extern "C" int stage_cat3_bwd_dw_rag_work_groups(void) { return cw_grid(); }
extern "C" size_t stage_cat3_bwd_dw_rag_ws_bytes(int groups, int Lqa) {
    return cw_ws(nullptr, cw_grid(), (size_t)cw_grid() + groups, Lqa).bytes;
}
// As stage_cat3_dx_ln_bwd_rag, plus dW (D, 3D) and dc (D); wtab is REQUIRED and must be built for stage_cat3_bwd_dw_rag_work_groups().
extern "C" int stage_cat3_bwd_dw_rag(const float* dy, const unsigned* relu_mask, const float* W, const float* a, const float* b_fc,
                                     const float* mean, const float* rstd, const float* gamma, const float* beta, float* da,
                                     float* db_fc, float* dgamma, float* dbeta, float* dW, float* dc, const int* gdesc, const int* wtab,
                                     long long rows, long long fc_rows, int D, int groups, int max_frames, int Lqa, float p_drop,
                                     unsigned long long seed, void* ws, size_t ws_bytes, void* stream) {
    if (!stage_cat3_bwd_dw_rag_supported(rows, fc_rows, D, groups, max_frames, Lqa) || !wtab || !gdesc) return STAGE_ERR_SHAPE;
    if (ws_bytes < stage_cat3_bwd_dw_rag_ws_bytes(groups, Lqa)) return STAGE_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int grid = cw_grid();
    const CwWs w = cw_ws(ws, grid, (size_t)grid + groups, Lqa);
    hipLaunchKernelGGL(cw_prep_w_kernel, dim3(6), dim3(1024), 0, st, W, w.img, w.w_up);
    const uint32_t th = drop_thresh16(p_drop);
    const float inv_keep = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    static bool attr_done = false;
    if (!attr_done) {
        cw_set_lds(cw_bwd_kernel<true, 3>); cw_set_lds(cw_bwd_kernel<false, 3>);
        attr_done = true;
    }
#define CW_LAUNCH3(DR)                                                                                                               \
    hipLaunchKernelGGL((cw_bwd_kernel<DR, 3>), dim3(grid), dim3(512), CW_LDS, st, dy, relu_mask, w.img, w.w_up, a, b_fc, mean, rstd,    \
                       gamma, beta, w.slabs, db_fc, w.part, w.partW, w.partB, (long)rows, max_frames, Lqa, 1, 0, 0, (uint64_t)seed, th, \
                       inv_keep, (const int4*)gdesc, (long)fc_rows, (long)groups * Lqa, wtab)
    if (p_drop > 0.f) CW_LAUNCH3(true); else CW_LAUNCH3(false);
#undef CW_LAUNCH3
    STAGE_LAUNCH_CHECK();
    const int rc = cw_finish(w, grid, dgamma, dbeta, dW, dc, st);
    if (rc) return rc;
    const long inner4 = (long)Lqa * CW_D / 4;
    hipLaunchKernelGGL(cw_reduce_seg_kernel, dim3(stage_grid_for((long long)groups * inner4, 256, 4096)), dim3(256), 0, st, w.slabs, da,
                       reinterpret_cast<const int2*>(wtab + CW_WTAB_GSEG(grid)), (long)groups, inner4);
    STAGE_LAUNCH_CHECK();
    return 0;
}

#ifdef CW_PROF
extern "C" int stage_cw_prof(unsigned long long* out32) {
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(cw_prof_buf), 32 * sizeof(unsigned long long)) != hipSuccess) return -1;
    return 0;
}
#endif
