// Backward of  z = drop(LN_3D([a, b, a*b]));  y = ReLU(z W^T + c)   (model/stage.py:381-385 c2q_down_projection, :276-279 concat_fc)
// WITHOUT the 3D-wide gradient tensor: the input gradient of the Linear, dz = (dy .* relu') W  (rows x 3D = 1.47 GB at the full
// configuration), used to be written by the dX GEMM and read back by the LayerNorm backward.  Here one workgroup owns complete
// 384-wide rows: the product lands in MFMA accumulators, the LayerNorm backward runs on them in registers and only the D-wide
// gradients of a and b leave the compute unit:
//     HBM traffic per row (D = 128):  read dy 512 B (+ 16 B mask bits) + b 512 B (+ a 512 B when it is not broadcast)
//                                     write db 512 B (+ da 512 B)            instead of  + 1536 B written + 1536 B read
// Layout.  A workgroup (4 waves) walks tiles of 64 rows.  The dy tile is staged ONCE per workgroup: gated by the ReLU bit mask,
// scaled per row by a power of two and split into two fp16 planes (DESIGN.md finding 20) that sit in LDS in MFMA A-operand
// order.  Wave w owns output column tile w of each third (columns 32w..32w+31 of the a-, b- and a*b-part), so that
// da = dz_a + dz_ab * b and db = dz_b + dz_ab * a are lane-local; its weight fragments (3 tiles x 8 k-steps x 2 planes) come from
// a pre-split image in global memory (192 KB for all waves: L2 resident; 196 KB do not fit the LDS next to anything).
// Row statistics of the LayerNorm backward (sum g, sum g * x_hat over the 384 columns) are reduced inside the wave's 32 columns
// and exchanged between the four waves through LDS.  Broadcast `a` (rep > 1: shared by the frames of one (example, candidate)):
// tiles follow the FRAMES, so that a lane's accumulator slot (row r, column c) means the same position of `a` in every frame and
// the gradient of `a` accumulates in registers in frame order (deterministic; an LDS image with per-row position look-ups
// measured +500 us).  A tile is four passes of 8 rows, each pass with its own first row: inner <= 32: one (padded) tile per frame;
// inner == 40: per four frames four "main" tiles (rows 0..31 of a frame) and one "rest" tile whose pass p holds rows 32..39 of
// frame p -- the C/D layout of the 32x32 MFMA puts pass (r >> 2) into accumulator register r, so the rest tile's registers
// r, r + 4, r + 8, r + 12 are the same position of `a` in four frames.  No padded matrix work, 20 gradient registers per lane.
// Workgroup slabs [group][chunk of frames][inner][D] are summed in fixed order by stage_reduce_rep.
#include "common.h"
#include "../../include/stage_hip.h"

#ifndef CF_ABL
#define CF_ABL 0      // developer ablation bits (timing only, results wrong): 1 no weight-fragment loads / MFMAs, 2 no LayerNorm epilogue
#endif                // (statistics + output pass), 4 no staging of the dy tile, 8 output pass without the second a / b read
namespace {
constexpr int CF_D = 128;            // the kernel is specialised to hsz = 128 (3D = 384 columns = 12 MFMA tiles)
constexpr int CF_KS = CF_D / 16;     // k-steps of the 32x32x16 MFMA
constexpr int CF_WFRAG = 2 * 12 * CF_KS * 64;   // uint4 fragments of the pre-split weight image

// Weight image: Wimg[plane][column tile ct < 12][k-step ks < 8][lane] = the 8 fp16 of B-operand lane (col = 32 ct + (lane & 31),
// k = 16 ks + 8 (lane >> 5) + e) of W[k][col]  (W = the Linear's (D, 3D) weight: dz = dy_gated . W).  One workgroup.
__global__ __launch_bounds__(1024) void cf_prep_w_kernel(const float* __restrict__ W, uint4* __restrict__ img, int* __restrict__ w_up_out) {
    __shared__ float red[16];
    const int tid = threadIdx.x;
    float m = 0.f;
    for (int e = tid; e < CF_D * 3 * CF_D; e += 1024) m = fmaxf(m, fabsf(W[e]));
    m = wave_max(m);
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = red[0];
#pragma unroll
    for (int i = 1; i < 16; i++) m = fmaxf(m, red[i]);
    const int w_up = h_up_field((int)(__float_as_uint(m) >> 23) & 0xff);
    const float sc = __uint_as_float((unsigned)w_up << 23);
    if (tid == 0) w_up_out[0] = w_up;
    for (int f = tid; f < 12 * CF_KS * 64; f += 1024) {
        const int lane = f & 63, ks = (f >> 6) % CF_KS, ct = f / (64 * CF_KS);
        const int col = 32 * ct + (lane & 31), k0 = 16 * ks + 8 * (lane >> 5);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = W[(long)(k0 + e) * (3 * CF_D) + col];
        uint4 hi, lo;
        h_split2(v[0], v[1], sc, hi.x, lo.x);
        h_split2(v[2], v[3], sc, hi.y, lo.y);
        h_split2(v[4], v[5], sc, hi.z, lo.z);
        h_split2(v[6], v[7], sc, hi.w, lo.w);
        img[f] = hi;
        img[12 * CF_KS * 64 + f] = lo;
    }
}

// value of quad lane `sel` (0..3) in every lane of the quad (DPP quad_perm broadcast; `sel` folds to a constant after unrolling)
__device__ __forceinline__ unsigned cf_quad_bcast(unsigned v, int sel) {
    switch (sel) {
        case 0: return __builtin_amdgcn_update_dpp(0u, v, 0x00, 0xf, 0xf, true);
        case 1: return __builtin_amdgcn_update_dpp(0u, v, 0x55, 0xf, 0xf, true);
        case 2: return __builtin_amdgcn_update_dpp(0u, v, 0xAA, 0xf, 0xf, true);
        default: return __builtin_amdgcn_update_dpp(0u, v, 0xFF, 0xf, 0xf, true);
    }
}

// MODE 0: rep == 1, tiles of 32 consecutive rows, grid-stride.  MODE 1: rep > 1, inner <= 32: one tile per frame (rows past `inner`
// are padding).  MODE 2: rep > 1, inner == 40: per four frames, four main tiles + one rest tile (file comment).
template <bool DROP, int MODE>
__global__ __launch_bounds__(256, 2) void cf_bwd_kernel(const float* __restrict__ dy, const unsigned* __restrict__ rmask,
                                                        const uint4* __restrict__ wimg, const int* __restrict__ w_up_p,
                                                        const float* __restrict__ a, const float* __restrict__ b,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, float* __restrict__ da,
                                                        float* __restrict__ db, float* __restrict__ part, long M, int rep, int inner,
                                                        int CH, int frames_per_chunk, uint64_t seed, uint32_t th, float inv_keep) {
    constexpr bool REP = MODE > 0;
    // LDS: A planes [ks][plane][lane] uint4 (16 KB) | row scale exponent fields [32] | row mean / rstd [32][2] | partial row
    //      statistics [wave][32][2]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint4* Ap = reinterpret_cast<uint4*>(smem_raw);
    int* row_up = reinterpret_cast<int*>(smem_raw + CF_KS * 2 * 64 * 16);
    float* row_ms = reinterpret_cast<float*>(row_up + 32);
    float* st_part = row_ms + 2 * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int w_up = w_up_p[0];
    constexpr int K3 = 3 * CF_D;
    const float invK = 1.0f / (float)K3;
    // buffer addressing: descriptors in SGPRs, ONE 32-bit lane offset per access pattern, the row step as a scalar offset (64-bit
    // per-load addresses cost two VGPRs each: 64 operand loads per tile spilled hundreds of registers); rows past the end read 0
    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, (int)(M * CF_D * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_mk = __builtin_amdgcn_make_buffer_rsrc((void*)rmask, 0, (int)(M * (CF_D / 32) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)b, 0, (int)(M * CF_D * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, (int)((REP ? M / rep : M) * CF_D * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_mean = __builtin_amdgcn_make_buffer_rsrc((void*)mean, 0, (int)(M * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_rstd = __builtin_amdgcn_make_buffer_rsrc((void*)rstd, 0, (int)(M * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_db = __builtin_amdgcn_make_buffer_rsrc((void*)db, 0, (int)(M * CF_D * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_da = __builtin_amdgcn_make_buffer_rsrc((void*)da, 0, REP ? 0 : (int)(M * CF_D * 4), 0x00020000);
    // work of this workgroup.  REP: frames [f_beg, f_end) of group g; else tiles blockIdx.x, + gridDim.x, ...
    const long GR = REP ? (long)rep * inner : 0;
    int g = 0, f_beg = 0, f_end = 0;
    long n_tiles;
    if (REP) {
        g = blockIdx.x / CH;
        const int ch = blockIdx.x % CH;
        f_beg = ch * frames_per_chunk;
        f_end = min(rep, f_beg + frames_per_chunk);
        const int nf = max(f_end - f_beg, 0);
        n_tiles = MODE == 1 ? nf : 5l * ((nf + 3) / 4);
    } else {
        const long all = (M + 31) / 32;
        n_tiles = all > (long)blockIdx.x ? (all - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    }
    const int c = 32 * wave + l31;                        // this lane's column inside each third
    float gm[3], ag[3] = {0.f, 0.f, 0.f}, ab[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 3; t++) gm[t] = gamma[t * CF_D + c];
    // Per-lane bases; everything else inside the tile loop is a compile-time constant on top of them.  (Written as base + constant:
    // the row index `constant | 4 h` that the optimiser forms otherwise does not fold into the LDS offset field and every one of the
    // rows x arrays became its own loop-invariant address register -- 200+ spills.)
    const int h4 = 4 * h;
    const float* ms_h = row_ms + 2 * h4;
    const int* up_h = row_up + h4;
    float* stw_h = st_part + (wave * 32 + h4) * 2;
    const float* sta_h = st_part + h4 * 2;
    const int vo_row = (h4 * CF_D + c) * 4;               // byte offset of (row 4 h, column c) in a (rows, D) tensor
    // gradient of the broadcast operand: slot r of a main tile = position 8 (r >> 2) + (r & 3) + 4 h; the rest tile (MODE 2) adds
    // its registers r, r + 4, r + 8, r + 12 (four frames) into slot r & 3 = position 32 + (r & 3) + 4 h
    float dacc[16], dacc_rest[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; r++) dacc[r] = 0.f;

    for (long it = 0; it < n_tiles; it++) {
        // ---- the tile: four passes of 8 rows; pass p covers rows pb[p] .. pb[p] + nv[p] - 1 (uniform values) ----
        long pb[4];
        int nv[4];
        bool rest = false;                                // MODE 2: the tile that gathers rows 32..39 of four frames
        if (MODE == 0) {
            const long t0 = ((long)blockIdx.x + it * gridDim.x) * 32;
#pragma unroll
            for (int p2 = 0; p2 < 4; p2++) { pb[p2] = t0 + 8 * p2; nv[p2] = (int)max(0l, min(8l, M - pb[p2])); }
        } else if (MODE == 1) {
            const long t0 = (long)g * GR + (long)(f_beg + it) * inner;
#pragma unroll
            for (int p2 = 0; p2 < 4; p2++) { pb[p2] = t0 + 8 * p2; nv[p2] = max(0, min(8, inner - 8 * p2)); }
        } else {
            const int quad = (int)(it / 5), k = (int)(it - 5l * quad);
            rest = k == 4;
            const int f0 = f_beg + 4 * quad;
#pragma unroll
            for (int p2 = 0; p2 < 4; p2++) {
                const int f = rest ? f0 + p2 : f0 + k;
                pb[p2] = (long)g * GR + (long)f * inner + (rest ? 32 : 8 * p2);
                nv[p2] = f < f_end ? 8 : 0;
            }
        }
        // (with a loop-invariant `inner` the optimiser would hoist the 16 per-slot validity selects out of the tile loop and hold
        // them in registers -- which then spill; behind an empty asm the counts look loop-variant)
#pragma unroll
        for (int p2 = 0; p2 < 4; p2++) {
            nv[p2] = __builtin_amdgcn_readfirstlane(nv[p2]);
            asm volatile("" : "+s"(nv[p2]));
        }
        __syncthreads();                                  // the previous tile's LDS is free
        // ---- stage the dy tile: 32 lanes per row (float4 each), 8 rows per pass.  All loads of the tile are requested before the
        // first one is used (no load sits inside a branch) ----
        {
            // the staging indices are recomputed per tile (behind an empty asm they are not loop invariants: five registers that
            // would otherwise be held -- and spilled -- across the whole kernel)
            int tid_i = tid;
            asm volatile("" : "+v"(tid_i));
            const int sj = tid_i >> 5, sl = tid_i & 31;   // row inside a pass, float4 index inside the row
            const int vo_dy = (sj * CF_D + 4 * sl) * 4, vo_mk = (int)(((long)(sl >> 3) * M + sj) * 4);
            uint2* st_dst = reinterpret_cast<uint2*>(Ap + (sl >> 2) * 2 * 64 + sj + 32 * ((sl >> 1) & 1)) + (sl & 1);
            typedef unsigned cf_u4 __attribute__((ext_vector_type(4)));
            cf_u4 raw[4];
            unsigned wb[4];
            float mu_[4], rs_[4];
#pragma unroll
            for (int pass = 0; pass < 4; pass++) {
                raw[pass] = __builtin_amdgcn_raw_buffer_load_b128(rs_dy, vo_dy, (int)(pb[pass] * CF_D * 4), 0);
                wb[pass] = __builtin_amdgcn_raw_buffer_load_b32(rs_mk, vo_mk, (int)(pb[pass] * 4), 0);   // [word][row] mask
                mu_[pass] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_mean, sj * 4, (int)(pb[pass] * 4), 0));
                rs_[pass] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_rstd, sj * 4, (int)(pb[pass] * 4), 0));
            }
#pragma unroll
            for (int pass = 0; pass < ((CF_ABL & 4) ? 0 : 4); pass++) {
                const bool ok = sj < nv[pass];
                const unsigned wbits = wb[pass] >> (4 * (sl & 7));       // the bits of this lane's 4 columns
                float4 v = make_float4(__uint_as_float(raw[pass][0]), __uint_as_float(raw[pass][1]), __uint_as_float(raw[pass][2]),
                                       __uint_as_float(raw[pass][3]));
                v.x = (ok && (wbits & 1u)) ? v.x : 0.f;
                v.y = (ok && (wbits & 2u)) ? v.y : 0.f;
                v.z = (ok && (wbits & 4u)) ? v.z : 0.f;
                v.w = (ok && (wbits & 8u)) ? v.w : 0.f;
                float m = h_amax3(h_amax3(v.x, v.y, v.z), v.w, v.w);
                m = group_max(m, 32);
                const int up = h_up_field((int)(__float_as_uint(m) >> 23) & 0xff);
                const float sc = __uint_as_float((unsigned)up << 23);
                unsigned h01, l01, h23, l23;
                h_split2(v.x, v.y, sc, h01, l01);
                h_split2(v.z, v.w, sc, h23, l23);
                // element (row rl = 8 pass + sj, k = 4 sl + e): k-step sl >> 2, operand lane rl + 32 ((sl >> 1) & 1), position 4 (sl & 1) + e
                uint2* dst = st_dst + 8 * pass * 2;
                dst[0] = make_uint2(h01, h23);
                dst[2 * 64] = make_uint2(l01, l23);       // plane 1: + 64 uint4 = + 128 uint2
                if (sl == 0) {                            // LDS writes only: nothing is loaded inside this branch
                    row_up[pass * 8 + sj] = up;
                    row_ms[2 * (pass * 8 + sj)] = mu_[pass];
                    row_ms[2 * (pass * 8 + sj) + 1] = ok ? rs_[pass] : 0.f;
                }
            }
        }
        __syncthreads();
        // ---- operands and dropout bits of this lane's 16 (row, column) slots: requested / computed BEFORE the product, so that the
        // load latency hides behind the matrix phase and the hash temporaries are dead when the accumulators come alive.
        // Slot r = row pb[r >> 2] + (r & 3) + 4 h (the C/D layout of the 32x32 MFMA), column c of each third ----
        float av[16], bv[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int so = (int)(pb[r >> 2] * CF_D * 4) + (r & 3) * CF_D * 4;       // uniform
            if (!REP) av[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_a, vo_row, so, 0));
            bv[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_b, vo_row, so, 0));
        }
        // dropout stream of stage_cat3_layernorm_fwd: element row * 3D + t * D + c, one hash per 4 consecutive columns = the 4 lanes of a
        // quad: quad lane q hashes for the registers 4 j + q (row pb[j] + q + 4 h); bits[t][j] = its four keep bits
        unsigned kbits[4] = {0u, 0u, 0u, 0u};             // [j]: the three thirds' nibbles packed (bits 4 t .. 4 t + 3)
        if (DROP) {
            const unsigned drop_lane = (unsigned)(((l31 & 3) + h4) * (K3 / 4) + (c >> 2));
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint64_t base = (uint64_t)pb[j] * (uint64_t)(K3 / 4) + (uint64_t)drop_lane;
#pragma unroll
                for (int t = 0; t < 3; t++) kbits[j] |= drop4_bits(seed, base + (uint64_t)(t * (CF_D / 4)), th) << (4 * t);
                __builtin_amdgcn_sched_barrier(0);        // three hashes in flight, not twelve (64-bit temporaries)
            }
        }
        // ---- the product: acc[third] = dy tile (32 x 128) . W[:, columns 32 w .. 32 w + 31 of each third] ----
        f32x16 acc[3];
#pragma unroll
        for (int t = 0; t < 3; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
        // weight fragments: double-buffered by hand, two k-steps per trip of a loop the compiler must NOT unroll (fully unrolled, the
        // scheduler hoists all 48 fragment loads = 192 registers above the first MFMA and spills)
        auto load_b = [&](sf16x8 (&bf)[3][2], int ks) {
#pragma unroll
            for (int t = 0; t < 3; t++)
#pragma unroll
                for (int p2 = 0; p2 < 2; p2++)
                    bf[t][p2] = __builtin_bit_cast(sf16x8, wimg[(size_t)p2 * 12 * CF_KS * 64 + ((4 * t + wave) * CF_KS + ks) * 64 + lane]);
        };
        auto mul_b = [&](const sf16x8 (&bf)[3][2], int ks) {
            sf16x8 af[2];
#pragma unroll
            for (int p2 = 0; p2 < 2; p2++) af[p2] = __builtin_bit_cast(sf16x8, Ap[(ks * 2 + p2) * 64 + lane]);
#pragma unroll
            for (int t = 0; t < 3; t++) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1], bf[t][0], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0], bf[t][1], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0], bf[t][0], acc[t], 0, 0, 0);
            }
        };
        if (!REP) {
            sf16x8 bfa[3][2], bfb[3][2];
            load_b(bfa, 0);
#pragma unroll 1
            for (int ks = 0; ks < ((CF_ABL & 1) ? 0 : CF_KS); ks += 2) {
                load_b(bfb, ks + 1);
                mul_b(bfa, ks);
                load_b(bfa, ks + 2 < CF_KS ? ks + 2 : 0);  // (the last request is a harmless re-read of k-step 0)
                mul_b(bfb, ks + 1);
            }
        } else {
            // the broadcast variants carry 20 more persistent registers (the gradient of `a`): one fragment set in flight instead of
            // two keeps them free of spills; the second workgroup of the CU covers the exposed L2 latency
            sf16x8 bfa[3][2];
#pragma unroll 1
            for (int ks = 0; ks < ((CF_ABL & 1) ? 0 : CF_KS); ks++) {
                load_b(bfa, ks);
                mul_b(bfa, ks);
            }
        }
        // ---- LayerNorm backward on the accumulators.  C/D layout: column l31, row 8 (r >> 2) + (r & 3) + 4 h of the tile ----
        if (REP) {
            // the broadcast operand: the same (inner x D) block for every frame of the group -- cache resident, read after the matrix
            // phase (16 registers fewer across it: the gradient of `a` already occupies 20)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int pos0 = MODE == 1 ? 8 * (r >> 2) : (rest ? 32 : 8 * (r >> 2));   // uniform; + (r & 3) + 4 h
                av[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_a, vo_row, (g * inner + pos0 + (r & 3)) * CF_D * 4, 0));
            }
        }
        if (!(CF_ABL & 2)) {
            const int q = l31 & 3;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int rl = 8 * (r >> 2) + (r & 3);    // + 4 h
                // (1) true units (undo the two power-of-two scales); rows past the end of their pass contribute nothing
                float un = ((r & 3) + h4 < nv[r >> 2]) ? __builtin_ldexpf(1.0f, 254 - up_h[rl] - w_up) : 0.f;
                // (2) dropout: lane l reads bit (l & 3) of the word owned by quad lane (r & 3)
                if (DROP) {
                    const unsigned bw = cf_quad_bcast(kbits[r >> 2], r & 3) >> q;
                    un *= inv_keep;
#pragma unroll
                    for (int t = 0; t < 3; t++) acc[t][r] = ((bw >> (4 * t)) & 1u) ? acc[t][r] * un : 0.f;
                } else {
#pragma unroll
                    for (int t = 0; t < 3; t++) acc[t][r] *= un;
                }
                // (3) partial row sums over this wave's 32 columns (x 3 thirds) -> LDS [wave][row][2]; column partials of d gamma / d beta
                const float mu = ms_h[2 * rl], rs = ms_h[2 * rl + 1];
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int t = 0; t < 3; t++) {
                    const float x = t == 0 ? av[r] : (t == 1 ? bv[r] : av[r] * bv[r]);
                    const float xh = (x - mu) * rs;
                    const float dd = acc[t][r];
                    const float gq = dd * gm[t];
                    s1 += gq;
                    s2 += gq * xh;
                    ag[t] += dd * xh;
                    ab[t] += dd;
                }
                s1 = group_sum(s1, 32);
                s2 = group_sum(s2, 32);
                if (l31 == 0) {
                    stw_h[rl * 2] = s1;
                    stw_h[rl * 2 + 1] = s2;
                }
                if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // four rows in flight at a time: bounded register pressure
            }
        }
        __syncthreads();
        if (!(CF_ABL & 2)) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int rl = 8 * (r >> 2) + (r & 3);
                const bool ok = (r & 3) + h4 < nv[r >> 2];
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int w2 = 0; w2 < 4; w2++) {          // fixed order: identical totals in all four waves
                    s1 += sta_h[(w2 * 32 + rl) * 2];
                    s2 += sta_h[(w2 * 32 + rl) * 2 + 1];
                }
                s1 *= invK;
                s2 *= invK;
                const float mu = ms_h[2 * rl], rs = ms_h[2 * rl + 1];
                float dz[3];
#pragma unroll
                for (int t = 0; t < 3; t++) {
                    const float x = t == 0 ? av[r] : (t == 1 ? bv[r] : av[r] * bv[r]);
                    const float xh = (x - mu) * rs;
                    dz[t] = rs * (acc[t][r] * gm[t] - s1 - xh * s2);
                }
                // z = [a, b, a*b]:  da = dz0 + dz2 * b ; db = dz1 + dz2 * a
                const float da_v = dz[0] + dz[2] * bv[r], db_v = dz[1] + dz[2] * av[r];
                const int so = (int)(pb[r >> 2] * CF_D * 4) + (r & 3) * CF_D * 4;
                // straight-line stores: an invalid slot (a row of the next frame / past the end) gets an out-of-range lane offset
                // and is dropped by the descriptor's bounds check
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(db_v), rs_db, ok ? vo_row : 0x7ffffff0, so, 0);
                if (!REP) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(da_v), rs_da, ok ? vo_row : 0x7ffffff0, so, 0);
                else if (MODE == 2 && rest) dacc_rest[r & 3] += ok ? da_v : 0.f;
                else dacc[r] += ok ? da_v : 0.f;
                if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // ---- column partials of d gamma / d beta: the two lane halves hold different rows of the same columns ----
    float* prow = part + (size_t)blockIdx.x * 2 * K3;
#pragma unroll
    for (int t = 0; t < 3; t++) {
        const float sg = xsum32(ag[t], ag[t]), sb = xsum32(ab[t], ab[t]);
        if (h == 0) {
            prow[t * CF_D + c] = sg;
            prow[K3 + t * CF_D + c] = sb;
        }
    }
    if (REP) {                                            // da slab [G][CH][inner][D] of this (group, chunk of frames)
        float* dst = da + (size_t)blockIdx.x * inner * CF_D;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int pos = 8 * (r >> 2) + (r & 3) + h4;
            if (pos < min(inner, 32)) dst[pos * CF_D + c] = dacc[r];
        }
        if (MODE == 2) {
#pragma unroll
            for (int r = 0; r < 4; r++) dst[(32 + r + h4) * CF_D + c] = dacc_rest[r];
        }
    }
}

void cf_chunks(long long groups, int rep, int mode, int* CH, int* fpc) {
    // ~8 workgroups per resident slot (2 per CU): with about one workgroup per slot the last few run alone.  MODE 2 walks the
    // frames four at a time: chunks of a multiple of four frames
    int ch = (int)((4096 + groups - 1) / groups);
    if (ch > rep) ch = rep;
    if (ch < 1) ch = 1;
    int f = (rep + ch - 1) / ch;
    if (mode == 2) f = (f + 3) / 4 * 4;
    *fpc = f;
    *CH = (rep + f - 1) / f;
}
inline size_t cf_align(size_t v) { return (v + 255) & ~(size_t)255; }
inline int cf_mode(int rep, int inner) { return rep == 1 ? 0 : (inner <= 32 ? 1 : (inner == 40 ? 2 : -1)); }
}  // namespace

extern "C" int stage_cat3_dx_ln_bwd_supported(long long rows, int D, int rep, int inner) {
    if (getenv("STAGE_NO_CAT3_FUSED")) return 0;
    return (D == CF_D && rows >= 4096 && rows * (long long)D * 4 < (1ll << 31) && rep >= 1 && inner >= 1 && cf_mode(rep, inner) >= 0 &&
            rows % ((long long)rep * inner) == 0) ? 1 : 0;
}

extern "C" size_t stage_cat3_dx_ln_bwd_ws_bytes(long long rows, int D, int rep, int inner) {
    if (D != CF_D || rep < 1 || inner < 1 || cf_mode(rep, inner) < 0) return 0;
    size_t wg;
    if (rep > 1) {
        int CH, fpc;
        cf_chunks(rows / ((long long)rep * inner), rep, cf_mode(rep, inner), &CH, &fpc);
        wg = (size_t)(rows / ((long long)rep * inner)) * CH;
    } else {
        wg = 1024;
    }
    size_t b = cf_align((size_t)CF_WFRAG * sizeof(uint4)) + 256;                 // weight image + scale word
    b += cf_align(wg * 2 * 3 * CF_D * sizeof(float));                            // d gamma / d beta partials
    if (rep > 1) b += cf_align(wg * (size_t)inner * CF_D * sizeof(float));       // da slabs
    return b;
}

// dy (rows, D) = gradient of the Linear's output BEFORE the ReLU gate, relu_mask = the forward's ReLU bit mask ([D/32][rows]),
// W (D, 3D) the Linear's weight.  Outputs as stage_cat3_layernorm_bwd_reduced (rep > 1: da (rows / rep, D)) / stage_cat3_layernorm_bwd
// (rep == 1: da (rows, D)).  D == 128; rep > 1 takes inner <= 32 or inner == 40.
extern "C" int stage_cat3_dx_ln_bwd(const float* dy, const unsigned* relu_mask, const float* W, const float* a, const float* b,
                                    const float* mean, const float* rstd, const float* gamma, float* da, float* db, float* dgamma,
                                    float* dbeta, long long rows, int D, int rep, int inner, float p_drop, unsigned long long seed,
                                    void* ws, size_t ws_bytes, void* stream) {
    if (!stage_cat3_dx_ln_bwd_supported(rows, D, rep, inner)) return STAGE_ERR_SHAPE;
    if (ws_bytes < stage_cat3_dx_ln_bwd_ws_bytes(rows, D, rep, inner)) return STAGE_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char* wsp = (char*)ws;
    uint4* img = (uint4*)wsp;
    int* w_up = (int*)(wsp + cf_align((size_t)CF_WFRAG * sizeof(uint4)));
    wsp += cf_align((size_t)CF_WFRAG * sizeof(uint4)) + 256;
    hipLaunchKernelGGL(cf_prep_w_kernel, dim3(1), dim3(1024), 0, st, W, img, w_up);
    const uint32_t th = drop_thresh16(p_drop);
    const float inv_keep = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    const bool drop = p_drop > 0.f;
    const int K3 = 3 * CF_D;
    float* part = (float*)wsp;
    int grid, CH = 1, fpc = 0;
    float* da_out = da;
    const int mode = cf_mode(rep, inner);
    const size_t lds = (size_t)CF_KS * 2 * 64 * 16 + (size_t)32 * (4 + 8 + 32);
    const long long groups = rows / ((long long)rep * inner);
    if (rep > 1) {
        cf_chunks(groups, rep, mode, &CH, &fpc);
        grid = (int)(groups * CH);
        wsp += cf_align((size_t)grid * 2 * K3 * sizeof(float));
        da_out = (float*)wsp;                                  // slabs [G][CH][inner][D]
    } else {
        const long tiles = (long)((rows + 31) / 32);
        grid = (int)(tiles < 1024 ? tiles : 1024);
    }
#define CF_LAUNCH(DR, MD)                                                                                                          \
    hipLaunchKernelGGL((cf_bwd_kernel<DR, MD>), dim3(grid), dim3(256), lds, st, dy, relu_mask, img, w_up, a, b, mean, rstd, gamma,    \
                       da_out, db, part, (long)rows, rep, inner, CH, fpc, (uint64_t)seed, th, inv_keep)
    if (drop) { if (mode == 0) CF_LAUNCH(true, 0); else if (mode == 1) CF_LAUNCH(true, 1); else CF_LAUNCH(true, 2); }
    else { if (mode == 0) CF_LAUNCH(false, 0); else if (mode == 1) CF_LAUNCH(false, 1); else CF_LAUNCH(false, 2); }
#undef CF_LAUNCH
    STAGE_LAUNCH_CHECK();
    // d gamma / d beta: ordered sum of the workgroup partials; da: ordered sum of the chunk slabs of every group
    stage_colreduce2(part, dgamma, 2 * K3, K3, part + K3, dbeta, 2 * K3, K3, grid, st);
    STAGE_LAUNCH_CHECK();
    if (rep > 1) return stage_reduce_rep(da_out, da, groups, CH, (long long)inner * CF_D, st);
    return 0;
}
