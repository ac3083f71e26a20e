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

#ifndef CFF_ABL
#define CFF_ABL 0     // the same for cff_fwd_kernel: 1 no weight-fragment loads / MFMAs, 2 no dropout hashes, 4 no y stores
#endif
#ifndef CF_ABL
#define CF_ABL 0      // developer ablation bits (timing only, results wrong): 1 no weight-fragment loads / MFMAs, 2 no LayerNorm epilogue
#endif                // (statistics + output pass), 4 no staging of the dy tile, 8 output pass without the second a / b read
namespace {
constexpr int CF_D = 128;            // the kernel is specialised to hsz = 128 (3D = 384 columns = 12 MFMA tiles)
constexpr int CF_KS = CF_D / 16;     // k-steps of the 32x32x16 MFMA
constexpr int CF_WFRAG = 2 * 12 * CF_KS * 64;   // uint4 fragments of the pre-split weight image
constexpr int CF_RAG_WGS = 512;                 // persistent workgroups of the balanced ragged launch (2 per CU on 256 CUs)
// work table (int32, tvqaplus_amd/ragged.py: RaggedTables.work_table): [first segment of workgroup w: W + 1 entries, padded to a multiple
// of 4] [(first slab, slabs) of every group: 2 G] [segments (group, first frame, end frame, workgroup): 4 each]
#define CF_WTAB_GSEG(W) ((((W) + 1) + 3) & ~3)
#define CF_WTAB_SEG(W, G) (CF_WTAB_GSEG(W) + ((2 * (G) + 3) & ~3))

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
    if (tid == 0 && blockIdx.x == 0) w_up_out[0] = w_up;
    // (every workgroup finds the largest magnitude itself -- 192 KB from L2 -- and splits its share of the fragments: one workgroup
    // for the whole image took 19 us)
    for (int f = blockIdx.x * 1024 + tid; f < 12 * CF_KS * 64; f += gridDim.x * 1024) {
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
// MODE 3: RAGGED token rows (include/stage_hip.h): group g = (example, candidate) keeps gdesc[g] = (first compact row, live words
// Lc <= 40, frame slots of its example = live frames + 1, first frame-compact sequence); its rows are [live frame][word < Lc]; dy /
// mask / statistics are compact rows, b and db rows of the FRAME-COMPACT tensor (sequence * Lqa + word; db may alias b: a row is read
// before it is written, by the same lane), `a` rows g * Lqa + word.  Lc <= 32 walks like MODE 1, longer groups like MODE 2 with
// Lc - 32 rows in the rest tile; `inner` = Lqa (slab pitch), `rep` = the largest frame count (chunking).
template <bool DROP, int MODE>
__global__ __launch_bounds__(256, 2) void cf_bwd_kernel(const float* __restrict__ dy, const unsigned* __restrict__ rmask,
                                                        const uint4* __restrict__ wimg, const int* __restrict__ w_up_p,
                                                        const float* __restrict__ a, const float* __restrict__ b,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, float* __restrict__ da,
                                                        float* __restrict__ db, float* __restrict__ part, long M, int rep, int inner,
                                                        int CH, int frames_per_chunk, uint64_t seed, uint32_t th, float inv_keep,
                                                        const int4* __restrict__ gdesc, long b_rows, long a_rows,
                                                        const int* __restrict__ wtab = nullptr) {
    constexpr bool REP = MODE > 0;
    constexpr bool RAG = MODE == 3;
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
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)b, 0, (int)((RAG ? b_rows : M) * CF_D * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, (int)((RAG ? a_rows : (REP ? M / rep : M)) * CF_D * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_mean = __builtin_amdgcn_make_buffer_rsrc((void*)mean, 0, (int)(M * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_rstd = __builtin_amdgcn_make_buffer_rsrc((void*)rstd, 0, (int)(M * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_db = __builtin_amdgcn_make_buffer_rsrc((void*)db, 0, (int)((RAG ? b_rows : M) * CF_D * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_da = __builtin_amdgcn_make_buffer_rsrc((void*)da, 0, REP ? 0 : (int)(M * CF_D * 4), 0x00020000);
    // work of this workgroup.  REP: frames [f_beg, f_end) of group g; else tiles blockIdx.x, + gridDim.x, ...
    // RAG with a work table (wtab, tvqaplus_amd/ragged.py: RaggedTables.work_table): the workgroup is one of CF_RAG_WGS persistent
    // ones and walks its SEGMENTS seg[s_beg .. s_end) = (group, first frame, end frame, .) -- equal tile counts per workgroup whatever
    // the groups' live frames and word counts are (the (group, chunk) grid left 23 % of the kernel to the last round of workgroups)
    const long GR = REP ? (long)rep * inner : 0;
    int g = 0, f_beg = 0, f_end = 0;
    long n_tiles = 0;
    long rg_row0 = 0, rg_seq0 = 0;                        // RAG: first compact row / first frame-compact sequence of the group
    int rg_lc = 0;                                        //      live words per frame
    bool rg_big = false;
    const bool persistent = RAG && wtab != nullptr;
    int s_beg = 0, s_end = 1;
    const int4* segs = nullptr;
    if (persistent) {
        s_beg = __builtin_amdgcn_readfirstlane(wtab[blockIdx.x]);
        s_end = __builtin_amdgcn_readfirstlane(wtab[blockIdx.x + 1]);
        segs = reinterpret_cast<const int4*>(wtab + CF_WTAB_SEG(gridDim.x, (int)(a_rows / inner)));
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

    // One loop over the tiles of all segments (a nested segment / tile loop made the allocator spill 39 registers): the segment
    // switch -- slab of the finished segment out, descriptors of the next one in -- is a uniform branch at the head of an iteration,
    // where only the persistent accumulators are live
    auto write_slab = [&](int slab) {                     // da slab [inner][D] of a (group, chunk of frames) / segment
        float* dst = da + (size_t)slab * inner * CF_D;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int pos = 8 * (r >> 2) + (r & 3) + h4;
            if (pos < min(inner, 32)) dst[pos * CF_D + c] = dacc[r];
        }
        if (MODE == 2) {
#pragma unroll
            for (int r = 0; r < 4; r++) dst[(32 + r + h4) * CF_D + c] = dacc_rest[r];
        }
        if (RAG) {                                        // every one of the Lqa (<= 40) positions is written: zeros past the live words
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (32 + r + h4 < inner) dst[(32 + r + h4) * CF_D + c] = dacc_rest[r];
        }
    };
    int sg = s_beg - 1;                                   // current segment (persistent) -- s_beg - 1: none yet
    long it = 0;
    for (;;) {
        if (it >= n_tiles) {                              // (uniform) next segment
            if (REP && sg >= s_beg) {
                write_slab(persistent ? sg : (int)blockIdx.x);
#pragma unroll
                for (int r = 0; r < 16; r++) dacc[r] = 0.f;
#pragma unroll
                for (int r = 0; r < 4; r++) dacc_rest[r] = 0.f;
            }
            if (++sg >= s_end) break;
            it = 0;
            if (REP) {
                int frames = rep;
                if (persistent) {
                    const int4 sd = segs[sg];
                    g = __builtin_amdgcn_readfirstlane(sd.x);
                    f_beg = __builtin_amdgcn_readfirstlane(sd.y);
                    f_end = __builtin_amdgcn_readfirstlane(sd.z);
                } else {
                    g = blockIdx.x / CH;
                    f_beg = (blockIdx.x % CH) * frames_per_chunk;
                }
                if (RAG) {
                    const int4 gd = gdesc[g];
                    rg_row0 = __builtin_amdgcn_readfirstlane(gd.x);
                    rg_lc = __builtin_amdgcn_readfirstlane(gd.y);
                    frames = rg_lc > 0 ? __builtin_amdgcn_readfirstlane(gd.z) - 1 : 0;
                    rg_seq0 = __builtin_amdgcn_readfirstlane(gd.w);
                    rg_big = rg_lc > 32;
                }
                if (!persistent) f_end = min(frames, f_beg + frames_per_chunk);
                const int nf = max(f_end - f_beg, 0);
                n_tiles = (MODE == 1 || (RAG && !rg_big)) ? nf : 5l * ((nf + 3) / 4);
            } else {
                const long all = (M + 31) / 32;
                n_tiles = all > (long)blockIdx.x ? (all - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
            }
            continue;
        }
        // ---- the tile: four passes of 8 rows; pass p covers rows pb[p] .. pb[p] + nv[p] - 1 (uniform values) ----
        long pb[4];
        long pbB[4];                                      // the same passes in the rows of b / db (RAG: frame-compact rows)
        int nv[4];
        bool rest = false;                                // MODE 2: the tile that gathers rows 32..39 of four frames
        if (RAG) {
            if (!rg_big) {
                const long f = f_beg + it;
#pragma unroll
                for (int p2 = 0; p2 < 4; p2++) {
                    pb[p2] = rg_row0 + f * rg_lc + 8 * p2;
                    pbB[p2] = (rg_seq0 + f) * inner + 8 * p2;
                    nv[p2] = max(0, min(8, rg_lc - 8 * p2));
                }
            } else {
                const int quad = (int)(it / 5), k = (int)(it - 5l * quad);
                rest = k == 4;
                const int f0 = f_beg + 4 * quad;
#pragma unroll
                for (int p2 = 0; p2 < 4; p2++) {
                    const int f = rest ? f0 + p2 : f0 + k;
                    pb[p2] = rg_row0 + (long)f * rg_lc + (rest ? 32 : 8 * p2);
                    pbB[p2] = (rg_seq0 + f) * inner + (rest ? 32 : 8 * p2);
                    nv[p2] = f < f_end ? (rest ? rg_lc - 32 : 8) : 0;
                }
            }
        } else if (MODE == 0) {
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
        if (!RAG) {
#pragma unroll
            for (int p2 = 0; p2 < 4; p2++) pbB[p2] = pb[p2];
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
            // operand slot inside its [ks][plane] block: (rl + 32 kh) ^ 2 ks ^ kh (see cff_fwd_kernel: spreads the 16 (k-step, k-half)
            // pairs a wave stores at once over all banks)
            uint2* st_e = reinterpret_cast<uint2*>(Ap + (sl >> 2) * 2 * 64 + ((sj + 32 * ((sl >> 1) & 1)) ^ (2 * (sl >> 2)) ^ ((sl >> 1) & 1))) + (sl & 1);
            uint2* st_o = reinterpret_cast<uint2*>(Ap + (sl >> 2) * 2 * 64 + ((sj + 8 + 32 * ((sl >> 1) & 1)) ^ (2 * (sl >> 2)) ^ ((sl >> 1) & 1))) + (sl & 1);
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
                uint2* dst = ((pass & 1) ? st_o : st_e) + 16 * (pass >> 1) * 2;
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
            const int so = (int)(pbB[r >> 2] * CF_D * 4) + (r & 3) * CF_D * 4;       // uniform
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
            for (int p2 = 0; p2 < 2; p2++) af[p2] = __builtin_bit_cast(sf16x8, Ap[(ks * 2 + p2) * 64 + (lane ^ (2 * ks) ^ h)]);
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
                const int pos0 = MODE == 1 ? 8 * (r >> 2) : (rest ? 32 : 8 * (r >> 2));   // uniform; + (r & 3) + 4 h  (RAG: rest is false for short groups)
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
                const int so = (int)(pbB[r >> 2] * CF_D * 4) + (r & 3) * CF_D * 4;
                // straight-line stores: an invalid slot (a row of the next frame / past the end) gets an out-of-range lane offset
                // and is dropped by the descriptor's bounds check
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(db_v), rs_db, ok ? vo_row : 0x7ffffff0, so, 0);
                if (!REP) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(da_v), rs_da, ok ? vo_row : 0x7ffffff0, so, 0);
                else if (MODE >= 2 && rest) dacc_rest[r & 3] += ok ? da_v : 0.f;
                else dacc[r] += ok ? da_v : 0.f;
                if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        it++;
    }
    // ---- column partials of d gamma / d beta: the two lane halves hold different rows of the same columns ----
    float* prow = part + (size_t)blockIdx.x * 2 * K3;
#pragma unroll
    for (int t = 0; t < 3; t++) {
        const float sg2 = xsum32(ag[t], ag[t]), sb = xsum32(ab[t], ab[t]);
        if (h == 0) {
            prow[t * CF_D + c] = sg2;
            prow[K3 + t * CF_D + c] = sb;
        }
    }
}

// out[g][e] = sum of the slabs gseg[g] = (first, count) in order (zeros for a group without a slab); e < inner_elems (multiple of 4)
__global__ __launch_bounds__(256) void cf_reduce_seg_kernel(const float* __restrict__ in, float* __restrict__ out, const int2* __restrict__ gseg,
                                                            long groups, long inner4) {
    const long total = groups * inner4;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long g = e / inner4, q = e % inner4;
        const int2 gs = gseg[g];
        const float* p = in + ((long)gs.x * inner4 + q) * 4;
        float4 acc = f4zero();
        for (int r = 0; r < gs.y; r++) acc = f4add(acc, ld4(p + (long)r * inner4 * 4));
        st4(out + e * 4, acc);
    }
}

void cf_chunks(long long groups, int rep, int mode, int* CH, int* fpc) {
    // ~4 workgroups per resident slot (2 per CU): with about one workgroup per slot the last few run alone.  MODE 2 walks the
    // frames four at a time: chunks of a multiple of four frames
    static const long wg_target = getenv("STAGE_CF_WGS") ? atol(getenv("STAGE_CF_WGS")) : 2048;       // (developer sweep: 1536-2560 measured 1-2 % faster than 4096, 1024 10 % slower)
    int ch = (int)((wg_target + groups - 1) / groups);
    if (ch > rep) ch = rep;
    if (ch < 1) ch = 1;
    int f = (rep + ch - 1) / ch;
    if (mode == 2) f = (f + 3) / 4 * 4;
    *fpc = f;
    *CH = (rep + f - 1) / f;
}
inline size_t cf_align(size_t v) { return (v + 255) & ~(size_t)255; }
inline long cf_flat_grid() {                          // MODE 0: workgroups that stride the 32-row tiles (developer sweep: STAGE_CF_FLAT_GRID)
    static const long g = getenv("STAGE_CF_FLAT_GRID") ? atol(getenv("STAGE_CF_FLAT_GRID")) : 1024;
    return g;
}
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
        wg = (size_t)cf_flat_grid();
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
    hipLaunchKernelGGL(cf_prep_w_kernel, dim3(6), dim3(1024), 0, st, W, img, w_up);
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
        grid = (int)(tiles < cf_flat_grid() ? tiles : cf_flat_grid());
    }
#define CF_LAUNCH(DR, MD)                                                                                                          \
    hipLaunchKernelGGL((cf_bwd_kernel<DR, MD>), dim3(grid), dim3(256), lds, st, dy, relu_mask, img, w_up, a, b, mean, rstd, gamma,    \
                       da_out, db, part, (long)rows, rep, inner, CH, fpc, (uint64_t)seed, th, inv_keep, (const int4*)nullptr, 0l, 0l)
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

// ---- ragged token rows (MODE 3) -------------------------------------------------------------------------------------
// persistent workgroups of the balanced ragged launch: what the work table is built for.  With the dW-inside backward enabled
// (cat3_bwd_dw.hip: one workgroup per compute unit) both kernels take ITS count, so that one table serves either.
static int cf_rag_wgs() {
    return stage_cat3_bwd_dw_rag_supported(1, 1, CF_D, 1, 1, 1) ? stage_cat3_bwd_dw_rag_work_groups() : CF_RAG_WGS;
}
extern "C" int stage_cat3_dx_ln_bwd_rag_supported(long long rows, long long fc_rows, int D, int groups, int max_frames, int Lqa) {
    if (getenv("STAGE_NO_CAT3_FUSED")) return 0;
    return (D == CF_D && rows >= 1 && rows * (long long)D * 4 < (1ll << 31) && fc_rows * (long long)D * 4 < (1ll << 31) && groups >= 1 &&
            max_frames >= 1 && Lqa >= 1 && Lqa <= 40) ? 1 : 0;
}
extern "C" size_t stage_cat3_dx_ln_bwd_rag_ws_bytes(int groups, int max_frames, int Lqa) {
    int CH, fpc;
    cf_chunks(groups, max_frames, 2, &CH, &fpc);
    size_t wg = (size_t)groups * CH;
    if (wg < (size_t)cf_rag_wgs() + groups) wg = (size_t)cf_rag_wgs() + groups;  // the balanced launch: <= work groups + groups slabs
    size_t b = cf_align((size_t)CF_WFRAG * sizeof(uint4)) + 256;
    b += cf_align(wg * 2 * 3 * CF_D * sizeof(float));
    b += cf_align(wg * (size_t)Lqa * CF_D * sizeof(float));
    return b;
}
extern "C" int stage_cat3_rag_work_groups(void) { return cf_rag_wgs(); }
// As stage_cat3_dx_ln_bwd with a broadcast `a`, on ragged token rows: dy / relu_mask / mean / rstd are compact (rows), b_fc and db_fc
// rows of the frame-compact tensor (fc_rows; db_fc may be b_fc itself), a and da (groups, Lqa, D); gdesc: see cf_bwd_kernel MODE 3.
// wtab (may be NULL: one workgroup per (group, chunk of max_frames)): the balanced work table for stage_cat3_rag_work_groups()
// workgroups (layout: CF_WTAB_* above; built by tvqaplus_amd/ragged.py from the same lengths as gdesc).
extern "C" int stage_cat3_dx_ln_bwd_rag(const float* dy, const unsigned* relu_mask, const float* W, const float* a, const float* b_fc,
                                        const float* mean, const float* rstd, const float* gamma, float* da, float* db_fc,
                                        float* dgamma, float* dbeta, const int* gdesc, const int* wtab, long long rows, long long fc_rows,
                                        int D, int groups, int max_frames, int Lqa, float p_drop, unsigned long long seed, void* ws,
                                        size_t ws_bytes, void* stream) {
    if (!stage_cat3_dx_ln_bwd_rag_supported(rows, fc_rows, D, groups, max_frames, Lqa)) return STAGE_ERR_SHAPE;
    if (ws_bytes < stage_cat3_dx_ln_bwd_rag_ws_bytes(groups, max_frames, Lqa)) return STAGE_ERR_WORKSPACE;
    static const bool no_balance = getenv("STAGE_CF_NO_BALANCE") != nullptr;     // developer switch: the (group, chunk) grid
    if (no_balance) wtab = nullptr;
    hipStream_t st = (hipStream_t)stream;
    char* wsp = (char*)ws;
    uint4* img = (uint4*)wsp;
    int* w_up = (int*)(wsp + cf_align((size_t)CF_WFRAG * sizeof(uint4)));
    wsp += cf_align((size_t)CF_WFRAG * sizeof(uint4)) + 256;
    hipLaunchKernelGGL(cf_prep_w_kernel, dim3(6), dim3(1024), 0, st, W, img, w_up);
    const uint32_t th = drop_thresh16(p_drop);
    const float inv_keep = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    const int K3 = 3 * CF_D;
    int CH, fpc;
    cf_chunks(groups, max_frames, 2, &CH, &fpc);
    const int grid = wtab ? cf_rag_wgs() : groups * CH;
    const size_t slabs = wtab ? (size_t)cf_rag_wgs() + groups : (size_t)grid;
    float* part = (float*)wsp;
    wsp += cf_align(slabs * 2 * K3 * sizeof(float));
    float* da_out = (float*)wsp;                               // slabs [G][CH][Lqa][D] / one per segment
    const size_t lds = (size_t)CF_KS * 2 * 64 * 16 + (size_t)32 * (4 + 8 + 32);
#define CF_LAUNCH3(DR)                                                                                                                \
    hipLaunchKernelGGL((cf_bwd_kernel<DR, 3>), dim3(grid), dim3(256), lds, st, dy, relu_mask, img, w_up, a, b_fc, mean, rstd, gamma,    \
                       da_out, db_fc, part, (long)rows, max_frames, Lqa, CH, fpc, (uint64_t)seed, th, inv_keep, (const int4*)gdesc,    \
                       (long)fc_rows, (long)groups * Lqa, wtab)
    if (p_drop > 0.f) CF_LAUNCH3(true); else CF_LAUNCH3(false);
#undef CF_LAUNCH3
    STAGE_LAUNCH_CHECK();
    stage_colreduce2(part, dgamma, 2 * K3, K3, part + K3, dbeta, 2 * K3, K3, grid, st);
    STAGE_LAUNCH_CHECK();
    if (!wtab) return stage_reduce_rep(da_out, da, groups, CH, (long long)Lqa * CF_D, st);
    const long inner4 = (long)Lqa * CF_D / 4;
    hipLaunchKernelGGL(cf_reduce_seg_kernel, dim3(stage_grid_for((long long)groups * inner4, 256, 4096)), dim3(256), 0, st, da_out, da,
                       reinterpret_cast<const int2*>(wtab + CF_WTAB_GSEG(cf_rag_wgs())), (long)groups, inner4);
    STAGE_LAUNCH_CHECK();
    return 0;
}

// =====================================================================================================================
// Forward twin:  z = drop(LN_3D([a, b, a*b])) ;  y = ReLU(z W^T + c)  in ONE pass over a and b.
// The two kernels it replaces (stage_cat3_layernorm_fwd -> 1.47 GB tensor -> stage_gemm_nt_mask) write z and read it back; here the
// normalised tile is built in registers, written once (the weight-gradient GEMM of the backward still contracts over it) and
// handed to the matrix cores through LDS: HBM traffic per row 512 B (b) [+ 512 B a] read, 1536 B (z) + 512 B (y) written, instead
// of + 1536 B read.  Same arithmetic and summation order as ln_fwd_fast_kernel<1> (rowops.hip) for the LayerNorm -- z, mean and
// rstd are bit-identical -- and the two-way fp16 split of gemm_nt_stream_kernel for the product (one power-of-two scale per row of
// z, one for the weight).  A workgroup (4 waves) walks 32-row tiles; wave w owns output columns 32 w .. 32 w + 31; its weight
// fragments (24 k-steps x 2 planes, 48 KB) stream from a pre-split image in global memory (L2 resident).
// =====================================================================================================================
#ifndef CFF_NT
#define CFF_NT 0      // cache policy of the z stores (2 = non-temporal)
#endif
namespace {
constexpr int CFF_KS = 3 * CF_D / 16;             // 24 k-steps of the 32x32x16 MFMA
constexpr int CFF_WFRAG = 2 * 4 * CFF_KS * 64;    // uint4 fragments of the forward weight image

// Wimg[plane][wave w][k-step ks][lane] = the 8 fp16 of B-operand lane (col n = 32 w + (lane & 31), k = 16 ks + 8 (lane >> 5) + e) of
// W[n][k]  (W = the Linear's (D, 3D) weight)
__global__ __launch_bounds__(1024) void cff_prep_w_kernel(const float* __restrict__ W, uint4* __restrict__ img, int* __restrict__ w_up_out) {
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
    if (tid == 0 && blockIdx.x == 0) w_up_out[0] = w_up;
    for (int f = blockIdx.x * 1024 + tid; f < 4 * CFF_KS * 64; f += gridDim.x * 1024) {
        const int lane = f & 63, ks = (f >> 6) % CFF_KS, w = f / (64 * CFF_KS);
        const float* src = W + (long)(32 * w + (lane & 31)) * (3 * CF_D) + 16 * ks + 8 * (lane >> 5);
        const float4 v0 = ld4(src), v1 = ld4(src + 4);
        uint4 hi, lo;
        h_split2(v0.x, v0.y, sc, hi.x, lo.x);
        h_split2(v0.z, v0.w, sc, hi.y, lo.y);
        h_split2(v1.x, v1.y, sc, hi.z, lo.z);
        h_split2(v1.z, v1.w, sc, hi.w, lo.w);
        img[f] = hi;
        img[4 * CFF_KS * 64 + f] = lo;
    }
}

// RAG: ragged token rows (include/stage_hip.h): row r of the compact output takes a[rowinfo[r].x] and b[rowinfo[r].y] (the QA row of
// its word / its row in the frame-compact attention output); a has a_rows rows, b has b_rows.
template <bool DROP, bool RAG = false>
__global__ __launch_bounds__(256, 3) void cff_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const uint4* __restrict__ wimg, const int* __restrict__ w_up_p,
                                                         const float* __restrict__ bias, float* __restrict__ z,
                                                         float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ y,
                                                         unsigned* __restrict__ mask_out, long M, int rep, int inner, float eps,
                                                         uint64_t seed, uint32_t th, float inv_keep,
                                                         const int4* __restrict__ rowinfo = nullptr, long a_rows = 0, long b_rows = 0) {
    // LDS: A planes [ks][plane][lane] uint4 (48 KB) | row scale exponent fields [32]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint4* Ap = reinterpret_cast<uint4*>(smem_raw);
    int* row_up = reinterpret_cast<int*>(smem_raw + CFF_KS * 2 * 64 * 16);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int w_up = w_up_p[0];
    constexpr int K3 = 3 * CF_D, K4 = K3 / 4, D4 = CF_D / 4;
    const float invK = 1.0f / (float)K3;
    const long GR = (long)rep * inner;
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)b, 0, (int)((RAG ? b_rows : M) * CF_D * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, (int)((RAG ? a_rows : M / rep) * CF_D * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc((void*)z, 0, (int)(M * K3 * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, (int)(M * CF_D * 4), 0x00020000);
    const int sj = tid >> 5, sl = tid & 31;               // staging: row inside a pass, float4 index inside each third
    float4 gm[3], bt[3];
#pragma unroll
    for (int t = 0; t < 3; t++) {
        gm[t] = ld4(gamma + 4 * (t * D4 + sl));
        bt[t] = ld4(beta + 4 * (t * D4 + sl));
    }
    const int n = 32 * wave + l31;                        // this lane's output column
    const float bsv = bias ? bias[n] : 0.f;
    // Operand slot of (row rl, k-step ks, k-half kh) inside its [ks][plane] block of 64: (rl + 32 kh) ^ 2 (ks & 7) ^ kh.  Unswizzled, the
    // 16 (k-step, k-half) pairs a wave stores at once sit 512 B apart = on the same banks (16-way conflict: 68 % of the kernel's LDS
    // cycles -- though not of its run time: the stores are off the critical path); the XOR spreads them over all 64 banks and keeps
    // the matrix phase's 1 KB block reads conflict free (a constant XOR permutes the slots of a 16-lane group among themselves).
    // Even / odd passes differ in bit 3 of the row: two pointers.
    uint2* st_e = reinterpret_cast<uint2*>(Ap + (sl >> 2) * 2 * 64 + ((sj + 32 * ((sl >> 1) & 1)) ^ (2 * (sl >> 2)) ^ ((sl >> 1) & 1))) + (sl & 1);
    uint2* st_o = reinterpret_cast<uint2*>(Ap + (sl >> 2) * 2 * 64 + ((sj + 8 + 32 * ((sl >> 1) & 1)) ^ (2 * (sl >> 2)) ^ ((sl >> 1) & 1))) + (sl & 1);
    const long n_tiles = (M + 31) / 32;

    for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long t0 = tile * 32;
        __syncthreads();                                  // the previous tile's A planes are free
        // ---- normalise 32 rows: 32 lanes per row (one float4 of each third), 8 rows per pass; loads of the tile first ----
        typedef unsigned cf_u4 __attribute__((ext_vector_type(4)));
        cf_u4 ra[4], rb[4];
        // broadcast operand: row -> (row / (rep * inner)) * inner + row % inner.  Uniform part once per tile, per row an add and
        // at most three conditional subtractions (inner >= 11)
        const long g0 = rep > 1 ? t0 / GR : 0;
        const int r0 = rep > 1 ? (int)(t0 - g0 * GR) : 0, p0 = rep > 1 ? r0 % inner : 0;
        if (RAG) {
            int2 ri[4];
#pragma unroll
            for (int pass = 0; pass < 4; pass++) {
                const long rc = min(t0 + 8 * pass + sj, M - 1);          // rows past the end: any valid pair (their results are dropped)
                ri[pass] = *reinterpret_cast<const int2*>(rowinfo + rc);
            }
#pragma unroll
            for (int pass = 0; pass < 4; pass++) {
                ra[pass] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, ri[pass].x * (CF_D * 4) + 16 * sl, 0, 0);
                rb[pass] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, ri[pass].y * (CF_D * 4) + 16 * sl, 0, 0);
            }
        } else
#pragma unroll
        for (int pass = 0; pass < 4; pass++) {
            const int rl = 8 * pass + sj;
            int ao;
            if (rep > 1) {
                int pos = p0 + rl;
                pos -= pos >= inner ? inner : 0;
                pos -= pos >= inner ? inner : 0;
                pos -= pos >= inner ? inner : 0;
                const long arow = (g0 + ((long)r0 + rl >= GR ? 1 : 0)) * inner + pos;
                ao = (int)(arow * CF_D * 4) + 16 * sl;
            } else ao = (int)((t0 + rl) * CF_D * 4) + 16 * sl;
            ra[pass] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, ao, 0, 0);
            rb[pass] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, (sj * CF_D + 4 * sl) * 4, (int)((t0 + 8 * pass) * CF_D * 4), 0);
        }
        // dropout counter of (row t0 + sj, third 0, this lane's column quad): every hash of the tile is this plus a constant (common.h)
        const uint64_t zlin = mix64_lin(seed, (uint64_t)(t0 + sj) * K4 + (uint64_t)sl);
#pragma unroll
        for (int pass = 0; pass < 4; pass++) {
            const int rl = 8 * pass + sj;
            const long row = t0 + rl;
            const bool ok = row < M;
            float4 v[3];
            v[0] = make_float4(__uint_as_float(ra[pass][0]), __uint_as_float(ra[pass][1]), __uint_as_float(ra[pass][2]), __uint_as_float(ra[pass][3]));
            v[1] = make_float4(__uint_as_float(rb[pass][0]), __uint_as_float(rb[pass][1]), __uint_as_float(rb[pass][2]), __uint_as_float(rb[pass][3]));
            v[2] = f4mul(v[0], v[1]);
            // (the arithmetic of ln_fwd_fast_kernel<1>: same operations, same order)
            float s = f4hsum(v[0]) + f4hsum(v[1]) + f4hsum(v[2]);
            s = group_sum(s, 32);
            const float mu = s * invK;
            float q = 0.f;
#pragma unroll
            for (int t = 0; t < 3; t++) {
                const float4 d = make_float4(v[t].x - mu, v[t].y - mu, v[t].z - mu, v[t].w - mu);
                q += f4hsum(f4mul(d, d));
            }
            q = group_sum(q, 32);
            const float rs = 1.0f / sqrtf(q * invK + eps);
            if (ok && sl == 0) {
                mean[row] = mu;
                rstd[row] = rs;
            }
            float4 o[3];
            float m = 0.f;
#pragma unroll
            for (int t = 0; t < 3; t++) {
                o[t].x = (v[t].x - mu) * rs * gm[t].x + bt[t].x;
                o[t].y = (v[t].y - mu) * rs * gm[t].y + bt[t].y;
                o[t].z = (v[t].z - mu) * rs * gm[t].z + bt[t].z;
                o[t].w = (v[t].w - mu) * rs * gm[t].w + bt[t].w;
                if (DROP && !(CFF_ABL & 2)) o[t] = f4mul(o[t], drop4_fin(zlin + (uint64_t)(pass * 8 * K4 + t * D4) * MIX64_C0, th, inv_keep));   // idx = row * K4 + t * D4 + sl
                if (!ok) o[t] = f4zero();
                // z is written once: the weight-gradient GEMM of the backward contracts over it (rows past the end: dropped by bounds).
                // z == NULL (uniform): the backward rebuilds it (cat3_bwd_dw.hip) -- nothing is stored
                if (z) __builtin_amdgcn_raw_buffer_store_b128((cf_u4){__float_as_uint(o[t].x), __float_as_uint(o[t].y), __float_as_uint(o[t].z), __float_as_uint(o[t].w)},
                                                       rs_z, (sj * K3 + t * CF_D + 4 * sl) * 4, (int)((t0 + 8 * pass) * K3 * 4), CFF_NT);
                m = h_amax3(h_amax3(m, o[t].x, o[t].y), o[t].z, o[t].w);
            }
            m = group_max(m, 32);
            const int up = h_up_field((int)(__float_as_uint(m) >> 23) & 0xff);
            const float sc = __uint_as_float((unsigned)up << 23);
#pragma unroll
            for (int t = 0; t < 3; t++) {
                unsigned h01, l01, h23, l23;
                h_split2(o[t].x, o[t].y, sc, h01, l01);
                h_split2(o[t].z, o[t].w, sc, h23, l23);
                // element (row rl, k = 128 t + 4 sl + e): k-step 8 t + (sl >> 2), operand lane rl + 32 ((sl >> 1) & 1), position 4 (sl & 1) + e
                uint2* dst = ((pass & 1) ? st_o : st_e) + ((8 * t) * 2 * 64 + 16 * (pass >> 1)) * 2;
                dst[0] = make_uint2(h01, h23);
                dst[2 * 64] = make_uint2(l01, l23);
            }
            if (sl == 0) row_up[rl] = up;
        }
        __syncthreads();
        // ---- the product: acc = z tile (32 x 384) . W[32 w .. 32 w + 31, :]^T ----
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
        const uint4* wp = wimg + (size_t)wave * CFF_KS * 64 + lane;
        auto load_b = [&](sf16x8 (&bf)[4][2], int kg) {       // four k-steps
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int p2 = 0; p2 < 2; p2++) bf[k][p2] = __builtin_bit_cast(sf16x8, wp[(size_t)p2 * 4 * CFF_KS * 64 + (4 * kg + k) * 64]);
        };
        auto mul_b = [&](const sf16x8 (&bf)[4][2], int kg, int odd) {      // odd = kg & 1, a literal at both call sites
#pragma unroll
            for (int k = 0; k < 4; k++) {
                sf16x8 af[2];
#pragma unroll
                for (int p2 = 0; p2 < 2; p2++)
                    af[p2] = __builtin_bit_cast(sf16x8, Ap[((4 * kg + k) * 2 + p2) * 64 + (lane ^ (2 * (4 * odd + k)) ^ h)]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1], bf[k][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0], bf[k][1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0], bf[k][0], acc, 0, 0, 0);
            }
        };
        {
            sf16x8 bfa[4][2], bfb[4][2];
            load_b(bfa, 0);
#pragma unroll 1
            for (int kg = 0; kg < ((CFF_ABL & 1) ? 0 : CFF_KS / 4); kg += 2) {
                load_b(bfb, kg + 1);
                mul_b(bfa, kg, 0);
                load_b(bfa, kg + 2 < CFF_KS / 4 ? kg + 2 : 0);
                mul_b(bfb, kg + 1, 1);
            }
        }
        // ---- epilogue: true units, bias, ReLU, ReLU bit mask, store.  C/D layout: column l31, row (r & 3) + 8 (r >> 2) + 4 h ----
        const int* up_h = row_up + 4 * h;
        unsigned myw = 0u;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int rl = (r & 3) + 8 * (r >> 2);
            float v = acc[r] * __builtin_ldexpf(1.0f, 254 - up_h[rl] - w_up) + bsv;
            v = fmaxf(v, 0.f);
            acc[r] = v;
            // one ballot per register = the 32 columns of two rows (lane halves); lane (l31 = r, h) keeps the word of its half
            const unsigned long long bal = __ballot(v > 0.f);
            const unsigned wsel = h ? (unsigned)(bal >> 32) : (unsigned)bal;
            if (l31 == r) myw = wsel;
        }
        const long mrow = t0 + 4 * h + (l31 & 3) + 8 * ((l31 & 15) >> 2);
        if (l31 < 16 && mrow < M) mask_out[(long)wave * M + mrow] = myw;      // [word][row]
#pragma unroll
        for (int r = 0; r < ((CFF_ABL & 4) ? 1 : 16); r++)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[r]), rs_y, (4 * h * CF_D + n) * 4,
                                                  (int)((t0 + (r & 3) + 8 * (r >> 2)) * CF_D * 4), 0);
    }
}
}  // namespace

extern "C" int stage_cat3_ln_gemm_fwd_supported(long long rows, int D, int rep, int inner) {
    if (getenv("STAGE_NO_CAT3_FUSED") || getenv("STAGE_NO_CAT3_FUSED_FWD")) return 0;
    return (D == CF_D && rows >= 4096 && rows * 3ll * D * 4 < (1ll << 31) && rep >= 1 && inner >= 1 && (rep == 1 || inner >= 11) &&
            rows % ((long long)rep * inner) == 0) ? 1 : 0;
}
extern "C" size_t stage_cat3_ln_gemm_fwd_ws_bytes(void) { return cf_align((size_t)CFF_WFRAG * sizeof(uint4)) + 256; }
extern "C" int stage_cat3_ln_gemm_fwd_rag_supported(long long rows, long long a_rows, long long b_rows, int D) {
    if (getenv("STAGE_NO_CAT3_FUSED") || getenv("STAGE_NO_CAT3_FUSED_FWD")) return 0;
    return (D == CF_D && rows >= 1 && rows * 3ll * D * 4 < (1ll << 31) && a_rows >= 1 && a_rows * (long long)D * 4 < (1ll << 31) &&
            b_rows >= 1 && b_rows * (long long)D * 4 < (1ll << 31)) ? 1 : 0;
}

// z (rows, 3D), mean / rstd (rows), y (rows, D) = ReLU(z W^T + bias), relu_mask_out [D/32][rows] (as stage_gemm_nt_mask).  W (D, 3D).
extern "C" int stage_cat3_ln_gemm_fwd(const float* a, const float* b, const float* gamma, const float* beta, const float* W,
                                      const float* bias, float* z, float* mean, float* rstd, float* y, unsigned* relu_mask_out,
                                      long long rows, int D, int rep, int inner, float eps, float p_drop, unsigned long long seed,
                                      void* ws, size_t ws_bytes, void* stream) {
    if (!stage_cat3_ln_gemm_fwd_supported(rows, D, rep, inner)) return STAGE_ERR_SHAPE;
    if (ws_bytes < stage_cat3_ln_gemm_fwd_ws_bytes()) return STAGE_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    uint4* img = (uint4*)ws;
    int* w_up = (int*)((char*)ws + cf_align((size_t)CFF_WFRAG * sizeof(uint4)));
    hipLaunchKernelGGL(cff_prep_w_kernel, dim3(6), dim3(1024), 0, st, W, img, w_up);
    const uint32_t th = drop_thresh16(p_drop);
    const float inv_keep = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    const size_t lds = (size_t)CFF_KS * 2 * 64 * 16 + 32 * 4;
    const long tiles = (long)((rows + 31) / 32);
    static const long grid_cap = getenv("STAGE_CFF_GRID") ? atol(getenv("STAGE_CFF_GRID")) : 3072;   // (developer sweep)
    const int grid = (int)(tiles < grid_cap ? tiles : grid_cap);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)cff_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        (void)hipFuncSetAttribute((const void*)cff_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        attr_done = true;
    }
    if (p_drop > 0.f)
        hipLaunchKernelGGL(cff_fwd_kernel<true>, dim3(grid), dim3(256), lds, st, a, b, gamma, beta, img, w_up, bias, z, mean, rstd, y,
                           relu_mask_out, (long)rows, rep, inner, eps, (uint64_t)seed, th, inv_keep, (const int4*)nullptr, 0l, 0l);
    else
        hipLaunchKernelGGL(cff_fwd_kernel<false>, dim3(grid), dim3(256), lds, st, a, b, gamma, beta, img, w_up, bias, z, mean, rstd, y,
                           relu_mask_out, (long)rows, rep, inner, eps, (uint64_t)seed, th, inv_keep, (const int4*)nullptr, 0l, 0l);
    STAGE_LAUNCH_CHECK();
    return 0;
}

// The same on ragged token rows: row r of z / mean / rstd / y / relu_mask_out (rows) is built from a[rowinfo[r].x] (a_rows x D) and
// b[rowinfo[r].y] (b_rows x D); rowinfo (rows, 4) int32 from stage_rag_rowinfo.
extern "C" int stage_cat3_ln_gemm_fwd_rag(const float* a, const float* b, const float* gamma, const float* beta, const float* W,
                                          const float* bias, float* z, float* mean, float* rstd, float* y, unsigned* relu_mask_out,
                                          const int* rowinfo, long long rows, long long a_rows, long long b_rows, int D, float eps,
                                          float p_drop, unsigned long long seed, void* ws, size_t ws_bytes, void* stream) {
    if (!stage_cat3_ln_gemm_fwd_rag_supported(rows, a_rows, b_rows, D) || !rowinfo) return STAGE_ERR_SHAPE;
    if (ws_bytes < stage_cat3_ln_gemm_fwd_ws_bytes()) return STAGE_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    uint4* img = (uint4*)ws;
    int* w_up = (int*)((char*)ws + cf_align((size_t)CFF_WFRAG * sizeof(uint4)));
    hipLaunchKernelGGL(cff_prep_w_kernel, dim3(6), dim3(1024), 0, st, W, img, w_up);
    const uint32_t th = drop_thresh16(p_drop);
    const float inv_keep = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    const size_t lds = (size_t)CFF_KS * 2 * 64 * 16 + 32 * 4;
    const long tiles = (long)((rows + 31) / 32);
    static const long grid_cap = getenv("STAGE_CFF_GRID") ? atol(getenv("STAGE_CFF_GRID")) : 3072;
    const int grid = (int)(tiles < grid_cap ? tiles : grid_cap);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)cff_fwd_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        (void)hipFuncSetAttribute((const void*)cff_fwd_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        attr_done = true;
    }
    if (p_drop > 0.f)
        hipLaunchKernelGGL((cff_fwd_kernel<true, true>), dim3(grid), dim3(256), lds, st, a, b, gamma, beta, img, w_up, bias, z, mean, rstd, y,
                           relu_mask_out, (long)rows, 1, 1, eps, (uint64_t)seed, th, inv_keep, (const int4*)rowinfo, (long)a_rows, (long)b_rows);
    else
        hipLaunchKernelGGL((cff_fwd_kernel<false, true>), dim3(grid), dim3(256), lds, st, a, b, gamma, beta, img, w_up, bias, z, mean, rstd, y,
                           relu_mask_out, (long)rows, 1, 1, eps, (uint64_t)seed, th, inv_keep, (const int4*)rowinfo, (long)a_rows, (long)b_rows);
    STAGE_LAUNCH_CHECK();
    return 0;
}
