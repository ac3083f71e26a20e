// K1 forward, register-resident version for D = 128 and Lr <= 32 (the video stream: 20 regions per frame) --
// StructuredAttention (model/context_query_attention.py:35-101).  Same math and output contract as str_attn_fwd.hip:
//
//   S  = Cn.Qn^T - 1e10*(1 - cm (x) qm) ,  S_ = softmax(scale*S, -1) * (cm (x) qm) ,  A = S_ . Q
//
// No LDS at all.  A work item is (frame, slice of the NA*Lqa context rows), one wave per item.  Both views of the frame's
// region tile live in registers for the whole item, loaded from global directly in MFMA operand layout:
//   qa  stage-1 A operand   lane (c15, g): region row c15 (permuted for the last region tile), 16 B at chunk 4m+g:
//                           normalised (x * 1/|row|, the norm is two cross-lane-group shuffles) and dropped ONCE per item
//   q2  stage-2 B operand   lane (c15, g): raw region row Rk(g, k), 16 B at d = 64b + 4 c15  (d-permutation: output
//                           column j of tile e is d = 64b + 4j + e, so one float4 feeds four 16-wide tiles)
// and the 16-row context tiles stream past them:
//   stage 1  S^T tile (regions x ctx) = Qn . Cn^T  (v_mfma_f32_16x16x4_f32; B = Cn fragments, prefetched one tile ahead by
//            asm loads that are issued BEFORE the tile's stores and awaited with a counted vmcnt -- vmcnt retires in
//            issue order, a compiler-tracked load behind the stores would drain them)
//            -> lane (c15, g) owns context row c15 and 4 regions per region tile: masked softmax in registers, and the
//               weights ARE the A operand of stage 2 (contraction index g <-> region Rk(g, k); both operands agree)
//   stage 2  A tile (ctx x d) = S_ . Q -> lane owns 4 consecutive d of context rows 4g+reg: 16-B stores, 256 B per row.
// Region permutation of the LAST region tile (PERM/KL) as in str_attn_fwd.hip: tile row 4g+k holds region base + g + 4k.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "../../include/stage_hip.h"

#ifndef K1_ABL
#define K1_ABL 0        // developer ablation bits (tools/ubench/k1_abl.hip): 1 no A stores, 2 no S stores, 4 no stage-2 MFMA,
#endif                  // 8 no stage-1 MFMA -- results are wrong with any bit set, timing experiments only
// The A tile (491 MB per launch, full 256-byte row segments, never re-read here) is stored non-temporally: -7 % kernel
// time.  The S / S_ stores stay plain: their 80-byte rows are partial lines and measured slower with the nt hint.
#ifndef K1_PLAIN_STORES
#define ST4_OUT(p, v) do { float4 v__ = (v); __builtin_nontemporal_store((f32x4){v__.x, v__.y, v__.z, v__.w}, reinterpret_cast<f32x4*>(p)); } while (0)
#else
#define ST4_OUT(p, v) st4((p), (v))
#endif
__device__ __forceinline__ void st4_out(float* p, float4 v) { ST4_OUT(p, v); }
__device__ __forceinline__ void st4_out(stage_bf16* p, float4 v) { stv4(p, v); }   // bf16 storage: 8 bytes per lane, 128 per row half
#ifndef K1_F16
#define K1_F16 1        // both products run as two-way fp16 splits on v_mfma_f32_16x16x32_f16 (either storage type: the operands are fp32 in registers)
#endif                  // (3 instructions of 16 cycles per 32 d instead of 8 of 32 cycles; common.h, DESIGN.md findings 20, 23)
#define RD 128          // row width (floats)
#define RNCH 8          // 4-float chunks per lane group

// TQ: storage type of Q and A (float, or bf16 in the bf16 storage mode; Cn -- small, L2 resident -- and the score maps stay fp32)
template <int RT, int KL, bool PERM, bool TRAIN, bool VEC_S, typename TQ = float, bool FC = false>
__global__ __launch_bounds__(256, 2) void str_attn_fwd_reg_kernel(
    const float* __restrict__ Cn, const TQ* __restrict__ Q, const float* __restrict__ cmask,
    const float* __restrict__ qmask, TQ* __restrict__ A, float* __restrict__ S, float* __restrict__ Sn, int N,
    int NA, int Li, int Lqa, int Lr, float scale, int slices, int tiles_per_slice, uint64_t seed, uint32_t th,
    float inv_keep, unsigned int* __restrict__ ticket, unsigned int ticket_base, int static_rounds,
    const int* __restrict__ fmap, const int2* __restrict__ cq) {
    // cq (FC only, may be NULL): COMPACT region rows -- frame f holds cq[f].y <= Lr rows starting at row cq[f].x of Q (the valid regions
    // + the halo the input encoder's convolutions need; the rows behind them do not exist).  q_mask stays dense.
    // FC (fmap != NULL; a template parameter so that the dense kernels stay exactly the code they were: their hand-counted waits
    // are sensitive to any change of the tile loop): A is FRAME-COMPACT (include/stage_hip.h, "ragged token rows"): example n keeps slots = fmap[N*Li + n] frame slots per
    // candidate (its live frames + one dump slot), first sequence fmap[N*Li + N + n]; fmap[frame] = slot of the frame, < 0: dead (the
    // frame's A rows are never read: they go to the dump slot so that the store count of the tile loop stays exact).  S / S_ stay dense.
    constexpr int NK2 = (RT - 1) * 4 + KL;            // stage-2 k-steps (4 regions each)
    constexpr int base_last = (RT - 1) * 16;
    // A last region tile with <= 4 regions (the headline Lr = 20) does not pay a 16-row stage-1 tile for them: its scores
    // come from v_mfma_f32_4x4x1 (16 independent 4x4 blocks, 8 cycles instead of 32): block (c15 >> 2, g) = 4 regions x
    // context rows 4 (c15 >> 2) .. +3 over the k values lane group g holds; the 4 partial sums per score are folded by a
    // 3-instruction transpose-reduce (v_permlane16_swap / v_permlane32_swap) that leaves lane (c15, g) with region base + g of context row c15 -- the PERM layout.
    constexpr bool T4 = PERM && KL == 1;
    constexpr int RF = T4 ? RT - 1 : RT;              // full 16-region tiles of stage 1
    // Stage 1 of the full tiles on fp16 pairs: both operands are bounded (the region rows are normalised here: |x| <= 1/keep;
    // the context rows get one power-of-two scale per row from their largest magnitude), the contraction index of a
    // 32-wide MFMA step j is d = 16 (2j + c) + 4 g + e (c = 0, 1: the two float4 a lane already holds) for BOTH operands.
    constexpr bool F16S1 = K1_F16 && RF > 0 && !(K1_ABL & 8);
    // Stage 2 likewise: the <= 8 contraction slots of a lane group (NK2 regions) are ONE 32-wide MFMA step.
    // Weights (softmax output, <= 1) scaled by 2^11, the raw region rows by one power of two per frame (undone on the outputs).
    // The 16 operand fragments of the frame (8 d tiles x hi / lo) do not fit next to the stage-1 operands, so every lane parks
    // them in a private 256-byte column of LDS (written once per item, read back per tile as aligned 4-register tuples; a lane
    // only ever reads what it wrote: no barrier) -- the fp32 operands they replace held 40 registers.
    constexpr bool F16S2 = K1_F16 && !(K1_ABL & 4);
    extern __shared__ __attribute__((aligned(16))) uint4 qpark[];      // F16S2: [wave][16 fragments][64 lanes]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int c15 = lane & 15, g = lane >> 4;
    const int CR = NA * Lqa, CT = (CR + 15) >> 4;
    const float inv_lqa = 1.0f / (float)Lqa;

    // region held by this lane in register k of region tile rt after stage 1 (C layout: row 4g + k); recomputed where
    // needed (8 live index registers are 8 too many here), validity kept as one bit mask
    auto Rk = [&](int rt, int k) -> int { return (PERM && rt == RT - 1) ? base_last + g + 4 * k : rt * 16 + 4 * g + k; };
    unsigned vmask = 0u;
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int k = 0; k < 4; k++) vmask |= (Rk(rt, k) < Lr ? 1u : 0u) << (rt * 4 + k);

    // output row of context row c of example n, frame i:  ((n*NA + a)*Li + i)*Lqa + w ,  c = a*Lqa + w
    //   = c + a * (Li - 1) * Lqa + (n*NA*Li + i) * Lqa :  one float reciprocal for a (exact for c < 2^22), one full-rate 24-bit
    // multiply, two adds.  The 64-bit form of the same expression was 3 v_mul_lo_u32 + v_mad_u64_u32 per row (quarter
    // rate), five rows per tile: ~450 VALU cycles of a 2560-cycle tile.  The launcher guarantees U < 2^24 rows.
    const unsigned row_k1 = (unsigned)(Li - 1) * (unsigned)Lqa;
    unsigned row_k0 = 0u;                           // per item
    auto out_row = [&](int c) -> unsigned {
        const int a = (int)(((float)c + 0.5f) * inv_lqa);
        return (unsigned)c + __umul24((unsigned)a, row_k1) + row_k0;
    };
    // the same for the rows of A: dense (fmap == NULL) they are the rows above; frame-compact: ((first + a * slots + slot) * Lqa + w
    unsigned arow_k1 = row_k1, arow_k0 = 0u;        // per item
    auto a_row = [&](int c) -> unsigned {
        if (!FC) return out_row(c);
        const int a = (int)(((float)c + 0.5f) * inv_lqa);
        return (unsigned)c + __umul24((unsigned)a, arow_k1) + arow_k0;
    };

    const long n_items = (long)N * Li * slices;
    const long n_waves = (long)gridDim.x * wpb;
    // Work distribution: the first `static_rounds` rounds are a plain stride (item = wave + round * #waves), only the
    // tail is handed out by tickets (one relaxed atomic per item, drawn at the top of the PREVIOUS item so that its
    // return travels with the frame loads).  All-dynamic costs ~85 us here: every atomic hits the same L2 word (~9 ns
    // each, 2048 of them queued at kernel start) and, vmcnt being in-order, the frame loads behind it cannot retire
    // before it returns.  All-static leaves the waves resident for only ~74 % of the kernel with ragged frames (items
    // without a valid region are almost free).
    long item = (long)blockIdx.x * wpb + wave;
    int round = 0;
    while (item < n_items) {
        round++;
        long next_item;
        if ((K1_ABL & 256) || round < static_rounds) {
            next_item = item + n_waves;
        } else {
            const unsigned drawn = lane == 0 ? __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            next_item = (long)static_rounds * n_waves + (long)(unsigned)__builtin_amdgcn_readfirstlane((int)(drawn - ticket_base));
        }
        const long frame = item / slices;           // n*Li + i
        const int slice = (int)(item % slices);
        const int n = (int)(frame / Li), i = (int)(frame % Li);
        const int tile0 = slice * tiles_per_slice;
        const int tile1 = min(CT, tile0 + tiles_per_slice);
        const TQ* qf = Q + frame * Lr * RD;
        long qrow0 = frame * Lr;                    // first row of the frame in Q (also the dropout counter's row)
        int Lrf = Lr;                               // rows the frame has
        if (FC && cq) {
            const int2 qd = cq[frame];
            qrow0 = __builtin_amdgcn_readfirstlane(qd.x);
            Lrf = __builtin_amdgcn_readfirstlane(qd.y);
            qf = Q + qrow0 * RD;
        }
        const int Lrc = max(Lrf - 1, 0);            // clamp for the rows that do not exist (masked: their values never count)
        row_k0 = ((unsigned)n * NA * Li + i) * Lqa;
        arow_k0 = row_k0;
        bool dead = false;
        if (FC) {
            const int slots = __builtin_amdgcn_readfirstlane(fmap[(long)N * Li + n]);
            const int first = __builtin_amdgcn_readfirstlane(fmap[(long)N * Li + N + n]);
            int slot = __builtin_amdgcn_readfirstlane(fmap[frame]);
            dead = slot < 0;
            slot = dead ? slots - 1 : slot;
            arow_k1 = (unsigned)(slots - 1) * (unsigned)Lqa;
            arow_k0 = (unsigned)(first + slot) * (unsigned)Lqa;
        }

        // region fed by this lane as stage-1 A row (i = c15), per region tile.  Derived per item from an opaque copy of
        // c15: hoisted out of the item loop, the 64-bit row offsets built from it stay live across the whole kernel and
        // push the allocation over 256 VGPRs
        int c15i = c15;
        asm volatile("" : "+v"(c15i));
        int areg[RT];
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
            areg[rt] = (T4 && rt == RT - 1) ? base_last + (c15i & 3)
                       : (PERM && rt == RT - 1) ? base_last + (c15i >> 2) + 4 * (c15i & 3) : rt * 16 + c15i;
        // ---- the frame's operands (compiler-tracked loads: the one full vmcnt drain per item) ----
        float4 qa[RT][RNCH];
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            const int rc = FC ? min(areg[rt], Lrc) : min(areg[rt], Lr - 1);
#pragma unroll
            for (int m = 0; m < RNCH; m++) qa[rt][m] = (K1_ABL & 128) ? make_float4(rc, m, g, 1.f) : ldv4(qf + rc * RD + 4 * (4 * m + g));
        }
        float4 q2[NK2][2];
        float qmk[RT][4];
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
#pragma unroll
            for (int k = 0; k < ((rt == RT - 1) ? KL : 4); k++) {
                const int rc = FC ? min(Rk(rt, k), Lrc) : min(Rk(rt, k), Lr - 1);
#pragma unroll
                for (int b = 0; b < 2; b++) q2[rt * 4 + k][b] = (K1_ABL & 128) ? make_float4(rc, b, c15, 1.f) : ldv4(qf + rc * RD + 64 * b + 4 * c15);
            }
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
#pragma unroll
            for (int k = 0; k < 4; k++) qmk[rt][k] = (K1_ABL & 128) ? 1.f : qmask[frame * Lr + min(Rk(rt, k), Lr - 1)];
        unsigned long long anyb = 0ull;
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (!((vmask >> (rt * 4 + k)) & 1u)) qmk[rt][k] = 0.f;
                anyb |= __ballot(qmk[rt][k] != 0.f);
            }
        // T4: are any of the (<= 4) regions of the last region tile valid in THIS frame?  If not, their scores are exactly
        // -1e10 / 0 whatever the operands are, so the 4x4 blocks of stage 1 and the last k-step of stage 2 are skipped
        // (ragged frames: 8..20 valid regions -> the tail is empty in ~2/3 of the frames)
        bool tail_any = true;
        if (T4) {
            unsigned long long tb = 0ull;
#pragma unroll
            for (int k = 0; k < 4; k++) tb |= __ballot(qmk[RT - 1][k] != 0.f);
            tail_any = tb != 0ull;
        }
        if (anyb == 0ull) {
            // no valid region in this frame: S = -1e10 (cos - 1e10 rounds to -1e10), S_ = 0, A = 0 for the whole slice
            const int c_lo = tile0 * 16, c_hi = min(CR, tile1 * 16);
            const int sq = lane & 31;
            for (int c = c_lo + (lane >> 5); c < c_hi; c += 2) {
                const unsigned orow = out_row(c);
                if (!dead) stv4(A + (size_t)a_row(c) * RD + 4 * sq, f4zero());
                for (int r = sq; r < Lr; r += 32) { S[(size_t)orow * Lr + r] = STAGE_NEG; Sn[(size_t)orow * Lr + r] = 0.f; }
            }
            item = next_item;
            continue;
        }
        // rows >= Lr of the region tile are zero operands (their scores are masked out of the softmax anyway)
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
#pragma unroll
            for (int k = 0; k < ((rt == RT - 1) ? KL : 4); k++)
                if (!((vmask >> (rt * 4 + k)) & 1u)) q2[rt * 4 + k][0] = q2[rt * 4 + k][1] = f4zero();
        // normalise (x * 1/|row|: 1 ulp from x / |row|) and drop the stage-1 operand once
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            float ss = 0.f;
#pragma unroll
            for (int m = 0; m < RNCH; m++) ss += f4hsum(f4mul(qa[rt][m], qa[rt][m]));
            ss = cross_row_sum(ss);
            const float ri = areg[rt] < (FC ? Lrf : Lr) ? 1.0f / fmaxf(sqrtf(ss), 1e-12f) : 0.f;
#pragma unroll
            for (int m = 0; m < RNCH; m++) {
                qa[rt][m] = f4scale(qa[rt][m], ri);
                if (TRAIN)
                    qa[rt][m] = f4mul(qa[rt][m], drop4(seed, (uint64_t)((FC ? qrow0 : frame * Lr) + areg[rt]) * 32 + 4 * m + g, th, inv_keep));
            }
        }

        // fp16 pairs of the normalised region rows (scale 2^qexp: the largest magnitude 1/keep lands below 2^12)
        const int qexp = 11 - (int)((__float_as_uint(inv_keep) >> 23) & 0xff) + 126;   // 11 - ceil(log2(1/keep)) (1/keep = 1: 10)
        unsigned qh[F16S1 ? RF : 1][4][4], ql[F16S1 ? RF : 1][4][4];
        if (F16S1) {
            const float qsc = __uint_as_float((unsigned)(127 + qexp) << 23);
#pragma unroll
            for (int rt = 0; rt < RF; rt++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float4 u = qa[rt][2 * j], w = qa[rt][2 * j + 1];
                    h_split2(u.x, u.y, qsc, qh[rt][j][0], ql[rt][j][0]);
                    h_split2(u.z, u.w, qsc, qh[rt][j][1], ql[rt][j][1]);
                    h_split2(w.x, w.y, qsc, qh[rt][j][2], ql[rt][j][2]);
                    h_split2(w.z, w.w, qsc, qh[rt][j][3], ql[rt][j][3]);
                }
        }

        uint4* const myq = qpark + (size_t)wave * 16 * 64 + lane;
        float inv2 = 1.f;
        if (F16S2) {
            float qmx = 0.f;
#pragma unroll
            for (int ks = 0; ks < NK2; ks++)
#pragma unroll
                for (int b = 0; b < 2; b++) qmx = h_amax3(h_amax3(qmx, q2[ks][b].x, q2[ks][b].y), q2[ks][b].z, q2[ks][b].w);
            qmx = wave_max(qmx);
            const int qu = h_up_field((int)(__float_as_uint(qmx) >> 23) & 0xff);
            const float sc2 = __uint_as_float((unsigned)qu << 23);
            inv2 = __builtin_ldexpf(1.0f, 127 - qu - 11);
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float c8[8];
#pragma unroll
                    for (int ks = 0; ks < 8; ks++) {
                        if (ks < NK2) {
                            const float4 q = q2[ks][b];
                            c8[ks] = e == 0 ? q.x : (e == 1 ? q.y : (e == 2 ? q.z : q.w));
                        } else c8[ks] = 0.f;
                    }
                    uint4 vh = make_uint4(0u, 0u, 0u, 0u), vl = make_uint4(0u, 0u, 0u, 0u);
                    h_split2(c8[0], c8[1], sc2, vh.x, vl.x);
                    if (NK2 > 2) h_split2(c8[2], c8[3], sc2, vh.y, vl.y);
                    if (NK2 > 4) h_split2(c8[4], c8[5], sc2, vh.z, vl.z);
                    if (NK2 > 6) h_split2(c8[6], c8[7], sc2, vh.w, vl.w);
                    myq[((b * 4 + e) * 2 + 0) * 64] = vh;
                    myq[((b * 4 + e) * 2 + 1) * 64] = vl;
                }
        }

        // ---- Cn fragments + context mask of a tile: untracked loads, counted waits ----
        f32x4 cf[RNCH];
        float cmv;
        auto issue_cf = [&](int tile) {
            const int c = min(tile * 16 + c15, CR - 1);  // rows past the end alias the last valid row
            const float* src = Cn + ((long)n * CR + c) * RD + 4 * g;
            const float* pm = cmask + (long)n * CR + c;
#pragma unroll
            for (int m = 0; m < RNCH; m++)
                asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(cf[m]) : "v"(src), "n"(64 * m) : "memory");
            asm volatile("global_load_dword %0, %1, off" : "=v"(cmv) : "v"(pm) : "memory");
        };
        // Stores every tile issues AFTER the next tile's fragments were requested.  The count must be EXACT to be useful:
        // an undercount of 2 makes every tile wait for two of its own stores to be acknowledged (~ the HBM write
        // latency per tile), an overcount would read stale fragments.  The lane-predicated stores of a permuted last
        // region tile are always issued: for k < KL region base + g + 4k is valid at least for g = 0, so the exec mask
        // is never empty.
        constexpr int NST = 8 + (VEC_S ? 2 : 8) * (PERM ? RT - 1 : RT) + (PERM ? 2 * KL : 0);
#define WAIT_CF(n_after)                                                                                            \
    asm volatile("s_waitcnt vmcnt(%9)"                                                                              \
                 : "+v"(cf[0]), "+v"(cf[1]), "+v"(cf[2]), "+v"(cf[3]), "+v"(cf[4]), "+v"(cf[5]), "+v"(cf[6]),       \
                   "+v"(cf[7]), "+v"(cmv)                                                                           \
                 : "n"(n_after)                                                                                     \
                 : "memory")
        issue_cf(tile0);
        WAIT_CF(0);
        for (int t = tile0; t < ((K1_ABL & 64) ? tile0 + 1 : tile1); t++) {
            // ---- stage 1 ----
            f32x4 acc[RT];
#pragma unroll
            for (int rt = 0; rt < RF; rt++) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            f32x4 tl = {0.f, 0.f, 0.f, 0.f};            // T4: 4x4 blocks (one chain: a 16x16 MFMA sits between two links)
#define S1_BODY(TAIL)                                                                                                 \
    _Pragma("unroll") for (int m = 0; m < ((K1_ABL & 8) ? 1 : RNCH); m++) {                                           \
        _Pragma("unroll") for (int rt = 0; rt < RF; rt++) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[rt][m].x, cf[m][0], acc[rt], 0, 0, 0); \
        if (TAIL) tl = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[RT - 1][m].x, cf[m][0], tl, 0, 0, 0);                    \
        _Pragma("unroll") for (int rt = 0; rt < RF; rt++) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[rt][m].y, cf[m][1], acc[rt], 0, 0, 0); \
        if (TAIL) tl = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[RT - 1][m].y, cf[m][1], tl, 0, 0, 0);                    \
        _Pragma("unroll") for (int rt = 0; rt < RF; rt++) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[rt][m].z, cf[m][2], acc[rt], 0, 0, 0); \
        if (TAIL) tl = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[RT - 1][m].z, cf[m][2], tl, 0, 0, 0);                    \
        _Pragma("unroll") for (int rt = 0; rt < RF; rt++) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[rt][m].w, cf[m][3], acc[rt], 0, 0, 0); \
        if (TAIL) tl = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[RT - 1][m].w, cf[m][3], tl, 0, 0, 0);                    \
    }
            if (F16S1) {
                // scale of this lane's context row: its largest magnitude (the row is spread over the 4 lane groups) -> [2^11, 2^12)
                float cmx = 0.f;
#pragma unroll
                for (int m = 0; m < RNCH; m++) cmx = h_amax3(h_amax3(cmx, cf[m][0], cf[m][1]), cf[m][2], cf[m][3]);
                cmx = xmax32(xmax16(cmx));
                const int cu = h_up_field((int)(__float_as_uint(cmx) >> 23) & 0xff);
                const float csc = __uint_as_float((unsigned)cu << 23);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    unsigned bh[4], bl[4];
                    h_split2(cf[2 * j][0], cf[2 * j][1], csc, bh[0], bl[0]);
                    h_split2(cf[2 * j][2], cf[2 * j][3], csc, bh[1], bl[1]);
                    h_split2(cf[2 * j + 1][0], cf[2 * j + 1][1], csc, bh[2], bl[2]);
                    h_split2(cf[2 * j + 1][2], cf[2 * j + 1][3], csc, bh[3], bl[3]);
                    h_operands_ready(bh[0], bh[1], bh[2], bh[3]);
                    h_operands_ready(bl[0], bl[1], bl[2], bl[3]);
                    const sf16x8 vbh = __builtin_bit_cast(sf16x8, make_uint4(bh[0], bh[1], bh[2], bh[3]));
                    const sf16x8 vbl = __builtin_bit_cast(sf16x8, make_uint4(bl[0], bl[1], bl[2], bl[3]));
#pragma unroll
                    for (int rt = 0; rt < RF; rt++) {
                        const sf16x8 vah = __builtin_bit_cast(sf16x8, make_uint4(qh[rt][j][0], qh[rt][j][1], qh[rt][j][2], qh[rt][j][3]));
                        const sf16x8 val = __builtin_bit_cast(sf16x8, make_uint4(ql[rt][j][0], ql[rt][j][1], ql[rt][j][2], ql[rt][j][3]));
                        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(val, vbh, acc[rt], 0, 0, 0);
                        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vah, vbl, acc[rt], 0, 0, 0);
                        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vah, vbh, acc[rt], 0, 0, 0);
                    }
                }
                const float inv = __builtin_ldexpf(1.0f, 127 - cu - qexp);   // back to true units (this lane's context row)
#pragma unroll
                for (int rt = 0; rt < RF; rt++)
#pragma unroll
                    for (int k = 0; k < 4; k++) acc[rt][k] *= inv;
                if (T4 && tail_any) {
#pragma unroll
                    for (int m = 0; m < RNCH; m++) {
                        tl = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[RT - 1][m].x, cf[m][0], tl, 0, 0, 0);
                        tl = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[RT - 1][m].y, cf[m][1], tl, 0, 0, 0);
                        tl = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[RT - 1][m].z, cf[m][2], tl, 0, 0, 0);
                        tl = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[RT - 1][m].w, cf[m][3], tl, 0, 0, 0);
                    }
                }
            } else if (T4 && tail_any) { S1_BODY(true) } else { S1_BODY(false) }   // one uniform branch per tile
#undef S1_BODY
            if (T4) {
                // tl[i] = partial score (this lane group's k values) of region base + i, context row c15
                // fold the 4 lane groups: rows g = 0/1 keep regions 0/1, then halves keep {0,1} / {2,3} (common.h: xsum16 /
                // xsum32 with two inputs are transpose-reduce steps) -> lane (c15, g) holds region base + g, all k
                acc[RT - 1] = (f32x4){xsum32(xsum16(tl[0], tl[1]), xsum16(tl[2], tl[3])), 0.f, 0.f, 0.f};
            }
            const float cm_cur = cmv;
            // the MFMAs above have read cf (in-order issue): request the next tile now, before this tile's stores.
            // Unconditional (the last step re-requests its own tile): an asm result defined inside a branch is merged
            // by register copies, and copying a register whose load is still in flight copies garbage.
            if (!(K1_ABL & 32)) issue_cf(min(t + 1, tile1 - 1));

            const int c = min(t * 16 + c15, CR - 1);
            const size_t srow = __umul24(out_row(c), (unsigned)Lr);   // element offset of the S / S_ row (< 2^29)
            float rv[RT][4], pv[RT][4];
            {   // ---- mask + softmax over regions; pv becomes the stage-2 A operand ----
#pragma clang fp contract(off)  // scale*raw must be ONE rounded value for both the max and the exponent: a contracted
                                // fma(raw, scale, -mx) sees -1e11 exactly vs the rounded max -> exp(-2048) = 0 -> 0/0
                float mx = -INFINITY;
                float msk[RT][4], xs[RT][4];
#pragma unroll
                for (int rt = 0; rt < RT; rt++)
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        msk[rt][k] = cm_cur * qmk[rt][k];
                        rv[rt][k] = acc[rt][k] - 1e10f * (1.0f - msk[rt][k]);
                        xs[rt][k] = rv[rt][k] * scale;
                        if (((vmask >> (rt * 4 + k)) & 1u)) mx = fmaxf(mx, xs[rt][k]);
                    }
                mx = cross_row_max(mx);
                float sum = 0.f;
#pragma unroll
                for (int rt = 0; rt < RT; rt++)
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        pv[rt][k] = (((vmask >> (rt * 4 + k)) & 1u)) ? __expf(xs[rt][k] - mx) : 0.f;  // v_exp_f32 path: ~1e-6 relative
                        sum += pv[rt][k];
                    }
                sum = cross_row_sum(sum);
                const float rsum = __builtin_amdgcn_rcpf(sum);  // 1 ulp
#pragma unroll
                for (int rt = 0; rt < RT; rt++)
#pragma unroll
                    for (int k = 0; k < 4; k++) pv[rt][k] = pv[rt][k] * rsum * msk[rt][k];
            }
            // ---- stores of S / S_ (context row c15 of the tile) ----
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                if (PERM && rt == RT - 1) {
#pragma unroll
                    for (int k = 0; k < KL; k++)
                        if (((vmask >> (rt * 4 + k)) & 1u)) {
                            S[srow + Rk(rt, k)] = rv[rt][k];
                            Sn[srow + Rk(rt, k)] = pv[rt][k];
                        }
                } else if (VEC_S) {
                    if ((K1_ABL & 2) && rv[rt][0] != 1.2345e30f) continue;
                    st4(S + srow + rt * 16 + 4 * g, make_float4(rv[rt][0], rv[rt][1], rv[rt][2], rv[rt][3]));
                    st4(Sn + srow + rt * 16 + 4 * g, make_float4(pv[rt][0], pv[rt][1], pv[rt][2], pv[rt][3]));
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        S[srow + Rk(rt, k)] = rv[rt][k];
                        Sn[srow + Rk(rt, k)] = pv[rt][k];
                    }
                }
            }
            // ---- stage 2: A tile (ctx x d), one 64-wide d block at a time (4 independent accumulator chains) ----
            // C layout: lane (c15, g) holds context rows 4g + reg, columns d = 64b + 4 c15 + e
            size_t arow4[4];
#pragma unroll
            for (int reg = 0; reg < 4; reg++)   // rows past the end alias the last valid row
                arow4[reg] = (size_t)a_row(min(t * 16 + 4 * g + reg, CR - 1)) * RD + 4 * c15;
            uint4 wh4 = make_uint4(0u, 0u, 0u, 0u), wl4 = make_uint4(0u, 0u, 0u, 0u);
            if (F16S2) {   // weights in slot order ks = 4 rt + k (masked / padded slots are exactly 0)
                float w8[8];
#pragma unroll
                for (int ks = 0; ks < 8; ks++) w8[ks] = ks < NK2 ? pv[ks >> 2][ks & 3] : 0.f;
                h_split2(w8[0], w8[1], 2048.f, wh4.x, wl4.x);
                if (NK2 > 2) h_split2(w8[2], w8[3], 2048.f, wh4.y, wl4.y);
                if (NK2 > 4) h_split2(w8[4], w8[5], 2048.f, wh4.z, wl4.z);
                if (NK2 > 6) h_split2(w8[6], w8[7], 2048.f, wh4.w, wl4.w);
                h_operands_ready(wh4.x, wh4.y, wh4.z, wh4.w);
                h_operands_ready(wl4.x, wl4.y, wl4.z, wl4.w);
            }
#pragma unroll
            for (int b = 0; b < 2; b++) {
                f32x4 o[4];
#pragma unroll
                for (int e = 0; e < 4; e++) o[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (F16S2) {
                    const sf16x8 wh8 = __builtin_bit_cast(sf16x8, wh4), wl8 = __builtin_bit_cast(sf16x8, wl4);
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const sf16x8 qh8 = __builtin_bit_cast(sf16x8, myq[((b * 4 + e) * 2 + 0) * 64]);
                        const sf16x8 ql8 = __builtin_bit_cast(sf16x8, myq[((b * 4 + e) * 2 + 1) * 64]);
                        o[e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh8, ql8, o[e], 0, 0, 0);
                        o[e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl8, qh8, o[e], 0, 0, 0);
                        o[e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh8, qh8, o[e], 0, 0, 0);
                    }
#pragma unroll
                    for (int e = 0; e < 4; e++)
#pragma unroll
                        for (int reg = 0; reg < 4; reg++) o[e][reg] *= inv2;
                }
#pragma unroll
                for (int rt = 0; rt < (F16S2 ? 0 : RT); rt++)
#pragma unroll
                    for (int k = 0; k < ((K1_ABL & 4) ? (rt == 0 ? 1 : 0) : ((rt == RT - 1) ? KL : 4)); k++) {
                        if (T4 && rt == RT - 1 && !tail_any) continue;   // all weights of this k-step are exactly 0
                        const float4 q = q2[rt * 4 + k][b];
                        const float p = pv[rt][k];
                        o[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, q.x, o[0], 0, 0, 0);
                        o[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, q.y, o[1], 0, 0, 0);
                        o[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, q.z, o[2], 0, 0, 0);
                        o[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, q.w, o[3], 0, 0, 0);
                    }
#pragma unroll
                for (int reg = 0; reg < 4; reg++)
                    if (!(K1_ABL & 1) || o[0][reg] == 1.2345e30f)
                        st4_out(A + arow4[reg] + 64 * b, make_float4(o[0][reg], o[1][reg], o[2][reg], o[3][reg]));
            }
            if (!(K1_ABL & 32)) WAIT_CF(NST);   // only this tile's stores may still be in flight
        }
#undef WAIT_CF
        item = next_item;
    }
}

template <int RT, int KL, bool PERM, bool TRAIN, bool VEC_S, typename TQ>
static int launch_reg_t(const float* Cn, const TQ* Q, const float* cm, const float* qm, TQ* A, float* S, float* Sn,
                        int N, int NA, int Li, int Lqa, int Lr, float scale, float p_drop, unsigned long long seed,
                        hipStream_t st, const int* fmap, const int* cq) {
    const int CR = NA * Lqa, CT = (CR + 15) / 16;
    // slices of the context tiles: enough work items (frames x slices) to balance ~2048 waves, >= 3 tiles per item
    int slices = 1;
    if (getenv("STAGE_K1_SLICES")) slices = atoi(getenv("STAGE_K1_SLICES"));
    else while (slices < 4 && (long)N * Li * slices < 8192 && CT / (slices + 1) >= 3) slices++;
    if (slices < 1) slices = 1;
    const int tps = (CT + slices - 1) / slices;
    slices = (CT + tps - 1) / tps;
    const long items = (long)N * Li * slices;
    long blocks = 512;                                           // 256 CUs x 8 waves, 4 waves per workgroup
    if (blocks * 4 > items) blocks = (items + 3) / 4;
    uint32_t th = TRAIN ? drop_thresh16(p_drop) : 0u;
    if (TRAIN && th == 0u) th = 1u;
    const float ik = TRAIN ? 1.0f / (1.0f - p_drop) : 1.0f;
    // ~70 % of the items by static stride, the tail by tickets (at least one dynamic round)
    int static_rounds = (int)((items * 7) / (blocks * 4 * 10));
    if (getenv("STAGE_K1_STATIC")) static_rounds = atoi(getenv("STAGE_K1_STATIC"));
    if (static_rounds < 1) static_rounds = 1;
    const long n_waves = blocks * 4;
    while (static_rounds > 1 && (long)static_rounds * n_waves > items) static_rounds--;
    // tickets drawn by this launch (common.h): one per item processed in a round >= static_rounds.  Every wave that has a
    // first item reaches that round (static_rounds * n_waves <= items, or static_rounds == 1), the rounds before it take
    // (static_rounds - 1) items per wave
    const long entering = n_waves < items ? n_waves : items;
    const long draws = items - (long)(static_rounds - 1) * entering;
    const StageTicket tk = stage_next_ticket((unsigned int)draws);
    if (!tk.word) return (int)hipErrorOutOfMemory;
    const size_t park = K1_F16 ? (size_t)4 * 16 * 64 * sizeof(uint4) : 0;   // 64 KB per workgroup
    if constexpr (std::is_same<TQ, float>::value) {      // the frame-compact layout exists for fp32 storage only
        if (fmap) {
            hipLaunchKernelGGL((str_attn_fwd_reg_kernel<RT, KL, PERM, TRAIN, VEC_S, TQ, true>), dim3((unsigned)blocks), dim3(256), park, st, Cn, Q,
                               cm, qm, A, S, Sn, N, NA, Li, Lqa, Lr, scale, slices, tps, (uint64_t)seed, th, ik, tk.word, tk.base,
                               static_rounds, fmap, (const int2*)cq);
            STAGE_LAUNCH_CHECK_TICKET(tk);
            return 0;
        }
    }
    hipLaunchKernelGGL((str_attn_fwd_reg_kernel<RT, KL, PERM, TRAIN, VEC_S, TQ, false>), dim3((unsigned)blocks), dim3(256), park, st, Cn, Q,
                       cm, qm, A, S, Sn, N, NA, Li, Lqa, Lr, scale, slices, tps, (uint64_t)seed, th, ik, tk.word, tk.base,
                       static_rounds, fmap, (const int2*)nullptr);
    STAGE_LAUNCH_CHECK_TICKET(tk);
    return 0;
}

template <int RT, bool TRAIN, typename TQ>
static int launch_reg(const float* Cn, const TQ* Q, const float* cm, const float* qm, TQ* A, float* S, float* Sn,
                      int N, int NA, int Li, int Lqa, int Lr, float scale, float p_drop, unsigned long long seed,
                      hipStream_t st, const int* fmap, const int* cq) {
    const int rem = Lr - 16 * (RT - 1);
    static const bool no_vec8 = getenv("STAGE_K1_NO_VEC8") != nullptr;       // (see str_attn_fwd.hip: 16-byte stores at 8-byte row starts)
    const bool vec = (Lr & 3) == 0 || ((Lr & 1) == 0 && !no_vec8);
#define ARGS Cn, Q, cm, qm, A, S, Sn, N, NA, Li, Lqa, Lr, scale, p_drop, seed, st, fmap, cq
    if (rem == 16) return vec ? launch_reg_t<RT, 4, false, TRAIN, true, TQ>(ARGS) : launch_reg_t<RT, 4, false, TRAIN, false, TQ>(ARGS);
    switch ((rem + 3) / 4) {
        case 1: return vec ? launch_reg_t<RT, 1, true, TRAIN, true, TQ>(ARGS) : launch_reg_t<RT, 1, true, TRAIN, false, TQ>(ARGS);
        case 2: return vec ? launch_reg_t<RT, 2, true, TRAIN, true, TQ>(ARGS) : launch_reg_t<RT, 2, true, TRAIN, false, TQ>(ARGS);
        case 3: return vec ? launch_reg_t<RT, 3, true, TRAIN, true, TQ>(ARGS) : launch_reg_t<RT, 3, true, TRAIN, false, TQ>(ARGS);
        default: return vec ? launch_reg_t<RT, 4, true, TRAIN, true, TQ>(ARGS) : launch_reg_t<RT, 4, true, TRAIN, false, TQ>(ARGS);
    }
#undef ARGS
}

// returns 1 when the shape is not handled here (caller falls back to the LDS-staged kernel), 0 on launch, other = error
template <typename TQ>
static int str_attn_fwd_reg_t(const float* Cn, const TQ* Q, const float* c_mask, const float* q_mask, TQ* A, float* S_raw,
                              float* S_norm, int N, int NA, int Li, int Lqa, int Lr, int D, float scale, float p_drop,
                              unsigned long long seed, void* stream, const int* fmap = nullptr, const int* cq = nullptr) {
    if (D != RD || Lr > 32 || (long)NA * Lqa >= (1 << 22)) return 1;
    if ((long)N * NA * (Li + 1) * Lqa >= (1l << 24) || (long)Li * Lqa >= (1l << 24)) return 1;   // 24-bit row arithmetic (+1: the dump slots of the frame-compact layout)
    hipStream_t st = (hipStream_t)stream;
    const bool train = p_drop > 0.f;
#define ARGS Cn, Q, c_mask, q_mask, A, S_raw, S_norm, N, NA, Li, Lqa, Lr, scale, p_drop, seed, st, fmap, cq
    if (Lr <= 16) return train ? launch_reg<1, true, TQ>(ARGS) : launch_reg<1, false, TQ>(ARGS);
    return train ? launch_reg<2, true, TQ>(ARGS) : launch_reg<2, false, TQ>(ARGS);
#undef ARGS
}

int stage_str_attn_fwd_reg(const float* Cn, const float* Q, const float* c_mask, const float* q_mask, float* A,
                           float* S_raw, float* S_norm, int N, int NA, int Li, int Lqa, int Lr, int D, float scale,
                           float p_drop, unsigned long long seed, void* stream, const int* fmap, const int* cq) {
    if (cq && !fmap) return STAGE_ERR_SHAPE;
    return str_attn_fwd_reg_t<float>(Cn, Q, c_mask, q_mask, A, S_raw, S_norm, N, NA, Li, Lqa, Lr, D, scale, p_drop, seed, stream, fmap, cq);
}
int stage_str_attn_fwd_reg_bf16(const float* Cn, const void* Q, const float* c_mask, const float* q_mask, void* A, float* S_raw,
                                float* S_norm, int N, int NA, int Li, int Lqa, int Lr, int D, float scale, float p_drop,
                                unsigned long long seed, void* stream) {
    return str_attn_fwd_reg_t<stage_bf16>(Cn, (const stage_bf16*)Q, c_mask, q_mask, (stage_bf16*)A, S_raw, S_norm, N, NA, Li, Lqa,
                                          Lr, D, scale, p_drop, seed, stream);
}
