// K1 backward, ONE pass over dA for D = 128 -- StructuredAttention (model/context_query_attention.py:58-61, 81, 95-101).
//
// With P = S_ (saved by the forward), dP = dA . Q^T, G = dS = scale * P * (dP - <P, dP>) (+ the gradient arriving on
// raw_s), the four products of the backward are
//     dP   [c, r] = sum_d dA[c, d] Q [r, d]            dQraw[r, d] = sum_c P[c, r] dA[c, d]
//     dCn  [c, d] = sum_{i, r} G[(c, i), r] Qn[i, r, d] dQn  [r, d] = sum_c G[c, r] Cn[c, d]        (c = NA*Lqa context rows)
// The three-kernel version (str_attn.hip: ds -> dq -> dc) reads dA from HBM twice and writes dS (77 / 192 MB) to read it back
// twice.  Here a workgroup owns (example n, a strided set of frames) and never lets G leave the CU:
//
//   phase 1 (waves split the 16-row context tiles, as in the forward)       for each of the wave's <= 2 tiles:
//       dP^T tile (regions x ctx) = Qraw . dA^T     A operand = Qraw rows from LDS, B operand = dA fragments from global
//       -> lane (c15, g) owns context row c15 and regions rt*16 + 4g + k: softmax backward in registers, G^T is ALREADY
//          the B operand (k = regions) of
//       dCn^T tile (d x ctx) += Qn^T . G^T          A operand = Qn^T from LDS (stored transposed: one ds_read_b128 = 4 k-steps)
//          accumulated in registers over ALL frames of the workgroup (64 accumulator registers per wave), written once as
//          a slab [chunk][n*CR + c][d]; a fixed-order slab sum finishes dCn (deterministic)
//       G tile -> LDS (row c, region columns; xor swizzle so that phase 2 reads it conflict-free)
//   phase 2 (waves split the OUTPUT: wave = (raw | normalised path, 32-wide d block), contraction over all context rows)
//       dQraw[r, d] = sum_c P[c, r] dA[c, d]        A operand = S_ from global (64-byte row pieces, L2-hot from phase 1),
//                                                   B operand = dA re-read in k = c layout (L2 / infinity-cache hot)
//       dQn  [r, d] = sum_c G[c, r] Cn[c, d]        A operand = G from LDS, B operand = Cn (100 KB per example, L2 resident)
//
// Region tiles without a valid region contribute exact zeros (P = 0 => G = 0) and are skipped per frame (ragged frames:
// two thirds of the video frames need one tile of two, fully padded frames cost two zero stores); with an external
// gradient on raw_s nothing is skipped (it may touch padded regions).  Every inner loop has compile-time trip counts
// (NRT = region tiles actually computed is a template parameter of the frame body).
#include <stdlib.h>
#include "common.h"
#include "../../include/stage_hip.h"

#define FD 128          // row width (floats)
#define FLDQ 132        // padded LDS row of the raw region tile
#define FUS_MAX_WGS 1024
// Phase 1 on fp16 pairs (round 4; finding 39): the two products of phase 1 -- dP^T = Q . dA^T and dCn^T += Qn^T . G^T -- run on
// v_mfma_f32_16x16x32_f16 with every fp32 operand split into hi + lo (three products per fp32 product: 5.3 x the rate of the fp32
// matrix instructions they replace).  Q / Qn^T sit in LDS as two fp16 planes each, written once per frame by the staging pass; dA and
// G are split in registers by the lane that owns the context row.  Power-of-two scales: Q one per frame (largest magnitude through an
// LDS atomic max between the two staging halves), Qn fixed 2^12 (its rows are unit vectors times the dropout scale), dA one per context row and frame, G one
// per context row that only ever grows over the frames of a workgroup (the dCn accumulators are rescaled when it does: TN-GEMM rule).
#ifndef FUS_P1_F16
#define FUS_P1_F16 1
#endif
// Phase 2 on fp16 pairs as well (needs FUS_P1_F16; Lqa a multiple of 8, 8-wave workgroups): dQraw = P^T . dA and dQn = G^T . Cn contract
// over the context rows, so a lane's 8-element operand is 8 consecutive context rows of one column -- eight strided loads, the same
// number as the k-steps of 4 they replace -- and both operands carry ONE scale per frame: P <= 1 and |Cn| <= 1 / keep are fixed,
// dA and G use the frame's largest magnitude, published by phase 1 through LDS atomic maxima.
#ifndef FUS_P2_F16
#define FUS_P2_F16 FUS_P1_F16
#endif
#define FUS_UP_P 141    // 2^14: P = S_ <= 1
#define FUS_UP_CN 139   // 2^12: |Cn| <= 1 / (1 - p_drop) < 4
#define FQLD 136        // halfs per row of a Q plane: 272 B -- the 16 rows of a ds_read_b128 lane group sit on 16 different 16-byte slots
#define FUS_UP_QN 139   // scale field (biased exponent) of the Qn planes: 2^12 (|Qn| <= 1 / (1 - p_drop): room up to p = 0.93)
#define FUS_TABLE_BYTES(G, N) ((((size_t)(G) * sizeof(int4) + (size_t)(N) * sizeof(int2)) + 255) & ~(size_t)255)
#define FUS_FRAME_BYTES(N, Li) (((size_t)(N) * (size_t)(Li) + 255) & ~(size_t)255)   // one byte per frame (fus_ext_scan_kernel)
#ifndef FUS_ABL
#define FUS_ABL 0   // developer ablation bits (tools/build_variant.sh; results are wrong with any bit set): 1 no phase 2, 2 no dP MFMAs,
#endif              // 4 no dCn MFMAs, 8 no LDS copy of dA, 16 no G store, 32 no phase-2 MFMAs, 64 no dQ stores
// developer phase timers (STAGE_K1_BWD_TIM = device pointer to 6 uint64): wave cycles per phase, summed over waves
#define TICK(ph) do { if (tim) { unsigned long long tn = __builtin_readcyclecounter(); tacc[ph] += tn - tlast; tlast = tn; } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int fchunk(int g, int m) { return m + 8 * (g >> 1) + 16 * (g & 1); }   // as dchunk(g, m, 8)
__device__ __forceinline__ float2 ld2(const float* p) { return *reinterpret_cast<const float2*>(p); }
__device__ __forceinline__ void st2(float* p, float2 v) { *reinterpret_cast<float2*>(p) = v; }

template <int RT> struct FusLay {
    static constexpr int LT = RT * 16 + 4;            // Qn^T row stride: 4 * odd -> the 16 d-rows of a ds_read_b128 group hit 64 banks
    static constexpr int LG = RT * 16;                // G row stride
    static constexpr bool SWZ = (RT % 2) == 0;        // LG % 32 == 0: rows c, c+1 would share banks -> xor 16 on odd rows
    static constexpr int QR_FLOATS = RT * 16 * FLDQ;
    static constexpr int QT_FLOATS = FD * LT;
};
template <int RT> __device__ __forceinline__ int gcol(int c, int col) { return FusLay<RT>::SWZ ? (col ^ ((c & 1) << 4)) : col; }

// ---------------------------------------------------------------------------------------------------------------------
// phase 1, one 16-row context tile: operands are fetched one tile ahead (fus_p1_fetch of tile s+1 is issued before the
// MFMAs of tile s)
// ---------------------------------------------------------------------------------------------------------------------
template <int PT, bool HAS_EXT> struct FusTile {   // PT = region tiles fetched (>= the NRT a frame body uses)
    float4 gv[8];            // dA fragments: context row c15, floats 4*fchunk(g, m) .. +3
    float2 pq[PT][2];        // S_ pieces: regions rt*16 + 4g + {0,1}, {2,3}
    float2 eq[HAS_EXT ? PT : 1][2];
};
// column swizzle of the LDS copy of dA (pipelined kernel): phase 2 reads rows 4k+g (two rows per 32-lane half) as 8-byte
// pieces of one 128-byte segment -> odd rows move to the other bank half, rows 2,3 mod 4 by 8 floats
// column swizzle (floats) of the LDS copy of dA; three access patterns, all conflict free: coalesced row writes (8 lanes = 128
// contiguous bytes), phase-1 MFMA layout (16 lanes = 16 rows at one column -> 16 distinct 16-byte bank groups), phase-2 rows
// 4k+g as 8-byte pieces (two rows per 32-lane half -> odd rows on the other bank half)
__device__ __forceinline__ int fswz(int c) { return ((c & 1) << 5) | (((c >> 1) & 7) << 2); }

// dAf / Snf / extf: row 0 = output row ((n*NA)*Li + i)*Lqa of the frame (uniform -> scalar base registers); rel = this lane's
// row offset a*Li*Lqa + w (32 bits)
// COAL (with the LDS copy of dA): gv[i] = row 2i + (lane >> 5) of the tile, floats 4*(lane & 31) .. +3 -- every load covers
// two full rows (1 KB contiguous).  In the MFMA layout (lane = row c15, eight 16-byte pieces of one 128-byte line) every
// instruction touches 64 different lines and the eight instructions of a tile re-touch them; with 100+ KB of tiles in flight
// per CU the 32 KB vector L1 evicts them in between and refills them up to eight times (measured ~9 GB/s per CU instead of
// ~25).  The tile is transposed through the LDS copy that phase 2 needs anyway.
// relA / LiA: the same for the rows of dA, whose frames may be compacted (LiA = frame slots of the example; dense: relA = rel, LiA = Li)
template <int NRT, bool HAS_EXT, bool COAL = false, typename TD = float>   // NRT = pieces fetched; TD = storage type of dA
__device__ __forceinline__ void fus_p1_fetch(FusTile<NRT, HAS_EXT>& T, const TD* __restrict__ dAf, const float* __restrict__ Snf,
                                             const float* __restrict__ extf, unsigned rel, unsigned relA, int Lr, int g, int tile = 0, int lane = 0,
                                             int NA = 0, int LiA = 0, int Lqa = 1) {
    if (COAL) {
        const int CR = NA * Lqa;
        int c = tile * 16 + (lane >> 5);
        int a = c / Lqa, w = c - a * Lqa;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const bool ok = c < CR;
            const unsigned r = ok ? (unsigned)(a * LiA * Lqa + w) : (unsigned)((NA - 1) * LiA * Lqa + Lqa - 1);
            T.gv[i] = ldv4(dAf + (r * FD + 4u * (lane & 31)));
            c += 2; w += 2;
            const bool wrap = w >= Lqa;        // Lqa >= 4: at most one wrap per step
            w -= wrap ? Lqa : 0;
            a += wrap ? 1 : 0;
        }
    } else {
        const unsigned offa = relA * FD + 4u * fchunk(g, 0);
#pragma unroll
        for (int m = 0; m < 8; m++) T.gv[m] = ldv4(dAf + (offa + 4u * m));
    }
    const unsigned offs = rel * (unsigned)Lr;
#pragma unroll
    for (int rt = 0; rt < NRT; rt++)
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            const unsigned r0 = (unsigned)min(rt * 16 + 4 * g + 2 * hh, Lr - 2);
            T.pq[rt][hh] = ld2(Snf + (offs + r0));
            if (HAS_EXT) T.eq[rt][hh] = ld2(extf + (offs + r0));
        }
}

#if FUS_P1_F16
// phase 1, one 16-row context tile, fp16-pair products.  Qp: the two Q planes [plane][Lr][FQLD] halfs; Tp: the two Qn^T planes
// [plane][128][LT] halfs; upQ: scale field of this frame's Q planes; geb: running exponent of |G| of this lane's context row
// (all four lanes of a row agree); dcn accumulates in units of 2^(FUS_UP_QN + field(geb) - 254).
template <int RT, int NRT, bool HAS_EXT, int PT = NRT, bool LDSA = false>
__device__ __forceinline__ void fus_p1_tile(const FusTile<PT, HAS_EXT>& T, const float* Qr, const float* QnT, float* Gs, int c,
                                            bool cvalid, int Lr, float scale, f32x4 (&dcn)[8], int c15, int g, int upQ, int& geb,
                                            float* dAs = nullptr, int CRr = 0, unsigned* fmx = nullptr) {
    constexpr int LT = FusLay<RT>::LT, LG = FusLay<RT>::LG;
    const char* Qp = reinterpret_cast<const char*>(Qr);
    const char* Tp = reinterpret_cast<const char*>(QnT);
    float4 gl[8];      // dA fragments: context row c15, floats 4*fchunk(g, m) .. +3 = this lane's 32 consecutive floats 32 sigma(g) ..
    if (LDSA) {        // coalesced rows -> LDS copy (kept for phase 2) -> this wave's fragments; no barrier: its own rows
        const int lane = c15 + 16 * g, tile0 = c - c15;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int row = tile0 + 2 * i + (lane >> 5);
            if (row < CRr) st4(&dAs[row * FD + ((4 * (lane & 31)) ^ fswz(row))], T.gv[i]);
        }
        const int rr = min(c, CRr - 1);           // padded context rows alias the last one (their columns are discarded)
#pragma unroll
        for (int m = 0; m < 8; m++) gl[m] = ld4(&dAs[rr * FD + ((4 * fchunk(g, m)) ^ fswz(rr))]);
    } else {
#pragma unroll
        for (int m = 0; m < 8; m++) gl[m] = T.gv[m];
    }
    // ---- dA row -> B operands: K-block kb = floats 8 kb .. 8 kb + 7 of the lane's 32 (k-slot (g, e) <-> d = 32 sigma(g) + 8 kb + e) ----
    float am = 0.f;
#pragma unroll
    for (int m = 0; m < 8; m++) am = h_amax3(h_amax3(am, gl[m].x, gl[m].y), gl[m].z, gl[m].w);
    am = cross_row_max(am);
    if (FUS_P2_F16 && fmx) {                                  // phase 2 splits dA under ONE scale per frame: its largest magnitude
        const float wm = wave_max(am);
        if (c15 == 0 && g == 0) atomicMax(&fmx[0], __float_as_uint(wm));
    }
    const int up_c = h_up_field((int)(__float_as_uint(am) >> 23) & 0xff);
    const float sc_c = __uint_as_float((unsigned)up_c << 23);
    uint4 bh[4], bl[4];
#pragma unroll
    for (int kb = 0; kb < 4; kb++) {
        h_split2(gl[2 * kb].x, gl[2 * kb].y, sc_c, bh[kb].x, bl[kb].x);
        h_split2(gl[2 * kb].z, gl[2 * kb].w, sc_c, bh[kb].y, bl[kb].y);
        h_split2(gl[2 * kb + 1].x, gl[2 * kb + 1].y, sc_c, bh[kb].z, bl[kb].z);
        h_split2(gl[2 * kb + 1].z, gl[2 * kb + 1].w, sc_c, bh[kb].w, bl[kb].w);
    }
#pragma unroll
    for (int kb = 0; kb < 4; kb++) {
        h_operands_ready(bh[kb].x, bh[kb].y, bh[kb].z, bh[kb].w);
        h_operands_ready(bl[kb].x, bl[kb].y, bl[kb].z, bl[kb].w);
    }
    // ---- dP^T tile (regions x ctx) = Q . dA^T ----
    f32x4 acc[NRT];
    const int plq = Lr * FQLD * 2;                            // bytes of one Q plane
    const int qoff = (4 * fchunk(g, 0)) * 2;                  // this lane group's 32-float slice of a Q row
#pragma unroll
    for (int rt = 0; rt < NRT; rt++) {
        acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const char* qr = Qp + min(rt * 16 + c15, Lr - 1) * (FQLD * 2) + qoff;   // pad rows alias the last region (P = 0 there)
#pragma unroll
        for (int kb = 0; kb < 4; kb++) {
            const sf16x8 ah = __builtin_bit_cast(sf16x8, *reinterpret_cast<const uint4*>(qr + 16 * kb));
            const sf16x8 al = __builtin_bit_cast(sf16x8, *reinterpret_cast<const uint4*>(qr + plq + 16 * kb));
            const sf16x8 vbh = __builtin_bit_cast(sf16x8, bh[kb]), vbl = __builtin_bit_cast(sf16x8, bl[kb]);
            acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, vbh, acc[rt], 0, 0, 0);
            acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, vbl, acc[rt], 0, 0, 0);
            acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, vbh, acc[rt], 0, 0, 0);
        }
    }
    // dP[c][r = rt*16 + 4g + k] in true units
    const int dpe = 254 - upQ - up_c;
    float p[NRT][4], dp[NRT][4], dot = 0.f;
#pragma unroll
    for (int rt = 0; rt < NRT; rt++) {
        const bool rok0 = rt * 16 + 4 * g < Lr, rok1 = rt * 16 + 4 * g + 2 < Lr;
        p[rt][0] = rok0 ? T.pq[rt][0].x : 0.f;
        p[rt][1] = rok0 ? T.pq[rt][0].y : 0.f;
        p[rt][2] = rok1 ? T.pq[rt][1].x : 0.f;
        p[rt][3] = rok1 ? T.pq[rt][1].y : 0.f;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            dp[rt][k] = __builtin_ldexpf(acc[rt][k], dpe);
            dot += p[rt][k] * dp[rt][k];
        }
    }
    dot = cross_row_sum(dot);
    f32x4 G[NRT];
    float gm = 0.f;
#pragma unroll
    for (int rt = 0; rt < NRT; rt++) {
        const bool rok0 = rt * 16 + 4 * g < Lr, rok1 = rt * 16 + 4 * g + 2 < Lr;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float v = scale * p[rt][k] * (dp[rt][k] - dot);
            if (HAS_EXT) {
                const float e = k == 0 ? T.eq[rt][0].x : (k == 1 ? T.eq[rt][0].y : (k == 2 ? T.eq[rt][1].x : T.eq[rt][1].y));
                v += (k < 2 ? rok0 : rok1) ? e : 0.f;
            }
            G[rt][k] = v;
        }
        gm = h_amax3(h_amax3(gm, G[rt][0], G[rt][1]), G[rt][2], G[rt][3]);
        if (cvalid) *reinterpret_cast<f32x4*>(&Gs[c * LG + gcol<RT>(c, rt * 16 + 4 * g)]) = G[rt];
    }
    // ---- G^T of the row -> B operands of dCn^T += Qn^T . G^T: K-block p = region tiles 2p, 2p + 1, k-slot (g, e) <-> region
    // 16 (2p + (e >> 2)) + 4 g + (e & 3) -- exactly the values this lane holds.  Running scale of the row (TN-GEMM rule). ----
    gm = cross_row_max(gm);
    if (FUS_P2_F16 && fmx) {                                  // ... and G
        const float wm = wave_max(gm);
        if (c15 == 0 && g == 0) atomicMax(&fmx[1], __float_as_uint(wm));
    }
    {
        const int ec = (int)(__float_as_uint(gm) >> 23) & 0xff;
        const int neb = ec > geb + 3 ? ec : geb;
        const int d = h_up_field(neb) - h_up_field(geb);
        geb = neb;
        if (__any(d != 0)) {                                  // this row's scale moved: bring its accumulators along
#pragma unroll
            for (int dt = 0; dt < 8; dt++)
#pragma unroll
                for (int k = 0; k < 4; k++) dcn[dt][k] = __builtin_ldexpf(dcn[dt][k], d);
        }
    }
    const float sc_g = __uint_as_float((unsigned)h_up_field(geb) << 23);
    constexpr int NKB = (NRT + 1) / 2;
    uint4 gh[NKB], gq[NKB];
#pragma unroll
    for (int pb = 0; pb < NKB; pb++) {
        h_split2(G[2 * pb][0], G[2 * pb][1], sc_g, gh[pb].x, gq[pb].x);
        h_split2(G[2 * pb][2], G[2 * pb][3], sc_g, gh[pb].y, gq[pb].y);
        if (2 * pb + 1 < NRT) {
            h_split2(G[(2 * pb + 1) < NRT ? 2 * pb + 1 : 0][0], G[(2 * pb + 1) < NRT ? 2 * pb + 1 : 0][1], sc_g, gh[pb].z, gq[pb].z);
            h_split2(G[(2 * pb + 1) < NRT ? 2 * pb + 1 : 0][2], G[(2 * pb + 1) < NRT ? 2 * pb + 1 : 0][3], sc_g, gh[pb].w, gq[pb].w);
        } else {
            gh[pb].z = gh[pb].w = gq[pb].z = gq[pb].w = 0u;   // odd tile count: the upper half of the last K-block is zero
        }
        h_operands_ready(gh[pb].x, gh[pb].y, gh[pb].z, gh[pb].w);
        h_operands_ready(gq[pb].x, gq[pb].y, gq[pb].z, gq[pb].w);
    }
    const int plt = FD * LT * 2;                              // bytes of one Qn^T plane
#pragma unroll
    for (int pb = 0; pb < NKB; pb++) {
        const sf16x8 vgh = __builtin_bit_cast(sf16x8, gh[pb]), vgl = __builtin_bit_cast(sf16x8, gq[pb]);
#pragma unroll
        for (int dt = 0; dt < 8; dt++) {
            const char* tr = Tp + ((dt * 16 + c15) * LT + 32 * pb + 4 * g) * 2;
            // (odd tile count: the upper half of the last K-block is not read -- behind the last row it would leave the plane)
            const bool up = 2 * pb + 1 < NRT;
            const uint2 h0 = *reinterpret_cast<const uint2*>(tr), h1 = up ? *reinterpret_cast<const uint2*>(tr + 32) : make_uint2(0u, 0u);
            const uint2 l0 = *reinterpret_cast<const uint2*>(tr + plt), l1 = up ? *reinterpret_cast<const uint2*>(tr + plt + 32) : make_uint2(0u, 0u);
            const sf16x8 ah = __builtin_bit_cast(sf16x8, make_uint4(h0.x, h0.y, h1.x, h1.y));
            const sf16x8 al = __builtin_bit_cast(sf16x8, make_uint4(l0.x, l0.y, l1.x, l1.y));
            dcn[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, vgh, dcn[dt], 0, 0, 0);
            dcn[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, vgl, dcn[dt], 0, 0, 0);
            dcn[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, vgh, dcn[dt], 0, 0, 0);
        }
    }
}
#else
template <int RT, int NRT, bool HAS_EXT, int PT = NRT, bool LDSA = false>
__device__ __forceinline__ void fus_p1_tile(const FusTile<PT, HAS_EXT>& T, const float* Qr, const float* QnT, float* Gs, int c,
                                            bool cvalid, int Lr, float scale, f32x4 (&dcn)[8], int c15, int g, int upQ, int& geb,
                                            float* dAs = nullptr, int CRr = 0, unsigned* fmx = nullptr) {
    constexpr int LT = FusLay<RT>::LT, LG = FusLay<RT>::LG;
    float4 gl[8];      // dA fragments in MFMA layout: context row c15, floats 4*fchunk(g, m) .. +3
    if (LDSA) {        // coalesced rows -> LDS copy (kept for phase 2) -> this wave's MFMA fragments; no barrier: its own rows
        const int lane = c15 + 16 * g, tile0 = c - c15;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int row = tile0 + 2 * i + (lane >> 5);
            if (row < CRr) st4(&dAs[row * FD + ((4 * (lane & 31)) ^ fswz(row))], T.gv[i]);
        }
        const int rr = min(c, CRr - 1);           // padded context rows alias the last one (their columns are discarded)
#pragma unroll
        for (int m = 0; m < 8; m++) gl[m] = ld4(&dAs[rr * FD + ((4 * fchunk(g, m)) ^ fswz(rr))]);
    } else {
#pragma unroll
        for (int m = 0; m < 8; m++) gl[m] = T.gv[m];
    }
    constexpr int NCHAIN = NRT == 1 ? 2 : 1;   // a single accumulator would be one dependent chain (40 instead of 32 cycles per MFMA)
    f32x4 acc[NRT][NCHAIN];
#pragma unroll
    for (int rt = 0; rt < NRT; rt++)
#pragma unroll
        for (int h = 0; h < NCHAIN; h++) acc[rt][h] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int qrow[NRT];
#pragma unroll
    for (int rt = 0; rt < NRT; rt++) qrow[rt] = min(rt * 16 + c15, Lr - 1) * FLDQ;   // pad rows alias the last region (P = 0 there)
#pragma unroll
    for (int m = 0; m < 8; m++) {
        const float gj[4] = {gl[m].x, gl[m].y, gl[m].z, gl[m].w};
        float4 qv[NRT];
#pragma unroll
        for (int rt = 0; rt < NRT; rt++) qv[rt] = ld4(&Qr[qrow[rt] + 4 * fchunk(g, m)]);
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int rt = 0; rt < NRT; rt++) {
                const float qa = j == 0 ? qv[rt].x : (j == 1 ? qv[rt].y : (j == 2 ? qv[rt].z : qv[rt].w));
                if (!(FUS_ABL & 2)) acc[rt][j % NCHAIN] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa, gj[j], acc[rt][j % NCHAIN], 0, 0, 0);
                else acc[rt][j % NCHAIN][j] += qa * gj[j];
            }
    }
    // dP[c][r = rt*16 + 4g + k] = sum of the chains
    float p[NRT][4], dp[NRT][4], dot = 0.f;
#pragma unroll
    for (int rt = 0; rt < NRT; rt++) {
        const bool rok0 = rt * 16 + 4 * g < Lr, rok1 = rt * 16 + 4 * g + 2 < Lr;
        p[rt][0] = rok0 ? T.pq[rt][0].x : 0.f;
        p[rt][1] = rok0 ? T.pq[rt][0].y : 0.f;
        p[rt][2] = rok1 ? T.pq[rt][1].x : 0.f;
        p[rt][3] = rok1 ? T.pq[rt][1].y : 0.f;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            dp[rt][k] = NCHAIN == 2 ? acc[rt][0][k] + acc[rt][NCHAIN - 1][k] : acc[rt][0][k];
            dot += p[rt][k] * dp[rt][k];
        }
    }
    dot = cross_row_sum(dot);
    f32x4 G[NRT];
#pragma unroll
    for (int rt = 0; rt < NRT; rt++) {
        const bool rok0 = rt * 16 + 4 * g < Lr, rok1 = rt * 16 + 4 * g + 2 < Lr;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float v = scale * p[rt][k] * (dp[rt][k] - dot);
            if (HAS_EXT) {
                const float e = k == 0 ? T.eq[rt][0].x : (k == 1 ? T.eq[rt][0].y : (k == 2 ? T.eq[rt][1].x : T.eq[rt][1].y));
                v += (k < 2 ? rok0 : rok1) ? e : 0.f;
            }
            G[rt][k] = v;
        }
        if (cvalid && !(FUS_ABL & 16)) *reinterpret_cast<f32x4*>(&Gs[c * LG + gcol<RT>(c, rt * 16 + 4 * g)]) = G[rt];
    }
    // dCn^T (d x ctx) += Qn^T . G^T ; k outer so that consecutive MFMAs hit different accumulators; columns of padded
    // context rows are never stored
#pragma unroll
    for (int rt = 0; rt < NRT; rt++)
#pragma unroll
        for (int dh = 0; dh < 8; dh += 4) {
            float4 a4[4];
#pragma unroll
            for (int dt = 0; dt < 4; dt++) a4[dt] = ld4(&QnT[((dh + dt) * 16 + c15) * LT + rt * 16 + 4 * g]);
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int dt = 0; dt < 4; dt++) {
                    const float a = k == 0 ? a4[dt].x : (k == 1 ? a4[dt].y : (k == 2 ? a4[dt].z : a4[dt].w));
                    if (!(FUS_ABL & 4)) dcn[dh + dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, G[rt][k], dcn[dh + dt], 0, 0, 0);
                    else dcn[dh + dt][k] += a * G[rt][k];
                }
        }
}

#endif   // FUS_P1_F16

// ---------------------------------------------------------------------------------------------------------------------
// phase 2: this wave's d block (E floats per lane: 64-wide for 4-wave workgroups, 32-wide for 8-wave ones) of dQraw (RAW)
// or dQn, contraction over the CR context rows in k-steps of 4
// ---------------------------------------------------------------------------------------------------------------------
#define FUS_U 5   // k-steps per fetch group (CR = 200 -> 50 steps = 10 groups)

template <int E> struct FusVec;
template <> struct FusVec<4> { typedef float4 T; };
template <> struct FusVec<2> { typedef float2 T; };
template <int E> __device__ __forceinline__ typename FusVec<E>::T fus_ldv(const float* p) { return *reinterpret_cast<const typename FusVec<E>::T*>(p); }
// bf16 storage: E consecutive bf16 -> E floats
template <int E> __device__ __forceinline__ typename FusVec<E>::T fus_ldv(const stage_bf16* p);
template <> __device__ __forceinline__ float4 fus_ldv<4>(const stage_bf16* p) { return ldv4(p); }
template <> __device__ __forceinline__ float2 fus_ldv<2>(const stage_bf16* p) {
    const unsigned u = *reinterpret_cast<const unsigned*>(p);
    return make_float2(__uint_as_float(u << 16), __uint_as_float(u & 0xFFFF0000u));
}
__device__ __forceinline__ float fus_elt(const float4& v, int e) { return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w)); }
__device__ __forceinline__ float fus_elt(const float2& v, int e) { return e == 0 ? v.x : v.y; }

template <int RT, int NRT, int E, bool RAW, bool PIPE, typename TD = float>
__device__ __forceinline__ void fus_p2(const TD* __restrict__ dAf, const float* __restrict__ Sn, const float* __restrict__ Cn,
                                       const float* Gs, float* __restrict__ out, long frame, int n, int i, int NA, int Li,
                                       int Lqa, int Lr, int d0, int c15, int g, int LiA, int Lrs) {
    typedef typename FusVec<E>::T vec_t;
    constexpr int LG = FusLay<RT>::LG;
    const int CR = NA * Lqa;
    const int nsteps = (CR + 3) >> 2;
    const int ngroups = (nsteps + FUS_U - 1) / FUS_U;
    f32x4 acc[NRT][E];
#pragma unroll
    for (int rt = 0; rt < NRT; rt++)
#pragma unroll
        for (int e = 0; e < E; e++) acc[rt][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // this lane's context row c = 4*step + g, walked as (rel = a*Li*Lqa + w, w); output row = nbase + rel
    const long nbase = ((long)n * NA * Li + i) * Lqa;
    const int jump = (Li - 1) * Lqa;                      // extra row offset when w wraps into the next answer
    const int rel_last = (NA - 1) * Li * Lqa + Lqa - 1;   // clamp target for c >= CR
    const int jumpA = (LiA - 1) * Lqa, relA_last = (NA - 1) * LiA * Lqa + Lqa - 1;   // the rows of dA (dAf = the frame's first row)
    int c = g, w = g, rel = g, relA = g;                  // Lqa >= 4 > g
    int rcl[NRT];
#pragma unroll
    for (int rt = 0; rt < NRT; rt++) rcl[rt] = min(rt * 16 + c15, Lr - 1);

    auto fetch = [&](vec_t (&bv)[FUS_U], float (&av)[FUS_U][NRT], bool (&okv)[FUS_U]) {
#pragma unroll
        for (int u = 0; u < FUS_U; u++) {
            const bool ok = c < CR;
            okv[u] = ok;
            const int cc = ok ? c : CR - 1;
            const long orow = nbase + (ok ? rel : rel_last);
            if (RAW) {
                bv[u] = fus_ldv<E>(dAf + (long)(ok ? relA : relA_last) * FD + d0);
#pragma unroll
                for (int rt = 0; rt < NRT; rt++) av[u][rt] = Sn[orow * Lr + rcl[rt]];
            } else {
                bv[u] = fus_ldv<E>(Cn + ((long)n * CR + cc) * FD + d0);
#pragma unroll
                for (int rt = 0; rt < NRT; rt++) av[u][rt] = Gs[cc * LG + gcol<RT>(cc, rt * 16 + c15)];
            }
            c += 4; w += 4; rel += 4; relA += 4;
            const bool wrap = w >= Lqa;      // Lqa >= 4: at most one wrap per step; selects, not a divergent loop
            w -= wrap ? Lqa : 0;
            rel += wrap ? jump : 0;
            relA += wrap ? jumpA : 0;
        }
    };
    auto mul = [&](const vec_t (&bv)[FUS_U], const float (&av)[FUS_U][NRT], const bool (&okv)[FUS_U]) {
#pragma unroll
        for (int u = 0; u < FUS_U; u++) {
            float b[E];
#pragma unroll
            for (int e = 0; e < E; e++) b[e] = okv[u] ? fus_elt(bv[u], e) : 0.f;
#pragma unroll
            for (int rt = 0; rt < NRT; rt++) {
                const float a = (okv[u] && rt * 16 + c15 < Lr) ? av[u][rt] : 0.f;
#pragma unroll
                for (int e = 0; e < E; e++) acc[rt][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[e], acc[rt][e], 0, 0, 0);
            }
        }
    };
    if (PIPE) {   // 2 waves per SIMD: groups double-buffered in registers
        vec_t b0[FUS_U], b1[FUS_U];
        float a0[FUS_U][NRT], a1[FUS_U][NRT];
        bool k0[FUS_U], k1[FUS_U];
        fetch(b0, a0, k0);
        for (int gi = 0; gi < ngroups; gi += 2) {
            fetch(b1, a1, k1);          // group gi + 1 (clamped + zeroed past the end)
            mul(b0, a0, k0);
            fetch(b0, a0, k0);          // group gi + 2
            if (gi + 1 < ngroups) mul(b1, a1, k1);
        }
    } else {      // 4 waves per SIMD cover the latency of a group, the registers go to the dCn accumulators instead
        for (int gi = 0; gi < ngroups; gi++) {
            vec_t b0[FUS_U];
            float a0[FUS_U][NRT];
            bool k0[FUS_U];
            fetch(b0, a0, k0);
            mul(b0, a0, k0);
        }
    }
    // C layout: row 4g + reg = region inside the tile, column c15 -> d = d0 + e.  `out` = the frame's first output row; Lrs = rows it has
    float* o = out + d0;
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int r = rt * 16 + 4 * g + reg;
            if (r < Lrs) {
                constexpr int Z = 0;
                const int ra = rt < NRT ? rt : Z;
                if (E == 4) {
                    st4(o + r * FD, rt < NRT ? make_float4(acc[ra][0][reg], acc[ra][1][reg], acc[ra][E - 2][reg], acc[ra][E - 1][reg]) : f4zero());
                } else {
                    st2(o + r * FD, rt < NRT ? make_float2(acc[ra][0][reg], acc[ra][1][reg]) : make_float2(0.f, 0.f));
                }
            }
        }
}

// The same with a UNIFORM row walk (Lqa % 4 == 0 and CR/4 a multiple of FUS_U: the published shapes): the four lane groups
// of a k-step sit in the same answer block, so the row pointer advances in scalar registers and every load is
// (scalar base) + (constant lane offset) -- no vector address arithmetic, no validity selects between the MFMAs.
// dAf / Snf / outf are the frame bases of fus_p1_fetch, Cnn = Cn + n*CR*128.
template <int RT, int NRT, int E, bool RAW, bool PIPE, bool LDSA = false, int UG = FUS_U, typename TD = float>
__device__ __forceinline__ void fus_p2_unif(const TD* __restrict__ dAf, const float* __restrict__ Snf,
                                            const float* __restrict__ Cnn, const float* Gs, float* __restrict__ outf, int NA,
                                            int Li, int Lqa, int Lr, int d0, int c15, int g, const float* dAs = nullptr, int LiA = 0,
                                            int Lrs = 0) {
    typedef typename FusVec<E>::T vec_t;
    constexpr int LG = FusLay<RT>::LG;
    const int ngroups = (NA * Lqa) / (4 * UG);
    f32x4 acc[NRT][E];
#pragma unroll
    for (int rt = 0; rt < NRT; rt++)
#pragma unroll
        for (int e = 0; e < E; e++) acc[rt][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned offB = (RAW && LDSA) ? (unsigned)(g * FD + (d0 ^ fswz(g))) : (unsigned)(g * FD + d0);
    unsigned offA[NRT];
#pragma unroll
    for (int rt = 0; rt < NRT; rt++)
        offA[rt] = RAW ? (unsigned)(g * Lr + min(rt * 16 + c15, Lr - 1)) : (unsigned)(g * LG + gcol<RT>(g, rt * 16 + c15));
    const TD* pBd = dAf;                                  // row 4*step of the B operand (uniform): dA rows (RAW) ...
    const float* pBc = Cnn;                               // ... or Cn rows
    const float* pA = Snf;
    const long jumpB = (long)(LiA - 1) * Lqa * FD, jumpA = (long)(Li - 1) * Lqa * Lr;
    const int wq_n = Lqa >> 2;
    int wq = 0, gs = 0, bs = 0;

    auto fetch = [&](vec_t (&bv)[UG], float (&av)[UG][NRT]) {
#pragma unroll
        for (int u = 0; u < UG; u++) {
            // fswz(4k + g) = fswz(g) ^ (((2k) & 7) << 2) (g < 4 sets disjoint bits); bs = 4k * FD
            bv[u] = (RAW && LDSA) ? fus_ldv<E>(dAs + bs + (offB ^ (unsigned)(((bs >> 8) & 7) << 2))) : (RAW ? fus_ldv<E>(pBd + offB) : fus_ldv<E>(pBc + offB));
#pragma unroll
            for (int rt = 0; rt < NRT; rt++) av[u][rt] = RAW ? pA[offA[rt]] : Gs[gs + offA[rt]];
            pBd += 4 * FD;
            pBc += 4 * FD;
            gs += 4 * LG;
            bs += 4 * FD;
            if (RAW) {
                pA += 4 * Lr;
                const bool wrap = ++wq == wq_n;       // uniform
                wq = wrap ? 0 : wq;
                pBd += wrap ? jumpB : 0;
                pA += wrap ? jumpA : 0;
            }
        }
    };
    auto mul = [&](const vec_t (&bv)[UG], const float (&av)[UG][NRT]) {
#pragma unroll
        for (int u = 0; u < UG; u++)
#pragma unroll
            for (int rt = 0; rt < NRT; rt++)
#pragma unroll
                for (int e = 0; e < E; e++)   // A rows >= Lr only feed output rows that are never stored
                    if (!(FUS_ABL & 32)) acc[rt][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][rt], fus_elt(bv[u], e), acc[rt][e], 0, 0, 0);
                    else acc[rt][e][u & 3] += av[u][rt] * fus_elt(bv[u], e);
    };
    if (PIPE) {
        vec_t b0[UG], b1[UG];
        float a0[UG][NRT], a1[UG][NRT];
        fetch(b0, a0);
        for (int gi = 0; gi < ngroups; gi += 2) {
            if (gi + 1 < ngroups) fetch(b1, a1);
            mul(b0, a0);
            if (gi + 2 < ngroups) fetch(b0, a0);
            if (gi + 1 < ngroups) mul(b1, a1);
        }
    } else {
        for (int gi = 0; gi < ngroups; gi++) {
            vec_t b0[UG];
            float a0[UG][NRT];
            fetch(b0, a0);
            mul(b0, a0);
        }
    }
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int r = rt * 16 + 4 * g + reg;
            if (r < Lrs) {
                constexpr int Z = 0;
                const int ra = rt < NRT ? rt : Z;
                if (E == 4) {
                    st4(outf + r * FD + d0, rt < NRT ? make_float4(acc[ra][0][reg], acc[ra][1][reg], acc[ra][E - 2][reg], acc[ra][E - 1][reg]) : f4zero());
                } else {
                    st2(outf + r * FD + d0, rt < NRT ? make_float2(acc[ra][0][reg], acc[ra][1][reg]) : make_float2(0.f, 0.f));
                }
            }
        }
}

#if FUS_P2_F16
// phase 2, fp16 pairs: this wave's 32-wide d block (d = d0 + e, e < 2; d0 = 32 (wave % 4) + 2 c15) of dQraw (RAW: A = P from global,
// B = dA from its LDS copy or global) or dQn (A = G from LDS, B = Cn), contraction over the context rows in K-blocks of 32
// (k-slot (g, j) <-> row 32 kb + 8 g + j).  fA / fB: scale fields of the two operands.
template <int RT, int NRT, bool RAW, bool LDSA, typename TD>
__device__ __forceinline__ void fus_p2_f16(const TD* __restrict__ dAf, const float* __restrict__ Snf, const float* __restrict__ Cnn,
                                           const float* Gs, float* __restrict__ outf, int NA, int Li, int Lqa, int Lr, int d0, int c15,
                                           int g, const float* dAs, int LiA, int Lrs, int fA, int fB) {
    constexpr int LG = FusLay<RT>::LG;
    const int CR = NA * Lqa, nkb = (CR + 31) >> 5;
    f32x4 acc[NRT][2];
#pragma unroll
    for (int rt = 0; rt < NRT; rt++)
#pragma unroll
        for (int e = 0; e < 2; e++) acc[rt][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float sA = __uint_as_float((unsigned)fA << 23), sB = __uint_as_float((unsigned)fB << 23);
    int colA[NRT];
#pragma unroll
    for (int rt = 0; rt < NRT; rt++) colA[rt] = min(rt * 16 + c15, Lr - 1);
    // one K-block ahead: the B rows (8 x float2) and the row base of the A columns; the A values of a region tile are requested one
    // tile ahead of their split (a full double buffer of A -- 8 NRT registers more -- spills)
    struct RawB { float2 b[8]; };
    auto rows = [&](int kb, bool& ok, int& cc, int& an, int& w) {
        const int c0 = 32 * kb + 8 * g;                   // 8 rows of one answer (Lqa % 8 == 0): valid or past the end as a whole
        ok = c0 < CR;
        cc = ok ? c0 : 0;
        an = cc / Lqa;
        w = cc - an * Lqa;
    };
    auto fetch_b = [&](RawB& R, int kb) {
        bool ok; int cc, an, w;
        rows(kb, ok, cc, an, w);
        if (RAW) {
            if (LDSA) {
#pragma unroll
                for (int j = 0; j < 8; j++) R.b[j] = ld2(&dAs[(cc + j) * FD + (d0 ^ fswz(cc + j))]);
            } else {
                const TD* pb = dAf + (long)(an * LiA * Lqa + w) * FD + d0;
#pragma unroll
                for (int j = 0; j < 8; j++) R.b[j] = fus_ldv<2>(pb + j * FD);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) R.b[j] = ld2(Cnn + (long)(cc + j) * FD + d0);
        }
    };
    auto fetch_a = [&](float (&a)[8], int kb, int rt) {
        bool ok; int cc, an, w;
        rows(kb, ok, cc, an, w);
        if (RAW) {
            const float* pa = Snf + (long)(an * Li * Lqa + w) * Lr + colA[rt < NRT ? rt : 0];
#pragma unroll
            for (int j = 0; j < 8; j++) a[j] = pa[j * Lr];
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) a[j] = Gs[(cc + j) * LG + gcol<RT>(cc + j, (rt < NRT ? rt : 0) * 16 + c15)];
        }
    };
    auto split_b = [&](const RawB& R, uint4 (&bh)[2], uint4 (&bl)[2]) {
        h_split2(R.b[0].x, R.b[1].x, sB, bh[0].x, bl[0].x);
        h_split2(R.b[2].x, R.b[3].x, sB, bh[0].y, bl[0].y);
        h_split2(R.b[4].x, R.b[5].x, sB, bh[0].z, bl[0].z);
        h_split2(R.b[6].x, R.b[7].x, sB, bh[0].w, bl[0].w);
        h_split2(R.b[0].y, R.b[1].y, sB, bh[1].x, bl[1].x);
        h_split2(R.b[2].y, R.b[3].y, sB, bh[1].y, bl[1].y);
        h_split2(R.b[4].y, R.b[5].y, sB, bh[1].z, bl[1].z);
        h_split2(R.b[6].y, R.b[7].y, sB, bh[1].w, bl[1].w);
#pragma unroll
        for (int e = 0; e < 2; e++) {
            h_operands_ready(bh[e].x, bh[e].y, bh[e].z, bh[e].w);
            h_operands_ready(bl[e].x, bl[e].y, bl[e].z, bl[e].w);
        }
    };
    {
        RawB B0, B1;
        float a0[8], a1[8];
        fetch_b(B0, 0);
        fetch_a(a0, 0, 0);
#pragma unroll 1
        for (int kb = 0; kb < nkb; kb++) {
            const bool more = kb + 1 < nkb;
            if (more) fetch_b(B1, kb + 1);
            uint4 bh[2], bl[2];
            split_b(B0, bh, bl);
            const float sa = (32 * kb + 8 * g < CR) ? sA : 0.f;            // rows past the end: the A operand is zero
#pragma unroll
            for (int rt = 0; rt < NRT; rt++) {
                // the A values of the next tile (or of tile 0 of the next K-block) are requested before this tile is multiplied
                if (rt + 1 < NRT) fetch_a((rt & 1) ? a0 : a1, kb, rt + 1);
                else if (more) fetch_a((rt & 1) ? a0 : a1, kb + 1, 0);
                const float (&av)[8] = (rt & 1) ? a1 : a0;
                uint4 ah, al;
                h_split2(av[0], av[1], sa, ah.x, al.x);
                h_split2(av[2], av[3], sa, ah.y, al.y);
                h_split2(av[4], av[5], sa, ah.z, al.z);
                h_split2(av[6], av[7], sa, ah.w, al.w);
                h_operands_ready(ah.x, ah.y, ah.z, ah.w);
                h_operands_ready(al.x, al.y, al.z, al.w);
                const sf16x8 vah = __builtin_bit_cast(sf16x8, ah), val = __builtin_bit_cast(sf16x8, al);
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const sf16x8 vbh = __builtin_bit_cast(sf16x8, bh[e]), vbl = __builtin_bit_cast(sf16x8, bl[e]);
                    acc[rt][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(val, vbh, acc[rt][e], 0, 0, 0);
                    acc[rt][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vah, vbl, acc[rt][e], 0, 0, 0);
                    acc[rt][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vah, vbh, acc[rt][e], 0, 0, 0);
                }
            }
            // (NRT odd: tile 0 of the next block went into the other buffer -- keep the roles aligned)
            if (NRT & 1) {
#pragma unroll
                for (int j = 0; j < 8; j++) { const float t = a0[j]; a0[j] = a1[j]; a1[j] = t; }
            }
#pragma unroll
            for (int j = 0; j < 8; j++) B0.b[j] = B1.b[j];
        }
    }
    const int de = 254 - fA - fB;                         // back to true units
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int r = rt * 16 + 4 * g + reg;
            if (r < Lrs) {
                constexpr int Z = 0;
                const int ra = rt < NRT ? rt : Z;
                st2(outf + r * FD + d0, rt < NRT ? make_float2(__builtin_ldexpf(acc[ra][0][reg], de), __builtin_ldexpf(acc[ra][1][reg], de))
                                                 : make_float2(0.f, 0.f));
            }
        }
}
#endif

// one frame: phase 1 over the wave's tiles (slot s -> tile wave + NW*s), barrier, phase 2
template <int RT, int NRT, int NW, bool HAS_EXT, bool PIPE, bool LDSA, typename TD = float>
__device__ __forceinline__ void fus_frame(const TD* __restrict__ dA, const float* __restrict__ ext,
                                          const float* __restrict__ Cn, const float* __restrict__ Sn, const float* Qr,
                                          const float* QnT, float* Gs, float* dAs, float* __restrict__ dQraw, float* __restrict__ dQn,
                                          long frame, int n, int i, int NA, int Li, int Lqa, int Lr, float scale,
                                          const unsigned (&orel)[16 / NW], int ntiles, f32x4 (&dcn)[16 / NW][8], int wave, int lane,
                                          unsigned long long* tim, unsigned long long (&tacc)[6], unsigned long long& tlast,
                                          const unsigned (&orelA)[16 / NW], long arow0, int LiA, long qrow0, int Lrs, int upQ,
                                          int (&geb)[16 / NW], unsigned* fmx) {
    constexpr int TPW = 16 / NW, E = NW == 4 ? 4 : 2;
    const int CR = NA * Lqa;
    // c15 / g re-derived from an opaque copy of the lane id per frame: hoisted out of the frame loop, the address
    // arithmetic built on them stays live across the kernel and is spilled to scratch
    int l = lane;
    asm volatile("" : "+v"(l));
    const int c15 = l & 15, g = l >> 4;
    const long rowbase = ((long)n * NA * Li + i) * Lqa;     // uniform
    const TD* dAf = dA + arow0 * FD;                        // first row of the frame in dA (dense: rowbase)
    const float* Snf = Sn + rowbase * Lr;
    const float* extf = HAS_EXT ? ext + rowbase * Lr : nullptr;
    if (PIPE) {
        FusTile<NRT, HAS_EXT> Ta, Tb;
        if (ntiles > 0) fus_p1_fetch<NRT, HAS_EXT, LDSA, TD>(Ta, dAf, Snf, extf, orel[0], orelA[0], Lr, g, wave, l, NA, LiA, Lqa);
#pragma unroll
        for (int s = 0; s < TPW; s++) {
            if (s < ntiles) {
                const int c = (wave + NW * s) * 16 + c15;
                if (s & 1) {
                    if (s + 1 < TPW && s + 1 < ntiles) fus_p1_fetch<NRT, HAS_EXT, LDSA, TD>(Ta, dAf, Snf, extf, orel[s + 1 < TPW ? s + 1 : 0], orelA[s + 1 < TPW ? s + 1 : 0], Lr, g, wave + NW * (s + 1), l, NA, LiA, Lqa);
                    fus_p1_tile<RT, NRT, HAS_EXT, NRT, LDSA>(Tb, Qr, QnT, Gs, c, c < CR, Lr, scale, dcn[s], c15, g, upQ, geb[s], dAs, CR, fmx);
                } else {
                    if (s + 1 < TPW && s + 1 < ntiles) fus_p1_fetch<NRT, HAS_EXT, LDSA, TD>(Tb, dAf, Snf, extf, orel[s + 1 < TPW ? s + 1 : 0], orelA[s + 1 < TPW ? s + 1 : 0], Lr, g, wave + NW * (s + 1), l, NA, LiA, Lqa);
                    fus_p1_tile<RT, NRT, HAS_EXT, NRT, LDSA>(Ta, Qr, QnT, Gs, c, c < CR, Lr, scale, dcn[s], c15, g, upQ, geb[s], dAs, CR, fmx);
                }
            }
        }
    } else {
#pragma unroll
        for (int s = 0; s < TPW; s++) {
            if (s < ntiles) {
                const int c = (wave + NW * s) * 16 + c15;
                FusTile<NRT, HAS_EXT> T;
                fus_p1_fetch<NRT, HAS_EXT, LDSA, TD>(T, dAf, Snf, extf, orel[s], orelA[s], Lr, g, wave + NW * s, l, NA, LiA, Lqa);
                fus_p1_tile<RT, NRT, HAS_EXT, NRT, LDSA>(T, Qr, QnT, Gs, c, c < CR, Lr, scale, dcn[s], c15, g, upQ, geb[s], dAs, CR, fmx);
            }
        }
    }
    TICK(2);
    __syncthreads();   // G of the whole frame is in LDS
    TICK(3);
    const int d0 = (16 * E) * (wave % (NW / 2)) + E * c15;
    // one region tile: 2 MFMAs per k-step -- the L2 latency of the S_ / Cn operands needs 10 k-steps per group in flight
    constexpr int UG = (PIPE && NRT == 1) ? 10 : FUS_U;
    const bool unif = (Lqa & 3) == 0 && ((CR >> 2) % UG) == 0;
    if (FUS_ABL & 1) return;
#if FUS_P2_F16
    if (E == 2 && (Lqa & 7) == 0) {
        const int upD = h_up_field((int)(fmx[0] >> 23) & 0xff), upG = h_up_field((int)(fmx[1] >> 23) & 0xff);
        if (wave < NW / 2) fus_p2_f16<RT, NRT, true, LDSA, TD>(dAf, Snf, nullptr, Gs, dQraw + qrow0 * FD, NA, Li, Lqa, Lr, d0, c15, g, dAs, LiA, Lrs, FUS_UP_P, upD);
        else fus_p2_f16<RT, NRT, false, false, TD>(dAf, Snf, Cn + (long)n * CR * FD, Gs, dQn + qrow0 * FD, NA, Li, Lqa, Lr, d0, c15, g, nullptr, LiA, Lrs, upG, FUS_UP_CN);
        return;
    }
#endif
    if (unif) {
        if (wave < NW / 2) fus_p2_unif<RT, NRT, E, true, PIPE, LDSA, UG, TD>(dAf, Snf, nullptr, Gs, dQraw + qrow0 * FD, NA, Li, Lqa, Lr, d0, c15, g, dAs, LiA, Lrs);
        else fus_p2_unif<RT, NRT, E, false, PIPE, false, UG, TD>(dAf, Snf, Cn + (long)n * CR * FD, Gs, dQn + qrow0 * FD, NA, Li, Lqa, Lr, d0, c15, g, nullptr, LiA, Lrs);
    } else {
        if (wave < NW / 2) fus_p2<RT, NRT, E, true, PIPE, TD>(dAf, Sn, Cn, Gs, dQraw + qrow0 * FD, frame, n, i, NA, Li, Lqa, Lr, d0, c15, g, LiA, Lrs);
        else fus_p2<RT, NRT, E, false, PIPE, TD>(dAf, Sn, Cn, Gs, dQn + qrow0 * FD, frame, n, i, NA, Li, Lqa, Lr, d0, c15, g, LiA, Lrs);
    }
}

// TD: storage type of dA, Q and Qn (float, or bf16 in the bf16 storage mode: converted as they are loaded; the LDS images, the
// score maps, Cn and all three gradients are fp32)
template <int RT, int NW, bool HAS_EXT, int OCC, bool LDSA, typename TD = float>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void str_attn_bwd_fused_kernel(
    const TD* __restrict__ dA, const float* __restrict__ ext, const float* __restrict__ Cn, const TD* __restrict__ Q,
    const TD* __restrict__ Qn, const float* __restrict__ Sn, const float* __restrict__ qmask, float* __restrict__ dQraw,
    float* __restrict__ dQn, float* __restrict__ part, int N, int NA, int Li, int Lqa, int Lr, float scale,
    const int4* __restrict__ sched, const unsigned char* __restrict__ fnv, unsigned long long* __restrict__ tim,
    const int* __restrict__ fmap, const int2* __restrict__ cq) {
    // cq != NULL: Q, Qn, dQraw, dQn hold COMPACT region rows (frame f = rows cq[f].x .. + cq[f].y - 1); q_mask stays dense
    // fmap != NULL: dA is frame-compact (str_attn_fwd_reg.hip, include/stage_hip.h "ragged token rows"): the frames of example n sit in
    // fmap[N*Li + n] slots per candidate from sequence fmap[N*Li + N + n]; a dead frame (fmap[frame] < 0) reads the example's dump
    // slot, which the caller has zeroed -- its dA is exactly zero by construction (the statement mask blocks the gradient)
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tlast = tim ? __builtin_readcyclecounter() : 0;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int LT = FusLay<RT>::LT, TPW = 16 / NW, NT = 64 * NW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int CR = NA * Lqa, CT = (CR + 15) >> 4;
#if FUS_P1_F16
    float* Qr = lds;                                  // [2 planes][Lr][FQLD] halfs   raw regions, hi / lo (reads of pad rows are clamped to the last region)
    float* QnT = Qr + Lr * FQLD;                      // [2 planes][128][LT] halfs    normalised regions, transposed, pad columns zero
#else
    float* Qr = lds;                                  // [Lr][FLDQ]     raw regions (reads of pad rows are clamped to the last region)
    float* QnT = Qr + Lr * FLDQ;                      // [128][LT]      normalised regions, transposed, pad columns zero
#endif
    float* Gs = QnT + FusLay<RT>::QT_FLOATS;          // [CR][LG]       dS of the frame
    float* dAs = Gs + CR * FusLay<RT>::LG;            // [CR][128]      dA of the frame (LDSA: phase 2 re-reads it from here)
    // two words behind everything else: largest |Q| of the frame being staged (float bits, LDS atomic max), alternating per frame
    unsigned* qmx = reinterpret_cast<unsigned*>(dAs + (LDSA ? CR * FD : 0));
    // work assignment from the schedule kernel: example n, frames chunk, chunk + nchunks, ... (workgroups are dealt to the
    // examples in proportion to their non-empty frames)
    const int4 job = sched[blockIdx.x];
    const int n = job.x, chunk = job.y, nchunks = job.z;
    if (n < 0) return;

    for (int e = tid; e < FusLay<RT>::QT_FLOATS; e += NT) QnT[e] = 0.f;
#if FUS_P1_F16
    for (int e = tid; e < Lr * FQLD; e += NT) Qr[e] = 0.f;            // rows a frame does not have keep what an earlier frame left: finite
    if (tid < 6) qmx[tid] = 0u;                                       // qmx[0..1]: |Q|; qmx[2 + 2 par + k]: |dA| (k = 0), |G| (k = 1) of a frame
    int geb[TPW];                                                     // running |G| exponent of this lane's context rows (fus_p1_tile)
#pragma unroll
    for (int s = 0; s < TPW; s++) geb[s] = 0;
    int par = 0, fpar = 0;
#else
    if (cq) for (int e = tid; e < Lr * FLDQ; e += NT) Qr[e] = 0.f;    // rows a frame does not have keep what an earlier frame left: finite
    int geb[TPW] = {};
    int fpar = 0;
#endif
    int upQ = 254;

    // this wave's context tiles (phase 1): slot s -> tile wave + NW*s
    unsigned orel[TPW], orelA[TPW];   // row offset of context row c inside a frame's row block: a*Li*Lqa + w (score maps) / a*LiA*Lqa + w (dA)
    const int ntiles = wave < CT ? (CT - 1 - wave) / NW + 1 : 0;
    const int LiA = fmap ? fmap[(long)N * Li + n] : Li;
    const long afirst = fmap ? (long)fmap[(long)N * Li + N + n] : (long)n * NA * Li;
#pragma unroll
    for (int s = 0; s < TPW; s++) {
        const int c = (wave + NW * s) * 16 + (lane & 15);
        const int cc = c < CR ? c : CR - 1;
        orel[s] = (unsigned)((cc / Lqa) * Li * Lqa + cc % Lqa);
        orelA[s] = (unsigned)((cc / Lqa) * LiA * Lqa + cc % Lqa);
    }
    f32x4 dcn[TPW][8];
#pragma unroll
    for (int s = 0; s < TPW; s++)
#pragma unroll
        for (int dt = 0; dt < 8; dt++) dcn[s][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    // staging map: a wave-step covers 8 region rows x 8 sixteen-byte pieces (128 contiguous bytes per row); 8 rows per
    // 32-lane half keep the transposed ds_write_b32 of Qn^T at 2-way bank conflicts (free) instead of 16-way
    constexpr int UNITS = RT * 2 * 4, NST = (UNITS + NW - 1) / NW;
    for (int i = chunk; i < Li; i += nchunks) {
        const long frame = (long)n * Li + i;
        // columns to process: up to the last valid region; with a gradient on raw_s the per-frame table of fus_ext_scan_kernel
        // (the last valid region, or Lr when that gradient is non-zero in a column that would be skipped)
        int nvalid;
        if (HAS_EXT) {
            nvalid = (int)fnv[frame];
        } else {
            const float mv = lane < Lr ? qmask[frame * Lr + lane] : 0.f;
            const unsigned long long bal = __ballot(mv != 0.f);
            nvalid = bal ? 64 - __builtin_clzll(bal) : 0;
        }
        long qrow0 = frame * Lr;                 // first row of the frame in Q / Qn / dQraw / dQn
        int Lrs = Lr;                            // rows it has
        if (cq) {
            const int2 qd = cq[frame];
            qrow0 = __builtin_amdgcn_readfirstlane(qd.x);
            Lrs = __builtin_amdgcn_readfirstlane(qd.y);
        }
        if (nvalid == 0) {   // uniform over the workgroup: no region contributes, both gradients of the frame are zero
            for (int e = tid; e < Lrs * 32; e += NT) {
                st4(dQraw + qrow0 * FD + 4 * e, f4zero());
                st4(dQn + qrow0 * FD + 4 * e, f4zero());
            }
            continue;
        }
        // ---- stage the frame's regions ----
#if FUS_P1_F16
        {
            float4 vq[NST], vn[NST];
            float qm = 0.f;
#pragma unroll
            for (int s = 0; s < NST; s++) {
                const int u = wave + NW * s;
                const int r = 8 * (u >> 2) + (lane & 7), q = 8 * (u & 3) + (lane >> 3);
                if (u < UNITS && r < Lrs) {
                    vq[s] = ldv4(Q + (qrow0 + r) * FD + 4 * q);
                    vn[s] = ldv4(Qn + (qrow0 + r) * FD + 4 * q);
                } else {
                    vq[s] = vn[s] = f4zero();
                }
                qm = h_amax3(h_amax3(qm, vq[s].x, vq[s].y), vq[s].z, vq[s].w);
            }
            qm = wave_max(qm);
            if (lane == 0) atomicMax(&qmx[par], __float_as_uint(qm));     // (non-negative floats order like their bit patterns)
            __syncthreads();
            upQ = h_up_field((int)(qmx[par] >> 23) & 0xff);
            if (tid == 0) {                                                // the next frame's words: last read before the barrier above
                qmx[par ^ 1] = 0u;
                qmx[2 + 2 * (par ^ 1)] = 0u;
                qmx[3 + 2 * (par ^ 1)] = 0u;
            }
            fpar = par;
            par ^= 1;
            const float sq = __uint_as_float((unsigned)upQ << 23), sn = __uint_as_float((unsigned)FUS_UP_QN << 23);
            char* Qp = reinterpret_cast<char*>(Qr);
            unsigned short* Th = reinterpret_cast<unsigned short*>(QnT);
            unsigned short* Tl = Th + FD * LT;
#pragma unroll
            for (int s = 0; s < NST; s++) {
                const int u = wave + NW * s;
                const int r = 8 * (u >> 2) + (lane & 7), q = 8 * (u & 3) + (lane >> 3);
                if (u < UNITS && r < Lrs) {
                    uint2 qh, ql, nh, nl;
                    h_split2(vq[s].x, vq[s].y, sq, qh.x, ql.x);
                    h_split2(vq[s].z, vq[s].w, sq, qh.y, ql.y);
                    h_split2(vn[s].x, vn[s].y, sn, nh.x, nl.x);
                    h_split2(vn[s].z, vn[s].w, sn, nh.y, nl.y);
                    *reinterpret_cast<uint2*>(Qp + (r * FQLD + 4 * q) * 2) = qh;
                    *reinterpret_cast<uint2*>(Qp + (Lr * FQLD + r * FQLD + 4 * q) * 2) = ql;
                    Th[(4 * q + 0) * LT + r] = (unsigned short)(nh.x & 0xffffu);
                    Th[(4 * q + 1) * LT + r] = (unsigned short)(nh.x >> 16);
                    Th[(4 * q + 2) * LT + r] = (unsigned short)(nh.y & 0xffffu);
                    Th[(4 * q + 3) * LT + r] = (unsigned short)(nh.y >> 16);
                    Tl[(4 * q + 0) * LT + r] = (unsigned short)(nl.x & 0xffffu);
                    Tl[(4 * q + 1) * LT + r] = (unsigned short)(nl.x >> 16);
                    Tl[(4 * q + 2) * LT + r] = (unsigned short)(nl.y & 0xffffu);
                    Tl[(4 * q + 3) * LT + r] = (unsigned short)(nl.y >> 16);
                }
            }
        }
#else
#pragma unroll
        for (int s = 0; s < NST; s++) {
            const int u = wave + NW * s;
            const int r = 8 * (u >> 2) + (lane & 7), q = 8 * (u & 3) + (lane >> 3);
            if (u < UNITS && r < Lrs) {
                const float4 vq = ldv4(Q + (qrow0 + r) * FD + 4 * q);
                const float4 vn = ldv4(Qn + (qrow0 + r) * FD + 4 * q);
                st4(&Qr[r * FLDQ + 4 * q], vq);
                QnT[(4 * q + 0) * LT + r] = vn.x;
                QnT[(4 * q + 1) * LT + r] = vn.y;
                QnT[(4 * q + 2) * LT + r] = vn.z;
                QnT[(4 * q + 3) * LT + r] = vn.w;
            }
        }
#endif
        TICK(0);
        __syncthreads();
        TICK(1);
        const int nrt = (nvalid + 15) >> 4;
        int aslot = i;
        if (fmap) {
            aslot = __builtin_amdgcn_readfirstlane(fmap[frame]);
            aslot = aslot < 0 ? LiA - 1 : aslot;
        }
        const long arow0 = (afirst + aslot) * Lqa;
#define FUS_FRAME(NRTV)                                                                                                  \
    fus_frame<RT, (NRTV) <= RT ? (NRTV) : RT, NW, HAS_EXT, (OCC == 2 && NW == 8), LDSA, TD>(dA, ext, Cn, Sn, Qr, QnT, Gs, dAs, dQraw, dQn, frame, n, i, NA, Li, \
                                                         Lqa, Lr, scale, orel, ntiles, dcn, wave, lane, tim, tacc, tlast, orelA, arow0, LiA, qrow0, Lrs, upQ, geb, qmx + 2 + 2 * fpar)
        if (RT == 1 || nrt == 1) FUS_FRAME(1);
        else if (RT == 2 || nrt == 2) FUS_FRAME(2);
        else if (RT == 3 || nrt == 3) FUS_FRAME(3);
        else FUS_FRAME(4);
#undef FUS_FRAME
        TICK(4);
        // no barrier here: the next frame's staging writes Qr / Qn^T (last read before the phase barrier above), G is only
        // overwritten after the next staging barrier
    }
    TICK(0);
    // dCn slab of this workgroup: [workgroup][c][d] (the slabs of one example are consecutive), lane (c15, g) holds d = dt*16 + 4g .. +3 of context row c15
#pragma unroll
    for (int s = 0; s < TPW; s++) {
        const int c = (wave + NW * s) * 16 + (lane & 15);
        if (s < ntiles && c < CR) {
            float* dst = part + ((size_t)blockIdx.x * CR + c) * FD + 4 * (lane >> 4);
#if FUS_P1_F16
            const int de = 254 - FUS_UP_QN - h_up_field(geb[s]);          // back to true units
#pragma unroll
            for (int dt = 0; dt < 8; dt++)
#pragma unroll
                for (int k = 0; k < 4; k++) dcn[s][dt][k] = __builtin_ldexpf(dcn[s][dt][k], de);
#endif
#pragma unroll
            for (int dt = 0; dt < 8; dt++)
                st4(dst + dt * 16, make_float4(dcn[s][dt][0], dcn[s][dt][1], dcn[s][dt][2], dcn[s][dt][3]));
        }
    }
    TICK(5);
    if (tim && lane == 0) for (int ph = 0; ph < 6; ph++) atomicAdd(tim + ph, tacc[ph]);
}

// dCn[n][c][:] = sum over the slabs of example n (workgroups first .. first + count - 1, fixed order: deterministic)
__global__ __launch_bounds__(256) void fus_slab_sum_kernel(const float* __restrict__ part, const int2* __restrict__ per_n,
                                                           float* __restrict__ out, int N, long C4n) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)N * C4n) return;
    const int n = (int)(e / C4n);
    const long ce = e - (long)n * C4n;
    const int2 fc = per_n[n];
    float4 acc = f4zero();
    for (int b = 0; b < fc.y; b++) acc = f4add(acc, ld4s(part + ((size_t)(fc.x + b) * C4n + ce) * 4));
    st4(out + e * 4, acc);
}

// Gradient on raw_s (dS_raw_ext): which columns of a frame must be processed.  Region tiles behind the last valid region are
// skipped by the main kernel (P = 0 there, so the softmax part of dS is exactly 0) -- unless the gradient on raw_s is
// non-zero in them: the scores of padded regions are cos - 1e10, d/dcos = 1, so such a gradient reaches Qn and Cn like any
// other.  One workgroup per frame scans the columns that would be skipped (thread c = context row c) and writes
// fnv[frame] = last valid region + 1, or Lr on a hit.  The supervised-attention loss only touches labelled (valid)
// regions: no hits, empty frames and padded tiles stay skipped, and the scan reads about a third of the tensor.
__global__ __launch_bounds__(256) void fus_ext_scan_kernel(const float* __restrict__ qmask, const float* __restrict__ ext,
                                                           unsigned char* __restrict__ fnv, int NA, int Li, int Lqa, int Lr) {
    const long frame = blockIdx.x;                 // n*Li + i
    const int n = (int)(frame / Li), i = (int)(frame % Li), tid = threadIdx.x, CR = NA * Lqa;
    const float mv = tid < Lr ? qmask[frame * Lr + tid] : 0.f;      // Lr <= 64: wave 0 holds the whole mask row
    const unsigned long long bal = __ballot(mv != 0.f);
    __shared__ int nv_sh;
    if (tid == 0) nv_sh = bal ? 64 - __builtin_clzll(bal) : 0;
    __syncthreads();
    const int nv = nv_sh, thr = ((nv + 15) >> 4) << 4;
    int hit = 0;
    if (thr < Lr && tid < CR) {
        const float* er = ext + ((((long)n * NA + tid / Lqa) * Li + i) * Lqa + tid % Lqa) * Lr;
        for (int col = thr; col < Lr; col += 2) {
            const float2 v = ld2(er + col);
            hit |= (v.x != 0.f) | (v.y != 0.f);
        }
    }
    hit = __syncthreads_or(hit);
    if (tid == 0) fnv[frame] = (unsigned char)(hit ? Lr : nv);
}

// Schedule (one workgroup, runs in front of the main kernel): example n gets W_n of the G workgroups, W_n ~ G * V_n / sum V
// (V_n = frames of n that are not skipped), at least one.  With one workgroup
// per CU and equal shares an example with 300 valid frames ran 1.5x longer than one with 200 and the CUs of the short ones
// idled (~17 % of the kernel at the synthetic TVQA+ length distribution).  Output: sched[b] = (n, chunk, W_n, 0) for
// workgroup b (n = -1: unused), per_n[n] = (first workgroup, W_n).  Depends on the masks only: run-to-run deterministic.
// Schedule, step 1: frames with a valid region per example (one workgroup per example; one thread per frame, all Lr mask values of a
// frame requested at once -- Lr is even: 8-byte loads).  The count lands in per_n[n].x for step 2.  (One 1024-thread workgroup used to
// do both steps: 23 us at the video shape, 70 us at the subtitle shape -- 10 % of the backward -- most of it the scan of N Li Lr mask
// values through one CU and thread 0 writing the G schedule entries one by one.)
__global__ __launch_bounds__(256) void fus_count_kernel(const float* __restrict__ qmask, const unsigned char* __restrict__ fnv, int Li, int Lr,
                                                        int2* __restrict__ per_n) {
    __shared__ int red[4];
    const int n = blockIdx.x;
    int cnt = 0;
    for (int i = threadIdx.x; i < Li; i += 256) {
        const long f = (long)n * Li + i;
        float nz = 0.f;
        if (fnv) {                      // gradient on raw_s: the per-frame column counts of fus_ext_scan_kernel
            nz = (float)fnv[f];
        } else {
            const float2* qm = reinterpret_cast<const float2*>(qmask + f * Lr);
#pragma unroll 8
            for (int r = 0; r < (Lr >> 1); r++) { const float2 v = qm[r]; nz += fabsf(v.x) + fabsf(v.y); }
        }
        cnt += nz != 0.f ? 1 : 0;
    }
    cnt = (int)wave_sum((float)cnt);    // (counts <= Li: exact in fp32)
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) per_n[n] = make_int2(red[0] + red[1] + red[2] + red[3], 0);
}

// Schedule, step 2 (one workgroup): workgroups per example in proportion to its non-empty frames, then the G entries in parallel
__global__ __launch_bounds__(256) void fus_schedule_kernel(int N, int Li, int G, int4* __restrict__ sched, int2* __restrict__ per_n) {
    extern __shared__ int sh[];        // V[N], W[N], first[N + 1]
    int* V = sh;
    int* W = sh + N;
    int* first = sh + 2 * N;
    for (int n = threadIdx.x; n < N; n += blockDim.x) V[n] = per_n[n].x;
    __syncthreads();
    if (threadIdx.x == 0) {
        long total = 0;
        for (int n = 0; n < N; n++) total += V[n];
        const int R = G - N;            // one workgroup each, the rest in proportion (G >= N by construction)
        int given = 0;
        for (int n = 0; n < N; n++) {
            const int q = total > 0 ? (int)(((long)R * V[n]) / total) : 0;
            W[n] = 1 + q;
            given += q;
        }
        // remainders: one more to the examples with the most frames per workgroup until the budget is used
        for (int left = R - given; left > 0 && total > 0; left--) {
            int best = 0;
            float load = -1.f;
            for (int n = 0; n < N; n++) {
                const float l = (float)V[n] / (float)W[n];
                if (l > load) { load = l; best = n; }
            }
            W[best]++;
        }
        int f0 = 0;
        for (int n = 0; n < N; n++) {
            if (W[n] > Li) W[n] = Li;
            first[n] = f0;
            f0 += W[n];
        }
        first[N] = f0;
    }
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += blockDim.x) per_n[n] = make_int2(first[n], W[n]);
    for (int b = threadIdx.x; b < G; b += blockDim.x) {
        int n = -1;
        for (int m = 0; m < N; m++)
            if (b >= first[m] && b < first[m + 1]) n = m;
        sched[b] = n >= 0 ? make_int4(n, b - first[n], W[n], 0) : make_int4(-1, 0, 1, 0);
    }
}

static int fus_num_wgs(int N, int Li = 1 << 20) {
    static int cus = 0;
    if (!cus) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cus = v;
    }
    static const int env_wgs = getenv("STAGE_K1_BWD_WGS") ? atoi(getenv("STAGE_K1_BWD_WGS")) : 0;   // developer switch
    int G = env_wgs > 0 ? env_wgs : cus;      // one workgroup per CU (256 registers, up to 157 KB of LDS)
    if (G > FUS_MAX_WGS) G = FUS_MAX_WGS;
    const long few = ((long)N * Li + 3) / 4;          // small problems: at least ~4 frames per workgroup (per-workgroup slab + set-up)
    if (G > few) G = (int)few;
    return G > N ? G : N;
}

template <int RT, int NW, int OCC, typename TD>
static int fus_launch(const TD* dA, const float* ext, const float* Cn, const TD* Q, const TD* Qn, const float* Sn,
                      const float* qmask, float* dQraw, float* dQn, float* dCn, int N, int NA, int Li, int Lqa, int Lr,
                      float scale, void* ws, hipStream_t st, const int* fmap, const int* cq) {
    const int CR = NA * Lqa;
    const int G = fus_num_wgs(N, Li);
    int4* sched = (int4*)ws;
    int2* per_n = (int2*)(sched + G);
    unsigned char* fnv = (unsigned char*)ws + FUS_TABLE_BYTES(G, N);
    float* part = (float*)((char*)ws + FUS_TABLE_BYTES(G, N) + FUS_FRAME_BYTES(N, Li));
    if (ext) {
        hipLaunchKernelGGL(fus_ext_scan_kernel, dim3((unsigned)((long)N * Li)), dim3(256), 0, st, qmask, ext, fnv, NA, Li, Lqa, Lr);
        STAGE_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(fus_count_kernel, dim3(N), dim3(256), 0, st, qmask, ext ? (const unsigned char*)fnv : nullptr, Li, Lr, per_n);
    STAGE_LAUNCH_CHECK();
    hipLaunchKernelGGL(fus_schedule_kernel, dim3(1), dim3(256), (3 * N + 1) * sizeof(int), st, N, Li, G, sched, per_n);
    STAGE_LAUNCH_CHECK();
    const size_t base = ((size_t)Lr * (FUS_P1_F16 ? FQLD : FLDQ) + FusLay<RT>::QT_FLOATS + (size_t)CR * FusLay<RT>::LG) * sizeof(float) + 32;
    const size_t with_da = base + (size_t)CR * FD * sizeof(float);
    // dA of a frame stays in LDS between the phases when it fits (the video shape) and the uniform phase 2 applies
    static const bool no_ldsa = getenv("STAGE_K1_BWD_NOLDSA") != nullptr;   // developer switch
    const bool unif = (Lqa & 3) == 0 && ((CR >> 2) % 10) == 0;
    const bool ldsa = !no_ldsa && unif && NW == 8 && OCC == 2 && with_da <= 160 * 1024;
    const size_t lds = ldsa ? with_da : base;
    const dim3 grid(G), block(64 * NW);
    unsigned long long* tim = (unsigned long long*)(getenv("STAGE_K1_BWD_TIM") ? strtoull(getenv("STAGE_K1_BWD_TIM"), 0, 0) : 0ull);
#define FUS_GO(EXTV, LDSAV)                                                                                                     \
    do {                                                                                                                        \
        if (lds > 64 * 1024)                                                                                                    \
            (void)hipFuncSetAttribute((const void*)str_attn_bwd_fused_kernel<RT, NW, EXTV, OCC, LDSAV, TD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((str_attn_bwd_fused_kernel<RT, NW, EXTV, OCC, LDSAV, TD>), grid, block, lds, st, dA, ext, Cn, Q, Qn, Sn, qmask, \
                           dQraw, dQn, part, N, NA, Li, Lqa, Lr, scale, (const int4*)sched, (const unsigned char*)fnv, tim, fmap,                          \
                           (const int2*)cq);                                                                                                               \
    } while (0)
    if (ext) { if (ldsa) FUS_GO(true, (NW == 8 && OCC == 2)); else FUS_GO(true, false); }
    else { if (ldsa) FUS_GO(false, (NW == 8 && OCC == 2)); else FUS_GO(false, false); }
#undef FUS_GO
    STAGE_LAUNCH_CHECK();
    const long C4n = (long)CR * (FD / 4);
    hipLaunchKernelGGL(fus_slab_sum_kernel, dim3((unsigned)(((long)N * C4n + 255) / 256)), dim3(256), 0, st, (const float*)part,
                       (const int2*)per_n, dCn, N, C4n);
    STAGE_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t stage_str_attn_bwd_fused_ws_bytes(int N, int NA, int Li, int Lqa, int D) {
    const int G = fus_num_wgs(N);     // schedule tables + per-frame column counts + one dCn slab per workgroup
    return FUS_TABLE_BYTES(G, N) + FUS_FRAME_BYTES(N, Li) + (size_t)G * NA * Lqa * D * sizeof(float);
}

template <typename TD>
static int str_attn_bwd_fused_t(const TD* dA, const float* dS_raw_ext, const float* Cn, const TD* Q, const TD* Qn,
                                const float* S_norm, const float* q_mask, float* dQraw, float* dQn, float* dCn, int N, int NA,
                                int Li, int Lqa, int Lr, int D, float scale, void* ws, size_t ws_bytes, void* stream,
                                const int* fmap = nullptr, const int* cq = nullptr) {
    if (N <= 0 || Li <= 0) return 0;
    if (D != FD || Lr < 2 || Lr > 64 || (Lr & 1) || Lqa < 4 || NA < 1 || NA * Lqa > 256 ||
        (long)NA * (Li + 1) * Lqa * FD >= (1l << 29)) return STAGE_ERR_SHAPE;   // 32-bit element offsets inside an example (+1: dump slot)
    if (ws_bytes < stage_str_attn_bwd_fused_ws_bytes(N, NA, Li, Lqa, D)) return STAGE_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int RT = (Lr + 15) / 16;
#define FUS_ARGS dA, dS_raw_ext, Cn, Q, Qn, S_norm, q_mask, dQraw, dQn, dCn, N, NA, Li, Lqa, Lr, scale, ws, st, fmap, cq
    switch (RT) {
        case 1: return fus_launch<1, 8, 2, TD>(FUS_ARGS);
        case 2: return fus_launch<2, 8, 2, TD>(FUS_ARGS);
        case 3: return fus_launch<3, 8, 2, TD>(FUS_ARGS);
        default: return fus_launch<4, 8, 2, TD>(FUS_ARGS);
    }
#undef FUS_ARGS
}

extern "C" int stage_str_attn_bwd_fused(const float* dA, const float* dS_raw_ext, const float* Cn, const float* Q,
                                        const float* Qn, const float* S_norm, const float* q_mask, float* dQraw,
                                        float* dQn, float* dCn, int N, int NA, int Li, int Lqa, int Lr, int D, float scale,
                                        void* ws, size_t ws_bytes, void* stream) {
    return str_attn_bwd_fused_t<float>(dA, dS_raw_ext, Cn, Q, Qn, S_norm, q_mask, dQraw, dQn, dCn, N, NA, Li, Lqa, Lr, D, scale, ws,
                                       ws_bytes, stream);
}

// dA frame-compact (ragged token rows, include/stage_hip.h): the rows of dA are addressed through `fmap`; everything else as above
extern "C" int stage_str_attn_bwd_fused_fc(const float* dA_fc, const float* dS_raw_ext, const float* Cn, const float* Q,
                                           const float* Qn, const float* S_norm, const float* q_mask, float* dQraw,
                                           float* dQn, float* dCn, const int* fmap, const int* cq, int N, int NA, int Li, int Lqa, int Lr,
                                           int D, float scale, void* ws, size_t ws_bytes, void* stream) {
    if (!fmap) return STAGE_ERR_SHAPE;
    return str_attn_bwd_fused_t<float>(dA_fc, dS_raw_ext, Cn, Q, Qn, S_norm, q_mask, dQraw, dQn, dCn, N, NA, Li, Lqa, Lr, D, scale, ws,
                                       ws_bytes, stream, fmap, cq);
}

// bf16 storage mode: dA, Q, Qn are bf16; the score maps, Cn and the three gradients (dQraw, dQn, dCn) stay fp32
extern "C" int stage_str_attn_bwd_fused_bf16(const void* dA, const float* dS_raw_ext, const float* Cn, const void* Q,
                                             const void* Qn, const float* S_norm, const float* q_mask, float* dQraw,
                                             float* dQn, float* dCn, int N, int NA, int Li, int Lqa, int Lr, int D, float scale,
                                             void* ws, size_t ws_bytes, void* stream) {
    typedef stage_bf16 B;
    return str_attn_bwd_fused_t<B>((const B*)dA, dS_raw_ext, Cn, (const B*)Q, (const B*)Qn, S_norm, q_mask, dQraw, dQn, dCn, N, NA,
                                   Li, Lqa, Lr, D, scale, ws, ws_bytes, stream);
}
