// K1 backward, the two reduction GEMMs on the matrix cores (D % 64 == 0; other widths use the scalar kernels in
// str_attn.hip).  Derivation in str_attn.hip: with P = S_ (saved), G = dS (from the ds kernel),
//     dQraw[n,i,r,:] = sum_c P[c,r] dA[c,:]         dQn[n,i,r,:] = sum_c G[c,r] Cn[n,c,:]        (c = NA*Lqa context rows)
//     dCn[n,c,:]     = sum_i sum_r G[(c,i),r] Qn[n,i,r,:]
// Both are tall-skinny products whose operands already sit in HBM in a layout the MFMA can take directly:
//   * A operand (16 x 4, one f32 per lane) = a 4-row x 16-column patch of P / G  (one dword per lane)
//   * B operand (4 x 16) = 4 rows of dA / Cn / Qn.  Column j of output tile e is mapped to d = 64 b + 4 j + e, so the
//     four tiles e = 0..3 of a 64-wide d block share ONE float4 load per lane, 16 lanes cover 256 contiguous bytes of a
//     row, and a lane ends up holding 4 consecutive d of an output row -> float4 stores.  No LDS, no barriers.
// Fixed summation order (k-steps in order inside a wave; frame chunks reduced by the slab-sum kernel): deterministic.
#include "common.h"
#include "../../include/stage_hip.h"

// ---------------------------------------------------------------------------------------------------------------
// dQraw / dQn: one wave per (frame, 64-wide d block).  RT = region tiles (Lr <= 16*RT).
// ---------------------------------------------------------------------------------------------------------------
// T4: the last region tile holds <= 4 regions (Lr = 20, 50): its 8 MFMAs per k-step run as v_mfma_f32_4x4x1 blocks (8 cycles
// instead of 32; block (c15 >> 2, g) = 4 regions x 4 output columns x the context row of lane group g) and the four
// partial sums per output are folded ONCE per item by the transpose-reduce of common.h (lane group g keeps region g).
template <int RT, bool T4>
__global__ __launch_bounds__(256) void str_attn_bwd_dq_mfma_kernel(const float* __restrict__ dA, const float* __restrict__ Sn,
                                                                   const float* __restrict__ dS, const float* __restrict__ Cn,
                                                                   float* __restrict__ dQraw, float* __restrict__ dQn,
                                                                   int N, int NA, int Li, int Lqa, int Lr, int D) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int c15 = lane & 15, g = lane >> 4;
    const int CR = NA * Lqa, DB = D >> 6;
    const long n_items = (long)N * Li * DB;
    const long n_waves = (long)gridDim.x * wpb;
    for (long item = (long)blockIdx.x * wpb + wave; item < n_items; item += n_waves) {
        const long frame = item / DB;
        const int b = (int)(item % DB);
        const int n = (int)(frame / Li), i = (int)(frame % Li);
        f32x4 ar[RT][4], an[RT][4];
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
#pragma unroll
            for (int e = 0; e < 4; e++) ar[rt][e] = an[rt][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // this lane's k row: c = 4*ks + g, walked incrementally as (a, w)
        int a = g / Lqa, w = g % Lqa;

        for (int c0 = 0; c0 < CR; c0 += 4) {
            const int c = c0 + g;
            const bool ok = c < CR;
            // clamped addresses, all loads issued back to back, validity applied by selects afterwards: a guarded load
            // (`ok ? load : 0`) is an exec-masked branch with its own wait -- the loads of a k-step would serialise
            const int cc = ok ? c : CR - 1;
            const long orow = ok ? ((long)(n * NA + a) * Li + i) * Lqa + w : ((long)(n * NA + NA - 1) * Li + i) * Lqa + Lqa - 1;
            float p[RT], gs[RT];
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                const int r = (T4 && rt == RT - 1) ? rt * 16 + (c15 & 3) : rt * 16 + c15;
                const int rc = r < Lr ? r : Lr - 1;
                p[rt] = Sn[orow * Lr + rc];
                gs[rt] = dS[orow * Lr + rc];
            }
            float4 da = ld4(dA + orow * D + 64 * b + 4 * c15);
            float4 cn = ld4(Cn + ((long)n * CR + cc) * D + 64 * b + 4 * c15);
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                const int r = (T4 && rt == RT - 1) ? rt * 16 + (c15 & 3) : rt * 16 + c15;
                const bool rok = ok && r < Lr;
                p[rt] = rok ? p[rt] : 0.f;
                gs[rt] = rok ? gs[rt] : 0.f;
            }
            if (!ok) da = cn = f4zero();
            const float dav[4] = {da.x, da.y, da.z, da.w}, cnv[4] = {cn.x, cn.y, cn.z, cn.w};
#pragma unroll
            for (int rt = 0; rt < RT; rt++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    if (T4 && rt == RT - 1) {
                        ar[rt][e] = __builtin_amdgcn_mfma_f32_4x4x1f32(p[rt], dav[e], ar[rt][e], 0, 0, 0);
                        an[rt][e] = __builtin_amdgcn_mfma_f32_4x4x1f32(gs[rt], cnv[e], an[rt][e], 0, 0, 0);
                    } else {
                        ar[rt][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(p[rt], dav[e], ar[rt][e], 0, 0, 0);
                        an[rt][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(gs[rt], cnv[e], an[rt][e], 0, 0, 0);
                    }
                }
            w += 4;
            while (w >= Lqa) { w -= Lqa; a++; }
        }
        if (T4) {
            // register i of a 4x4 block = region base + i, summed over this lane group's context rows only: fold the four
            // groups, lane group g keeps region base + g (column c15 as in the 16x16 layout)
            const int r = (RT - 1) * 16 + g;
            float4 qr, qn;
            float* qrv = &qr.x;
            float* qnv = &qn.x;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                qrv[e] = xsum32(xsum16(ar[RT - 1][e][0], ar[RT - 1][e][1]), xsum16(ar[RT - 1][e][2], ar[RT - 1][e][3]));
                qnv[e] = xsum32(xsum16(an[RT - 1][e][0], an[RT - 1][e][1]), xsum16(an[RT - 1][e][2], an[RT - 1][e][3]));
            }
            if (r < Lr) {
                const long off = (frame * Lr + r) * D + 64 * b + 4 * c15;
                st4(dQraw + off, qr);
                st4(dQn + off, qn);
            }
        }
        // C layout: row i = 4g + reg (region), col j = c15 (-> d = 64b + 4 c15 + e)
#pragma unroll
        for (int rt = 0; rt < (T4 ? RT - 1 : RT); rt++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int r = rt * 16 + 4 * g + reg;
                if (r < Lr) {
                    const long off = (frame * Lr + r) * D + 64 * b + 4 * c15;
                    st4(dQraw + off, make_float4(ar[rt][0][reg], ar[rt][1][reg], ar[rt][2][reg], ar[rt][3][reg]));
                    st4(dQn + off, make_float4(an[rt][0][reg], an[rt][1][reg], an[rt][2][reg], an[rt][3][reg]));
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// dCn partial slabs: one wave per (n, pair of 16-row context tiles, frame chunk); all of D (NB = D/64 blocks) per wave.
// part layout: [chunk][N*CR][D]
// ---------------------------------------------------------------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(256) void str_attn_bwd_dc_mfma_kernel(const float* __restrict__ dS, const float* __restrict__ Qn,
                                                                   float* __restrict__ part, int N, int NA, int Li, int Lqa,
                                                                   int Lr, int frames_per_chunk, int nchunks) {
    constexpr int D = 64 * NB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int c15 = lane & 15, g = lane >> 4;
    const int CR = NA * Lqa, CT = (CR + 15) >> 4, CP = (CT + 1) >> 1;
    const long n_items = (long)N * CP * nchunks;
    const long n_waves = (long)gridDim.x * wpb;
    const int KR = (Lr + 3) >> 2;
    for (long item = (long)blockIdx.x * wpb + wave; item < n_items; item += n_waves) {
        const int chunk = (int)(item % nchunks);
        const int cp = (int)((item / nchunks) % CP);
        const int n = (int)(item / ((long)nchunks * CP));
        const int fa = chunk * frames_per_chunk, fb = min(Li, fa + frames_per_chunk);
        f32x4 acc[2][NB][4];
        long obase[2];   // row base of dS for (c, frame 0): ((n*NA + a)*Li)*Lqa + w ; add i*Lqa per frame
        bool cok[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int c = (2 * cp + u) * 16 + c15;
            cok[u] = c < CR;
            const int cc = cok[u] ? c : 0;
            obase[u] = ((long)(n * NA + cc / Lqa) * Li) * Lqa + cc % Lqa;
#pragma unroll
            for (int b = 0; b < NB; b++)
#pragma unroll
                for (int e = 0; e < 4; e++) acc[u][b][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        // steps s = (frame - fa) * KR + kr, software pipelined: the operands of step s + 1 are requested (straight-line,
        // clamped addresses) before the MFMAs of step s; two register sets, loop unrolled by two
        const int nsteps = (fb - fa) * KR;
        auto fetch = [&](int sidx, float (&a_op)[2], float4 (&q)[NB]) {
            const int sc = sidx < nsteps ? sidx : nsteps - 1;
            const int i = fa + sc / KR, kr = sc % KR;
            const int r = min(4 * kr + g, Lr - 1);
#pragma unroll
            for (int u = 0; u < 2; u++) a_op[u] = dS[(obase[u] + (long)i * Lqa) * Lr + r];
            const float* qrow = Qn + (((long)n * Li + i) * Lr + r) * D + 4 * c15;
#pragma unroll
            for (int b = 0; b < NB; b++) q[b] = ld4(qrow + 64 * b);
        };
        auto mul = [&](int sidx, const float (&a_op)[2], const float4 (&q)[NB]) {
            const bool ok = sidx < nsteps && 4 * (sidx % KR) + g < Lr;
            float av[2];
#pragma unroll
            for (int u = 0; u < 2; u++) av[u] = (ok && cok[u]) ? a_op[u] : 0.f;
#pragma unroll
            for (int b = 0; b < NB; b++) {
                const float qv[4] = {q[b].x, q[b].y, q[b].z, q[b].w};
#pragma unroll
                for (int u = 0; u < 2; u++)
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        acc[u][b][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], qv[e], acc[u][b][e], 0, 0, 0);
            }
        };
        if (nsteps > 0) {
            float a0[2], a1[2];
            float4 q0[NB], q1[NB];
            fetch(0, a0, q0);
            for (int sidx = 0; sidx < nsteps; sidx += 2) {
                fetch(sidx + 1, a1, q1);
                mul(sidx, a0, q0);
                fetch(sidx + 2, a0, q0);
                mul(sidx + 1, a1, q1);
            }
        }
        // C layout: row = 4g + reg (context row inside the tile), col j = c15 -> d = 64b + 4 c15 + e
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int c = (2 * cp + u) * 16 + 4 * g + reg;
                if (c < CR) {
                    float* dst = part + (((size_t)chunk * N * CR) + (size_t)n * CR + c) * D + 4 * c15;
#pragma unroll
                    for (int b = 0; b < NB; b++)
                        st4(dst + 64 * b, make_float4(acc[u][b][0][reg], acc[u][b][1][reg], acc[u][b][2][reg], acc[u][b][3][reg]));
                }
            }
    }
}

int stage_str_attn_bwd_dq_mfma(const float* dA, const float* Sn, const float* dS, const float* Cn, float* dQraw,
                                          float* dQn, int N, int NA, int Li, int Lqa, int Lr, int D, void* stream) {
    if (D % 64 != 0 || Lr > 64) return STAGE_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const long items = (long)N * Li * (D / 64);
    long blocks = (items + 3) / 4;
    if (blocks > 768) blocks = 768;   // ~3 workgroups of 4 waves per CU, waves stride the items
    const dim3 grid((unsigned)blocks), block(256);
    const int rt = (Lr + 15) / 16;
    const bool t4 = Lr - 16 * (rt - 1) <= 4 && !getenv("STAGE_K1_DQ_NO_T4");
#define LAUNCH_DQ(R, T) hipLaunchKernelGGL((str_attn_bwd_dq_mfma_kernel<R, T>), grid, block, 0, st, dA, Sn, dS, Cn, dQraw, dQn, N, NA, Li, Lqa, Lr, D)
    switch (rt) {
        case 1: if (t4) LAUNCH_DQ(1, true); else LAUNCH_DQ(1, false); break;
        case 2: if (t4) LAUNCH_DQ(2, true); else LAUNCH_DQ(2, false); break;
        case 3: if (t4) LAUNCH_DQ(3, true); else LAUNCH_DQ(3, false); break;
        default: if (t4) LAUNCH_DQ(4, true); else LAUNCH_DQ(4, false); break;
    }
#undef LAUNCH_DQ
    STAGE_LAUNCH_CHECK();
    return 0;
}

// part must hold nchunks * N * NA * Lqa * D floats; returns the number of chunks written through *nchunks_out
int stage_str_attn_bwd_dc_mfma(const float* dS, const float* Qn, float* part, int N, int NA, int Li, int Lqa,
                                          int Lr, int D, int max_chunks, int* nchunks_out, void* stream) {
    if (D % 64 != 0 || D > 256) return STAGE_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    int fpc = (Li + max_chunks - 1) / max_chunks;
    if (fpc < 1) fpc = 1;
    const int nchunks = (Li + fpc - 1) / fpc;
    *nchunks_out = nchunks;
    const int CR = NA * Lqa, CT = (CR + 15) / 16, CP = (CT + 1) / 2;
    const long items = (long)N * CP * nchunks;
    long blocks = (items + 3) / 4;
    if (blocks > 768) blocks = 768;
    const dim3 grid((unsigned)blocks), block(256);
    switch (D / 64) {
        case 1: hipLaunchKernelGGL((str_attn_bwd_dc_mfma_kernel<1>), grid, block, 0, st, dS, Qn, part, N, NA, Li, Lqa, Lr, fpc, nchunks); break;
        case 2: hipLaunchKernelGGL((str_attn_bwd_dc_mfma_kernel<2>), grid, block, 0, st, dS, Qn, part, N, NA, Li, Lqa, Lr, fpc, nchunks); break;
        case 3: hipLaunchKernelGGL((str_attn_bwd_dc_mfma_kernel<3>), grid, block, 0, st, dS, Qn, part, N, NA, Li, Lqa, Lr, fpc, nchunks); break;
        default: hipLaunchKernelGGL((str_attn_bwd_dc_mfma_kernel<4>), grid, block, 0, st, dS, Qn, part, N, NA, Li, Lqa, Lr, fpc, nchunks); break;
    }
    STAGE_LAUNCH_CHECK();
    return 0;
}
