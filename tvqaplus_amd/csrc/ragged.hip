// Ragged token rows: the index tables behind the live-range execution of the (N, 5, Li, Lqa, .) kernels (include/stage_hip.h,
// DESIGN.md "ragged token rows").
//
// The reference computes every padded row of the (N, 5, Li, Lqa, D) tensors (model/stage.py:365-387, 276-279, 484-505: no masking
// inside LayerNorm / Linear / the convolutions of model/encoder.py:35-52).  What reaches an output or a gradient is less:
//   * model/stage.py:503 takes the max over the words of  statement * mask + (1 - mask) * -1e10 ; a frame whose mask is all zero
//     comes out as the constant -1e10 and its gradient is dout * mask = 0: NOTHING of such a frame is ever used -- dead frame;
//   * of a live frame only the words w < Lv (last valid word + 1) are used, and the classifier encoder's depthwise convolutions
//     (n_conv layers of width k, no mask: model/cnn.py:42-47) let words up to Lv + n_conv * (k / 2) - 1 leak into them, forward
//     and backward -- the halo.  Words behind Lc = min(Lqa, Lv + halo) are dead.
// The host (tvqaplus_amd/ragged.py) turns the masks into small int32 tables; the kernels here expand them on the device:
//   compact rows        [group g = (n, a)] [live frame] [word < Lc(g)]                     every row kernel behind the attention
//   frame-compact rows  [sequence = first(n) + a * slots(n) + slot] [word < Lqa]           A / dA of the attention kernels
//                       (slots(n) = live frames of example n + 1 dump slot that dead frames write to / read zeros from)
#include "common.h"
#include "../../include/stage_hip.h"

namespace {

// rowinfo[row] = (QA row g * Lqa + w, frame-compact row, dense output row g * Li + i, w) for every compact row
__global__ __launch_bounds__(256) void rag_rowinfo_kernel(const int4* __restrict__ seq, const int* __restrict__ seqfc, long S, int Lqa,
                                                          int4* __restrict__ rowinfo) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long s = t / Lqa;
    const int l = (int)(t - s * Lqa);
    if (s >= S) return;
    const int4 sq = seq[s];
    if (l >= sq.y) return;
    rowinfo[(long)sq.x + l] = make_int4(sq.z * Lqa + l, seqfc[s] * Lqa + l, sq.w, l);
}

// pooled output of the frames that have no group: what the dense kernel computes for an all-masked group (the constant -1e10, first row)
__global__ __launch_bounds__(256) void rag_fill_kernel(float4* __restrict__ out, int4* __restrict__ idx, long n4) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n4) return;
    out[e] = make_float4(STAGE_NEG, STAGE_NEG, STAGE_NEG, STAGE_NEG);
    idx[e] = make_int4(0, 0, 0, 0);
}

// zero the dump slot of every (example, candidate) in a frame-compact (., D) tensor: the backward reads the rows of dead frames there
__global__ __launch_bounds__(256) void rag_zero_dump_kernel(float4* __restrict__ A, const int* __restrict__ fmap, int N, int NA, int Li,
                                                            int Lqa, int D4) {
    const int g = blockIdx.x, n = g / NA, a = g - n * NA;
    const int slots = fmap[(long)N * Li + n], first = fmap[(long)N * Li + N + n];
    float4* dst = A + ((long)(first + a * slots + slots - 1) * Lqa) * D4;
    for (int e = threadIdx.x; e < Lqa * D4; e += blockDim.x) dst[e] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// source row (in the padded (frames, L, .) feature tensor) of every compact context row: frame f holds rows cq[f].x .. + cq[f].y - 1
__global__ __launch_bounds__(256) void rag_ctx_rows_kernel(const int2* __restrict__ cq, long frames, int L, int* __restrict__ src) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long f = t / L;
    const int l = (int)(t - f * L);
    if (f >= frames) return;
    const int2 q = cq[f];
    if (l < q.y) src[(long)q.x + l] = (int)(f * L + l);
}

}  // namespace

extern "C" int stage_rag_ctx_rows(const int* cq, long long frames, int L, int* src_rows, void* stream) {
    if (frames <= 0) return 0;
    if (L < 1 || frames * L >= (1ll << 31)) return STAGE_ERR_SHAPE;
    const long total = (long)frames * L;
    hipLaunchKernelGGL(rag_ctx_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const int2*)cq,
                       (long)frames, L, src_rows);
    STAGE_LAUNCH_CHECK();
    return 0;
}

extern "C" int stage_rag_rowinfo(const int* seq, const int* seqfc, long long S, int Lqa, int* rowinfo, void* stream) {
    if (S <= 0) return 0;
    if (Lqa < 1) return STAGE_ERR_SHAPE;
    const long total = (long)S * Lqa;
    hipLaunchKernelGGL(rag_rowinfo_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const int4*)seq, seqfc,
                       (long)S, Lqa, (int4*)rowinfo);
    STAGE_LAUNCH_CHECK();
    return 0;
}

extern "C" int stage_rag_fill_pooled(float* out, int* argmax, long long rows, int D, void* stream) {
    if (rows <= 0) return 0;
    if (D % 4) return STAGE_ERR_SHAPE;
    const long n4 = (long)rows * (D / 4);
    hipLaunchKernelGGL(rag_fill_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (float4*)out, (int4*)argmax, n4);
    STAGE_LAUNCH_CHECK();
    return 0;
}

extern "C" int stage_rag_zero_dump(float* A_fc, const int* fmap, int N, int NA, int Li, int Lqa, int D, void* stream) {
    if (N <= 0 || NA <= 0) return 0;
    if (D % 4 || !fmap) return STAGE_ERR_SHAPE;
    hipLaunchKernelGGL(rag_zero_dump_kernel, dim3((unsigned)(N * NA)), dim3(256), 0, (hipStream_t)stream, (float4*)A_fc, fmap, N, NA, Li, Lqa,
                       D / 4);
    STAGE_LAUNCH_CHECK();
    return 0;
}
