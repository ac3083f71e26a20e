// Issue rate of v_mfma_f32_16x16x4_f32 / v_mfma_f32_4x4x1_16b_f32: 256 workgroups of W waves (W = 4: one wave per SIMD, 8: two),
// each wave runs `iters` x 16 MFMAs on C independent accumulators (C = 1: a fully dependent chain, 2, 4).
// hipcc --offload-arch=gfx950 -O3 -o mfma_f32 mfma_f32.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int C, bool SMALL>
__global__ void k(float* out, int iters) {
    extern __shared__ float pin[];
    f4 acc[C];
    for (int i = 0; i < C; i++) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16 / C; r++)
#pragma unroll
            for (int i = 0; i < C; i++)
                acc[i] = SMALL ? __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < C; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 1.2345e30f) out[threadIdx.x] = s + pin[0];
}
template <int C, bool SMALL>
static float run(float* d, int waves, int iters) {
    hipFuncSetAttribute((const void*)k<C, SMALL>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    k<C, SMALL><<<256, 64 * waves, 100 * 1024>>>(d, 10);
    hipEventRecord(s);
    k<C, SMALL><<<256, 64 * waves, 100 * 1024>>>(d, iters);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    return ms * 1e6f / ((float)iters * 16);   // ns per MFMA of one wave
}
int main() {
    float* d; hipMalloc(&d, 4096);
    const int iters = 20000;
    printf("ns per MFMA per wave (2.2 GHz: 32 cycles = 14.5 ns, 8 cycles = 3.6 ns)\n");
    printf("16x16x4  1 wave/SIMD: chain %.1f  2 acc %.1f  4 acc %.1f | 2 waves/SIMD: chain %.1f  2 acc %.1f  4 acc %.1f\n",
           run<1, false>(d, 4, iters), run<2, false>(d, 4, iters), run<4, false>(d, 4, iters),
           run<1, false>(d, 8, iters), run<2, false>(d, 8, iters), run<4, false>(d, 8, iters));
    printf("4x4x1    1 wave/SIMD: chain %.1f  2 acc %.1f  4 acc %.1f | 2 waves/SIMD: chain %.1f  2 acc %.1f  4 acc %.1f\n",
           run<1, true>(d, 4, iters), run<2, true>(d, 4, iters), run<4, true>(d, 4, iters),
           run<1, true>(d, 8, iters), run<2, true>(d, 8, iters), run<4, true>(d, 8, iters));
    return 0;
}
