// Does the sustained matrix-core rate depend on the operand DATA?  256 workgroups x 8 waves, 16 independent
// v_mfma_f32_32x32x16_bf16 per iteration, operands: (0) constants, (1) random bf16 bit patterns per lane, refreshed from registers
// every MFMA (8 different operand pairs rotate).  hipcc --offload-arch=gfx950 -O3 -o mfma_power mfma_power.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, const unsigned* rnd, int iters) {
    extern __shared__ float pin[];
    bf8 a[4], b[4];
    for (int i = 0; i < 4; i++) {
        u4 ua, ub;
        for (int j = 0; j < 4; j++) {
            const unsigned ra = rnd[(threadIdx.x * 37 + i * 8 + j) & 4095], rb = rnd[(threadIdx.x * 53 + i * 8 + j + 4) & 4095];
            // random mantissa / sign, exponent kept near 1.0 so that nothing overflows
            ua[j] = MODE ? ((ra & 0x807F807Fu) | 0x3F803F80u) : 0x3F803F80u;
            ub[j] = MODE ? ((rb & 0x807F807Fu) | 0x3F803F80u) : 0x3F803F80u;
        }
        a[i] = __builtin_bit_cast(bf8, ua);
        b[i] = __builtin_bit_cast(bf8, ub);
    }
    f16v acc[4];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 16; j++) acc[i][j] = 0.f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + r) & 3], b[(i + 2 * r) & 3], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 16; j++) s += acc[i][j];
    if (s == 1.2345e30f) out[threadIdx.x] = s + pin[0];
}
template <int MODE>
static float run(float* d, unsigned* r, int iters) {
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    k<MODE><<<256, 512, 100 * 1024>>>(d, r, 10);
    hipEventRecord(s);
    k<MODE><<<256, 512, 100 * 1024>>>(d, r, iters);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    return ms * 1e3f;
}
int main() {
    float* d; hipMalloc(&d, 4096);
    unsigned h[4096]; unsigned x = 12345u;
    for (int i = 0; i < 4096; i++) { x = x * 1664525u + 1013904223u; h[i] = x; }
    unsigned* r; hipMalloc(&r, sizeof(h)); hipMemcpy(r, h, sizeof(h), hipMemcpyHostToDevice);
    const int iters = 40000;
    for (int rep = 0; rep < 3; rep++) {
        const float t0 = run<0>(d, r, iters), t1 = run<1>(d, r, iters);
        printf("constant operands %.0f us (%.2f GHz)   random operands %.0f us (%.2f GHz)\n", t0, 2.0 * iters * 512 / (t0 * 1e3), t1,
               2.0 * iters * 512 / (t1 * 1e3));
    }
    return 0;
}
