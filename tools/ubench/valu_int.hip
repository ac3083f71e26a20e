// Issue cost of the integer VALU instructions a counter-based dropout hash is made of, against v_fma_f32 (CDNA4).
// 256 workgroups x 512 threads (2 waves per SIMD), 8 independent chains per lane, 64 instructions per iteration.
// Prints SIMD cycles per wave-instruction at the measured wall time (2.4 GHz nominal).  hipcc --offload-arch=gfx950 -O3 -o valu_int valu_int.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define OP1(name, asmstr)                                                                     \
    __global__ __launch_bounds__(512, 1) void k_##name(unsigned* out, int iters, unsigned c) { \
        unsigned v[8];                                                                        \
        for (int i = 0; i < 8; i++) v[i] = threadIdx.x * 2654435761u + i * 40503u + c;        \
        unsigned long long w[8];                                                              \
        for (int i = 0; i < 8; i++) w[i] = v[i];                                              \
        for (int it = 0; it < iters; it++) {                                                  \
            _Pragma("unroll") for (int r = 0; r < 8; r++) {                                   \
                _Pragma("unroll") for (int i = 0; i < 8; i++) { asmstr; }                     \
            }                                                                                 \
        }                                                                                     \
        unsigned s = 0;                                                                       \
        for (int i = 0; i < 8; i++) s ^= v[i] ^ (unsigned)w[i] ^ (unsigned)(w[i] >> 32);      \
        if (s == 0x12345u) out[threadIdx.x] = s;                                              \
    }
OP1(mul_lo, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[i]) : "v"(c)))
OP1(mul_hi, asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(v[i]) : "v"(c)))
OP1(mul_u24, asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v[i]) : "v"(c)))
OP1(mad_u24, asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(v[i]) : "v"(c)))
OP1(mad_u64, asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(v[i]), "v"(c) : "vcc"))
OP1(xor_, asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[i]) : "v"(c)))
OP1(lshr, asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(v[i])))
OP1(lshr64, asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(w[i])))
OP1(alignbit, asm volatile("v_alignbit_b32 %0, %0, %0, 13" : "+v"(v[i])))
OP1(xad, asm volatile("v_xad_u32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(c)))
OP1(lshl_add, asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(v[i]) : "v"(c)))
OP1(add3, asm volatile("v_add3_u32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(c)))
OP1(perm, asm volatile("v_perm_b32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(c)))
OP1(bfe, asm volatile("v_bfe_u32 %0, %0, 3, 16" : "+v"(v[i])))
OP1(fma, asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(c)))
OP1(cvt_pk, asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c)))
OP1(cmp_sel, asm volatile("v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(c) : "vcc"))

template <typename K> static void run(const char* name, K kern, unsigned* d, int per_iter_mult) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    const int iters = 4000;
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, d, 10, 7u);
    hipEventRecord(s);
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, d, iters, 7u);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    // per SIMD: 2 waves x iters x 64 instructions
    const double insts = 2.0 * iters * 64 * per_iter_mult;
    printf("%-10s %8.1f us  %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, ms * 1e3, ms * 1e-3 * 2.4e9 / insts);
}
int main() {
    unsigned* d; hipMalloc(&d, 4096);
#define R(n, m) run(#n, k_##n, d, m)
    R(fma, 1); R(xor_, 1); R(lshr, 1); R(lshr64, 1); R(alignbit, 1); R(xad, 1); R(lshl_add, 1); R(add3, 1); R(perm, 1); R(bfe, 1);
    R(cvt_pk, 1); R(cmp_sel, 2); R(mul_lo, 1); R(mul_hi, 1); R(mul_u24, 1); R(mad_u24, 1); R(mad_u64, 1);
    return 0;
}
