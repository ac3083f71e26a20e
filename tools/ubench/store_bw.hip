// Micro-benchmark: pure-store bandwidth of the K1 output pattern vs a fully coalesced stream (same bytes).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ __launch_bounds__(256) void k_pattern(float* A, long rows, int mode, int iters_per_wave) {
    const int lane = threadIdx.x & 63, c15 = lane & 15, g = lane >> 4;
    const long nw = (long)gridDim.x * 4, w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    float4 v = make_float4(1.f, 2.f, 3.f, (float)lane);
    // each "tile" = 16 rows x 128 floats (8 KB)
    for (long tile = w; tile < rows / 16; tile += nw) {
        float* base = A + tile * 16 * 128;
#pragma unroll
        for (int dt = 0; dt < 8; dt++) {
            if (mode == 0) *(float4*)(base + c15 * 128 + dt * 16 + 4 * g) = v;      // K1 pattern: 16 rows x 64 B per instr
            else if (mode == 1) *(float4*)(base + dt * 256 + lane * 4) = v;         // 1 KB contiguous per instr
            else *(float4*)(base + (dt >> 1) * 512 + (lane >> 5) * 128 * 0 + ((dt & 1) * 64 + lane) * 4 % 512 + ((dt>>1)*0)) = v;
        }
    }
}
__global__ __launch_bounds__(256) void k_stream(float4* A, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) A[i] = make_float4(1, 2, 3, 4);
}
int main() {
    const long rows = 16L * 5 * 300 * 40;  // 960000 rows x 128 floats = 491.5 MB
    float* A; hipMalloc(&A, rows * 128 * 4);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    for (int mode = 0; mode < 2; mode++) for (int blocks : {256, 512, 1024, 2048, 4096}) {
        for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_pattern, dim3(blocks), dim3(256), 0, 0, A, rows, mode, 0);
        hipEventRecord(s); for (int r = 0; r < 10; r++) hipLaunchKernelGGL(k_pattern, dim3(blocks), dim3(256), 0, 0, A, rows, mode, 0);
        hipEventRecord(e); hipEventSynchronize(e); float ms; hipEventElapsedTime(&ms, s, e);
        printf("mode %d blocks %4d: %.1f us  %.2f TB/s\n", mode, blocks, ms * 100, rows * 512.0 / (ms / 10 * 1e-3) / 1e12);
    }
    for (int blocks : {1024, 4096, 16384}) {
        for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(256), 0, 0, (float4*)A, rows * 32);
        hipEventRecord(s); for (int r = 0; r < 10; r++) hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(256), 0, 0, (float4*)A, rows * 32);
        hipEventRecord(e); hipEventSynchronize(e); float ms; hipEventElapsedTime(&ms, s, e);
        printf("stream blocks %5d: %.1f us  %.2f TB/s\n", blocks, ms * 100, rows * 512.0 / (ms / 10 * 1e-3) / 1e12);
    }
    return 0;
}
