// Micro-benchmark: what a 960000 x 128 fp32 "read once, write once" kernel can reach on MI355X with different access
// shapes.  mode 0: grid-stride float4 copy; 1: read only (sum); 2: GEMM-tile pattern (128-row tile, 4 chunks of 32 floats
// per row, one chunk in flight, scalar-dword epilogue stores 2 rows x 128 B per instr); 3: same reads, all 4 chunks
// issued up front; 4: tile pattern with full 512-B row reads (float4, 32 lanes per row) and float4 stores.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ X, float4* __restrict__ Y, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) Y[i] = X[i];
}
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ X, float* __restrict__ Y, long n4) {
    float4 a = make_float4(0, 0, 0, 0);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) { float4 v = X[i]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
    if (a.x + a.y + a.z + a.w == 1.2345f) Y[0] = a.x;
}
template <int MODE>
__global__ __launch_bounds__(256, 2) void k_tile(const float* __restrict__ X, float* __restrict__ Y, long M) {
    __shared__ float sm[128 * 33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long m0 = (long)blockIdx.x * 128;
    const int lrow = tid >> 3, lcol = (tid & 7) * 4;
    float acc[64];
#pragma unroll
    for (int i = 0; i < 64; i++) acc[i] = 0.f;
    if (MODE == 2) {
        float4 p[4];
#pragma unroll
        for (int q = 0; q < 4; q++) p[q] = *(const float4*)(X + (m0 + lrow + 32 * q) * 128 + lcol);
        for (int k0 = 0; k0 < 128; k0 += 32) {
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; q++) { float* d = &sm[(lrow + 32 * q) * 33 + lcol]; d[0] = p[q].x; d[1] = p[q].y; d[2] = p[q].z; d[3] = p[q].w; }
            __syncthreads();
            if (k0 + 32 < 128) {
#pragma unroll
                for (int q = 0; q < 4; q++) p[q] = *(const float4*)(X + (m0 + lrow + 32 * q) * 128 + k0 + 32 + lcol);
            }
#pragma unroll
            for (int j = 0; j < 16; j++) acc[(k0 / 32) * 16 + j] += sm[((wave * 32 + (lane & 31)) % 128) * 33 + j + 16 * (lane >> 5)];
        }
    } else if (MODE == 3) {
        float4 p[4][4];
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int q = 0; q < 4; q++) p[c][q] = *(const float4*)(X + (m0 + lrow + 32 * q) * 128 + 32 * c + lcol);
#pragma unroll
        for (int c = 0; c < 4; c++) {
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; q++) { float* d = &sm[(lrow + 32 * q) * 33 + lcol]; d[0] = p[c][q].x; d[1] = p[c][q].y; d[2] = p[c][q].z; d[3] = p[c][q].w; }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 16; j++) acc[c * 16 + j] += sm[((wave * 32 + (lane & 31)) % 128) * 33 + j + 16 * (lane >> 5)];
        }
    }
    if (MODE == 2 || MODE == 3) {
        // epilogue like the GEMM: wave (wm, wn) owns 64x64; per instr 2 rows x 32 floats
        const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
#pragma unroll
        for (int mi = 0; mi < 2; mi++)
#pragma unroll
            for (int ni = 0; ni < 2; ni++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const long m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    Y[m * 128 + wn * 64 + ni * 32 + l31] = acc[(mi * 2 + ni) * 16 + r];
                }
    }
    if (MODE == 4) {
        // full-row pattern: thread -> row tid>>5 (+8 per pass), 16-B column tid&31; 16 passes
        float4 p[16];
#pragma unroll
        for (int q = 0; q < 16; q++) p[q] = *(const float4*)(X + (m0 + (tid >> 5) + 8 * q) * 128 + (tid & 31) * 4);
#pragma unroll
        for (int q = 0; q < 16; q++) { p[q].x += 1.f; *(float4*)(Y + (m0 + (tid >> 5) + 8 * q) * 128 + (tid & 31) * 4) = p[q]; }
    }
}
// mode 5: wave-private 32-row tiles, A operand loaded straight in MFMA layout (lane = (row, k-half), 16 B per lane:
// 32 rows x 32 B per instruction), MFMA C-layout dword stores
template <int PREF>
__global__ __launch_bounds__(512, 2) void k_direct(const float* __restrict__ X, float* __restrict__ Y, long M) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    const long nw = (long)gridDim.x * (blockDim.x >> 6), w0 = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    for (long t = w0; t < M / 32; t += nw) {
        const float* xr = X + (t * 32 + l31) * 128 + 4 * h;
        float4 v[16];
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = *(const float4*)(xr + 8 * i);
        float acc[4][16];
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[nt][r] = v[(nt * 4 + (r >> 2)) & 15].x + v[r].y * (float)nt;
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++) Y[(t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 128 + nt * 32 + l31] = acc[nt][r];
    }
}
template <typename F> void timeit(const char* name, F f, double bytes) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    for (int r = 0; r < 3; r++) f();
    hipEventRecord(s); for (int r = 0; r < 10; r++) f();
    hipEventRecord(e); hipEventSynchronize(e); float ms; hipEventElapsedTime(&ms, s, e);
    printf("%-40s %8.1f us  %.2f TB/s\n", name, ms * 100, bytes / (ms / 10 * 1e-3) / 1e12);
}
int main() {
    const long M = 960000; const double B = M * 512.0;
    float *X, *Y; hipMalloc(&X, M * 512); hipMalloc(&Y, M * 512); hipMemset(X, 0, M * 512);
    for (int blocks : {1024, 2048, 8192, 65536})
        { char n[64]; sprintf(n, "copy float4 grid-stride blocks=%d", blocks); timeit(n, [&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, (const float4*)X, (float4*)Y, M * 32); }, 2 * B); }
    for (int blocks : {1024, 2048, 8192})
        { char n[64]; sprintf(n, "read-only float4 blocks=%d", blocks); timeit(n, [&] { hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, (const float4*)X, Y, M * 32); }, B); }
    timeit("tile: 1 chunk in flight, dword stores", [&] { hipLaunchKernelGGL(k_tile<2>, dim3(M / 128), dim3(256), 0, 0, X, Y, M); }, 2 * B);
    timeit("tile: 4 chunks up front, dword stores", [&] { hipLaunchKernelGGL(k_tile<3>, dim3(M / 128), dim3(256), 0, 0, X, Y, M); }, 2 * B);
    timeit("tile: full rows float4 in/out", [&] { hipLaunchKernelGGL(k_tile<4>, dim3(M / 128), dim3(256), 0, 0, X, Y, M); }, 2 * B);
    for (int blocks : {256, 512})
        { char n[64]; sprintf(n, "direct MFMA-layout loads, blocks=%d x512", blocks); timeit(n, [&] { hipLaunchKernelGGL(k_direct<0>, dim3(blocks), dim3(512), 0, 0, X, Y, M); }, 2 * B); }
    return 0;
}
