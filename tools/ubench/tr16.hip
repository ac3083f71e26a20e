// What ds_read_b64_tr_b16 returns (gfx950): LDS holds h[r][c] = 256 r + c (16-bit), 64 halfs per row.  Lane l = 16 g + i supplies the
// address of row 4 g + (i >> 2), columns 4 (i & 3) .. + 3  (a [4 rows][16 columns] block per 16-lane group, 4 contiguous halfs per lane);
// prints the four halfs every lane gets.  hipcc --offload-arch=gfx950 -O3 -o tr16 tr16.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __fp16 h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__global__ void k(unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short h[16 * 64];
    for (int e = threadIdx.x; e < 16 * 64; e += 64) h[e] = (unsigned short)(256 * (e / 64) + (e % 64));
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, i = l & 15;
    const unsigned short* p = &h[(4 * g + (i >> 2)) * 64 + 4 * (i & 3)];
    h4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)p);
    unsigned short u[4];
    __builtin_memcpy(u, &v, 8);
    for (int j = 0; j < 4; j++) out[l * 4 + j] = u[j];
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    k<<<1, 64>>>(d);
    unsigned short hst[256]; hipMemcpy(hst, d, sizeof(hst), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l++) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; j++) printf(" (r%d,c%d)", hst[l * 4 + j] >> 8, hst[l * 4 + j] & 255);
        printf("\n");
    }
    return 0;
}
