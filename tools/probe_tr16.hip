#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out, int rowstride_elems) {
    __shared__ __attribute__((aligned(16))) short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int lane = threadIdx.x, i = lane & 15, grp = lane >> 4;
    // hypothesis (b): lane i of a 16-lane group addresses row i / 4, column quad i % 4 of a [4][16] block
    short* p = lds + grp * 4 * rowstride_elems + (i >> 2) * rowstride_elems + (i & 3) * 4;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * 2);
    for (int rs : {16, 32}) {
        k<<<1, 64>>>(d, rs);
        short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("row stride %d elements:\n", rs);
        for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %4d", h[l * 4 + j]); printf("\n"); }
    }
    return 0;
}
