// Does the register file of the MFMA operands matter?  acc in AGPRs; B operand from a VGPR vs an AGPR.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0: A v, B v   1: A v, B a   2: A a, B a
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = a0 + threadIdx.x + i; b[i] = b0 + threadIdx.x * 0.5f + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[r]), "v"(b[r]));
                if (MODE == 1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[r]), "a"(b[r]));
                if (MODE == 2) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "a"(a[r]), "a"(b[r]));
            }
    }
    f32x4 s = acc[0];
    for (int i = 1; i < 8; ++i) s += acc[i];
    *(f32x4*)(out + ((size_t)blockIdx.x * 256 + threadIdx.x) * 4) = s;
}
template <class K>
void time_it(const char* name, K kern, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f; const int iters = 20000, blocks = 256;
    for (int t = 0; t < 3; ++t) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (t > 0 && ms < best) best = ms;
    }
    printf("%-30s %8.3f ms  %.1f TFLOP/s\n", name, best, 2.0 * 16 * 16 * 4 * 32.0 * iters * blocks * 4 / best / 1e9);
}
int main() {
    float* out; hipMalloc(&out, (size_t)1024 * 256 * 16);
    time_it("A vgpr, B vgpr, C agpr", k<0>, out);
    time_it("A vgpr, B agpr, C agpr", k<1>, out);
    time_it("A agpr, B agpr, C agpr", k<2>, out);
    return 0;
}
