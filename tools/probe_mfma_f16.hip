// Pure-MFMA ceiling probe for the 16-bit matrix instructions a split-precision (fp16 x 3) version of the fp32
// GEMMs would use: v_mfma_f32_16x16x32_f16 (gfx950) with fp32 accumulation, no memory traffic.
// Three of these per 16x16x32 block replace eight v_mfma_f32_16x16x4_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_f16_kernel(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (_Float16)(a0 + 0.001f * (threadIdx.x + j));
        b[j] = (_Float16)(b0 + 0.002f * (threadIdx.x + j));
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    }
    f32x4 s = acc[0];
    for (int i = 1; i < NACC; ++i) s += acc[i];
    *(f32x4*)(out + ((size_t)blockIdx.x * 256 + threadIdx.x) * 4) = s;
}

int main() {
    float* out;
    (void)hipMalloc(&out, (size_t)4096 * 256 * 4 * sizeof(float));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int blocks = 1024, iters = 20000;
    float best = 1e30f;
    for (int t = 0; t < 3; ++t) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(mfma_f16_kernel<8>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (t > 0 && ms < best) best = ms;
    }
    const double mfmas = (double)iters * 4 * 8 * blocks * 4;           // per wave: iters x 4 x NACC
    const double flops = mfmas * 2.0 * 16 * 16 * 32;
    printf("v_mfma_f32_16x16x32_f16: %.3f ms, %.1f TFLOP/s raw; as fp16x3 (3 MFMAs per fp32 product block): %.1f "
           "fp32-equivalent TFLOP/s\n", best, flops / best / 1e9, flops / 3.0 / best / 1e9);
    return 0;
}
