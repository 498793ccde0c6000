// Pure-MFMA ceiling probe: how many fp32 TFLOP/s do v_mfma_f32_16x16x4_f32 / 32x32x2_f32 sustain
// with no memory traffic, for short and long (tens of ms) kernels?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma16_kernel(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    f32x4 s = acc[0];
    for (int i = 1; i < NACC; ++i) s += acc[i];
    *(f32x4*)(out + ((size_t)blockIdx.x * 256 + threadIdx.x) * 4) = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void mfma32_kernel(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    f32x16 s = acc[0];
    for (int i = 1; i < NACC; ++i) s += acc[i];
    *(f32x4*)(out + ((size_t)blockIdx.x * 256 + threadIdx.x) * 4) = f32x4{s[0], s[1], s[2], s[3]};
}

template <class K>
void time_it(const char* name, K kern, int blocks, int iters, double flop_per_iter_per_wave, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    for (int t = 0; t < 3; ++t) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (t > 0 && ms < best) best = ms;
    }
    const double flops = flop_per_iter_per_wave * iters * blocks * 4.0;
    printf("%-44s %8.3f ms  %.1f TFLOP/s\n", name, best, flops / best / 1e9);
}

int main() {
    float* out;
    hipMalloc(&out, (size_t)4096 * 256 * 16);
    const double f16 = 2.0 * 16 * 16 * 4, f32 = 2.0 * 32 * 32 * 2;
    time_it("16x16x4  8 acc, 1 wg/CU, short", mfma16_kernel<8>, 256, 2000, f16 * 4 * 8, out);
    time_it("16x16x4  8 acc, 1 wg/CU, long", mfma16_kernel<8>, 256, 40000, f16 * 4 * 8, out);
    time_it("16x16x4  8 acc, 3 wg/CU, long", mfma16_kernel<8>, 768, 20000, f16 * 4 * 8, out);
    time_it("16x16x4 16 acc, 3 wg/CU, long", mfma16_kernel<16>, 768, 10000, f16 * 4 * 16, out);
    time_it("16x16x4  2 acc, 1 wg/CU, long", mfma16_kernel<2>, 256, 160000, f16 * 4 * 2, out);
    time_it("32x32x2  4 acc, 1 wg/CU, short", mfma32_kernel<4>, 256, 2000, f32 * 4 * 4, out);
    time_it("32x32x2  4 acc, 1 wg/CU, long", mfma32_kernel<4>, 256, 40000, f32 * 4 * 4, out);
    time_it("32x32x2  4 acc, 3 wg/CU, long", mfma32_kernel<4>, 768, 20000, f32 * 4 * 4, out);
    return 0;
}
