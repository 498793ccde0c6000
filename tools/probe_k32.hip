// Is one v_mfma_f32_16x16x32_{f16,bf16} on the concatenation of two K = 16 operands the same contraction as the two
// v_mfma_f32_16x16x16 instructions (fsn_mma_k32 against fsn_mma_k16 twice)?  Random operands, fp32 results compared.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "../include/fsn_hip.h"
#include "../fullsubnet_amd/csrc/fsn_common.h"
template <int AR>
__global__ void k(const float* src, float* out16, float* out32) {
    const int lane = threadIdx.x;
    f32x4 v[4];
    for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(src + (i * 64 + lane) * 4);
    const auto a0 = fsn_operand<AR>(v[0]), a1 = fsn_operand<AR>(v[1]), b0 = fsn_operand<AR>(v[2]), b1 = fsn_operand<AR>(v[3]);
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    const f32x4 d16 = fsn_mma_k16<AR>(a1, b1, fsn_mma_k16<AR>(a0, b0, c));
    const f32x4 d32 = fsn_mma_k32<AR>(a0, a1, b0, b1, c);
    *reinterpret_cast<f32x4*>(out16 + lane * 4) = d16;
    *reinterpret_cast<f32x4*>(out32 + lane * 4) = d32;
}
int main() {
    float h[1024], r16[256], r32[256], *d, *o16, *o32;
    for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 7919 % 1013) - 506) / 300.0f;
    hipMalloc(&d, sizeof h); hipMalloc(&o16, sizeof r16); hipMalloc(&o32, sizeof r32);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    for (int ar = 0; ar < 2; ++ar) {
        if (ar == 0) hipLaunchKernelGGL(k<FSN_ARITH_F16>, dim3(1), dim3(64), 0, 0, d, o16, o32);
        else hipLaunchKernelGGL(k<FSN_ARITH_BF16>, dim3(1), dim3(64), 0, 0, d, o16, o32);
        hipMemcpy(r16, o16, sizeof r16, hipMemcpyDeviceToHost);
        hipMemcpy(r32, o32, sizeof r32, hipMemcpyDeviceToHost);
        double m = 0, s = 0;
        for (int i = 0; i < 256; ++i) { m = fmax(m, fabs((double)r16[i] - r32[i])); s = fmax(s, fabs((double)r16[i])); }
        printf("%s: max |K32 - 2 x K16| = %.3e (values up to %.3f)\n", ar ? "bf16" : "f16", m, s);
    }
    return 0;
}
