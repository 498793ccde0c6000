// Timing probe for the fp32 weight-gradient product (gemm_tn_kernel, not part of the library): config 3's shape
// C[1536][384] = sum over 395 264 rows of dgates^T h, 16 K splits of 192 x 192 tiles on 256 CUs.
#include <cstdio>
#include <cstdlib>
#include "../fullsubnet_amd/csrc/lstm_train_kernels.hip"
void fsn_set_error(const char*, ...) {}
int fsn_check_launch(const char*) { return hipGetLastError() == hipSuccess ? 0 : -3; }
__global__ void fill_kernel(float* p, size_t n, unsigned seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 747796405u + seed; x ^= x >> 16; x *= 2246822519u; x ^= x >> 13;
        p[i] = ((x & 0xffff) / 32768.0f - 1.0f) * scale;
    }
}
template <class Kern>
float run(Kern k, const float* A, const float* B, float* part, float* asum, int M, int Nc, long K, int mb, int nb, int splits) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    const long kps = ((K / splits) + 15) & ~15L;
    for (int it = 0; it < 4; ++it) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(mb * nb * splits), dim3(256), 96 * 1024, 0, A, (long)M, B, (long)Nc, part, M, Nc, K, kps, mb, nb, asum, 1);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (it > 0 && ms < best) best = ms;
    }
    return best;
}
int main() {
    const int M = 1536, Nc = 384; const long K = 193L * 2048;
    float *A, *B, *part, *asum;
    hipMalloc(&A, (size_t)K * M * 4); hipMalloc(&B, (size_t)K * Nc * 4); hipMalloc(&part, (size_t)32 * M * Nc * 4); hipMalloc(&asum, (size_t)32 * M * 4);
    fill_kernel<<<4096, 256>>>(A, (size_t)K * M, 1, 0.1f); fill_kernel<<<4096, 256>>>(B, (size_t)K * Nc, 2, 0.5f);
    hipDeviceSynchronize();
    const double flops = 2.0 * M * Nc * (double)K;
    auto rep = [&](const char* what, float ms) { printf("  %-58s: %.3f ms = %.1f TFLOP/s (%.3f of 157.3)\n", what, ms, flops / ms * 1e-9, flops / ms * 1e-9 / 157.3); };
    rep("gemm_tn_kernel: 192 x 192 tiles (6 x 6 per wave), ring depth 2", run(gemm_tn_kernel<6, 6, 2, 2, FSN_ARITH_F32, 2>, A, B, part, asum, M, Nc, K, 8, 2, 16));
    rep("  operands not loaded (matrix stream + structure alone)", run(gemm_tn_kernel<6, 6, 2, 2, FSN_ARITH_F32, 2, 1>, A, B, part, asum, M, Nc, K, 8, 2, 16));
    rep("  ring depth 1", run(gemm_tn_kernel<6, 6, 2, 2, FSN_ARITH_F32, 1>, A, B, part, asum, M, Nc, K, 8, 2, 16));
    rep("256 x 128 tiles (8 x 4 per wave), 18 tiles x 14 splits", run(gemm_tn_kernel<8, 4, 2, 2, FSN_ARITH_F32, 2>, A, B, part, asum, M, Nc, K, 6, 3, 14));
    rep("  operands not loaded", run(gemm_tn_kernel<8, 4, 2, 2, FSN_ARITH_F32, 2, 1>, A, B, part, asum, M, Nc, K, 6, 3, 14));
    rep("shipped again", run(gemm_tn_kernel<6, 6, 2, 2, FSN_ARITH_F32, 2>, A, B, part, asum, M, Nc, K, 8, 2, 16));
    return 0;
}
