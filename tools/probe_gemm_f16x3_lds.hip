// Timing + accuracy probe for the two forms of the split-precision GEMM (not part of the library).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/probe_gemm_f16x3_lds.hip -o probe ; ./probe [rows]
//   FSN_F16X3_GEMM_DIRECT=1 ./probe   -> the first (register-direct) form
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../fullsubnet_amd/csrc/gemm_f16x3_kernels.hip"
void fsn_set_error(const char*, ...) {}
int fsn_check_launch(const char*) { return hipGetLastError() == hipSuccess ? 0 : -3; }
__global__ void fill_kernel(float* p, size_t n, unsigned seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 747796405u + seed; x ^= x >> 16; x *= 2246822519u; x ^= x >> 13;
        p[i] = ((x & 0xffff) / 32768.0f - 1.0f) * scale;
    }
}
int main(int argc, char** argv) {
    const int K = 384, N = 1536;
    const long M = argc > 1 ? atol(argv[1]) : 190L * 16448;
    float *A, *W, *C, *bias;
    void* packed;
    (void)hipMalloc(&A, (size_t)M * K * 4);
    (void)hipMalloc(&W, (size_t)N * K * 4);
    (void)hipMalloc(&C, (size_t)M * N * 4);
    (void)hipMalloc(&bias, N * 4);
    (void)hipMalloc(&packed, fsn_f16x3_packed_halves(N, K) * 2);
    fill_kernel<<<4096, 256>>>(A, (size_t)M * K, 1, 1.0f);
    fill_kernel<<<256, 256>>>(W, (size_t)N * K, 2, 0.05f);
    fill_kernel<<<8, 256>>>(bias, N, 3, 0.1f);
    fsn_launch_pack_f16x3(W, packed, N, K, 0, 256.f);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int it = 0; it < 4; ++it) {
        (void)hipEventRecord(e0, 0);
        if (fsn_launch_gemm_f16x3(A, K, packed, bias, C, M / 16, N, K, 0) != 0) { printf("launch failed\n"); return 1; }
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (it > 0 && ms < best) best = ms;
    }
    const int SR = 48;
    std::vector<float> hW((size_t)N * K), hb(N);
    (void)hipMemcpy(hW.data(), W, hW.size() * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hb.data(), bias, N * 4, hipMemcpyDeviceToHost);
    double max_err = 0;
    const long starts[3] = {0, M / 3 / 16 * 16, M - SR};
    for (long r0 : starts) {
        std::vector<float> hA((size_t)SR * K), hC((size_t)SR * N);
        (void)hipMemcpy(hA.data(), A + r0 * K, hA.size() * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hC.data(), C + r0 * N, hC.size() * 4, hipMemcpyDeviceToHost);
        for (int r = 0; r < SR; ++r)
            for (int n = 0; n < N; ++n) {
                double ref = hb[n];
                for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)r * K + k] * hW[(size_t)n * K + k];
                const size_t fo = (((size_t)(r / 16) * (N / 16) + n / 16) * 64 + (n % 16) + 16 * ((r % 16) / 4)) * 4 + r % 4;
                max_err = fmax(max_err, fabs(hC[fo] - ref));
            }
    }
    printf("f16x3 GEMM (%s) %ld x %d x %d: %.3f ms = %.1f fp32-equivalent TFLOP/s; max |err| vs fp64 %.3e\n",
           getenv("FSN_F16X3_GEMM_DIRECT") ? "direct" : "lds", M, K, N, best, 2.0 * M * K * N / best / 1e9, max_err);
    return 0;
}
