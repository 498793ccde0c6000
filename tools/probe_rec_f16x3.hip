// Timing probe for the split-precision recurrent kernel (not part of the library): where does a step go?
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -DFSN_PROBE_ABLATE=k tools/probe_rec_f16x3.hip -o probe   (k = 0..3)
#include <cstdio>
#include <cstdlib>
#include "../fullsubnet_amd/csrc/lstm_f16x3_kernels.hip"
void fsn_set_error(const char*, ...) {}
int fsn_check_launch(const char*) { return hipGetLastError() == hipSuccess ? 0 : -3; }
__global__ void fill_kernel(float* p, size_t n, unsigned seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 747796405u + seed; x ^= x >> 16; x *= 2246822519u; x ^= x >> 13;
        p[i] = ((x & 0xffff) / 32768.0f - 1.0f) * scale;
    }
}
__global__ void fill_half_kernel(_Float16* p, size_t n, unsigned seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 747796405u + seed; x ^= x >> 16; x *= 2246822519u; x ^= x >> 13;
        p[i] = (_Float16)(((x & 0xffff) / 32768.0f - 1.0f) * scale);
    }
}
int main(int argc, char** argv) {
    const int Tp = argc > 1 ? atoi(argv[1]) : 190;
    const int H = 384, tiles = 1028, Npad = tiles * 16;
    float *gx, *crm, *fcw, *fcb;
    _Float16* w;
    hipMalloc(&gx, (size_t)Tp * Npad * 4 * H * 4);
    hipMalloc(&w, (size_t)2 * 4 * H * H * 2);
    hipMalloc(&crm, (size_t)2 * 64 * Tp * 272 * 4);
    hipMalloc(&fcw, 16 * H * 4);
    hipMalloc(&fcb, 64);
    fill_kernel<<<4096, 256>>>(gx, (size_t)Tp * Npad * 4 * H, 1, 1.0f);
    fill_half_kernel<<<256, 256>>>(w, (size_t)2 * 4 * H * H, 2, 12.0f);
    fill_kernel<<<16, 256>>>(fcw, 16 * H, 3, 0.1f);
    fill_kernel<<<1, 64>>>(fcb, 16, 4, 0.1f);
    hipDeviceSynchronize();
    FsnRecFc fc{};
    fc.w_p = fcw; fc.bias = fcb; fc.crm_r = crm; fc.crm_i = crm + (size_t)64 * Tp * 272;
    fc.N = 64 * 257; fc.F = 257; fc.FP = 272; fc.T = Tp; fc.la = 0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0, 0);
        fsn_launch_lstm_rec_f16x3(gx, w, Tp, Npad, H, 4, 255, &fc, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (it > 0 && ms < best) best = ms;
    }
    const double flops = 3 * 2.0 * 255 * 64 * 384.0 * 1536 * Tp;
    printf("rec f16x3 ablate=%d: %.3f ms  %.1f TFLOP/s of 16-bit MFMA (floor at 2088: %.3f ms)\n", FSN_PROBE_ABLATE, best,
           flops / best / 1e9, flops / 2088e9);
    return 0;
}
