// Timing probe for the sub-band recurrent kernels (not part of the library).
#include <cstdio>
#include <cstdlib>
#include "../fullsubnet_amd/csrc/lstm_kernels.hip"
#include "experimental_rec1.hip"
void fsn_set_error(const char*, ...) {}
int fsn_check_launch(const char*) { return hipGetLastError() == hipSuccess ? 0 : -3; }
__global__ void fill_kernel(float* p, size_t n, unsigned seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 747796405u + seed; x ^= x >> 16; x *= 2246822519u; x ^= x >> 13;
        p[i] = ((x & 0xffff) / 32768.0f - 1.0f) * scale;
    }
}
int main(int argc, char** argv) {
    const int Tp = argc > 1 ? atoi(argv[1]) : 190;
    const int H = 384, tiles = 1028, Npad = tiles * 16;
    float *gx, *w, *hseq;
    hipMalloc(&gx, (size_t)Tp * Npad * 4 * H * 4);
    hipMalloc(&hseq, (size_t)Tp * Npad * H * 4);
    hipMalloc(&w, (size_t)4 * H * H * 4);
    fill_kernel<<<4096, 256>>>(gx, (size_t)Tp * Npad * 4 * H, 1, 1.0f);
    fill_kernel<<<256, 256>>>(w, (size_t)4 * H * H, 2, 0.05f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int ver = 0; ver < 5; ++ver) {
        float best = 1e30f;
        for (int it = 0; it < 3; ++it) {
            hipEventRecord(e0, 0);
            if (ver == 0) launch_rec1<384, 4>(gx, w, hseq, Tp, Npad, 256, 0);
            else if (ver == 1) launch_rec<384, 4, false, 2>(gx, nullptr, w, hseq, Tp, Npad, 256, 0);
            else if (ver == 2) launch_rec<384, 4, false, 3>(gx, nullptr, w, hseq, Tp, Npad, 256, 0);
            else if (ver == 3) launch_rec<384, 4, false, 4>(gx, nullptr, w, hseq, Tp, Npad, 256, 0);
            else launch_rec<384, 4, false, 6>(gx, nullptr, w, hseq, Tp, Npad, 256, 0);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (it > 0 && ms < best) best = ms;
        }
        const double flops = 2.0 * 256 * 64 * 384.0 * 1536 * Tp;
        const char* names[5] = {"rec1 (1 wave/SIMD, pinned)", "rec UG=2 (12 waves)", "rec UG=3 (8 waves)", "rec UG=4 (6 waves)",
                                "rec UG=6 (4 waves)"};
        printf("%s ablate=%d: %.3f ms  %.1f TFLOP/s (ideal %.3f ms)\n", names[ver], FSN_REC1_ABLATE, best,
               flops / best / 1e9, flops / 156e9);
    }
    return 0;
}
