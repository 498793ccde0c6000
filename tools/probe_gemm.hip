// Stand-alone timing probe for the fp32 MFMA GEMM family (not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/probe_gemm.hip -o gpurun_out/probe_gemm
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ unsigned g_place[65536];
#define FSN_GEMM_PLACEMENT g_place
#include <map>
#include "../fullsubnet_amd/csrc/gemm_kernels.hip"

void fsn_set_error(const char*, ...) {}
int fsn_check_launch(const char*) { return hipGetLastError() == hipSuccess ? 0 : -3; }

__global__ void fill_kernel(float* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 747796405u + seed;
        x ^= x >> 16;
        x *= 2246822519u;
        x ^= x >> 13;
        p[i] = ((x & 0xffff) / 32768.0f - 1.0f) * 0.5f;
    }
}

template <int PF, int RTW = 4, int CTW = 4, int WR = 2, int WC = 2, int AK = 0>
void run(const char* name, const FsnGemmA& a, const float* wp, const FsnGemmC& c, int row_tiles, int col_tiles,
         int kchunks, int wg_per_cu = 0) {
    int nb = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gemm_kernel<AK, 0, RTW, CTW, WR, WC, PF>, WR * WC * 64, 0);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    for (int it = 0; it < 4; ++it) {
        hipEventRecord(e0, 0);
        launch<AK, 0, RTW, CTW, WR, WC, PF>(a, wp, c, row_tiles, col_tiles, kchunks, 0, wg_per_cu);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (it > 0 && ms < best) best = ms;
    }
    {   // placement histogram of the last launch: workgroups per (xcc, se/sh/cu)
        static unsigned hp[65536];
        hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_place), sizeof(hp));
        const long nrb = ((long)row_tiles + WR * RTW - 1) / (WR * RTW), ncb = (col_tiles + WC * CTW - 1) / (WC * CTW);
        long grid = 256L * (wg_per_cu > 0 ? wg_per_cu : nb);
        if (grid > nrb * ncb) grid = nrb * ncb;
        std::map<unsigned, int> per_cu;
        for (long b = 0; b < grid && b < 32768; ++b) per_cu[((hp[2 * b + 1] & 0xf) << 16) | (hp[2 * b] & 0xff00)]++;
        int hist[16] = {0};
        for (auto& kv : per_cu) hist[kv.second < 15 ? kv.second : 15]++;
        printf("   grid %ld on %zu CUs; CUs with k WGs:", grid, per_cu.size());
        for (int k = 1; k < 16; ++k) if (hist[k]) printf(" %dx%d", hist[k], k);
        printf("\n");
    }
    const double flops = 2.0 * row_tiles * 16.0 * col_tiles * 16.0 * kchunks * 16.0;
    printf("%-28s blocks/CU(api)=%d  %.3f ms  %.1f TFLOP/s\n", name, nb, best, flops / best / 1e9);
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 16;
    const long Npad = ((long)B * 257 + 79) / 80 * 80, Tp = 190;
    const long R = Tp * Npad;
    const int H = 384, row_tiles = (int)(R / 16), col_tiles = 4 * H / 16, kchunks = H / 16;
    float *A, *W, *G, *bias;
    hipMalloc(&A, R * H * 4);
    hipMalloc(&W, (size_t)4 * H * H * 4);
    hipMalloc(&G, R * 4 * H * 4);
    hipMalloc(&bias, 4 * H * 4);
    fill_kernel<<<2048, 256>>>(A, R * H, 1);
    fill_kernel<<<256, 256>>>(W, (size_t)4 * H * H, 2);
    fill_kernel<<<8, 256>>>(bias, 4 * H, 3);
    hipDeviceSynchronize();
    FsnGemmA a{};
    a.kind = 0;
    a.p0 = A;
    a.ld = H;
    FsnGemmC c{};
    c.kind = 0;
    c.p0 = G;
    c.bias = bias;
    printf("rows %ld (B=%d), K=%d, cols=%d\n", R, B, H, 4 * H);
    run<2, 4, 4, 2, 2, 3>("PF2 4x4 2x2 1/CU", a, W, c, row_tiles, col_tiles, kchunks, 1);
    run<3, 4, 4, 2, 2, 3>("PF3 4x4 2x2 1/CU", a, W, c, row_tiles, col_tiles, kchunks, 1);
    run<4, 4, 4, 2, 2, 3>("PF4 4x4 2x2 1/CU", a, W, c, row_tiles, col_tiles, kchunks, 1);
    run<2, 4, 8, 2, 2, 3>("PF2 4x8 2x2 1/CU", a, W, c, row_tiles, col_tiles, kchunks, 1);
    run<3, 4, 8, 2, 2, 3>("PF3 4x8 2x2 1/CU", a, W, c, row_tiles, col_tiles, kchunks, 1);
    run<2, 8, 4, 2, 2, 3>("PF2 8x4 2x2 1/CU", a, W, c, row_tiles, col_tiles, kchunks, 1);
    run<2, 6, 6, 2, 2, 3>("PF2 6x6 2x2 1/CU", a, W, c, row_tiles, col_tiles, kchunks, 1);
    run<2, 8, 6, 2, 2, 3>("PF2 8x6 2x2 1/CU", a, W, c, row_tiles, col_tiles, kchunks, 1);
    run<2, 8, 8, 2, 2, 3>("PF2 8x8 2x2 1/CU", a, W, c, row_tiles, col_tiles, kchunks, 1);
    run<2, 4, 8, 2, 2, 0>("PF2 4x8 2x2 1/CU rowmajor", a, W, c, row_tiles, col_tiles, kchunks, 1);
    run<2, 4, 8, 4, 1, 3>("PF2 4x8 4x1 1/CU", a, W, c, row_tiles, col_tiles, kchunks, 1);
    run<2, 4, 8, 1, 4, 3>("PF2 4x8 1x4 1/CU", a, W, c, row_tiles, col_tiles, kchunks, 1);
    return 0;
}
