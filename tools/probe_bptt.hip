// Timing probe for lstm2_group_bptt_kernel (not part of the library): the shipped kernel and ablations (template
// parameter ABL, see the kernel) on random operands at config 3's per-rank shape (2064 rows = 32 clusters + 16 rows).
#include <cstdio>
#include <cstdlib>
#include "../fullsubnet_amd/csrc/lstm_group_bptt_kernels.hip"
void fsn_set_error(const char*, ...) {}
bool fsn_persistent_allowed() { return true; }
bool fsn_grid_fits(const void*, int, unsigned) { return true; }
void fsn_persist_admit(const void*, int, unsigned) {}
unsigned long long fsn_spin_ticks() { return 1ull << 31; }
unsigned* fsn_ctx_sticky() { return nullptr; }
int fsn_check_launch(const char*) { return hipGetLastError() == hipSuccess ? 0 : -3; }
int fsn_launch_zero_words(unsigned* p, size_t n, hipStream_t s) { return hipMemsetAsync(p, 0, n * 4, s) == hipSuccess ? 0 : -3; }
__global__ void fill_kernel(float* p, size_t n, unsigned seed, float scale, float offset) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 747796405u + seed; x ^= x >> 16; x *= 2246822519u; x ^= x >> 13;
        p[i] = ((x & 0xffff) / 32768.0f - 1.0f) * scale + offset;
    }
}
#ifndef PROBE_AR
#define PROBE_AR 0  // -DPROBE_AR=2 / 3: fp16 / bf16 matrix-core operands
#endif
template <int ABL>
float run(BpttArgs a, int clusters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int it = 0; it < 3; ++it) {
        hipMemsetAsync(a.flags, 0, fsn_lstm2_group_bptt_flag_words(clusters) * 4, 0);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((lstm2_group_bptt_kernel<ABL, PROBE_AR>), dim3(clusters * BM * 2), dim3(256), 0, 0, a);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (it > 0 && ms < best) best = ms;
    }
    return best;
}
int main(int argc, char** argv) {
    const int Tp = argc > 1 ? atoi(argv[1]) : 193, clusters = argc > 2 ? atoi(argv[2]) : 32, N = clusters * 64 + 16;
    const size_t TN = (size_t)Tp * N;
    float *dh1, *w, *sv0, *sv1, *dg, *dx; unsigned* flags;
    hipMalloc(&dh1, TN * BH * 4); hipMalloc(&w, (size_t)3 * BH * BG * 4); hipMalloc(&sv0, TN * 5 * BH * 4); hipMalloc(&sv1, TN * 5 * BH * 4);
    hipMalloc(&dg, 2 * TN * BG * 4); hipMalloc(&dx, TN * BH * 4); hipMalloc(&flags, fsn_lstm2_group_bptt_flag_words(clusters) * 4);
    fill_kernel<<<1024, 256>>>(dh1, TN * BH, 1, 0.01f, 0.f);
    fill_kernel<<<1024, 256>>>(w, (size_t)3 * BH * BG, 2, 0.05f, 0.f);
    fill_kernel<<<1024, 256>>>(sv0, TN * 5 * BH, 3, 0.4f, 0.5f);
    fill_kernel<<<1024, 256>>>(sv1, TN * 5 * BH, 4, 0.4f, 0.5f);
    hipDeviceSynchronize();
    BpttArgs a{};
    a.dh1 = dh1; a.wbase = w; a.o_whh1T = 0; a.o_wih1T = BH * BG; a.o_whh0T = 2 * BH * BG;
    a.gates0 = sv0; a.cseq0 = sv0 + TN * BG; a.gates1 = sv1; a.cseq1 = sv1 + TN * BG; a.dg1 = dg; a.dg0 = dg + TN * BG; a.dx = dx;
    a.flags = flags; a.status = flags + (size_t)clusters * 2 * BFS; a.spin_ticks = 1ull << 31; a.Tp = Tp; a.Nrows = N;
    const double mfma_us = 2.0 * 64 * 48 * (3.0 * BG) / (64.0 * 4 * 2.4e3);  // per step and CU (one member of each layer) at 2.4 GHz
    const float t0 = run<0>(a, clusters);
    unsigned st = 0; hipMemcpy(&st, a.status, 4, hipMemcpyDeviceToHost);
    printf("arithmetic %d: ", PROBE_AR);
    printf("lstm2_group_bptt_kernel, %d clusters, %d steps: %.3f ms = %.1f us per step (MFMA alone %.1f us), status %u\n", clusters, Tp, t0, 1e3 * t0 / Tp, mfma_us, st);
#define V(abl, what) { const float t = run<abl>(a, clusters); printf("  %-52s: %.3f ms = %.1f us per step\n", what, t, 1e3 * t / Tp); }
    V(16, "plain instead of write-through stores");
    V(2, "no gate-gradient / dx stores");
    V(8, "saved activations not loaded");
    V(32, "no tanhf");
    V(4, "A fragments not loaded");
    V(1, "no flag polling");
    V(2 + 8, "no stores, no saved activations");
    V(2 + 8 + 4, "... and no A loads");
    V(1 + 2 + 8 + 4 + 32, "... no flags, no tanhf (K loops, barriers, B through LDS)");
    V(0, "shipped again");
    return 0;
}
