// Timing probe for fb_chain_kernel (not part of the library): the shipped kernel and ablations (template parameter
// ABL, see the kernel) on random operands: where does a step of the full-band chain spend its time?
#include <cstdio>
#include <cstdlib>
#include "../fullsubnet_amd/csrc/fb_chain_kernels.hip"
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
template <int KS, int ABL>
float run(ChainArgs a) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int it = 0; it < 4; ++it) {
        hipMemsetAsync(a.flags, 0, fsn_fb_chain_flag_words() * 4, 0);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((fb_chain_kernel<512, KS, ABL>), dim3(2 * 128), dim3(256), 0, 0, a);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (it > 0 && ms < best) best = ms;
    }
    return best;
}
template <int KS>
void sweep(ChainArgs a, int Tp) {
    const float t0 = run<KS, 0>(a);
    unsigned st = 0; hipMemcpy(&st, a.status, 4, hipMemcpyDeviceToHost);
    printf("fb_chain_kernel<KS=%d>, %d row tiles, %d steps: %.3f ms = %.2f us per step, status %u\n", KS, a.RT, Tp, t0, 1e3 * t0 / Tp, st);
#define V(abl, what) { const float t = run<KS, abl>(a); printf("  %-46s: %.3f ms = %.2f us per step\n", what, t, 1e3 * t / Tp); }
    V(16, "no drain before the flag store");
    V(8, "plain instead of write-through stores");
    V(2, "no h / gx1 stores");
    V(64, "no layer-1 projection in L0");
    V(1, "no flag polling");
    V(1 + 32, "no flag polling, no flag stores");
    V(1 + 32 + 16, "... and no drain");
    V(1 + 32 + 16 + 2, "... and no h stores");
    V(1 + 32 + 16 + 2 + 4, "... and no A loads (MFMA, cell, barriers)");
    V(1 + 32 + 16 + 4, "no sync, no A loads, but h stores");
    V(4, "no A loads only");
#undef V
}
int main(int argc, char** argv) {
    const int Tp = argc > 1 ? atoi(argv[1]) : 190, Npad = argc > 2 ? atoi(argv[2]) : 64;
    const int H = 512;
    float *gx0, *w, *b1, *ex, *hseq; unsigned* flags;
    hipMalloc(&gx0, (size_t)Tp * Npad * 4 * H * 4); hipMalloc(&w, (size_t)3 * 4 * H * H * 4); hipMalloc(&b1, 4 * H * 4);
    hipMalloc(&ex, fsn_fb_chain_exchange_floats(Tp, Npad) * 4); hipMalloc(&hseq, (size_t)Tp * Npad * H * 4);
    hipMalloc(&flags, fsn_fb_chain_flag_words() * 4);
    fill_kernel<<<1024, 256>>>(gx0, (size_t)Tp * Npad * 4 * H, 1, 1.0f, 0.f);
    fill_kernel<<<1024, 256>>>(w, (size_t)3 * 4 * H * H, 2, 0.05f, 0.f);
    fill_kernel<<<8, 256>>>(b1, 4 * H, 3, 0.1f, 0.f);
    hipDeviceSynchronize();
    ChainArgs a{};
    a.gx0 = gx0; a.whh0_p = w; a.wih1_p = w + (size_t)4 * H * H; a.whh1_p = w + (size_t)8 * H * H; a.b1 = b1;
    a.hx0 = ex; a.hx1 = ex + (size_t)Tp * Npad * H; a.gx1 = ex + (size_t)2 * Tp * Npad * H; a.hseq1 = hseq;
    a.flags = flags; a.status = flags + fsn_fb_chain_status_word(); a.spin_ticks = 1ull << 31; a.Tp = Tp; a.RT = Npad / 16; a.Npad = Npad;
    if (a.RT == 1) sweep<4>(a, Tp);
    else if (a.RT == 2) sweep<2>(a, Tp);
    else sweep<1>(a, Tp);
    return 0;
}
