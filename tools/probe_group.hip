// Timing probe for lstm2_group_kernel (not part of the library): the shipped kernel and ablations (template parameter
// ABL, see the kernel) on random operands at the shape of an 8-utterance shard (32 clusters, 190 steps).
#include <cstdio>
#include <cstdlib>
#include "../fullsubnet_amd/csrc/lstm_group_kernels.hip"
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
template <int ABL>
float run(GrpArgs a, int clusters, size_t flag_words) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int it = 0; it < 3; ++it) {
        hipMemsetAsync(a.flags, 0, flag_words * 4, 0);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(lstm2_group_kernel<ABL>, dim3(clusters * GM * 2), dim3(256), 0, 0, a);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (it > 0 && ms < best) best = ms;
    }
    return best;
}
int main(int argc, char** argv) {
    const int Tp = argc > 1 ? atoi(argv[1]) : 190, clusters = argc > 2 ? atoi(argv[2]) : 32;
    const int B = 8, F = 257, FP = 272, H = 384, T = Tp - 2;
    float *mag, *fb, *den, *w, *bias, *fcw, *fcb, *cr, *ci, *ex;
    unsigned* flags;
    hipMalloc(&mag, (size_t)B * Tp * FP * 4); hipMalloc(&fb, (size_t)B * Tp * FP * 4); hipMalloc(&den, 64 * 4);
    hipMalloc(&w, ((size_t)4 * H * 32 + 3 * (size_t)4 * H * H) * 4); hipMalloc(&bias, 8 * H * 4);
    hipMalloc(&fcw, 16 * H * 4); hipMalloc(&fcb, 64);
    hipMalloc(&cr, (size_t)B * T * FP * 4); hipMalloc(&ci, (size_t)B * T * FP * 4);
    hipMalloc(&ex, fsn_lstm2_group_exchange_floats(clusters) * 4);
    const size_t fw = fsn_lstm2_group_flag_words(clusters);
    hipMalloc(&flags, fw * 4);
    fill_kernel<<<1024, 256>>>(mag, (size_t)B * Tp * FP, 1, 0.5f, 0.6f);
    fill_kernel<<<1024, 256>>>(fb, (size_t)B * Tp * FP, 2, 0.5f, 0.6f);
    fill_kernel<<<1, 64>>>(den, 64, 3, 0.0f, 1.0f);
    fill_kernel<<<1024, 256>>>(w, (size_t)4 * H * 32 + 3 * (size_t)4 * H * H, 4, 0.05f, 0.f);
    fill_kernel<<<8, 256>>>(bias, 8 * H, 5, 0.1f, 0.f);
    fill_kernel<<<8, 256>>>(fcw, 16 * H, 6, 0.1f, 0.f);
    hipMemset(fcb, 0, 64);
    hipDeviceSynchronize();
    GrpArgs a{};
    a.xin.mag = mag; a.xin.fb_out = fb; a.xin.den = den; a.xin.bias = bias; a.xin.den_mode = 0;
    a.xin.B = B; a.xin.Tp = Tp; a.xin.F = F; a.xin.FP = FP; a.xin.N = B * F < clusters * 64 ? B * F : clusters * 64; a.xin.nb = 15; a.xin.kin_chunks = 2;
    a.wbase = w; a.o_wih0 = 0; a.o_whh0 = 4 * H * 32; a.o_wih1 = a.o_whh0 + 4 * H * H; a.o_whh1 = a.o_wih1 + 4 * H * H;
    a.bias1 = bias + 4 * H; a.hx0 = ex; a.hx1 = ex + (size_t)clusters * GD0 * 64 * H; a.flags = flags; a.status = flags + (size_t)clusters * 2 * GFS; a.spin_ticks = 1ull << 31;
    a.fc.w_p = fcw; a.fc.bias = fcb; a.fc.crm_r = cr; a.fc.crm_i = ci; a.fc.N = a.xin.N; a.fc.F = F; a.fc.FP = FP; a.fc.T = T; a.fc.la = 2;
    a.Tp = Tp;
    const double mfma_us = 2.0 * 64 * (1536.0 / 8) * (416 + 768) / (64.0 * 4 * 2.4e3);  // per iteration and CU at 2.4 GHz
    const float t0 = run<0>(a, clusters, fw);
    unsigned st = 0; hipMemcpy(&st, a.status, 4, hipMemcpyDeviceToHost);
    printf("lstm2_group_kernel, %d clusters, %d steps: %.3f ms = %.1f us per iteration (MFMA alone %.1f us), status %u\n", clusters, Tp, t0, 1e3 * t0 / (Tp + 2), mfma_us, st);
    printf("  without the acquire fences   : %.3f ms\n", run<1>(a, clusters, fw));
    printf("  without fences and polls     : %.3f ms\n", run<3>(a, clusters, fw));
    printf("  plain instead of sc1 stores  : %.3f ms\n", run<4>(a, clusters, fw));
    printf("  without gate non-linearities : %.3f ms\n", run<8>(a, clusters, fw));
    printf("  without the output layer     : %.3f ms\n", run<16>(a, clusters, fw));
    printf("  without A-fragment loads     : %.3f ms\n", run<32>(a, clusters, fw));
    printf("  without layer-1 priority     : %.3f ms\n", run<64>(a, clusters, fw));
    printf("  without all of them          : %.3f ms\n", run<63>(a, clusters, fw));
    printf("  shipped again                : %.3f ms\n", run<0>(a, clusters, fw));
    return 0;
}
