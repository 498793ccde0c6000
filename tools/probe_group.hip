// Timing probe for lstm2_group_kernel (not part of the library): the shipped kernel and ablations (template parameter
// ABL, see the kernel) on random operands at the shape of an 8-utterance shard (32 clusters, 190 steps).
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
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
    if (argc > 3) {  // timeline of cluster 0: clock stamps at the phase boundaries of every step (ABL 4096)
        const size_t nd = (size_t)2 * GM * (Tp + 1) * 8;
        hipMalloc(&a.dbg, nd * 8); hipMemset(a.dbg, 0, nd * 8);
        const float ms = atoi(argv[3]) == 1 ? run<4096 + 8192>(a, clusters, fw) : run<4096>(a, clusters, fw);
        std::vector<unsigned long long> d(nd);
        hipMemcpy(d.data(), a.dbg, nd * 8, hipMemcpyDeviceToHost);
        int rate_khz = 100000; hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
        const double us = 1e3 / rate_khz;
        auto D = [&](int l, int m, int t, int e) { return (double)(long long)(d[(((size_t)l * GM + m) * (Tp + 1) + t) * 8 + e] - d[0]) * us; };
        printf("timeline run: %.3f ms, clock %d kHz\n", ms, rate_khz);
        const int t0 = 40, t1 = Tp - 20;
        const char* n0[] = {"input loads + wait h0[t-1]", "K loop (416)", "ring wait", "cell + stores", "publish (drain + flag)", "to next start"};
        const char* n1[] = {"wait h0[s]", "K loop 1 (384, x W_ih)", "wait h1[s-1]", "output layer", "K loop 2 (384, h W_hh)", "cell + stores", "publish", "to next start"};
        for (int l = 0; l < 2; ++l) {
            const int ne = l ? 8 : 6;
            printf("layer %d, mean phase durations over steps %d..%d [us], members 0..7 and their mean:\n", l, t0, t1);
            double tot = 0;
            for (int e = 0; e < ne; ++e) {
                double mm = 0;
                printf("  %-28s", l ? n1[e] : n0[e]);
                for (int m = 0; m < GM; ++m) {
                    double acc = 0;
                    for (int t = t0; t < t1; ++t) acc += (e + 1 < ne ? D(l, m, t, e + 1) : D(l, m, t + 1, 0)) - D(l, m, t, e);
                    acc /= (t1 - t0); mm += acc / GM;
                    printf(" %6.2f", acc);
                }
                printf("  | %6.2f\n", mm); tot += mm;
            }
            printf("  period %.2f us\n", tot);
        }
        // hand-off latencies: from the LAST member's publish-done stamp to a consumer's wait-passed stamp
        double h00 = 0, h01 = 0, h11 = 0, lead = 0;
        for (int t = t0; t < t1; ++t) {
            double p0 = -1e30, p0p = -1e30, p1p = -1e30, w0 = 0, w1 = 0, w11 = 0;
            for (int m = 0; m < GM; ++m) { p0 = std::max(p0, D(0, m, t, 5)); p0p = std::max(p0p, D(0, m, t - 1, 5)); p1p = std::max(p1p, D(1, m, t - 1, 7)); }
            for (int m = 0; m < GM; ++m) { w0 += D(0, m, t, 1) / GM; w1 += D(1, m, t, 1) / GM; w11 += D(1, m, t, 3) / GM; }
            h00 += w0 - p0p; h01 += w1 - p0; h11 += w11 - p1p;
            lead += D(1, 0, t, 0) - D(0, 0, t, 0);
        }
        const int n = t1 - t0;
        printf("mean [us]: L0 wait passed - last publish of h0[t-1]: %.2f; L1 wait passed - last publish of h0[s]: %.2f; L1 wait passed - last publish of h1[s-1]: %.2f; L1 starts step s this long after L0 started step s: %.2f\n", h00 / n, h01 / n, h11 / n, lead / n);
        printf("steps 100..102 of member 0 (us from step 100's layer-0 start):\n");
        const double o = D(0, 0, 100, 0);
        for (int t = 100; t < 103; ++t) {
            printf("  L0 t=%d:", t); for (int e = 0; e < 6; ++e) printf(" %7.2f", D(0, 0, t, e) - o); printf("\n");
            printf("  L1 s=%d:", t); for (int e = 0; e < 8; ++e) printf(" %7.2f", D(1, 0, t, e) - o); printf("\n");
        }
        return 0;
    }
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
    printf("  weight fragments not loaded  : %.3f ms\n", run<128>(a, clusters, fw));
    printf("  no weight loads, no LDS stage: %.3f ms\n", run<384>(a, clusters, fw));
    printf("  all of them + no weight loads: %.3f ms\n", run<63 + 128>(a, clusters, fw));
    printf("  all + no loads + no LDS stage: %.3f ms\n", run<63 + 384>(a, clusters, fw));
    printf("  members share an XCD         : %.3f ms\n", run<512>(a, clusters, fw));
    printf("  ... + without A loads        : %.3f ms\n", run<512 + 32>(a, clusters, fw));
    printf("  ... + without all of them    : %.3f ms\n", run<512 + 63>(a, clusters, fw));
    printf("  layer 0 has the priority     : %.3f ms\n", run<1024>(a, clusters, fw));
    printf("  dynamic priority of layer 0  : %.3f ms\n", run<2048>(a, clusters, fw));
    printf("  A fragments as 1 KB blocks   : %.3f ms\n", run<16384>(a, clusters, fw));
    printf("  A request before the weights : %.3f ms\n", run<32768>(a, clusters, fw));
    printf("  ... with members per XCD     : %.3f ms\n", run<32768 + 512>(a, clusters, fw));
    printf("  layer 0 boosted in its tail  : %.3f ms\n", run<8192>(a, clusters, fw));
    printf("  shipped again                : %.3f ms\n", run<0>(a, clusters, fw));
    return 0;
}
