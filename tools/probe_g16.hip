// Timing probe for the 16-bit arithmetic's group kernels (lstm_group16_kernels.hip; not part of the library): the shipped
// kernels and ablations (template parameter ABL, see the kernels) on random operands at config 3's per-rank shape
// (32 clusters of 64 rows, 195 / 193 steps).   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe_g16.hip -o tools/bin/probe_g16
#include <cstdio>
#include <cstdlib>
#include "../fullsubnet_amd/csrc/lstm_group16_kernels.hip"
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
#define PROBE_AR 2
#endif
#ifndef PROBE_SV
#define PROBE_SV 0  // 1: the gates saved / the hand-off in 16 bits (FSN_ARITH_SAVES16, the default under AMP since round 6)
#endif
static unsigned* g_flags; static int g_clusters;
template <int ABL>
float run_fwd(G16FwdArgs a) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int it = 0; it < 3; ++it) {
        hipMemsetAsync(g_flags, 0, fsn_lstm2_g16_flag_words(g_clusters) * 4, 0);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((lstm2_g16_fwd_kernel<PROBE_AR, ABL, PROBE_SV>), dim3(g_clusters * QM * 2), dim3(256), 0, 0, a);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (it > 0 && ms < best) best = ms;
    }
    return best;
}
template <int ABL>
float run_bwd(G16BwdArgs a) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int it = 0; it < 3; ++it) {
        hipMemsetAsync(g_flags, 0, fsn_lstm2_g16_flag_words(g_clusters) * 4, 0);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((lstm2_g16_bwd_kernel<PROBE_AR, ABL, PROBE_SV>), dim3(g_clusters * QM * 2), dim3(256), 0, 0, a);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (it > 0 && ms < best) best = ms;
    }
    return best;
}
int main(int argc, char** argv) {
    const int Tp = argc > 1 ? atoi(argv[1]) : 195, clusters = argc > 2 ? atoi(argv[2]) : 32, N = clusters * 64 + 16;
    const int which = argc > 3 ? atoi(argv[3]) : 3;  // 1 forward, 2 bptt, 3 both
    g_clusters = clusters;
    const size_t TN = (size_t)Tp * N;
    float *x, *w, *bias, *h0, *h1, *sv0, *sv1, *dg, *dh1, *part; unsigned short* w16; unsigned* flags;
    hipMalloc(&x, TN * 32 * 4); hipMalloc(&w, (size_t)4 * QG * QH * 4); hipMalloc(&bias, 2 * QG * 4);
    hipMalloc(&h0, TN * QH * 4); hipMalloc(&h1, TN * QH * 4); hipMalloc(&sv0, TN * 5 * QH * 4); hipMalloc(&sv1, TN * 5 * QH * 4);
    hipMalloc(&dg, 2 * TN * QG * 4); hipMalloc(&dh1, TN * QH * 4); hipMalloc(&part, fsn_lstm2_g16_partial_floats(clusters) * 4);
    hipMalloc(&w16, (size_t)4 * QG * QH * 4); hipMalloc(&flags, fsn_lstm2_g16_flag_words(clusters) * 4);
    g_flags = flags;
    fill_kernel<<<1024, 256>>>(x, TN * 32, 1, 1.0f, 0.f);
    fill_kernel<<<1024, 256>>>(w, (size_t)4 * QG * QH, 2, 0.05f, 0.f);
    fill_kernel<<<64, 256>>>(bias, 2 * QG, 5, 0.05f, 0.f);
    fill_kernel<<<1024, 256>>>(sv0, TN * 5 * QH, 3, 0.4f, 0.5f);
    fill_kernel<<<1024, 256>>>(sv1, TN * 5 * QH, 4, 0.4f, 0.5f);
    fill_kernel<<<1024, 256>>>(dh1, TN * QH, 6, 0.01f, 0.f);
    hipDeviceSynchronize();
    unsigned short *p_a = w16, *p_b = p_a + (size_t)QG * 32, *p_c = p_b + (size_t)QG * QH, *p_d = p_c + (size_t)QG * QH;
#if PROBE_AR != 0
    if (which & 1) {
        g16_pack_fwd<PROBE_AR>(w, p_a, 32, 32, 0); g16_pack_fwd<PROBE_AR>(w, p_b, QH, QH, 0);
        g16_pack_fwd<PROBE_AR>(w + (size_t)QG * QH, p_c, QH, QH, 0); g16_pack_fwd<PROBE_AR>(w + (size_t)2 * QG * QH, p_d, QH, QH, 0);
        G16FwdArgs a{};
        a.x = x; a.x_step = N; a.w16 = w16; a.o_ih0 = 0; a.o_hh0 = (unsigned)((p_b - p_a) * 2); a.o_ih1 = (unsigned)((p_c - p_a) * 2);
        a.o_hh1 = (unsigned)((p_d - p_a) * 2); a.bias0 = bias; a.bias1 = bias + QG; a.hseq0 = h0; a.hseq1 = h1;
        a.gates0 = sv0; a.cseq0 = sv0 + TN * QG; a.gates1 = sv1; a.cseq1 = sv1 + TN * QG; a.flags = flags;
        a.status = flags + fsn_lstm2_g16_status_word(clusters); a.spin_ticks = 1ull << 31; a.Tp = Tp; a.Nrows = N;
        const float t0 = run_fwd<0>(a);
        unsigned st = 0; hipMemcpy(&st, a.status, 4, hipMemcpyDeviceToHost);
        printf("arithmetic %d: lstm2_g16_fwd_kernel, %d clusters, %d steps: %.3f ms = %.1f us per step, status %u\n", PROBE_AR, clusters, Tp, t0, 1e3 * t0 / Tp, st);
#define VF(abl, what) { const float t = run_fwd<abl>(a); printf("  %-60s: %.3f ms = %.1f us per step\n", what, t, 1e3 * t / Tp); }
        VF(64, "payload at device scope (what any other placement takes)");
        VF(128, "members (not clusters) share an XCD");
        VF(128 + 8, "... no saves");
        VF(8, "no saves");
        VF(4, "no weight loads");
        VF(2, "partners' tiles not loaded (constants staged)");
        VF(1, "no flag waits");
        VF(8 + 4, "no saves, no weight loads");
        VF(8 + 4 + 2, "... and no tile loads");
        VF(8 + 4 + 2 + 1, "... and no flag waits");
        VF(8 + 4 + 2 + 1 + 16, "... and no h stores (K loops, LDS traffic, barriers, cell)");
        VF(0, "shipped again");
    }
#endif
    if (which & 2) {
        const size_t wb = (size_t)QG * QH * 2;
        unsigned char* wp = reinterpret_cast<unsigned char*>(w16);
        g16_pack_bptt<PROBE_AR>(w, wp, 0); g16_pack_bptt<PROBE_AR>(w + (size_t)QG * QH, wp + wb, 0);
        g16_pack_bptt<PROBE_AR>(w + (size_t)2 * QG * QH, wp + 2 * wb, 0);
        fill_kernel<<<1024, 256>>>(sv0, TN * 5 * QH, 3, 0.4f, 0.5f);
        fill_kernel<<<1024, 256>>>(sv1, TN * 5 * QH, 4, 0.4f, 0.5f);
        G16BwdArgs b{};
        b.dh1 = dh1; b.w16 = w16; b.o_hh1 = 0; b.o_ih1 = (unsigned)wb; b.o_hh0 = (unsigned)(2 * wb);
        b.gates0 = sv0; b.cseq0 = sv0 + TN * QG; b.gates1 = sv1; b.cseq1 = sv1 + TN * QG; b.dg1 = dg; b.dg0 = dg + TN * QG;
        b.x1 = part; b.x0 = reinterpret_cast<unsigned char*>(part) + (size_t)clusters * QDX * q_xslot<PROBE_AR>();
        unsigned short* dg16; float* dbp; hipMalloc(&dg16, 2 * TN * QG * 2); hipMalloc(&dbp, (size_t)2 * clusters * QG * 4);
        b.dg16_0 = dg16; b.dg16_1 = dg16 + TN * QG; b.dbp = dbp; b.dg1_f32 = 0;
        b.flags = flags; b.status = flags + fsn_lstm2_g16_status_word(clusters); b.spin_ticks = 1ull << 31; b.Tp = Tp; b.Nrows = N;
        const float t0 = run_bwd<0>(b);
        unsigned st = 0; hipMemcpy(&st, b.status, 4, hipMemcpyDeviceToHost);
        printf("arithmetic %d: lstm2_g16_bwd_kernel, %d clusters, %d steps: %.3f ms = %.1f us per step, status %u\n", PROBE_AR, clusters, Tp, t0, 1e3 * t0 / Tp, st);
#define VB(abl, what) { const float t = run_bwd<abl>(b); printf("  %-60s: %.3f ms = %.1f us per step\n", what, t, 1e3 * t / Tp); }
        VB(128, "members (not clusters) share an XCD");
        VB(8, "no gate-gradient stores");
        VB(2, "saved activations not loaded");
        VB(4, "no weight loads");
        VB(16, "no exchange stores");
        VB(32, "exchanged operand not loaded");
        VB(1, "no flag waits");
        VB(8 + 2, "no saved loads, no gate-gradient stores");
        VB(8 + 2 + 16 + 32, "... and no exchange traffic");
        VB(8 + 2 + 16 + 32 + 4, "... and no weight loads");
        VB(8 + 2 + 16 + 32 + 4 + 1, "... and no flag waits (K loops, LDS, barriers, cell derivative)");
        VB(0, "shipped again");
    }
    return 0;
}
