// EXPERIMENT, not part of libfsn_hip.so: the sub-band recurrence laid out for one wave per SIMD.
// Included by tools/probe_rec.hip after fullsubnet_amd/csrc/lstm_kernels.hip; result in
// profiles/r01_gemm_probe.md (31.9 ms vs 29.5 ms for the shipped 3-waves-per-SIMD kernel: correct,
// not faster, and its 512 registers leave no room for the left-over tiles next to it).
#ifndef FSN_REC1_ABLATE
#define FSN_REC1_ABLATE 0  // probe-only bit mask: 1 no gx init loads, 2 no L2 touch-prefetch, 4 no activations,
#endif                     // 8 no B refill, 16 no hseq copy-out, 32 no MFMAs
namespace {
// ---------------------------------------------------------------------------------------------
// lstm_rec1_kernel: the same recurrence laid out for ONE wave per SIMD (4 waves per workgroup, one
// workgroup per CU), after the GEMM probe showed that MFMA-streaming waves sharing a SIMD lose a
// large part of the matrix pipe to instruction-by-instruction alternation.  Each wave owns H/4
// hidden units (UG = H/64 unit groups) x all four gates x all 16 RT rows and has the whole 512-entry
// register file: accumulators, the gx tile being added, c and one temporary stay in registers.
// Everything a wave waits for is requested one K-chunk ahead and pinned there with
// sched_barrier(0):
//   * A fragments (h_{t-1}) from LDS, double buffered;
//   * B fragments (W_hh) from L2, refilled IN PLACE right after the 4 RT MFMAs that consumed them,
//     walking one linear sequence gate 0..3 x chunk 0..KC-1 that wraps into the next time step;
//   * the gx tiles of a gate are requested when its K loop starts and added when it ends.
template <int H, int RT>
__global__ __launch_bounds__(256, 1) void lstm_rec1_kernel(const float* __restrict__ gx,
                                                            const float* __restrict__ whh_p,
                                                            float* __restrict__ hseq, int Tp, int Npad) {
    constexpr int UG = H / 64;   // unit groups per wave
    constexpr int KC = H / 16;   // k chunks == unit groups of the layer
    constexpr int CT = 4 * KC;
    constexpr int HS = H + 4;
    constexpr int ROWS = RT * 16;
    extern __shared__ __attribute__((aligned(16))) float hl[];  // [ROWS][HS]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const long n0 = (long)blockIdx.x * ROWS;

    f32x4 cst[RT][UG], tmp[RT][UG], acc[RT][UG], bfr0[UG], bfr1[UG], a_cur[RT], a_nxt[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int u = 0; u < UG; ++u) cst[rt][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < ROWS * HS; i += 256) hl[i] = 0.f;

    // per-lane offsets (floats): this wave's first fragment inside a [k chunk] slab of W_hh and
    // inside a gate's gx tile row; fragment (rt, u) sits a compile-time constant further on
    const int boff0 = wave * UG * KC * 256 + lane * 4;
    const int goff0 = wave * UG * 256 + lane * 4;
    const float* ap0 = hl + lr * HS + 4 * lq;
    // first two B fragment sets of the sequence (gate 0, chunks 0 and 1)
#pragma unroll
    for (int u = 0; u < UG; ++u) {
        bfr0[u] = *reinterpret_cast<const f32x4*>(whh_p + boff0 + u * KC * 256);
        bfr1[u] = *reinterpret_cast<const f32x4*>(whh_p + 256 + boff0 + u * KC * 256);
    }
    __syncthreads();

    for (int t = 0; t < Tp; ++t) {
        const float* gx_t = gx + (((long)t * Npad + n0) >> 4) * CT * 256;
#pragma unroll 1
        for (int g = 0; g < 4; ++g) {  // PyTorch gate order i, f, g, o
            const float* gx_g = gx_t + g * KC * 256;
            // accumulators start from the input projection
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int u = 0; u < UG; ++u)
#if FSN_REC1_ABLATE & 1
                    acc[rt][u] = f32x4{0.f, 0.f, 0.f, 0.f};
#else
                    acc[rt][u] = *reinterpret_cast<const f32x4*>(gx_g + goff0 + (rt * CT + u) * 256);
#endif
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) a_cur[rt] = *reinterpret_cast<const f32x4*>(ap0 + rt * 16 * HS);
            __builtin_amdgcn_sched_barrier(0);
            // one K chunk: 4 RT MFMAs per unit group, then that group's B slot is refilled in place
            // with the fragment two elements further on in the (gate, chunk) sequence
            auto chunk = [&](const f32x4(&ac)[RT], f32x4(&bs)[UG], const float* bsrc) {
#pragma unroll
                for (int u = 0; u < UG; ++u) {
#if !(FSN_REC1_ABLATE & 32)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) acc[rt][u] = mfma16(ac[rt][j], bs[u][j], acc[rt][u]);
#else
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) asm volatile("" ::"v"(ac[rt]), "v"(bs[u]));
#endif
                    __builtin_amdgcn_sched_barrier(0);
#if !(FSN_REC1_ABLATE & 8)
                    bs[u] = *reinterpret_cast<const f32x4*>(bsrc + boff0 + u * KC * 256);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            // (step 0 multiplies the zero-initialised h_{-1}: 0.5 % extra work, no branch)
#pragma unroll 1
            for (int kc = 0; kc < KC; kc += 2) {
                const int k2 = kc + 2 < KC ? kc + 2 : 0, k3 = kc + 3 < KC ? kc + 3 : 1;
                const int g2 = kc + 2 < KC ? g : ((g + 1) & 3);
                const float* bsrc2 = whh_p + ((long)g2 * KC * KC + k2) * 256;
                const float* bsrc3 = whh_p + ((long)g2 * KC * KC + k3) * 256;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    a_nxt[rt] = *reinterpret_cast<const f32x4*>(ap0 + (kc + 1) * 16 + rt * 16 * HS);
                __builtin_amdgcn_sched_barrier(0);
                chunk(a_cur, bfr0, bsrc2);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    a_cur[rt] = *reinterpret_cast<const f32x4*>(ap0 + k2 * 16 + rt * 16 * HS);
                __builtin_amdgcn_sched_barrier(0);
                chunk(a_nxt, bfr1, bsrc3);
            }
            // one uniform branch per gate, straight-line vector code inside
#if FSN_REC1_ABLATE & 4
            if (g == 3) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int u = 0; u < UG; ++u) tmp[rt][u] = acc[rt][u] * 1e-3f;
            }
#else
#define FSN_FOR_ALL(body)                          \
    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) \
    _Pragma("unroll") for (int u = 0; u < UG; ++u)    \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) { body; }
            if (g == 0) {
                FSN_FOR_ALL(tmp[rt][u][i] = sigmoid_fast(acc[rt][u][i]))
            } else if (g == 1) {
                FSN_FOR_ALL(cst[rt][u][i] = sigmoid_fast(acc[rt][u][i]) * cst[rt][u][i])
            } else if (g == 2) {
                FSN_FOR_ALL(cst[rt][u][i] = cst[rt][u][i] + tmp[rt][u][i] * tanh_fast(acc[rt][u][i]))
            } else {
                FSN_FOR_ALL(tmp[rt][u][i] = sigmoid_fast(acc[rt][u][i]) * tanh_fast(cst[rt][u][i]))
            }
#undef FSN_FOR_ALL
#endif
        }
        __syncthreads();  // every wave has finished reading h_{t-1}
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int u = 0; u < UG; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    hl[(rt * 16 + 4 * lq + i) * HS + (wave * UG + u) * 16 + lr] = tmp[rt][u][i];
        __syncthreads();  // h_t complete in LDS
#if !(FSN_REC1_ABLATE & 16)
        float* dst = hseq + ((long)t * Npad + n0) * H;
        for (int i = threadIdx.x; i < ROWS * (H / 4); i += 256) {
            const int row = i / (H / 4), c4 = i % (H / 4);
            *reinterpret_cast<f32x4*>(dst + (long)row * H + c4 * 4) =
                *reinterpret_cast<const f32x4*>(hl + row * HS + c4 * 4);
        }
#endif
    }
}

template <int H, int RT>
int launch_rec1(const float* gx, const float* whh_p, float* hseq, int Tp, int Npad, int main_wgs, hipStream_t s) {
    // at least 81 KB so that exactly one workgroup (= one wave per SIMD) is resident per CU
    size_t lds = (size_t)RT * 16 * (H + 4) * sizeof(float);
    if (lds < 84 * 1024) lds = 84 * 1024;
    auto kern = lstm_rec1_kernel<H, RT>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) {
        fsn_set_error("lstm_rec1: cannot reserve %zu bytes of LDS", lds);
        return FSN_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)main_wgs), dim3(256), lds, s, gx, whh_p, hseq, Tp, Npad);
    return fsn_check_launch("lstm_rec1_kernel");
}

}  // namespace
