// EXPERIMENT for the next round - not part of the library, never measured yet.
// lstm_rec_kernel<H, RT, UG, false> (precomputed projection, hidden sequence stored) with two changes:
//   (a) the projection tiles of pass p + 1 are fetched into the accumulator tile acc[rt][u] as soon as the cell
//       update of pass p has consumed it (same registers, no extra ones), instead of all of them at the start of
//       pass p + 1 where the first MFMA waits for HBM with every sibling wave in the same state; the tiles of the
//       next step's first pass are fetched the same way during the last update of this step;
//   (b) the two barriers of a step only order LDS traffic (fence on the "local" address space): a __syncthreads()
//       also waits for vmcnt(0), i.e. for the prefetch of (a).
// Included by tools/probe_rec.hip after lstm_kernels.hip (f32x4, mfma16, sigmoid_fast, tanh_fast come from there);
// the probe times it against the shipped kernel and compares the stored hidden sequences bit for bit.
namespace {

// SKEW (third experiment): the three waves a SIMD holds leave the step barrier together and reach
// every pass boundary together, so their cell updates (VALU / transcendental work, ~0.8 us per wave and pass) run back
// to back with the matrix pipe idle.  SKEW > 0 delays wave slot k of a SIMD by k * SKEW * 64 clocks after the barrier
// (s_sleep): the updates of one wave then fall under the MFMAs of the other two.
template <int H, int RT, int UG, int SKEW>
__global__ __launch_bounds__((H / (16 * UG)) * 64) void lstm_rec_pf_kernel(const float* __restrict__ gx,
                                                                           const float* __restrict__ whh_p,
                                                                           float* __restrict__ hseq, int Tp, int Npad) {
    constexpr int NW = H / (16 * UG);
    constexpr int KC = H / 16;
    constexpr int CT = 4 * KC;
    constexpr int HS = H + 4;
    constexpr int ROWS = RT * 16;
    constexpr int UNR = RT >= 3 ? 1 : (RT == 2 ? 2 : 4);
    extern __shared__ __attribute__((aligned(16))) float hl[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const long n0 = (long)blockIdx.x * ROWS;

    float cst[RT][UG][4], tmp[RT][UG][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int u = 0; u < UG; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) cst[rt][u][i] = 0.f;
    for (int i = threadIdx.x; i < ROWS * HS; i += NW * 64) hl[i] = 0.f;

    // projection tile (step t, gate g, row tile rt, unit group of this wave + u) in accumulator-fragment order
    auto gx_tile = [&](int t, int g, int rt, int u) -> const f32x4* {
        const long rt0 = ((long)t * Npad + n0) >> 4;
        return reinterpret_cast<const f32x4*>(gx + (((rt0 + rt) * CT + g * KC + wave * UG + u) * 64 + lane) * 4);
    };
    f32x4 acc[RT][UG];  // carried round the loops: always holds the NEXT pass's projection tiles at a pass boundary
    {
        int g0 = 1;
        asm volatile("" : "+s"(g0));
#pragma unroll
        for (int u = 0; u < UG; ++u)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt][u] = *gx_tile(0, g0, rt, u);
    }
    __syncthreads();

    for (int t = 0; t < Tp; ++t) {
        if (SKEW > 0) {  // waves w, w + 4, w + 8 share SIMD w % 4: slot = wave / 4
            if (wave >= 8) __builtin_amdgcn_s_sleep(2 * SKEW);
            else if (wave >= 4) __builtin_amdgcn_s_sleep(SKEW);
        }
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            __builtin_amdgcn_sched_barrier(0);
            int g = pass == 0 ? 1 : (pass == 1 ? 0 : pass);          // f, i, g, o
            int gn = pass == 0 ? 0 : (pass == 1 ? 2 : (pass == 2 ? 3 : 1));  // the gate after this one
            asm volatile("" : "+s"(g));
            asm volatile("" : "+s"(gn));
            const int tn = pass == 3 ? t + 1 : t;                     // ... and its step
            const bool more = tn < Tp;
            unsigned bo[UG];
#pragma unroll
            for (int u = 0; u < UG; ++u) bo[u] = (unsigned)(((g * KC + wave * UG + u) * KC * 64 + lane) * 4);
            if (t > 0) {
                f32x4 bn[UG];
#pragma unroll
                for (int u = 0; u < UG; ++u) bn[u] = *reinterpret_cast<const f32x4*>(whh_p + bo[u]);
#pragma unroll UNR
                for (int kc = 0; kc < KC; ++kc) {
                    f32x4 bc[UG];
#pragma unroll
                    for (int u = 0; u < UG; ++u) bc[u] = bn[u];
                    if (kc + 1 < KC) {
#pragma unroll
                        for (int u = 0; u < UG; ++u)
                            bn[u] = *reinterpret_cast<const f32x4*>(whh_p + (bo[u] + (unsigned)(kc + 1) * 256u));
                    }
                    const float* ap = hl + lr * HS + kc * 16 + 4 * lq;
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(ap + rt * 16 * HS);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int u = 0; u < UG; ++u) acc[rt][u] = mfma16(a[j], bc[u][j], acc[rt][u]);
                    }
                }
            }
            // cell update, one accumulator tile at a time; the tile's registers then take the next pass's projection
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int u = 0; u < UG; ++u) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float x = acc[rt][u][i];
                        if (pass == 0) cst[rt][u][i] = sigmoid_fast(x) * cst[rt][u][i];
                        else if (pass == 1) tmp[rt][u][i] = sigmoid_fast(x);
                        else if (pass == 2) cst[rt][u][i] = cst[rt][u][i] + tmp[rt][u][i] * tanh_fast(x);
                        else tmp[rt][u][i] = sigmoid_fast(x) * tanh_fast(cst[rt][u][i]);
                        if (pass == 0 || pass == 2) asm volatile("" : "+v"(cst[rt][u][i]));
                        else asm volatile("" : "+v"(tmp[rt][u][i]));
                    }
                    if (more) acc[rt][u] = *gx_tile(tn, gn, rt, u);
                    __builtin_amdgcn_sched_barrier(0);  // the fetch stays here, under the remaining updates
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        // every wave has finished reading h_{t-1} (its LDS reads were waited for before the MFMAs that used them)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int u = 0; u < UG; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    hl[(rt * 16 + 4 * lq + i) * HS + (wave * UG + u) * 16 + lr] = tmp[rt][u][i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();  // h_t complete in LDS
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        float* dst = hseq + ((long)t * Npad + n0) * H;
        for (int i = threadIdx.x; i < ROWS * (H / 4); i += NW * 64) {
            const int row = i / (H / 4), c4 = i % (H / 4);
            *reinterpret_cast<f32x4*>(dst + (long)row * H + c4 * 4) = *reinterpret_cast<const f32x4*>(hl + row * HS + c4 * 4);
        }
    }
}

template <int H, int RT, int UG = 2, int SKEW = 0>
int launch_rec_pf(const float* gx, const float* whh_p, float* hseq, int Tp, int Npad, int wgs, hipStream_t s) {
    constexpr int NW = H / (16 * UG);
    const size_t lds = (size_t)RT * 16 * (H + 4) * sizeof(float);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_rec_pf_kernel<H, RT, UG, SKEW>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return -3;
    hipLaunchKernelGGL((lstm_rec_pf_kernel<H, RT, UG, SKEW>), dim3((unsigned)wgs), dim3(NW * 64), lds, s, gx, whh_p, hseq, Tp, Npad);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace
