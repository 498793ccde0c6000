// Full-band model (fullsubnet/model.py:95, sequence_model.py:106-125): both LSTM layers over all frames as ONE
// persistent launch.
//
// Why: the full-band LSTM has B <= 64 rows (one per utterance) and 2 x 512 hidden units - 0.1 TFLOP per batch - but
// its frames are a dependent chain, so it ran as a wavefront of T' + 1 launches of ~10.6 us each: 2.0 ms that no other
// work of the forward can hide (the sub-band input needs the mean of the WHOLE full-band output, model.py:110).  A
// launch boundary costs a drain, a dispatch and cold operand fetches; here the chain stays resident and a step costs one
// hand-off through memory (~3 us) plus its MFMAs:
//   - three stages of H / 4 = 128 workgroups each: L0 (layer 0: h0_t from gx0_t + h0_{t-1} W_hh0), L1x (feed-forward:
//     gx1_t = h0_t W_ih1 + b1, one step behind L0) and L1h (layer 1: h1_t from gx1_t + h1_{t-1} W_hh1).  Splitting
//     layer 1 in two keeps every stage at K = 512 per step and takes the input half of layer 1 off its recurrence;
//   - a workgroup owns FOUR hidden units = one 16-column MFMA tile (4 gates x 4 units) for all rows: its weight slice
//     (512 x 16 floats) is read ONCE into registers and stays there for all frames - the K loop has no weight traffic;
//   - wave w = row tile w (B > 32), or K is split over the waves (2 row tiles x 2 halves, 1 tile x 4 quarters) and the
//     partial tiles are summed in a fixed order through LDS;
//   - h_t goes to per-step buffers in A-fragment order (a wave's K chunk is one contiguous 1 KB block) with the
//     write-through / flag recipe of the CDNA guide (Guideline 16, R1: sc1 stores, every storing wave drains, ONE flag
//     store; one wave polls the 128 flags of the producing stage, barrier, sc1 loads).  No buffer is ever reused, so
//     there is no back-pressure and the dependence graph is acyclic: with all 384 workgroups resident (two per CU
//     fit) the launch cannot deadlock; every spin is bounded anyway (status raised, results garbage, never a hang).
#include "fsn_common.h"

namespace {

constexpr int CH = 512;         // hidden units per layer
constexpr int CKC = CH / 16;    // K chunks of an H-wide operand
constexpr int CNW = CH / 4;     // workgroups per stage
constexpr unsigned kChainSpin = 1u << 21;

struct ChainArgs {
    const float* gx0;     // layer-0 projection incl. bias, fragment order: tile (t * RT + rt, ct) = [64][4]
    const float* whh0_p;  // packed [4H/16][KC][64][4]
    const float* wih1_p;
    const float* whh1_p;
    const float* b1;      // [4H]
    float* hx0;           // [Tp][RT][KC][64][4]: h of layer 0 in A-fragment order
    float* hx1;           // likewise layer 1
    float* gx1;           // [Tp][CNW][4 waves][64][4]: partial projection tiles of layer 1
    float* hseq1;         // [Tp][Npad][H] row-major (the output layer's A operand)
    unsigned* flags;      // [3][CNW]: steps published by (stage, workgroup); stage 0 = L0, 1 = L1h, 2 = L1x
    unsigned* status;
    int Tp, RT, Npad;
};

// wave 0: all 128 flags of a stage >= epoch, and (optionally) one more flag >= its epoch
__device__ __forceinline__ bool chain_wait(const unsigned* flags, unsigned epoch, const unsigned* one, unsigned one_epoch,
                                           unsigned* status) {
    const int lane = threadIdx.x & 63;
    for (unsigned spins = 0;; ++spins) {
        bool ok = true;
        if (epoch > 0) {
            const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(flags) + lane,
                                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = (unsigned)v >= epoch && (unsigned)(v >> 32) >= epoch;
        }
        if (one && lane == 0) ok = ok && __hip_atomic_load(one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= one_epoch;
        if (__all((int)ok)) return true;
        if ((spins & 255u) == 255u) {
            const unsigned st = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (st != 0 || spins >= kChainSpin) {
                if (lane == 0 && st == 0) __hip_atomic_store(status, 1u + epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

__device__ __forceinline__ void chain_store(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sc1: write-through
}

// KS = K split over the waves: 1 (up to 4 row tiles), 2 (up to 2), 4 (one row tile)
template <int KS>
__global__ __launch_bounds__(256, 2) void fb_chain_kernel(const ChainArgs a) {
    constexpr int RTW = 4 / KS;    // row tiles a workgroup can hold
    constexpr int CW = CKC / KS;   // K chunks per wave
    constexpr int AD = CW < 16 ? CW : 16;  // A fragments in flight
    __shared__ f32x4 red[KS > 1 ? (KS - 1) * RTW * 64 : 1];

    const int stage = (int)blockIdx.x / CNW, j = (int)blockIdx.x % CNW;  // stage 2 (L1x) is dispatched last: it shares CUs
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rt = wave % RTW, kp = wave / RTW;
    const int lr = lane & 15, lq = lane >> 4, g = lr >> 2, ul = lr & 3;
    const int Tp = a.Tp, RT = a.RT;
    const bool active = rt < RT;

    unsigned* fl0 = a.flags;
    unsigned* fl1 = a.flags + CNW;
    unsigned* flx = a.flags + 2 * CNW;

    // this workgroup's weight slice: column tile = (gate g, units 4 j .. 4 j + 3), K chunks kp CW .. kp CW + CW - 1
    f32x4 wreg[CW];
    {
        const float* wsel = stage == 0 ? a.whh0_p : (stage == 1 ? a.whh1_p : a.wih1_p);
        const float* wp = wsel + ((size_t)(g * CKC + (j >> 2)) * CKC * 64 + lq * 16 + 4 * (j & 3) + ul) * 4;
#pragma unroll
        for (int q = 0; q < CW; ++q) wreg[q] = *reinterpret_cast<const f32x4*>(wp + (size_t)(kp * CW + q) * 256);
    }
    const float bias1 = a.b1[g * CH + 4 * j + ul];

    const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(a.hx0, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(a.hx1, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(a.gx1, 0, 0x7fffffff, 0x00020000);
    const unsigned lane16 = (unsigned)lane * 16u;
    // acc += h[tile (ts, rt)][K part kp] W: A fragments by sc1 buffer loads (written through by other CUs), AD in flight
    auto kpart = [&](f32x4 acc, const __amdgpu_buffer_rsrc_t& r, int ts) -> f32x4 {
        const unsigned base = (unsigned)((((size_t)ts * RT + rt) * CKC + kp * CW) * 1024);
        f32x4 ar[AD];
#pragma unroll
        for (int d = 0; d < AD; ++d)
            ar[d] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, lane16, base + d * 1024u, 16));
        __builtin_amdgcn_sched_barrier(0);  // all AD requests leave before the first MFMA waits for one of them
#pragma unroll
        for (int q = 0; q < CW; ++q) {
            const f32x4 av = ar[q % AD];
            if (q + AD < CW) {
                ar[q % AD] = __builtin_bit_cast(
                    f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, lane16, base + (unsigned)(q + AD) * 1024u, 16));
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc = mfma16(av[jj], wreg[q][jj], acc);
        }
        return acc;
    };
    // partial tiles of the K parts -> the kp = 0 wave of each row tile, summed in the fixed order 1, 2, 3
    auto reduce = [&](f32x4 acc) -> f32x4 {
        if (KS > 1) {
            if (kp > 0) red[((kp - 1) * RTW + rt) * 64 + lane] = acc;
            __syncthreads();
            if (kp == 0) {
#pragma unroll
                for (int p = 1; p < KS; ++p) {
                    const f32x4 o = red[((p - 1) * RTW + rt) * 64 + lane];
                    acc = f32x4{acc[0] + o[0], acc[1] + o[1], acc[2] + o[2], acc[3] + o[3]};
                }
            }
        }
        return acc;
    };
    auto publish = [&](unsigned* flag, unsigned epoch) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

    if (stage == 2) {
        // ---- L1x: gx1_t = h0_t W_ih1^T + b1, partial per K part (summed by L1h's reduction) ---------------------
        float* out = a.gx1 + ((size_t)j * 4 + wave) * 256 + lane * 4;
        for (int t = 0; t < Tp; ++t) {
            if (wave == 0) (void)chain_wait(fl0, (unsigned)t + 1, nullptr, 0, a.status);
            __syncthreads();
            if (active) {
                const float b = kp == 0 ? bias1 : 0.f;
                const f32x4 acc = kpart(f32x4{b, b, b, b}, r0, t);
                float* o = out + (size_t)t * CNW * 1024;
#pragma unroll
                for (int i = 0; i < 4; ++i) chain_store(o + i, acc[i]);
            }
            publish(flx + j, (unsigned)t + 1);
        }
        return;
    }

    // ---- L0 / L1h: one LSTM layer's recurrence for units 4 j .. 4 j + 3 ------------------------------------------
    const bool l1 = stage == 1;
    unsigned* flown = l1 ? fl1 : fl0;
    float* hx = l1 ? a.hx1 : a.hx0;
    // the projection tile of this lane: rows 4 lq + i, column (g, 4 j + ul) = lane lq 16 + 4 (j & 3) + ul of column
    // tile g KC + j / 4
    const float* gx0p = a.gx0 + ((size_t)(g * CKC + (j >> 2)) * 64 + lq * 16 + 4 * (j & 3) + ul) * 4;
    const size_t gx0_step = (size_t)RT * (4 * CKC) * 256;
    const bool owner = active && kp == 0;
    float c[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4 gxn = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!l1 && owner) gxn = *reinterpret_cast<const f32x4*>(gx0p + (size_t)rt * (4 * CKC) * 256);
    for (int t = 0; t < Tp; ++t) {
        f32x4 acc = gxn;
        if (!l1 && owner && t + 1 < Tp)
            gxn = *reinterpret_cast<const f32x4*>(gx0p + (size_t)(t + 1) * gx0_step + (size_t)rt * (4 * CKC) * 256);
        if (l1 || t > 0) {
            if (wave == 0) (void)chain_wait(flown, (unsigned)t, l1 ? flx + j : nullptr, (unsigned)t + 1, a.status);
            __syncthreads();
        }
        if (l1) {
            acc = f32x4{0.f, 0.f, 0.f, 0.f};
            if (active)
                acc = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                    rx, lane16, (unsigned)((((size_t)t * CNW + j) * 4 + wave) * 1024), 16));
        }
        if (active && t > 0) acc = kpart(acc, l1 ? r1 : r0, t - 1);
        acc = reduce(acc);
        if (owner) {
            float act[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) act[i] = g == 2 ? tanh_fast(acc[i]) : sigmoid_fast(acc[i]);
            // lanes ul (gate i) gather f, g, o of their unit from lanes ul + 4, + 8, + 12 of the same 16-lane row group
            float* hdst = hx + (((size_t)t * RT + rt) * CKC + (j >> 2)) * 256 + ((j & 3) * 16 + 4 * lq) * 4 + ul;
            float* hrow = a.hseq1 + ((size_t)t * a.Npad + rt * 16 + 4 * lq) * CH + 4 * j + ul;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float fg = __shfl(act[i], lane + 4, 64);
                const float gg = __shfl(act[i], lane + 8, 64);
                const float og = __shfl(act[i], lane + 12, 64);
                if (g == 0) {
                    const float cn = fg * c[i] + act[i] * gg;
                    c[i] = cn;
                    const float hv = og * tanh_fast(cn);
                    chain_store(hdst + i * 4, hv);
                    if (l1) hrow[(size_t)i * CH] = hv;
                }
            }
        }
        publish(flown + j, (unsigned)t + 1);
    }
}

}  // namespace

bool fsn_fb_chain_supported(int H, int Npad) { return H == CH && Npad >= 16 && Npad <= 64 && Npad % 16 == 0; }
size_t fsn_fb_chain_exchange_floats(int Tp, int Npad) {
    return (size_t)2 * Tp * Npad * CH + (size_t)Tp * CNW * 1024;  // hx0, hx1, gx1
}
size_t fsn_fb_chain_flag_words() { return (size_t)3 * CNW + 16; }

// gx0: fragment-order projection of layer 0 (bias included); hseq1 [Tp][Npad][H] row-major out.
int fsn_launch_fb_chain(const float* gx0, const float* whh0_p, const float* wih1_p, const float* whh1_p, const float* b1,
                        float* exchange, unsigned* flags, float* hseq1, int Tp, int Npad, int H, hipStream_t s) {
    if (!fsn_fb_chain_supported(H, Npad) || Tp < 1) {
        fsn_set_error("fb_chain: built for H = 512 and at most 64 rows");
        return FSN_ERR_ARG;
    }
    if (hipMemsetAsync(flags, 0, fsn_fb_chain_flag_words() * sizeof(unsigned), s) != hipSuccess) {
        fsn_set_error("fb_chain: cannot clear the flags");
        return FSN_ERR_LAUNCH;
    }
    ChainArgs a{};
    a.gx0 = gx0;
    a.whh0_p = whh0_p;
    a.wih1_p = wih1_p;
    a.whh1_p = whh1_p;
    a.b1 = b1;
    a.hx0 = exchange;
    a.hx1 = exchange + (size_t)Tp * Npad * CH;
    a.gx1 = exchange + (size_t)2 * Tp * Npad * CH;
    a.hseq1 = hseq1;
    a.flags = flags;
    a.status = flags + 3 * CNW;
    a.Tp = Tp;
    a.RT = Npad / 16;
    a.Npad = Npad;
    const dim3 grid(3 * CNW), block(256);
    if (a.RT == 1) hipLaunchKernelGGL(fb_chain_kernel<4>, grid, block, 0, s, a);
    else if (a.RT == 2) hipLaunchKernelGGL(fb_chain_kernel<2>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(fb_chain_kernel<1>, grid, block, 0, s, a);
    return fsn_check_launch("fb_chain_kernel");
}
