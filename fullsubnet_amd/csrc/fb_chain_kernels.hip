// Full-band model (fullsubnet/model.py:95, sequence_model.py:106-125): both LSTM layers over all frames as ONE
// persistent launch.
//
// Why: the full-band LSTM has B <= 64 rows (one per utterance) and 2 x 512 hidden units - 0.1 TFLOP per batch - but
// its frames are a dependent chain, so it ran as a wavefront of T' + 1 launches of ~10.6 us each: 2.0 ms that no other
// work of the forward can hide (the sub-band input needs the mean of the WHOLE full-band output, model.py:110).  A
// launch boundary costs a drain, a dispatch and cold operand fetches; here the chain stays resident and a step costs
// its MFMAs plus one hand-off through memory:
//   - two stages of H / 4 workgroups (128 at H = 512; the kernel is a template over H, built for 512 and 384), one
//     workgroup per CU: L0 (layer 0) and L1 (layer 1's recurrence).  A workgroup owns FOUR hidden units = one 16-column
//     MFMA tile (4 gates x 4 units) for all rows; its weight slices (H x 16 floats each) are read ONCE into registers
//     and stay there for all frames - no weight traffic at all;
//   - the input half of layer 1 (h0_t W_ih1 + b1) has the same A operand as layer 0's recurrence at step t + 1
//     (h0_t W_hh0): the L0 workgroup computes both from one set of A fragments - its own gates first (critical path),
//     cell update, h0 store, flag, and then layer 1's projection tile while the partners' flags are on their way.
//     Layer 1 is left with K = 512 per step like layer 0;
//   - wave w = row tile w (B > 32), or K is split over the waves (2 row tiles x 2 halves, 1 tile x 4 quarters) and the
//     partial tiles are summed in a fixed order through LDS;
//   - h_t goes to per-step buffers in A-fragment order (a wave's K chunk is one contiguous 1 KB block) with the
//     write-through / flag recipe of the CDNA guide (Guideline 16, R1: sc1 stores, every storing wave drains, ONE flag
//     store per copy; one wave polls the 128 flags of the producing stage, barrier, sc1 loads).  No buffer is ever
//     reused, so there is no back-pressure and the dependence graph is acyclic: with all 256 workgroups resident the
//     launch cannot deadlock; every spin is bounded anyway (status raised, never a hang; the host then turns the
//     output into NaN, fsn_launch_poison_if).  The host serialises persistent launches of different streams
//     (fsn_api.hip) so that two of them never share the CUs.  Up to 4095 steps (the reach of a buffer resource).
// Measured with tools/probe_chain.hip (190 steps): see DESIGN.md 4.6.
#include "fsn_common.h"

namespace {

constexpr int CHMAX = 512;      // largest hidden size built (384 and 512 are)
constexpr int CFS = CHMAX / 4;  // words per copy of a stage's flag array (H / 4 workgroups per stage use it)
constexpr int CREP = 16;        // copies of a stage's flag array: a poller reads copy (its index % CREP)

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
    float* hseq0;         // training (SAVE): layer 0's hidden sequence, row-major
    float* gates0;        // training: activated gates i | f | g | o of layer 0, [Tp][Npad][4H]
    float* cseq0;         // training: cell sequence of layer 0, [Tp][Npad][H]
    float* gates1;
    float* cseq1;
    unsigned* flags;      // [2][CREP][CFS] steps published by (stage 0 = L0 / 1 = L1, workgroup), CREP copies
    unsigned* status;
    unsigned long long spin_ticks;  // wait bound (fsn_spin_ticks)
    int Tp, RT, Npad;
};

// wave 0: all 128 flags of a stage >= epoch and (optionally) one more flag >= its epoch, both looked at in the same
// round trip; bounded
__device__ __forceinline__ bool chain_wait(const unsigned* flags, int nflags, unsigned epoch, const unsigned* one,
                                           unsigned one_epoch, unsigned* status, unsigned long long ticks) {
    const int lane = threadIdx.x & 63;
    const unsigned long long* f = reinterpret_cast<const unsigned long long*>(flags) + lane;
    unsigned long long t0 = 0;
    for (unsigned spins = 0;; ++spins) {
        unsigned long long v = ~0ull;
        unsigned w = ~0u;
        if (epoch > 0 && 2 * lane < nflags) v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (one && lane == 0) w = __hip_atomic_load(one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all((int)((unsigned)v >= epoch && (unsigned)(v >> 32) >= epoch && w >= one_epoch))) return true;
        if ((spins & 255u) == 255u && fsn_wait_give_up(status, spins, t0, ticks, 1u + epoch)) return false;
    }
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// 16-byte write-through store (sc1): whole 16-byte groups, never single dwords - a step of the chain publishes ~0.8 MB,
// and as dword stores that was 200 k partial-line write transactions per step (measured: 2 us of a 9.6 us step)
__device__ __forceinline__ void chain_store16(const __amdgpu_buffer_rsrc_t& r, unsigned voff, unsigned soff, f32x4 v,
                                              bool plain) {
    if (plain) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0);
    else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 16);  // aux 16 = sc1
}
// value of quad lane Q (lanes 4 k .. 4 k + 3 form a quad) in every lane of the quad
template <int Q>
__device__ __forceinline__ float quad_bcast(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), Q * 0x55, 0xf, 0xf, true));
}

// KS = K split over the waves: 1 (up to 4 row tiles), 2 (up to 2), 4 (one row tile)
// ABL: experiment knob of tools/probe_chain.hip (0 in the library; any bit set gives WRONG results): 1 no flag polling,
// 2 no h / projection stores, 4 A fragments not loaded, 8 plain instead of write-through stores, 16 no drain before
// the flag store, 32 no flag stores, 64 no layer-1 projection in L0 (L1 does not wait for it)
// SAVE: the training form - every step also keeps the activated gates, the cell state and layer 0's hidden sequence
// (fsn_lstm_layer_backward's inputs, the layouts of fsn_lstm_layer_forward)
// CELL: 0 = LSTM (gate tile i | f | g | o); 1 = GRU (audio_zen/model/module/sequence_model.py:59-66) written as a FOUR-gate cell
// so that every product, hand-off and buffer of the chain stays as it is: tile r | z | nx | nh with
//   r = sigmoid(W_ir x + b_ir + W_hr h + b_hr), z likewise, nx = W_in x + b_in (no recurrent part), nh = W_hn h + b_hn (no
//   input part), n = tanh(nx + r nh), h' = n + z (h - n)
// - the caller expands nn.GRU's [3H] gate rows to [4H] with zero blocks (fsn_gru2_forward); the cell state registers hold h.
template <int CH, int KS, int ABL = 0, bool SAVE = false, int CELL = 0>
__global__ __launch_bounds__(256, 1) void fb_chain_kernel(const ChainArgs a) {
    static_assert(!(SAVE && CELL != 0), "the training form is built for the LSTM cell");
    constexpr int CKC = CH / 16;   // K chunks of an H-wide operand (and column tiles per gate)
    constexpr int CNW = CH / 4;    // workgroups per stage
    static_assert(CKC % 4 == 0 && CNW % 2 == 0 && CNW <= CFS, "hidden size: a multiple of 64, at most 512");
    constexpr int RTW = 4 / KS;    // row tiles a workgroup can hold
    constexpr int CW = CKC / KS;   // K chunks per wave
    __shared__ f32x4 red[KS > 1 ? (KS - 1) * RTW * 64 : 1];

    const int stage = (int)blockIdx.x / CNW, j = (int)blockIdx.x % CNW;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rt = wave % RTW, kp = wave / RTW;
    const int lr = lane & 15, lq = lane >> 4, g = lr >> 2, ul = lr & 3;
    const int Tp = a.Tp, RT = a.RT;
    const bool active = rt < RT, owner = active && kp == 0;

    // Every workgroup polls ALL flags of a stage.  With one copy of the flags that is hundreds of pollers on the same
    // four cache lines (one memory channel each): measured, 128 extra pollers doubled the step time.  So a producer
    // writes CREP copies (one store instruction, CREP lanes) and a consumer polls copy (its index % CREP).
    unsigned* fl0 = a.flags;               // [CREP][CFS]
    unsigned* fl1 = a.flags + CREP * CFS;  // [CREP][CFS]
    const int rep = j % CREP;

    const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(a.hx0, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(a.hx1, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(a.gx1, 0, 0x7fffffff, 0x00020000);
    const unsigned lane16 = (unsigned)lane * 16u;

    // weight slice of column tile (gate g, units 4 j .. 4 j + 3), K chunks kp CW .. kp CW + CW - 1, from the packed
    // [4H/16][KC][64][4] order: this lane's fragment of chunk kc is one 16-byte group
    auto load_w = [&](const float* packed, f32x4 (&w)[CW]) {
        const float* wp = packed + ((size_t)(g * CKC + (j >> 2)) * CKC * 64 + lq * 16 + 4 * (j & 3) + ul) * 4;
#pragma unroll
        for (int q = 0; q < CW; ++q) w[q] = *reinterpret_cast<const f32x4*>(wp + (size_t)(kp * CW + q) * 256);
    };
    // all A fragments of tile (ts, rt), K part kp: sc1 buffer loads (the producers wrote through), all in flight at once
    auto load_a = [&](f32x4 (&ar)[CW], const __amdgpu_buffer_rsrc_t& r, int ts) {
        const unsigned base = (unsigned)((((size_t)ts * RT + rt) * CKC + kp * CW) * 1024);
#pragma unroll
        for (int q = 0; q < CW; ++q) {
            if (ABL & 4) ar[q] = f32x4{0.5f, 0.25f, -0.125f, 0.0625f};
            else ar[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, lane16, base + q * 1024u, 16));
        }
        __builtin_amdgcn_sched_barrier(0);  // every request leaves before the first MFMA waits for one of them
    };
    auto mac = [&](f32x4 acc, const f32x4 (&ar)[CW], const f32x4 (&w)[CW]) -> f32x4 {
#pragma unroll
        for (int q = 0; q < CW; ++q)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc = mfma16(ar[q][jj], w[q][jj], acc);
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
    // every wave drains its stores, then one flag store per copy
    auto publish = [&](unsigned* flags, unsigned epoch) {
        if (!(ABL & 16)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if ((int)threadIdx.x < CREP && !(ABL & 32))
            __hip_atomic_store(flags + (size_t)threadIdx.x * CFS + j, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto hstore = [&](const __amdgpu_buffer_rsrc_t& r, unsigned voff, unsigned soff, f32x4 v) {
        if (!(ABL & 2)) chain_store16(r, voff, soff, v, (ABL & 8) != 0);
    };
    // lane (u', lq) of a quad holds v[i] = x[row 4 lq + i][unit u']: 4 x 4 transpose inside the quad ->
    // x[row 4 lq + ul][units 0..3], one 16-byte group of a row-major (or A-fragment) buffer per lane
    auto quad_transpose = [&](const float (&v)[4]) -> f32x4 {
        f32x4 o;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float t0 = u == 0 ? quad_bcast<0>(v[0]) : u == 1 ? quad_bcast<1>(v[0]) : u == 2 ? quad_bcast<2>(v[0]) : quad_bcast<3>(v[0]);
            const float t1 = u == 0 ? quad_bcast<0>(v[1]) : u == 1 ? quad_bcast<1>(v[1]) : u == 2 ? quad_bcast<2>(v[1]) : quad_bcast<3>(v[1]);
            const float t2 = u == 0 ? quad_bcast<0>(v[2]) : u == 1 ? quad_bcast<1>(v[2]) : u == 2 ? quad_bcast<2>(v[2]) : quad_bcast<3>(v[2]);
            const float t3 = u == 0 ? quad_bcast<0>(v[3]) : u == 1 ? quad_bcast<1>(v[3]) : u == 2 ? quad_bcast<2>(v[3]) : quad_bcast<3>(v[3]);
            o[u] = ul == 0 ? t0 : ul == 1 ? t1 : ul == 2 ? t2 : t3;
        }
        return o;
    };
    const int hrow = 4 * lq + ul;                                     // the row a lane stores after the transpose
    // LSTM cell of this wave's 16 rows x 4 units from the gate tile; returns h[row 4 lq + ul][units 0..3] (lanes g = 0).
    // SAVE: activated gates -> gates_t [Npad][4H] (every lane: its gate, 4 units of row 4 lq + ul), c_t -> cseq_t.
    auto cell = [&](f32x4 acc, float (&c)[4], float* gates_t, float* cseq_t) -> f32x4 {
        float act[4], hq[4];
        if constexpr (CELL == 1) {  // GRU: lanes g = 0 hold r and gather z, nx, nh of their unit; c[] is h_{t-1} of the unit
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = g < 2 ? sigmoid_f(acc[i]) : acc[i];
                const float z = __shfl(v, lane + 4, 64);
                const float nx = __shfl(v, lane + 8, 64);
                const float nh = __shfl(v, lane + 12, 64);
                const float n = tanhf(nx + v * nh);
                const float hn = n + z * (c[i] - n);
                c[i] = hn;
                hq[i] = hn;
            }
            return quad_transpose(hq);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) act[i] = g == 2 ? tanh_fast(acc[i]) : sigmoid_fast(acc[i]);
        // lanes ul (gate i, g = 0) gather f, g, o of their unit from lanes ul + 4, + 8, + 12 of the same 16-lane row
        // group; the other lanes compute along on garbage (no branch)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float fg = __shfl(act[i], lane + 4, 64);
            const float gg = __shfl(act[i], lane + 8, 64);
            const float og = __shfl(act[i], lane + 12, 64);
            const float cn = fg * c[i] + act[i] * gg;
            c[i] = cn;
            hq[i] = og * tanh_fast(cn);
        }
        if (SAVE) {
            const size_t row = (size_t)rt * 16 + hrow;
            *reinterpret_cast<f32x4*>(gates_t + row * (4 * CH) + g * CH + 4 * j) = quad_transpose(act);
            const f32x4 cv = quad_transpose(c);
            if (g == 0) *reinterpret_cast<f32x4*>(cseq_t + row * CH + 4 * j) = cv;
        }
        return quad_transpose(hq);
    };
    const unsigned hvoff = (unsigned)(((j & 3) * 16 + hrow) * 16);    // its 16-byte group inside chunk j / 4
    auto hsoff = [&](int t) { return (unsigned)((((size_t)t * RT + rt) * CKC + (j >> 2)) * 1024); };
    const unsigned gx1_wave = (unsigned)(((size_t)j * 4 + wave) * 1024);  // this wave's projection tile inside a step
    const unsigned gx1_step = (unsigned)CNW * 4096u;
    float c[4] = {0.f, 0.f, 0.f, 0.f};

    if (stage == 0) {
        // ---- L0: layer 0's recurrence for units 4 j .. 4 j + 3, and layer 1's input projection for ITS units 4 j .. --
        f32x4 whh[CW], wih[CW], ar[CW];
        load_w(a.whh0_p, whh);
        load_w(a.wih1_p, wih);
        const float bias1 = a.b1[g * CH + 4 * j + ul];
        // the layer-0 projection tile of this lane: rows 4 lq + i, column (g, 4 j + ul) = lane lq 16 + 4 (j & 3) + ul of
        // column tile g KC + j / 4
        const float* gx0p = a.gx0 + ((size_t)(g * CKC + (j >> 2)) * 64 + lq * 16 + 4 * (j & 3) + ul) * 4 +
                            (size_t)rt * (4 * CKC) * 256;
        const size_t gx0_step = (size_t)RT * (4 * CKC) * 256;
        f32x4 gxn = f32x4{0.f, 0.f, 0.f, 0.f};
        if (owner) gxn = *reinterpret_cast<const f32x4*>(gx0p);
        // iteration t: h0_t (t < Tp) and the projection of h0_{t-1} (t > 0), both from the A fragments of h0_{t-1}
        for (int t = 0; t <= Tp; ++t) {
            f32x4 acc = gxn;
            if (owner && t + 1 < Tp) gxn = *reinterpret_cast<const f32x4*>(gx0p + (size_t)(t + 1) * gx0_step);
            if (t > 0) {
                if (wave == 0 && !(ABL & 1)) (void)chain_wait(fl0 + rep * CFS, CNW, (unsigned)t, nullptr, 0, a.status, a.spin_ticks);
                __syncthreads();
                if (active) load_a(ar, r0, t - 1);
            }
            if (t < Tp) {
                if (active && t > 0) acc = mac(acc, ar, whh);
                acc = reduce(acc);
                if (owner) {
                    const f32x4 hv = cell(acc, c, SAVE ? a.gates0 + (size_t)t * a.Npad * 4 * CH : nullptr,
                                          SAVE ? a.cseq0 + (size_t)t * a.Npad * CH : nullptr);
                    if (g == 0) {
                        hstore(r0, hvoff, hsoff(t), hv);
                        if (SAVE) *reinterpret_cast<f32x4*>(a.hseq0 + ((size_t)t * a.Npad + rt * 16 + hrow) * CH + 4 * j) = hv;
                    }
                }
            }
            // Layer 1's projection tile of step t - 1 (partial per K part; L1's reduction sums the parts) AFTER the flag
            // of this step: its MFMAs run while the partners' flags are on their way (measured: before the flag they
            // cost their full 1.7 us at 64 rows - a write-through store is acknowledged faster than that).  The tile's
            // store is awaited by the next drain: tile s is complete once this workgroup has published s + 3.  Inline
            // asm: a store the compiler sees makes every later wait for a load a wait for ALL memory operations.
            const bool proj = t > 0 && active && !(ABL & 64);
            const float b = kp == 0 ? bias1 : 0.f;
            if (t < Tp) publish(fl0, (unsigned)t + 1);
            if (proj) {
                const f32x4 accx = mac(f32x4{b, b, b, b}, ar, wih);
                const unsigned so = (unsigned)(t - 1) * gx1_step + gx1_wave;
                if (ABL & 2) {
                } else if (ABL & 8) {
                    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" ::"v"(accx), "v"(lane16), "s"(rx), "s"(so) : "memory");
                } else {
                    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen sc1" ::"v"(accx), "v"(lane16), "s"(rx), "s"(so) : "memory");
                }
            }
            if (t == Tp) publish(fl0, (unsigned)Tp + 2);
        }
        return;
    }

    // ---- L1: layer 1's recurrence for units 4 j .. 4 j + 3 ------------------------------------------------------------
    f32x4 whh[CW], ar[CW];
    load_w(a.whh1_p, whh);
    for (int s = 0; s < Tp; ++s) {
        // h1_{s-1} of all workgroups, and the projection tile of step s from L0 workgroup j (complete at flag s + 3)
        if (wave == 0 && !(ABL & 1))
            (void)chain_wait(fl1 + rep * CFS, CNW, (unsigned)s, (ABL & 64) ? nullptr : fl0 + rep * CFS + j, (unsigned)s + 3,
                             a.status, a.spin_ticks);
        __syncthreads();
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        if (active) {
            acc = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, lane16, (unsigned)s * gx1_step + gx1_wave, 16));
            if (s > 0) {
                load_a(ar, r1, s - 1);
                acc = mac(acc, ar, whh);
            }
        }
        acc = reduce(acc);
        if (owner) {
            const f32x4 hv = cell(acc, c, SAVE ? a.gates1 + (size_t)s * a.Npad * 4 * CH : nullptr,
                                  SAVE ? a.cseq1 + (size_t)s * a.Npad * CH : nullptr);
            if (g == 0) {
                hstore(r1, hvoff, hsoff(s), hv);
                *reinterpret_cast<f32x4*>(a.hseq1 + ((size_t)s * a.Npad + rt * 16 + hrow) * CH + 4 * j) = hv;
            }
        }
        publish(fl1, (unsigned)s + 1);
    }
}

}  // namespace

// H = 384 or 512, up to 64 rows, and a device with one CU per workgroup (2 x H / 4: the 64-row variant needs a CU's
// whole register file)
namespace {
template <int CH>
bool chain_grid_fits(int RT) {
    const unsigned grid = 2 * (CH / 4);
    const void *k, *ks;
    if (RT == 1) k = (const void*)fb_chain_kernel<CH, 4, 0, false>, ks = (const void*)fb_chain_kernel<CH, 4, 0, true>;
    else if (RT == 2) k = (const void*)fb_chain_kernel<CH, 2, 0, false>, ks = (const void*)fb_chain_kernel<CH, 2, 0, true>;
    else k = (const void*)fb_chain_kernel<CH, 1, 0, false>, ks = (const void*)fb_chain_kernel<CH, 1, 0, true>;
    const void* kg = RT == 1   ? (const void*)fb_chain_kernel<CH, 4, 0, false, 1>
                     : RT == 2 ? (const void*)fb_chain_kernel<CH, 2, 0, false, 1>
                               : (const void*)fb_chain_kernel<CH, 1, 0, false, 1>;
    return fsn_grid_fits(k, 256, grid) && fsn_grid_fits(ks, 256, grid) && fsn_grid_fits(kg, 256, grid);
}
}  // namespace
bool fsn_fb_chain_supported(int H, int Npad) {
    if ((H != 512 && H != 384) || Npad < 16 || Npad > 64 || Npad % 16 != 0) return false;
    if (!fsn_persistent_allowed()) return false;
    // residency contract: all 2 x H / 4 workgroups at once, by the compiled kernels' own occupancy on this device
    return H == 512 ? chain_grid_fits<512>(Npad / 16) : chain_grid_fits<384>(Npad / 16);
}
// the per-step hand-off buffers are addressed through buffer resources, whose offsets reach 2 GB: 4095 steps
int fsn_fb_chain_max_steps() { return (int)(0x7fffffffu / ((unsigned)(CHMAX / 4) * 4096u)); }
size_t fsn_fb_chain_exchange_floats(int Tp, int Npad) {
    return (size_t)2 * Tp * Npad * CHMAX + (size_t)Tp * (CHMAX / 4) * 1024;  // hx0, hx1, gx1 (sized for H = 512)
}
size_t fsn_fb_chain_flag_words() { return (size_t)2 * CREP * CFS + 16; }
size_t fsn_fb_chain_status_word() { return (size_t)2 * CREP * CFS; }

namespace {
template <int CH, bool SAVE, int CELL = 0>
void chain_launch(const ChainArgs& a, hipStream_t s) {
    const dim3 grid(2 * (CH / 4)), block(256);
    if (a.RT == 1) FSN_PERSIST_LAUNCH((fb_chain_kernel<CH, 4, 0, SAVE, CELL>), grid, block, s, a);
    else if (a.RT == 2) FSN_PERSIST_LAUNCH((fb_chain_kernel<CH, 2, 0, SAVE, CELL>), grid, block, s, a);
    else FSN_PERSIST_LAUNCH((fb_chain_kernel<CH, 1, 0, SAVE, CELL>), grid, block, s, a);
}
}  // namespace

// gx0: fragment-order projection of layer 0 (bias included); hseq1 [Tp][Npad][H] row-major out.
// Training form: hseq0, save0, save1 non-NULL (save = gates [Tp][Npad][4H] followed by the cell sequence [Tp][Npad][H],
// the layout of fsn_lstm_layer_forward).
int fsn_launch_fb_chain(const float* gx0, const float* whh0_p, const float* wih1_p, const float* whh1_p, const float* b1,
                        float* exchange, unsigned* flags, float* hseq1, int Tp, int Npad, int H, hipStream_t s,
                        float* hseq0, float* save0, float* save1, int cell) {
    if (!fsn_fb_chain_supported(H, Npad) || Tp < 1 || Tp > fsn_fb_chain_max_steps()) {
        fsn_set_error("fb_chain: built for H = 384 / 512, at most 64 rows and 4095 steps");
        return FSN_ERR_ARG;
    }
    const bool save = hseq0 || save0 || save1;
    if (cell != 0 && (cell != 1 || save)) {
        fsn_set_error("fb_chain: cell 0 (LSTM) or 1 (GRU as a four-gate cell, inference form only)");
        return FSN_ERR_ARG;
    }
    if (save && !(hseq0 && save0 && save1)) {
        fsn_set_error("fb_chain: the training form needs hseq0, save0 and save1");
        return FSN_ERR_ARG;
    }
    // flags and status: zero before EVERY launch (a kernel, not hipMemsetAsync: see fsn_launch_zero_words)
    if (fsn_launch_zero_words(flags, fsn_fb_chain_flag_words(), s) != FSN_OK) return FSN_ERR_LAUNCH;
    ChainArgs a{};
    a.gx0 = gx0;
    a.whh0_p = whh0_p;
    a.wih1_p = wih1_p;
    a.whh1_p = whh1_p;
    a.b1 = b1;
    a.hx0 = exchange;
    a.hx1 = exchange + (size_t)Tp * Npad * H;
    a.gx1 = exchange + (size_t)2 * Tp * Npad * H;
    a.hseq1 = hseq1;
    a.hseq0 = hseq0;
    a.gates0 = save0;
    a.cseq0 = save0 ? save0 + (size_t)Tp * Npad * 4 * H : nullptr;
    a.gates1 = save1;
    a.cseq1 = save1 ? save1 + (size_t)Tp * Npad * 4 * H : nullptr;
    a.flags = flags;
    a.status = flags + fsn_fb_chain_status_word();
    a.spin_ticks = fsn_spin_ticks();
    a.Tp = Tp;
    a.RT = Npad / 16;
    a.Npad = Npad;
    if (H == 512) {
        if (cell) chain_launch<512, false, 1>(a, s);
        else if (save) chain_launch<512, true>(a, s);
        else chain_launch<512, false>(a, s);
    } else {
        if (cell) chain_launch<384, false, 1>(a, s);
        else if (save) chain_launch<384, true>(a, s);
        else chain_launch<384, false>(a, s);
    }
    return fsn_check_launch("fb_chain_kernel");
}
