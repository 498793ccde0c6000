// Back-propagation through time of the sub-band model's two LSTM layers (training step, fullsubnet/trainer.py:56-63
// through sequence_model.py:52-58) as ONE persistent launch: the mirror image of lstm_group_kernels.hip.
//
// Per step t = T-1 .. 0 (formulas: lstm_train_kernels.hip):
//   layer 1: dh1_t = dH1_t (from the output layer) + dgates1_{t+1} W_hh1          -> cell derivative -> dgates1_t
//   layer 0: dh0_t = dgates1_t W_ih1 (the layer-to-layer dX, no GEMM) + dgates0_{t+1} W_hh0 -> ... -> dgates0_t
// As 2 x 193 launches of bptt_step_kernel plus a dX GEMM this cost 14.6 + 1.9 ms of a 50.5 ms training step (one step:
// a 2064 x 1536 x 384 product, 15 us of MFMA work, 37 us as a launch).  Here the same work split stays resident:
//   - workgroup (cluster c, member m, layer l) owns hidden units [48 m, 48 m + 48) of layer l for the 64 rows of
//     cluster c: wave w = row tile w, 3 accumulator tiles (dh of 3 unit groups), dc_t in registers; 16 workgroups per
//     cluster, two per CU, layer 1 (the leading chain) in the first half of the grid;
//   - the exchange buffers ARE the outputs: a member stores its 4 x 48 gate-gradient columns of step t into
//     dgates[t] (write-through), which every member reads as the A operand of step t - 1 (K = 1536 per product) and the
//     weight-gradient GEMMs read afterwards; nothing is reused, so there is no back-pressure;
//   - W^T fragments (the [k][out] order nn.LSTM stores) are fetched once per workgroup and shared by its four waves
//     through a two-stage LDS buffer, four K chunks x three column tiles per stage (one barrier per 48 MFMAs, the
//     cadence of the forward kernel); A fragments come through an eight-deep register ring;
//   - the saved activations of a step (gates, c_t, c_{t-1}, dH: 7 values per element) are requested before the flag
//     wait;
//   - flags / bounded spins / status exactly as in lstm_group_kernels.hip.
#include "fsn_common.h"

#ifndef FSN_BPTT_A_AUX
// 16: sc1 loads of the A operand (never cached: 0.3 GB per step through the fabric); 0: ordinary loads after an
// agent-scope acquire (buffer_inv sc1), shared through the XCD's L2 - measured slower (48.5 against 47.1 ms per training
// step: the invalidates of 64 workgroups per XCD and step take the weights out of the L2 as well)
#define FSN_BPTT_A_AUX 16
#ifndef FSN_BPTT_TURN
#define FSN_BPTT_TURN 2  // x 4 chunks = A fragments in flight (must divide 24 stages); measured: 2 -> 12.5 ms, 3 -> 12.9, 4 -> 13.2
#endif
#endif

namespace {

constexpr int BH = 384;           // hidden units (both layers)
constexpr int BG = 4 * BH;        // gate columns = K of every product
constexpr int BKC = BG / 16;      // K chunks (96)
constexpr int BM = 8;             // members per cluster and layer
constexpr int BU = BH / 16 / BM;  // 16-unit groups per member (3)
constexpr int BROWS = 64;         // rows per cluster
#ifndef FSN_BPTT_TURN16
#define FSN_BPTT_TURN16 2  // the same under the 16-bit arithmetic (the K loop is 8x shorter: the A operand's latency shows)
#endif
#ifndef FSN_BPTT_BCH
#define FSN_BPTT_BCH 4  // probe: 4 -> 12.8 ms, 6 -> 13.0
#endif
constexpr int BCH = FSN_BPTT_BCH;  // K chunks per LDS stage
constexpr int BFS = 32;           // words between flag groups (one cache line each)

struct BpttArgs {
    const float* dh1;     // [Tp][N][H]   d loss / d hseq1
    const float* wbase;   // packed W^T matrices ([H/16][KC][64][4]) live in one buffer: element offsets
    unsigned o_whh1T, o_wih1T, o_whh0T;
    const float *gates0, *cseq0, *gates1, *cseq1;  // saved by the forward pass: [Tp][N][4H], [Tp][N][H]
    float *dg0, *dg1;     // [Tp][N][4H]: gate gradients (outputs and exchange buffers)
    float* dx;            // [Tp][N][H]: dgates1_t W_ih1 = layer 0's dH, produced by layer 1 (see below)
    unsigned* flags;      // [clusters][2][BFS]: steps published by (layer 1 | layer 0, member)
    unsigned* status;
    unsigned long long spin_ticks;  // wait bound (fsn_spin_ticks)
    int Tp, Nrows;
};

__device__ __forceinline__ void bptt_store_sc1(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // write-through
}

__device__ __forceinline__ bool bptt_poll(unsigned* flags8, unsigned epoch, unsigned* status, unsigned long long ticks) {
    const int lane = threadIdx.x & 63;
    unsigned long long t0 = 0;
    for (unsigned spins = 0;; ++spins) {
        unsigned v = epoch;
        if (lane < BM) v = __hip_atomic_load(flags8 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all((int)(v >= epoch))) return true;
        if ((spins & 255u) == 255u && fsn_wait_give_up(status, spins, t0, ticks, 1u + epoch)) return false;
        __builtin_amdgcn_s_sleep(1);
    }
}

// ABL: experiment knob of tools/probe_bptt.hip (0 in the library; any bit set gives WRONG results): 1 no flag polling,
// 2 no gate-gradient / dx stores, 4 A fragments not loaded, 8 saved activations not loaded, 16 plain instead of
// write-through stores, 32 no tanhf in the cell derivative
// AR: arithmetic of the products (fsn_mma_k16); everything stored stays fp32
template <int LAYER, int ABL, int AR>
__device__ __forceinline__ void bptt_body(const BpttArgs& a, int cluster, int member,
                                          typename FsnWFrag<AR>::type (*bsh)[BCH * 2 * BU][64]) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const int Tp = a.Tp;
    const size_t N = (size_t)a.Nrows;
    unsigned* fl1 = a.flags + ((size_t)cluster * 2 + 0) * BFS;
    unsigned* fl0 = a.flags + ((size_t)cluster * 2 + 1) * BFS;
    // this lane's A fragment inside a cluster's [64][4H] tile of a gate-gradient buffer (byte offset)
    const unsigned a_off = (unsigned)(((wave * 16 + lr) * BG + 4 * lq) * 4);
    // the cluster's [64][4H] tile of step t of a gate-gradient buffer as a buffer resource (one per step: the whole
    // buffer - 2.5 GB at config 3's shape - is beyond the 2 GB a resource's offsets reach)
    auto tile = [&](float* dg, int t) {
        return __builtin_amdgcn_make_buffer_rsrc(dg + ((size_t)t * N + (size_t)cluster * BROWS) * BG, 0, BROWS * BG * 4,
                                                 0x00020000);
    };
    // likewise the cluster's [64][H] tile of a [Tp][N][H] buffer (cell states, dH, dx)
    auto tileh = [&](const float* p, int t) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p) + ((size_t)t * N + (size_t)cluster * BROWS) * BH, 0,
                                                 BROWS * BH * 4, 0x00020000);
    };
    // Element (row 4 lq + i of this wave's tile, unit 16 u + lr of this member) of such tiles: ONE lane offset each and
    // compile-time scalar offsets for (i, u, gate) - no per-element address registers (84 loads and 60 stores per step)
    const unsigned voff_g = (unsigned)(((wave * 16 + 4 * lq) * BG + member * BU * 16 + lr) * 4);
    const unsigned voff_h = (unsigned)(((wave * 16 + 4 * lq) * BH + member * BU * 16 + lr) * 4);
    auto ldf = [&](const __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
    };
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wbase), 0, 0x7fffffff, 0x00020000);

    // acc[u] += A(16 rows x K) W^T(K x 16 units of group u): n chunks (a multiple of 2 BCH) of the tile behind `xr`
    // (sc1 loads: the partners wrote through) against the packed matrix at element offset b of the weight buffer.
    // (The descriptor is passed by value: a reference to one of two descriptors chosen at run time puts both on the
    // stack.)
    // NT = 3: one matrix (element offset b); NT = 6: two matrices against the SAME A fragments (tiles 0..2 from b, 3..5
    // from b2) - layer 1 forms dgates1_{t+1} W_hh1 (its own dh) and dgates1_{t+1} W_ih1 (layer 0's dH) in one pass.
    // `mid` runs once, after the fourth stage: the step's saved activations are requested THERE - loads return in order,
    // so requested before the loop they hold up the first A fragment for an HBM latency every step (measured: 9 us of
    // a 69 us step), while behind sixteen chunks of buffered MFMA work the same wait is covered.
    auto kloop = [&](auto& acc, const __amdgpu_buffer_rsrc_t xr, unsigned b, unsigned b2, int n, auto&& mid) {
        constexpr int NT = (int)(sizeof(acc) / sizeof(f32x4));
        constexpr int TURN = AR == FSN_ARITH_F32 ? FSN_BPTT_TURN : FSN_BPTT_TURN16;  // stages per turn of the A ring
        constexpr int AD = TURN * BCH;       // A fragments in flight (write-through data of other CUs: first touch is far)
        constexpr int NB = BU;       // B fragments a wave holds in registers at a time (NT = 6: fetched in two halves)
        f32x4 ar[AD];
        typename FsnWFrag<AR>::type bn[NB];
        auto fetch_a = [&](int k) -> f32x4 {
            const int kc = k < n ? k : n - 1;
            if (ABL & 4) return f32x4{0.5f, 0.25f, -0.125f, 0.0625f};
            return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, a_off, (unsigned)kc * 64u, FSN_BPTT_A_AUX));
        };
        // Stage s holds chunks BCH s .. BCH s + BCH - 1, fragment (c, u) at index c NT + u.  The BCH NT fragments of a
        // stage are fetched in batches of NB (the registers a wave spends on them): batch id = fragments NB id ..;
        // wave w takes batches w, w + 4, ... - its q-th one at chunk q QSTEP of the stage before, parked in LDS when the
        // registers are needed again or at the end of that stage.
        constexpr int NBATCH = BCH * NT / NB, QMAX = (NBATCH + 3) / 4, QSTEP = BCH / QMAX;
        auto fetch_b = [&](int s, int q) {
            const int id = wave + 4 * q;
            if (id < NBATCH) {
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int f = id * NB + j, c = f / NT, u = f % NT;
                    int k = s * BCH + c;
                    k = k < n ? k : n - 1;
                    const unsigned ofs = (u < BU ? b : b2) + ((unsigned)(member * BU + u % BU) * BKC + (unsigned)k) * 256u;
                    bn[j] = fsn_load_wfrag<AR>(wrsrc, (unsigned)lane, ofs);
                }
            }
        };
        auto park_b = [&](int buf, int q) {
            const int id = wave + 4 * q;
            if (id < NBATCH) {
#pragma unroll
                for (int j = 0; j < NB; ++j) bsh[buf][id * NB + j][lane] = bn[j];
            }
        };
#pragma unroll
        for (int d = 0; d < AD; ++d) ar[d] = fetch_a(d);
#pragma unroll
        for (int q = 0; q < QMAX; ++q) {
            fetch_b(0, q);
            park_b(0, q);
        }
        __syncthreads();
        for (int s0 = 0; s0 < n / BCH; s0 += TURN) {  // TURN stages = one turn of the A ring (statically indexed)
            if (s0 == 2 * TURN) mid();
#pragma unroll
            for (int d = 0; d < AD; ++d) {
                const int ds = d / BCH, c = d % BCH, s = s0 + ds, buf = s & 1;
                if (c % QSTEP == 0 && c / QSTEP < QMAX) {  // the next stage's fragments, batch by batch
                    if (c > 0) park_b(buf ^ 1, c / QSTEP - 1);
                    __builtin_amdgcn_sched_barrier(0);  // requests first, pinned under this stage's MFMAs
                    fetch_b(s + 1, c / QSTEP);
                }
                const f32x4 av = ar[d];
                ar[d] = fetch_a(s * BCH + c + AD);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (AR == FSN_ARITH_F32) {
#pragma unroll
                    for (int u = 0; u < NT; ++u) {
                        const f32x4 bf = bsh[buf][c * NT + u][lane];
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[u] = mfma16(av[j], bf[j], acc[u]);
                    }
                } else {  // 16-bit operands: one matrix instruction per tile and K chunk
                    const typename FsnOperand<AR>::type ao = fsn_operand<AR>(av);
#pragma unroll
                    for (int u = 0; u < NT; ++u) acc[u] = fsn_mma_k16<AR>(ao, fsn_wfrag_operand<AR>(bsh[buf][c * NT + u][lane]), acc[u]);
                }
                if (c == BCH - 1) {
                    park_b(buf ^ 1, QMAX - 1);
                    __syncthreads();
                }
            }
        }
    };

    auto peek = [&](unsigned* flags8) -> unsigned {
        unsigned v = 0xffffffffu;
        if (wave == 0 && lane < BM) v = __hip_atomic_load(flags8 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return v;
    };
    auto wait_peeked = [&](unsigned v, unsigned* flags8, unsigned epoch) {
        if (wave == 0 && !(ABL & 1) && !__all((int)(v >= epoch))) (void)bptt_poll(flags8, epoch, a.status, a.spin_ticks);
        __syncthreads();
#if FSN_BPTT_A_AUX == 0
        // agent-scope acquire (buffer_inv sc1): the A operand is then read with ordinary loads, which the 16 workgroups
        // of a cluster - on one XCD when the grid allows it - share through that XCD's L2
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
    };
    auto publish = [&](unsigned* flag, unsigned epoch) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

    auto gstore = [&](const __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float v) {
        if (ABL & 2) return;
        if (ABL & 16) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
        else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 16);  // sc1: write-through
    };
    const float* gates = LAYER ? a.gates1 : a.gates0;
    const float* cseq = LAYER ? a.cseq1 : a.cseq0;
    float* dgout = LAYER ? a.dg1 : a.dg0;
    float dc[BU][4];
#pragma unroll
    for (int u = 0; u < BU; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) dc[u][i] = 0.f;

    // Work split: layer 0's dh is dgates1_t W_ih1 + dgates0_{t+1} W_hh0 - twice layer 1's K.  The first product has the
    // same A operand as layer 1's own (dgates1 of the step before), so LAYER 1 forms it, for its member's 48 units, in
    // the same pass over the A fragments (six accumulator tiles instead of three) and hands it over through `dx`;
    // layer 0 is left with one product, and nobody reads a dgates1 tile twice.  Layer 1 runs one extra iteration
    // (t = -1) that only produces dx_0.
    unsigned seen1 = LAYER ? 0xffffffffu : peek(fl1);
    for (int t = Tp - 1; t >= (LAYER ? -1 : 0); --t) {
        const unsigned done = (unsigned)(Tp - 1 - t);  // steps every member has published when step t + 1 is complete
        const int tt = t < 0 ? 0 : t;
        // saved activations of step t for this lane's 3 x 4 elements (requested inside the K loop, see kloop)
        float e_g[BU][4][4], e_ct[BU][4], e_cp[BU][4], e_dh[BU][4];
        auto load_saved = [&]() {
            const __amdgpu_buffer_rsrc_t rg = tile(const_cast<float*>(gates), tt), rc = tileh(cseq, tt),
                                         rp = tileh(cseq, t > 0 ? t - 1 : 0), rd = tileh(LAYER ? a.dh1 : cseq, tt);
#pragma unroll
            for (int u = 0; u < BU; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        e_g[u][i][g] = (ABL & 8) ? 0.4f : ldf(rg, voff_g, (unsigned)((i * BG + g * BH + u * 16) * 4));
                    const unsigned so = (unsigned)((i * BH + u * 16) * 4);
                    e_ct[u][i] = (ABL & 8) ? 0.3f : ldf(rc, voff_h, so);
                    e_cp[u][i] = (ABL & 8) ? 0.2f : (t > 0 ? ldf(rp, voff_h, so) : 0.f);
                    e_dh[u][i] = (ABL & 8) ? 0.1f : (LAYER ? ldf(rd, voff_h, so) : 0.f);
                }
        };
        f32x4 acc[BU];
#pragma unroll
        for (int u = 0; u < BU; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (LAYER) {
            if (t < Tp - 1) {
                f32x4 acc6[2 * BU];
#pragma unroll
                for (int u = 0; u < 2 * BU; ++u) acc6[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                wait_peeked(peek(fl1), fl1, done);  // dgates1_{t+1} of all members (just published: polls)
                kloop(acc6, tile(a.dg1, t + 1), a.o_whh1T, a.o_wih1T, BKC, load_saved);
                const __amdgpu_buffer_rsrc_t rx = tileh(a.dx, t + 1);
#pragma unroll
                for (int u = 0; u < BU; ++u) {
                    acc[u] = acc6[u];
#pragma unroll
                    for (int i = 0; i < 4; ++i)  // dx_{t+1}: layer 0's dH of step t + 1
                        gstore(rx, voff_h, (unsigned)((i * BH + u * 16) * 4), acc6[BU + u][i]);
                }
            } else {
                load_saved();
            }
        } else {
            // dx_t was produced by layer 1's iteration t - 1: all its members have published Tp - (t - 1)
            wait_peeked(seen1, fl1, done + 2);
            {
                const __amdgpu_buffer_rsrc_t rx = tileh(a.dx, t);
#pragma unroll
                for (int u = 0; u < BU; ++u)
#pragma unroll
                    for (int i = 0; i < 4; ++i)  // sc1: written through by layer 1's workgroup
                        acc[u][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                                  rx, voff_h, (unsigned)((i * BH + u * 16) * 4), 16));
            }
            if (t < Tp - 1) {
                wait_peeked(peek(fl0), fl0, done);  // dgates0_{t+1} of all members
                kloop(acc, tile(a.dg0, t + 1), a.o_whh0T, 0u, BKC, load_saved);
            } else {
                load_saved();
            }
            seen1 = peek(fl1);  // for the next step: layer 1 is ahead
        }
        // cell derivative of this wave's 16 rows x 48 units -> dgates_t (write-through: the partners' next A operand)
        if (t >= 0) {
            const __amdgpu_buffer_rsrc_t ro = tile(dgout, t);
#pragma unroll
            for (int u = 0; u < BU; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float ig = e_g[u][i][0], fg = e_g[u][i][1], gg = e_g[u][i][2], og = e_g[u][i][3];
                    const float dh = e_dh[u][i] + acc[u][i];
                    const float tc = (ABL & 32) ? e_ct[u][i] : tanhf(e_ct[u][i]);
                    const float d_o = dh * tc;
                    const float dct = dc[u][i] + dh * og * (1.f - tc * tc);
                    const unsigned so = (unsigned)((i * BG + u * 16) * 4);
                    gstore(ro, voff_g, so, dct * gg * ig * (1.f - ig));
                    gstore(ro, voff_g, so + BH * 4, dct * e_cp[u][i] * fg * (1.f - fg));
                    gstore(ro, voff_g, so + 2 * BH * 4, dct * ig * (1.f - gg * gg));
                    gstore(ro, voff_g, so + 3 * BH * 4, d_o * og * (1.f - og));
                    dc[u][i] = dct * fg;
                }
        }
        publish((LAYER ? fl1 : fl0) + member, done + 1);
    }
}

template <int ABL, int AR = FSN_ARITH_F32>
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_num_vgpr(116))) void lstm2_group_bptt_kernel(const BpttArgs a) {
    __shared__ typename FsnWFrag<AR>::type bsh[2][BCH * 2 * BU][64];  // two stages x (4 chunks x up to 6 column tiles) x 1 KB (512 B in 16 bits)
    // first half of the grid: layer 1 (the leading chain), second half: layer 0; cluster members on one XCD when the
    // cluster count allows it (speed only), as in lstm2_group_kernel
    const int half = gridDim.x >> 1;
    const int second = (int)blockIdx.x >= half ? 1 : 0;
    const int bid = (int)blockIdx.x - second * half;
    const int nclusters = half / BM;
    int cluster, member;
    if (nclusters % 8 == 0) {
        const int xcd = bid & 7, j = bid >> 3;
        cluster = xcd * (nclusters / 8) + j / BM;
        member = j % BM;
    } else {
        cluster = bid / BM;
        member = bid % BM;
    }
    if (!second) {
        __builtin_amdgcn_s_setprio(2);  // layer 1 is the longer chain (six tiles per A fragment against three)
        bptt_body<1, ABL, AR>(a, cluster, member, bsh);
    } else {
        bptt_body<0, ABL, AR>(a, cluster, member, bsh);
    }
}

}  // namespace

// one cluster per workgroup set: at most CUs / 8 clusters
int fsn_lstm2_group_bptt_clusters(int tiles) {
    int cus = 0, dev = 0;
    if (!fsn_persistent_allowed() || hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return 0;
    // residency contract: two workgroups per CU, by the compiled kernel's occupancy
    for (const void* k : {(const void*)lstm2_group_bptt_kernel<0>, (const void*)lstm2_group_bptt_kernel<0, FSN_ARITH_F16>,
                          (const void*)lstm2_group_bptt_kernel<0, FSN_ARITH_BF16>})
        if (!fsn_grid_fits(k, 256, 2u * (unsigned)cus)) return 0;
    const int cap = cus / BM, c = tiles / 4;
    return c < cap ? c : cap;
}
size_t fsn_lstm2_group_bptt_flag_words(int clusters) { return (size_t)clusters * 2 * BFS + 16; }
size_t fsn_lstm2_group_bptt_status_word(int clusters) { return (size_t)clusters * 2 * BFS; }

// Rows [0, 64 clusters) of the two layers: dh1 [Tp][Nrows][H]; whh1T_p / wih1T_p / whh0T_p = W_hh1 / W_ih1 / W_hh0
// packed TRANSPOSED ([H/16][4H/16][64][4], fsn_launch_pack(..., transposed = 1)) in one buffer; save0 / save1 in
// fsn_lstm_layer_forward's layout; dg0 / dg1 [Tp][Nrows][4H] out; dx [Tp][Nrows][H] scratch (layer 0's dH).
int fsn_launch_lstm2_group_bptt(const float* dh1, const float* whh1T_p, const float* wih1T_p, const float* whh0T_p,
                                const float* save0, const float* save1, float* dg0, float* dg1, float* dx, unsigned* flags,
                                int Tp, int Nrows, int clusters, int H, hipStream_t s, int arith, const void* w16) {
    if (H != BH || clusters < 1 || clusters > fsn_lstm2_group_bptt_clusters(Nrows / 16)) {
        fsn_set_error("lstm2_group_bptt: H = 384, clusters * 64 <= rows, one cluster per eight CUs at most");
        return FSN_ERR_ARG;
    }
    if (fsn_launch_zero_words(flags, fsn_lstm2_group_bptt_flag_words(clusters), s) != FSN_OK) return FSN_ERR_LAUNCH;
    const float* lo = whh1T_p;
    for (const float* q : {wih1T_p, whh0T_p}) lo = q < lo ? q : lo;
    for (const float* q : {whh1T_p, wih1T_p, whh0T_p})
        if (q - lo > 0x1fffffffL) {
            fsn_set_error("lstm2_group_bptt: the packed weight matrices must share one buffer");
            return FSN_ERR_ARG;
        }
    if (arith != FSN_ARITH_F32 && !w16) {
        fsn_set_error("lstm2_group_bptt: 16-bit arithmetic needs the 16-bit copy of the packed weights");
        return FSN_ERR_ARG;
    }
    BpttArgs a{};
    a.dh1 = dh1;
    a.wbase = arith == FSN_ARITH_F32 ? lo : static_cast<const float*>(w16);  // the 16-bit mirror of the buffer from `lo`
    a.o_whh1T = (unsigned)(whh1T_p - lo);
    a.o_wih1T = (unsigned)(wih1T_p - lo);
    a.o_whh0T = (unsigned)(whh0T_p - lo);
    a.gates0 = save0;
    a.cseq0 = save0 + (size_t)Tp * Nrows * BG;
    a.gates1 = save1;
    a.cseq1 = save1 + (size_t)Tp * Nrows * BG;
    a.dg0 = dg0;
    a.dg1 = dg1;
    a.dx = dx;
    a.flags = flags;
    a.status = flags + (size_t)clusters * 2 * BFS;
    a.spin_ticks = fsn_spin_ticks();
    a.Tp = Tp;
    a.Nrows = Nrows;
    const dim3 grid((unsigned)clusters * BM * 2), block(256);
    if (arith == FSN_ARITH_F16) FSN_PERSIST_LAUNCH((lstm2_group_bptt_kernel<0, FSN_ARITH_F16>), grid, block, s, a);
    else if (arith == FSN_ARITH_BF16) FSN_PERSIST_LAUNCH((lstm2_group_bptt_kernel<0, FSN_ARITH_BF16>), grid, block, s, a);
    else if (arith == FSN_ARITH_F32) FSN_PERSIST_LAUNCH(lstm2_group_bptt_kernel<0>, grid, block, s, a);
    else {
        fsn_set_error("lstm2_group_bptt: arithmetic %d unknown", arith);
        return FSN_ERR_ARG;
    }
    return fsn_check_launch("lstm2_group_bptt_kernel");
}
