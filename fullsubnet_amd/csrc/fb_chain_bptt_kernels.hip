// Back-propagation through time of the full-band model's two LSTM layers (training step; H = 512; 16 rows = one row tile,
// up to QMAXT row tiles per launch - Fast FullSubNet's decoder pair at its TOML's batch of 72, a full-band model at more
// than 16 utterances per rank) as ONE persistent launch: the backward counterpart of fb_chain_kernels.hip, with the work split of
// lstm_group_bptt_kernels.hip.
//
//   layer 1: dh1_t = dH1_t + dgates1_{t+1} W_hh1 -> cell derivative -> dgates1_t
//   layer 0: dh0_t = dgates1_t W_ih1 + dgates0_{t+1} W_hh0 -> ... -> dgates0_t
// As 2 x 193 launches of bptt_step_kernel<1, 1, 16> (7.8 us each) this was 3.0 ms of a 45 ms training step.  Here:
//   - three stages of H / 16 = 32 workgroups: L1, X and L0.  A workgroup owns 16 hidden units = one 16-column MFMA tile
//     of dh for the 16 rows; K = 2048 gate columns is split over its four waves (fixed-order sum through LDS); its slice
//     of the ONE W^T it needs (2048 x 16 floats) is read once into registers;
//   - every stage forms one product per step: L1 dgates1_{t+1} W_hh1, X dgates1_t W_ih1 (layer 0's dH of step t, handed
//     to L0 workgroup j as four partial tiles that L0's reduction sums), L0 dgates0_{t+1} W_hh0.  Round 2 had L1 form
//     X's product too, after publishing, from the same A fragments: 1.7 us of MFMAs per step that sat on L1's own chain
//     (the next step's flags arrived while it was still multiplying) - 9.3 us per step; as a stage of its own X follows
//     L1 by one step and L0 follows X, and a step is one product long;
//   - the gate-gradient buffers dgates[t] ([16][2048] per step; the weight-gradient GEMMs read them afterwards) are the
//     exchange buffers: write-through stores, drain, flag copies; one wave polls; sc1 loads; nothing is reused;
//   - the saved activations of a step are requested AFTER the A fragments (loads return in order: requested first they
//     would hold the MFMAs up for an HBM latency);
//   - several row tiles (round 6): the rows are independent sequences sharing the weights.  A workgroup forms its product
//     for every tile of the step with the ONE W^T slice in its registers (the next tile's A fragments travel under this
//     tile's MFMAs), the partial sums of all tiles go through LDS in one pass, wave w finishes tiles w and w + 4 (cell
//     derivative, write-through stores), and the step is handed on ONCE: one drain, one flag, one poll per step whatever
//     the tile count (walking the tiles as separate chains inside a step - one hand-off each - was measured first: 11.4 ms
//     for a 72-row two-layer stack's forward + backward against 9.4 ms step by step).
#include "fsn_common.h"

namespace {

constexpr int QH = 512;           // hidden units per layer
constexpr int QG = 4 * QH;        // gate columns = K
constexpr int QKC = QG / 16;      // K chunks (128)
constexpr int QCW = QKC / 4;      // K chunks per wave (32)
constexpr int QNW = QH / 16;      // workgroups per stage (32)
constexpr int QREP = 4;           // copies of a stage's flag array
constexpr int QMAXT = 5;          // row tiles per launch
constexpr int QOWN = (QMAXT + 3) / 4;  // tiles a wave finishes (tile k belongs to wave k % 4)

struct ChainBpttArgs {
    const float* dh1;     // [Tp][N][H]  d loss / d hseq1
    const float *whh1T_p, *wih1T_p, *whh0T_p;  // W^T packed [H/16][4H/16][64][4]
    const float *gates0, *cseq0, *gates1, *cseq1;  // saved by the forward pass: [Tp][N][4H], [Tp][N][H]
    float *dg0, *dg1;     // [Tp][N][4H]: gate gradients (outputs and exchange buffers)
    float* dx;            // [tiles][Tp][QNW][4 waves][64][4]: partial tiles of dgates1_t W_ih1 (layer 0's dH)
    unsigned* flags;      // [3][QREP][QNW]: steps published by (L1 | L0 | X, workgroup)
    unsigned* status;
    unsigned long long spin_ticks;  // wait bound (fsn_spin_ticks)
    int Tp;
    int N;                // rows of the buffers: 16 per tile
};

// wave 0: all 32 flags of a stage copy >= epoch and (optionally) one more flag >= its epoch; bounded
__device__ __forceinline__ bool qwait(const unsigned* flags, unsigned epoch, const unsigned* one, unsigned one_epoch,
                                      unsigned* status, unsigned long long ticks) {
    const int lane = threadIdx.x & 63;
    unsigned long long t0 = 0;
    for (unsigned spins = 0;; ++spins) {
        unsigned v = ~0u, w = ~0u;
        if (epoch > 0 && lane < QNW) v = __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (one && lane == 0) w = __hip_atomic_load(one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all((int)(v >= epoch && w >= one_epoch))) return true;
        if ((spins & 255u) == 255u && fsn_wait_give_up(status, spins, t0, ticks, 1u + epoch)) return false;
    }
}

__global__ __launch_bounds__(256, 1) void fb_chain_bptt_kernel(const ChainBpttArgs a) {
    extern __shared__ f32x4 red[];           // partial sums of (wave, tile): [4][nt][64] (4 KB per row tile: the launch beside the
                                             // weight-gradient products' 144 KB workgroups needs the CU's last 16 KB)
    const int role = (int)blockIdx.x / QNW, j = (int)blockIdx.x % QNW;  // thirds of the grid: L1, X, L0
    const int l1 = role == 0 ? 1 : 0;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const int Tp = a.Tp;
    const int nt = a.N >> 4;                 // row tiles of this launch (1 .. QMAXT)
    const size_t nrows = (size_t)a.N;
    unsigned* fl1 = a.flags;                 // [QREP][QNW]
    unsigned* fl0 = a.flags + QREP * QNW;
    unsigned* flx = a.flags + 2 * QREP * QNW;
    const int rep = j % QREP;
    const unsigned lane16 = (unsigned)lane * 16u;

    // this wave's K quarter of column tile j of a packed W^T: chunk kc's fragment is 1 KB at ((j KC + kc) 64 + lane) 4
    auto load_w = [&](const float* packed, f32x4 (&w)[QCW]) {
        const float* wp = packed + ((size_t)j * QKC * 64 + lane) * 4 + (size_t)(wave * QCW) * 256;
#pragma unroll
        for (int q = 0; q < QCW; ++q) w[q] = *reinterpret_cast<const f32x4*>(wp + (size_t)q * 256);
    };
    // rows [16 tile, 16 tile + 16) of step t of a [Tp][N][...] buffer as a buffer resource
    auto slab = [&](const float* p, int t, int width, int tile) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p) + ((size_t)t * nrows + (size_t)16 * tile) * width, 0,
                                                 16 * width * 4, 0x00020000);
    };
    // this wave's A fragments of a tile of dgates[t] (row lr, k = 16 kc + 4 lq ..) by HALVES of its K quarter (QCW / 2 chunks =
    // 16 KB per wave in flight): sc1 loads; half s + 1 travels under half s' MFMAs (two register sets of 64)
    constexpr int QHC = QCW / 2;
    const unsigned a_off = (unsigned)((lr * QG + 4 * lq) * 4);
    auto load_half = [&](f32x4 (&ar)[QHC], const float* dg, int t, int stage) {  // stage = 2 tile + half
        const __amdgpu_buffer_rsrc_t r = slab(dg, t, QG, stage >> 1);
        const int q0 = wave * QCW + (stage & 1) * QHC;
#pragma unroll
        for (int q = 0; q < QHC; ++q)
            ar[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, a_off, (unsigned)((q0 + q) * 64), 16));
        __builtin_amdgcn_sched_barrier(0);
    };
    auto publish = [&](unsigned* flags, unsigned epoch) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if ((int)threadIdx.x < QREP)
            __hip_atomic_store(flags + (size_t)threadIdx.x * QNW + j, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // element (row 4 lq + i, unit 16 j + lr) of a step's slabs: one lane offset, compile-time scalar offsets
    const unsigned voff_g = (unsigned)(((4 * lq) * QG + j * 16 + lr) * 4);
    const unsigned voff_h = (unsigned)(((4 * lq) * QH + j * 16 + lr) * 4);
    auto ldf = [&](const __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
    };
    const float* gates = l1 ? a.gates1 : a.gates0;
    const float* cseq = l1 ? a.cseq1 : a.cseq0;
    float* dgout = l1 ? a.dg1 : a.dg0;
    const unsigned dx_wave = (unsigned)(((size_t)j * 4 + wave) * 1024);  // this wave's partial tile inside a step of dx
    const unsigned dx_step = (unsigned)QNW * 4096u;
    auto rdx = [&](int tile) {  // dx of row tile `tile`: [Tp][QNW][4 waves][64][4]
        return __builtin_amdgcn_make_buffer_rsrc(a.dx + (size_t)tile * Tp * QNW * 1024, 0, 0x7fffffff, 0x00020000);
    };
    float dc[QOWN][4];  // cell-state gradient of the tiles this wave finishes
#pragma unroll
    for (int o = 0; o < QOWN; ++o)
#pragma unroll
        for (int i = 0; i < 4; ++i) dc[o][i] = 0.f;

    f32x4 whh[QCW], ah0[QHC], ah1[QHC];
    load_w(role == 0 ? a.whh1T_p : role == 1 ? a.wih1T_p : a.whh0T_p, whh);

    // acc[k] += (tile k of dg[t]) x (this wave's K quarter of W^T) for every tile, chunk by chunk in K order; the first half
    // tile must have been requested (load_half(ah0, dg, t, 0)) by the caller - which places its other requests behind it
    auto products = [&](f32x4 (&acc)[QMAXT], const float* dg, int t) {
#pragma unroll
        for (int st = 0; st < 2 * QMAXT; ++st) {
            if ((st >> 1) >= nt) break;
            if (st + 1 < 2 * nt) load_half((st & 1) ? ah0 : ah1, dg, t, st + 1);
            const int h = st & 1;
#pragma unroll
            for (int q = 0; q < QHC; ++q)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    acc[st >> 1] = mfma16(((st & 1) ? ah1 : ah0)[q][jj], whh[h * QHC + q][jj], acc[st >> 1]);
        }
    };
    // cell derivative of tile `tile`, rows 4 lq + i, unit 16 j + lr, from dh -> dgates_t (write-through)
    auto cell = [&](int t, int tile, float (&dcs)[4], f32x4 dh, const float (&e_g)[4][4], const float (&e_ct)[4], const float (&e_cp)[4]) {
        const __amdgpu_buffer_rsrc_t ro = slab(dgout, t, QG, tile);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float ig = e_g[i][0], fg = e_g[i][1], gg = e_g[i][2], og = e_g[i][3];
            const float tc = tanhf(e_ct[i]);
            const float d_o = dh[i] * tc;
            const float dct = dcs[i] + dh[i] * og * (1.f - tc * tc);
            const unsigned so = (unsigned)(i * QG * 4);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dct * gg * ig * (1.f - ig)), ro, voff_g, so, 16);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dct * e_cp[i] * fg * (1.f - fg)), ro, voff_g, so + QH * 4, 16);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dct * ig * (1.f - gg * gg)), ro, voff_g, so + 2 * QH * 4, 16);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, d_o * og * (1.f - og)), ro, voff_g, so + 3 * QH * 4, 16);
            dcs[i] = dct * fg;
        }
    };
    // saved activations of step t, tile `tile`, for the finishing wave's 4 elements (requested after the A fragments)
    auto load_saved = [&](int t, int tile, float (&e_g)[4][4], float (&e_ct)[4], float (&e_cp)[4], float (&e_dh)[4]) {
        const __amdgpu_buffer_rsrc_t rg = slab(gates, t, QG, tile), rc = slab(cseq, t, QH, tile),
                                     rp = slab(cseq, t > 0 ? t - 1 : 0, QH, tile), rd = slab(l1 ? a.dh1 : cseq, t, QH, tile);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int g = 0; g < 4; ++g) e_g[i][g] = ldf(rg, voff_g, (unsigned)((i * QG + g * QH) * 4));
            e_ct[i] = ldf(rc, voff_h, (unsigned)(i * QH * 4));
            e_cp[i] = t > 0 ? ldf(rp, voff_h, (unsigned)(i * QH * 4)) : 0.f;
            e_dh[i] = l1 ? ldf(rd, voff_h, (unsigned)(i * QH * 4)) : 0.f;
        }
    };
    // The end of a step of L1 / L0: the four waves' partial sums of every tile through LDS (summed in the fixed order wave 0,
    // 1, 2, 3), wave w finishes tiles w and w + 4 (the first one's saved activations were requested before the products, the
    // second one's - wave 0 with five tiles - only now: one register set), then ONE hand-off.
    auto finish = [&](int t, f32x4 (&acc)[QMAXT], float (&e_g)[4][4], float (&e_ct)[4], float (&e_cp)[4], float (&e_dh)[4],
                      unsigned* flags, unsigned epoch) {
#pragma unroll
        for (int k = 0; k < QMAXT; ++k)
            if (k < nt) red[(wave * nt + k) * 64 + lane] = acc[k];
        __syncthreads();
#pragma unroll
        for (int o = 0; o < QOWN; ++o) {
            const int tile = wave + 4 * o;
            if (tile < nt) {
                if (o > 0) load_saved(t, tile, e_g, e_ct, e_cp, e_dh);
                f32x4 v = red[tile * 64 + lane];
#pragma unroll
                for (int p = 1; p < 4; ++p) {
                    const f32x4 r = red[(p * nt + tile) * 64 + lane];
                    v = f32x4{v[0] + r[0], v[1] + r[1], v[2] + r[2], v[3] + r[3]};
                }
                cell(t, tile, dc[o], f32x4{v[0] + e_dh[0], v[1] + e_dh[1], v[2] + e_dh[2], v[3] + e_dh[3]}, e_g, e_ct, e_cp);
            }
        }
        publish(flags, epoch);
    };

    if (role == 0) {
        // ---- L1: dgates1_t from dH1_t + dgates1_{t+1} W_hh1; epoch Tp - t is published once dgates1_t is stored, drained
        for (int t = Tp - 1; t >= 0; --t) {
            const unsigned done = (unsigned)(Tp - 1 - t);
            f32x4 acc[QMAXT];
#pragma unroll
            for (int k = 0; k < QMAXT; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            float e_g[4][4], e_ct[4], e_cp[4], e_dh[4];
            if (t < Tp - 1) {
                if (wave == 0) (void)qwait(fl1 + rep * QNW, done, nullptr, 0, a.status, a.spin_ticks);
                __syncthreads();
                load_half(ah0, a.dg1, t + 1, 0);
            }
            if (wave < nt) load_saved(t, wave, e_g, e_ct, e_cp, e_dh);
            if (t < Tp - 1) products(acc, a.dg1, t + 1);
            finish(t, acc, e_g, e_ct, e_cp, e_dh, fl1, done + 1);
        }
        return;
    }
    if (role == 1) {
        // ---- X: the partial tiles of dx_t = dgates1_t W_ih1 (layer 0's dH of step t), one per wave and row tile, for L0
        // workgroup j; dgates1_t is complete at L1's epoch Tp - t; this workgroup publishes the same epoch once its tiles
        // are drained
        for (int t = Tp - 1; t >= 0; --t) {
            const unsigned epoch = (unsigned)(Tp - t);
            if (wave == 0) (void)qwait(fl1 + rep * QNW, epoch, nullptr, 0, a.status, a.spin_ticks);
            __syncthreads();
            f32x4 accx[QMAXT];
#pragma unroll
            for (int k = 0; k < QMAXT; ++k) accx[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            load_half(ah0, a.dg1, t, 0);
            products(accx, a.dg1, t);
            const unsigned so = (unsigned)t * dx_step + dx_wave;
#pragma unroll
            for (int k = 0; k < QMAXT; ++k) {
                if (k >= nt) break;
                const __amdgpu_buffer_rsrc_t rx = rdx(k);
                asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen sc1" ::"v"(accx[k]), "v"(lane16), "s"(rx), "s"(so) : "memory");
            }
            publish(flx, epoch);
        }
        return;
    }
    // ---- L0: dh0_t = dx_t (four partial tiles per row tile from X workgroup j, one per wave) + dgates0_{t+1} W_hh0 ----------
    for (int t = Tp - 1; t >= 0; --t) {
        const unsigned done = (unsigned)(Tp - 1 - t);
        // dx_t: stored by X workgroup j, complete at its epoch Tp - t = done + 1
        if (wave == 0) (void)qwait(fl0 + rep * QNW, done, flx + rep * QNW + j, done + 1, a.status, a.spin_ticks);
        __syncthreads();
        f32x4 acc[QMAXT];
#pragma unroll
        for (int k = 0; k < QMAXT; ++k)
            acc[k] = k < nt ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rdx(k), lane16, (unsigned)t * dx_step + dx_wave, 16))
                            : f32x4{0.f, 0.f, 0.f, 0.f};
        float e_g[4][4], e_ct[4], e_cp[4], e_dh[4];
        if (t < Tp - 1) load_half(ah0, a.dg0, t + 1, 0);
        if (wave < nt) load_saved(t, wave, e_g, e_ct, e_cp, e_dh);
        if (t < Tp - 1) products(acc, a.dg0, t + 1);
        finish(t, acc, e_g, e_ct, e_cp, e_dh, fl0, done + 1);
    }
}

}  // namespace

bool fsn_fb_chain_bptt_supported(int H, int N) {
    if (H != QH || N < 16 || N % 16 || N > 16 * QMAXT || !fsn_persistent_allowed()) return false;
    return fsn_grid_fits((const void*)fb_chain_bptt_kernel, 256, 3 * QNW);  // residency contract
}
// dx is addressed as t * 128 KB through 32-bit byte offsets
int fsn_fb_chain_bptt_max_steps() { return (int)(0x7fffffffu / ((unsigned)QNW * 4096u)) - 1; }
size_t fsn_fb_chain_bptt_dx_floats(int Tp, int N) { return (size_t)(N / 16) * Tp * QNW * 1024; }
size_t fsn_fb_chain_bptt_flag_words() { return (size_t)3 * QREP * QNW + 16; }
size_t fsn_fb_chain_bptt_status_word() { return (size_t)3 * QREP * QNW; }

// dh1 [Tp][N][H]; W^T matrices packed by fsn_launch_pack(..., transposed = 1); save0 / save1 in
// fsn_lstm_layer_forward's layout; dg0 / dg1 [Tp][N][4H] out; dx: fsn_fb_chain_bptt_dx_floats(Tp, N) scratch.
int fsn_launch_fb_chain_bptt(const float* dh1, const float* whh1T_p, const float* wih1T_p, const float* whh0T_p,
                             const float* save0, const float* save1, float* dg0, float* dg1, float* dx, unsigned* flags,
                             int Tp, int N, int H, hipStream_t s) {
    if (!fsn_fb_chain_bptt_supported(H, N) || Tp < 1 || Tp > fsn_fb_chain_bptt_max_steps()) {
        fsn_set_error("fb_chain_bptt: built for H = 512, 16 .. %d rows in tiles of 16 and at most %d steps", 16 * QMAXT,
                      fsn_fb_chain_bptt_max_steps());
        return FSN_ERR_ARG;
    }
    if (fsn_launch_zero_words(flags, fsn_fb_chain_bptt_flag_words(), s) != FSN_OK) return FSN_ERR_LAUNCH;
    ChainBpttArgs a{};
    a.dh1 = dh1;
    a.whh1T_p = whh1T_p;
    a.wih1T_p = wih1T_p;
    a.whh0T_p = whh0T_p;
    a.gates0 = save0;
    a.cseq0 = save0 + (size_t)Tp * N * QG;
    a.gates1 = save1;
    a.cseq1 = save1 + (size_t)Tp * N * QG;
    a.dg0 = dg0;
    a.dg1 = dg1;
    a.dx = dx;
    a.flags = flags;
    a.status = flags + fsn_fb_chain_bptt_status_word();
    a.spin_ticks = fsn_spin_ticks();
    a.Tp = Tp;
    a.N = N;
    fsn_persist_admit((const void*)fb_chain_bptt_kernel, 256, 3u * QNW);
    hipLaunchKernelGGL(fb_chain_bptt_kernel, dim3(3 * QNW), dim3(256), (size_t)(N / 16) * 4 * 64 * sizeof(f32x4), s, a);
    return fsn_check_launch("fb_chain_bptt_kernel");
}
