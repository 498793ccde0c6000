// Recurrent half of nn.LSTM (audio_zen/model/module/sequence_model.py:52-58,116-117) for gfx950.
//
// The input half  W_ih x_t + b_ih + b_hh  of every time step is produced beforehand by
// gemm_kernels.hip in accumulator-fragment order ("gx").  What is left is, per step,
//     gates = gx[t] + h_{t-1} W_hh^T ;  c = sig(f) c + sig(i) tanh(g) ;  h = sig(o) tanh(c)
// with PyTorch's gate order (i, f, g, o) along the 4H axis and h_0 = c_0 = 0.
//
// Two kernels, one per regime:
//   lstm_rec_kernel  - MANY independent sequences (the sub-band model: N = B*F = 16 448 rows).
//     A workgroup owns 16*RT rows for the whole utterance: h lives in LDS, c in registers, W_hh
//     streams from L2 as pre-tiled B fragments (2.4 MB, re-read once per step by each workgroup,
//     ~15 GB/s per CU at RT = 5).  Each wave owns 32 hidden units x all four gates, so the cell
//     update is lane-local in MFMA accumulator layout; the four gates are accumulated one after
//     the other to keep accumulators + c + one temporary inside the 168-VGPR budget of 3 waves
//     per SIMD.  MFMA-bound: 16*RT x 384 x 1536 MAC per step per workgroup.
//   lstm_step_kernel - FEW sequences (the full-band model: N = B rows), one launch per time step,
//     work split over hidden units x row tiles x 4-way split-K so that 128+ workgroups share a
//     64-row problem; h and c live in global memory (L2 resident).
#include "fsn_common.h"
#include <stdlib.h>

namespace {

// ---------------------------------------------------------------------------------------------
// One frame of the sub-band model input for the rows of this workgroup, written to LDS as
// xl[row][0 .. 16 kin_chunks) (zero padded): freq_unfold + cat + norm of fullsubnet/model.py:98-111.
template <int NTHREADS>
__device__ __forceinline__ void stage_sb_input(const FsnSbInput& x, float* xl, int xs, long n0, int rows, int t) {
    const int kin = 16 * x.kin_chunks;
    for (int i = threadIdx.x; i < rows * kin; i += NTHREADS) {
        const int row = i / kin, c = i % kin;
        const float v = fsn_sb_input_value(x, n0 + row, c, t);
        xl[row * xs + c] = v;
    }
}

template <int H, int RT, int UG, bool XIN>
__global__ __launch_bounds__((H / (16 * UG)) * 64) void lstm_rec_kernel(const float* __restrict__ gx,
                                                                        const FsnSbInput xin,
                                                                        const float* __restrict__ whh_p,
                                                                        float* __restrict__ hseq, int Tp, int Npad,
                                                                        const FsnRecFc fc) {
    constexpr int NW = H / (16 * UG);   // waves per workgroup
    constexpr int KC = H / 16;          // k chunks == unit groups
    constexpr int CT = 4 * KC;          // column tiles of the gate matrix
    constexpr int HS = H + 4;           // LDS row stride (floats): 16 B aligned, breaks the 64-bank period
    constexpr int ROWS = RT * 16;
    constexpr int UNR = RT >= 3 ? 1 : (RT == 2 ? 2 : 4);  // K-loop unroll: bound the in-flight B fragments
    extern __shared__ __attribute__((aligned(16))) float hl[];  // [ROWS][HS] (+ 2 x [ROWS][XS] when XIN)
    // XIN: the first sub-band layer builds its input projection itself (K = 2nb+2 = 32: two more
    // chunks per gate) from a double-buffered LDS tile of the unfolded, normalised input, instead of
    // reading a 19.2 GB precomputed gx that an HBM-write-bound GEMM would have to produce first.
    const int XS = XIN ? 16 * xin.kin_chunks + 4 : 0;
    float* xl = hl + ROWS * HS;
    float* wl = xl;  // !XIN with the output layer fused: its two weight rows [2][H] sit here instead

    // wave-uniform, and told so: everything derived from it (unit group, weight / projection tile bases) then lives in
    // scalar registers and the loads take the scalar-base + 32-bit lane offset form
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const long n0 = (long)blockIdx.x * ROWS;
    const bool fuse_fc = !XIN && fc.w_p != nullptr;
    if (fuse_fc) {  // un-tile rows 0 / 1 of the packed output weights: element (c, k) of fragment order
        for (int i = threadIdx.x; i < 2 * H; i += NW * 64) {
            const int c = i / H, k = i % H;
            wl[i] = fc.w_p[(((k >> 4) * 64) + ((k & 15) >> 2) * 16 + c) * 4 + (k & 3)];
        }
    }

    float cst[RT][UG][4], tmp[RT][UG][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int u = 0; u < UG; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) cst[rt][u][i] = 0.f;
    for (int i = threadIdx.x; i < ROWS * HS; i += NW * 64) hl[i] = 0.f;
    if (XIN) stage_sb_input<NW * 64>(xin, xl, XS, n0, ROWS, 0);
    __syncthreads();

    for (int t = 0; t < Tp; ++t) {
        const long gx_rt0 = ((long)t * Npad + n0) >> 4;
        // frame t+1 goes into the other x buffer; it was last read in step t-1, which ended with
        // two barriers, and is first read after the two barriers that end this step
        if (XIN && t + 1 < Tp) stage_sb_input<NW * 64>(xin, xl + ((t + 1) & 1) * ROWS * XS, XS, n0, ROWS, t + 1);
        const float* xt = xl + (t & 1) * ROWS * XS;
        // gate order of evaluation: f (1), i (0), g (2), o (3)
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            // The four passes are unrolled, so that the cell update of each is straight-line code that updates c and
            // the temporary in place.  Two things keep the register count of the rolled loop: nothing is scheduled
            // across a pass boundary, and the gate index is opaque to the optimiser - as a constant, the per-gate
            // operand addresses of all four passes are hoisted out of the time loop and held live (85+ spills).
            __builtin_amdgcn_sched_barrier(0);
            int g = pass == 0 ? 1 : (pass == 1 ? 0 : pass);
            asm volatile("" : "+s"(g));
            f32x4 acc[RT][UG];
            unsigned bo[UG];  // 32-bit element offsets from the (uniform) weight base: one register per stream
#pragma unroll
            for (int u = 0; u < UG; ++u) {
                const int ug = wave * UG + u;
                bo[u] = (unsigned)(((g * KC + ug) * KC * 64 + lane) * 4);
                if (!XIN) {
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
                        acc[rt][u] = *reinterpret_cast<const f32x4*>(
                            gx + (((gx_rt0 + rt) * CT + g * KC + ug) * 64 + lane) * 4);
                } else {
                    const float bias = xin.bias[(g * KC + ug) * 16 + lr];
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[rt][u] = f32x4{bias, bias, bias, bias};
                }
            }
            if (XIN) {  // W_ih x_t: the same fragment scheme with A from the staged input tile
                for (int kx = 0; kx < xin.kin_chunks; ++kx) {
                    f32x4 bx[UG];
#pragma unroll
                    for (int u = 0; u < UG; ++u)
                        bx[u] = *reinterpret_cast<const f32x4*>(
                            xin.wih_p + (((long)(g * KC + wave * UG + u) * xin.kin_chunks + kx) * 64 + lane) * 4);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(xt + (rt * 16 + lr) * XS + kx * 16 + 4 * lq);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int u = 0; u < UG; ++u) acc[rt][u] = mfma16(a[j], bx[u][j], acc[rt][u]);
                    }
                }
            }
            if (t > 0) {  // h_{-1} = 0
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
            // the branch on the (uniform) pass sits OUTSIDE the unrolled element loops, and c / the temporary are
            // scalar arrays rather than 4-vectors: written per element on vectors, hipcc emits a four-way scalar
            // branch tree and register-tuple copies around every single value
#define FSN_REC_EPILOGUE(VAR, EXPR)                                                                   \
    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                                 \
    _Pragma("unroll") for (int u = 0; u < UG; ++u)                                                    \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                   \
        VAR[rt][u][i] = EXPR;                                                                         \
        asm volatile("" : "+v"(VAR[rt][u][i])); /* computed HERE: not sunk towards its use two passes later */ \
    }
            if (pass == 0) {
                FSN_REC_EPILOGUE(cst, sigmoid_fast(acc[rt][u][i]) * cst[rt][u][i])
            } else if (pass == 1) {
                FSN_REC_EPILOGUE(tmp, sigmoid_fast(acc[rt][u][i]))
            } else if (pass == 2) {
                FSN_REC_EPILOGUE(cst, cst[rt][u][i] + tmp[rt][u][i] * tanh_fast(acc[rt][u][i]))
            } else {
                FSN_REC_EPILOGUE(tmp, sigmoid_fast(acc[rt][u][i]) * tanh_fast(cst[rt][u][i]))
            }
#undef FSN_REC_EPILOGUE
            __builtin_amdgcn_sched_barrier(0);
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
        if (fuse_fc) {
            // output layer on the spot: 4 threads per (row, output), a quarter of K each, joined by two
            // lane shuffles; frame t - la of the mask (the first la steps are the look-ahead warm-up)
            const int tid = threadIdx.x;
            if (tid < ROWS * 8) {
                const int part = tid & 3, c = (tid >> 2) & 1, row = tid >> 3;
                const float* hp = hl + row * HS + part * (H / 4);
                const float* wp = wl + c * H + part * (H / 4);
                float a0 = 0.f, a1 = 0.f;
#pragma unroll 2  // deeper unrolling costs the registers the left-over step kernels need beside this one
                for (int k = 0; k < H / 4; k += 8) {
                    const f32x4 h0 = *reinterpret_cast<const f32x4*>(hp + k), w0 = *reinterpret_cast<const f32x4*>(wp + k);
                    const f32x4 h1 = *reinterpret_cast<const f32x4*>(hp + k + 4),
                                w1 = *reinterpret_cast<const f32x4*>(wp + k + 4);
                    a0 = fmaf(h0[0], w0[0], a0);
                    a0 = fmaf(h0[1], w0[1], a0);
                    a0 = fmaf(h0[2], w0[2], a0);
                    a0 = fmaf(h0[3], w0[3], a0);
                    a1 = fmaf(h1[0], w1[0], a1);
                    a1 = fmaf(h1[1], w1[1], a1);
                    a1 = fmaf(h1[2], w1[2], a1);
                    a1 = fmaf(h1[3], w1[3], a1);
                }
                float v = a0 + a1;
                v += __shfl_xor(v, 1, 64);
                v += __shfl_xor(v, 2, 64);
                const long n = n0 + row;
                if (part == 0 && t >= fc.la && n < fc.N) {
                    const long ng = n + fc.row0;
                    const int b = (int)(ng / fc.F), f = (int)(ng % fc.F);
                    (c ? fc.crm_i : fc.crm_r)[((long)b * fc.T + (t - fc.la)) * fc.FP + f] = v + fc.bias[c];
                }
            }
        } else {
            // stream h_t out as whole rows: hseq[t][n0 + row][0..H)
            float* dst = hseq + ((long)t * Npad + n0) * H;
            for (int i = threadIdx.x; i < ROWS * (H / 4); i += NW * 64) {
                const int row = i / (H / 4), c4 = i % (H / 4);
                *reinterpret_cast<f32x4*>(dst + (long)row * H + c4 * 4) =
                    *reinterpret_cast<const f32x4*>(hl + row * HS + c4 * 4);
            }
        }
    }
}

// Gate non-linearities on PAIRS of values (round 5): fp32 MFMAs and vector instructions do not overlap on a SIMD
// (tools/probe_overlap.hip: the times add whoever issues them), so every vector instruction of the persistent kernels is
// paid in full; v_pk_mul / v_pk_add / v_pk_fma_f32 do two lanes' worth of the multiplies and adds around the
// transcendentals per issue.  Same IEEE operations as sigmoid_fast / tanh_fast (1 - 2 r == fma(-2, r, 1) exactly).
__device__ __forceinline__ f32x2 sigmoid_fast2(f32x2 x) {
    const f32x2 t = x * f32x2{-1.4426950408889634f, -1.4426950408889634f};
    const f32x2 d = f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + f32x2{1.0f, 1.0f};
    return f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}
__device__ __forceinline__ f32x2 tanh_fast2(f32x2 x) {
    const f32x2 t = x * f32x2{2.8853900817779268f, 2.8853900817779268f};
    const f32x2 d = f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + f32x2{1.0f, 1.0f};
    const f32x2 r = f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    return __builtin_elementwise_fma(r, f32x2{-2.0f, -2.0f}, f32x2{1.0f, 1.0f});
}
__device__ __forceinline__ f32x2 lo2(f32x4 v) { return __builtin_shufflevector(v, v, 0, 1); }
__device__ __forceinline__ f32x2 hi2(f32x4 v) { return __builtin_shufflevector(v, v, 2, 3); }
__device__ __forceinline__ f32x4 cat2(f32x2 a, f32x2 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3); }
// cell state / staged values of a wave: scalars (the round-2 form) or whole accumulator-shaped vectors (the packed form)
template <bool PK> struct RecState { typedef float type[4]; };
template <> struct RecState<true> { typedef f32x4 type; };

// FSN_REC_GRU (a bit of the two persistent kernels' OPT / ABL parameter): nn.GRU (audio_zen/model/module/sequence_model.py:
// 59-66) on the same kernels, written as a FOUR-gate cell whose gate slots follow the kernels' order of evaluation
// (slot 1, 0, 2, 3):  slot 1 = r (W_ir x + W_hr h + b_ir + b_hr), slot 0 = nh (W_hn h + b_hn: zero input block),
// slot 2 = nx (W_in x + b_in: zero recurrent block), slot 3 = z (fsn_launch_gru_expand4, order 1).  Passes:
//   r:  tmp = sig(a)      nh: tmp = tmp a      nx: tmp = tanh(a + tmp) = n      z: h = n + sig(a) (h_{t-1} - n)
// (= (1 - z) n + z h_{t-1}); `cst` holds h in fp32 where the LSTM holds c.  The products of the two zero blocks are
// skipped (nx: no recurrent K loop; nh: no input chunks / slices from step 1 on): 3/4 of the LSTM's matrix work.
#define FSN_REC_GRU (1 << 20)

// ---------------------------------------------------------------------------------------------
// Last sub-band layer with its INPUT PROJECTION INSIDE: gates = b + x_t W_ih^T + h_{t-1} W_hh^T with x_t = h_t of the
// layer below, read from that layer's hidden sequence (4.8 GB at config 2).  The separate K = 384 projection GEMM and
// its 19.2 GB fragment-ordered `gx` round trip (written once, read back once per batch) are gone; the K loop of a gate
// pass is twice as long (768), so the per-pass costs - cell update, pass boundary, barriers - weigh half as much.
//
// h_{t-1} stays in LDS as in lstm_rec_kernel (16 RT x (H + 4) floats); x_t does not fit next to it, so it streams
// through a two-stage LDS ring in K slices of SK chunks (RT x SK fragments of 1 KB per stage).  The ring is filled by
// LDS-DMA (global_load_lds_dwordx4: no registers, nothing for the waves to wait on), one fragment per instruction:
// lane (row i, quarter q) fetches the 16 bytes x[row i][16 kc + 4 q ..] so that a stage holds the A fragments in the
// lane order ds_read_b128 wants (conflict-free by construction).  A gate pass walks the four slices of x_t and then
// h_{t-1}; the four passes of a step re-stream the same 96 KB tile (L2 hits), the next slice always in flight behind
// the current one.  Slice boundaries are LDS-only barriers (the DMA a wave issued a whole slice earlier has long
// landed: it is older than a dozen weight fragments the wave has consumed since, and loads return in order).
// Output layer fused exactly as in lstm_rec_kernel (the hidden sequence of this layer is never stored).
// ---------------------------------------------------------------------------------------------
// One frame of the sub-band model input for the persistent first-layer kernel, in two halves so that the memory
// latency is never exposed: `issue` requests ALL of a thread's elements (and their divisors) at once, `commit`
// divides and writes them to the LDS tile a gate pass later.  No 64-bit division per element: the rows of a
// workgroup are consecutive, so (b, f) of a row follow from (b0, f0) of the workgroup's first row by a carry.
// Same operands, same IEEE division as fsn_sb_input_value: bit-identical values.
// ROWSIN: the plain row-major form of the layer input (FsnSbInput::x_rows, COLS = 16 or 32 columns, zero padded in
// memory) - requested the same way, nothing to divide.
template <int NTHREADS, int ROWS, int EPT, int COLS = 32, bool ROWSIN = false>
struct SbStage {
    static constexpr int LOGC = COLS == 32 ? 5 : 4;
    static_assert(COLS == 32 || COLS == 16, "one or two K chunks");
    float raw[EPT], den[ROWSIN ? 1 : EPT];
    __device__ __forceinline__ void issue(const FsnSbInput& x, long n0, int b0, int f0, int t) {
        // the element indices do not depend on the step; left visible, the optimiser computes them once and keeps
        // a dozen registers live through the whole kernel
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        if constexpr (ROWSIN) {
            const float* frame = x.x_rows + ((long)t * x.x_step + n0) * x.x_ld;  // wave-uniform
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const int i = tid + e * NTHREADS;
                const int row = i >> LOGC, c = i & (COLS - 1);
                const bool ok = i < ROWS * COLS && n0 + row < x.N;
                raw[e] = frame[ok ? (long)row * x.x_ld + c : 0];  // branch-free; replaced by zero in commit()
            }
            return;
        }
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = tid + e * NTHREADS;
            const int row = i >> LOGC, c = i & (COLS - 1);  // 32 input columns (two K chunks)
            const long n = n0 + row;             // local row (validity, per-row divisors); (b, f) are global
            const bool ok = i < ROWS * COLS && n < x.N && c <= 2 * x.nb + 1;
            const unsigned fr = (unsigned)(f0 + row), q = fr / (unsigned)x.F;  // the carry: rows are consecutive
            const int b = b0 + (int)q, f = (int)(fr - q * (unsigned)x.F);
            const long fo = ((long)b * x.Tp + t) * x.FP;
            int j = f + c - x.nb;
            j = j < 0 ? -j : j;
            j = j >= x.F ? 2 * (x.F - 1) - j : j;
            const float* src = c <= 2 * x.nb ? x.mag + fo + j : x.fb_out + fo + f;
            // branch-free: an element that is not there reads element 0 and is replaced by zero in commit()
            raw[e] = *(ok ? src : x.mag);
            den[e] = x.den[ok ? (x.den_mode ? (long)t * x.den_stride + (n + x.row0) : (long)b) : 0];
        }
    }
    __device__ __forceinline__ void commit(const FsnSbInput& x, float* xl, int xs, long n0) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = tid + e * NTHREADS;
            const int row = i >> LOGC, c = i & (COLS - 1);
            if constexpr (ROWSIN) {
                if (i < ROWS * COLS) xl[row * xs + c] = n0 + row < x.N ? raw[e] : 0.f;
            } else {
                const bool ok = n0 + row < x.N && c <= 2 * x.nb + 1;
                if (i < ROWS * COLS) xl[row * xs + c] = ok ? raw[e] / den[e] : 0.f;
            }
        }
    }
};

// First sub-band layer, persistent (the successor of lstm_rec_kernel<.., XIN = true> for the 32-column sub-band
// input): gates = b + x_t W_ih^T (K = 32, x_t = freq_unfold ++ fb_output, normalised: fullsubnet/model.py:98-111,
// gathered into a double-buffered LDS tile) + h_{t-1} W_hh^T.  Against its predecessor:
//   - weight fragments travel in a two-deep buffer-load ring that runs through the x chunks, the h chunks, and on
//     into the next pass / step (a gate pass never starts cold; no per-load address arithmetic);
//   - the next frame's gather is requested at the start of a step and written to LDS one gate pass later, into
//     registers that are dead during that pass (its latency used to be exposed on all 12 waves once per step);
//   - the two barriers of a step only order LDS traffic;
//   - the bias of the next pass is requested a pass ahead.
// W_hh must follow W_ih in one packed buffer (element offset whh_off).  hseq [Tp][Npad][H] receives h_t.
// KX: K chunks of the layer input (2: the sub-band model's 32 columns; 1 or 2 with ROWSIN, the plain row-major input of
// any stacked LSTM's first layer - Fast FullSubNet's bottleneck is 16 columns wide: fast_fullsubnet/model.py:66-74).  With
// ONE x chunk a pass walks an odd number of K chunks, so the two weight-fragment registers sets swap roles from pass to pass
// (four passes per step: every step starts the same way).
#ifndef FSN_REC_IN_VCAP
#define FSN_REC_IN_VCAP 76  // x 2 = 152 registers: three waves per SIMD + a step workgroup beside them (tests/test_host_cpu.py)
#endif
template <int H, int RT, int UG, int OPT = 0, int KX = 2, bool ROWSIN = false>
__global__ __launch_bounds__((H / (16 * UG)) * 64) __attribute__((amdgpu_num_vgpr(FSN_REC_IN_VCAP))) void lstm_rec_in_kernel(
    const FsnSbInput xin, const float* __restrict__ w_p, unsigned whh_off, float* __restrict__ hseq, int Tp, int Npad) {
    constexpr int NW = H / (16 * UG);
    constexpr int KC = H / 16;
    static_assert(KX == 2 || (KX == 1 && ROWSIN), "the gathered sub-band input is two chunks wide");
    constexpr int HS = H + 4;
    constexpr int XS = 16 * KX + 4;  // row stride of the x tile
    constexpr int ROWS = RT * 16;
    constexpr int EPT = (ROWS * 16 * KX + NW * 64 - 1) / (NW * 64);
    extern __shared__ __attribute__((aligned(16))) float hl[];  // [ROWS][HS] | xl [2][ROWS][XS]
    float* xl = hl + ROWS * HS;

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const long n0 = (long)blockIdx.x * ROWS;
    const int sb_b0 = ROWSIN ? 0 : (int)((n0 + xin.row0) / xin.F);  // (b, f) of the workgroup's first row, once
    const int sb_f0 = ROWSIN ? 0 : (int)((n0 + xin.row0) - (long)sb_b0 * xin.F);

    constexpr bool PK = (OPT & 256) != 0;    // gate non-linearities on pairs (v_pk_*_f32), see sigmoid_fast2
    constexpr bool GRU = (OPT & FSN_REC_GRU) != 0;  // the GRU as a four-gate cell (see FSN_REC_GRU below lstm_rec_x_kernel's bits)
    static_assert(!GRU || PK, "the GRU cell is written on the packed epilogue");
    constexpr bool KOPT = (OPT & 4096) != 0 && RT >= 2 && RT <= 4 && KC % 6 == 0;  // see lstm_rec_x_kernel
    typename RecState<PK>::type cst[RT][UG], tmp[RT][UG];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int u = 0; u < UG; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) cst[rt][u][i] = 0.f;
    for (int i = threadIdx.x; i < ROWS * HS; i += NW * 64) hl[i] = 0.f;
    SbStage<NW * 64, ROWS, EPT, 16 * KX, ROWSIN> stage;
    stage.issue(xin, n0, sb_b0, sb_f0, 0);
    stage.commit(xin, xl, XS, n0);
    __syncthreads();

    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w_p), 0, 0x7fffffff, 0x00020000);
    const unsigned lane16 = (unsigned)lane * 16u;
    auto wload = [&](unsigned ofs) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane16, ofs * 4u, 0));
    };
    auto wxofs = [&](int g, int u) { return (unsigned)((g * KC + wave * UG + u) * KX) * 256u; };
    auto whofs = [&](int g, int u) { return whh_off + (unsigned)((g * KC + wave * UG + u) * KC) * 256u; };
    f32x4 b0[UG], b1[UG];
    float bias_n[UG];
    {
        int g0 = 1;
        asm volatile("" : "+s"(g0));
#pragma unroll
        for (int u = 0; u < UG; ++u) {
            b0[u] = wload(wxofs(g0, u));
            bias_n[u] = xin.bias[(g0 * KC + wave * UG + u) * 16 + lr];
        }
    }
    auto mma = [&](f32x4 (&acc)[RT][UG], const float* a, int a_rt_stride, const f32x4 (&b)[UG]) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(a + rt * a_rt_stride);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int u = 0; u < UG; ++u) acc[rt][u] = mfma16(av[jj], b[u][jj], acc[rt][u]);
        }
    };
    for (int t = 0; t < Tp; ++t) {
        // frame t + 1: requested now, written to the other x buffer after the first gate pass (that buffer was last
        // read in step t - 1 and is first read after the two barriers that end this step)
        const bool more = t + 1 < Tp && !(OPT & 2);  // (OPT & 2, probe: the price of the input gather)
        if (more) stage.issue(xin, n0, sb_b0, sb_f0, t + 1);
        const float* xa = xl + (t & 1) * ROWS * XS + lr * XS + 4 * lq;
        const float* ha = hl + lr * HS + 4 * lq;
        // gate order of evaluation: f (1), i (0), g (2), o (3)
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            __builtin_amdgcn_sched_barrier(0);
            int g = pass == 0 ? 1 : (pass == 1 ? 0 : pass);
            int gn = pass == 0 ? 0 : (pass == 1 ? 2 : (pass == 2 ? 3 : 1));  // the gate after this one
            asm volatile("" : "+s"(g));  // opaque: see lstm_rec_kernel
            asm volatile("" : "+s"(gn));
            const bool hpart = t > 0 && !(GRU && pass == 2);  // h_{-1} = 0; the GRU's nx gate has no recurrent part
            // ... and its nh gate no input part: skipped from step 1 on in the two-chunk form (with ONE x chunk the skipped pass
            // would change the parity of the fragment sets' roles; the zero block is multiplied there)
            const bool xpart = !(GRU && KX == 2 && pass == 1 && t > 0);
            const bool next_h_first = GRU && KX == 2 && pass == 0 && t > 0;  // the next pass opens with its recurrent product
            f32x4 acc[RT][UG];
            // B0: the fragments this pass starts with, B1: the other set; C0 / C1: the same for the recurrent product
            const bool SW = KX == 1 && (pass & 1);  // (a constant once the passes are unrolled)
            f32x4 (&B0)[UG] = SW ? b1 : b0;
            f32x4 (&B1)[UG] = SW ? b0 : b1;
            f32x4 (&C0)[UG] = KX == 1 ? B1 : B0;
            f32x4 (&C1)[UG] = KX == 1 ? B0 : B1;
            unsigned wx[UG], wh[UG], wxn[UG];
#pragma unroll
            for (int u = 0; u < UG; ++u) {
                wx[u] = wxofs(g, u);
                wh[u] = whofs(g, u);
                wxn[u] = next_h_first ? whofs(gn, u) : wxofs(gn, u);
                const float b = bias_n[u];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt][u] = f32x4{b, b, b, b};
                bias_n[u] = xin.bias[(gn * KC + wave * UG + u) * 16 + lr];  // a pass ahead
            }
            // ---- x_t W_ih^T: two chunks (or one) --------------------------------------------------
            if constexpr (KX == 2) {
              if (xpart) {
#pragma unroll
                for (int u = 0; u < UG; ++u) B1[u] = wload(wx[u] + 256u);
                __builtin_amdgcn_sched_barrier(0);
                mma(acc, xa, 16 * XS, B0);
#pragma unroll
                for (int u = 0; u < UG; ++u) B0[u] = wload(hpart ? wh[u] : wxn[u]);
                __builtin_amdgcn_sched_barrier(0);
                mma(acc, xa + 16, 16 * XS, B1);
              }
            } else {
#pragma unroll
                for (int u = 0; u < UG; ++u) B1[u] = wload(hpart ? wh[u] : wxn[u]);
                __builtin_amdgcn_sched_barrier(0);
                mma(acc, xa, 16 * XS, B0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- h_{t-1} W_hh^T (h_{-1} = 0) -------------------------------------------------------
            if (KOPT && hpart) {
                typedef const __attribute__((address_space(3))) float* lds_cptr;
                unsigned hb01 = (unsigned)(size_t)(lds_cptr)(hl + lr * HS + 4 * lq), hb23 = hb01 + 32u * HS * 4u;
                asm volatile("" : "+v"(hb01));
                asm volatile("" : "+v"(hb23));
                lds_cptr ha01 = (lds_cptr)(size_t)hb01;
                lds_cptr ha23 = (lds_cptr)(size_t)hb23;
                auto mmah = [&](int kofs, const f32x4 (&b)[UG]) {
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const f32x4 av = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>((rt < 2 ? ha01 : ha23) + (rt & 1) * 16 * HS + kofs);
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                            for (int u = 0; u < UG; ++u) acc[rt][u] = mfma16(av[jj], b[u][jj], acc[rt][u]);
                    }
                };
                if constexpr ((OPT & 32768) != 0) {
                    // APF: the first row tile's A fragment of block k + 1 is requested right behind block k's first-tile MFMAs
                    // (into the registers they have just read) instead of at the head of block k + 1, where the block's first
                    // MFMA waits for it; tiles 1 - 3 have the MFMAs before them to land
                    auto lda = [&](int rt, int kofs) {
                        return *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>((rt < 2 ? ha01 : ha23) + (rt & 1) * 16 * HS + kofs);
                    };
                    f32x4 apre = lda(0, 0);
                    auto blk = [&](int kofs, const f32x4 (&b)[UG], bool more) {
                        f32x4 av[RT];
#pragma unroll
                        for (int rt = 1; rt < RT; ++rt) av[rt] = lda(rt, kofs);
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                            for (int u = 0; u < UG; ++u) acc[0][u] = mfma16(apre[jj], b[u][jj], acc[0][u]);
                        __builtin_amdgcn_sched_barrier(0);
                        if (more) apre = lda(0, kofs + 16);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int rt = 1; rt < RT; ++rt)
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                                for (int u = 0; u < UG; ++u) acc[rt][u] = mfma16(av[rt][jj], b[u][jj], acc[rt][u]);
                    };
#pragma unroll 1
                    for (int hs = 0; hs < KC / 6; ++hs) {
#pragma unroll
                        for (int kk = 0; kk < 6; kk += 2) {
                            const int kc = hs * 6 + kk;
#pragma unroll
                            for (int u = 0; u < UG; ++u) C1[u] = wload(wh[u] + (unsigned)(kc + 1) * 256u);
                            __builtin_amdgcn_sched_barrier(0);
                            blk(kk * 16, C0, true);
                            __builtin_amdgcn_sched_barrier(0);
                            const bool more_h = kc + 2 < KC;
#pragma unroll
                            for (int u = 0; u < UG; ++u) C0[u] = wload(more_h ? wh[u] + (unsigned)(kc + 2) * 256u : wxn[u]);
                            __builtin_amdgcn_sched_barrier(0);
                            blk((kk + 1) * 16, C1, more_h);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        ha01 += 6 * 16;
                        ha23 += 6 * 16;
                    }
                } else
#pragma unroll 1
                for (int hs = 0; hs < KC / 6; ++hs) {
#pragma unroll
                    for (int kk = 0; kk < 6; kk += 2) {
                        const int kc = hs * 6 + kk;
#pragma unroll
                        for (int u = 0; u < UG; ++u) C1[u] = wload(wh[u] + (unsigned)(kc + 1) * 256u);
                        __builtin_amdgcn_sched_barrier(0);
                        mmah(kk * 16, C0);
                        __builtin_amdgcn_sched_barrier(0);
                        const bool more_h = kc + 2 < KC;
#pragma unroll
                        for (int u = 0; u < UG; ++u) C0[u] = wload(more_h ? wh[u] + (unsigned)(kc + 2) * 256u : wxn[u]);
                        __builtin_amdgcn_sched_barrier(0);
                        mmah((kk + 1) * 16, C1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    ha01 += 6 * 16;
                    ha23 += 6 * 16;
                }
            } else if (hpart) {
#pragma unroll 1
                for (int kc = 0; kc < KC; kc += 2) {
#pragma unroll
                    for (int u = 0; u < UG; ++u) C1[u] = wload(wh[u] + (unsigned)(kc + 1) * 256u);
                    __builtin_amdgcn_sched_barrier(0);
                    mma(acc, ha + kc * 16, 16 * HS, C0);
                    const bool more_h = kc + 2 < KC;
#pragma unroll
                    for (int u = 0; u < UG; ++u) C0[u] = wload(more_h ? wh[u] + (unsigned)(kc + 2) * 256u : wxn[u]);
                    __builtin_amdgcn_sched_barrier(0);
                    mma(acc, ha + (kc + 1) * 16, 16 * HS, C1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#define FSN_REC_EPILOGUE(VAR, EXPR)                                                                   \
    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                                 \
    _Pragma("unroll") for (int u = 0; u < UG; ++u)                                                    \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                   \
        VAR[rt][u][i] = EXPR;                                                                         \
        asm volatile("" : "+v"(VAR[rt][u][i]));                                                       \
    }
#define FSN_REC_EPILOGUE2(VAR, EXPR)                                                                  \
    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                                 \
    _Pragma("unroll") for (int u = 0; u < UG; ++u) {                                                  \
        const f32x4 A = acc[rt][u], C = cst[rt][u], M = tmp[rt][u];                                   \
        (void)A, (void)C, (void)M;                                                                    \
        auto half = [&](f32x2 a, f32x2 c, f32x2 m) { (void)a, (void)c, (void)m; return EXPR; };       \
        VAR[rt][u] = cat2(half(lo2(A), lo2(C), lo2(M)), half(hi2(A), hi2(C), hi2(M)));                \
        asm volatile("" : "+v"(VAR[rt][u]));                                                          \
    }
            if constexpr (GRU) {  // cst = h_{t-1} (and h_t after the z pass); tmp: r, r * (W_hn h + b_hn), n
                if (pass == 0) {
                    FSN_REC_EPILOGUE2(tmp, sigmoid_fast2(a))
                    if (more) stage.commit(xin, xl + ((t + 1) & 1) * ROWS * XS, XS, n0);
                } else if (pass == 1) {
                    FSN_REC_EPILOGUE2(tmp, m * a)
                } else if (pass == 2) {
                    FSN_REC_EPILOGUE2(tmp, tanh_fast2(a + m))
                } else {
                    FSN_REC_EPILOGUE2(cst, m + sigmoid_fast2(a) * (c - m))
                }
            } else if constexpr (PK) {
                if (pass == 0) {
                    FSN_REC_EPILOGUE2(cst, sigmoid_fast2(a) * c)
                    if (more) stage.commit(xin, xl + ((t + 1) & 1) * ROWS * XS, XS, n0);
                } else if (pass == 1) {
                    FSN_REC_EPILOGUE2(tmp, sigmoid_fast2(a))
                } else if (pass == 2) {
                    FSN_REC_EPILOGUE2(cst, c + m * tanh_fast2(a))
                } else {
                    FSN_REC_EPILOGUE2(tmp, sigmoid_fast2(a) * tanh_fast2(c))
                }
            } else if (pass == 0) {
                FSN_REC_EPILOGUE(cst, sigmoid_fast(acc[rt][u][i]) * cst[rt][u][i])
                if (more) stage.commit(xin, xl + ((t + 1) & 1) * ROWS * XS, XS, n0);  // tmp's registers are free here
            } else if (pass == 1) {
                FSN_REC_EPILOGUE(tmp, sigmoid_fast(acc[rt][u][i]))
            } else if (pass == 2) {
                FSN_REC_EPILOGUE(cst, cst[rt][u][i] + tmp[rt][u][i] * tanh_fast(acc[rt][u][i]))
            } else {
                FSN_REC_EPILOGUE(tmp, sigmoid_fast(acc[rt][u][i]) * tanh_fast(cst[rt][u][i]))
            }
#undef FSN_REC_EPILOGUE
#undef FSN_REC_EPILOGUE2
            __builtin_amdgcn_sched_barrier(0);
        }
        // every wave has finished reading h_{t-1}
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        {
            unsigned hwb = (unsigned)((4 * lq) * HS + (wave * UG) * 16 + lr);
            asm volatile("" : "+v"(hwb));  // re-derived every step: see lstm_rec_x_kernel
            float* hw = hl + hwb;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int u = 0; u < UG; ++u)
#pragma unroll
                    for (int i = 0; i < 4; ++i) hw[(rt * 16 + i) * HS + u * 16] = GRU ? cst[rt][u][i] : tmp[rt][u][i];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();  // h_t complete in LDS
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        // stream h_t out as whole rows: hseq[t][n0 + row][0..H)
        float* dst = hseq + ((long)t * Npad + n0) * H;
        // (Round 5, measured and not kept: the gather's step-independent index arithmetic read back from an LDS table, and
        // these stores from one address pair per step - ~250 fewer vector instructions per wave and step, no change in
        // time: the step's ends wait on the barriers anyway.  Without the stores 0.1 ms, without the gather 0.2.)
        if ((OPT & 1) && t + 1 < Tp) continue;  // probe: the price of streaming the hidden sequence out
        for (int i = threadIdx.x; i < ROWS * (H / 4); i += NW * 64) {
            const int row = i / (H / 4), c4 = i % (H / 4);
            const f32x4 v = *reinterpret_cast<const f32x4*>(hl + row * HS + c4 * 4);
            *reinterpret_cast<f32x4*>(dst + (long)row * H + c4 * 4) = v;
        }
    }
}

// One 16-byte-per-lane LDS-DMA fragment (1 KB per wave): lane l's 16 bytes at `g` land at LDS byte address
// lds_base + 16 l.  Written as asm so that the compiler neither serialises later LDS reads behind it (it cannot tell
// the ring stages apart and would wait for vmcnt(0) before every ds_read) nor counts it in its own vmcnt bookkeeping
// (an extra, OLDER request in the queue can only make its counted waits longer, never too short).
// (Non-temporal / sc0 sc1 fills were measured in round 5: +0.4 .. +0.8 ms - the slices are re-read from L2 / Infinity Cache.)
__device__ __forceinline__ void lds_dma_fragment(const float* g, unsigned lds_base) {
    unsigned saved;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, off\n\t"
        "s_nop 0\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(saved)
        : "s"(lds_base), "v"(g)
        : "memory");
}

// The same with the source address as a wave-uniform base (scalar registers) + this lane's byte offset: no per-fragment
// vector arithmetic at all.
__device__ __forceinline__ void lds_dma_fragment_s(const float* sbase, unsigned lane_bytes, unsigned lds_base) {
    unsigned saved;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %3\n\t"
        "s_nop 0\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(saved)
        : "s"(lds_base), "v"(lane_bytes), "s"(sbase)
        : "memory");
}

// ABL / OPT: the bits of the two persistent kernels' template parameter (tools/probe_rec_x.hip, probe_rec_in.hip).
//   Pricing bits, results WRONG: 1 no slice barriers, 2 no gate non-linearities, 4 no output layer, 8 no ring fills,
//   16 no step barriers (lstm_rec_in_kernel: 1 no hidden-sequence stores, 2 no input gather).
//   Forms with bit-identical results: 64 the barrier that opens a pass' first ring slice taken inside the previous pass'
//   recurrent product; 256 gate non-linearities on pairs (v_pk_*_f32); 4096 (+ 8192: h part only) the K loop without
//   per-chunk vector instructions; 32768 (layer 0) the next block's first-tile A fragment requested a block early;
//   131072 a slice's ring fills behind the first row tile of its first block; 262144 the output layer's tail without a
//   64-bit division per step; 524288 the fills' addresses from scalar registers.
// What each is worth, and the variants that were measured and removed from this file again (wave priorities, fills by
// one wave per SIMD in rotation, non-temporal fills / stores, zero-seeded accumulators, k-step-major MFMA order, two
// rolling A fragment sets, merged reciprocals, a step without its barriers): profiles/r05_rec_probes.md.
#ifndef FSN_REC_X_OPT
#define FSN_REC_X_OPT (64 | 256 | 4096 | 131072 | 262144 | 524288)  // 52.9 (round 4) -> 50.3 ms stand-alone
#endif
#ifndef FSN_REC_IN_OPT
#define FSN_REC_IN_OPT (256 | 4096 | 32768)  // 28.65 -> 27.4 ms stand-alone
#endif
#ifndef FSN_REC_VCAP
#define FSN_REC_VCAP 76  // x 2 on gfx950's unified register file = 152: three waves per SIMD + room for a step workgroup
#endif
// HSEQ: the layer is not the last one of its stack (or its output layer is not the fused two-row one): h_t is streamed
// out as whole rows to hseq_out [Tp][Npad][H], like lstm_rec_in_kernel does, and no output layer is formed - every
// stacked nn.LSTM layer of a SequenceModel (sequence_model.py:52-58) then takes its input from the layer below with
// no projection GEMM and no gx round trip (Fast FullSubNet's bottleneck: fast_fullsubnet/model.py:66-74).
template <int H, int RT, int UG, int ABL = 0, bool HSEQ = false>
__global__ __launch_bounds__((H / (16 * UG)) * 64) __attribute__((amdgpu_num_vgpr(FSN_REC_VCAP))) void lstm_rec_x_kernel(const float* __restrict__ xseq,
                                                                          const float* __restrict__ w_p,
                                                                          unsigned whh_off,
                                                                          const float* __restrict__ bias, int Tp,
                                                                          int Npad, const FsnRecFc fc) {
    // (HSEQ: the destination travels in fc.crm_r - the kernel's signature, and with it the register allocation of the
    // fused form, stays what it was)
    constexpr int NW = H / (16 * UG);
    constexpr int KC = H / 16;
    constexpr int HS = H + 4;
    constexpr int ROWS = RT * 16;
    constexpr int SK = 6;              // K chunks per x slice
    constexpr int NSL = KC / SK;       // slices per pass
    constexpr int NF = RT * SK;        // 1 KB fragments per ring stage
    static_assert(KC % SK == 0 && SK % 2 == 0 && KC % 2 == 0, "slice width must divide the K range; chunks go in pairs");
    // xs [2][NF][256] | hl [ROWS][HS] | wl [2][H].  The ring comes first: its LDS-DMA destination travels in M0,
    // and byte addresses below 64 KB are safe whatever width of M0 the DMA path honours.
    extern __shared__ __attribute__((aligned(16))) float xs[];
    float* hl = xs + 2 * NF * 256;
    float* wl = hl + ROWS * HS;

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const long n0 = (long)blockIdx.x * ROWS;
    if (!HSEQ)
        for (int i = threadIdx.x; i < 2 * H; i += NW * 64) {  // rows 0 / 1 of the packed output weights, un-tiled
            const int c = i / H, k = i % H;
            wl[i] = fc.w_p[(((k >> 4) * 64) + ((k & 15) >> 2) * 16 + c) * 4 + (k & 3)];
        }
    // FCT (ABL & 262144): the output layer's tail without its per-step overhead - (b, f) of the workgroup's first row once
    // (rows are consecutive; the per-step form divided two 64-bit integers per thread and step), the bias from scalar
    // registers, the quad sums by DPP instead of ds_bpermute, and the two product chains kept scalar (hipcc paired them
    // into v_pk_fma_f32 at the price of three v_mov per product).  Same arithmetic, same order.
    constexpr bool FCT = (ABL & 262144) != 0 && !HSEQ;
    int fc_b0 = 0, fc_f0 = 0;
    float fc_bias0 = 0.f, fc_bias1 = 0.f;
    if (FCT) {
        const long ng0 = n0 + fc.row0;
        fc_b0 = (int)(ng0 / fc.F);
        fc_f0 = (int)(ng0 - (long)fc_b0 * fc.F);
        fc_bias0 = fc.bias[0];
        fc_bias1 = fc.bias[1];
    }
    // KOPT: the recurrent product's K loop without per-chunk vector instructions.  Vector instructions and fp32 MFMAs share
    // the SIMD (tools/probe_overlap.hip), and the rolled loop spent 11 of them per 64 MFMAs: the hidden state sits beyond
    // the 64 KB an LDS read's immediate offset reaches, so every row tile's address was re-derived per chunk (6 v_add), and
    // the refilled weight fragments landed in fresh registers that were then copied (4 v_mov_b64 behind a vmcnt(0)).  Now:
    // two base registers (row tiles 0-1 / 2-3) advanced once per 6 chunks, immediates inside, and the refill of a
    // fragment pinned behind the last MFMA that reads it, so that it returns into the same registers.
    constexpr bool KOPT = (ABL & 4096) != 0 && RT >= 2 && RT <= 4;
    constexpr bool PK = (ABL & 256) != 0;          // gate non-linearities on pairs (v_pk_*_f32)
    constexpr bool GRU = (ABL & FSN_REC_GRU) != 0;  // the GRU as a four-gate cell (FSN_REC_GRU)
    static_assert(!GRU || (PK && (ABL & 64) && !(ABL & 2)), "the GRU cell is written on the shipped form");
    typename RecState<PK>::type cst[RT][UG], tmp[RT][UG];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int u = 0; u < UG; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) cst[rt][u][i] = 0.f;
    for (int i = threadIdx.x; i < ROWS * HS; i += NW * 64) hl[i] = 0.f;

    // ring stage `buf` <- slice `sl` of x_t: this wave's share of the NF fragments
    const unsigned xlane = (unsigned)(lr * H + 4 * lq);  // lane part of the source address; the rest is uniform
    const unsigned xs_lds = (unsigned)(size_t)(__attribute__((address_space(3))) float*)xs;  // LDS byte address
    auto fill = [&](int buf, int t, int sl) {
        const float* src = xseq + ((long)t * Npad + n0) * H + sl * (SK * 16);  // wave-uniform
        for (int f = wave; f < NF; f += NW) {
            const int rt = f / SK, kcl = f - rt * SK;
            lds_dma_fragment(src + (rt * 16 * H + kcl * 16) + xlane,
                             __builtin_amdgcn_readfirstlane(xs_lds + (unsigned)((buf * NF + f) * 1024)));
        }
    };
    // FSA (ABL & 524288): the fills' addresses from scalar registers - a wave's fragments of a stage are the same (row
    // tile, chunk) pairs all through the kernel, so their offsets are formed once, and the source is base + lane offset
    // (the loop form spent a 64-bit vector add, a vector add and a v_readfirstlane per fragment, inside a real loop)
    constexpr bool FSA = (ABL & 524288) != 0;
    constexpr int FPW = (NF + NW - 1) / NW;  // fragments per wave and stage (the last one only in the first waves if NW does not divide NF)
    int fsa_src[FPW], fsa_lds[FPW];
#pragma unroll
    for (int i = 0; i < FPW; ++i) {
        const int f = wave + i * NW, rt = f / SK, kcl = f - rt * SK;
        fsa_src[i] = __builtin_amdgcn_readfirstlane(rt * 16 * H + kcl * 16);
        fsa_lds[i] = __builtin_amdgcn_readfirstlane((int)xs_lds + f * 1024);
    }
    const unsigned xlane_bytes = xlane * 4u;
    auto fill_s = [&](int buf, int t, int sl) {
        const float* src = xseq + ((long)t * Npad + n0) * H + sl * (SK * 16);  // wave-uniform
#pragma unroll
        for (int i = 0; i < FPW; ++i)
            if (NF % NW == 0 || wave + i * NW < NF)
                lds_dma_fragment_s(src + fsa_src[i], xlane_bytes, (unsigned)(fsa_lds[i] + buf * (NF * 1024)));
    };
    fill(0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // Weight fragments travel in a two-deep ring (b0 / b1, written out by hand so that no copy is needed and every
    // wait is for the older of two requests); b0 always holds - or has in flight - the first fragments the next
    // K chunk needs, across slice, pass and step boundaries (the first chunk of a pass never starts cold).
    f32x4 b0[UG], b1[UG];
    // fragment (gate g, unit group u of this wave, chunk kc): element offset from w_p (W_ih, with W_hh whh_off
    // elements behind it) = a wave-uniform part (scalar registers) + 256 kc, + this lane's 16 bytes
    // buffer loads (T8): resource descriptor + scalar byte offset + this lane's constant 16 l - no per-load VGPR
    // address arithmetic at all
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w_p), 0, 0x7fffffff, 0x00020000);
    const unsigned lane16 = (unsigned)lane * 16u;
    auto wofs = [&](int g, int u) { return (unsigned)((g * KC + wave * UG + u) * KC) * 256u; };
    auto wload = [&](unsigned ofs) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane16, ofs * 4u, 0));
    };
    float bias_n[UG];  // the bias of the coming pass, requested a pass ahead
    {
        int g0 = 1;
        asm volatile("" : "+s"(g0));
#pragma unroll
        for (int u = 0; u < UG; ++u) {
            b0[u] = wload(wofs(g0, u));
            bias_n[u] = bias[(g0 * KC + wave * UG + u) * 16 + lr];
        }
    }
    // acc[rt][u] += A(16 rows x 16 k) B(16 k x 16 units): a = this lane's A fragment address of row tile 0
    auto mma = [&](f32x4 (&acc)[RT][UG], const float* a, int a_rt_stride, const f32x4 (&b)[UG]) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(a + rt * a_rt_stride);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int u = 0; u < UG; ++u) acc[rt][u] = mfma16(av[jj], b[u][jj], acc[rt][u]);
        }
    };

    // FLATE (ABL & 131072): a slice's ring fills are issued BEHIND the first row tile's MFMAs of the slice's first block
    // instead of ahead of the block: the block that follows a slice barrier - where all twelve waves stand together - opens
    // with two LDS reads and eight MFMAs, and the fills' instructions run under the other waves' MFMAs (the same gain was
    // measured with the fills left out: it is the shape of the block, not the fills' latency).
    constexpr bool FLATE = (ABL & 131072) != 0;
    auto mma_fill = [&](f32x4 (&acc)[RT][UG], const float* a, int a_rt_stride, const f32x4 (&b)[UG], int buf, int nt, int nsl) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(a), a1 = *reinterpret_cast<const f32x4*>(a + a_rt_stride);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int u = 0; u < UG; ++u) acc[0][u] = mfma16(a0[jj], b[u][jj], acc[0][u]);
        __builtin_amdgcn_sched_barrier(0);
        if (nt < Tp && !(ABL & 8)) {
            if (FSA) fill_s(buf, nt, nsl);
            else fill(buf, nt, nsl);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rt = 1; rt < RT; ++rt) {
            const f32x4 av = rt == 1 ? a1 : *reinterpret_cast<const f32x4*>(a + rt * a_rt_stride);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int u = 0; u < UG; ++u) acc[rt][u] = mfma16(av[jj], b[u][jj], acc[rt][u]);
        }
    };

    for (int t = 0; t < Tp; ++t) {
        // gate order of evaluation: f (1), i (0), g (2), o (3)
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            __builtin_amdgcn_sched_barrier(0);
            int g = pass == 0 ? 1 : (pass == 1 ? 0 : pass);
            int gn = pass == 0 ? 0 : (pass == 1 ? 2 : (pass == 2 ? 3 : 1));  // the gate after this one
            asm volatile("" : "+s"(g));  // opaque: see lstm_rec_kernel
            asm volatile("" : "+s"(gn));
            // GRU: the nx gate (pass 2) has no recurrent part, the nh gate (pass 1) no input part - its x slices are
            // skipped from step 1 on (at step 0 nothing else runs in that pass and the zero block keeps the ring's order)
            const bool hpart = t > 0 && !(GRU && pass == 2);
            const bool xpart = !(GRU && pass == 1 && t > 0);
            const bool prev_h = t > 0 && !(GRU && pass == 3);  // the pass before this one had a recurrent product
            f32x4 acc[RT][UG];
            unsigned wx[UG], wh[UG], wxn[UG];  // uniform offsets: W_ih / W_hh of this gate, the next pass' first fragment
#pragma unroll
            for (int u = 0; u < UG; ++u) {
                wx[u] = wofs(g, u);
                wh[u] = wx[u] + whh_off;
                wxn[u] = wofs(gn, u) + ((GRU && pass == 0 && t > 0) ? whh_off : 0u);
                const float b = bias_n[u];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt][u] = f32x4{b, b, b, b};
                bias_n[u] = bias[(gn * KC + wave * UG + u) * 16 + lr];
            }
            // ---- x_t W_ih^T, slice by slice ------------------------------------------------------
#pragma unroll 1
            for (int sl = 0; sl < (xpart ? NSL : 0); ++sl) {
                const int j = pass * NSL + sl;  // slice counter of the step: ring stage j & 1
                // (ABL & 64, experiment: the barrier that opens a pass' first slice is taken in the middle of the previous
                // pass' recurrent product instead - its fills were issued before that product began - so that no barrier
                // follows the gate non-linearities)
                if (j > 0 && !(ABL & 1) && !((ABL & 64) && sl == 0 && prev_h)) {
                    // this wave's fills of stage j & 1 were issued a slice ago, before UG SK weight fragments it has
                    // consumed since; at most the UG prefetched ones are still in flight
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(UG) : "memory");
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
                }
                // next slice into the other stage (last read in slice j - 1, which every wave has left):
                // same x_t for the next pass, x_{t+1} after the last pass
                const int nsl = sl + 1 < NSL ? sl + 1 : 0;
                const int nt = (sl + 1 < NSL || pass < 3) ? t : t + 1;
                if (!FLATE) {
                    if (nt < Tp && !(ABL & 8)) fill((j + 1) & 1, nt, nsl);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const float* xa = xs + ((j & 1) * NF) * 256 + lane * 4;
#pragma unroll
                for (int kk = 0; kk < SK; kk += 2) {
                    const int kc = sl * SK + kk;
#pragma unroll
                    for (int u = 0; u < UG; ++u)
                        b1[u] = wload(wx[u] + (unsigned)(kc + 1) * 256u);
                    __builtin_amdgcn_sched_barrier(0);  // requests first, pinned: hipcc otherwise sinks them to their use
                    if (FLATE && kk == 0) mma_fill(acc, xa, SK * 256, b0, (j + 1) & 1, nt, nsl);
                    else mma(acc, xa + kk * 256, SK * 256, b0);
                    if (KOPT && !(ABL & 8192)) __builtin_amdgcn_sched_barrier(0);
                    // chunk kc + 2: W_ih, or the first chunk of W_hh, or (h_{-1} = 0: no W_hh product) of the next pass
                    const bool more_x = kc + 2 < KC;
#pragma unroll
                    for (int u = 0; u < UG; ++u) {
                        b0[u] = wload(more_x ? wx[u] + (unsigned)(kc + 2) * 256u : (hpart ? wh[u] : wxn[u]));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    mma(acc, xa + (kk + 1) * 256, SK * 256, b1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // ---- h_{t-1} W_hh^T (h_{-1} = 0) -------------------------------------------------------
            if (KOPT && hpart) {
                // two LDS byte addresses, opaque to the optimiser (it would fold the tile's own offset into the immediates
                // and overflow them again): row tiles 0-1 / 2-3
                typedef const __attribute__((address_space(3))) float* lds_cptr;
                unsigned hb01 = (unsigned)(size_t)(lds_cptr)(hl + lr * HS + 4 * lq), hb23 = hb01 + 32u * HS * 4u;
                asm volatile("" : "+v"(hb01));
                asm volatile("" : "+v"(hb23));
                lds_cptr ha01 = (lds_cptr)(size_t)hb01;
                lds_cptr ha23 = (lds_cptr)(size_t)hb23;
                auto mmah = [&](int kofs, const f32x4 (&b)[UG]) {
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const f32x4 av = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>((rt < 2 ? ha01 : ha23) + (rt & 1) * 16 * HS + kofs);
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                            for (int u = 0; u < UG; ++u) acc[rt][u] = mfma16(av[jj], b[u][jj], acc[rt][u]);
                    }
                };
#pragma unroll 1
                for (int hs = 0; hs < NSL; ++hs) {
                    if ((ABL & 64) && pass < 3 && hs == NSL / 2) {
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(UG) : "memory");
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
                        __builtin_amdgcn_s_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
                    }
#pragma unroll
                    for (int kk = 0; kk < SK; kk += 2) {
                        const int kc = hs * SK + kk;
#pragma unroll
                        for (int u = 0; u < UG; ++u) b1[u] = wload(wh[u] + (unsigned)(kc + 1) * 256u);
                        __builtin_amdgcn_sched_barrier(0);
                        mmah(kk * 16, b0);
                        __builtin_amdgcn_sched_barrier(0);  // the refill behind the last MFMA that reads b0: same registers
                        const bool more_h = kc + 2 < KC;
#pragma unroll
                        for (int u = 0; u < UG; ++u) b0[u] = wload(more_h ? wh[u] + (unsigned)(kc + 2) * 256u : wxn[u]);
                        __builtin_amdgcn_sched_barrier(0);
                        mmah((kk + 1) * 16, b1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    ha01 += SK * 16;
                    ha23 += SK * 16;
                }
            } else if (hpart) {
                const float* ha = hl + lr * HS + 4 * lq;
#pragma unroll 1
                for (int kc = 0; kc < KC; kc += 2) {
                    if ((ABL & 64) && pass < 3 && kc == KC / 2) {
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(UG) : "memory");
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
                        __builtin_amdgcn_s_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
                    }
#pragma unroll
                    for (int u = 0; u < UG; ++u)
                        b1[u] = wload(wh[u] + (unsigned)(kc + 1) * 256u);
                    __builtin_amdgcn_sched_barrier(0);
                    mma(acc, ha + kc * 16, 16 * HS, b0);
                    const bool more_h = kc + 2 < KC;
#pragma unroll
                    for (int u = 0; u < UG; ++u) {
                        b0[u] = wload(more_h ? wh[u] + (unsigned)(kc + 2) * 256u : wxn[u]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    mma(acc, ha + (kc + 1) * 16, 16 * HS, b1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#define FSN_REC_EPILOGUE(VAR, EXPR)                                                                   \
    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                                 \
    _Pragma("unroll") for (int u = 0; u < UG; ++u)                                                    \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                   \
        VAR[rt][u][i] = EXPR;                                                                         \
        asm volatile("" : "+v"(VAR[rt][u][i]));                                                       \
    }
#define FSN_REC_EPILOGUE2(VAR, EXPR)                                                                  \
    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                                 \
    _Pragma("unroll") for (int u = 0; u < UG; ++u) {                                                  \
        const f32x4 A = acc[rt][u], C = cst[rt][u], M = tmp[rt][u];                                   \
        (void)A, (void)C, (void)M;                                                                    \
        auto half = [&](f32x2 a, f32x2 c, f32x2 m) { (void)a, (void)c, (void)m; return EXPR; };       \
        VAR[rt][u] = cat2(half(lo2(A), lo2(C), lo2(M)), half(hi2(A), hi2(C), hi2(M)));                \
        asm volatile("" : "+v"(VAR[rt][u]));                                                          \
    }
            if constexpr (GRU) {  // cst = h_{t-1} (h_t after the z pass); tmp: r, r * (W_hn h + b_hn), n
                if (pass == 0) {
                    FSN_REC_EPILOGUE2(tmp, sigmoid_fast2(a))
                } else if (pass == 1) {
                    FSN_REC_EPILOGUE2(tmp, m * a)
                } else if (pass == 2) {
                    FSN_REC_EPILOGUE2(tmp, tanh_fast2(a + m))
                } else {
                    FSN_REC_EPILOGUE2(cst, m + sigmoid_fast2(a) * (c - m))
                }
            } else if constexpr (PK && !(ABL & 2)) {
                if (pass == 0) {
                    FSN_REC_EPILOGUE2(cst, sigmoid_fast2(a) * c)
                } else if (pass == 1) {
                    FSN_REC_EPILOGUE2(tmp, sigmoid_fast2(a))
                } else if (pass == 2) {
                    FSN_REC_EPILOGUE2(cst, c + m * tanh_fast2(a))
                } else {
                    FSN_REC_EPILOGUE2(tmp, sigmoid_fast2(a) * tanh_fast2(c))
                }
            } else if constexpr (PK) {
            } else if (ABL & 2) {
                if (pass == 0 || pass == 2) {
                    FSN_REC_EPILOGUE(cst, acc[rt][u][i] * 0.5f)
                } else {
                    FSN_REC_EPILOGUE(tmp, acc[rt][u][i] * 0.5f)
                }
            } else if (pass == 0) {
                FSN_REC_EPILOGUE(cst, sigmoid_fast(acc[rt][u][i]) * cst[rt][u][i])
            } else if (pass == 1) {
                FSN_REC_EPILOGUE(tmp, sigmoid_fast(acc[rt][u][i]))
            } else if (pass == 2) {
                FSN_REC_EPILOGUE(cst, cst[rt][u][i] + tmp[rt][u][i] * tanh_fast(acc[rt][u][i]))
            } else {
                FSN_REC_EPILOGUE(tmp, sigmoid_fast(acc[rt][u][i]) * tanh_fast(cst[rt][u][i]))
            }
#undef FSN_REC_EPILOGUE
#undef FSN_REC_EPILOGUE2
            __builtin_amdgcn_sched_barrier(0);
        }
        // every wave has finished reading h_{t-1}; the fill of the next step's first slice stays in flight
        if (!(ABL & 16)) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        }
        {
            // one base register, re-derived every step (opaque to the optimiser): hoisted out of the time loop the
            // 32 store addresses become 32 live registers that end up in scratch
            unsigned hwb = (unsigned)((4 * lq) * HS + (wave * UG) * 16 + lr);
            asm volatile("" : "+v"(hwb));
            float* hw = hl + hwb;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int u = 0; u < UG; ++u)
#pragma unroll
                    for (int i = 0; i < 4; ++i) hw[(rt * 16 + i) * HS + u * 16] = GRU ? cst[rt][u][i] : tmp[rt][u][i];
        }
        if (!(ABL & 16)) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            __builtin_amdgcn_s_barrier();  // h_t complete in LDS
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        }
        if (HSEQ) {
            // stream h_t out as whole rows: hseq_out[t][n0 + row][0..H)
            float* dst = fc.crm_r + ((long)t * Npad + n0) * H;
            for (int i = threadIdx.x; i < ROWS * (H / 4); i += NW * 64) {
                const int row = i / (H / 4), c4 = i % (H / 4);
                *reinterpret_cast<f32x4*>(dst + (long)row * H + c4 * 4) =
                    *reinterpret_cast<const f32x4*>(hl + row * HS + c4 * 4);
            }
        } else if (FCT && !(ABL & 4)) {
            const int tid = threadIdx.x;
            if (tid < ROWS * 8) {
                const int part = tid & 3, c = (tid >> 2) & 1, row = tid >> 3;
                const float* hp = hl + row * HS + part * (H / 4);
                const float* wp = wl + c * H + part * (H / 4);
                float a0 = 0.f, a1 = 0.f;
#pragma unroll 2
                for (int k = 0; k < H / 4; k += 8) {
                    const f32x4 h0 = *reinterpret_cast<const f32x4*>(hp + k), w0 = *reinterpret_cast<const f32x4*>(wp + k);
                    const f32x4 h1 = *reinterpret_cast<const f32x4*>(hp + k + 4),
                                w1 = *reinterpret_cast<const f32x4*>(wp + k + 4);
                    a0 = fmaf(h0[0], w0[0], a0);
                    a0 = fmaf(h0[1], w0[1], a0);
                    a0 = fmaf(h0[2], w0[2], a0);
                    a0 = fmaf(h0[3], w0[3], a0);
                    asm volatile("" : "+v"(a0));  // not a twin of the other chain any more: no pairing
                    a1 = fmaf(h1[0], w1[0], a1);
                    a1 = fmaf(h1[1], w1[1], a1);
                    a1 = fmaf(h1[2], w1[2], a1);
                    a1 = fmaf(h1[3], w1[3], a1);
                }
                float v = a0 + a1;
                v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
                v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
                if (part == 0 && t >= fc.la && (int)n0 + row < fc.N) {
                    int f = fc_f0 + row, b = fc_b0;
                    while (f >= fc.F) {
                        f -= fc.F;
                        ++b;
                    }
                    (c ? fc.crm_i : fc.crm_r)[((long)b * fc.T + (t - fc.la)) * fc.FP + f] = v + (c ? fc_bias1 : fc_bias0);
                }
            }
        } else if (!(ABL & 4)) {
            // output layer on the spot (nn.Linear(H, 2)): 4 threads per (row, output), a quarter of K each
            const int tid = threadIdx.x;
            if (tid < ROWS * 8) {
                const int part = tid & 3, c = (tid >> 2) & 1, row = tid >> 3;
                const float* hp = hl + row * HS + part * (H / 4);
                const float* wp = wl + c * H + part * (H / 4);
                float a0 = 0.f, a1 = 0.f;
#pragma unroll 2
                for (int k = 0; k < H / 4; k += 8) {
                    const f32x4 h0 = *reinterpret_cast<const f32x4*>(hp + k), w0 = *reinterpret_cast<const f32x4*>(wp + k);
                    const f32x4 h1 = *reinterpret_cast<const f32x4*>(hp + k + 4),
                                w1 = *reinterpret_cast<const f32x4*>(wp + k + 4);
                    a0 = fmaf(h0[0], w0[0], a0);
                    a0 = fmaf(h0[1], w0[1], a0);
                    a0 = fmaf(h0[2], w0[2], a0);
                    a0 = fmaf(h0[3], w0[3], a0);
                    a1 = fmaf(h1[0], w1[0], a1);
                    a1 = fmaf(h1[1], w1[1], a1);
                    a1 = fmaf(h1[2], w1[2], a1);
                    a1 = fmaf(h1[3], w1[3], a1);
                }
                float v = a0 + a1;
                v += __shfl_xor(v, 1, 64);
                v += __shfl_xor(v, 2, 64);
                const long n = n0 + row;
                if (part == 0 && t >= fc.la && n < fc.N) {
                    const long ng = n + fc.row0;
                    const int b = (int)(ng / fc.F), f = (int)(ng % fc.F);
                    (c ? fc.crm_i : fc.crm_r)[((long)b * fc.T + (t - fc.la)) * fc.FP + f] = v + fc.bias[c];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Small-N variant (RT <= 2, i.e. fewer row tiles than ~2 per CU: small batches, per-rank shards of a
// strong-scaled batch, single utterances).  With 16-32 rows per workgroup a K chunk is only 8-16 MFMAs
// per gate, far shorter than the L2 latency of its B fragments, and lstm_rec_kernel's four sequential
// gate passes leave the step latency-bound (52 us per step at RT = 1, 10 us of MFMA).  Registers are
// plentiful here, so all four gates accumulate in ONE pass: a quarter of the dependent chunk
// iterations, four times the MFMA work between two waits, no temporaries in the cell update.
template <int H, int RT, int UG, bool XIN>
__global__ __launch_bounds__((H / (16 * UG)) * 64) void lstm_rec_small_kernel(const float* __restrict__ gx,
                                                                              const FsnSbInput xin,
                                                                              const float* __restrict__ whh_p,
                                                                              float* __restrict__ hseq, int Tp,
                                                                              int Npad, const FsnRecFc) {
    constexpr int NW = H / (16 * UG);
    constexpr int KC = H / 16;
    constexpr int CT = 4 * KC;
    constexpr int HS = H + 4;
    constexpr int ROWS = RT * 16;
    extern __shared__ __attribute__((aligned(16))) float hl[];
    const int XS = XIN ? 16 * xin.kin_chunks + 4 : 0;
    float* xl = hl + ROWS * HS;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const long n0 = (long)blockIdx.x * ROWS;

    f32x4 cst[RT][UG];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int u = 0; u < UG; ++u) cst[rt][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < ROWS * HS; i += NW * 64) hl[i] = 0.f;
    if (XIN) stage_sb_input<NW * 64>(xin, xl, XS, n0, ROWS, 0);
    __syncthreads();

    const float* bp[4][UG];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int u = 0; u < UG; ++u) bp[g][u] = whh_p + ((long)(g * KC + wave * UG + u) * KC * 64 + lane) * 4;

    for (int t = 0; t < Tp; ++t) {
        const long gx_rt0 = ((long)t * Npad + n0) >> 4;
        if (XIN && t + 1 < Tp) stage_sb_input<NW * 64>(xin, xl + ((t + 1) & 1) * ROWS * XS, XS, n0, ROWS, t + 1);
        const float* xt = xl + (t & 1) * ROWS * XS;
        f32x4 acc[4][RT][UG];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int u = 0; u < UG; ++u) {
                const int ug = wave * UG + u;
                if (!XIN) {
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
                        acc[g][rt][u] = *reinterpret_cast<const f32x4*>(
                            gx + (((gx_rt0 + rt) * CT + g * KC + ug) * 64 + lane) * 4);
                } else {
                    const float bias = xin.bias[(g * KC + ug) * 16 + lr];
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[g][rt][u] = f32x4{bias, bias, bias, bias};
                }
            }
        if (XIN) {
            for (int kx = 0; kx < xin.kin_chunks; ++kx) {
                f32x4 a[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    a[rt] = *reinterpret_cast<const f32x4*>(xt + (rt * 16 + lr) * XS + kx * 16 + 4 * lq);
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int u = 0; u < UG; ++u) {
                        const f32x4 bx = *reinterpret_cast<const f32x4*>(
                            xin.wih_p + (((long)(g * KC + wave * UG + u) * xin.kin_chunks + kx) * 64 + lane) * 4);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int rt = 0; rt < RT; ++rt) acc[g][rt][u] = mfma16(a[rt][j], bx[j], acc[g][rt][u]);
                    }
            }
        }
        if (t > 0) {  // h_{-1} = 0
            f32x4 bn[4][UG];
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int u = 0; u < UG; ++u) bn[g][u] = *reinterpret_cast<const f32x4*>(bp[g][u]);
#pragma unroll 1
            for (int kc = 0; kc < KC; ++kc) {
                f32x4 bc[4][UG], a[RT];
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int u = 0; u < UG; ++u) bc[g][u] = bn[g][u];
                const int kn = kc + 1 < KC ? kc + 1 : kc;  // clamped: branch-free, counted waits
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int u = 0; u < UG; ++u)
                        bn[g][u] = *reinterpret_cast<const f32x4*>(bp[g][u] + (long)kn * 256);
                const float* ap = hl + lr * HS + kc * 16 + 4 * lq;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) a[rt] = *reinterpret_cast<const f32x4*>(ap + rt * 16 * HS);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                            for (int u = 0; u < UG; ++u)
                                acc[g][rt][u] = mfma16(a[rt][j], bc[g][u][j], acc[g][rt][u]);
            }
        }
        f32x4 hv[RT][UG];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int u = 0; u < UG; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float ig = sigmoid_fast(acc[0][rt][u][i]), fg = sigmoid_fast(acc[1][rt][u][i]);
                    const float gg = tanh_fast(acc[2][rt][u][i]), og = sigmoid_fast(acc[3][rt][u][i]);
                    const float cn = fg * cst[rt][u][i] + ig * gg;
                    cst[rt][u][i] = cn;
                    hv[rt][u][i] = og * tanh_fast(cn);
                }
        __syncthreads();
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int u = 0; u < UG; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    hl[(rt * 16 + 4 * lq + i) * HS + (wave * UG + u) * 16 + lr] = hv[rt][u][i];
        __syncthreads();
        float* dst = hseq + ((long)t * Npad + n0) * H;
        for (int i = threadIdx.x; i < ROWS * (H / 4); i += NW * 64) {
            const int row = i / (H / 4), c4 = i % (H / 4);
            *reinterpret_cast<f32x4*>(dst + (long)row * H + c4 * 4) =
                *reinterpret_cast<const f32x4*>(hl + row * HS + c4 * 4);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// One time step.  grid = (H/16 unit groups, ceil(row tiles / RTS)); a workgroup owns RTS 16-row
// tiles x one 16-unit group x the four gates; its 4 waves split K four ways (each wave holds the
// RTS x 4 partial tiles, so one W_hh fragment load feeds RTS MFMAs) and the partials are reduced
// through LDS in a fixed order (deterministic); wave w < RTS then finishes row tile w.  RTS = 1 for
// the few-row launches of inference (full-band model, left-over tiles), larger for the training step.
template <int RTS>
__device__ __forceinline__ void lstm_step_body(const float* __restrict__ gx, const float* __restrict__ whh_p,
                                               const float* __restrict__ h_prev, float* __restrict__ h_out,
                                               const float* c_prev, float* c, float* __restrict__ gates_out,
                                               long gx_rt0, int row_tiles, int H, int first) {
    __shared__ f32x4 red[4][RTS][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int ug = blockIdx.x, rtile0 = blockIdx.y * RTS;
    const int KC = H >> 4, CT = 4 * KC;
    f32x4 acc[RTS][4];
#pragma unroll
    for (int rt = 0; rt < RTS; ++rt)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[rt][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    // what the finishing wave's epilogue reads (projection tiles, previous cell state) is requested before the K
    // loop: cold lines, each of which would otherwise cost a memory round trip after the barrier
    const int ftile = rtile0 + wave < row_tiles ? rtile0 + wave : row_tiles - 1;
    f32x4 addv[4];
    float c_old[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
        addv[g] = *reinterpret_cast<const f32x4*>(gx + (((gx_rt0 + ftile) * CT + g * KC + ug) * 64 + lane) * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) c_old[i] = first ? 0.f : c_prev[((long)ftile * 16 + 4 * lq + i) * H + ug * 16 + lr];
    if (!first) {
        const int kc0 = wave * (KC >> 2), kc1 = kc0 + (KC >> 2);
        const float* ap[RTS];
#pragma unroll
        for (int rt = 0; rt < RTS; ++rt) {
            int rtile = rtile0 + rt;
            rtile = rtile < row_tiles ? rtile : row_tiles - 1;
            ap[rt] = h_prev + ((long)rtile * 16 + lr) * H + 4 * lq;
        }
#pragma unroll 2
        for (int kc = kc0; kc < kc1; ++kc) {
            f32x4 a[RTS], b[4];
#pragma unroll
            for (int rt = 0; rt < RTS; ++rt) a[rt] = *reinterpret_cast<const f32x4*>(ap[rt] + kc * 16);
#pragma unroll
            for (int g = 0; g < 4; ++g)
                b[g] = *reinterpret_cast<const f32x4*>(whh_p + (((long)(g * KC + ug) * KC + kc) * 64 + lane) * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int rt = 0; rt < RTS; ++rt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[rt][g] = mfma16(a[rt][j], b[g][j], acc[rt][g]);
        }
#pragma unroll
        for (int rt = 0; rt < RTS; ++rt)
#pragma unroll
            for (int g = 0; g < 4; ++g) red[wave][rt][g][lane] = acc[rt][g];
        __syncthreads();
    }
    const int rt = wave, rtile = rtile0 + rt;
    if (rt >= RTS || rtile >= row_tiles) return;
    f32x4 pre[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (!first) {  // partials summed in wave order 0, 1, 2, 3
            v = red[0][rt][g][lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const f32x4 r = red[w][rt][g][lane];
                v = f32x4{v[0] + r[0], v[1] + r[1], v[2] + r[2], v[3] + r[3]};
            }
        }
        const f32x4 x = addv[g];
        pre[g] = f32x4{v[0] + x[0], v[1] + x[1], v[2] + x[2], v[3] + x[3]};
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long row = (long)rtile * 16 + 4 * lq + i;
        const long idx = row * H + ug * 16 + lr;
        // hardware exp / rcp forms (fsn_common.h): the libm ones were ~12 % of this kernel at 129 row tiles
        const float ig = sigmoid_fast(pre[0][i]), fg = sigmoid_fast(pre[1][i]);
        const float gg = tanh_fast(pre[2][i]), og = sigmoid_fast(pre[3][i]);
        const float cn = fg * c_old[i] + ig * gg;
        c[idx] = cn;
        h_out[idx] = og * tanh_fast(cn);
        if (gates_out) {  // training: keep the activated gates for the backward pass, [row][4H]
            float* gp = gates_out + row * 4 * H + ug * 16 + lr;
            gp[0] = ig;
            gp[H] = fg;
            gp[2 * H] = gg;
            gp[3 * H] = og;
        }
    }
}

template <int RTS>
__global__ __launch_bounds__(256) void lstm_step_kernel(const float* __restrict__ gx,
                                                        const float* __restrict__ whh_p,
                                                        const float* __restrict__ h_prev,
                                                        float* __restrict__ h_out, const float* c_prev,
                                                        float* c, float* __restrict__ gates_out, long gx_rt0,
                                                        int row_tiles, int H, int first) {
    lstm_step_body<RTS>(gx, whh_p, h_prev, h_out, c_prev, c, gates_out, gx_rt0, row_tiles, H, first);
}

// Many rows (the training step's 129 row tiles and more): no split-K at all.  A workgroup takes RW * 4 row
// tiles of one unit group; every wave owns RW of them for the whole K range and finishes them itself, so there
// is no partial-sum exchange, no barrier, and the cell update runs on all four waves instead of one.  The four
// waves read the same W_hh fragments (L1 hits after the first).
template <int RW>
__global__ __launch_bounds__(256) void lstm_step_rows_kernel(const float* __restrict__ gx,
                                                             const float* __restrict__ whh_p,
                                                             const float* __restrict__ h_prev,
                                                             float* __restrict__ h_out, const float* c_prev, float* c,
                                                             float* __restrict__ gates_out, long gx_rt0, int row_tiles,
                                                             int H, int first) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int ug = blockIdx.x, rtile0 = (blockIdx.y * 4 + wave) * RW;
    if (rtile0 >= row_tiles) return;
    const int KC = H >> 4, CT = 4 * KC;
    f32x4 acc[RW][4];
    int rtile[RW];
#pragma unroll
    for (int rt = 0; rt < RW; ++rt) {
        rtile[rt] = rtile0 + rt < row_tiles ? rtile0 + rt : row_tiles - 1;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            acc[rt][g] = *reinterpret_cast<const f32x4*>(gx + (((gx_rt0 + rtile[rt]) * CT + g * KC + ug) * 64 + lane) * 4);
    }
    // previous cell state requested before the K loop: read in the epilogue, every element would pay its own
    // memory round trip (and the wait for it also waits for the stores of the element before)
    float c_old[RW][4];
#pragma unroll
    for (int rt = 0; rt < RW; ++rt)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            c_old[rt][i] = first ? 0.f : c_prev[((long)rtile[rt] * 16 + 4 * lq + i) * H + ug * 16 + lr];
    if (!first) {
        const float* ap[RW];
#pragma unroll
        for (int rt = 0; rt < RW; ++rt) ap[rt] = h_prev + ((long)rtile[rt] * 16 + lr) * H + 4 * lq;
        const float* bp = whh_p + ((long)ug * KC * 64 + lane) * 4;
        const long gstride = (long)KC * KC * 256;
#pragma unroll 2
        for (int kc = 0; kc < KC; ++kc) {
            f32x4 a[RW], b[4];
#pragma unroll
            for (int rt = 0; rt < RW; ++rt) a[rt] = *reinterpret_cast<const f32x4*>(ap[rt] + kc * 16);
#pragma unroll
            for (int g = 0; g < 4; ++g) b[g] = *reinterpret_cast<const f32x4*>(bp + g * gstride + (long)kc * 256);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int rt = 0; rt < RW; ++rt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[rt][g] = mfma16(a[rt][j], b[g][j], acc[rt][g]);
        }
    }
#pragma unroll
    for (int rt = 0; rt < RW; ++rt) {
        if (rtile0 + rt >= row_tiles) break;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long row = (long)rtile[rt] * 16 + 4 * lq + i;
            const long idx = row * H + ug * 16 + lr;
            const float ig = sigmoid_fast(acc[rt][0][i]), fg = sigmoid_fast(acc[rt][1][i]);
            const float gg = tanh_fast(acc[rt][2][i]), og = sigmoid_fast(acc[rt][3][i]);
            const float cn = fg * c_old[rt][i] + ig * gg;
            c[idx] = cn;
            h_out[idx] = og * tanh_fast(cn);
            if (gates_out) {
                float* gp = gates_out + row * 4 * H + ug * 16 + lr;
                gp[0] = ig;
                gp[H] = fg;
                gp[2 * H] = gg;
                gp[3 * H] = og;
            }
        }
    }
}

// K loop of lstm_step_cu_kernel: acc[u][g] += A(16 rows x H) W_hh(u, g)^T for the
// wave's row tile, UGW unit groups and four gates.  The four waves of the workgroup need the SAME 4 UGW weight
// fragments per K chunk (they differ in the row tile only): fetched per wave that is 13 KB per wave and chunk
// against 48 MFMAs, 35 B/clk per CU on the vector memory path.  Here every wave
// fetches UGW of the fragments, parks them in LDS (two stages, one barrier per chunk) and all four read them from
// there; only the A fragment is per wave.  ap: this lane's A address for chunk 0; bp: packed W_hh at unit group
// ug0 (+ lane * 4).
template <int UGW>
__device__ __forceinline__ void cu_kloop(f32x4 (&acc)[UGW][4], const float* ap, const float* bp, int KC,
                                         f32x4 (*bsh)[UGW * 4][64]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long gstride = (long)KC * KC * 256, ustride = (long)KC * 256;
    f32x4 an, bn[UGW];
    auto fetch = [&](int kc) {
        an = *reinterpret_cast<const f32x4*>(ap + kc * 16);
#pragma unroll
        for (int k = 0; k < UGW; ++k) {
            const int f = wave * UGW + k, u = f >> 2, g = f & 3;  // fragment f = (unit group u, gate g)
            bn[k] = *reinterpret_cast<const f32x4*>(bp + g * gstride + u * ustride + (long)kc * 256);
        }
    };
    fetch(0);
#pragma unroll
    for (int k = 0; k < UGW; ++k) bsh[0][wave * UGW + k][lane] = bn[k];
    f32x4 a = an;
    __syncthreads();
    for (int kc = 0; kc < KC; ++kc) {
        __builtin_amdgcn_sched_barrier(0);  // next chunk's global fetch first, pinned under this chunk's MFMAs
        fetch(kc + 1 < KC ? kc + 1 : kc);
        __builtin_amdgcn_sched_barrier(0);
        const int buf = kc & 1;
#pragma unroll
        for (int u = 0; u < UGW; ++u) {
            f32x4 b[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) b[g] = bsh[buf][u * 4 + g][lane];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[u][g] = mfma16(a[j], b[g][j], acc[u][g]);
        }
        // the other stage was last read in the previous iteration, which ended with the barrier below
#pragma unroll
        for (int k = 0; k < UGW; ++k) bsh[buf ^ 1][wave * UGW + k][lane] = bn[k];
        a = an;
        __syncthreads();
    }
}

// One step for a row count that fills the chip about once (2 - 9 utterances, 4 x groups of row tiles): the
// launch is shaped so that every CU gets ONE workgroup of four waves, one wave per SIMD, and every wave the same
// work - row tile w of its group x UGW hidden-unit groups x all four gates - with the operands of the next K
// chunk fetched (pinned) while the 16 UGW MFMAs of this one issue.  Against lstm_step_rows_kernel (one unit
// group per wave, 3.1 workgroups per CU at 129 tiles, loads and MFMAs of a chunk back to back): 40 -> 2x us per
// step at 128 tiles.  c_prev / c / gates_out as in lstm_step_rows_kernel (inference: c in place, no gates);
// row_tiles must be a multiple of 4.
template <int UGW>
__global__ __launch_bounds__(256) void lstm_step_cu_kernel(const float* __restrict__ gx,
                                                           const float* __restrict__ whh_p,
                                                           const float* __restrict__ h_prev,
                                                           float* __restrict__ h_out, const float* c_prev,
                                                           float* c, float* __restrict__ gates_out, long gx_rt0,
                                                           int H, int first) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int ug0 = blockIdx.x * UGW;
    const long rtile = (long)blockIdx.y * 4 + wave;
    if (first) c_prev = c;  // any valid address: the value is not used on the first step
    const int KC = H >> 4, CT = 4 * KC;
    f32x4 acc[UGW][4];
#pragma unroll
    for (int u = 0; u < UGW; ++u)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            acc[u][g] = *reinterpret_cast<const f32x4*>(gx + (((gx_rt0 + rtile) * CT + g * KC + ug0 + u) * 64 + lane) * 4);
    // the previous cell state is asked for now: these are cold lines, and read in the epilogue each one would
    // cost its own memory round trip (the value is unused on the first step; the buffer exists either way)
    float c_old[UGW][4];
#pragma unroll
    for (int u = 0; u < UGW; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) c_old[u][i] = c_prev[(rtile * 16 + 4 * lq + i) * H + (ug0 + u) * 16 + lr];
    if (!first) {  // uniform over the workgroup (barriers inside)
        __shared__ f32x4 bsh[2][UGW * 4][64];
        cu_kloop<UGW>(acc, h_prev + (rtile * 16 + lr) * H + 4 * lq, whh_p + ((long)ug0 * KC * 64 + lane) * 4, KC, bsh);
    }
#pragma unroll
    for (int u = 0; u < UGW; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long idx = (rtile * 16 + 4 * lq + i) * H + (ug0 + u) * 16 + lr;
            const float ig = sigmoid_fast(acc[u][0][i]), fg = sigmoid_fast(acc[u][1][i]);
            const float gg = tanh_fast(acc[u][2][i]), og = sigmoid_fast(acc[u][3][i]);
            const float cn = fg * (first ? 0.f : c_old[u][i]) + ig * gg;
            c[idx] = cn;
            h_out[idx] = og * tanh_fast(cn);
            if (gates_out) {  // training: the activated gates for the backward pass, [row][4H]
                float* gp = gates_out + (rtile * 16 + 4 * lq + i) * 4 * H + (ug0 + u) * 16 + lr;
                gp[0] = ig;
                gp[H] = fg;
                gp[2 * H] = gg;
                gp[3 * H] = og;
            }
        }
}

// The single-tile form also runs the left-over tiles of the sub-band model NEXT TO the resident
// persistent workgroups (12 waves x 152 registers = 456 of the 512 per SIMD lane for the layer-0
// kernel): it only gets a slot there if it needs <= 56 registers and <= 12 KB of LDS - hence this
// inference-only instance without the training outputs (40 + 16 registers; checked by
// tests/test_host_cpu.py on the code object).  With more it silently waits for the 32 ms persistent
// kernel to end (measured: +1.4 ms per batch).
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(24))) void lstm_step1_kernel(const float* __restrict__ gx,
                                                         const float* __restrict__ whh_p,
                                                         const float* __restrict__ h_prev,
                                                         float* __restrict__ h_out, float* __restrict__ c,
                                                         long gx_rt0, int H, int first) {
    // split-K partials meet pairwise (8 KB of LDS instead of 12): next to lstm_rec_x_kernel's 151.5 KB there is
    // room for exactly one such workgroup per CU, and this chain has ~10 x slack against the kernel it runs beside.
    // (Round 5: a grid of one slot per CU with the step's 96 tasks rotating through the slots, so that no CU pays every
    // step's matrix work, measured no different: 81.8 / 82.0 against 81.6 / 81.9 ms per batch.)
    __shared__ f32x4 red[2][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int ug = blockIdx.x, rtile = blockIdx.y;
    const int KC = H >> 4, CT = 4 * KC;
    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!first) {
        const int kc0 = wave * (KC >> 2), kc1 = kc0 + (KC >> 2);
        const float* ap = h_prev + ((long)rtile * 16 + lr) * H + 4 * lq;
#pragma unroll 4
        for (int kc = kc0; kc < kc1; ++kc) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(ap + kc * 16);
            f32x4 b[4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                b[g] = *reinterpret_cast<const f32x4*>(whh_p + (((long)(g * KC + ug) * KC + kc) * 64 + lane) * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = mfma16(a[j], b[g][j], acc[g]);
        }
        if (wave >= 2) {
#pragma unroll
            for (int g = 0; g < 4; ++g) red[wave - 2][g][lane] = acc[g];
        }
        __syncthreads();
        if (wave >= 2) return;
#pragma unroll
        for (int g = 0; g < 4; ++g) {  // wave 0 += wave 2, wave 1 += wave 3
            const f32x4 r = red[wave][g][lane];
            acc[g] = f32x4{acc[g][0] + r[0], acc[g][1] + r[1], acc[g][2] + r[2], acc[g][3] + r[3]};
        }
        __syncthreads();
        if (wave == 1) {
#pragma unroll
            for (int g = 0; g < 4; ++g) red[0][g][lane] = acc[g];
        }
        __syncthreads();
    }
    if (wave != 0) return;
    // everything below accumulates in place: a second live copy of the 16 partial sums is what pushes the
    // templated form over the register budget
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (!first) {  // (w0 + w2) + (w1 + w3)
            const f32x4 r = red[0][g][lane];
            acc[g] = f32x4{acc[g][0] + r[0], acc[g][1] + r[1], acc[g][2] + r[2], acc[g][3] + r[3]};
        }
        const f32x4 x = *reinterpret_cast<const f32x4*>(gx + (((gx_rt0 + rtile) * CT + g * KC + ug) * 64 + lane) * 4);
        acc[g] = f32x4{acc[g][0] + x[0], acc[g][1] + x[1], acc[g][2] + x[2], acc[g][3] + x[3]};
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long idx = ((long)rtile * 16 + 4 * lq + i) * H + ug * 16 + lr;
        const float c_old = first ? 0.f : c[idx];
        const float ig = sigmoid_fast(acc[0][i]), fg = sigmoid_fast(acc[1][i]);
        const float gg = tanh_fast(acc[2][i]), og = sigmoid_fast(acc[3][i]);
        const float cn = fg * c_old + ig * gg;
        c[idx] = cn;
        h_out[idx] = og * tanh_fast(cn);
    }
}

// ---------------------------------------------------------------------------------------------
// Two-layer wavefront step: ONE launch advances layer 0 by step i and layer 1 by step i - 1
// (blockIdx.z picks the job), so a two-layer LSTM over T frames is T + 1 dependent launches instead
// of 2 T.  Used where the recurrence is a chain of tiny latency-bound launches: the full-band model
// (N = B rows) and the sub-band model of small batches.  The layer-1 job has no precomputed input
// projection (its input row h0_t has only just been produced): it accumulates x W_ih^T and
// h W_hh^T in the same 4-way split-K pass and adds the bias in the epilogue.
struct FsnStepJob {
    const float* add;    // fragment-ordered tiles added to the accumulators: the layer-0 projection incl. bias
                         // (tile (add_rt0 + rtile) * CT + column tile), or the layer-1 bias tiles (add_rs = 0)
    const float* xw_p;   // layer-1 form: packed W_ih [4H/16][H/16][64][4]; NULL for the layer-0 form
    const float* x;      // layer-1 form: input rows [rows][H] (h of the layer below at this step)
    const float* whh_p;
    const float* h_prev;
    float* h_out;
    float* c;
    long add_rt0;
    int add_rs, first, active;
    int H;          // hidden units of this job's layer (row stride of h / c); the two jobs may differ
    int kx_chunks;  // layer-1 form: input width / 16 (= hidden units / 16 of the layer below)
    int x_ld;       // layer-1 form: row stride of x
    int row_tiles;  // 16-row tiles of this job (the grid is sized for the job with the most)
};
struct FsnStepJobs {
    FsnStepJob j[2];
};
template <class Jobs>
__device__ __forceinline__ void lstm_step2_body(const Jobs& jobs) {
    const FsnStepJob job = jobs.j[blockIdx.z];  // one uniform kernarg fetch, no per-member branching
    const int H = job.H;
    if (!job.active || (int)blockIdx.x * 16 >= H || (int)blockIdx.y >= job.row_tiles) return;  // grid: the widest / tallest job
    __shared__ f32x4 red[3][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int ug = blockIdx.x, rtile = blockIdx.y;
    const int KC = H >> 4, CT = 4 * KC;
    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the finishing wave asks for everything its epilogue needs (projection / bias tiles, previous cell
    // state) before the K loop: these are cold lines whose latency would otherwise follow the barrier
    f32x4 addv[4];
    float c_old[4];
    const long cidx = ((long)rtile * 16 + 4 * lq) * H + ug * 16 + lr;
    if (wave == 0) {
        const float* ap = job.add + (((job.add_rt0 + (long)rtile * job.add_rs) * CT + ug) * 64 + lane) * 4;
#pragma unroll
        for (int g = 0; g < 4; ++g) addv[g] = *reinterpret_cast<const f32x4*>(ap + (long)g * KC * 256);
        const float* cp = job.first ? job.add : job.c + cidx;  // first step: any valid address, value unused
        const long cs = job.first ? 0 : H;
#pragma unroll
        for (int i = 0; i < 4; ++i) c_old[i] = cp[i * cs];
    }
    const int kc0 = wave * (KC >> 2), kc1 = kc0 + (KC >> 2);
    const float* ah = job.h_prev + ((long)rtile * 16 + lr) * H + 4 * lq;
    const float* bh = job.whh_p + ((long)ug * KC * 64 + lane) * 4;
    const long gstride = (long)KC * KC * 256;  // gate g of unit group ug: column tile g KC + ug
    const int KX = job.kx_chunks;
    const long xstride = (long)KC * KX * 256;
    if (job.xw_p && !job.first && KX == KC) {
        // layer-1 job in steady state: x W_ih^T and h W_hh^T share one loop, so that the loads of both
        // products are in flight together (two back-to-back loops would pay the L2 latency twice)
        const float* ax = job.x + ((long)rtile * 16 + lr) * job.x_ld + 4 * lq;
        const float* bx = job.xw_p + ((long)ug * KC * 64 + lane) * 4;
#pragma unroll 2
        for (int kc = kc0; kc < kc1; ++kc) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(ax + kc * 16);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(ah + kc * 16);
            f32x4 b0[4], b1[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                b0[g] = *reinterpret_cast<const f32x4*>(bx + g * gstride + (long)kc * 256);
                b1[g] = *reinterpret_cast<const f32x4*>(bh + g * gstride + (long)kc * 256);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = mfma16(a0[j], b0[g][j], acc[g]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = mfma16(a1[j], b1[g][j], acc[g]);
        }
    } else {
        // the two products one after the other: layer 0 (h W_hh^T only), layer 1 at its first step
        // (x W_ih^T only), or a layer 1 whose input width differs from its own (blocks of different widths)
        if (job.xw_p) {
            const int x0 = wave * (KX >> 2), x1 = x0 + (KX >> 2);
            const float* a1p = job.x + ((long)rtile * 16 + lr) * job.x_ld + 4 * lq;
            const float* b1p = job.xw_p + ((long)ug * KX * 64 + lane) * 4;
#pragma unroll 2
            for (int kc = x0; kc < x1; ++kc) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(a1p + kc * 16);
                f32x4 b[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) b[g] = *reinterpret_cast<const f32x4*>(b1p + g * xstride + (long)kc * 256);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[g] = mfma16(a[j], b[g][j], acc[g]);
            }
        }
        if (!job.first) {
#pragma unroll 2
            for (int kc = kc0; kc < kc1; ++kc) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(ah + kc * 16);
                f32x4 b[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) b[g] = *reinterpret_cast<const f32x4*>(bh + g * gstride + (long)kc * 256);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[g] = mfma16(a[j], b[g][j], acc[g]);
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) red[wave - 1][g][lane] = acc[g];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int w = 0; w < 3; ++w) {
            const f32x4 r = red[w][g][lane];
            acc[g] = f32x4{acc[g][0] + r[0], acc[g][1] + r[1], acc[g][2] + r[2], acc[g][3] + r[3]};
        }
        acc[g] = f32x4{acc[g][0] + addv[g][0], acc[g][1] + addv[g][1], acc[g][2] + addv[g][2], acc[g][3] + addv[g][3]};
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long idx = cidx + (long)i * H;
        // hardware exp / rcp forms as in the persistent kernel: the libm ones are ~2 us of this one-wave
        // epilogue, a sixth of the whole step
        const float ig = sigmoid_fast(acc[0][i]), fg = sigmoid_fast(acc[1][i]);
        const float gg = tanh_fast(acc[2][i]), og = sigmoid_fast(acc[3][i]);
        const float cn = fg * (job.first ? 0.f : c_old[i]) + ig * gg;
        job.c[idx] = cn;
        job.h_out[idx] = og * tanh_fast(cn);
    }
}

__global__ __launch_bounds__(256) void lstm_step2_kernel(const FsnStepJobs jobs) { lstm_step2_body(jobs); }

// The same step beside the group kernel of lstm_group_kernels.hip (two 216-register workgroups per CU): capped at the 80
// registers per lane that are left there; it spills a little, on a chain that has ten times the slack.
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(40))) void lstm_step2_small_kernel(const FsnStepJobs jobs) {
    lstm_step2_body(jobs);
}

template <int H, int RT, bool XIN, int UG = 2>
int launch_rec(const float* gx, const FsnSbInput* xin, const float* whh_p, float* hseq, int Tp, int Npad,
               int main_wgs, hipStream_t s, const FsnRecFc* fc = nullptr) {
    constexpr int NW = H / (16 * UG);
    size_t lds = (size_t)RT * 16 * (H + 4) * sizeof(float);
    if (XIN) lds += (size_t)2 * RT * 16 * (16 * xin->kin_chunks + 4) * sizeof(float);
    const bool fuse = fc && fc->w_p;
    if (fuse && (XIN || RT <= 1)) {
        fsn_set_error("lstm_rec: the output layer can only be fused into the 4-pass kernel without input staging");
        return FSN_ERR_ARG;
    }
    if (fuse) lds += (size_t)2 * H * sizeof(float);
    // RT == 1 (fewer row tiles than CUs): the one-pass-all-gates variant, ~10 % faster there (8.8 vs
    // 10.0 ms per layer; a 16-row workgroup still owes 9216 MFMAs = 31 us per step, so small batches
    // stay bound by one tile per CU until the hidden units of a tile are split across CUs).  At
    // RT = 2 it spills and loses.
    void (*kern)(const float*, const FsnSbInput, const float*, float*, int, int, const FsnRecFc) =
        lstm_rec_kernel<H, RT, UG, XIN>;
    if constexpr (RT <= 1) kern = lstm_rec_small_kernel<H, RT, UG, XIN>;
    if (lds > 160 * 1024 ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) {
        fsn_set_error("lstm_rec: cannot reserve %zu bytes of LDS", lds);
        return FSN_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)main_wgs), dim3(NW * 64), lds, s, gx, XIN ? *xin : FsnSbInput{}, whh_p,
                       hseq, Tp, Npad, fuse ? *fc : FsnRecFc{});
    return fsn_check_launch("lstm_rec_kernel");
}

template <int H, int RT, bool HSEQ, int CELL = 0, int UG = 2>
int launch_rec_x(const float* xseq, const float* wih_p, const float* whh_p, const float* bias, int Tp, int Npad,
                 int main_wgs, hipStream_t s, const FsnRecFc* fc, float* hseq_out) {
    constexpr int NW = H / (16 * UG);
    const size_t lds = ((size_t)RT * 16 * (H + 4) + 2 * H + (size_t)2 * RT * 6 * 256) * sizeof(float);
    auto kern = lstm_rec_x_kernel<H, RT, UG, FSN_REC_X_OPT | (CELL ? FSN_REC_GRU : 0), HSEQ>;
    if (lds > 160 * 1024 ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) {
        fsn_set_error("lstm_rec_x: cannot reserve %zu bytes of LDS", lds);
        return FSN_ERR_LAUNCH;
    }
    if (whh_p < wih_p || whh_p - wih_p > 0x3fffffffL) {
        fsn_set_error("lstm_rec_x: W_hh must follow W_ih in one packed buffer");
        return FSN_ERR_ARG;
    }
    FsnRecFc a{};
    if (fc) a = *fc;
    if (HSEQ) a.crm_r = hseq_out;
    hipLaunchKernelGGL(kern, dim3((unsigned)main_wgs), dim3(NW * 64), lds, s, xseq, wih_p, (unsigned)(whh_p - wih_p), bias,
                       Tp, Npad, a);
    return fsn_check_launch("lstm_rec_x_kernel");
}

template <int H, int RT, int KX = 2, bool ROWSIN = false, int CELL = 0, int UG = 2>
int launch_rec_in(const FsnSbInput* xin, const float* whh_p, float* hseq, int Tp, int Npad, int main_wgs, hipStream_t s) {
    constexpr int NW = H / (16 * UG);
    const size_t lds = ((size_t)RT * 16 * (H + 4) + (size_t)2 * RT * 16 * (16 * KX + 4)) * sizeof(float);
    auto kern = lstm_rec_in_kernel<H, RT, UG, FSN_REC_IN_OPT | (CELL ? FSN_REC_GRU : 0), KX, ROWSIN>;
    if (lds > 160 * 1024 ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) {
        fsn_set_error("lstm_rec_in: cannot reserve %zu bytes of LDS", lds);
        return FSN_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)main_wgs), dim3(NW * 64), lds, s, *xin, xin->wih_p,
                       (unsigned)(whh_p - xin->wih_p), hseq, Tp, Npad);
    return fsn_check_launch("lstm_rec_in_kernel");
}

}  // namespace

// First layer of a stack on lstm_rec_in_kernel: the 32-column gathered sub-band input (two K chunks) or a plain row-major
// input of one or two chunks, H = 384, 2 - 4 row tiles per workgroup, W_hh packed right behind W_ih.  Anything else stays
// on lstm_rec_kernel<.., XIN = true>.
bool fsn_lstm_rec_in_supported(const FsnSbInput* xin, const float* whh_p, int H, int RT) {
    const bool shape = xin && (xin->x_rows ? (xin->kin_chunks == 1 || xin->kin_chunks == 2) && xin->x_ld >= 16 * xin->kin_chunks
                                           : xin->kin_chunks == 2);
    return shape && H == 384 && RT >= 2 && RT <= 4 && whh_p > xin->wih_p && whh_p - xin->wih_p < 0x3fffffffL;
}

int fsn_launch_lstm_rec_in(const FsnSbInput* xin, const float* whh_p, float* hseq, int Tp, int Npad, int H, int RT,
                           int main_wgs, hipStream_t s, int cell) {
    if (!fsn_lstm_rec_in_supported(xin, whh_p, H, RT) || (cell && !xin->x_rows)) {
        fsn_set_error("lstm_rec_in: unsupported configuration");
        return FSN_ERR_ARG;
    }
    if (cell) {  // GRU (FSN_REC_GRU): the row-major input forms, weights expanded by fsn_launch_gru_expand4(.., order 1)
#define FSN_REC_IN_GRU(R)                                                                                             \
    if (RT == R)                                                                                                      \
        return xin->kin_chunks == 2 ? launch_rec_in<384, R, 2, true, 1>(xin, whh_p, hseq, Tp, Npad, main_wgs, s)      \
                                    : launch_rec_in<384, R, 1, true, 1>(xin, whh_p, hseq, Tp, Npad, main_wgs, s);
        FSN_REC_IN_GRU(2)
        FSN_REC_IN_GRU(3)
        FSN_REC_IN_GRU(4)
#undef FSN_REC_IN_GRU
    }
#define FSN_REC_IN_CASE(R)                                                                                         \
    if (RT == R) {                                                                                                 \
        if (!xin->x_rows) return launch_rec_in<384, R>(xin, whh_p, hseq, Tp, Npad, main_wgs, s);                   \
        if (xin->kin_chunks == 2) return launch_rec_in<384, R, 2, true>(xin, whh_p, hseq, Tp, Npad, main_wgs, s);  \
        return launch_rec_in<384, R, 1, true>(xin, whh_p, hseq, Tp, Npad, main_wgs, s);                            \
    }
    FSN_REC_IN_CASE(2)
    FSN_REC_IN_CASE(3)
    FSN_REC_IN_CASE(4)
#undef FSN_REC_IN_CASE
    fsn_set_error("lstm_rec_in: unsupported row tiles %d", RT);
    return FSN_ERR_ARG;
}

// The last sub-band layer with its input projection inside (lstm_rec_x_kernel): built for H = 384 and 2 - 4 row
// tiles per workgroup (at 5 the x ring no longer fits beside the hidden state; the caller then keeps the
// projection GEMM + lstm_rec_kernel pair).
bool fsn_lstm_rec_x_supported(int H, int RT) { return H == 384 && RT >= 2 && RT <= 4; }

int fsn_launch_lstm_rec_x(const float* xseq, const float* wih_p, const float* whh_p, const float* bias, int Tp, int Npad,
                          int H, int RT, int main_wgs, hipStream_t s, const FsnRecFc* fc, float* hseq_out, int cell) {
    if (((!fc || !fc->w_p) && !hseq_out) || !fsn_lstm_rec_x_supported(H, RT)) {
        fsn_set_error("lstm_rec_x: needs the fused output layer or a hidden-sequence buffer, H = 384 and 2 - 4 row tiles "
                      "(got H %d, RT %d)", H, RT);
        return FSN_ERR_ARG;
    }
    if (cell) {  // GRU (FSN_REC_GRU)
        if (hseq_out) {
            if (RT == 2) return launch_rec_x<384, 2, true, 1>(xseq, wih_p, whh_p, bias, Tp, Npad, main_wgs, s, nullptr, hseq_out);
            if (RT == 3) return launch_rec_x<384, 3, true, 1>(xseq, wih_p, whh_p, bias, Tp, Npad, main_wgs, s, nullptr, hseq_out);
            return launch_rec_x<384, 4, true, 1>(xseq, wih_p, whh_p, bias, Tp, Npad, main_wgs, s, nullptr, hseq_out);
        }
        fsn_set_error("lstm_rec_x: the GRU cell is instantiated for the hidden-sequence form (its output layer is a separate launch)");
        return FSN_ERR_ARG;
    }
    if (hseq_out) {  // a layer inside a stack: h_t stored, no output layer
        if (RT == 2) return launch_rec_x<384, 2, true>(xseq, wih_p, whh_p, bias, Tp, Npad, main_wgs, s, nullptr, hseq_out);
        if (RT == 3) return launch_rec_x<384, 3, true>(xseq, wih_p, whh_p, bias, Tp, Npad, main_wgs, s, nullptr, hseq_out);
        return launch_rec_x<384, 4, true>(xseq, wih_p, whh_p, bias, Tp, Npad, main_wgs, s, nullptr, hseq_out);
    }
    if (RT == 2) return launch_rec_x<384, 2, false>(xseq, wih_p, whh_p, bias, Tp, Npad, main_wgs, s, fc, nullptr);
    if (RT == 3) return launch_rec_x<384, 3, false>(xseq, wih_p, whh_p, bias, Tp, Npad, main_wgs, s, fc, nullptr);
    return launch_rec_x<384, 4, false>(xseq, wih_p, whh_p, bias, Tp, Npad, main_wgs, s, fc, nullptr);
}

// How the N sub-band sequences are laid out on the chip.  One workgroup per CU (LDS-bound), RT
// 16-row tiles per workgroup, so a single launch is worth max-RT tile-times and
// N = B F = 64 * 257 = 1028 tiles is the worst case for 256 CUs: 4.016 tiles per CU.  Instead of
// paying a fifth tile on every CU (206 workgroups x 5 tiles, 50 CUs idle), the main kernel takes
// floor(tiles / CUs) tiles per CU on all CUs and the few left-over tiles (4 of 1028) run
// concurrently as per-step lstm_step_kernel launches on an auxiliary stream: those small workgroups
// fit next to the resident main workgroup (40 VGPRs, 12 KB LDS) and add 0.4 % of MFMA work.
FsnRecPlan fsn_lstm_rec_plan(int N, int H) {
    (void)H;
    const int rt_max = 5;    // LDS: 16 RT (H + 4) floats = 124 KB at H = 384
    const int left_max = 16;  // tiles worth handing to the step kernels
    int cus = 256, dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus < 1) cus = 256;
    FsnRecPlan p;
    p.tiles = (N + 15) / 16;
    p.npad = p.tiles * 16;
    // Few rows: one 16-row tile per CU leaves most CUs idle and makes every busy one stream the whole
    // W_hh (2.4 MB) from L2 per step (~31 us/step, measured); below this many tiles the per-step
    // kernels, which spread a step over (H/16) x tiles workgroups, are faster (batch 1: 21.4 -> 8.4 ms
    // for 3 s of audio; break-even at ~160 tiles = batch 10).
    constexpr int step_below = 160;
    if (p.tiles < step_below) {
        p.rt = 1;
        p.main_wgs = 0;
        p.left_tiles = p.tiles;
        return p;
    }
    if (p.tiles <= cus) {
        p.rt = 1;
        p.main_wgs = p.tiles;
        p.left_tiles = 0;
        return p;
    }
    const int rt_floor = p.tiles / cus < rt_max ? p.tiles / cus : rt_max;
    const int left = p.tiles - cus * rt_floor;
    if (left <= left_max) {
        p.rt = rt_floor;
        p.main_wgs = cus;
        p.left_tiles = left;
        return p;
    }
    // general case: whole rounds, pick the RT with the smallest makespan (rounds x RT) ...
    int best = 1;
    long best_cost = -1;
    for (int rt = 1; rt <= rt_max; ++rt) {
        const long wgs = (p.tiles + rt - 1) / rt;
        const long rounds = (wgs + cus - 1) / cus;
        const long cost = rounds * rt * 64 + rounds;
        if (best_cost < 0 || cost < best_cost || (cost == best_cost && rt > best)) {
            best = rt;
            best_cost = cost;
        }
    }
    // ... unless FULL rounds of some RT leave only a few tiles for the step kernels: 128 utterances are 2056 tiles = two
    // full rounds at 4 tiles per workgroup + 8 left over, where whole rounds would take three of RT = 3 (188 ms against
    // 2 x 84); 96 utterances two rounds of 3 + 6 tiles instead of three
    for (int rt = rt_max; rt >= 1; --rt) {
        const long rounds = p.tiles / ((long)cus * rt);
        if (rounds < 1) continue;
        const long left2 = p.tiles - rounds * cus * rt;
        const long cost = rounds * rt * 64 + rounds;
        if (left2 <= left_max && cost < best_cost) {
            p.rt = rt;
            p.main_wgs = (int)(rounds * cus);
            p.left_tiles = (int)left2;
            return p;
        }
    }
    p.rt = best;
    p.main_wgs = (p.tiles + best - 1) / best;
    p.left_tiles = 0;
    p.npad = p.main_wgs * best * 16;
    p.tiles = p.npad / 16;
    return p;
}

bool fsn_lstm_rec_can_fuse_fc(int RT, bool xin) { return !xin && RT >= 2; }

int fsn_launch_lstm_rec(const float* gx, const FsnSbInput* xin, const float* whh_p, float* hseq, int Tp, int Npad,
                        int H, int RT, int main_wgs, hipStream_t s, const FsnRecFc* fc) {
#define FSN_REC_CASE(HH, R)                                                                              \
    if (H == HH && RT == R)                                                                              \
        return xin ? launch_rec<HH, R, true>(gx, xin, whh_p, hseq, Tp, Npad, main_wgs, s, fc)            \
                   : launch_rec<HH, R, false>(gx, xin, whh_p, hseq, Tp, Npad, main_wgs, s, fc);
    FSN_REC_CASE(384, 1)
    FSN_REC_CASE(384, 2)
    FSN_REC_CASE(384, 3)
    FSN_REC_CASE(384, 4)
    FSN_REC_CASE(384, 5)
#undef FSN_REC_CASE
    fsn_set_error("lstm_rec: unsupported hidden size %d / row tiles %d (built for H = 384)", H, RT);
    return FSN_ERR_ARG;
}

// One step for `row_tiles` 16-row tiles: gx tiles gx_rt0 .. gx_rt0 + row_tiles - 1 of the fragment-
// ordered projection, h_prev / h_out / c point at the first of those rows.
int fsn_launch_lstm_step(const float* gx, const float* whh_p, const float* h_prev, float* h_out, float* c,
                         long gx_rt0, int row_tiles, int H, int first, hipStream_t s, int beside_persistent) {
    if (beside_persistent) {  // must fit next to a resident persistent workgroup, see lstm_step1_kernel
        if (H % 64 != 0) {
            fsn_set_error("lstm_step: hidden size %d must be a multiple of 64", H);
            return FSN_ERR_ARG;
        }
        hipLaunchKernelGGL(lstm_step1_kernel, dim3(H / 16, row_tiles), dim3(256), 0, s, gx, whh_p, h_prev, h_out, c,
                           gx_rt0, H, first);
        return fsn_check_launch("lstm_step1_kernel");
    }
    return fsn_launch_lstm_step_train(gx, whh_p, h_prev, h_out, c, c, nullptr, gx_rt0, row_tiles, H, first, s);
}

// One step on row_tiles (a multiple of 4) tiles with the one-workgroup-per-CU kernel; picks the unit groups per
// wave that fill the chip best: cost = rounds of workgroups over the CUs x work per workgroup.
int fsn_launch_lstm_step_cu(const float* gx, const float* whh_p, const float* h_prev, float* h_out, float* c,
                            long gx_rt0, int row_tiles, int H, int first, hipStream_t s, const float* c_prev,
                            float* gates_out) {
    if (!c_prev) c_prev = c;
    if (H % 16 != 0 || row_tiles % 4 != 0 || row_tiles <= 0) {
        fsn_set_error("lstm_step_cu: H %d must be a multiple of 16 and row_tiles %d a positive multiple of 4", H, row_tiles);
        return FSN_ERR_ARG;
    }
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int ugs = H / 16, groups = row_tiles / 4;
    int best = 1;
    long best_cost = -1;
    for (int ugw = 1; ugw <= 3; ++ugw) {
        if (ugs % ugw) continue;
        const long wgs = (long)groups * (ugs / ugw);
        const long cost = ((wgs + cus - 1) / cus) * ugw;
        if (best_cost < 0 || cost <= best_cost) {  // ties: the wider wave tile (fewer operand loads per MFMA)
            best = ugw;
            best_cost = cost;
        }
    }
#define FSN_STEP_CU(U)                                                                                             \
    hipLaunchKernelGGL(lstm_step_cu_kernel<U>, dim3(ugs / U, groups), dim3(256), 0, s, gx, whh_p, h_prev, h_out,    \
                       c_prev, c, gates_out, gx_rt0, H, first)
    if (best == 3) FSN_STEP_CU(3);
    else if (best == 2) FSN_STEP_CU(2);
    else FSN_STEP_CU(1);
#undef FSN_STEP_CU
    return fsn_check_launch("lstm_step_cu_kernel");
}

// Training form: c_{t-1} is read from c_prev, c_t written to c_out (the saved cell sequence) and the
// activated gates i, f, g, o to gates_out [rows][4H].
int fsn_launch_lstm_step_train(const float* gx, const float* whh_p, const float* h_prev, float* h_out,
                               const float* c_prev, float* c_out, float* gates_out, long gx_rt0, int row_tiles, int H,
                               int first, hipStream_t s) {
    if (H % 64 != 0) {
        fsn_set_error("lstm_step: hidden size %d must be a multiple of 64", H);
        return FSN_ERR_ARG;
    }
    if (row_tiles >= 64 && row_tiles % 4 == 0)  // config 3: 16 x 128 bins = 128 tiles = one workgroup per CU
        return fsn_launch_lstm_step_cu(gx, whh_p, h_prev, h_out, c_out, gx_rt0, row_tiles, H, first, s, c_prev, gates_out);
    if (row_tiles >= 64) {  // measured at 129 tiles: 65.1 ms per training step against 66.3 for the split-K form
        hipLaunchKernelGGL(lstm_step_rows_kernel<1>, dim3(H / 16, (row_tiles + 3) / 4), dim3(256), 0, s, gx, whh_p, h_prev,
                           h_out, c_prev, c_out, gates_out, gx_rt0, row_tiles, H, first);
        return fsn_check_launch("lstm_step_rows_kernel");
    }
#ifndef FSN_STEP_RTS2_FROM
#define FSN_STEP_RTS2_FROM 16
#endif
    const int rts = row_tiles >= FSN_STEP_RTS2_FROM ? 2 : 1;  // measured: 2 is the best at 129 tiles, 4 no better
#define FSN_STEP_CASE(R)                                                                                         \
    hipLaunchKernelGGL(lstm_step_kernel<R>, dim3(H / 16, (row_tiles + R - 1) / R), dim3(256), 0, s, gx, whh_p, h_prev, \
                       h_out, c_prev, c_out, gates_out, gx_rt0, row_tiles, H, first)
    if (rts == 2) FSN_STEP_CASE(2);
    else
        FSN_STEP_CASE(1);
#undef FSN_STEP_CASE
    return fsn_check_launch("lstm_step_kernel");
}

// Two LSTM layers over T steps on `row_tiles` 16-row tiles, advanced in a wavefront (see lstm_step2_kernel).
// Layer 0 has H0 units, layer 1 H1 units and H0 inputs.  gx0: layer-0 projection, tile (t, i) at
// t * gx_stride + gx_off + i; wih1_p: layer-1 input weights [4 H1 / 16][H0 / 16][64][4]; bias1_frag: b_ih + b_hh
// of layer 1 as fragment tiles; hseq0 / hseq1: [T][hs_stride rows][H0 / H1] with this launch's rows starting at
// row hs_off; c0 / c1: [row_tiles * 16][H0 / H1].  state_h0 / state_h1: streaming continuation (see header).
int fsn_launch_lstm_wavefront2w(const float* gx0, long gx_stride, long gx_off, const float* whh0_p,
                                const float* wih1_p, const float* bias1_frag, const float* whh1_p, float* hseq0,
                                float* hseq1, long hs_stride, long hs_off, float* c0, float* c1, int T, int row_tiles,
                                int H0, int H1, hipStream_t s, float* state_h0, float* state_h1, int beside_group) {
    if (H0 % 64 != 0 || H1 % 64 != 0 || (state_h0 == nullptr) != (state_h1 == nullptr)) {
        fsn_set_error("lstm_wavefront2: hidden sizes %d / %d must be multiples of 64 (and both states or none)", H0, H1);
        return FSN_ERR_ARG;
    }
    const bool cont = state_h0 != nullptr;
    const size_t step0 = (size_t)hs_stride * H0, step1 = (size_t)hs_stride * H1;
    float* h0 = hseq0 + (size_t)hs_off * H0;
    float* h1 = hseq1 + (size_t)hs_off * H1;
    const int Hmax = H0 > H1 ? H0 : H1;
    for (int i = 0; i <= T; ++i) {
        FsnStepJobs jobs{};
        FsnStepJob& a = jobs.j[0];
        FsnStepJob& b = jobs.j[1];
        a.H = H0;
        b.H = H1;
        a.row_tiles = b.row_tiles = row_tiles;
        if (i < T) {
            a.active = 1;
            a.add = gx0;
            a.add_rt0 = (long)i * gx_stride + gx_off;
            a.add_rs = 1;
            a.whh_p = whh0_p;
            a.h_prev = i ? h0 + (i - 1) * step0 : (cont ? state_h0 : h0);
            a.h_out = h0 + i * step0;
            a.c = c0;
            a.first = i == 0 && !cont;
        }
        if (i >= 1) {
            const int t = i - 1;
            b.active = 1;
            b.add = bias1_frag;
            b.add_rt0 = 0;
            b.add_rs = 0;
            b.xw_p = wih1_p;
            b.x = h0 + t * step0;
            b.x_ld = H0;
            b.kx_chunks = H0 / 16;
            b.whh_p = whh1_p;
            b.h_prev = t ? h1 + (t - 1) * step1 : (cont ? state_h1 : h1);
            b.h_out = h1 + t * step1;
            b.c = c1;
            b.first = t == 0 && !cont;
        }
        // (a 16-wave split of the K range - every wave's operands in one round trip - was tried for the full-band
        // model and lost: 13.3 us per step against 10.5; dispatching and joining 16 waves costs more than it saves)
        if (beside_group) hipLaunchKernelGGL(lstm_step2_small_kernel, dim3(Hmax / 16, row_tiles, 2), dim3(256), 0, s, jobs);
        else hipLaunchKernelGGL(lstm_step2_kernel, dim3(Hmax / 16, row_tiles, 2), dim3(256), 0, s, jobs);
        FSN_TRY_LAUNCH("lstm_step2_kernel");
    }
    if (cont) {
        const size_t rows = (size_t)row_tiles * 16;
        if (hipMemcpyAsync(state_h0, h0 + (size_t)(T - 1) * step0, rows * H0 * sizeof(float), hipMemcpyDeviceToDevice,
                           s) != hipSuccess ||
            hipMemcpyAsync(state_h1, h1 + (size_t)(T - 1) * step1, rows * H1 * sizeof(float), hipMemcpyDeviceToDevice,
                           s) != hipSuccess) {
            fsn_set_error("lstm_wavefront2: state copy failed");
            return FSN_ERR_LAUNCH;
        }
    }
    return FSN_OK;
}

int fsn_launch_lstm_wavefront2(const float* gx0, long gx_stride, long gx_off, const float* whh0_p, const float* wih1_p,
                               const float* bias1_frag, const float* whh1_p, float* hseq0, float* hseq1, long hs_stride,
                               long hs_off, float* c0, float* c1, int T, int row_tiles, int H, hipStream_t s,
                               float* state_h0, float* state_h1, int beside_group) {
    return fsn_launch_lstm_wavefront2w(gx0, gx_stride, gx_off, whh0_p, wih1_p, bias1_frag, whh1_p, hseq0, hseq1,
                                       hs_stride, hs_off, c0, c1, T, row_tiles, H, H, s, state_h0, state_h1, beside_group);
}
