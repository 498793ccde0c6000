// Shared device/host helpers for libfsn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fsn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define FSN_WAVE 64

static inline int fsn_round_up(int x, int m) { return (x + m - 1) / m * m; }
static inline size_t fsn_round_up_sz(size_t x, size_t m) { return (x + m - 1) / m * m; }

// frequency axis padded to a multiple of 16 floats: rows stay 16-byte aligned for float4 operand
// loads and the zero padding doubles as the K padding of the full-band input projection.
static inline int fsn_fpad(int F) { return fsn_round_up(F, 16); }

void fsn_set_error(const char* fmt, ...);

// Opened by every entry point that enqueues work (fsn_api.hip): makes the device of the caller's stream the
// current one for the duration of the call and selects the per-(device, stream) record of the library.
struct FsnCallScope {
    int prev;
    bool switched;
    explicit FsnCallScope(void* stream);
    ~FsnCallScope();
};
int fsn_check_launch(const char* what);

#define FSN_TRY_LAUNCH(what)                        \
    do {                                            \
        const int _rc = fsn_check_launch(what);     \
        if (_rc != FSN_OK) return _rc;              \
    } while (0)

#define FSN_REQUIRE(cond, ...)          \
    do {                                \
        if (!(cond)) {                  \
            fsn_set_error(__VA_ARGS__); \
            return FSN_ERR_ARG;         \
        }                               \
    } while (0)

// D = A(16x4) * B(4x16) + C, exact fp32 (k-ordered fmaf chain).  Lane l supplies A[l&15][l>>4]
// and B[l>>4][l&15]; D/C: col = l&15, row = 4*(l>>4) + reg.
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// One K = 16 block of a product, D = A(16 x 16) B(16 x 16) + C, from the fp32 fragments every kernel of this library
// moves (lane (i, q) holds A[i][4q .. 4q+3] / B[4q .. 4q+3][i]), in the arithmetic AR of the call:
//   FSN_ARITH_F32   four v_mfma_f32_16x16x4_f32: exact fp32 products, the arithmetic of every parity claim;
//   FSN_ARITH_F16   both operands rounded to fp16 (round to nearest even) AT THE MATRIX CORE'S INPUT, one
//   FSN_ARITH_BF16  v_mfma_f32_16x16x16_{f16,bf16}, fp32 accumulation - what torch.autocast does to nn.LSTM / nn.Linear
//                   (recipes/dns_interspeech_2020/fullsubnet/trainer.py:56), an eighth of the matrix-core time.
// Everything around the product (storage, cell update, reductions) is fp32 in every mode.
typedef _Float16 fsn_f16x4 __attribute__((ext_vector_type(4)));
typedef short fsn_s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned fsn_u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 fsn_bf16x2 __attribute__((ext_vector_type(2)));
template <int AR>
struct FsnOperand {
    typedef f32x4 type;
};
template <>
struct FsnOperand<FSN_ARITH_F16> {
    typedef fsn_f16x4 type;
};
template <>
struct FsnOperand<FSN_ARITH_BF16> {
    typedef fsn_s16x4 type;
};
// A weight fragment as it travels L2 -> registers -> LDS under arithmetic AR: fp32 (16 bytes per lane), or - 16-bit
// arithmetic - the 16-bit copy of the packed weights (8 bytes per lane: half the L2 traffic and half the LDS of a K
// loop whose matrix work is an eighth); fragment ORDER is the same, so element offsets carry over.
template <int AR>
struct FsnWFrag {
    typedef f32x4 type;
};
template <>
struct FsnWFrag<FSN_ARITH_F16> {
    typedef fsn_u32x2 type;
};
template <>
struct FsnWFrag<FSN_ARITH_BF16> {
    typedef fsn_u32x2 type;
};
// the lane's fragment at element offset `ofs` of the packed weight buffer behind resource r
template <int AR>
__device__ __forceinline__ typename FsnWFrag<AR>::type fsn_load_wfrag(const __amdgpu_buffer_rsrc_t r, unsigned lane,
                                                                     unsigned ofs) {
    if constexpr (AR == FSN_ARITH_F32)
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16u, ofs * 4u, 0));
    else
        return __builtin_bit_cast(fsn_u32x2, __builtin_amdgcn_raw_buffer_load_b64(r, lane * 8u, ofs * 2u, 0));
}
template <int AR>
__device__ __forceinline__ typename FsnOperand<AR>::type fsn_wfrag_operand(const typename FsnWFrag<AR>::type w) {
    if constexpr (AR == FSN_ARITH_F32) return w;
    else return __builtin_bit_cast(typename FsnOperand<AR>::type, w);
}
template <int AR>
__device__ __forceinline__ typename FsnOperand<AR>::type fsn_operand(const f32x4 v) {
    if constexpr (AR == FSN_ARITH_F16) {
        return __builtin_convertvector(v, fsn_f16x4);  // v_cvt_pk_f16_f32 x 2
    } else if constexpr (AR == FSN_ARITH_BF16) {
        // pairs through __builtin_convertvector: one v_cvt_pk_bf16_f32 each (round to nearest even), and - unlike the
        // same instruction from inline asm, which produced NaNs here - the compiler knows it is a VALU write feeding a
        // matrix instruction and inserts the wait state that hazard needs
        const fsn_bf16x2 lo = __builtin_convertvector(f32x2{v[0], v[1]}, fsn_bf16x2);
        const fsn_bf16x2 hi = __builtin_convertvector(f32x2{v[2], v[3]}, fsn_bf16x2);
        const fsn_u32x2 r = {__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
        return __builtin_bit_cast(fsn_s16x4, r);
    } else {
        return v;
    }
}
template <int AR>
__device__ __forceinline__ f32x4 fsn_mma_k16(const typename FsnOperand<AR>::type a, const typename FsnOperand<AR>::type b,
                                             f32x4 c) {
    if constexpr (AR == FSN_ARITH_F16) {
        return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
    } else if constexpr (AR == FSN_ARITH_BF16) {
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) c = mfma16(a[j], b[j], c);
        return c;
    }
}

// Two K = 16 blocks of the same product at once, as ONE K = 32 instruction of gfx950 (v_mfma_f32_16x16x32_{f16,bf16}: the
// K = 16 shapes issue at half its rate there).  The matrix instruction sums a_lane(i, q)[j] b_lane(n, q)[j] over the four
// lane groups q and the lane's elements j: as long as A and B give the same k to the same (q, j), ANY assignment of k to
// (q, j) is the same contraction - so the eight values of a lane are simply block 0's four followed by block 1's, and the
// result is fsn_mma_k16(a0, b0, fsn_mma_k16(a1, b1, c)) up to the order of the fp32 sums inside the instruction.
typedef _Float16 fsn_f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 fsn_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned fsn_u32x4 __attribute__((ext_vector_type(4)));
template <int AR>
__device__ __forceinline__ f32x4 fsn_mma_k32(const typename FsnOperand<AR>::type a0, const typename FsnOperand<AR>::type a1,
                                             const typename FsnOperand<AR>::type b0, const typename FsnOperand<AR>::type b1,
                                             f32x4 c) {
    if constexpr (AR == FSN_ARITH_F16) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7),
                                                      __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7), c, 0, 0, 0);
    } else if constexpr (AR == FSN_ARITH_BF16) {
        const fsn_u32x2 pa0 = __builtin_bit_cast(fsn_u32x2, a0), pa1 = __builtin_bit_cast(fsn_u32x2, a1);
        const fsn_u32x2 pb0 = __builtin_bit_cast(fsn_u32x2, b0), pb1 = __builtin_bit_cast(fsn_u32x2, b1);
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(
            __builtin_bit_cast(fsn_bf16x8, fsn_u32x4{pa0[0], pa0[1], pa1[0], pa1[1]}),
            __builtin_bit_cast(fsn_bf16x8, fsn_u32x4{pb0[0], pb0[1], pb1[0], pb1[1]}), c, 0, 0, 0);
    } else {
        return fsn_mma_k16<AR>(a1, b1, fsn_mma_k16<AR>(a0, b0, c));
    }
}

// gfx950 store-data hazard (measured: tools/probe_store_hazard.hip, profiles/r06_store_hazard.md).  A 12 / 16-byte-per-lane
// store reads its data registers lane quad by lane quad over the cycles after issue; a vector instruction that writes them
// again 1 - 2 issue slots later (global stores, buffer stores with an immediate soffset) or 1 slot later (buffer stores with
// an SGPR soffset) reaches memory in lanes 8 - 15 / 12 - 15 of every 16.  hipcc (ROCm 7.2) keeps two wait states in the
// first case and none in the second (its hazard recogniser exempts a register soffset): lstm2_g16_bwd_kernel's layer-0 gate
// gradients were 6e-2 off in round 5 behind `buffer_store_dwordx4 v[38:41], .., s0 offen ; v_pk_add_f32 v[38:39], ..`.
// Call this right behind such a store when the value is re-used as an accumulator: the data registers stay allocated and
// untouched for two more issue slots (the distance that was never wrong is 3).  tests/test_host_cpu.py scans EVERY kernel
// of the shipped library for vector writes inside the unsafe distance (tools/check_store_hazard.py).
// FSN_HOLD_MODE (diagnosis builds only, tools/build_variant.py): 1 = nothing (the failing build), 2 = kept allocated, no wait.
#ifndef FSN_HOLD_MODE
#define FSN_HOLD_MODE 0
#endif
__device__ __forceinline__ void fsn_hold_store_data(const f32x4& v) {
#if FSN_HOLD_MODE == 0
    asm volatile("s_nop 1" ::"v"(v) : "memory");
#elif FSN_HOLD_MODE == 2
    asm volatile("" ::"v"(v) : "memory");
#endif
}

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// Gate non-linearities of every LSTM forward kernel (persistent, per-step, wavefront) on the hardware transcendentals
// (v_exp_f32 / v_rcp_f32, ~1 ULP each): 4-5 VALU instructions per value instead of ~30 for the
// libm-accurate forms.  Absolute error <= ~2e-7, the size of one fp32 rounding of the gate
// pre-activation itself; the end-to-end effect on the compressed mask is measured by the parity
// tests (tests/test_gpu_parity.py).  -DFSN_ACCURATE_ACT=1 switches back to expf / tanhf.
#ifndef FSN_ACCURATE_ACT
#define FSN_ACCURATE_ACT 0
#endif
__device__ __forceinline__ float sigmoid_fast(float x) {
#if FSN_ACCURATE_ACT
    return sigmoid_f(x);
#else
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
#endif
}
__device__ __forceinline__ float tanh_fast(float x) {
#if FSN_ACCURATE_ACT
    return tanhf(x);
#else
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
#endif
}

// Bijective XCD-aware block remap: block b runs on XCD b % 8 (observed, speed only); give each
// XCD a contiguous chunk of the virtual grid so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
    const unsigned q = nblk >> 3, r = nblk & 7u;
    const unsigned xcd = bid & 7u, slot = bid >> 3;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

// ---- kernels' host launchers (one per translation unit) -------------------------------------
struct FsnStream {
    hipStream_t s;
};

// fft_kernels.hip
int fsn_launch_stft(const float* y, int B, int L, const float* window, float* re, float* im, float* mag,
                    int T, int Tp, int F, int FP, bool frame_major, hipStream_t s);
int fsn_launch_mask_irfft(const float* re, const float* im, const float* crm_r, const float* crm_i,
                          int B, int T, int F, int FP, bool frame_major, const float* window,
                          float* wframes, hipStream_t s);
int fsn_launch_ola(const float* wframes, const float* window, int B, int T, int length, float* y,
                   hipStream_t s);

// dft_kernels.hip (any even n_fft / any hop: direct fp64 DFT; reference layout [B][F][T] only)
int fsn_launch_dft_stft(const float* y, int B, int L, const float* window, float* re, float* im, float* mag, int T,
                        int N, int hop, hipStream_t s);
int fsn_launch_dft_istft(const float* re, const float* im, const float* window, float* wframes, float* y, int B, int T,
                         int N, int hop, int length, hipStream_t s);

// elementwise_kernels.hip
int fsn_launch_decompress(const float* in, float* out, size_t n, hipStream_t s);
int fsn_launch_compress(const float* in, float* out, size_t n, hipStream_t s);
int fsn_launch_build_cirm(const float* nr, const float* ni, const float* cr, const float* ci, float* out,
                          size_t n, hipStream_t s);
int fsn_launch_transpose(const float* in, float* out, int batch, int R, int C, long ld_in, long bs_in,
                         long ld_out, long bs_out, int R_valid, int C_valid, hipStream_t s);
int fsn_launch_crm_rows(const float* crm_r, const float* crm_i, float* out, long r0, long n, int F, int FP, int T,
                        hipStream_t s);
int fsn_launch_binsum(const float* mag, double* binsum, int B, int Tp, int FP, hipStream_t s);
int fsn_launch_offline_den(const double* binsum, const float* fb_out, float* den_fb, float* den_sb, int B,
                           int Tp, int F, int FP, int nb, int which, hipStream_t s);
// carry / t0: streaming continuation (running sums of the t0 frames already seen, updated in place); NULL / 0 offline
int fsn_launch_cumulative_den_fb(const float* mag, float* den, int B, int Tp, int F, int FP, hipStream_t s,
                                 double* carry = nullptr, int t0 = 0);
int fsn_launch_cumulative_den_sb(const float* mag, const float* fb_out, float* den, int B, int Tp, int F,
                                 int FP, int nb, int Npad, hipStream_t s, double* carry = nullptr, int t0 = 0);

// gemm_kernels.hip
struct FsnGemmA {  // A operand description
    int kind;      // 0 row-major, 1 full-band input, 2 sub-band input
    const float* p0;  // row-major: matrix; fb/sb: mag [B][Tp][FP]
    const float* p1;  // sb: fb_out [B][Tp][FP]
    const float* den;  // fb/sb: divisor
    int den_mode;      // 0: den[b]   1: den[b*Tp + t] (fb) / den[t*den_stride + n] (sb)
    long ld;           // row-major leading dimension
    int B, Tp, F, FP, Npad, N, nb;
    int n_offset;      // sb: first unit of this launch (rows are n_offset .. n_offset + Npad - 1)
    int den_stride;    // sb, den_mode 1: row stride of den
};
struct FsnGemmC {  // C store description
    int kind;      // 0 fragment-order + bias, 1 fb_out rows (bias + relu), 2 crm planes, 3 plain row-major
    float* p0;
    float* p1;
    const float* bias;
    int B, Tp, T, F, FP, Npad, N, la;
    int n_off;     // kind 2: first unit of the rows (row r is step r / Npad, unit n_off + r % Npad)
    long ld;       // kind 3: leading dimension of p0
    int rows, cols;  // kind 3: valid extent
};
// input of one band section of Improved FullSubNet, normalised, in the LSTM entries' layout (section_kernels.hip)
size_t fsn_section_input_workspace_floats(int B, int F);
int fsn_launch_section_input(const float* noisy, const float* fb, int B, int F, int T, int lower, int units, int sc, int sn,
                             int fc, int fn, int u_lo, int u_hi, float eps, float* out, int Np, int ldo, void* workspace,
                             hipStream_t s);
void fsn_tn_plan_splits(int M, int Nc, long K, int arith, int* splits, long* bound);  // lstm_train_kernels.hip (test hook)
// nn.Linear with O <= 4 outputs and I % 64 == 0 inputs as row dot products / outer products (gemm_kernels.hip)
bool fsn_linear_small_out_ok(int I, int O, long ldx);
int fsn_launch_linear_small_out(const float* x, long ldx, const float* w, const float* b, float* y, long R, int I, int O,
                                int relu, hipStream_t s);
int fsn_launch_linear_small_dx(const float* dy, long lddy, const float* w, float* dx, long lddx, long R, int I, int O,
                               hipStream_t s);
int fsn_launch_gemm(const FsnGemmA& a, const float* w_packed, const FsnGemmC& c, int row_tiles, int col_tiles,
                    int k_chunks, hipStream_t s);
// W [n_out][k] (transposed = 0) or W^T stored as [k][n_out] (transposed = 1, row stride ldw) -> B fragments
int fsn_launch_pack(const float* w, float* wp, int n_out, int k, int n_out_pad, int k_pad, hipStream_t s,
                    int transposed = 0, int ldw = 0);
int fsn_launch_bias_sum(const float* a, const float* b, float* out, int n, int n_pad, hipStream_t s);
int fsn_launch_bias_frag(const float* bias, float* frag, int n, hipStream_t s);

size_t fsn_lstm2_group_bptt_flag_words(int clusters);  // lstm_group_bptt_kernels.hip
int fsn_lstm2_group_bptt_clusters(int tiles);
int fsn_launch_lstm2_group_bptt(const float* dh1, const float* whh1T_p, const float* wih1T_p, const float* whh0T_p,
                                const float* save0, const float* save1, float* dg0, float* dg1, float* dx, unsigned* flags,
                                int Tp, int Nrows, int clusters, int H, hipStream_t s, int arith = FSN_ARITH_F32,
                                const void* w16 = nullptr);
// lstm_group16_kernels.hip: the same two-layer forward-with-saves / BPTT for the 16-bit training arithmetic (autocast),
// built for it: transposed products, per-wave weight streams, 16-byte hand-offs, K-split BPTT (see the file)
int fsn_lstm2_g16_clusters(int tiles);
size_t fsn_lstm2_g16_flag_words(int clusters);
size_t fsn_lstm2_g16_status_word(int clusters);
size_t fsn_lstm2_g16_partial_floats(int clusters);
size_t fsn_lstm2_g16_fwd_weight_halves(int Ipad);
size_t fsn_lstm2_g16_bwd_weight_bytes();
int fsn_launch_lstm2_g16_train(const float* x, int I, int Nrows, const float* w_ih0, const float* w_hh0, const float* w_ih1,
                               const float* w_hh1, const float* bias0, const float* bias1, float* hseq0, float* hseq1,
                               float* save0, float* save1, unsigned* flags, void* w16, int Tp, int clusters, int H,
                               hipStream_t s, int arith);
int fsn_launch_lstm2_g16_bptt(const float* dh1, const float* w_hh1, const float* w_ih1, const float* w_hh0, const float* save0,
                              const float* save1, float* dg0, float* dg1, float* exchange, unsigned* flags, void* wbuf, int Tp,
                              int Nrows, int clusters, int H, hipStream_t s, int arith, void* dg16_0, void* dg16_1, float* dbp,
                              int dg1_f32, int dg0_f32 = 1);
int fsn_launch_g16_left_to16(const float* dg, void* dg16, int Tp, int Nrows, long row0, int left, hipStream_t s, int arith);
int fsn_launch_lstm2_g16_finish(const float* dg1, const float* dg0, void* dg16_1, void* dg16_0, const float* dbp, int clusters, int Tp,
                                int Nrows, int left, float* db1, float* db0, hipStream_t s, int arith);
// fb_chain_bptt_kernels.hip: BPTT of the full-band model's two layers (16 rows, H = 512) as one persistent launch
bool fsn_fb_chain_bptt_supported(int H, int N);
int fsn_fb_chain_bptt_max_steps();
size_t fsn_fb_chain_bptt_dx_floats(int Tp, int N);
size_t fsn_fb_chain_bptt_flag_words();
size_t fsn_fb_chain_bptt_status_word();
int fsn_launch_fb_chain_bptt(const float* dh1, const float* whh1T_p, const float* wih1T_p, const float* whh0T_p,
                             const float* save0, const float* save1, float* dg0, float* dg1, float* dx, unsigned* flags,
                             int Tp, int N, int H, hipStream_t s);
int fsn_launch_zero_words(unsigned* p, size_t n, hipStream_t s);  // elementwise_kernels.hip
// dst[i] = src[i] rounded to fp16 / bf16 (arith = FSN_ARITH_F16 / _BF16): the 16-bit copy of a packed weight buffer
int fsn_launch_to16(const float* src, void* dst, size_t n, int arith, hipStream_t s);
// out[0..n) = NaN if *status != 0 (a persistent kernel hit its wait bound); status words of the kernels' flag arrays.
// The same kernel reports the event to the host: the sticky status record of the caller's stream (fsn_stream_status).
int fsn_launch_poison_if(const unsigned* status, float* out, size_t n, hipStream_t s);
int fsn_launch_hog(int workgroups, int lds_bytes, int heavy, unsigned long long ticks, float* sink, hipStream_t s);

// ---- residency contract of the persistent kernels whose workgroups wait for each other ----------
// (fsn_api.hip)  fb_chain_kernel, lstm2_group_kernel, lstm2_group_bptt_kernel and fb_chain_bptt_kernel make progress
// only when their WHOLE grid is resident.  The library guarantees what it can decide alone - the grid fits an idle
// device (fsn_grid_fits, from the compiled kernel's occupancy), its own persistent launches of different streams
// share the chip only when the set is provably placeable (PersistLaunch / fsn_persist_admit) - and bounds what it cannot: a foreign kernel (RCCL, another stream of the process) that holds CUs
// delays the missing workgroups until it ends; the resident ones wait for them, by the CLOCK (fsn_spin_ticks, not by a
// poll count), and a wait that runs out raises the launch's status word: outputs become NaN and the host learns of it
// through the stream's sticky status (FSN_ERR_TIMEOUT).
bool fsn_persistent_allowed();        // false: the caller switched these kernels off (fsn_set_persistent_mode)
unsigned long long fsn_spin_ticks();  // the wait bound in ticks of wall_clock64() on the current device
// whole grid co-resident on an idle device?  (blocks per CU from hipOccupancyMaxActiveBlocksPerMultiprocessor, cached)
bool fsn_grid_fits(const void* kernel, int block_threads, unsigned grid);
// every launch of such a kernel reports its footprint first: the launch then waits for earlier persistent launches on
// other streams until the set that may run beside it is provably placeable (fsn_api.hip, PersistLaunch)
void fsn_persist_admit(const void* kernel, int block_threads, unsigned grid);
#define FSN_PERSIST_LAUNCH(kernel, grid, block, s, ...)                                   \
    do {                                                                                  \
        fsn_persist_admit((const void*)(kernel), (int)(block).x, (unsigned)(grid).x);     \
        hipLaunchKernelGGL((kernel), grid, block, 0, s, __VA_ARGS__);                     \
    } while (0)
unsigned* fsn_ctx_sticky();           // device-visible sticky status record {status, events} of the running call's stream

#ifdef __HIPCC__
// One 256-poll round of a bounded wait has passed: give up?  The first round only takes the time (t0), so that a wait
// that succeeds at once never touches the clock.  `code` identifies the wait (1 + step) in the status word.
__device__ __forceinline__ bool fsn_wait_give_up(unsigned* status, unsigned spins, unsigned long long& t0,
                                                 unsigned long long ticks, unsigned code) {
    const unsigned st = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long now = (unsigned long long)wall_clock64();
    if (spins < 256u) t0 = now;
    if (st == 0 && now - t0 <= ticks) return false;
    if ((threadIdx.x & 63) == 0 && st == 0) __hip_atomic_store(status, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}
#endif
size_t fsn_fb_chain_status_word();
size_t fsn_lstm2_group_status_word(int clusters);
size_t fsn_lstm2_group_bptt_status_word(int clusters);

// one stack of fsn_launch_lstm2_group_multi (lstm_group_kernels.hip)
struct FsnGroupStack {
    const float* gx;
    const float *whh0_p, *wih1_p, *whh1_p, *bias1;
    float *hseq0, *hseq1;
    int N;
};
int fsn_lstm2_group_multi_cap();
int fsn_launch_lstm2_group_multi(int n, const FsnGroupStack* st, unsigned* flags, int Tp, int H, hipStream_t s);
int fsn_launch_lstm2_group_train(const float* x, long x_ld, int x_cols, int Nrows, const float* wih0_p, const float* whh0_p,
                                 const float* wih1_p, const float* whh1_p, const float* bias0, const float* bias1,
                                 float* hseq0, float* hseq1, float* save0, float* save1, unsigned* flags, int Tp,
                                 int clusters, int H, hipStream_t s, int arith = FSN_ARITH_F32,
                                 const void* w16 = nullptr);  // lstm_group_kernels.hip; w16: fsn_launch_to16 of the packed weights

// fb_chain_kernels.hip: the full-band model's two LSTM layers over all frames as one persistent launch
bool fsn_fb_chain_supported(int H, int Npad);
int fsn_fb_chain_max_steps();
size_t fsn_fb_chain_exchange_floats(int Tp, int Npad);
size_t fsn_fb_chain_flag_words();
int fsn_launch_fb_chain(const float* gx0, const float* whh0_p, const float* wih1_p, const float* whh1_p, const float* b1,
                        float* exchange, unsigned* flags, float* hseq1, int Tp, int Npad, int H, hipStream_t s,
                        float* hseq0 = nullptr, float* save0 = nullptr, float* save1 = nullptr, int cell = 0);
// gru_kernels.hip: nn.GRU's [3H] gate rows (r, z, n) of one layer as the FOUR-gate cell the chain kernel runs (r | z | nx | nh):
// w_ih4 [4H][I] = W_ir; W_iz; W_in; 0   w_hh4 [4H][H] = W_hr; W_hz; 0; W_hn   b4 [4H] = b_ir + b_hr; b_iz + b_hz; b_in; b_hn
// order 1: the gate slots of the many-row persistent kernels (lstm_kernels.hip, FSN_REC_GRU): nh | r | nx | z
int fsn_launch_gru_expand4(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, float* w_ih4, float* w_hh4,
                           float* b4, int I, int H, hipStream_t s, int order = 0);

// lstm_train_kernels.hip (training step: BPTT pieces)
// C [M][Nc] = sum_k A[k][M]^T B[k][Nc]   (both operands row-major over k; split-K, deterministic 2-pass)
size_t fsn_gemm_tn_workspace_bytes(int M, int Nc, long K);
int fsn_launch_gemm_tn(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int Nc, long K,
                       void* workspace, hipStream_t s, float* colsum_out = nullptr, int arith = FSN_ARITH_F32);
int fsn_launch_colsum(const float* A, long lda, float* out, int cols, long rows, void* workspace, hipStream_t s);
// the same product with both operands 16-bit in memory (LDS-DMA staging, transposing LDS reads); workspace as above
bool fsn_gemm_tn16h_supported(int M, int Nc, long K);
bool fsn_gemm_tn16n_supported(int M, int Nc, long K);  // the narrow form (Nc <= 32) and dx from the 16-bit gate gradients
int fsn_launch_gemm_tn16n(const void* A16, long lda, const void* B16, long ldb, float* C, long ldc, int M, int Nc, long K,
                          void* workspace, hipStream_t s, int arith);
bool fsn_gemm_dx16_supported(long rows, int G, int I);
int fsn_launch_gemm_dx16(const void* dg16, long ld16, const float* w, void* wfrag, float* dx, long lddx, long rows, int G, int I,
                         hipStream_t s, int arith);
int fsn_launch_gemm_tn16h(const void* A16, long lda, const void* B16, long ldb, float* C, long ldc, int M, int Nc, long K,
                          void* workspace, hipStream_t s, int arith);
size_t fsn_colsum_workspace_bytes(int cols, long rows);
int fsn_launch_bptt_step(const float* dh_out, const float* dgates_next, const float* whhT_p, float* dc,
                         const float* gates, const float* c_t, const float* c_prev, float* dgates, int row_tiles, int H,
                         int last, int first, hipStream_t s);
int fsn_launch_bptt_elem(const float* dh_out, const float* dh_rec, float* dc, const float* gates, const float* c_t,
                         const float* c_prev, float* dgates, long n_elems, int H, int last, int first, hipStream_t s);

// gru_kernels.hip
int fsn_launch_gru_step(const float* gx, const float* whh_p, const float* b_hn, const float* h_prev, float* h_out,
                        float* save, long gx_rt0, int row_tiles, int H, int first, hipStream_t s, int beside_persistent = 0);
int fsn_launch_gru_bptt_step(const float* dh_out, const float* dgx_next, const float* dghn_next, const float* whhT_p,
                             float* carry, const float* save, const float* h_prev, float* dgx, float* dghn,
                             int row_tiles, int H, int last, int first, hipStream_t s);

// lstm_kernels.hip
// Sub-band model input (fullsubnet/model.py:98-111) for kernels that build it on the fly:
// channel c < 2nb+1 of unit n = b F + f at frame t is mag[b][t][reflect(f + c - nb)], channel 2nb+1
// is fb_out[b][t][f]; everything divided by den (den_mode 0: den[b], 1: den[t * den_stride + n]).
struct FsnSbInput {
    const float* mag;
    const float* fb_out;
    const float* den;
    const float* wih_p;  // packed W_ih [4H/16][kin_pad/16][64][4]
    const float* bias;   // b_ih + b_hh [4H]
    int den_mode, den_stride;
    int B, Tp, F, FP, N, nb, kin_chunks;
    // generic form (x_rows != NULL): the layer input is a plain row-major matrix, element (t, n, c) at
    // x_rows[(t * x_step + n) * x_ld + c] with zero padding up to 16 kin_chunks columns; N = valid rows
    const float* x_rows;
    long x_ld, x_step;
    // row-range calls (fsn_fullsubnet_forward_rows): local row n of this launch is row n + row0 of the flattened
    // (b, f) index space; N stays the number of valid LOCAL rows.  0 everywhere else.
    long row0;
};

#ifdef __HIPCC__
// Element (row n, column c) of frame t of the sub-band model input: freq_unfold + cat + norm of
// fullsubnet/model.py:98-111 (reflect-padded neighbours of the noisy magnitude, then the full-band output,
// divided by the norm statistic), or the plain row-major form; zero beyond the valid rows / columns.
__device__ __forceinline__ float fsn_sb_input_value(const FsnSbInput& x, long n, int c, int t) {
    if (x.x_rows) return n < x.N ? x.x_rows[((long)t * x.x_step + n) * x.x_ld + c] : 0.f;
    if (n >= x.N || c > 2 * x.nb + 1) return 0.f;
    n += x.row0;
    const int b = (int)(n / x.F), f = (int)(n % x.F);
    const long fo = ((long)b * x.Tp + t) * x.FP;
    int j = f + c - x.nb;
    j = j < 0 ? -j : j;
    j = j >= x.F ? 2 * (x.F - 1) - j : j;
    const float raw = c <= 2 * x.nb ? x.mag[fo + j] : x.fb_out[fo + f];
    return raw / x.den[x.den_mode ? (long)t * x.den_stride + n : b];
}
#endif

// Output layer (nn.Linear(H, 2), fullsubnet/model.py:53-61 + the reshape of :129-135) fused into the
// last sub-band recurrent layer: the two mask values of a row are formed from h_t while it sits in LDS
// and go straight to the compressed-mask planes; the hidden sequence of that layer is never written.
struct FsnRecFc {
    const float* w_p;   // packed output weights [1][H/16][64][4] (rows 0, 1); NULL: not fused
    const float* bias;  // [16]
    float* crm_r;
    float* crm_i;       // [B][T][FP]
    int N, F, FP, T, la;
    long row0;          // first row of a row-range call in the flattened (b, f) space (N = valid local rows)
};

struct FsnRecPlan {
    int rt;          // 16-row tiles per workgroup of the persistent recurrent kernel
    int main_wgs;    // its grid: rows [0, main_wgs * rt * 16)
    int left_tiles;  // trailing tiles handled step by step on the auxiliary stream
    int tiles;       // all tiles = npad / 16
    int npad;        // padded row count (row stride of every [t][n] buffer)
};
FsnRecPlan fsn_lstm_rec_plan(int N, int H);
// beside_persistent: the launch runs concurrently with a resident lstm_rec_kernel (left-over tiles) and must
// use the small-footprint single-tile kernel to get a slot next to it
int fsn_launch_lstm_step(const float* gx, const float* whh_p, const float* h_prev, float* h_out, float* c,
                         long gx_rt0, int row_tiles, int H, int first, hipStream_t s, int beside_persistent = 0);
int fsn_launch_lstm_step_cu(const float* gx, const float* whh_p, const float* h_prev, float* h_out, float* c,
                            long gx_rt0, int row_tiles, int H, int first, hipStream_t s, const float* c_prev = nullptr,
                            float* gates_out = nullptr);
int fsn_launch_lstm_step_train(const float* gx, const float* whh_p, const float* h_prev, float* h_out,
                               const float* c_prev, float* c_out, float* gates_out, long gx_rt0, int row_tiles, int H,
                               int first, hipStream_t s);
// xin == NULL: accumulators start from the precomputed projection gx; otherwise the (K = 2nb+2)
// input projection of the sub-band model's first layer is computed inside the kernel from xin.
int fsn_launch_lstm_rec(const float* gx, const FsnSbInput* xin, const float* whh_p, float* hseq, int Tp, int Npad,
                        int H, int RT, int main_wgs, hipStream_t s, const FsnRecFc* fc = nullptr);
bool fsn_lstm_rec_can_fuse_fc(int RT, bool xin);
// first sub-band layer on the persistent kernel with the weight ring / deferred input staging (see the kernel)
bool fsn_lstm_rec_in_supported(const FsnSbInput* xin, const float* whh_p, int H, int RT);
int fsn_launch_lstm_rec_in(const FsnSbInput* xin, const float* whh_p, float* hseq, int Tp, int Npad, int H, int RT,
                           int main_wgs, hipStream_t s, int cell = 0);
// a layer with its input projection inside: xseq [Tp][Npad][H] is the hidden sequence of the layer below,
// wih_p / whh_p the packed weights, bias = b_ih + b_hh [4H]; either the output layer (fc) is fused and nothing else is
// stored (the last layer of the sub-band model), or hseq_out [Tp][Npad][H] receives h_t (a layer inside a stack)
bool fsn_lstm_rec_x_supported(int H, int RT);
int fsn_launch_lstm_rec_x(const float* xseq, const float* wih_p, const float* whh_p, const float* bias, int Tp, int Npad,
                          int H, int RT, int main_wgs, hipStream_t s, const FsnRecFc* fc, float* hseq_out = nullptr, int cell = 0);
// state_h0 / state_h1 (may be NULL): streaming continuation - the hidden states before this call ([rows][H],
// updated to the last step's on return); c0 / c1 then hold the carried cell states.
int fsn_launch_lstm_wavefront2(const float* gx0, long gx_stride, long gx_off, const float* whh0_p, const float* wih1_p,
                               const float* bias1_frag, const float* whh1_p, float* hseq0, float* hseq1, long hs_stride,
                               long hs_off, float* c0, float* c1, int T, int row_tiles, int H, hipStream_t s,
                               float* state_h0 = nullptr, float* state_h1 = nullptr, int beside_group = 0);
int fsn_launch_lstm_wavefront2w(const float* gx0, long gx_stride, long gx_off, const float* whh0_p,
                                const float* wih1_p, const float* bias1_frag, const float* whh1_p, float* hseq0,
                                float* hseq1, long hs_stride, long hs_off, float* c0, float* c1, int T, int row_tiles,
                                int H0, int H1, hipStream_t s, float* state_h0 = nullptr, float* state_h1 = nullptr,
                                int beside_group = 0);

// lstm_group_kernels.hip: the sub-band model for few rows as ONE persistent launch (clusters of 8 workgroups per 64
// rows, both layers + output layer; see the file).  exchange: fsn_lstm2_group_exchange_floats(clusters) floats,
// flags: fsn_lstm2_group_flag_words(clusters) 32-bit words (cleared inside the launcher).
size_t fsn_lstm2_group_exchange_floats(int clusters);
size_t fsn_lstm2_group_flag_words(int clusters);
int fsn_lstm2_group_clusters(int tiles);
int fsn_launch_lstm2_group(const FsnSbInput* xin, const float* whh0_p, const float* wih1_p, const float* whh1_p,
                           const float* bias1, float* exchange, unsigned* flags, const FsnRecFc* fc, int Tp, int clusters,
                           int H, hipStream_t s);

// gemm_f16x3_kernels.hip (experimental, opt-in: FSN_F16X3=1)
size_t fsn_f16x3_packed_halves(int n_out, int k);
int fsn_launch_pack_f16x3(const float* w, void* packed, int n_out, int k, hipStream_t s, float scale = 256.f);
int fsn_launch_gemm_f16x3(const float* A, long lda, const void* packed, const float* bias, float* C, long row_tiles,
                          int n_out, int k, hipStream_t s);
// lstm_f16x3_kernels.hip (experimental, opt-in: FSN_F16X3=1)
int fsn_launch_lstm_rec_f16x3(const float* gx, const void* packed, int Tp, int Npad, int H, int RT, int main_wgs,
                              const FsnRecFc* fc, hipStream_t s);
float fsn_f16x3_wih0_scale();
int fsn_launch_lstm_rec_xin_f16x3(const FsnSbInput* xin, const void* wih_packed, const void* whh_packed, float* hseq,
                                  int Tp, int Npad, int H, int RT, int main_wgs, hipStream_t s);
