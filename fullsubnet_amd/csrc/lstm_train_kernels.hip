// Training-step pieces of nn.LSTM (recipes/dns_interspeech_2020/fullsubnet/trainer.py:56-63 ->
// autograd through audio_zen/model/module/sequence_model.py:52-58): back-propagation through time.
//
// Per layer, with the activated gates i,f,g,o and the cell sequence c_t saved by the forward pass:
//   for t = T-1 .. 0:
//     dh      = dH_t (from the layer above) + dh_rec (from step t+1)
//     do      = dh * tanh(c_t);  dc = dc_carry + dh * o * (1 - tanh(c_t)^2)
//     di, dg, df = dc*g, dc*i, dc*c_{t-1};  dc_carry = dc * f
//     dgates_t = [di i(1-i), df f(1-f), dg (1-g^2), do o(1-o)]          (bptt_elem_kernel)
//     dh_rec  = dgates_t W_hh                                            (gemm_kernel, K = 4H)
//   dX = dgates W_ih (one GEMM over all steps);  dW_ih = dgates^T X;  dW_hh = dgates_{1..}^T H_{0..T-2};
//   db = column sums of dgates                                           (gemm_tn_kernel / colsum)
// First, correctness-first version: one elementwise launch + one small GEMM per step.
#include <stdlib.h>

#include "fsn_common.h"

namespace {

__global__ __launch_bounds__(256) void bptt_elem_kernel(const float* __restrict__ dh_out,
                                                        const float* __restrict__ dh_rec, float* __restrict__ dc,
                                                        const float* __restrict__ gates,
                                                        const float* __restrict__ c_t,
                                                        const float* __restrict__ c_prev,
                                                        float* __restrict__ dgates, long n_elems, int H, int last,
                                                        int first) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_elems) return;
    const long row = idx / H;
    const int u = (int)(idx % H);
    const float* gp = gates + row * 4 * H + u;
    const float ig = gp[0], fg = gp[H], gg = gp[2 * H], og = gp[3 * H];
    const float dh = dh_out[idx] + (last ? 0.f : dh_rec[idx]);
    const float tc = tanhf(c_t[idx]);
    const float d_o = dh * tc;
    const float dct = (last ? 0.f : dc[idx]) + dh * og * (1.f - tc * tc);
    const float cp = first ? 0.f : c_prev[idx];
    float* dg = dgates + row * 4 * H + u;
    dg[0] = dct * gg * ig * (1.f - ig);
    dg[H] = dct * cp * fg * (1.f - fg);
    dg[2 * H] = dct * ig * (1.f - gg * gg);
    dg[3 * H] = d_o * og * (1.f - og);
    dc[idx] = dct * fg;
}

// One BPTT step, fused: dh_rec = dgates_{t+1} W_hh for RTS 16-row tiles x CTS 16-unit groups (K = 4H,
// 4-way split-K over the waves: one dgates fragment feeds CTS MFMAs, one W_hh^T fragment RTS of them;
// partials reduced through LDS in a fixed order) followed by the cell derivative of those blocks ->
// dgates_t.  The mirror image of lstm_step_kernel; grid = (H/16/CTS, ceil(row tiles / RTS)).
template <int RTS, int CTS, int NW = 4>
__global__ __launch_bounds__(NW * 64) void bptt_step_kernel(const float* __restrict__ dh_out,
                                                        const float* __restrict__ dgates_next,
                                                        const float* __restrict__ whhT_p, float* __restrict__ dc,
                                                        const float* __restrict__ gates,
                                                        const float* __restrict__ c_t,
                                                        const float* __restrict__ c_prev,
                                                        float* __restrict__ dgates, int row_tiles, int H, int last,
                                                        int first) {
    __shared__ f32x4 red[NW][RTS][CTS][64];  // NW-way split-K (16 for the full-band model's single row tile)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int ug0 = blockIdx.x * CTS, rtile0 = blockIdx.y * RTS;
    const int G = 4 * H, KC = G >> 4;
    static_assert(RTS * CTS <= NW, "one finished tile per wave at most");
    // The wave that finishes tile (ert, ect) asks for everything the element-wise part reads - saved gates, dh, c_t,
    // dc, c_{t-1}: 8 cold values per element - before the K loop instead of after the barrier.
    const int ert = wave / CTS, ect = wave % CTS;
    const bool fin = wave < RTS * CTS && rtile0 + ert < row_tiles;
    float e_gate[4][4], e_dh[4], e_ct[4], e_dc[4], e_cp[4];
    if (fin) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long row = (long)(rtile0 + ert) * 16 + 4 * lq + i;
            const int u = (ug0 + ect) * 16 + lr;
            const float* gp = gates + row * G + u;
#pragma unroll
            for (int g = 0; g < 4; ++g) e_gate[i][g] = gp[(long)g * H];
            e_dh[i] = dh_out[row * H + u];
            e_ct[i] = c_t[row * H + u];
            e_dc[i] = last ? 0.f : dc[row * H + u];
            e_cp[i] = first ? 0.f : c_prev[row * H + u];
        }
    }
    if (!last) {
        f32x4 acc[RTS][CTS];
#pragma unroll
        for (int rt = 0; rt < RTS; ++rt)
#pragma unroll
            for (int ct = 0; ct < CTS; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int kc0 = wave * (KC / NW), kc1 = kc0 + KC / NW;
        const float* ap[RTS];
        const float* bp[CTS];
#pragma unroll
        for (int rt = 0; rt < RTS; ++rt) {
            int rtile = rtile0 + rt;
            rtile = rtile < row_tiles ? rtile : row_tiles - 1;
            ap[rt] = dgates_next + ((long)rtile * 16 + lr) * G + 4 * lq;
        }
#pragma unroll
        for (int ct = 0; ct < CTS; ++ct) bp[ct] = whhT_p + ((long)(ug0 + ct) * KC * 64 + lane) * 4;
#pragma unroll 2
        for (int kc = kc0; kc < kc1; ++kc) {
            f32x4 a[RTS], b[CTS];
#pragma unroll
            for (int rt = 0; rt < RTS; ++rt) a[rt] = *reinterpret_cast<const f32x4*>(ap[rt] + kc * 16);
#pragma unroll
            for (int ct = 0; ct < CTS; ++ct) b[ct] = *reinterpret_cast<const f32x4*>(bp[ct] + (long)kc * 256);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int rt = 0; rt < RTS; ++rt)
#pragma unroll
                    for (int ct = 0; ct < CTS; ++ct) acc[rt][ct] = mfma16(a[rt][j], b[ct][j], acc[rt][ct]);
        }
#pragma unroll
        for (int rt = 0; rt < RTS; ++rt)
#pragma unroll
            for (int ct = 0; ct < CTS; ++ct) red[wave][rt][ct][lane] = acc[rt][ct];
        __syncthreads();
    }
    if (fin) {
        const int rt = ert, ct = ect;
        const int rtile = rtile0 + rt;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (!last) {
            v = red[0][rt][ct][lane];
#pragma unroll
            for (int w = 1; w < NW; ++w) {
                const f32x4 r = red[w][rt][ct][lane];
                v = f32x4{v[0] + r[0], v[1] + r[1], v[2] + r[2], v[3] + r[3]};
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long row = (long)rtile * 16 + 4 * lq + i;
            const int u = (ug0 + ct) * 16 + lr;
            const long idx = row * H + u;
            const float ig = e_gate[i][0], fg = e_gate[i][1], gg = e_gate[i][2], og = e_gate[i][3];
            const float dh = e_dh[i] + v[i];
            const float tc = tanhf(e_ct[i]);
            const float d_o = dh * tc;
            const float dct = e_dc[i] + dh * og * (1.f - tc * tc);
            const float cp = e_cp[i];
            float* dg = dgates + row * G + u;
            dg[0] = dct * gg * ig * (1.f - ig);
            dg[H] = dct * cp * fg * (1.f - fg);
            dg[2 * H] = dct * ig * (1.f - gg * gg);
            dg[3 * H] = d_o * og * (1.f - og);
            dc[idx] = dct * fg;
        }
    }
}

// C[M][Nc] (partial, per K split) = sum_k A[k][m] * B[k][n]; A, B row-major over k (K % 16 == 0).
// Same execution shape as the forward GEMM (gemm_kernels.hip): ONE 4-wave workgroup per CU, one
// wave per SIMD with a large accumulator tile (RTW x CTW MFMA tiles) and a 2-deep register ring of
// operand chunks whose refills are pinned right behind the MFMAs that free them.  Lane (r = l&15,
// q = l>>4) feeds A[k0 + 4q + j][m0 + r] / B[k0 + 4q + j][n0 + r] to the j-th MFMA of a 16-deep
// chunk: 16 lanes read 64 contiguous bytes of one k row, the row base is wave-uniform (SGPR) and
// the lane part a fixed 32-bit offset.  Out-of-range columns are clamped on load and never stored.
// AR: arithmetic of the products (fsn_mma_k16): the lane's four k of a chunk ARE the 16-bit instruction's operand.
// ABL: experiment knob of tools/probe_tn.hip (0 in the library; a set bit gives WRONG results): 1 operands not loaded
template <int RTW, int CTW, int WM, int WN, int AR = FSN_ARITH_F32, int PF = 2, int ABL = 0>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const float* __restrict__ A, long lda,
                                                      const float* __restrict__ B, long ldb,
                                                      float* __restrict__ part, int M, int Nc, long K, long k_per_split,
                                                      int m_blocks, int n_blocks, float* __restrict__ asum_part,
                                                      int xcd_grouped) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;  // PF: operand chunks in flight (register ring)
    const int lr = lane & 15, lq = lane >> 4;
    const int wm = wave / WN, wn = wave % WN;
    int tile = blockIdx.x % (m_blocks * n_blocks), split = blockIdx.x / (m_blocks * n_blocks);
    if (xcd_grouped) {
        // all tiles of a K split on ONE XCD (block b runs on XCD b % 8: observed, speed only): the split's A rows are
        // read by n_blocks workgroups and its B rows by m_blocks - from that XCD's L2 after the first touch instead of
        // once each from HBM (the 16-bit forms are bandwidth-bound: 11 GB per GEMM at config 3's shape otherwise)
        const int tiles = m_blocks * n_blocks, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        tile = j % tiles;
        split = xcd * ((int)(gridDim.x >> 3) / tiles) + j / tiles;
    }
    const int mb = tile / n_blocks, nb = tile % n_blocks;
    const int m0 = (mb * WM + wm) * RTW * 16, n0 = (nb * WN + wn) * CTW * 16;
    const long k_begin = (long)split * k_per_split;
    long k_end = k_begin + k_per_split;
    k_end = k_end < K ? k_end : K;
    const int chunks = (int)((k_end - k_begin) >> 4), last = chunks - 1;

    int aoff[RTW], boff[CTW];
#pragma unroll
    for (int i = 0; i < RTW; ++i) {
        const int m = m0 + i * 16 + lr;
        aoff[i] = 4 * lq * (int)lda + (m < M ? m : M - 1);
    }
#pragma unroll
    for (int i = 0; i < CTW; ++i) {
        const int n = n0 + i * 16 + lr;
        boff[i] = 4 * lq * (int)ldb + (n < Nc ? n : Nc - 1);
    }
    const float* a0 = A + k_begin * lda;
    const float* b0 = B + k_begin * ldb;

    f32x4 acc[RTW][CTW];
#pragma unroll
    for (int i = 0; i < RTW; ++i)
#pragma unroll
        for (int j = 0; j < CTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // column sums of A over k (= the bias gradient when A is dgates) ride along for free: the A fragments are in
    // registers anyway; 8 adds per chunk next to 128 MFMAs.  Written by the n-block-0 / wave-column-0 waves only.
    float asum[RTW];
#pragma unroll
    for (int i = 0; i < RTW; ++i) asum[i] = 0.f;
    float abuf[PF][RTW][4], bbuf[PF][CTW][4];
    auto fetch = [&](int p, int kc) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* ar = a0 + ((long)kc * 16 + j) * lda;  // wave-uniform row bases
            const float* br = b0 + ((long)kc * 16 + j) * ldb;
#pragma unroll
            for (int i = 0; i < RTW; ++i) abuf[p][i][j] = (ABL & 1) ? 0.25f * (float)kc : ar[aoff[i]];
#pragma unroll
            for (int i = 0; i < CTW; ++i) bbuf[p][i][j] = (ABL & 1) ? 0.5f : br[boff[i]];
        }
    };
    auto consume = [&](int p) {
        if constexpr (AR == FSN_ARITH_F32) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < RTW; ++i)
#pragma unroll
                    for (int jj = 0; jj < CTW; ++jj) acc[i][jj] = mfma16(abuf[p][i][j], bbuf[p][jj][j], acc[i][jj]);
        } else {
            typename FsnOperand<AR>::type ao[RTW], bo[CTW];
#pragma unroll
            for (int i = 0; i < RTW; ++i)
                ao[i] = fsn_operand<AR>(f32x4{abuf[p][i][0], abuf[p][i][1], abuf[p][i][2], abuf[p][i][3]});
#pragma unroll
            for (int jj = 0; jj < CTW; ++jj)
                bo[jj] = fsn_operand<AR>(f32x4{bbuf[p][jj][0], bbuf[p][jj][1], bbuf[p][jj][2], bbuf[p][jj][3]});
#pragma unroll
            for (int i = 0; i < RTW; ++i)
#pragma unroll
                for (int jj = 0; jj < CTW; ++jj) acc[i][jj] = fsn_mma_k16<AR>(ao[i], bo[jj], acc[i][jj]);
        }
#pragma unroll
        for (int i = 0; i < RTW; ++i) asum[i] += (abuf[p][i][0] + abuf[p][i][1]) + (abuf[p][i][2] + abuf[p][i][3]);
    };
#pragma unroll
    for (int p = 0; p < PF; ++p) fetch(p, p < last ? p : last);
    const int k_main = (chunks / PF) * PF;
    for (int kc0 = 0; kc0 < k_main; kc0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            consume(p);
            __builtin_amdgcn_sched_barrier(0);
            const int kn = kc0 + p + PF;
            fetch(p, kn < last ? kn : last);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int p = 0; p < PF - 1; ++p)  // the left-over chunks (fewer than PF) are already in slots 0 ..
        if (k_main + p < chunks) consume(p);

    float* out = part + (long)split * M * Nc;
#pragma unroll
    for (int i = 0; i < RTW; ++i)
#pragma unroll
        for (int jj = 0; jj < CTW; ++jj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + i * 16 + 4 * lq + r, n = n0 + jj * 16 + lr;
                if (m < M && n < Nc) out[(long)m * Nc + n] = acc[i][jj][r];
            }
    if (asum_part && nb == 0 && wn == 0) {
#pragma unroll
        for (int i = 0; i < RTW; ++i) {
            float v = asum[i];  // lanes r, r + 16, r + 32, r + 48 hold the four k phases of column m0 + 16 i + r
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int m = m0 + i * 16 + lr;
            if (lq == 0 && m < M) asum_part[(long)split * M + m] = v;
        }
    }
}

// The same product under the 16-bit training arithmetic, built for it (round 4).  With 16-bit operands the matrix work of
// a 192 x 192 tile is 36 instructions of 8 cycles per 16 k, and gemm_tn_kernel feeds them with 48 DWORD loads per lane
// (a lane's operand is four k of one column: four rows of a K-major matrix) - it ran at 14 % of the 16-bit peak, bound
// by its load instructions.  Here both operand slabs of a 32-k chunk are fetched as 16-BYTE row pieces (12 per thread
// instead of 96 dword loads), rounded to 16 bits once on the way into LDS - [k step][column tile][16 k][16 columns], k
// rows of 32 bytes - and every wave reads its operands with ds_read_b64_tr_b16 (gfx950's transposing LDS read: lane (lr,
// lq) of a 16-lane group addresses row lr / 4, column quad lr % 4 of a [4 k][16 columns] block and receives column lr of
// the four rows = the matrix instruction's operand A[m = lr][k = 4 lq + j]; measured, tools/probe_tr16.hip).  Two LDS
// stages, one barrier per chunk, the next chunk's loads in flight under the current chunk's matrix work.  Same products
// (operands rounded exactly as fsn_mma_k16 rounds them), same k order, same K splits: bit-identical partial sums.  The
// column sums of A (the bias gradient) ride on the staging threads' fp32 values.
typedef short tq_s16x4 __attribute__((ext_vector_type(4)));
constexpr int TQ_TS = 544;             // bytes per [16 k][16 columns] 16-bit subtile (512 + 32: spreads the staging writes over banks)
constexpr int TQ_OP = 2 * 12 * TQ_TS;  // one operand of one chunk: 2 k steps x 12 column tiles
constexpr int TQ_STAGE = 2 * TQ_OP;    // A then B
constexpr int TQ_LDS = 2 * TQ_STAGE + 4 * 48 * 16;
constexpr int TQ_PF = 2;               // chunks of operand rows in flight per staging thread
template <int AR>
__global__ __launch_bounds__(256) void gemm_tn16_kernel(const float* __restrict__ A, long lda, const float* __restrict__ B,
                                                        long ldb, float* __restrict__ part, int M, int Nc, long K,
                                                        long k_per_split, int m_blocks, int n_blocks,
                                                        float* __restrict__ asum_part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tq_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    // all tiles of a K split on ONE XCD (block b runs on XCD b % 8: observed, speed only), as gemm_tn_kernel's grouped form
    const int tiles = m_blocks * n_blocks, xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int tile = jb % tiles, split = xcd * ((int)(gridDim.x >> 3) / tiles) + jb / tiles;
    const int mb = tile / n_blocks, nb = tile % n_blocks;
    const int m0 = mb * 192, n0 = nb * 192;
    const long k_begin = (long)split * k_per_split;
    long k_end = k_begin + k_per_split;
    k_end = k_end < K ? k_end : K;
    const int chunks = (int)((k_end - k_begin + 31) >> 5);

    // staging: thread t < 192 owns the 16-byte piece q = t % 48 of the 192 columns for the k rows 8 (t / 48) + i, i < 8
    const bool stager = tid < 192;
    const int q = tid % 48, kp = (tid / 48) & 3;
    const float* ap = A + (k_begin + kp * 8) * lda + m0 + 4 * q;
    const float* bp = B + (k_begin + kp * 8) * ldb + n0 + 4 * q;
    // TQ_PF chunks of operand rows in flight per staging thread (register sets, statically indexed): the slabs come from
    // HBM (3 GB per product, each element read once: PMC FETCH_SIZE = 3.4 GB), and a chunk's matrix work is ~0.25 us - one
    // chunk ahead left the kernel at 2.3 TB/s, latency-bound; two: 2.65.  Three do not fit the 256 architectural
    // registers a load can target beside the operands (the accumulators live in the other half of the file).
    f32x4 va[TQ_PF][8], vb[TQ_PF][8], asum = {0.f, 0.f, 0.f, 0.f};
    auto load = [&](int set, int c) {
        if (!stager) return;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const long k = k_begin + (long)c * 32 + kp * 8 + i;
            const bool ok = k < k_end;
            va[set][i] = ok ? *reinterpret_cast<const f32x4*>(ap + ((long)c * 32 + i) * lda) : f32x4{0.f, 0.f, 0.f, 0.f};
            vb[set][i] = ok ? *reinterpret_cast<const f32x4*>(bp + ((long)c * 32 + i) * ldb) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store = [&](int set, int stage) {
        if (!stager) return;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int kk = kp * 8 + i;
            unsigned char* d = tq_lds + stage * TQ_STAGE + ((kk >> 4) * 12 + (q >> 2)) * TQ_TS + (kk & 15) * 32 + (q & 3) * 8;
            *reinterpret_cast<fsn_u32x2*>(d) = __builtin_bit_cast(fsn_u32x2, fsn_operand<AR>(va[set][i]));
            *reinterpret_cast<fsn_u32x2*>(d + TQ_OP) = __builtin_bit_cast(fsn_u32x2, fsn_operand<AR>(vb[set][i]));
            asum += va[set][i];
        }
    };
    f32x4 acc[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int lane_off = (4 * lq + (lr >> 2)) * 32 + (lr & 3) * 8;
    auto tr = [&](const unsigned char* p) {
        return __builtin_bit_cast(typename FsnOperand<AR>::type,
                                  __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tq_s16x4*)p));
    };
    auto compute = [&](int stage) {
        // both k steps of the chunk as ONE K = 32 matrix instruction per tile (fsn_mma_k32: round 5)
        typename FsnOperand<AR>::type a[2][6], b[2][6];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const unsigned char* base = tq_lds + stage * TQ_STAGE + ks * 12 * TQ_TS + lane_off;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                a[ks][i] = tr(base + (wm * 6 + i) * TQ_TS);
                b[ks][i] = tr(base + TQ_OP + (wn * 6 + i) * TQ_TS);
            }
        }
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[i][j] = fsn_mma_k32<AR>(a[0][i], a[1][i], b[0][j], b[1][j], acc[i][j]);
    };
    // chunk c lives in register set c % TQ_PF and LDS stage c & 1; rows beyond k_end (and whole chunks beyond the last) load zeros
#pragma unroll
    for (int d = 0; d < TQ_PF; ++d) load(d, d);
    store(0, 0);
    __syncthreads();
    for (int c0 = 0; c0 < chunks; c0 += 2 * TQ_PF) {  // 2 TQ_PF: both the register set and the LDS stage of a chunk are static
#pragma unroll
        for (int d = 0; d < 2 * TQ_PF; ++d) {
            const int c = c0 + d;
            if (c < chunks) {  // uniform
                load(d % TQ_PF, c + TQ_PF);           // set of chunk c (already in LDS) is free: chunk c + TQ_PF takes it
                compute(d & 1);
                store((d + 1) % TQ_PF, (d + 1) & 1);  // chunk c + 1 (zeros beyond the end) into the other stage
                __syncthreads();
            }
        }
    }

    float* out = part + (long)split * M * Nc;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + (wm * 6 + i) * 16 + 4 * lq + r, n = n0 + (wn * 6 + j) * 16 + lr;
                out[(long)m * Nc + n] = acc[i][j][r];
            }
    if (asum_part && nb == 0) {  // the four k phases of a column quad meet in a fixed order
        f32x4* red = reinterpret_cast<f32x4*>(tq_lds + 2 * TQ_STAGE);
        if (stager) red[kp * 48 + q] = asum;
        __syncthreads();
        if (tid < 48) {
            f32x4 v = red[tid];
#pragma unroll
            for (int k = 1; k < 4; ++k) v += red[k * 48 + tid];
            *reinterpret_cast<f32x4*>(asum_part + (long)split * M + m0 + 4 * tid) = v;
        }
    }
}

// ... and with the operands ALREADY in 16 bits in memory (the BPTT kernel writes the 16-bit gate gradients it forms for the
// exchange anyway; the hidden sequences are converted once): half the HBM bytes, and no conversion pass - so the slabs go
// global -> LDS by LDS-DMA (no registers: four chunks of 24 KB in flight per CU instead of the two that fit the
// architectural registers), each DMA instruction building two [16 k][16 columns] subtiles of the image
// ds_read_b64_tr_b16 wants (lane l fetches row (l & 31) >> 1, half l & 1 of subtile l >> 5).  K in chunks of 32 (the
// caller passes K rounded down to 32; the reduce kernel adds the tail rows from the same 16-bit operands).
constexpr int TH_STAGES = 4;             // (6 stages = 5 chunks in flight measured SLOWER, 0.76 against 0.66 ms: the LDS-DMA path lands ~29 GB/s per CU whatever is in flight)
// WN: waves along the N side - 2: 192 x 192 tile, four waves (the square plan); 4: 192 x 384 tile, eight waves - a weight
// gradient's whole 384-column side in one workgroup, so that the [k][1536] gate-gradient rows are fetched once per 192 of
// their columns instead of twice (the launch is bound by the bytes its DMAs land: 36 KB per chunk for 192 x 384 outputs
// against 2 x 24 KB)
template <int WN>
constexpr int th_stage_bytes() { return 2 * (12 + 6 * WN) * 512; }  // (A 12 + B 6 WN column tiles) x 2 k steps x 512 B
__device__ __forceinline__ void th_lds_dma(const unsigned short* g, unsigned lds_base) {
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
template <int AR, int WN>
__global__ __launch_bounds__(WN * 128) void gemm_tn16h_kernel(const unsigned short* __restrict__ A, long lda,
                                                              const unsigned short* __restrict__ B, long ldb,
                                                              float* __restrict__ part, int M, int Nc, long K, long k_per_split,
                                                              int m_blocks, int n_blocks) {
    constexpr int TH_STAGE = th_stage_bytes<WN>();
    constexpr int NTB = 6 * WN;        // B column tiles of the workgroup
    constexpr int B0 = 12 * 1024;      // byte offset of the B block inside a stage (A: 2 k steps x 6 KB)
    extern __shared__ __attribute__((aligned(16))) unsigned char th_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const int wm = wave / WN, wn = wave % WN;
    const int tiles = m_blocks * n_blocks, xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int tile = jb % tiles, split = xcd * ((int)(gridDim.x >> 3) / tiles) + jb / tiles;
    const int mb = tile / n_blocks, nb = tile % n_blocks;
    const int m0 = mb * 192, n0 = nb * (96 * WN);
    const long k_begin = (long)split * k_per_split;
    long k_end = k_begin + k_per_split;
    k_end = k_end < K ? k_end : K;
    const int chunks = (int)((k_end - k_begin) >> 5);  // whole chunks of 32 (K and k_per_split are multiples of 32)
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)th_lds;

    // staging roles (six DMA instructions per chunk each, one per pair of column tiles): role 0, 1 = A, k step 0 / 1;
    // role 2 + 2 h + ks = B columns [192 h, 192 h + 192), k step ks.  Wave w takes role w (2 + WN roles: with eight waves the
    // last two stage nothing).  Lane l -> row 16 ks + ((l & 31) >> 1), columns 16 (2 p + (l >> 5)) + 8 (l & 1) .. + 7
    const bool loader = wave < 2 + WN;
    const int s_op = wave >= 2 ? 1 : 0, s_ks = wave & 1, s_half = wave >= 2 ? (wave - 2) >> 1 : 0;
    const long s_ld = s_op ? ldb : lda;
    const unsigned short* sp = (s_op ? B + n0 + 192 * s_half : A + m0) + (k_begin + 16 * s_ks + ((lane & 31) >> 1)) * s_ld +
                               16 * (lane >> 5) + 8 * (lane & 1);
    const unsigned s_dst = s_op ? (unsigned)(B0 + (s_ks * (NTB / 2) + s_half * 6) * 1024) : (unsigned)(s_ks * 6 * 1024);
    auto issue = [&](int c) {  // chunk c into stage c % TH_STAGES
        if (!loader) return;
        const unsigned dst = lds0 + (unsigned)((c % TH_STAGES) * TH_STAGE) + s_dst;
#pragma unroll
        for (int p = 0; p < 6; ++p) th_lds_dma(sp + (long)c * 32 * s_ld + 32 * p, dst + (unsigned)(p * 1024));
    };
    f32x4 acc[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int lane_off = (4 * lq + (lr >> 2)) * 32 + (lr & 3) * 8;
    auto tr = [&](const unsigned char* p) {
        return __builtin_bit_cast(typename FsnOperand<AR>::type,
                                  __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tq_s16x4*)p));
    };
    auto compute = [&](int stage) {
        typename FsnOperand<AR>::type a[2][6], b[2][6];
        const unsigned char* base = th_lds + stage * TH_STAGE + lane_off;  // column tile i of an operand block at + i * 512
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                a[ks][i] = tr(base + ks * 6 * 1024 + (wm * 6 + i) * 512);
                b[ks][i] = tr(base + B0 + ks * (NTB / 2) * 1024 + (wn * 6 + i) * 512);
            }
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[i][j] = fsn_mma_k32<AR>(a[0][i], a[1][i], b[0][j], b[1][j], acc[i][j]);
    };
    // TH_STAGES - 1 chunks in flight; the DMAs are invisible to the compiler's counter, so the waits are stated here: a
    // staging wave issues 6 per chunk, in order, and nothing else that counts
#pragma unroll
    for (int c = 0; c < TH_STAGES - 1; ++c)
        if (c < chunks) issue(c);
    for (int c = 0; c < chunks; ++c) {
        static_assert(TH_STAGES == 4, "the counted wait below: TH_STAGES - 2 younger chunks x 6 DMAs per wave");
        if (c + TH_STAGES - 1 <= chunks) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  // two younger chunks may still be in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // chunk c of every wave has landed; everyone has left stage (c - 1) % TH_STAGES
        if (c + TH_STAGES - 1 < chunks) issue(c + TH_STAGES - 1);
        compute(c % TH_STAGES);
    }
    float* out = part + (long)split * M * Nc;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + (wm * 6 + i) * 16 + 4 * lq + r, n = n0 + (wn * 6 + j) * 16 + lr;
                out[(long)m * Nc + n] = acc[i][j][r];
            }
}
// its epilogue: sum of the split partials (fixed order) + the tail rows (K % 32) from the same 16-bit operands
template <int AR>
__global__ void tn16h_reduce_kernel(const float* __restrict__ part, float* __restrict__ C, long ldc, int M, int Nc, int splits,
                                    const unsigned short* __restrict__ A, long lda, const unsigned short* __restrict__ B,
                                    long ldb, long k_tail0, long K) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)M * Nc) return;
    const int m = (int)(i / Nc), n = (int)(i % Nc);
    float acc = 0.f;
    for (int s = 0; s < splits; ++s) acc += part[(long)s * M * Nc + i];
    for (long k = k_tail0; k < K; ++k) {
        float a, b;
        if constexpr (AR == FSN_ARITH_F16) {
            a = (float)__builtin_bit_cast(_Float16, A[k * lda + m]);
            b = (float)__builtin_bit_cast(_Float16, B[k * ldb + n]);
        } else {
            a = __builtin_bit_cast(float, (unsigned)A[k * lda + m] << 16);
            b = __builtin_bit_cast(float, (unsigned)B[k * ldb + n] << 16);
        }
        acc = fmaf(a, b, acc);
    }
    C[(long)m * ldc + n] = acc;
}

// column sums riding on gemm_tn: sum of the split partials (fixed order) + the K % 16 tail rows
__global__ void tn_colsum_reduce_kernel(const float* __restrict__ asum_part, float* __restrict__ out, int M, int splits,
                                        const float* __restrict__ A, long lda, long k_tail0, long K) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    float acc = 0.f;
    for (int s = 0; s < splits; ++s) acc += asum_part[(long)s * M + m];
    for (long k = k_tail0; k < K; ++k) acc += A[k * lda + m];
    out[m] = acc;
}

// out[i] = sum_s part[s][i] in a fixed order
__global__ void reduce_splits_kernel(const float* __restrict__ part, float* __restrict__ C, long ldc, int M, int Nc,
                                     int splits) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)M * Nc) return;
    float acc = 0.f;
    for (int s = 0; s < splits; ++s) acc += part[(long)s * M * Nc + i];
    C[(i / Nc) * ldc + (i % Nc)] = acc;
}

// gemm_tn epilogue: sum of the split partials (fixed order) + the K % 16 tail rows the MFMA kernel
// does not cover; `transposed`: the partials hold C^T ([Nc][M], operands were swapped).
__global__ void tn_reduce_kernel(const float* __restrict__ part, float* __restrict__ C, long ldc, int M, int Nc,
                                 int splits, int transposed, const float* __restrict__ A, long lda,
                                 const float* __restrict__ B, long ldb, long k_tail0, long K) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)M * Nc) return;
    const int m = (int)(i / Nc), n = (int)(i % Nc);
    const long pi = transposed ? (long)n * M + m : i;
    float acc = 0.f;
    for (int s = 0; s < splits; ++s) acc += part[(long)s * M * Nc + pi];
    for (long k = k_tail0; k < K; ++k) acc += A[k * lda + m] * B[k * ldb + n];
    C[(long)m * ldc + n] = acc;
}

// partial column sums over blocks of rows: part[rb][c] = sum_{r in block rb} A[r][c]
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ A, long lda,
                                                             float* __restrict__ part, int cols, long rows,
                                                             long rows_per_block) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    const long r0 = (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block;
    r1 = r1 < rows ? r1 : rows;
    float acc = 0.f;
    for (long r = r0; r < r1; ++r) acc += A[r * lda + c];
    part[(long)blockIdx.y * cols + c] = acc;
}

// the same for at most 16 columns (the 2-wide output layer): 16 row groups x 16 columns per block instead of two
// busy threads; the row groups are combined in a fixed order
__global__ __launch_bounds__(256) void colsum_narrow_kernel(const float* __restrict__ A, long lda,
                                                            float* __restrict__ part, int cols, long rows,
                                                            long rows_per_block) {
    __shared__ float red[16][17];
    const int c = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const long r0 = (long)blockIdx.x * rows_per_block;
    long r1 = r0 + rows_per_block;
    r1 = r1 < rows ? r1 : rows;
    float acc = 0.f;
    if (c < cols)
        for (long r = r0 + rg; r < r1; r += 16) acc += A[r * lda + c];
    red[rg][c] = acc;
    __syncthreads();
    if (threadIdx.x < 16 && c < cols) {
        float t = 0.f;
        for (int g = 0; g < 16; ++g) t += red[g][c];
        part[(long)blockIdx.x * cols + c] = t;
    }
}

struct TnPlan {
    int m_blocks, n_blocks, splits, narrow;
    int square;  // 192 x 192 tiles, every K split's tiles on one XCD (bandwidth-bound under the 16-bit arithmetic)
    long k_per_split;
};
// Workgroup tile 256 x 128 (wave tile 8 x 4, waves 2 x 2) or, for narrow outputs (the K = 2nb+2
// input projection), 512 x 32 (wave tile 8 x 2, waves 4 x 1).  K is split so that the grid is one
// workgroup per CU (or as close below it as the tile count allows).
TnPlan tn_plan(int M, int Nc, long K, int arith = FSN_ARITH_F32, bool allow_square = true) {
    TnPlan p;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    p.narrow = Nc <= 32;
    p.square = 0;
    (void)arith;  // every arithmetic: in fp32 the square plan is worth 0.3 ms of a 42 ms step, under the 16-bit one 1 ms per GEMM
    if (allow_square && M % 192 == 0 && Nc % 192 == 0 && cus % 8 == 0 && (cus / 8) % ((M / 192) * (Nc / 192)) == 0 &&
        K >= (long)(cus / ((M / 192) * (Nc / 192))) * 128) {
        const long s = cus / ((M / 192) * (Nc / 192));  // whole splits per XCD, one workgroup per CU
        const long kps = ((K + s - 1) / s + 15) / 16 * 16;
        if ((K + kps - 1) / kps == s) {  // every split non-empty (the kernel's grid is fixed by the XCD mapping)
            p.square = 1;
            p.m_blocks = M / 192;
            p.n_blocks = Nc / 192;
            p.k_per_split = kps;
            p.splits = (int)s;
            return p;
        }
    }
    p.m_blocks = p.narrow ? (M + 511) / 512 : (M + 255) / 256;
    p.n_blocks = p.narrow ? (Nc + 31) / 32 : (Nc + 127) / 128;
    const long tiles = (long)p.m_blocks * p.n_blocks;
    long s = cus / tiles;
    const long max_s = (K + 127) / 128;  // at least 8 chunks per split
    s = s < max_s ? s : max_s;
    s = s < 1 ? 1 : s;
    p.k_per_split = ((K + s - 1) / s + 15) / 16 * 16;
    p.splits = (int)((K + p.k_per_split - 1) / p.k_per_split);
    return p;
}
#ifndef FSN_TN_PF32
#define FSN_TN_PF32 2  // operand chunks in flight of the fp32 192 x 192 form (measured r04: 2 -> 3.76 ms, 3 -> 3.72: not latency-bound)
// (r04, tools/probe_tn.hip, profiles/r04_tn_probe.txt: without its operand loads this kernel's matrix stream runs at 0.98 of
// the fp32 peak, 3.03 ms, with them at 0.80, 3.67.  Three LDS-staged forms of the same product - 16-byte row pieces through
// registers with one / two chunks in flight, and by LDS-DMA with software-pipelined ds_read_b32 operands - measured 4.6 /
// 4.2 / 3.67 ms: the staged form only reaches the register ring's time (its DMA costs 0.3 ms, its LDS reads 0.2, its
// barrier 0.15), so the ring stays.)
#endif

// ---- layer 0's INPUT-side products from the 16-bit gate gradients (round 6) -------------------------------------------------
// With these two the BPTT launch of the 16-bit arithmetic stores no fp32 gate gradients at all (2.45 GB less written per
// step at config 3's shape, 2 x 2.45 GB less read): dW_ih0 = dgates0^T x and dx = dgates0 W_ih0 take the row-major 16-bit
// copies the launch writes for the large products anyway - both operands of both products rounded to 16 bits, as every other
// product of the autocast arithmetic (dx used to be an fp32 product of the fp32 gradients: wider than the reference's own).
//
// (1) dW_ih0 [M = 4H][<= 32] = A^T B: gemm_tn16h_kernel's staging (LDS-DMA into the image ds_read_b64_tr_b16 wants, four stages)
// on a 384 x 32 tile: four waves x 96 gate columns, B = the 32 input columns.  A chunk of 32 k: A 2 k steps x 24 column
// tiles x 512 B, B 2 x 2 x 512 B.  Wave w stages A's k step w & 1, column pairs 6 (w >> 1) .. + 5; waves 0, 1 also B's k step w.
constexpr int TNN_STAGE = 2 * (24 + 2) * 512;
template <int AR>
__global__ __launch_bounds__(256) void gemm_tn16n_kernel(const unsigned short* __restrict__ A, long lda,
                                                         const unsigned short* __restrict__ B, long ldb, float* __restrict__ part,
                                                         int M, long K, long k_per_split, int m_blocks) {
    constexpr int B0 = 2 * 24 * 512;
    extern __shared__ __attribute__((aligned(16))) unsigned char th_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const int mb = (int)blockIdx.x % m_blocks, split = (int)blockIdx.x / m_blocks;
    const int m0 = mb * 384;
    const long k_begin = (long)split * k_per_split;
    long k_end = k_begin + k_per_split;
    k_end = k_end < K ? k_end : K;
    const int chunks = k_end > k_begin ? (int)((k_end - k_begin) >> 5) : 0;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)th_lds;
    const int s_ks = wave & 1, s_half = wave >> 1;
    const long s_row = k_begin + 16 * s_ks + ((lane & 31) >> 1);
    const int s_col = 16 * (lane >> 5) + 8 * (lane & 1);
    const unsigned short* spa = A + m0 + 192 * s_half + s_row * lda + s_col;
    const unsigned short* spb = B + s_row * ldb + s_col;
    const unsigned dst_a = (unsigned)(s_ks * 24 * 512 + s_half * 6 * 1024), dst_b = (unsigned)(B0 + s_ks * 2 * 512);
    auto issue = [&](int c) {
        const unsigned st = lds0 + (unsigned)((c % TH_STAGES) * TNN_STAGE);
#pragma unroll
        for (int p = 0; p < 6; ++p) th_lds_dma(spa + (long)c * 32 * lda + 32 * p, st + dst_a + (unsigned)(p * 1024));
        if (wave < 2) th_lds_dma(spb + (long)c * 32 * ldb, st + dst_b);
    };
    f32x4 acc[6][2];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int lane_off = (4 * lq + (lr >> 2)) * 32 + (lr & 3) * 8;
    auto tr = [&](const unsigned char* p) {
        return __builtin_bit_cast(typename FsnOperand<AR>::type,
                                  __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tq_s16x4*)p));
    };
#pragma unroll
    for (int c = 0; c < TH_STAGES - 1; ++c)
        if (c < chunks) issue(c);
    for (int c = 0; c < chunks; ++c) {
        static_assert(TH_STAGES == 4, "the counted waits below: two younger chunks of 7 (waves 0, 1) / 6 DMAs per wave");
        if (c + TH_STAGES - 1 <= chunks) {
            if (wave < 2) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();  // chunk c of every wave has landed; everyone has left stage (c - 1) % TH_STAGES
        if (c + TH_STAGES - 1 < chunks) issue(c + TH_STAGES - 1);
        const unsigned char* base = th_lds + (c % TH_STAGES) * TNN_STAGE + lane_off;
        typename FsnOperand<AR>::type a[2][6], b[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < 6; ++i) a[ks][i] = tr(base + ks * 24 * 512 + (wave * 6 + i) * 512);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[ks][j] = tr(base + B0 + ks * 2 * 512 + j * 512);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = fsn_mma_k32<AR>(a[0][i], a[1][i], b[0][j], b[1][j], acc[i][j]);
    }
    float* out = part + (long)split * M * 32;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(long)(m0 + (wave * 6 + i) * 16 + 4 * lq + r) * 32 + j * 16 + lr] = acc[i][j][r];
}
// its epilogue: the split partials [split][M][32] (fixed order) + the K % 32 tail rows, the first Nc <= 32 columns out
template <int AR>
__global__ void tn16n_reduce_kernel(const float* __restrict__ part, float* __restrict__ C, long ldc, int M, int Nc, int splits,
                                    const unsigned short* __restrict__ A, long lda, const unsigned short* __restrict__ B,
                                    long ldb, long k_tail0, long K) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)M * Nc) return;
    const int m = (int)(i / Nc), n = (int)(i % Nc);
    float acc = 0.f;
    for (int s = 0; s < splits; ++s) acc += part[((long)s * M + m) * 32 + n];
    for (long k = k_tail0; k < K; ++k) {
        float a, b;
        if constexpr (AR == FSN_ARITH_F16) {
            a = (float)__builtin_bit_cast(_Float16, A[k * lda + m]);
            b = (float)__builtin_bit_cast(_Float16, B[k * ldb + n]);
        } else {
            a = __builtin_bit_cast(float, (unsigned)A[k * lda + m] << 16);
            b = __builtin_bit_cast(float, (unsigned)B[k * ldb + n] << 16);
        }
        acc = fmaf(a, b, acc);
    }
    C[(long)m * ldc + n] = acc;
}

// (2) dx [rows][I <= 32] = dg16 [rows][G] W [G][I], formed transposed like every product of the 16-bit kernels:
// D^T[input column][row] = W^T (A operand: fragments packed once, resident in LDS) x dg16^T (B operand: a lane's eight
// consecutive gate columns of one row = one 16-byte load straight from the row-major copy), so a lane ends up with four
// consecutive input columns of one row: 16-byte stores.  A wave walks 16-row tiles; its operand loads run twelve K blocks
// ahead.  HBM-bound: 2 G bytes per row in, 128 out.
constexpr int DX16_DEPTH = 12;
template <int AR>
__global__ void dx16_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int G, int I) {
    // fragment (kb, j): lane (lr, lq) holds W[32 kb + 8 lq + e][16 j + lr], e = 0..7 (zero beyond I)
    const long n = (long)(G / 32) * 2 * 64;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63), j = (int)((i >> 6) & 1), kb = (int)(i >> 7);
        const int col = 16 * j + (lane & 15), k0 = 32 * kb + 8 * (lane >> 4);
        f32x4 lo, hi;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            lo[e] = col < I ? w[(long)(k0 + e) * I + col] : 0.f;
            hi[e] = col < I ? w[(long)(k0 + 4 + e) * I + col] : 0.f;
        }
        const fsn_u32x2 a = __builtin_bit_cast(fsn_u32x2, fsn_operand<AR>(lo)), b = __builtin_bit_cast(fsn_u32x2, fsn_operand<AR>(hi));
        reinterpret_cast<fsn_u32x4*>(out)[i] = fsn_u32x4{a[0], a[1], b[0], b[1]};
    }
}
template <int AR>
__global__ __launch_bounds__(256) void gemm_dx16_kernel(const unsigned short* __restrict__ dg16, long ld16,
                                                        const unsigned short* __restrict__ wfrag, float* __restrict__ dx, long lddx,
                                                        long tiles, int I, int KB) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wl[];  // [KB][2][64 lanes][16 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lq = lane >> 4;
    for (int i = tid; i < KB * 2 * 64; i += 256) reinterpret_cast<fsn_u32x4*>(wl)[i] = reinterpret_cast<const fsn_u32x4*>(wfrag)[i];
    __syncthreads();
    auto opnd = [](const fsn_u32x4 v, int h) {
        return fsn_wfrag_operand<AR>(fsn_u32x2{v[2 * h], v[2 * h + 1]});
    };
    for (long tile = (long)blockIdx.x * 4 + wave; tile < tiles; tile += (long)gridDim.x * 4) {
        const unsigned short* p = dg16 + (tile * 16 + lr) * ld16 + lq * 8;
        fsn_u32x4 ring[DX16_DEPTH];
#pragma unroll
        for (int d = 0; d < DX16_DEPTH; ++d) ring[d] = *reinterpret_cast<const fsn_u32x4*>(p + d * 32);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        for (int kb0 = 0; kb0 < KB; kb0 += DX16_DEPTH) {
#pragma unroll
            for (int d = 0; d < DX16_DEPTH; ++d) {
                const int kb = kb0 + d;
                const fsn_u32x4 b = ring[d];
                if (kb + DX16_DEPTH < KB) ring[d] = *reinterpret_cast<const fsn_u32x4*>(p + (kb + DX16_DEPTH) * 32);
                const fsn_u32x4 a0 = reinterpret_cast<const fsn_u32x4*>(wl)[(kb * 2 + 0) * 64 + lane];
                const fsn_u32x4 a1 = reinterpret_cast<const fsn_u32x4*>(wl)[(kb * 2 + 1) * 64 + lane];
                acc0 = fsn_mma_k32<AR>(opnd(a0, 0), opnd(a0, 1), opnd(b, 0), opnd(b, 1), acc0);
                acc1 = fsn_mma_k32<AR>(opnd(a1, 0), opnd(a1, 1), opnd(b, 0), opnd(b, 1), acc1);
            }
        }
        float* o = dx + (tile * 16 + lr) * lddx + 4 * lq;
        if (I == 32) {
            *reinterpret_cast<f32x4*>(o) = acc0;
            *reinterpret_cast<f32x4*>(o + 16) = acc1;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (4 * lq + r < I) o[r] = acc0[r];
                if (16 + 4 * lq + r < I) o[16 + r] = acc1[r];
            }
        }
    }
}

constexpr size_t kTnOnePerCu = 96 * 1024;  // LDS reservation (never touched): one workgroup per CU
constexpr long kColsumRows = 2048;
// rows per block of a column-sum launch: 2048, or fewer when that would leave most of the chip idle - the full-band
// output layer's 257 columns x 3120 rows ran on 4 workgroups walking 2048 rows each (0.46 ms; 49 blocks of 64: ~0.02)
static long colsum_rows_per_block(int cols, long rows) {
    long r = kColsumRows;
    const long col_blocks = cols <= 16 ? 1 : (cols + 255) / 256;
    while (r > 64 && ((rows + r - 1) / r) * col_blocks < 256) r >>= 1;
    return r;
}

}  // namespace

// The most K splits ANY plan of this (M, Nc) can take, whatever K: callers size one scratch buffer for several products
// of the same shape and slightly different K ((T - 1) N against T N rows), and the two plans split K differently (one
// workgroup per CU over 256 x 128 tiles, or over 192 x 192 tiles).  fsn_launch_gemm_tn refuses a plan beyond it.
static long tn_max_splits(int M, int Nc) {
    const bool swap = M <= 32 && Nc > 32;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int m = swap ? Nc : M, n = swap ? M : Nc;
    const bool narrow = n <= 32;
    const long tiles = narrow ? (long)((m + 511) / 512) * ((n + 31) / 32) : (long)((m + 255) / 256) * ((n + 127) / 128);
    long splits = cus / tiles > 1 ? cus / tiles : 1;
    if (!swap && M % 192 == 0 && Nc % 192 == 0) {
        const long sq = cus / ((long)(M / 192) * (Nc / 192));
        splits = sq > splits ? sq : splits;
    }
    if (!swap && M % 192 == 0 && Nc % 384 == 0) {  // the 192 x 384 form of the 16-bit-operand products: half as many tiles
        const long wq = cus / ((long)(M / 192) * (Nc / 384));
        splits = wq > splits ? wq : splits;
    }
    return splits;
}
// test hook (fsn_debug_tn_plan): the K splits fsn_launch_gemm_tn would take for this product and the bound its scratch is
// sized by
void fsn_tn_plan_splits(int M, int Nc, long K, int arith, int* splits, long* bound) {
    const bool swap = M <= 32 && Nc > 32;
    const long K16 = K & ~15L;
    const TnPlan p = K16 <= 0 ? TnPlan{} : swap ? tn_plan(Nc, M, K16, arith, false) : tn_plan(M, Nc, K16, arith);
    if (splits) *splits = K16 <= 0 ? 1 : p.splits;
    if (bound) *bound = tn_max_splits(M, Nc);
}
size_t fsn_gemm_tn_workspace_bytes(int M, int Nc, long K) {
    if ((K & ~15L) <= 0) return (size_t)M * (Nc + 1) * sizeof(float);
    return (size_t)tn_max_splits(M, Nc) * M * (Nc + 1) * sizeof(float);  // + one column-sum row per split
}

// colsum_out (may be NULL): also out[m] = sum_k A[k][m], from the same pass over A (not with a narrow M)
int fsn_launch_gemm_tn(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int Nc, long K,
                       void* workspace, hipStream_t s, float* colsum_out, int arith) {
    if (arith != FSN_ARITH_F32 && arith != FSN_ARITH_F16 && arith != FSN_ARITH_BF16) {
        fsn_set_error("gemm_tn: arithmetic %d unknown", arith);
        return FSN_ERR_ARG;
    }
    if (K <= 0 || lda * 16 > 0x7fffffffL || ldb * 16 > 0x7fffffffL) {
        fsn_set_error("gemm_tn: bad K = %ld or leading dimension", K);
        return FSN_ERR_ARG;
    }
    // a narrow M (the 2-row dW of the sub-band output layer) goes on the narrow side of the tile
    const bool swap = M <= 32 && Nc > 32;
    if (swap && colsum_out) {
        fsn_set_error("gemm_tn: fused column sums are not available for M <= 32");
        return FSN_ERR_ARG;
    }
    const long K16 = K & ~15L;
    float* part = static_cast<float*>(workspace);
    float* asum_part = nullptr;
    int splits = 0;
    if (K16 > 0) {
        const TnPlan p = swap ? tn_plan(Nc, M, K16, arith, false) : tn_plan(M, Nc, K16, arith);
        if (p.splits > tn_max_splits(M, Nc)) {  // the scratch buffer is sized by that bound
            fsn_set_error("gemm_tn: plan of %d splits for %d x %d exceeds the workspace bound", p.splits, M, Nc);
            return FSN_ERR_WORKSPACE;
        }
        if (colsum_out) asum_part = part + (size_t)p.splits * M * Nc;
        auto wide = arith == FSN_ARITH_F16    ? gemm_tn_kernel<8, 4, 2, 2, FSN_ARITH_F16>
                    : arith == FSN_ARITH_BF16 ? gemm_tn_kernel<8, 4, 2, 2, FSN_ARITH_BF16>
                                              : gemm_tn_kernel<8, 4, 2, 2>;
        auto narrow = arith == FSN_ARITH_F16    ? gemm_tn_kernel<8, 2, 4, 1, FSN_ARITH_F16>
                      : arith == FSN_ARITH_BF16 ? gemm_tn_kernel<8, 2, 4, 1, FSN_ARITH_BF16>
                                                : gemm_tn_kernel<8, 2, 4, 1>;
        static bool attr_set[4] = {false, false, false, false};
        if (!attr_set[arith]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(wide), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)kTnOnePerCu) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(narrow), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)kTnOnePerCu) != hipSuccess) {
                fsn_set_error("gemm_tn: cannot reserve %zu bytes of LDS", kTnOnePerCu);
                return FSN_ERR_LAUNCH;
            }
            attr_set[arith] = true;
        }
        const dim3 grid((unsigned)(p.m_blocks * p.n_blocks * p.splits));
        if (p.square && arith != FSN_ARITH_F32 && lda % 4 == 0 && ldb % 4 == 0 && ((size_t)A & 15) == 0 && ((size_t)B & 15) == 0) {
            // the 16-bit arithmetic's own kernel: 16-byte operand loads, transposing LDS reads
            auto k16 = arith == FSN_ARITH_F16 ? gemm_tn16_kernel<FSN_ARITH_F16> : gemm_tn16_kernel<FSN_ARITH_BF16>;
            static bool k16_set[4] = {false, false, false, false};
            if (!k16_set[arith]) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(k16), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)kTnOnePerCu) != hipSuccess) {
                    fsn_set_error("gemm_tn: cannot reserve %zu bytes of LDS", kTnOnePerCu);
                    return FSN_ERR_LAUNCH;
                }
                k16_set[arith] = true;
            }
            static_assert(TQ_LDS <= (int)kTnOnePerCu, "the reservation that keeps one workgroup per CU holds the stages");
            hipLaunchKernelGGL(k16, grid, dim3(256), kTnOnePerCu, s, A, lda, B, ldb, part, M, Nc, K16, p.k_per_split, p.m_blocks,
                               p.n_blocks, asum_part);
        } else if (p.square) {
            auto square = arith == FSN_ARITH_F16    ? gemm_tn_kernel<6, 6, 2, 2, FSN_ARITH_F16>
                          : arith == FSN_ARITH_BF16 ? gemm_tn_kernel<6, 6, 2, 2, FSN_ARITH_BF16>
                                                    : gemm_tn_kernel<6, 6, 2, 2, FSN_ARITH_F32, FSN_TN_PF32>;
            static bool sq_set[4] = {false, false, false, false};
            if (!sq_set[arith]) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(square), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)kTnOnePerCu) != hipSuccess) {
                    fsn_set_error("gemm_tn: cannot reserve %zu bytes of LDS", kTnOnePerCu);
                    return FSN_ERR_LAUNCH;
                }
                sq_set[arith] = true;
            }
            hipLaunchKernelGGL(square, grid, dim3(256), kTnOnePerCu, s, A, lda, B, ldb, part, M, Nc, K16, p.k_per_split,
                               p.m_blocks, p.n_blocks, asum_part, 1);
        } else if (swap)
            hipLaunchKernelGGL(p.narrow ? narrow : wide, grid, dim3(256), kTnOnePerCu, s, B, ldb, A, lda, part, Nc, M,
                               K16, p.k_per_split, p.m_blocks, p.n_blocks, (float*)nullptr, 0);
        else
            hipLaunchKernelGGL(p.narrow ? narrow : wide, grid, dim3(256), kTnOnePerCu, s, A, lda, B, ldb, part, M, Nc,
                               K16, p.k_per_split, p.m_blocks, p.n_blocks, asum_part, 0);
        FSN_TRY_LAUNCH("gemm_tn_kernel");
        splits = p.splits;
    }
    const long n = (long)M * Nc;
    hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, part, C, ldc, M, Nc, splits,
                       swap ? 1 : 0, A, lda, B, ldb, K16, K);
    FSN_TRY_LAUNCH("tn_reduce_kernel");
    if (colsum_out) {
        hipLaunchKernelGGL(tn_colsum_reduce_kernel, dim3((M + 255) / 256), dim3(256), 0, s, asum_part, colsum_out, M,
                           splits, A, lda, K16, K);
        return fsn_check_launch("tn_colsum_reduce_kernel");
    }
    return FSN_OK;
}

// C [M][Nc] = sum_k A[k][m] B[k][n] with BOTH operands 16-bit in memory (fp16 / bf16 per `arith`), fp32 accumulation:
// M and Nc multiples of 192, one workgroup per CU over the 192 x 192 tiles with every K split's tiles on one XCD (the
// plan of fsn_launch_gemm_tn's square form); false when the shape has no such plan.  workspace: fsn_gemm_tn_workspace_bytes.
static int g_tn16h_wide = 1;  // fsn_tn16h_wide(0): the 192 x 192 tiles also where the 192 x 384 form applies (A/B measurements, tests)
void fsn_tn16h_wide(int on) { g_tn16h_wide = on; }
// the 192 x 384 form: Nc a multiple of 384, whole K splits per XCD with one (eight-wave) workgroup per CU
static bool tn16h_wide_plan(int M, int Nc, long K32, TnPlan* out) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (!g_tn16h_wide || M % 192 || Nc % 384 || cus % 8) return false;
    const int tiles = (M / 192) * (Nc / 384);
    if ((cus / 8) % tiles) return false;
    const long s = cus / tiles;
    const long kps = ((K32 + s - 1) / s + 31) / 32 * 32;
    if (kps < 128 || (K32 + kps - 1) / kps != s || s > tn_max_splits(M, Nc)) return false;
    if (out) {
        out->square = 1;
        out->narrow = 0;
        out->m_blocks = M / 192;
        out->n_blocks = Nc / 384;
        out->splits = (int)s;
        out->k_per_split = kps;
    }
    return true;
}
bool fsn_gemm_tn16h_supported(int M, int Nc, long K) {
    const long K32 = K & ~31L;
    if (K32 <= 0) return false;
    return tn16h_wide_plan(M, Nc, K32, nullptr) || tn_plan(M, Nc, K32, FSN_ARITH_F16).square != 0;
}
int fsn_launch_gemm_tn16h(const void* A16, long lda, const void* B16, long ldb, float* C, long ldc, int M, int Nc, long K,
                          void* workspace, hipStream_t s, int arith) {
    const long K32 = K & ~31L;
    if ((arith != FSN_ARITH_F16 && arith != FSN_ARITH_BF16) || !fsn_gemm_tn16h_supported(M, Nc, K) || lda % 8 || ldb % 8 ||
        ((size_t)A16 & 15) || ((size_t)B16 & 15)) {
        fsn_set_error("gemm_tn16h: 16-bit arithmetic, M and Nc multiples of 192 with a one-workgroup-per-CU plan, 16-byte aligned rows");
        return FSN_ERR_ARG;
    }
    TnPlan p{};
    const bool wide = tn16h_wide_plan(M, Nc, K32, &p);
    if (!wide) {
        p = tn_plan(M, Nc, K32, arith);
        p.k_per_split = (p.k_per_split + 31) / 32 * 32;  // whole chunks; the last split takes what is left (K32 is a multiple of 32)
        if ((K32 + p.k_per_split - 1) / p.k_per_split != p.splits || p.splits > tn_max_splits(M, Nc)) {
            fsn_set_error("gemm_tn16h: no plan for %d x %d, K = %ld", M, Nc, K);
            return FSN_ERR_ARG;
        }
    }
    float* part = static_cast<float*>(workspace);
    const unsigned short *a = static_cast<const unsigned short*>(A16), *b = static_cast<const unsigned short*>(B16);
    const dim3 grid((unsigned)(p.m_blocks * p.n_blocks * p.splits));
    if (wide) {
        constexpr size_t kLdsW = (size_t)TH_STAGES * th_stage_bytes<4>();  // 144 KB: one workgroup per CU by itself
        auto kern = arith == FSN_ARITH_F16 ? gemm_tn16h_kernel<FSN_ARITH_F16, 4> : gemm_tn16h_kernel<FSN_ARITH_BF16, 4>;
        static bool setw[4] = {false, false, false, false};
        if (!setw[arith]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsW) !=
                hipSuccess) {
                fsn_set_error("gemm_tn16h: cannot reserve %zu bytes of LDS", kLdsW);
                return FSN_ERR_LAUNCH;
            }
            setw[arith] = true;
        }
        hipLaunchKernelGGL(kern, grid, dim3(512), kLdsW, s, a, lda, b, ldb, part, M, Nc, K32, p.k_per_split, p.m_blocks, p.n_blocks);
    } else {
        constexpr size_t kLds = kTnOnePerCu;  // 4 stages of 24 KB = the reservation that keeps one workgroup per CU
        static_assert(TH_STAGES * th_stage_bytes<2>() <= (int)kTnOnePerCu, "the stages fit it");
        auto kern = arith == FSN_ARITH_F16 ? gemm_tn16h_kernel<FSN_ARITH_F16, 2> : gemm_tn16h_kernel<FSN_ARITH_BF16, 2>;
        static bool set[4] = {false, false, false, false};
        if (!set[arith]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds) !=
                hipSuccess) {
                fsn_set_error("gemm_tn16h: cannot reserve %zu bytes of LDS", kLds);
                return FSN_ERR_LAUNCH;
            }
            set[arith] = true;
        }
        hipLaunchKernelGGL(kern, grid, dim3(256), kLds, s, a, lda, b, ldb, part, M, Nc, K32, p.k_per_split, p.m_blocks, p.n_blocks);
    }
    FSN_TRY_LAUNCH("gemm_tn16h_kernel");
    const long n = (long)M * Nc;
    if (arith == FSN_ARITH_F16)
        hipLaunchKernelGGL(tn16h_reduce_kernel<FSN_ARITH_F16>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, part, C, ldc, M, Nc,
                           p.splits, a, lda, b, ldb, K32, K);
    else
        hipLaunchKernelGGL(tn16h_reduce_kernel<FSN_ARITH_BF16>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, part, C, ldc, M, Nc,
                           p.splits, a, lda, b, ldb, K32, K);
    return fsn_check_launch("tn16h_reduce_kernel");
}


// dW [M][Nc <= 32] = A16^T B16 over K rows, both operands 16-bit row-major in memory (A [K][lda >= M], B [K][ldb >= 32], its
// columns beyond Nc zero); M a multiple of 384.  workspace: fsn_gemm_tn_workspace_bytes(M, Nc, K).
bool fsn_gemm_tn16n_supported(int M, int Nc, long K) { return M % 384 == 0 && Nc >= 1 && Nc <= 32 && (K & ~31L) >= 32 * 64; }
int fsn_launch_gemm_tn16n(const void* A16, long lda, const void* B16, long ldb, float* C, long ldc, int M, int Nc, long K,
                          void* workspace, hipStream_t s, int arith) {
    if ((arith != FSN_ARITH_F16 && arith != FSN_ARITH_BF16) || !fsn_gemm_tn16n_supported(M, Nc, K) || lda % 8 || ldb % 8 || ldb < 32 ||
        ((size_t)A16 & 15) || ((size_t)B16 & 15)) {
        fsn_set_error("gemm_tn16n: 16-bit arithmetic, M a multiple of 384, Nc <= 32 in rows of >= 32 16-bit columns, 16-byte aligned rows");
        return FSN_ERR_ARG;
    }
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const long K32 = K & ~31L;
    const int m_blocks = M / 384;
    long splits = cus / m_blocks > 1 ? cus / m_blocks : 1;
    const long bound = (long)(fsn_gemm_tn_workspace_bytes(M, Nc, K) / ((size_t)M * 32 * sizeof(float)));
    splits = splits < bound ? splits : bound;
    const long kps = ((K32 + splits - 1) / splits + 31) / 32 * 32;
    splits = (K32 + kps - 1) / kps;
    constexpr size_t kLds = (size_t)TH_STAGES * TNN_STAGE;  // 104 KB: one workgroup per CU by itself
    auto kern = arith == FSN_ARITH_F16 ? gemm_tn16n_kernel<FSN_ARITH_F16> : gemm_tn16n_kernel<FSN_ARITH_BF16>;
    static bool set[4] = {false, false, false, false};
    if (!set[arith]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds) != hipSuccess) {
            fsn_set_error("gemm_tn16n: cannot reserve %zu bytes of LDS", kLds);
            return FSN_ERR_LAUNCH;
        }
        set[arith] = true;
    }
    float* part = static_cast<float*>(workspace);
    const unsigned short *a = static_cast<const unsigned short*>(A16), *b = static_cast<const unsigned short*>(B16);
    hipLaunchKernelGGL(kern, dim3((unsigned)(m_blocks * splits)), dim3(256), kLds, s, a, lda, b, ldb, part, M, K32, kps, m_blocks);
    FSN_TRY_LAUNCH("gemm_tn16n_kernel");
    const long n = (long)M * Nc;
    if (arith == FSN_ARITH_F16)
        hipLaunchKernelGGL(tn16n_reduce_kernel<FSN_ARITH_F16>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, part, C, ldc, M, Nc,
                           (int)splits, a, lda, b, ldb, K32, K);
    else
        hipLaunchKernelGGL(tn16n_reduce_kernel<FSN_ARITH_BF16>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, part, C, ldc, M, Nc,
                           (int)splits, a, lda, b, ldb, K32, K);
    return fsn_check_launch("tn16n_reduce_kernel");
}

// dx [rows][I <= 32] (row stride lddx) = dg16 [rows][G] W [G][I] with both operands rounded to 16 bits; rows a multiple of 16,
// G a multiple of 32 up to 2048; wfrag: G * 32 16-bit words of scratch for the packed weight (written here).
bool fsn_gemm_dx16_supported(long rows, int G, int I) {
    return rows > 0 && rows % 16 == 0 && G % (32 * DX16_DEPTH) == 0 && G <= 2048 && I >= 1 && I <= 32;
}
int fsn_launch_gemm_dx16(const void* dg16, long ld16, const float* w, void* wfrag, float* dx, long lddx, long rows, int G, int I,
                         hipStream_t s, int arith) {
    if ((arith != FSN_ARITH_F16 && arith != FSN_ARITH_BF16) || !fsn_gemm_dx16_supported(rows, G, I) || ld16 % 8 || lddx % 4 ||
        ((size_t)dg16 & 15) || ((size_t)dx & 15) || ((size_t)wfrag & 15)) {
        fsn_set_error("gemm_dx16: 16-bit arithmetic, rows %% 16 == 0, G %% 384 == 0 (<= 2048), I <= 32, 16-byte aligned rows");
        return FSN_ERR_ARG;
    }
    const int KB = G / 32;
    const size_t lds = (size_t)KB * 2 * 1024;
    unsigned short* wf = static_cast<unsigned short*>(wfrag);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const long tiles = rows / 16;
    const unsigned grid = (unsigned)((tiles + 3) / 4 < cus ? (tiles + 3) / 4 : cus);
    if (arith == FSN_ARITH_F16) {
        hipLaunchKernelGGL(dx16_pack_kernel<FSN_ARITH_F16>, dim3((unsigned)((KB * 128 + 255) / 256)), dim3(256), 0, s, w, wf, G, I);
        FSN_TRY_LAUNCH("dx16_pack_kernel");
        static bool set = false;
        if (!set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_dx16_kernel<FSN_ARITH_F16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    128 * 1024) != hipSuccess) {
                fsn_set_error("gemm_dx16: cannot reserve LDS");
                return FSN_ERR_LAUNCH;
            }
            set = true;
        }
        hipLaunchKernelGGL(gemm_dx16_kernel<FSN_ARITH_F16>, dim3(grid), dim3(256), lds, s, static_cast<const unsigned short*>(dg16), ld16, wf,
                           dx, lddx, tiles, I, KB);
    } else {
        hipLaunchKernelGGL(dx16_pack_kernel<FSN_ARITH_BF16>, dim3((unsigned)((KB * 128 + 255) / 256)), dim3(256), 0, s, w, wf, G, I);
        FSN_TRY_LAUNCH("dx16_pack_kernel");
        static bool set = false;
        if (!set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_dx16_kernel<FSN_ARITH_BF16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    128 * 1024) != hipSuccess) {
                fsn_set_error("gemm_dx16: cannot reserve LDS");
                return FSN_ERR_LAUNCH;
            }
            set = true;
        }
        hipLaunchKernelGGL(gemm_dx16_kernel<FSN_ARITH_BF16>, dim3(grid), dim3(256), lds, s, static_cast<const unsigned short*>(dg16), ld16, wf,
                           dx, lddx, tiles, I, KB);
    }
    return fsn_check_launch("gemm_dx16_kernel");
}

size_t fsn_colsum_workspace_bytes(int cols, long rows) {
    const long r = colsum_rows_per_block(cols, rows);
    return (size_t)((rows + r - 1) / r) * cols * sizeof(float);
}

int fsn_launch_colsum(const float* A, long lda, float* out, int cols, long rows, void* workspace, hipStream_t s) {
    const long rpb = colsum_rows_per_block(cols, rows);
    const int rb = (int)((rows + rpb - 1) / rpb);
    float* part = static_cast<float*>(workspace);
    if (cols <= 16)
        hipLaunchKernelGGL(colsum_narrow_kernel, dim3(rb), dim3(256), 0, s, A, lda, part, cols, rows, rpb);
    else
        hipLaunchKernelGGL(colsum_partial_kernel, dim3((cols + 255) / 256, rb), dim3(256), 0, s, A, lda, part, cols, rows,
                           rpb);
    FSN_TRY_LAUNCH("colsum_partial_kernel");
    hipLaunchKernelGGL(reduce_splits_kernel, dim3((cols + 255) / 256), dim3(256), 0, s, part, out, (long)cols, 1, cols,
                       rb);
    return fsn_check_launch("reduce_splits_kernel");
}

// The BPTT step in the one-workgroup-per-CU shape of lstm_step_cu_kernel (lstm_kernels.hip), for row counts that
// fill the chip at least once (used from 192 row tiles): a workgroup = four row tiles x CTW
// column tiles of dh_rec = dgates_{t+1} W_hh, one row tile per wave, the whole K = 4H range per wave (no split-K
// exchange).  A stage is four K chunks: wave w fetches chunk w's CTW weight fragments for everybody (two-stage LDS
// buffer, one barrier per stage) and its own four A fragments; the next stage's fetch is pinned under the 16 CTW
// MFMAs of this one.  Everything the element-wise part reads (saved gates, dh, c_t, dc, c_{t-1}) is requested
// before the K loop - with one wave per SIMD there are registers to spare, and read afterwards each of the 8 CTW x 4
// values per lane would pay its own memory round trip.
template <int CTW>
__global__ __launch_bounds__(256) void bptt_step_cu_kernel(const float* __restrict__ dh_out,
                                                           const float* __restrict__ dgates_next,
                                                           const float* __restrict__ whhT_p, float* __restrict__ dc,
                                                           const float* __restrict__ gates,
                                                           const float* __restrict__ c_t, const float* c_prev,
                                                           float* __restrict__ dgates, int H, int last, int first) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int ug0 = blockIdx.x * CTW;
    const long rtile = (long)blockIdx.y * 4 + wave;
    const int G = 4 * H, KC = G >> 4, stages = KC >> 2;
    if (first) c_prev = c_t;  // any valid address: the value is not used at t = 0
    float e_gate[CTW][4][4], e_dh[CTW][4], e_ct[CTW][4], e_dc[CTW][4], e_cp[CTW][4];
#pragma unroll
    for (int ct = 0; ct < CTW; ++ct)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long row = rtile * 16 + 4 * lq + i;
            const int u = (ug0 + ct) * 16 + lr;
            const float* gp = gates + row * G + u;
#pragma unroll
            for (int g = 0; g < 4; ++g) e_gate[ct][i][g] = gp[(long)g * H];
            e_dh[ct][i] = dh_out[row * H + u];
            e_ct[ct][i] = c_t[row * H + u];
            e_dc[ct][i] = last ? 0.f : dc[row * H + u];
            e_cp[ct][i] = c_prev[row * H + u];
        }
    f32x4 acc[CTW];
#pragma unroll
    for (int ct = 0; ct < CTW; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!last) {  // uniform over the workgroup (barriers inside)
        __shared__ f32x4 bsh[2][4 * CTW][64];
        const float* ap = dgates_next + (rtile * 16 + lr) * G + 4 * lq;
        const float* bp = whhT_p + ((long)ug0 * KC * 64 + lane) * 4;
        f32x4 an[4], bn[CTW];
        auto fetch = [&](int st) {
#pragma unroll
            for (int q = 0; q < 4; ++q) an[q] = *reinterpret_cast<const f32x4*>(ap + (st * 4 + q) * 16);
#pragma unroll
            for (int ct = 0; ct < CTW; ++ct)
                bn[ct] = *reinterpret_cast<const f32x4*>(bp + ((long)ct * KC + st * 4 + wave) * 256);
        };
        fetch(0);
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct) bsh[0][wave * CTW + ct][lane] = bn[ct];
        f32x4 a[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = an[q];
        __syncthreads();
        for (int st = 0; st < stages; ++st) {
            __builtin_amdgcn_sched_barrier(0);
            fetch(st + 1 < stages ? st + 1 : st);
            __builtin_amdgcn_sched_barrier(0);
            const int buf = st & 1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 b[CTW];
#pragma unroll
                for (int ct = 0; ct < CTW; ++ct) b[ct] = bsh[buf][q * CTW + ct][lane];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int ct = 0; ct < CTW; ++ct) acc[ct] = mfma16(a[q][j], b[ct][j], acc[ct]);
            }
#pragma unroll
            for (int ct = 0; ct < CTW; ++ct) bsh[buf ^ 1][wave * CTW + ct][lane] = bn[ct];
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = an[q];
            __syncthreads();
        }
    }
#pragma unroll
    for (int ct = 0; ct < CTW; ++ct)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long row = rtile * 16 + 4 * lq + i;
            const int u = (ug0 + ct) * 16 + lr;
            const long idx = row * H + u;
            const float ig = e_gate[ct][i][0], fg = e_gate[ct][i][1], gg = e_gate[ct][i][2], og = e_gate[ct][i][3];
            const float dh = e_dh[ct][i] + acc[ct][i];
            const float tc = tanhf(e_ct[ct][i]);
            const float d_o = dh * tc;
            const float dct = e_dc[ct][i] + dh * og * (1.f - tc * tc);
            const float cp = first ? 0.f : e_cp[ct][i];
            float* dg = dgates + row * G + u;
            dg[0] = dct * gg * ig * (1.f - ig);
            dg[H] = dct * cp * fg * (1.f - fg);
            dg[2 * H] = dct * ig * (1.f - gg * gg);
            dg[3 * H] = d_o * og * (1.f - og);
            dc[idx] = dct * fg;
        }
}

#ifndef FSN_BPTT_SPLIT16_TILES
#define FSN_BPTT_SPLIT16_TILES 8  // measured (round 6, 5 tiles x 512 units, Fast FullSubNet's decoder at batch 72): see DESIGN 7.4
#endif
int fsn_launch_bptt_step(const float* dh_out, const float* dgates_next, const float* whhT_p, float* dc,
                         const float* gates, const float* c_t, const float* c_prev, float* dgates, int row_tiles, int H,
                         int last, int first, hipStream_t s) {
    // measured at 129 row tiles (tools/bench_train.py): 2 x 2 72.2 ms per training step, 1 x 2 73.0, 2 x 1 75.1,
    // 1 x 1 76.3, 4 x 2 76.8, 2 x 4 77.8, 4 x 4 89.0; a no-split-K form (a wave per tile for the whole K = 4H
    // range, which pays off in the forward step) is 10 % slower here: K is four times longer
    // measured (tools/bench_train.py): 256 row tiles 103.3 -> 96.8 ms per training step; at 128 tiles (one workgroup
    // per CU, nothing left to overlap its element-wise part with) 57.9 against 57.0 for the split-K form below
    if (row_tiles >= 192 && row_tiles % 4 == 0 && H % 48 == 0) {
        hipLaunchKernelGGL(bptt_step_cu_kernel<3>, dim3(H / 48, row_tiles / 4), dim3(256), 0, s, dh_out, dgates_next,
                           whhT_p, dc, gates, c_t, c_prev, dgates, H, last, first);
        return fsn_check_launch("bptt_step_cu_kernel");
    }
    const int cfg = row_tiles >= 64 && H % 32 == 0 ? 22 : 11;
#define FSN_BPTT_CASE(R, C)                                                                                        \
    hipLaunchKernelGGL((bptt_step_kernel<R, C>), dim3(H / 16 / C, (row_tiles + R - 1) / R), dim3(256), 0, s, dh_out, \
                       dgates_next, whhT_p, dc, gates, c_t, c_prev, dgates, row_tiles, H, last, first)
    if (cfg == 22) FSN_BPTT_CASE(2, 2);
    else if (row_tiles <= FSN_BPTT_SPLIT16_TILES && (4 * H / 16) % 16 == 0)  // a handful of rows (full-band model, the sibling models' blocks): 16-way split-K
        hipLaunchKernelGGL((bptt_step_kernel<1, 1, 16>), dim3(H / 16, row_tiles), dim3(1024), 0, s, dh_out, dgates_next,
                           whhT_p, dc, gates, c_t, c_prev, dgates, row_tiles, H, last, first);
    else FSN_BPTT_CASE(1, 1);
#undef FSN_BPTT_CASE
    return fsn_check_launch("bptt_step_kernel");
}

int fsn_launch_bptt_elem(const float* dh_out, const float* dh_rec, float* dc, const float* gates, const float* c_t,
                         const float* c_prev, float* dgates, long n_elems, int H, int last, int first, hipStream_t s) {
    hipLaunchKernelGGL(bptt_elem_kernel, dim3((unsigned)((n_elems + 255) / 256)), dim3(256), 0, s, dh_out, dh_rec, dc,
                       gates, c_t, c_prev, dgates, n_elems, H, last, first);
    return fsn_check_launch("bptt_elem_kernel");
}
