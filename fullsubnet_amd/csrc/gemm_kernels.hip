// fp32 MFMA GEMM family: C[R, Nout] = A[R, K] * W[Nout, K]^T (+ bias, epilogue).
//
// Used for every non-recurrent contraction of the path (sequence_model.py:116-123):
//   * LSTM input projections  W_ih x_t + (b_ih + b_hh)  for all (t, n) at once,
//   * the nn.Linear output layers (+ReLU for the full-band model).
//
// v_mfma_f32_16x16x4_f32 is exact fp32 at 64 FLOP/clk/SIMD; it needs ONE A and ONE B dword per lane
// per 32-cycle instruction, so operands are fed straight from global memory / L2 as 16-byte
// fragments (lane (i, q) reads 4 consecutive k of row i) without an LDS stage: at a 64x64 wave
// tile the kernel needs ~16 B/clk/CU from L1/L2, a quarter of what the vector memory path gives.
// Both operands are K-contiguous, so nn.LSTM's own [4H, K] weight layout is already the B layout;
// weights are re-tiled once ("packed") so that a wave's B fragment is one contiguous 1 KB line
// group.  K order inside a 16-chunk is permuted identically for A and B (j-th MFMA of a chunk
// contracts k = 16 kc + 4 q + j), which is free because fp32 addition order is ours to choose.
//
// The A operand is produced on the fly for the two model inputs, so the 387 MB freq_unfold tensor
// and the 400 MB concatenated / normalised sub-band input of the reference
// (base_model.py:31-44, fullsubnet/model.py:110-111) are never materialised.
#include "fsn_common.h"

#ifndef FSN_GEMM_ABLATE
#define FSN_GEMM_ABLATE 0  // probe-only: 1 = no operand loads in the K loop, 2 = no MFMAs
#endif

namespace {

__device__ __forceinline__ int reflect_idx(int j, int F) {
    j = j < 0 ? -j : j;
    return j >= F ? 2 * (F - 1) - j : j;
}

// ------------------------------------------------------------------------------------------
// A-operand providers.  prepare() resolves the row once; load() returns 4 consecutive k.
// ------------------------------------------------------------------------------------------
template <int KIND>
struct ARow;

template <>
struct ARow<0> {  // row-major matrix [R][ld]
    const float* base;
    __device__ __forceinline__ void prepare(const FsnGemmA& a, long row, long nrows) {
        const long lim = a.N > 0 ? (long)a.N : nrows;  // a.N: valid rows when not a multiple of 16
        row = row < lim ? row : lim - 1;  // clamped rows are computed and discarded
        base = a.p0 + row * a.ld;
    }
    __device__ __forceinline__ f32x4 load(const FsnGemmA&, int k0) const {
        return *reinterpret_cast<const f32x4*>(base + k0);
    }
};

template <>
struct ARow<3> {  // A already in fragment order: [rtile][kchunk][lane][4] (what lstm_rec_kernel emits)
    const float* base;
    long kstride;
    __device__ __forceinline__ void prepare(const FsnGemmA& a, long row, long nrows) {
        row = row < nrows ? row : nrows - 1;
        // a.ld = K (floats per row); tile = row / 16, lane-in-tile handled by the caller's lane id
        base = a.p0 + (row >> 4) * (a.ld * 16) + (threadIdx.x & 63) * 4;
    }
    __device__ __forceinline__ f32x4 load(const FsnGemmA&, int k0) const {
        return *reinterpret_cast<const f32x4*>(base + (long)(k0 >> 4) * 256);
    }
};

template <>
struct ARow<1> {  // full-band model input: row (t, b) = mag[b][t][:] / den   (fullsubnet/model.py:92-94)
    const float* base;
    float den;
    bool valid;
    __device__ __forceinline__ void prepare(const FsnGemmA& a, long row, long nrows) {
        const int t = (int)(row / a.Npad), b = (int)(row % a.Npad);
        valid = row < nrows && b < a.B;
        const int bb = valid ? b : 0, tt = valid ? t : 0;
        base = a.p0 + ((long)bb * a.Tp + tt) * a.FP;
        den = a.den[a.den_mode ? (long)bb * a.Tp + tt : bb];
    }
    __device__ __forceinline__ f32x4 load(const FsnGemmA&, int k0) const {
        f32x4 v = *reinterpret_cast<const f32x4*>(base + k0);  // columns >= F are stored as zeros
        if (!valid) return f32x4{0.f, 0.f, 0.f, 0.f};
        return f32x4{v[0] / den, v[1] / den, v[2] / den, v[3] / den};
    }
};

template <>
struct ARow<2> {  // sub-band model input: row (t, n = b F + f), 2 nb + 2 channels
    // channel c < 2nb+1 : mag[b][t][reflect(f + c - nb)]      (base_model.py:31-44, N = nb)
    // channel 2nb+1     : fb_out[b][t][f]                      (model.py:98-101,110)
    // all divided by the sub-band norm divisor                 (model.py:111)
    const float* mrow;
    const float* frow;
    float den;
    int f;
    bool valid;
    __device__ __forceinline__ void prepare(const FsnGemmA& a, long row, long nrows) {
        const int t = (int)(row / a.Npad), n = a.n_offset + (int)(row % a.Npad);
        valid = row < nrows && n < a.N;
        const int nn = valid ? n : 0, tt = valid ? t : 0;
        const int b = nn / a.F;
        f = nn % a.F;
        mrow = a.p0 + ((long)b * a.Tp + tt) * a.FP;
        frow = a.p1 + ((long)b * a.Tp + tt) * a.FP;
        den = a.den[a.den_mode ? (long)tt * a.den_stride + nn : b];
    }
    __device__ __forceinline__ f32x4 load(const FsnGemmA& a, int k0) const {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = k0 + j;
            float x = 0.f;
            if (c <= 2 * a.nb) x = mrow[reflect_idx(f + c - a.nb, a.F)];
            else if (c == 2 * a.nb + 1) x = frow[f];
            v[j] = valid ? x / den : 0.f;
        }
        return v;
    }
};

// ------------------------------------------------------------------------------------------
// C stores.  acc register i of lane l is C[16 rtile + 4 (l>>4) + i][16 ctile + (l&15)].
// ------------------------------------------------------------------------------------------
template <int KIND>
__device__ __forceinline__ void store_tile(const FsnGemmC& c, f32x4 acc, float bias, long rtile, int ctile,
                                           int col_tiles, int lane) {
    const int col = ctile * 16 + (lane & 15);
    if (KIND == 0) {
        // fragment order: tile (rtile, ctile) is one contiguous 1 KB block [lane][reg]; this is the
        // accumulator-init layout of the recurrent kernels.
        f32x4 v = {acc[0] + bias, acc[1] + bias, acc[2] + bias, acc[3] + bias};
        *reinterpret_cast<f32x4*>(c.p0 + ((rtile * col_tiles + ctile) * 64 + lane) * 4) = v;
    } else if (KIND == 3) {  // plain row-major C (+ optional bias), used by the training-step GEMMs
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long row = rtile * 16 + 4 * (lane >> 4) + i;
            if (row < c.rows && col < c.cols) {
                const float v = acc[i] + bias;
                c.p0[row * c.ld + col] = (c.la && v < 0.f) ? 0.f : v;  // la doubles as the ReLU flag here
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long row = rtile * 16 + 4 * (lane >> 4) + i;
            const int t = (int)(row / c.Npad), n = (int)(row % c.Npad) + (KIND == 2 ? c.n_off : 0);
            if (KIND == 1) {  // full-band output layer: ReLU(h W^T + b) -> fb_out[b][t][f]
                if (n < c.B && col < c.FP) {
                    // ReLU as torch.relu: a NaN stays a NaN (fmaxf would turn it into 0 - and with it the poison that
                    // a full-band chain launch which ran out of time leaves in its output, fsn_launch_poison_if)
                    const float pre = acc[i] + bias;
                    const float v = col < c.F ? (pre < 0.f ? 0.f : pre) : 0.f;
                    c.p0[((long)n * c.Tp + t) * c.FP + col] = v;
                }
            } else {  // sub-band output layer: compressed cIRM planes, look-ahead frames dropped
                      // (model.py:128-135): crm_{r,i}[b][t - la][f]
                if (n < c.N && t >= c.la && col < 2) {
                    const int b = n / c.F, f = n % c.F;
                    float* dst = col == 0 ? c.p0 : c.p1;
                    dst[((long)b * c.T + (t - c.la)) * c.FP + f] = acc[i] + bias;
                }
            }
        }
    }
}

template <int AKIND, int CKIND, int RTW, int CTW, int WR, int WC, int PF>
__global__ __launch_bounds__(WR* WC * 64) void gemm_kernel(FsnGemmA a, const float* __restrict__ wp, FsnGemmC c,
                                                           int row_tiles, int col_tiles, int k_chunks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave / WC, wc = wave % WC;
    const unsigned ncb = (col_tiles + WC * CTW - 1) / (WC * CTW);
    const unsigned nrb = (unsigned)(((long)row_tiles + WR * RTW - 1) / (WR * RTW));
    const unsigned ntiles = nrb * ncb;
    const long nrows = (long)row_tiles * 16;
    const int kq = 4 * (lane >> 4);
    // Persistent workgroups: the grid is sized to fill the chip once (short-lived workgroups were
    // measured at ~half the residency the register budget allows), each workgroup walks over
    // output tiles.  XCD x (block b runs on XCD b % 8 - speed only) owns one contiguous range of
    // the tile list, rows-major with the column blocks of a row block adjacent, so the workgroups
    // of one XCD work on neighbouring tiles at the same time and share A rows / B panels in its L2.
#ifdef FSN_GEMM_PLACEMENT
    if (threadIdx.x == 0) {
        FSN_GEMM_PLACEMENT[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_ID
        FSN_GEMM_PLACEMENT[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // XCC_ID
    }
#endif
    const unsigned xcd = blockIdx.x & 7u, lid = blockIdx.x >> 3, lstride = (gridDim.x + 7u - xcd) >> 3;
    const unsigned tq = ntiles >> 3, tr = ntiles & 7u;
    const unsigned tbeg = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const unsigned tcnt = tq + (xcd < tr ? 1u : 0u);
    for (unsigned ti = lid; ti < tcnt; ti += lstride) {
    const unsigned v = tbeg + ti;
    const unsigned rb = v / ncb, cb = v % ncb;
    const long rtile0 = ((long)rb * WR + wr) * RTW;
    const int ctile0 = ((int)cb * WC + wc) * CTW;

    ARow<AKIND> arow[RTW];
#pragma unroll
    for (int rt = 0; rt < RTW; ++rt) arow[rt].prepare(a, (rtile0 + rt) * 16 + (lane & 15), nrows);
    const float* bptr[CTW];
    float biasv[CTW];
#pragma unroll
    for (int ct = 0; ct < CTW; ++ct) {
        int ctile = ctile0 + ct;
        ctile = ctile < col_tiles ? ctile : col_tiles - 1;
        bptr[ct] = wp + ((long)ctile * k_chunks * 64 + lane) * 4;
        biasv[ct] = c.bias ? c.bias[ctile * 16 + (lane & 15)] : 0.f;
    }

    f32x4 acc[RTW][CTW];
#pragma unroll
    for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Register ring of PF operand chunks.  The steady-state loop is branch-free (tail loads are
    // clamped to the last chunk and simply re-read it) so that the compiler can count its own
    // outstanding loads: it then waits with vmcnt(8 (PF-1)) instead of draining the queue before
    // every chunk.  Chunk kc + PF is requested right after chunk kc has been consumed.
    f32x4 abuf[PF][RTW], bbuf[PF][CTW];
    const int last = k_chunks - 1;
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        const int kc = p < last ? p : last;
#pragma unroll
        for (int rt = 0; rt < RTW; ++rt) abuf[p][rt] = arow[rt].load(a, kc * 16 + kq);
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct) bbuf[p][ct] = *reinterpret_cast<const f32x4*>(bptr[ct] + (long)kc * 256);
    }
    const int k_main = (k_chunks / PF) * PF;
    for (int kc0 = 0; kc0 < k_main; kc0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
#if FSN_GEMM_ABLATE != 2
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
                    for (int ct = 0; ct < CTW; ++ct)
                        acc[rt][ct] = mfma16(abuf[p][rt][j], bbuf[p][ct][j], acc[rt][ct]);
#else
#pragma unroll
            for (int rt = 0; rt < RTW; ++rt) asm volatile("" ::"v"(abuf[p][rt]));
#pragma unroll
            for (int ct = 0; ct < CTW; ++ct) asm volatile("" ::"v"(bbuf[p][ct]));
#endif
#if FSN_GEMM_ABLATE != 1
            // pin the refill of slot p right here: hipcc otherwise sinks these loads down to their
            // first use (one ring turn later), which turns the ring into load -> wait -> use
            __builtin_amdgcn_sched_barrier(0);
            int kn = kc0 + p + PF;
            kn = kn < last ? kn : last;
#pragma unroll
            for (int rt = 0; rt < RTW; ++rt) abuf[p][rt] = arow[rt].load(a, kn * 16 + kq);
#pragma unroll
            for (int ct = 0; ct < CTW; ++ct) bbuf[p][ct] = *reinterpret_cast<const f32x4*>(bptr[ct] + (long)kn * 256);
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
    }
    // remainder chunks (k_chunks % PF of them) are already sitting in the ring, in order
#pragma unroll
    for (int p = 0; p < PF - 1; ++p) {
        if (k_main + p < k_chunks) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
                    for (int ct = 0; ct < CTW; ++ct)
                        acc[rt][ct] = mfma16(abuf[p][rt][j], bbuf[p][ct][j], acc[rt][ct]);
        }
    }

#pragma unroll
    for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct)
            if (rtile0 + rt < row_tiles && ctile0 + ct < col_tiles)
                store_tile<CKIND>(c, acc[rt][ct], biasv[ct], rtile0 + rt, ctile0 + ct, col_tiles, lane);
    }  // tile loop
}

// W [n_out][k] (nn.LSTM / nn.Linear layout) -> B-fragment order [n_out_pad/16][k_pad/16][64][4]
__global__ void pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int n_out, int k, int ctiles,
                            int kchunks, int transposed, int ldw) {
    const long total = (long)ctiles * kchunks * 256;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 3), lane = (int)((i >> 2) & 63);
        const long blk = i >> 8;
        const int kc = (int)(blk % kchunks), ct = (int)(blk / kchunks);
        const int row = ct * 16 + (lane & 15), col = kc * 16 + 4 * (lane >> 4) + j;
        // transposed: the source holds W^T, i.e. element (row, col) of W sits at w[col][row]
        wp[i] = (row < n_out && col < k) ? (transposed ? w[(long)col * ldw + row] : w[(long)row * k + col]) : 0.f;
    }
}

__global__ void bias_sum_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                int n, int n_pad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_pad) out[i] = i < n ? (b ? a[i] + b[i] : a[i]) : 0.f;
}

// Launch geometry.  Measured on MI355X (tools/probe_gemm.hip, profiles/r01_gemm_probe.md): two waves
// that both stream MFMAs on one SIMD alternate instruction by instruction and lose ~45 % of the
// matrix pipe, while ONE wave per SIMD with a large register tile (4x8 tiles = 64x128 outputs, 128
// accumulator VGPRs + a 2-deep operand ring) sustains ~140 TFLOP/s (90 % of the fp32 MFMA peak).
// So the big GEMMs run as exactly one 4-wave workgroup per CU (enforced with an LDS reservation
// the kernel never touches), persistent over the tile list.
template <int AKIND, int CKIND, int RTW, int CTW, int WR, int WC, int PF = 2>
int launch(const FsnGemmA& a, const float* wp, const FsnGemmC& c, int row_tiles, int col_tiles, int k_chunks,
           hipStream_t s, int wg_per_cu = 0, size_t lds_reserve = 0) {
    const long nrb = ((long)row_tiles + WR * RTW - 1) / (WR * RTW);
    const long ncb = (col_tiles + WC * CTW - 1) / (WC * CTW);
    auto kern = gemm_kernel<AKIND, CKIND, RTW, CTW, WR, WC, PF>;
    static int occ = 0;  // per instantiation
    if (occ == 0) {
        int n = 0;
        if (lds_reserve > 64 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_reserve) != hipSuccess) {
            fsn_set_error("gemm: cannot reserve %zu bytes of LDS", lds_reserve);
            return FSN_ERR_LAUNCH;
        }
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, WR * WC * 64, lds_reserve) != hipSuccess || n < 1)
            n = 1;
        occ = n;
    }
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    long grid = (long)cus * (wg_per_cu > 0 ? wg_per_cu : occ);
    if (grid > nrb * ncb) grid = nrb * ncb;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WR * WC * 64), lds_reserve, s, a, wp, c, row_tiles,
                       col_tiles, k_chunks);
    return fsn_check_launch("gemm_kernel");
}

constexpr size_t kOnePerCu = 96 * 1024;  // > 160 KB / 2: at most one such workgroup fits a CU

}  // namespace

int fsn_launch_gemm(const FsnGemmA& a, const float* wp, const FsnGemmC& c, int row_tiles, int col_tiles,
                    int k_chunks, hipStream_t s) {
    if (a.kind == 2 && c.kind == 0)
        return launch<2, 0, 4, 8, 2, 2>(a, wp, c, row_tiles, col_tiles, k_chunks, s, 1, kOnePerCu);
    if (a.kind == 1 && c.kind == 0)
        return launch<1, 0, 4, 8, 2, 2>(a, wp, c, row_tiles, col_tiles, k_chunks, s, 1, kOnePerCu);
    if (a.kind == 0 && c.kind == 0)
        return launch<0, 0, 4, 8, 2, 2>(a, wp, c, row_tiles, col_tiles, k_chunks, s, 1, kOnePerCu);
    if (a.kind == 0 && c.kind == 1)
        return launch<0, 1, 4, 8, 2, 2>(a, wp, c, row_tiles, col_tiles, k_chunks, s, 1, kOnePerCu);
    // 2-column output layer: HBM-bound on reading the hidden sequence -> many waves in flight
    if (a.kind == 0 && c.kind == 2) return launch<0, 2, 4, 1, 4, 1>(a, wp, c, row_tiles, col_tiles, k_chunks, s);
    // training-step GEMMs (row-major C).  All-steps GEMMs (dX, output layers): the one-workgroup-per-CU
    // shape with a 256 x 128 tile (sb hidden 384 = 3 column blocks); per-step ones: few row tiles per
    // launch -> small workgroup tiles, many of them.
    if (a.kind == 0 && c.kind == 3) {
        if (row_tiles >= 2048 && col_tiles >= 4)
            return launch<0, 3, 8, 4, 2, 2>(a, wp, c, row_tiles, col_tiles, k_chunks, s, 1, kOnePerCu);
        // a one-tile-wide output layer over many rows is HBM-bound on reading its input: many waves in flight
        if (row_tiles >= 2048 && col_tiles == 1) return launch<0, 3, 4, 1, 4, 1>(a, wp, c, row_tiles, col_tiles, k_chunks, s);
        // two tiles wide over many rows (the sub-band model's input gradient: 402 k rows x 32 columns, K = 1536, 2.5 GB of
        // gate gradients to read): every wave owns both column tiles of four row tiles - as 2 x 2 waves of 2 x 2 tiles half
        // of the workgroup had no columns: 1.08 -> 0.63 ms (deeper prefetch or 2 row tiles per wave: no better)
        if (row_tiles >= 2048 && col_tiles == 2) return launch<0, 3, 4, 2, 4, 1>(a, wp, c, row_tiles, col_tiles, k_chunks, s);
        return launch<0, 3, 2, 2, 2, 2>(a, wp, c, row_tiles, col_tiles, k_chunks, s);
    }
    fsn_set_error("fsn_launch_gemm: unsupported operand kinds A=%d C=%d", a.kind, c.kind);
    return FSN_ERR_ARG;
}

int fsn_launch_pack(const float* w, float* wp, int n_out, int k, int n_out_pad, int k_pad, hipStream_t s,
                    int transposed, int ldw) {
    const int ctiles = n_out_pad / 16, kchunks = k_pad / 16;
    const long total = (long)ctiles * kchunks * 256;
    const unsigned grid = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_kernel, dim3(grid), dim3(256), 0, s, w, wp, n_out, k, ctiles, kchunks, transposed, ldw);
    return fsn_check_launch("pack_kernel");
}

// bias[n] -> accumulator-fragment tiles [n/16][64][4]: lane l of column tile ct holds bias[16 ct + (l & 15)] x 4
__global__ void bias_frag_kernel(const float* __restrict__ bias, float* __restrict__ frag, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n * 16) frag[i] = bias[(i >> 8) * 16 + ((i >> 2) & 15)];
}

int fsn_launch_bias_frag(const float* bias, float* frag, int n, hipStream_t s) {
    hipLaunchKernelGGL(bias_frag_kernel, dim3((n * 16 + 255) / 256), dim3(256), 0, s, bias, frag, n);
    return fsn_check_launch("bias_frag_kernel");
}

int fsn_launch_bias_sum(const float* a, const float* b, float* out, int n, int n_pad, hipStream_t s) {
    hipLaunchKernelGGL(bias_sum_kernel, dim3((n_pad + 255) / 256), dim3(256), 0, s, a, b, out, n, n_pad);
    return fsn_check_launch("bias_sum_kernel");
}

// ---- nn.Linear with a handful of outputs (the sub-band output layer: 384 -> 2, sequence_model.py:82-84) ------------------
// As a GEMM it is 16 padded columns for 2 real ones and runs at a third of the memory bandwidth it is bound by (0.37 ms
// for 618 MB at config 3's shape, in each direction).  Forward: one 16-lane group per row, lane p owns the 16-byte groups
// p, p + 16, ... of the row (one coalesced 256-byte segment per load), fmaf chains in k order, a fixed butterfly.
// dX = dY W: every thread writes one 16-byte group of a row.
namespace {
template <int O>
__global__ __launch_bounds__(256) void linear_small_out_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ w,
                                                               const float* __restrict__ b, float* __restrict__ y, long R, int I,
                                                               int relu) {
    const long r = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int p = threadIdx.x & 15;
    float acc[O];
#pragma unroll
    for (int o = 0; o < O; ++o) acc[o] = 0.f;
    if (r < R) {
        const float* xr = x + r * ldx;
        for (int k = 4 * p; k < I; k += 64) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + k);
#pragma unroll
            for (int o = 0; o < O; ++o) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (long)o * I + k);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[o] = fmaf(xv[j], wv[j], acc[o]);
            }
        }
    }
#pragma unroll
    for (int o = 0; o < O; ++o) {
        acc[o] += __shfl_xor(acc[o], 1, 64);
        acc[o] += __shfl_xor(acc[o], 2, 64);
        acc[o] += __shfl_xor(acc[o], 4, 64);
        acc[o] += __shfl_xor(acc[o], 8, 64);
    }
    if (p == 0 && r < R) {
#pragma unroll
        for (int o = 0; o < O; ++o) {
            const float v = acc[o] + b[o];
            y[r * O + o] = relu && v < 0.f ? 0.f : v;
        }
    }
}
template <int O>
__global__ __launch_bounds__(256) void linear_small_dx_kernel(const float* __restrict__ dy, long lddy, const float* __restrict__ w,
                                                              float* __restrict__ dx, long lddx, long R, int I) {
    const int groups = I >> 2;  // 16-byte groups per row
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * groups) return;
    const long r = i / groups;
    const int k = (int)(i - r * groups) * 4;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int o = 0; o < O; ++o) {
        const float d = dy[r * lddy + o];
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (long)o * I + k);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = fmaf(d, wv[j], acc[j]);
    }
    *reinterpret_cast<f32x4*>(dx + r * lddx + k) = acc;
}
}  // namespace

bool fsn_linear_small_out_ok(int I, int O, long ldx) { return O >= 1 && O <= 4 && I % 64 == 0 && ldx % 4 == 0; }
int fsn_launch_linear_small_out(const float* x, long ldx, const float* w, const float* b, float* y, long R, int I, int O,
                                int relu, hipStream_t s) {
    const dim3 grid((unsigned)((R + 15) / 16)), block(256);
    switch (O) {
    case 1: hipLaunchKernelGGL(linear_small_out_kernel<1>, grid, block, 0, s, x, ldx, w, b, y, R, I, relu); break;
    case 2: hipLaunchKernelGGL(linear_small_out_kernel<2>, grid, block, 0, s, x, ldx, w, b, y, R, I, relu); break;
    case 3: hipLaunchKernelGGL(linear_small_out_kernel<3>, grid, block, 0, s, x, ldx, w, b, y, R, I, relu); break;
    case 4: hipLaunchKernelGGL(linear_small_out_kernel<4>, grid, block, 0, s, x, ldx, w, b, y, R, I, relu); break;
    default: fsn_set_error("linear_small_out: O = %d", O); return FSN_ERR_ARG;
    }
    return fsn_check_launch("linear_small_out_kernel");
}
int fsn_launch_linear_small_dx(const float* dy, long lddy, const float* w, float* dx, long lddx, long R, int I, int O,
                               hipStream_t s) {
    const long n = R * (I >> 2);
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    switch (O) {
    case 1: hipLaunchKernelGGL(linear_small_dx_kernel<1>, grid, block, 0, s, dy, lddy, w, dx, lddx, R, I); break;
    case 2: hipLaunchKernelGGL(linear_small_dx_kernel<2>, grid, block, 0, s, dy, lddy, w, dx, lddx, R, I); break;
    case 3: hipLaunchKernelGGL(linear_small_dx_kernel<3>, grid, block, 0, s, dy, lddy, w, dx, lddx, R, I); break;
    case 4: hipLaunchKernelGGL(linear_small_dx_kernel<4>, grid, block, 0, s, dy, lddy, w, dx, lddx, R, I); break;
    default: fsn_set_error("linear_small_dx: O = %d", O); return FSN_ERR_ARG;
    }
    return fsn_check_launch("linear_small_dx_kernel");
}
