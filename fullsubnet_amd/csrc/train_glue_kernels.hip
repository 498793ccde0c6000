// The glue of the FullSubNet TRAINING graph as hand-written kernels (recipes/dns_interspeech_2020/fullsubnet/model.py:72-136
// under autograd, fullsubnet/trainer.py:41-71): everything between the transforms, the four LSTM layers, the two output
// layers and the loss that round 3 still ran as ATen tensor algebra -
//   * look-ahead pad + Laplace norm of the full-band input, written time-major for the LSTM entries
//     (model.py:85-95; offline: audio_zen/model/base_model.py:204-218, cumulative: :221-251);
//   * the sub-band model's input: freq_unfold (reflect-padded neighbours, base_model.py:14-46) ++ full-band output, the
//     norm over the unfolded tensor - offline: ONE mean per utterance, taken analytically from per-bin sums (the 31-fold
//     unfolded tensor is never formed); cumulative (the shipped train_cumulativeLaplaceNorm.toml): every
//     unit (b, f) is its own "sample" whose 2 nb + 2 rows are the "frequencies", a running mean over the frames so far -
//     restricted to the rows drop_band keeps (audio_zen/acoustics/feature.py:309-345) - forward and backward (the
//     gradient reaches the full-band output directly and through the mean / the running means of all later frames);
//   * the mask's reshape / look-ahead slice (model.py:129-135) and its gradient;
//   * the training target: complex ideal ratio mask, compressed, band-dropped like the prediction
//     (audio_zen/acoustics/mask.py:7-44, trainer.py:51-53).
// HBM-bound gathers and reductions (a few MB per step): coalesced along the frequency axis of time-major tensors,
// statistics in fp64 in a fixed order (bit-reproducible).  Row order of the sub-band tensors: drop_band's - group i holds
// samples i, i + g, ... at bins i, i + g, ... (< F - F % g), the groups concatenated along the batch axis.
#include "fsn_common.h"

namespace {

struct TrDims {
    int B, F, T, la, nb, g;   // g = 1: no band dropping (B == 1 or num_groups <= 1)
    int Tp, Fd, Fs, R;
    int cum;                  // 1: cumulative_laplace_norm (divisors per frame), 0: offline_laplace_norm (per utterance)
};
__host__ __device__ inline TrDims tr_dims(const fsn_train_dims* d) {
    TrDims t;
    t.B = d->B, t.F = d->F, t.T = d->T, t.la = d->look_ahead, t.nb = d->nb;
    t.g = (d->B > 1 && d->groups > 1) ? d->groups : 1;
    t.Tp = t.T + t.la;
    t.Fd = t.F - t.F % t.g;
    t.Fs = t.g > 1 ? t.Fd / t.g : t.F;
    t.R = t.B * t.Fs;
    t.cum = d->norm == FSN_NORM_CUMULATIVE_LAPLACE;
    return t;
}
// row of the band-dropped (b_out, fs) space -> (sample b, bin f)
__device__ __forceinline__ void tr_row_bf(const TrDims& d, int r, int& b, int& f) {
    const int bo = r / d.Fs, fs = r % d.Fs;
    if (d.g == 1) {
        b = bo, f = fs;
        return;
    }
    int i = 0, off = 0;
    for (; i < d.g; ++i) {
        const int n = (d.B - i + d.g - 1) / d.g;
        if (bo < off + n) break;
        off += n;
    }
    b = i + d.g * (bo - off);
    f = i + d.g * fs;
}
// (sample b, bin f) -> its row, or -1 when drop_band leaves it out
__device__ __forceinline__ int tr_bf_row(const TrDims& d, int b, int f) {
    if (d.g == 1) return b * d.Fs + f;
    const int i = b % d.g;
    if (f >= d.Fd || f % d.g != i) return -1;
    int off = 0;
    for (int k = 0; k < i; ++k) off += (d.B - k + d.g - 1) / d.g;
    return (off + b / d.g) * d.Fs + f / d.g;
}
__device__ __forceinline__ int tr_reflect(int j, int F) {
    j = j < 0 ? -j : j;
    return j >= F ? 2 * (F - 1) - j : j;
}
__device__ __forceinline__ double tr_block_sum(double v, double* sh) {  // 256 threads, valid in thread 0
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
    __syncthreads();
    return t;
}

// rowsum[b F + f] = sum_t mag[b][f][t]  (one wave per row)
__global__ __launch_bounds__(256) void tr_rowsum_kernel(const float* __restrict__ mag, double* __restrict__ rowsum, int rows, int T) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    double acc = 0.0;
    for (int t = lane; t < T; t += 64) acc += (double)mag[(size_t)row * T + t];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (lane == 0) rowsum[row] = acc;
}
// total[b] = sum_f rowsum[b][f]
__global__ __launch_bounds__(256) void tr_total_kernel(const double* __restrict__ rowsum, double* __restrict__ total, int F) {
    __shared__ double sh[4];
    double acc = 0.0;
    for (int f = threadIdx.x; f < F; f += 256) acc += rowsum[(size_t)blockIdx.x * F + f];
    const double t = tr_block_sum(acc, sh);
    if (threadIdx.x == 0) total[blockIdx.x] = t;
}
// x_tm[t][b][f] = pad(mag)[b][f][t] / (mean_b + 1e-5), mag_tm[t][b][f] = pad(mag)[b][f][t]; zero beyond (B, F); one block
// per (t, 32-bin slab, b): a 32 x 32 tile through LDS so that both sides are coalesced
// cumulative_laplace_norm of the padded full-band input (base_model.py:221-251): colsum[b][t] = sum_f mag[b][f][t]
// (a block = 64 frames x 4 bin groups: threads along t, coalesced; the four partial sums meet in a fixed order), then
// cden[b][t] = (sum_{tau <= t} colsum[b][tau]) / (F (t + 1)) + EPSILON; the look-ahead frames are zeros that still count.
__global__ __launch_bounds__(256) void tr_fb_colsum_kernel(const float* __restrict__ mag, double* __restrict__ colsum, TrDims d) {
    __shared__ double part[4][64];
    const int b = blockIdx.y, t = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    double acc = 0.0;
    if (t < d.T)
        for (int f = q; f < d.F; f += 4) acc += (double)mag[((size_t)b * d.F + f) * d.T + t];
    part[q][threadIdx.x & 63] = acc;
    __syncthreads();
    if (q == 0 && t < d.Tp) colsum[(size_t)b * d.Tp + t] = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x];
}
// one wave per utterance: a scan over the frames in chunks of 64 (Hillis-Steele inside the chunk, fp64, fixed order)
__global__ __launch_bounds__(64) void tr_fb_cum_den_kernel(const double* __restrict__ colsum, float* __restrict__ cden, TrDims d) {
    const int b = blockIdx.x, lane = threadIdx.x;
    double carry = 0.0;
    for (int t0 = 0; t0 < d.Tp; t0 += 64) {
        const int t = t0 + lane;
        double v = t < d.Tp ? colsum[(size_t)b * d.Tp + t] : 0.0;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double u = __shfl_up(v, o, 64);
            if (lane >= o) v += u;
        }
        v += carry;
        if (t < d.Tp) cden[(size_t)b * d.Tp + t] = (float)(v / ((double)d.F * (t + 1))) + 1.1920928955078125e-07f;
        carry = __shfl(v, 63, 64);
    }
}
__global__ __launch_bounds__(256) void tr_fb_input_kernel(const float* __restrict__ mag, const double* __restrict__ total,
                                                         const float* __restrict__ cden, float* __restrict__ x_tm,
                                                         float* __restrict__ mag_tm, TrDims d, int Bp, int Fp) {
    __shared__ float tile[32][33];
    const int t0 = blockIdx.x * 32, f0 = blockIdx.y * 32, b = blockIdx.z;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {  // rows of the tile = bins, columns = frames (contiguous in mag)
        const int f = f0 + i, t = t0 + tx;
        tile[i][tx] = (b < d.B && f < d.F && t < d.T) ? mag[((size_t)b * d.F + f) * d.T + t] : 0.f;
    }
    __syncthreads();
    float den = 1.f;
    if (b < d.B && !d.cum) den = (float)(total[b] / ((double)d.F * d.Tp)) + 1e-5f;
    for (int i = ty; i < 32; i += 8) {  // rows = frames, columns = bins (contiguous in the outputs)
        const int t = t0 + i, f = f0 + tx;
        if (t < d.Tp && f < Fp) {
            const float v = tile[tx][i];
            if (d.cum && b < d.B) den = cden[(size_t)b * d.Tp + t];
            const size_t o = ((size_t)t * Bp + b) * Fp + f;
            mag_tm[o] = v;
            x_tm[o] = (b < d.B && f < d.F) ? v / den : 0.f;
        }
    }
}
// fbsum[b] = sum over (t, f) of fb_out_tm[t][b][f]: partial per (t, b), then per b (fixed order)
__global__ __launch_bounds__(256) void tr_fbsum_partial_kernel(const float* __restrict__ fb, long ld, double* __restrict__ partial,
                                                              TrDims d, int Bp) {
    __shared__ double sh[4];
    const int t = blockIdx.x, b = blockIdx.y;
    double acc = 0.0;
    for (int f = threadIdx.x; f < d.F; f += 256) acc += (double)fb[((size_t)t * Bp + b) * ld + f];
    const double s = tr_block_sum(acc, sh);
    if (threadIdx.x == 0) partial[(size_t)b * d.Tp + t] = s;
}
// den[b] = mean of the unfolded sub-band tensor + 1e-5: (sum_j mult[j] rowsum[b][j] + sum fb_out[b]) / (F (2 nb + 2) Tp),
// mult[j] = number of (unit f, window position) pairs that read bin j (reflection without edge repeat)
__global__ __launch_bounds__(256) void tr_sb_den_kernel(const double* __restrict__ rowsum, const double* __restrict__ partial,
                                                       float* __restrict__ den, TrDims d) {
    __shared__ double sh[4];
    const int b = blockIdx.x;
    double acc = 0.0;
    for (int j = threadIdx.x; j < d.F; j += 256) {
        int lo = j - d.nb, hi = j + d.nb;
        lo = lo < 0 ? 0 : lo;
        hi = hi > d.F - 1 ? d.F - 1 : hi;
        int m = hi - lo + 1;                                  // units whose window holds j directly
        if (j > 0 && d.nb - j >= 0) m += d.nb - j + 1;        // ... below bin 0, mirrored onto j: units f <= nb - j
        const int ju = d.F - 1 - j;
        if (ju > 0 && d.nb - ju >= 0) m += d.nb - ju + 1;     // ... above bin F - 1
        acc += (double)m * rowsum[(size_t)b * d.F + j];
    }
    for (int t = threadIdx.x; t < d.Tp; t += 256) acc += partial[(size_t)b * d.Tp + t];
    const double s = tr_block_sum(acc, sh);
    if (threadIdx.x == 0) den[b] = (float)(s / ((double)d.F * (2 * d.nb + 2) * d.Tp)) + 1e-5f;
}
// sb_in[t][r][c] (c < 32 = 2 nb + 2 padded to the LSTM entries' 32 columns; rows beyond R zero): 8 rows x 32 columns per wave-pass
__device__ __forceinline__ float tr_sb_raw(const float* __restrict__ mag_tm, const float* __restrict__ fb, long ld_fb, const TrDims& d,
                                           int Bp, int Fp, int t, int b, int f, int c) {
    return c < 2 * d.nb + 1 ? mag_tm[((size_t)t * Bp + b) * Fp + tr_reflect(f + c - d.nb, d.F)] : fb[((size_t)t * Bp + b) * ld_fb + f];
}
__device__ __forceinline__ double tr_sum32(double v) {  // over the 32 lanes that share a row (either half of the wave)
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// den: [B] (offline) or [Tp][Rp] (cumulative)
__global__ __launch_bounds__(256) void tr_sb_input_kernel(const float* __restrict__ mag_tm, const float* __restrict__ fb, long ld_fb,
                                                         const float* __restrict__ den, float* __restrict__ out, TrDims d, int Bp,
                                                         int Fp, int Rp) {
    const int t = blockIdx.y;
    const int c = threadIdx.x & 31;
    const int W = 2 * d.nb + 1;
    for (int r = blockIdx.x * 8 + (threadIdx.x >> 5); r < Rp; r += gridDim.x * 8) {
        float v = 0.f;
        if (r < d.R && c <= W) {
            int b, f;
            tr_row_bf(d, r, b, f);
            v = tr_sb_raw(mag_tm, fb, ld_fb, d, Bp, Fp, t, b, f, c) / (d.cum ? den[(size_t)t * Rp + r] : den[b]);
        }
        out[((size_t)t * Rp + r) * 32 + c] = v;
    }
}
// cumulative norm of the sub-band tensor, forward: S[t][r] = sum over the 2 nb + 2 columns of the raw row (fp64) ...
__global__ __launch_bounds__(256) void tr_sb_cum_sum_kernel(const float* __restrict__ mag_tm, const float* __restrict__ fb, long ld_fb,
                                                           double* __restrict__ S, TrDims d, int Bp, int Fp, int Rp) {
    const int t = blockIdx.y;
    const int c = threadIdx.x & 31;
    for (int r0 = blockIdx.x * 8; r0 < Rp; r0 += gridDim.x * 8) {  // (uniform trip count: the shuffles see whole waves)
        const int r = r0 + (threadIdx.x >> 5);
        double v = 0.0;
        if (r < d.R && c <= 2 * d.nb + 1) {
            int b, f;
            tr_row_bf(d, r, b, f);
            v = (double)tr_sb_raw(mag_tm, fb, ld_fb, d, Bp, Fp, t, b, f, c);
        }
        v = tr_sum32(v);
        if (c == 0 && r < Rp) S[(size_t)t * Rp + r] = v;
    }
}
// ... den[t][r] = (sum_{tau <= t} S[tau][r]) / ((2 nb + 2) (t + 1)) + EPSILON (base_model.py:230-251 with the units as samples:
// quirk Q4).  A block = 32 rows x 8 time segments: every thread sums its segment, the eight segment sums of a row meet in
// LDS, every thread walks its segment again from its prefix - a serial depth of 2 Tp / 8 instead of Tp; rows coalesced.
constexpr int TR_SEG = 8;
__global__ __launch_bounds__(256) void tr_sb_cum_scan_kernel(const double* __restrict__ S, float* __restrict__ den, TrDims d, int Rp) {
    __shared__ double seg[TR_SEG][32];
    const int rl = threadIdx.x & 31, q = threadIdx.x >> 5, r = blockIdx.x * 32 + rl;
    const int L = (d.Tp + TR_SEG - 1) / TR_SEG, ta = q * L, tb = ta + L < d.Tp ? ta + L : d.Tp;
    double run = 0.0;
    if (r < Rp)
        for (int t = ta; t < tb; ++t) run += S[(size_t)t * Rp + r];
    seg[q][rl] = run;
    __syncthreads();
    if (r >= Rp) return;
    run = 0.0;
    for (int k = 0; k < q; ++k) run += seg[k][rl];
    const double C = (double)(2 * d.nb + 2);
    for (int t = ta; t < tb; ++t) {
        run += S[(size_t)t * Rp + r];
        den[(size_t)t * Rp + r] = (float)(run / (C * (t + 1))) + 1.1920928955078125e-07f;
    }
}
// backward: y = raw / D, D[t] = (sum_{tau <= t} S[tau]) / (C (t + 1)) + eps  =>  d loss / d S[tau] = sum_{t >= tau} P[t] with
// P[t] = - (sum_c dy[t][c] y[t][c]) / D[t] / (C (t + 1)); every raw element of frame tau receives that on top of dy / D.
__global__ __launch_bounds__(256) void tr_sb_cum_bwd_p_kernel(const float* __restrict__ dx, const float* __restrict__ sb_in,
                                                             const float* __restrict__ den, double* __restrict__ P, TrDims d, int Rp) {
    // a thread = four columns of a row (16-byte loads: a wave reads 2 x 1 KB runs), eight threads per row, 32 rows per block
    const int t = blockIdx.y, c4 = (threadIdx.x & 7) * 4, r = blockIdx.x * 32 + (threadIdx.x >> 3);
    const int C = 2 * d.nb + 2;
    double v = 0.0;
    if (r < d.R) {
        const size_t i = ((size_t)t * Rp + r) * 32 + c4;
        const f32x4 a = *reinterpret_cast<const f32x4*>(dx + i), b = *reinterpret_cast<const f32x4*>(sb_in + i);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (c4 + k < C) v += (double)a[k] * (double)b[k];  // (the padding columns of dx are never written: not read into the sum)
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 7) == 0 && r < Rp)
        P[(size_t)t * Rp + r] = r < d.R ? -v / (double)den[(size_t)t * Rp + r] / ((double)C * (t + 1)) : 0.0;
}
__global__ __launch_bounds__(256) void tr_sb_cum_bwd_scan_kernel(const double* __restrict__ P, float* __restrict__ G, TrDims d, int Rp) {
    __shared__ double seg[TR_SEG][32];
    const int rl = threadIdx.x & 31, q = threadIdx.x >> 5, r = blockIdx.x * 32 + rl;
    const int L = (d.Tp + TR_SEG - 1) / TR_SEG, ta = q * L, tb = ta + L < d.Tp ? ta + L : d.Tp;
    double run = 0.0;
    if (r < Rp)
        for (int t = ta; t < tb; ++t) run += P[(size_t)t * Rp + r];
    seg[q][rl] = run;
    __syncthreads();
    if (r >= Rp) return;
    run = 0.0;
    for (int k = TR_SEG - 1; k > q; --k) run += seg[k][rl];  // everything behind this segment
    for (int t = tb - 1; t >= ta; --t) {
        run += P[(size_t)t * Rp + r];
        G[(size_t)t * Rp + r] = (float)run;
    }
}
// backward, pass 1: partial[bo][t] = sum over the rows of band-dropped sample bo and the columns of dx[t][r][c] sb_in[t][r][c]
__global__ __launch_bounds__(256) void tr_sb_bwd_partial_kernel(const float* __restrict__ dx, const float* __restrict__ sb_in,
                                                               double* __restrict__ partial, TrDims d, int Rp) {
    __shared__ double sh[4];
    const int t = blockIdx.x, bo = blockIdx.y;
    const size_t base = ((size_t)t * Rp + (size_t)bo * d.Fs) * 32;
    double acc = 0.0;
    const int C = 2 * d.nb + 2;  // columns beyond are padding: the dX product does not write them (uninitialised memory)
    for (int i = threadIdx.x; i < d.Fs * 32; i += 256)
        if ((i & 31) < C) acc += (double)dx[base + i] * (double)sb_in[base + i];
    const double s = tr_block_sum(acc, sh);
    if (threadIdx.x == 0) partial[(size_t)bo * d.Tp + t] = s;
}
// pass 2: dmu[b] = - (sum_t partial) / den[b]: every element's share of d loss / d mean times its 1 / count
__global__ __launch_bounds__(256) void tr_sb_bwd_dmu_kernel(const double* __restrict__ partial, const float* __restrict__ den,
                                                           float* __restrict__ dmu, TrDims d) {
    __shared__ double sh[4];
    const int bo = blockIdx.x;
    double acc = 0.0;
    for (int t = threadIdx.x; t < d.Tp; t += 256) acc += partial[(size_t)bo * d.Tp + t];
    const double s = tr_block_sum(acc, sh);
    if (threadIdx.x == 0) {
        int b, f;
        tr_row_bf(d, bo * d.Fs, b, f);
        dmu[b] = (float)(-s / (double)den[b] / ((double)d.F * (2 * d.nb + 2) * d.Tp));
    }
}
// pass 3: d fb_out[t][b][f] = [row kept] dx[t][r][2 nb + 1] / den[b] + dmu[b], through the ReLU of the full-band output
// layer (fb_out > 0), as the padded dy of fsn_linear_backward ([Tp Bp][ld_d], zeros beyond (B, F))
// (cumulative norm: a unit's divisors hang on its own rows only, so dropped units receive nothing; dmu = G [Tp][Rp])
__global__ __launch_bounds__(256) void tr_sb_bwd_dfb_kernel(const float* __restrict__ dx, const float* __restrict__ den,
                                                           const float* __restrict__ dmu, const float* __restrict__ fb, long ld_fb,
                                                           float* __restrict__ dfb, long ld_d, TrDims d, int Bp, int Rp) {
    const int t = blockIdx.y, b = blockIdx.z;
    for (int f = blockIdx.x * 256 + threadIdx.x; f < ld_d; f += gridDim.x * 256) {
        float v = 0.f;
        if (b < d.B && f < d.F && fb[((size_t)t * Bp + b) * ld_fb + f] > 0.f) {
            const int r = tr_bf_row(d, b, f);
            if (d.cum) {
                if (r >= 0) v = dx[((size_t)t * Rp + r) * 32 + 2 * d.nb + 1] / den[(size_t)t * Rp + r] + dmu[(size_t)t * Rp + r];
            } else {
                v = dmu[b];
                if (r >= 0) v += dx[((size_t)t * Rp + r) * 32 + 2 * d.nb + 1] / den[b];
            }
        }
        dfb[((size_t)t * Bp + b) * ld_d + f] = v;
    }
}
// mask[bo][c][fs][t] = y[t + la][bo Fs + fs][c]  (model.py:129-135): a 32 (rows) x 32 (frames) tile per block
__global__ __launch_bounds__(256) void tr_mask_out_kernel(const float* __restrict__ y, float* __restrict__ mask, TrDims d, int Rp) {
    __shared__ float tile[2][32][33];
    const int r0 = blockIdx.x * 32, t0 = blockIdx.y * 32;
    for (int i = threadIdx.x; i < 32 * 64; i += 256) {  // 64 floats per frame: 32 rows x 2 outputs, contiguous in y
        const int tt = i >> 6, rc = i & 63, r = r0 + (rc >> 1), t = t0 + tt;
        tile[rc & 1][rc >> 1][tt] = (r < d.R && t < d.T) ? y[((size_t)(t + d.la) * Rp + r) * 2 + (rc & 1)] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * 32 * 32; i += 256) {
        const int tt = i & 31, rr = (i >> 5) & 31, c = i >> 10, r = r0 + rr, t = t0 + tt;
        if (r < d.R && t < d.T) mask[(((size_t)(r / d.Fs) * 2 + c) * d.Fs + r % d.Fs) * d.T + t] = tile[c][rr][tt];
    }
}
// dy[t][r][c] (ld columns, zero beyond the two outputs, beyond R and for the look-ahead frames) = d mask[bo][c][fs][t - la]
__global__ __launch_bounds__(256) void tr_mask_grad_kernel(const float* __restrict__ dmask, float* __restrict__ dy, TrDims d, int Rp,
                                                          int ld) {
    const int t = blockIdx.y;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)Rp * ld; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / ld), c = (int)(i % ld);
        float v = 0.f;
        if (r < d.R && c < 2 && t >= d.la) v = dmask[(((size_t)(r / d.Fs) * 2 + c) * d.Fs + r % d.Fs) * d.T + (t - d.la)];
        dy[(size_t)t * Rp * ld + i] = v;
    }
}
// target[bo][c][fs][t] = compress(cIRM)[b][f][t][c]  (mask.py:7-44 + drop_band, in the layout of the prediction)
__device__ __forceinline__ float tr_compress(float m) {  // mask.py:32-44, K = 10, C = 0.1
    m = m <= -100.0f ? -100.0f : m;
    const float e = expf(-0.1f * m);
    return 10.0f * (1.0f - e) / (1.0f + e);
}
__global__ __launch_bounds__(256) void tr_target_kernel(const float* __restrict__ nr, const float* __restrict__ ni,
                                                       const float* __restrict__ cr, const float* __restrict__ ci,
                                                       float* __restrict__ target, TrDims d) {
    const float eps = 1.1920928955078125e-07f;  // audio_zen/constant.py:9
    const int r = blockIdx.y;
    int b, f;
    tr_row_bf(d, r, b, f);
    const size_t src = ((size_t)b * d.F + f) * d.T;
    const size_t dst = ((size_t)(r / d.Fs) * 2 * d.Fs + r % d.Fs) * d.T;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < d.T; t += gridDim.x * 256) {
        const float a = nr[src + t], bb = ni[src + t], c = cr[src + t], dd = ci[src + t];
        const float den = a * a + bb * bb + eps;
        target[dst + t] = tr_compress((a * c + bb * dd) / den);
        target[dst + (size_t)d.Fs * d.T + t] = tr_compress((a * dd - bb * c) / den);
    }
}
// Time-major rows [T][N][W] <-> `n` pieces [n][T][rows][W] of whole clusters (rows beyond N: zeros going in, dropped coming
// back): a batch of more sub-band rows than one persistent training launch holds runs piece by piece (train.py:
// lstm2_train_chunks).  W is a multiple of 2 (float2 moves).  TO_PIECES: rows -> pieces, else pieces -> rows.
template <bool TO_PIECES>
__global__ __launch_bounds__(256) void tr_pieces_kernel(const float* __restrict__ src, float* __restrict__ dst, int T, long N, int W2,
                                                        int rows, int n) {
    const long per_t = (long)n * rows * W2;  // float2 elements of one step across all pieces
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per_t; i += (long)gridDim.x * blockDim.x) {
        const int t = blockIdx.y;
        const long r = i / W2;              // padded row index = piece * rows + row
        const int w = (int)(i - r * W2);
        const long piece = r / rows, row = r - piece * rows;
        const long po = ((piece * T + t) * rows + row) * W2 + w, ro = ((long)t * N + r) * W2 + w;
        const float2* s2 = reinterpret_cast<const float2*>(src);
        float2* d2 = reinterpret_cast<float2*>(dst);
        if (TO_PIECES) d2[po] = r < N ? s2[ro] : float2{0.f, 0.f};
        else if (r < N) d2[ro] = s2[po];
    }
}

__global__ void tr_scale_kernel(const float* __restrict__ x, const float* __restrict__ s, float* __restrict__ y, size_t n) {
    const float k = *s;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = x[i] * k;
}

struct TrWs {
    double *rowsum, *total, *partial;
    float* dmu;
    // cumulative norm only
    double* rowframe;  // [Tp][Rcap]: S (forward) / P (backward)
    float* G;          // [Tp][Rcap]
    float* cden;       // [B][Tp]
};
static size_t tr_rcap(const TrDims& d) { return ((size_t)d.R + 63) / 64 * 64; }  // padded row counts up to this fit the workspace
static size_t tr_ws_bytes(const TrDims& d) {
    size_t n = ((size_t)d.B * d.F + d.B + (size_t)d.B * d.Tp) * sizeof(double) + (((size_t)d.B + 1) / 2 * 2) * sizeof(float) + 256;
    if (d.cum) n += (size_t)d.Tp * tr_rcap(d) * (sizeof(double) + sizeof(float)) + (size_t)d.B * d.Tp * sizeof(float);
    return n;
}
static TrWs tr_carve(const TrDims& d, void* ws) {
    TrWs w;
    w.rowsum = static_cast<double*>(ws);
    w.total = w.rowsum + (size_t)d.B * d.F;
    w.partial = w.total + d.B;
    w.dmu = reinterpret_cast<float*>(w.partial + (size_t)d.B * d.Tp);
    w.rowframe = reinterpret_cast<double*>(w.dmu + ((size_t)d.B + 1) / 2 * 2);
    w.G = reinterpret_cast<float*>(w.rowframe + (size_t)d.Tp * tr_rcap(d));
    w.cden = w.G + (size_t)d.Tp * tr_rcap(d);
    return w;
}
static bool tr_check(const fsn_train_dims* dd) {
    return dd && dd->B >= 1 && dd->F >= 2 && dd->T >= 1 && dd->look_ahead >= 0 && dd->nb >= 0 && 2 * dd->nb + 2 <= 32 && dd->nb < dd->F &&
           dd->groups >= 1 && (dd->norm == FSN_NORM_OFFLINE_LAPLACE || dd->norm == FSN_NORM_CUMULATIVE_LAPLACE);
}

}  // namespace

#define TR_REQUIRE_DIMS(dd) \
    FSN_REQUIRE(tr_check(dd), "train glue: bad dimensions (B, F >= 2, T >= 1, 2 nb + 2 <= 32, nb < F, groups >= 1, norm offline / cumulative Laplace)")

extern "C" int fsn_train_rows(const fsn_train_dims* dims, int* Fs, int* R) {
    TR_REQUIRE_DIMS(dims);
    const TrDims d = tr_dims(dims);
    if (Fs) *Fs = d.Fs;
    if (R) *R = d.R;
    return FSN_OK;
}
extern "C" size_t fsn_train_glue_workspace_bytes(const fsn_train_dims* dims) { return tr_check(dims) ? tr_ws_bytes(tr_dims(dims)) : 0; }
extern "C" size_t fsn_train_den_elems(const fsn_train_dims* dims, int Rp) {
    if (!tr_check(dims)) return 0;
    const TrDims d = tr_dims(dims);
    return d.cum ? (size_t)d.Tp * (size_t)Rp : (size_t)d.B;
}

extern "C" int fsn_train_fb_input(const fsn_train_dims* dims, const float* mag, float* x_tm, float* mag_tm, int Bp, int Fp,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    FsnCallScope scope(stream);
    TR_REQUIRE_DIMS(dims);
    const TrDims d = tr_dims(dims);
    FSN_REQUIRE(mag && x_tm && mag_tm && workspace && Bp >= d.B && Fp >= d.F, "train fb input: NULL pointer / padded sizes below (B, F)");
    if (workspace_bytes < tr_ws_bytes(d)) {
        fsn_set_error("train fb input: workspace too small");
        return FSN_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const TrWs w = tr_carve(d, workspace);
    if (d.cum) {
        hipLaunchKernelGGL(tr_fb_colsum_kernel, dim3((unsigned)((d.Tp + 63) / 64), (unsigned)d.B), dim3(256), 0, s, mag, w.partial, d);
        FSN_TRY_LAUNCH("tr_fb_colsum_kernel");
        hipLaunchKernelGGL(tr_fb_cum_den_kernel, dim3((unsigned)d.B), dim3(64), 0, s, w.partial, w.cden, d);
        FSN_TRY_LAUNCH("tr_fb_cum_den_kernel");
    } else {
        hipLaunchKernelGGL(tr_rowsum_kernel, dim3((unsigned)((d.B * d.F + 3) / 4)), dim3(256), 0, s, mag, w.rowsum, d.B * d.F, d.T);
        FSN_TRY_LAUNCH("tr_rowsum_kernel");
        hipLaunchKernelGGL(tr_total_kernel, dim3((unsigned)d.B), dim3(256), 0, s, w.rowsum, w.total, d.F);
        FSN_TRY_LAUNCH("tr_total_kernel");
    }
    hipLaunchKernelGGL(tr_fb_input_kernel, dim3((unsigned)((d.Tp + 31) / 32), (unsigned)((Fp + 31) / 32), (unsigned)Bp), dim3(256), 0, s,
                       mag, w.total, w.cden, x_tm, mag_tm, d, Bp, Fp);
    return fsn_check_launch("tr_fb_input_kernel");
}

extern "C" int fsn_train_sb_input(const fsn_train_dims* dims, const float* mag_tm, const float* fb_out_tm, long ld_fb, int Bp, int Fp,
                                  float* sb_in, int Rp, float* den, void* workspace, size_t workspace_bytes, void* stream) {
    FsnCallScope scope(stream);
    TR_REQUIRE_DIMS(dims);
    const TrDims d = tr_dims(dims);
    FSN_REQUIRE(mag_tm && fb_out_tm && sb_in && den && workspace && Bp >= d.B && Fp >= d.F && ld_fb >= d.F && Rp >= d.R,
                "train sb input: NULL pointer / padded sizes below (B, F, rows)");
    if (workspace_bytes < tr_ws_bytes(d)) {
        fsn_set_error("train sb input: workspace too small");
        return FSN_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const TrWs w = tr_carve(d, workspace);  // rowsum: left there by fsn_train_fb_input of the same step
    const unsigned gx = (unsigned)((Rp + 7) / 8 < 1024 ? (Rp + 7) / 8 : 1024);
    if (d.cum) {
        FSN_REQUIRE((size_t)Rp <= tr_rcap(d), "train sb input: cumulative norm takes at most the rows rounded up to 64 as padded rows");
        hipLaunchKernelGGL(tr_sb_cum_sum_kernel, dim3(gx, (unsigned)d.Tp), dim3(256), 0, s, mag_tm, fb_out_tm, ld_fb, w.rowframe, d, Bp, Fp, Rp);
        FSN_TRY_LAUNCH("tr_sb_cum_sum_kernel");
        hipLaunchKernelGGL(tr_sb_cum_scan_kernel, dim3((unsigned)((Rp + 31) / 32)), dim3(256), 0, s, w.rowframe, den, d, Rp);
        FSN_TRY_LAUNCH("tr_sb_cum_scan_kernel");
    } else {
        hipLaunchKernelGGL(tr_fbsum_partial_kernel, dim3((unsigned)d.Tp, (unsigned)d.B), dim3(256), 0, s, fb_out_tm, ld_fb, w.partial, d, Bp);
        FSN_TRY_LAUNCH("tr_fbsum_partial_kernel");
        hipLaunchKernelGGL(tr_sb_den_kernel, dim3((unsigned)d.B), dim3(256), 0, s, w.rowsum, w.partial, den, d);
        FSN_TRY_LAUNCH("tr_sb_den_kernel");
    }
    hipLaunchKernelGGL(tr_sb_input_kernel, dim3(gx, (unsigned)d.Tp), dim3(256), 0, s, mag_tm, fb_out_tm, ld_fb, den, sb_in, d, Bp, Fp, Rp);
    return fsn_check_launch("tr_sb_input_kernel");
}

extern "C" int fsn_train_sb_input_backward(const fsn_train_dims* dims, const float* dx, const float* sb_in, int Rp, const float* den,
                                           const float* fb_out_tm, long ld_fb, int Bp, float* d_fb, long ld_dfb, void* workspace,
                                           size_t workspace_bytes, void* stream) {
    FsnCallScope scope(stream);
    TR_REQUIRE_DIMS(dims);
    const TrDims d = tr_dims(dims);
    FSN_REQUIRE(dx && sb_in && den && fb_out_tm && d_fb && workspace && Bp >= d.B && ld_fb >= d.F && ld_dfb >= d.F && Rp >= d.R,
                "train sb input backward: NULL pointer / padded sizes below (B, F, rows)");
    if (workspace_bytes < tr_ws_bytes(d)) {
        fsn_set_error("train sb input backward: workspace too small");
        return FSN_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const TrWs w = tr_carve(d, workspace);
    if (d.cum) {
        FSN_REQUIRE((size_t)Rp <= tr_rcap(d), "train sb input backward: cumulative norm takes at most the rows rounded up to 64 as padded rows");
        hipLaunchKernelGGL(tr_sb_cum_bwd_p_kernel, dim3((unsigned)((Rp + 31) / 32), (unsigned)d.Tp), dim3(256), 0, s, dx, sb_in, den, w.rowframe, d,
                           Rp);
        FSN_TRY_LAUNCH("tr_sb_cum_bwd_p_kernel");
        hipLaunchKernelGGL(tr_sb_cum_bwd_scan_kernel, dim3((unsigned)((Rp + 31) / 32)), dim3(256), 0, s, w.rowframe, w.G, d, Rp);
        FSN_TRY_LAUNCH("tr_sb_cum_bwd_scan_kernel");
    } else {
        hipLaunchKernelGGL(tr_sb_bwd_partial_kernel, dim3((unsigned)d.Tp, (unsigned)d.B), dim3(256), 0, s, dx, sb_in, w.partial, d, Rp);
        FSN_TRY_LAUNCH("tr_sb_bwd_partial_kernel");
        hipLaunchKernelGGL(tr_sb_bwd_dmu_kernel, dim3((unsigned)d.B), dim3(256), 0, s, w.partial, den, w.dmu, d);
        FSN_TRY_LAUNCH("tr_sb_bwd_dmu_kernel");
    }
    hipLaunchKernelGGL(tr_sb_bwd_dfb_kernel, dim3((unsigned)((ld_dfb + 255) / 256), (unsigned)d.Tp, (unsigned)Bp), dim3(256), 0, s, dx, den,
                       d.cum ? w.G : w.dmu, fb_out_tm, ld_fb, d_fb, ld_dfb, d, Bp, Rp);
    return fsn_check_launch("tr_sb_bwd_dfb_kernel");
}

extern "C" int fsn_train_mask_out(const fsn_train_dims* dims, const float* y, int Rp, float* mask, void* stream) {
    FsnCallScope scope(stream);
    TR_REQUIRE_DIMS(dims);
    const TrDims d = tr_dims(dims);
    FSN_REQUIRE(y && mask && Rp >= d.R, "train mask out: NULL pointer / fewer padded rows than rows");
    hipLaunchKernelGGL(tr_mask_out_kernel, dim3((unsigned)((d.R + 31) / 32), (unsigned)((d.T + 31) / 32)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), y, mask, d, Rp);
    return fsn_check_launch("tr_mask_out_kernel");
}

extern "C" int fsn_train_mask_grad(const fsn_train_dims* dims, const float* d_mask, float* dy, int Rp, int ld, void* stream) {
    FsnCallScope scope(stream);
    TR_REQUIRE_DIMS(dims);
    const TrDims d = tr_dims(dims);
    FSN_REQUIRE(d_mask && dy && Rp >= d.R && ld >= 2, "train mask grad: NULL pointer / bad padded sizes");
    const size_t n = (size_t)Rp * ld;
    hipLaunchKernelGGL(tr_mask_grad_kernel, dim3((unsigned)((n + 255) / 256 < 512 ? (n + 255) / 256 : 512), (unsigned)d.Tp), dim3(256), 0,
                       static_cast<hipStream_t>(stream), d_mask, dy, d, Rp, ld);
    return fsn_check_launch("tr_mask_grad_kernel");
}

extern "C" int fsn_train_cirm_target(const fsn_train_dims* dims, const float* noisy_real, const float* noisy_imag,
                                     const float* clean_real, const float* clean_imag, float* target, void* stream) {
    FsnCallScope scope(stream);
    TR_REQUIRE_DIMS(dims);
    const TrDims d = tr_dims(dims);
    FSN_REQUIRE(noisy_real && noisy_imag && clean_real && clean_imag && target, "train target: NULL pointer argument");
    hipLaunchKernelGGL(tr_target_kernel, dim3((unsigned)((d.T + 255) / 256), (unsigned)d.R), dim3(256), 0, static_cast<hipStream_t>(stream),
                       noisy_real, noisy_imag, clean_real, clean_imag, target, d);
    return fsn_check_launch("tr_target_kernel");
}

extern "C" int fsn_train_rows_pieces(const float* src, float* dst, int T, long N, int W, int rows, int n, int to_pieces, void* stream) {
    FsnCallScope scope(stream);
    FSN_REQUIRE(src && dst && T >= 1 && T <= 65535 && N >= 1 && W >= 2 && W % 2 == 0 && rows >= 1 && n >= 1 && (long)n * rows >= N,
                "train rows <-> pieces: NULL pointer / bad sizes (W even, n rows >= N, T <= 65535)");
    const long per_t = (long)n * rows * (W / 2);
    const unsigned gx = (unsigned)((per_t + 255) / 256 < 2048 ? (per_t + 255) / 256 : 2048);
    if (to_pieces)
        hipLaunchKernelGGL(tr_pieces_kernel<true>, dim3(gx, (unsigned)T), dim3(256), 0, static_cast<hipStream_t>(stream), src, dst, T, N,
                           W / 2, rows, n);
    else
        hipLaunchKernelGGL(tr_pieces_kernel<false>, dim3(gx, (unsigned)T), dim3(256), 0, static_cast<hipStream_t>(stream), src, dst, T, N,
                           W / 2, rows, n);
    return fsn_check_launch("tr_pieces_kernel");
}

extern "C" int fsn_scale_by_scalar(const float* x, const float* scale, float* y, size_t n, void* stream) {
    FsnCallScope scope(stream);
    FSN_REQUIRE(x && scale && y, "scale: NULL pointer argument");
    if (n == 0) return FSN_OK;
    const size_t g = (n + 255) / 256;
    hipLaunchKernelGGL(tr_scale_kernel, dim3((unsigned)(g < 2048 ? g : 2048)), dim3(256), 0, static_cast<hipStream_t>(stream), x, scale, y, n);
    return fsn_check_launch("tr_scale_kernel");
}
