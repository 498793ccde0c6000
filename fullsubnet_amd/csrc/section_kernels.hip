// Input of one band section of Improved FullSubNet's sub-band model (improved_fullsubnet/model.py:315-440): unit u of the
// band [lower, upper) sees the noisy bins lower + u c - n .. lower + (u + 1) c + n - 1 (c centre bins, n neighbours on
// each side, reflected at the two ends of the spectrum) and the same window of the full-band model's output (its own c, n),
// concatenated along the bin axis and divided by (mean over the WHOLE section of an utterance + eps):
// offline_laplace_norm of model.py:124-150.
//
// As tensor algebra this is two gathers, a concat, three norm kernels, a fill and a transposing copy per section - nine
// launches and eight passes over the unfolded tensor (48 MB for the first section at batch 32), 36 launches for the four
// sections in front of every recurrence.  Here: the unfolded tensor is never formed.
//   1. row sums S[b][f] = sum_t x[b][f][t] of both inputs in fp64 (one wave per row, a fixed butterfly);
//   2. mu[b] = (sum over the section's (unit, column) pairs of S at the gathered bin) / (units x columns x T): the mean
//      of the unfolded tensor is a multiplicity-weighted sum of row sums (exact in fp64, rounded once like fsn_norm);
//   3. out[t][b units_loc + u][col] = x[b][bin(u, col)][t] / (mu[b] + eps), time-major with zero padding, which is the
//      layout the LSTM entries take (fsn_lstm2_forward, fsn_lstm2_forward_multi): a transposing gather through LDS.
#include "fsn_common.h"

namespace {

struct SectionArgs {
    const float* noisy;  // [B][F][T]
    const float* fb;     // [B][F][T]
    int B, F, T;
    int lower, units;            // band start, units of the whole section
    int sc, sn, fc, fn;          // centre / neighbour bins of the noisy and of the full-band window
    int u_lo, u_hi;              // units produced by this call (frequency-axis shard)
    float eps;
};

__device__ __forceinline__ int reflect_bin(int j, int F) {
    j = j < 0 ? -j : j;
    return j > F - 1 ? 2 * (F - 1) - j : j;
}
// column col of unit u: which input (0 noisy, 1 full-band output) and which bin
__device__ __forceinline__ int section_bin(const SectionArgs& a, int u, int col, int& src) {
    const int ws = a.sc + 2 * a.sn;
    if (col < ws) {
        src = 0;
        return reflect_bin(a.lower - a.sn + a.sc * u + col, a.F);
    }
    src = 1;
    return reflect_bin(a.lower - a.fn + a.fc * u + (col - ws), a.F);
}

// S[b * F + f] = sum_t x[b][f][t] (fp64), one wave per row; rows of noisy first, then of fb
__global__ __launch_bounds__(256) void section_rowsum_kernel(const float* __restrict__ noisy, const float* __restrict__ fb,
                                                             double* __restrict__ S, long rows, int T) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= 2 * rows) return;
    const float* p = (r < rows ? noisy + r * T : fb + (r - rows) * T);
    double s = 0.0;
    for (int t = lane; t < T; t += 64) s += (double)p[t];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0) S[r] = s;
}

// den[b] = float(sum / count) + eps: one workgroup per utterance, the (unit, column) pairs strided over its threads,
// partial sums met in a fixed order
__global__ __launch_bounds__(256) void section_mean_kernel(const SectionArgs a, const double* __restrict__ S,
                                                           float* __restrict__ den) {
    __shared__ double red[256];
    const int b = blockIdx.x, W = a.sc + 2 * a.sn + a.fc + 2 * a.fn;
    const long rows = (long)a.B * a.F;
    double s = 0.0;
    for (int i = threadIdx.x; i < a.units * W; i += 256) {
        int src;
        const int bin = section_bin(a, i / W, i % W, src);
        s += S[(src ? rows : 0) + (long)b * a.F + bin];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double n = (double)a.units * W * a.T;
        den[b] = (float)(red[0] / n) + a.eps;
    }
}

// out[t][row][col], row = b units_loc + (u - u_lo) < rows_valid, col < W: the gathered value / den[b]; zero elsewhere
// (rows up to Np, columns up to ldo).  A workgroup = one row x 64 frames: reads run along t, writes along col.
__global__ __launch_bounds__(256) void section_gather_kernel(const SectionArgs a, const float* __restrict__ den,
                                                             float* __restrict__ out, int Np, int ldo) {
    extern __shared__ float tile[];  // [64][ldo + 1]
    const int W = a.sc + 2 * a.sn + a.fc + 2 * a.fn, uloc = a.u_hi - a.u_lo;
    const int row = blockIdx.y, t0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool valid = row < a.B * uloc;
    const int pitch = ldo + 1;
    if (valid) {
        const int b = row / uloc, u = a.u_lo + row % uloc;
        const float d = den[b];
        const int t = t0 + lane;
        for (int col = wave; col < W; col += 4) {
            int src;
            const int bin = section_bin(a, u, col, src);
            const float* p = (src ? a.fb : a.noisy) + ((long)b * a.F + bin) * a.T;
            tile[lane * pitch + col] = t < a.T ? p[t] / d : 0.f;
        }
    }
    __syncthreads();
    const int nt = a.T - t0 < 64 ? a.T - t0 : 64;
    for (int i = threadIdx.x; i < nt * ldo; i += 256) {
        const int tt = i / ldo, col = i - tt * ldo;
        out[((long)(t0 + tt) * Np + row) * ldo + col] = (valid && col < W) ? tile[tt * pitch + col] : 0.f;
    }
}

}  // namespace

size_t fsn_section_input_workspace_floats(int B, int F) {
    return (size_t)2 * 2 * B * F + (size_t)B + 16;  // row sums (doubles) | den
}

int fsn_launch_section_input(const float* noisy, const float* fb, int B, int F, int T, int lower, int units, int sc, int sn,
                             int fc, int fn, int u_lo, int u_hi, float eps, float* out, int Np, int ldo, void* workspace,
                             hipStream_t s) {
    SectionArgs a{};
    a.noisy = noisy;
    a.fb = fb;
    a.B = B, a.F = F, a.T = T;
    a.lower = lower, a.units = units;
    a.sc = sc, a.sn = sn, a.fc = fc, a.fn = fn;
    a.u_lo = u_lo, a.u_hi = u_hi;
    a.eps = eps;
    double* S = static_cast<double*>(workspace);
    float* den = reinterpret_cast<float*>(S + (size_t)2 * B * F);
    const long rows = (long)B * F;
    hipLaunchKernelGGL(section_rowsum_kernel, dim3((unsigned)((2 * rows + 3) / 4)), dim3(256), 0, s, noisy, fb, S, rows, T);
    FSN_TRY_LAUNCH("section_rowsum_kernel");
    hipLaunchKernelGGL(section_mean_kernel, dim3((unsigned)B), dim3(256), 0, s, a, S, den);
    FSN_TRY_LAUNCH("section_mean_kernel");
    const size_t lds = (size_t)64 * (ldo + 1) * sizeof(float);
    hipLaunchKernelGGL(section_gather_kernel, dim3((unsigned)((T + 63) / 64), (unsigned)Np), dim3(256), lds, s, a, den, out, Np, ldo);
    return fsn_check_launch("section_gather_kernel");
}
