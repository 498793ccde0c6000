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

// ---- round 5: the glue AROUND the models of the composed families as kernels -------------------------------------------------
// Improved FullSubNet's forward (improved_fullsubnet/model.py:541-591) still ran 21 tensor-algebra launches of the host
// framework per call (tools/diag_aten_ops.py): mag ** fdrc, the last-bin slice (a non-contiguous view that every consumer
// copied: eight times), the [B, F, T] <-> time-major transposes around a SequenceModel, the sections' output re-ordering,
// cat, F.pad and the two mask products.  Four kernels, all pure data movement + the same IEEE operations (sqrtf, one
// multiply): bit-identical to the tensor algebra.
namespace {

// out [B][F - 1][T] = mag[b][f][t] ** fdrc for fdrc = 0.5 (sqrtf: what torch.pow dispatches to for that exponent) or 1
__global__ __launch_bounds__(256) void improved_front_kernel(const float* __restrict__ mag, float* __restrict__ out, int B, int F, int T,
                                                            int mode) {
    const long n = (long)B * (F - 1) * T;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long row = i / T;
        const int t = (int)(i - row * T);
        const long b = row / (F - 1), f = row - b * (F - 1);
        const float v = mag[(b * F + f) * T + t];
        out[i] = mode ? sqrtf(v) : v;
    }
}
// h [T][Np][Ip] = x[b][f][t] (zero for b >= B, f >= F): 32 x 32 tiles through LDS, both sides coalesced
__global__ __launch_bounds__(256) void bft_to_rows_kernel(const float* __restrict__ x, float* __restrict__ h, int B, int F, int T, int Np,
                                                         int Ip, int z0) {
    __shared__ float tile[32][33];
    const int t0 = blockIdx.x * 32, f0 = blockIdx.y * 32, b = z0 + (int)blockIdx.z;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int f = f0 + i, t = t0 + tx;
        tile[i][tx] = (b < B && f < F && t < T) ? x[((long)b * F + f) * T + t] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int t = t0 + i, f = f0 + tx;
        if (t < T && f < Ip) h[((long)t * Np + b) * Ip + f] = tile[tx][i];
    }
}
// y [B][O][T] = o[t][b][c] (o: [T][Np][ld])
__global__ __launch_bounds__(256) void rows_to_bft_kernel(const float* __restrict__ o, float* __restrict__ y, int T, int Np, int ld, int B,
                                                         int O, int z0) {
    __shared__ float tile[32][33];
    const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32, b = z0 + (int)blockIdx.z;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int t = t0 + i, c = c0 + tx;
        tile[i][tx] = (t < T && c < O) ? o[((long)t * Np + b) * ld + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, t = t0 + tx;
        if (c < O && t < T) y[((long)b * O + c) * T + t] = tile[tx][i];
    }
}
struct MaskSections {
    fsn_mask_section s[8];
    int n;
};
// bins no section covers (the last bin: model.py:566, 575 F.pad): zero in both planes
__global__ __launch_bounds__(256) void mask_zero_uncovered_kernel(const MaskSections ms, float* __restrict__ er, float* __restrict__ ei,
                                                                 int F, int T) {
    const int f = blockIdx.x, b = blockIdx.y;
    for (int i = 0; i < ms.n; ++i)
        if (f >= ms.s[i].lower && f < ms.s[i].lower + ms.s[i].units * ms.s[i].center) return;
    const long base = ((long)b * F + f) * T;
    for (int t = threadIdx.x; t < T; t += 256) er[base + t] = 0.f, ei[base + t] = 0.f;
}
// one section: o [T][Np][ld], row b units + u, column comp center + cc  ->  er / ei [B][F][T] at bin lower + u center + cc,
// times the noisy real / imaginary part (model.py:576-577: no complex product).  A workgroup = 8 rows x 32 frames.
__global__ __launch_bounds__(256) void mask_apply_kernel(const fsn_mask_section sec, const float* __restrict__ real,
                                                        const float* __restrict__ imag, float* __restrict__ er, float* __restrict__ ei,
                                                        int B, int F, int T, int R, int y0) {
    extern __shared__ float tile[];  // [32 frames][R rows * ld + 1], R = rows per workgroup (8 for narrow sections, fewer for wide)
    const int W = 2 * sec.center, pitch = R * sec.ld + 1;
    const int t0 = blockIdx.x * 32, r0 = (y0 + (int)blockIdx.y) * R;
    const int rows = B * sec.units;
    const int nt = T - t0 < 32 ? T - t0 : 32;
    const float* o = static_cast<const float*>(sec.o);
    for (int i = threadIdx.x; i < nt * R * sec.ld; i += 256) {  // R rows x ld columns are contiguous for a frame
        const int tt = i / (R * sec.ld), j = i - tt * (R * sec.ld);
        tile[tt * pitch + j] = (r0 + j / sec.ld < sec.Np) ? o[((long)(t0 + tt) * sec.Np + r0) * sec.ld + j] : 0.f;
    }
    __syncthreads();
    const int tt = threadIdx.x & 31;
    for (int p = threadIdx.x >> 5; p < R * W; p += 8) {  // (row, column) pairs, 32 frames each: 128-byte runs in the planes
        const int rr = p / W, col = p - rr * W, row = r0 + rr;
        if (row >= rows || tt >= nt) continue;
        const int b = row / sec.units, u = row - b * sec.units;
        const int comp = col / sec.center, cc = col - comp * sec.center;
        const long idx = ((long)b * F + sec.lower + u * sec.center + cc) * T + t0 + tt;
        const float m = tile[tt * pitch + rr * sec.ld + col];
        if (comp == 0) er[idx] = m * real[idx];
        else ei[idx] = m * imag[idx];
    }
}

}  // namespace

extern "C" int fsn_improved_front(const float* mag, int B, int F, int T, int sqrt_mode, float* out, void* stream) {
    FsnCallScope scope(stream);
    FSN_REQUIRE(mag && out && B >= 1 && F >= 2 && T >= 1 && (sqrt_mode == 0 || sqrt_mode == 1), "improved front: bad arguments");
    const long n = (long)B * (F - 1) * T;
    const long g = (n + 255) / 256;
    hipLaunchKernelGGL(improved_front_kernel, dim3((unsigned)(g < 4096 ? g : 4096)), dim3(256), 0, static_cast<hipStream_t>(stream), mag, out,
                       B, F, T, sqrt_mode);
    return fsn_check_launch("improved_front_kernel");
}
extern "C" int fsn_bft_to_rows(const float* x, int B, int F, int T, float* h, int Np, int Ip, void* stream) {
    FsnCallScope scope(stream);
    FSN_REQUIRE(x && h && B >= 1 && F >= 1 && T >= 1 && Np >= B && Ip >= F, "bft_to_rows: bad arguments");
    for (int z0 = 0; z0 < Np; z0 += 65535) {  // the row index is grid.z: at most 65535 per launch
        const int nz = Np - z0 < 65535 ? Np - z0 : 65535;
        hipLaunchKernelGGL(bft_to_rows_kernel, dim3((unsigned)((T + 31) / 32), (unsigned)((Ip + 31) / 32), (unsigned)nz), dim3(256), 0,
                           static_cast<hipStream_t>(stream), x, h, B, F, T, Np, Ip, z0);
        FSN_TRY_LAUNCH("bft_to_rows_kernel");
    }
    return FSN_OK;
}
extern "C" int fsn_rows_to_bft(const float* o, int T, int Np, int ld, int B, int O, float* y, void* stream) {
    FsnCallScope scope(stream);
    FSN_REQUIRE(o && y && B >= 1 && O >= 1 && T >= 1 && Np >= B && ld >= O, "rows_to_bft: bad arguments");
    for (int z0 = 0; z0 < B; z0 += 65535) {
        const int nz = B - z0 < 65535 ? B - z0 : 65535;
        hipLaunchKernelGGL(rows_to_bft_kernel, dim3((unsigned)((T + 31) / 32), (unsigned)((O + 31) / 32), (unsigned)nz), dim3(256), 0,
                           static_cast<hipStream_t>(stream), o, y, T, Np, ld, B, O, z0);
        FSN_TRY_LAUNCH("rows_to_bft_kernel");
    }
    return FSN_OK;
}
extern "C" int fsn_improved_mask_apply(int n, const fsn_mask_section* sections, const float* real, const float* imag, int B, int F, int T,
                                       float* er, float* ei, void* stream) {
    FsnCallScope scope(stream);
    FSN_REQUIRE(n >= 1 && n <= 8 && sections && real && imag && er && ei && B >= 1 && F >= 1 && T >= 1 && B <= 65535,
                "improved mask apply: 1 .. 8 sections, non-NULL planes");
    MaskSections ms{};
    ms.n = n;
    for (int i = 0; i < n; ++i) {
        const fsn_mask_section& q = sections[i];
        FSN_REQUIRE(q.o && q.center >= 1 && q.units >= 0 && q.lower >= 0 && q.lower + q.units * q.center <= F && q.ld >= 2 * q.center &&
                        q.ld <= 480 && q.Np >= B * q.units,
                    "improved mask apply: section %d out of range", i);
        ms.s[i] = q;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(mask_zero_uncovered_kernel, dim3((unsigned)F, (unsigned)B), dim3(256), 0, s, ms, er, ei, F, T);
    FSN_TRY_LAUNCH("mask_zero_uncovered_kernel");
    for (int i = 0; i < n; ++i) {
        const fsn_mask_section& q = ms.s[i];
        if (q.units == 0) continue;
        int R = 480 / q.ld;  // rows per workgroup: at most 480 floats per frame in the tile (61.6 KB of LDS: below the 64 KB a launch gets unasked)
        R = R > 8 ? 8 : (R < 1 ? 1 : R);
        const size_t lds = (size_t)32 * (R * q.ld + 1) * sizeof(float);
        const long groups = ((long)B * q.units + R - 1) / R;  // grid.y: at most 65535 per launch
        for (long y0 = 0; y0 < groups; y0 += 65535) {
            const long ny = groups - y0 < 65535 ? groups - y0 : 65535;
            hipLaunchKernelGGL(mask_apply_kernel, dim3((unsigned)((T + 31) / 32), (unsigned)ny), dim3(256), lds, s, q, real, imag, er, ei, B,
                               F, T, R, (int)y0);
            FSN_TRY_LAUNCH("mask_apply_kernel");
        }
    }
    return FSN_OK;
}
