// The five feature norms of audio_zen/model/base_model.py (norm_wrapper :356-372) on a [B, C, F, T] tensor, as the
// composed models use them between their SequenceModel blocks (Fast / Improved FullSubNet, the full-band baseline, every
// FullSubNet constructor combination outside the fused kernels):
//   offline_laplace_norm    :204-218   x / (mean_{c,f,t} x + 1e-5)                                per sample
//   cumulative_laplace_norm :221-251   x / (sum_{tau<=t} sum_f x / (F (t+1)) + eps)               per (sample, channel)
//   offline_gaussian_norm   :295-310   (x - mean) / (std + 1e-5), std unbiased (torch.std)        per sample
//   cumulative_layer_norm   :312-354   (x - mean_t) / sqrt(var_t + eps), running over frames      per (sample, channel)
//   forgetting_norm         :103-151   x / (mu_t + 1e-10), mu_t an exponentially forgetting mean  per sample
// HBM-bound streaming: the tensor is read twice (statistics, then the division) and written once.  Three kernels in the
// style of binsum / cumulative_den: (1) per-frame sums of x and x^2 over the F-like extent (threads along t: coalesced;
// fp64 accumulation in a fixed order, so results do not depend on the launch shape), (2) one thread per row turns them
// into a per-frame (shift, divisor) pair - totals for the offline norms, a running scan for the cumulative ones, the
// reference's fp32 recurrence for the forgetting norm, (3) y = (x - shift) / divisor.
#include "fsn_common.h"

namespace {

constexpr float kEpsF32 = 1.1920928955078125e-07f;  // torch.finfo(torch.float32).eps = audio_zen.constant.EPSILON

// x viewed as [R][Fr][T] (T contiguous): s1[r][t] = sum_f x, s2[r][t] = sum_f x^2.  A workgroup = 64 frames x 4 slices
// of the Fr extent (a wave each; 64 lanes x 4 B = one 256-byte row segment per load), the slices combined through LDS in
// a fixed order; blockIdx.z splits Fr further when there are few rows (the offline norms of a small batch: R = B), the
// parts being added by norm_scan_kernel in index order.  Fixed order everywhere: results do not depend on timing.
__global__ __launch_bounds__(256) void norm_frame_sums_kernel(const float* __restrict__ x, double* __restrict__ s1,
                                                              double* __restrict__ s2, int Fr, int T, int want_sq, long R,
                                                              int f_per_part) {
    __shared__ double red[2][3][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + lane;
    const long r = blockIdx.y;
    const int f0 = blockIdx.z * f_per_part;
    int f1 = f0 + f_per_part;
    f1 = f1 < Fr ? f1 : Fr;
    double a = 0.0, b = 0.0;
    if (t < T) {
        const float* p = x + r * (long)Fr * T + t;
        for (int f = f0 + w; f < f1; f += 4) {
            const double v = p[(long)f * T];
            a += v;
            if (want_sq) b += v * v;
        }
    }
    if (w > 0) {
        red[0][w - 1][lane] = a;
        red[1][w - 1][lane] = b;
    }
    __syncthreads();
    if (w == 0 && t < T) {
        for (int k = 0; k < 3; ++k) {
            a += red[0][k][lane];
            b += red[1][k][lane];
        }
        const long o = ((long)blockIdx.z * R + r) * T + t;
        s1[o] = a;
        if (want_sq) s2[o] = b;
    }
}

// one workgroup per row: (shift, divisor) per frame.  The threads fetch the frames' sums (coalesced, the Fr parts
// added in index order) into LDS; the recurrence over the frames then runs in thread 0 out of LDS - it is sequential by
// nature (a running sum, or the forgetting norm's fp32 recurrence) but no longer pays a memory round trip per frame.
// The arithmetic after the sums follows the reference's fp32 tensor operations (same operand types, same order) - the
// sums themselves are exactly-summed fp64 values rounded once.
__global__ __launch_bounds__(256) void norm_scan_kernel(const double* __restrict__ s1, const double* __restrict__ s2,
                                                        float* __restrict__ shift, float* __restrict__ den, int R, int Fr,
                                                        int T, int norm_type, int sample_length, float eps, int parts,
                                                        int want_sq) {
    extern __shared__ double row[];  // a[T] | b[T] | shift[T] | den[T] (the last two as floats)
    __shared__ float stat[2];
    const int r = blockIdx.x, lane = threadIdx.x;  // "lane": thread of the workgroup; the recurrences run in thread 0
    double* a = row;
    double* b = row + T;
    float* sh = reinterpret_cast<float*>(row + 2 * T);
    float* dn = sh + T;
    // the Fr parts of a frame's sums, added in index order; four waves share the frames and eight loads are in flight per
    // thread (with up to 64 parts and one wave this gather alone was 70 us for a 301-frame row)
    for (int t = lane; t < T; t += 256) {
        double va = 0.0, vb = 0.0;
        const double* p1 = s1 + (long)r * T + t;
        const double* p2 = s2 + (long)r * T + t;
        const long zs = (long)R * T;
#pragma unroll 8
        for (int z = 0; z < parts; ++z) {
            va += p1[z * zs];
            if (want_sq) vb += p2[z * zs];
        }
        a[t] = va;
        b[t] = vb;
    }
    __syncthreads();
    if (norm_type == FSN_NORM_OFFLINE_LAPLACE || norm_type == FSN_NORM_OFFLINE_GAUSSIAN) {
        // one statistic per row: the lanes of wave 0 add their frames (t = lane, lane + 64, ...), then a fixed butterfly -
        // the same order in every run (a serial walk over LDS by one lane cost 60 us at batch 1)
        if (lane < 64) {
            double tot = 0.0, tot2 = 0.0;
            for (int t = lane; t < T; t += 64) {
                tot += a[t];
                if (norm_type == FSN_NORM_OFFLINE_GAUSSIAN) tot2 += b[t];
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                tot += __shfl_xor(tot, m, 64);
                tot2 += __shfl_xor(tot2, m, 64);
            }
            const double n = (double)Fr * T;
            const float mu = (float)(tot / n);
            float s = 0.f, d;
            if (norm_type == FSN_NORM_OFFLINE_LAPLACE) {
                d = mu + eps;
            } else {
                const double m = tot / n;
                double var = (tot2 - n * m * m) / (n - 1.0);  // torch.std: unbiased
                var = var > 0.0 ? var : 0.0;
                s = mu;
                d = (float)sqrt(var) + eps;
            }
            if (lane == 0) stat[0] = s, stat[1] = d;
        }
        __syncthreads();
        const float s = stat[0], d = stat[1];
        for (int t = lane; t < T; t += 256) {
            shift[(long)r * T + t] = s;
            den[(long)r * T + t] = d;
        }
        return;
    }
    if (lane == 0) {
    if (norm_type == FSN_NORM_CUMULATIVE_LAPLACE || norm_type == FSN_NORM_CUMULATIVE_LAYER) {
        double c1 = 0.0, c2 = 0.0;
        for (int t = 0; t < T; ++t) {
            c1 += a[t];
            const float cum1 = (float)c1;
            const float count = (float)((long)Fr * (t + 1));
            const float mean = cum1 / count;
            if (norm_type == FSN_NORM_CUMULATIVE_LAPLACE) {
                sh[t] = 0.f;
                dn[t] = mean + eps;
            } else {
                c2 += b[t];
                const float cum2 = (float)c2;
                const float var = (cum2 - 2.f * mean * cum1) / count + mean * mean;  // base_model.py:341-345
                sh[t] = mean;
                dn[t] = sqrtf(var + eps);
            }
        }
    } else {  // forgetting_norm: the reference's fp32 recurrence, operation by operation
        const double alpha_d = (double)(sample_length - 1) / (double)(sample_length + 1);
        const float alpha = (float)alpha_d;
        float mu = 0.f;
        for (int t = 0; t < T; ++t) {
            const float fm = (float)(a[t] / (double)Fr);
            float al = alpha, oma = (float)(1.0 - alpha_d);  // t >= sample_length: Python doubles, rounded when they meet the tensor
            if (t < sample_length) {
                const float a_t = (float)((double)(t - 1) / (double)(t + 1));  // torch.tensor([...]) rounds to fp32
                al = a_t < alpha ? a_t : alpha;
                oma = 1.f - al;  // (1 - alp) on an fp32 tensor
            }
            mu = al * mu + oma * fm;
            sh[t] = 0.f;
            dn[t] = mu + eps;
        }
    }
    }
    __syncthreads();
    for (int t = lane; t < T; t += 256) {
        shift[(long)r * T + t] = sh[t];
        den[(long)r * T + t] = dn[t];
    }
}

// y[r][f][t] = (x - shift[rs][t]) / den[rs][t], rs = r / rows_per_stat (a statistic row may span several tensor rows)
__global__ __launch_bounds__(256) void norm_apply_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         const float* __restrict__ shift, const float* __restrict__ den,
                                                         long rows, int T, int rows_per_stat, int has_shift) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    for (long row = blockIdx.y; row < rows; row += gridDim.y) {
        const long rs = row / rows_per_stat;
        const float d = den[rs * T + t];
        const float v = x[row * T + t];
        y[row * T + t] = has_shift ? (v - shift[rs * T + t]) / d : v / d;
    }
}

struct NormDims {
    long R;     // statistic rows
    int Fr;     // extent summed per frame
    int parts;  // the Fr extent is summed in this many parts (more workgroups when there are few rows)
    int f_per_part;
};
NormDims norm_dims(int norm_type, int B, int C, int F, int T) {
    const bool per_channel = norm_type == FSN_NORM_CUMULATIVE_LAPLACE || norm_type == FSN_NORM_CUMULATIVE_LAYER;
    NormDims d = per_channel ? NormDims{(long)B * C, F, 1, F} : NormDims{(long)B, C * F, 1, C * F};
    const long blocks = d.R * ((T + 63) / 64);
    int parts = 1;
    while (parts < 64 && blocks * parts < 1024 && d.Fr / (parts * 2) >= 16) parts *= 2;
    d.parts = parts;
    d.f_per_part = (d.Fr + parts - 1) / parts;
    return d;
}

}  // namespace

extern "C" size_t fsn_norm_workspace_bytes(int norm_type, int B, int C, int F, int T) {
    if (norm_type < FSN_NORM_OFFLINE_LAPLACE || norm_type > FSN_NORM_FORGETTING || B < 1 || C < 1 || F < 1 || T < 1) return 0;
    const NormDims d = norm_dims(norm_type, B, C, F, T);
    return fsn_round_up_sz((size_t)d.R * T * ((size_t)d.parts * 2 * sizeof(double) + 2 * sizeof(float)), 256);
}

extern "C" int fsn_norm(const float* x, float* y, int norm_type, int B, int C, int F, int T, int sample_length, float eps,
                        void* workspace, size_t workspace_bytes, void* stream) {
    FsnCallScope scope(stream);
    FSN_REQUIRE(x && y && workspace, "NULL pointer argument");
    FSN_REQUIRE(norm_type >= FSN_NORM_OFFLINE_LAPLACE && norm_type <= FSN_NORM_FORGETTING, "norm_type %d unknown", norm_type);
    FSN_REQUIRE(B >= 1 && C >= 1 && F >= 1 && T >= 1 && (long)B * C <= 0x7fffffffL && (long)C * F <= 0x7fffffffL,
                "norm: bad shape [%d, %d, %d, %d]", B, C, F, T);
    FSN_REQUIRE(norm_type != FSN_NORM_FORGETTING || sample_length >= 1, "forgetting_norm: sample_length %d < 1", sample_length);
    FSN_REQUIRE(norm_type != FSN_NORM_OFFLINE_GAUSSIAN || (long)C * F * T >= 2, "offline_gaussian_norm: needs two values");
    if (workspace_bytes < fsn_norm_workspace_bytes(norm_type, B, C, F, T)) {
        fsn_set_error("norm: workspace too small");
        return FSN_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const NormDims d = norm_dims(norm_type, B, C, F, T);
    const size_t scan_lds = (size_t)T * (2 * sizeof(double) + 2 * sizeof(float));
    FSN_REQUIRE(d.R <= 0x7fffffffL && scan_lds <= 144 * 1024, "norm: too many statistic rows or frames (T <= %d)",
                FSN_NORM_MAX_FRAMES);
    if (scan_lds > 64 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(norm_scan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)scan_lds);
        (void)hipGetLastError();
    }
    double* s1 = static_cast<double*>(workspace);
    double* s2 = s1 + (size_t)d.parts * d.R * T;
    float* shift = reinterpret_cast<float*>(s2 + (size_t)d.parts * d.R * T);
    float* den = shift + d.R * T;
    const int want_sq = norm_type == FSN_NORM_OFFLINE_GAUSSIAN || norm_type == FSN_NORM_CUMULATIVE_LAYER;
    // grid.y carries the rows: chunked for row counts beyond the 65535 limit (a chunk's parts land where the whole
    // launch's would: the row index inside the kernel is offset through the pointers, R stays the full count)
    for (long r0 = 0; r0 < d.R; r0 += 65535) {
        const long nr = d.R - r0 < 65535 ? d.R - r0 : 65535;
        hipLaunchKernelGGL(norm_frame_sums_kernel, dim3((unsigned)((T + 63) / 64), (unsigned)nr, (unsigned)d.parts), dim3(256), 0, s,
                           x + r0 * (long)d.Fr * T, s1 + r0 * T, s2 + r0 * T, d.Fr, T, want_sq, d.R, d.f_per_part);
    }
    FSN_TRY_LAUNCH("norm_frame_sums_kernel");
    if (!(eps > 0.f))  // the constants of audio_zen/model/base_model.py
        eps = (norm_type == FSN_NORM_OFFLINE_LAPLACE || norm_type == FSN_NORM_OFFLINE_GAUSSIAN) ? 1e-5f
              : norm_type == FSN_NORM_FORGETTING                                               ? 1e-10f
                                                                                               : kEpsF32;
    hipLaunchKernelGGL(norm_scan_kernel, dim3((unsigned)d.R), dim3(256), scan_lds, s, s1,
                       s2, shift, den, (int)d.R, d.Fr, T, norm_type, sample_length, eps, d.parts, want_sq);
    FSN_TRY_LAUNCH("norm_scan_kernel");
    const long rows = (long)B * C * F;
    const unsigned gy = (unsigned)(rows < 32768 ? rows : 32768);
    hipLaunchKernelGGL(norm_apply_kernel, dim3((unsigned)((T + 255) / 256), gy), dim3(256), 0, s, x, y, shift, den, rows, T, d.Fr,
                       norm_type == FSN_NORM_OFFLINE_GAUSSIAN || norm_type == FSN_NORM_CUMULATIVE_LAYER);
    return fsn_check_launch("norm_apply_kernel");
}
