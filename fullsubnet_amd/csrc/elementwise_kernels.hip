// Streaming (HBM-bound) kernels of the path: cIRM algebra, layout changes at the API boundary and
// the normalisation statistics.  All fp32 unless stated; written without fma contraction so that
// the elementwise results follow the reference's operation order.
#include "fsn_common.h"

namespace {

// ---- audio_zen/acoustics/mask.py ------------------------------------------------------------
__device__ __forceinline__ float decompress1(float m) {  // mask.py:47-64, K = 10, limit = 9.9
    const float lim = 9.9f;
    m = m >= lim ? lim : (m <= -lim ? -lim : m);
    return -10.0f * logf((10.0f - m) / (10.0f + m));
}
__device__ __forceinline__ float compress1(float m) {  // mask.py:32-44, K = 10, C = 0.1
    m = m <= -100.0f ? -100.0f : m;
    const float e = expf(-0.1f * m);
    return 10.0f * (1.0f - e) / (1.0f + e);
}

__global__ void decompress_kernel(const float* __restrict__ in, float* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = decompress1(in[i]);
}
__global__ void compress_kernel(const float* __restrict__ in, float* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = compress1(in[i]);
}
// mask.py:7-29
__global__ void build_cirm_kernel(const float* __restrict__ nr, const float* __restrict__ ni,
                                  const float* __restrict__ cr, const float* __restrict__ ci,
                                  float* __restrict__ out, size_t n) {
    const float eps = 1.1920928955078125e-07f;  // audio_zen/constant.py:9
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float a = nr[i], b = ni[i], c = cr[i], d = ci[i];
        const float den = a * a + b * b + eps;
        const float mr = (a * c + b * d) / den;
        const float mi = (a * d - b * c) / den;
        f32x2 o = {compress1(mr), compress1(mi)};
        *reinterpret_cast<f32x2*>(out + 2 * i) = o;
    }
}

// ---- batched 2-D transpose with zero fill: out[b][c][r] = in[b][r][c] ----------------------
// (r < R_valid, c < C_valid come from `in`, the rest of the R x C output tile is zero)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                        int R, int C, long ld_in, long bs_in, long ld_out,
                                                        long bs_out, int R_valid, int C_valid) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R_valid && c < C_valid) ? in[b * bs_in + r * ld_in + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < C && r < R) out[b * bs_out + c * ld_out + r] = tile[tx][i];
    }
}

// ---- normalisation statistics ---------------------------------------------------------------
// binsum[b][f] = sum_t mag[b][t][f], fp64 accumulation, fixed order (deterministic).  A workgroup owns 64 bins of
// one utterance; its 8 wave-rows each walk every 8th frame (four independent loads in flight per thread) and the
// eight partial sums meet in LDS in a fixed order.  (One thread per bin walking all frames serially was a 190-deep
// chain of dependent cache misses on 128 workgroups: 46 us for 13 MB, 0.29 TB/s.)
constexpr int kBinsumTG = 8;
__global__ __launch_bounds__(64 * kBinsumTG) void binsum_kernel(const float* __restrict__ mag,
                                                                double* __restrict__ binsum, int Tp, int FP) {
    __shared__ double part[kBinsumTG][64];
    const int b = blockIdx.y, fl = threadIdx.x & 63, tg = threadIdx.x >> 6;
    const int f = blockIdx.x * 64 + fl;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (f < FP) {
        const float* p = mag + (long)b * Tp * FP + f;
        int t = tg;
        for (; t + 3 * kBinsumTG < Tp; t += 4 * kBinsumTG) {
            const float v0 = p[(long)t * FP], v1 = p[(long)(t + kBinsumTG) * FP];
            const float v2 = p[(long)(t + 2 * kBinsumTG) * FP], v3 = p[(long)(t + 3 * kBinsumTG) * FP];
            a0 += (double)v0;
            a1 += (double)v1;
            a2 += (double)v2;
            a3 += (double)v3;
        }
        for (; t < Tp; t += kBinsumTG) a0 += (double)p[(long)t * FP];
    }
    part[tg][fl] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (tg == 0 && f < FP) {
        double acc = part[0][fl];
#pragma unroll
        for (int k = 1; k < kBinsumTG; ++k) acc += part[k][fl];
        binsum[(long)b * FP + f] = acc;
    }
}

// reflect(j) of F.pad(mode="reflect") for j in [-N, F+N)
__device__ __forceinline__ int reflect_idx(int j, int F) {
    j = j < 0 ? -j : j;
    return j >= F ? 2 * (F - 1) - j : j;
}

__device__ __forceinline__ double block_sum(double v, double* scratch) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    double tot = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += scratch[w];
    return tot;
}

// offline_laplace_norm (base_model.py:204-218): one mean per utterance, eps 1e-5.
//   which == 0: den_fb[b] = mean_{f,t}(mag) + 1e-5                        (fullsubnet/model.py:92)
//   which == 1: den_sb[b] = mean over the concatenated [F, 2nb+2, Tp] sub-band tensor + 1e-5
//               (model.py:110-111) = (sum_f m[f] binsum[f] + sum fb_out) / (F (2nb+2) Tp) with
//               m[f] = number of (unit, row) pairs of freq_unfold (base_model.py:31-44) hitting bin f.
__global__ __launch_bounds__(256) void offline_den_kernel(const double* __restrict__ binsum,
                                                          const float* __restrict__ fb_out,
                                                          float* __restrict__ den_fb, float* __restrict__ den_sb,
                                                          int Tp, int F, int FP, int nb, int which) {
    __shared__ double scratch[4];
    const int b = blockIdx.x;
    double acc = 0.0;
    if (which == 0) {
        for (int f = threadIdx.x; f < F; f += blockDim.x) acc += binsum[(long)b * FP + f];
        const double tot = block_sum(acc, scratch);
        if (threadIdx.x == 0) den_fb[b] = (float)(tot / ((double)F * Tp)) + 1e-5f;
    } else {
        for (int f = threadIdx.x; f < F; f += blockDim.x) {
            // m[f] in closed form: pairs (u, k), u in [0, F), k in [-nb, nb], whose source bin is f -
            // directly (u + k = f), mirrored at the low edge (u + k = -f) or at the high edge
            // (u + k = 2 (F - 1) - f); checked against the brute-force count in tests/test_host_cpu.py
            const int direct = min(nb, f) - max(-nb, f - (F - 1)) + 1;
            const int low = f >= 1 ? max(0, min(F - 1, nb - f) + 1) : 0;
            const int high = f <= F - 2 ? max(0, F - max(0, 2 * (F - 1) - f - nb)) : 0;
            acc += (double)(direct + low + high) * binsum[(long)b * FP + f];
        }
        // columns F..FP-1 of fb_out are written as zeros by the output layer: whole rows can be summed
        const f32x4* p = reinterpret_cast<const f32x4*>(fb_out + (long)b * Tp * FP);
        for (long i = threadIdx.x; i < (long)Tp * FP / 4; i += blockDim.x) {
            const f32x4 v = p[i];
            acc += ((double)v[0] + (double)v[1]) + ((double)v[2] + (double)v[3]);
        }
        const double tot = block_sum(acc, scratch);
        if (threadIdx.x == 0) den_sb[b] = (float)(tot / ((double)F * (2 * nb + 2) * Tp)) + 1e-5f;
    }
}

// cumulative_laplace_norm (base_model.py:221-251) for the full-band input [B,1,F,Tp]:
// den[b][t] = (sum_{tau<=t} sum_f mag[b][tau][f]) / (F (t+1)) + EPSILON.  One block per utterance.
// `carry` (may be NULL) / `t0`: streaming - the running sum of the t0 frames seen before this call, updated.
__global__ __launch_bounds__(256) void cumulative_den_fb_kernel(const float* __restrict__ mag,
                                                                float* __restrict__ den, int Tp, int F, int FP,
                                                                double* __restrict__ carry, int t0) {
    __shared__ double scratch[4];
    const int b = blockIdx.x;
    double run = carry ? carry[b] : 0.0;
    for (int t = 0; t < Tp; ++t) {
        double acc = 0.0;
        for (int f = threadIdx.x; f < F; f += blockDim.x) acc += (double)mag[((long)b * Tp + t) * FP + f];
        run += block_sum(acc, scratch);
        if (threadIdx.x == 0)
            den[(long)b * Tp + t] = (float)(run / ((double)F * (t0 + t + 1))) + 1.1920928955078125e-07f;
    }
    if (carry && threadIdx.x == 0) carry[b] = run;
}

// cumulative_laplace_norm on the 4-D sub-band tensor [B, F, 2nb+2, Tp] (quirk Q4): every unit
// (b, f) is its own "batch" entry with 2nb+2 "frequencies":
// den[t][n] = (sum_{tau<=t} (sum_k mag[b][tau][refl(f+k)] + fb_out[b][tau][f])) / ((2nb+2)(t+1)) + EPS
// with n = b F + f; stored [Tp][Npad] to match the row order of the input projection.
__global__ __launch_bounds__(256) void cumulative_den_sb_kernel(const float* __restrict__ mag,
                                                                const float* __restrict__ fb_out,
                                                                float* __restrict__ den, int B, int Tp, int F,
                                                                int FP, int nb, int Npad, double* __restrict__ carry,
                                                                int t0) {
    const long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= (long)B * F) return;
    const int b = (int)(n / F), f = (int)(n % F);
    double run = carry ? carry[n] : 0.0;
    for (int t = 0; t < Tp; ++t) {
        const float* row = mag + ((long)b * Tp + t) * FP;
        double acc = (double)fb_out[((long)b * Tp + t) * FP + f];
        for (int k = -nb; k <= nb; ++k) acc += (double)row[reflect_idx(f + k, F)];
        run += acc;
        den[(long)t * Npad + n] = (float)(run / ((double)(2 * nb + 2) * (t0 + t + 1))) + 1.1920928955078125e-07f;
    }
    if (carry) carry[n] = run;
}

// Rows [r0, r0 + n) of the flattened (b, f) index space out of the frame-major mask planes [B][T][FP] into
// [row][2][T] (fullsubnet/model.py:129-135 restricted to a row range): 16 rows x 64 frames per workgroup through
// LDS, so that both the plane reads (along f) and the row writes (along t) are contiguous runs.
__global__ __launch_bounds__(256) void crm_rows_kernel(const float* __restrict__ crm_r, const float* __restrict__ crm_i,
                                                       float* __restrict__ out, long r0, long n, int F, int FP, int T) {
    __shared__ float tile[2][64][17];
    const long row0 = (long)blockIdx.x * 16;
    const int t0 = blockIdx.y * 64;
    for (int i = threadIdx.x; i < 2 * 64 * 16; i += 256) {
        const int r = i & 15, t = (i >> 4) & 63, c = i >> 10;
        const long row = row0 + r;
        float v = 0.f;
        if (row < n && t0 + t < T) {
            const long ng = r0 + row;
            const long b = ng / F;
            const int f = (int)(ng % F);
            v = (c ? crm_i : crm_r)[(b * T + t0 + t) * FP + f];
        }
        tile[c][t][r] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * 64 * 16; i += 256) {
        const int t = i & 63, c = (i >> 6) & 1, r = i >> 7;
        const long row = row0 + r;
        if (row < n && t0 + t < T) out[(row * 2 + c) * T + t0 + t] = tile[c][t][r];
    }
}

}  // namespace

static unsigned ew_grid(size_t n) {
    size_t g = (n + 255) / 256;
    return (unsigned)(g < 2048 ? (g ? g : 1) : 2048);
}

int fsn_launch_decompress(const float* in, float* out, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(decompress_kernel, dim3(ew_grid(n)), dim3(256), 0, s, in, out, n);
    return fsn_check_launch("decompress_kernel");
}
int fsn_launch_compress(const float* in, float* out, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(compress_kernel, dim3(ew_grid(n)), dim3(256), 0, s, in, out, n);
    return fsn_check_launch("compress_kernel");
}
int fsn_launch_build_cirm(const float* nr, const float* ni, const float* cr, const float* ci, float* out, size_t n,
                          hipStream_t s) {
    hipLaunchKernelGGL(build_cirm_kernel, dim3(ew_grid(n)), dim3(256), 0, s, nr, ni, cr, ci, out, n);
    return fsn_check_launch("build_cirm_kernel");
}
int fsn_launch_transpose(const float* in, float* out, int batch, int R, int C, long ld_in, long bs_in, long ld_out,
                         long bs_out, int R_valid, int C_valid, hipStream_t s) {
    dim3 grid((C + 31) / 32, (R + 31) / 32, batch);
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, s, in, out, R, C, ld_in, bs_in, ld_out, bs_out, R_valid,
                       C_valid);
    return fsn_check_launch("transpose_kernel");
}
int fsn_launch_crm_rows(const float* crm_r, const float* crm_i, float* out, long r0, long n, int F, int FP, int T,
                        hipStream_t s) {
    hipLaunchKernelGGL(crm_rows_kernel, dim3((unsigned)((n + 15) / 16), (unsigned)((T + 63) / 64)), dim3(256), 0, s, crm_r,
                       crm_i, out, r0, n, F, FP, T);
    return fsn_check_launch("crm_rows_kernel");
}
int fsn_launch_binsum(const float* mag, double* binsum, int B, int Tp, int FP, hipStream_t s) {
    hipLaunchKernelGGL(binsum_kernel, dim3((FP + 63) / 64, B), dim3(64 * kBinsumTG), 0, s, mag, binsum, Tp, FP);
    return fsn_check_launch("binsum_kernel");
}
int fsn_launch_offline_den(const double* binsum, const float* fb_out, float* den_fb, float* den_sb, int B, int Tp,
                           int F, int FP, int nb, int which, hipStream_t s) {
    hipLaunchKernelGGL(offline_den_kernel, dim3(B), dim3(256), 0, s, binsum, fb_out, den_fb, den_sb, Tp, F, FP, nb,
                       which);
    return fsn_check_launch("offline_den_kernel");
}
int fsn_launch_cumulative_den_fb(const float* mag, float* den, int B, int Tp, int F, int FP, hipStream_t s,
                                 double* carry, int t0) {
    hipLaunchKernelGGL(cumulative_den_fb_kernel, dim3(B), dim3(256), 0, s, mag, den, Tp, F, FP, carry, t0);
    return fsn_check_launch("cumulative_den_fb_kernel");
}
int fsn_launch_cumulative_den_sb(const float* mag, const float* fb_out, float* den, int B, int Tp, int F, int FP,
                                 int nb, int Npad, hipStream_t s, double* carry, int t0) {
    const long n = (long)B * F;
    hipLaunchKernelGGL(cumulative_den_sb_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, mag, fb_out, den,
                       B, Tp, F, FP, nb, Npad, carry, t0);
    return fsn_check_launch("cumulative_den_sb_kernel");
}

// Zero n 32-bit words (flags of the persistent kernels).  A kernel instead of hipMemsetAsync: measured on ROCm 7.2,
// the memset node of a captured graph took effect in the first replay only (tests/test_gpu_streaming.py: replays on new
// input read the previous replay's flags and hand-off buffers), a kernel node is replayed every time.
__global__ void zero_words_kernel(unsigned* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
int fsn_launch_zero_words(unsigned* p, size_t n, hipStream_t s) {
    const unsigned blocks = (unsigned)((n + 255) / 256 < 64 ? (n + 255) / 256 : 64);
    hipLaunchKernelGGL(zero_words_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s, p, n);
    return fsn_check_launch("zero_words_kernel");
}

// The persistent kernels whose workgroups wait for each other bound every spin; when a bound is hit (a workgroup could
// not be placed because something else held the CUs for seconds) they raise `status` and finish with garbage.  This
// makes that visible: the output becomes NaN (nothing is written when status is 0: the kernel costs its launch).
// It also tells the host: sticky (pinned host memory, may be NULL) = {status of the first such launch, how many}.  A
// launch with several outputs is reported once per output; the count is a count of poisoned tensors.
__global__ void poison_if_kernel(const unsigned* __restrict__ status, float* __restrict__ out, size_t n,
                                 unsigned* __restrict__ sticky) {
    const unsigned st = *status;
    if (st == 0u) return;
    if (sticky && blockIdx.x == 0 && threadIdx.x == 0) {
        unsigned expected = 0u;
        (void)__hip_atomic_compare_exchange_strong(sticky, &expected, st, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_SYSTEM);
        (void)__hip_atomic_fetch_add(sticky + 1, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const float nan = __builtin_nanf("");
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = nan;
}
int fsn_launch_poison_if(const unsigned* status, float* out, size_t n, hipStream_t s) {
    if (!status || !out || n == 0) return FSN_OK;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(poison_if_kernel, dim3(blocks), dim3(256), 0, s, status, out, n, fsn_ctx_sticky());
    return fsn_check_launch("poison_if_kernel");
}

template <int AR>
__global__ void to16_kernel(const f32x4* __restrict__ src, fsn_u32x2* __restrict__ dst, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = __builtin_bit_cast(fsn_u32x2, fsn_operand<AR>(src[i]));
}
int fsn_launch_to16(const float* src, void* dst, size_t n, int arith, hipStream_t s) {
    if (n % 4 != 0 || (arith != FSN_ARITH_F16 && arith != FSN_ARITH_BF16)) {
        fsn_set_error("to16: element count %zu must be a multiple of 4, arithmetic fp16 / bf16", n);
        return FSN_ERR_ARG;
    }
    const size_t n4 = n / 4;
    const unsigned blocks = (unsigned)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    if (arith == FSN_ARITH_F16)
        hipLaunchKernelGGL(to16_kernel<FSN_ARITH_F16>, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const f32x4*>(src),
                           static_cast<fsn_u32x2*>(dst), n4);
    else
        hipLaunchKernelGGL(to16_kernel<FSN_ARITH_BF16>, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const f32x4*>(src),
                           static_cast<fsn_u32x2*>(dst), n4);
    return fsn_check_launch("to16_kernel");
}

// Test hook (fsn_debug_hog): a foreign kernel that holds CUs for a while.  `heavy`: every wave keeps ~200 registers
// live (one wave per SIMD then excludes the 216-register group workgroups from that SIMD); LDS is whatever the
// launch asks for dynamically.  Spins on the constant-rate counter until `ticks` have passed.
template <bool HEAVY>
__global__ __launch_bounds__(256) void hog_kernel(unsigned long long ticks, float* sink) {
    extern __shared__ float hog_lds[];
    const unsigned long long t0 = (unsigned long long)wall_clock64();
    float acc[HEAVY ? 192 : 4];
    constexpr int NA = HEAVY ? 192 : 4;
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = (float)(threadIdx.x + i);
    if (threadIdx.x == 0) hog_lds[0] = 1.0f;
    __syncthreads();
    while ((unsigned long long)wall_clock64() - t0 < ticks) {
        const float m = hog_lds[0];
#pragma unroll
        for (int i = 0; i < NA; ++i) acc[i] = __builtin_fmaf(acc[i], m, 1e-9f);
        __builtin_amdgcn_s_sleep(8);
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NA; ++i) sum += acc[i];
    if (sum == 12345.678f) sink[0] = sum;  // keeps the registers live; never true in practice
}
int fsn_launch_hog(int workgroups, int lds_bytes, int heavy, unsigned long long ticks, float* sink, hipStream_t s) {
    if (lds_bytes > 64 * 1024) {  // above 64 KB of dynamic LDS a kernel opts in
        (void)hipFuncSetAttribute(heavy ? (const void*)hog_kernel<true> : (const void*)hog_kernel<false>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        (void)hipGetLastError();
    }
    if (heavy) hipLaunchKernelGGL((hog_kernel<true>), dim3(workgroups), dim3(256), (size_t)lds_bytes, s, ticks, sink);
    else hipLaunchKernelGGL((hog_kernel<false>), dim3(workgroups), dim3(256), (size_t)lds_bytes, s, ticks, sink);
    return fsn_check_launch("hog_kernel");
}
