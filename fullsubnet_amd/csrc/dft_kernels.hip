// STFT / iSTFT for transform sizes and hops other than the FullSubNet recipe's 512 / 256
// (audio_zen/acoustics/feature.py:9-91 with other TOML values; improved_fullsubnet/model.py:550-557,
// 582-589 uses 512 / 128 at 16 kHz and 960 / 480 at 48 kHz).  The transform is < 0.1 % of the path's
// work, so the general case is a direct O(N^2) DFT in fp64 from an exact twiddle table in LDS: any even
// N, no radix restrictions, and - like the radix-8 fast path - the result is the correctly rounded
// transform of the fp32 windowed frame (N fp64 rounding errors of 2^-53 each before the final
// rounding to fp32).  ~1.3 GFLOP fp64 per 1000 frames at N = 960: microseconds next to the LSTMs.
#include "fsn_common.h"

namespace {

__device__ __forceinline__ void fill_twiddles(double* c, double* s, int N) {
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        double sv, cv;
        sincospi(2.0 * (double)n / (double)N, &sv, &cv);
        c[n] = cv;
        s[n] = sv;
    }
}

// grid (B * T, ceil(F / 256)); re / im / mag [B][F][T] (any may be NULL)
__global__ __launch_bounds__(256) void dft_stft_kernel(const float* __restrict__ y, const float* __restrict__ window,
                                                       float* __restrict__ re, float* __restrict__ im,
                                                       float* __restrict__ mag, int L, int T, int N, int hop, int F) {
    extern __shared__ double sh[];
    double *x = sh, *c = sh + N, *s = sh + 2 * N;
    const int b = blockIdx.x / T, t = blockIdx.x % T;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        int j = hop * t + n - N / 2;  // centre padding, reflect without repeating the edge
        j = j < 0 ? -j : j;
        j = j >= L ? 2 * (L - 1) - j : j;
        x[n] = (double)(y[(long)b * L + j] * window[n]);  // the frame is rounded to fp32 like ATen's
    }
    fill_twiddles(c, s, N);
    __syncthreads();
    const int f = blockIdx.y * blockDim.x + threadIdx.x;
    if (f >= F) return;
    double ar = 0.0, ai = 0.0;
    int idx = 0;
    for (int n = 0; n < N; ++n) {
        const double xv = x[n];
        ar = fma(xv, c[idx], ar);
        ai = fma(-xv, s[idx], ai);
        idx += f;
        idx = idx >= N ? idx - N : idx;
    }
    const float fr = (float)ar, fi = (float)ai;
    const long o = ((long)b * F + f) * T + t;
    if (re) re[o] = fr;
    if (im) im[o] = fi;
    if (mag) mag[o] = (float)sqrt((double)fr * fr + (double)fi * fi);
}

// grid (B * T, ceil(N / 256)); re / im [B][F][T] -> windowed time frames wframes [B][T][N]
__global__ __launch_bounds__(256) void dft_irfft_kernel(const float* __restrict__ re, const float* __restrict__ im,
                                                        const float* __restrict__ window, float* __restrict__ wframes,
                                                        int T, int N, int F) {
    extern __shared__ double sh[];
    double *xr = sh, *xi = sh + F, *c = sh + 2 * F, *s = sh + 2 * F + N;
    const int b = blockIdx.x / T, t = blockIdx.x % T;
    for (int f = threadIdx.x; f < F; f += blockDim.x) {
        const long o = ((long)b * F + f) * T + t;
        const bool edge = f == 0 || f == N / 2;  // C2R: imaginary parts of DC / Nyquist are ignored
        const double k = edge ? 1.0 : 2.0;
        xr[f] = k * (double)re[o];
        xi[f] = edge ? 0.0 : k * (double)im[o];
    }
    fill_twiddles(c, s, N);
    __syncthreads();
    const int n = blockIdx.y * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double acc = 0.0;
    int idx = 0;
    for (int f = 0; f < F; ++f) {
        acc = fma(xr[f], c[idx], acc);
        acc = fma(-xi[f], s[idx], acc);
        idx += n;
        idx = idx >= N ? idx - N : idx;
    }
    const float v = (float)(acc / (double)N);
    wframes[((long)b * T + t) * N + n] = v * window[n];
}

// ---- two-level form: N = P Q -------------------------------------------------------------------------------------
// The direct transform above costs N^2 / 2 multiply-adds per frame and rebuilds its twiddle table (N sincospi in fp64)
// in every workgroup: 10.8 % of Improved FullSubNet's step at 48 kHz (960 / 480; profiles/r02_kernel_stats_improved48_b32.md).
// Any composite N splits once, Cooley-Tukey style, into Q transforms of length P and P of length Q, each still direct,
// still fp64, still from the exact N-entry table (W_P^a = W_N^{Q a}, W_Q^b = W_N^{P b}): N (P + Q) instead of N^2 -
// 62 N at 960 = 30 x 32 - and a workgroup builds the table ONCE and walks over many frames.  No radix-specific code:
// 400 = 20 x 20, 512 = 16 x 32, 960 = 30 x 32, 1536 = 32 x 48 all take this path.
//   forward, n = Q n1 + n2, k = k1 + P k2:
//     Bt[n2][k1] = W_N^{n2 k1} sum_{n1} x[Q n1 + n2] W_N^{Q n1 k1}         (x real: 2 fma per term)
//     X[k1 + P k2] = sum_{n2} Bt[n2][k1] W_N^{P n2 k2}                       (k <= N / 2 only)
//   inverse (C2R), the same index split with conjugate twiddles:
//     Ct[n2][k1] = conj(W_N^{n2 k1}) sum_{k2} X[k1 + P k2] conj(W_N^{P n2 k2}), X completed by Hermitian symmetry
//     x[Q n1 + n2] = Re sum_{k1} Ct[n2][k1] conj(W_N^{Q n1 k1})              (2 fma per term)
// LDS: table 2 N + frame N (2 N complex for the inverse) + intermediate 2 N doubles: 38 / 46 KB at N = 960.
__global__ __launch_bounds__(256) void dft2_stft_kernel(const float* __restrict__ y, const float* __restrict__ window,
                                                        float* __restrict__ re, float* __restrict__ im,
                                                        float* __restrict__ mag, int L, int T, int N, int hop, int F, int P,
                                                        int Q, long frames) {
    extern __shared__ double sh[];
    double *c = sh, *s = sh + N, *x = sh + 2 * N, *br = sh + 3 * N, *bi = sh + 4 * N;
    fill_twiddles(c, s, N);
    for (long fr = blockIdx.x; fr < frames; fr += gridDim.x) {
        const int b = (int)(fr / T), t = (int)(fr % T);
        __syncthreads();  // table ready / previous frame's stage 2 done with x and Bt
        for (int n = threadIdx.x; n < N; n += blockDim.x) {
            int j = hop * t + n - N / 2;  // centre padding, reflect without repeating the edge
            j = j < 0 ? -j : j;
            j = j >= L ? 2 * (L - 1) - j : j;
            x[n] = (double)(y[(long)b * L + j] * window[n]);  // the frame is rounded to fp32 like ATen's
        }
        __syncthreads();
        for (int i = threadIdx.x; i < N; i += blockDim.x) {
            const int k1 = i / Q, n2 = i - k1 * Q;
            double ar = 0.0, ai = 0.0;
            int idx = 0;
            const int step = (int)(((long)Q * k1) % N);
            for (int n1 = 0; n1 < P; ++n1) {
                const double xv = x[Q * n1 + n2];
                ar = fma(xv, c[idx], ar);
                ai = fma(-xv, s[idx], ai);
                idx += step;
                idx = idx >= N ? idx - N : idx;
            }
            const int tw = (int)(((long)n2 * k1) % N);
            const double wr = c[tw], wi = -s[tw];
            br[n2 * P + k1] = ar * wr - ai * wi;
            bi[n2 * P + k1] = ar * wi + ai * wr;
        }
        __syncthreads();
        for (int k = threadIdx.x; k < F; k += blockDim.x) {
            const int k2 = k / P, k1 = k - k2 * P;
            double ar = 0.0, ai = 0.0;
            int idx = 0;
            const int step = (int)(((long)P * k2) % N);
            for (int n2 = 0; n2 < Q; ++n2) {
                const double vr = br[n2 * P + k1], vi = bi[n2 * P + k1], wr = c[idx], wi = -s[idx];
                ar = fma(vr, wr, ar);
                ar = fma(-vi, wi, ar);
                ai = fma(vr, wi, ai);
                ai = fma(vi, wr, ai);
                idx += step;
                idx = idx >= N ? idx - N : idx;
            }
            const float fr32 = (float)ar, fi32 = (float)ai;
            const long o = ((long)b * F + k) * T + t;
            if (re) re[o] = fr32;
            if (im) im[o] = fi32;
            if (mag) mag[o] = (float)sqrt((double)fr32 * fr32 + (double)fi32 * fi32);
        }
    }
}

__global__ __launch_bounds__(256) void dft2_irfft_kernel(const float* __restrict__ re, const float* __restrict__ im,
                                                         const float* __restrict__ window, float* __restrict__ wframes,
                                                         int T, int N, int F, int P, int Q, long frames) {
    extern __shared__ double sh[];
    double *c = sh, *s = sh + N, *xr = sh + 2 * N, *xi = sh + 3 * N, *cr = sh + 4 * N, *ci = sh + 5 * N;
    fill_twiddles(c, s, N);
    for (long fr = blockIdx.x; fr < frames; fr += gridDim.x) {
        const int b = (int)(fr / T), t = (int)(fr % T);
        __syncthreads();
        for (int f = threadIdx.x; f < F; f += blockDim.x) {
            const long o = ((long)b * F + f) * T + t;
            const bool edge = f == 0 || f == N / 2;  // C2R: imaginary parts of DC / Nyquist are ignored
            const double vr = (double)re[o], vi = edge ? 0.0 : (double)im[o];
            xr[f] = vr;
            xi[f] = vi;
            if (!edge) {  // X[N - f] = conj(X[f])
                xr[N - f] = vr;
                xi[N - f] = -vi;
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < N; i += blockDim.x) {
            const int k1 = i / Q, n2 = i - k1 * Q;
            double ar = 0.0, ai = 0.0;
            int idx = 0;
            const int step = (int)(((long)P * n2) % N);
            for (int k2 = 0; k2 < Q; ++k2) {
                const double vr = xr[k1 + P * k2], vi = xi[k1 + P * k2], wr = c[idx], wi = s[idx];  // conj(W) = c + i s
                ar = fma(vr, wr, ar);
                ar = fma(-vi, wi, ar);
                ai = fma(vr, wi, ai);
                ai = fma(vi, wr, ai);
                idx += step;
                idx = idx >= N ? idx - N : idx;
            }
            const int tw = (int)(((long)n2 * k1) % N);
            const double wr = c[tw], wi = s[tw];
            cr[n2 * P + k1] = ar * wr - ai * wi;
            ci[n2 * P + k1] = ar * wi + ai * wr;
        }
        __syncthreads();
        for (int n = threadIdx.x; n < N; n += blockDim.x) {
            const int n1 = n / Q, n2 = n - n1 * Q;
            double acc = 0.0;
            int idx = 0;
            const int step = (int)(((long)Q * n1) % N);
            for (int k1 = 0; k1 < P; ++k1) {
                acc = fma(cr[n2 * P + k1], c[idx], acc);  // Re (Ct conj(W))
                acc = fma(-ci[n2 * P + k1], s[idx], acc);
                idx += step;
                idx = idx >= N ? idx - N : idx;
            }
            const float v = (float)(acc / (double)N);
            wframes[((long)b * T + t) * N + n] = v * window[n];
        }
    }
}

// the split with P <= Q, both > 3, P as large as possible; 0: N is (twice) a prime or too large for the LDS: direct form
int dft2_factor(int N) {
    if (N > 2048) return 0;
    int best = 0;
    for (int p = 4; (long)p * p <= N; ++p)
        if (N % p == 0) best = p;
    return best;
}

// overlap-add (ascending frame order, fp32), division by the overlap-added squared window, centre
// trim and length handling of torch.istft
__global__ __launch_bounds__(256) void ola_generic_kernel(const float* __restrict__ wframes,
                                                          const float* __restrict__ window, float* __restrict__ y,
                                                          int B, int T, int N, int hop, int length) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)B * length) return;
    const int b = (int)(gid / length), j = (int)(gid % length);
    const int p = j + N / 2;
    const long total = (long)N + (long)hop * (T - 1);
    float out = 0.f;
    if (p < total) {
        int t0 = p - N + 1;
        t0 = t0 <= 0 ? 0 : (t0 + hop - 1) / hop;
        int t1 = p / hop;
        t1 = t1 < T ? t1 : T - 1;
        float acc = 0.f, env = 0.f;
        for (int t = t0; t <= t1; ++t) {
            const int n = p - hop * t;
            const float w = window[n];
            acc = acc + wframes[((long)b * T + t) * N + n];
            env = env + w * w;
        }
        out = acc / env;
    }
    y[gid] = out;
}

}  // namespace

int fsn_launch_dft_stft(const float* y, int B, int L, const float* window, float* re, float* im, float* mag, int T,
                        int N, int hop, hipStream_t s) {
    const int F = N / 2 + 1;
    if (const int P = dft2_factor(N)) {
        const long frames = (long)B * T;
        const unsigned grid = (unsigned)(frames < 1024 ? frames : 1024);
        if (5 * N * sizeof(double) > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dft2_stft_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)(5 * N * sizeof(double)));
        hipLaunchKernelGGL(dft2_stft_kernel, dim3(grid), dim3(256), 5 * N * sizeof(double), s, y, window, re, im, mag, L, T, N,
                           hop, F, P, N / P, frames);
        return fsn_check_launch("dft2_stft_kernel");
    }
    hipLaunchKernelGGL(dft_stft_kernel, dim3((unsigned)(B * T), (F + 255) / 256), dim3(256), 3 * N * sizeof(double), s, y,
                       window, re, im, mag, L, T, N, hop, F);
    return fsn_check_launch("dft_stft_kernel");
}

int fsn_launch_dft_istft(const float* re, const float* im, const float* window, float* wframes, float* y, int B, int T,
                         int N, int hop, int length, hipStream_t s) {
    const int F = N / 2 + 1;
    if (const int P = dft2_factor(N)) {
        const long frames = (long)B * T;
        const unsigned grid = (unsigned)(frames < 1024 ? frames : 1024);
        if (6 * N * sizeof(double) > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dft2_irfft_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)(6 * N * sizeof(double)));
        hipLaunchKernelGGL(dft2_irfft_kernel, dim3(grid), dim3(256), 6 * N * sizeof(double), s, re, im, window, wframes, T, N, F,
                           P, N / P, frames);
        FSN_TRY_LAUNCH("dft2_irfft_kernel");
    } else {
        hipLaunchKernelGGL(dft_irfft_kernel, dim3((unsigned)(B * T), (N + 255) / 256), dim3(256),
                           (2 * F + 2 * N) * sizeof(double), s, re, im, window, wframes, T, N, F);
        FSN_TRY_LAUNCH("dft_irfft_kernel");
    }
    const long n = (long)B * length;
    hipLaunchKernelGGL(ola_generic_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, wframes, window, y, B, T,
                       N, hop, length);
    return fsn_check_launch("ola_generic_kernel");
}
