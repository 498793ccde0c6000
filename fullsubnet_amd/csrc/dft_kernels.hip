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
    hipLaunchKernelGGL(dft_stft_kernel, dim3((unsigned)(B * T), (F + 255) / 256), dim3(256), 3 * N * sizeof(double), s, y,
                       window, re, im, mag, L, T, N, hop, F);
    return fsn_check_launch("dft_stft_kernel");
}

int fsn_launch_dft_istft(const float* re, const float* im, const float* window, float* wframes, float* y, int B, int T,
                         int N, int hop, int length, hipStream_t s) {
    const int F = N / 2 + 1;
    hipLaunchKernelGGL(dft_irfft_kernel, dim3((unsigned)(B * T), (N + 255) / 256), dim3(256),
                       (2 * F + 2 * N) * sizeof(double), s, re, im, window, wframes, T, N, F);
    FSN_TRY_LAUNCH("dft_irfft_kernel");
    const long n = (long)B * length;
    hipLaunchKernelGGL(ola_generic_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, wframes, window, y, B, T,
                       N, hop, length);
    return fsn_check_launch("ola_generic_kernel");
}
