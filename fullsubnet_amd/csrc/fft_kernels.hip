// STFT / iSTFT kernels for gfx950.
//
// Replaces torch.stft / torch.istft as called from audio_zen/acoustics/feature.py:33-40,84-91
// (n_fft = win_length = 512, hop = 256, periodic Hann, center=True / reflect, onesided).
//
// Design: one 64-lane wavefront transforms TWO real frames with ONE 512-point complex FFT
// (frame A in the real part, frame B in the imaginary part, split afterwards through the
// Hermitian symmetry), 8 points per lane, three radix-8 Stockham passes through LDS.  The DFT is
// evaluated in fp64 (twiddles from sincospi): the frame*window product is rounded to fp32 exactly
// as ATen does, and the transform of that product is then correct to < 1 fp32 ULP, which is what
// the "within 2 ULP" budget of the north star is spent against (MKL's own fp32 FFT is up to 3 ULP
// from this value at frame-max scale).  The kernels are HBM-bound streaming kernels;
// fp64 costs nothing that matters (a 512-point FFT per 2 KB of input).
#include "fsn_common.h"

namespace {

struct cd {
    double x, y;
};
__device__ __forceinline__ cd cadd(cd a, cd b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cd csub(cd a, cd b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cd cmul(cd a, cd b) {
    return {fma(a.x, b.x, -(a.y * b.y)), fma(a.x, b.y, a.y * b.x)};
}
// multiply by -i (forward) or +i (inverse)
template <bool INV>
__device__ __forceinline__ cd mul_mi(cd a) {
    return INV ? cd{-a.y, a.x} : cd{a.y, -a.x};
}
#define FFT2(a, b)         \
    {                      \
        cd _t = a;         \
        a = cadd(_t, b);   \
        b = csub(_t, b);   \
    }

// 8-point DFT, decimation in frequency; result in natural order in v[0..7].
template <bool INV>
__device__ __forceinline__ void fft8(cd* v) {
    const double h = 0.70710678118654752440;
    FFT2(v[0], v[4]);
    FFT2(v[1], v[5]);
    FFT2(v[2], v[6]);
    FFT2(v[3], v[7]);
    // v5 *= e^{-+i pi/4}, v6 *= -+i, v7 *= e^{-+3i pi/4}
    if (!INV) {
        v[5] = cd{(v[5].x + v[5].y) * h, (v[5].y - v[5].x) * h};
        v[7] = cd{(v[7].y - v[7].x) * h, -(v[7].x + v[7].y) * h};
    } else {
        v[5] = cd{(v[5].x - v[5].y) * h, (v[5].y + v[5].x) * h};
        v[7] = cd{-(v[7].x + v[7].y) * h, (v[7].x - v[7].y) * h};
    }
    v[6] = mul_mi<INV>(v[6]);
    FFT2(v[0], v[2]);
    FFT2(v[1], v[3]);
    FFT2(v[4], v[6]);
    FFT2(v[5], v[7]);
    v[3] = mul_mi<INV>(v[3]);
    v[7] = mul_mi<INV>(v[7]);
    FFT2(v[0], v[1]);
    FFT2(v[2], v[3]);
    FFT2(v[4], v[5]);
    FFT2(v[6], v[7]);
    // bit-reversed -> natural
    cd t1 = v[1], t3 = v[3], t4 = v[4], t6 = v[6];
    v[1] = t4;
    v[4] = t1;
    v[3] = t6;
    v[6] = t3;
}

// LDS index of point i: one double of padding per eight.  The radix-8 passes store with strides of 8 and 64 doubles
// (16- and 8-way bank conflicts on 8-byte accesses unpadded); with the padding a half-wave's stores fall on distinct banks.
__device__ __forceinline__ int lpad(int i) { return i + (i >> 3); }
constexpr int kLdsPoints = 512 + 64;

// One Stockham radix-8 pass of a 512-point FFT held by one wave (lane j owns butterfly j).
// v holds in[j + 64 r]; on return the outputs are written to (sre, sim) in the pass's order.
template <bool INV, int NS>
__device__ __forceinline__ void pass8(cd* v, double* sre, double* sim, int j) {
    const int k = j % NS;
    if (NS > 1) {
        // twiddle r = w^r with w = e^{-+ 2 pi i k / (8 NS)}: ONE sincospi per lane and pass, the powers by fp64 products
        // (a few 1e-16 relative: 2^-29 of an fp32 ULP).  Seven sincospi per pass made these kernels compute-bound on the
        // fp64 vector pipe (35 us for 49 MB, round 6); the transform is a streaming stage.
        double s, c;
        sincospi((INV ? 1.0 : -1.0) * (double)k / (double)(4 * NS), &s, &c);
        const cd w1{c, s};
        const cd w2 = cmul(w1, w1), w3 = cmul(w2, w1), w4 = cmul(w2, w2);
        const cd w5 = cmul(w4, w1), w6 = cmul(w3, w3), w7 = cmul(w4, w3);
        v[1] = cmul(v[1], w1);
        v[2] = cmul(v[2], w2);
        v[3] = cmul(v[3], w3);
        v[4] = cmul(v[4], w4);
        v[5] = cmul(v[5], w5);
        v[6] = cmul(v[6], w6);
        v[7] = cmul(v[7], w7);
    }
    fft8<INV>(v);
    const int j0 = (j / NS) * NS * 8 + k;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        sre[lpad(j0 + r * NS)] = v[r].x;
        sim[lpad(j0 + r * NS)] = v[r].y;
    }
}

__device__ __forceinline__ void load8(cd* v, const double* sre, const double* sim, int j) {
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = cd{sre[lpad(j + 64 * r)], sim[lpad(j + 64 * r)]};
}

constexpr int kWavesPerBlock = 4;

// A wave exchanges its points through ITS OWN LDS region: LDS instructions of one wave execute in order, so the passes
// need no workgroup barrier - only the compiler must keep the order (fences at wavefront scope emit no instruction).
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------------------------
// y [B][L] -> re, im [B][T][FP] (frame-major) or [B][F][T] (reference layout); mag likewise, with
// Tp >= T frames in the frame-major layout (frames >= T are the look-ahead zeros of
// fullsubnet/model.py:85).  Columns F..FP-1 of the frame-major rows are written as zeros.
// ---------------------------------------------------------------------------------------------
template <bool FRAME_MAJOR>
__global__ __launch_bounds__(256) void stft_kernel(const float* __restrict__ y, const float* __restrict__ window,
                                                   float* __restrict__ re, float* __restrict__ im,
                                                   float* __restrict__ mag, int B, int L, int T, int Tp, int F,
                                                   int FP) {
    __shared__ double lds[kWavesPerBlock][2][kLdsPoints];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pairs_per_b = (Tp + 1) >> 1;
    const long p = (long)blockIdx.x * kWavesPerBlock + wave;
    const int b = (int)(p / pairs_per_b);
    const int tA = 2 * (int)(p % pairs_per_b), tB = tA + 1;
    const bool live = b < B;
    double* sre = lds[wave][0];
    double* sim = lds[wave][1];

    cd v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int n = lane + 64 * r;
        const float w = window[n];
        float xa = 0.f, xb = 0.f;
        if (live && tA < T) {
            int j = 256 * tA + n - 256;
            j = j < 0 ? -j : j;
            j = j >= L ? 2 * (L - 1) - j : j;
            xa = y[(long)b * L + j] * w;
        }
        if (live && tB < T) {
            int j = 256 * tB + n - 256;
            j = j < 0 ? -j : j;
            j = j >= L ? 2 * (L - 1) - j : j;
            xb = y[(long)b * L + j] * w;
        }
        v[r] = cd{(double)xa, (double)xb};
    }
    pass8<false, 1>(v, sre, sim, lane);
    wave_lds_sync();
    load8(v, sre, sim, lane);
    wave_lds_sync();
    pass8<false, 8>(v, sre, sim, lane);
    wave_lds_sync();
    load8(v, sre, sim, lane);
    wave_lds_sync();
    pass8<false, 64>(v, sre, sim, lane);
    wave_lds_sync();

    if (!live) return;
    // split: X_A[k] = (Z[k] + conj Z[N-k]) / 2 ; X_B[k] = (Z[k] - conj Z[N-k]) / (2i)
#pragma unroll
    for (int m = 0; m < 5; ++m) {
        const int k = lane + 64 * m;
        if (k >= FP) break;
        float ar = 0.f, ai = 0.f, br = 0.f, bi = 0.f;
        if (k < F) {
            const int kn = (512 - k) & 511;
            const double zr = sre[lpad(k)], zi = sim[lpad(k)], wr = sre[lpad(kn)], wi = sim[lpad(kn)];
            ar = (float)(0.5 * (zr + wr));
            ai = (float)(0.5 * (zi - wi));
            br = (float)(0.5 * (zi + wi));
            bi = (float)(-0.5 * (zr - wr));
        }
        const float am = (float)sqrt((double)ar * ar + (double)ai * ai);
        const float bm = (float)sqrt((double)br * br + (double)bi * bi);
        if (FRAME_MAJOR) {
            if (tA < T) {
                const long o = ((long)b * T + tA) * FP + k;
                if (re) re[o] = ar;
                if (im) im[o] = ai;
            }
            if (tB < T) {
                const long o = ((long)b * T + tB) * FP + k;
                if (re) re[o] = br;
                if (im) im[o] = bi;
            }
            if (mag) {
                if (tA < Tp) mag[((long)b * Tp + tA) * FP + k] = am;  // zero for tA >= T
                if (tB < Tp) mag[((long)b * Tp + tB) * FP + k] = bm;
            }
        } else if (k < F) {
            if (tA < T) {
                const long o = ((long)b * F + k) * T + tA;
                if (re) re[o] = ar;
                if (im) im[o] = ai;
                if (mag) mag[o] = am;
            }
            if (tB < T) {
                const long o = ((long)b * F + k) * T + tB;
                if (re) re[o] = br;
                if (im) im[o] = bi;
                if (mag) mag[o] = bm;
            }
        }
    }
}

// mask.py:47-64 decompress_cIRM(K = 10, limit = 9.9) in fp32, written like the reference.
__device__ __forceinline__ float decompress1(float m) {
    const float lim = 9.9f;
    m = m >= lim ? lim : (m <= -lim ? -lim : m);
    return -10.0f * logf((10.0f - m) / (10.0f + m));
}

// ---------------------------------------------------------------------------------------------
// (decompress cIRM, complex mask) + inverse real FFT + synthesis window for a pair of frames:
// re/im (and optionally crm_r/crm_i) -> wframes [B][T][512] = irfft(S) * window, rounded to fp32
// after the irfft and after the window product like ATen does.
// inferencer.py:137-140 + feature.py:84-91 (torch.istft up to the overlap-add).
// ---------------------------------------------------------------------------------------------
template <bool FRAME_MAJOR>
__global__ __launch_bounds__(256) void mask_irfft_kernel(const float* __restrict__ re, const float* __restrict__ im,
                                                         const float* __restrict__ crm_r,
                                                         const float* __restrict__ crm_i,
                                                         const float* __restrict__ window,
                                                         float* __restrict__ wframes, int B, int T, int F, int FP) {
    __shared__ double lds[kWavesPerBlock][2][kLdsPoints];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pairs_per_b = (T + 1) >> 1;
    const long p = (long)blockIdx.x * kWavesPerBlock + wave;
    const int b = (int)(p / pairs_per_b);
    const int tA = 2 * (int)(p % pairs_per_b), tB = tA + 1;
    const bool live = b < B;
    double* sre = lds[wave][0];
    double* sim = lds[wave][1];

#pragma unroll
    for (int m = 0; m < 5; ++m) {
        const int k = lane + 64 * m;
        if (k > 256) break;
        float s[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int t = q ? tB : tA;
            if (live && t < T) {
                const long o = FRAME_MAJOR ? ((long)b * T + t) * FP + k : ((long)b * F + k) * T + t;
                const float xr = re[o], xi = im[o];
                if (crm_r) {
                    const float mr = decompress1(crm_r[o]), mi = decompress1(crm_i[o]);
                    s[q][0] = mr * xr - mi * xi;
                    s[q][1] = mi * xr + mr * xi;
                } else {
                    s[q][0] = xr;
                    s[q][1] = xi;
                }
            }
        }
        if (k == 0 || k == 256) {  // C2R ignores the imaginary part of DC / Nyquist
            s[0][1] = 0.f;
            s[1][1] = 0.f;
        }
        const double ar = s[0][0], ai = s[0][1], br = s[1][0], bi = s[1][1];
        sre[lpad(k)] = ar - bi;
        sim[lpad(k)] = ai + br;
        if (k > 0 && k < 256) {
            sre[lpad(512 - k)] = ar + bi;
            sim[lpad(512 - k)] = br - ai;
        }
    }
    wave_lds_sync();
    cd v[8];
    load8(v, sre, sim, lane);
    wave_lds_sync();
    pass8<true, 1>(v, sre, sim, lane);
    wave_lds_sync();
    load8(v, sre, sim, lane);
    wave_lds_sync();
    pass8<true, 8>(v, sre, sim, lane);
    wave_lds_sync();
    load8(v, sre, sim, lane);
    wave_lds_sync();
    pass8<true, 64>(v, sre, sim, lane);
    wave_lds_sync();
    if (!live) return;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int n = lane + 64 * r;
        const float w = window[n];
        const float fa = (float)(sre[lpad(n)] * (1.0 / 512.0));
        const float fb = (float)(sim[lpad(n)] * (1.0 / 512.0));
        if (tA < T) wframes[((long)b * T + tA) * 512 + n] = fa * w;
        if (tB < T) wframes[((long)b * T + tB) * 512 + n] = fb * w;
    }
}

// Overlap-add of the windowed frames, division by the overlap-added squared window, centre trim
// and length handling of torch.istft (feature.py:84-91).  y [B][length].
__global__ __launch_bounds__(256) void ola_kernel(const float* __restrict__ wframes,
                                                  const float* __restrict__ window, float* __restrict__ y, int B,
                                                  int T, int length) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)B * length) return;
    const int b = (int)(gid / length), j = (int)(gid % length);
    const int p = j + 256;
    const int total = 512 + 256 * (T - 1);
    float out = 0.f;
    if (p < total) {
        const int t_hi = p >> 8, t_lo = t_hi - 1;
        float acc = 0.f, env = 0.f;
        if (t_lo >= 0 && t_lo < T) {
            const int n = p - 256 * t_lo;
            const float w = window[n];
            acc = wframes[((long)b * T + t_lo) * 512 + n];
            env = w * w;
        }
        if (t_hi < T) {
            const int n = p - 256 * t_hi;
            const float w = window[n];
            acc = acc + wframes[((long)b * T + t_hi) * 512 + n];
            env = env + w * w;
        }
        out = acc / env;
    }
    y[gid] = out;
}

}  // namespace

int fsn_launch_stft(const float* y, int B, int L, const float* window, float* re, float* im, float* mag, int T,
                    int Tp, int F, int FP, bool frame_major, hipStream_t s) {
    const long pairs = (long)B * ((Tp + 1) / 2);
    const unsigned grid = (unsigned)((pairs + kWavesPerBlock - 1) / kWavesPerBlock);
    if (frame_major)
        hipLaunchKernelGGL(stft_kernel<true>, dim3(grid), dim3(256), 0, s, y, window, re, im, mag, B, L, T, Tp, F,
                           FP);
    else
        hipLaunchKernelGGL(stft_kernel<false>, dim3(grid), dim3(256), 0, s, y, window, re, im, mag, B, L, T, Tp, F,
                           FP);
    return fsn_check_launch("stft_kernel");
}

int fsn_launch_mask_irfft(const float* re, const float* im, const float* crm_r, const float* crm_i, int B, int T,
                          int F, int FP, bool frame_major, const float* window, float* wframes, hipStream_t s) {
    const long pairs = (long)B * ((T + 1) / 2);
    const unsigned grid = (unsigned)((pairs + kWavesPerBlock - 1) / kWavesPerBlock);
    if (frame_major)
        hipLaunchKernelGGL(mask_irfft_kernel<true>, dim3(grid), dim3(256), 0, s, re, im, crm_r, crm_i, window,
                           wframes, B, T, F, FP);
    else
        hipLaunchKernelGGL(mask_irfft_kernel<false>, dim3(grid), dim3(256), 0, s, re, im, crm_r, crm_i, window,
                           wframes, B, T, F, FP);
    return fsn_check_launch("mask_irfft_kernel");
}

int fsn_launch_ola(const float* wframes, const float* window, int B, int T, int length, float* y, hipStream_t s) {
    const long n = (long)B * length;
    const unsigned grid = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(ola_kernel, dim3(grid), dim3(256), 0, s, wframes, window, y, B, T, length);
    return fsn_check_launch("ola_kernel");
}
