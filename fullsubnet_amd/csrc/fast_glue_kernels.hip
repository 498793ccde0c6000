// Fast FullSubNet's tensor glue (recipes/dns_interspeech_2020/fast_fullsubnet/model.py:108-140, 143-202) as HIP kernels:
// everything between the model's LSTM / Linear blocks - look-ahead pad, the norm in front of the encoder, the unit
// windows of the mel spectrum and of the encoder output, real_time_downsampling, the norm in front of the bottleneck,
// real_time_upsampling, the two concatenations, the final reshape + look-ahead slice.
//
// As tensor algebra this was ~30 ATen launches per forward (gather, cats, two mean reductions, mul / div, fills and ten
// transposing copies between [B, C, T] and the time-major layout the LSTM entries take): 1.2 ms of a 51 ms step at batch
// 256.  Here every tensor between the blocks stays TIME-MAJOR ([T][rows][columns]: a block's output is the next block's
// input as it lies), and the glue is seven kernels:
//   spec_rows       [B][F][T0] magnitude -> [T][Bp][Fp] rows of the mel product (T = T0 + look_ahead zero frames)
//   mean / scale    x / (mean over an utterance's (band, frame) values + 1e-5): offline_laplace_norm, base_model.py:204-218
//   ds              frame 0 kept, then means over consecutive blocks of `shrink` frames (a shorter last block over what it
//                   has) of both sources, and the utterance's mean of the UNFOLDED, down-sampled tensor as a multiplicity-
//                   weighted sum of these (fp64, rounded once like fsn_norm) - the unfolded tensor is never formed
//   units           out[ts][b M + m][w] = ds(src(w))[ts][b][reflect(m - n + w')] / (mean_b + 1e-5), zero-padded columns
//   decoder_input   out[t][b] = enc[t][b] | bottleneck output held for `shrink` frames (frame t takes low-rate frame t / s)
//   mask_out        [T][Bp][2F] -> [B][2][F][T - look_ahead] (the first look_ahead frames dropped)
// Statistics in fp64 with a fixed summation order: two runs are bit-identical.
#include "fsn_common.h"
#include "../../include/fsn_hip.h"

namespace {

constexpr float kFastEps = 1e-5f;  // offline_laplace_norm's epsilon

__device__ __forceinline__ int fast_reflect(int j, int M) {
    j = j < 0 ? -j : j;
    return j > M - 1 ? 2 * (M - 1) - j : j;
}

// fixed-order sum of one fp64 value per thread of a 256-thread workgroup; the result in every thread
__device__ __forceinline__ double fast_block_sum(double v, double* sh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    return ((sh[0] + sh[1]) + sh[2]) + sh[3];
}

// [B][F][T0] -> [T][Bp][Fp]; zero beyond (T0, B, F): transposing copy through LDS, both sides coalesced
__global__ __launch_bounds__(256) void fast_spec_rows_kernel(const float* __restrict__ mag, float* __restrict__ rows, int B,
                                                             int F, int T0, int T, int Bp, int Fp) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, f0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int j = ty; j < 32; j += 8) {
        const int f = f0 + j, t = t0 + tx;
        tile[j][tx] = (b < B && f < F && t < T0) ? mag[((size_t)b * F + f) * T0 + t] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = ty; j < 32; j += 8) {
        const int t = t0 + j, f = f0 + tx;
        if (t < T && f < Fp) rows[((size_t)t * Bp + b) * Fp + f] = tile[tx][j];
    }
}

// den[b] = float(sum of x[t][b][c] / (T C)) + eps, one workgroup per utterance
__global__ __launch_bounds__(256) void fast_mean_kernel(const float* __restrict__ x, float* __restrict__ den, int T, int Bp,
                                                        int C) {
    __shared__ double sh[4];
    const int b = blockIdx.x;
    double s = 0.0;
    const long n = (long)T * C;
    for (long p = threadIdx.x; p < n; p += 256) {
        const long t = p / C, c = p % C;
        s += (double)x[(t * Bp + b) * C + c];
    }
    const double tot = fast_block_sum(s, sh);
    if (threadIdx.x == 0) den[b] = (float)(tot / (double)n) + kFastEps;
}

// out = x / den[b], rows beyond B zero
__global__ __launch_bounds__(256) void fast_scale_kernel(const float* __restrict__ x, const float* __restrict__ den,
                                                         float* __restrict__ out, long n, int B, int Bp, int C) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int b = (int)((i / C) % Bp);
    out[i] = b < B ? x[i] / den[b] : 0.f;
}

struct FastUnits {
    int T, Ts, B, Bp, M, n_mel, n_enc, shrink;
};

// ds[src][ts][b][m] and den[b]: one workgroup per utterance
__global__ __launch_bounds__(256) void fast_ds_kernel(const float* __restrict__ mel, const float* __restrict__ enc, long ld_enc,
                                                      float* __restrict__ ds, float* __restrict__ den, const FastUnits a) {
    extern __shared__ int mult[];  // [2][M]: how many (unit, window column) pairs read band m
    __shared__ double sh[4];
    const int b = blockIdx.x, M = a.M;
    for (int i = threadIdx.x; i < 2 * M; i += 256) mult[i] = 0;
    __syncthreads();
    const int w_mel = 2 * a.n_mel + 1, w_enc = 2 * a.n_enc + 1;
    for (int p = threadIdx.x; p < M * (w_mel + w_enc); p += 256) {
        const int u = p / (w_mel + w_enc), w = p % (w_mel + w_enc);
        if (w < w_mel) atomicAdd(&mult[fast_reflect(u - a.n_mel + w, M)], 1);
        else atomicAdd(&mult[M + fast_reflect(u - a.n_enc + (w - w_mel), M)], 1);
    }
    __syncthreads();
    double s = 0.0;
    const size_t half = (size_t)a.Ts * a.B * M;
    for (int p = threadIdx.x; p < a.Ts * M; p += 256) {
        const int ts = p / M, m = p % M;
        const int f0 = ts == 0 ? 0 : 1 + (ts - 1) * a.shrink;
        int f1 = ts == 0 ? 1 : f0 + a.shrink;
        f1 = f1 < a.T ? f1 : a.T;
        float sm = 0.f, se = 0.f;
        for (int t = f0; t < f1; ++t) {
            sm += mel[((size_t)t * a.Bp + b) * M + m];
            se += enc[((size_t)t * a.Bp + b) * ld_enc + m];
        }
        const float factor = 1.f / (float)(f1 - f0);  // torch.mean: sum x (1 / count)
        const float vm = f1 - f0 > 1 ? sm * factor : sm, ve = f1 - f0 > 1 ? se * factor : se;
        ds[((size_t)ts * a.B + b) * M + m] = vm;
        ds[half + ((size_t)ts * a.B + b) * M + m] = ve;
        s += (double)mult[m] * (double)vm + (double)mult[M + m] * (double)ve;
    }
    const double tot = fast_block_sum(s, sh);
    if (threadIdx.x == 0) den[b] = (float)(tot / ((double)M * (w_mel + w_enc) * a.Ts)) + kFastEps;
}

// out[ts][b M + m][w]: one workgroup per (ts, utterance); rows beyond B M are zeroed by the last utterance's workgroup
__global__ __launch_bounds__(256) void fast_units_kernel(const float* __restrict__ ds, const float* __restrict__ den,
                                                         float* __restrict__ out, const FastUnits a, int Np, int Wp) {
    extern __shared__ float band[];  // [2][M]
    const int ts = blockIdx.x, b = blockIdx.y, M = a.M;
    const size_t half = (size_t)a.Ts * a.B * M;
    for (int i = threadIdx.x; i < 2 * M; i += 256)
        band[i] = ds[(i < M ? 0 : half) + ((size_t)ts * a.B + b) * M + (i < M ? i : i - M)];
    __syncthreads();
    const float d = den[b];
    const int w_mel = 2 * a.n_mel + 1, w_enc = 2 * a.n_enc + 1;
    float* o = out + ((size_t)ts * Np + (size_t)b * M) * Wp;
    for (int p = threadIdx.x; p < M * Wp; p += 256) {
        const int m = p / Wp, w = p % Wp;
        float v = 0.f;
        if (w < w_mel) v = band[fast_reflect(m - a.n_mel + w, M)] / d;
        else if (w < w_mel + w_enc) v = band[M + fast_reflect(m - a.n_enc + (w - w_mel), M)] / d;
        o[p] = v;
    }
    if (b == a.B - 1) {
        const size_t pad = (size_t)(Np - a.B * M) * Wp;
        float* z = out + ((size_t)ts * Np + (size_t)a.B * M) * Wp;
        for (size_t p = threadIdx.x; p < pad; p += 256) z[p] = 0.f;
    }
}

// out[t][b][0 .. M) = enc[t][b]; out[t][b][M .. 2M) = slow[t / shrink][b M + m]; rows beyond B zero
__global__ __launch_bounds__(256) void fast_decoder_input_kernel(const float* __restrict__ enc, long ld_enc,
                                                                 const float* __restrict__ slow, long ld_slow_t, long ld_slow_r,
                                                                 float* __restrict__ out, int T, int B, int Bp, int M,
                                                                 int shrink, int relu) {
    const long per_t = (long)Bp * 2 * M;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int t = blockIdx.y;
    if (i >= per_t) return;
    const int b = (int)(i / (2 * M)), c = (int)(i % (2 * M));
    float v = 0.f;
    if (b < B) {
        if (c < M) {
            v = enc[((size_t)t * Bp + b) * ld_enc + c];
        } else {  // the bottleneck's output, held; `relu`: it arrives as the output layer's pre-activation
            v = slow[(size_t)(t / shrink) * ld_slow_t + ((size_t)b * M + (c - M)) * ld_slow_r];
            if (relu) v = fmaxf(v, 0.f);
        }
    }
    out[(size_t)t * per_t + i] = v;
}

// o[T][Bp][ld] (2F columns used) -> mask[B][2F][T0], T0 = T - la, frame t0 of the mask = frame t0 + la of o
__global__ __launch_bounds__(256) void fast_mask_out_kernel(const float* __restrict__ o, long ld, float* __restrict__ mask, int T0,
                                                            int la, int Bp, int cols) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, c0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int j = ty; j < 32; j += 8) {
        const int t = t0 + j, c = c0 + tx;
        tile[j][tx] = (t < T0 && c < cols) ? o[((size_t)(t + la) * Bp + b) * ld + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, t = t0 + tx;
        if (c < cols && t < T0) mask[((size_t)b * cols + c) * T0 + t] = tile[tx][j];
    }
}

int fast_low_rate_frames(int T, int shrink) { return 1 + (T - 1 + shrink - 1) / shrink; }

}  // namespace

extern "C" int fsn_fast_low_rate_frames(int T, int shrink) { return T >= 2 && shrink >= 1 ? fast_low_rate_frames(T, shrink) : 0; }

extern "C" size_t fsn_fast_glue_workspace_bytes(int T, int B, int num_mels, int shrink) {
    if (T < 2 || B < 1 || num_mels < 1 || shrink < 1) return 0;
    const size_t Ts = (size_t)fast_low_rate_frames(T, shrink);
    return fsn_round_up_sz(2 * Ts * B * num_mels * sizeof(float), 256) + fsn_round_up_sz((size_t)B * sizeof(float), 256);
}

extern "C" int fsn_fast_spec_rows(const float* mag, int B, int F, int T0, int look_ahead, float* rows, int Bp, int Fp, void* stream) {
    FsnCallScope scope(stream);
    FSN_REQUIRE(mag && rows, "NULL pointer argument");
    FSN_REQUIRE(B >= 1 && F >= 1 && T0 >= 1 && look_ahead >= 0 && Bp >= B && Fp >= F && Bp <= 65535,
                "fast spec rows: need B, F, T0 >= 1, look_ahead >= 0 and padded sizes not below (B, F)");
    const int T = T0 + look_ahead;
    hipLaunchKernelGGL(fast_spec_rows_kernel, dim3((unsigned)((T + 31) / 32), (unsigned)((Fp + 31) / 32), (unsigned)Bp), dim3(256), 0,
                       static_cast<hipStream_t>(stream), mag, rows, B, F, T0, T, Bp, Fp);
    return fsn_check_launch("fast_spec_rows_kernel");
}

extern "C" int fsn_fast_norm_rows(const float* x, int T, int B, int Bp, int C, float* out, void* workspace, size_t workspace_bytes,
                                  void* stream) {
    FsnCallScope scope(stream);
    FSN_REQUIRE(x && out && workspace, "NULL pointer argument");
    FSN_REQUIRE(T >= 1 && B >= 1 && Bp >= B && C >= 1, "fast norm rows: need T, B, C >= 1 and Bp >= B");
    if (workspace_bytes < fsn_round_up_sz((size_t)B * sizeof(float), 256)) {
        fsn_set_error("fast norm rows: workspace too small");
        return FSN_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* den = static_cast<float*>(workspace);
    hipLaunchKernelGGL(fast_mean_kernel, dim3((unsigned)B), dim3(256), 0, s, x, den, T, Bp, C);
    FSN_TRY_LAUNCH("fast_mean_kernel");
    const long n = (long)T * Bp * C;
    hipLaunchKernelGGL(fast_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, den, out, n, B, Bp, C);
    return fsn_check_launch("fast_scale_kernel");
}

extern "C" int fsn_fast_bottleneck_input(const float* mel, const float* enc, long ld_enc, int T, int B, int Bp, int num_mels,
                                         int mel_neighbors, int enc_neighbors, int shrink, float* out, int Np, int Wp,
                                         void* workspace, size_t workspace_bytes, void* stream) {
    FsnCallScope scope(stream);
    FSN_REQUIRE(mel && enc && out && workspace, "NULL pointer argument");
    FSN_REQUIRE(T >= 2 && B >= 1 && Bp >= B && B <= 65535 && num_mels >= 2 && num_mels <= 4096 && ld_enc >= num_mels && shrink >= 1,
                "fast bottleneck input: need T >= 2, 1 <= B <= Bp, 2 <= num_mels <= 4096, ld_enc >= num_mels, shrink >= 1");
    FSN_REQUIRE(mel_neighbors >= 0 && enc_neighbors >= 0 && mel_neighbors < num_mels && enc_neighbors < num_mels,
                "fast bottleneck input: neighbours must be in [0, num_mels)");
    const int W = 2 * mel_neighbors + 1 + 2 * enc_neighbors + 1;
    FSN_REQUIRE(Wp >= W && (long)Np >= (long)B * num_mels, "fast bottleneck input: padded sizes below (B num_mels, unit width %d)", W);
    if (workspace_bytes < fsn_fast_glue_workspace_bytes(T, B, num_mels, shrink)) {
        fsn_set_error("fast bottleneck input: workspace too small");
        return FSN_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    FastUnits a{};
    a.T = T;
    a.Ts = fast_low_rate_frames(T, shrink);
    a.B = B;
    a.Bp = Bp;
    a.M = num_mels;
    a.n_mel = mel_neighbors;
    a.n_enc = enc_neighbors;
    a.shrink = shrink;
    float* ds = static_cast<float*>(workspace);
    float* den = reinterpret_cast<float*>(static_cast<char*>(workspace) + fsn_round_up_sz((size_t)2 * a.Ts * B * num_mels * sizeof(float), 256));
    const size_t lds = (size_t)2 * num_mels * sizeof(float);
    hipLaunchKernelGGL(fast_ds_kernel, dim3((unsigned)B), dim3(256), lds, s, mel, enc, ld_enc, ds, den, a);
    FSN_TRY_LAUNCH("fast_ds_kernel");
    hipLaunchKernelGGL(fast_units_kernel, dim3((unsigned)a.Ts, (unsigned)B), dim3(256), lds, s, ds, den, out, a, Np, Wp);
    return fsn_check_launch("fast_units_kernel");
}

extern "C" int fsn_fast_decoder_input(const float* enc, long ld_enc, const float* slow, long ld_slow_frame, long ld_slow_row, int relu,
                                      int T, int B, int Bp, int num_mels, int shrink, float* out, void* stream) {
    FsnCallScope scope(stream);
    FSN_REQUIRE(enc && slow && out, "NULL pointer argument");
    FSN_REQUIRE(T >= 1 && B >= 1 && Bp >= B && num_mels >= 1 && ld_enc >= num_mels && shrink >= 1 && ld_slow_row >= 1 &&
                    ld_slow_frame >= (long)B * num_mels * ld_slow_row && T <= 65535,
                "fast decoder input: need 1 <= T <= 65535, 1 <= B <= Bp, strides not below the sizes");
    const long per_t = (long)Bp * 2 * num_mels;
    hipLaunchKernelGGL(fast_decoder_input_kernel, dim3((unsigned)((per_t + 255) / 256), (unsigned)T), dim3(256), 0,
                       static_cast<hipStream_t>(stream), enc, ld_enc, slow, ld_slow_frame, ld_slow_row, out, T, B, Bp, num_mels, shrink, relu);
    return fsn_check_launch("fast_decoder_input_kernel");
}

extern "C" int fsn_fast_mask_out(const float* o, long ld, int T, int B, int Bp, int F, int look_ahead, float* mask, void* stream) {
    FsnCallScope scope(stream);
    FSN_REQUIRE(o && mask, "NULL pointer argument");
    FSN_REQUIRE(B >= 1 && Bp >= B && B <= 65535 && F >= 1 && ld >= 2 * (long)F && look_ahead >= 0 && T > look_ahead,
                "fast mask out: need 1 <= B <= Bp, ld >= 2 F, 0 <= look_ahead < T");
    const int T0 = T - look_ahead;
    hipLaunchKernelGGL(fast_mask_out_kernel, dim3((unsigned)((T0 + 31) / 32), (unsigned)((2 * F + 31) / 32), (unsigned)B), dim3(256), 0,
                       static_cast<hipStream_t>(stream), o, ld, mask, T0, look_ahead, Bp, 2 * F);
    return fsn_check_launch("fast_mask_out_kernel");
}
