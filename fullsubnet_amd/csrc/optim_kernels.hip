// Tail of the training step (recipes/dns_interspeech_2020/fullsubnet/trainer.py:62-69):
//   loss = MSELoss()(cIRM, cRM)                       audio_zen/loss.py:4
//   clip_grad_norm_(model.parameters(), 10)           trainer.py:65-67
//   Adam.step()                                       train.py:55-59 (torch.optim.Adam, no weight decay)
// HBM-bound streaming over 5.6 M parameters (22.6 MB): the ~60 small launches of the eager versions
// become two (sum of squares, clip + Adam), both multi-tensor: the kernel argument carries the table
// of tensor pointers, a block looks its 4096-element chunk up in the prefix table.
// Reductions are deterministic: fixed chunk partials (fp64) summed in a fixed order by every block.
#include "fsn_common.h"

namespace {

constexpr int kChunk = 4096;  // elements per block
constexpr int kMaxTensors = FSN_ADAM_MAX_TENSORS;

struct TensorTable {
    float* p[kMaxTensors];
    float* g[kMaxTensors];
    float* m[kMaxTensors];
    float* v[kMaxTensors];
    long numel[kMaxTensors];
    int chunk0[kMaxTensors + 1];  // first chunk of tensor i; chunk0[n] = total chunks
    int n;
};

__device__ __forceinline__ int find_tensor(const TensorTable& tt, int chunk) {
    int i = 0;
    while (i + 1 < tt.n && chunk >= tt.chunk0[i + 1]) ++i;
    return i;
}

__device__ __forceinline__ double block_sum(double x, double* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sh[wave] = x;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
    return t;  // valid in thread 0
}

// skipped (may be NULL): {updates skipped so far, snapshot of that count taken here for the update kernel}
// grad_scale (may be NULL): the loss scale the gradients carry; the norm is that of g / scale
__global__ __launch_bounds__(256) void grad_sumsq_kernel(TensorTable tt, double* __restrict__ partial,
                                                         unsigned* __restrict__ skipped,
                                                         const float* __restrict__ grad_scale) {
    __shared__ double sh[4];
    if (skipped && blockIdx.x == 0 && threadIdx.x == 0) skipped[1] = skipped[0];
    const float inv_scale = grad_scale ? 1.0f / *grad_scale : 1.0f;
    const int ti = find_tensor(tt, blockIdx.x);
    const long base = (long)(blockIdx.x - tt.chunk0[ti]) * kChunk;
    const float* g = tt.g[ti];
    const long n = tt.numel[ti];
    double acc = 0.0;
    for (int i = threadIdx.x; i < kChunk; i += 256) {
        const long k = base + i;
        if (k < n) {
            const double x = g[k] * inv_scale;  // the fp32 value GradScaler.unscale_ would leave in g
            acc += x * x;
        }
    }
    const double t = block_sum(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

struct AdamScalars {
    float one_minus_beta1, beta2, one_minus_beta2, step_size, bc2_sqrt, eps, max_norm;
    double lr, beta1_d, beta2_d;  // for the bias corrections of a step count that skipped updates have shifted
    int step;
};

// every block: total norm from the partials (same order everywhere), clip coefficient as
// torch.nn.utils.clip_grad_norm_ computes it (max_norm / (norm + 1e-6), clamped to 1), then the
// Adam update of its chunk in torch.optim.Adam's single-tensor operation order:
//   m.lerp_(g, 1-b1); v.mul_(b2).addcmul_(g, g, 1-b2); denom = sqrt(v)/sqrt(bc2) + eps; p += -step_size * m/denom
// A non-finite gradient norm (an overflow, or the NaN outputs of a persistent kernel that ran out of time) skips the
// whole update - parameters, both moments and the gradients stay as they are - like GradScaler.step() does in the
// reference (fullsubnet/trainer.py:69), and counts it in skipped[0]; the skipped updates do not advance Adam's step
// count (the host keeps counting calls: the kernel subtracts the snapshot skipped[1]).
__global__ __launch_bounds__(256) void clip_adam_kernel(TensorTable tt, const double* __restrict__ partial,
                                                        int n_partials, AdamScalars a, float* __restrict__ norm_out,
                                                        unsigned* __restrict__ skipped,
                                                        const float* __restrict__ grad_scale,
                                                        const float* __restrict__ found_inf) {
    __shared__ double sh[4];
    __shared__ float coef_sh, step_size_sh, bc2_sqrt_sh;
    __shared__ int ok_sh;
    double acc = 0.0;
    for (int i = threadIdx.x; i < n_partials; i += 256) acc += partial[i];
    const double t = block_sum(acc, sh);
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(t);
        float coef = a.max_norm / (norm + 1e-6f);
        coef_sh = coef < 1.0f ? coef : 1.0f;
        // found_inf: GradScaler's verdict over ALL parameter groups of the optimizer - a group whose own gradients
        // are finite must skip with the others (scaler.step skips the whole optimizer.step, trainer.py:69)
        const bool ok = __builtin_isfinite(norm) && !(found_inf && *found_inf != 0.f);
        ok_sh = ok ? 1 : 0;
        step_size_sh = a.step_size;
        bc2_sqrt_sh = a.bc2_sqrt;
        const unsigned before = skipped ? skipped[1] : 0u;
        if (before != 0u) {
            const int k = a.step - (int)before > 1 ? a.step - (int)before : 1;
            step_size_sh = (float)(a.lr / (1.0 - pow(a.beta1_d, (double)k)));
            bc2_sqrt_sh = (float)sqrt(1.0 - pow(a.beta2_d, (double)k));
        }
        if (blockIdx.x == 0) {
            if (norm_out) *norm_out = norm;
            if (!ok && skipped) skipped[0] = before + 1u;
        }
    }
    __syncthreads();
    if (!ok_sh) return;
    a.step_size = step_size_sh;
    a.bc2_sqrt = bc2_sqrt_sh;
    const float coef = a.max_norm > 0.f ? coef_sh : 1.0f;
    const float inv_scale = grad_scale ? 1.0f / *grad_scale : 1.0f;
    const int ti = find_tensor(tt, blockIdx.x);
    const long base = (long)(blockIdx.x - tt.chunk0[ti]) * kChunk;
    const long n = tt.numel[ti];
    float *p = tt.p[ti], *g = tt.g[ti], *m = tt.m[ti], *v = tt.v[ti];
    for (int i = threadIdx.x; i < kChunk; i += 256) {
        const long k = base + i;
        if (k >= n) break;
        const float gk = (grad_scale ? g[k] * inv_scale : g[k]) * coef;  // unscale (GradScaler.unscale_), then clip
        const float mk = m[k] + a.one_minus_beta1 * (gk - m[k]);
        const float vk = v[k] * a.beta2 + a.one_minus_beta2 * (gk * gk);
        const float denom = sqrtf(vk) / a.bc2_sqrt + a.eps;
        g[k] = gk;
        m[k] = mk;
        v[k] = vk;
        p[k] = p[k] - a.step_size * (mk / denom);
    }
}

// MSE: partial[b] = sum over the block's chunk of (x - y)^2; grad = 2 (x - y) / n
__global__ __launch_bounds__(256) void mse_partial_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                          long n, float inv_n2, double* __restrict__ partial,
                                                          float* __restrict__ grad) {
    __shared__ double sh[4];
    const long base = (long)blockIdx.x * kChunk;
    double acc = 0.0;
    for (int i = threadIdx.x; i < kChunk; i += 256) {
        const long k = base + i;
        if (k < n) {
            const float d = x[k] - y[k];
            acc += (double)d * d;
            if (grad) grad[k] = d * inv_n2;
        }
    }
    const double t = block_sum(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void mse_final_kernel(const double* __restrict__ partial, int n_partials, double inv_n,
                                                        float* __restrict__ loss) {
    __shared__ double sh[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n_partials; i += 256) acc += partial[i];
    const double t = block_sum(acc, sh);
    if (threadIdx.x == 0) *loss = (float)(t * inv_n);
}

int build_table(TensorTable& tt, int n, float* const* p, float* const* g, float* const* m, float* const* v,
                const size_t* numel) {
    tt.n = n;
    int chunks = 0;
    for (int i = 0; i < n; ++i) {
        tt.p[i] = p ? p[i] : nullptr;
        tt.g[i] = g[i];
        tt.m[i] = m ? m[i] : nullptr;
        tt.v[i] = v ? v[i] : nullptr;
        tt.numel[i] = (long)numel[i];
        tt.chunk0[i] = chunks;
        chunks += (int)((numel[i] + kChunk - 1) / kChunk);
    }
    tt.chunk0[n] = chunks;
    return chunks;
}

}  // namespace

extern "C" size_t fsn_clip_adam_workspace_bytes(int n_tensors, const size_t* numel) {
    size_t chunks = 0;
    for (int i = 0; i < n_tensors; ++i) chunks += (numel[i] + kChunk - 1) / kChunk;
    return fsn_round_up_sz((chunks ? chunks : 1) * sizeof(double), 256);
}

extern "C" int fsn_clip_adam_step(int n_tensors, float* const* params, float* const* grads, float* const* exp_avg,
                                  float* const* exp_avg_sq, const size_t* numel, const fsn_adam_cfg* cfg,
                                  float* total_norm_out, const float* grad_scale, const float* found_inf,
                                  unsigned* skipped_steps, void* workspace, size_t workspace_bytes, void* stream) {
    FsnCallScope scope(stream);
    FSN_REQUIRE(params && grads && exp_avg && exp_avg_sq && numel && cfg && workspace, "NULL pointer argument");
    FSN_REQUIRE(n_tensors >= 1 && n_tensors <= kMaxTensors, "clip_adam: 1..%d tensors per call (got %d)", kMaxTensors,
                n_tensors);
    FSN_REQUIRE(cfg->step >= 1 && cfg->lr > 0.f && cfg->beta1 >= 0.f && cfg->beta1 < 1.f && cfg->beta2 >= 0.f &&
                    cfg->beta2 < 1.f && cfg->eps > 0.f,
                "clip_adam: bad hyper-parameters");
    for (int i = 0; i < n_tensors; ++i)
        FSN_REQUIRE(params[i] && grads[i] && exp_avg[i] && exp_avg_sq[i] && numel[i] > 0, "clip_adam: tensor %d is empty", i);
    if (workspace_bytes < fsn_clip_adam_workspace_bytes(n_tensors, numel)) {
        fsn_set_error("clip_adam: workspace too small");
        return FSN_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    TensorTable tt;
    const int chunks = build_table(tt, n_tensors, params, grads, exp_avg, exp_avg_sq, numel);
    double* partial = static_cast<double*>(workspace);
    hipLaunchKernelGGL(grad_sumsq_kernel, dim3(chunks), dim3(256), 0, s, tt, partial, skipped_steps, grad_scale);
    FSN_TRY_LAUNCH("grad_sumsq_kernel");
    // scalars exactly as torch.optim.adam._single_tensor_adam forms them (Python doubles -> fp32 kernel scalars)
    const double b1 = cfg->beta1, b2 = cfg->beta2;
    const double bc1 = 1.0 - pow(b1, cfg->step), bc2 = 1.0 - pow(b2, cfg->step);
    AdamScalars a;
    a.one_minus_beta1 = (float)(1.0 - b1);
    a.beta2 = cfg->beta2;
    a.one_minus_beta2 = (float)(1.0 - b2);
    a.step_size = (float)((double)cfg->lr / bc1);
    a.bc2_sqrt = (float)sqrt(bc2);
    a.eps = cfg->eps;
    a.max_norm = cfg->max_norm;
    a.lr = (double)cfg->lr;
    a.beta1_d = b1;
    a.beta2_d = b2;
    a.step = cfg->step;
    hipLaunchKernelGGL(clip_adam_kernel, dim3(chunks), dim3(256), 0, s, tt, partial, chunks, a, total_norm_out,
                       skipped_steps, grad_scale, found_inf);
    return fsn_check_launch("clip_adam_kernel");
}

extern "C" size_t fsn_mse_loss_workspace_bytes(size_t n) {
    const size_t chunks = (n + kChunk - 1) / kChunk;
    return fsn_round_up_sz((chunks ? chunks : 1) * sizeof(double), 256);
}

extern "C" int fsn_mse_loss(const float* input, const float* target, size_t n, float* loss, float* grad_input,
                            void* workspace, size_t workspace_bytes, void* stream) {
    FsnCallScope scope(stream);
    FSN_REQUIRE(input && target && loss && workspace, "NULL pointer argument");
    FSN_REQUIRE(n >= 1, "mse_loss: empty input");
    if (workspace_bytes < fsn_mse_loss_workspace_bytes(n)) {
        fsn_set_error("mse_loss: workspace too small");
        return FSN_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int chunks = (int)((n + kChunk - 1) / kChunk);
    double* partial = static_cast<double*>(workspace);
    hipLaunchKernelGGL(mse_partial_kernel, dim3(chunks), dim3(256), 0, s, input, target, (long)n, (float)(2.0 / (double)n),
                       partial, grad_input);
    FSN_TRY_LAUNCH("mse_partial_kernel");
    hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(256), 0, s, partial, chunks, 1.0 / (double)n, loss);
    return fsn_check_launch("mse_final_kernel");
}
