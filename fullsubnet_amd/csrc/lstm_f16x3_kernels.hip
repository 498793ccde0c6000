// EXPERIMENTAL, off by default (FSN_F16X3=1, see gemm_f16x3_kernels.hip): the persistent recurrent kernel of the
// LAST sub-band layer (projection precomputed, output layer fused) with the h W_hh^T product on the 16-bit matrix
// cores at fp32 accuracy.  h_t is kept in LDS as two fp16 planes (hi, lo; pre-scaled by 64), W_hh comes pre-split
// and pre-tiled (hi, lo; pre-scaled by 256) from L2, three v_mfma_f32_16x16x32_f16 per product block accumulate in
// fp32 on top of gx * 2^14; everything else - 12 waves, four gate passes, cell state in registers, two barriers per
// step, the fused nn.Linear(H, 2) - is lstm_rec_kernel's.  Left-over tiles stay on the fp32 step kernels.
#include "fsn_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr float kSA = 64.f, kSW = 256.f, kS = kSA * kSW;

template <int H, int RT, int UG>
__global__ __launch_bounds__((H / (16 * UG)) * 64) void lstm_rec_f16x3_kernel(const float* __restrict__ gx,
                                                                              const f16x8* __restrict__ whi,
                                                                              const f16x8* __restrict__ wlo, int Tp,
                                                                              int Npad, const FsnRecFc fc) {
    constexpr int NW = H / (16 * UG);
    constexpr int KC16 = H / 16, KC32 = H / 32, CT = 4 * KC16;
    constexpr int HSH = H + 8;  // halves per LDS row: 16-byte aligned rows, off the 256-byte bank period
    constexpr int ROWS = RT * 16;
    extern __shared__ __attribute__((aligned(16))) _Float16 sh[];
    _Float16* hh = sh;               // [ROWS][HSH] high halves of 64 h
    _Float16* hlo = sh + ROWS * HSH;  // low halves
    float* wl = reinterpret_cast<float*>(hlo + ROWS * HSH);  // [2][H] output-layer weights

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const long n0 = (long)blockIdx.x * ROWS;
    for (int i = threadIdx.x; i < 2 * H; i += NW * 64) {
        const int c = i / H, k = i % H;
        wl[i] = fc.w_p[(((k >> 4) * 64) + ((k & 15) >> 2) * 16 + c) * 4 + (k & 3)];
    }
    for (int i = threadIdx.x; i < 2 * ROWS * HSH; i += NW * 64) sh[i] = (_Float16)0.f;
    f32x4 cst[RT][UG], tmp[RT][UG];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int u = 0; u < UG; ++u) cst[rt][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    for (int t = 0; t < Tp; ++t) {
        const long gx_rt0 = ((long)t * Npad + n0) >> 4;
#pragma unroll 1
        for (int pass = 0; pass < 4; ++pass) {
            const int g = pass == 0 ? 1 : (pass == 1 ? 0 : pass);  // f, i, g, o
            f32x4 acc[RT][UG];
            long bo[UG];
#pragma unroll
            for (int u = 0; u < UG; ++u) {
                const int ug = wave * UG + u;
                bo[u] = (long)(g * KC16 + ug) * KC32 * 64 + lane;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const f32x4 x =
                        *reinterpret_cast<const f32x4*>(gx + (((gx_rt0 + rt) * CT + g * KC16 + ug) * 64 + lane) * 4);
                    acc[rt][u] = f32x4{x[0] * kS, x[1] * kS, x[2] * kS, x[3] * kS};
                }
            }
            if (t > 0) {  // h_{-1} = 0
                f16x8 bhn[UG], bln[UG];
#pragma unroll
                for (int u = 0; u < UG; ++u) {
                    bhn[u] = whi[bo[u]];
                    bln[u] = wlo[bo[u]];
                }
#pragma unroll 1
                for (int kc = 0; kc < KC32; ++kc) {
                    f16x8 bh[UG], bl[UG];
#pragma unroll
                    for (int u = 0; u < UG; ++u) {
                        bh[u] = bhn[u];
                        bl[u] = bln[u];
                    }
                    const int kn = kc + 1 < KC32 ? kc + 1 : kc;  // clamped: branch-free
#pragma unroll
                    for (int u = 0; u < UG; ++u) {
                        bhn[u] = whi[bo[u] + (long)kn * 64];
                        bln[u] = wlo[bo[u] + (long)kn * 64];
                    }
                    const _Float16* ap = hh + lr * HSH + kc * 32 + 8 * lq;
                    const _Float16* lp = hlo + lr * HSH + kc * 32 + 8 * lq;
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const f16x8 ah = *reinterpret_cast<const f16x8*>(ap + rt * 16 * HSH);
                        const f16x8 al = *reinterpret_cast<const f16x8*>(lp + rt * 16 * HSH);
#pragma unroll
                        for (int u = 0; u < UG; ++u) {
                            acc[rt][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[u], acc[rt][u], 0, 0, 0);
                            acc[rt][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[u], acc[rt][u], 0, 0, 0);
                            acc[rt][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[u], acc[rt][u], 0, 0, 0);
                        }
                    }
                }
            }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int u = 0; u < UG; ++u)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float x = acc[rt][u][i] * (1.0f / kS);
                        if (pass == 0) cst[rt][u][i] = sigmoid_fast(x) * cst[rt][u][i];
                        else if (pass == 1) tmp[rt][u][i] = sigmoid_fast(x);
                        else if (pass == 2) cst[rt][u][i] = cst[rt][u][i] + tmp[rt][u][i] * tanh_fast(x);
                        else tmp[rt][u][i] = sigmoid_fast(x) * tanh_fast(cst[rt][u][i]);
                    }
        }
        __syncthreads();  // every wave has finished reading h_{t-1}
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int u = 0; u < UG; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = tmp[rt][u][i] * kSA;
                    const _Float16 hi = (_Float16)v;
                    const int o = (rt * 16 + 4 * lq + i) * HSH + (wave * UG + u) * 16 + lr;
                    hh[o] = hi;
                    hlo[o] = (_Float16)(v - (float)hi);
                }
        __syncthreads();  // h_t complete in LDS
        {   // fused output layer (see lstm_rec_kernel), h = (hi + lo) / 64
            const int tid = threadIdx.x;
            if (tid < ROWS * 8) {
                const int part = tid & 3, c = (tid >> 2) & 1, row = tid >> 3;
                const _Float16* hp = hh + row * HSH + part * (H / 4);
                const _Float16* lp = hlo + row * HSH + part * (H / 4);
                const float* wp = wl + c * H + part * (H / 4);
                float a0 = 0.f;
#pragma unroll 2
                for (int k = 0; k < H / 4; k += 8) {
                    const f16x8 x = *reinterpret_cast<const f16x8*>(hp + k), y = *reinterpret_cast<const f16x8*>(lp + k);
                    const f32x4 w0 = *reinterpret_cast<const f32x4*>(wp + k), w1 = *reinterpret_cast<const f32x4*>(wp + k + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) a0 = fmaf((float)x[j] + (float)y[j], w0[j], a0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) a0 = fmaf((float)x[4 + j] + (float)y[4 + j], w1[j], a0);
                }
                float v = a0 * (1.0f / kSA);
                v += __shfl_xor(v, 1, 64);
                v += __shfl_xor(v, 2, 64);
                const long n = n0 + row;
                if (part == 0 && t >= fc.la && n < fc.N) {
                    const int b = (int)(n / fc.F), f = (int)(n % fc.F);
                    (c ? fc.crm_i : fc.crm_r)[((long)b * fc.T + (t - fc.la)) * fc.FP + f] = v + fc.bias[c];
                }
            }
        }
    }
}

template <int RT>
int launch(const float* gx, const void* packed, int Tp, int Npad, int main_wgs, const FsnRecFc& fc, hipStream_t s) {
    constexpr int H = 384, UG = 2, NW = H / (16 * UG);
    const size_t lds = (size_t)2 * RT * 16 * (H + 8) * sizeof(_Float16) + (size_t)2 * H * sizeof(float);
    auto kern = lstm_rec_f16x3_kernel<H, RT, UG>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess) {
        fsn_set_error("lstm_rec_f16x3: cannot reserve %zu bytes of LDS", lds);
        return FSN_ERR_LAUNCH;
    }
    const f16x8* whi = static_cast<const f16x8*>(packed);
    const f16x8* wlo = whi + (size_t)4 * H * H / 8;
    hipLaunchKernelGGL(kern, dim3((unsigned)main_wgs), dim3(NW * 64), lds, s, gx, whi, wlo, Tp, Npad, fc);
    return fsn_check_launch("lstm_rec_f16x3_kernel");
}

}  // namespace

// packed: W_hh [4H][H] split by fsn_launch_pack_f16x3; fc: required (this variant never stores the hidden sequence)
int fsn_launch_lstm_rec_f16x3(const float* gx, const void* packed, int Tp, int Npad, int H, int RT, int main_wgs,
                              const FsnRecFc* fc, hipStream_t s) {
    if (H != 384 || !fc || !fc->w_p || RT < 2 || RT > 5) {
        fsn_set_error("lstm_rec_f16x3: built for H = 384, 2..5 row tiles per workgroup and a fused output layer");
        return FSN_ERR_ARG;
    }
    switch (RT) {
        case 2: return launch<2>(gx, packed, Tp, Npad, main_wgs, *fc, s);
        case 3: return launch<3>(gx, packed, Tp, Npad, main_wgs, *fc, s);
        case 4: return launch<4>(gx, packed, Tp, Npad, main_wgs, *fc, s);
        default: return launch<5>(gx, packed, Tp, Npad, main_wgs, *fc, s);
    }
}
