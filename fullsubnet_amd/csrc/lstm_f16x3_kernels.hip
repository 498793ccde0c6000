// EXPERIMENTAL, off by default (FSN_F16X3=1, see gemm_f16x3_kernels.hip): the persistent recurrent kernels of the
// sub-band model - the LAST layer (projection precomputed, output layer fused) and, with XIN, the FIRST layer (K = 32
// input projection built in-kernel from a staged input tile, hidden sequence stored) - with the h W_hh^T product on
// the 16-bit matrix cores at fp32 accuracy.  h_t is kept in LDS as two fp16 planes (hi, lo; pre-scaled by 64), W_hh comes pre-split
// and pre-tiled (hi, lo; pre-scaled by 256) from L2, three v_mfma_f32_16x16x32_f16 per product block accumulate in
// fp32 on top of gx * 2^14; everything else - 12 waves, four gate passes, cell state in registers, two barriers per
// step, the fused nn.Linear(H, 2) - is lstm_rec_kernel's.  Left-over tiles stay on the fp32 step kernels.
#include "fsn_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#ifndef FSN_F16X3_UG
#define FSN_F16X3_UG 2  // hidden-unit groups of 16 per wave: 2 -> 12 waves, 3 -> 8 waves
#endif
#ifndef FSN_PROBE_ABLATE  // tools/probe_rec_f16x3.hip only: 1 no W refills, 2 no LDS operand reads, 3 no cell math
#define FSN_PROBE_ABLATE 0
#endif

namespace {

constexpr float kSA = 64.f, kSW = 256.f, kS = kSA * kSW;
// layer-0 input tile: the normalised magnitudes have no bound like |h| < 1 (one loud bin of an otherwise quiet
// utterance is hundreds of times the mean), so they get the small pre-scale and W_ih the large one; same product
constexpr float kSX = 4.f, kSWX = kS / kSX;
constexpr int kXSH = 40;  // halves per LDS row of the 32-wide input tile

template <int H, int RT, int UG, bool XIN>
__global__ __launch_bounds__((H / (16 * UG)) * 64) void lstm_rec_f16x3_kernel(const float* __restrict__ gx,
                                                                              const FsnSbInput xin,
                                                                              const f16x8* __restrict__ wxh,
                                                                              const f16x8* __restrict__ wxl,
                                                                              const f16x8* __restrict__ whi,
                                                                              const f16x8* __restrict__ wlo,
                                                                              float* __restrict__ hseq, int Tp,
                                                                              int Npad, const FsnRecFc fc) {
    constexpr int NW = H / (16 * UG);
    constexpr int KC16 = H / 16, KC32 = H / 32, CT = 4 * KC16;
    constexpr int HSH = H + 8;  // halves per LDS row: 16-byte aligned rows, off the 256-byte bank period
    constexpr int ROWS = RT * 16;
    extern __shared__ __attribute__((aligned(16))) _Float16 sh[];
    _Float16* hh = sh;               // [ROWS][HSH] high halves of 64 h
    _Float16* hlo = sh + ROWS * HSH;  // low halves
    float* wl = reinterpret_cast<float*>(hlo + ROWS * HSH);  // !XIN: [2][H] output-layer weights
    _Float16* xh = hlo + ROWS * HSH;                          // XIN: [2 buffers][ROWS][kXSH] high halves of 4 x
    _Float16* xlo = xh + 2 * ROWS * kXSH;                     //      low halves
    auto stage = [&](int t) {
        _Float16* dh = xh + (t & 1) * ROWS * kXSH;
        _Float16* dl = xlo + (t & 1) * ROWS * kXSH;
        for (int i = threadIdx.x; i < ROWS * 32; i += NW * 64) {
            const int row = i >> 5, c = i & 31;
            // saturate instead of overflowing to inf: a bin 16 000 times the utterance mean (a lone tone burst in
            // digital silence) drives every gate it touches far into saturation either way
            const float v = fminf(fsn_sb_input_value(xin, (long)blockIdx.x * ROWS + row, c, t) * kSX, 65504.f);
            const _Float16 hi = (_Float16)v;
            dh[row * kXSH + c] = hi;
            dl[row * kXSH + c] = (_Float16)(v - (float)hi);
        }
    };

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const long n0 = (long)blockIdx.x * ROWS;
    for (int i = threadIdx.x; i < 2 * ROWS * HSH; i += NW * 64) sh[i] = (_Float16)0.f;
    if (XIN) {
        stage(0);
    } else {
        for (int i = threadIdx.x; i < 2 * H; i += NW * 64) {
            const int c = i / H, k = i % H;
            wl[i] = fc.w_p[(((k >> 4) * 64) + ((k & 15) >> 2) * 16 + c) * 4 + (k & 3)];
        }
    }
    f32x4 cst[RT][UG], tmp[RT][UG];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int u = 0; u < UG; ++u) cst[rt][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    for (int t = 0; t < Tp; ++t) {
        const long gx_rt0 = ((long)t * Npad + n0) >> 4;
        // frame t+1 goes into the other buffer: last read in step t-1, first read after this step's two barriers
        if (XIN && t + 1 < Tp) stage(t + 1);
#pragma unroll 1
        for (int pass = 0; pass < 4; ++pass) {
            const int g = pass == 0 ? 1 : (pass == 1 ? 0 : pass);  // f, i, g, o
            f32x4 acc[RT][UG];
            long bo[UG];
#pragma unroll
            for (int u = 0; u < UG; ++u) {
                const int ug = wave * UG + u;
                bo[u] = (long)(g * KC16 + ug) * KC32 * 64 + lane;
                if (!XIN) {
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const f32x4* xp = reinterpret_cast<const f32x4*>(
                            gx + (((gx_rt0 + rt) * CT + g * KC16 + ug) * 64 + lane) * 4);
                        const f32x4 x = __builtin_nontemporal_load(xp);  // streamed once: keep W_hh in L2 instead
                        acc[rt][u] = f32x4{x[0] * kS, x[1] * kS, x[2] * kS, x[3] * kS};
                    }
                } else {
                    const float bias = xin.bias[(g * KC16 + ug) * 16 + lr] * kS;
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[rt][u] = f32x4{bias, bias, bias, bias};
                }
            }
            if (XIN) {  // W_ih x_t: one 32-wide chunk, the same three-term product
                f16x8 bxh[UG], bxl[UG];
#pragma unroll
                for (int u = 0; u < UG; ++u) {
                    bxh[u] = wxh[(long)(g * KC16 + wave * UG + u) * 64 + lane];
                    bxl[u] = wxl[(long)(g * KC16 + wave * UG + u) * 64 + lane];
                }
                const _Float16* xp = xh + (t & 1) * ROWS * kXSH + lr * kXSH + 8 * lq;
                const _Float16* xq = xlo + (t & 1) * ROWS * kXSH + lr * kXSH + 8 * lq;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const f16x8 ah = *reinterpret_cast<const f16x8*>(xp + rt * 16 * kXSH);
                    const f16x8 al = *reinterpret_cast<const f16x8*>(xq + rt * 16 * kXSH);
#pragma unroll
                    for (int u = 0; u < UG; ++u) {
                        acc[rt][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bxh[u], acc[rt][u], 0, 0, 0);
                        acc[rt][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bxl[u], acc[rt][u], 0, 0, 0);
                        acc[rt][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bxh[u], acc[rt][u], 0, 0, 0);
                    }
                }
            }
            if (t > 0) {  // h_{-1} = 0
                int kcur = 0;
                f16x8 bhn[UG], bln[UG];
#pragma unroll
                for (int u = 0; u < UG; ++u) {
                    bhn[u] = whi[bo[u] + (long)kcur * 64];
                    bln[u] = wlo[bo[u] + (long)kcur * 64];
                }
#pragma unroll 1
                for (int it = 0; it < KC32; ++it) {
                    f16x8 bh[UG], bl[UG];
#pragma unroll
                    for (int u = 0; u < UG; ++u) {
                        bh[u] = bhn[u];
                        bl[u] = bln[u];
                    }
                    const int kc = kcur;
                    kcur = kcur + 1 < KC32 ? kcur + 1 : 0;  // the last refill re-reads the first chunk: branch-free
                    // pinned: without the fences the compiler sinks these loads to the top of the NEXT iteration,
                    // right in front of the MFMAs that wait for them (an L2 round trip exposed per chunk)
                    __builtin_amdgcn_sched_barrier(0);
                    if (FSN_PROBE_ABLATE != 1) {
#pragma unroll
                        for (int u = 0; u < UG; ++u) {
                            bhn[u] = whi[bo[u] + (long)kcur * 64];
                            bln[u] = wlo[bo[u] + (long)kcur * 64];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const _Float16* ap = hh + lr * HSH + (FSN_PROBE_ABLATE == 2 ? 0 : kc * 32) + 8 * lq;
                    const _Float16* lp = hlo + lr * HSH + (FSN_PROBE_ABLATE == 2 ? 0 : kc * 32) + 8 * lq;
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
            // one branch on the pass around the whole update, not one per element
#define FSN_CELL(STMT)                                                \
    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                 \
    _Pragma("unroll") for (int u = 0; u < UG; ++u)                    \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                   \
        const float x = acc[rt][u][i] * (1.0f / kS);                  \
        STMT;                                                         \
    }
            if (FSN_PROBE_ABLATE == 3) { FSN_CELL(tmp[rt][u][i] = x + cst[rt][u][i]) }
            else if (pass == 0) { FSN_CELL(cst[rt][u][i] = sigmoid_fast(x) * cst[rt][u][i]) }
            else if (pass == 1) { FSN_CELL(tmp[rt][u][i] = sigmoid_fast(x)) }
            else if (pass == 2) { FSN_CELL(cst[rt][u][i] = cst[rt][u][i] + tmp[rt][u][i] * tanh_fast(x)) }
            else { FSN_CELL(tmp[rt][u][i] = sigmoid_fast(x) * tanh_fast(cst[rt][u][i])) }
#undef FSN_CELL
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
        if (XIN) {  // stream h_t = (hi + lo) / 64 out as whole fp32 rows: hseq[t][n0 + row][0..H)
            float* dst = hseq + ((long)t * Npad + n0) * H;
            for (int i = threadIdx.x; i < ROWS * (H / 8); i += NW * 64) {
                const int row = i / (H / 8), c8 = i % (H / 8);
                const f16x8 x = *reinterpret_cast<const f16x8*>(hh + row * HSH + c8 * 8);
                const f16x8 y = *reinterpret_cast<const f16x8*>(hlo + row * HSH + c8 * 8);
                f32x4 o0, o1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o0[j] = ((float)x[j] + (float)y[j]) * (1.0f / kSA);
                    o1[j] = ((float)x[4 + j] + (float)y[4 + j]) * (1.0f / kSA);
                }
                *reinterpret_cast<f32x4*>(dst + (long)row * H + c8 * 8) = o0;
                *reinterpret_cast<f32x4*>(dst + (long)row * H + c8 * 8 + 4) = o1;
            }
        } else {  // fused output layer (see lstm_rec_kernel), h = (hi + lo) / 64
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
                    const long ng = n + fc.row0;
                    const int b = (int)(ng / fc.F), f = (int)(ng % fc.F);
                    (c ? fc.crm_i : fc.crm_r)[((long)b * fc.T + (t - fc.la)) * fc.FP + f] = v + fc.bias[c];
                }
            }
        }
    }
}

template <int RT, bool XIN>
int launch(const float* gx, const FsnSbInput& xin, const void* wih_packed, const void* packed, float* hseq, int Tp,
           int Npad, int main_wgs, const FsnRecFc& fc, hipStream_t s) {
    constexpr int H = 384, UG = FSN_F16X3_UG, NW = H / (16 * UG);
    const size_t lds = (size_t)2 * RT * 16 * (H + 8) * sizeof(_Float16) +
                       (XIN ? (size_t)4 * RT * 16 * kXSH * sizeof(_Float16) : (size_t)2 * H * sizeof(float));
    auto kern = lstm_rec_f16x3_kernel<H, RT, UG, XIN>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess) {
        fsn_set_error("lstm_rec_f16x3: cannot reserve %zu bytes of LDS", lds);
        return FSN_ERR_LAUNCH;
    }
    const f16x8* whi = static_cast<const f16x8*>(packed);
    const f16x8* wlo = whi + (size_t)4 * H * H / 8;
    const f16x8* wxh = static_cast<const f16x8*>(wih_packed);
    const f16x8* wxl = wxh ? wxh + (size_t)4 * H * 32 / 8 : nullptr;
    hipLaunchKernelGGL(kern, dim3((unsigned)main_wgs), dim3(NW * 64), lds, s, gx, xin, wxh, wxl, whi, wlo, hseq, Tp, Npad,
                       fc);
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
    const FsnSbInput none{};
    switch (RT) {
        case 2: return launch<2, false>(gx, none, nullptr, packed, nullptr, Tp, Npad, main_wgs, *fc, s);
        case 3: return launch<3, false>(gx, none, nullptr, packed, nullptr, Tp, Npad, main_wgs, *fc, s);
        case 4: return launch<4, false>(gx, none, nullptr, packed, nullptr, Tp, Npad, main_wgs, *fc, s);
        default: return launch<5, false>(gx, none, nullptr, packed, nullptr, Tp, Npad, main_wgs, *fc, s);
    }
}

// first sub-band layer: wih_packed = W_ih [4H][32] split with scale 4096 (kSWX), whh_packed = W_hh [4H][H] (scale 256)
int fsn_launch_lstm_rec_xin_f16x3(const FsnSbInput* xin, const void* wih_packed, const void* whh_packed, float* hseq,
                                  int Tp, int Npad, int H, int RT, int main_wgs, hipStream_t s) {
    if (H != 384 || !xin || xin->kin_chunks != 2 || RT < 2 || RT > 5) {
        fsn_set_error("lstm_rec_xin_f16x3: built for H = 384, a 32-wide layer input and 2..5 row tiles per workgroup");
        return FSN_ERR_ARG;
    }
    const FsnRecFc none{};
    switch (RT) {
        case 2: return launch<2, true>(nullptr, *xin, wih_packed, whh_packed, hseq, Tp, Npad, main_wgs, none, s);
        case 3: return launch<3, true>(nullptr, *xin, wih_packed, whh_packed, hseq, Tp, Npad, main_wgs, none, s);
        case 4: return launch<4, true>(nullptr, *xin, wih_packed, whh_packed, hseq, Tp, Npad, main_wgs, none, s);
        default: return launch<5, true>(nullptr, *xin, wih_packed, whh_packed, hseq, Tp, Npad, main_wgs, none, s);
    }
}

float fsn_f16x3_wih0_scale() { return kSWX; }
