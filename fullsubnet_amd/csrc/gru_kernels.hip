// nn.GRU branch of SequenceModel (audio_zen/model/module/sequence_model.py:59-66), one layer,
// unidirectional, h0 = 0.  PyTorch's cell (gate rows r, z, n of weight_ih / weight_hh):
//   r = sigmoid(W_ir x + b_ir + W_hr h + b_hr)        z = sigmoid(W_iz x + b_iz + W_hz h + b_hz)
//   n = tanh(W_in x + b_in + r * (W_hn h + b_hn))     h' = n + z * (h - n)
// Same execution shape as the LSTM step kernels (lstm_kernels.hip / lstm_train_kernels.hip): the input
// projection of all steps is one GEMM (fragment-ordered gx, bias = b_ih + [b_hr, b_hz, 0]); each step is
// one launch whose workgroups own RTS 16-row tiles x one 16-unit group x the three gates, 4 waves =
// 4-way split-K reduced through LDS in a fixed order.  Training keeps r, z, n and hn = W_hn h + b_hn.
#include <stdlib.h>

#include "fsn_common.h"

namespace {

template <int RTS>
__global__ __launch_bounds__(256) void gru_step_kernel(const float* __restrict__ gx, const float* __restrict__ whh_p,
                                                       const float* __restrict__ b_hn,
                                                       const float* __restrict__ h_prev, float* __restrict__ h_out,
                                                       float* __restrict__ save, long gx_rt0, int row_tiles, int H,
                                                       int first) {
    __shared__ f32x4 red[4][RTS][3][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int ug = blockIdx.x, rtile0 = blockIdx.y * RTS;
    const int KC = H >> 4, CT = 3 * KC;
    if (!first) {
        f32x4 acc[RTS][3];
#pragma unroll
        for (int rt = 0; rt < RTS; ++rt)
#pragma unroll
            for (int g = 0; g < 3; ++g) acc[rt][g] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int kc0 = wave * (KC >> 2), kc1 = kc0 + (KC >> 2);
        const float* ap[RTS];
#pragma unroll
        for (int rt = 0; rt < RTS; ++rt) {
            int rtile = rtile0 + rt;
            rtile = rtile < row_tiles ? rtile : row_tiles - 1;
            ap[rt] = h_prev + ((long)rtile * 16 + lr) * H + 4 * lq;
        }
#pragma unroll 2
        for (int kc = kc0; kc < kc1; ++kc) {
            f32x4 a[RTS], b[3];
#pragma unroll
            for (int rt = 0; rt < RTS; ++rt) a[rt] = *reinterpret_cast<const f32x4*>(ap[rt] + kc * 16);
#pragma unroll
            for (int g = 0; g < 3; ++g)
                b[g] = *reinterpret_cast<const f32x4*>(whh_p + (((long)(g * KC + ug) * KC + kc) * 64 + lane) * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int rt = 0; rt < RTS; ++rt)
#pragma unroll
                    for (int g = 0; g < 3; ++g) acc[rt][g] = mfma16(a[rt][j], b[g][j], acc[rt][g]);
        }
#pragma unroll
        for (int rt = 0; rt < RTS; ++rt)
#pragma unroll
            for (int g = 0; g < 3; ++g) red[wave][rt][g][lane] = acc[rt][g];
        __syncthreads();
    }
    const int rt = wave, rtile = rtile0 + rt;
    if (rt >= RTS || rtile >= row_tiles) return;
    f32x4 hh[3], xg[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (!first) {
            v = red[0][rt][g][lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const f32x4 r = red[w][rt][g][lane];
                v = f32x4{v[0] + r[0], v[1] + r[1], v[2] + r[2], v[3] + r[3]};
            }
        }
        hh[g] = v;
        xg[g] = *reinterpret_cast<const f32x4*>(gx + (((gx_rt0 + rtile) * CT + g * KC + ug) * 64 + lane) * 4);
    }
    const int u = ug * 16 + lr;
    const float bn = b_hn[u];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long row = (long)rtile * 16 + 4 * lq + i;
        const long idx = row * H + u;
        const float hp = first ? 0.f : h_prev[idx];
        const float r = sigmoid_f(xg[0][i] + hh[0][i]);
        const float z = sigmoid_f(xg[1][i] + hh[1][i]);
        const float hn = hh[2][i] + bn;
        const float n = tanhf(xg[2][i] + r * hn);
        h_out[idx] = n + z * (hp - n);
        if (save) {  // training: [row][4H] = r | z | n | hn
            float* sp = save + row * 4 * H + u;
            sp[0] = r;
            sp[H] = z;
            sp[2 * H] = n;
            sp[3 * H] = hn;
        }
    }
}

// The same step for the left-over row tiles BESIDE a resident workgroup of the persistent many-row kernels (lstm_kernels.hip,
// FSN_REC_GRU: 3 x 152 of a SIMD's 512 registers and up to 148 KB of a CU's 160 KB of LDS are taken): one row tile per
// workgroup, at most 48 registers, 6 KB of LDS - the split-K partial sums travel in two rounds (gates r and z, then the n gate's
// recurrent part) through one small buffer.  Same operations in the same order as gru_step_kernel<1>: bit-identical results.
// (gru_step_kernel<1> itself takes 64 registers and 12 KB: its launches waited for the persistent launch to END - measured, a
// GRU FullSubNet at batch 64: 190 launches per layer behind each of the two persistent launches, +2.9 ms of 65.)
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(24))) void gru_step1_kernel(
    const float* __restrict__ gx, const float* __restrict__ whh_p, const float* __restrict__ b_hn,
    const float* __restrict__ h_prev, float* __restrict__ h_out, long gx_rt0, int row_tiles, int H, int first) {
    __shared__ f32x4 red[3][2][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int ug = blockIdx.x, rtile = blockIdx.y;
    const int KC = H >> 4, CT = 3 * KC;
    f32x4 acc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!first) {
        const int kc0 = wave * (KC >> 2), kc1 = kc0 + (KC >> 2);
        const float* ap = h_prev + ((long)rtile * 16 + lr) * H + 4 * lq;
#pragma unroll 1
        for (int kc = kc0; kc < kc1; ++kc) {
            f32x4 b[3];
            const f32x4 a = *reinterpret_cast<const f32x4*>(ap + kc * 16);
#pragma unroll
            for (int g = 0; g < 3; ++g)
                b[g] = *reinterpret_cast<const f32x4*>(whh_p + (((long)(g * KC + ug) * KC + kc) * 64 + lane) * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[g] = mfma16(a[j], b[g][j], acc[g]);
        }
        // ((wave 0 + wave 1) + wave 2) + wave 3, gate by gate: gru_step_kernel's order
        if (wave > 0) {
            red[wave - 1][0][lane] = acc[0];
            red[wave - 1][1][lane] = acc[1];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int w = 0; w < 3; ++w) {
                    const f32x4 r = red[w][g][lane];
                    acc[g] = f32x4{acc[g][0] + r[0], acc[g][1] + r[1], acc[g][2] + r[2], acc[g][3] + r[3]};
                }
        }
        __syncthreads();
        if (wave > 0) red[wave - 1][0][lane] = acc[2];
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                const f32x4 r = red[w][0][lane];
                acc[2] = f32x4{acc[2][0] + r[0], acc[2][1] + r[1], acc[2][2] + r[2], acc[2][3] + r[3]};
            }
        }
    }
    if (wave != 0 || rtile >= row_tiles) return;
    f32x4 xg[3];
#pragma unroll
    for (int g = 0; g < 3; ++g)
        xg[g] = *reinterpret_cast<const f32x4*>(gx + (((gx_rt0 + rtile) * CT + g * KC + ug) * 64 + lane) * 4);
    const int u = ug * 16 + lr;
    const float bn = b_hn[u];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long idx = ((long)rtile * 16 + 4 * lq + i) * H + u;
        const float hp = first ? 0.f : h_prev[idx];
        const float r = sigmoid_f(xg[0][i] + acc[0][i]);
        const float z = sigmoid_f(xg[1][i] + acc[1][i]);
        const float hn = acc[2][i] + bn;
        const float n = tanhf(xg[2][i] + r * hn);
        h_out[idx] = n + z * (hp - n);
    }
}

// One BPTT step: rec = [dgx_{t+1}[:, :2H] | dghn_{t+1}] W_hh (K = 3H) for RTS x CTS blocks, then
//   dh = dh_out_t + rec + carry;  dn = dh (1 - z);  dz = dh (h_{t-1} - n);  carry' = dh z
//   dn_pre = dn (1 - n^2);  dr_pre = dn_pre hn r (1 - r);  dz_pre = dz z (1 - z)
//   dgx_t = [dr_pre, dz_pre, dn_pre],  dghn_t = dn_pre r
template <int RTS, int CTS>
__global__ __launch_bounds__(256) void gru_bptt_step_kernel(
    const float* __restrict__ dh_out, const float* __restrict__ dgx_next, const float* __restrict__ dghn_next,
    const float* __restrict__ whhT_p, float* __restrict__ carry, const float* __restrict__ save,
    const float* __restrict__ h_prev, float* __restrict__ dgx, float* __restrict__ dghn, int row_tiles, int H, int last,
    int first) {
    __shared__ f32x4 red[4][RTS][CTS][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int ug0 = blockIdx.x * CTS, rtile0 = blockIdx.y * RTS;
    const int G = 3 * H, KC = G >> 4, KX = (2 * H) >> 4;
    if (!last) {
        f32x4 acc[RTS][CTS];
#pragma unroll
        for (int rt = 0; rt < RTS; ++rt)
#pragma unroll
            for (int ct = 0; ct < CTS; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int kc0 = wave * (KC >> 2), kc1 = kc0 + (KC >> 2);
        long arow[RTS];
        const float* bp[CTS];
#pragma unroll
        for (int rt = 0; rt < RTS; ++rt) {
            int rtile = rtile0 + rt;
            rtile = rtile < row_tiles ? rtile : row_tiles - 1;
            arow[rt] = (long)rtile * 16 + lr;
        }
#pragma unroll
        for (int ct = 0; ct < CTS; ++ct) bp[ct] = whhT_p + ((long)(ug0 + ct) * KC * 64 + lane) * 4;
#pragma unroll 2
        for (int kc = kc0; kc < kc1; ++kc) {
            f32x4 a[RTS], b[CTS];
#pragma unroll
            for (int rt = 0; rt < RTS; ++rt) {
                const float* src = kc < KX ? dgx_next + arow[rt] * G + kc * 16 : dghn_next + arow[rt] * H + (kc - KX) * 16;
                a[rt] = *reinterpret_cast<const f32x4*>(src + 4 * lq);
            }
#pragma unroll
            for (int ct = 0; ct < CTS; ++ct) b[ct] = *reinterpret_cast<const f32x4*>(bp[ct] + (long)kc * 256);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int rt = 0; rt < RTS; ++rt)
#pragma unroll
                    for (int ct = 0; ct < CTS; ++ct) acc[rt][ct] = mfma16(a[rt][j], b[ct][j], acc[rt][ct]);
        }
#pragma unroll
        for (int rt = 0; rt < RTS; ++rt)
#pragma unroll
            for (int ct = 0; ct < CTS; ++ct) red[wave][rt][ct][lane] = acc[rt][ct];
        __syncthreads();
    }
    for (int tt = wave; tt < RTS * CTS; tt += 4) {
        const int rt = tt / CTS, ct = tt % CTS;
        const int rtile = rtile0 + rt;
        if (rtile >= row_tiles) continue;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (!last) {
            v = red[0][rt][ct][lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const f32x4 r = red[w][rt][ct][lane];
                v = f32x4{v[0] + r[0], v[1] + r[1], v[2] + r[2], v[3] + r[3]};
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long row = (long)rtile * 16 + 4 * lq + i;
            const int u = (ug0 + ct) * 16 + lr;
            const long idx = row * H + u;
            const float* sp = save + row * 4 * H + u;
            const float r = sp[0], z = sp[H], n = sp[2 * H], hn = sp[3 * H];
            const float dh = dh_out[idx] + v[i] + (last ? 0.f : carry[idx]);
            const float hp = first ? 0.f : h_prev[idx];
            const float dn_pre = dh * (1.f - z) * (1.f - n * n);
            const float dz_pre = dh * (hp - n) * z * (1.f - z);
            const float dr_pre = dn_pre * hn * r * (1.f - r);
            float* dg = dgx + row * 3 * H + u;
            dg[0] = dr_pre;
            dg[H] = dz_pre;
            dg[2 * H] = dn_pre;
            dghn[idx] = dn_pre * r;
            carry[idx] = dh * z;
        }
    }
}

}  // namespace

int fsn_launch_gru_step(const float* gx, const float* whh_p, const float* b_hn, const float* h_prev, float* h_out,
                        float* save, long gx_rt0, int row_tiles, int H, int first, hipStream_t s, int beside_persistent) {
    if (H % 64 != 0) {
        fsn_set_error("gru_step: hidden size %d must be a multiple of 64", H);
        return FSN_ERR_ARG;
    }
    if (beside_persistent && !save) {  // must fit next to a resident persistent workgroup, see gru_step1_kernel
        hipLaunchKernelGGL(gru_step1_kernel, dim3(H / 16, row_tiles), dim3(256), 0, s, gx, whh_p, b_hn, h_prev, h_out, gx_rt0,
                           row_tiles, H, first);
        return fsn_check_launch("gru_step1_kernel");
    }
    if (row_tiles >= 16)
        hipLaunchKernelGGL(gru_step_kernel<2>, dim3(H / 16, (row_tiles + 1) / 2), dim3(256), 0, s, gx, whh_p, b_hn, h_prev,
                           h_out, save, gx_rt0, row_tiles, H, first);
    else
        hipLaunchKernelGGL(gru_step_kernel<1>, dim3(H / 16, row_tiles), dim3(256), 0, s, gx, whh_p, b_hn, h_prev, h_out,
                           save, gx_rt0, row_tiles, H, first);
    return fsn_check_launch("gru_step_kernel");
}

int fsn_launch_gru_bptt_step(const float* dh_out, const float* dgx_next, const float* dghn_next, const float* whhT_p,
                             float* carry, const float* save, const float* h_prev, float* dgx, float* dghn,
                             int row_tiles, int H, int last, int first, hipStream_t s) {
    if (row_tiles >= 64 && H % 32 == 0)
        hipLaunchKernelGGL((gru_bptt_step_kernel<2, 2>), dim3(H / 32, (row_tiles + 1) / 2), dim3(256), 0, s, dh_out,
                           dgx_next, dghn_next, whhT_p, carry, save, h_prev, dgx, dghn, row_tiles, H, last, first);
    else
        hipLaunchKernelGGL((gru_bptt_step_kernel<1, 1>), dim3(H / 16, row_tiles), dim3(256), 0, s, dh_out, dgx_next,
                           dghn_next, whhT_p, carry, save, h_prev, dgx, dghn, row_tiles, H, last, first);
    return fsn_check_launch("gru_bptt_step_kernel");
}

// nn.GRU's gate rows as a four-gate cell (fb_chain_kernel<.., CELL = 1>, fsn_gru2_forward: slots r | z | nx | nh; the many-row
// persistent kernels, FSN_REC_GRU in lstm_kernels.hip: slots nh | r | nx | z - `order` 1)
namespace {
__global__ void gru_expand4_kernel(const float* __restrict__ w_ih, const float* __restrict__ w_hh, const float* __restrict__ b_ih,
                                   const float* __restrict__ b_hh, float* __restrict__ w_ih4, float* __restrict__ w_hh4,
                                   float* __restrict__ b4, int I, int H, int order) {
    const long ni = (long)4 * H * I, nh = (long)4 * H * H, n = ni + nh + 4 * H;
    // kind of the gate in slot s: 0 = r, 1 = z, 2 = nx (input part of n), 3 = nh (recurrent part of n)
    const int kinds = order ? (3 | 0 << 2 | 2 << 4 | 1 << 6) : (0 | 1 << 2 | 2 << 4 | 3 << 6);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        if (i < ni) {
            const int row = (int)(i / I), col = (int)(i % I), kind = (kinds >> (2 * (row / H))) & 3, u = row % H;
            w_ih4[i] = kind < 3 ? w_ih[(long)(kind * H + u) * I + col] : 0.f;
        } else if (i < ni + nh) {
            const long k = i - ni;
            const int row = (int)(k / H), col = (int)(k % H), kind = (kinds >> (2 * (row / H))) & 3, u = row % H;
            w_hh4[k] = kind < 2 ? w_hh[(long)(kind * H + u) * H + col] : kind == 2 ? 0.f : w_hh[(long)(2 * H + u) * H + col];
        } else {
            const int row = (int)(i - ni - nh), kind = (kinds >> (2 * (row / H))) & 3, u = row % H;
            b4[row] = kind < 2 ? b_ih[kind * H + u] + b_hh[kind * H + u] : kind == 2 ? b_ih[2 * H + u] : b_hh[2 * H + u];
        }
    }
}
}  // namespace
int fsn_launch_gru_expand4(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, float* w_ih4, float* w_hh4,
                           float* b4, int I, int H, hipStream_t s, int order) {
    const long n = (long)4 * H * (I + H + 1);
    hipLaunchKernelGGL(gru_expand4_kernel, dim3((unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, s, w_ih, w_hh,
                       b_ih, b_hh, w_ih4, w_hh4, b4, I, H, order);
    return fsn_check_launch("gru_expand4_kernel");
}
