// EXPERIMENTAL, off by default (FSN_F16X3=1 in the environment of the process switches it on for the sub-band
// layer-1 input projection only): the fp32 GEMM C = A W^T evaluated on the 16-bit matrix cores at fp32 accuracy.
// Both operands are split into two fp16 halves (x = x_hi + x_lo after a power-of-two pre-scale that keeps the low
// halves out of the fp16 subnormals); three v_mfma_f32_16x16x32_f16 per product block (a_hi w_hi + a_hi w_lo +
// a_lo w_hi; the dropped a_lo w_lo term is 2^-22 relative) accumulate in fp32.  Measured (profiles/r01_gemm_probe.md):
// 269 fp32-equivalent TFLOP/s against 138 for the fp32 MFMA kernel, maximum error against an fp64 reference 6e-7
// - smaller than a plain fp32 fma chain's 1.3e-6, because the 384-term sum sees 36 instead of 384 roundings.
// It stays opt-in until it is decided whether "fp32 within 1e-4" (the north star) admits 16-bit matrix
// instructions; DESIGN.md §10.  Same execution shape as gemm_kernel: one 4-wave workgroup per CU, persistent over an
// XCD-partitioned tile list, 4 x 8 wave tile, operands straight from global / L2, refills pinned behind the MFMAs.
#include "fsn_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr float kScaleA = 64.f, kScaleW = 256.f;  // |A| < 1 (hidden states), |W| of order 0.1 - 10

// W [n_out][k] fp32 -> whi / wlo in B-fragment order [n_out/16][k/32][64 lanes][8 halves]:
// lane l of tile (ct, kc) holds W[16 ct + (l & 15)][32 kc + 8 (l >> 4) .. + 7] * kScaleW
__global__ void pack_f16x3_kernel(const float* __restrict__ w, _Float16* __restrict__ whi, _Float16* __restrict__ wlo,
                                  int n_out, int k) {
    const int kc32 = k / 32;
    const long total = (long)(n_out / 16) * kc32 * 512;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
        const long blk = i >> 9;
        const int kc = (int)(blk % kc32), ct = (int)(blk / kc32);
        const float v = w[(long)(ct * 16 + (lane & 15)) * k + kc * 32 + 8 * (lane >> 4) + j] * kScaleW;
        const _Float16 h = (_Float16)v;
        whi[i] = h;
        wlo[i] = (_Float16)(v - (float)h);
    }
}

__device__ __forceinline__ void split8(const f32x4 x0, const f32x4 x1, f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = (j < 4 ? x0[j] : x1[j - 4]) * kScaleA;
        const _Float16 h = (_Float16)v;
        hi[j] = h;
        lo[j] = (_Float16)(v - (float)h);
    }
}

// A row-major [rows][lda] fp32, C = fragment-ordered tiles + bias (the gx layout of the recurrent kernels)
template <int RTW, int CTW>
__global__ __launch_bounds__(256) void gemm_f16x3_kernel(const float* __restrict__ A, long lda,
                                                         const f16x8* __restrict__ whi, const f16x8* __restrict__ wlo,
                                                         const float* __restrict__ bias, float* __restrict__ C,
                                                         long row_tiles, int col_tiles, int kc32) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const long rows = row_tiles * 16;
    const unsigned ncb = (unsigned)((col_tiles + 2 * CTW - 1) / (2 * CTW));
    const unsigned nrb = (unsigned)((row_tiles + 2 * RTW - 1) / (2 * RTW));
    const unsigned ntiles = nrb * ncb;
    const unsigned xcd = blockIdx.x & 7u, lid = blockIdx.x >> 3, lstride = (gridDim.x + 7u - xcd) >> 3;
    const unsigned tq = ntiles >> 3, tr = ntiles & 7u;
    const unsigned tbeg = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const unsigned tcnt = tq + (xcd < tr ? 1u : 0u);
    for (unsigned ti = lid; ti < tcnt; ti += lstride) {
        const unsigned v = tbeg + ti;
        const unsigned rb = v / ncb, cb = v % ncb;
        const long rtile0 = ((long)rb * 2 + wr) * RTW;
        const int ctile0 = ((int)cb * 2 + wc) * CTW;
        const float* arow[RTW];
#pragma unroll
        for (int rt = 0; rt < RTW; ++rt) {
            long row = (rtile0 + rt) * 16 + (lane & 15);
            row = row < rows ? row : rows - 1;
            arow[rt] = A + row * lda + 8 * (lane >> 4);
        }
        long boff[CTW];
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct) {
            int c = ctile0 + ct;
            c = c < col_tiles ? c : col_tiles - 1;
            boff[ct] = (long)c * kc32 * 64 + lane;
        }
        f32x4 acc[RTW][CTW];
#pragma unroll
        for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
            for (int ct = 0; ct < CTW; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 araw[RTW][2];
        f16x8 bh[CTW], bl[CTW];
        auto fetch = [&](int kc) {
#pragma unroll
            for (int rt = 0; rt < RTW; ++rt) {
                araw[rt][0] = *reinterpret_cast<const f32x4*>(arow[rt] + kc * 32);
                araw[rt][1] = *reinterpret_cast<const f32x4*>(arow[rt] + kc * 32 + 4);
            }
#pragma unroll
            for (int ct = 0; ct < CTW; ++ct) {
                bh[ct] = whi[boff[ct] + (long)kc * 64];
                bl[ct] = wlo[boff[ct] + (long)kc * 64];
            }
        };
        fetch(0);
        for (int kc = 0; kc < kc32; ++kc) {
            f16x8 ah[RTW], al[RTW], ch[CTW], cl[CTW];
#pragma unroll
            for (int rt = 0; rt < RTW; ++rt) split8(araw[rt][0], araw[rt][1], ah[rt], al[rt]);
#pragma unroll
            for (int ct = 0; ct < CTW; ++ct) {
                ch[ct] = bh[ct];
                cl[ct] = bl[ct];
            }
            __builtin_amdgcn_sched_barrier(0);  // refill right away; the clamped re-read at the end keeps it branch-free
            fetch(kc + 1 < kc32 ? kc + 1 : kc);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
                for (int ct = 0; ct < CTW; ++ct)
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[rt], ch[ct], acc[rt][ct], 0, 0, 0);
#pragma unroll
            for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
                for (int ct = 0; ct < CTW; ++ct)
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[rt], cl[ct], acc[rt][ct], 0, 0, 0);
#pragma unroll
            for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
                for (int ct = 0; ct < CTW; ++ct)
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[rt], ch[ct], acc[rt][ct], 0, 0, 0);
        }
        const float unscale = 1.0f / (kScaleA * kScaleW);
#pragma unroll
        for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
            for (int ct = 0; ct < CTW; ++ct)
                if (rtile0 + rt < row_tiles && ctile0 + ct < col_tiles) {
                    const f32x4 a = acc[rt][ct];
                    const float b = bias ? bias[(ctile0 + ct) * 16 + (lane & 15)] : 0.f;
                    *reinterpret_cast<f32x4*>(C + (((rtile0 + rt) * col_tiles + ctile0 + ct) * 64 + lane) * 4) =
                        f32x4{a[0] * unscale + b, a[1] * unscale + b, a[2] * unscale + b, a[3] * unscale + b};
                }
    }
}

}  // namespace

// 2 x n_out x k halves: whi then wlo
size_t fsn_f16x3_packed_halves(int n_out, int k) { return (size_t)2 * n_out * k; }

int fsn_launch_pack_f16x3(const float* w, void* packed, int n_out, int k, hipStream_t s) {
    if (n_out % 16 || k % 32) {
        fsn_set_error("f16x3 pack: n_out %d must be a multiple of 16 and k %d of 32", n_out, k);
        return FSN_ERR_ARG;
    }
    _Float16* whi = static_cast<_Float16*>(packed);
    hipLaunchKernelGGL(pack_f16x3_kernel, dim3(1024), dim3(256), 0, s, w, whi, whi + (size_t)n_out * k, n_out, k);
    return fsn_check_launch("pack_f16x3_kernel");
}

int fsn_launch_gemm_f16x3(const float* A, long lda, const void* packed, const float* bias, float* C, long row_tiles,
                          int n_out, int k, hipStream_t s) {
    auto kern = gemm_f16x3_kernel<4, 8>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                96 * 1024) != hipSuccess) {
            fsn_set_error("gemm_f16x3: cannot reserve LDS");
            return FSN_ERR_LAUNCH;
        }
        attr_set = true;
    }
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const f16x8* whi = static_cast<const f16x8*>(packed);
    const f16x8* wlo = whi + (size_t)n_out * k / 8;
    hipLaunchKernelGGL(kern, dim3(cus), dim3(256), 96 * 1024, s, A, lda, whi, wlo, bias, C, row_tiles, n_out / 16, k / 32);
    return fsn_check_launch("gemm_f16x3_kernel");
}
