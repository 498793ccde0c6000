// EXPERIMENTAL, off by default (FSN_F16X3=1 in the environment of the process switches it on for the sub-band
// layer-1 input projection only): the fp32 GEMM C = A W^T evaluated on the 16-bit matrix cores at fp32 accuracy.
// Both operands are split into two fp16 halves (x = x_hi + x_lo after a power-of-two pre-scale that keeps the low
// halves out of the fp16 subnormals); three v_mfma_f32_16x16x32_f16 per product block (a_hi w_hi + a_hi w_lo +
// a_lo w_hi; the dropped a_lo w_lo term is 2^-22 relative) accumulate in fp32.  Measured (profiles/r01_gemm_probe.md):
// 269 fp32-equivalent TFLOP/s against 138 for the fp32 MFMA kernel, maximum error against an fp64 reference 6e-7
// - smaller than a plain fp32 fma chain's 1.3e-6, because the 384-term sum sees 36 instead of 384 roundings.
// It stays opt-in until it is decided whether "fp32 within 1e-4" (the north star) admits 16-bit matrix
// instructions; profiles/HISTORY.md §8 (split precision).  Same execution shape as gemm_kernel: one 4-wave workgroup per CU, persistent over an
// XCD-partitioned tile list, 4 x 8 wave tile, operands straight from global / L2, refills pinned behind the MFMAs.
#include "fsn_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr float kScaleA = 64.f, kScaleW = 256.f;  // |A| < 1 (hidden states), |W| of order 0.1 - 10

// W [n_out][k] fp32 -> whi / wlo in B-fragment order [n_out/16][k/32][64 lanes][8 halves]:
// lane l of tile (ct, kc) holds W[16 ct + (l & 15)][32 kc + 8 (l >> 4) .. + 7] * kScaleW
__global__ void pack_f16x3_kernel(const float* __restrict__ w, _Float16* __restrict__ whi, _Float16* __restrict__ wlo,
                                  int n_out, int k, float scale) {
    const int kc32 = k / 32;
    const long total = (long)(n_out / 16) * kc32 * 512;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
        const long blk = i >> 9;
        const int kc = (int)(blk % kc32), ct = (int)(blk / kc32);
        float v = w[(long)(ct * 16 + (lane & 15)) * k + kc * 32 + 8 * (lane >> 4) + j] * scale;
        v = fminf(fmaxf(v, -65504.f), 65504.f);  // |w| >= 256 (16 for the layer-0 input weights): saturate, no inf
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

// A row-major [rows][lda] fp32, C = fragment-ordered tiles + bias (the gx layout of the recurrent kernels).
// Operands are staged through LDS: the workgroup (128 rows x 256 columns) fetches every byte once (48 KB per K
// chunk), splits A into fp16 halves once, and the waves read fragments from LDS.  Two stages of 48 KB, one barrier
// per K chunk; the loop runs over (tile, chunk) pairs so that the first chunk of the next tile is already in flight
// while a tile finishes.  (A first, register-direct form - every wave pulling its own operands from L2, tools/
// probe_gemm_f16x3.hip - ran at 15.1 ms on the 3.1 M-row projection; this one 13.0 ms; without the C stores 9.1.)
constexpr int kLdsRT = 8, kLdsCT = 16;                                   // workgroup tile in 16 x 16 tiles
constexpr int kLdsWHalves = 2 * kLdsCT * 512, kLdsAHalves = 2 * kLdsRT * 512;  // halves per stage: W, A (hi + lo)
constexpr int kLdsStageHalves = kLdsWHalves + kLdsAHalves;               // 24576 halves = 48 KB

__global__ __launch_bounds__(512) void gemm_f16x3_lds_kernel(const float* __restrict__ A, long lda,
                                                             const f16x8* __restrict__ whi,
                                                             const f16x8* __restrict__ wlo,
                                                             const float* __restrict__ bias, float* __restrict__ C,
                                                             long row_tiles, int col_tiles, int kc32) {
    // 8 waves = 2 per SIMD, 2 x 4 over the workgroup tile, 4 x 4 tiles each: while one wave of a SIMD waits for
    // LDS, a barrier or its stores, the other one keeps the matrix core busy
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const long rows = row_tiles * 16;
    const unsigned ncb = (unsigned)((col_tiles + kLdsCT - 1) / kLdsCT);
    const unsigned nrb = (unsigned)((row_tiles + kLdsRT - 1) / kLdsRT);
    const unsigned ntiles = nrb * ncb;
    const unsigned xcd = blockIdx.x & 7u, lid = blockIdx.x >> 3, lstride = (gridDim.x + 7u - xcd) >> 3;
    const unsigned tq = ntiles >> 3, tr = ntiles & 7u;
    const unsigned tbeg = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const unsigned tcnt = tq + (xcd < tr ? 1u : 0u);
    if (lid >= tcnt) return;

    // staging registers of one (tile, chunk): this wave's share = W tiles ct = 2 wave, 2 wave + 1 (both halves) and
    // A tile rt = wave (fp32, split on the way into LDS)
    f16x8 wreg[2][2];
    f32x4 areg[2];
    auto load_stage = [&](unsigned ti, int kc) {
        const unsigned v = tbeg + ti;
        const unsigned rb = v / ncb, cb = v % ncb;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int ct = (int)cb * kLdsCT + wave * 2 + j;
            ct = ct < col_tiles ? ct : col_tiles - 1;
            const long o = ((long)ct * kc32 + kc) * 64 + lane;
            wreg[0][j] = whi[o];
            wreg[1][j] = wlo[o];
        }
        long row = ((long)rb * kLdsRT + wave) * 16 + (lane & 15);
        row = row < rows ? row : rows - 1;
        const float* ap = A + row * lda + kc * 32 + 8 * (lane >> 4);
        areg[0] = *reinterpret_cast<const f32x4*>(ap);
        areg[1] = *reinterpret_cast<const f32x4*>(ap + 4);
    };
    auto store_stage = [&](int buf) {
        _Float16* wb = lds + buf * kLdsStageHalves;
        _Float16* ab = wb + kLdsWHalves;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            *reinterpret_cast<f16x8*>(wb + ((wave * 2 + j) * 64 + lane) * 8) = wreg[0][j];
            *reinterpret_cast<f16x8*>(wb + ((kLdsCT + wave * 2 + j) * 64 + lane) * 8) = wreg[1][j];
        }
        f16x8 hi, lo;
        split8(areg[0], areg[1], hi, lo);
        *reinterpret_cast<f16x8*>(ab + (wave * 64 + lane) * 8) = hi;
        *reinterpret_cast<f16x8*>(ab + ((kLdsRT + wave) * 64 + lane) * 8) = lo;
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 outv[4][4];
    long o_rt0 = -1;
    int o_ct0 = 0;
    auto store_out = [&](int rt, int ct) {
        if (o_rt0 + rt < row_tiles && o_ct0 + ct < col_tiles)
            __builtin_nontemporal_store(outv[rt][ct], reinterpret_cast<f32x4*>(C + (((o_rt0 + rt) * col_tiles + o_ct0 + ct) * 64 + lane) * 4));
    };

    unsigned ti = lid;
    int kc = 0, buf = 0;
    load_stage(ti, 0);
    store_stage(0);
    __syncthreads();
    for (;;) {
        unsigned tn = ti;
        int kn = kc + 1;
        if (kn == kc32) {
            kn = 0;
            tn = ti + lstride;
        }
        const bool more = tn < tcnt;
        if (more) load_stage(tn, kn);
        __builtin_amdgcn_sched_barrier(0);
        {
            const _Float16* wb = lds + buf * kLdsStageHalves;
            const _Float16* ab = wb + kLdsWHalves;
            f16x8 ah[4], al[4];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                ah[rt] = *reinterpret_cast<const f16x8*>(ab + ((wr * 4 + rt) * 64 + lane) * 8);
                al[rt] = *reinterpret_cast<const f16x8*>(ab + ((kLdsRT + wr * 4 + rt) * 64 + lane) * 8);
            }
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const f16x8 bh = *reinterpret_cast<const f16x8*>(wb + ((wc * 4 + ct) * 64 + lane) * 8);
                const f16x8 bl = *reinterpret_cast<const f16x8*>(wb + ((kLdsCT + wc * 4 + ct) * 64 + lane) * 8);
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[rt], bh, acc[rt][ct], 0, 0, 0);
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[rt], bl, acc[rt][ct], 0, 0, 0);
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[rt], bh, acc[rt][ct], 0, 0, 0);
            }
        }
        const bool parked_now = kc == kc32 - 1;
        if (parked_now) {  // tile finished: park C = acc / (sA sW) + bias
            if (o_rt0 >= 0 && kc32 <= 8) {  // short K: what the per-chunk stores below did not reach
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    if (q >= 2 * (kc32 - 1)) store_out(q >> 2, q & 3);
            }
            const unsigned v = tbeg + ti;
            const unsigned rb = v / ncb, cb = v % ncb;
            o_rt0 = (long)rb * kLdsRT + wr * 4;
            o_ct0 = (int)cb * kLdsCT + wc * 4;
            const float unscale = 1.0f / (kScaleA * kScaleW);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const int c = o_ct0 + ct < col_tiles ? o_ct0 + ct : col_tiles - 1;
                const float b = bias ? bias[c * 16 + (lane & 15)] : 0.f;
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    const f32x4 a = acc[rt][ct];
                    outv[rt][ct] = f32x4{a[0] * unscale + b, a[1] * unscale + b, a[2] * unscale + b, a[3] * unscale + b};
                    acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        if (!more) break;
        store_stage(buf ^ 1);  // last read one iteration ago, before the barrier that ended it
        // The finished tile does not leave in one burst (16 KB per wave all at once stalls the wave on the store
        // queue, measured 7 of 16 ms): it is parked in registers and goes out two 1 KB stores per K chunk of the
        // NEXT tile - issued after the wait for this iteration's staging loads, so that no wait covers a fresh store.
        if (o_rt0 >= 0 && kc < 8 && !parked_now) {
            switch (kc) {
                case 0: store_out(0, 0); store_out(0, 1); break;
                case 1: store_out(0, 2); store_out(0, 3); break;
                case 2: store_out(1, 0); store_out(1, 1); break;
                case 3: store_out(1, 2); store_out(1, 3); break;
                case 4: store_out(2, 0); store_out(2, 1); break;
                case 5: store_out(2, 2); store_out(2, 3); break;
                case 6: store_out(3, 0); store_out(3, 1); break;
                default: store_out(3, 2); store_out(3, 3); break;
            }
        }
        __syncthreads();
        buf ^= 1;
        ti = tn;
        kc = kn;
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) store_out(q >> 2, q & 3);  // the last tile
}

}  // namespace

// 2 x n_out x k halves: whi then wlo
size_t fsn_f16x3_packed_halves(int n_out, int k) { return (size_t)2 * n_out * k; }

int fsn_launch_pack_f16x3(const float* w, void* packed, int n_out, int k, hipStream_t s, float scale) {
    if (n_out % 16 || k % 32) {
        fsn_set_error("f16x3 pack: n_out %d must be a multiple of 16 and k %d of 32", n_out, k);
        return FSN_ERR_ARG;
    }
    _Float16* whi = static_cast<_Float16*>(packed);
    hipLaunchKernelGGL(pack_f16x3_kernel, dim3(1024), dim3(256), 0, s, w, whi, whi + (size_t)n_out * k, n_out, k, scale);
    return fsn_check_launch("pack_f16x3_kernel");
}

int fsn_launch_gemm_f16x3(const float* A, long lda, const void* packed, const float* bias, float* C, long row_tiles,
                          int n_out, int k, hipStream_t s) {
    const size_t lds = (size_t)2 * kLdsStageHalves * sizeof(_Float16);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16x3_lds_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            fsn_set_error("gemm_f16x3: cannot reserve %zu bytes of LDS", lds);
            return FSN_ERR_LAUNCH;
        }
        attr_set = true;
    }
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const f16x8* whi = static_cast<const f16x8*>(packed);
    const f16x8* wlo = whi + (size_t)n_out * k / 8;
    hipLaunchKernelGGL(gemm_f16x3_lds_kernel, dim3(cus), dim3(512), lds, s, A, lda, whi, wlo, bias, C, row_tiles,
                       n_out / 16, k / 32);
    return fsn_check_launch("gemm_f16x3_lds_kernel");
}
