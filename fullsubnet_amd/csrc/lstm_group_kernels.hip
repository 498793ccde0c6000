// Sub-band model for FEW rows (6 - 9 utterances: the per-rank share of a strong-scaled batch, small serving batches):
// both LSTM layers and the output layer as ONE persistent launch in which clusters of workgroups share a group of 64
// rows (fullsubnet/model.py:121-128: the rows are independent sequences).
//
// Why: with ~8 rows per CU the persistent kernels of lstm_kernels.hip cannot be used - a workgroup that owns rows
// must stream ALL of W_hh (2.4 MB) from L2 every step whatever its row count, which bounds a step at ~31 us - so
// this regime ran as one launch per step and layer (hidden units x row tiles spread over all CUs, h and c through
// L2): 29 us per step against 15 us of MFMA work (launch boundary, cold operand fetches, drain), plus a separate
// projection GEMM per layer with its gx round trip.  Here the same work split stays resident:
//   - workgroup (cluster c, member m, layer l) owns hidden units [48 m, 48 m + 48) of layer l for the 64 rows of
//     cluster c: wave w = row tile w, 3 unit groups x 4 gates = 12 accumulator tiles, c_t in registers.  Eight members
//     x two layers = 16 workgroups per cluster, two per CU: the layer-0 and the layer-1 workgroups of a CU are
//     independent instruction streams, so one's barriers, cell update and hand-off waits run under the other's MFMAs;
//   - each layer's K loop contains its input projection (K = 32 + 384 and 384 + 384: no projection GEMMs, no gx);
//     layer 1 follows layer 0 at a distance of up to two steps (the depth of layer 0's exchange buffer);
//   - members exchange h slices through a small global buffer per cluster (48 columns written, 384 read) with the
//     write-through / flag / acquire recipe of the CDNA guide (Guideline 16, R1: sc1 stores, drained, ONE flag store;
//     relaxed poll by one wave, ONE agent-scope acquire, barrier, plain loads);
//   - weight fragments are fetched once per workgroup and shared by its four waves through a two-stage LDS buffer
//     (the K loop of lstm_step_cu_kernel), 0.9 MB per CU and step instead of 2.4 - 4.7 MB; the A fragments (other
//     CUs' write-through data: first touch comes from the Infinity Cache / HBM) are requested four chunks ahead.
// The output layer of step s is computed by layer-1 member m for rows 8 m .. 8 m + 7 when h1_s has been gathered for the
// next step anyway.  Every spin is bounded: on a timeout the workgroup raises `status` and all waits fall through (the
// results are then garbage, never a hang); flags and status are zeroed by the host before every launch.
// (Round 6: both layers of a member as ONE instruction stream per CU - lstm2_duo_kernel, commit 1df4f24 - was built, is
// bit-identical and SLOWER: 73 us per step against 57; its floor with every load, poll and non-linearity removed is 53 us, one
// wave per SIMD hides none of its own latencies.  profiles/r06_duo_probe.md.)
#include "fsn_common.h"

#ifndef FSN_GRP_AD
#define FSN_GRP_AD 6        // A fragments in flight (probe, 32 clusters: 4 -> 11.34 ms, 6 -> 11.2, 8 -> 11.4 with spills)
#endif
#ifndef FSN_GRP_CPS
#define FSN_GRP_CPS 2       // K chunks per LDS stage of the weight fragments = per workgroup barrier (probe: 1 -> 11.09 ms, 2 -> 10.88)
#endif
#ifndef FSN_GRP_CPS16
#define FSN_GRP_CPS16 4     // the same under the 16-bit arithmetic: fragments are half as large and a chunk is 8x fewer MFMA cycles
#endif
#ifndef FSN_GRP_AD16
#define FSN_GRP_AD16 8      // a multiple of FSN_GRP_CPS16
#endif
#ifndef FSN_GRP_VGPR
#define FSN_GRP_VGPR 108    // register cap of the two-workgroups-per-CU forms
#endif
#ifndef FSN_GRP_BOOST
#define FSN_GRP_BOOST 12    // experiment (ABL 8192): first chunk of layer 0's K loop at the raised priority
#endif
#ifndef FSN_GRP_LEAD
#define FSN_GRP_LEAD 2      // experiment (ABL 2048)
#endif
#ifndef FSN_GRP_D0
#define FSN_GRP_D0 4
#endif
#ifndef FSN_GRP_BIAS_LDS
#define FSN_GRP_BIAS_LDS 1  // biases in LDS also with one cluster per workgroup set (frees 12 registers for the ring)
#endif

namespace {

template <int AR>
constexpr int grp_cps() { return AR == FSN_ARITH_F32 ? FSN_GRP_CPS : FSN_GRP_CPS16; }
template <int AR>
constexpr int grp_ad() { return AR == FSN_ARITH_F32 ? FSN_GRP_AD : FSN_GRP_AD16; }

constexpr int GH = 384;          // hidden units (both layers)
constexpr int GKC = GH / 16;     // K chunks of an H-wide operand
constexpr int GM = 8;            // members per cluster and layer
constexpr int GU = GH / 16 / GM; // unit groups per member (3)
constexpr int GROWS = 64;        // rows per cluster
constexpr int GD0 = FSN_GRP_D0;           // depth of layer 0's exchange buffer: layer 0 may run GD0 - 2 steps ahead of layer 1
constexpr int GFS = 32;          // words between the flag groups of (cluster, layer): one 128-byte line each

struct GrpArgs {
    FsnSbInput xin;        // layer-0 input (mag, fb_out, den, row0, N ...); xin.bias = b0
    const float* wbase;    // the four packed weight matrices live in one buffer: element offsets from wbase
    unsigned o_wih0;       // packed [4H/16][2][64][4]
    unsigned o_whh0;       // packed [4H/16][KC][64][4]
    unsigned o_wih1, o_whh1;
    const float* bias1;    // b_ih + b_hh of layer 1 [4H]
    float* hx0;            // [clusters][GD0][64][H]   h of layer 0
    float* hx1;            // [clusters][2][64][H]     h of layer 1
    unsigned* flags;       // [clusters][2][GFS]: steps published so far by (layer, member), GM words used per group
    unsigned* status;      // 0 = fine
    unsigned long long spin_ticks;  // wait bound (fsn_spin_ticks)
    FsnRecFc fc;           // output layer; fc.N = valid local rows
    int Tp;
    // training form (TRAIN): the exchange buffers ARE the hidden sequences - hx0 / hx1 = hseq0 / hseq1 [Tp][Nrows][H],
    // a cluster's tile of step t at rows 64 c .. of step t (nothing is reused) - and every step keeps the activated
    // gates [Tp][Nrows][4H] and the cell state [Tp][Nrows][H] (the layouts of fsn_lstm_layer_forward)
    float *gates0, *cseq0, *gates1, *cseq1;
    int Nrows;
    int nclusters;         // clusters of this launch (the grid has min(nclusters, CUs / 8) workgroup sets)
    // GX form (with TRAIN): layer 0 starts from its projection, computed beforehand by a GEMM (any input width) - gx as
    // fragment tiles [Tp][gx_tiles][4H/16][64][4], bias included - and Nrows need not be a multiple of 64 (the last
    // cluster's missing 16-row tiles load a valid tile's rows and store nothing)
    const float* gx;
    int gx_tiles;
    unsigned long long* dbg;  // tools/probe_group.hip's timeline (ABL 4096): [layer][member][Tp + 1][8] clock stamps of cluster 0
};

// Several weight sets in one launch (GX form): the sections of improved_fullsubnet/model.py:402-449 are independent
// two-layer stacks over the same frames, each with its own weights, input width and row count.  Set i owns clusters
// [cluster0_i, cluster0_{i+1}).
constexpr int GSETS = 8;
struct GrpSet {
    const float* gx;
    float *hseq0, *hseq1;
    const float* bias1;
    unsigned o_whh0, o_wih1, o_whh1;  // element offsets from GrpArgs::wbase
    int N;                            // rows of the set (a multiple of 16)
    int cluster0;
};
struct GrpSets {
    GrpSet s[GSETS];
    int n;
};

__device__ __forceinline__ void store_sc1(float* p, float v) {
    // write-through store (sc1): the line leaves this XCD's L2, any CU of the chip reads it after an agent-scope acquire
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One wave polls the eight member flags of a layer (relaxed agent-scope loads, never served by this CU's L1) until all
// have reached `epoch`; bounded.  Returns false after a timeout (status raised).
__device__ __forceinline__ bool grp_poll(unsigned* flags8, unsigned epoch, unsigned* status, unsigned long long ticks) {
    const int lane = threadIdx.x & 63;
    unsigned long long t0 = 0;
    for (unsigned spins = 0;; ++spins) {
        unsigned v = epoch;
        if (lane < GM) v = __hip_atomic_load(flags8 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all((int)(v >= epoch))) return true;
        if ((spins & 255u) == 255u && fsn_wait_give_up(status, spins, t0, ticks, 1u + epoch)) return false;
        __builtin_amdgcn_s_sleep(1);
    }
}

// ABL: experiment knob of tools/probe_group.hip (0 in the library; any bit set gives WRONG results): 1 no acquire
// fence, 2 no flag polling, 4 plain instead of write-through stores, 8 no gate non-linearities, 16 no output layer,
// 32 A fragments not loaded (a constant instead), 128 weight fragments not loaded, 256 no LDS stage of the weight fragments
// (operands from registers, no K-loop barriers), 512 members (not clusters) share an XCD.
// What a workgroup keeps per cluster.  A workgroup serves one cluster, or TWO alternately (batches of 9 - 16 utterances:
// more clusters than the chip holds at once): step t of cluster A, step t of cluster B, step t + 1 of A ... - while it
// works on one cluster the partners' flags and write-through data of the other are on their way.
struct GrpCl {
    int cluster;
    int row_l;    // this lane's A-operand row (local to the launch)
    bool row_ok;
    bool tile_ok;  // GX form: this wave's 16-row tile exists (wave-uniform)
    long ng;
    int xb, xf;   // layer-0 input of this lane's row: (b, f)
    float *hx0, *hx1;
    unsigned *fl0, *fl1;
    __amdgpu_buffer_rsrc_t xrsrc0, xrsrc1;
    float c[GU][4];
    unsigned seen0;
};

// TRAIN: the general two-layer form (plain row-major input, hidden sequences = exchange buffers, no output layer);
// SAVE (with TRAIN): also keep the activated gates and cell states (the training forward)
// AR: arithmetic of the products (fsn_mma_k16: FSN_ARITH_F32, or 16-bit operands with fp32 accumulation for autocast
// training); data movement and everything stored are the same in every mode
template <int LAYER, int ABL, bool TRAIN, int NCL, bool SAVE, int AR, bool GX = false>
__device__ __forceinline__ void group_body(const GrpArgs& a, int cluster_a, int cluster_b, int member,
                                           typename FsnWFrag<AR>::type (*bsh)[GU * 4 * grp_cps<AR>()][64], float (*bias_sh)[16]) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const int Tp = a.Tp;
    const FsnSbInput& x = a.xin;
    auto stamp = [&](int cl, int t, int e) {
        if constexpr ((ABL & 4096) != 0) {
            if (cl == 0 && threadIdx.x == 0) {
                __builtin_amdgcn_sched_barrier(0);
                a.dbg[(((size_t)LAYER * GM + member) * (Tp + 1) + t) * 8 + e] = (unsigned long long)wall_clock64();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // byte offset of the tile that holds h0_t / h1_t inside hx0 / hx1
    const unsigned step_bytes = TRAIN ? (unsigned)a.Nrows * GH * 4u : 0u;
    auto slot0 = [&](int t) { return TRAIN ? (unsigned)t * step_bytes : (unsigned)((t % GD0) * GROWS * GH * 4); };
    auto slot1 = [&](int t) { return TRAIN ? (unsigned)t * step_bytes : (unsigned)((t & 1) * GROWS * GH * 4); };
    auto init = [&](GrpCl& k, int cluster) {
        k.cluster = cluster;
        k.row_l = cluster * GROWS + wave * 16 + lr;
        k.hx0 = a.hx0 + (TRAIN ? (size_t)cluster * GROWS * GH : (size_t)cluster * GD0 * GROWS * GH);
        k.hx1 = a.hx1 + (TRAIN ? (size_t)cluster * GROWS * GH : (size_t)cluster * 2 * GROWS * GH);
        k.fl0 = a.flags + ((size_t)cluster * 2 + 0) * GFS;
        k.fl1 = a.flags + ((size_t)cluster * 2 + 1) * GFS;
        k.xrsrc0 = __builtin_amdgcn_make_buffer_rsrc(k.hx0, 0, 0x7fffffff, 0x00020000);
        k.xrsrc1 = __builtin_amdgcn_make_buffer_rsrc(k.hx1, 0, 0x7fffffff, 0x00020000);
        k.row_ok = k.row_l < x.N;
        k.tile_ok = !GX || cluster * GROWS + wave * 16 < a.Nrows;
        k.ng = k.row_l + x.row0;
        k.xb = k.row_ok ? (int)(k.ng / x.F) : 0;
        k.xf = k.row_ok ? (int)(k.ng % x.F) : 0;
#pragma unroll
        for (int u = 0; u < GU; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) k.c[u][i] = 0.f;
        k.seen0 = 0u;
    };
    // this lane's A fragment inside a [64][H] tile of the exchange buffers (byte offset), read with sc1 buffer loads: the
    // partners stored write-through (sc1), so an sc1 load - never served by this CU's L1 - needs no acquire fence
    // (GX form: a missing tile of the last cluster reads the cluster's first tile instead)
    static_assert(!GX || (TRAIN && NCL == 1 && !SAVE), "the GX form is the general inference form, one cluster per set");
    const bool my_tile = !GX || cluster_a * GROWS + wave * 16 < a.Nrows;
    const unsigned a_off = (unsigned)((((my_tile ? wave * 16 : 0) + lr) * GH + 4 * lq) * 4);
    auto xload = [&](const __amdgpu_buffer_rsrc_t& r, unsigned voff, unsigned soff) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 16));  // aux 16 = sc1
    };

    // biases of this member's 12 column tiles: registers, or LDS when the workgroup keeps two clusters' state (12
    // registers decide between fitting and spilling there; with one cluster the registers are faster)
    constexpr bool kBiasLds = NCL > 1 || FSN_GRP_BIAS_LDS;
    float bias[GU][4];
    if (GX && LAYER == 0) {
        // the projection tiles carry layer 0's bias
    } else if (kBiasLds) {
        if (threadIdx.x < GU * 4 * 16) {
            const int f = threadIdx.x >> 4, u = f >> 2, g = f & 3, l = threadIdx.x & 15;
            bias_sh[f][l] = (LAYER ? a.bias1 : a.xin.bias)[(g * GKC + member * GU + u) * 16 + l];
        }
        __syncthreads();
    } else {
#pragma unroll
        for (int u = 0; u < GU; ++u)
#pragma unroll
            for (int g = 0; g < 4; ++g) bias[u][g] = (LAYER ? a.bias1 : a.xin.bias)[(g * GKC + member * GU + u) * 16 + lr];
    }

    // ---- K loop: acc += A(16 rows x 16 n) B(16 n x [3 unit groups x 4 gates x 16]) over two operand segments -------
    // segment 1: n1 chunks, A from registers (xa) or from a1 (global, row stride GH), B from b1 (chunk stride 256,
    // column-tile stride s1); segment 2: n2 chunks, A from a2, B from b2 (column-tile stride s2).  Fragment f of a chunk
    // = (unit group f >> 2, gate f & 3); wave w fetches fragments 3 w .. 3 w + 2 and parks them in LDS.
    // weight fragments by buffer load: resource descriptor + scalar byte offset + this lane's 16 l (no per-fragment
    // pointers in vector registers)
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wbase), 0, 0x7fffffff, 0x00020000);
    // A operand: xa (registers, layer-0 input) or tile `at1` / `at2` (byte offset) of exchange buffer ab1 / ab2 (0 / 1)
    // (descriptors by value: a reference to one of two descriptors puts both on the stack)
    auto kloop = [&](f32x4 (&acc)[GU][4], const f32x4* xa, const __amdgpu_buffer_rsrc_t r1, unsigned at1, unsigned b1,
                     unsigned s1, int n1, const __amdgpu_buffer_rsrc_t r2, unsigned at2, unsigned b2, unsigned s2, int n2) {
        const int n = n1 + n2;
        // The A fragments come from the exchange buffers, which were written through to memory by other CUs: their
        // first touch after the acquire costs a trip to the Infinity Cache / HBM, several chunks of MFMA time.  They
        // are therefore requested AD chunks ahead (a register ring, indexed statically by unrolling the loop AD-fold);
        // the weight fragments (L2 hits) one chunk ahead, through LDS.
        constexpr int AD = grp_ad<AR>();
        f32x4 ar[AD];
        typename FsnWFrag<AR>::type bn[GU];
        auto fetch_a = [&](int k) -> f32x4 {
            const int kc = k < n ? k : n - 1;
            if (ABL & 32) return f32x4{0.5f, 0.25f, 0.125f, 0.0625f};
            if constexpr ((ABL & 16384) != 0) {  // experiment (wrong data): a wave's fragment as ONE contiguous 1 KB block of the tile
                const unsigned fo = (unsigned)lane * 16u, blk = (unsigned)wave * 1024u;
                if (kc < n1) {
                    if (xa) return kc == 0 ? xa[0] : xa[1];
                    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r1, fo, at1 + (unsigned)kc * 4096u + blk, 16));
                }
                return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r2, fo, at2 + (unsigned)(kc - n1) * 4096u + blk, 16));
            }
            if (kc < n1) {
                if (xa) return kc == 0 ? xa[0] : xa[1];
                return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r1, a_off, at1 + (unsigned)kc * 64u, 16));
            }
            return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r2, a_off, at2 + (unsigned)(kc - n1) * 64u, 16));
        };
        auto fetch_b = [&](int k) {
            const int kc = k < n ? k : n - 1;
            const bool first = kc < n1;
            const int kk = first ? kc : kc - n1;
            const unsigned bb = first ? b1 : b2, cs = first ? s1 : s2;
#pragma unroll
            for (int j = 0; j < GU; ++j) {
                const int f = wave * GU + j, u = f >> 2, g = f & 3;
                const unsigned ofs = bb + ((unsigned)(g * GKC + member * GU + u) * cs + (unsigned)kk) * 256u;
                if constexpr ((ABL & 128) != 0 && AR == FSN_ARITH_F32) bn[j] = f32x4{0.01f, 0.02f, -0.01f, 0.005f};
                else bn[j] = fsn_load_wfrag<AR>(wrsrc, (unsigned)lane, ofs);
            }
        };
#pragma unroll
        for (int d = 0; d < AD; ++d) ar[d] = fetch_a(d);
        constexpr int CPS = grp_cps<AR>();
        static_assert(AD % CPS == 0, "the A ring turns in whole stages");
#pragma unroll
        for (int c = 0; c < CPS; ++c) {
            fetch_b(c);
#pragma unroll
            for (int j = 0; j < GU; ++j)
                if (!(ABL & 256)) bsh[0][c * GU * 4 + wave * GU + j][lane] = bn[j];
        }
        if (!(ABL & 256)) __syncthreads();
        for (int k0 = 0; k0 < n; k0 += AD) {
            if constexpr ((ABL & 8192) != 0 && LAYER == 0) {  // experiment: layer 0 issues first in the tail of its K loop
                if (k0 == FSN_GRP_BOOST) __builtin_amdgcn_s_setprio(3);
            }
#pragma unroll
            for (int d = 0; d < AD; ++d) {
                const int k = k0 + d;
                if (k < n) {  // uniform (n is a multiple of CPS)
                    const int c = d % CPS, buf = (k / CPS) & 1;
                    __builtin_amdgcn_sched_barrier(0);  // requests first, pinned under this chunk's MFMAs
                    const f32x4 av = ar[d];
                    if constexpr ((ABL & 32768) != 0) {  // experiment: the A request first (the wait for the weight fragments then covers it)
                        ar[d] = fetch_a(k + AD);
                        __builtin_amdgcn_sched_barrier(0);
                        fetch_b(k + CPS);
                    } else {
                        fetch_b(k + CPS);                   // the same chunk of the next stage
                        ar[d] = fetch_a(k + AD);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (AR == FSN_ARITH_F32) {
#pragma unroll
                        for (int u = 0; u < GU; ++u) {
                            f32x4 b[4];
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                if constexpr ((ABL & 256) != 0) b[g] = bn[(u + g) % GU];
                                else b[g] = bsh[buf][c * GU * 4 + u * 4 + g][lane];
                            }
#pragma unroll
                            for (int j = 0; j < 4; ++j)
#pragma unroll
                                for (int g = 0; g < 4; ++g) acc[u][g] = mfma16(av[j], b[g][j], acc[u][g]);
                        }
                    } else {  // 16-bit operands: one matrix instruction per tile and K chunk
                        const typename FsnOperand<AR>::type ao = fsn_operand<AR>(av);
#pragma unroll
                        for (int u = 0; u < GU; ++u)
#pragma unroll
                            for (int g = 0; g < 4; ++g)
                                acc[u][g] = fsn_mma_k16<AR>(ao, fsn_wfrag_operand<AR>(bsh[buf][c * GU * 4 + u * 4 + g][lane]), acc[u][g]);
                    }
#pragma unroll
                    for (int j = 0; j < GU; ++j)
                        if (!(ABL & 256)) bsh[buf ^ 1][c * GU * 4 + wave * GU + j][lane] = bn[j];
                    if (c == CPS - 1 && !(ABL & 256)) __syncthreads();
                }
            }
        }
        if (n % CPS && !(ABL & 256)) __syncthreads();  // a last, partial stage (layer 0's 2 + 24 chunks at four per stage): close it as well
    };

    // Flags are looked at EARLY (before a K loop) and checked after it: in the steady state the early look already
    // shows the awaited epoch and the check costs nothing; only otherwise does wave 0 poll.  No acquire fence: every
    // load of exchanged data is an sc1 load (see xload).
    auto peek = [&](unsigned* flags8) -> unsigned {
        unsigned v = 0xffffffffu;
        if (wave == 0 && lane < GM) v = __hip_atomic_load(flags8 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return v;
    };
    auto wait_peeked = [&](unsigned v, unsigned* flags8, unsigned epoch) {
        if (wave == 0 && !(ABL & 2) && !__all((int)(v >= epoch))) (void)grp_poll(flags8, epoch, a.status, a.spin_ticks);
        __syncthreads();  // one wave looked for all four
    };
    // h slice of this step is in flight (write-through): every wave drains, then one lane bumps the flag
    auto publish = [&](unsigned* flag, unsigned epoch) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // cell update of this wave's 16 rows x 48 units; h_t slice -> exchange buffer
    auto cell = [&](GrpCl& k, f32x4 (&acc)[GU][4], float* hdst, float* gates_t, float* cseq_t) {
#pragma unroll
        for (int u = 0; u < GU; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float ig, fg, gg, og;
                if (ABL & 8) {
                    ig = acc[u][0][i], fg = acc[u][1][i], gg = acc[u][2][i], og = acc[u][3][i];
                } else {
                    ig = sigmoid_fast(acc[u][0][i]), fg = sigmoid_fast(acc[u][1][i]);
                    gg = tanh_fast(acc[u][2][i]), og = sigmoid_fast(acc[u][3][i]);
                }
                const float cn = fg * k.c[u][i] + ig * gg;
                k.c[u][i] = cn;
                float* hp = hdst + (size_t)(wave * 16 + 4 * lq + i) * GH + (member * GU + u) * 16 + lr;
                const float hv = (ABL & 8) ? og * cn : og * tanh_fast(cn);
                if (GX && !k.tile_ok) continue;
                if (ABL & 4) *hp = hv;
                else store_sc1(hp, hv);
                if (TRAIN && SAVE) {  // kept in place of the pre-activations for save_cell (after the hand-off)
                    acc[u][0][i] = ig, acc[u][1][i] = fg, acc[u][2][i] = gg, acc[u][3][i] = og;
                }
            }
    };
    // Training: the activated gates and the cell state of the step, stored AFTER the hand-off - publish() drains the
    // wave's stores before the flag goes out, and only the 12 h values belong to the hand-off: with the 60 saved values
    // issued first every step's flag waited for their trip to HBM as well (same values, same addresses: bit-equal)
    auto save_cell = [&](GrpCl& k, f32x4 (&acc)[GU][4], float* gates_t, float* cseq_t) {
#pragma unroll
        for (int u = 0; u < GU; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) {  // rows of this cluster inside step t's [Nrows][...] slabs
                const size_t row = (size_t)k.cluster * GROWS + wave * 16 + 4 * lq + i;
                const int unit = (member * GU + u) * 16 + lr;
                float* gp = gates_t + row * (4 * GH) + unit;
                gp[0] = acc[u][0][i];
                gp[GH] = acc[u][1][i];
                gp[2 * GH] = acc[u][2][i];
                gp[3 * GH] = acc[u][3][i];
                cseq_t[row * GH + unit] = k.c[u][i];
            }
    };

    if (LAYER == 0) {
        auto iter0 = [&](int t, GrpCl& k) {
            // the layer-0 input of this lane's row at frame t (two A fragments: columns 4 lq .. and 16 + 4 lq ..):
            // requested now, divided after the wait below
            float raw[8];
            f32x4 acc[GU][4];
            stamp(k.cluster, t, 0);
            const float den = (k.row_ok && !TRAIN) ? x.den[x.den_mode ? (long)t * x.den_stride + k.ng : (long)k.xb] : 1.f;
            if (GX) {  // the projection tiles of this wave's rows at frame t (bias included): requested before the wait
                const int tile = k.cluster * (GROWS / 16) + (k.tile_ok ? wave : 0);
                const float* gp = a.gx + (((size_t)t * a.gx_tiles + tile) * (4 * GKC)) * 256 + lane * 4;
#pragma unroll
                for (int u = 0; u < GU; ++u)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        acc[u][g] = *reinterpret_cast<const f32x4*>(gp + (size_t)(g * GKC + member * GU + u) * 256);
            } else if (TRAIN) {  // plain row-major input [Tp][x_step][x_ld], 32 columns (zero-padded by the caller)
                const float* xr = x.x_rows + ((long)t * x.x_step + (k.row_ok ? k.row_l : 0)) * x.x_ld + 4 * lq;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(xr);
                const f32x4 v1 = x.kin_chunks > 1 ? *reinterpret_cast<const f32x4*>(xr + 16) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) raw[e] = v0[e], raw[4 + e] = v1[e];
            } else {
                const long fo = ((long)k.xb * x.Tp + t) * x.FP;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int cc = (e >> 2) * 16 + 4 * lq + (e & 3);
                    int jj = k.xf + cc - x.nb;
                    jj = jj < 0 ? -jj : jj;
                    jj = jj >= x.F ? 2 * (x.F - 1) - jj : jj;
                    const bool ok = k.row_ok && cc <= 2 * x.nb + 1;
                    const float* src = cc <= 2 * x.nb ? x.mag + fo + jj : x.fb_out + fo + k.xf;
                    raw[e] = *(ok ? src : x.mag);
                }
            }
            if (t > 0) wait_peeked(peek(k.fl0), k.fl0, (unsigned)t);  // h0_{t-1} of all members (just published: polls)
            stamp(k.cluster, t, 1);
            f32x4 xa[2];
            if (!GX) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int cc = (e >> 2) * 16 + 4 * lq + (e & 3);
                    if (TRAIN) xa[e >> 2][e & 3] = k.row_ok ? raw[e] : 0.f;
                    else xa[e >> 2][e & 3] = (k.row_ok && cc <= 2 * x.nb + 1) ? raw[e] / den : 0.f;
                }
#pragma unroll
                for (int u = 0; u < GU; ++u)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float b = kBiasLds ? bias_sh[u * 4 + g][lr] : bias[u][g];
                        acc[u][g] = f32x4{b, b, b, b};
                    }
            }
            const unsigned ring = t >= GD0 ? peek(k.fl1) : 0xffffffffu;
            if constexpr ((ABL & 2048) != 0) {  // experiment: layer 0 issues first while it is less than two steps ahead
                unsigned v = lane < GM ? __hip_atomic_load(k.fl1 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
                v = min(v, (unsigned)__shfl_xor((int)v, 1, 64));
                v = min(v, (unsigned)__shfl_xor((int)v, 2, 64));
                v = min(v, (unsigned)__shfl_xor((int)v, 4, 64));
                const int done1 = __builtin_amdgcn_readfirstlane((int)v);  // steps layer 1 has published
                if (t - done1 < FSN_GRP_LEAD) __builtin_amdgcn_s_setprio(2);
                else __builtin_amdgcn_s_setprio(0);
            }
            if (!GX) kloop(acc, xa, k.xrsrc0, 0, a.o_wih0, 2, 2, k.xrsrc0, t > 0 ? slot0(t - 1) : 0u, a.o_whh0, GKC, t > 0 ? GKC : 0);
            else if (t > 0) kloop(acc, nullptr, k.xrsrc0, 0, 0, 0, 0, k.xrsrc0, slot0(t - 1), a.o_whh0, GKC, GKC);
            // slot t % GD0 still holds h0_{t-GD0}: layer 1 must have finished its step t - GD0 (it reads that slot
            // there) - all eight layer-1 members, i.e. they have published step t - GD0 + 1
            stamp(k.cluster, t, 2);
            if (t >= GD0) wait_peeked(ring, k.fl1, (unsigned)(t - GD0 + 1));
            stamp(k.cluster, t, 3);
            cell(k, acc, reinterpret_cast<float*>(reinterpret_cast<char*>(k.hx0) + slot0(t)),
                 TRAIN ? a.gates0 + (size_t)t * a.Nrows * 4 * GH : nullptr, TRAIN ? a.cseq0 + (size_t)t * a.Nrows * GH : nullptr);
            stamp(k.cluster, t, 4);
            publish(k.fl0 + member, (unsigned)t + 1);
            if constexpr ((ABL & 8192) != 0) __builtin_amdgcn_s_setprio(0);
            stamp(k.cluster, t, 5);
            if (TRAIN && SAVE) save_cell(k, acc, a.gates0 + (size_t)t * a.Nrows * 4 * GH, a.cseq0 + (size_t)t * a.Nrows * GH);
        };
        GrpCl ka, kb;
        init(ka, cluster_a);
        if (NCL > 1 && cluster_b >= 0) init(kb, cluster_b);
        for (int t = 0; t < Tp; ++t) {
            iter0(t, ka);
            if (NCL > 1 && cluster_b >= 0) iter0(t, kb);
        }
    } else {
        // output layer: thread (d = tid >> 4, p = tid & 15) owns 24 of the 384 terms of dot product d (row d >> 1 of
        // this member's eight rows, output d & 1)
        const int d = threadIdx.x >> 4, p = threadIdx.x & 15;
        const int frow = member * 8 + (d >> 1), fcc = d & 1;
        auto iter1 = [&](int s, GrpCl& k) {
            // Step s: x_s W_ih^T first - it only needs h0_s, which layer 0 published long ago - so that the partners'
            // h1_{s-1}, published a moment ago, has a whole half K loop to arrive before anyone waits for it.  The extra
            // iteration s = Tp only computes the output layer of the last step.
            f32x4 acc[GU][4];
            unsigned seen1 = 0xffffffffu;
            stamp(k.cluster, s, 0);
            if (s < Tp) {
                wait_peeked(k.seen0, k.fl0, (unsigned)s + 1);
                stamp(k.cluster, s, 1);
                seen1 = peek(k.fl1);
#pragma unroll
                for (int u = 0; u < GU; ++u)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float b = kBiasLds ? bias_sh[u * 4 + g][lr] : bias[u][g];
                        acc[u][g] = f32x4{b, b, b, b};
                    }
                kloop(acc, nullptr, k.xrsrc0, slot0(s), a.o_wih1, GKC, GKC, k.xrsrc0, 0, 0, 0, 0);
            } else {
                seen1 = peek(k.fl1);
            }
            stamp(k.cluster, s, 2);
            if (s > 0) {
                wait_peeked(seen1, k.fl1, (unsigned)s);  // h1_{s-1} of all members
                stamp(k.cluster, s, 3);
                // Output layer (nn.Linear(384, 2)) of step s - 1 for rows 8 m .. 8 m + 7 of the cluster, from h1_{s-1} as
                // it has just been gathered: 16 dot products x 16 threads (~1 us, covered by the layer-0 workgroup
                // that shares the CU)
                if (!(ABL & 16) && !TRAIN) {
                    const unsigned hoff = (unsigned)((((s - 1) & 1) * GROWS * GH + frow * GH + p * 24) * 4);
                    f32x4 hv[6], fw[6];
#pragma unroll
                    for (int q = 0; q < 6; ++q) {
                        const int kk = p * 24 + 4 * q;  // four consecutive k: one 16-byte group of the packed weights
                        hv[q] = xload(k.xrsrc1, hoff, 16u * q);
                        fw[q] = *reinterpret_cast<const f32x4*>(a.fc.w_p + (((kk >> 4) * 64) + ((kk & 15) >> 2) * 16 + fcc) * 4);
                    }
                    float acc1 = 0.f;
#pragma unroll
                    for (int q = 0; q < 6; ++q)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc1 = fmaf(hv[q][j], fw[q][j], acc1);
                    acc1 += __shfl_xor(acc1, 1, 64);
                    acc1 += __shfl_xor(acc1, 2, 64);
                    acc1 += __shfl_xor(acc1, 4, 64);
                    acc1 += __shfl_xor(acc1, 8, 64);
                    const long n = (long)k.cluster * GROWS + frow;
                    const int so = s - 1;
                    if (p == 0 && so >= a.fc.la && n < a.fc.N) {
                        const long ngl = n + a.fc.row0;
                        const int b = (int)(ngl / a.fc.F), f = (int)(ngl % a.fc.F);
                        (fcc ? a.fc.crm_i : a.fc.crm_r)[((long)b * a.fc.T + (so - a.fc.la)) * a.fc.FP + f] = acc1 + a.fc.bias[fcc];
                    }
                }
            }
            k.seen0 = peek(k.fl0);  // for the next step: layer 0 is ahead, this usually shows s + 2 already
            stamp(k.cluster, s, 4);
            if (s > 0 && s < Tp)
                kloop(acc, nullptr, k.xrsrc1, slot1(s - 1), a.o_whh1, GKC, GKC, k.xrsrc1, 0, 0, 0, 0);
            stamp(k.cluster, s, 5);
            if (s < Tp) {
                // slot s & 1 held h1_{s-2}: read by every member in step s - 1, which they have left (flag1 >= s above)
                cell(k, acc, reinterpret_cast<float*>(reinterpret_cast<char*>(k.hx1) + slot1(s)),
                     TRAIN ? a.gates1 + (size_t)s * a.Nrows * 4 * GH : nullptr, TRAIN ? a.cseq1 + (size_t)s * a.Nrows * GH : nullptr);
                stamp(k.cluster, s, 6);
                publish(k.fl1 + member, (unsigned)s + 1);
                stamp(k.cluster, s, 7);
                if (TRAIN && SAVE) save_cell(k, acc, a.gates1 + (size_t)s * a.Nrows * 4 * GH, a.cseq1 + (size_t)s * a.Nrows * GH);
            }
        };
        GrpCl ka, kb;
        init(ka, cluster_a);
        ka.seen0 = peek(ka.fl0);
        if (NCL > 1 && cluster_b >= 0) {
            init(kb, cluster_b);
            kb.seen0 = peek(kb.fl0);
        }
        for (int s = 0; s <= (TRAIN ? Tp - 1 : Tp); ++s) {
            iter1(s, ka);
            if (NCL > 1 && cluster_b >= 0) iter1(s, kb);
        }
    }
}

template <int ABL, bool TRAIN = false, int NCL = 1, bool SAVE = TRAIN, int AR = FSN_ARITH_F32>
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_num_vgpr(FSN_GRP_VGPR))) void lstm2_group_kernel(const GrpArgs a) {
    // weight fragments of one K chunk, shared by the four waves: two stages x 12 fragments x 1 KB
    __shared__ typename FsnWFrag<AR>::type bsh[2][GU * 4 * grp_cps<AR>()][64];
    __shared__ float bias_sh[GU * 4][16];
    // The first half of the grid runs layer 0, the second half layer 1: blocks are handed out in order, one per CU
    // before any CU gets its second, so that every CU ends up with one workgroup of each layer (speed only).  Within a
    // half: observed, block b runs on XCD b % 8; when the cluster count allows it the eight members of a cluster are
    // blocks with the same b % 8, i.e. share one L2 (speed only as well).
    // The grid holds `slots` = min(clusters, CUs / 8) workgroup sets; slot s serves cluster s and, when there are more
    // clusters than slots, cluster s + slots as well (alternately, see GrpCl).
    const int half = gridDim.x >> 1;
    const int layer = (int)blockIdx.x >= half ? 1 : 0;
    const int bid = (int)blockIdx.x - layer * half;
    const int slots = half / GM;
    int slot, member;
    if ((ABL & 512) && slots % 8 == 0) {
        member = bid & 7;
        slot = bid >> 3;
    } else if (slots % 8 == 0) {
        const int xcd = bid & 7, j = bid >> 3;  // j-th block of that XCD
        slot = xcd * (slots / 8) + j / GM;
        member = j % GM;
    } else {
        slot = bid / GM;
        member = bid % GM;
    }
    const int cluster = slot, cluster_b = slot + slots < a.nclusters ? slot + slots : -1;
    // Layer 1 is the longer dependent chain (K = 768 per step against 416) and layer 0 is throttled to stay within
    // GD0 - 2 steps of it: layer 1's waves issue first, layer 0's fill the gaps.
    if ((ABL & 1024) && layer == 0) __builtin_amdgcn_s_setprio(2);
    else if ((ABL & 2048) && layer == 1) __builtin_amdgcn_s_setprio(1);
    else if (!(ABL & (1024 | 2048)) && layer == 1 && !(ABL & 64)) __builtin_amdgcn_s_setprio(2);
    if (layer == 0) group_body<0, ABL, TRAIN, NCL, SAVE, AR>(a, cluster, cluster_b, member, bsh, bias_sh);
    else group_body<1, ABL, TRAIN, NCL, SAVE, AR>(a, cluster, cluster_b, member, bsh, bias_sh);
}

// Several independent two-layer stacks (weight sets) in one launch, GX form: workgroup set `slot` serves cluster `slot`
// of the launch, which belongs to the set whose cluster range contains it.
template <int ABL>
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_num_vgpr(FSN_GRP_VGPR))) void lstm2_group_multi_kernel(const GrpArgs a0,
                                                                                                          const GrpSets sets) {
    __shared__ typename FsnWFrag<FSN_ARITH_F32>::type bsh[2][GU * 4 * FSN_GRP_CPS][64];
    __shared__ float bias_sh[GU * 4][16];
    const int half = gridDim.x >> 1;
    const int layer = (int)blockIdx.x >= half ? 1 : 0;
    const int bid = (int)blockIdx.x - layer * half;
    const int slots = half / GM;
    int slot, member;
    if (slots % 8 == 0) {
        const int xcd = bid & 7, j = bid >> 3;
        slot = xcd * (slots / 8) + j / GM;
        member = j % GM;
    } else {
        slot = bid / GM;
        member = bid % GM;
    }
    int si = 0;
    for (int i = 1; i < sets.n; ++i)
        if (slot >= sets.s[i].cluster0) si = i;
    const GrpSet& q = sets.s[si];
    GrpArgs a = a0;
    a.gx = q.gx;
    a.gx_tiles = q.N / 16;
    a.hx0 = q.hseq0;
    a.hx1 = q.hseq1;
    a.bias1 = q.bias1;
    a.o_whh0 = q.o_whh0;
    a.o_wih1 = q.o_wih1;
    a.o_whh1 = q.o_whh1;
    a.Nrows = q.N;
    a.flags = a0.flags + (size_t)q.cluster0 * 2 * GFS;
    const int cluster = slot - q.cluster0;  // within the set
    if (layer == 1 && !(ABL & 64)) __builtin_amdgcn_s_setprio(2);
    if (layer == 0) group_body<0, ABL, true, 1, false, FSN_ARITH_F32, true>(a, cluster, -1, member, bsh, bias_sh);
    else group_body<1, ABL, true, 1, false, FSN_ARITH_F32, true>(a, cluster, -1, member, bsh, bias_sh);
}

}  // namespace

size_t fsn_lstm2_group_exchange_floats(int clusters) { return (size_t)clusters * (GD0 + 2) * GROWS * GH; }
size_t fsn_lstm2_group_flag_words(int clusters) { return (size_t)clusters * 2 * GFS + 16; }
size_t fsn_lstm2_group_status_word(int clusters) { return (size_t)clusters * 2 * GFS; }

// Clusters of 64 rows that run on the group kernel for `tiles` 16-row tiles: at most one cluster per eight CUs (its 16
// workgroups, two per CU, must all be resident at once); what is left runs step by step beside it.
static int grp_slots_cap() {
    int cus = 0, dev = 0;
    if (!fsn_persistent_allowed() || hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return 0;
    // residency contract: two workgroups per CU of EVERY form of the kernel (compiled occupancy, not an assumption)
    const unsigned grid = 2u * (unsigned)cus;
    const void* forms[] = {(const void*)lstm2_group_kernel<0, false, 1>,      (const void*)lstm2_group_kernel<0, false, 2>,
                           (const void*)lstm2_group_kernel<0, true, 1, true>, (const void*)lstm2_group_kernel<0, true, 2, true>,
                           (const void*)lstm2_group_kernel<0, true, 1, false>, (const void*)lstm2_group_kernel<0, true, 2, false>,
                           (const void*)lstm2_group_kernel<0, true, 1, true, FSN_ARITH_F16>,
                           (const void*)lstm2_group_kernel<0, true, 2, true, FSN_ARITH_F16>,
                           (const void*)lstm2_group_kernel<0, true, 1, true, FSN_ARITH_BF16>,
                           (const void*)lstm2_group_kernel<0, true, 2, true, FSN_ARITH_BF16>,
                           (const void*)lstm2_group_multi_kernel<0>};
    for (const void* k : forms)
        if (!fsn_grid_fits(k, 256, grid)) return 0;
    return cus / GM;
}
// One cluster per workgroup set (one set per eight CUs: its 16 workgroups, two per CU, must all be resident), or two
// when (nearly) every set gets two: a launch takes the time of its fullest set - measured 22.3 ms for 64 clusters
// against 11.7 ms for 32, so 33 - 55 clusters are better off with one launch of 32 and the rest elsewhere.
int fsn_lstm2_group_clusters(int tiles) {
    const int cap = grp_slots_cap();
    const int c = tiles / 4;
    if (cap == 0) return 0;
    if (c >= 2 * cap - cap / 4) return c < 2 * cap ? c : 2 * cap;
    return c < cap ? c : cap;
}

int fsn_launch_lstm2_group(const FsnSbInput* xin, const float* whh0_p, const float* wih1_p, const float* whh1_p,
                           const float* bias1, float* exchange, unsigned* flags, const FsnRecFc* fc, int Tp, int clusters,
                           int H, hipStream_t s) {
    if (H != GH || !xin || xin->x_rows || xin->kin_chunks != 2 || !fc || !fc->w_p || clusters < 1) {
        fsn_set_error("lstm2_group: built for the sub-band model (H = 384, 32 input columns, fused output layer)");
        return FSN_ERR_ARG;
    }
    // flags and status: zero before EVERY launch (a kernel, not hipMemsetAsync: see fsn_launch_zero_words)
    if (fsn_launch_zero_words(flags, fsn_lstm2_group_flag_words(clusters), s) != FSN_OK) return FSN_ERR_LAUNCH;
    // the four matrices sit in one packed blob: address them as offsets from the lowest pointer
    const float* lo = xin->wih_p;
    for (const float* q : {whh0_p, wih1_p, whh1_p}) lo = q < lo ? q : lo;
    for (const float* q : {xin->wih_p, whh0_p, wih1_p, whh1_p})
        if (q - lo > 0x1fffffffL) {
            fsn_set_error("lstm2_group: the packed weight matrices must share one buffer");
            return FSN_ERR_ARG;
        }
    GrpArgs a{};
    a.xin = *xin;
    a.wbase = lo;
    a.o_wih0 = (unsigned)(xin->wih_p - lo);
    a.o_whh0 = (unsigned)(whh0_p - lo);
    a.o_wih1 = (unsigned)(wih1_p - lo);
    a.o_whh1 = (unsigned)(whh1_p - lo);
    a.bias1 = bias1;
    a.hx0 = exchange;
    a.hx1 = exchange + (size_t)clusters * GD0 * GROWS * GH;
    a.flags = flags;
    a.status = flags + (size_t)clusters * 2 * GFS;
    a.spin_ticks = fsn_spin_ticks();
    a.fc = *fc;
    a.Tp = Tp;
    a.nclusters = clusters;
    const int cap = grp_slots_cap(), slots = clusters < cap ? clusters : cap;
    if (cap == 0 || clusters > 2 * cap) {
        fsn_set_error("lstm2_group: %d clusters cannot be co-resident on this device (persistent kernels off, or occupancy)", clusters);
        return FSN_ERR_ARG;
    }
    if (clusters > slots) FSN_PERSIST_LAUNCH((lstm2_group_kernel<0, false, 2>), dim3((unsigned)slots * GM * 2), dim3(256), s, a);
    else FSN_PERSIST_LAUNCH((lstm2_group_kernel<0, false, 1>), dim3((unsigned)slots * GM * 2), dim3(256), s, a);
    return fsn_check_launch("lstm2_group_kernel");
}

// General two-layer form: rows [0, 64 clusters) of x [Tp][Nrows][x_ld] (16 or 32 zero-padded input columns: x_cols;
// wih0_p packed 32 columns wide either way) through both layers; hseq0 / hseq1 [Tp][Nrows][H].  Training: save0 / save1
// non-NULL = gates [Tp][Nrows][4H] followed by the cell sequence [Tp][Nrows][H] (fsn_lstm_layer_forward's layouts).
// bias0 / bias1 = b_ih + b_hh.
int fsn_launch_lstm2_group_train(const float* x, long x_ld, int x_cols, int Nrows, const float* wih0_p, const float* whh0_p,
                                 const float* wih1_p, const float* whh1_p, const float* bias0, const float* bias1,
                                 float* hseq0, float* hseq1, float* save0, float* save1, unsigned* flags, int Tp,
                                 int clusters, int H, hipStream_t s, int arith, const void* w16) {
    if (H != GH || clusters < 1 || (long)clusters * GROWS > Nrows || (size_t)Tp * Nrows * GH * 4 > 0x7fffffffull) {
        fsn_set_error("lstm2_group (training): H = 384, clusters * 64 <= rows, hidden sequence below 2 GB");
        return FSN_ERR_ARG;
    }
    if (arith != FSN_ARITH_F32 && !((arith == FSN_ARITH_F16 || arith == FSN_ARITH_BF16) && save0 && save1 && w16)) {
        fsn_set_error("lstm2_group: arithmetic %d is built for the training form (fp16 / bf16 operands, with the 16-bit copy "
                      "of the packed weights) only", arith);
        return FSN_ERR_ARG;
    }
    if (fsn_launch_zero_words(flags, fsn_lstm2_group_flag_words(clusters), s) != FSN_OK) return FSN_ERR_LAUNCH;
    const float* lo = wih0_p;
    for (const float* q : {whh0_p, wih1_p, whh1_p}) lo = q < lo ? q : lo;
    for (const float* q : {wih0_p, whh0_p, wih1_p, whh1_p})
        if (q - lo > 0x1fffffffL) {
            fsn_set_error("lstm2_group: the packed weight matrices must share one buffer");
            return FSN_ERR_ARG;
        }
    GrpArgs a{};
    a.xin.x_rows = x;
    a.xin.x_ld = x_ld;
    a.xin.x_step = Nrows;
    a.xin.N = clusters * GROWS;
    a.xin.F = 1;
    a.xin.kin_chunks = x_cols > 16 ? 2 : 1;
    a.xin.bias = bias0;
    // 16-bit arithmetic: the kernel fetches its weight fragments from w16, the 16-bit mirror (element for element) of
    // the packed buffer that starts at the lowest of the four matrices
    a.wbase = arith == FSN_ARITH_F32 ? lo : static_cast<const float*>(w16);
    a.o_wih0 = (unsigned)(wih0_p - lo);
    a.o_whh0 = (unsigned)(whh0_p - lo);
    a.o_wih1 = (unsigned)(wih1_p - lo);
    a.o_whh1 = (unsigned)(whh1_p - lo);
    a.bias1 = bias1;
    a.hx0 = hseq0;
    a.hx1 = hseq1;
    a.flags = flags;
    a.status = flags + (size_t)clusters * 2 * GFS;
    a.spin_ticks = fsn_spin_ticks();
    a.Tp = Tp;
    const bool save = save0 && save1;
    a.gates0 = save0;
    a.cseq0 = save ? save0 + (size_t)Tp * Nrows * 4 * GH : nullptr;
    a.gates1 = save1;
    a.cseq1 = save ? save1 + (size_t)Tp * Nrows * 4 * GH : nullptr;
    a.Nrows = Nrows;
    a.nclusters = clusters;
    const int cap = grp_slots_cap(), slots = clusters < cap ? clusters : cap;
    if (cap == 0 || clusters > 2 * cap) {
        fsn_set_error("lstm2_group: %d clusters cannot be co-resident on this device (persistent kernels off, or occupancy)", clusters);
        return FSN_ERR_ARG;
    }
    const dim3 grid((unsigned)slots * GM * 2), block(256);
    if (save && arith == FSN_ARITH_F16) {
        if (clusters > slots) FSN_PERSIST_LAUNCH((lstm2_group_kernel<0, true, 2, true, FSN_ARITH_F16>), grid, block, s, a);
        else FSN_PERSIST_LAUNCH((lstm2_group_kernel<0, true, 1, true, FSN_ARITH_F16>), grid, block, s, a);
    } else if (save && arith == FSN_ARITH_BF16) {
        if (clusters > slots) FSN_PERSIST_LAUNCH((lstm2_group_kernel<0, true, 2, true, FSN_ARITH_BF16>), grid, block, s, a);
        else FSN_PERSIST_LAUNCH((lstm2_group_kernel<0, true, 1, true, FSN_ARITH_BF16>), grid, block, s, a);
    } else if (save) {
        if (clusters > slots) FSN_PERSIST_LAUNCH((lstm2_group_kernel<0, true, 2, true>), grid, block, s, a);
        else FSN_PERSIST_LAUNCH((lstm2_group_kernel<0, true, 1, true>), grid, block, s, a);
    } else {
        if (clusters > slots) FSN_PERSIST_LAUNCH((lstm2_group_kernel<0, true, 2, false>), grid, block, s, a);
        else FSN_PERSIST_LAUNCH((lstm2_group_kernel<0, true, 1, false>), grid, block, s, a);
    }
    return fsn_check_launch("lstm2_group_kernel (training)");
}

// Several two-layer stacks with their own weights over the same Tp frames as ONE launch (GX form): stack i has N[i] rows
// (a multiple of 16), its layer-0 projection gx[i] as fragment tiles [Tp][N[i] / 16][4H / 16][64][4] (bias included),
// packed W_hh0 / W_ih1 / W_hh1 (all stacks' matrices inside one buffer), bias1 = b_ih + b_hh of layer 1 and its hidden
// sequences hseq0 / hseq1 [Tp][N[i]][H].  Clusters: sum of ceil(N[i] / 64) <= fsn_lstm2_group_multi_cap().
int fsn_lstm2_group_multi_cap() { return grp_slots_cap(); }
int fsn_launch_lstm2_group_multi(int n, const FsnGroupStack* st, unsigned* flags, int Tp, int H, hipStream_t s) {
    if (H != GH || n < 1 || n > GSETS || !st || !flags) {
        fsn_set_error("lstm2_group (several stacks): H = 384, 1 .. %d stacks", GSETS);
        return FSN_ERR_ARG;
    }
    GrpSets sets{};
    sets.n = n;
    const float* lo = st[0].whh0_p;
    for (int i = 0; i < n; ++i)
        for (const float* q : {st[i].whh0_p, st[i].wih1_p, st[i].whh1_p}) lo = q < lo ? q : lo;
    int clusters = 0;
    for (int i = 0; i < n; ++i) {
        const FsnGroupStack& q = st[i];
        if (q.N < 16 || q.N % 16 || (size_t)Tp * q.N * GH * 4 > 0x7fffffffull || !q.gx || !q.hseq0 || !q.hseq1 || !q.bias1) {
            fsn_set_error("lstm2_group (several stacks): stack %d: rows a multiple of 16, hidden sequence below 2 GB", i);
            return FSN_ERR_ARG;
        }
        for (const float* w : {q.whh0_p, q.wih1_p, q.whh1_p})
            if (w - lo > 0x1fffffffL) {
                fsn_set_error("lstm2_group: the packed weight matrices must share one buffer");
                return FSN_ERR_ARG;
            }
        GrpSet& g = sets.s[i];
        g.gx = q.gx;
        g.hseq0 = q.hseq0;
        g.hseq1 = q.hseq1;
        g.bias1 = q.bias1;
        g.o_whh0 = (unsigned)(q.whh0_p - lo);
        g.o_wih1 = (unsigned)(q.wih1_p - lo);
        g.o_whh1 = (unsigned)(q.whh1_p - lo);
        g.N = q.N;
        g.cluster0 = clusters;
        clusters += (q.N + GROWS - 1) / GROWS;
    }
    const int cap = grp_slots_cap();
    if (cap == 0 || clusters > cap) {
        fsn_set_error("lstm2_group: %d clusters cannot be co-resident on this device (persistent kernels off, or occupancy)", clusters);
        return FSN_ERR_ARG;
    }
    if (fsn_launch_zero_words(flags, fsn_lstm2_group_flag_words(clusters), s) != FSN_OK) return FSN_ERR_LAUNCH;
    GrpArgs a{};
    a.wbase = lo;
    a.flags = flags;
    a.status = flags + (size_t)clusters * 2 * GFS;
    a.spin_ticks = fsn_spin_ticks();
    a.Tp = Tp;
    a.nclusters = clusters;
    FSN_PERSIST_LAUNCH((lstm2_group_multi_kernel<0>), dim3((unsigned)clusters * GM * 2), dim3(256), s, a, sets);
    return fsn_check_launch("lstm2_group_multi_kernel");
}
