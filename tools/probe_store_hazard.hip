// Stand-alone probe of the gfx950 store-data hazard (profiles/r06_store_hazard.md).
//
// Question: how many issue slots after a 12 / 16-byte-per-lane store may a wave write the store's DATA registers again?
// hipcc (ROCm 7.2, GCNHazardRecognizer) keeps 2 wait states between a > 64-bit VMEM store and a VALU write of its data
// registers - EXCEPT for buffer stores whose soffset operand is an SGPR, where it keeps none ("the instruction takes an
// extra cycle", a rule from the first GCN parts).  The 16-bit BPTT kernel of round 5 went wrong exactly where the compiler
// had used that exemption (buffer_store_dwordx4 v[38:41], .., s0 offen ; v_pk_add_f32 v[38:39], ..).
//
// Every wave issues, per iteration, NB stores of 1 KB (64 lanes x 16 bytes, every dword = 1.0f) to its own addresses and
// rewrites data registers of each store D issue slots behind it; a checker counts the dwords in memory that are not 1.0f.
// Swept: store kind (buffer with immediate / SGPR soffset, sc1, global with vector / scalar base, dwordx3), the rewriting
// instruction (v_mov of the last / first dword, v_pk_add_f32, DPP move, an LDS read landing in the registers), what fills
// the slots in between (s_nop, independent vector instructions, another store), burst length, workgroups, and a bandwidth
// hog on a second stream.
//
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/probe_store_hazard tools/probe_store_hazard.hip
// run:   tools/bin/probe_store_hazard [csv path] > profiles/r06_store_hazard.md
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr unsigned GOOD = 0x3f800000u;  // 1.0f
constexpr unsigned BAD = 0x40400000u;   // 3.0f (v_pk_add_f32 of 1.0 + 2.0 produces it too)

// explicit data registers v[32 .. 32 + 4 NB) of the asm blocks
#define CLOBBER_V32_127                                                                                                   \
    "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", \
        "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64",    \
        "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80",    \
        "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96",    \
        "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110",       \
        "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124",    \
        "v125", "v126", "v127"

// KIND: 0 buffer, soffset immediate   1 buffer, soffset SGPR   2 buffer, soffset SGPR, sc1   3 global, 64-bit vector address
//       4 global, 32-bit vector offset + scalar base   5 buffer_store_dwordx3, soffset SGPR   6 buffer dwordx4, soffset SGPR, sc0 sc1
//       7 / 8 buffer_store_dwordx2 with immediate / SGPR soffset (the controls: no hazard is documented for 8-byte stores)
// MODE: 0 v_mov_b32 of the LAST data dword   1 v_mov_b32 of the FIRST   2 v_pk_add_f32 of the last two   3 v_mov_b32_dpp of the
//       last   4 ds_read_b128 into all four (LDS holds 3.0f)
// FILL: 0 one s_nop D-2   1 D-1 independent v_mov_b32   2 one buffer_store_dwordx2 of other registers, then s_nop
#define PROBE_ASM(STORE_LINE)                                                                                          \
    asm volatile(                                                                                                      \
        ".set R, 32\n\t"                                                                                               \
        ".rept %[nb]\n\t"                                                                                              \
        "v_mov_b32 v[R], %[g]\n\t v_mov_b32 v[R+1], %[g]\n\t v_mov_b32 v[R+2], %[g]\n\t v_mov_b32 v[R+3], %[g]\n\t"       \
        ".set R, R+4\n\t"                                                                                              \
        ".endr\n\t"                                                                                                    \
        "s_nop 7\n\t"                                                                                                  \
        ".set R, 32\n\t"                                                                                               \
        ".rept %[nb]\n\t" STORE_LINE                                                                                   \
        "\n\t"                                                                                                         \
        ".if %[fill] == 0\n\t"                                                                                         \
        "  .if %[d] > 1\n\t s_nop %[d]-2\n\t .endif\n\t"                                                               \
        ".elseif %[fill] == 1\n\t"                                                                                     \
        "  .rept %[d]-1\n\t v_mov_b32 %[t], %[t]\n\t .endr\n\t"                                                        \
        ".else\n\t"                                                                                                    \
        "  .if %[d] > 1\n\t buffer_store_dwordx2 %[x2], %[vo], %[rs2], 0 offen\n\t .endif\n\t"                         \
        "  .if %[d] > 2\n\t s_nop %[d]-3\n\t .endif\n\t"                                                               \
        ".endif\n\t"                                                                                                   \
        ".if %[mode] == 0\n\t v_mov_b32 v[R+%[last]], %[b]\n\t"                                                        \
        ".elseif %[mode] == 1\n\t v_mov_b32 v[R], %[b]\n\t"                                                            \
        ".elseif %[mode] == 2\n\t v_pk_add_f32 v[R+%[pk]:R+%[pk]+1], v[R+%[pk]:R+%[pk]+1], %[bp]\n\t"                                      \
        ".elseif %[mode] == 3\n\t v_mov_b32_dpp v[R+%[last]], %[b] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" \
        ".else\n\t ds_read_b128 v[R:R+3], %[la]\n\t"                                                                   \
        ".endif\n\t"                                                                                                   \
        "v_add_u32 %[vo], 0x400, %[vo]\n\t"                                                                            \
        "v_lshl_add_u64 %[va], %[va], 0, %[k1024]\n\t"                                                                 \
        ".set R, R+4\n\t"                                                                                              \
        ".endr\n\t"                                                                                                    \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                     \
        : [vo] "+v"(vo), [va] "+v"(va), [t] "+v"(tmp)                                                                  \
        : [rs] "s"(rs), [rs2] "s"(rs2), [so] "s"(so), [sb] "s"(sbase), [g] "v"(GOOD), [b] "v"(BAD), [bp] "v"(bp),     \
          [x2] "v"(x2), [la] "v"(la), [k1024] "s"(k1024), [d] "n"(D), [nb] "n"(NB), [fill] "n"(FILL), [mode] "n"(MODE), \
          [last] "n"(KIND == 5 ? 2 : KIND >= 7 ? 1 : 3), [pk] "n"(KIND >= 7 ? 0 : 2)                                                                                \
        : "memory", CLOBBER_V32_127)

template <int KIND, int MODE, int D, int NB, int FILL>
__global__ __launch_bounds__(256) void probe_kernel(unsigned* out, unsigned* scratch2, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned lds[256 * 4];
    for (int i = threadIdx.x; i < 256 * 4; i += 256) lds[i] = BAD;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const size_t wave_bytes = (size_t)iters * NB * 1024;
    unsigned char* const wbase = reinterpret_cast<unsigned char*>(out) + (size_t)wave * wave_bytes;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(wbase, 0, (int)wave_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(scratch2 + (size_t)wave * 128, 0, 512, 0x00020000);
    const f32x2 bp = {2.0f, 2.0f}, x2 = {5.0f, 6.0f};
    const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)lds + (threadIdx.x & 255) * 16;
    const unsigned long long k1024 = 1024;
    unsigned tmp = lane;
    for (int it = 0; it < iters; ++it) {
        const unsigned itoff = __builtin_amdgcn_readfirstlane((unsigned)it * NB * 1024u);
        // buffer kinds with an SGPR soffset carry the iteration offset there; everything else in the vector offset / address
        unsigned so = (KIND == 1 || KIND == 2 || KIND == 5 || KIND == 6 || KIND == 8) ? itoff : 0u;
        unsigned vo = lane * 16u + ((KIND == 1 || KIND == 2 || KIND == 5 || KIND == 6 || KIND == 8) ? 0u : itoff);
        unsigned long long va = (unsigned long long)(size_t)wbase + itoff + lane * 16u;
        const unsigned long long sbase = (unsigned long long)(size_t)wbase;
        if constexpr (KIND == 0) PROBE_ASM("buffer_store_dwordx4 v[R:R+3], %[vo], %[rs], 0 offen");
        else if constexpr (KIND == 1) PROBE_ASM("buffer_store_dwordx4 v[R:R+3], %[vo], %[rs], %[so] offen");
        else if constexpr (KIND == 2) PROBE_ASM("buffer_store_dwordx4 v[R:R+3], %[vo], %[rs], %[so] offen sc1");
        else if constexpr (KIND == 3) PROBE_ASM("global_store_dwordx4 %[va], v[R:R+3], off");
        else if constexpr (KIND == 4) PROBE_ASM("global_store_dwordx4 %[vo], v[R:R+3], %[sb]");
        else if constexpr (KIND == 5) PROBE_ASM("buffer_store_dwordx3 v[R:R+2], %[vo], %[rs], %[so] offen");
        else if constexpr (KIND == 6) PROBE_ASM("buffer_store_dwordx4 v[R:R+3], %[vo], %[rs], %[so] offen sc0 sc1");
        else if constexpr (KIND == 7) PROBE_ASM("buffer_store_dwordx2 v[R:R+1], %[vo], %[rs], 0 offen");
        else PROBE_ASM("buffer_store_dwordx2 v[R:R+1], %[vo], %[rs], %[so] offen");
    }
    if (tmp == 0xffffffffu) out[0] = tmp;
}

// counts[dword 0..3][lane quad of 16: 0..3] of wrong dwords, counts[16] = dwords never written (zero)
__global__ void check_kernel(const unsigned* __restrict__ out, size_t ndw, int ndata, unsigned long long* counts) {
    __shared__ unsigned long long sh[17];
    if (threadIdx.x < 17) sh[threadIdx.x] = 0;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ndw; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned v = out[i];
        const int dw = (int)(i & 3), lane = (int)((i >> 2) & 63);
        if (dw >= ndata) continue;  // dwordx3 / dwordx2 stores leave the rest of a lane's 16 bytes untouched
        if (v == 0u) atomicAdd(&sh[16], 1ull);
        else if (v != GOOD) atomicAdd(&sh[dw * 4 + ((lane & 15) >> 2)], 1ull);
    }
    __syncthreads();
    if (threadIdx.x < 17 && sh[threadIdx.x]) atomicAdd(&counts[threadIdx.x], sh[threadIdx.x]);
}

__global__ void hog_kernel(const float4* __restrict__ a, float4* __restrict__ b, size_t n, int reps) {
    for (int r = 0; r < reps; ++r)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

struct Result {
    int kind, mode, d, nb, fill, wgs, hog;
    unsigned long long bad[16], unwritten, total;
};

static unsigned* g_out;
static unsigned* g_s2;
static unsigned long long* g_counts;
static float4 *g_ha, *g_hb;
static hipStream_t g_sa, g_sb;
static std::vector<Result> g_results;
constexpr int ITERS = 16;
constexpr size_t HOG_N = (size_t)1 << 26;  // 1 GiB per buffer

template <int KIND, int MODE, int D, int NB, int FILL>
static void run_one(int wgs, int hog) {
    const size_t bytes = (size_t)wgs * 4 * ITERS * NB * 1024;
    CK(hipMemsetAsync(g_out, 0, bytes, g_sa));
    CK(hipMemsetAsync(g_counts, 0, 17 * sizeof(unsigned long long), g_sa));
    CK(hipStreamSynchronize(g_sa));
    if (hog) hog_kernel<<<512, 256, 0, g_sb>>>(g_ha, g_hb, HOG_N, 2);
    probe_kernel<KIND, MODE, D, NB, FILL><<<wgs, 256, 0, g_sa>>>(g_out, g_s2, ITERS);
    CK(hipGetLastError());
    check_kernel<<<1024, 256, 0, g_sa>>>(g_out, bytes / 4, KIND == 5 ? 3 : KIND >= 7 ? 2 : 4, g_counts);
    CK(hipStreamSynchronize(g_sa));
    CK(hipStreamSynchronize(g_sb));
    unsigned long long c[17];
    CK(hipMemcpy(c, g_counts, sizeof(c), hipMemcpyDeviceToHost));
    Result r{KIND, MODE, D, NB, FILL, wgs, hog, {}, c[16], (unsigned long long)wgs * 4 * ITERS * NB};
    for (int i = 0; i < 16; ++i) r.bad[i] = c[i];
    g_results.push_back(r);
}

template <int KIND, int MODE, int D, int NB, int FILL>
static void run_cfg() {
    for (int wgs : {256, 1024})
        for (int hog : {0, 1}) run_one<KIND, MODE, D, NB, FILL>(wgs, hog);
}
template <int KIND, int MODE, int D>
static void run_d() {
    run_cfg<KIND, MODE, D, 1, 0>();
    run_cfg<KIND, MODE, D, 24, 0>();
    run_cfg<KIND, MODE, D, 24, 1>();
    if constexpr (D >= 2) run_cfg<KIND, MODE, D, 24, 2>();
}
template <int KIND, int MODE>
static void run_mode() {
    run_d<KIND, MODE, 1>();
    run_d<KIND, MODE, 2>();
    run_d<KIND, MODE, 3>();
    run_d<KIND, MODE, 4>();
    run_d<KIND, MODE, 5>();
    run_d<KIND, MODE, 6>();
    run_d<KIND, MODE, 8>();
}
template <int KIND>
static void run_kind() {
    run_mode<KIND, 0>();
    run_mode<KIND, 1>();
    run_mode<KIND, 2>();
    run_mode<KIND, 3>();
    run_mode<KIND, 4>();
}

int main(int argc, char** argv) {
    CK(hipSetDevice(0));
    CK(hipStreamCreate(&g_sa));
    CK(hipStreamCreate(&g_sb));
    CK(hipMalloc(&g_out, (size_t)1024 * 4 * ITERS * 24 * 1024));
    CK(hipMalloc(&g_s2, (size_t)1024 * 4 * 512));
    CK(hipMalloc(&g_counts, 17 * sizeof(unsigned long long)));
    CK(hipMalloc(&g_ha, HOG_N * sizeof(float4)));
    CK(hipMalloc(&g_hb, HOG_N * sizeof(float4)));
    CK(hipMemset(g_ha, 1, HOG_N * sizeof(float4)));
    run_kind<0>();
    run_kind<1>();
    run_kind<2>();
    run_kind<3>();
    run_kind<4>();
    run_kind<5>();
    run_kind<6>();
    run_kind<7>();
    run_kind<8>();

    const char* kinds[] = {"buffer_store_dwordx4, soffset immediate", "buffer_store_dwordx4, soffset SGPR",
                           "buffer_store_dwordx4, soffset SGPR, sc1", "global_store_dwordx4, vector address",
                           "global_store_dwordx4, vector offset + scalar base", "buffer_store_dwordx3, soffset SGPR",
                           "buffer_store_dwordx4, soffset SGPR, sc0 sc1", "buffer_store_dwordx2, soffset immediate (control)",
                           "buffer_store_dwordx2, soffset SGPR (control)"};
    const char* modes[] = {"v_mov_b32 last dword", "v_mov_b32 first dword", "v_pk_add_f32 last two dwords", "v_mov_b32_dpp last dword",
                           "ds_read_b128 into all four"};
    const char* fills[] = {"s_nop", "v_mov x (D-1)", "store_dwordx2 + s_nop"};
    const int ds[] = {1, 2, 3, 4, 5, 6, 8};
    if (argc > 1) {
        FILE* f = fopen(argv[1], "w");
        fprintf(f, "kind,mode,fill,burst,workgroups,hog,distance,stores,unwritten_dwords,wrong_dwords,wrong_by_dword0,dword1,dword2,dword3,"
                   "wrong_by_lane_quad0,quad1,quad2,quad3\n");
        for (const Result& r : g_results) {
            unsigned long long all = 0, bydw[4] = {}, byq[4] = {};
            for (int i = 0; i < 16; ++i) all += r.bad[i], bydw[i / 4] += r.bad[i], byq[i % 4] += r.bad[i];
            fprintf(f, "%d,%d,%d,%d,%d,%d,%d,%llu,%llu,%llu,%llu,%llu,%llu,%llu,%llu,%llu,%llu,%llu\n", r.kind, r.mode, r.fill, r.nb, r.wgs,
                    r.hog, r.d, r.total, r.unwritten, all, bydw[0], bydw[1], bydw[2], bydw[3], byq[0], byq[1], byq[2], byq[3]);
        }
        fclose(f);
    }
    printf("# gfx950 store-data hazard probe (tools/probe_store_hazard.hip)\n\n");
    printf("Entries: fraction of the rewritten data dwords that reached memory with the NEW value, i.e. wrong dwords / (stores x 64\n"
           "lanes x rewritten data dwords per lane; a store is 64 lanes x 16 bytes, every dword 1.0f), worst case over\n"
           "{256, 1024 workgroups} x {alone, beside a 2 x 1 GiB copy on a second stream}; `-` = no wrong dword in any of them.\n"
           "D = issue slots between the store and the instruction that writes its data registers again (D = 1: the very next).\n\n");
    unsigned long long unwritten = 0;
    for (int k = 0; k < 9; ++k) {
        printf("## %s\n\n| rewritten by | burst | slots filled with |", kinds[k]);
        for (int d : ds) printf(" D=%d |", d);
        printf("\n|---|---|---|");
        for (size_t i = 0; i < sizeof(ds) / sizeof(ds[0]); ++i) printf("---|");
        printf("\n");
        for (int m = 0; m < 5; ++m)
            for (int nb : {1, 24})
                for (int fl = 0; fl < 3; ++fl) {
                    if (nb == 1 && fl) continue;
                    printf("| %s | %d | %s |", modes[m], nb, fills[fl]);
                    for (int d : ds) {
                        double worst = -1;
                        for (const Result& r : g_results)
                            if (r.kind == k && r.mode == m && r.nb == nb && r.fill == fl && r.d == d) {
                                unsigned long long all = 0;
                                for (int i = 0; i < 16; ++i) all += r.bad[i];
                                unwritten += r.unwritten;
                                const double per = m == 2 ? (k == 5 ? 1 : 2) : m == 4 ? (k == 5 ? 3 : k >= 7 ? 2 : 4) : 1;
                                const double f = (double)all / ((double)r.total * 64 * per);
                                if (f > worst) worst = f;
                            }
                        if (worst < 0) printf(" n/a |");
                        else if (worst == 0) printf(" - |");
                        else printf(" %.3g |", worst);
                    }
                    printf("\n");
                }
        printf("\n");
    }
    // which lanes / dwords go wrong where anything does
    unsigned long long bydw[4] = {}, byq[4] = {};
    for (const Result& r : g_results)
        if (r.mode == 4)
            for (int i = 0; i < 16; ++i) bydw[i / 4] += r.bad[i], byq[i % 4] += r.bad[i];
    printf("ds_read_b128 rewrites, wrong dwords by data dword 0..3: %llu %llu %llu %llu; by lane quad (lane %% 16) / 4 = 0..3: %llu %llu %llu %llu\n\n",
           bydw[0], bydw[1], bydw[2], bydw[3], byq[0], byq[1], byq[2], byq[3]);
    for (int m = 0; m < 4; ++m) {
        unsigned long long q[4] = {};
        for (const Result& r : g_results)
            if (r.mode == m)
                for (int i = 0; i < 16; ++i) q[i % 4] += r.bad[i];
        printf("%s: wrong dwords by lane quad 0..3: %llu %llu %llu %llu\n", modes[m], q[0], q[1], q[2], q[3]);
    }
    printf("\ndwords never written (must be 0: the probe's own sanity): %llu; configurations run: %zu\n", unwritten, g_results.size());
    return 0;
}
