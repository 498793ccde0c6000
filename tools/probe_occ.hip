// How many 256-thread workgroups of a kernel with V arch VGPRs are co-resident per CU on this device?  Every workgroup
// checks in and waits (bounded) until all of the grid have: only possible if the whole grid is resident at once.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int V>
__global__ __launch_bounds__(256) void occ_kernel(unsigned* counter, unsigned* ok, long long ticks) {

    if (V == 96) asm volatile("" ::: "v95");
    if (V == 128) asm volatile("" ::: "v127");
    if (V == 64) asm volatile("" ::: "v63");
    if (V == 80) asm volatile("" ::: "v79");
    if (V == 172) asm volatile("" ::: "v171");
    __shared__ int dummy[512];
    dummy[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(counter, 1u);
        const long long t0 = wall_clock64();
        bool all = false;
        while (wall_clock64() - t0 < ticks) {
            if (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= gridDim.x) { all = true; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        if (!all) atomicAdd(ok, 1u);  // counts the workgroups that gave up
    }
    __syncthreads();
    if (dummy[(threadIdx.x + 1) & 255] == -1) ok[1] = 1;
}
template <int V>
void run(const char* name) {
    unsigned *c, *ok; hipMalloc(&c, 4); hipMalloc(&ok, 8);
    int api = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, occ_kernel<V>, 256, 0);
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void*)occ_kernel<V>);
    printf("%s: numRegs %d, API occupancy %d per CU; resident workgroups per CU that rendezvous:", name, fa.numRegs, api);
    for (int k = 1; k <= 8; ++k) {
        hipMemset(c, 0, 4); hipMemset(ok, 0, 8);
        occ_kernel<V><<<256 * k, 256>>>(c, ok, 100000000LL / 10);  // 100 MHz clock: 100 ms bound
        hipDeviceSynchronize();
        unsigned gave_up = 0; hipMemcpy(&gave_up, ok, 4, hipMemcpyDeviceToHost);
        printf(" %d:%s", k, gave_up ? "no" : "yes");
        if (gave_up) break;
    }
    printf("\n");
}
int main() {
    run<64>("64 VGPRs"); run<80>("80 VGPRs"); run<96>("96 VGPRs"); run<128>("128 VGPRs"); run<172>("172 VGPRs");
    return 0;
}
