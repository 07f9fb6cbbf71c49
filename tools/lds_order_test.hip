// lds_order_test.hip -- developer experiment (not part of the product): are the lanes of ONE wave-instruction
// `ds_add_rtn_u32` that hit the same LDS address served in ascending lane order on gfx950?
//
// The stable tile partition needs, for every event of a 64-event batch, its rank among the batch's events of the
// same tile.  If the hardware serves same-address lanes in lane order, one returning LDS atomic gives that rank.
//
//   hipcc -O3 --offload-arch=gfx950 tools/lds_order_test.hip -o gpurun_out/lds_order_test && gpurun_out/lds_order_test
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// every wave: `iters` batches; lane picks address mix(seed) % n_addr; rank by atomic vs rank by counting lower lanes
__global__ __launch_bounds__(1024) void k_order(int n_addr, int iters, unsigned long long *violations,
                                                unsigned long long *conflicts)
{
    extern __shared__ uint32_t cnt[]; // [waves][n_addr]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t *mine = cnt + (size_t)wv * n_addr;
    unsigned long long bad = 0, conf = 0;
    for (int it = 0; it < iters; ++it) {
        for (int a = lane; a < n_addr; a += 64) mine[a] = 0;
        __builtin_amdgcn_wave_barrier();
        const uint32_t addr = mix((blockIdx.x * 1024u + tid) * 2654435761u + it * 40503u) % (uint32_t)n_addr;
        const uint32_t got = atomicAdd(&mine[addr], 1u);
        uint32_t want = 0;
        for (int l = 0; l < 64; ++l) {
            const uint32_t other = __shfl(addr, l);
            if (l < lane && other == addr) ++want;
        }
        if (got != want) ++bad;
        if (want) ++conf;
        __builtin_amdgcn_wave_barrier();
    }
    if (bad) atomicAdd(violations, bad);
    if (conf) atomicAdd(conflicts, conf);
}

int main()
{
    unsigned long long *d, h[2];
    hipMalloc(&d, 16);
    const int addrs[] = {1, 2, 3, 7, 16, 33, 150, 450, 2048};
    int rc = 0;
    for (int n_addr : addrs) {
        for (int threads : {64, 256, 1024}) {
            hipMemset(d, 0, 16);
            const size_t lds = (size_t)(threads / 64) * n_addr * 4;
            hipLaunchKernelGGL(k_order, dim3(512), dim3(threads), lds, 0, n_addr, 200, d, d + 1);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
            hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            printf("n_addr %5d threads %4d: conflicts %llu violations %llu\n", n_addr, threads, h[1], h[0]);
            if (h[0]) rc = 1;
        }
    }
    printf(rc ? "RESULT: NOT lane-ordered\n" : "RESULT: lane-ordered in every trial\n");
    return rc;
}
