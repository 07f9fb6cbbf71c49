// lds_atomic_rate.hip -- developer experiment: throughput of LDS atomics on gfx950 (cycles per wave-instruction per CU).
//   hipcc -O3 --offload-arch=gfx950 tools/lds_atomic_rate.hip -o /tmp/lar && /tmp/lar
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int MODE> // 0: ds_add_u32 no return, 1: ds_add_rtn_u32, 2: ds_add_f32, 3: plain read+write, 4: ds_add_rtn_f32
__global__ __launch_bounds__(1024) void k(int n_addr, int iters, int same, unsigned long long *cyc, float *sink)
{
    __shared__ uint32_t mem[4096];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += blockDim.x) mem[i] = 0;
    __syncthreads();
    uint32_t h = tid * 2654435761u + 12345u;
    float facc = 0.f;
    uint32_t uacc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
        const uint32_t a = same ? (uint32_t)((tid >> 6) * 64 + (h % (uint32_t)same)) % n_addr : h % (uint32_t)n_addr;
        if (MODE == 0) atomicAdd(&mem[a], 1u);
        if (MODE == 1) uacc += atomicAdd(&mem[a], 1u);
        if (MODE == 2) atomicAdd((float *)&mem[a], 1.0f);
        if (MODE == 3) { volatile uint32_t *m = mem; m[a] = m[a] + 1u; }
        if (MODE == 4) facc += atomicAdd((float *)&mem[a], 1.0f);
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) atomicAdd(cyc, t1 - t0);
    if (sink) sink[tid] = facc + uacc + mem[tid];
}

int main()
{
    unsigned long long *d, h;
    hipMalloc(&d, 8);
    const char *names[] = {"ds_add_u32", "ds_add_rtn_u32", "ds_add_f32", "read+write", "ds_add_rtn_f32"};
    const int iters = 2000;
    for (int threads : {64, 256, 1024}) {
        for (int n_addr : {4096, 256, 16}) {
            for (int mode = 0; mode < 5; ++mode) {
                hipMemset(d, 0, 8);
                const int blocks = 256;
                switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, n_addr, iters, 0, d, nullptr); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, n_addr, iters, 0, d, nullptr); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(threads), 0, 0, n_addr, iters, 0, d, nullptr); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(threads), 0, 0, n_addr, iters, 0, d, nullptr); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(threads), 0, 0, n_addr, iters, 0, d, nullptr); break;
                }
                hipDeviceSynchronize();
                hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
                const double per_block = (double)h / blocks;
                printf("threads %4d addrs %4d %-15s: %7.1f cycles per wave-instruction per CU (%.1f per iteration of the block)\n", threads,
                       n_addr, names[mode], per_block / iters / (threads / 64), per_block / iters);
            }
        }
    }
    return 0;
}
