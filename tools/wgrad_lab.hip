// wgrad_lab.hip -- stand-alone timing harness for the fp32-MFMA weight-gradient kernel (csrc/wgrad_mfma.h).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I frlw-evd_amd/csrc tools/wgrad_lab.hip -o build/wgrad_lab
//   build/wgrad_lab [B] [reps] [first n shapes]     per-shape time / TFLOP/s and a checksum of the partial tiles' sum
// Developer tool: nothing imports it.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define FRLW_DEV_BUILD 1
namespace {
#include "conv_mfma.h"
#include "wgrad_mfma.h"

__global__ void k_fill(float *p, long long n, uint32_t seed, float scale)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = ((int)(h & 0xFFFF) - 32768) * scale / 32768.0f;
    }
}
__global__ void k_checksum(const float *p, long long per, int splits, double *out)
{
    double s = 0;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < per; i += (long long)gridDim.x * blockDim.x) {
        float v = 0; for (int z = 0; z < splits; ++z) v += p[z * per + i];
        s += (double)v * (1 + (i % 7));
    }
    atomicAdd(out, s);
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct Shape { int H, W, Cin, Cout, k, s; };
} // namespace

int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 64;
    const int reps = argc > 2 ? atoi(argv[2]) : 10;
    const int nshapes = argc > 3 ? atoi(argv[3]) : 1000;
    int done = 0;
    const Shape shapes[] = {{32, 40, 256, 256, 3, 1}, {16, 20, 256, 256, 3, 1}, {8, 10, 256, 256, 3, 1}, {16, 20, 128, 128, 3, 1},
                            {32, 40, 64, 64, 3, 1}, {64, 80, 64, 128, 3, 2}, {16, 20, 256, 256, 1, 1}, {32, 40, 128, 128, 1, 1},
                            {64, 80, 64, 64, 1, 1}, {16, 20, 128, 128, 1, 1}, {8, 10, 512, 512, 1, 1}, {128, 160, 32, 64, 3, 2}};
    hipStream_t st; CK(hipStreamCreate(&st));
    float *scratch; const long long scratch_floats = 96ll << 20; CK(hipMalloc(&scratch, scratch_floats * 4));
    double *cs; CK(hipMalloc(&cs, 8));
    double tot = 0;
    for (const Shape &sh : shapes) {
        if (done++ >= nshapes) break;
        const int Ho = sh.H / sh.s, Wo = sh.W / sh.s;
        const long long nx = (long long)B * sh.H * sh.W * sh.Cin, nz = (long long)B * Ho * Wo * sh.Cout;
        float *x, *dz; CK(hipMalloc(&x, nx * 4)); CK(hipMalloc(&dz, nz * 4));
        k_fill<<<1024, 256, 0, st>>>(x, nx, 1u, 1.0f); k_fill<<<1024, 256, 0, st>>>(dz, nz, 7u, 0.05f);
        WgradArgs a{};
        a.x = x; a.H = sh.H; a.W = sh.W; a.Cin = sh.Cin; a.x_bs = (long long)sh.H * sh.W * sh.Cin; a.x_cs = sh.Cin;
        a.dz = dz; a.Ho = Ho; a.Wo = Wo; a.Cout = sh.Cout; a.dz_bs = (long long)Ho * Wo * sh.Cout; a.dz_cs = sh.Cout;
        a.k = sh.k; a.stride = sh.s; a.pad = (sh.k - 1) / 2; a.R = sh.k * sh.k * sh.Cin; a.M = B * Ho * Wo;
        a.partial = scratch;
        a.prec = (int)dev_knob("FRLW_WGRAD_PREC", 1); // 1: three bf16 MFMAs per product
        const int target = (int)dev_knob("FRLW_WGRAD_TARGET", 1024);
        auto run = [&]() { launch_wgrad(a, target, scratch_floats, st); };
        for (int i = 0; i < 2; ++i) run();
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) run();
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        CK(hipMemsetAsync(cs, 0, 8, st)); k_checksum<<<256, 256, 0, st>>>(scratch, (long long)a.R * sh.Cout, a.splits, cs);
        double h; CK(hipMemcpyAsync(&h, cs, 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
        const double fl = 2.0 * a.M * sh.Cout * a.R;
        printf("%3dx%-3d %4d->%-4d k%d s%d: %8.1f us %7.1f TFLOP/s  splits %3d  checksum %.6e\n", sh.H, sh.W, sh.Cin, sh.Cout, sh.k, sh.s,
               ms * 1e3, fl / ms / 1e9, a.splits, h);
        tot += ms;
        CK(hipFree(x)); CK(hipFree(dz));
    }
    printf("sum %.0f us\n", tot * 1e3);
    return 0;
}
