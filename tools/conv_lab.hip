// conv_lab.hip -- stand-alone timing harness for the fp32-MFMA convolution kernel (csrc/conv_mfma.h).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I frlw-evd_amd/csrc tools/conv_lab.hip -o build/conv_lab
//   build/conv_lab [B]          prints per-shape time / TFLOP/s and the workgroups-per-CU spread of the launch
// Developer tool: nothing imports it.  Variants are selected with -D flags of conv_mfma.h.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define FRLW_DEV_BUILD 1
namespace {
#include "conv_mfma.h"

__global__ void k_fill(float *p, long long n, uint32_t seed, float scale)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = ((int)(h & 0xFFFF) - 32768) * scale / 32768.0f;
    }
}
__global__ void k_checksum(const float *p, long long n, double *out)
{
    double s = 0;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) s += (double)p[i] * (1 + (i % 7));
    atomicAdd(out, s);
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct Shape { int H, W, Cin, Cout, k, s; };
} // namespace

int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 32;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const int nshapes = argc > 3 ? atoi(argv[3]) : 1000;
    const int prec = argc > 4 ? atoi(argv[4]) : 0; // 1: three bf16 MFMAs per product (split weight operand); prints the error against prec 0
    int done = 0;
    const Shape shapes[] = {{32, 40, 256, 256, 3, 1}, {16, 20, 256, 256, 3, 1}, {8, 10, 256, 256, 3, 1}, {16, 20, 128, 128, 3, 1},
                            {32, 40, 64, 64, 3, 1}, {16, 20, 256, 256, 1, 1}, {32, 40, 128, 128, 1, 1}, {64, 80, 64, 64, 1, 1},
                            {16, 20, 128, 128, 1, 1}, {32, 40, 64, 64, 1, 1}, {8, 10, 256, 256, 1, 1}, {8, 10, 512, 512, 1, 1},
                            {32, 40, 512, 7, 1, 1}, {16, 20, 512, 7, 1, 1}, {8, 10, 512, 7, 1, 1},
                            {16, 20, 16, 128, 1, 1}, {8, 10, 16, 64, 1, 1}, // (these two: one k-tile -- the floor of a launch)
                            {64, 80, 64, 128, 3, 1}, {64, 80, 128, 128, 1, 1}, {32, 40, 256, 128, 1, 1}}; // wide-M layers of the train step
    hipStream_t st; CK(hipStreamCreate(&st));
    float *scratch; const long long scratch_floats = 64ll << 20; CK(hipMalloc(&scratch, scratch_floats * 4));
    double *cs; CK(hipMalloc(&cs, 8));
    double tot = 0;
    for (const Shape &sh : shapes) {
        if (done++ >= nshapes) break;
        const int Ho = sh.H / sh.s, Wo = sh.W / sh.s, npad = (sh.Cout + 31) / 32 * 32, K = sh.k * sh.k * sh.Cin;
        const long long nx = (long long)B * sh.H * sh.W * sh.Cin, nw = (long long)K * npad, ny = (long long)B * Ho * Wo * sh.Cout;
        float *x, *w, *y; CK(hipMalloc(&x, nx * 4)); CK(hipMalloc(&w, nw * 4)); CK(hipMalloc(&y, ny * 4));
        k_fill<<<1024, 256, 0, st>>>(x, nx, 1u, 1.0f); k_fill<<<1024, 256, 0, st>>>(w, nw, 7u, 0.05f);
        ConvArgs c{};
        c.x = x; c.H = sh.H; c.W = sh.W; c.Cin = sh.Cin; c.x_cs = sh.Cin; c.x_co = 0; c.x_bs = (long long)sh.H * sh.W * sh.Cin;
        c.w = w; c.bias = nullptr; c.Cout = sh.Cout; c.Npad = npad; c.k = sh.k; c.stride = sh.s; c.pad = sh.k / 2;
        c.y = y; c.Ho = Ho; c.Wo = Wo; c.y_cs = sh.Cout; c.y_co = 0; c.y_bs = (long long)Ho * Wo * sh.Cout;
        c.res = nullptr; c.act = ACT_SILU; c.sig_from = 0; c.M = B * Ho * Wo; c.K = K; c.tstride = 0; c.kw = 0; c.y_rp = 0;
        float *yref = nullptr; uint4 *wsplit = nullptr;
        if (prec == 1) {
            CK(hipMalloc(&yref, ny * 4)); CK(hipMalloc(&wsplit, (long long)(K + 15) / 16 * 4 * npad * 16));
            c.y = yref; launch_conv(c, scratch, scratch_floats, st); c.y = y;
            k_conv_split_operand<<<1024, 256, 0, st>>>(w, K, npad, wsplit);
            c.w = (const float *)wsplit; c.prec = 1;
        }
        for (int i = 0; i < 3; ++i) launch_conv(c, scratch, scratch_floats, st);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) launch_conv(c, scratch, scratch_floats, st);
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        CK(hipMemsetAsync(cs, 0, 8, st)); k_checksum<<<256, 256, 0, st>>>(y, ny, cs);
        double h; CK(hipMemcpyAsync(&h, cs, 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
#ifdef CONV_LAB_PROBE
        {
            const int maxwg = 1 << 16;
            uint32_t *pr; unsigned long long *pt, *pe; CK(hipMalloc(&pr, maxwg * 4)); CK(hipMalloc(&pt, maxwg * 8)); CK(hipMalloc(&pe, maxwg * 8));
            CK(hipMemsetAsync(pr, 0xFF, maxwg * 4, st));
            CK(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_probe), &pr, sizeof(pr), 0, hipMemcpyHostToDevice, st));
            CK(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_probe_t), &pt, sizeof(pt), 0, hipMemcpyHostToDevice, st));
            CK(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_probe_e), &pe, sizeof(pe), 0, hipMemcpyHostToDevice, st));
            unsigned long long *ph; CK(hipMalloc(&ph, maxwg * 64)); CK(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_probe_ph), &ph, sizeof(ph), 0, hipMemcpyHostToDevice, st));
            launch_conv(c, scratch, scratch_floats, st);
            std::vector<unsigned long long> hph(maxwg * 8); CK(hipMemcpyAsync(hph.data(), ph, maxwg * 64, hipMemcpyDeviceToHost, st));
            std::vector<uint32_t> hp(maxwg); std::vector<unsigned long long> ht(maxwg), he(maxwg);
            CK(hipMemcpyAsync(hp.data(), pr, maxwg * 4, hipMemcpyDeviceToHost, st)); CK(hipMemcpyAsync(ht.data(), pt, maxwg * 8, hipMemcpyDeviceToHost, st)); CK(hipMemcpyAsync(he.data(), pe, maxwg * 8, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
            uint32_t *nul = nullptr; CK(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_probe), &nul, sizeof(nul), 0, hipMemcpyHostToDevice, st));
            int cnt[4096] = {0}, n = 0; unsigned long long tmin = ~0ull, tmax = 0, emax = 0, dmax = 0; double dsum = 0;
            for (int i = 0; i < maxwg; ++i) if (hp[i] != 0xFFFFFFFFu) { ++cnt[hp[i] & 4095]; ++n; if (ht[i] < tmin) tmin = ht[i]; if (ht[i] > tmax) tmax = ht[i]; if (he[i] > emax) emax = he[i]; dsum += he[i] - ht[i]; if (he[i] - ht[i] > dmax) dmax = he[i] - ht[i]; }
            int cus = 0, mn = 1 << 30, mx = 0, hist[64] = {0};
            for (int i = 0; i < 4096; ++i) if (cnt[i]) { ++cus; if (cnt[i] < mn) mn = cnt[i]; if (cnt[i] > mx) mx = cnt[i]; ++hist[cnt[i] < 63 ? cnt[i] : 63]; }
            printf("   probe: %d workgroups on %d CUs, per CU min %d max %d, start spread %.1f us, span %.1f us, workgroup mean %.1f max %.1f us; histogram:", n, cus, mn, mx, (tmax - tmin) / 100.0, (emax - tmin) / 100.0, dsum / n / 100.0, dmax / 100.0);
            for (int i = 0; i < 64; ++i) if (hist[i]) printf(" %dx%d", hist[i], i);
            printf("\n");
            { // per CU: when does its first / last workgroup finish (relative to the launch's first start)
              static unsigned long long cmin[4096], cmax[4096]; for (int i = 0; i < 4096; ++i) { cmin[i] = ~0ull; cmax[i] = 0; }
              for (int i = 0; i < maxwg; ++i) if (hp[i] != 0xFFFFFFFFu) { const int c2 = hp[i] & 4095; if (he[i] < cmin[c2]) cmin[c2] = he[i]; if (he[i] > cmax[c2]) cmax[c2] = he[i]; }
              double sfirst = 0, slast = 0, lo = 1e30, hi = 0; int nc = 0;
              for (int i = 0; i < 4096; ++i) if (cmax[i]) { ++nc; sfirst += cmin[i] - tmin; slast += cmax[i] - tmin; if (cmax[i] - tmin < lo) lo = cmax[i] - tmin; if (cmax[i] - tmin > hi) hi = cmax[i] - tmin; }
              printf("   per CU: first workgroup done at %.1f us (mean), last at %.1f us (mean; min %.1f max %.1f over CUs)\n", sfirst / nc / 100, slast / nc / 100, lo / 100, hi / 100); }
            { double a[5] = {0}; for (int i = 0; i < maxwg; ++i) if (hp[i] != 0xFFFFFFFFu) { a[0] += hph[i * 8] - ht[i]; a[1] += hph[i * 8 + 1] - hph[i * 8]; a[2] += hph[i * 8 + 2] - hph[i * 8 + 1]; a[3] += hph[i * 8 + 3] - hph[i * 8 + 2]; a[4] += he[i] - hph[i * 8 + 3]; }
              printf("   phases (mean us): index setup %.2f, first loads issued %.2f, first tile in LDS %.2f, k-loop %.2f, epilogue %.2f\n", a[0] / n / 100, a[1] / n / 100, a[2] / n / 100, a[3] / n / 100, a[4] / n / 100); }
            CK(hipFree(ph));
            CK(hipFree(pr)); CK(hipFree(pt)); CK(hipFree(pe));
        }
#endif
        if (prec == 1) {
            std::vector<float> a(ny), b(ny);
            CK(hipMemcpy(a.data(), y, ny * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), yref, ny * 4, hipMemcpyDeviceToHost));
            double md = 0, mx = 0;
            for (long long i = 0; i < ny; ++i) { md = fmax(md, fabs((double)a[i] - b[i])); mx = fmax(mx, fabs((double)b[i])); }
            printf("   prec 1 vs float32 MFMA: max |diff| %.3e of max |y| %.3e = %.2e\n", md, mx, md / mx);
            CK(hipFree(yref)); CK(hipFree(wsplit));
        }
        const double fl = 2.0 * c.M * sh.Cout * K;
        printf("%3dx%-3d %4d->%-4d k%d: %8.1f us %7.1f TFLOP/s  splits %d  checksum %.6e\n", sh.H, sh.W, sh.Cin, sh.Cout, sh.k, ms * 1e3,
               fl / ms / 1e9, c.splits, h);
        tot += ms;
        CK(hipFree(x)); CK(hipFree(w)); CK(hipFree(y));
    }
    printf("sum %.0f us\n", tot * 1e3);
    return 0;
}
