// train_ops.hip -- training-mode operators of BaseConv = Conv2d(bias=False) + BatchNorm2d (batch statistics) + SiLU
// (core/yolox/models/network_blocks.py:33-65) for the train step of core/exp.py:283-315, on NHWC float32 tensors
// (torch channels_last storage, so the autograd wrapper in frlw-evd_amd/yolox/train_ops.py passes pointers as is):
//
//   frlw_conv_weight_layouts   torch (Cout, Cin, k, k) -> the two GEMM operands: forward (k*k*Cin, Npad(Cout)) and
//                              data-gradient (k*k*Cout, Npad(Cin)) with the taps flipped
//   frlw_conv2d_fwd            z = conv(x, w)                    k_conv_mfma (conv_mfma.h), fp32 MFMA
//   frlw_conv2d_dgrad          dx = conv^T(dz, w)                the same kernel on the flipped operand; stride 2 through
//                              the transposed gather (tstride = 2)
//   frlw_conv2d_wgrad          dw = x^T * dz                     k_wgrad_mfma: M = k*k*Cin, N = Cout, contraction over
//                              the B*Ho*Wo output pixels split over blockIdx.z, deterministic two-stage reduction
//                              straight into torch's (Cout, Cin, k, k) layout
//   frlw_bn_stats              per-channel mean / biased variance (float64 accumulation, two-stage, deterministic)
//   frlw_bn_silu_fwd           y = silu(gamma * (z - mean) * invstd + beta)
//   frlw_bn_silu_bwd           dz, dgamma, dbeta from dy and z (u recomputed, nothing but z saved)
//
// MI355X: every contraction runs on the matrix cores, like the inference forward, in the arithmetic the caller names
// (`precision`: 0 = v_mfma_f32_32x32x2_f32, 1 = float32 products from three bf16 MFMAs on split operands: conv_mfma.h);
// the elementwise passes are HBM-bound float4 streams over the (M, C) view of the tensor.

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "frlw_evd.h"

namespace {

#include "conv_mfma.h"

#define TRY_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
        FRLW_DEV_LOG("frlw_evd train_ops: %s -> %s\n", #expr, hipGetErrorString(e_)); \
        return FRLW_ERR_HIP; } } while (0)

// ---- weight layouts -----------------------------------------------------------------------------
// fwd[(ky*k + kx)*Cin + ci][co] = w[co][ci][ky][kx];  dgr[((k-1-ky)*k + (k-1-kx))*Cout + co][ci] = w[co][ci][ky][kx]
// One thread per element of the padded forward operand (rows (ky, kx, ci), np_f columns) and of the padded
// data-gradient operand: every element, padding included, is written, so no memset is needed.
// `parity` != 0 (stride-2, k = 3): the data-gradient operand is grouped by output parity class (py, px) in the order
// (0,0) (0,1) (1,0) (1,1) with 1, 2, 2, 4 taps -- row blocks starting at 0, 1, 3, 5 times Cout; inside a class the rows
// are (t_ky, t_kx, co) where tap t reads dz[o' + t] and stands for kernel index 1 (parity 0) or 2, 0 (parity 1, t = 0, 1).
// element (row, col) of the forward (which = 0) or data-gradient (which = 1) operand; zero in the padding (rows past K too)
// (WPair: two BaseConvs stacked along the output channels -- channels >= split come from the second block's weight)
struct WPair { const float *w, *w2; int split; };
__device__ __forceinline__ const float *wpair_of(const WPair &p, int co, int Cin, int kk)
{
    return (p.split > 0 && co >= p.split) ? p.w2 - (long long)p.split * Cin * kk : p.w; // indexed with the stacked co below
}
__device__ __forceinline__ float weight_operand_value(const WPair wp, int Cout, int Cin, int k, int which, int parity, long long row, int col)
{
    const int kk = k * k;
    if (which == 0) {
        const int ci = (int)(row % Cin), tap = (int)(row / Cin);
        return (col < Cout && tap < kk) ? wpair_of(wp, col, Cin, kk)[((long long)col * Cin + ci) * kk + tap] : 0.0f;
    }
    const int co = (int)(row % Cout), tapf = (int)(row / Cout); // tapf = flipped tap index
    if (tapf >= kk || col >= Cin) return 0.0f;
    const float *w = wpair_of(wp, co, Cin, kk);
    int tap = kk - 1 - tapf;
    if (parity) {
        const int cls = tapf < 1 ? 0 : (tapf < 3 ? 1 : (tapf < 5 ? 2 : 3));
        const int tl = tapf - (cls == 0 ? 0 : (cls == 1 ? 1 : (cls == 2 ? 3 : 5)));
        const int py = cls >> 1, px = cls & 1, kwc = px ? 2 : 1;
        const int t_ky = tl / kwc, t_kx = tl - t_ky * kwc;
        const int ky = py ? (t_ky == 0 ? 2 : 0) : 1, kx = px ? (t_kx == 0 ? 2 : 0) : 1;
        tap = ky * 3 + kx;
    }
    return w[((long long)co * Cin + col) * kk + tap];
}

__host__ __device__ inline long long operand_rows(long long K, int prec) { return prec == 1 ? (K + 15) / 16 * 16 : K; }

// float32 operands: one thread per element i of [forward | data gradient]
__device__ __forceinline__ void weight_layout_elem(const WPair w, int Cout, int Cin, int k, float *fwd, int np_f, float *dgr, int np_d,
                                                   int parity, long long nf, long long i)
{
    if (i < nf) fwd[i] = weight_operand_value(w, Cout, Cin, k, 0, 0, i / np_f, (int)(i % np_f));
    else dgr[i - nf] = weight_operand_value(w, Cout, Cin, k, 1, parity, (i - nf) / np_d, (int)((i - nf) % np_d));
}

// split operands (conv_mfma.h, prec = 1): one thread per PAIR of 16-byte records (hi and lo of the same eight values) =
// float-equivalents i .. i + 7 of [forward | data gradient], each operand ceil16(K) rows; i % 8 == 0
__device__ __forceinline__ void weight_layout_record(const WPair w, int Cout, int Cin, int k, float *fwd, int np_f, float *dgr, int np_d,
                                                     int parity, long long nf, long long i)
{
    const int which = i < nf ? 0 : 1;
    const long long r2 = (which ? i - nf : i) >> 3;
    const int np = which ? np_d : np_f;
    const int n = (int)(r2 % np), h = (int)((r2 / np) & 1);
    const long long kt = r2 / (2ll * np);
    uint32_t hi[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float x[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) x[q] = weight_operand_value(w, Cout, Cin, k, which, parity, kt * 16 + conv_split_kmem(h, 2 * e + q), n);
        hi[e] = conv_bf16_pair(x[0], x[1]);
        lo[e] = conv_bf16_pair(x[0] - __builtin_bit_cast(float, hi[e] << 16), x[1] - __builtin_bit_cast(float, hi[e] & 0xFFFF0000u));
    }
    uint4 *out = (uint4 *)(which ? dgr : fwd) + ((kt * 2 + h) * 2) * np + n;
    out[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    out[np] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
}

__global__ void k_weight_layouts(const WPair w, int Cout, int Cin, int k, float *fwd, int np_f, float *dgr, int np_d, int parity, int prec)
{
    const int kk = k * k;
    const long long nf = fwd ? operand_rows((long long)kk * Cin, prec) * np_f : 0, nd = dgr ? operand_rows((long long)kk * Cout, prec) * np_d : 0;
    if (prec == 1) {
        for (long long i = 8 * (blockIdx.x * (long long)blockDim.x + threadIdx.x); i < nf + nd; i += 8ll * gridDim.x * blockDim.x)
            weight_layout_record(w, Cout, Cin, k, fwd, np_f, dgr, np_d, parity, nf, i);
        return;
    }
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nf + nd; i += (long long)gridDim.x * blockDim.x)
        weight_layout_elem(w, Cout, Cin, k, fwd, np_f, dgr, np_d, parity, nf, i);
}

// every weight of a model in one launch: element i belongs to the entry with the largest `first` <= i.  A workgroup walks
// tiles of 2048 consecutive elements and looks the entry of a tile's first element up ONCE (wave-uniform: scalar loads);
// only the elements of a tile that straddles two entries search again.  (Entries with split operands: 512 records per tile.)
__global__ __launch_bounds__(256) void k_weight_layouts_batch(const frlw_weight_layout_item_t *items, int n, long long total)
{
    constexpr int kTile = 2048;
    for (long long base = (long long)blockIdx.x * kTile; base < total; base += (long long)gridDim.x * kTile) {
        int lo = 0, hi = n;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (items[mid].first <= base) lo = mid; else hi = mid;
        }
        const frlw_weight_layout_item_t cur = items[lo];
        const long long next = lo + 1 < n ? items[lo + 1].first : total;
        if (cur.precision == 1 && base + kTile <= next) { // a whole tile of one split entry: one record pair per thread
            const int np_f = (cur.Cout + 31) / 32 * 32, np_d = (cur.Cin + 31) / 32 * 32;
            const long long nf = cur.w_fwd ? operand_rows((long long)cur.k * cur.k * cur.Cin, 1) * np_f : 0;
            weight_layout_record(WPair{cur.w, cur.w2, cur.split}, cur.Cout, cur.Cin, cur.k, cur.w_fwd, np_f, cur.w_dgrad, np_d, (cur.dgrad_parity && cur.k == 3) ? 1 : 0, nf,
                                 base - cur.first + 8 * threadIdx.x);
            continue;
        }
#pragma unroll
        for (int u = 0; u < kTile / 256; ++u) {
            const long long i = base + u * 256 + threadIdx.x;
            if (i >= total) break;
            frlw_weight_layout_item_t it = cur;
            if (i >= next) {
                int j = lo + 1;
                while (j + 1 < n && items[j + 1].first <= i) ++j;
                it = items[j];
            }
            const int np_f = (it.Cout + 31) / 32 * 32, np_d = (it.Cin + 31) / 32 * 32;
            const long long nf = it.w_fwd ? operand_rows((long long)it.k * it.k * it.Cin, it.precision) * np_f : 0;
            const int parity = (it.dgrad_parity && it.k == 3) ? 1 : 0;
            if (it.precision == 1) { // every eighth thread-slot lays a record pair (entries start at multiples of 32 elements)
                if ((i & 7) == 0) weight_layout_record(WPair{it.w, it.w2, it.split}, it.Cout, it.Cin, it.k, it.w_fwd, np_f, it.w_dgrad, np_d, parity, nf, i - it.first);
            } else {
                weight_layout_elem(WPair{it.w, it.w2, it.split}, it.Cout, it.Cin, it.k, it.w_fwd, np_f, it.w_dgrad, np_d, parity, nf, i - it.first);
            }
        }
    }
}

// ---- weight gradient ------------------------------------------------------------------------------
// dW[r = (ky, kx, ci)][co] = sum over pixels m of x[m shifted by the tap][ci] * dz[m][co].
// 64 x 64 tile of (r, co) per workgroup, 4 wavefronts 2 x 2, k-tiles of 16 pixels, blockIdx.z owns a slice of the
// pixels and writes a partial tile (always: the reduction kernel also converts to torch's layout).
#include "wgrad_mfma.h"

// Thin layers have few output tiles and therefore many pixel splits: first add groups of kWgradGroup partial tiles
// in parallel (same layout), then the final pass below runs over the group sums.  Fixed order -> deterministic.
constexpr int kWgradGroup = 16;
__global__ void k_wgrad_group_sum(const float *partial, int splits, long long per, float *out)
{
    const int g = blockIdx.y;
    const int z0 = g * kWgradGroup, z1 = z0 + kWgradGroup < splits ? z0 + kWgradGroup : splits;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < per; i += (long long)gridDim.x * blockDim.x) {
        float t[kWgradGroup];
#pragma unroll
        for (int u = 0; u < kWgradGroup; ++u) t[u] = z0 + u < z1 ? partial[(long long)(z0 + u) * per + i] : 0.0f;
        float v = 0.0f;
#pragma unroll
        for (int u = 0; u < kWgradGroup; ++u) v += t[u];
        out[(long long)g * per + i] = v;
    }
}

// dw[co][ci][ky][kx] = sum over splits of partial[z][(ky*k + kx)*Cin + ci][co].  32 x 32 tiles through LDS: the
// partials are read along co (coalesced), the gradient is written along ci (contiguous for 1x1, stride k*k else).
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float *partial, int splits, int R, int Cout, int Cin, int k,
                                                      float *dw)
{
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const long long per = (long long)R * Cout;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = r0 + ty + 8 * j, co = c0 + tx;
        float v = 0.0f;
        if (r < R && co < Cout) {
            const float *src = partial + (long long)r * Cout + co;
            int z = 0;
            for (; z + 8 <= splits; z += 8) { // eight independent loads in flight, fixed summation order
                float t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = src[(long long)(z + u) * per];
#pragma unroll
                for (int u = 0; u < 8; ++u) v += t[u];
            }
            for (; z < splits; ++z) v += src[(long long)z * per];
        }
        tile[ty + 8 * j][tx] = v;
    }
    __syncthreads();
    const int kk = k * k;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = r0 + tx, co = c0 + ty + 8 * j;
        if (r < R && co < Cout) {
            const int tap = r / Cin, ci = r - tap * Cin;
            dw[((long long)co * Cin + ci) * kk + tap] = tile[tx][ty + 8 * j];
        }
    }
}

// ---- BatchNorm (batch statistics) + SiLU ------------------------------------------------------------
// The tensor is an (M, C) row-major matrix (NHWC), C % 4 == 0.  Reductions over M are two-stage and deterministic:
// a workgroup of 256 threads = C4 = min(C / 4, 256) float4 columns x 256 / C4 row lanes takes bn_rows_per_wg(M) rows (column
// chunks of 1024 channels), every thread keeps float64 sums of its four channels with four rows in flight, the row
// lanes are combined through LDS and the workgroup writes partial[wg][c][2]; a second kernel adds the partials of a
// channel with 16 lanes in parallel.
// rows per workgroup: enough workgroups to fill the chip on the mid-size layers, at most 512 rows
__host__ __device__ inline int bn_rows_per_wg(long long M)
{
    long long r = M / 1024;
    r = (r + 31) / 32 * 32;
    if (r < 32) r = 32;
    if (r > 512) r = 512;
    return (int)r;
}

// F32_GROUPS: the four rows in flight are added in float32 before they enter the float64 sums (one conversion and one f64 add
// per four rows instead of four each; f64 runs at half rate on this part).  Used for the backward sums only: the statistics
// pass keeps every term in float64 (sum of squares minus squared mean cancels).
template <bool F32_GROUPS = false, typename F>
__device__ __forceinline__ void bn_reduce_rows(long long M, int C, double *partial, F f)
{
    __shared__ double red[256][8];
    const int tid = threadIdx.x;
    const int rows = bn_rows_per_wg(M);
    const long long row0 = (long long)blockIdx.x * rows;
    long long row1 = row0 + rows;
    if (row1 > M) row1 = M;
    for (int c0 = 0; c0 < C; c0 += 1024) {
        const int c4n = (C - c0) / 4 < 256 ? (C - c0) / 4 : 256; // float4 columns in this chunk
        const int lanes = 256 / c4n;                              // row lanes (>= 1)
        const int q = tid % c4n, rl = tid / c4n;
        const int c = c0 + 4 * q;
        double acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.0;
        if (rl < lanes) {
            long long r = row0 + rl;
            for (; r + 3ll * lanes < row1; r += 4ll * lanes) { // four independent rows in flight
                float a0[8], a1[8], a2[8], a3[8];
                f(r, c, a0); f(r + lanes, c, a1); f(r + 2ll * lanes, c, a2); f(r + 3ll * lanes, c, a3);
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    acc[i] += F32_GROUPS ? (double)((a0[i] + a1[i]) + (a2[i] + a3[i])) : ((double)a0[i] + (double)a1[i]) + ((double)a2[i] + (double)a3[i]);
            }
            for (; r < row1; r += lanes) {
                float a0[8];
                f(r, c, a0);
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] += (double)a0[i];
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) red[tid][i] = acc[i];
        __syncthreads();
        if (tid < c4n) {
            for (int l = 1; l < lanes; ++l)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] += red[tid + l * c4n][i];
            double *dst = partial + ((long long)blockIdx.x * C + c) * 2;
#pragma unroll
            for (int j = 0; j < 4; ++j) { dst[2 * j] = acc[j]; dst[2 * j + 1] = acc[4 + j]; }
        }
        __syncthreads();
    }
}

// partial[wg][c] = {sum, sum of squares} in float64
__global__ __launch_bounds__(256) void k_bn_stats_partial(const float *z, long long M, int C, double *partial)
{
    bn_reduce_rows(M, C, partial, [=](long long r, int c, float (&o)[8]) {
        const float4 v = *(const float4 *)(z + r * C + c);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
        o[4] = v.x * v.x; o[5] = v.y * v.y; o[6] = v.z * v.z; o[7] = v.w * v.w;
    });
}

// adds the partials of kFinCh channels per workgroup, 256 / kFinCh lanes per channel, in a fixed order
// -> tot[c] = {sum a, sum b}.  (16 channels x 16 lanes left 16 workgroups on the chip for a 256-channel layer and took
// 14 us, twice the pass that produced the partials; 4 x 64: C / 4 workgroups, a quarter of the rows per lane.)
constexpr int kFinCh = 4;
__device__ __forceinline__ void bn_sum_partials(const double *partial, int n_wg, int C, double &s0, double &s1, int &c_out)
{
    constexpr int L = 256 / kFinCh;
    __shared__ double red2[256][2];
    const int tid = threadIdx.x, cl = tid % kFinCh, lane = tid / kFinCh;
    const int c = blockIdx.x * kFinCh + cl;
    double a = 0.0, b = 0.0;
    if (c < C) {
        int w = lane;
        for (; w + 3 * L < n_wg; w += 4 * L) { // four independent rows in flight
            const double2 p0 = *(const double2 *)(partial + ((long long)w * C + c) * 2);
            const double2 p1 = *(const double2 *)(partial + ((long long)(w + L) * C + c) * 2);
            const double2 p2 = *(const double2 *)(partial + ((long long)(w + 2 * L) * C + c) * 2);
            const double2 p3 = *(const double2 *)(partial + ((long long)(w + 3 * L) * C + c) * 2);
            a += (p0.x + p1.x) + (p2.x + p3.x);
            b += (p0.y + p1.y) + (p2.y + p3.y);
        }
        for (; w < n_wg; w += L) { a += partial[((long long)w * C + c) * 2]; b += partial[((long long)w * C + c) * 2 + 1]; }
    }
    red2[tid][0] = a;
    red2[tid][1] = b;
    __syncthreads();
    for (int half = L / 2; half >= 1; half >>= 1) { // fixed tree over the lanes of a channel
        if (lane < half) { red2[tid][0] += red2[tid + half * kFinCh][0]; red2[tid][1] += red2[tid + half * kFinCh][1]; }
        __syncthreads();
    }
    s0 = red2[cl][0]; s1 = red2[cl][1]; c_out = (lane == 0 && c < C) ? c : -1;
}

// mean, biased variance, invstd = 1 / sqrt(var + eps) in float32 (what BatchNorm2d normalises with)
// (+ the running statistics of nn.BatchNorm2d when given: momentum update with the unbiased variance)
// (pair: two BaseConvs stacked along the channels -- channels >= split update the second block's running statistics)
struct RunStats2 { float *run_mean2, *run_var2; long long *tracked2; int split; };
__global__ __launch_bounds__(256) void k_bn_stats_final(const double *partial, int n_wg, int C, long long M, float eps,
                                                        float *mean, float *var, float *invstd, float *run_mean,
                                                        float *run_var, float momentum, long long *batches_tracked, RunStats2 pr)
{
    double s, ss;
    int c;
    if (batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) *batches_tracked += 1; // nn.BatchNorm2d.num_batches_tracked
    if (pr.tracked2 && blockIdx.x == 0 && threadIdx.x == 0) *pr.tracked2 += 1;
    bn_sum_partials(partial, n_wg, C, s, ss, c);
    if (c < 0) return;
    const double mu = s / (double)M;
    double v = ss / (double)M - mu * mu;
    if (v < 0.0) v = 0.0;
    mean[c] = (float)mu;
    var[c] = (float)v;
    invstd[c] = (float)(1.0 / sqrt(v + (double)eps));
    if (run_mean) {
        const float unbiased = (float)v * ((float)M / (float)(M > 1 ? M - 1 : 1));
        float *rm = run_mean + c, *rv = run_var + c;
        if (pr.split > 0 && c >= pr.split) { rm = pr.run_mean2 + (c - pr.split); rv = pr.run_var2 + (c - pr.split); }
        *rm = *rm * (1.0f - momentum) + momentum * (float)mu;
        *rv = *rv * (1.0f - momentum) + momentum * unbiased;
    }
}

// sigmoid by the hardware exp2 / reciprocal (~1 ulp each), like the inference epilogue (conv_mfma.h): the exact expf + IEEE
// division are 25 instructions per element and made the backward reduction pass VALU-bound (1.18 ms per step against 0.65 ms of
// traffic); tolerance of the train step against torch is 1e-3, these differ from it by ~1e-7 relative.
__device__ __forceinline__ float sigmoid_fast(float u) { return __frcp_rn(1.0f + __expf(-u)); }
__device__ __forceinline__ float silu_f(float u) { return u * sigmoid_fast(u); }

// Two BaseConvs stacked along the channels (frlw_baseconv_fuse_t::split): channels [0, split) use the first block's gamma / beta
// and rows, channels [split, C) the second block's.  split == 0: one block.  (split % 4 == 0: a float4 never straddles.)
struct Aff2 { const float *g, *b, *g2, *b2; int split; };
__device__ __forceinline__ void aff_of(const Aff2 &a, int c, const float *&g, const float *&b)
{
    const bool second = a.split > 0 && c >= a.split;
    g = second ? a.g2 - a.split : a.g;
    b = second ? a.b2 - a.split : a.b;
}
struct Rows2 { const float *p; long long rs; const float *p2; long long rs2; int split; };
__device__ __forceinline__ const float *row_of(const Rows2 &d, long long r, int c)
{
    return (d.split > 0 && c >= d.split) ? d.p2 + r * d.rs2 + (c - d.split) : d.p + r * d.rs + c;
}

// y = silu(gamma * (z - mean) * invstd + beta) [+ res]
// FUSE: the rows of y and res may be channel slices of wider NHWC tensors (y_rs / res_rs floats between two rows), res may be NULL
template <bool FUSE>
__global__ void k_bn_silu_fwd(const float *z, long long n4, int C, const Aff2 ab, const float *mean,
                              const float *invstd, float *y, long long y_rs, const float *res, long long res_rs, float *y2, long long y2_rs)
{
    const int C4 = C >> 2;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        const float4 v = ((const float4 *)z)[i];
        const float *gamma = ab.g, *beta = ab.b;
        if (FUSE) aff_of(ab, c, gamma, beta);
        float4 o;
        o.x = silu_f(gamma[c + 0] * ((v.x - mean[c + 0]) * invstd[c + 0]) + beta[c + 0]);
        o.y = silu_f(gamma[c + 1] * ((v.y - mean[c + 1]) * invstd[c + 1]) + beta[c + 1]);
        o.z = silu_f(gamma[c + 2] * ((v.z - mean[c + 2]) * invstd[c + 2]) + beta[c + 2]);
        o.w = silu_f(gamma[c + 3] * ((v.w - mean[c + 3]) * invstd[c + 3]) + beta[c + 3]);
        if (FUSE) {
            const long long r = i / C4;
            if (res) {
                const float4 rr = *(const float4 *)(res + r * res_rs + c);
                o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
            }
            if (ab.split > 0 && c >= ab.split) *(float4 *)(y2 + r * y2_rs + (c - ab.split)) = o;
            else *(float4 *)(y + r * y_rs + c) = o;
        } else {
            ((float4 *)y)[i] = o;
        }
    }
}

// du = dy * d silu(u) / du with u = gamma * zhat + beta
__device__ __forceinline__ float dsilu_times(float dy, float u)
{
    const float s = sigmoid_fast(u);
    return dy * (s * (1.0f + u * (1.0f - s)));
}

// partial[wg][c] = {sum du, sum du * zhat} in float64
// (dy_rs: floats between two rows of dy -- C when dense, more when dy is a channel slice of a wider NHWC gradient)
template <bool PAIR>
__global__ __launch_bounds__(256) void k_bn_silu_bwd_partial(const Rows2 dyr, const float *z, long long M, int C,
                                                             const Aff2 ab, const float *mean,
                                                             const float *invstd, double *partial)
{
    bn_reduce_rows<true>(M, C, partial, [=](long long r, int c, float (&o)[8]) {
        const float *gamma = ab.g, *beta = ab.b;
        if (PAIR) aff_of(ab, c, gamma, beta);
        const float4 zv = *(const float4 *)(z + r * C + c), gv = *(const float4 *)(PAIR ? row_of(dyr, r, c) : dyr.p + r * dyr.rs + c);
        const float zz[4] = {zv.x, zv.y, zv.z, zv.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float zh = (zz[j] - mean[c + j]) * invstd[c + j];
            const float du = dsilu_times(gg[j], gamma[c + j] * zh + beta[c + j]);
            o[j] = du;
            o[4 + j] = du * zh;
        }
    });
}

// dbeta = sum du, dgamma = sum du * zhat (float32 outputs); sums[c] = {dbeta / M, dgamma / M} for the apply pass
__global__ __launch_bounds__(256) void k_bn_silu_bwd_final(const double *partial, int n_wg, int C, long long M,
                                                           float *dgamma, float *dbeta, float *sums)
{
    double s, sz;
    int c;
    bn_sum_partials(partial, n_wg, C, s, sz, c);
    if (c < 0) return;
    dbeta[c] = (float)s;
    dgamma[c] = (float)sz;
    sums[2 * c + 0] = (float)(s / (double)M);
    sums[2 * c + 1] = (float)(sz / (double)M);
}

// dz = gamma * invstd * (du - mean(du) - zhat * mean(du * zhat))
template <bool PAIR>
__global__ void k_bn_silu_bwd_apply(const Rows2 dyr, const float *z, long long n4, int C, const Aff2 ab,
                                    const float *mean, const float *invstd, const float *sums, float *dz)
{
    const int C4 = C >> 2;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        const float4 zv = ((const float4 *)z)[i];
        const float *gamma = ab.g, *beta = ab.b;
        if (PAIR) aff_of(ab, c, gamma, beta);
        const float *dy = dyr.p;
        const long long dy_rs = dyr.rs;
        const float4 gv = PAIR ? *(const float4 *)row_of(dyr, i / C4, c)
                               : (dy_rs == C ? ((const float4 *)dy)[i] : *(const float4 *)(dy + (i / C4) * dy_rs + c));
        const float zz[4] = {zv.x, zv.y, zv.z, zv.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float g = gamma[c + j], is = invstd[c + j];
            const float zh = (zz[j] - mean[c + j]) * is;
            const float du = dsilu_times(gg[j], g * zh + beta[c + j]);
            o[j] = g * is * (du - sums[2 * (c + j)] - zh * sums[2 * (c + j) + 1]);
        }
        ((float4 *)dz)[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

inline int npad32(int n) { return (n + 31) / 32 * 32; }

} // namespace

extern "C" {

int frlw_conv2d_dgrad_parity(int k, int stride, int H, int W)
{
    return (stride == 2 && k == 3 && !(H & 1) && !(W & 1)) ? 1 : 0;
}

// precision 1 (split operands) needs the parity classes of the data-gradient operand to start on k-tiles of 16 rows
static inline bool precision_ok(int precision, int Cout, int dgrad_parity)
{
    return precision == 0 || (precision == 1 && (!dgrad_parity || Cout % 16 == 0));
}

int64_t frlw_conv_operand_floats(int K, int N, int precision)
{
    if (K < 1 || N < 1) return 0;
    return operand_rows(K, precision) * (int64_t)npad32(N);
}

static int weight_layouts_impl(const float *w, const float *w2, int split, int Cout, int Cin, int k, int dgrad_parity, float *w_fwd,
                               float *w_dgrad, int precision, frlw_stream_t stream)
{
    (void)hipGetLastError(); // other libraries in the process (torch's BLAS look-ups) leave stale errors behind
    if (!w || Cout < 1 || Cin < 1 || k < 1 || (!w_fwd && !w_dgrad)) return FRLW_ERR_ARG;
    if (split > 0 && (!w2 || split >= Cout)) return FRLW_ERR_ARG;
    if (!precision_ok(precision, Cout, dgrad_parity && w_dgrad)) return FRLW_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const long long total = (w_fwd ? frlw_conv_operand_floats(k * k * Cin, Cout, precision) : 0) + (w_dgrad ? frlw_conv_operand_floats(k * k * Cout, Cin, precision) : 0);
    hipLaunchKernelGGL(k_weight_layouts, dim3(conv_grid_1d(precision == 1 ? total / 8 : total)), dim3(256), 0, s, WPair{w, w2, split > 0 ? split : 0}, Cout, Cin, k, w_fwd, npad32(Cout),
                       w_dgrad, npad32(Cin), (dgrad_parity && k == 3) ? 1 : 0, precision);
    TRY_HIP(hipGetLastError());
    return FRLW_OK;
}

int frlw_conv_weight_layouts(const float *w, int Cout, int Cin, int k, int dgrad_parity, float *w_fwd, float *w_dgrad,
                             int precision, frlw_stream_t stream)
{
    return weight_layouts_impl(w, nullptr, 0, Cout, Cin, k, dgrad_parity, w_fwd, w_dgrad, precision, stream);
}

int frlw_conv_weight_layouts_batch(const frlw_weight_layout_item_t *items, int n, int64_t total, frlw_stream_t stream)
{
    (void)hipGetLastError();
    if (!items || n < 1 || total < 1) return FRLW_ERR_ARG;
    const long long tiles = (total + 2047) / 2048;
    hipLaunchKernelGGL(k_weight_layouts_batch, dim3((unsigned)(tiles < 8192 ? tiles : 8192)), dim3(256), 0, (hipStream_t)stream, items, n, (long long)total);
    TRY_HIP(hipGetLastError());
    return FRLW_OK;
}

static int conv_common(const float *x, int B, int H, int W, int Cin, const float *w_gemm, int Cout, int k, int stride,
                       int tstride, int Ho, int Wo, float *y, float *scratch, int64_t scratch_floats, int precision, hipStream_t s,
                       double *stats = nullptr, int *stats_rows = nullptr, int *sk_counters = nullptr, const float *res = nullptr,
                       long long res_rs = 0)
{
    if (precision != 0 && precision != 1) return FRLW_ERR_ARG;
    (void)hipGetLastError(); // other libraries in the process (torch's BLAS look-ups) leave stale errors behind
    if (!x || !w_gemm || !y || B < 1 || H < 1 || W < 1 || Cin < 4 || (Cin & 3) || Cout < 1 || k < 1) return FRLW_ERR_ARG;
    ConvArgs c = {};
    c.x = x; c.H = H; c.W = W; c.Cin = Cin; c.x_cs = Cin; c.x_co = 0; c.x_bs = (long long)H * W * Cin;
    c.w = w_gemm; c.bias = nullptr; c.Cout = Cout; c.Npad = npad32(Cout); c.k = k; c.stride = stride; c.pad = (k - 1) / 2;
    c.y = y; c.Ho = Ho; c.Wo = Wo; c.y_cs = Cout; c.y_co = 0; c.y_bs = (long long)Ho * Wo * Cout;
    c.res = res; c.r_cs = (int)(res_rs > 0 ? res_rs : Cout); c.r_co = 0; c.r_bs = (long long)Ho * Wo * c.r_cs;
    c.act = ACT_NONE; c.sig_from = 0;
    c.M = B * Ho * Wo; c.K = k * k * Cin;
    c.tstride = tstride;
    c.prec = precision;
    c.stats = stats;
    if (!launch_conv(c, scratch, scratch ? scratch_floats : 0, s, sk_counters)) return FRLW_ERR_UNSUPPORTED;
    if (stats_rows) *stats_rows = c.stats ? c.stats_rows : 0;
    if (hipGetLastError() != hipSuccess) return FRLW_ERR_HIP;
    return FRLW_OK;
}

int frlw_conv2d_fwd(const float *x, int B, int H, int W, int Cin, const float *w_fwd, int Cout, int k, int stride, float *z,
                    float *scratch, int64_t scratch_floats, int precision, frlw_stream_t stream)
{
    if (stride != 1 && stride != 2) return FRLW_ERR_UNSUPPORTED;
    const int pad = (k - 1) / 2;
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    return conv_common(x, B, H, W, Cin, w_fwd, Cout, k, stride, 0, Ho, Wo, z, scratch, scratch_floats, precision, (hipStream_t)stream);
}

static int conv2d_dgrad_impl(const float *dz, int B, int Ho, int Wo, int Cout, const float *w_dgrad, int Cin, int k, int stride,
                             int H, int W, float *dx, float *scratch, int64_t scratch_floats, int precision, frlw_stream_t stream,
                             int *sk_counters, const float *dx_add = nullptr, int64_t dx_add_rs = 0)
{
    if (precision != 0 && precision != 1) return FRLW_ERR_ARG;
    // dx_add: dx = data gradient + dx_add in the epilogue (the gradient a second consumer of x has produced already; rows of
    // dx_add_rs floats).  Stride-1 layers only: the parity classes of a stride-2 layer write interleaved rows.
    if (dx_add && (stride != 1 || (dx_add_rs > 0 && (dx_add_rs < Cin || (dx_add_rs & 3))) || ((uintptr_t)dx_add & 15))) return FRLW_ERR_UNSUPPORTED;
    // dx[iy][ix][ci] = sum dz[(iy + pad - ky) / s][(ix + pad - kx) / s][co] * w[co][ci][ky][kx]: a stride-1 convolution of
    // dz (transposed gather for s = 2) with the flipped operand and padding k - 1 - pad = pad (odd k)
    if (stride != 1 && stride != 2) return FRLW_ERR_UNSUPPORTED;
    if (!(k & 1)) return FRLW_ERR_UNSUPPORTED;
    if (frlw_conv2d_dgrad_parity(k, stride, H, W)) {
        // four stride-1 convolutions of dz, one per output parity class, with 1 / 2 / 2 / 4 of the nine taps (no
        // multiplications by the zeros of the up-sampled gradient); class (py, px) writes dx[2 o' + (py, px)]
        if (Ho * 2 != H || Wo * 2 != W) return FRLW_ERR_ARG;
        if (!dz || !w_dgrad || !dx || B < 1 || Cout < 4 || (Cout & 3) || Cin < 1) return FRLW_ERR_ARG;
        if (!precision_ok(precision, Cout, 1)) return FRLW_ERR_UNSUPPORTED;
        (void)hipGetLastError();
        static const int row0[4] = {0, 1, 3, 5};
        for (int cls = 0; cls < 4; ++cls) {
            const int py = cls >> 1, px = cls & 1, kh = py ? 2 : 1, kwc = px ? 2 : 1;
            ConvArgs c = {};
            c.x = dz; c.H = Ho; c.W = Wo; c.Cin = Cout; c.x_cs = Cout; c.x_co = 0; c.x_bs = (long long)Ho * Wo * Cout;
            c.w = w_dgrad + (long long)row0[cls] * Cout * npad32(Cin); c.bias = nullptr;
            c.Cout = Cin; c.Npad = npad32(Cin); c.k = kh; c.kw = kwc; c.stride = 1; c.pad = 0;
            c.y = dx; c.Ho = Ho; c.Wo = Wo; c.y_cs = 2 * Cin; c.y_rp = 2 * W * Cin; c.y_co = (py * W + px) * Cin;
            c.y_bs = (long long)H * W * Cin;
            c.res = nullptr; c.act = ACT_NONE;
            c.prec = precision;
            c.M = B * Ho * Wo; c.K = kh * kwc * Cout;
            if (!launch_conv(c, scratch, scratch ? scratch_floats : 0, (hipStream_t)stream, sk_counters)) return FRLW_ERR_UNSUPPORTED;
        }
        if (hipGetLastError() != hipSuccess) return FRLW_ERR_HIP;
        return FRLW_OK;
    }
    return conv_common(dz, B, Ho, Wo, Cout, w_dgrad, Cin, k, 1, stride == 2 ? 2 : 0, H, W, dx, scratch, scratch_floats,
                       precision, (hipStream_t)stream, nullptr, nullptr, sk_counters, dx_add, dx_add_rs);
}

int frlw_conv2d_dgrad(const float *dz, int B, int Ho, int Wo, int Cout, const float *w_dgrad, int Cin, int k, int stride,
                      int H, int W, float *dx, float *scratch, int64_t scratch_floats, int precision, frlw_stream_t stream)
{
    return conv2d_dgrad_impl(dz, B, Ho, Wo, Cout, w_dgrad, Cin, k, stride, H, W, dx, scratch, scratch_floats, precision, stream, nullptr);
}

int64_t frlw_conv2d_wgrad_scratch_floats(int B, int Ho, int Wo, int Cin, int Cout, int k)
{
    const long long R = (long long)k * k * Cin;
    static const long long target = dev_knob("FRLW_WGRAD_TARGET", 1024ll); // four 128 x 128 workgroups per CU (wgrad_want_splits rounds DOWN to it)
    const long long sp = wgrad_want_splits(R, Cout, (long long)B * Ho * Wo, target);
    const long long groups = sp > 64 ? (sp + kWgradGroup - 1) / kWgradGroup : 0;
    return (sp + groups) * R * Cout;
}

static long long wgrad_splits_for(long long scratch_floats, long long per, long long want_total)
{
    // want_total = (sp + groups) * per from the query; recover the largest sp that fits the given scratch
    long long sp = want_total / per;
    if (sp > 64) { // undo the group allowance: sp + ceil(sp / 16) <= want_total / per
        long long s = sp * kWgradGroup / (kWgradGroup + 1);
        while (s + (s + kWgradGroup - 1) / kWgradGroup > sp) --s;
        sp = s;
    }
    long long fit = scratch_floats / per;
    if (fit > 64) { long long s = fit * kWgradGroup / (kWgradGroup + 1); while (s + (s + kWgradGroup - 1) / kWgradGroup > fit) --s; fit = s > 64 ? s : 64; }
    return sp < fit ? sp : fit;
}

int frlw_conv2d_wgrad(const float *x, int B, int H, int W, int Cin, const float *dz, int Ho, int Wo, int Cout, int k,
                      int stride, float *dw, float *scratch, int64_t scratch_floats, int precision, frlw_stream_t stream)
{
    (void)hipGetLastError(); // other libraries in the process (torch's BLAS look-ups) leave stale errors behind
    if (!x || !dz || !dw || !scratch || B < 1 || (Cin & 3) || (Cout & 3) || Cin < 4 || Cout < 4) return FRLW_ERR_ARG;
    if (precision != 0 && precision != 1) return FRLW_ERR_ARG;
    WgradArgs a = {};
    a.prec = precision;
    a.x = x; a.H = H; a.W = W; a.Cin = Cin; a.x_bs = (long long)H * W * Cin; a.x_cs = Cin;
    a.dz = dz; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout; a.dz_bs = (long long)Ho * Wo * Cout; a.dz_cs = Cout;
    a.k = k; a.stride = stride; a.pad = (k - 1) / 2;
    a.R = k * k * Cin; a.M = B * Ho * Wo;
    const long long per = (long long)a.R * Cout;
    const long long sp = wgrad_splits_for(scratch_floats, per, frlw_conv2d_wgrad_scratch_floats(B, Ho, Wo, Cin, Cout, k));
    if (sp < 1) return FRLW_ERR_WORKSPACE;
    a.splits = (int)sp;
    a.partial = scratch;
    hipStream_t s = (hipStream_t)stream;
    if (!launch_wgrad_tiles(a, s)) return FRLW_ERR_UNSUPPORTED; // a tensor beyond 3.7 GB: split the batch
    const float *final_src = scratch;
    int final_n = a.splits;
    if (a.splits > 64) { // group sums go behind the partial tiles (the scratch query reserves the room)
        const int groups = (a.splits + kWgradGroup - 1) / kWgradGroup;
        float *gs = scratch + (long long)a.splits * per;
        hipLaunchKernelGGL(k_wgrad_group_sum, dim3(conv_grid_1d(per), groups), dim3(256), 0, s, scratch, a.splits, per, gs);
        final_src = gs;
        final_n = groups;
    }
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((a.R + 31) / 32, (Cout + 31) / 32), dim3(256), 0, s, final_src, final_n, a.R, Cout, Cin, k, dw);
    TRY_HIP(hipGetLastError());
    return FRLW_OK;
}

int64_t frlw_bn_scratch_doubles(int64_t M, int C)
{
    const int rows = bn_rows_per_wg(M);
    const int64_t pass = ((M + rows - 1) / rows) * (int64_t)C * 2;
    const int64_t fused = ((M + 63) / 64) * (int64_t)C * 2; // the convolution epilogue's slabs of >= 64 rows (conv_mfma.h: ConvArgs::stats)
    return pass > fused ? pass : fused;
}

static int bn_stats_impl(const float *z, int64_t M, int C, float eps, float *mean, float *var, float *invstd,
                         double *scratch, float *run_mean, float *run_var, float momentum, long long *batches_tracked,
                         frlw_stream_t stream, int have_partials = 0, RunStats2 pr = RunStats2{nullptr, nullptr, nullptr, 0});

int frlw_bn_stats(const float *z, int64_t M, int C, float eps, float *mean, float *var, float *invstd, double *scratch,
                  frlw_stream_t stream)
{
    return bn_stats_impl(z, M, C, eps, mean, var, invstd, scratch, nullptr, nullptr, 0.0f, nullptr, stream);
}

static int bn_stats_impl(const float *z, int64_t M, int C, float eps, float *mean, float *var, float *invstd,
                         double *scratch, float *run_mean, float *run_var, float momentum, long long *batches_tracked,
                         frlw_stream_t stream, int have_partials, RunStats2 pr)
{
    (void)hipGetLastError(); // other libraries in the process (torch's BLAS look-ups) leave stale errors behind
    if (!z || !mean || !var || !invstd || !scratch || M < 1 || C < 4 || (C & 3)) return FRLW_ERR_ARG;
    int n_wg = (int)((M + bn_rows_per_wg(M) - 1) / bn_rows_per_wg(M));
    hipStream_t s = (hipStream_t)stream;
    if (have_partials > 0) n_wg = have_partials; // the convolution's epilogue has written `have_partials` slabs into scratch
    else hipLaunchKernelGGL(k_bn_stats_partial, dim3(n_wg), dim3(256), 0, s, z, (long long)M, C, scratch);
    hipLaunchKernelGGL(k_bn_stats_final, dim3((C + kFinCh - 1) / kFinCh), dim3(256), 0, s, scratch, n_wg, C, (long long)M, eps, mean,
                       var, invstd, run_mean, run_var, momentum, batches_tracked, pr);
    TRY_HIP(hipGetLastError());
    return FRLW_OK;
}

// split > 0: channels [split, C) are a second block's -- gamma2 / beta2, rows of y2 (y2_rs floats apart; the first block's y rows
// are then `split` wide by default)
static int bn_silu_fwd_impl(const float *z, int64_t M, int C, const float *gamma, const float *beta, const float *mean,
                            const float *invstd, float *y, int64_t y_rs, const float *res, int64_t res_rs, frlw_stream_t stream,
                            int split = 0, const float *gamma2 = nullptr, const float *beta2 = nullptr, float *y2 = nullptr, int64_t y2_rs = 0)
{
    (void)hipGetLastError(); // other libraries in the process (torch's BLAS look-ups) leave stale errors behind
    if (!z || !y || !gamma || !beta || !mean || !invstd || M < 1 || C < 4 || (C & 3)) return FRLW_ERR_ARG;
    const int c1 = split > 0 ? split : C; // channels of the (first) block
    if (split > 0 && ((split & 3) || split >= C || !gamma2 || !beta2 || !y2 || res)) return FRLW_ERR_ARG;
    if (y_rs <= 0) y_rs = c1;
    if (res_rs <= 0) res_rs = C;
    if (y2_rs <= 0) y2_rs = C - c1;
    if (y_rs < c1 || (y_rs & 3) || ((uintptr_t)y & 15) || (res && (res_rs < C || (res_rs & 3) || ((uintptr_t)res & 15)))) return FRLW_ERR_ARG;
    if (split > 0 && (y2_rs < C - c1 || (y2_rs & 3) || ((uintptr_t)y2 & 15))) return FRLW_ERR_ARG;
    const long long n4 = (long long)M * C / 4;
    const Aff2 ab = {gamma, beta, gamma2, beta2, split > 0 ? split : 0};
    if (res || y_rs != C || split > 0)
        hipLaunchKernelGGL(k_bn_silu_fwd<true>, dim3(conv_grid_1d(n4)), dim3(256), 0, (hipStream_t)stream, z, n4, C, ab, mean,
                           invstd, y, (long long)y_rs, res, (long long)res_rs, y2, (long long)y2_rs);
    else
        hipLaunchKernelGGL(k_bn_silu_fwd<false>, dim3(conv_grid_1d(n4)), dim3(256), 0, (hipStream_t)stream, z, n4, C, ab, mean,
                           invstd, y, (long long)C, (const float *)nullptr, (long long)C, (float *)nullptr, (long long)0);
    TRY_HIP(hipGetLastError());
    return FRLW_OK;
}

int frlw_bn_silu_fwd(const float *z, int64_t M, int C, const float *gamma, const float *beta, const float *mean,
                     const float *invstd, float *y, frlw_stream_t stream)
{
    return bn_silu_fwd_impl(z, M, C, gamma, beta, mean, invstd, y, 0, nullptr, 0, stream);
}

// split > 0: channels [split, C) are a second block's -- gamma2 / beta2, gradient rows dy2 (dy2_rs floats apart; dy's rows are
// then `split` wide by default); z, dz, mean, invstd, dgamma, dbeta stay stacked (C wide)
static int bn_silu_bwd_impl(const float *dy, int64_t dy_row_stride, const float *z, int64_t M, int C, const float *gamma, const float *beta,
                            const float *mean, const float *invstd, float *dz, float *dgamma, float *dbeta, double *scratch,
                            float *sums, frlw_stream_t stream, int split = 0, const float *gamma2 = nullptr, const float *beta2 = nullptr,
                            const float *dy2 = nullptr, int64_t dy2_row_stride = 0)
{
    (void)hipGetLastError(); // other libraries in the process (torch's BLAS look-ups) leave stale errors behind
    if (!dy || !z || !dz || !dgamma || !dbeta || !scratch || !sums || M < 1 || C < 4 || (C & 3)) return FRLW_ERR_ARG;
    const int c1 = split > 0 ? split : C;
    if (split > 0 && ((split & 3) || split >= C || !gamma2 || !beta2 || !dy2)) return FRLW_ERR_ARG;
    const long long dy_rs = dy_row_stride > 0 ? dy_row_stride : c1;
    const long long dy2_rs = dy2_row_stride > 0 ? dy2_row_stride : C - c1;
    if (dy_rs < c1 || (dy_rs & 3) || ((uintptr_t)dy & 15)) return FRLW_ERR_ARG;
    if (split > 0 && (dy2_rs < C - c1 || (dy2_rs & 3) || ((uintptr_t)dy2 & 15))) return FRLW_ERR_ARG;
    const int n_wg = (int)((M + bn_rows_per_wg(M) - 1) / bn_rows_per_wg(M));
    hipStream_t s = (hipStream_t)stream;
    const Aff2 ab = {gamma, beta, gamma2, beta2, split > 0 ? split : 0};
    const Rows2 dyr = {dy, dy_rs, dy2, dy2_rs, split > 0 ? split : 0};
    const long long n4 = (long long)M * C / 4;
    if (split > 0) {
        hipLaunchKernelGGL(k_bn_silu_bwd_partial<true>, dim3(n_wg), dim3(256), 0, s, dyr, z, (long long)M, C, ab, mean, invstd, scratch);
        hipLaunchKernelGGL(k_bn_silu_bwd_final, dim3((C + kFinCh - 1) / kFinCh), dim3(256), 0, s, scratch, n_wg, C, (long long)M, dgamma, dbeta, sums);
        hipLaunchKernelGGL(k_bn_silu_bwd_apply<true>, dim3(conv_grid_1d(n4)), dim3(256), 0, s, dyr, z, n4, C, ab, mean, invstd, sums, dz);
    } else {
        hipLaunchKernelGGL(k_bn_silu_bwd_partial<false>, dim3(n_wg), dim3(256), 0, s, dyr, z, (long long)M, C, ab, mean, invstd, scratch);
        hipLaunchKernelGGL(k_bn_silu_bwd_final, dim3((C + kFinCh - 1) / kFinCh), dim3(256), 0, s, scratch, n_wg, C, (long long)M, dgamma, dbeta, sums);
        hipLaunchKernelGGL(k_bn_silu_bwd_apply<false>, dim3(conv_grid_1d(n4)), dim3(256), 0, s, dyr, z, n4, C, ab, mean, invstd, sums, dz);
    }
    TRY_HIP(hipGetLastError());
    return FRLW_OK;
}

int frlw_bn_silu_bwd(const float *dy, int64_t dy_row_stride, const float *z, int64_t M, int C, const float *gamma, const float *beta,
                     const float *mean, const float *invstd, float *dz, float *dgamma, float *dbeta, double *scratch,
                     float *sums, frlw_stream_t stream)
{
    return bn_silu_bwd_impl(dy, dy_row_stride, z, M, C, gamma, beta, mean, invstd, dz, dgamma, dbeta, scratch, sums, stream);
}

/* ---- one call per BaseConv and direction (the per-operator entry points above stay for tests / other callers) ---- */
int64_t frlw_baseconv_weight_cache_floats(int Cin, int Cout, int k, int precision)
{
    return frlw_conv_operand_floats(k * k * Cin, Cout, precision) + frlw_conv_operand_floats(k * k * Cout, Cin, precision);
}

int64_t frlw_baseconv_train_scratch_bytes(int B, int H, int W, int Cin, int Cout, int k, int stride)
{
    const int pad = (k - 1) / 2;
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    const int64_t M = (int64_t)B * Ho * Wo;
    int64_t bytes = 0;
    bytes += sizeof(float) * frlw_conv_operand_floats(k * k * Cin, Cout, 1);   // forward operand (either precision fits)
    bytes += sizeof(float) * frlw_conv_operand_floats(k * k * Cout, Cin, 1);   // data-gradient operand
    bytes += sizeof(double) * frlw_bn_scratch_doubles(M, Cout);     // reduction partials
    bytes += sizeof(float) * 2 * Cout;                              // sums
    bytes += sizeof(float) * frlw_conv2d_wgrad_scratch_floats(B, Ho, Wo, Cin, Cout, k);
    bytes += sizeof(float) * 8ll * 1024 * 1024;                     // split-K partials of forward / data gradient
    return bytes + 6 * 256;
}

namespace {
struct TrainScratch { float *w_fwd, *w_dg, *sums, *wgrad, *splitk; double *red; int64_t wgrad_floats, splitk_floats; };
inline TrainScratch carve(void *scratch, int B, int Ho, int Wo, int Cin, int Cout, int k)
{
    char *p = (char *)scratch;
    auto take = [&](int64_t bytes) { char *r = p; p += (bytes + 255) / 256 * 256; return r; };
    TrainScratch t;
    t.w_fwd = (float *)take(sizeof(float) * frlw_conv_operand_floats(k * k * Cin, Cout, 1));
    t.w_dg = (float *)take(sizeof(float) * frlw_conv_operand_floats(k * k * Cout, Cin, 1));
    t.red = (double *)take(sizeof(double) * frlw_bn_scratch_doubles((int64_t)B * Ho * Wo, Cout));
    t.sums = (float *)take(sizeof(float) * 2 * Cout);
    t.wgrad_floats = frlw_conv2d_wgrad_scratch_floats(B, Ho, Wo, Cin, Cout, k);
    t.wgrad = (float *)take(sizeof(float) * t.wgrad_floats);
    t.splitk_floats = 8ll * 1024 * 1024;
    t.splitk = (float *)take(sizeof(float) * t.splitk_floats);
    return t;
}
} // namespace

/* y = silu(bn(conv(x, w))) with batch statistics; z (the convolution output), mean, var (biased), invstd are outputs
 * the backward needs.  scratch: frlw_baseconv_train_scratch_bytes bytes, contents not needed afterwards. */
int frlw_baseconv_train_fwd(const float *x, const float *w, const float *gamma, const float *beta, float eps, int B, int H,
                            int W, int Cin, int Cout, int k, int stride, float *z, float *y, float *mean, float *var,
                            float *invstd, float *running_mean, float *running_var, float momentum,
                            int64_t *num_batches_tracked, float *w_cache, void *scratch, int64_t scratch_bytes,
                            int *splitk_counters, const frlw_baseconv_fuse_t *fuse, int precision, frlw_stream_t stream)
{
    if (!x || (!w && !w_cache) || !gamma || !beta || !z || !y || !mean || !var || !invstd || !scratch) return FRLW_ERR_ARG;
    if (fuse && fuse->struct_size != (int32_t)sizeof(frlw_baseconv_fuse_t)) return FRLW_ERR_ARG;
    const int split = fuse ? fuse->split : 0; // > 0: two blocks stacked along the output channels (frlw_evd.h)
    if (split < 0 || (split > 0 && ((split & 3) || split >= Cout || !fuse->gamma2 || !fuse->beta2 || !fuse->y2 || fuse->residual || (w && !fuse->w2))))
        return FRLW_ERR_ARG;
    if (scratch_bytes < frlw_baseconv_train_scratch_bytes(B, H, W, Cin, Cout, k, stride)) return FRLW_ERR_WORKSPACE;
    const int pad = (k - 1) / 2;
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    const int64_t M = (int64_t)B * Ho * Wo;
    TrainScratch t = carve(scratch, B, Ho, Wo, Cin, Cout, k);
    int rc;
    const float *w2 = split > 0 ? fuse->w2 : nullptr;
    const float *w_fwd = t.w_fwd;
    if (w_cache && !w) { // the caller has laid both operands out already (frlw_conv_weight_layouts_batch)
        w_fwd = w_cache;
    } else if (w_cache) { // both operands in ONE launch, kept by the caller: the backward of this step finds its operand ready
        float *w_dg = w_cache + frlw_conv_operand_floats(k * k * Cin, Cout, precision);
        if ((rc = weight_layouts_impl(w, w2, split, Cout, Cin, k, frlw_conv2d_dgrad_parity(k, stride, H, W), w_cache, w_dg, precision, stream)) != FRLW_OK) return rc;
        w_fwd = w_cache;
    } else if ((rc = weight_layouts_impl(w, w2, split, Cout, Cin, k, 0, t.w_fwd, nullptr, precision, stream)) != FRLW_OK) return rc;
    if (stride != 1 && stride != 2) return FRLW_ERR_UNSUPPORTED;
    int stat_rows = 0; // > 0: the convolution's epilogue left the column sums of its output slabs in t.red
    if ((rc = conv_common(x, B, H, W, Cin, w_fwd, Cout, k, stride, 0, Ho, Wo, z, t.splitk, t.splitk_floats, precision, (hipStream_t)stream,
                          t.red, &stat_rows, splitk_counters)) != FRLW_OK) return rc;
    RunStats2 pr = {nullptr, nullptr, nullptr, 0};
    if (split > 0 && running_mean) {
        if (!fuse->running_mean2 || !fuse->running_var2) return FRLW_ERR_ARG;
        pr = RunStats2{fuse->running_mean2, fuse->running_var2, (long long *)fuse->num_batches_tracked2, split};
    }
    if ((rc = bn_stats_impl(z, M, Cout, eps, mean, var, invstd, t.red, running_mean, running_mean ? running_var : nullptr,
                            momentum, (long long *)num_batches_tracked, stream, stat_rows, pr)) != FRLW_OK) return rc;
    return bn_silu_fwd_impl(z, M, Cout, gamma, beta, mean, invstd, y, fuse ? fuse->y_row_stride : 0, fuse ? fuse->residual : nullptr,
                            fuse ? fuse->residual_row_stride : 0, stream, split, split > 0 ? fuse->gamma2 : nullptr,
                            split > 0 ? fuse->beta2 : nullptr, split > 0 ? fuse->y2 : nullptr, split > 0 ? fuse->y2_row_stride : 0);
}

/* Gradients of the block: dx (NULL: not needed), dw (Cout, Cin, k, k), dgamma, dbeta.  dz: (B, Ho, Wo, Cout) work buffer. */
int frlw_baseconv_train_bwd(const float *dy, int64_t dy_row_stride, const float *x, const float *z, const float *w, const float *gamma,
                            const float *beta, const float *mean, const float *invstd, int B, int H, int W, int Cin,
                            int Cout, int k, int stride, float *dz, float *dx, float *dw, float *dgamma, float *dbeta,
                            const float *w_cache, void *scratch, int64_t scratch_bytes, int *splitk_counters,
                            const frlw_baseconv_fuse_t *fuse, int precision, frlw_stream_t stream)
{
    if (!dy || !x || !z || !w || !gamma || !beta || !mean || !invstd || !dz || !dw || !dgamma || !dbeta || !scratch)
        return FRLW_ERR_ARG;
    if (fuse && fuse->struct_size != (int32_t)sizeof(frlw_baseconv_fuse_t)) return FRLW_ERR_ARG;
    if (fuse && fuse->dx_add && (!dx || stride != 1)) return FRLW_ERR_UNSUPPORTED;
    const int split = fuse ? fuse->split : 0;
    if (split < 0 || (split > 0 && ((split & 3) || split >= Cout || !fuse->gamma2 || !fuse->beta2 || !fuse->dy2 || (!w_cache && !fuse->w2))))
        return FRLW_ERR_ARG;
    if (scratch_bytes < frlw_baseconv_train_scratch_bytes(B, H, W, Cin, Cout, k, stride)) return FRLW_ERR_WORKSPACE;
    const int pad = (k - 1) / 2;
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    const int64_t M = (int64_t)B * Ho * Wo;
    TrainScratch t = carve(scratch, B, Ho, Wo, Cin, Cout, k);
    int rc;
    if ((rc = bn_silu_bwd_impl(dy, dy_row_stride, z, M, Cout, gamma, beta, mean, invstd, dz, dgamma, dbeta, t.red, t.sums, stream, split,
                               split > 0 ? fuse->gamma2 : nullptr, split > 0 ? fuse->beta2 : nullptr, split > 0 ? fuse->dy2 : nullptr,
                               split > 0 ? fuse->dy2_row_stride : 0)) != FRLW_OK) return rc;
    if (dx) {
        const float *w_dg = t.w_dg;
        if (w_cache) w_dg = w_cache + frlw_conv_operand_floats(k * k * Cin, Cout, precision); // laid out by the forward of this step
        else if ((rc = weight_layouts_impl(w, split > 0 ? fuse->w2 : nullptr, split, Cout, Cin, k, frlw_conv2d_dgrad_parity(k, stride, H, W), nullptr, t.w_dg, precision, stream)) != FRLW_OK) return rc;
        if ((rc = conv2d_dgrad_impl(dz, B, Ho, Wo, Cout, w_dg, Cin, k, stride, H, W, dx, t.splitk, t.splitk_floats, precision, stream, splitk_counters,
                                    fuse ? fuse->dx_add : nullptr, fuse ? fuse->dx_add_row_stride : 0)) != FRLW_OK) return rc;
    }
    return frlw_conv2d_wgrad(x, B, H, W, Cin, dz, Ho, Wo, Cout, k, stride, dw, t.wgrad, t.wgrad_floats, precision, stream);
}

} // extern "C"
