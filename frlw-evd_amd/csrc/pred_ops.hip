// pred_ops.hip -- the three biased 1x1 prediction convolutions of one head level in training mode, forward and backward.
//
// Replaces, per level k of YOLOXHead (core/yolox/models/yolo_head.py:160-186, training branch),
//     torch.cat([reg_preds[k](reg_feat), obj_preds[k](reg_feat), cls_preds[k](cls_feat)], 1)
// and its autograd.  With 4 + 1 + nc output channels these are not GEMMs worth a matrix pipe: every input row (C floats
// of reg_feat and of cls_feat) is read once and dotted with 5 + nc weight rows -- HBM-bound streaming, 2 * M * C * 4 bytes
// forward, twice that backward (the rows again for the weight gradient, the feature gradients out).
//
//   forward   out[m][0:4] = reg_feat[m] . w_reg + b_reg, out[m][4] = reg_feat[m] . w_obj + b_obj,
//             out[m][5:]  = cls_feat[m] . w_cls + b_cls                       (M, 5 + nc) row-major
//   backward  d_reg_feat[m] = dout[m][0:5] . [w_reg; w_obj], d_cls_feat[m] = dout[m][5:] . w_cls,
//             dw[j][c] = sum_m dout[m][j] * feat_j[m][c], db[j] = sum_m dout[m][j]
// One wavefront owns whole rows: lane l holds the float4 chunks l, l + 64 (C <= 512) of a row and of every weight row
// (registers, loaded once); the 5 + nc partial dots of a row are folded over the 64 lanes with a halving butterfly
// (8 values -> 4 -> 2 -> 1 while the partner distance goes 32 -> 16 -> 8, then 3 plain steps).  The weight gradient is
// accumulated per wavefront in registers over a FIXED set of rows; the eight wavefronts of a workgroup are added in
// wavefront order through LDS and the workgroup partials in workgroup order by a second kernel: deterministic.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "frlw_evd.h"

namespace {

constexpr int kPredWaves = 8;      // wavefronts per workgroup
constexpr int kPredMaxWg = 1024;   // workgroups of a launch (rows are dealt round-robin to wavefronts)

struct PredArgs {
    const float *reg_feat, *cls_feat;  // (M, C)
    const float *w_reg, *w_obj, *w_cls; // (4, C), (1, C), (nc, C)
    const float *b_reg, *b_obj, *b_cls;
    long long M;
    int C, nc;
    float *out;                         // forward: (M, 5 + nc)
    const float *dout;                  // backward
    float *d_reg, *d_cls;               // (M, C)
    float *partial;                     // [workgroup][5 + nc][C] weight-gradient partials, then [workgroup][16] bias partials
    int n_waves;
};

__device__ __forceinline__ float lane_xchg(float v, int mask) { return __shfl_xor(v, mask, 64); }

// Sum of v[j] over the 64 lanes for j < 8, two groups of 8 at most (NOUT <= 16).  Returns in `res` the total of value
// index (lane >> 3) & 7 of each group -- every lane ends with ONE value per group.
template <int NG>
__device__ __forceinline__ void fold8(float (&v)[NG][8], float (&res)[NG], int lane)
{
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        // distance 32: keep values 0-3 (lanes < 32) or 4-7 (lanes >= 32)
        float a[4];
        const bool hi = lane & 32;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float mine = hi ? v[g][4 + t] : v[g][t], theirs = hi ? v[g][t] : v[g][4 + t];
            a[t] = mine + lane_xchg(theirs, 32);
        }
        // distance 16: keep 2 of the 4
        float b[2];
        const bool hi2 = lane & 16;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float mine = hi2 ? a[2 + t] : a[t], theirs = hi2 ? a[t] : a[2 + t];
            b[t] = mine + lane_xchg(theirs, 16);
        }
        // distance 8: keep 1 of the 2
        const bool hi3 = lane & 8;
        const float mine = hi3 ? b[1] : b[0], theirs = hi3 ? b[0] : b[1];
        float c = mine + lane_xchg(theirs, 8);
        c += lane_xchg(c, 4);
        c += lane_xchg(c, 2);
        c += lane_xchg(c, 1);
        res[g] = c; // total of value index ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1)
    }
}
__device__ __forceinline__ int fold8_index(int lane) { return ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1); }

// CPL: float4 chunks per lane (C <= 256: 1, C <= 512: 2); NG: groups of 8 outputs (5 + nc <= 8: 1, else 2)
template <int CPL, int NG>
__global__ __launch_bounds__(64 * kPredWaves) void k_pred_fwd(PredArgs a)
{
    const int lane = threadIdx.x & 63, wave = blockIdx.x * kPredWaves + (threadIdx.x >> 6);
    const int nout = 5 + a.nc, c4n = a.C / 4;
    float4 w[NG * 8][CPL];
    float bias[NG];
#pragma unroll
    for (int j = 0; j < NG * 8; ++j)
#pragma unroll
        for (int u = 0; u < CPL; ++u) {
            const int ch = lane + 64 * u;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < nout && ch < c4n) {
                const float *src = j < 4 ? a.w_reg + (long long)j * a.C : j == 4 ? a.w_obj : a.w_cls + (long long)(j - 5) * a.C;
                t = *(const float4 *)(src + 4 * ch);
            }
            w[j][u] = t;
        }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int j = g * 8 + fold8_index(lane);
        bias[g] = j < 4 ? a.b_reg[j] : j == 4 ? a.b_obj[0] : j < nout ? a.b_cls[j - 5] : 0.0f;
    }
    for (long long m = wave; m < a.M; m += a.n_waves) {
        float4 xr[CPL], xc[CPL];
#pragma unroll
        for (int u = 0; u < CPL; ++u) {
            const int ch = lane + 64 * u;
            xr[u] = ch < c4n ? *(const float4 *)(a.reg_feat + m * a.C + 4 * ch) : make_float4(0.f, 0.f, 0.f, 0.f);
            xc[u] = ch < c4n ? *(const float4 *)(a.cls_feat + m * a.C + 4 * ch) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float v[NG][8];
#pragma unroll
        for (int j = 0; j < NG * 8; ++j) {
            float s = 0.0f;
#pragma unroll
            for (int u = 0; u < CPL; ++u) {
                const float4 x = j < 5 ? xr[u] : xc[u];
                s += x.x * w[j][u].x + x.y * w[j][u].y + x.z * w[j][u].z + x.w * w[j][u].w;
            }
            v[j >> 3][j & 7] = s;
        }
        float res[NG];
        fold8<NG>(v, res, lane);
        if ((lane & 7) == 0) {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int j = g * 8 + fold8_index(lane);
                if (j < nout) a.out[m * nout + j] = res[g] + bias[g];
            }
        }
    }
}

template <int CPL, int NG>
__global__ __launch_bounds__(64 * kPredWaves) void k_pred_bwd(PredArgs a)
{
    const int lane = threadIdx.x & 63, wave = blockIdx.x * kPredWaves + (threadIdx.x >> 6);
    const int nout = 5 + a.nc, c4n = a.C / 4;
    float4 w[NG * 8][CPL], dw[NG * 8][CPL];
    float db[NG * 8];
#pragma unroll
    for (int j = 0; j < NG * 8; ++j) {
        db[j] = 0.0f;
#pragma unroll
        for (int u = 0; u < CPL; ++u) {
            const int ch = lane + 64 * u;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < nout && ch < c4n) {
                const float *src = j < 4 ? a.w_reg + (long long)j * a.C : j == 4 ? a.w_obj : a.w_cls + (long long)(j - 5) * a.C;
                t = *(const float4 *)(src + 4 * ch);
            }
            w[j][u] = t;
            dw[j][u] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    for (long long m = wave; m < a.M; m += a.n_waves) {
        float4 xr[CPL], xc[CPL];
#pragma unroll
        for (int u = 0; u < CPL; ++u) {
            const int ch = lane + 64 * u;
            xr[u] = ch < c4n ? *(const float4 *)(a.reg_feat + m * a.C + 4 * ch) : make_float4(0.f, 0.f, 0.f, 0.f);
            xc[u] = ch < c4n ? *(const float4 *)(a.cls_feat + m * a.C + 4 * ch) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float g[NG * 8];
#pragma unroll
        for (int j = 0; j < NG * 8; ++j) g[j] = j < nout ? a.dout[m * nout + j] : 0.0f; // wave-uniform addresses: broadcast loads
        float4 dr[CPL], dc[CPL];
#pragma unroll
        for (int u = 0; u < CPL; ++u) { dr[u] = make_float4(0.f, 0.f, 0.f, 0.f); dc[u] = make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
        for (int j = 0; j < NG * 8; ++j) {
            db[j] += g[j];
#pragma unroll
            for (int u = 0; u < CPL; ++u) {
                const float4 x = j < 5 ? xr[u] : xc[u];
                dw[j][u].x += g[j] * x.x; dw[j][u].y += g[j] * x.y; dw[j][u].z += g[j] * x.z; dw[j][u].w += g[j] * x.w;
                float4 &d = j < 5 ? dr[u] : dc[u];
                d.x += g[j] * w[j][u].x; d.y += g[j] * w[j][u].y; d.z += g[j] * w[j][u].z; d.w += g[j] * w[j][u].w;
            }
        }
#pragma unroll
        for (int u = 0; u < CPL; ++u) {
            const int ch = lane + 64 * u;
            if (ch < c4n) {
                *(float4 *)(a.d_reg + m * a.C + 4 * ch) = dr[u];
                *(float4 *)(a.d_cls + m * a.C + 4 * ch) = dc[u];
            }
        }
    }
    // the workgroup's eight wavefronts are added in wavefront order through LDS, one output row j at a time
    __shared__ __attribute__((aligned(16))) float red[kPredWaves][512];
    __shared__ float redb[kPredWaves][16];
    const int wl = threadIdx.x >> 6;
    const int n_wg = a.n_waves / kPredWaves;
    float *pw = a.partial + (long long)blockIdx.x * nout * a.C;
#pragma unroll
    for (int j = 0; j < NG * 8; ++j) {
        if (j < nout) { // block-uniform
#pragma unroll
            for (int u = 0; u < CPL; ++u) {
                const int ch = lane + 64 * u;
                if (ch < c4n) *(float4 *)&red[wl][4 * ch] = dw[j][u];
            }
            __syncthreads();
            for (int c = threadIdx.x; c < a.C; c += 64 * kPredWaves) {
                float sum = 0.0f;
#pragma unroll
                for (int q = 0; q < kPredWaves; ++q) sum += red[q][c];
                pw[(long long)j * a.C + c] = sum;
            }
            __syncthreads();
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < NG * 8; ++j) redb[wl][j] = db[j];
    }
    __syncthreads();
    if (threadIdx.x < NG * 8) {
        float sum = 0.0f;
#pragma unroll
        for (int q = 0; q < kPredWaves; ++q) sum += redb[q][threadIdx.x];
        a.partial[(long long)n_wg * nout * a.C + (long long)blockIdx.x * 16 + threadIdx.x] = sum;
    }
}

// dw[j][c] = sum over workgroups of the partials; db likewise.  16 elements x 16 segments of the workgroup range per block:
// every thread adds its segment front to back, the 16 segment sums are added in segment order -- a fixed order.
__global__ __launch_bounds__(256) void k_pred_bwd_final(const float *partial, int n_wg, int nout, int C, float *dw_reg,
                                                        float *dw_obj, float *dw_cls, float *db_reg, float *db_obj, float *db_cls)
{
    __shared__ float seg_sum[16][17];
    const int el = threadIdx.x & 15, seg = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + el;
    const int per = nout * C, total = per + nout;
    const int len = (n_wg + 15) / 16, w0 = seg * len, w1 = w0 + len < n_wg ? w0 + len : n_wg;
    float s = 0.0f;
    if (e < per) {
        int wv = w0;
        for (; wv + 4 <= w1; wv += 4) {
            float t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] = partial[(long long)(wv + u) * per + e];
#pragma unroll
            for (int u = 0; u < 4; ++u) s += t[u];
        }
        for (; wv < w1; ++wv) s += partial[(long long)wv * per + e];
    } else if (e < total) {
        const float *pb = partial + (long long)n_wg * per;
        for (int wv = w0; wv < w1; ++wv) s += pb[(long long)wv * 16 + (e - per)];
    }
    seg_sum[seg][el] = s;
    __syncthreads();
    if (seg == 0 && e < total) {
        float t = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += seg_sum[q][el];
        if (e < per) {
            const int j = e / C, c = e - j * C;
            if (j < 4) dw_reg[j * C + c] = t;
            else if (j == 4) dw_obj[c] = t;
            else dw_cls[(j - 5) * C + c] = t;
        } else {
            const int j = e - per;
            if (j < 4) db_reg[j] = t;
            else if (j == 4) db_obj[0] = t;
            else db_cls[j - 5] = t;
        }
    }
}

inline int pred_waves(long long M)
{
    long long wg = (M + 8 * kPredWaves - 1) / (8 * kPredWaves); // at least ~8 rows per wavefront
    if (wg > kPredMaxWg) wg = kPredMaxWg;
    if (wg < 1) wg = 1;
    return (int)wg * kPredWaves;
}

} // namespace

extern "C" {

int64_t frlw_pred_bwd_scratch_floats(int64_t M, int C, int nc)
{
    const int n_wg = pred_waves(M) / kPredWaves;
    return (int64_t)n_wg * (5 + nc) * C + (int64_t)n_wg * 16;
}

int frlw_pred_fwd(const float *reg_feat, const float *cls_feat, int64_t M, int C, int nc, const float *w_reg,
                  const float *b_reg, const float *w_obj, const float *b_obj, const float *w_cls, const float *b_cls,
                  float *out, frlw_stream_t stream)
{
    (void)hipGetLastError();
    if (!reg_feat || !cls_feat || !w_reg || !b_reg || !w_obj || !b_obj || !w_cls || !b_cls || !out || M < 1) return FRLW_ERR_ARG;
    if (C < 4 || (C & 3) || C > 512 || nc < 1 || nc > 11) return FRLW_ERR_UNSUPPORTED;
    PredArgs a = {};
    a.reg_feat = reg_feat; a.cls_feat = cls_feat; a.w_reg = w_reg; a.w_obj = w_obj; a.w_cls = w_cls;
    a.b_reg = b_reg; a.b_obj = b_obj; a.b_cls = b_cls; a.M = M; a.C = C; a.nc = nc; a.out = out;
    a.n_waves = pred_waves(M);
    const dim3 grid(a.n_waves / kPredWaves), block(64 * kPredWaves);
    hipStream_t s = (hipStream_t)stream;
    const bool two_c = C > 256, two_g = 5 + nc > 8;
    if (!two_c && !two_g) hipLaunchKernelGGL((k_pred_fwd<1, 1>), grid, block, 0, s, a);
    else if (two_c && !two_g) hipLaunchKernelGGL((k_pred_fwd<2, 1>), grid, block, 0, s, a);
    else if (!two_c) hipLaunchKernelGGL((k_pred_fwd<1, 2>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_pred_fwd<2, 2>), grid, block, 0, s, a);
    return hipGetLastError() == hipSuccess ? FRLW_OK : FRLW_ERR_HIP;
}

int frlw_pred_bwd(const float *reg_feat, const float *cls_feat, const float *dout, int64_t M, int C, int nc,
                  const float *w_reg, const float *w_obj, const float *w_cls, float *d_reg_feat, float *d_cls_feat,
                  float *dw_reg, float *db_reg, float *dw_obj, float *db_obj, float *dw_cls, float *db_cls, float *scratch,
                  int64_t scratch_floats, frlw_stream_t stream)
{
    (void)hipGetLastError();
    if (!reg_feat || !cls_feat || !dout || !w_reg || !w_obj || !w_cls || !d_reg_feat || !d_cls_feat || !dw_reg || !db_reg ||
        !dw_obj || !db_obj || !dw_cls || !db_cls || !scratch || M < 1)
        return FRLW_ERR_ARG;
    if (C < 4 || (C & 3) || C > 512 || nc < 1 || nc > 11) return FRLW_ERR_UNSUPPORTED;
    if (scratch_floats < frlw_pred_bwd_scratch_floats(M, C, nc)) return FRLW_ERR_WORKSPACE;
    PredArgs a = {};
    a.reg_feat = reg_feat; a.cls_feat = cls_feat; a.w_reg = w_reg; a.w_obj = w_obj; a.w_cls = w_cls;
    a.M = M; a.C = C; a.nc = nc; a.dout = dout; a.d_reg = d_reg_feat; a.d_cls = d_cls_feat; a.partial = scratch;
    a.n_waves = pred_waves(M);
    const dim3 grid(a.n_waves / kPredWaves), block(64 * kPredWaves);
    hipStream_t s = (hipStream_t)stream;
    const bool two_c = C > 256, two_g = 5 + nc > 8;
    if (!two_c && !two_g) hipLaunchKernelGGL((k_pred_bwd<1, 1>), grid, block, 0, s, a);
    else if (two_c && !two_g) hipLaunchKernelGGL((k_pred_bwd<2, 1>), grid, block, 0, s, a);
    else if (!two_c) hipLaunchKernelGGL((k_pred_bwd<1, 2>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_pred_bwd<2, 2>), grid, block, 0, s, a);
    const int nout = 5 + nc, total = nout * C + nout;
    hipLaunchKernelGGL(k_pred_bwd_final, dim3((total + 15) / 16), dim3(256), 0, s, scratch, a.n_waves / kPredWaves, nout, C, dw_reg, dw_obj,
                       dw_cls, db_reg, db_obj, db_cls);
    return hipGetLastError() == hipSuccess ? FRLW_OK : FRLW_ERR_HIP;
}

} // extern "C"

// ---- SPP max-pools of the train step ---------------------------------------------------------------------------------
// SPPBottleneck (network_blocks.py:139-151): cat[x, maxpool5(x), maxpool9(x), maxpool13(x)] (stride 1, "same" padding) and
// its gradient, NHWC.  ATen's channels_last max-pool backward costs 0.11 ms per pool on these 8 x 10 maps (measured); here
// the maps of 64 channels sit in LDS, every output scans its window row-major with torch's update rule
// (`v > best || isnan(v)`: the FIRST maximum wins a tie) and remembers the arg-max pixel; the backward GATHERS (every input
// pixel looks at the outputs whose window holds it, k = 5, 9, 13, row-major) -- a fixed summation order, no float atomics.
namespace {

// channels per workgroup: the widest of 64 / 32 / 16 whose maps (backward: 3 x (float + uint16) per element) fit the LDS AND
// that still gives the chip 1024 workgroups; otherwise the narrowest that fits.  (The window scans are chains of dependent
// LDS reads: with 64 channels the 8 x 10 x 256 maps of a batch of 64 were 256 workgroups = one wavefront per SIMD, 290 us.)
inline int spp_train_ch(int HW, int per_elem_bytes, int C, int B)
{
    int fit = 0;
    for (int ch = 64; ch >= 16; ch >>= 1) {
        if ((size_t)HW * ch * per_elem_bytes > 144 * 1024) continue;
        fit = ch;
        if ((long long)((C + ch - 1) / ch) * B >= 1024) return ch;
    }
    return fit;
}

__global__ __launch_bounds__(256) void k_spp_train_fwd(const float *x, int H, int W, int C, int ch, float *out, uint16_t *arg)
{
    extern __shared__ float sp_x[]; // [H * W][ch]
    const int b = blockIdx.y, c0 = blockIdx.x * ch, HW = H * W;
    const int cl = threadIdx.x % ch, prow = threadIdx.x / ch, pstep = 256 / ch;
    const bool c_ok = c0 + cl < C;
    for (int p = prow; p < HW; p += pstep) sp_x[p * ch + cl] = c_ok ? x[((long long)b * HW + p) * C + c0 + cl] : 0.0f;
    __syncthreads();
    if (!c_ok) return;
    for (int p = prow; p < HW; p += pstep) {
        const int y = p / W, xx = p - y * W;
        float *o = out + ((long long)b * HW + p) * 4 * C + c0 + cl;
        o[0] = sp_x[p * ch + cl];
#pragma unroll
        for (int ki = 0; ki < 3; ++ki) {
            const int r = 2 + 2 * ki; // window radius 2, 4, 6
            float best = -INFINITY;
            int at = -1;
            for (int yy = y - r > 0 ? y - r : 0; yy <= (y + r < H - 1 ? y + r : H - 1); ++yy)
                for (int xc = xx - r > 0 ? xx - r : 0; xc <= (xx + r < W - 1 ? xx + r : W - 1); ++xc) {
                    const float v = sp_x[(yy * W + xc) * ch + cl];
                    if (v > best || v != v || at < 0) { best = v; at = yy * W + xc; } // (at < 0: the first element, like ATen)
                }
            o[(ki + 1) * C] = best;
            arg[(((long long)b * HW + p) * 3 + ki) * C + c0 + cl] = (uint16_t)at;
        }
    }
}

__global__ __launch_bounds__(256) void k_spp_train_bwd(const float *g, const uint16_t *arg, int H, int W, int C, int ch, float *dx)
{
    extern __shared__ float sp_lds[];
    const int b = blockIdx.y, c0 = blockIdx.x * ch, HW = H * W;
    float *sg = sp_lds;                               // [3][H * W][ch] gradients of the three pools
    uint16_t *sa = (uint16_t *)(sg + 3 * HW * ch);    // [3][H * W][ch] their arg-max pixels
    const int cl = threadIdx.x % ch, prow = threadIdx.x / ch, pstep = 256 / ch;
    const bool c_ok = c0 + cl < C;
    for (int p = prow; p < HW; p += pstep)
#pragma unroll
        for (int ki = 0; ki < 3; ++ki) {
            sg[(ki * HW + p) * ch + cl] = c_ok ? g[((long long)b * HW + p) * 4 * C + (ki + 1) * C + c0 + cl] : 0.0f;
            sa[(ki * HW + p) * ch + cl] = c_ok ? arg[(((long long)b * HW + p) * 3 + ki) * C + c0 + cl] : (uint16_t)0xffff;
        }
    __syncthreads();
    if (!c_ok) return;
    for (int p = prow; p < HW; p += pstep) {
        const int y = p / W, xx = p - y * W;
        float s = g[((long long)b * HW + p) * 4 * C + c0 + cl]; // the identity branch
#pragma unroll
        for (int ki = 0; ki < 3; ++ki) {
            const int r = 2 + 2 * ki;
            for (int yy = y - r > 0 ? y - r : 0; yy <= (y + r < H - 1 ? y + r : H - 1); ++yy)
                for (int xc = xx - r > 0 ? xx - r : 0; xc <= (xx + r < W - 1 ? xx + r : W - 1); ++xc) {
                    const int q = yy * W + xc;
                    if (sa[(ki * HW + q) * ch + cl] == (uint16_t)p) s += sg[(ki * HW + q) * ch + cl];
                }
        }
        dx[((long long)b * HW + p) * C + c0 + cl] = s;
    }
}

} // namespace

extern "C" {

int frlw_spp_train_fwd(const float *x, int B, int H, int W, int C, float *out, uint16_t *argmax, frlw_stream_t stream)
{
    (void)hipGetLastError();
    if (!x || !out || !argmax || B < 1 || H < 1 || W < 1 || C < 1) return FRLW_ERR_ARG;
    const int ch = spp_train_ch(H * W, 3 * 6, C, B); // the same chunking as the backward, which needs the most LDS
    if (H * W > 65535 || ch == 0) return FRLW_ERR_UNSUPPORTED;
    const size_t lds = (size_t)H * W * ch * sizeof(float);
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)k_spp_train_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_spp_train_fwd, dim3((C + ch - 1) / ch, B), dim3(256), lds, (hipStream_t)stream, x, H, W, C, ch, out, argmax);
    return hipGetLastError() == hipSuccess ? FRLW_OK : FRLW_ERR_HIP;
}

int frlw_spp_train_bwd(const float *dout, const uint16_t *argmax, int B, int H, int W, int C, float *dx, frlw_stream_t stream)
{
    (void)hipGetLastError();
    if (!dout || !argmax || !dx || B < 1 || H < 1 || W < 1 || C < 1) return FRLW_ERR_ARG;
    const int ch = spp_train_ch(H * W, 3 * 6, C, B);
    if (H * W > 65535 || ch == 0) return FRLW_ERR_UNSUPPORTED;
    const size_t lds = (size_t)3 * H * W * ch * (sizeof(float) + sizeof(uint16_t));
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)k_spp_train_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_spp_train_bwd, dim3((C + ch - 1) / ch, B), dim3(256), lds, (hipStream_t)stream, dout, argmax, H, W, C, ch, dx);
    return hipGetLastError() == hipSuccess ? FRLW_OK : FRLW_ERR_HIP;
}

} // extern "C"
