// elementwise.hip -- the harness' stand-alone elementwise steps: leaky_transform
// (generate_taf.py:69-76), uint8 truncation (generate_taf.py:232), nearest resize
// (generate_eventvolume.py:149).  HBM-bound streaming kernels, grid-stride.

#include "frlw_common.h"

using namespace frlw;

namespace {

__global__ void k_leaky(const float *in, long long n, float *out_f32, uint8_t *out_u8)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        float v = leaky_f(in[i]);
        if (out_f32) out_f32[i] = v;
        if (out_u8) out_u8[i] = f32_to_u8(v);
    }
}

__global__ void k_quantize(const float *in, long long n, int clip255, uint8_t *out)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        float v = in[i];
        if (clip255 && v > 255.0f) v = 255.0f;
        out[i] = f32_to_u8(v);
    }
}

template <typename T>
__global__ void k_resize_nearest(const T *in, int C, int H, int W, int Ho, int Wo, float sh, float sw,
                                 T *out)
{
    const long long total = (long long)C * Ho * Wo;
    for (long long o = blockIdx.x * (long long)blockDim.x + threadIdx.x; o < total;
         o += (long long)gridDim.x * blockDim.x) {
        const int xo = (int)(o % Wo);
        const int yo = (int)((o / Wo) % Ho);
        const int c = (int)(o / ((long long)Wo * Ho));
        int ys = (int)floorf((float)yo * sh);
        int xs = (int)floorf((float)xo * sw);
        if (ys > H - 1) ys = H - 1;
        if (xs > W - 1) xs = W - 1;
        out[o] = in[((long long)c * H + ys) * W + xs];
    }
}

// Sample transform of the training loader (data/dataset.py:217-231): nearest resize of the (C, H, W) uint8 volume to
// (Hr, Wr) -- torch's nearest: src = min(floor(dst * float(in) / out), in - 1) -- then / 255, crop at (y0, x0), flip.
// par[b] = {Hr, Wr, y0, x0, flip}.  One thread per output element, x fastest.
__global__ void k_sample_transform(const uint8_t *in, int B, int C, int H, int W, const int *par, float *out)
{
    const long long total = (long long)B * C * H * W;
    for (long long o = blockIdx.x * (long long)blockDim.x + threadIdx.x; o < total;
         o += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(o % W), y = (int)((o / W) % H);
        const long long bc = o / ((long long)W * H);
        const int b = (int)(bc / C);
        const int *p = par + 5 * b;
        const int Hr = p[0], Wr = p[1], y0 = p[2], x0 = p[3], flip = p[4];
        const int xs = flip ? W - 1 - x : x; // img[:, :, ::-1] after the crop
        const float sh = (float)H / (float)Hr, sw = (float)W / (float)Wr;
        int sy = (int)floorf((float)(y + y0) * sh), sx = (int)floorf((float)(xs + x0) * sw);
        if (sy > H - 1) sy = H - 1;
        if (sx > W - 1) sx = W - 1;
        out[o] = (float)in[(bc * H + sy) * W + sx] / 255.0f;
    }
}

// Evaluator hand-off (evaluate/evaluator.py:56-63 transform_dt + evaluate/src/io/box_filtering.py:17-39): detection
// rows [cx, cy, w, h, class, score] in detector pixels -> [t, x, y, w, h, class, score, 0] in sensor pixels (float32
// like the reference's tensors) and the Prophesee filter mask ts > skip, w^2 + h^2 >= diag^2, w >= min_w, h >= min_h.
__global__ void k_eval_transform_dt(const float *dets, const int *img_of_row, const long long *ts, long long n, float rw,
                                    float rh, float skip_ts, float diag2, float min_w, float min_h, float *out,
                                    uint8_t *keep)
{
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *d = dets + 6 * i;
    const float t = (float)ts[img_of_row[i]];
    const float x = (d[0] - d[2] / 2.0f) * rw, y = (d[1] - d[3] / 2.0f) * rh, w = d[2] * rw, h = d[3] * rh;
    float *o = out + 8 * i;
    o[0] = t; o[1] = x; o[2] = y; o[3] = w; o[4] = h; o[5] = d[4]; o[6] = d[5]; o[7] = 0.0f;
    keep[i] = (t > skip_ts && w * w + h * h >= diag2 && w >= min_w && h >= min_h) ? 1 : 0;
}

} // namespace

extern "C" {

int frlw_eval_transform_dt(const float *dets, const int32_t *img_of_row, const int64_t *timestamps, int64_t n, float rw,
                           float rh, float skip_ts, float min_diag_sq, float min_w, float min_h, float *out,
                           uint8_t *keep, frlw_stream_t stream)
{
    if (n < 0 || (n > 0 && (!dets || !img_of_row || !timestamps || !out || !keep))) return FRLW_ERR_ARG;
    if (n == 0) return FRLW_OK;
    (void)hipGetLastError(); // stale errors of other libraries
    hipLaunchKernelGGL(k_eval_transform_dt, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dets,
                       img_of_row, (const long long *)timestamps, (long long)n, rw, rh, skip_ts, min_diag_sq, min_w,
                       min_h, out, keep);
    HIP_TRY(hipGetLastError());
    return FRLW_OK;
}

int frlw_sample_transform_u8(const uint8_t *in, int B, int C, int H, int W, const int32_t *params, float *out,
                             frlw_stream_t stream)
{
    if (!in || !params || !out || B < 0 || C < 1 || H < 1 || W < 1) return FRLW_ERR_ARG;
    const long long n = (long long)B * C * H * W;
    if (n == 0) return FRLW_OK;
    (void)hipGetLastError(); // stale errors of other libraries
    hipLaunchKernelGGL(k_sample_transform, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, in, B, C, H, W,
                       params, out);
    HIP_TRY(hipGetLastError());
    return FRLW_OK;
}

int frlw_leaky_transform(const float *in, int64_t n, float *out_f32, uint8_t *out_u8,
                         frlw_stream_t stream)
{
    if (!in || (!out_f32 && !out_u8) || n < 0) return FRLW_ERR_ARG;
    if (n == 0) return FRLW_OK;
    (void)hipGetLastError(); // stale errors of other libraries
    hipLaunchKernelGGL(k_leaky, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, in,
                       (long long)n, out_f32, out_u8);
    HIP_TRY(hipGetLastError());
    return FRLW_OK;
}

int frlw_quantize_u8(const float *in, int64_t n, int clip255, uint8_t *out, frlw_stream_t stream)
{
    if (!in || !out || n < 0) return FRLW_ERR_ARG;
    if (n == 0) return FRLW_OK;
    (void)hipGetLastError(); // stale errors of other libraries
    hipLaunchKernelGGL(k_quantize, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, in,
                       (long long)n, clip255, out);
    HIP_TRY(hipGetLastError());
    return FRLW_OK;
}

int frlw_resize_nearest_f32(const float *in, int C, int H, int W, int Ho, int Wo, float *out,
                            frlw_stream_t stream)
{
    if (!in || !out || C <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return FRLW_ERR_ARG;
    const long long total = (long long)C * Ho * Wo;
    (void)hipGetLastError(); // stale errors of other libraries
    hipLaunchKernelGGL(k_resize_nearest<float>, dim3(grid_for(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, in, C, H, W, Ho, Wo, (float)H / (float)Ho,
                       (float)W / (float)Wo, out);
    HIP_TRY(hipGetLastError());
    return FRLW_OK;
}

int frlw_resize_nearest_u8(const uint8_t *in, int C, int H, int W, int Ho, int Wo, uint8_t *out,
                           frlw_stream_t stream)
{
    if (!in || !out || C <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return FRLW_ERR_ARG;
    const long long total = (long long)C * Ho * Wo;
    (void)hipGetLastError(); // stale errors of other libraries
    hipLaunchKernelGGL(k_resize_nearest<uint8_t>, dim3(grid_for(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, in, C, H, W, Ho, Wo, (float)H / (float)Ho,
                       (float)W / (float)Wo, out);
    HIP_TRY(hipGetLastError());
    return FRLW_OK;
}

} // extern "C"
