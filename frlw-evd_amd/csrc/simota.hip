// simota.hip -- SimOTA label assignment of the YOLOX training branch for a whole batch, without host
// round trips (reference: core/yolox/models/yolo_head.py:482-584 get_assignments, :586-669
// get_in_boxes_info, :671-707 dynamic_k_matching, core/yolox/utils/boxes.py:79-102 bboxes_iou).
//
// The reference walks the batch in Python with an `.item()` sync per image and per ground-truth box;
// here the three steps are three launches over the batch:
//   k_candidates  (image, anchor)     candidate = centre inside any GT box OR any GT centre square
//   k_rows        (image, GT) per WG  IoU + cost row, k = clamp(int(sum of top-10 IoU), 1), k cheapest anchors
//   k_resolve     (image, anchor)     an anchor claimed by several GTs goes to the cheapest GT (over all GTs)
// Dtypes follow the reference: labels are float64, predictions float32; arithmetic on the prediction side
// is done in float32 first and promoted, IoU / cost are float64.

#include "frlw_common.h"

using namespace frlw;

namespace {

constexpr int kRowThreads = 256;

struct SimotaWs {
    double *cost;   // (B, G, A)
    double *iou;    // (B, G, A)
    int *count;     // (B, A) number of GTs that picked the anchor
    int *picker;    // (B, A) one GT that picked it (exact when count == 1)
    uint8_t *cand;  // (B, A)
    int *nlabel;    // (B)
};

__host__ __device__ inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

inline size_t ws_layout(int B, int A, int G, SimotaWs *w, uint8_t *base)
{
    size_t off = 0;
    const size_t row = (size_t)B * G * A * sizeof(double);
    if (w) w->cost = (double *)(base + off);
    off += align256(row);
    if (w) w->iou = (double *)(base + off);
    off += align256(row);
    if (w) w->count = (int *)(base + off);
    off += align256((size_t)B * A * sizeof(int));
    if (w) w->picker = (int *)(base + off);
    off += align256((size_t)B * A * sizeof(int));
    if (w) w->cand = base + off;
    off += align256((size_t)B * A);
    if (w) w->nlabel = (int *)(base + off);
    off += align256((size_t)B * sizeof(int));
    return off;
}

// number of labels of an image: rows whose five fields sum to > 0 (yolo_head.py:330); the first
// `n` rows are then taken as the boxes (:349-350)
__global__ void k_nlabel(const double *labels, int B, int G, int *nlabel, int *num_fg)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int n = 0;
    for (int g = 0; g < G; ++g) {
        const double *l = labels + ((size_t)b * G + g) * 5;
        if (l[0] + l[1] + l[2] + l[3] + l[4] > 0.0) ++n;
    }
    nlabel[b] = n;
    num_fg[b] = 0;
}

__device__ __forceinline__ void anchor_centre(const float *xs, const float *ys, const float *st, int a, double &xc,
                                              double &yc, float &s)
{
    s = st[a];
    xc = (double)(xs[a] * s + 0.5f * s); // float32 on the anchor side (yolo_head.py:600-611), then promoted
    yc = (double)(ys[a] * s + 0.5f * s);
}

__device__ __forceinline__ void in_box_center(const double *gt, double xc, double yc, float s, float radius,
                                              bool &in_box, bool &in_center)
{
    const double cx = gt[1], cy = gt[2], w = gt[3], h = gt[4];
    const double l = xc - (cx - 0.5 * w), t = yc - (cy - 0.5 * h), r = (cx + 0.5 * w) - xc, b = (cy + 0.5 * h) - yc;
    in_box = fmin(fmin(l, t), fmin(r, b)) > 0.0; // :626-629
    const double rr = (double)(radius * s);      // radius * stride in float32 (:633)
    const double cl = xc - (cx - rr), ct = yc - (cy - rr), cr = (cx + rr) - xc, cb = (cy + rr) - yc;
    in_center = fmin(fmin(cl, ct), fmin(cr, cb)) > 0.0; // :650-653
}

__global__ void k_candidates(const double *labels, const float *xs, const float *ys, const float *st, int B, int A,
                             int G, float radius, SimotaWs w)
{
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= (long long)B * A) return;
    const int b = (int)(i / A), a = (int)(i - (long long)b * A);
    double xc, yc;
    float s;
    anchor_centre(xs, ys, st, a, xc, yc, s);
    bool any = false;
    const int n = w.nlabel[b];
    for (int g = 0; g < n; ++g) {
        bool ib, ic;
        in_box_center(labels + ((size_t)b * G + g) * 5, xc, yc, s, radius, ib, ic);
        any |= ib | ic; // :656-657 is_in_boxes_anchor_or_center
    }
    w.cand[i] = any ? 1 : 0;
    w.count[i] = 0;
    w.picker[i] = -1;
}

__device__ __forceinline__ float sigmoid_f(float v) { return 1.0f / (1.0f + expf(-v)); }

// block-wide arg-best over the row held in LDS; ties go to the lower anchor index.  BEST_IS_MAX selects
// maximum (IoU) or minimum (cost).  Returns the index (or -1 when every entry is exhausted) to all threads.
template <bool BEST_IS_MAX>
__device__ int block_argbest(const double *row, int A, double *red_v, int *red_i, double &best)
{
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const double worst = BEST_IS_MAX ? -1.0e300 : 1.0e300;
    double v = worst;
    int idx = -1;
    for (int a = t; a < A; a += kRowThreads) {
        const double x = row[a];
        const bool better = BEST_IS_MAX ? (x > v) : (x < v);
        if (better) { v = x; idx = a; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(v, off);
        const int oi = __shfl_xor(idx, off);
        const bool better = oi >= 0 && (idx < 0 || (BEST_IS_MAX ? (ov > v) : (ov < v)) || (ov == v && oi < idx));
        if (better) { v = ov; idx = oi; }
    }
    if (lane == 0) { red_v[wv] = v; red_i[wv] = idx; }
    __syncthreads();
    v = red_v[0];
    idx = red_i[0];
#pragma unroll
    for (int k = 1; k < kRowThreads / 64; ++k) {
        const double ov = red_v[k];
        const int oi = red_i[k];
        const bool better = oi >= 0 && (idx < 0 || (BEST_IS_MAX ? (ov > v) : (ov < v)) || (ov == v && oi < idx));
        if (better) { v = ov; idx = oi; }
    }
    __syncthreads();
    best = v;
    return idx;
}

// One WG per (image, GT).  Dynamic LDS: two float64 rows of A entries.
__global__ __launch_bounds__(kRowThreads) void k_rows(const float *preds, const double *labels, const float *xs,
                                                      const float *ys, const float *st, int B, int A, int G, int nc,
                                                      float radius, SimotaWs w)
{
    extern __shared__ double lds_rows[];
    double *liou = lds_rows, *lcost = lds_rows + A;
    __shared__ double red_v[kRowThreads / 64];
    __shared__ int red_i[kRowThreads / 64];
    const int b = blockIdx.x / G, g = blockIdx.x - b * G;
    if (g >= w.nlabel[b]) return;
    const int t = threadIdx.x;
    const double *gt = labels + ((size_t)b * G + g) * 5;
    const int gcls = (int)gt[0];
    const double gcx = gt[1], gcy = gt[2], gw = gt[3], gh = gt[4];
    const double garea = gw * gh; // torch.prod(bboxes_a[:, 2:], 1)
    const int P = 5 + nc;
    const size_t rowoff = ((size_t)b * G + g) * A;
    for (int a = t; a < A; a += kRowThreads) {
        double iou = -1.0, cost = 1.0e300;
        if (w.cand[(size_t)b * A + a]) {
            const float *p = preds + ((size_t)b * A + a) * P;
            // prediction side in float32 (boxes.py:92-99 on a float32 tensor), then promoted
            const float px = p[0], py = p[1], pw = p[2], ph = p[3];
            const double ptlx = (double)(px - pw / 2.0f), ptly = (double)(py - ph / 2.0f);
            const double pbrx = (double)(px + pw / 2.0f), pbry = (double)(py + ph / 2.0f);
            const double parea = (double)(pw * ph);
            const double tlx = fmax(gcx - gw / 2.0, ptlx), tly = fmax(gcy - gh / 2.0, ptly);
            const double brx = fmin(gcx + gw / 2.0, pbrx), bry = fmin(gcy + gh / 2.0, pbry);
            const double en = (tlx < brx && tly < bry) ? 1.0 : 0.0;
            const double ai = (brx - tlx) * (bry - tly) * en;
            iou = ai / (garea + parea - ai);
            // class cost: BCE(sqrt(sigmoid(cls) * sigmoid(obj)), one-hot) summed over classes, float32
            // (yolo_head.py:541-551); torch clamps each log at -100
            const float so = sigmoid_f(p[4]);
            float cc = 0.0f;
            for (int c = 0; c < nc; ++c) {
                const float q = sqrtf(sigmoid_f(p[5 + c]) * so);
                const float lg = c == gcls ? fmaxf(logf(q), -100.0f) : fmaxf(logf(1.0f - q), -100.0f);
                cc = cc - lg;
            }
            double xc, yc;
            float s;
            anchor_centre(xs, ys, st, a, xc, yc, s);
            bool ib, ic;
            in_box_center(gt, xc, yc, s, radius, ib, ic);
            cost = (double)cc + 3.0 * (-log(iou + 1e-8)) + ((ib && ic) ? 0.0 : 100000.0); // :553-557
        }
        liou[a] = iou;
        lcost[a] = cost;
        w.iou[rowoff + a] = iou;
        w.cost[rowoff + a] = cost;
    }
    __syncthreads();
    // dynamic k = clamp(int(sum of the top-10 candidate IoUs), min 1)  (:676-679)
    double sum = 0.0;
    for (int r = 0; r < 10; ++r) {
        double v;
        const int idx = block_argbest<true>(liou, A, red_v, red_i, v);
        if (idx < 0 || v < 0.0) break; // fewer than 10 candidates
        sum += v;
        if (t == 0) liou[idx] = -2.0;
        __syncthreads();
    }
    int k = (int)sum;
    if (k < 1) k = 1;
    for (int r = 0; r < k; ++r) { // the k cheapest candidates (:681-684)
        double v;
        const int idx = block_argbest<false>(lcost, A, red_v, red_i, v);
        if (idx < 0 || v >= 1.0e300) break;
        if (t == 0) {
            lcost[idx] = 2.0e300;
            atomicAdd(&w.count[(size_t)b * A + idx], 1);
            atomicMax(&w.picker[(size_t)b * A + idx], g);
        }
        __syncthreads();
    }
}

__global__ void k_resolve(int B, int A, int G, SimotaWs w, uint8_t *fg, int *matched_gt, double *matched_iou,
                          int *num_fg)
{
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= (long long)B * A) return;
    const int b = (int)(i / A), a = (int)(i - (long long)b * A);
    const int c = w.count[i];
    int g = -1;
    if (c == 1) {
        g = w.picker[i];
    } else if (c > 1) { // argmin of the cost over ALL GTs of the image (:688-692)
        double best = 1.0e301;
        const int n = w.nlabel[b];
        for (int q = 0; q < n; ++q) {
            const double v = w.cost[((size_t)b * G + q) * A + a];
            if (v < best) { best = v; g = q; }
        }
    }
    fg[i] = g >= 0 ? 1 : 0;
    matched_gt[i] = g;
    matched_iou[i] = g >= 0 ? w.iou[((size_t)b * G + g) * A + a] : 0.0;
    if (g >= 0) atomicAdd(&num_fg[b], 1);
}

} // namespace

extern "C" {

size_t frlw_simota_workspace_bytes(int B, int A, int G)
{
    if (B <= 0 || A <= 0 || G <= 0) return 0;
    return ws_layout(B, A, G, nullptr, nullptr);
}

int frlw_simota_assign(const float *preds, const double *labels, const float *x_shifts, const float *y_shifts,
                       const float *strides, int B, int A, int G, int num_classes, float radius, uint8_t *fg,
                       int32_t *matched_gt, double *matched_iou, int32_t *num_fg, int32_t *nlabel, void *workspace,
                       size_t workspace_bytes, frlw_stream_t stream)
{
    (void)hipGetLastError(); // clear stale errors of other libraries in the process
    if (!preds || !labels || !x_shifts || !y_shifts || !strides || !fg || !matched_gt || !matched_iou || !num_fg ||
        !workspace || B <= 0 || A <= 0 || G <= 0 || num_classes <= 0)
        return FRLW_ERR_ARG;
    const size_t lds = (size_t)2 * A * sizeof(double);
    if (lds > 150 * 1024) return FRLW_ERR_UNSUPPORTED; // two float64 rows must fit the 160 KB LDS
    SimotaWs w;
    if (ws_layout(B, A, G, &w, (uint8_t *)workspace) > workspace_bytes) return FRLW_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_nlabel, dim3((B + 63) / 64), dim3(64), 0, s, labels, B, G, w.nlabel, num_fg);
    const long long n = (long long)B * A;
    hipLaunchKernelGGL(k_candidates, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, labels, x_shifts, y_shifts,
                       strides, B, A, G, radius, w);
    if (lds > 48 * 1024) // above the default dynamic-LDS limit
        HIP_TRY(hipFuncSetAttribute((const void *)k_rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_rows, dim3(B * G), dim3(kRowThreads), lds, s, preds, labels, x_shifts, y_shifts, strides, B,
                       A, G, num_classes, radius, w);
    hipLaunchKernelGGL(k_resolve, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, B, A, G, w, fg, matched_gt,
                       matched_iou, num_fg);
    if (nlabel) HIP_TRY(hipMemcpyAsync(nlabel, w.nlabel, sizeof(int) * B, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipGetLastError());
    return FRLW_OK;
}

} // extern "C"
