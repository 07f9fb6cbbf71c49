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
__global__ __launch_bounds__(64) void k_nlabel(const double *labels, int B, int G, int *nlabel, int *num_fg)
{
    const int b = blockIdx.x, lane = threadIdx.x; // one wavefront per image
    int n = 0;
    for (int g = lane; g < G; g += 64) {
        const double *l = labels + ((size_t)b * G + g) * 5;
        if (l[0] + l[1] + l[2] + l[3] + l[4] > 0.0) ++n;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off);
    if (lane == 0) {
        nlabel[b] = n;
        num_fg[b] = 0;
    }
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


// ---- the loss of a batch on the raw level outputs --------------------------------------------------------------------------
// (yolo_head.py:237-256 decode, :305-473 get_losses, losses.py:16-36 IOUloss "iou", nn.BCEWithLogitsLoss): decode + cat in
// one launch, the assignment above, then ONE pass over (image, anchor) for the three sums and ONE pass for the gradient of
// the raw outputs -- instead of ~200 small tensor operations and their autograd nodes.

constexpr int kMaxLevels = 4;
constexpr int kLossThreads = 256;

struct LossGeom {
    const float *raw[kMaxLevels]; // (B, h, w, P) rows of a level: cat[reg, obj, cls], no sigmoid
    float *grad[kMaxLevels];
    int w[kMaxLevels];
    int a_off[kMaxLevels + 1];    // first anchor of a level in the concatenated (A) axis
    float stride[kMaxLevels];
    int n_levels, B, A, P;
};

__device__ __forceinline__ int level_of(const LossGeom &g, int a)
{
    int l = 0;
    while (l + 1 < g.n_levels && a >= g.a_off[l + 1]) ++l;
    return l;
}

// preds (B, A, P): xy = (xy + grid) * stride, wh = square(wh) * stride (yolo_head.py:237-256 as this fork has it), the
// rest as is; x_shifts / y_shifts / strides (A) of :254-256.  One thread per output float: both sides coalesced.
__global__ __launch_bounds__(kLossThreads) void k_loss_decode(LossGeom g, float *preds, float *xs, float *ys, float *st)
{
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const long long total = (long long)g.B * g.A * g.P;
    if (i >= total) return;
    const int c = (int)(i % g.P);
    const long long ba = i / g.P;
    const int a = (int)(ba % g.A), b = (int)(ba / g.A);
    const int l = level_of(g, a);
    const int cell = a - g.a_off[l], hw = g.a_off[l + 1] - g.a_off[l];
    const float s = g.stride[l];
    const float gx = (float)(cell % g.w[l]), gy = (float)(cell / g.w[l]);
    float v = g.raw[l][((size_t)b * hw + cell) * g.P + c];
    if (c == 0) v = (v + gx) * s;
    else if (c == 1) v = (v + gy) * s;
    else if (c < 4) v = (v * v) * s;
    preds[i] = v;
    if (b == 0 && c == 0) {
        xs[a] = gx;
        ys[a] = gy;
        st[a] = s;
    }
}

// log_sigmoid of ATen (min(0, x) - log1p(exp(-|x|))) in float32: BCEWithLogitsLoss = (1 - t) * x - log_sigmoid(x)
__device__ __forceinline__ float log_sigmoid_f(float x) { return fminf(0.0f, x) - log1pf(expf(-fabsf(x))); }

struct IouTerm {
    double iou, u, d0, d1, en;
    bool tl_p[2], br_p[2], tl_eq[2], br_eq[2]; // which side max / min took (ties split the gradient like torch.max does)
};

// losses.py:16-36 on (float32 prediction, float64 target): the prediction side's corners and area in float32, then promoted
__device__ __forceinline__ IouTerm iou_term(const float *p, const double *gt)
{
    IouTerm r;
    const float pw2 = p[2] / 2.0f, ph2 = p[3] / 2.0f;
    const double ptl[2] = {(double)(p[0] - pw2), (double)(p[1] - ph2)};
    const double pbr[2] = {(double)(p[0] + pw2), (double)(p[1] + ph2)};
    const double ttl[2] = {gt[1] - gt[3] / 2.0, gt[2] - gt[4] / 2.0};
    const double tbr[2] = {gt[1] + gt[3] / 2.0, gt[2] + gt[4] / 2.0};
    double tl[2], br[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        tl[k] = fmax(ptl[k], ttl[k]);
        br[k] = fmin(pbr[k], tbr[k]);
        r.tl_p[k] = ptl[k] > ttl[k];
        r.tl_eq[k] = ptl[k] == ttl[k];
        r.br_p[k] = pbr[k] < tbr[k];
        r.br_eq[k] = pbr[k] == tbr[k];
    }
    const double area_p = (double)(p[2] * p[3]), area_g = gt[3] * gt[4];
    r.en = ((tl[0] < br[0]) ? 1.0 : 0.0) * ((tl[1] < br[1]) ? 1.0 : 0.0);
    r.d0 = br[0] - tl[0];
    r.d1 = br[1] - tl[1];
    const double area_i = r.d0 * r.d1 * r.en;
    r.u = area_p + area_g - area_i + 1e-16;
    r.iou = area_i / r.u;
    return r;
}

__device__ __forceinline__ int class_of(const double *gt, int nc)
{
    long long c = (long long)gt[0]; // .to(torch.int64).clamp(0, nc - 1)
    return (int)(c < 0 ? 0 : (c > nc - 1 ? nc - 1 : c));
}

// per WG: float64 partial sums {iou, obj, cls} over its (image, anchor) pairs
__global__ __launch_bounds__(kLossThreads) void k_loss_fwd(const float *preds, const double *labels, const uint8_t *fg,
                                                           const int *matched_gt, const double *matched_iou, int B, int A,
                                                           int G, int nc, double *partial)
{
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    double s_iou = 0.0, s_obj = 0.0, s_cls = 0.0;
    if (i < (long long)B * A) {
        const int P = 5 + nc;
        const float *p = preds + (size_t)i * P;
        const bool f = fg[i] != 0;
        const float x = p[4], t = f ? 1.0f : 0.0f;
        s_obj = (double)((1.0f - t) * x - log_sigmoid_f(x)); // float32 elements (yolo_head.py:441-443)
        if (f) {
            const int b = (int)(i / A);
            const double *gt = labels + ((size_t)b * G + matched_gt[i]) * 5;
            const IouTerm r = iou_term(p, gt);
            s_iou = 1.0 - r.iou * r.iou;
            const int cls = class_of(gt, nc);
            const double miou = matched_iou[i];
            for (int c = 0; c < nc; ++c) { // float64 target = one_hot * IoU (:383-385): the element is float64
                const float xc = p[5 + c];
                const double tc = c == cls ? miou : 0.0;
                s_cls += (1.0 - tc) * (double)xc - (double)log_sigmoid_f(xc);
            }
        }
    }
    __shared__ double red[3][kLossThreads / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s_iou += __shfl_xor(s_iou, off);
        s_obj += __shfl_xor(s_obj, off);
        s_cls += __shfl_xor(s_cls, off);
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { red[0][wv] = s_iou; red[1][wv] = s_obj; red[2][wv] = s_cls; }
    __syncthreads();
    if (threadIdx.x < 3) {
        double v = 0.0;
        for (int k = 0; k < kLossThreads / 64; ++k) v += red[threadIdx.x][k];
        partial[(size_t)blockIdx.x * 3 + threadIdx.x] = v;
    }
}

// result: {loss, 5 * loss_iou, loss_obj, loss_cls, num_fg / num_gt, num_fg}  (yolo_head.py:445-473), partials summed in
// WG order: the same bits on every run
__global__ __launch_bounds__(kLossThreads) void k_loss_final(const double *partial, int n_wg, const int *num_fg,
                                                             const int *nlabel, int B, double *result)
{
    __shared__ double red[3][kLossThreads];
    double v[3] = {0.0, 0.0, 0.0};
    const int t = threadIdx.x;
    const int per = (n_wg + kLossThreads - 1) / kLossThreads;
    for (int k = t * per; k < min(n_wg, (t + 1) * per); ++k)
        for (int q = 0; q < 3; ++q) v[q] += partial[(size_t)k * 3 + q];
    for (int q = 0; q < 3; ++q) red[q][t] = v[q];
    __syncthreads();
    if (t == 0) {
        double s[3] = {0.0, 0.0, 0.0};
        for (int k = 0; k < kLossThreads; ++k)
            for (int q = 0; q < 3; ++q) s[q] += red[q][k];
        long long nf = 0, ng = 0;
        for (int b = 0; b < B; ++b) { nf += num_fg[b]; ng += nlabel[b]; }
        const double num_fg_d = (double)(nf < 1 ? 1 : nf);
        const double l_iou = s[0] / num_fg_d, l_obj = s[1] / num_fg_d, l_cls = s[2] / num_fg_d;
        result[0] = 5.0 * l_iou + l_obj + l_cls + 0.0;
        result[1] = 5.0 * l_iou;
        result[2] = l_obj;
        result[3] = l_cls;
        result[4] = num_fg_d / (double)(ng < 1 ? 1 : ng);
        result[5] = num_fg_d;
    }
}

// gradient of the raw level outputs for the upstream gradient of result[0..3] (result[4] carries none)
__global__ __launch_bounds__(kLossThreads) void k_loss_bwd(LossGeom g, const double *labels, const uint8_t *fg,
                                                           const int *matched_gt, const double *matched_iou, int G,
                                                           const double *result, const double *grad_result)
{
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= (long long)g.B * g.A) return;
    const int a = (int)(i % g.A), b = (int)(i / g.A);
    const int l = level_of(g, a);
    const int cell = a - g.a_off[l], hw = g.a_off[l + 1] - g.a_off[l];
    const size_t row = ((size_t)b * hw + cell) * g.P;
    const float *raw = g.raw[l] + row;
    float *out = g.grad[l] + row;
    const int nc = g.P - 5;
    const double nfg = result[5];
    const double g_iou = 5.0 * (grad_result[0] + grad_result[1]) / nfg; // d / d(sum of the IoU terms)
    const double g_obj = (grad_result[0] + grad_result[2]) / nfg;
    const double g_cls = (grad_result[0] + grad_result[3]) / nfg;
    const bool f = fg[i] != 0;
    const float xo = raw[4];
    out[4] = (sigmoid_f(xo) - (f ? 1.0f : 0.0f)) * (float)g_obj; // the objectness term is float32 end to end
    if (!f) {
        out[0] = out[1] = out[2] = out[3] = 0.0f;
        for (int c = 0; c < nc; ++c) out[5 + c] = 0.0f;
        return;
    }
    const double *gt = labels + ((size_t)b * G + matched_gt[i]) * 5;
    const int cls = class_of(gt, nc);
    const double miou = matched_iou[i];
    for (int c = 0; c < nc; ++c) {
        const double tc = c == cls ? miou : 0.0;
        out[5 + c] = (float)(((double)sigmoid_f(raw[5 + c]) - tc) * g_cls);
    }
    const float s = g.stride[l];
    const float gx = (float)(cell % g.w[l]), gy = (float)(cell / g.w[l]);
    const float p[4] = {(raw[0] + gx) * s, (raw[1] + gy) * s, (raw[2] * raw[2]) * s, (raw[3] * raw[3]) * s};
    const IouTerm r = iou_term(p, gt);
    // L = 1 - iou^2, iou = I / U, U = area_p + area_g - I + eps, I = d0 * d1 * en
    const double d_iou = -2.0 * r.iou * g_iou;
    const double area_i = r.d0 * r.d1 * r.en;
    const double d_u = -d_iou * area_i / (r.u * r.u);
    const double d_i = d_iou / r.u - d_u;
    const double dd[2] = {d_i * r.d1 * r.en, d_i * r.d0 * r.en};
    float g_tl[2], g_br[2]; // gradients of the float32 corners (cast where the forward promoted them)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        g_br[k] = (float)(r.br_p[k] ? dd[k] : (r.br_eq[k] ? 0.5 * dd[k] : 0.0));
        g_tl[k] = (float)(r.tl_p[k] ? -dd[k] : (r.tl_eq[k] ? -0.5 * dd[k] : 0.0));
    }
    const float g_area = (float)d_u;
    const float g_px = g_tl[0] + g_br[0], g_py = g_tl[1] + g_br[1];
    const float g_pw = (g_br[0] - g_tl[0]) / 2.0f + g_area * p[3];
    const float g_ph = (g_br[1] - g_tl[1]) / 2.0f + g_area * p[2];
    out[0] = g_px * s;
    out[1] = g_py * s;
    out[2] = (g_pw * s) * (2.0f * raw[2]);
    out[3] = (g_ph * s) * (2.0f * raw[3]);
}

struct LossWs {
    float *xs, *ys, *st; // (A)
    int *num_fg;         // (B)
    int *nlabel;         // (B)
    double *partial;     // (ceil(B * A / kLossThreads), 3)
    uint8_t *simota;
};

inline size_t loss_ws_layout(int B, int A, int G, LossWs *w, uint8_t *base)
{
    size_t off = 0;
    if (w) w->xs = (float *)(base + off);
    off += align256((size_t)A * 4);
    if (w) w->ys = (float *)(base + off);
    off += align256((size_t)A * 4);
    if (w) w->st = (float *)(base + off);
    off += align256((size_t)A * 4);
    if (w) w->num_fg = (int *)(base + off);
    off += align256((size_t)B * 4);
    if (w) w->nlabel = (int *)(base + off);
    off += align256((size_t)B * 4);
    if (w) w->partial = (double *)(base + off);
    off += align256(((size_t)B * A + kLossThreads - 1) / kLossThreads * 3 * sizeof(double));
    if (w) w->simota = base + off;
    off += ws_layout(B, A, G, nullptr, nullptr);
    return off;
}

int loss_geom(const float *const *raw, float *const *grad, const int32_t *h, const int32_t *wd, const float *strides,
              int n_levels, int B, int nc, LossGeom *g)
{
    if (!raw || !h || !wd || !strides || n_levels < 1 || n_levels > kMaxLevels || B <= 0 || nc <= 0) return FRLW_ERR_ARG;
    g->n_levels = n_levels;
    g->B = B;
    g->P = 5 + nc;
    g->a_off[0] = 0;
    for (int l = 0; l < kMaxLevels; ++l) {
        const bool on = l < n_levels;
        if (on && (!raw[l] || h[l] <= 0 || wd[l] <= 0 || (grad && !grad[l]))) return FRLW_ERR_ARG;
        g->raw[l] = on ? raw[l] : nullptr;
        g->grad[l] = on && grad ? grad[l] : nullptr;
        g->w[l] = on ? wd[l] : 1;
        g->stride[l] = on ? strides[l] : 0.0f;
        g->a_off[l + 1] = g->a_off[l] + (on ? h[l] * wd[l] : 0);
    }
    g->A = g->a_off[n_levels];
    return FRLW_OK;
}

int simota_launch(const float *preds, const double *labels, const float *x_shifts, const float *y_shifts,
                  const float *strides, int B, int A, int G, int num_classes, float radius, uint8_t *fg,
                  int32_t *matched_gt, double *matched_iou, int32_t *num_fg, const SimotaWs &w, hipStream_t s)
{
    const size_t lds = (size_t)2 * A * sizeof(double);
    if (lds > 150 * 1024) return FRLW_ERR_UNSUPPORTED; // two float64 rows must fit the 160 KB LDS
    hipLaunchKernelGGL(k_nlabel, dim3(B), dim3(64), 0, s, labels, B, G, w.nlabel, num_fg);
    const long long n = (long long)B * A;
    hipLaunchKernelGGL(k_candidates, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, labels, x_shifts, y_shifts,
                       strides, B, A, G, radius, w);
    if (lds > 48 * 1024) // above the default dynamic-LDS limit
        HIP_TRY(hipFuncSetAttribute((const void *)k_rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_rows, dim3(B * G), dim3(kRowThreads), lds, s, preds, labels, x_shifts, y_shifts, strides, B,
                       A, G, num_classes, radius, w);
    hipLaunchKernelGGL(k_resolve, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, B, A, G, w, fg, matched_gt,
                       matched_iou, num_fg);
    return FRLW_OK;
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
    SimotaWs w;
    if (ws_layout(B, A, G, &w, (uint8_t *)workspace) > workspace_bytes) return FRLW_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int rc = simota_launch(preds, labels, x_shifts, y_shifts, strides, B, A, G, num_classes, radius, fg, matched_gt,
                                 matched_iou, num_fg, w, s);
    if (rc != FRLW_OK) return rc;
    if (nlabel) HIP_TRY(hipMemcpyAsync(nlabel, w.nlabel, sizeof(int) * B, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipGetLastError());
    return FRLW_OK;
}


size_t frlw_yolox_loss_workspace_bytes(int B, int A, int G)
{
    if (B <= 0 || A <= 0 || G <= 0) return 0;
    return loss_ws_layout(B, A, G, nullptr, nullptr);
}

int frlw_yolox_loss_fwd(const float *const *raw, const int32_t *h, const int32_t *w, const float *strides, int n_levels,
                        int B, int num_classes, const double *labels, int G, float radius, float *preds, uint8_t *fg,
                        int32_t *matched_gt, double *matched_iou, double *result, void *workspace,
                        size_t workspace_bytes, frlw_stream_t stream)
{
    (void)hipGetLastError();
    LossGeom g;
    int rc = loss_geom(raw, nullptr, h, w, strides, n_levels, B, num_classes, &g);
    if (rc != FRLW_OK) return rc;
    if (!labels || G <= 0 || !preds || !fg || !matched_gt || !matched_iou || !result || !workspace) return FRLW_ERR_ARG;
    LossWs lw;
    if (loss_ws_layout(B, g.A, G, &lw, (uint8_t *)workspace) > workspace_bytes) return FRLW_ERR_WORKSPACE;
    SimotaWs sw;
    ws_layout(B, g.A, G, &sw, lw.simota);
    sw.nlabel = lw.nlabel;
    hipStream_t s = (hipStream_t)stream;
    const long long n = (long long)B * g.A;
    hipLaunchKernelGGL(k_loss_decode, dim3((unsigned)((n * g.P + kLossThreads - 1) / kLossThreads)), dim3(kLossThreads), 0, s,
                       g, preds, lw.xs, lw.ys, lw.st);
    rc = simota_launch(preds, labels, lw.xs, lw.ys, lw.st, B, g.A, G, num_classes, radius, fg, matched_gt, matched_iou,
                       lw.num_fg, sw, s);
    if (rc != FRLW_OK) return rc;
    const int n_wg = (int)((n + kLossThreads - 1) / kLossThreads);
    hipLaunchKernelGGL(k_loss_fwd, dim3(n_wg), dim3(kLossThreads), 0, s, preds, labels, fg, matched_gt, matched_iou, B, g.A,
                       G, num_classes, lw.partial);
    hipLaunchKernelGGL(k_loss_final, dim3(1), dim3(kLossThreads), 0, s, lw.partial, n_wg, lw.num_fg, lw.nlabel, B, result);
    HIP_TRY(hipGetLastError());
    return FRLW_OK;
}

int frlw_yolox_loss_bwd(const float *const *raw, const int32_t *h, const int32_t *w, const float *strides, int n_levels,
                        int B, int num_classes, const double *labels, int G, const uint8_t *fg,
                        const int32_t *matched_gt, const double *matched_iou, const double *result,
                        const double *grad_result, float *const *grad_raw, frlw_stream_t stream)
{
    (void)hipGetLastError();
    LossGeom g;
    const int rc = loss_geom(raw, grad_raw, h, w, strides, n_levels, B, num_classes, &g);
    if (rc != FRLW_OK) return rc;
    if (!grad_raw || !labels || G <= 0 || !fg || !matched_gt || !matched_iou || !result || !grad_result) return FRLW_ERR_ARG;
    const long long n = (long long)B * g.A;
    hipLaunchKernelGGL(k_loss_bwd, dim3((unsigned)((n + kLossThreads - 1) / kLossThreads)), dim3(kLossThreads), 0,
                       (hipStream_t)stream, g, labels, fg, matched_gt, matched_iou, G, result, grad_result);
    HIP_TRY(hipGetLastError());
    return FRLW_OK;
}

} // extern "C"
