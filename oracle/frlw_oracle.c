/*
 * frlw_oracle.c -- CPU restatement of the FRLW-EvD event encoders.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for the HIP encoders.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product path (frlw-evd_amd/) never does.
 *
 * Parity is PINNED: every function below is checked bit-for-bit (f32) against outputs of the
 * reference's own Python functions run single-threaded on CPU in the build container
 * (tests/golden/make_golden.py generated tests/golden/<case>.npz; tests/test_oracle_golden.py).
 * The two transcendental epilogues (expf in SAE, log1pf in leaky_transform) are pinned on the
 * quantised uint8 artefact with the mismatch budget stated in the tests, because torch-CPU
 * uses SLEEF and this file uses libm.
 *
 * Semantics = sequential f32 accumulation in stream order (torch index_add_ on one CPU thread),
 * last-writer-wins for index_put_.  Build with -ffp-contract=off (see oracle/Makefile).
 *
 * Citations are file:line in the reference tree (HarmoniaLeo/FRLW-EvD).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_ERR_INDEX (-1) /* torch would raise IndexError */
#define ORC_ERR_ARG (-2)
#define ORC_ERR_NOMEM (-3)

/* (N, stride) float64 rows [x, y, t, p, (z)] -- generate_eventvolume.py:135 */
#define EV_X(ev, i, s) ((ev)[(size_t)(i) * (s) + 0])
#define EV_Y(ev, i, s) ((ev)[(size_t)(i) * (s) + 1])
#define EV_T(ev, i, s) ((ev)[(size_t)(i) * (s) + 2])
#define EV_P(ev, i, s) ((ev)[(size_t)(i) * (s) + 3])

/* ---------------------------------------------------------------------------------------------
 * Event Count Image: generate_eventframe, generate_eventcountimage.py:19-41
 *   idx = 2x + 2Wy + p (:32); img[idx] += 0.05f per event; img > 1 -> 1 (:34);
 *   (H,W,2) -> (2,H,W) (:36); * 255 (:41).
 * ------------------------------------------------------------------------------------------- */
int orc_eventframe(const double *ev, int64_t n, int stride, int H, int W, float *out)
{
    const int64_t cells = (int64_t)H * W * 2;
    float *img = (float *)calloc((size_t)cells, sizeof(float));
    if (!img) return ORC_ERR_NOMEM;
    const float inc = 0.05f; /* zeros_like(x).float() + 0.05 */
    for (int64_t i = 0; i < n; ++i) {
        int64_t x = (int64_t)EV_X(ev, i, stride), y = (int64_t)EV_Y(ev, i, stride);
        int64_t p = (int64_t)EV_P(ev, i, stride); /* .long() truncates toward zero (:28) */
        int64_t idx = 2 * x + 2 * (int64_t)W * y + p;
        if (idx < 0 || idx >= cells) { free(img); return ORC_ERR_INDEX; }
        img[idx] = img[idx] + inc;
    }
    for (int64_t c = 0; c < cells; ++c)
        if (img[c] > 1.0f) img[c] = 1.0f;
    for (int c = 0; c < 2; ++c)
        for (int64_t px = 0; px < (int64_t)H * W; ++px)
            out[(int64_t)c * H * W + px] = img[px * 2 + c] * 255.0f;
    free(img);
    return ORC_OK;
}

/* ---------------------------------------------------------------------------------------------
 * Event Volume: generate_agile_event_volume_cuda, generate_eventvolume.py:15-42
 *   t* = bins * float(t) (:23); for k = 1..bins: w = 1 - |k - t*| (:27-28) times [p, 1-p];
 *   negatives -> 0 (:29); row x + W*y gets the 2*bins weights added (:31-32);
 *   (H*W, bins, 2) -> (2*bins, H, W) (:35); / 5 * 255 (:37, the 5 is hard-coded).
 * Adding a +-0 weight never changes an accumulator that starts at +0, so only w*pol > 0 is added.
 * ------------------------------------------------------------------------------------------- */
int orc_event_volume(const double *ev, int64_t n, int stride, int H, int W, int bins, float *out)
{
    const int64_t rows = (int64_t)H * W;
    const int C = 2 * bins;
    float *img = (float *)calloc((size_t)(rows * C), sizeof(float));
    if (!img) return ORC_ERR_NOMEM;
    for (int64_t i = 0; i < n; ++i) {
        int64_t x = (int64_t)EV_X(ev, i, stride), y = (int64_t)EV_Y(ev, i, stride);
        int64_t p = (int64_t)EV_P(ev, i, stride);
        float ts = (float)bins * (float)EV_T(ev, i, stride);
        int64_t row = x + (int64_t)W * y;
        if (row < 0 || row >= rows) { free(img); return ORC_ERR_INDEX; }
        float pol[2] = { (float)p, (float)(1 - p) };
        for (int k = 1; k <= bins; ++k) {
            float d = (float)k - ts;
            float w = 1.0f - fabsf(d);
            for (int c = 0; c < 2; ++c) {
                float a = w * pol[c];
                if (!(a >= 0.0f)) a = 0.0f;
                float *cell = &img[row * C + (k - 1) * 2 + c];
                *cell = *cell + a;
            }
        }
    }
    for (int c = 0; c < C; ++c)
        for (int64_t px = 0; px < rows; ++px)
            out[(int64_t)c * rows + px] = img[px * C + c] / 5.0f * 255.0f;
    free(img);
    return ORC_OK;
}

/* ---------------------------------------------------------------------------------------------
 * Surface of Active Events: generate_leaky_cuda -> taf_cuda,
 * generate_surfaceofactiveevents.py:44-80
 *   drop events with x >= W or y >= H (:72); t_img = float(now) - 5e6 everywhere (:48);
 *   t_img[p,y,x] = float(t), last writer wins (:49); max with previous memory (:51-52);
 *   memory <- t_img (:54); exp(float(lamda) * (t_img - float(now))) * 255 per lamda (:55-63).
 * ------------------------------------------------------------------------------------------- */
int orc_sae(const double *ev, int64_t n, int stride, int H, int W, const double *lamdas, int nl,
            const float *mem_in, int64_t now, float *out, float *mem_out)
{
    const int64_t plane = (int64_t)H * W, cells = 2 * plane;
    const float nowf = (float)now;
    const float init = (0.0f + nowf) - 5000000.0f;
    for (int64_t c = 0; c < cells; ++c) mem_out[c] = init;
    for (int64_t i = 0; i < n; ++i) {
        double xd = EV_X(ev, i, stride), yd = EV_Y(ev, i, stride);
        if (!(xd < (double)W && yd < (double)H)) continue;
        int64_t x = (int64_t)xd, y = (int64_t)yd, p = (int64_t)EV_P(ev, i, stride);
        if (x < 0 || y < 0 || p < 0 || p > 1) return ORC_ERR_INDEX;
        mem_out[p * plane + y * W + x] = (float)EV_T(ev, i, stride);
    }
    if (mem_in)
        for (int64_t c = 0; c < cells; ++c)
            if (!(mem_out[c] > mem_in[c])) mem_out[c] = mem_in[c];
    for (int l = 0; l < nl; ++l) {
        const float lam = (float)lamdas[l];
        for (int64_t c = 0; c < cells; ++c)
            out[(int64_t)l * cells + c] = expf(lam * (mem_out[c] - nowf)) * 255.0f;
    }
    return ORC_OK;
}

/* ---------------------------------------------------------------------------------------------
 * Temporal Active Focus, one window: generate_taf_cuda -> taf_cuda, generate_taf.py:19-67
 *   idx = p + 2x + 2Wy; cnt[idx] += 1 (:23-24); sum[idx] += float(t) - 1 (:25-26);
 *   mean = sum / (cnt + 1e-8) (:27); forward = cnt == 0 (:35);
 *   all cells empty -> state unchanged (:40-41); else on cat([old, mean]) for i = K..1:
 *   e[i-1] -= 1; e[i] = forward ? e[i-1] : e[i] (:45-47); drop slot 0 (:48-49).
 *   view = state.permute(3,2,0,1) -> (2K, H, W), channel 2k + p (:55).
 * state: (H, W, 2, K) f32, updated in place.  view may be NULL.
 * ------------------------------------------------------------------------------------------- */
int orc_taf_window(const double *ev, int64_t n, int stride, int H, int W, int K, float *state,
                   float *view)
{
    const int64_t cells = (int64_t)H * W * 2;
    float *cnt = (float *)calloc((size_t)cells, sizeof(float));
    float *sum = (float *)calloc((size_t)cells, sizeof(float));
    float *e = (float *)malloc(sizeof(float) * (size_t)(K + 1));
    if (!cnt || !sum || !e) { free(cnt); free(sum); free(e); return ORC_ERR_NOMEM; }
    for (int64_t i = 0; i < n; ++i) {
        int64_t x = (int64_t)EV_X(ev, i, stride), y = (int64_t)EV_Y(ev, i, stride);
        int64_t p = (int64_t)EV_P(ev, i, stride);
        int64_t idx = p + 2 * x + 2 * (int64_t)W * y;
        if (idx < 0 || idx >= cells) { free(cnt); free(sum); free(e); return ORC_ERR_INDEX; }
        float tv = (float)EV_T(ev, i, stride) - 1.0f;
        cnt[idx] = cnt[idx] + 1.0f;
        sum[idx] = sum[idx] + tv;
    }
    int any = 0;
    for (int64_t c = 0; c < cells && !any; ++c) any = cnt[c] != 0.0f;
    if (any) {
        const float eps = (float)1e-8;
        for (int64_t c = 0; c < cells; ++c) {
            float mean = sum[c] / (cnt[c] + eps);
            int fwd = cnt[c] == 0.0f;
            float *s = &state[c * K];
            for (int k = 0; k < K; ++k) e[k] = s[k];
            e[K] = mean;
            for (int i = K; i >= 1; --i) {
                e[i - 1] = e[i - 1] - 1.0f;
                if (fwd) e[i] = e[i - 1];
            }
            for (int k = 0; k < K; ++k) s[k] = e[k + 1];
        }
    }
    if (view) {
        const int64_t plane = (int64_t)H * W;
        for (int k = 0; k < K; ++k)
            for (int p = 0; p < 2; ++p)
                for (int64_t px = 0; px < plane; ++px)
                    view[((int64_t)k * 2 + p) * plane + px] = state[(px * 2 + p) * K + k];
    }
    free(cnt); free(sum); free(e);
    return ORC_OK;
}

/* leaky_transform, generate_taf.py:69-76: 255 * max(0, 1 - log1p(-v) / 8.7) */
void orc_leaky_transform(const float *in, int64_t n, float *out)
{
    const float den = (float)8.7;
    for (int64_t i = 0; i < n; ++i) {
        float v = log1pf(-in[i]);
        v = 1.0f - v / den;
        if (v < 0.0f) v = 0.0f;
        out[i] = v * 255.0f;
    }
}

/* F.interpolate(mode='nearest') as the harnesses call it (generate_eventvolume.py:149):
 * src = min(floor(dst * (in / out)), in - 1) with the scale held in f32 (ATen
 * nearest_neighbor_compute_source_index). */
void orc_resize_nearest(const float *in, int C, int H, int W, int Ho, int Wo, float *out)
{
    const float sh = (float)H / (float)Ho, sw = (float)W / (float)Wo;
    for (int c = 0; c < C; ++c)
        for (int yo = 0; yo < Ho; ++yo) {
            int ys = (int)floorf((float)yo * sh);
            if (ys > H - 1) ys = H - 1;
            for (int xo = 0; xo < Wo; ++xo) {
                int xs = (int)floorf((float)xo * sw);
                if (xs > W - 1) xs = W - 1;
                out[((int64_t)c * Ho + yo) * Wo + xo] = in[((int64_t)c * H + ys) * W + xs];
            }
        }
}

/* numpy .astype(np.uint8) on values in [0, 256): truncation (generate_taf.py:232,235).
 * clip255 = np.where(v > 255, 255, v) first (generate_eventvolume.py:156). */
void orc_quantize_u8(const float *in, int64_t n, int clip255, uint8_t *out)
{
    for (int64_t i = 0; i < n; ++i) {
        float v = in[i];
        if (clip255 && v > 255.0f) v = 255.0f;
        out[i] = (uint8_t)(int)v;
    }
}

/* ---------------------------------------------------------------------------------------------
 * Harness glue for a raw DAT stream.
 *
 * DAT Event2D record: t:u32, then x = w & 16383, y = (w & 268419072) >> 14,
 * p = (w & 268435456) >> 28 (src/io/dat_events_tools.py:16,96-98).
 * ------------------------------------------------------------------------------------------- */
static inline void dat_unpack(const uint8_t *rec, int64_t i, int64_t *t, int64_t *x, int64_t *y,
                              int64_t *p)
{
    uint32_t tt, w;
    memcpy(&tt, rec + 8 * i, 4);
    memcpy(&w, rec + 8 * i + 4, 4);
    *t = (int64_t)tt;
    *x = (int64_t)(w & 16383u);
    *y = (int64_t)((w & 268419072u) >> 14);
    *p = (int64_t)((w & 268435456u) >> 28);
}

/* Window index of one event, generate_taf.py:197-203: z starts at 0 and every window i with
 * start + i*win <= t <= start + (i+1)*win overwrites it, so the LAST matching i wins. */
static inline int taf_window_of(int64_t t, int64_t t_start, int64_t win, int n_windows)
{
    int z = 0;
    for (int i = 0; i < n_windows; ++i)
        if (t >= t_start + (int64_t)i * win && t <= t_start + (int64_t)(i + 1) * win) z = i;
    return z;
}

/*
 * TAF over a DAT stream, generate_taf.py:193-227:
 *   per event z (:197-203); per window: select z == iter in stream order (:212),
 *   t <- (t - t_min) / (t_max - t_min + 1e-8) in f64 (:213-215), optional coordinate
 *   down-scale x*rw, y*rh in f64 then truncation (:216-219, when Hs > H), generate_taf_cuda.
 * state (H, W, 2, K) in/out.  view (2K, H, W) optional = the last window's ecd_viewed.
 * Hs, Ws: sensor shape; H, W: encode shape (== sensor, or the smaller target).
 */
int orc_taf_stream_dat8(const uint8_t *rec, int64_t n, int Hs, int Ws, int H, int W, int K,
                        int64_t t_start, int64_t window_us, int n_windows, float *state,
                        float *view)
{
    if (n_windows <= 0 || window_us <= 0) return ORC_ERR_ARG;
    int32_t *z = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    int64_t *start = (int64_t *)calloc((size_t)n_windows + 1, sizeof(int64_t));
    int64_t *order = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    if (!z || !start || !order) { free(z); free(start); free(order); return ORC_ERR_NOMEM; }
    for (int64_t i = 0; i < n; ++i) {
        int64_t t, x, y, p;
        dat_unpack(rec, i, &t, &x, &y, &p);
        z[i] = taf_window_of(t, t_start, window_us, n_windows);
        start[z[i] + 1]++;
    }
    for (int w = 0; w < n_windows; ++w) start[w + 1] += start[w];
    {
        int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * (size_t)n_windows);
        if (!cur) { free(z); free(start); free(order); return ORC_ERR_NOMEM; }
        for (int w = 0; w < n_windows; ++w) cur[w] = start[w];
        for (int64_t i = 0; i < n; ++i) order[cur[z[i]]++] = i; /* stable */
        free(cur);
    }
    const double rw = (double)W / (double)Ws, rh = (double)H / (double)Hs;
    const int scale = H < Hs;
    int rc = ORC_OK;
    int64_t maxn = 0;
    for (int w = 0; w < n_windows; ++w)
        if (start[w + 1] - start[w] > maxn) maxn = start[w + 1] - start[w];
    double *ev = (double *)malloc(sizeof(double) * 4 * (size_t)(maxn > 0 ? maxn : 1));
    if (!ev) { free(z); free(start); free(order); return ORC_ERR_NOMEM; }
    for (int w = 0; w < n_windows && rc == ORC_OK; ++w) {
        const int64_t m = start[w + 1] - start[w];
        const double t_min = (double)(t_start + (int64_t)w * window_us);
        const double den = (double)window_us + 1e-8; /* t_max - t_min + 1e-8 */
        for (int64_t j = 0; j < m; ++j) {
            int64_t t, x, y, p;
            dat_unpack(rec, order[start[w] + j], &t, &x, &y, &p);
            double xd = (double)x, yd = (double)y;
            if (scale) { xd = xd * rw; yd = yd * rh; }
            ev[4 * j + 0] = xd;
            ev[4 * j + 1] = yd;
            ev[4 * j + 2] = ((double)t - t_min) / den;
            ev[4 * j + 3] = (double)p;
        }
        rc = orc_taf_window(ev, m, 4, H, W, K, state, (w == n_windows - 1) ? view : NULL);
    }
    free(ev); free(z); free(start); free(order);
    return rc;
}

/*
 * Event Volume over a DAT stream, generate_eventvolume.py:139-148:
 *   keep t > end - window (:139); t <- (t - (end - window)) / window in f64 (:141);
 *   optional coordinate down-scale (:143-145); generate_agile_event_volume_cuda.
 */
int orc_ev_stream_dat8(const uint8_t *rec, int64_t n, int Hs, int Ws, int H, int W, int bins,
                       int64_t t_end, int64_t window_us, float *out)
{
    double *ev = (double *)malloc(sizeof(double) * 4 * (size_t)(n > 0 ? n : 1));
    if (!ev) return ORC_ERR_NOMEM;
    const double rw = (double)W / (double)Ws, rh = (double)H / (double)Hs;
    const int scale = H < Hs;
    const int64_t t0 = t_end - window_us;
    int64_t m = 0;
    for (int64_t i = 0; i < n; ++i) {
        int64_t t, x, y, p;
        dat_unpack(rec, i, &t, &x, &y, &p);
        if (!(t > t0)) continue;
        double xd = (double)x, yd = (double)y;
        if (scale) { xd = xd * rw; yd = yd * rh; }
        ev[4 * m + 0] = xd;
        ev[4 * m + 1] = yd;
        ev[4 * m + 2] = ((double)t - (double)t0) / (double)window_us;
        ev[4 * m + 3] = (double)p;
        ++m;
    }
    int rc = orc_event_volume(ev, m, 4, H, W, bins, out);
    free(ev);
    return rc;
}

/* Event Count Image over a DAT stream, generate_eventcountimage.py:155-163: the last
 * `count` events of the stream (events[-events_window:]), optional coordinate down-scale. */
int orc_eci_stream_dat8(const uint8_t *rec, int64_t n, int Hs, int Ws, int H, int W, float *out)
{
    double *ev = (double *)malloc(sizeof(double) * 4 * (size_t)(n > 0 ? n : 1));
    if (!ev) return ORC_ERR_NOMEM;
    const double rw = (double)W / (double)Ws, rh = (double)H / (double)Hs;
    const int scale = H < Hs;
    for (int64_t i = 0; i < n; ++i) {
        int64_t t, x, y, p;
        dat_unpack(rec, i, &t, &x, &y, &p);
        double xd = (double)x, yd = (double)y;
        if (scale) { xd = xd * rw; yd = yd * rh; }
        ev[4 * i + 0] = xd; ev[4 * i + 1] = yd; ev[4 * i + 2] = (double)t; ev[4 * i + 3] = (double)p;
    }
    int rc = orc_eventframe(ev, n, 4, H, W, out);
    free(ev);
    return rc;
}

/* SAE over a DAT stream, generate_surfaceofactiveevents.py:183-191: keep t > now - window (:183),
 * optional coordinate down-scale, generate_leaky_cuda with the absolute microsecond stamps. */
int orc_sae_stream_dat8(const uint8_t *rec, int64_t n, int Hs, int Ws, int H, int W,
                        const double *lamdas, int nl, const float *mem_in, int64_t now,
                        int64_t window_us, float *out, float *mem_out)
{
    double *ev = (double *)malloc(sizeof(double) * 4 * (size_t)(n > 0 ? n : 1));
    if (!ev) return ORC_ERR_NOMEM;
    const double rw = (double)W / (double)Ws, rh = (double)H / (double)Hs;
    const int scale = H < Hs;
    int64_t m = 0;
    for (int64_t i = 0; i < n; ++i) {
        int64_t t, x, y, p;
        dat_unpack(rec, i, &t, &x, &y, &p);
        if (!(t > now - window_us)) continue;
        double xd = (double)x, yd = (double)y;
        if (scale) { xd = xd * rw; yd = yd * rh; }
        ev[4 * m + 0] = xd; ev[4 * m + 1] = yd; ev[4 * m + 2] = (double)t; ev[4 * m + 3] = (double)p;
        ++m;
    }
    int rc = orc_sae(ev, m, 4, H, W, lamdas, nl, mem_in, now, out, mem_out);
    free(ev);
    return rc;
}
