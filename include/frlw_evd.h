/*
 * frlw_evd.h -- C-ABI of the MI355X-native FRLW-EvD hot path (libfrlw_evd.so, gfx950 only).
 *
 * The reference (HarmoniaLeo/FRLW-EvD) has no FFI on this path: its boundary is a set of Python
 * functions that take a device tensor of events and return fresh device tensors.  Each entry point
 * below replaces the body of one of those functions; the Python shims in frlw-evd_amd/ keep the
 * reference's names and argument meaning and call these through ctypes (INTEGRATION.md shows
 * the binding).  Citations are file:line in the reference tree.
 *
 * Conventions
 *   - every pointer except frlw_events_t itself, `lamdas` and `status_out` is DEVICE memory
 *     owned by the caller; nothing is allocated, retained or freed by the library;
 *   - calls are asynchronous on `stream` (a hipStream_t; NULL = the default stream),
 *     re-entrant (no globals) and contain no host synchronisation, so they can be captured
 *     into a hipGraph;
 *   - scratch comes from an explicit workspace: query frlw_encoder_workspace_bytes(), pass
 *     a device buffer of at least that size; one workspace per in-flight call;
 *   - return value: FRLW_OK or a negative FRLW_ERR_* for errors detectable at launch time.
 *     Data-dependent errors (coordinate out of range = the IndexError torch raises in
 *     index_add_, generate_eventvolume.py:32) are recorded in the workspace and read back
 *     with frlw_encoder_status(), which synchronises the stream.  After such an error the
 *     outputs are unspecified but no out-of-bounds access has happened.
 */
#ifndef FRLW_EVD_H
#define FRLW_EVD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *frlw_stream_t; /* hipStream_t */

enum {
    FRLW_OK = 0,
    FRLW_ERR_ARG = -1,       /* bad argument (NULL pointer, non-positive shape, ...) */
    FRLW_ERR_INDEX = -2,     /* event coordinate outside the encode shape (torch: IndexError) */
    FRLW_ERR_WORKSPACE = -3, /* workspace too small */
    FRLW_ERR_HIP = -4,       /* a HIP runtime call failed */
    FRLW_ERR_POLARITY = -5,  /* polarity outside {0, 1} */
    FRLW_ERR_UNSUPPORTED = -6
};

/* Event array layouts. */
enum {
    /* (N, row_stride) float64 rows [x, y, t, p, ...]: the tensor the reference functions take
     * (generate_eventvolume.py:135, generate_taf.py:195,203).  x, y, p are truncated toward
     * zero like .long(); t is used as the harness left it (already normalised / absolute). */
    FRLW_LAYOUT_XYTP_F64 = 0,
    /* raw 8-byte Prophesee DAT Event2D records: t:u32, then x = w & 16383,
     * y = (w >> 14) & 16383, p = (w >> 28) & 1 (src/io/dat_events_tools.py:16,96-98).
     * The window selection / f64 time normalisation the reference harness does on the host
     * tensor (generate_taf.py:197-215, generate_eventvolume.py:139-141) then runs on device. */
    FRLW_LAYOUT_DAT8 = 1
};

typedef struct frlw_events {
    const void *data;  /* device pointer to the event array */
    int64_t n;         /* number of events */
    int32_t layout;    /* FRLW_LAYOUT_* */
    int32_t row_stride; /* XYTP_F64: doubles per row (>= 4); ignored for DAT8 */
    /* Optional coordinate maps for DAT8 (device, uint16): x <- xmap[x], y <- ymap[y].  They carry
     * the harness' down-scale `x * rw, y * rh` + truncation (generate_taf.py:216-219); NULL =
     * identity.  map_w / map_h = table lengths (the sensor width / height). */
    const uint16_t *xmap;
    const uint16_t *ymap;
    int32_t map_w;
    int32_t map_h;
} frlw_events_t;

#define FRLW_MAX_WINDOWS 64
#define FRLW_MAX_LAMDAS 8
#define FRLW_MAX_BINS 8 /* Event Volume bins and TAF K: register-resident per cell */

/* frlw_taf_encode flags */
#define FRLW_TAF_U8_FLIP_K 1 /* write out_u8 newest slot first (np.flip(axis=0), generate_taf.py:229) */

/* Bytes of device workspace any encoder needs for `n_events` events on an H x W encode shape. */
size_t frlw_encoder_workspace_bytes(int64_t n_events, int H, int W);

/* Synchronise `stream` and fetch the data-dependent status of the last encoder call that used
 * `workspace`: FRLW_OK, FRLW_ERR_INDEX or FRLW_ERR_POLARITY. */
int frlw_encoder_status(const void *workspace, frlw_stream_t stream, int *status_out);

/*
 * Event Count Image -- replaces generate_eventframe(events, shape),
 * generate_eventcountimage.py:19-41.  out_f32: (2, H, W), channel = polarity, value * 255.
 * out_u8 (optional): the same after .astype(uint8) (generate_eventcountimage.py:178-180).
 * All n events are used (the harness has already cut events[-events_window:], :155).
 */
int frlw_eci_encode(const frlw_events_t *ev, int H, int W, float *out_f32, uint8_t *out_u8,
                    void *workspace, size_t workspace_bytes, frlw_stream_t stream);

/*
 * Event Volume -- replaces generate_agile_event_volume_cuda(events, shape, events_window,
 * volume_bins), generate_eventvolume.py:15-42.  out_f32: (2*bins, H, W), channel
 * 2*(k-1) + (0 if p == 1 else 1), scaled / 5 * 255.  out_u8 (optional): clipped at 255 then
 * truncated (generate_eventvolume.py:155-157).
 * XYTP_F64: column t is the normalised time, t_end / window_us are ignored.
 * DAT8: events with t <= t_end - window_us are dropped and t <- (t - (t_end - window_us)) /
 * window_us in f64 (generate_eventvolume.py:139-141).
 */
int frlw_ev_encode(const frlw_events_t *ev, int H, int W, int bins, int64_t t_end,
                   int64_t window_us, float *out_f32, uint8_t *out_u8, void *workspace,
                   size_t workspace_bytes, frlw_stream_t stream);

/*
 * Surface of Active Events -- replaces generate_leaky_cuda(events, shape, lamdas, memory, now),
 * generate_surfaceofactiveevents.py:44-80.  Events with x >= W or y >= H are dropped (:72).
 * mem_in: previous (2, H, W) memory or NULL; mem_out: new memory (2, H, W), required, may alias
 * mem_in.  out_f32: (2*n_lamda, H, W) lamda-major then polarity, * 255; out_u8 optional.
 * lamdas: HOST array of n_lamda doubles (cast to f32 like torch does for a Python scalar).
 * DAT8: events with t <= now - window_us are dropped (:183); window_us <= 0 keeps all.
 */
int frlw_sae_encode(const frlw_events_t *ev, int H, int W, const double *lamdas, int n_lamda,
                    const float *mem_in, float *mem_out, int64_t now, int64_t window_us,
                    float *out_f32, uint8_t *out_u8, void *workspace, size_t workspace_bytes,
                    frlw_stream_t stream);

/*
 * Temporal Active Focus -- replaces generate_taf_cuda(events, shape, past_volume, volume_bins)
 * (generate_taf.py:19-67) and, for DAT8 streams, the harness loop around it (:197-227) plus
 * leaky_transform (:69-76) and the uint8 truncation (:228-235) in ONE pass over the events.
 *   state   : (H, W, 2, K) f32, read and updated in place (the reference returns a new tensor;
 *             the shim clones first).  Initialise to -6000 for a fresh sequence (:205-209).
 *   view_f32: optional (2K, H, W) f32 = ecd_viewed after the last window, channel 2k + p.
 *   out_u8  : optional (K, 2, H, W) = uint8(leaky_transform(view)); FRLW_TAF_U8_FLIP_K reverses k.
 * XYTP_F64: one window; column t is the window-relative time in [0, 1]; n_windows must be 1 and
 * t_start / window_us are ignored.
 * DAT8: window of an event z = last i in [0, n_windows) with t_start + i*window_us <= t <=
 * t_start + (i+1)*window_us, else 0 (:197-203); t <- (t - t_min) / (window_us + 1e-8) in f64 (:215).
 * A window without any event leaves the state untouched (:40-41).
 */
int frlw_taf_encode(const frlw_events_t *ev, int H, int W, int K, int64_t t_start,
                    int64_t window_us, int n_windows, float *state, float *view_f32,
                    uint8_t *out_u8, int flags, void *workspace, size_t workspace_bytes,
                    frlw_stream_t stream);

/* leaky_transform(ecd), generate_taf.py:69-76, on n floats; either output may be NULL. */
int frlw_leaky_transform(const float *in, int64_t n, float *out_f32, uint8_t *out_u8,
                         frlw_stream_t stream);

/* F.interpolate(mode='nearest') of a (C, H, W) volume to (C, Ho, Wo) as the harnesses use it
 * (generate_eventvolume.py:149): src = min(floor(dst * f32(in / out)), in - 1). */
int frlw_resize_nearest_f32(const float *in, int C, int H, int W, int Ho, int Wo, float *out,
                            frlw_stream_t stream);
int frlw_resize_nearest_u8(const uint8_t *in, int C, int H, int W, int Ho, int Wo, uint8_t *out,
                           frlw_stream_t stream);

/* np.where(v > 255, 255, v).astype(uint8) (clip255 != 0) or plain .astype(uint8) truncation. */
int frlw_quantize_u8(const float *in, int64_t n, int clip255, uint8_t *out, frlw_stream_t stream);

/* Library identification: "frlw_evd <version> gfx950". */
const char *frlw_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FRLW_EVD_H */
