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
    FRLW_ERR_UNSUPPORTED = -6,
    FRLW_ERR_SPAN = -7       /* frlw_taf_encode_batch: an event outside [t_start, t_start + n_windows * window_us] */
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

/* Optional per-call overrides of the launch heuristics (experiments, and tests that force a rarely taken path).
 * Every field: < 0 = the library's choice.  There is no process-wide state: the knobs travel with the call.
 * `struct_size` = sizeof(frlw_tuning_t) as the CALLER compiled it: the struct has grown and will grow at its end, and the
 * library reads only the fields that lie inside `struct_size` (the others take their defaults), so a caller built against
 * an older header never has its stack read past the struct.  A struct_size below 8 or above 4096 is FRLW_ERR_ARG. */
typedef struct frlw_tuning {
    int32_t struct_size;     /* = sizeof(frlw_tuning_t) */
    int32_t tile_width_log2; /* 6..8 */
    int32_t batches_per_wave; /* 1..8: 64-event batches per wavefront of a partition workgroup */
    int32_t hot_tile_records; /* a tile with more records than this is shared by several workgroups (EV / TAF) */
    int32_t staged_scatter;   /* 0 / 1: records leave the partition through an LDS staging area */
    int32_t quarter_below;    /* frames with at most this many wavefronts run every tile as four quarter workgroups */
    int32_t no_value_table;   /* 1: TAF DAT8 computes the f64 division per event instead of the per-call table */
    int32_t taf_tile_walk;    /* frlw_taf_encode_batch / frlw_ev_encode_batch: 1 = tiles are split in LDS by the kernel that walks
                               * them (kf_taf_tile / kf_ev_tile: less HBM traffic, measured slower), 0 = split pass + sub-tile
                               * kernel (the default) */
    int32_t direct_bins;      /* frlw_taf_encode_batch / frlw_ev_encode_batch: 1 = the partition's bins are the 256-cell sub-tiles
                               * wherever the frame allows it (at most 64 tiles: the 304x240 class) -- no second-level split pass
                               * at all; 0 = always tile bins + split pass; default: sub-tile bins for calls with fewer than 512
                               * (sequence, tile) pairs and more than 8192 events per pair on average */
    int32_t chunk_major;      /* frlw_taf_encode_batch / frlw_ev_encode_batch: 1 = the chunk-major partition (no histogram pass: the
                               * scatter writes every chunk sorted by bin where it stands plus a directory row, the consumers
                               * gather -- the default wherever a sequence has at most 4096 chunks), 0 = histogram + scans +
                               * bin-major scatter (the only form for longer sequences and for the tile walk) */
    int32_t ev_lds_float_atomics; /* frlw_ev_encode_batch, direct mode of the chunk-major partition: 1 = one wavefront per sub-tile
                               * sums its list with LDS float atomics in stream order (kf_ev_fadd: short latency chain, 3 x the
                               * cycles per record -- the default up to 3 M events per call), 0 = the ticket-sort kernel */
    int32_t walk_window_table; /* frlw_taf_encode_batch, chunk-major partition with tile bins: 1 = the split kernel leaves every
                               * sub-tile list's window starts in a table and the walk reads them (the default), 0 = the walk
                               * finds them with a scan of its list, as it does for every other partition form */
} frlw_tuning_t;

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
    const frlw_tuning_t *tuning; /* HOST pointer or NULL */
} frlw_events_t;

#define FRLW_MAX_WINDOWS 64
#define FRLW_MAX_LAMDAS 8
#define FRLW_MAX_BINS 8 /* Event Volume bins and TAF K: register-resident per cell */

/* frlw_taf_encode flags */
#define FRLW_TAF_U8_FLIP_K 1 /* write out_u8 newest slot first (np.flip(axis=0), generate_taf.py:229) */

/* Bytes of device workspace any encoder needs for `n_events` events on an H x W encode shape: the larger of the general
 * plan and of the two-launch form frlw_sae_encode / frlw_eci_encode take for eligible calls (a workspace of exactly this
 * size gets the same path a larger one gets). */
size_t frlw_encoder_workspace_bytes(int64_t n_events, int H, int W);

/* Which form frlw_sae_encode / frlw_eci_encode launched, counted per process since it loaded the library (diagnostics and
 * tests; the results are the same bits either way): counts[0] = Surface of Active Events through the two-launch form,
 * [1] = through the general path, [2] = Event Count Image through the two-launch form, [3] = through the single-launch
 * scan or the general path. */
int frlw_encoder_path_counts(uint64_t counts[4]);

/* Synchronise `stream` and fetch the data-dependent status of the last encoder call that used
 * `workspace`: FRLW_OK, FRLW_ERR_INDEX, FRLW_ERR_POLARITY, FRLW_ERR_SPAN, or FRLW_ERR_HIP when a workgroup of a
 * fast-path launch gave up waiting for the launch's header reset (a bounded wait; the call's outputs are then incomplete). */
int frlw_encoder_status(const void *workspace, frlw_stream_t stream, int *status_out);

/* Callers that do not synchronise after every call (a training loop that encodes batch after batch) still must not
 * train on a stale state: every encoder call also ORs its data-dependent status into a word of the workspace header
 * that survives the next call.  frlw_workspace_init() zeroes the header once after the workspace is allocated;
 * frlw_encoder_deferred_status() synchronises `stream`, returns the accumulated status of all calls since (FRLW_OK or
 * the first of FRLW_ERR_INDEX / _POLARITY / _SPAN / _HIP) and clears it.  The reference raises at the offending call
 * (torch's IndexError in index_add_, generate_taf.py:24); this is the same information, one host sync per batch. */
int frlw_workspace_init(void *workspace, size_t workspace_bytes, frlw_stream_t stream);
int frlw_encoder_deferred_status(void *workspace, frlw_stream_t stream, int *status_out);

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

/*
 * Temporal Active Focus, fast path for a BATCH of independent sequences (csrc/taf_fast.hip) -- the same harness loop
 * (generate_taf.py:193-235) for n_seq <= FRLW_MAX_SEQUENCES streams in one launch sequence.  Sequence s owns the DAT8
 * records [seq_offsets[s], seq_offsets[s + 1]) of ev->data (HOST array of n_seq + 1 event indices), starts at
 * t_start[s] (HOST array) and has its own FIFO state, its own outputs and its own "window without any event leaves the
 * state untouched" rule (:40-41 is evaluated per file, :143-160):
 *   state (n_seq, H, W, 2, K) in place; view_f32 (n_seq, 2K, H, W) or NULL; out_u8 (n_seq, K, 2, H, W) or NULL.
 * Bit-identical to n_seq calls of frlw_taf_encode.  Differences in contract:
 *   - DAT8 only; 12 + ceil(log2 n_windows) + bits(window_us) <= 32 (else FRLW_ERR_UNSUPPORTED at launch);
 *   - every event must lie inside [t_start[s], t_start[s] + n_windows * window_us] (the harness cuts the stream with
 *     seek_time, :162-193).  Checked on device: frlw_encoder_status() then reports FRLW_ERR_SPAN (or FRLW_ERR_INDEX)
 *     and NOTHING has been written -- state and outputs are untouched, so the caller can fall back to
 *     frlw_taf_encode, which places such events like the reference does.
 * The stream may be in any order (sums follow stream order); a time-sorted stream is the fast case.
 * Workspace: frlw_taf_batch_workspace_bytes(total events, n_seq, H, W, window_us); 0 = unsupported shape.
 */
#define FRLW_MAX_SEQUENCES 64
size_t frlw_taf_batch_workspace_bytes(int64_t n_events, int n_seq, int H, int W, int64_t window_us);
int frlw_taf_encode_batch(const frlw_events_t *ev, const int64_t *seq_offsets, const int64_t *t_start, int n_seq, int H,
                          int W, int K, int64_t window_us, int n_windows, float *state, float *view_f32,
                          uint8_t *out_u8, int flags, void *workspace, size_t workspace_bytes, frlw_stream_t stream);

/*
 * Row-stripe sharding of ONE frame over several GPUs ("route events by y; each GPU owns its stripe of state /
 * output -- no halo, no exchange"): every rank holds the stream, encodes rows [y_lo, y_lo + rows) of the H_full x W frame
 * (events of other rows are skipped) into ITS stripe of the state / outputs -- (n_seq, rows, W, 2, K) and so on.  One quantity
 * is global: "a window without any event in the whole frame leaves the state untouched" (generate_taf.py:40-41).  So the
 * encode runs in two halves around one 8-byte-per-sequence exchange:
 *   frlw_taf_stripe_partition(...)            hist, scans, scatter of this stripe; leaves the stripe's window masks in the workspace
 *   frlw_taf_stripe_window_masks(workspace)   device pointer to the n_seq 64-bit masks: OR-reduce them IN PLACE over the ranks
 *                                             (RCCL all-reduce with BOR through torch.distributed in the shim), same stream order
 *   frlw_taf_stripe_finish(...)               split + walk with the reduced masks; same arguments, same workspace
 * The stripes put together are bit-identical to frlw_taf_encode_batch on the whole frame.  Workspace:
 * frlw_taf_batch_workspace_bytes(total events, n_seq, rows, W, window_us).
 */
int frlw_taf_stripe_partition(const frlw_events_t *ev, const int64_t *seq_offsets, const int64_t *t_start, int n_seq, int H_full,
                              int W, int y_lo, int rows, int K, int64_t window_us, int n_windows, void *workspace,
                              size_t workspace_bytes, frlw_stream_t stream);
unsigned long long *frlw_taf_stripe_window_masks(void *workspace);
int frlw_taf_stripe_finish(const frlw_events_t *ev, const int64_t *seq_offsets, const int64_t *t_start, int n_seq, int H_full, int W,
                           int y_lo, int rows, int K, int64_t window_us, int n_windows, float *state, float *view_f32,
                           uint8_t *out_u8, int flags, void *workspace, size_t workspace_bytes, frlw_stream_t stream);

/*
 * Event Volume for a batch of independent streams -- the harness lines generate_eventvolume.py:139-157 (window cut,
 * f64 normalisation) around generate_agile_event_volume_cuda (:15-42) for n_seq <= FRLW_MAX_SEQUENCES label windows in one
 * launch sequence.  Sequence s owns the DAT8 records [seq_offsets[s], seq_offsets[s + 1]) and ends at t_end[s] (HOST
 * arrays): events with t <= t_end[s] - window_us are dropped (:139), t <- (t - (t_end[s] - window_us)) / window_us in f64
 * (:141).  out_f32 (n_seq, 2 * bins, H, W) and / or out_u8 (the same clipped at 255 and truncated, :155-157).
 * Bit-identical to n_seq calls of frlw_ev_encode.  Differences in contract: DAT8 only; window_us < 2^20; an event with
 * t > t_end[s] is reported by frlw_encoder_status() as FRLW_ERR_SPAN and NOTHING is written (frlw_ev_encode places it).
 * Workspace: frlw_ev_batch_workspace_bytes(total events, n_seq, H, W, window_us); 0 = unsupported shape.
 */
size_t frlw_ev_batch_workspace_bytes(int64_t n_events, int n_seq, int H, int W, int64_t window_us);
int frlw_ev_encode_batch(const frlw_events_t *ev, const int64_t *seq_offsets, const int64_t *t_end, int n_seq, int H, int W,
                         int bins, int64_t window_us, float *out_f32, uint8_t *out_u8, void *workspace,
                         size_t workspace_bytes, frlw_stream_t stream);

/* The verdict of the one-time LDS lane-order self-test of the fast paths (frlw_taf_encode_batch, frlw_ev_encode_batch, the
 * two-launch forms of frlw_sae_encode / frlw_eci_encode) for the CURRENT device: *ok_out = 1 when the property held, 0 when
 * it did not (those entry points then answer FRLW_ERR_UNSUPPORTED and callers take the general path -- same bits, different
 * step time).  The test runs once per process and device (0.7 ms, one host synchronisation), on the first fast-path call or
 * here, whichever comes first: a job of N ranks runs it N times, once on each GPU.  Ranks that must keep the same step time
 * agree on the MINIMUM of their verdicts before the first encode (frlw_evd_amd.dist.agree_fast_path: one all-reduce).
 * `workspace`: any initialised encoder workspace (>= 1 KB). */
int frlw_fast_path_verdict(void *workspace, size_t workspace_bytes, frlw_stream_t stream, int *ok_out);

/* Self-test of the gfx950 properties frlw_taf_encode_batch relies on, for the lanes of one LDS atomic instruction that
 * hit the same address: a returning integer add serves them in ascending lane order, and ds_add_f32 applies them in that
 * order with v_add_f32 rounding (= the reference's sequential f32 sum).  256 workgroups x `iters` batches of random
 * addresses in [0, n_addr <= 512); out_dev (device, 3 x uint64): [0] = rank mismatches (must be 0), [1] = lanes that
 * shared their address with a lower lane (the sample size), [2] = f32 sums that differ from the sequential sum (must be 0). */
int frlw_selftest_lds_atomic_order(int n_addr, int iters, unsigned long long *out_dev, frlw_stream_t stream);
/* frlw_taf_encode_batch runs that self-test by itself on its first call per device (one host synchronisation for the
 * lifetime of the process, result cached) and returns FRLW_ERR_UNSUPPORTED from then on if the property does not hold, so
 * that callers take frlw_taf_encode. */
#ifdef FRLW_DEV_BUILD
/* Developer builds only (-DFRLW_DEV_BUILD -> libfrlw_evd_dev.so; NOT a symbol of the product library): tests force the
 * cached verdict of the self-test: value 0 = "does not hold", 1 = "holds", -1 = forget. */
int frlw_debug_force_lds_order(int value);
#endif

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

/* ---------------------------------------------------------------------------------------------
 * YOLOX detector forward (eval): replaces the forward of the reference's nn.Modules
 * CSPDarknet / YOLOPAFPN / YOLOXHead (core/yolox/models/, all files) called from core/model.py:40-58.
 *
 * The network is a plan of kernel launches built once from the module tree (frlw-evd_amd/detector.py
 * walks the reference-named modules, folds BatchNorm into the weights and uploads them) and replayed
 * natively by frlw_det_run.  Tensors are NHWC f32; a tensor argument is (buffer index, pixel stride
 * in floats, channel offset), so a concat is just several producers writing disjoint channel slices
 * of one buffer.  Buffer indices refer to the `bufs` array passed to frlw_det_run.
 * ------------------------------------------------------------------------------------------- */
typedef struct frlw_detector frlw_detector_t;

enum { FRLW_ACT_NONE = 0, FRLW_ACT_SILU = 1, FRLW_ACT_SIGMOID = 2 };

frlw_detector_t *frlw_det_create(void);
void frlw_det_destroy(frlw_detector_t *d);
int frlw_det_num_ops(const frlw_detector_t *d);

/* Independent sub-graphs (the three head levels) can run concurrently: ops added after frlw_det_set_lane(d, l)
 * with l in 1..2 are launched on library-owned side streams between a fork (side streams wait for everything
 * launched so far on the caller's stream) and a join (the caller's stream waits for the side streams). */
int frlw_det_set_lane(frlw_detector_t *d, int lane);

/* Arithmetic of the convolutions added from now on (default 0):
 *   0  v_mfma_f32_32x32x2_f32 on float32 operands: every product exact, the float32 MFMA rate (157 TFLOP/s peak);
 *   1  float32 products from THREE bf16 MFMAs: x = hi + lo with hi = bf16(x), lo = bf16(x - hi);
 *      a * b ~ a_hi b_hi + a_hi b_lo + a_lo b_hi, float32 accumulation -- <= 2^-16 relative error per product (observed
 *      3e-6 of max |y| per layer; the detector's stated tolerance is 1e-3), 5.3 x the matrix rate.  The activations stay
 *      float32 in memory (split in registers); `w_dev` of frlw_det_add_conv is then the SPLIT IMAGE of the [K][Npad]
 *      operand, made by frlw_conv_split_operand (frlw_conv_split_operand_bytes(K, Npad) bytes, caller-allocated). */
int frlw_det_set_precision(frlw_detector_t *d, int precision);
size_t frlw_conv_split_operand_bytes(int K, int Npad);
int frlw_conv_split_operand(const float *w, int K, int Npad, void *out, frlw_stream_t stream);
int frlw_det_add_fork(frlw_detector_t *d);
int frlw_det_add_join(frlw_detector_t *d);

/* Optional scratch buffer (index into bufs; n_floats floats PER LANE, 3 lanes) for split-K partial sums of the
 * convolutions whose output grid would leave most CUs idle (8x10 feature maps).  The buffer must be ZERO when it is first
 * handed to frlw_det_run and is left alone by the caller afterwards: the last 1024 words of every lane's region are the
 * arrival counters of the in-kernel reduction (k_conv_mfma_sk: the split that arrives last at a tile's counter sums all
 * splits and resets it). */
int frlw_det_set_scratch(frlw_detector_t *d, int buf, int64_t n_floats);

/* Focus space-to-depth (network_blocks.py:205-217): NCHW (B, C, H, W) -> NHWC (B, H/2, W/2, 4C). */
int frlw_det_add_focus(frlw_detector_t *d, int src_buf, int C, int H, int W, int dst_buf);

/* BFM stem of the yolox_taf_bfm recipes, per-pixel part + Focus layout in one kernel
 * (core/Others/Temporal_Active_Focus.py:62-127 Temporal_Active_Focus_connect.forward up to self.conv; eval mode):
 * NCHW (B, C, H, W) with C = 2 T in {4, 8, 16} -> NHWC (B, H/2, W/2, 4 * 4 log2(T)).
 * `weights` (device, kept by reference): for every stage i its effective grouped 1x1 weight g * v / |v| as
 * (out_channels_i, in_channels_per_group_i) row-major then its bias; then trans_up weight (4E, E) + bias and
 * trans_down weight (E, 4E) + bias, E = 4 log2(T).  frlw_det_bfm_weight_count(C) floats (0: unsupported C). */
/* Focus space-to-depth (network_blocks.py:205-217) as a stand-alone call: (B, C, H, W) NCHW -> (B, H/2, W/2, 4C) NHWC with
 * the channel blocks TL, BL, TR, BR -- what the train step feeds its first BaseConv (the torch.cat + layout copy it
 * replaces cost 0.55 ms per step at batch 64). */
int frlw_focus_nhwc(const float *x, int B, int C, int H, int W, float *y, frlw_stream_t stream);

/* Focus + the stem's 3x3 BaseConv (darknet.py:292) fused: the space-to-depth image stays in LDS.  w_dev / bias_dev as for
 * frlw_det_add_conv with Cin = 4 C, k = 3, Npad = 32.  C in {10, 16} (TAF K = 5 / 8), Cout <= 32, else
 * FRLW_ERR_UNSUPPORTED (use frlw_det_add_focus + frlw_det_add_conv). */
int frlw_det_add_focus_stem(frlw_detector_t *d, int src_buf, int C, int H, int W, const float *w_dev, const float *bias_dev,
                            int Cout, int dst_buf, int dst_cs, int dst_co);
int frlw_det_bfm_weight_count(int C);
int frlw_det_add_bfm_stem(frlw_detector_t *d, int src_buf, int C, int H, int W, const float *weights, int n_weights,
                          int dst_buf);

/* nn.Upsample(scale_factor=2, mode="nearest") of a channel slice (yolo_pafpn.py:29).  When the slice is the output of the
 * convolution added just before (same lane; the FPN's lateral / reduce 1x1 convolutions), no launch is added: that
 * convolution's epilogue stores the upsampled copy too. */
int frlw_det_add_upsample(frlw_detector_t *d, int src_buf, int cs_src, int co_src, int C, int H, int W,
                          int dst_buf, int cs_dst, int co_dst);

/* SPPBottleneck pools (network_blocks.py:139-151): channels [0, C) -> max-pool 5 / 9 / 13 into [C, 4C). */
int frlw_det_add_spp_pool(frlw_detector_t *d, int buf, int cs, int C, int H, int W);

/*
 * BaseConv (network_blocks.py:33-65) with BatchNorm folded: y = act(conv(x, w) + bias) [+ res].
 *   w_dev   : device, (k*k*Cin, Npad) row-major, row (ky*k + kx)*Cin + ci, Npad = Cout rounded up to 32
 *   bias_dev: device, Cout floats (or NULL)
 *   k in {1, 3}, stride in {1, 2}, padding (k-1)/2; Cin, src_cs, src_co multiples of 4
 *   dst_bs  : floats between images in the destination (0 = dense Ho*Wo*dst_cs); lets the prediction
 *             convs write straight into the (B, A, 5 + nc) head tensor (yolo_head.py:229-231)
 *   res_buf : < 0 for none; Bottleneck shortcut added AFTER the activation (network_blocks.py:108-110)
 *   act     : FRLW_ACT_*; with FRLW_ACT_SIGMOID only channels >= sig_from are squashed (yolo_head.py:209-211)
 *   group_n : 0, or a multiple of 128: grouped convolution -- output channels [g * group_n, (g + 1) * group_n) read the Cin
 *             input channels starting at src_co + g * Cin (two towers of the head in one launch)
 */
int frlw_det_add_conv(frlw_detector_t *d, int src_buf, int src_cs, int src_co, int Cin, int H, int W,
                      const float *w_dev, const float *bias_dev, int Cout, int Npad, int k, int stride,
                      int dst_buf, int dst_cs, int dst_co, int64_t dst_bs, int res_buf, int res_cs,
                      int res_co, int act, int sig_from, int group_n);

/*
 * decode_outputs (yolo_head.py:258-303): xy = (xy + grid) * stride, wh = square(wh) * stride, keep
 * obj > obj_thr, class-agnostic NMS at iou_thr (the documented torchvision.ops.nms semantics: descending
 * score, suppress IoU > thr), emit [cx, cy, w, h, argmax cls, obj * max cls] in descending-score order.
 *   raw_buf    : (B, A, 5 + nc) f32     decoded_buf: optional (B, A, 5 + nc), < 0 for none
 *   dets_buf   : (B, A, 6) f32          counts_buf : (B, 1 + A) int32; [b][0] = detections of image b: 0 = nothing
 *                passed (the reference then returns one all-zero row), -1 = more than 8192 candidates (only reachable
 *                with A > 8192; not handled on device); [b][1..] = scratch (the candidates' anchors in score order)
 *   nms_buf    : (B, frlw_det_nms_workspace_floats(A)) f32 of scratch: per image the candidate count, the sorted corner
 *                boxes and the bit matrix "box i suppresses box j" (A = 6720: 5.8 MB per image).  Three launches: decode +
 *                sort per image, the bit matrix over the whole GPU, one small workgroup per image walking it in score order.
 */
long long frlw_det_nms_workspace_floats(int A);

/* The three biased 1x1 prediction convolutions of one head level in eval mode (yolo_head.py:205-231) as ONE streaming
 * pass: rows 0..4 of w_dev (F = 5 + nc rows of C floats: reg x4, obj, cls x nc) read channels [src_co, src_co + C) of the
 * level's feature buffer, rows 5.. read [src_co + C, src_co + 2 C); sigmoid on outputs >= 4; the result lands at anchors
 * [first_anchor, first_anchor + hw) of the (B, A, F) head tensor (dst_bs = A * F).  C <= 256, F <= 16, else
 * FRLW_ERR_UNSUPPORTED (use frlw_det_add_conv with the block weight matrix). */
int frlw_det_add_pred(frlw_detector_t *d, int src_buf, int src_cs, int src_co, int C, int hw, const float *w_dev,
                      const float *bias_dev, int F, int dst_buf, int first_anchor, int64_t dst_bs);

int frlw_det_add_decode_nms(frlw_detector_t *d, int raw_buf, int A, int nc, int n_levels, const int *lvl_h,
                            const int *lvl_w, const int *lvl_stride, float obj_thr, float iou_thr,
                            int decoded_buf, int dets_buf, int counts_buf, int nms_buf);

/* Launch ops [first, last) (last < 0: to the end) for B images on `stream`; bufs: n_bufs device pointers. */
int frlw_det_run(const frlw_detector_t *d, int B, void *const *bufs, int n_bufs, int first, int last,
                 frlw_stream_t stream);

/* The three biased 1x1 prediction convolutions of ONE head level in training mode and their gradients (csrc/pred_ops.hip)
 * -- replaces torch.cat([reg_preds[k](reg_feat), obj_preds[k](reg_feat), cls_preds[k](cls_feat)], 1) of
 * core/yolox/models/yolo_head.py:160-186 (training branch, no sigmoid) and its autograd.
 *   reg_feat, cls_feat (M = B*H*W, C) NHWC rows; w_reg (4, C), w_obj (1, C), w_cls (nc, C) as torch stores the 1x1 weights;
 *   out / dout (M, 5 + nc) rows [reg 0:4 | obj | cls]; C % 4 == 0, C <= 512, 1 <= nc <= 11.
 * Deterministic (fixed summation order for the weight and bias gradients). */
int64_t frlw_pred_bwd_scratch_floats(int64_t M, int C, int nc);
int frlw_pred_fwd(const float *reg_feat, const float *cls_feat, int64_t M, int C, int nc, const float *w_reg,
                  const float *b_reg, const float *w_obj, const float *b_obj, const float *w_cls, const float *b_cls,
                  float *out, frlw_stream_t stream);
int frlw_pred_bwd(const float *reg_feat, const float *cls_feat, const float *dout, int64_t M, int C, int nc,
                  const float *w_reg, const float *w_obj, const float *w_cls, float *d_reg_feat, float *d_cls_feat,
                  float *dw_reg, float *db_reg, float *dw_obj, float *db_obj, float *dw_cls, float *db_cls, float *scratch,
                  int64_t scratch_floats, frlw_stream_t stream);

/* The pools of SPPBottleneck in training mode (network_blocks.py:139-151): out (B, H, W, 4C) NHWC = cat[x, maxpool5, maxpool9,
 * maxpool13] (stride 1, "same" padding, ATen's tie rule: the first maximum in row-major window order), argmax (B, H, W, 3, C)
 * uint16 input pixel of every pooled value; the backward gathers dx (B, H, W, C) from dout (B, H, W, 4C) in a fixed order.
 * The maps of 64, 32 or 16 channels must fit the LDS (H * W <= 512), else FRLW_ERR_UNSUPPORTED. */
int frlw_spp_train_fwd(const float *x, int B, int H, int W, int C, float *out, uint16_t *argmax, frlw_stream_t stream);
int frlw_spp_train_bwd(const float *dout, const uint16_t *argmax, int B, int H, int W, int C, float *dx, frlw_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * SimOTA label assignment of the YOLOX training branch, whole batch, no host round trips.
 * Replaces the per-image Python of core/yolox/models/yolo_head.py:482-584 (get_assignments),
 * :586-669 (get_in_boxes_info), :671-707 (dynamic_k_matching) and core/yolox/utils/boxes.py:79-102.
 *
 *   preds     (B, A, 5 + num_classes) f32: decoded cx, cy, w, h (yolo_head.py:237-256), obj and class logits
 *   labels    (B, G, 5) f64 rows [class, cx, cy, w, h]; an image's boxes are its first n rows, n = number of
 *             rows with a positive field sum (yolo_head.py:330,349-350)
 *   x_shifts, y_shifts, strides  (A) f32: grid coordinates and stride of every anchor (yolo_head.py:254-256)
 *   radius    centre-sampling radius in strides (core/exp.py:377-384: 5 for GEN1, 2.5 for 1 Mpx)
 * outputs (caller-allocated):
 *   fg (B, A) u8, matched_gt (B, A) i32 (-1 = background), matched_iou (B, A) f64 (IoU of the prediction
 *   with its matched box, the class target), num_fg (B) i32, nlabel (B) i32 (may be NULL).
 * (frlw_yolox_loss_fwd below runs this assignment as a part of the whole loss.) */
size_t frlw_simota_workspace_bytes(int B, int A, int G);
int frlw_simota_assign(const float *preds, const double *labels, const float *x_shifts, const float *y_shifts,
                       const float *strides, int B, int A, int G, int num_classes, float radius, uint8_t *fg,
                       int32_t *matched_gt, double *matched_iou, int32_t *num_fg, int32_t *nlabel, void *workspace,
                       size_t workspace_bytes, frlw_stream_t stream);

/* The training loss of a batch on the raw level outputs (core/yolox/models/yolo_head.py:237-256 get_output_and_grid,
 * :305-473 get_losses; core/yolox/models/losses.py:16-36 IOUloss "iou"; nn.BCEWithLogitsLoss for objectness and class):
 * decode + concatenation, the SimOTA assignment above, the three loss sums and -- in _bwd -- the gradient of the raw
 * outputs, as six + one launches with no host synchronisation.
 *   raw, h, w, strides  HOST arrays of n_levels (<= 4) entries: device pointer to the level's (B, h, w, 5 + nc) f32 rows
 *                       cat[reg, obj, cls] (no sigmoid), its grid shape and stride
 *   labels (B, G, 5) f64 as for frlw_simota_assign
 * _fwd outputs (caller-allocated, kept for _bwd): preds (B, A, 5 + nc) f32 decoded, A = sum h*w; fg, matched_gt,
 *   matched_iou as above; result (6) f64 = {loss, 5 * loss_iou, loss_obj, loss_cls, num_fg / max(num_gt, 1),
 *   max(num_fg, 1)} -- the tuple get_losses returns (its loss_l1 is the constant 0.0) and the divisor.
 * _bwd: grad_result (4+) f64 on the device = upstream gradient of result[0..3]; grad_raw: HOST array of n_levels device
 *   pointers, each receives the (B, h, w, 5 + nc) f32 gradient of its level (every element written).
 * Dtypes as in the reference: IoU and class terms float64 (float64 labels), objectness float32; sums in float64 in a
 * fixed order (bit-reproducible). */
size_t frlw_yolox_loss_workspace_bytes(int B, int A, int G);
int frlw_yolox_loss_fwd(const float *const *raw, const int32_t *h, const int32_t *w, const float *strides, int n_levels,
                        int B, int num_classes, const double *labels, int G, float radius, float *preds, uint8_t *fg,
                        int32_t *matched_gt, double *matched_iou, double *result, void *workspace,
                        size_t workspace_bytes, frlw_stream_t stream);
int frlw_yolox_loss_bwd(const float *const *raw, const int32_t *h, const int32_t *w, const float *strides, int n_levels,
                        int B, int num_classes, const double *labels, int G, const uint8_t *fg,
                        const int32_t *matched_gt, const double *matched_iou, const double *result,
                        const double *grad_result, float *const *grad_raw, frlw_stream_t stream);

/* Sample transform of the training loader for a batch (data/dataset.py:217-231 in propheseeDataset.__getitem__):
 * (B, C, H, W) uint8 -> (B, C, H, W) f32 = flip(crop(nearest_resize(x, (Hr, Wr)) / 255)).
 * params (device): B rows of five int32 {Hr = int(H * sr), Wr = int(W * sr), y0 = -cy, x0 = -cx, flip}. */
int frlw_sample_transform_u8(const uint8_t *in, int B, int C, int H, int W, const int32_t *params, float *out,
                             frlw_stream_t stream);

/* Evaluator hand-off for the detections of a batch (evaluate/evaluator.py:56-63 transform_dt, and the mask of
 * evaluate/src/io/box_filtering.py:17-39): n packed rows [cx, cy, w, h, class, score] (detector pixels) ->
 * out (n, 8) f32 [t, x, y, w, h, class, score, 0] (sensor pixels; t = timestamps[img_of_row[i]]) and
 * keep[i] = t > skip_ts && w^2 + h^2 >= min_diag_sq && w >= min_w && h >= min_h. */
int frlw_eval_transform_dt(const float *dets, const int32_t *img_of_row, const int64_t *timestamps, int64_t n, float rw,
                           float rh, float skip_ts, float min_diag_sq, float min_w, float min_h, float *out,
                           uint8_t *keep, frlw_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Training-mode BaseConv = Conv2d(bias=False) + BatchNorm2d(batch statistics) + SiLU
 * (core/yolox/models/network_blocks.py:33-65) for the train step of core/exp.py:283-315: forward, data gradient,
 * weight gradient, BatchNorm + SiLU forward / backward.  All tensors NHWC float32 (= torch channels_last storage),
 * dense (pixel stride = channels).  Cin % 4 == 0 and Cout % 4 == 0.  `scratch` buffers are caller-owned.
 * `precision` (every contraction below): 0 = float32 MFMA, 1 = float32 products from three bf16 MFMAs (see
 * frlw_det_set_precision; the weight operands are then split images of ceil16(K) rows -- frlw_conv_operand_floats --
 * and the data-gradient operand of a parity-grouped layer needs Cout % 16 == 0, else FRLW_ERR_UNSUPPORTED: use 0 there).
 * ------------------------------------------------------------------------------------------- */
/* floats (4-byte units) of a GEMM operand with K rows and N columns: rows(K, precision) * pad32(N). */
int64_t frlw_conv_operand_floats(int K, int N, int precision);
/* torch weight (Cout, Cin, k, k) -> forward operand (k*k*Cin, pad32(Cout)) and / or data-gradient operand
 * (k*k*Cout, pad32(Cin)) with flipped taps; with dgrad_parity = frlw_conv2d_dgrad_parity(k, stride, H, W) != 0 the
 * rows of the latter are grouped by output parity class (see frlw_conv2d_dgrad).  pad32(n) = n rounded up to 32.
 * Either output may be NULL. */
int frlw_conv2d_dgrad_parity(int k, int stride, int H, int W);
int frlw_conv_weight_layouts(const float *w, int Cout, int Cin, int k, int dgrad_parity, float *w_fwd, float *w_dgrad,
                             int precision, frlw_stream_t stream);
/* The same for MANY weights in one launch (the ~74 BaseConv weights of the detector after an optimizer step: 74 launches
 * of ~5 us each otherwise).  `items`: DEVICE array of n entries, `first` = running sum of the entries' element counts
 * (frlw_conv_operand_floats(k*k*Cin, Cout, precision) + frlw_conv_operand_floats(k*k*Cout, Cin, precision) each; either
 * operand pointer may be NULL and then counts 0), `total` = the sum. */
typedef struct frlw_weight_layout_item {
    const float *w;
    float *w_fwd, *w_dgrad;
    int32_t Cout, Cin, k, dgrad_parity;
    int32_t precision, reserved;
    int64_t first;
    const float *w2;   /* split > 0: output channels [split, Cout) come from this second weight (Cout - split, Cin, k, k): the */
    int32_t split, reserved2; /* operands of two BaseConvs stacked along the output channels (frlw_baseconv_fuse_t::split) */
} frlw_weight_layout_item_t;
int frlw_conv_weight_layouts_batch(const frlw_weight_layout_item_t *items, int n, int64_t total, frlw_stream_t stream);
/* z (B, Ho, Wo, Cout) = conv2d(x (B, H, W, Cin), w), padding (k - 1) / 2, stride 1 or 2.  scratch: optional split-K
 * partial sums (scratch_floats floats; NULL = never split). */
int frlw_conv2d_fwd(const float *x, int B, int H, int W, int Cin, const float *w_fwd, int Cout, int k, int stride, float *z,
                    float *scratch, int64_t scratch_floats, int precision, frlw_stream_t stream);
/* dx (B, H, W, Cin) = gradient of the convolution above with respect to x, from dz (B, Ho, Wo, Cout).  Stride 2 with
 * k = 3 and even H, W (frlw_conv2d_dgrad_parity) runs as four stride-1 convolutions, one per output parity class
 * (1 / 2 / 2 / 4 taps) and needs the parity-grouped operand; other stride-2 shapes use a transposed gather. */
int frlw_conv2d_dgrad(const float *dz, int B, int Ho, int Wo, int Cout, const float *w_dgrad, int Cin, int k, int stride,
                      int H, int W, float *dx, float *scratch, int64_t scratch_floats, int precision, frlw_stream_t stream);
/* dw in torch's (Cout, Cin, k, k) layout = gradient with respect to the weight; scratch is REQUIRED
 * (frlw_conv2d_wgrad_scratch_floats floats for full parallelism; fewer = fewer splits). */
int64_t frlw_conv2d_wgrad_scratch_floats(int B, int Ho, int Wo, int Cin, int Cout, int k);
int frlw_conv2d_wgrad(const float *x, int B, int H, int W, int Cin, const float *dz, int Ho, int Wo, int Cout, int k,
                      int stride, float *dw, float *scratch, int64_t scratch_floats, int precision, frlw_stream_t stream);
/* Per-channel batch statistics of z viewed as (M, C): mean, biased variance, invstd = 1 / sqrt(var + eps).
 * scratch: frlw_bn_scratch_doubles(M, C) doubles. */
int64_t frlw_bn_scratch_doubles(int64_t M, int C);
int frlw_bn_stats(const float *z, int64_t M, int C, float eps, float *mean, float *var, float *invstd, double *scratch,
                  frlw_stream_t stream);
/* y = silu(gamma * (z - mean) * invstd + beta) */
int frlw_bn_silu_fwd(const float *z, int64_t M, int C, const float *gamma, const float *beta, const float *mean,
                     const float *invstd, float *y, frlw_stream_t stream);
/* dz, dgamma (C), dbeta (C) from dy and the saved z; sums: 2 C floats of scratch.  dy_row_stride: floats between two rows
 * of dy (0 or C: dense; larger: dy is a channel slice of a wider NHWC gradient -- what the backward of a concatenation hands
 * out -- read in place; multiple of 4, dy 16-byte aligned). */
int frlw_bn_silu_bwd(const float *dy, int64_t dy_row_stride, const float *z, int64_t M, int C, const float *gamma, const float *beta,
                     const float *mean, const float *invstd, float *dz, float *dgamma, float *dbeta, double *scratch,
                     float *sums, frlw_stream_t stream);

/* The whole block in one call per direction (what frlw-evd_amd/yolox/train_ops.py uses): forward = weight layout +
 * convolution + batch statistics + BatchNorm/SiLU; backward = BatchNorm/SiLU backward + data gradient (dx may be NULL)
 * + weight gradient.  x (B, H, W, Cin), z / y / dy / dz (B, Ho, Wo, Cout) NHWC; w and dw in torch's (Cout, Cin, k, k).
 * mean / var (biased) / invstd: (Cout).  running_mean / running_var (may be NULL) are updated in place like
 * nn.BatchNorm2d: r = (1 - momentum) r + momentum * {mean, unbiased variance}; num_batches_tracked (device int64, may be
 * NULL) is incremented by one in the same launch sequence.
 * scratch: frlw_baseconv_train_scratch_bytes(...) bytes, reusable between calls.
 * w_cache (may be NULL): frlw_baseconv_weight_cache_floats(Cin, Cout, k, precision) floats owned by the caller, one per layer: the
 * forward then lays the weight out for itself AND for the data gradient in one launch, and the backward of the same
 * step (weights unchanged in between) reuses it instead of laying the weight out again.  w == NULL with a w_cache in
 * the forward: the cache holds both operands of the current weights already (frlw_conv_weight_layouts_batch: forward
 * operand first, the data-gradient operand frlw_conv_operand_floats(k*k*Cin, Cout, precision) floats behind it) and no layout kernel is launched.
 * splitk_counters (may be NULL): 1024 ints owned by the caller, ZERO when first handed over and left alone by the caller
 * afterwards -- one buffer for all layers and calls of a stream.  With it, a convolution that splits its contraction over
 * blockIdx.z (thin layers: few output tiles, long contractions) sums the splits INSIDE the kernel -- the split that arrives
 * last at its tile's counter adds all of them in split order, the same bits as the separate reduction launch, and sets the
 * counter back to zero -- instead of launching k_splitk_reduce behind it (NULL).  frlw_det_run does the same with the last
 * 1024 words of every lane's scratch region (frlw_det_set_scratch: the caller hands the buffer over zeroed).
 * fuse (may be NULL): what the blocks AROUND this BaseConv would otherwise do in launches of their own (round 6) --
 *   residual: y = silu(bn(conv(x))) + residual -- the Bottleneck shortcut (network_blocks.py:109-111) in the pass that writes y;
 *   y_row_stride: y is a channel slice of a wider NHWC tensor (the concatenation buffer of a CSPLayer, network_blocks.py:191-193):
 *     the block writes its output where the concatenation would have copied it;
 *   dx_add: dx = data gradient + dx_add in the convolution's epilogue -- the gradient another consumer of the same x has
 *     produced already (the shortcut's dy, the sibling 1x1 branch of a CSPLayer): autograd's accumulation launch is gone.
 *     Stride-1 layers only (FRLW_ERR_UNSUPPORTED otherwise).
 *   Row strides in floats, multiples of 4, 0 = dense; pointers 16-byte aligned.  The sums are the same IEEE additions the separate
 *   launches made (a + b = b + a): results are bit-identical to the unfused sequence. */
typedef struct frlw_baseconv_fuse {
    int32_t struct_size;         /* sizeof(frlw_baseconv_fuse_t) */
    int32_t reserved;
    const float *residual;       /* forward; (B, Ho, Wo, Cout) rows */
    int64_t residual_row_stride;
    int64_t y_row_stride;        /* forward */
    const float *dx_add;         /* backward; (B, H, W, Cin) rows */
    int64_t dx_add_row_stride;
    /* split > 0: the call is TWO BaseConvs reading the same x (conv1 | conv2 of a CSPLayer, network_blocks.py:191-193), stacked
     * along the output channels: Cout = both blocks' channels, channels [0, split) are the first block's (w, gamma, beta, running
     * statistics, num_batches_tracked, y, dy: the call's own arguments; their rows are `split` wide by default), channels
     * [split, Cout) the second block's (the fields below).  ONE convolution, ONE statistics / BatchNorm pass, ONE data gradient
     * (dx is the sum over both blocks by construction) and ONE weight gradient do the work of two of each; z, dz, mean, var,
     * invstd, dw, dgamma, dbeta are stacked (Cout wide; dw = (Cout, Cin, k, k): the second block's rows start at split).
     * split % 4 == 0; no residual.  w_cache (when the caller lays the operands out itself): the operands of the STACKED weight. */
    int32_t split, reserved2;
    const float *w2;             /* (Cout - split, Cin, k, k) */
    const float *gamma2, *beta2;
    float *running_mean2, *running_var2;
    int64_t *num_batches_tracked2;
    float *y2;                   /* forward */
    int64_t y2_row_stride;
    const float *dy2;            /* backward */
    int64_t dy2_row_stride;
} frlw_baseconv_fuse_t;
int64_t frlw_baseconv_weight_cache_floats(int Cin, int Cout, int k, int precision);
int64_t frlw_baseconv_train_scratch_bytes(int B, int H, int W, int Cin, int Cout, int k, int stride);
int frlw_baseconv_train_fwd(const float *x, const float *w, const float *gamma, const float *beta, float eps, int B, int H,
                            int W, int Cin, int Cout, int k, int stride, float *z, float *y, float *mean, float *var,
                            float *invstd, float *running_mean, float *running_var, float momentum,
                            int64_t *num_batches_tracked, float *w_cache, void *scratch, int64_t scratch_bytes,
                            int *splitk_counters, const frlw_baseconv_fuse_t *fuse, int precision, frlw_stream_t stream);
int frlw_baseconv_train_bwd(const float *dy, int64_t dy_row_stride, const float *x, const float *z, const float *w, const float *gamma,
                            const float *beta, const float *mean, const float *invstd, int B, int H, int W, int Cin,
                            int Cout, int k, int stride, float *dz, float *dx, float *dw, float *dgamma, float *dbeta,
                            const float *w_cache, void *scratch, int64_t scratch_bytes, int *splitk_counters,
                            const frlw_baseconv_fuse_t *fuse, int precision, frlw_stream_t stream);

/* Measurement aid: a bare loop of v_mfma_f32_32x32x2_f32 (the instruction of every convolution here) on `blocks`
 * workgroups of four wavefronts, iters x 32 MFMAs (= iters x 131072 FLOP) per wavefront, operands = the 256 floats of
 * `seed` (device).  Timed by the caller (bench.py): the fp32 matrix rate the chip SUSTAINS on non-trivial data, which is
 * what the convolutions can approach -- the data-sheet peak assumes the maximum clock. */
int frlw_selftest_mfma_f32_rate(int blocks, int iters, const float *seed, float *sink, frlw_stream_t stream);

/* Library identification: "frlw_evd <version> gfx950". */
const char *frlw_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FRLW_EVD_H */
