// frlw_common.h -- shared device helpers of the event encoders (gfx950, wave64).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>

#include "frlw_evd.h"

namespace frlw {

constexpr int kWave = 64;

// Inclusive prefix sum over the 64 lanes of a wavefront in the vector ALU (DPP row shifts inside the rows of 16 lanes, then
// the row totals by row_bcast15 / row_bcast31): six adds, no LDS crossbar round trips (a __shfl_up scan is six dependent
// ds_bpermute).  All lanes must be active.
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); // row_bcast15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); // row_bcast31 into rows 2 and 3
    return v;
}
// Maximum over the 64 lanes of a wavefront, returned wave-uniform (same DPP ladder, v_max instead of v_add; the total
// ends up in lane 63).  All lanes must be active.
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// A tile is (1 << twl) pixels wide and 8 rows high; twl = 8 for frames wider than 512 px, else 6
// (measured best on MI355X; frlw_tuning_t::tile_width_log2 overrides for experiments).
// One tile = one workgroup of the tile kernels = NT = 4 << twl threads owning 4 cells each
// (cell = (pixel, polarity)).  A tile row is a whole number of wavefronts of consecutive cells,
// so the (H, W, 2, K) state and the (C, H, W) outputs are read / written in full lines.
constexpr int kTileH = 8;
constexpr int kTileHLog = 3;
constexpr int kCellsPerThread = 4;
constexpr int kMaxTiles = 2048;   // LDS of the scatter workgroup: 52 B per tile
constexpr int kMaxTlut = 1 << 16; // longest TAF window (us) served by the value table
constexpr int kPartThreads = 1024; // partition workgroup = 16 wavefronts
constexpr int kPartWaves = kPartThreads / kWave;
constexpr int kMaxBpw = 8;        // batches of 64 events per wavefront per workgroup chunk (registers!)
constexpr int kSlabUnits = 32;    // workgroups per slab of the two-level column scan

enum Kind : int { KIND_ECI = 0, KIND_EV = 1, KIND_SAE = 2, KIND_TAF = 3 };
enum : int { ST_INDEX = 1, ST_POLARITY = 2, ST_SPAN = 4, /* 8: taf_fast.hip ST_MULBAD (not an error) */ ST_STALL = 16 /* a bounded device-side wait ran out: FRLW_ERR_HIP */ };

// First kHeaderBytes of the workspace.
constexpr int kMaxHot = 32;     // tiles per encode whose cells are split over several workgroups (skew)
constexpr int kSliceMult = 16;  // records per thread in one LDS counting-sort slice of the EV / TAF tile kernels
struct WsHeader {
    int32_t status; // ST_* flags
    uint32_t pad;
    unsigned long long wmask; // bit w set <=> TAF window w holds at least one encoded event
    uint32_t n_hot;           // tiles with more than hot_thr records (may exceed kMaxHot; only the listed ones are split)
    uint32_t hot_thr;
    uint32_t hot[kMaxHot];
};
constexpr size_t kHeaderBytes = 1024;
static_assert(sizeof(WsHeader) <= kHeaderBytes, "header");
// Two words at the end of the header outlive the per-call reset: `sticky` is the OR of the status of every encoder call
// since frlw_workspace_init / the last frlw_encoder_deferred_status (unchecked callers read it once per batch or epoch
// instead of synchronising after every call); the 24 bytes in front of it receive the one-time LDS self-test result.
constexpr size_t kStickyOffset = kHeaderBytes - 8;
// ... and `stall`, 8 bytes in front of `sticky`: the epoch of a fast-path call one of whose workgroups gave up its bounded wait
// for the header reset (ST_STALL).  It lies OUTSIDE the range workgroup 0 resets, so a workgroup 0 that starts late cannot
// wipe the verdict of its own call; workgroup 0 (and the captured form's reset kernel) clears it only when it belongs to an
// EARLIER call.  frlw_encoder_status reports FRLW_ERR_HIP while it is set.
constexpr size_t kStallOffset = kHeaderBytes - 16;
constexpr size_t kSelftestOffset = kHeaderBytes - 64;
static_assert(sizeof(WsHeader) <= kSelftestOffset, "header tail");
__device__ __forceinline__ void fold_sticky_status(void *hdr, int status)
{
    if (status) atomicOr((int *)((char *)hdr + kStickyOffset), status);
}

// How one event becomes (tile, cell, window, value).  Passed by value to the kernels.
struct Decode {
    const void *data;
    long long n;
    int row_stride;
    const uint16_t *xmap;
    const uint16_t *ymap;
    int map_w, map_h;
    int H, W;
    int twl;            // log2 of the tile width
    int tiles_x, n_tiles;
    long long t0;       // EV: t_end - window; SAE: now - window; TAF: t_start
    long long win;      // EV: window; TAF: window_us
    int n_windows;      // TAF
    int time_filter;    // DAT8 EV / SAE: drop t <= t0
    uint32_t win_magic; // floor(2^32 / win), TAF DAT8
    const float *tlut;  // TAF DAT8: tlut[r] = float(r / (win + 1e-8)) - 1 for r in [0, win], or NULL
};

struct Pos {
    int tile;      // < 0: not encoded
    uint32_t cell; // ((ly << twl | lx) << 1) | p
    int err;
};

// Bounds / polarity handling common to all layouts.
template <int KIND>
__device__ __forceinline__ Pos place(const Decode &P, int x, int y, int p)
{
    Pos r;
    r.tile = -1; r.cell = 0; r.err = 0;
    if ((unsigned)p > 1u) { r.err = ST_POLARITY; return r; }
    if ((unsigned)x >= (unsigned)P.W || (unsigned)y >= (unsigned)P.H) {
        // The reference indexes the FLAT pixel x + W*y (generate_eventvolume.py:32): x >= W aliases
        // into the next row and only a flat index outside [0, H*W) raises.  index_put_ (SAE,
        // generate_surfaceofactiveevents.py:49) checks each axis.
        if (KIND == KIND_SAE) { r.err = ST_INDEX; return r; }
        long long flat = (long long)x + (long long)P.W * y;
        if (flat < 0 || flat >= (long long)P.H * P.W) { r.err = ST_INDEX; return r; }
        y = (int)(flat / P.W);
        x = (int)(flat - (long long)y * P.W);
    }
    r.tile = (y >> kTileHLog) * P.tiles_x + (x >> P.twl);
    r.cell = (uint32_t)((((y & (kTileH - 1)) << P.twl) | (x & ((1 << P.twl) - 1))) << 1) | (uint32_t)p;
    return r;
}

// Position of a raw DAT record (src/io/dat_events_tools.py:96-98) incl. the optional coordinate
// maps and the time / range filters that drop an event without an error.
template <int KIND, bool HAS_MAP = true>
__device__ __forceinline__ Pos dat_pos(const Decode &P, uint2 r)
{
    int x = (int)(r.y & 16383u), y = (int)((r.y >> 14) & 16383u), p = (int)((r.y >> 28) & 1u);
    Pos o;
    o.tile = -1; o.cell = 0; o.err = 0;
    if (HAS_MAP && P.xmap) {
        if (x >= P.map_w || y >= P.map_h) { o.err = ST_INDEX; return o; }
        x = P.xmap[x];
        y = P.ymap[y];
    }
    if (KIND == KIND_SAE && (x >= P.W || y >= P.H)) return o; // generate_surfaceofactiveevents.py:72
    if ((KIND == KIND_EV || KIND == KIND_SAE) && P.time_filter && !((long long)r.x > P.t0)) return o;
    return place<KIND>(P, x, y, p);
}

// Window and f32 value of a raw DAT record.
template <int KIND>
__device__ __forceinline__ void dat_value(const Decode &P, uint2 r, int &window, float &val)
{
    window = 0;
    val = 0.0f;
    if (KIND == KIND_EV) {
        val = (float)((double)((long long)r.x - P.t0) / (double)P.win); // generate_eventvolume.py:141
    } else if (KIND == KIND_SAE) {
        val = (float)r.x; // float(t), generate_surfaceofactiveevents.py:76
    } else if (KIND == KIND_TAF) {
        // generate_taf.py:197-203: z = the last window i with start + i*w <= t <= start + (i+1)*w, else 0
        const long long rel = (long long)r.x - P.t0;
        const uint32_t winu = (uint32_t)P.win;
        double num;
        if (rel >= 0 && rel <= (long long)P.n_windows * P.win) {
            const uint32_t relu = (uint32_t)rel;
            uint32_t z = __umulhi(relu, P.win_magic); // floor(rel / win) - {0, 1, 2}
            uint32_t rem = relu - z * winu;
            if (rem >= winu) { ++z; rem -= winu; }
            if (rem >= winu) { ++z; rem -= winu; }
            if (z >= (uint32_t)P.n_windows) { z = (uint32_t)P.n_windows - 1u; rem = winu; } // t == end of the last window
            window = (int)z;
            if (P.tlut) { val = P.tlut[rem]; return; } // same f64 division, done once per distinct value
            num = (double)rem;
        } else {
            num = (double)rel; // outside the span: window 0, normalised time outside [0, 1]
        }
        const double tn = num / ((double)P.win + 1e-8); // generate_taf.py:215
        val = (float)tn - 1.0f;                         // t - 1, generate_taf.py:26
    }
}

// The reference's device tensor: (N, row_stride) float64 rows [x, y, t, p, ...].
template <int KIND>
__device__ __forceinline__ Pos f64_pos(const Decode &P, long long i, double &t)
{
    const double *r = (const double *)P.data + i * (long long)P.row_stride;
    const double xd = r[0], yd = r[1], pd = r[3];
    t = r[2];
    Pos o;
    o.tile = -1; o.cell = 0; o.err = 0;
    if (KIND == KIND_SAE && !(xd < (double)P.W && yd < (double)P.H)) return o;
    if (!(fabs(xd) < 1.0e9) || !(fabs(yd) < 1.0e9) || !(fabs(pd) < 1.0e9)) { o.err = ST_INDEX; return o; }
    return place<KIND>(P, (int)xd, (int)yd, (int)pd); // .long(): truncation toward zero
}

template <int KIND>
__device__ __forceinline__ float f64_value(double t)
{
    if (KIND == KIND_TAF) return (float)t - 1.0f; // generate_taf.py:26
    if (KIND == KIND_ECI) return 0.0f;
    return (float)t;
}

// Workgroups are handed to the 8 XCDs round-robin (block b runs on XCD b % 8) and every XCD has its own L2.  Chunk
// c of the stream is therefore given to block (c - start_k) * 8 + k with k the XCD that owns the contiguous chunk range
// [start_k, start_{k+1}): consecutive chunks -- whose runs are neighbours in every tile's record list -- are written
// through the SAME L2, which merges the lines they share before they reach the HBM.
__device__ __forceinline__ long long chunk_of_block(unsigned block, unsigned n_blocks)
{
#ifdef FRLW_NO_XCD_REMAP
    return block;
#else
    const unsigned k = block & 7u, idx = block >> 3, q = n_blocks >> 3, r = n_blocks & 7u;
    return (long long)k * q + (k < r ? k : r) + idx;
#endif
}

__device__ __forceinline__ uint64_t lanemask_lt()
{
    return (1ull << (threadIdx.x & 63)) - 1ull;
}

struct TileGeom {
    int x0, y0, nx, ny; // nx, ny: valid pixels of this tile
};

__device__ __forceinline__ TileGeom tile_geom(int tile, int tiles_x, int twl, int H, int W)
{
    TileGeom g;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int tw = 1 << twl;
    g.x0 = tx << twl;
    g.y0 = ty << kTileHLog;
    g.nx = W - g.x0 < tw ? W - g.x0 : tw;
    g.ny = H - g.y0 < kTileH ? H - g.y0 : kTileH;
    return g;
}

// float -> uint8 the way numpy's .astype(np.uint8) does it on the reference's x86 hosts: truncate to int32
// (cvttss2si: NaN / inf / out-of-range give INT_MIN), then keep the low byte.  v_cvt_i32_f32 saturates
// instead, so the out-of-range case is made explicit.  (Only reachable with state values > 0, i.e. events
// the harness placed after the last window.)
__device__ __forceinline__ uint8_t f32_to_u8(float v)
{
    const int i = (v >= -2147483648.0f && v < 2147483648.0f) ? (int)v : (int)0x80000000;
    return (uint8_t)i;
}

__device__ __forceinline__ float leaky_f(float v)
{
    float l = log1pf(-v);   // generate_taf.py:72
    l = 1.0f - l / 8.7f;    // :73
    if (l < 0.0f) l = 0.0f; // :74
    return l * 255.0f;      // :75
}

// uint8(leaky_transform(v)) is a non-increasing step function of x = -v >= 0 with at most 256 levels.
// thr[k] (k = 1..255) = bit pattern of the largest x with uint8(leaky_f(-x)) >= k, found by bisection on the
// SAME f32 pipeline (k_hist builds the table once per encode); thr[0] = +inf.  The lookup below returns
// exactly what evaluating leaky_f would, for ~10 instructions instead of a log1pf per value.
constexpr int kLeakyLevels = 256;

__device__ __forceinline__ uint32_t leaky_threshold_bits(int k)
{
    uint32_t lo = 0u, hi = 0x7f000000u; // f(0) = 255 >= k always; f(huge) = 0
    while (hi - lo > 1u) {
        const uint32_t mid = lo + (hi - lo) / 2u;
        if ((int)leaky_f(-__uint_as_float(mid)) >= k) lo = mid; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ uint8_t leaky_u8_lookup(float v, const uint32_t *thr)
{
    if (!(v <= 0.0f)) return f32_to_u8(leaky_f(v)); // outside the table's domain (events after the last window)
    const float x = fabsf(v); // v <= 0 here; fabsf also maps +0 to +0 (-(+0) = -0 would compare as a huge bit pattern)
    const uint32_t xb = __float_as_uint(x);
    // first guess from the hardware log2 (1 ulp-ish), then walk to the exact level
    float g = 255.0f * (1.0f - (__log2f(1.0f + x) * 0.69314718f) / 8.7f);
    int k = g < 0.0f ? 0 : (g > 255.0f ? 255 : (int)g);
    while (k < 255 && xb <= thr[k + 1]) ++k;
    while (k > 0 && xb > thr[k]) --k;
    return (uint8_t)k;
}

// The 256 thresholds are a constant of the device's f32 pipeline: built ONCE per device and process (partition.hip) into a
// device-resident table, on the stream of the first encoder call; calls on other streams wait for that fill through an
// event until it has completed (no host synchronisation; inside a stream capture the fill becomes a node of the graph).
// Until round 3 every TAF encode rebuilt the table in the last workgroup of its histogram kernel: 31 bisection steps of
// log1pf on one wavefront, ~8 us on the critical path of every small call.  nullptr: a HIP call failed.
const uint32_t *leaky_table(hipStream_t s);

// Out-of-line on purpose: it is the rare path of the batched look-up below, which would otherwise inline
// N copies of log1pf per cell.
__device__ __noinline__ uint8_t leaky_u8_exact(float v, const uint32_t *thr) { return leaky_u8_lookup(v, thr); }

// N look-ups at once: the first guess is off by at most one level, so thr[k] and thr[k + 1] decide it; both
// are read for all N values before any is used.  The (rare) values the pair does not pin down, and those
// outside the table's domain, take the exact scalar routine.
template <int N>
__device__ __forceinline__ void leaky_u8_lookup_n(const float (&v)[N], const uint32_t *thr, uint8_t (&out)[N])
{
    int k[N];
    uint32_t t0[N], t1[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float x = fabsf(v[i]);
        // first guess only (the table pair below decides): multiply instead of the exact routine's divide
        const float g = 255.0f - (__log2f(1.0f + x) * (255.0f * 0.69314718f / 8.7f));
        k[i] = g > 0.0f ? (g > 254.0f ? 254 : (int)g) : 0; // k + 1 stays inside the table
    }
#pragma unroll
    for (int i = 0; i < N; ++i) { t0[i] = thr[k[i]]; t1[i] = thr[k[i] + 1]; }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint32_t xb = __float_as_uint(fabsf(v[i]));
        // level k exactly: thr[k + 1] < x <= thr[k]  (thr[0] = +inf)
        out[i] = (uint8_t)k[i];
        if (!(v[i] <= 0.0f && xb > t1[i] && xb <= t0[i])) out[i] = leaky_u8_exact(v[i], thr);
    }
}

// The same step function through a bucket table (kf_taf_walk): the level changes by 20.3 per octave of 1 + x at most, so a bucket
// of 1/32 octave of x (the float's exponent and five mantissa bits) holds at most ONE threshold.  lut[b] = the level at the
// bucket's upper end; the level of x is lut[b] + (x <= thr[lut[b] + 1]): one byte and one word from LDS, a compare and an add --
// no transcendental, no first guess to repair.  Buckets start at 2^-6 (everything below shares bucket 0: level 254, or 255 for
// x = 0, which thr[255] = 0 decides) and end at 2^13 (level 0 from 6002 on).  k_leaky_fill builds lut[] from thr[] itself and
// CHECKS the one-threshold property for every bucket on the device's own table (word kLeakyOkWord); a device whose log1pf broke
// it would take the exact routine for every value.
constexpr int kLeakyBucketShift = 18;                         // 5 mantissa bits
constexpr int kLeakyBucket0 = (127 - 6) << 5;                 // bits(2^-6) >> 18
constexpr int kLeakyBuckets = ((127 + 13) << 5) - kLeakyBucket0; // 608
constexpr int kLeakyLutWord = kLeakyLevels;                   // lut bytes, four per word, behind the 256 thresholds
constexpr int kLeakyOkWord = kLeakyLutWord + kLeakyBuckets / 4;
constexpr int kLeakyTableWords = kLeakyOkWord + 1;            // 409 words: what leaky_table() points at

template <int N>
__device__ __forceinline__ void leaky_u8_bucket_n(const float (&v)[N], const uint32_t *tab, uint8_t (&out)[N])
{
    const uint8_t *lut = (const uint8_t *)(tab + kLeakyLutWord);
    const bool lut_ok = tab[kLeakyOkWord] != 0u; // (uniform)
    uint32_t xb[N], k[N], t1[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        xb[i] = __float_as_uint(fabsf(v[i]));
        int b = (int)(xb[i] >> kLeakyBucketShift) - kLeakyBucket0;
        b = b < 0 ? 0 : (b > kLeakyBuckets - 1 ? kLeakyBuckets - 1 : b);
        k[i] = lut[b];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) t1[i] = tab[k[i] + 1u];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        out[i] = (uint8_t)(k[i] + (xb[i] <= t1[i] ? 1u : 0u));
        if (!(v[i] <= 0.0f) || !lut_ok) out[i] = leaky_u8_exact(v[i], tab); // outside the table's domain (rare), or no valid lut
    }
}

// ---- host side ---------------------------------------------------------------------------------
struct Plan {
    int twl, tiles_x, tiles_y, n_tiles;
    int bpw;          // batches of 64 events per wavefront
    long long chunk;  // events per partition workgroup = 1024 * bpw
    int units, slabs; // partition workgroups, slabs of 32
    unsigned hot_thr; // a tile with more records than this is shared by several workgroups (EV / TAF)
    int staged;       // < 0: by size
    int quarter_below;
    int no_lut;
    size_t off_counts, off_slabtot, off_base, off_errs, off_tlut, off_records, bytes;
};

struct Partitioned {
    const uint2 *records;
    const uint32_t *base;
    WsHeader *hdr;
    const uint32_t *leaky_thr; // TAF: threshold table built by k_hist
    Plan plan;
};

bool make_plan(long long n, int H, int W, const frlw_tuning_t *tuning, Plan &p);

// One knob of the caller's frlw_tuning_t: a field that lies outside the caller's `struct_size` (an older header) or is
// negative takes the default.
inline int tuning_knob(const frlw_tuning_t *tu, int32_t frlw_tuning_t::*f, int dflt)
{
    if (!tu) return dflt;
    const size_t end = (size_t)((const char *)&(tu->*f) - (const char *)tu) + sizeof(int32_t);
    if ((size_t)tu->struct_size < end) return dflt;
    const int v = (int)(tu->*f);
    return v >= 0 ? v : dflt;
}
inline bool tuning_valid(const frlw_tuning_t *tu) { return !tu || (tu->struct_size >= 8 && tu->struct_size <= 4096); }
int hip_fail(hipError_t e, const char *what, int line);

// frlw_sae_encode through the chunk-major partition of taf_fast.hip (two launches): FRLW_OK = launched, 1 = not eligible
// (nothing launched), negative = error.
int sae_fast_try(const frlw_events_t *ev, int H, int W, const float *lam, int n_lamda, const float *mem_in, float *mem_out,
                 long long now, long long window_us, float *out_f32, uint8_t *out_u8, void *workspace, size_t workspace_bytes,
                 hipStream_t st);

// Workspace bytes the two-launch form asks for (0: the call is not eligible) and the per-process launch counters behind
// frlw_encoder_path_counts: [0] SAE two-launch, [1] SAE general, [2] ECI two-launch, [3] ECI single-launch scan / general.
size_t sae_fast_workspace_bytes(long long n, int H, int W);
extern std::atomic<unsigned long long> g_path_counts[4];

// hist -> scans -> stable scatter: tile-major 8-byte records {window << (twl + 4) | cell, f32 bits}.
int partition_events(const frlw_events_t *ev, int H, int W, int kind, long long t0, long long win,
                     int n_windows, int time_filter, void *ws, size_t ws_bytes, hipStream_t s,
                     Partitioned &out);

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return frlw::hip_fail(e_, #expr, __LINE__); } while (0)

inline int grid_for(long long n, int block)
{
    long long g = (n + block - 1) / block;
    if (g > 256 * 8) g = 256 * 8;
    if (g < 1) g = 1;
    return (int)g;
}

} // namespace frlw
