// encoders.hip -- event-stream -> dense-tensor encoders for MI355X (gfx950, wave64).
//
// Replaces the bodies of the reference's four encoder functions (file:line in the reference):
//   generate_eventframe               generate_eventcountimage.py:19-41      (ECI)
//   generate_agile_event_volume_cuda  generate_eventvolume.py:15-42          (Event Volume)
//   generate_leaky_cuda / taf_cuda    generate_surfaceofactiveevents.py:44-80 (SAE)
//   generate_taf_cuda / taf_cuda      generate_taf.py:19-67, leaky_transform :69-76 (TAF)
// and, for raw DAT streams, the harness glue around them (generate_taf.py:197-227).
//
// The reference accumulates with torch index_add_; its defined result is the single-thread one:
// f32 adds applied in STREAM ORDER.  Float atomics cannot reproduce that, so the pipeline is
//
//   1. k_hist     every wave ("unit") owns a contiguous chunk of the stream and histograms it over
//                 32x8-pixel tiles in LDS                          -> counts[unit][tile]
//   2. k_colscan  per tile: exclusive prefix over units            -> counts[unit][tile], total[tile]
//      k_tilescan exclusive prefix over tiles                      -> tile_base[tile]
//   3. k_scatter  same chunks again: stable rank of every event inside (unit, tile) by wavefront
//                 ballots, 8-byte records {cell|window, f32 value} written tile-major.  The
//                 partition is STABLE, so each tile's records are still in stream order.
//   4. k_*_tile   one wavefront per tile keeps the tile's accumulators (and, for TAF, its K-deep
//                 FIFO state) in LDS, consumes its records 64 at a time; lanes that hit the same
//                 cell are found with ballots and applied in ascending lane order = stream order;
//                 the epilogue (scale / exp / log1p / uint8 truncation / layout permute) is fused
//                 into the tile write-out.
//
// Everything is integer / exact-f32 arithmetic in the reference's operation order; this file is
// compiled with -ffp-contract=off so no f32 multiply-add is ever fused.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "frlw_evd.h"

namespace {

constexpr int kWave = 64;
constexpr int kTileW = 32;   // pixels
constexpr int kTileH = 8;
constexpr int kTileWLog = 5;
constexpr int kTileHLog = 3;
constexpr int kTilePx = kTileW * kTileH;     // 256
constexpr int kTileCells = 2 * kTilePx;      // 512 (pixel, polarity)
constexpr int kCellBits = 9;
constexpr int kMaxTiles = 16384;             // LDS histogram of one wave: 64 KiB

enum Kind : int { KIND_ECI = 0, KIND_EV = 1, KIND_SAE = 2, KIND_TAF = 3 };

enum : int { ST_INDEX = 1, ST_POLARITY = 2 };

// First kHeaderBytes of the workspace.
struct WsHeader {
    int32_t status;   // ST_* flags
    uint32_t n_valid; // events that survived filtering
    uint32_t wcount[FRLW_MAX_WINDOWS];
};
constexpr size_t kHeaderBytes = 1024;
static_assert(sizeof(WsHeader) <= kHeaderBytes, "header");

// How one event is turned into (tile, cell, window, value).  Passed by value to the kernels.
struct Decode {
    const void *data;
    long long n;
    int layout;
    int row_stride;
    const uint16_t *xmap;
    const uint16_t *ymap;
    int map_w, map_h;
    int H, W;
    int tiles_x, n_tiles;
    int tile_bits;       // ceil(log2(n_tiles))
    int kind;
    long long t0;        // EV: t_end - window; SAE: now - window; TAF: t_start
    long long win;       // EV: window; TAF: window_us
    int n_windows;       // TAF
    int time_filter;     // DAT8 EV / SAE: drop t <= t0
};

struct Ev {
    int tile;       // -1: not encoded (filtered or invalid)
    uint32_t meta;  // window << 9 | cell, cell = ((ly << 5 | lx) << 1) | p
    float val;
    int window;
    int err;
};

__device__ __forceinline__ Ev decode_event(const Decode &P, long long i)
{
    Ev e;
    e.tile = -1; e.meta = 0; e.val = 0.0f; e.window = 0; e.err = 0;
    long long x, y, p;
    double t = 0.0;
    long long ti = 0;
    if (P.layout == FRLW_LAYOUT_XYTP_F64) {
        const double *r = (const double *)P.data + i * (long long)P.row_stride;
        double xd = r[0], yd = r[1];
        t = r[2];
        double pd = r[3];
        if (P.kind == KIND_SAE) { // generate_surfaceofactiveevents.py:72
            if (!(xd < (double)P.W && yd < (double)P.H)) return e;
        }
        x = (long long)xd; y = (long long)yd; p = (long long)pd; // .long(): toward zero
    } else {
        uint2 r = ((const uint2 *)P.data)[i];
        ti = (long long)r.x;
        x = (long long)(r.y & 16383u);
        y = (long long)((r.y >> 14) & 16383u);
        p = (long long)((r.y >> 28) & 1u);
        if (P.xmap) {
            if (x >= P.map_w || y >= P.map_h) { e.err = ST_INDEX; return e; }
            x = P.xmap[x];
            y = P.ymap[y];
        }
        if (P.kind == KIND_SAE && (x >= P.W || y >= P.H)) return e;
        if (P.time_filter && !(ti > P.t0)) return e;
    }
    if (p < 0 || p > 1) { e.err = ST_POLARITY; return e; }
    if (x < 0 || x >= P.W || y < 0 || y >= P.H) {
        // The reference indexes the FLAT pixel x + W*y (generate_eventvolume.py:32): x >= W
        // aliases into the next row and only a flat index outside [0, H*W) raises.
        if (P.kind == KIND_SAE) { e.err = ST_INDEX; return e; } // index_put_ checks each axis
        long long flat = x + (long long)P.W * y;
        if (flat < 0 || flat >= (long long)P.H * P.W) { e.err = ST_INDEX; return e; }
        y = flat / P.W;
        x = flat - y * P.W;
    }
    int window = 0;
    float val = 0.0f;
    if (P.kind == KIND_EV) {
        if (P.layout == FRLW_LAYOUT_XYTP_F64) val = (float)t;
        else val = (float)((double)(ti - P.t0) / (double)P.win); // generate_eventvolume.py:141
    } else if (P.kind == KIND_SAE) {
        val = (P.layout == FRLW_LAYOUT_XYTP_F64) ? (float)t : (float)(double)ti;
    } else if (P.kind == KIND_TAF) {
        if (P.layout == FRLW_LAYOUT_XYTP_F64) {
            val = (float)t - 1.0f; // generate_taf.py:26
        } else {
            // generate_taf.py:197-203: the last window i with start+i*w <= t <= start+(i+1)*w
            long long rel = ti - P.t0;
            if (rel >= 0 && rel <= (long long)P.n_windows * P.win) {
                long long z = rel / P.win;
                window = (int)(z < P.n_windows ? z : P.n_windows - 1);
            }
            double t_min = (double)(P.t0 + (long long)window * P.win);
            double tn = ((double)ti - t_min) / ((double)P.win + 1e-8); // generate_taf.py:215
            val = (float)tn - 1.0f;
        }
    }
    int xi = (int)x, yi = (int)y;
    e.tile = (yi >> kTileHLog) * P.tiles_x + (xi >> kTileWLog);
    uint32_t cell = (uint32_t)((((yi & (kTileH - 1)) << kTileWLog) | (xi & (kTileW - 1))) << 1) | (uint32_t)p;
    e.meta = ((uint32_t)window << kCellBits) | cell;
    e.val = val;
    e.window = window;
    return e;
}

__device__ __forceinline__ uint64_t lanemask_lt()
{
    return (1ull << (threadIdx.x & 63)) - 1ull;
}

// For every active lane: the mask of active lanes holding the same key (low `bits` bits).
template <int BITS>
__device__ __forceinline__ uint64_t match_any_fixed(uint32_t key, bool act)
{
    uint64_t m = __ballot(act);
#pragma unroll
    for (int b = 0; b < BITS; ++b) {
        bool bit = (key >> b) & 1u;
        uint64_t bal = __ballot(bit && act);
        m &= bit ? bal : ~bal;
    }
    return m;
}

__device__ __forceinline__ uint64_t match_any_var(uint32_t key, int bits, bool act)
{
    uint64_t m = __ballot(act);
    for (int b = 0; b < bits; ++b) {
        bool bit = (key >> b) & 1u;
        uint64_t bal = __ballot(bit && act);
        m &= bit ? bal : ~bal;
    }
    return m;
}

// ---------------------------------------------------------------------------------------------
// 1. per-unit tile histogram
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWave) void k_hist(Decode P, long long chunk, uint32_t *counts,
                                                 WsHeader *hdr)
{
    extern __shared__ uint32_t lds[];
    uint32_t *hist = lds;                       // [n_tiles]
    uint32_t *wc = lds + P.n_tiles;             // [FRLW_MAX_WINDOWS]
    const int lane = threadIdx.x;
    const long long unit = blockIdx.x;
    for (int b = lane; b < P.n_tiles + FRLW_MAX_WINDOWS; b += kWave) lds[b] = 0;
    __syncthreads();
    const long long begin = unit * chunk;
    long long end = begin + chunk;
    if (end > P.n) end = P.n;
    int err = 0;
    for (long long i0 = begin; i0 < end; i0 += kWave) {
        long long i = i0 + lane;
        if (i < end) {
            Ev e = decode_event(P, i);
            err |= e.err;
            if (e.tile >= 0) {
                atomicAdd(&hist[e.tile], 1u);
                if (P.kind == KIND_TAF) atomicAdd(&wc[e.window], 1u);
            }
        }
    }
    __syncthreads();
    uint32_t *row = counts + unit * (long long)P.n_tiles;
    for (int b = lane; b < P.n_tiles; b += kWave) row[b] = hist[b];
    if (P.kind == KIND_TAF && lane < P.n_windows && wc[lane]) atomicAdd(&hdr->wcount[lane], wc[lane]);
    if (err) atomicOr(&hdr->status, err);
}

// ---------------------------------------------------------------------------------------------
// 2. counts[unit][tile] -> exclusive prefix over units (in place) + total[tile]
//    block = 64 tiles x kSlabs unit-slabs
// ---------------------------------------------------------------------------------------------
constexpr int kSlabs = 16;

__global__ __launch_bounds__(kWave * kSlabs) void k_colscan(uint32_t *counts, int units,
                                                             int n_tiles, uint32_t *total)
{
    __shared__ uint32_t part[kSlabs][kWave];
    const int tl = threadIdx.x & (kWave - 1);
    const int slab = threadIdx.x >> 6;
    const int tile = blockIdx.x * kWave + tl;
    const int per = (units + kSlabs - 1) / kSlabs;
    const int u0 = slab * per;
    int u1 = u0 + per;
    if (u1 > units) u1 = units;
    uint32_t s = 0;
    if (tile < n_tiles)
        for (int u = u0; u < u1; ++u) s += counts[(long long)u * n_tiles + tile];
    part[slab][tl] = s;
    __syncthreads();
    uint32_t run = 0;
    for (int k = 0; k < slab; ++k) run += part[k][tl];
    if (tile < n_tiles) {
        for (int u = u0; u < u1; ++u) {
            uint32_t *c = &counts[(long long)u * n_tiles + tile];
            uint32_t v = *c;
            *c = run;
            run += v;
        }
        if (slab == kSlabs - 1) total[tile] = run;
    }
}

// exclusive scan of total[0..n) -> base[0..n], one block of 1024 threads
__global__ __launch_bounds__(1024) void k_tilescan(const uint32_t *total, int n, uint32_t *base)
{
    __shared__ uint32_t sums[1024];
    const int tid = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int b0 = tid * per;
    int b1 = b0 + per;
    if (b1 > n) b1 = n;
    uint32_t s = 0;
    for (int b = b0; b < b1; ++b) s += total[b];
    sums[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) { // Hillis-Steele inclusive scan
        uint32_t v = (tid >= off) ? sums[tid - off] : 0u;
        __syncthreads();
        sums[tid] += v;
        __syncthreads();
    }
    uint32_t run = sums[tid] - s;
    for (int b = b0; b < b1; ++b) {
        base[b] = run;
        run += total[b];
    }
    if (tid == 1023) base[n] = sums[1023];
}

// ---------------------------------------------------------------------------------------------
// 3. stable scatter into tile-major records
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWave) void k_scatter(Decode P, long long chunk,
                                                    const uint32_t *counts, const uint32_t *base,
                                                    uint2 *records)
{
    extern __shared__ uint32_t lds[];
    uint32_t *cursor = lds; // [n_tiles]: next free record slot of (this unit, tile)
    const int lane = threadIdx.x;
    const long long unit = blockIdx.x;
    const uint32_t *row = counts + unit * (long long)P.n_tiles;
    for (int b = lane; b < P.n_tiles; b += kWave) cursor[b] = base[b] + row[b];
    __syncthreads();
    const long long begin = unit * chunk;
    long long end = begin + chunk;
    if (end > P.n) end = P.n;
    const uint64_t lt = lanemask_lt();
    for (long long i0 = begin; i0 < end; i0 += kWave) {
        long long i = i0 + lane;
        Ev e;
        e.tile = -1;
        if (i < end) e = decode_event(P, i);
        const bool act = e.tile >= 0;
        const uint64_t m = match_any_var((uint32_t)e.tile, P.tile_bits, act);
        if (act) {
            const uint32_t rank = (uint32_t)__popcll(m & lt);
            const uint32_t size = (uint32_t)__popcll(m);
            const uint32_t pos = cursor[e.tile] + rank;
            records[pos] = make_uint2(e.meta, __float_as_uint(e.val));
            // the group's last lane publishes the advanced cursor after every member has read it
            if (rank + 1 == size) cursor[e.tile] = pos + 1;
        }
        // LDS operations of one wave execute in order, so the next iteration's reads of cursor[]
        // see this write; the read above happens-before the write because both are issued by
        // this wave in program order and the write is the last LDS instruction of the iteration.
    }
}

// ---------------------------------------------------------------------------------------------
// 4. tile kernels.  cell -> (ly, lx, p); pixel (x0 + lx, y0 + ly)
// ---------------------------------------------------------------------------------------------
struct TileGeom {
    int x0, y0, nx, ny; // nx, ny: valid pixels of this tile
};

__device__ __forceinline__ TileGeom tile_geom(int tile, int tiles_x, int H, int W)
{
    TileGeom g;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    g.x0 = tx << kTileWLog;
    g.y0 = ty << kTileHLog;
    g.nx = W - g.x0 < kTileW ? W - g.x0 : kTileW;
    g.ny = H - g.y0 < kTileH ? H - g.y0 : kTileH;
    return g;
}

// ---- ECI -------------------------------------------------------------------------------------
struct EciParams {
    int H, W, tiles_x;
    float lut[21]; // value * 255 after n sequential +0.05f adds, clamped (n >= 20 -> 255)
    float *out_f32;
    uint8_t *out_u8;
};

__global__ __launch_bounds__(kWave) void k_eci_tile(const uint2 *rec, const uint32_t *base,
                                                     EciParams q)
{
    __shared__ uint32_t cnt[kTileCells];
    const int lane = threadIdx.x, tile = blockIdx.x;
    for (int c = lane; c < kTileCells; c += kWave) cnt[c] = 0;
    __syncthreads();
    const uint32_t beg = base[tile], end = base[tile + 1];
    for (uint32_t i = beg + lane; i < end; i += kWave) atomicAdd(&cnt[rec[i].x & (kTileCells - 1)], 1u);
    __syncthreads();
    const TileGeom g = tile_geom(tile, q.tiles_x, q.H, q.W);
    const long long plane = (long long)q.H * q.W;
    for (int o = lane; o < kTileCells; o += kWave) { // o = (p, ly, lx): rows of 32 contiguous pixels
        const int lx = o & 31, ly = (o >> 5) & 7, p = o >> 8;
        if (lx < g.nx && ly < g.ny) {
            uint32_t n = cnt[((ly << 5 | lx) << 1) | p];
            float v = q.lut[n > 20u ? 20u : n];
            long long idx = p * plane + (long long)(g.y0 + ly) * q.W + g.x0 + lx;
            if (q.out_f32) q.out_f32[idx] = v;
            if (q.out_u8) q.out_u8[idx] = (uint8_t)(int)v;
        }
    }
}

// ---- Event Volume ----------------------------------------------------------------------------
struct EvParams {
    int H, W, tiles_x, bins;
    float *out_f32;
    uint8_t *out_u8;
};

__global__ __launch_bounds__(kWave) void k_ev_tile(const uint2 *rec, const uint32_t *base,
                                                    EvParams q)
{
    extern __shared__ uint32_t lds[];
    float *acc = (float *)lds; // [(k * 2 + c)][256 pixels]
    const int lane = threadIdx.x, tile = blockIdx.x;
    const int C = 2 * q.bins;
    for (int c = lane; c < C * kTilePx; c += kWave) acc[c] = 0.0f;
    __syncthreads();
    const uint32_t beg = base[tile], end = base[tile + 1];
    const uint64_t lt = lanemask_lt();
    const float binsf = (float)q.bins;
    for (uint32_t i0 = beg; i0 < end; i0 += kWave) {
        const uint32_t i = i0 + lane;
        const bool act = i < end;
        uint2 r = make_uint2(0u, 0u);
        if (act) r = rec[i];
        const uint32_t cell = r.x & (kTileCells - 1);
        const float ts = binsf * __uint_as_float(r.y); // t* = bins * float(t), generate_eventvolume.py:23
        const uint64_t m = match_any_fixed<kCellBits>(cell, act);
        // the lowest lane of every group applies the group's events in ascending lane order
        uint64_t mm = (act && (m & lt) == 0) ? m : 0ull;
        const int px = (int)(cell >> 1);
        const int ch = (cell & 1u) ? 0 : 1; // weights [p, 1 - p]: channel 0 = p == 1
        while (__ballot(mm != 0ull)) {
            const int j = mm ? (__ffsll((long long)mm) - 1) : lane;
            const float tj = __shfl(ts, j);
            if (mm) {
                mm &= mm - 1ull;
                const int k0 = (int)floorf(tj);
#pragma unroll
                for (int dk = 0; dk < 2; ++dk) {
                    const int k = k0 + dk;
                    if (k >= 1 && k <= q.bins) {
                        const float d = (float)k - tj;
                        const float w = 1.0f - fabsf(d); // generate_eventvolume.py:28
                        if (w > 0.0f) {
                            float *a = &acc[((k - 1) * 2 + ch) * kTilePx + px];
                            *a = *a + w;
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    const TileGeom g = tile_geom(tile, q.tiles_x, q.H, q.W);
    const long long plane = (long long)q.H * q.W;
    for (int o = lane; o < C * kTilePx; o += kWave) {
        const int lx = o & 31, ly = (o >> 5) & 7, c = o >> 8;
        if (lx < g.nx && ly < g.ny) {
            float v = acc[c * kTilePx + (ly << 5 | lx)] / 5.0f * 255.0f; // generate_eventvolume.py:37
            long long idx = c * plane + (long long)(g.y0 + ly) * q.W + g.x0 + lx;
            if (q.out_f32) q.out_f32[idx] = v;
            if (q.out_u8) q.out_u8[idx] = (uint8_t)(int)(v > 255.0f ? 255.0f : v);
        }
    }
}

// ---- SAE -------------------------------------------------------------------------------------
struct SaeParams {
    int H, W, tiles_x, n_lamda;
    float lam[FRLW_MAX_LAMDAS];
    float nowf;
    const float *mem_in;
    float *mem_out;
    float *out_f32;
    uint8_t *out_u8;
};

__global__ __launch_bounds__(kWave) void k_sae_tile(const uint2 *rec, const uint32_t *base,
                                                     SaeParams q)
{
    // last writer in stream order per cell = max over (position in the tile's record list, t bits)
    __shared__ unsigned long long last[kTileCells];
    const int lane = threadIdx.x, tile = blockIdx.x;
    for (int c = lane; c < kTileCells; c += kWave) last[c] = 0ull;
    __syncthreads();
    const uint32_t beg = base[tile], end = base[tile + 1];
    for (uint32_t i = beg + lane; i < end; i += kWave) {
        uint2 r = rec[i];
        unsigned long long key = ((unsigned long long)(i - beg + 1u) << 32) | r.y;
        atomicMax(&last[r.x & (kTileCells - 1)], key);
    }
    __syncthreads();
    const TileGeom g = tile_geom(tile, q.tiles_x, q.H, q.W);
    const long long plane = (long long)q.H * q.W;
    const float init = (0.0f + q.nowf) - 5000000.0f; // generate_surfaceofactiveevents.py:48
    for (int o = lane; o < kTileCells; o += kWave) {
        const int lx = o & 31, ly = (o >> 5) & 7, p = o >> 8;
        if (lx < g.nx && ly < g.ny) {
            unsigned long long key = last[((ly << 5 | lx) << 1) | p];
            float t = key ? __uint_as_float((uint32_t)key) : init;
            long long idx = p * plane + (long long)(g.y0 + ly) * q.W + g.x0 + lx;
            if (q.mem_in) {
                float m = q.mem_in[idx];
                if (!(t > m)) t = m; // torch.where(t_img > memory, t_img, memory) :52
            }
            q.mem_out[idx] = t;
            const float dt = t - q.nowf;
            for (int l = 0; l < q.n_lamda; ++l) {
                float v = expf(q.lam[l] * dt) * 255.0f;
                long long oi = (long long)l * 2 * plane + idx;
                if (q.out_f32) q.out_f32[oi] = v;
                if (q.out_u8) q.out_u8[oi] = (uint8_t)(int)v;
            }
        }
    }
}

// ---- TAF -------------------------------------------------------------------------------------
struct TafParams {
    int H, W, tiles_x, K, n_windows, flip;
    const WsHeader *hdr;
    float *state;    // (H, W, 2, K)
    float *view_f32; // (2K, H, W) or NULL
    uint8_t *out_u8; // (K, 2, H, W) or NULL
};

__device__ __forceinline__ float leaky_f(float v)
{
    float l = log1pf(-v);           // generate_taf.py:72
    l = 1.0f - l / 8.7f;            // :73
    if (l < 0.0f) l = 0.0f;         // :74
    return l * 255.0f;              // :75
}

__global__ __launch_bounds__(kWave) void k_taf_tile(const uint2 *rec, const uint32_t *base,
                                                     TafParams q)
{
    extern __shared__ uint32_t lds[];
    float *acc_sum = (float *)lds;                         // [512]
    uint32_t *acc_cnt = lds + kTileCells;                  // [512]
    float *st = (float *)(lds + 2 * kTileCells);           // [K][512]: slot-major, conflict-free FIFO
    const int lane = threadIdx.x, tile = blockIdx.x;
    const int K = q.K;
    const TileGeom g = tile_geom(tile, q.tiles_x, q.H, q.W);
    const uint32_t beg = base[tile], end = base[tile + 1];
    const uint64_t lt = lanemask_lt();

    for (int attempt = 0; attempt < 2; ++attempt) {
        // attempt 0 assumes the tile's records are window-sorted (true for a time-sorted stream,
        // the partition being stable); attempt 1 re-scans the whole list once per window.
        const bool sorted = attempt == 0;
        // ---- load the tile's FIFO state: global (y, x, p, k) -> LDS [k][cell]
        for (int ly = 0; ly < g.ny; ++ly) {
            const float *src = q.state + (((long long)(g.y0 + ly) * q.W + g.x0) * 2) * K;
            const int nfl = g.nx * 2 * K;
            for (int f = lane; f < nfl; f += kWave) {
                const int cellrow = f / K, k = f - cellrow * K; // cellrow = lx * 2 + p
                st[k * kTileCells + ((ly << 6) | cellrow)] = src[f];
            }
        }
        for (int c = lane; c < kTileCells; c += kWave) { acc_sum[c] = 0.0f; acc_cnt[c] = 0u; }
        __syncthreads();

        uint32_t pos = beg;
        bool violated = false;
        for (int w = 0; w < q.n_windows; ++w) {
            // ---- accumulate the records of window w, in stream order
            uint32_t p0 = sorted ? pos : beg;
            while (p0 < end) {
                const uint32_t i = p0 + lane;
                const bool valid = i < end;
                uint2 r = make_uint2(0u, 0u);
                if (valid) r = rec[i];
                const int rw = (int)(r.x >> kCellBits);
                const bool act = valid && rw == w;
                const uint64_t actm = __ballot(act);
                const uint64_t validm = __ballot(valid);
                uint32_t advance = kWave;
                bool more = true;
                if (sorted) {
                    // the records of window w must be a prefix of what is left
                    const int nact = __popcll(actm);
                    const uint64_t prefix = nact == 64 ? ~0ull : ((1ull << nact) - 1ull);
                    if (actm != prefix) { violated = true; break; }
                    advance = (uint32_t)nact;
                    more = actm == validm; // else the next record belongs to another window
                }
                if (actm) {
                    const uint32_t cell = r.x & (kTileCells - 1);
                    const float val = __uint_as_float(r.y);
                    const uint64_t m = match_any_fixed<kCellBits>(cell, act);
                    const bool leader = act && (m & lt) == 0;
                    uint64_t mm = leader ? m : 0ull;
                    float s = 0.0f;
                    uint32_t c = 0u;
                    if (leader) { s = acc_sum[cell]; c = acc_cnt[cell]; }
                    while (__ballot(mm != 0ull)) {
                        const int j = mm ? (__ffsll((long long)mm) - 1) : lane;
                        const float v = __shfl(val, j);
                        if (mm) { s = s + v; c += 1u; mm &= mm - 1ull; } // sum += t - 1, cnt += 1
                    }
                    if (leader) { acc_sum[cell] = s; acc_cnt[cell] = c; }
                }
                p0 += advance;
                if (sorted) pos = p0;
                if (!more) break;
            }
            if (violated) break;
            __syncthreads();
            // ---- FIFO update (generate_taf.py:35-49); a globally empty window changes nothing
            const bool window_has_events = q.hdr->wcount[w] != 0u;
            for (int c = lane; c < kTileCells; c += kWave) {
                const uint32_t n = acc_cnt[c];
                if (window_has_events) {
                    if (n == 0u) {
                        for (int k = 0; k < K; ++k) st[k * kTileCells + c] = st[k * kTileCells + c] - 1.0f;
                    } else {
                        const float mean = acc_sum[c] / ((float)n + 1e-8f); // :27
                        for (int k = 0; k + 1 < K; ++k) st[k * kTileCells + c] = st[(k + 1) * kTileCells + c] - 1.0f;
                        st[(K - 1) * kTileCells + c] = mean;
                    }
                }
                acc_sum[c] = 0.0f;
                acc_cnt[c] = 0u;
            }
            __syncthreads();
        }
        if (sorted && !violated && pos != end) violated = true; // a record of an earlier window was left behind
        if (!violated) break;
        __syncthreads();
    }

    // ---- write-out: state, optional f32 view (2K, H, W), optional uint8 leaky transform
    for (int ly = 0; ly < g.ny; ++ly) {
        float *dst = q.state + (((long long)(g.y0 + ly) * q.W + g.x0) * 2) * K;
        const int nfl = g.nx * 2 * K;
        for (int f = lane; f < nfl; f += kWave) {
            const int cellrow = f / K, k = f - cellrow * K;
            dst[f] = st[k * kTileCells + ((ly << 6) | cellrow)];
        }
    }
    const long long plane = (long long)q.H * q.W;
    if (q.view_f32 || q.out_u8) {
        for (int o = lane; o < K * kTileCells; o += kWave) { // o = (k, p, ly, lx)
            const int lx = o & 31, ly = (o >> 5) & 7, p = (o >> 8) & 1, k = o >> 9;
            if (lx < g.nx && ly < g.ny) {
                const float v = st[k * kTileCells + (((ly << 5 | lx) << 1) | p)];
                const long long pix = (long long)(g.y0 + ly) * q.W + g.x0 + lx;
                if (q.view_f32) q.view_f32[(long long)(2 * k + p) * plane + pix] = v; // channel 2k + p, :55
                if (q.out_u8) {
                    const int ko = q.flip ? (K - 1 - k) : k;
                    q.out_u8[(long long)(2 * ko + p) * plane + pix] = (uint8_t)(int)leaky_f(v);
                }
            }
        }
    }
}

// ---- stand-alone elementwise helpers ---------------------------------------------------------
__global__ void k_leaky(const float *in, long long n, float *out_f32, uint8_t *out_u8)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        float v = leaky_f(in[i]);
        if (out_f32) out_f32[i] = v;
        if (out_u8) out_u8[i] = (uint8_t)(int)v;
    }
}

__global__ void k_quantize(const float *in, long long n, int clip255, uint8_t *out)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        float v = in[i];
        if (clip255 && v > 255.0f) v = 255.0f;
        out[i] = (uint8_t)(int)v;
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

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct Plan {
    int tiles_x, tiles_y, n_tiles, tile_bits;
    long long chunk;
    int units;
    size_t off_counts, off_total, off_base, off_records, bytes;
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int env_int(const char *name, int dflt)
{
    const char *s = getenv(name);
    return s && *s ? atoi(s) : dflt;
}

bool make_plan(long long n, int H, int W, Plan &p)
{
    if (H <= 0 || W <= 0 || n < 0) return false;
    p.tiles_x = (W + kTileW - 1) / kTileW;
    p.tiles_y = (H + kTileH - 1) / kTileH;
    p.n_tiles = p.tiles_x * p.tiles_y;
    if (p.n_tiles > kMaxTiles) return false;
    p.tile_bits = 0;
    while ((1 << p.tile_bits) < p.n_tiles) ++p.tile_bits;
    // one wave per unit; enough units to put >= 2 waves on every SIMD of the 256 CUs without
    // letting counts[unit][tile] outgrow the event array itself
    const int max_units = env_int("FRLW_UNITS", 2048);
    long long chunk = (n + max_units - 1) / max_units;
    if (chunk < 1024) chunk = 1024;
    chunk = (chunk + kWave - 1) / kWave * kWave;
    p.chunk = chunk;
    p.units = (int)((n + chunk - 1) / chunk);
    if (p.units < 1) p.units = 1;
    size_t off = kHeaderBytes;
    p.off_counts = off; off = align_up(off + (size_t)p.units * p.n_tiles * 4, 256);
    p.off_total = off;  off = align_up(off + (size_t)p.n_tiles * 4, 256);
    p.off_base = off;   off = align_up(off + (size_t)(p.n_tiles + 1) * 4, 256);
    p.off_records = off; off = align_up(off + (size_t)(n > 0 ? n : 1) * 8, 256);
    p.bytes = off;
    return true;
}

// FRLW_DEBUG=1 prints the failing HIP call to stderr; the ABI itself only returns the code.
int hip_fail(hipError_t e, const char *what, int line)
{
    if (env_int("FRLW_DEBUG", 0))
        fprintf(stderr, "frlw_evd: %s failed at line %d: %s\n", what, line, hipGetErrorString(e));
    return FRLW_ERR_HIP;
}
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return hip_fail(e_, #expr, __LINE__); } while (0)

struct Partitioned {
    const uint2 *records;
    const uint32_t *base;
    WsHeader *hdr;
    Plan plan;
};

// Steps 1-3 shared by every encoder.
int partition_events(const frlw_events_t *ev, int H, int W, int kind, long long t0, long long win,
                     int n_windows, int time_filter, void *ws, size_t ws_bytes, hipStream_t s,
                     Partitioned &out)
{
    if (!ev || !ws || (ev->n > 0 && !ev->data)) return FRLW_ERR_ARG;
    if (ev->layout != FRLW_LAYOUT_XYTP_F64 && ev->layout != FRLW_LAYOUT_DAT8) return FRLW_ERR_ARG;
    if (ev->layout == FRLW_LAYOUT_XYTP_F64 && ev->row_stride < 4) return FRLW_ERR_ARG;
    if ((ev->xmap == nullptr) != (ev->ymap == nullptr)) return FRLW_ERR_ARG;
    if (ev->n >= (1ll << 32)) return FRLW_ERR_UNSUPPORTED;
    Plan p;
    if (!make_plan(ev->n, H, W, p)) return FRLW_ERR_UNSUPPORTED;
    if (ws_bytes < p.bytes) return FRLW_ERR_WORKSPACE;
    char *w8 = (char *)ws;
    WsHeader *hdr = (WsHeader *)w8;
    uint32_t *counts = (uint32_t *)(w8 + p.off_counts);
    uint32_t *total = (uint32_t *)(w8 + p.off_total);
    uint32_t *base = (uint32_t *)(w8 + p.off_base);
    uint2 *records = (uint2 *)(w8 + p.off_records);

    Decode d;
    d.data = ev->data; d.n = ev->n; d.layout = ev->layout; d.row_stride = ev->row_stride;
    d.xmap = ev->layout == FRLW_LAYOUT_DAT8 ? ev->xmap : nullptr;
    d.ymap = ev->layout == FRLW_LAYOUT_DAT8 ? ev->ymap : nullptr;
    d.map_w = ev->map_w; d.map_h = ev->map_h;
    d.H = H; d.W = W; d.tiles_x = p.tiles_x; d.n_tiles = p.n_tiles; d.tile_bits = p.tile_bits;
    d.kind = kind; d.t0 = t0; d.win = win; d.n_windows = n_windows; d.time_filter = time_filter;

    HIP_TRY(hipMemsetAsync(hdr, 0, kHeaderBytes, s));
    const size_t lds_hist = (size_t)(p.n_tiles + FRLW_MAX_WINDOWS) * 4;
    const size_t lds_cur = (size_t)p.n_tiles * 4;
    hipLaunchKernelGGL(k_hist, dim3(p.units), dim3(kWave), lds_hist, s, d, p.chunk, counts, hdr);
    hipLaunchKernelGGL(k_colscan, dim3((p.n_tiles + kWave - 1) / kWave), dim3(kWave * kSlabs), 0, s,
                       counts, p.units, p.n_tiles, total);
    hipLaunchKernelGGL(k_tilescan, dim3(1), dim3(1024), 0, s, total, p.n_tiles, base);
    hipLaunchKernelGGL(k_scatter, dim3(p.units), dim3(kWave), lds_cur, s, d, p.chunk, counts, base,
                       records);
    HIP_TRY(hipGetLastError());
    out.records = records; out.base = base; out.hdr = hdr; out.plan = p;
    return FRLW_OK;
}

int grid_for(long long n, int block)
{
    long long g = (n + block - 1) / block;
    if (g > 256 * 8) g = 256 * 8;
    if (g < 1) g = 1;
    return (int)g;
}

} // namespace

// =============================================================================================
// C-ABI
// =============================================================================================
extern "C" {

const char *frlw_version(void) { return "frlw_evd 0.1.0 gfx950"; }

size_t frlw_encoder_workspace_bytes(int64_t n_events, int H, int W)
{
    Plan p;
    if (!make_plan(n_events, H, W, p)) return 0;
    return p.bytes;
}

int frlw_encoder_status(const void *workspace, frlw_stream_t stream, int *status_out)
{
    if (!workspace || !status_out) return FRLW_ERR_ARG;
    int32_t st = 0;
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(&st, workspace, sizeof(st), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    *status_out = (st & ST_INDEX) ? FRLW_ERR_INDEX : (st & ST_POLARITY) ? FRLW_ERR_POLARITY : FRLW_OK;
    return FRLW_OK;
}

int frlw_eci_encode(const frlw_events_t *ev, int H, int W, float *out_f32, uint8_t *out_u8,
                    void *workspace, size_t workspace_bytes, frlw_stream_t stream)
{
    if (!out_f32 && !out_u8) return FRLW_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    Partitioned pt;
    int rc = partition_events(ev, H, W, KIND_ECI, 0, 1, 1, 0, workspace, workspace_bytes, s, pt);
    if (rc != FRLW_OK) return rc;
    EciParams q;
    q.H = H; q.W = W; q.tiles_x = pt.plan.tiles_x; q.out_f32 = out_f32; q.out_u8 = out_u8;
    // generate_eventcountimage.py:32-34,41: n sequential f32 adds of 0.05f, > 1 -> 1, * 255
    volatile float acc = 0.0f;
    q.lut[0] = 0.0f;
    for (int n = 1; n <= 20; ++n) {
        acc = acc + 0.05f;
        float v = acc;
        q.lut[n] = (v > 1.0f ? 1.0f : v) * 255.0f;
    }
    hipLaunchKernelGGL(k_eci_tile, dim3(pt.plan.n_tiles), dim3(kWave), 0, s, pt.records, pt.base, q);
    HIP_TRY(hipGetLastError());
    return FRLW_OK;
}

int frlw_ev_encode(const frlw_events_t *ev, int H, int W, int bins, int64_t t_end,
                   int64_t window_us, float *out_f32, uint8_t *out_u8, void *workspace,
                   size_t workspace_bytes, frlw_stream_t stream)
{
    if ((!out_f32 && !out_u8) || bins < 1 || bins > FRLW_MAX_BINS) return FRLW_ERR_ARG;
    if (ev && ev->layout == FRLW_LAYOUT_DAT8 && window_us <= 0) return FRLW_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    Partitioned pt;
    int rc = partition_events(ev, H, W, KIND_EV, t_end - window_us, window_us, 1, 1, workspace,
                              workspace_bytes, s, pt);
    if (rc != FRLW_OK) return rc;
    EvParams q;
    q.H = H; q.W = W; q.tiles_x = pt.plan.tiles_x; q.bins = bins; q.out_f32 = out_f32; q.out_u8 = out_u8;
    hipLaunchKernelGGL(k_ev_tile, dim3(pt.plan.n_tiles), dim3(kWave), (size_t)2 * bins * kTilePx * 4, s,
                       pt.records, pt.base, q);
    HIP_TRY(hipGetLastError());
    return FRLW_OK;
}

int frlw_sae_encode(const frlw_events_t *ev, int H, int W, const double *lamdas, int n_lamda,
                    const float *mem_in, float *mem_out, int64_t now, int64_t window_us,
                    float *out_f32, uint8_t *out_u8, void *workspace, size_t workspace_bytes,
                    frlw_stream_t stream)
{
    if (!mem_out || !lamdas || n_lamda < 0 || n_lamda > FRLW_MAX_LAMDAS) return FRLW_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    Partitioned pt;
    const int filt = window_us > 0;
    int rc = partition_events(ev, H, W, KIND_SAE, now - window_us, 1, 1, filt, workspace,
                              workspace_bytes, s, pt);
    if (rc != FRLW_OK) return rc;
    SaeParams q;
    q.H = H; q.W = W; q.tiles_x = pt.plan.tiles_x; q.n_lamda = n_lamda;
    for (int l = 0; l < n_lamda; ++l) q.lam[l] = (float)lamdas[l];
    q.nowf = (float)now;
    q.mem_in = mem_in; q.mem_out = mem_out; q.out_f32 = out_f32; q.out_u8 = out_u8;
    hipLaunchKernelGGL(k_sae_tile, dim3(pt.plan.n_tiles), dim3(kWave), 0, s, pt.records, pt.base, q);
    HIP_TRY(hipGetLastError());
    return FRLW_OK;
}

int frlw_taf_encode(const frlw_events_t *ev, int H, int W, int K, int64_t t_start,
                    int64_t window_us, int n_windows, float *state, float *view_f32,
                    uint8_t *out_u8, int flags, void *workspace, size_t workspace_bytes,
                    frlw_stream_t stream)
{
    if (!state || K < 1 || K > FRLW_MAX_BINS || n_windows < 1 || n_windows > FRLW_MAX_WINDOWS)
        return FRLW_ERR_ARG;
    if (ev && ev->layout == FRLW_LAYOUT_XYTP_F64 && n_windows != 1) return FRLW_ERR_ARG;
    if (ev && ev->layout == FRLW_LAYOUT_DAT8 && window_us <= 0) return FRLW_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    Partitioned pt;
    int rc = partition_events(ev, H, W, KIND_TAF, t_start, window_us, n_windows, 0, workspace,
                              workspace_bytes, s, pt);
    if (rc != FRLW_OK) return rc;
    TafParams q;
    q.H = H; q.W = W; q.tiles_x = pt.plan.tiles_x; q.K = K; q.n_windows = n_windows;
    q.flip = (flags & FRLW_TAF_U8_FLIP_K) ? 1 : 0;
    q.hdr = pt.hdr; q.state = state; q.view_f32 = view_f32; q.out_u8 = out_u8;
    const size_t lds = (size_t)(2 * kTileCells + K * kTileCells) * 4;
    hipLaunchKernelGGL(k_taf_tile, dim3(pt.plan.n_tiles), dim3(kWave), lds, s, pt.records, pt.base, q);
    HIP_TRY(hipGetLastError());
    return FRLW_OK;
}

int frlw_leaky_transform(const float *in, int64_t n, float *out_f32, uint8_t *out_u8,
                         frlw_stream_t stream)
{
    if (!in || (!out_f32 && !out_u8) || n < 0) return FRLW_ERR_ARG;
    if (n == 0) return FRLW_OK;
    hipLaunchKernelGGL(k_leaky, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, in,
                       (long long)n, out_f32, out_u8);
    HIP_TRY(hipGetLastError());
    return FRLW_OK;
}

int frlw_quantize_u8(const float *in, int64_t n, int clip255, uint8_t *out, frlw_stream_t stream)
{
    if (!in || !out || n < 0) return FRLW_ERR_ARG;
    if (n == 0) return FRLW_OK;
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
    hipLaunchKernelGGL(k_resize_nearest<uint8_t>, dim3(grid_for(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, in, C, H, W, Ho, Wo, (float)H / (float)Ho,
                       (float)W / (float)Wo, out);
    HIP_TRY(hipGetLastError());
    return FRLW_OK;
}

} // extern "C"
