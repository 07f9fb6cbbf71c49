// encoders.hip -- tile kernels (step 4) and the C-ABI of the event-stream -> tensor encoders.
//
// Replaces the bodies of the reference's four encoder functions (file:line in the reference):
//   generate_eventframe               generate_eventcountimage.py:19-41      (ECI)
//   generate_agile_event_volume_cuda  generate_eventvolume.py:15-42          (Event Volume)
//   generate_leaky_cuda / taf_cuda    generate_surfaceofactiveevents.py:44-80 (SAE)
//   generate_taf_cuda / taf_cuda      generate_taf.py:19-67, leaky_transform :69-76 (TAF)
// and, for raw DAT streams, the harness glue around them (generate_taf.py:197-227).
//
// partition.hip has already grouped the events by tile ((1 << twl) x 8 pixels), keeping stream order
// inside a tile (records {window << (twl + 4) | cell, f32 value}).  One workgroup of NT = 4 << twl
// threads owns one tile; thread t owns the four cells t, t + NT, t + 2NT, t + 3NT and keeps their
// accumulators (and, for TAF, their K-deep FIFO) in registers:
//
//   ECI  integer counts by LDS atomics (order-free), 21-entry LUT epilogue.
//   SAE  last writer per cell = LDS atomicMax over (position, t bits) (order-free), exp epilogue.
//   EV / TAF  exact f32 sums in stream order: a slice of the tile's records is counting-sorted by cell
//        into LDS (LDS atomics only pick slots; each cell's short segment is then ordered by stream
//        position), and every owner thread adds its cells' segments sequentially -- no conflict
//        handling, no float atomics.  TAF does this per 10 ms window with FIFO ageing in registers;
//        leaky transform, uint8 truncation and the (2K, H, W) permute are fused into the write-out.
//        The bodies are templated on the cells per thread: 4 for a whole tile; 1 for a "quarter" -- four
//        workgroups per tile -- used for tiles the partition lists as hot (skew) and for every tile of a
//        frame too small to give each SIMD a wavefront.
//
// Everything is exact-f32 arithmetic in the reference's operation order (-ffp-contract=off).

#include "frlw_common.h"

using namespace frlw;

namespace {

#ifndef FRLW_SLICE_MULT
#define FRLW_SLICE_MULT kSliceMult
#endif
constexpr int CPT = kCellsPerThread;
constexpr int kMaxK = 8;

template <int C> struct Owner { // the C cells of thread t in a tile of NT threads
    int row[C];  // tile row of cell j
    int lx, p;   // pixel column inside the tile, polarity
    bool ok[C];
    long long pix[C]; // y * W + x
};

// Cell c = t + NT * j = (row << (twl + 1)) | (lx << 1) | p with a row of NT / 2 cells.  A whole-tile workgroup
// owns j = 0..3 (C = 4, j0 = 0); a quarter workgroup of a hot tile owns the single j = j0 (C = 1).
template <int NT, int C>
__device__ __forceinline__ Owner<C> make_owner(const TileGeom &g, int W, int j0)
{
    constexpr int RL = NT / 2;
    Owner<C> o;
    const int t = threadIdx.x;
    const int within = t & (RL - 1);
    o.lx = within >> 1;
    o.p = within & 1;
#pragma unroll
    for (int j = 0; j < C; ++j) {
        o.row[j] = 2 * (j0 + j) + (t >= RL ? 1 : 0);
        o.ok[j] = o.lx < g.nx && o.row[j] < g.ny;
        o.pix[j] = (long long)(g.y0 + o.row[j]) * W + g.x0 + o.lx;
    }
    return o;
}

// ---- skew: hot tiles ---------------------------------------------------------------------------------
// k_tilescan lists up to kMaxHot tiles that hold more than hot_thr records.  Their per-tile workgroups do
// nothing; 4 * kMaxHot workgroups in front of them in the grid give each listed tile to four workgroups of NT threads with ONE cell
// per thread (quarter q = the cells t + NT * q), i.e. four times the lanes on the per-cell sequential sums.
// Each quarter streams the tile's whole record list and keeps only its own cells.
__device__ __forceinline__ bool tile_is_listed(const WsHeader *hdr, const uint32_t *base, int tile)
{
    if (base[tile + 1] - base[tile] <= hdr->hot_thr) return false;
    const uint32_t n_hot = hdr->n_hot < (uint32_t)kMaxHot ? hdr->n_hot : (uint32_t)kMaxHot;
    for (uint32_t h = 0; h < n_hot; ++h) // more than kMaxHot hot tiles: the unlisted ones stay whole
        if ((int)hdr->hot[h] == tile) return true;
    return false;
}

__device__ __forceinline__ bool pick_quarter(const WsHeader *hdr, int idx, int &tile, int &quarter)
{
    const int h = idx >> 2;
    const uint32_t n_hot = hdr->n_hot < (uint32_t)kMaxHot ? hdr->n_hot : (uint32_t)kMaxHot;
    if ((uint32_t)h >= n_hot) return false;
    tile = (int)hdr->hot[h];
    quarter = idx & 3;
    return true;
}

#ifdef FRLW_TILE_PROF
// Developer build only (-DFRLW_TILE_PROF): per-phase cycle sums of the TAF tile kernel, thread 0 of each tile.
__device__ unsigned long long g_prof[16];
#define PROF_MARK(ph) do { const unsigned long long now_ = __builtin_readcyclecounter(); \
        if (threadIdx.x == 0) atomicAdd(&g_prof[ph], now_ - prof_t_); prof_t_ = now_; } while (0)
#define PROF_BEGIN unsigned long long prof_t_ = __builtin_readcyclecounter()
#define PROF_RESET prof_t_ = __builtin_readcyclecounter()
#else
#define PROF_MARK(ph) do { } while (0)
#define PROF_BEGIN do { } while (0)
#define PROF_RESET do { } while (0)
#endif

// ---- slice counting sort shared by EV and TAF --------------------------------------------------
// Sorts the records i in [s0, s0 + span) that `sel` maps to a local cell (sel(meta) >= 0; local cell =
// t + NT * j, j < C) by cell into slot[]; returns through (c[j], o[j]) the length and offset of the segments of
// this thread's C cells.  `cnt` has NT * C entries.
#ifndef FRLW_KBATCH
#define FRLW_KBATCH 8
#endif
constexpr int kBatch = FRLW_KBATCH; // global loads in flight per thread in the two passes of slice_sort
template <int NT, int C, typename Sel>
__device__ __forceinline__ void slice_sort(const uint2 *rec, uint32_t s0, uint32_t span, Sel sel, uint32_t *cnt,
                                           uint2 *slot, uint32_t *red, uint32_t (&c)[C], uint32_t (&o)[C], int cb = 0)
{
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    constexpr int NW = NT / kWave;
    PROF_BEGIN;
#pragma unroll
    for (int j = 0; j < C; ++j) cnt[t + NT * j] = 0;
    __syncthreads();
    // batches of kBatch loads in flight per thread
    for (uint32_t i0 = s0 + t; i0 < s0 + span; i0 += kBatch * NT) {
        uint32_t m[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const uint32_t i = i0 + u * NT;
            m[u] = i < s0 + span ? rec[i].x : 0u;
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int cell = sel(m[u]);
            if (i0 + u * NT < s0 + span && cell >= 0) atomicAdd(&cnt[cell], 1u);
        }
    }
    __syncthreads();
    PROF_MARK(1);
    // exclusive scan in thread-major order: a thread's segments are adjacent
    uint32_t tot = 0;
#pragma unroll
    for (int j = 0; j < C; ++j) { c[j] = cnt[t + NT * j]; tot += c[j]; }
    const uint32_t inc = wave_incl_scan(tot);
    if (lane == kWave - 1) red[16 + wv] = inc;
    __syncthreads();
    uint32_t run = inc - tot;
    for (int k = 0; k < wv && k < NW; ++k) run += red[16 + k];
#pragma unroll
    for (int j = 0; j < C; ++j) { o[j] = run; cnt[t + NT * j] = run; run += c[j]; }
    __syncthreads();
    PROF_MARK(2);
    // Rounds of NT records with a barrier in between: slots of a later round always come after those
    // of an earlier one, so a cell's segment is out of order only among the few records of one round.
    for (uint32_t g0 = s0; g0 < s0 + span; g0 += kBatch * NT) {
        uint2 r[kBatch]; // the loads of kBatch rounds are issued together
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const uint32_t i = g0 + u * NT + t;
            r[u] = i < s0 + span ? rec[i] : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            if (g0 + u * NT >= s0 + span) break; // block-uniform
            const uint32_t i = g0 + u * NT + t;
            const int cell = sel(r[u].x);
            if (i < s0 + span && cell >= 0) {
                const uint32_t sl = atomicAdd(&cnt[cell], 1u);
                // one 8-byte LDS entry per record: {value bits, position in the slice << 8 | window}
                slot[sl] = make_uint2(r[u].y, ((i - s0) << 8) | ((r[u].x >> cb) & 255u));
            }
            __syncthreads();
        }
    }
    PROF_MARK(3);
}

// Slots were handed out by LDS atomics in arrival order; restore stream order inside the C segments
// of this thread.  Only records of the same round (NT consecutive records) can be swapped, so a segment
// is a run of tiny permuted blocks -- nearly always pairs.  One bubble pass, the cells in lock-step
// (independent LDS reads in flight per step, selects instead of branches), repairs every adjacent inversion;
// a segment it cannot repair (an element that has to move two or more places: three or more records of
// one pixel within NT events) is finished by an insertion sort afterwards.
template <int C>
__device__ __forceinline__ void segments_order(const uint32_t (&c)[C], const uint32_t (&o)[C], uint2 *slot,
                                               uint32_t last_slot)
{
    uint32_t maxc = 0, below[C];
    uint2 top[C]; // largest element so far = what slot[o + x - 1] holds
    bool deep[C];
#pragma unroll
    for (int j = 0; j < C; ++j) {
        maxc = c[j] > maxc ? c[j] : maxc;
        top[j] = make_uint2(0u, 0u);
        below[j] = 0u;
        deep[j] = false;
    }
    for (uint32_t x = 0; x < maxc; ++x) {
        uint2 e[C];
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const uint32_t at = o[j] + x;
            e[j] = slot[at < last_slot ? at : last_slot];
        }
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const bool live = x < c[j];
            const bool inv = live && e[j].y < top[j].y; // .y orders by position in the slice
            const bool fwd = live && !inv;
            deep[j] |= inv && e[j].y < below[j];
            if (inv) { // adjacent inversion: e goes under top
                slot[o[j] + x - 1] = e[j];
                slot[o[j] + x] = top[j];
            }
            below[j] = inv ? e[j].y : (fwd ? top[j].y : below[j]);
            top[j].x = fwd ? e[j].x : top[j].x;
            top[j].y = fwd ? e[j].y : top[j].y;
        }
    }
#pragma unroll
    for (int j = 0; j < C; ++j) {
        if (!deep[j]) continue;
        for (uint32_t x = 1; x < c[j]; ++x) {
            const uint2 e = slot[o[j] + x];
            uint32_t b = x;
            while (b > 0 && slot[o[j] + b - 1].y > e.y) { slot[o[j] + b] = slot[o[j] + b - 1]; --b; }
            slot[o[j] + b] = e;
        }
    }
}

// local cell of a record for a workgroup that owns the cells j0 .. j0 + C - 1 of every thread (-1: not ours)
template <int NT, int C>
__device__ __forceinline__ int local_cell(uint32_t meta, int j0)
{
    const int cell = (int)(meta & (uint32_t)(NT * kCellsPerThread - 1));
    if (C == kCellsPerThread) return cell;
    return (cell / NT) == j0 ? (cell & (NT - 1)) : -1;
}

template <int NT> struct TileLds {
    static constexpr int NC = NT * CPT;
    static constexpr int SLICE = FRLW_SLICE_MULT * NT; // records counting-sorted per pass
};

// ---- ECI -------------------------------------------------------------------------------------
struct EciParams {
    int H, W, twl, tiles_x;
    float lut[21]; // value * 255 after n sequential +0.05f adds, clamped (n >= 20 -> 255)
    float *out_f32;
    uint8_t *out_u8;
};

template <int NT>
__global__ __launch_bounds__(NT) void k_eci_tile(const uint2 *rec, const uint32_t *base, EciParams q)
{
    constexpr int NC = NT * CPT;
    __shared__ uint32_t cnt[NC];
    const int t = threadIdx.x, tile = blockIdx.x;
#pragma unroll
    for (int j = 0; j < CPT; ++j) cnt[t + NT * j] = 0;
    __syncthreads();
    const uint32_t beg = base[tile], end = base[tile + 1];
    for (uint32_t i = beg + t; i < end; i += NT) atomicAdd(&cnt[rec[i].x & (NC - 1)], 1u);
    __syncthreads();
    const TileGeom g = tile_geom(tile, q.tiles_x, q.twl, q.H, q.W);
    const Owner<CPT> ow = make_owner<NT, CPT>(g, q.W, 0);
    const long long plane = (long long)q.H * q.W;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        if (ow.ok[j]) {
            const uint32_t n = cnt[t + NT * j];
            const float v = q.lut[n > 20u ? 20u : n];
            const long long idx = ow.p * plane + ow.pix[j];
            if (q.out_f32) q.out_f32[idx] = v;
            if (q.out_u8) q.out_u8[idx] = f32_to_u8(v);
        }
    }
}

// Small calls (BASELINE.json configs[0]: 100 000 events on 304x240): the five launches of the general path -- histogram, two
// scans, scatter, tile kernel -- are 29 us of launch chain for 1.4 MB of data.  A count image needs no order at all: ONE launch in
// which every workgroup owns 2048 consecutive pixels (4096 counters in LDS), reads the WHOLE event array (0.8 MB out of the L2,
// 36 workgroups at 304x240) and counts the events that fall into its stretch; the 21-entry table turns the counts into the
// image.  The first workgroup also owns the call's status.  Used while events x workgroups stays small (frlw_eci_encode).
constexpr int kEciScanPix = 2048;
constexpr int kEciScanThreads = 1024;
template <bool HAS_MAP>
__global__ __launch_bounds__(kEciScanThreads) void k_eci_scan(const uint2 *data, long long n, const uint16_t *xmap, const uint16_t *ymap,
                                                              int map_w, int map_h, EciParams q, WsHeader *hdr)
{
    __shared__ uint32_t cnt[2 * kEciScanPix];
    __shared__ int serr;
    const int t = threadIdx.x;
    for (int i = t; i < 2 * kEciScanPix; i += kEciScanThreads) cnt[i] = 0u;
    if (t == 0) serr = 0;
    __syncthreads();
    const long long total = (long long)q.H * q.W; // (< 2^31: the host checks)
    const long long p0 = (long long)blockIdx.x * kEciScanPix;
    const bool check_status = blockIdx.x == 0;
    int err = 0;
    constexpr int RU = 16; // clamped loads in flight per thread, masked below (a workgroup's pass over the array is a chain of
                           // dependent iterations: with four loads per iteration 100 000 events took 25 round trips, 21 us)
    for (long long i0 = 0; i0 < n; i0 += RU * kEciScanThreads) {
        uint2 r[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const long long i = i0 + (long long)u * kEciScanThreads + t;
            r[u] = data[i < n ? i : n - 1];
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const bool live = i0 + (long long)u * kEciScanThreads + t < n;
            int x = (int)(r[u].y & 16383u), y = (int)((r[u].y >> 14) & 16383u);
            const uint32_t p = (r[u].y >> 28) & 1u;
            bool ok = live;
            if (HAS_MAP) {
                ok = ok && x < map_w && y < map_h;
                x = xmap[x < map_w ? x : 0];
                y = ymap[y < map_h ? y : 0];
            }
            // the reference indexes the flat cell 2 x + 2 W y + p (generate_eventcountimage.py:32): x >= W aliases into the next
            // row, only a flat pixel outside the frame raises.  (The range test below drops such an event in every workgroup; the
            // STATUS is the first workgroup's business alone: 36 workgroups repeat this loop, every instruction counts.)
            const uint32_t flat = (uint32_t)x + (uint32_t)q.W * (uint32_t)y, loc = flat - (uint32_t)p0;
            if (check_status && live && (!ok || flat >= (uint32_t)total)) err |= ST_INDEX;
            if (ok && loc < (uint32_t)kEciScanPix && flat < (uint32_t)total) atomicAdd(&cnt[(loc << 1) | p], 1u);
        }
    }
    if (err) atomicOr(&serr, err);
    __syncthreads();
    if (blockIdx.x == 0 && t == 0) { // every workgroup has seen every event: the first one speaks for the call
        hdr->status = serr;
        fold_sticky_status(hdr, serr);
    }
    for (int i = t; i < 2 * kEciScanPix; i += kEciScanThreads) {
        const int pol = i >= kEciScanPix ? 1 : 0;
        const long long pix = p0 + (i - pol * kEciScanPix);
        if (pix >= total) continue;
        const uint32_t c = cnt[((uint32_t)(i - pol * kEciScanPix) << 1) | (uint32_t)pol];
        const float v = q.lut[c > 20u ? 20u : c];
        const long long idx = (long long)pol * total + pix; // view (H, W, 2) -> permute (2, H, W), :36
        if (q.out_f32) q.out_f32[idx] = v;
        if (q.out_u8) q.out_u8[idx] = f32_to_u8(v);
    }
}

// ---- SAE -------------------------------------------------------------------------------------
struct SaeParams {
    int H, W, twl, tiles_x, n_lamda;
    float lam[FRLW_MAX_LAMDAS];
    float nowf;
    const float *mem_in;
    float *mem_out;
    float *out_f32;
    uint8_t *out_u8;
};

template <int NT>
__global__ __launch_bounds__(NT) void k_sae_tile(const uint2 *rec, const uint32_t *base, SaeParams q)
{
    // last writer in stream order per cell = max over (position in the tile's record list, t bits)
    constexpr int NC = NT * CPT;
    __shared__ unsigned long long last[NC];
    const int t = threadIdx.x, tile = blockIdx.x;
#pragma unroll
    for (int j = 0; j < CPT; ++j) last[t + NT * j] = 0ull;
    __syncthreads();
    const uint32_t beg = base[tile], end = base[tile + 1];
    for (uint32_t i = beg + t; i < end; i += NT) {
        const uint2 r = rec[i];
        const unsigned long long key = ((unsigned long long)(i - beg + 1u) << 32) | r.y;
        atomicMax(&last[r.x & (NC - 1)], key);
    }
    __syncthreads();
    const TileGeom g = tile_geom(tile, q.tiles_x, q.twl, q.H, q.W);
    const Owner<CPT> ow = make_owner<NT, CPT>(g, q.W, 0);
    const long long plane = (long long)q.H * q.W;
    const float init = (0.0f + q.nowf) - 5000000.0f; // generate_surfaceofactiveevents.py:48
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        if (ow.ok[j]) {
            const unsigned long long key = last[t + NT * j];
            float tv = key ? __uint_as_float((uint32_t)key) : init;
            const long long idx = ow.p * plane + ow.pix[j];
            if (q.mem_in) {
                const float m = q.mem_in[idx];
                if (!(tv > m)) tv = m; // torch.where(t_img > memory, t_img, memory) :52
            }
            q.mem_out[idx] = tv;
            const float dt = tv - q.nowf;
            for (int l = 0; l < q.n_lamda; ++l) {
                const float v = expf(q.lam[l] * dt) * 255.0f;
                const long long oi = (long long)l * 2 * plane + idx;
                if (q.out_f32) q.out_f32[oi] = v;
                if (q.out_u8) q.out_u8[oi] = f32_to_u8(v);
            }
        }
    }
}

// ---- Event Volume ----------------------------------------------------------------------------
struct EvParams {
    int H, W, twl, tiles_x, bins;
    const WsHeader *hdr;
    float *out_f32;
    uint8_t *out_u8;
};

template <int NT, int C, int BINS>
__device__ __forceinline__ void ev_tile_body(const uint2 *rec, const uint32_t *base, const EvParams &q, int tile, int j0,
                                             uint32_t *cnt, uint2 *slot, uint32_t *red)
{
    constexpr int SLICE = TileLds<NT>::SLICE;
    const uint32_t beg = base[tile], end = base[tile + 1];
    float acc[C][BINS]; // cell (pixel, polarity) x time bin (BINS = 5 for the recipe's five bins, else kMaxK)
#pragma unroll
    for (int j = 0; j < C; ++j)
#pragma unroll
        for (int k = 0; k < BINS; ++k) acc[j][k] = 0.0f;
    const float binsf = (float)q.bins;
    for (uint32_t s0 = beg; s0 < end; s0 += SLICE) {
        const uint32_t span = end - s0 < (uint32_t)SLICE ? end - s0 : (uint32_t)SLICE;
        uint32_t c[C], o[C];
        slice_sort<NT, C>(rec, s0, span, [=](uint32_t m) { return local_cell<NT, C>(m, j0); }, cnt, slot, red, c, o);
        segments_order<C>(c, o, slot, SLICE - 1);
        uint32_t maxc = 0;
#pragma unroll
        for (int j = 0; j < C; ++j) maxc = c[j] > maxc ? c[j] : maxc;
        for (uint32_t a = 0; a < maxc; ++a) {
            uint32_t tv[C];
#pragma unroll
            for (int j = 0; j < C; ++j) {
                const uint32_t at = o[j] + a;
                tv[j] = slot[at < SLICE - 1 ? at : SLICE - 1].x;
            }
#pragma unroll
            for (int j = 0; j < C; ++j) {
                const bool live = a < c[j];
                const float ts = binsf * __uint_as_float(tv[j]); // t* = bins * float(t), generate_eventvolume.py:23
#pragma unroll
                for (int k = 0; k < BINS; ++k) { // bins beyond q.bins are accumulated but never stored
                    const float d = (float)(k + 1) - ts;
                    const float w = 1.0f - fabsf(d); // :28; negative weights -> 0 (:29) = skipped
                    const float na = acc[j][k] + w;
                    acc[j][k] = (live && w > 0.0f) ? na : acc[j][k];
                }
            }
        }
        __syncthreads();
    }
    const TileGeom g = tile_geom(tile, q.tiles_x, q.twl, q.H, q.W);
    const Owner<C> ow = make_owner<NT, C>(g, q.W, j0);
    const long long plane = (long long)q.H * q.W;
    const int ch = ow.p ? 0 : 1; // weights [p, 1 - p]: channel 0 = p == 1
#pragma unroll
    for (int j = 0; j < C; ++j) {
        if (ow.ok[j]) {
#pragma unroll
            for (int k = 0; k < BINS; ++k) {
                if (k < q.bins) {
                    const float v = acc[j][k] / 5.0f * 255.0f; // generate_eventvolume.py:37
                    const long long idx = (long long)(2 * k + ch) * plane + ow.pix[j];
                    if (q.out_f32) q.out_f32[idx] = v;
                    if (q.out_u8) q.out_u8[idx] = f32_to_u8(v > 255.0f ? 255.0f : v);
                }
            }
        }
    }
}

// grid = 4 * kMaxHot + n_tiles: the four quarters of every listed hot tile, then one workgroup per tile
template <int NT, int BINS>
__device__ __forceinline__ void ev_tile_kernel(const uint2 *rec, const uint32_t *base, const EvParams &q, int n_tiles)
{
    __shared__ uint32_t cnt[NT * CPT];
    __shared__ uint2 slot[TileLds<NT>::SLICE];
    __shared__ uint32_t red[32];
    if (n_tiles < 0) { // small frame (-n_tiles tiles): every tile as four quarters, grid = 4 * tiles
        ev_tile_body<NT, 1, BINS>(rec, base, q, (int)blockIdx.x >> 2, (int)blockIdx.x & 3, cnt, slot, red);
        return;
    }
    if ((int)blockIdx.x >= 4 * kMaxHot) {
        const int tile = (int)blockIdx.x - 4 * kMaxHot;
        if (tile_is_listed(q.hdr, base, tile)) return; // hot: left to its quarters
        ev_tile_body<NT, CPT, BINS>(rec, base, q, tile, 0, cnt, slot, red);
    } else { // the long-running workgroups come first in the grid
        int tile, quarter;
        if (!pick_quarter(q.hdr, (int)blockIdx.x, tile, quarter)) return;
        ev_tile_body<NT, 1, BINS>(rec, base, q, tile, quarter, cnt, slot, red);
    }
}
// (two instantiations: the recipe's five bins keep five accumulators per cell instead of eight -- the loop over the bins is the
// kernel's VALU work)
template <int NT>
__global__ __launch_bounds__(NT) void k_ev_tile(const uint2 *rec, const uint32_t *base, EvParams q, int n_tiles)
{
    ev_tile_kernel<NT, kMaxK>(rec, base, q, n_tiles);
}
template <int NT>
__global__ __launch_bounds__(NT) void k_ev_tile5(const uint2 *rec, const uint32_t *base, EvParams q, int n_tiles)
{
    ev_tile_kernel<NT, 5>(rec, base, q, n_tiles);
}

// ---- TAF -------------------------------------------------------------------------------------
struct TafParams {
    int H, W, twl, tiles_x, K, n_windows, flip;
    WsHeader *hdr;
    const uint32_t *leaky_thr; // level thresholds of uint8(leaky_transform(.)): the per-device table of partition.hip
    float *state;    // (H, W, 2, K)
    float *view_f32; // (2K, H, W) or NULL
    uint8_t *out_u8; // (K, 2, H, W) or NULL
};

// One FIFO step of one cell, generate_taf.py:27,35-49 (K <= 8, unused slots are never stored).
__device__ __forceinline__ void taf_fifo(float (&st)[kMaxK], int K, uint32_t n, float sum)
{
    // empty cell: every slot - 1; otherwise shift down (slot k+1 - 1) and the mean enters at K-1.  One
    // select-only body for both cases (the cells of a wave take both).
    const bool hit = n != 0u;
    const float mean = sum / ((float)n + 1e-8f);
#pragma unroll
    for (int k = 0; k < kMaxK; ++k) {
        const float nxt = k + 1 < kMaxK ? st[k + 1] : 0.0f;
        const float v = (hit ? nxt : st[k]) - 1.0f;
        st[k] = (hit && k == K - 1) ? mean : v;
    }
}

template <int NT, int C>
__device__ __forceinline__ void taf_tile_body(const uint2 *rec, const uint32_t *base, const TafParams &q, int tile,
                                              int j0, uint32_t *cnt, uint2 *slot, uint32_t *red, uint32_t *thr)
{
    constexpr int SLICE = TileLds<NT>::SLICE;
    const int t = threadIdx.x;
    const int K = q.K;
    for (int i = t; i < kLeakyLevels; i += NT) thr[i] = q.leaky_thr[i];
    __syncthreads(); // a tile without records reaches the write-out (which reads thr[]) without any other barrier
    const int cb = q.twl + 4; // cell bits
    const TileGeom g = tile_geom(tile, q.tiles_x, q.twl, q.H, q.W);
    const Owner<C> ow = make_owner<NT, C>(g, q.W, j0);
    const uint32_t beg = base[tile], end = base[tile + 1];
    const unsigned long long wmask = q.hdr->wmask; // windows that hold events (anywhere in the frame)
    float st[C][kMaxK], sum[C];
    uint32_t num[C];
    const auto sel = [=](uint32_t m) { return local_cell<NT, C>(m, j0); };

    auto load_state = [&]() {
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const float *src = q.state + (ow.pix[j] * 2 + ow.p) * K;
#pragma unroll
            for (int k = 0; k < kMaxK; ++k) st[j][k] = 0.0f;
            if (ow.ok[j]) {
                if (K == 8) {
                    const float4 a = ((const float4 *)src)[0], b = ((const float4 *)src)[1];
                    st[j][0] = a.x; st[j][1] = a.y; st[j][2] = a.z; st[j][3] = a.w;
                    st[j][4] = b.x; st[j][5] = b.y; st[j][6] = b.z; st[j][7] = b.w;
                } else {
#pragma unroll
                    for (int k = 0; k < kMaxK; ++k) if (k < K) st[j][k] = src[k];
                }
            }
            sum[j] = 0.0f;
            num[j] = 0u;
        }
    };
    // closes window w for the thread's cells: FIFO step (skipped when the window is empty in the whole
    // frame, generate_taf.py:40-41), accumulators back to zero
    auto close_window = [&](int w) {
        const bool has = (wmask >> w) & 1ull;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            if (has) taf_fifo(st[j], K, num[j], sum[j]);
            sum[j] = 0.0f;
            num[j] = 0u;
        }
    };

    // ---- fast path: the tile's records are window-sorted (always true for a time-sorted stream, the
    // partition being stable).  Slices of SLICE records, each counting-sorted by cell once, may span
    // several windows; the FIFO steps between them are block-uniform.
    PROF_BEGIN;
    load_state();
    bool bad = false;
    int cur_w = 0; // block-uniform: windows < cur_w are closed
    for (uint32_t s0 = beg; s0 < end; s0 += SLICE) {
        const uint32_t span = end - s0 < (uint32_t)SLICE ? end - s0 : (uint32_t)SLICE;
        const int wlo = (int)(rec[s0].x >> cb), whi = (int)(rec[s0 + span - 1].x >> cb);
        if (wlo < cur_w || whi < wlo || whi >= q.n_windows) { bad = true; break; }
        for (; cur_w < wlo; ++cur_w) close_window(cur_w);
        uint32_t c[C], o[C], a[C];
        PROF_MARK(0);
        slice_sort<NT, C>(rec, s0, span, sel, cnt, slot, red, c, o, cb);
        PROF_RESET;
        segments_order<C>(c, o, slot, SLICE - 1);
        PROF_MARK(4);
#pragma unroll
        for (int j = 0; j < C; ++j) a[j] = 0;
        for (int w = wlo; w <= whi; ++w) {
            // the cells advance together: one LDS read each per step, until none of them has a
            // record of window w next
            for (;;) {
                uint2 e[C];
#pragma unroll
                for (int j = 0; j < C; ++j) {
                    const uint32_t at = o[j] + a[j];
                    e[j] = slot[at < SLICE - 1 ? at : SLICE - 1];
                }
                uint32_t any = 0u;
#pragma unroll
                for (int j = 0; j < C; ++j) {
                    const bool take = a[j] < c[j] && (int)(e[j].y & 255u) == w;
                    const float nsum = sum[j] + __uint_as_float(e[j].x); // sum += t - 1 in stream order, generate_taf.py:26
                    sum[j] = take ? nsum : sum[j];
                    num[j] += take ? 1u : 0u;
                    a[j] += take ? 1u : 0u;
                    any |= take ? 1u : 0u;
                }
                if (!any) break;
            }
            if (w < whi) { close_window(w); cur_w = w + 1; }
        }
        PROF_MARK(5);
        bool viol = false; // a record whose window runs backwards inside its cell's segment
#pragma unroll
        for (int j = 0; j < C; ++j) viol |= a[j] != c[j];
        if (__syncthreads_or(viol)) { bad = true; break; }
        PROF_MARK(6);
    }
    if (!bad) {
        for (; cur_w < q.n_windows; ++cur_w) close_window(cur_w);
        PROF_MARK(7);
    } else {
        // ---- general path (stream not time-sorted): nothing has been written yet.  One pass per window
        // over the whole list, slices in list order, records selected by window.
        __syncthreads();
        if (t == 0) atomicAdd(&q.hdr->pad, 1u); // diagnostic: tiles on the general path
        load_state();
        for (int w = 0; w < q.n_windows; ++w) {
            for (uint32_t s0 = beg; s0 < end; s0 += SLICE) {
                const uint32_t span = end - s0 < (uint32_t)SLICE ? end - s0 : (uint32_t)SLICE;
                uint32_t c[C], o[C];
                slice_sort<NT, C>(rec, s0, span, [=](uint32_t m) { return (int)(m >> cb) == w ? sel(m) : -1; }, cnt, slot,
                                  red, c, o, cb);
                segments_order<C>(c, o, slot, SLICE - 1);
#pragma unroll
                for (int j = 0; j < C; ++j) {
                    for (uint32_t x = 0; x < c[j]; ++x) sum[j] = sum[j] + __uint_as_float(slot[o[j] + x].x);
                    num[j] += c[j];
                }
                __syncthreads();
            }
            close_window(w);
        }
    }

    // ---- write-out: state, optional f32 view (2K, H, W), optional uint8 leaky transform
    const long long plane = (long long)q.H * q.W;
#pragma unroll
    for (int j = 0; j < C; ++j) {
        if (!ow.ok[j]) continue;
        float *dst = q.state + (ow.pix[j] * 2 + ow.p) * K;
        if (K == 8) {
            ((float4 *)dst)[0] = make_float4(st[j][0], st[j][1], st[j][2], st[j][3]);
            ((float4 *)dst)[1] = make_float4(st[j][4], st[j][5], st[j][6], st[j][7]);
        } else {
#pragma unroll
            for (int k = 0; k < kMaxK; ++k) if (k < K) dst[k] = st[j][k];
        }
        if (q.view_f32) {
#pragma unroll
            for (int k = 0; k < kMaxK; ++k)
                if (k < K) q.view_f32[(long long)(2 * k + ow.p) * plane + ow.pix[j]] = st[j][k]; // :55
        }
        if (q.out_u8) {
            uint8_t lv[kMaxK];
            leaky_u8_lookup_n<kMaxK>(st[j], thr, lv); // the eight table look-ups in flight together
#pragma unroll
            for (int k = 0; k < kMaxK; ++k) {
                if (k < K) {
                    const int ko = q.flip ? (K - 1 - k) : k;
                    q.out_u8[(long long)(2 * ko + ow.p) * plane + ow.pix[j]] = lv[k];
                }
            }
        }
    }
    PROF_MARK(8);
}

// grid = 4 * kMaxHot + n_tiles: the four quarters of every listed hot tile, then one workgroup per tile
template <int NT>
__global__ __launch_bounds__(NT) void k_taf_tile(const uint2 *rec, const uint32_t *base, TafParams q, int n_tiles)
{
    __shared__ uint32_t cnt[NT * CPT];
    __shared__ uint2 slot[TileLds<NT>::SLICE];
    __shared__ uint32_t red[48];
    __shared__ uint32_t thr[kLeakyLevels];
    if (n_tiles < 0) { // small frame (-n_tiles tiles): every tile as four quarters, grid = 4 * tiles
        taf_tile_body<NT, 1>(rec, base, q, (int)blockIdx.x >> 2, (int)blockIdx.x & 3, cnt, slot, red, thr);
        return;
    }
    if ((int)blockIdx.x >= 4 * kMaxHot) {
        const int tile = (int)blockIdx.x - 4 * kMaxHot;
        if (tile_is_listed(q.hdr, base, tile)) return; // hot: left to its quarters
        taf_tile_body<NT, CPT>(rec, base, q, tile, 0, cnt, slot, red, thr);
    } else { // the long-running workgroups come first in the grid
        int tile, quarter;
        if (!pick_quarter(q.hdr, (int)blockIdx.x, tile, quarter)) return;
        taf_tile_body<NT, 1>(rec, base, q, tile, quarter, cnt, slot, red, thr);
    }
}

#define LAUNCH_TILE_GRID(KERNEL, GRID, PLAN, STREAM, ...)                                               \
    do {                                                                                                \
        if ((PLAN).twl == 8)                                                                            \
            hipLaunchKernelGGL(KERNEL<1024>, dim3(GRID), dim3(1024), 0, STREAM, __VA_ARGS__);           \
        else if ((PLAN).twl == 7)                                                                       \
            hipLaunchKernelGGL(KERNEL<512>, dim3(GRID), dim3(512), 0, STREAM, __VA_ARGS__);             \
        else                                                                                            \
            hipLaunchKernelGGL(KERNEL<256>, dim3(GRID), dim3(256), 0, STREAM, __VA_ARGS__);             \
    } while (0)
#define LAUNCH_TILE(KERNEL, PLAN, STREAM, ...) LAUNCH_TILE_GRID(KERNEL, (PLAN).n_tiles, PLAN, STREAM, __VA_ARGS__)
// With the four quarters of every listed hot tile in front of the per-tile workgroups (skew).  A frame whose tiles
// do not even give every SIMD of the chip one wavefront (GEN1: 150 tiles x 4 wavefronts on 1024 SIMDs) runs ALL its
// tiles as quarters: four times the wavefronts, one cell per thread, each quarter streaming its tile's records.
#define LAUNCH_TILE_Q(KERNEL, PLAN, STREAM, ...)                                                                   \
    do {                                                                                                           \
        if ((PLAN).n_tiles * ((4 << (PLAN).twl) / kWave) <= (PLAN).quarter_below)                                   \
            LAUNCH_TILE_GRID(KERNEL, (PLAN).n_tiles * 4, PLAN, STREAM, __VA_ARGS__, -(PLAN).n_tiles);              \
        else                                                                                                       \
            LAUNCH_TILE_GRID(KERNEL, (PLAN).n_tiles + kMaxHot * 4, PLAN, STREAM, __VA_ARGS__, (PLAN).n_tiles);     \
    } while (0)

} // namespace

// =============================================================================================
// C-ABI
// =============================================================================================
extern "C" {

#ifdef FRLW_TILE_PROF
int frlw_debug_prof(unsigned long long *out, int reset)
{
    if (out) hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = {}; hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof z); }
    return 0;
}
#endif

const char *frlw_version(void) { return "frlw_evd 0.5.0 gfx950"; }

size_t frlw_encoder_workspace_bytes(int64_t n_events, int H, int W)
{
    Plan p;
    if (!make_plan(n_events, H, W, nullptr, p)) return 0;
    // frlw_sae_encode / frlw_eci_encode take the two-launch form of taf_fast.hip when the call is eligible: its chunk-major tables
    // are a little larger than the general plan's (1 M events at 304x240: 372 against 347 KB on top of the same 8 n bytes of
    // records), and a caller that allocates exactly what this query returns must get that path too
    const size_t fast = sae_fast_workspace_bytes(n_events, H, W);
    return fast > p.bytes ? fast : p.bytes;
}

int frlw_encoder_path_counts(uint64_t counts[4])
{
    if (!counts) return FRLW_ERR_ARG;
    for (int i = 0; i < 4; ++i) counts[i] = (uint64_t)g_path_counts[i].load(std::memory_order_relaxed);
    return FRLW_OK;
}

int frlw_encoder_status(const void *workspace, frlw_stream_t stream, int *status_out)
{
    if (!workspace || !status_out) return FRLW_ERR_ARG;
    int32_t st = 0;
    uint32_t stall = 0u;
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(&st, workspace, sizeof(st), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&stall, (const char *)workspace + kStallOffset, sizeof(stall), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (stall) st |= ST_STALL; // (kept outside the range a late workgroup 0 resets: frlw_common.h kStallOffset)
    *status_out = (st & ST_INDEX) ? FRLW_ERR_INDEX : (st & ST_POLARITY) ? FRLW_ERR_POLARITY : (st & ST_SPAN) ? FRLW_ERR_SPAN : (st & ST_STALL) ? FRLW_ERR_HIP : FRLW_OK;
    return FRLW_OK;
}

int frlw_workspace_init(void *workspace, size_t workspace_bytes, frlw_stream_t stream)
{
    if (!workspace || workspace_bytes < kHeaderBytes) return FRLW_ERR_ARG;
    HIP_TRY(hipMemsetAsync(workspace, 0, kHeaderBytes, (hipStream_t)stream));
    return FRLW_OK;
}

int frlw_encoder_deferred_status(void *workspace, frlw_stream_t stream, int *status_out)
{
    if (!workspace || !status_out) return FRLW_ERR_ARG;
    int32_t st = 0;
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(&st, (char *)workspace + kStickyOffset, sizeof(st), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemsetAsync((char *)workspace + kStickyOffset, 0, sizeof(st), s));
    HIP_TRY(hipStreamSynchronize(s));
    *status_out = (st & ST_INDEX) ? FRLW_ERR_INDEX : (st & ST_POLARITY) ? FRLW_ERR_POLARITY : (st & ST_SPAN) ? FRLW_ERR_SPAN : (st & ST_STALL) ? FRLW_ERR_HIP : FRLW_OK;
    return FRLW_OK;
}

int frlw_eci_encode(const frlw_events_t *ev, int H, int W, float *out_f32, uint8_t *out_u8,
                    void *workspace, size_t workspace_bytes, frlw_stream_t stream)
{
    if (!out_f32 && !out_u8) return FRLW_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    EciParams q;
    q.H = H; q.W = W; q.twl = 0; q.tiles_x = 0; q.out_f32 = out_f32; q.out_u8 = out_u8;
    // generate_eventcountimage.py:32-34,41: n sequential f32 adds of 0.05f, > 1 -> 1, * 255
    volatile float acc = 0.0f;
    q.lut[0] = 0.0f;
    for (int n = 1; n <= 20; ++n) {
        acc = acc + 0.05f;
        float v = acc;
        q.lut[n] = (v > 1.0f ? 1.0f : v) * 255.0f;
    }
    // calls of the GEN1 class with at least 16 384 events: two launches through the chunk-major partition (taf_fast.hip:
    // kf_scatter_cm<.., 2> + kf_sae_sub<true>: 100 000 events 21 -> 12 us)
    if (tuning_knob(ev ? ev->tuning : nullptr, &frlw_tuning_t::staged_scatter, -1) != 0) {
        const int rc2 = sae_fast_try(ev, H, W, q.lut, 21, nullptr, nullptr, 0, 0, out_f32, out_u8, workspace, workspace_bytes, s);
        if (rc2 <= 0) return rc2;
    }
    // one launch for smaller calls: every workgroup of 2048 pixels reads all events (k_eci_scan); at most 8 M event reads in all
    // (36 workgroups x 100 000 events at 304x240: 29 -> ~8 us); frlw_tuning_t::staged_scatter = 0 keeps the general path (tests)
    if (ev && workspace && workspace_bytes >= kHeaderBytes && ev->layout == FRLW_LAYOUT_DAT8 && H > 0 && W > 0 && ev->n > 0 && ev->data &&
        tuning_valid(ev->tuning) && (ev->xmap == nullptr) == (ev->ymap == nullptr) &&
        tuning_knob(ev->tuning, &frlw_tuning_t::staged_scatter, -1) != 0) {
        const long long wgs = ((long long)H * W + kEciScanPix - 1) / kEciScanPix;
        // W <= 65536: k_eci_scan forms the flat pixel x + W * y in 32 bits, and x, y reach 65535 through the coordinate maps -- a
        // wider frame (which the general path places with 64-bit arithmetic) could wrap an out-of-frame event back into the frame
        if (wgs * ev->n <= (8ll << 20) && (long long)H * W < (1ll << 31) && W <= 65536) {
            (void)hipGetLastError();
            if (ev->xmap)
                hipLaunchKernelGGL(k_eci_scan<true>, dim3((unsigned)wgs), dim3(kEciScanThreads), 0, s, (const uint2 *)ev->data, (long long)ev->n,
                                   ev->xmap, ev->ymap, ev->map_w, ev->map_h, q, (WsHeader *)workspace);
            else
                hipLaunchKernelGGL(k_eci_scan<false>, dim3((unsigned)wgs), dim3(kEciScanThreads), 0, s, (const uint2 *)ev->data, (long long)ev->n,
                                   (const uint16_t *)nullptr, (const uint16_t *)nullptr, 0, 0, q, (WsHeader *)workspace);
            HIP_TRY(hipGetLastError());
            g_path_counts[3].fetch_add(1ull, std::memory_order_relaxed);
            return FRLW_OK;
        }
    }
    g_path_counts[3].fetch_add(1ull, std::memory_order_relaxed);
    Partitioned pt;
    int rc = partition_events(ev, H, W, KIND_ECI, 0, 1, 1, 0, workspace, workspace_bytes, s, pt);
    if (rc != FRLW_OK) return rc;
    q.twl = pt.plan.twl; q.tiles_x = pt.plan.tiles_x;
    LAUNCH_TILE(k_eci_tile, pt.plan, s, pt.records, pt.base, q);
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
    q.H = H; q.W = W; q.twl = pt.plan.twl; q.tiles_x = pt.plan.tiles_x; q.bins = bins;
    q.hdr = pt.hdr; q.out_f32 = out_f32; q.out_u8 = out_u8;
    if (bins <= 5) LAUNCH_TILE_Q(k_ev_tile5, pt.plan, s, pt.records, pt.base, q);
    else LAUNCH_TILE_Q(k_ev_tile, pt.plan, s, pt.records, pt.base, q);
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
    {   // single calls of the GEN1 class: two launches instead of five (taf_fast.hip: kf_scatter_cm<.., SAE> + kf_sae_sub)
        float lamf[FRLW_MAX_LAMDAS];
        for (int l = 0; l < n_lamda; ++l) lamf[l] = (float)lamdas[l];
        const int rc2 = sae_fast_try(ev, H, W, lamf, n_lamda, mem_in, mem_out, now, window_us, out_f32, out_u8, workspace, workspace_bytes, s);
        if (rc2 <= 0) return rc2;
    }
    g_path_counts[1].fetch_add(1ull, std::memory_order_relaxed);
    Partitioned pt;
    const int filt = window_us > 0;
    int rc = partition_events(ev, H, W, KIND_SAE, now - window_us, 1, 1, filt, workspace,
                              workspace_bytes, s, pt);
    if (rc != FRLW_OK) return rc;
    SaeParams q;
    q.H = H; q.W = W; q.twl = pt.plan.twl; q.tiles_x = pt.plan.tiles_x; q.n_lamda = n_lamda;
    for (int l = 0; l < n_lamda; ++l) q.lam[l] = (float)lamdas[l];
    q.nowf = (float)now;
    q.mem_in = mem_in; q.mem_out = mem_out; q.out_f32 = out_f32; q.out_u8 = out_u8;
    LAUNCH_TILE(k_sae_tile, pt.plan, s, pt.records, pt.base, q);
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
    q.H = H; q.W = W; q.twl = pt.plan.twl; q.tiles_x = pt.plan.tiles_x; q.K = K;
    q.n_windows = n_windows;
    q.flip = (flags & FRLW_TAF_U8_FLIP_K) ? 1 : 0;
    q.hdr = pt.hdr; q.state = state; q.view_f32 = view_f32; q.out_u8 = out_u8; q.leaky_thr = pt.leaky_thr;
    LAUNCH_TILE_Q(k_taf_tile, pt.plan, s, pt.records, pt.base, q);
    HIP_TRY(hipGetLastError());
    return FRLW_OK;
}

} // extern "C"
