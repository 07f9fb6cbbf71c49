// taf_fast.hip -- the fast Temporal Active Focus path: a batch of independent, in-span DAT streams.
//
// Replaces, for B sequences at once, the harness loop generate_taf.py:193-235 around taf_cuda (:19-58): window
// selection, f64 time normalisation, per-window count / mean-time accumulation, K-deep FIFO ageing, leaky transform,
// uint8 truncation.  Same contract as the general path (encoders.hip): f32 sums in STREAM ORDER, bit for bit.
//
// What makes it fast (measured reasons in DESIGN.md section 3):
//   * 4-byte records {r | window | cell}: r = t - window start (the f32 value is tlut[r], one table of win + 1
//     floats), 12-bit cell inside a 4096-cell tile.  Half the record traffic of the general path.
//   * the stable partition ranks a batch of 64 consecutive events with ONE returning LDS atomic per event: gfx950
//     serves the lanes of a wave-instruction that hit one LDS address in ascending lane order (tools/lds_order_test.hip,
//     tests/test_taf_fast_gpu.py::test_lds_atomic_lane_order), so the returned value IS the stream rank.
//   * the records of a tile (4096 cells) are split once more, stably, by sub-tile of 256 cells (kf_split_whole; tiles
//     that hold a large share of the stream are cut into segments of 8192 records with one workgroup each,
//     kf_split_place), so skew costs more workgroups, not a longer critical path.
//   * the sums of different windows do not depend on each other -- only the FIFO steps that consume them are
//     sequential.  kf_taf_walk gives a sub-tile to eight wavefronts that take ONE WINDOW EACH: tickets from per-cell
//     LDS counters (the same lane-ordered atomic) turn the window's records into per-cell segments without any
//     ordering pass, owner lanes add each segment front to back (the reference's sequential index_add_), and the FIFO
//     steps follow with one cell per lane.
//   * sequences of a batch are independent problems in one launch sequence: own tiles, own window mask
//     (generate_taf.py:40-41 is a per-sequence rule), own t_start.
//
// Requirements, checked on device (violations -> status, NOTHING is written, the caller falls back to frlw_taf_encode):
// every event inside [t_start, t_start + n_windows * window_us] of its sequence and inside the frame.

#include "frlw_common.h"

#include <atomic>
#include <stddef.h>
#include <string.h>
#include <type_traits>

using namespace frlw;

namespace {

// TAF through the tile walk (kf_taf_tile) instead of kf_split_whole + kf_taf_walk: bit-identical, 23 % less HBM traffic,
// but measured SLOWER on MI355X (141 vs 121 us of tile work at 10 M events) -- so it is opt-in
// (frlw_tuning_t::taf_tile_walk), and the Event Volume batch path, which has no other tile kernel, is its user.
#ifndef FRLW_TAF_TILE_WALK
#define FRLW_TAF_TILE_WALK 0
#endif
constexpr bool kTafTileWalk = FRLW_TAF_TILE_WALK != 0;
constexpr int kFT = 1024;                 // threads of every workgroup here
constexpr int kFW = kFT / kWave;          // 16 wavefronts
constexpr int kCellBits = 12;
constexpr int kCells = 1 << kCellBits;    // cells (pixel x polarity) per tile
constexpr int kSubCells = kCells / kFW;   // 256 cells owned by one wavefront of the tile kernel
constexpr int kPixLog = 11;               // log2 pixels per tile
constexpr int kMaxSeq = FRLW_MAX_SEQUENCES;
constexpr int kMaxFastTiles = 1024;       // tiles per sequence (LDS of the scatter workgroup: 72 B per tile)
constexpr int kMaxPairs = 8192;           // (sequence, tile) pairs per call (LDS of the tile scan: one round)
constexpr int kMaxBinPairs = 65536;       // (sequence, bin) pairs per call in the direct mode (the tile scan runs in rounds)
constexpr int kFastSlab = 32;             // chunks per slab of the two-level column scan
constexpr int ST_MULBAD = 8;              // per-chunk flag next to the ST_* error bits (not an error)
constexpr int kSplitSeg = 8192;           // records one workgroup of the sub-tile split handles
constexpr int kBigBpw = 20;               // 64-event batches per wavefront of the one-workgroup-per-CU form of kf_scatter_cm
#ifndef FRLW_WHOLE_SEGS
#define FRLW_WHOLE_SEGS 4 /* measured: 8 -> 4 takes 7 % (TAF hot spot at 10 M events) to 15 % (Event Volume batch with hot spots) off skewed calls, uniform calls unchanged; 3 sends ordinary 25 000-record GEN1 tiles through the segments (+9 %) */
#endif
constexpr int kSplitWhole = FRLW_WHOLE_SEGS * kSplitSeg; // tiles up to this many records are split by ONE workgroup (kf_split_whole) ...
#ifndef FRLW_FEW_PAIRS
#define FRLW_FEW_PAIRS 256
#endif
constexpr int kFewPairs = FRLW_FEW_PAIRS;             // ... unless the call has fewer (sequence, tile) pairs than this: one workgroup per
                                           // tile would leave most CUs idle (one GEN1 stream: 20 tiles of 50 000 records took
                                           // 31 us), so every tile above one segment goes through the segment kernels
__host__ __device__ inline uint32_t whole_max_of(int pairs) { return pairs < kFewPairs ? (uint32_t)kSplitSeg : (uint32_t)kSplitWhole; }
__host__ __device__ inline uint32_t split_segments(uint32_t n, uint32_t whole_max) { return n > whole_max ? (n + kSplitSeg - 1) / kSplitSeg : 0u; }

struct SeqTab { // kernel argument, built on the host
    int n_seq;
    int chunk0[kMaxSeq + 1];    // first chunk of sequence s; [n_seq] = total
    int slab0[kMaxSeq + 1];     // first slab
    long long ev0[kMaxSeq + 1]; // first event
    long long t0[kMaxSeq];      // t_start
};

struct FastGeom {
    const uint2 *data;
    const uint16_t *xmap, *ymap;
    int map_w, map_h;
    int H, W, twl, thl, tiles_x, T;
    int bpw;      // batches of 64 events per wavefront of a partition workgroup = ceil(run / 64)
    int chunk_ev; // events per chunk (one scatter workgroup), a multiple of 16, <= 8192
    int run;      // events per wavefront of the scatter workgroup = chunk_ev / 16
    long long n_total; // records in the array (loads never go past it)
    int order_check;   // TAF: flag sequences whose window index ever decreases (only the tile walk needs to know)
    int y_lo, H_full;  // row-stripe sharding of one frame: this call encodes rows [y_lo, y_lo + H) of an H_full-row frame; events
                       // of other rows are skipped (not an error); H_full == H, y_lo == 0: the whole frame
    int n_windows, wb;
    uint32_t win, win_magic;
    int bin_shift, bin_mask; // direct mode (FastPlan): bin = tile << 4 | sub-tile of the cell; otherwise 0, 0: bin = tile
    int simple;   // the call meets the conditions of the SIMPLE decode (below): decided on the host
    uint32_t span; // n_windows * win when that fits 32 bits (SIMPLE)
    double rcp; // 1 / (win + 1e-8) (TAF) or 1 / win (Event Volume), IEEE f64, computed ONCE on the host: kf_hist checks that
                // multiplying by it gives every r of the window the float the division gives; the tile kernels multiply
};

struct FastHeader {
    int32_t status; // ST_* flags; same offset as WsHeader::status (frlw_encoder_status reads it)
    uint32_t filtered_tiles; // diagnostic: sub-tiles whose records were not window-sorted (unsorted stream)
    unsigned long long wmask[kMaxSeq]; // bit w set <=> window w of the sequence holds at least one event
    uint32_t mul_bad; // != 0: float(r * (1 / den)) differs from float(r / den) for some r in [0, win]: use the table
    uint32_t unsorted[kMaxSeq]; // != 0: somewhere in the sequence an event's window is lower than its predecessor's (the
                                // tile walk needs window-sorted lists; such a sequence takes the split + sub-tile kernels)
    // chunk-major partition: where the next (sequence, tile) list goes in rec2[] / the next split segment id (the header is
    // zeroed by a memset node in front of kf_scatter_cm; placement order is whatever order the workgroups arrive in -- the
    // lists themselves, and everything computed from them, do not depend on it)
    uint32_t rec_cursor, seg_cursor;
    // kf_scatter_cm's first workgroup resets everything above and then publishes the call's epoch here; the other workgroups
    // touch the header only at their very end and only once they see that epoch (no memset node: 4.8 us of every call)
    uint32_t epoch;
};
static_assert(sizeof(FastHeader) <= kSelftestOffset, "header");

struct FastPlan {
    int twl, thl, tiles_x, tiles_y, T;
    // Direct mode (small calls on small frames): the partition's bins are the 256-cell SUB-TILES (16 per tile) instead of the tiles, so kf_scatter's output
    // already is sub-tile-major and the second-level split (kf_split_whole / kf_split_place: a read + write of every record)
    // is not run at all.  Possible while a sequence has at most kMaxFastTiles bins (the scatter workgroup keeps 16 counters
    // per bin in LDS): the GEN1 / 304x240 class of frames (36 tiles = 576 bins), not 1280x720 (450 tiles = 7200 bins).
    int direct, TB, bin_shift, pairs_b; // bins per sequence (T or 16 T), log2 of bins per tile, (sequence, bin) pairs
    int big;                            // chunk-major partition with chunks above 8192 events (kf_scatter_cm<.., kBigBpw>)
    int bpw, chunk;
    int chunks, slabs, pairs;
    size_t off_counts, off_slabtot, off_base, off_sub, off_seg0, off_segcnt, off_errs, off_tlut, off_records, off_records2, bytes;
    size_t off_sub_end, off_segdesc; // chunk-major partition: list ends, pair of every split segment
    size_t off_wst, off_wst_flag;    // TileP::wst / wst_flag
    int max_seq_chunks;              // chunks of the longest sequence (the column a chunk-major consumer keeps in LDS)
    int max_segs;
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Tile shape: 2^twl x 2^(11 - twl) pixels, the one that covers the frame with the fewest tiles (ties: the widest,
// longest contiguous rows).
enum : int { DIRECT_OFF = 0, DIRECT_AUTO = 1, DIRECT_FORCE = 2 };
bool fast_plan(long long n, int n_seq, int H, int W, FastPlan &p, int direct_mode = DIRECT_AUTO, int min_bpw = 0, bool cm = false)
{
    if (H <= 0 || W <= 0 || n < 0 || n_seq < 1 || n_seq > kMaxSeq || n >= (1ll << 31)) return false;
    long long best = -1;
    for (int twl = 5; twl <= 8; ++twl) {
        const int tw = 1 << twl, th = 1 << (kPixLog - twl);
        const long long t = (long long)((W + tw - 1) / tw) * ((H + th - 1) / th);
        if (best < 0 || t <= best) { best = t; p.twl = twl; }
    }
    p.thl = kPixLog - p.twl;
    p.tiles_x = (W + (1 << p.twl) - 1) >> p.twl;
    p.tiles_y = (H + (1 << p.thl) - 1) >> p.thl;
    p.T = p.tiles_x * p.tiles_y;
    if (p.T > kMaxFastTiles || (long long)p.T * n_seq > kMaxPairs) return false;
    p.pairs = p.T * n_seq;
    // AUTO: calls with few (sequence, tile) pairs whose tiles hold more than one split segment on average -- the launch-bound
    // ones (one GEN1 stream of 1 M events: 52 -> 42 us; eight: 150 -> 136 us; tools/time_direct.py).  With many pairs
    // kf_split_whole is the cheaper second level (64 GEN1 streams: 820 us against 876 direct), with few events per tile the
    // 576-bin scatter costs more than the whole-tile split it replaces (one stream of 250 k events: 40 us against 46):
    // there only when forced (frlw_tuning_t::direct_bins = 1).
    // Chunk-major partition (cm): the consumers of a direct-mode call gather their own lists -- two launches in all -- so every
    // call with few pairs goes that way, however few events a pair holds (5 sequences of 70 k events on 97x131: 36 us against 55).
    p.direct = (direct_mode != DIRECT_OFF && kFW * p.T <= kMaxFastTiles && (long long)kFW * p.T * n_seq <= kMaxBinPairs &&
                (direct_mode == DIRECT_FORCE || (p.pairs < 2 * kFewPairs && (cm || n >= (long long)kSplitSeg * p.pairs)))) ? 1 : 0;
    p.TB = p.direct ? kFW * p.T : p.T;
    p.bin_shift = p.direct ? 4 : 0;
    p.pairs_b = p.TB * n_seq;
    // Chunk size: the partition kernels run two workgroups per CU (512 at a time), and a grid that is not a whole number
    // of such rounds ends on a part-filled one (10 M events in chunks of 8192 = 2.4 rounds: the last one 38 % full).  So
    // the stream is cut into 512 * k chunks with the smallest k whose chunks fit the 8192-event staging area.
    // (cm with min_bpw above kMaxBpw: the one-workgroup-per-CU form of kf_scatter_cm, chunks up to kBigBpw batches per
    // wavefront in 256 * k chunks -- tile bins only, and only while staging area + counters fit the CU's LDS)
    // Default: frames with many tiles and streams long enough to fill two rounds of 256 such chunks -- where the ordinary chunks
    // would leave a consumer runs of a dozen records (10 M events at 1280x720: 164 us against 171 with 8192-event chunks and 177
    // with the histogram partition; 3 M events: 113 against 111 -- hence the lower bound; 64 GEN1 streams have 200-record runs
    // either way and lose 1.5 % to the lower occupancy)
    const bool big = cm && !p.direct && (min_bpw > kMaxBpw || (min_bpw == 0 && p.TB >= 256 && n >= 6000000)) &&
                     (long long)kFW * p.TB * 4 + (p.TB + 2) * 4 + (long long)kFT * kBigBpw * 4 + 64 <= 150 * 1024;
    p.big = big ? 1 : 0;
    long long cap = (long long)kFT * (big ? kBigBpw : kMaxBpw);
    const long long round = big ? 256 : 512;
    if (p.direct) { // 16 counters per bin: shorter chunks keep the scatter workgroup at two per CU (78 KB of LDS)
        const long long lds_cap = ((79ll * 1024 - 16 - (long long)kFW * p.TB * 4 - (p.TB + 2) * 4) / 6) / 16 * 16;
        if (lds_cap >= 2048 && lds_cap < cap) cap = lds_cap;
    }
    long long k = (n + round * cap - 1) / (round * cap);
    if (k < 1) k = 1;
    long long ce = (n + round * k - 1) / (round * k);
    ce = (ce + 15) / 16 * 16;
    // cm: a consumer gathers one run per chunk, so a call that cannot fill 512 workgroups anyway takes the largest chunks there are
    // (one GEN1 stream: 144 chunks of 6944 events instead of 509 of 1968: 32 us against 50)
    if (cm && k == 1) ce = cap;
    if (ce < 1024) ce = 1024; // tiny calls: keep whole 64-event batches per wavefront
    // frlw_tuning_t::batches_per_wave: at least this many 64-event batches per wavefront (only ever LARGER chunks than the
    // default: the workspace query budgets the default's chunk count)
    if (min_bpw > 0 && ce < (long long)min_bpw * kFT) ce = (long long)min_bpw * kFT;
    if (ce > cap) ce = cap;
    p.chunk = (int)ce;
    p.bpw = (p.chunk / kFW + kWave - 1) / kWave;
    return true;
}

// per-sequence chunk / slab tables + workspace layout
bool fast_layout(const int64_t *seq_offsets, const int64_t *t_start, int n_seq, FastPlan &p, SeqTab &S, uint32_t win)
{
    S.n_seq = n_seq;
    int c = 0, sl = 0;
    p.max_seq_chunks = 1;
    for (int s = 0; s < n_seq; ++s) {
        const long long n_s = seq_offsets[s + 1] - seq_offsets[s];
        if (n_s < 0) return false;
        S.chunk0[s] = c;
        S.slab0[s] = sl;
        S.ev0[s] = seq_offsets[s];
        S.t0[s] = t_start[s];
        int cs = (int)((n_s + p.chunk - 1) / p.chunk);
        if (cs < 1) cs = 1; // an empty sequence keeps one (empty) chunk: no special cases downstream
        if (cs > p.max_seq_chunks) p.max_seq_chunks = cs;
        c += cs;
        sl += (cs + kFastSlab - 1) / kFastSlab;
    }
    S.chunk0[n_seq] = c;
    S.slab0[n_seq] = sl;
    S.ev0[n_seq] = seq_offsets[n_seq];
    p.chunks = c;
    p.slabs = sl;
    const long long n = seq_offsets[n_seq] - seq_offsets[0];
    size_t off = kHeaderBytes;
    p.off_counts = off;  off = align_up(off + (size_t)(c > 0 ? c : 1) * p.TB * 4, 256);
    p.off_slabtot = off; off = align_up(off + (size_t)(sl > 0 ? sl : 1) * p.TB * 4, 256);
    p.off_base = off;    off = align_up(off + (size_t)(p.pairs_b + 1) * 4, 256);
    p.off_sub = off;     off = align_up(off + ((size_t)p.pairs * kFW + 1) * 4, 256);
    p.max_segs = 2 * (int)(n / kSplitSeg) + 1; // tiles above the whole-tile limit (>= one segment): full segments + one partial each
    p.off_seg0 = off;    off = align_up(off + (size_t)(p.pairs_b + 1) * 4, 256);
    p.off_segcnt = off;  off = align_up(off + (size_t)p.max_segs * kFW * 4, 256);
    p.off_errs = off;    off = align_up(off + (size_t)(c > 0 ? c : 1) * 4, 256);
    p.off_tlut = off;    off = align_up(off + (size_t)(win + 1) * 4, 256);
    p.off_records = off; off = align_up(off + (size_t)(n > 0 ? n : 1) * 4, 256);
    p.off_records2 = off; off = align_up(off + (size_t)(n > 0 ? n : 1) * 4, 256);
    p.off_sub_end = off; off = align_up(off + ((size_t)p.pairs * kFW + 1) * 4, 256);
    p.off_segdesc = off; off = align_up(off + (size_t)p.max_segs * 4, 256);
    p.off_wst = off;     off = align_up(off + (size_t)p.pairs * kFW * (FRLW_MAX_WINDOWS + 1) * 4, 256);
    p.off_wst_flag = off; off = align_up(off + (size_t)p.pairs * 4, 256);
    p.bytes = off;
    return true;
}

// ---- decode ----------------------------------------------------------------------------------------
struct FastEv {
    int tile;      // < 0: not encoded (err says why)
    uint32_t word; // r << (12 + wb) | window << 12 | cell
    uint32_t window;
    int err;
};

// src/io/dat_events_tools.py:96-98 (bit fields), generate_taf.py:197-203 (window), :215-219 (coordinate scaling via the
// maps); the flat index x + W * y of generate_taf.py:23 aliases x >= W into the next row like the general path.
// EV (Event Volume, generate_eventvolume.py:139-141): t0 = t_end - window; events with t <= t0 are dropped like the
// harness' `events_[:, 2] > end_time - time_window` filter, an event behind t_end is outside the contract (ST_SPAN);
// word = (t - t0) << 12 | cell, one "window".
// SIMPLE (chosen per call on the host, FastGeom::simple): whole frame (no row stripe), every sequence's t0 in [0, 2^32), the
// span n_windows * win below 2^32, win >= 2 -- then the time arithmetic is 32-bit, the stripe test disappears and the window
// needs ONE correction step after the multiply-high (floor(2^32 / win) under-estimates the quotient by less than one).  Same
// results as the general form on such calls; 14 of the decode's 52 VALU instructions less in kf_hist and kf_scatter.
// SAE (Surface of Active Events, generate_surfaceofactiveevents.py:72, :176-190; an EV-shaped decode): events outside the frame
// and events at or in front of t0 = now - window are dropped without an error, there is no upper time bound, and the record's
// time field is replaced by the caller with the event's position in its sequence (the consumer wants the LAST writer).
// SAE == 2 (Event Count Image, generate_eventcountimage.py:19-41): the Event Volume decode -- x >= W aliases into the next row,
// a flat pixel outside the frame is an error -- without any time bound (the host passes t0 = -1: every event is kept).
template <bool HAS_MAP, bool EV = false, bool SIMPLE = false, int SAE = 0>
__device__ __forceinline__ FastEv fast_decode(const FastGeom &G, uint2 r, long long t0)
{
    FastEv o;
    o.tile = -1; o.word = 0; o.window = 0; o.err = 0;
    int x = (int)(r.y & 16383u), y = (int)((r.y >> 14) & 16383u);
    const uint32_t p = (r.y >> 28) & 1u;
    if (HAS_MAP) {
        if (x >= G.map_w || y >= G.map_h) { o.err = ST_INDEX; return o; }
        x = G.xmap[x];
        y = G.ymap[y];
    }
    if (SAE == 1 && (x >= G.W || y >= G.H_full)) return o; // generate_surfaceofactiveevents.py:72
    if (x >= G.W || y >= G.H_full) {
        const long long flat = (long long)x + (long long)G.W * y;
        if (flat >= (long long)G.H_full * G.W) { o.err = ST_INDEX; return o; }
        y = (int)(flat / G.W);
        x = (int)(flat - (long long)y * G.W);
    }
    if (SIMPLE) {
        const uint32_t t0lo = (uint32_t)t0, relu = r.x - t0lo;
        if (EV) {
            if (r.x <= t0lo) return o; // generate_eventvolume.py:139: not an error, not encoded
            if (!SAE && relu > G.win) { o.err = ST_SPAN; return o; }
            const int tw1e = (1 << G.twl) - 1, th1e = (1 << G.thl) - 1;
            const uint32_t celle = (uint32_t)((((y & th1e) << G.twl) | (x & tw1e)) << 1) | p;
            o.tile = (((y >> G.thl) * G.tiles_x + (x >> G.twl)) << G.bin_shift) | (int)((celle >> 8) & (uint32_t)G.bin_mask);
            o.word = (relu << kCellBits) | celle;
            return o;
        }
        if (r.x < t0lo || relu > G.span) { o.err = ST_SPAN; return o; }
        uint32_t z = __umulhi(relu, G.win_magic); // floor(rel / win) or one less
        uint32_t rem = relu - z * G.win;
        if (rem >= G.win) { ++z; rem -= G.win; }
        if (z >= (uint32_t)G.n_windows) { z = (uint32_t)G.n_windows - 1u; rem = G.win; } // t == end of the last window
        const int tw1 = (1 << G.twl) - 1, th1 = (1 << G.thl) - 1;
        const uint32_t cell = (uint32_t)((((y & th1) << G.twl) | (x & tw1)) << 1) | p;
        o.tile = (((y >> G.thl) * G.tiles_x + (x >> G.twl)) << G.bin_shift) | (int)((cell >> 8) & (uint32_t)G.bin_mask);
        o.window = z;
        o.word = (rem << (kCellBits + G.wb)) | (z << kCellBits) | cell;
        return o;
    }
    y -= G.y_lo; // row-stripe sharding: another rank owns the rows outside [y_lo, y_lo + H)
    if ((unsigned)y >= (unsigned)G.H) return o;
    const long long rel = (long long)r.x - t0;
    if (EV) {
        if (rel <= 0) return o; // generate_eventvolume.py:139: not an error, not encoded
        if (!SAE && rel > (long long)G.win) { o.err = ST_SPAN; return o; }
        const int tw1e = (1 << G.twl) - 1, th1e = (1 << G.thl) - 1;
        const uint32_t celle = (uint32_t)((((y & th1e) << G.twl) | (x & tw1e)) << 1) | p;
        o.tile = (((y >> G.thl) * G.tiles_x + (x >> G.twl)) << G.bin_shift) | (int)((celle >> 8) & (uint32_t)G.bin_mask);
        o.word = ((uint32_t)rel << kCellBits) | celle;
        return o;
    }
    if (rel < 0 || rel > (long long)G.n_windows * G.win) { o.err = ST_SPAN; return o; }
    const uint32_t relu = (uint32_t)rel;
    uint32_t z = __umulhi(relu, G.win_magic); // floor(rel / win) - {0, 1, 2}
    uint32_t rem = relu - z * G.win;
    if (rem >= G.win) { ++z; rem -= G.win; }
    if (rem >= G.win) { ++z; rem -= G.win; }
    if (z >= (uint32_t)G.n_windows) { z = (uint32_t)G.n_windows - 1u; rem = G.win; } // t == end of the last window
    const int tw1 = (1 << G.twl) - 1, th1 = (1 << G.thl) - 1;
    const uint32_t cell = (uint32_t)((((y & th1) << G.twl) | (x & tw1)) << 1) | p;
    o.tile = (((y >> G.thl) * G.tiles_x + (x >> G.twl)) << G.bin_shift) | (int)((cell >> 8) & (uint32_t)G.bin_mask);
    o.window = z;
    o.word = (rem << (kCellBits + G.wb)) | (z << kCellBits) | cell;
    return o;
}

// largest s with first[s] <= v (first[0] = 0; empty sequences repeat a value: the last of them wins, like a linear walk).
// Bisection: the table sits in the kernel arguments, every probe is a DEPENDENT scalar load -- the linear walk this replaces
// cost a 64-sequence call up to 64 of them per chunk and wavefront (66 M scalar instructions in kf_hist for 64 M events).
__device__ __forceinline__ int seq_of(const int *first, int n_seq, int v)
{
    int lo = 0, hi = n_seq;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (v >= first[mid]) lo = mid; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ int seq_of_chunk(const SeqTab &S, int chunk) { return seq_of(S.chunk0, S.n_seq, chunk); }

// ---- 1. histogram ------------------------------------------------------------------------------------
// Persistent workgroups (two per CU) walk the chunks grid-stride; a thread takes eight records of a chunk as four 16-byte
// loads (the order inside a chunk does not matter for a histogram) and has the NEXT chunk's loads in flight while it
// decodes and counts this one: the read of the 8-byte records runs at the copy rate instead of in bursts.
struct HistSpan { // wave-uniform description of one chunk's records
    long long first; // index of the first record the loads cover (a 16-byte boundary; may lie one record in front of the chunk)
    long long begin, end;
    long long t0;
};

__device__ __forceinline__ HistSpan hist_span(const FastGeom &G, const SeqTab &S, int chunk, int n_chunks)
{
    HistSpan L;
    L.begin = L.end = L.first = 0;
    L.t0 = 0;
    if (chunk >= n_chunks) return L;
    const int s = seq_of_chunk(S, chunk);
    L.t0 = S.t0[s];
    L.begin = S.ev0[s] + (long long)(chunk - S.chunk0[s]) * G.chunk_ev;
    L.end = L.begin + G.chunk_ev < S.ev0[s + 1] ? L.begin + G.chunk_ev : S.ev0[s + 1];
    if (L.end < L.begin) L.end = L.begin;
    // record pairs on 16-byte boundaries: step one record back if the chunk starts on the odd half of a pair (stays
    // inside the array unless the array itself starts there: then the loads are merely unaligned)
    L.first = L.begin - (long long)((reinterpret_cast<uintptr_t>(G.data + L.begin) >> 3) & 1u);
    if (L.first < 0) L.first = L.begin;
    return L;
}

// Loads without branches (a load under a lane condition becomes its own basic block with its own s_waitcnt: eight
// serialized round trips to HBM): every thread reads SOME pair of the chunk -- its own, or the chunk's last one -- and the
// validity of the two records is decided afterwards from the indices.
__device__ __forceinline__ uint32_t hist_pairs(const HistSpan &L) // whole pairs inside [first, end): both records exist
{
    long long cover = L.end - L.first;
    if (cover > 2ll * (kMaxBpw / 2) * kFT) cover = 2ll * (kMaxBpw / 2) * kFT;
    return (uint32_t)(cover >> 1);
}

__device__ __forceinline__ void hist_issue(const FastGeom &G, const HistSpan &L, uint4 (&v)[kMaxBpw / 2])
{
    const uint32_t pairs = hist_pairs(L);
    if (pairs > 0) { // wave-uniform
        const uint4 *src = (const uint4 *)(G.data + L.first);
#pragma unroll
        for (int j = 0; j < kMaxBpw / 2; ++j) {
            const uint32_t pj = (uint32_t)(j * kFT) + threadIdx.x;
            v[j] = src[pj < pairs ? pj : pairs - 1u];
        }
    }
}

template <bool HAS_MAP, bool EV = false, bool SIMPLE = false>
__global__ __launch_bounds__(kFT) __attribute__((amdgpu_waves_per_eu(8, 8))) void kf_hist(FastGeom G, SeqTab S, uint32_t *counts,
                                                                                         int32_t *errs, float *tlut_w, int n_chunks)
{
    extern __shared__ uint32_t lds[];
    uint32_t *hist = lds; // [T]
    __shared__ int serr;
    const int tid = threadIdx.x;
    for (int b = tid; b < G.T; b += kFT) hist[b] = 0;
    if (tid == 0) serr = 0;
    int mul_err = 0;
    if (EV) {
        // Event Volume: tlut[r] = float(r / window) (generate_eventvolume.py:141, :23: t.float()), r = t - (t_end - window);
        // the same exhaustive check decides whether the tile kernels may multiply by 1 / window instead
        const double den = (double)G.win, rcp = G.rcp;
        for (long long r = (long long)blockIdx.x * kFT + tid; r <= (long long)G.win; r += (long long)gridDim.x * kFT) {
            const float exact = (float)((double)r / den);
            if ((float)((double)r * rcp) != exact) mul_err = ST_MULBAD;
            tlut_w[r] = exact;
        }
    } else if (tlut_w) {
        // tlut[r] = float(r / (win + 1e-8)) - 1 (generate_taf.py:215, :26): one correctly rounded f64 division per
        // distinct in-window time instead of one per event
        // The walk kernel would rather multiply by 1 / den than gather from the table: allowed only if that gives the
        // same float for EVERY r of the domain, which is checked right here, exhaustively, per call.
        const double den = (double)G.win + 1e-8, rcp = G.rcp;
        for (long long r = (long long)blockIdx.x * kFT + tid; r <= (long long)G.win; r += (long long)gridDim.x * kFT) {
            const float exact = (float)((double)r / den);
            if ((float)((double)r * rcp) != exact) mul_err = ST_MULBAD;
            tlut_w[r] = exact - 1.0f;
        }
    }
    uint4 cur[kMaxBpw / 2], nxt[kMaxBpw / 2];
    HistSpan Lc, Ln = hist_span(G, S, (int)blockIdx.x, n_chunks);
    hist_issue(G, Ln, nxt);
    if (mul_err) atomicOr(&serr, mul_err);
    __syncthreads();
    for (int chunk = (int)blockIdx.x; chunk < n_chunks; chunk += (int)gridDim.x) {
        Lc = Ln;
#pragma unroll
        for (int j = 0; j < kMaxBpw / 2; ++j) cur[j] = nxt[j];
        Ln = hist_span(G, S, chunk + (int)gridDim.x, n_chunks);
        hist_issue(G, Ln, nxt);
        int err = 0;
        const uint32_t pairs = hist_pairs(Lc);
        const bool skip_first = Lc.first != Lc.begin; // the first record of pair 0 lies in front of the chunk
#pragma unroll
        for (int j = 0; j < kMaxBpw / 2; ++j) {
            const uint32_t pj = (uint32_t)(j * kFT + tid);
            if (pj < pairs && !(skip_first && pj == 0u)) {
                const FastEv o = fast_decode<HAS_MAP, EV, SIMPLE>(G, make_uint2(cur[j].x, cur[j].y), Lc.t0);
                err |= o.err;
                if (o.tile >= 0) atomicAdd(&hist[o.tile], 1u);
            }
            if (pj < pairs) {
                const FastEv o = fast_decode<HAS_MAP, EV, SIMPLE>(G, make_uint2(cur[j].z, cur[j].w), Lc.t0);
                err |= o.err;
                if (o.tile >= 0) atomicAdd(&hist[o.tile], 1u);
            }
        }
        // what the whole pairs leave over -- the odd record at the chunk's end (also when the chunk starts on the odd half of
        // a pair and is covered from one record earlier): at most one, fetched by one thread
        if (tid == 0 && Lc.end > Lc.begin && Lc.first + 2ll * pairs < Lc.end) { // (an EMPTY chunk covered from one record earlier has nothing left over)
            const FastEv o = fast_decode<HAS_MAP, EV, SIMPLE>(G, G.data[Lc.end - 1], Lc.t0);
            err |= o.err;
            if (o.tile >= 0) atomicAdd(&hist[o.tile], 1u);
        }
        if (err) atomicOr(&serr, err);
        __syncthreads();
        uint32_t *row = counts + (long long)chunk * G.T;
        for (int b = tid; b < G.T; b += kFT) { row[b] = hist[b]; hist[b] = 0u; }
        if (tid == 0) { errs[chunk] = serr; serr = 0; }
        __syncthreads();
    }
}

// ---- 2. scans ----------------------------------------------------------------------------------------
// counts[c][b], c in one slab of 32 chunks of ONE sequence -> exclusive prefix over c (in place), slabtot[slab][b]
__device__ __forceinline__ void slabscan_one(const SeqTab &S, uint32_t *counts, int T, uint32_t *slabtot, int slab, int b)
{
    const int s = seq_of(S.slab0, S.n_seq, slab);
    const int c0 = S.chunk0[s] + (slab - S.slab0[s]) * kFastSlab, cend = S.chunk0[s + 1];
    uint32_t v[kFastSlab];
#pragma unroll
    for (int k = 0; k < kFastSlab; ++k) v[k] = (c0 + k < cend) ? counts[(long long)(c0 + k) * T + b] : 0u;
    uint32_t run = 0;
#pragma unroll
    for (int k = 0; k < kFastSlab; ++k) {
        const uint32_t t = v[k];
        v[k] = run;
        run += t;
    }
#pragma unroll
    for (int k = 0; k < kFastSlab; ++k)
        if (c0 + k < cend) counts[(long long)(c0 + k) * T + b] = v[k];
    slabtot[(long long)slab * T + b] = run;
}
__global__ __launch_bounds__(kWave) void kf_slabscan(SeqTab S, uint32_t *counts, int T, uint32_t *slabtot)
{
    const int b = blockIdx.x * kWave + threadIdx.x;
    if (b < T) slabscan_one(S, counts, T, slabtot, blockIdx.y, b);
}

// slabtot[slab][b] -> exclusive prefix over the slabs of each sequence (in place); exclusive scan over the
// (sequence, tile) pairs -> base[0..pairs]; resets the header and folds the per-chunk error flags into it.
// (small calls -- at most kInlineSlabScan (slab, tile) columns -- run the slab scan here too: one launch less)
constexpr int kInlineSlabScan = 8192;
__global__ __launch_bounds__(kFT) void kf_tilescan(SeqTab S, uint32_t *slabtot, int T, uint32_t *base, uint32_t *seg0,
                                                   FastHeader *hdr, const int32_t *errs, int chunks, uint32_t *counts_inline,
                                                   int slabs_inline, int no_segments)
{
    if (counts_inline) {
        for (int i = threadIdx.x; i < slabs_inline * T; i += kFT) slabscan_one(S, counts_inline, T, slabtot, i / T, i % T);
        __syncthreads(); // the totals are read back below by other threads of this workgroup
    }
    __shared__ uint32_t tot[kMaxPairs];
    __shared__ uint32_t wsum[kFW], wsum2[kFW];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int pairs = S.n_seq * T; // (T = bins per sequence: tiles, or sub-tiles in the direct mode)
    if (tid == 0) { hdr->status = 0; hdr->filtered_tiles = 0u; hdr->mul_bad = 0u; }
    if (tid < kMaxSeq) { hdr->wmask[tid] = 0ull; hdr->unsorted[tid] = 0u; }
    __syncthreads();
    {
        int e = 0;
        for (int c = tid; c < chunks; c += kFT) e |= errs[c];
        if (e & ~ST_MULBAD) atomicOr(&hdr->status, e & ~ST_MULBAD);
        if (e & ST_MULBAD) hdr->mul_bad = 1u;
    }
    // two exclusive scans over the (sequence, bin) pairs: records -> base[], split segments (kSplitSeg records each, at least
    // one per pair; none in the direct mode) -> seg0[]; rounds of kMaxPairs pairs (the direct mode has up to 65 536)
    const uint32_t whole_max = no_segments ? 0xffffffffu : whole_max_of(pairs);
    uint32_t carry = 0, scarry = 0;
    for (int p0 = 0; p0 < pairs; p0 += kMaxPairs) {
        const int np = pairs - p0 < kMaxPairs ? pairs - p0 : kMaxPairs;
        for (int idx = tid; idx < np; idx += kFT) {
            const int gi = p0 + idx, s = gi / T, b = gi - s * T;
            uint32_t run = 0;
            const int sl1 = S.slab0[s + 1];
            for (int sl = S.slab0[s]; sl < sl1; sl += 8) { // 8 independent loads in flight, then the 8 prefix stores
                uint32_t v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = sl + k < sl1 ? slabtot[(long long)(sl + k) * T + b] : 0u;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (sl + k < sl1) slabtot[(long long)(sl + k) * T + b] = run;
                    run += v[k];
                }
            }
            tot[idx] = run;
        }
        __syncthreads();
        const int per = (np + kFT - 1) / kFT;
        const int b0 = tid * per;
        int b1 = b0 + per;
        if (b1 > np) b1 = np;
        uint32_t sum = 0, ssum = 0;
        for (int b = b0; b < b1; ++b) { sum += tot[b]; ssum += split_segments(tot[b], whole_max); }
        uint32_t inc = sum, sinc = ssum;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const uint32_t v = __shfl_up(inc, off), v2 = __shfl_up(sinc, off);
            if (lane >= off) { inc += v; sinc += v2; }
        }
        if (lane == kWave - 1) { wsum[wv] = inc; wsum2[wv] = sinc; }
        __syncthreads();
        uint32_t pre = 0, spre = 0, all = 0, sall = 0;
        for (int k = 0; k < kFW; ++k) {
            if (k < wv) { pre += wsum[k]; spre += wsum2[k]; }
            all += wsum[k]; sall += wsum2[k];
        }
        uint32_t run = carry + pre + inc - sum, srun = scarry + spre + sinc - ssum;
        for (int b = b0; b < b1; ++b) {
            base[p0 + b] = run;
            seg0[p0 + b] = srun;
            run += tot[b];
            srun += split_segments(tot[b], whole_max);
        }
        carry += all;
        scarry += sall;
        __syncthreads(); // tot / wsum are reused by the next round
    }
    if (tid == 0) { base[pairs] = carry; seg0[pairs] = scarry; }
    if (tid == 0) fold_sticky_status(hdr, hdr->status); // all error flags are in since the barrier behind the fold above
}

// ---- 3. stable scatter ---------------------------------------------------------------------------------
// LDS (dynamic): wcnt[16][T] u32 | loff[T + 1] u32 | stage[chunk] u32 | stile[chunk] u16  (78.6 KB at T = 450 with
// 8192-event chunks: two workgroups per CU)
__host__ __device__ inline size_t scatter_lds_bytes(int T, int chunk)
{
    return (size_t)kFW * T * 4 + (size_t)(T + 2) * 4 + (size_t)chunk * 4 + (size_t)chunk * 2 + 16;
}

template <bool HAS_MAP, bool EV = false, bool ORDER = false, bool SIMPLE = false> // ORDER: flag sequences whose window index ever decreases (a
// template parameter on purpose: as a run-time flag the mere presence of the check cost this kernel 35 %, measured)
__global__ __launch_bounds__(kFT) void kf_scatter(FastGeom G, SeqTab S, const uint32_t *counts, const uint32_t *slabtot,
                                                  const uint32_t *base, uint32_t *records, FastHeader *hdr)
{
    extern __shared__ uint32_t lds[];
    const int T = G.T;
    uint32_t *wcnt_all = lds;                  // [16][T]: per-wavefront running counts, then prefixes
    uint32_t *loff = wcnt_all + (size_t)kFW * T; // [T + 1] slot of tile b's first record in the staged chunk; after
                                                 // the staging: global slot of that record MINUS its staged slot
    uint32_t *stage = loff + ((T + 2) & ~1);
    uint16_t *stile = (uint16_t *)(stage + G.chunk_ev);
    __shared__ uint32_t wtot[kFW];
    __shared__ unsigned long long wg_seen;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6; // (NOT readfirstlane: with the wavefront index in an SGPR this kernel ran 60 % longer, measured)
    const int chunk = (int)chunk_of_block(blockIdx.x, gridDim.x);
    const int s = seq_of_chunk(S, chunk);
    uint32_t *wcnt = wcnt_all + (size_t)wv * T;
    for (int b = tid; b < kFW * T; b += kFT) wcnt_all[b] = 0;
    if (tid == 0) wg_seen = 0ull;
    __syncthreads();

    const long long chunk_begin = S.ev0[s] + (long long)(chunk - S.chunk0[s]) * G.chunk_ev;
    const long long wave_begin = chunk_begin + (long long)wv * G.run; // wavefront w owns the w-th run of the chunk
    const long long left = S.ev0[s + 1] - wave_begin;
    const uint32_t nloc = left < (long long)G.run ? (uint32_t)(left < 0 ? 0 : left) : (uint32_t)G.run;
    const long long t0 = S.t0[s];
    // the event in front of this wavefront's run (same sequence), for the window-order check below: requested first, so
    // that it is back first (loads return in order)
    uint2 qprev = make_uint2(0u, 0u);
    const bool has_prev = ORDER && nloc > 0 && wave_begin > S.ev0[s];
    if (ORDER && nloc > 0) qprev = G.data[has_prev ? wave_begin - 1 : wave_begin];
    uint2 q[kMaxBpw];
    if (nloc > 0) { // wave-uniform.  No load under a lane condition (each would wait for its own data: eight serialized round
                    // trips): lanes behind the run's end re-read its last event and are masked by `i < nloc` below
        const uint2 *src = G.data + wave_begin;
#pragma unroll
        for (int j = 0; j < kMaxBpw; ++j) {
            const uint32_t i = (uint32_t)(j * kWave + lane);
            q[j] = src[i < nloc ? i : nloc - 1u];
        }
    }
    // global slot of this chunk's run in every tile (needed after the ranks: issue the loads now)
    const int slab = S.slab0[s] + (chunk - S.chunk0[s]) / kFastSlab;
    // (three loads without a lane condition, summed only where the sum is needed: inside an `if (tid < T)` the compiler
    // waits for them -- and for the event loads in front of them -- right here)
    const int tcl = tid < T ? tid : 0;
    const uint32_t gs_a = base[s * T + tcl], gs_b = slabtot[(long long)slab * T + tcl], gs_c = counts[(long long)chunk * T + tcl];
    // ---- phase A: stream rank of every event inside (wavefront, tile): batches of 64 consecutive events, one
    // returning LDS atomic each -- same-address lanes are served in lane order, and a wavefront's LDS instructions in
    // program order, so the returned count is the number of earlier events of the wavefront's run in the same tile.
    uint32_t where[kMaxBpw], word[kMaxBpw];
    unsigned long long wseen = 0ull;
    // window of the event in front of this wavefront's run (same sequence), for the order check below
    uint32_t carry = 0u;
    if (has_prev) carry = fast_decode<HAS_MAP, EV, SIMPLE>(G, qprev, t0).window;
    bool backwards = false;
#pragma unroll
    for (int j = 0; j < kMaxBpw; ++j) {
        where[j] = 0xffffffffu;
        word[j] = 0u;
        if (j < G.bpw) {
            const uint32_t i = (uint32_t)(j * kWave + lane);
            uint32_t win_j = 0xffffffffu; // lanes behind the run's end: larger than any window
            if (i < nloc) {
                const FastEv o = fast_decode<HAS_MAP, EV, SIMPLE>(G, q[j], t0);
                if (o.tile >= 0) {
                    const uint32_t r = atomicAdd(&wcnt[o.tile], 1u);
                    where[j] = ((uint32_t)o.tile << 16) | r;
                    word[j] = o.word;
                    wseen |= 1ull << o.window;
                    win_j = o.window;
                }
            }
            if (ORDER) { // does the stream ever step back into an earlier window?  (the tile walk relies on window-sorted lists)
                // Usual case: the 64 events of the batch lie in ONE window that is not below the previous batch's last --
                // two lane reads and a ballot.  Otherwise (a window boundary inside the batch, an event that was not
                // encoded, the ragged end of a run) every lane looks at its left neighbour.
                const uint32_t w_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)win_j);
                if (__ballot(win_j != w_first) == 0ull) {
                    if (w_first != 0xffffffffu) { backwards |= w_first < carry; carry = w_first; }
                } else {
                    uint32_t prev = (uint32_t)__shfl_up((int)win_j, 1);
                    if (lane == 0) prev = carry;
                    // lanes without an event (0xffffffff) neither compare nor serve as a neighbour: the ragged end of a run
                    // only has them behind its last event, an error event voids the whole call anyway
                    backwards |= win_j != 0xffffffffu && prev != 0xffffffffu && win_j < prev;
                    const unsigned long long have = __ballot(win_j != 0xffffffffu);
                    if (have) carry = (uint32_t)__builtin_amdgcn_readlane((int)win_j, 63 - __builtin_clzll(have));
                }
            }
        }
    }
    if (ORDER && __ballot(backwards) && lane == 0) atomicOr(&hdr->unsorted[s], 1u);
    __syncthreads();
    // ---- phase B: per tile, exclusive prefix of the 16 wavefront counts; chunk-local offsets of the tiles
    uint32_t mine = 0; // records of tile `tid` in this chunk
    for (int b = tid; b < T; b += kFT) {
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < kFW; ++w) {
            const uint32_t v = wcnt_all[(size_t)w * T + b];
            wcnt_all[(size_t)w * T + b] = run;
            run += v;
        }
        mine = run;
    }
    const uint32_t inc = wave_incl_scan(mine);
    if (lane == kWave - 1) wtot[wv] = inc;
    __syncthreads();
    uint32_t pre = 0, total = 0;
#pragma unroll
    for (int k = 0; k < kFW; ++k) { if (k < wv) pre += wtot[k]; total += wtot[k]; }
    if (tid < T) loff[tid] = pre + inc - mine;
    __syncthreads();
    // ---- phase C: stage the chunk tile-major in LDS, then leave in one linear sweep: consecutive threads write
    // consecutive records of a tile's run (whole lines instead of 64 scattered 4-byte stores)
#pragma unroll
    for (int j = 0; j < kMaxBpw; ++j) {
        if (j < G.bpw && where[j] != 0xffffffffu) {
            const uint32_t b = where[j] >> 16;
            const uint32_t slot = loff[b] + wcnt[b] + (where[j] & 0xffffu);
            stage[slot] = word[j];
            stile[slot] = (uint16_t)b;
        }
    }
    // which windows of the sequence hold events at all ("all(forward)", generate_taf.py:40): OR inside the wavefront,
    // inside the workgroup, and touch the global word only for bits it does not show yet
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)wseen, off), hi = __shfl_xor((unsigned)(wseen >> 32), off);
        wseen |= ((unsigned long long)hi << 32) | lo;
    }
    if (lane == 0 && wseen) atomicOr(&wg_seen, wseen);
    __syncthreads();
    if (tid < T) loff[tid] = (gs_a + gs_b + gs_c) - loff[tid]; // wraps around harmlessly (mod 2^32)
    __syncthreads();
    for (uint32_t qi = tid; qi < total; qi += kFT) records[loff[stile[qi]] + qi] = stage[qi];
    if (tid == 0) {
        const unsigned long long m = wg_seen;
        const unsigned long long have = __hip_atomic_load(&hdr->wmask[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (m & ~have) atomicOr(&hdr->wmask[s], m);
    }
}

// ---- 3'. chunk-major scatter: the partition WITHOUT a histogram pass (round 4) -----------------------------------------
// kf_hist exists only so that kf_scatter knows, before it writes, where every (chunk, bin) run goes in a bin-major array --
// a whole extra pass over the 8-byte events (80 MB and 19 us at 10 M events) plus two scan launches.  Here the scatter
// workgroup sorts its chunk by bin in LDS exactly as before and writes it out AS IT IS, chunk-major: chunk c's records
// occupy rec[first event of c - first event of the call ...) in one linear sweep (whole lines, no per-record address), and
// the chunk leaves one directory row dir[c][bin] = count << 16 | offset of the bin's run inside the chunk.  A bin's list is
// then the concatenation of its runs in chunk order -- still stream order -- and whoever consumes the bin reads its column
// of the directory (a few hundred to a few thousand entries), prefix-sums it in LDS and gathers the runs (col_* below).
// The kernel also does what kf_hist did on the side: the per-call value table + the check that multiplying by 1 / den gives
// the same floats, the data-dependent status (straight into the header: no per-chunk flags to fold), the window masks.
__host__ __device__ inline size_t scatter_cm_lds_bytes(int T, int chunk)
{
    return (size_t)kFW * T * 4 + (size_t)(T + 2) * 4 + (size_t)chunk * 4 + 16;
}

// MAXB: 64-event batches per wavefront the registers hold.  8: 64 VGPRs, two workgroups per CU, chunks up to 8192 events.
// kBigBpw: 128 VGPRs, ONE workgroup per CU, chunks up to 20 480 events (LDS: 80 KB of staging + the counters) -- for large
// calls with tile bins, where a consumer gathers one run per chunk: 10 M events at 1280x720 leave 512 chunks with 43-record
// runs instead of 1536 with 14-record ones.
#ifndef FRLW_SCATTER_AHEAD
#define FRLW_SCATTER_AHEAD 4
#endif
#if defined(FRLW_WALK_PROF) || defined(FRLW_SCAT_PROF) // developer timeline of kf_taf_walk / kf_scatter_cm (tools/enc_lab.cpp prints it): cycles between stamps, summed over workgroups
constexpr int kProfWgs = 131072;
__device__ unsigned long long g_walk_prof[kProfWgs * 9];
#define XPROF(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pd_[i] += t_ - tp_; tp_ = t_; } while (0)
#define XPROF_INIT() unsigned long long tp_ = __builtin_amdgcn_s_memtime(), pd_[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}
#define XPROF_END() do { if (threadIdx.x == 0 && blockIdx.x < kProfWgs) for (int i_ = 0; i_ < 9; ++i_) g_walk_prof[blockIdx.x * 9 + i_] += pd_[i_]; } while (0)
#endif
#ifdef FRLW_WALK_PROF
#define WPROF(i) XPROF(i)
#define WPROF_INIT() XPROF_INIT()
#define WPROF_END() XPROF_END()
#else
#define WPROF(i) do { } while (0)
#define WPROF_INIT() do { } while (0)
#define WPROF_END() do { } while (0)
#endif
#ifdef FRLW_SCAT_PROF
#define SPROF(i) XPROF(i)
#define SPROF_INIT() XPROF_INIT()
#define SPROF_END() XPROF_END()
#define SPROF_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define SPROF(i) do { } while (0)
#define SPROF_INIT() do { } while (0)
#define SPROF_END() do { } while (0)
#define SPROF_DRAIN() do { } while (0)
#endif
template <bool HAS_MAP, bool EV = false, bool SIMPLE = false, int MAXB = kMaxBpw, int SAE = 0>
__global__ __launch_bounds__(kFT) __attribute__((amdgpu_waves_per_eu(MAXB > kMaxBpw ? 4 : 8, MAXB > kMaxBpw ? 4 : 8))) void kf_scatter_cm(FastGeom G, SeqTab S, uint32_t *dir, uint32_t *records, FastHeader *hdr, float *tlut_w,
                                                     uint32_t epoch)
{
    extern __shared__ uint32_t lds[];
    const int T = G.T;
    uint32_t *wcnt_all = lds;                    // [16][T]: per-wavefront running counts, then prefixes
    uint32_t *loff = wcnt_all + (size_t)kFW * T; // [T + 1] slot of bin b's first record in the staged chunk
    uint32_t *stage = loff + ((T + 2) & ~1);
    __shared__ uint32_t wtot[kFW];
    __shared__ unsigned long long wg_seen;
    __shared__ int serr;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    SPROF_INIT();
    const int chunk = (int)chunk_of_block(blockIdx.x, gridDim.x);
    const int s = seq_of_chunk(S, chunk);
    uint32_t *wcnt = wcnt_all + (size_t)wv * T;
    for (int b = tid; b < kFW * T; b += kFT) wcnt_all[b] = 0;
    if (tid == 0) { wg_seen = 0ull; serr = 0; }
    if (blockIdx.x == 0 && epoch != 0u) {
        // The header's per-call words are reset HERE, by the workgroup the dispatcher starts first, instead of by a memset node
        // in front of the kernel.  Every other workgroup writes to the header only at its very end and only after it has seen
        // this call's epoch (published below, behind the reset): the first workgroup is resident before any other one starts,
        // so that wait always ends (and is bounded all the same, below).  epoch == 0: the call is being captured into a graph --
        // a host-made epoch would be baked into the node and every replay after the first would find it published already --
        // so launch_fast_cm put a reset kernel in front instead and nobody resets or waits here.
        uint32_t *h32 = (uint32_t *)hdr;
        for (int i = tid; i < (int)(offsetof(FastHeader, epoch) / 4); i += kFT) h32[i] = 0u;
        if (tid == 0) { // an EARLIER call's stall verdict goes; this call's own (a workgroup that gave up before we started) stays
            uint32_t *sw = (uint32_t *)((char *)hdr + kStallOffset);
            const uint32_t was = __hip_atomic_load(sw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (was != 0u && was != epoch) atomicCAS(sw, was, 0u);
        }
        __threadfence();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&hdr->epoch, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }

    const long long chunk_begin = S.ev0[s] + (long long)(chunk - S.chunk0[s]) * G.chunk_ev;
    const long long wave_begin = chunk_begin + (long long)wv * G.run; // wavefront w owns the w-th run of the chunk
    const long long left = S.ev0[s + 1] - wave_begin;
    const uint32_t nloc = left < (long long)G.run ? (uint32_t)(left < 0 ? 0 : left) : (uint32_t)G.run;
    const long long t0 = S.t0[s];
    // The chunk's events: kAhead batches per wavefront are requested here, batch j + kAhead when batch j is ranked (phase A).
    // All MAXB at once (the form until round 5) fills the CU's memory queue -- 160 KB per CU, every CU of the part in the same
    // burst -- and the wavefronts then stand at the ISSUE of their loads until HBM has served the queue: 8 of a workgroup's 26 us
    // in front of the first decoded event (developer timeline, -DFRLW_SCAT_PROF).
    constexpr int kAhead = MAXB > FRLW_SCATTER_AHEAD ? FRLW_SCATTER_AHEAD : MAXB;
    uint2 q[MAXB];
    const uint2 *src = G.data + wave_begin;
    if (nloc > 0) { // wave-uniform; no load under a lane condition: lanes behind the run's end re-read its last event
#pragma unroll
        for (int j = 0; j < kAhead; ++j) {
            const uint32_t i = (uint32_t)(j * kWave + lane);
            q[j] = src[i < nloc ? i : nloc - 1u];
        }
    }
    int pre_err = 0;
    // the per-call value table, spread over the grid while the event loads fly (generate_taf.py:215,:26 /
    // generate_eventvolume.py:141,:23): tlut[r] and the exhaustive check "float(r * (1 / den)) == float(r / den) for every r"
    if (tlut_w) {
        const double den = EV ? (double)G.win : (double)G.win + 1e-8, rcp = G.rcp;
        bool bad = false;
        for (long long r = (long long)blockIdx.x * kFT + tid; r <= (long long)G.win; r += (long long)gridDim.x * kFT) {
            const float exact = (float)((double)r / den);
            bad |= (float)((double)r * rcp) != exact;
            tlut_w[r] = EV ? exact : exact - 1.0f;
        }
        // kept in a register until phase A's flags are OR-ed in BEHIND the barrier below: `serr` is zeroed by thread 0 in front
        // of that barrier, and an atomicOr from another wavefront here could land before the zero and be lost
        if (bad) pre_err = ST_MULBAD;
    }
    // The barrier that publishes the zeroed counters must NOT wait for the event loads: __syncthreads() drains vmcnt, and the
    // burst of a whole chunk (160 KB per CU, every CU of the part at once: HBM-bound, 8 of a workgroup's 26 us) would have to
    // land before the first event is decoded.  A raw s_barrier behind the LDS writes only: the compiler's counted waits
    // (loads return in order) then let batch j be ranked while batches j + 1 ... are still on their way.
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    SPROF(0);
    SPROF(1);
    // ---- phase A: stream rank of every event inside (wavefront, bin), one returning LDS atomic each (lane order = stream order)
    uint32_t where[MAXB], word[MAXB];
    unsigned long long wseen = 0ull;
    int err = pre_err;
#pragma unroll
    for (int j = 0; j < MAXB; ++j) {
        where[j] = 0xffffffffu;
        word[j] = 0u;
        if (j + kAhead < MAXB) {
            if (nloc > 0 && j + kAhead < G.bpw) { // wave-uniform
                const uint32_t i2 = (uint32_t)((j + kAhead) * kWave + lane);
                q[j + kAhead] = src[i2 < nloc ? i2 : nloc - 1u];
            }
            asm volatile("" ::: "memory"); // (the request stays HERE: hoisted to the top it is the burst again)
        }
        if (j < G.bpw) {
            const uint32_t i = (uint32_t)(j * kWave + lane);
            if (i < nloc) {
                const FastEv o = fast_decode<HAS_MAP, EV, SIMPLE, SAE>(G, q[j], t0);
                err |= o.err;
                if (o.tile >= 0) {
                    const uint32_t r = atomicAdd(&wcnt[o.tile], 1u);
                    where[j] = ((uint32_t)o.tile << 16) | r;
                    // SAE: the event's position in its sequence + 1 (below 2^20: the host checks; a record is never 0) in place of the time field
                    word[j] = SAE ? ((uint32_t)(wave_begin - S.ev0[s] + (long long)i + 1) << kCellBits) | (o.word & (uint32_t)(kCells - 1)) : o.word;
                    wseen |= 1ull << o.window;
                }
            }
        }
    }
    if (err) atomicOr(&serr, err);
    SPROF(2);
    __syncthreads();
    SPROF(3);
    // ---- phase B: per bin (thread = bin: T <= kFT), exclusive prefix of the 16 wavefront counts; chunk-local offsets of the bins.
    // The prefixes stay in registers until the bin's slot in the staged chunk is known and go back to LDS ONCE, slot included:
    // phase C then reads one table per record
    uint32_t mine = 0; // records of bin `tid` in this chunk
    uint32_t pv[kFW];
    {
        const int b = tid < T ? tid : 0;
#pragma unroll
        for (int w = 0; w < kFW; ++w) {
            pv[w] = mine;
            mine += wcnt_all[(size_t)w * T + b];
        }
        if (tid >= T) mine = 0u;
    }
    const uint32_t inc = wave_incl_scan(mine);
    if (lane == kWave - 1) wtot[wv] = inc;
    __syncthreads();
    uint32_t pre = 0, total = 0;
#pragma unroll
    for (int k = 0; k < kFW; ++k) { if (k < wv) pre += wtot[k]; total += wtot[k]; }
    const uint32_t my_off = pre + inc - mine;
    if (tid < T) {
#pragma unroll
        for (int w = 0; w < kFW; ++w) wcnt_all[(size_t)w * T + tid] = pv[w] + my_off;
        // the directory is BIN-major, dir[bin][chunk] (every chunk writes its entry of all T bins: T scattered 4-byte stores
        // per workgroup): a consumer reads its bin's column as ONE contiguous stretch -- chunk-major rows made every consumer's
        // first step 512 loads from 512 lines (5 of the 30 us of a kf_split_whole workgroup)
        dir[(long long)tid * (long long)gridDim.x + chunk] = (mine << 16) | my_off;
    }
    __syncthreads();
    SPROF(4);
    // ---- phase C: stage the chunk bin-major in LDS
#pragma unroll
    for (int j = 0; j < MAXB; ++j) {
        if (j < G.bpw && where[j] != 0xffffffffu) {
            const uint32_t b = where[j] >> 16;
            stage[wcnt[b] + (where[j] & 0xffffu)] = word[j];
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)wseen, off), hi = __shfl_xor((unsigned)(wseen >> 32), off);
        wseen |= ((unsigned long long)hi << 32) | lo;
    }
    if (lane == 0 && wseen) atomicOr(&wg_seen, wseen);
    __syncthreads();
    SPROF(5);
    // ---- phase D: the staged chunk leaves as it is, one linear sweep into the chunk's own stretch of rec[]
    uint32_t *dst = records + (chunk_begin - S.ev0[0]);
    for (uint32_t qi = tid; qi < total; qi += kFT) dst[qi] = stage[qi];
    SPROF(6);
    SPROF_DRAIN();
    SPROF(7);
    SPROF_END();
    if (tid == 0) {
        if (epoch != 0u) {
            // bounded: ~2^22 polls of >= 128 cycles (a fraction of a second; the wait is normally over before it starts).  A part
            // or a scheduler that does not start workgroup 0 first ends the call with ST_STALL (FRLW_ERR_HIP) instead of a hang.
            uint32_t polls = 0u;
            while (__hip_atomic_load(&hdr->epoch, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                if (++polls > (1u << 22)) break;
                __builtin_amdgcn_s_sleep(2);
            }
            if (polls > (1u << 22)) { // (not hdr->status: a workgroup 0 that starts later would zero it; see kStallOffset)
                atomicExch((uint32_t *)((char *)hdr + kStallOffset), epoch);
                fold_sticky_status(hdr, ST_STALL);
                return;
            }
        }
        const unsigned long long m = wg_seen;
        const unsigned long long have = __hip_atomic_load(&hdr->wmask[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (m & ~have) atomicOr(&hdr->wmask[s], m);
        const int e = serr;
        if (e & ~ST_MULBAD) { atomicOr(&hdr->status, e & ~ST_MULBAD); fold_sticky_status(hdr, e & ~ST_MULBAD); }
        if (e & ST_MULBAD) hdr->mul_bad = 1u; // (every writer stores the same value)
    }
}

// ---- 4. per-tile split by sub-tile, 5. one wavefront per sub-tile -----------------------------------------
struct TileP {
    int H, W, twl, thl, tiles_x, T, K, n_windows, wb, flip;
    uint32_t win;
    const uint32_t *rec;   // tile-major records (scatter output)
    uint32_t *rec2;        // the same records, inside every tile sub-tile-major (split output)
    const uint32_t *base;  // [pairs + 1]
    uint32_t *sub;         // [pairs * 16 + 1] first record of every sub-tile in rec2
    uint32_t *sub_end;     // chunk-major partition: [pairs * 16] end of every sub-tile's list (lists are placed through a cursor,
                           // not back to back in pair order); NULL otherwise: a list ends where the next one starts
    const uint32_t *seg0;  // [pairs + 1] first split segment of every (sequence, tile) pair
    uint32_t *segcnt;      // [segments][16] records of every sub-tile in a segment, then their offsets inside the sub-tile
    int pairs;
    int direct;            // 1: rec2 / sub are the scatter's own output (sub-tile bins): no split kernel has run
    int skip_whole;        // 1: tiles up to the whole-tile limit are NOT re-sorted (a tile-walk kernel splits them in LDS) ...
    int tile_walk;         // ... TAF: unless their sequence is not window-sorted (hdr->unsorted)
    int first_block;       // kf_split_whole: block b does the work of block b + first_block
    int seg_grid;          // segment workgroups launched (they stride over the segments: most calls have none)
    uint32_t tile_max;     // tiles with more records than this go through the segment split
    const float *tlut;
    const uint32_t *leaky_thr;
    FastHeader *hdr;
    double rcp;      // FastGeom::rcp
    float *state;    // (B, H, W, 2, K)
    float *view_f32; // (B, 2K, H, W) or NULL
    uint8_t *out_u8; // (B, K, 2, H, W) or NULL
    // window starts, written by kf_split_whole<true> for the tiles it splits (TAF only; NULL otherwise): wst[sg * (n_windows + 1)
    // + w] = list position of the first record of window w in sub-tile list sg, 0xffffffff: the window has none -- what the walk's
    // own scan finds; wst_flag[pair] = 0: no table (the walk scans its list), 1: table valid, 2: a list of the tile is not
    // window-sorted (the walk filters the whole list per window)
    uint32_t *wst, *wst_flag;
};

// ---- the consumer side of the chunk-major partition: a bin's column of the directory ----------------------------------
constexpr int kColMax = 4096; // chunks per sequence a consumer keeps in LDS (32 KB); longer sequences take the histogram path

struct CmP { // kernel argument of the chunk-major consumers
    const uint32_t *dir; // [TB][n_chunks]: count << 16 | offset of the bin's run inside the chunk's records
    int n_chunks;        // chunks of the call (all sequences)
    const uint32_t *rec; // chunk-major records (kf_scatter_cm)
    int TB;              // bins per sequence
    int chunk_ev;        // events per chunk = records a chunk's stretch of rec[] can hold
    uint32_t *hot_start; // [pairs] skewed tiles: where the tile's 16 lists start in rec2[]
    uint32_t *hot_seg0;  // [pairs] ... and the id of the tile's first split segment
    uint32_t *segdesc;   // [max_segs] pair of every split segment
    int max_segs;
};

// Column of bin `b` of sequence `s`: L[c] = records of the bin in the sequence's chunks before c (L[C] = all of them, the
// return value), D[c] = index in rec[] of the bin's first record of chunk c MINUS L[c] -- list position i (stream order) is
// rec[D[c] + i] for the chunk c with L[c] <= i < L[c + 1].  NT threads, workgroup barriers inside; wsum: NT / 64 + 1 words.
template <int NT>
__device__ __forceinline__ uint32_t col_load(const CmP &m, const SeqTab &S, int s, int b, uint32_t *L, uint32_t *D, uint32_t *wsum)
{
    constexpr int NWV = NT / kWave;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int c0 = S.chunk0[s], C = S.chunk0[s + 1] - c0; // >= 1: an empty sequence keeps one (empty) chunk
    const uint32_t out0 = (uint32_t)(S.ev0[s] - S.ev0[0]);
    uint32_t carry = 0;
    for (int cb = 0; cb < C; cb += NT) { // passes of NT chunks, one per thread (workgroup-uniform trip count: usually one)
        const int c = cb + tid;
        // (clamped index, masked value: no load sits under a lane condition)
        const uint32_t v = m.dir[(long long)b * m.n_chunks + (c0 + (c < C ? c : C - 1))];
        const uint32_t e = c < C ? v : 0u, cnt = e >> 16;
        const uint32_t inc = wave_incl_scan(cnt);
        if (lane == kWave - 1) wsum[wv] = inc;
        __syncthreads();
        uint32_t run = carry + inc - cnt, total = 0;
#pragma unroll
        for (int k = 0; k < NWV; ++k) { if (k < wv) run += wsum[k]; total += wsum[k]; }
        if (c < C) {
            L[c] = run;
            D[c] = out0 + (uint32_t)c * (uint32_t)m.chunk_ev + (e & 0xffffu) - run;
        }
        carry += total;
        __syncthreads(); // wsum is reused by the next pass
    }
    if (tid == 0) L[C] = carry;
    __syncthreads();
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)carry); // (uniform for the compiler too: scalar branches downstream)
}

// idx[(i - lo) >> 4] = the chunk that holds list position i, for every i = lo (mod 16) ... in [lo, hi) -- lo a multiple of 16:
// a lane then finds its own position's chunk with the short walk of col_addr instead of a bisection.  Call after col_load.
template <int NT>
__device__ __forceinline__ void col_index(const uint32_t *L, int C, uint32_t lo, uint32_t hi, uint16_t *idx)
{
    for (int c = threadIdx.x; c < C; c += NT) {
        const uint32_t a = L[c] > lo ? L[c] : lo, z = L[c + 1] < hi ? L[c + 1] : hi;
        for (uint32_t i = (a + 15u) & ~15u; i < z; i += 16u) idx[(i - lo) >> 4] = (uint16_t)c;
    }
}

__device__ __forceinline__ uint32_t col_addr(const uint32_t *L, const uint32_t *D, uint32_t c, uint32_t i); // below

// The same for one wavefront's 64 consecutive positions lo + r0 .. lo + r0 + 63 of a range [lo, lo + nr) indexed by idx (this
// lane: lo + ric): when they all lie inside ONE run -- the rule for the long runs of a skewed tile -- the run is found once,
// with scalar compares, and a lane only adds.
__device__ __forceinline__ uint32_t col_addr_wave(const uint32_t *L, const uint32_t *D, const uint16_t *idx, uint32_t lo, uint32_t r0,
                                                  uint32_t ric, uint32_t nr)
{
    if (r0 >= nr) return D[0] + L[0]; // (wave-uniform: nothing of this wavefront's batch is inside the range; any address that exists)
    const uint32_t i0 = lo + r0, last = lo + (r0 + 63u < nr ? r0 + 63u : nr - 1u);
    uint32_t cs = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx[r0 >> 4]);
    uint32_t lnext = (uint32_t)__builtin_amdgcn_readfirstlane((int)L[cs + 1]);
    while (lnext <= i0) { ++cs; lnext = (uint32_t)__builtin_amdgcn_readfirstlane((int)L[cs + 1]); }
    if (lnext > last) return (uint32_t)__builtin_amdgcn_readfirstlane((int)D[cs]) + lo + ric;
    return col_addr(L, D, idx[ric >> 4], lo + ric);
}

__device__ __forceinline__ uint32_t col_addr(const uint32_t *L, const uint32_t *D, uint32_t c, uint32_t i) // c: a chunk at or in front of i's
{
    // (i < L[C]: ends inside the column; empty runs are stepped over).  Two steps without a branch -- 16 positions rarely span
    // more runs -- then the loop for whoever is still short
    c += L[c + 1] <= i ? 1u : 0u;
    c += L[c + 1] <= i ? 1u : 0u;
#ifndef COLX
    if (__builtin_expect(__ballot(L[c + 1] <= i) != 0ull, 0))
        while (L[c + 1] <= i) ++c;
#endif
    return D[c] + i;
}

constexpr int kMaxK = 8;

// One FIFO step of one cell, generate_taf.py:27,35-49.  Cell without events: every slot - 1; otherwise shift down
// (slot k + 1, - 1) and the mean enters at K - 1.  `has` false (window empty in the whole sequence, :40-41): unchanged.
__device__ __forceinline__ float fifo_mean(uint32_t n, float sum) { return sum / ((float)n + 1e-8f); } // generate_taf.py:27

__device__ __forceinline__ void fifo_step(float (&st)[kMaxK], int K, bool has, uint32_t n, float mean)
{
    const bool hit = n != 0u;
#pragma unroll
    for (int k = 0; k < kMaxK; ++k) {
        const float nxt = k + 1 < kMaxK ? st[k + 1] : 0.0f;
        const float v = (hit ? nxt : st[k]) - 1.0f;
        const float nv = (hit && k == K - 1) ? mean : v;
        st[k] = has ? nv : st[k];
    }
}

// The same step for K = 8 with TWO lanes per cell: the even lane holds slots 0..3, the odd lane slots 4..7 of the row
// (the whole workgroup works in phase 2, and a lane moves 16 bytes of the row).  Slot 3 takes over slot 4 from the
// partner lane through a DPP row shift; the float operations per slot are those of fifo_step.
__device__ __forceinline__ void fifo_step_half(float (&st)[4], bool upper, bool has, uint32_t n, float mean)
{
    const bool hit = n != 0u;
    const float up = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, st[0]), 0x101, 0xf, 0xf, false)); // row_shl:1: lane l reads lane l + 1
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float nxt = k < 3 ? st[k + 1] : (upper ? 0.0f : up);
        const float v = (hit ? nxt : st[k]) - 1.0f;
        const float nv = (hit && upper && k == 3) ? mean : v;
        st[k] = has ? nv : st[k];
    }
}

#define LDS_FENCE() asm volatile("" ::: "memory")
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory") // workgroup barrier that publishes LDS only: loads / stores in flight stay in flight

__device__ __forceinline__ int pair_of_segment(const uint32_t *seg0, int pairs, uint32_t seg)
{
    int lo = 0, hi = pairs; // largest g with seg0[g] <= seg
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (seg0[mid] <= seg) lo = mid; else hi = mid;
    }
    return lo;
}

// Count the records of rec[beg, end) per (wavefront, sub-tile) into wtot: eight loads in flight per thread (indices
// clamped, values masked -- a load under a lane condition, or one load per loop iteration, is one exposed round trip each:
// a 22 000-record tile took 22 of them).
__device__ __forceinline__ void count_subtiles(const uint32_t *rec, uint32_t beg, uint32_t end, uint32_t (*wtot)[kFW])
{
    const int tid = threadIdx.x, wv = tid >> 6;
    for (uint32_t c0 = beg; c0 < end; c0 += 8 * kFT) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t i = c0 + (uint32_t)(u * kFT + tid);
            v[u] = rec[i < end ? i : end - 1u];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (c0 + (uint32_t)(u * kFT + tid) < end) atomicAdd(&wtot[wv][(v[u] & (kCells - 1)) >> 8], 1u);
    }
}

__device__ __forceinline__ void split_count_segment(const TileP &q, uint32_t seg, uint32_t (*wtot)[kFW])
{
    const int tid = threadIdx.x, wv = tid >> 6;
    if (seg >= q.seg0[q.pairs]) return;
    const int g = pair_of_segment(q.seg0, q.pairs, seg);
    const uint32_t beg = q.base[g] + (seg - q.seg0[g]) * (uint32_t)kSplitSeg;
    const uint32_t end = q.base[g + 1] - beg < (uint32_t)kSplitSeg ? q.base[g + 1] : beg + kSplitSeg;
    if (tid < kFW * kFW) (&wtot[0][0])[tid] = 0u;
    __syncthreads();
    count_subtiles(q.rec, beg, end, wtot);
    __syncthreads();
    if (tid < kFW) {
        uint32_t t = 0;
#pragma unroll
        for (int w = 0; w < kFW; ++w) t += wtot[w][tid];
        q.segcnt[(long long)seg * kFW + tid] = t;
    }
}

// 4a. Tiles of ordinary size (at most kSplitWhole records): ONE workgroup per (sequence, tile) reorders the tile's
// records sub-tile-major, STABLY, in ONE pass over the list: every thread loads its (up to) 32 records at once -- all
// loads of the tile in flight together, the list is read once and stays in registers -- and takes one returning LDS
// atomic per record on a (sub-tile, batch) counter, batch = the 64 records of one wave-instruction: lanes of one
// instruction are served in lane order and the batches are numbered in stream order, so the ticket plus the prefix of the
// sub-tile's earlier batches is the record's stable slot.  (Until round 3 this kernel counted the sub-tiles in a first
// pass and then re-read the list in chunks of 8192: six dependent trips to memory per workgroup where this has one.)
constexpr int kWholeChunks = FRLW_WHOLE_SEGS;
constexpr int kWholeBatches = kWholeChunks * kSplitSeg / kWave; // 512 batches of 64 records
constexpr int kWholeRow = kWholeBatches + 1;                    // row stride of scnt: the 16 counters of one batch in 16 banks

// CM (chunk-major partition): the tile's list is not contiguous -- it is gathered from the tile's column of the directory
// (col_*), the 16 sub-tile lists go wherever the header's cursor says, and a skewed tile only books its space and its split
// segments here (kf_segcount_cm / kf_split_place<true> do the work: the segments are not known before this kernel runs).
template <bool CM>
__global__ __launch_bounds__(kFT) __attribute__((amdgpu_waves_per_eu(8, 8))) void kf_split_whole(TileP q, CmP cm, SeqTab S) // (64 VGPRs: two workgroups per CU)
{
    constexpr int RPT = kSplitSeg / kFT; // 8 records per thread and chunk of 8192
    // scnt [sub-tile][batch] tickets, then exclusive prefixes | stage: one chunk of records, sub-tile-major.  CM, while the
    // list is gathered (before either is used): the column L | D | the position index
    // CM: the column lies over the STAGING area (+ a tail of its own), not over the counters: the counters are zeroed in front of
    // the gather and the tickets are taken as the gathered records arrive -- no barrier, no drained load queue in between
    constexpr int kColWords = 2 * kColMax + 1 + kSplitWhole / 32 + 1;
    constexpr int kPoolTail = CM && kColWords > kSplitSeg ? kColWords - kSplitSeg : 0;
    __shared__ __attribute__((aligned(16))) uint32_t pool[kFW * kWholeRow + kSplitSeg + kPoolTail];
    uint32_t *scnt = pool, *stage = pool + kFW * kWholeRow;
    __shared__ uint32_t wtot[kFW][kFW];        // segment counting (4b): [wavefront][sub-tile]
    __shared__ uint32_t vtot[kFW];             // records of every sub-tile
    __shared__ uint32_t vbeg[kFW][kFW];        // [wavefront]: every wavefront's own copy of the sub-tile starts
    __shared__ uint32_t cE[kFW][kFW], cD[kFW][kFW]; // [wavefront]: per chunk, see step 4
    __shared__ uint32_t s_start, s_first;
    __shared__ int s_unsorted;
    __shared__ int s_fw[kWholeChunks][kFW], s_lw[kWholeChunks][kFW]; // (CM, TAF) first / last window of every sub-tile list inside every chunk, -1: no record
    __shared__ uint32_t s_wst[CM ? kFW * (FRLW_MAX_WINDOWS + 1) : 1];  // (CM, TAF) the tile's rows of TileP::wst while they are made
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // (CM: the status is REQUESTED here and tested behind the directory column -- a test in front of it made the column's load
    // wait for the header's round trip; the directory lies at addresses the plan fixes, reading it is safe whatever the status)
    const int32_t status0 = q.hdr->status;
    if (!CM && status0 != 0) return;
    const int blk = (int)blockIdx.x + q.first_block;
    if (!CM && blk >= q.pairs) { // the blocks behind the tiles: one segment of a skewed tile each (4b, counting)
        if (q.first_block && blk == q.pairs && tid == 0) q.sub[(long long)q.pairs * kFW] = q.base[q.pairs]; // (the last tile's block is not there to do it)
        const uint32_t nseg = q.seg0[q.pairs];
        for (uint32_t seg = (uint32_t)(blk - q.pairs); seg < nseg; seg += (uint32_t)q.seg_grid) {
            split_count_segment(q, seg, wtot);
            __syncthreads(); // wtot is reused
        }
        return;
    }
    const int g = blk;
    uint32_t beg, n;
    uint32_t m[kWholeChunks][RPT];
    if (CM) {
        uint32_t *colL = stage, *colD = stage + kColMax + 1;
        uint16_t *idx = (uint16_t *)(stage + 2 * kColMax + 1);
        const int s = g / q.T, C = S.chunk0[s + 1] - S.chunk0[s];
        for (int i = tid; i < kFW * kWholeRow; i += kFT) scnt[i] = 0u; // (published by the barriers of col_load)
        // (readfirstlane: the workgroup-uniform values that come out of LDS are uniform for the COMPILER too -- as vector values
        // they turned every "is this chunk of the list there at all" test below into divergent control flow, and 19 of the 32
        // records were spilled)
        n = col_load<kFT>(cm, S, s, g - s * q.T, colL, colD, &wtot[0][0]);
        if (status0 != 0) return;
        if (n == 0u) {
            if (tid < kFW) { q.sub[(long long)g * kFW + tid] = 0u; q.sub_end[(long long)g * kFW + tid] = 0u; }
            if (q.wst && tid == 0) q.wst_flag[g] = 0u; // (empty lists: the walk's scan finds nothing to read)
            return;
        }
        const bool hot = n > q.tile_max;
        if (tid == 0) {
            s_start = atomicAdd(&q.hdr->rec_cursor, n); // the tile's 16 lists: n records of rec2[] from here
            if (hot) s_first = atomicAdd(&q.hdr->seg_cursor, (n + kSplitSeg - 1) / kSplitSeg);
            if (q.wst && hot) q.wst_flag[g] = 0u; // (the segment kernels place this tile: no table)
            s_unsorted = 0;
        }
        if (q.wst && !hot) { // the tile's rows of the window table start at "no record" (published by the barriers below)
            if (tid < kWholeChunks * kFW) { (&s_fw[0][0])[tid] = -1; (&s_lw[0][0])[tid] = -1; }
            for (int i = tid; i < kFW * (q.n_windows + 1); i += kFT) s_wst[i] = 0xffffffffu;
        }
        if (hot) { // a skewed tile (or a call with few tiles): cut into segments of 8192 list positions, one workgroup each
            __syncthreads();
            const uint32_t nseg = (n + kSplitSeg - 1) / kSplitSeg, first = s_first;
            if (tid == 0) { cm.hot_start[g] = s_start; cm.hot_seg0[g] = first; }
            for (uint32_t k = tid; k < nseg; k += kFT)
                if (first + k < (uint32_t)cm.max_segs) cm.segdesc[first + k] = (uint32_t)g;
            return;
        }
        col_index<kFT>(colL, C, 0u, n, idx);
        __syncthreads();
        const uint32_t wvs = (uint32_t)__builtin_amdgcn_readfirstlane(wv); // (the wavefront index as a scalar: uniform branches below)
        // 1. the whole list into registers: list position -> chunk through the index + a walk over at most a few run boundaries
        // (two sweeps: every index first -- LDS work only, nothing in flight -- then the loads through one buffer descriptor with
        // 32-bit offsets: with 64-bit addresses next to the 32 records the compiler spilled 19 of them, each spill waiting for
        // its load)
        {
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)cm.rec, 0, 0xffffffffu, 0x00020000);
#pragma unroll
            for (int c = 0; c < kWholeChunks; ++c) {
#pragma unroll
                for (int u = 0; u < RPT; ++u) {
                    const uint32_t i = (uint32_t)(c * kSplitSeg + u * kFT + tid), ic = i < n ? i : n - 1u;
                    // the wavefront's 64 positions i0 .. i0 + 63: mostly inside ONE run when the runs are long (64 sequences of
                    // 1 M events: 226 records per run) -- then the run is found once, with scalar compares, and a lane only adds
                    const uint32_t i0 = (uint32_t)(c * kSplitSeg + u * kFT) + wvs * kWave;
                    uint32_t off = 0u;
                    if (i0 < n) { // wave-uniform
                        const uint32_t last = i0 + 63u < n ? i0 + 63u : n - 1u;
                        uint32_t cs = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx[i0 >> 4]);
                        uint32_t lnext = (uint32_t)__builtin_amdgcn_readfirstlane((int)colL[cs + 1]);
                        while (lnext <= i0) { ++cs; lnext = (uint32_t)__builtin_amdgcn_readfirstlane((int)colL[cs + 1]); }
                        if (lnext > last) off = (uint32_t)__builtin_amdgcn_readfirstlane((int)colD[cs]) + ic;
                        else off = col_addr(colL, colD, idx[ic >> 4], ic);
                    }
                    m[c][u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(off << 2), 0, 0);
                }
            }
        }
        beg = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_start);
        // (the column is dead once the last address is out; its space is written again in step 4, three barriers from here)
    } else {
        beg = q.base[g];
        const uint32_t end = q.base[g + 1];
        if (g == q.pairs - 1 && tid == 0) q.sub[(long long)q.pairs * kFW] = end; // end of the last sub-tile's list
        if (end - beg > whole_max_of(q.pairs) || q.skip_whole) return; // a skewed tile (or a call with few tiles): left to the segment kernels below
        if (q.tile_walk && q.hdr->unsorted[g / q.T] == 0u) return;     // split in LDS by kf_taf_tile
        n = end - beg;
        if (n == 0u) {
            if (tid < kFW) q.sub[(long long)g * kFW + tid] = beg;
            return;
        }
        // 1. the whole list into registers (indices clamped: no load sits under a lane condition)
#pragma unroll
        for (int c = 0; c < kWholeChunks; ++c)
#pragma unroll
            for (int u = 0; u < RPT; ++u) {
                const uint32_t i = (uint32_t)(c * kSplitSeg + u * kFT + tid);
                m[c][u] = q.rec[beg + (i < n ? i : n - 1u)];
            }
    }
    if (!CM) {
        for (int i = tid; i < kFW * kWholeRow; i += kFT) scnt[i] = 0u;
        __syncthreads();
    }
    // 2. tickets: batch (c, u, wv) of the stream, counter [sub-tile][batch]; packed four to a register (a ticket is < 64)
    uint32_t rk[kWholeChunks][RPT / 4];
#pragma unroll
    for (int c = 0; c < kWholeChunks; ++c) {
#pragma unroll
        for (int u4 = 0; u4 < RPT / 4; ++u4) rk[c][u4] = 0u;
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            const uint32_t i0 = (uint32_t)(c * kSplitSeg + u * kFT + wv * kWave); // first record of the batch: wave-uniform
            if (i0 < n) {
                const bool valid = i0 + (uint32_t)lane < n;
                const uint32_t b = valid ? (m[c][u] & (kCells - 1)) >> 8 : (uint32_t)(lane & 15); // (lanes behind the end add 0, spread over the counters)
                const uint32_t t = atomicAdd(&scnt[b * kWholeRow + (c * RPT + u) * kFW + wv], valid ? 1u : 0u);
                rk[c][u >> 2] |= t << (8 * (u & 3));
            }
        }
    }
    __syncthreads();
    // 3. wavefront b: exclusive prefix of sub-tile b's batch counts in stream order (eight scans of 64 batches)
    {
        uint32_t carry = 0;
#pragma unroll
        for (int k = 0; k < kWholeBatches / kWave; ++k) {
            if ((uint32_t)(k * kWave * kWave) < n) { // (batches behind the end of the list hold zeros)
                const uint32_t v = scnt[wv * kWholeRow + k * kWave + lane];
                const uint32_t inc = wave_incl_scan(v);
                scnt[wv * kWholeRow + k * kWave + lane] = carry + inc - v;
                carry += __shfl(inc, kWave - 1);
            }
        }
        if (lane == 0) vtot[wv] = carry;
    }
    __syncthreads();
    {
        // every wavefront: where the 16 lists start (its own copy: no further barrier)
        const uint32_t t = vtot[lane & 15];
        uint32_t inc = t;
#pragma unroll
        for (int o2 = 1; o2 < kFW; o2 <<= 1) {
            const uint32_t u = __shfl_up(inc, o2);
            if ((lane & 15) >= o2) inc += u;
        }
        if (lane < kFW) {
            vbeg[wv][lane] = beg + inc - t;
            if (wv == 0) {
                q.sub[(long long)g * kFW + lane] = beg + inc - t;
                if (CM) q.sub_end[(long long)g * kFW + lane] = beg + inc;
            }
        }
        LDS_FENCE();
    }
    // 4. records to their slots, chunk by chunk THROUGH LDS: the 8192 records of a chunk are laid out sub-tile-major in the
    // staging area (slot = the chunk's records of lower sub-tiles + the record's rank inside the chunk), then leave in one
    // linear sweep -- consecutive threads write consecutive records of a sub-tile's run (whole lines; the direct form wrote
    // 64 records of one instruction to 16 lists, ~16 bytes per line touched: 61 MB of write traffic for 40 MB of records).
    // Per chunk and sub-tile b (every wavefront keeps its own copy, no barrier for the tables):
    //   cE[b] = (records of sub-tiles < b in the chunk) - (prefix of b at the chunk's first batch)   -> slot = cE[b] + prefix + ticket
    //   cD[b] = (start of b's list) + (prefix of b at the chunk's first batch) - (records of sub-tiles < b)  -> address = cD[b] + slot
    const bool wtab = CM && q.wst != nullptr; // (kernel-uniform)
#pragma unroll
    for (int c = 0; c < kWholeChunks; ++c) {
        if ((uint32_t)(c * kSplitSeg) >= n) break; // workgroup-uniform
        const uint32_t nch = n - (uint32_t)(c * kSplitSeg) < (uint32_t)kSplitSeg ? n - (uint32_t)(c * kSplitSeg) : (uint32_t)kSplitSeg;
        {
            const int b = lane & 15;
            const uint32_t p0 = scnt[b * kWholeRow + c * (kSplitSeg / kWave)];
            const uint32_t p1 = (uint32_t)((c + 1) * kSplitSeg) < n ? scnt[b * kWholeRow + (c + 1) * (kSplitSeg / kWave)] : vtot[b];
            const uint32_t cnt = p1 - p0;
            uint32_t inc = cnt;
#pragma unroll
            for (int o2 = 1; o2 < kFW; o2 <<= 1) {
                const uint32_t u = __shfl_up(inc, o2);
                if (b >= o2) inc += u;
            }
            if (lane < kFW) {
                cE[wv][lane] = (inc - cnt) - p0;
                cD[wv][lane] = vbeg[wv][lane] + p0 - (inc - cnt);
            }
            LDS_FENCE();
        }
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            const uint32_t i = (uint32_t)(u * kFT + tid);
            if (i < nch) {
                const uint32_t b = (m[c][u] & (kCells - 1)) >> 8;
                const uint32_t t = (rk[c][u >> 2] >> (8 * (u & 3))) & 255u;
                stage[cE[wv][b] + scnt[b * kWholeRow + (c * RPT + u) * kFW + wv] + t] = m[c][u];
            }
        }
        __syncthreads();
        // (CM, TAF) the window starts kf_taf_walk needs, so that it does not have to scan its list for them.  The staged chunk is
        // sub-tile-major and stable: neighbours of one sub-tile are neighbours of that sub-tile's LIST.  A record whose sub-tile or
        // window differs from its staged predecessor's (and the chunk's first record) is a candidate for "first record of its
        // window in its list": the minimum list position over the candidates IS that record (a candidate that is no true start --
        // the first record of a sub-tile in a later chunk -- has an earlier record of its window in front of it, a candidate
        // too), taken with one LDS atomicMin -- a handful per chunk; two LDS reads + four instructions per record otherwise.
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            const uint32_t i = (uint32_t)(u * kFT + tid);
            if (i < nch) {
                const uint32_t r = stage[i];
                q.rec2[cD[wv][(r & (kCells - 1)) >> 8] + i] = r;
            }
        }
        if (wtab) { // (a pass of its own: a branch per record inside the sweep above kept its LDS reads from being issued together)
            const uint32_t dmask = (((1u << q.wb) - 1u) << kCellBits) | (uint32_t)(kCells - 1) >> 8 << 8;
            uint32_t cand = 0u; // bit u: record u * 1024 + tid of the chunk is a candidate
#pragma unroll
            for (int u0 = 0; u0 < RPT; u0 += 2) { // two records' reads issued together (clamped indices; the empty asm keeps the
                uint32_t r[2], rp[2];               // compiler from putting each read under its own `i < nch` branch, one LDS round trip each;
#pragma unroll                                      // four at a time spilled five of the later chunks' records)
                for (int k = 0; k < 2; ++k) {
                    const uint32_t i = (uint32_t)((u0 + k) * kFT + tid);
                    r[k] = stage[i];                            // (i < 8192: inside the staging area whatever nch is; what lies
                    rp[k] = stage[(i - 1u) & (kSplitSeg - 1)];  // behind the chunk's end, or in front of record 0, is masked below)
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) asm volatile("" : "+v"(r[k]), "+v"(rp[k]));
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const uint32_t i = (uint32_t)((u0 + k) * kFT + tid);
                    cand |= (i < nch && (((r[k] ^ rp[k]) & dmask) != 0u || i == 0u || i == nch - 1u) ? 1u : 0u) << (u0 + k);
                }
            }
            while (cand) { // rare: a handful of records per chunk
                const int u = __builtin_ctz(cand);
                cand &= cand - 1u;
                const uint32_t i = (uint32_t)(u * kFT + tid);
                const uint32_t r = stage[i], rp = stage[i > 0u ? i - 1u : 0u];
                const int bsub = (int)((r & (kCells - 1)) >> 8), bp = (int)((rp & (kCells - 1)) >> 8);
                const int wc = (int)__builtin_amdgcn_ubfe(r, kCellBits, q.wb), wpn = (int)__builtin_amdgcn_ubfe(rp, kCellBits, q.wb);
                const bool same = i > 0u && bp == bsub; // the staged predecessor is the list predecessor
                // (an LDS atomic: global ones sat in front of every chunk's barrier, which waits for the memory queue to drain)
                if (!same || wc != wpn) atomicMin(&s_wst[bsub * (q.n_windows + 1) + wc], cD[wv][bsub] + i - vbeg[wv][bsub]);
                if (same && wc < wpn) s_unsorted = 1;
                if (!same) { s_fw[c][bsub] = wc; if (i > 0u) s_lw[c][bp] = wpn; } // first window of list bsub / last of list bp in this chunk
                if (i == nch - 1u) s_lw[c][bsub] = wc;
            }
        }
        // the staging area is reused by the next chunk; behind the LAST chunk only step 5 follows, which touches LDS alone: a raw
        // barrier there (a __syncthreads() would make the workgroup wait for the drain of its last 32 KB of stores)
        if (wtab && (uint32_t)((c + 1) * kSplitSeg) >= n) LDS_BARRIER();
        else __syncthreads();
    }
    // 5. (CM, TAF) a window index that DEcreases between list neighbours marks the tile unsorted (the walk then filters the whole
    // list per window, as it does after its own scan): inside a chunk the sweep saw it at the neighbour; across chunks it is the
    // list's last window in one chunk against its first in the next one that has any.
    if (wtab) {
        if (tid < kFW) {
            int prev = -1;
            for (int c = 0; c < kWholeChunks; ++c)
                if (s_fw[c][tid] >= 0) { if (s_fw[c][tid] < prev) s_unsorted = 1; prev = s_lw[c][tid]; }
        }
        uint32_t *const wr = q.wst + (long long)g * kFW * (q.n_windows + 1);
        for (int i = tid; i < kFW * (q.n_windows + 1); i += kFT) wr[i] = s_wst[i];
        LDS_BARRIER();
        if (tid == 0) q.wst_flag[g] = s_unsorted ? 2u : 1u;
    }
}

// 4b. Skewed tiles (more than kSplitWhole records).  Reorders every tile's records sub-tile-major (sub-tile = the 256 cells [256 v, 256 v + 256) one workgroup of
// kf_taf_walk owns), STABLY, so that every sub-tile's list is still in stream order.  A tile's list is cut into segments
// of 8192 records, one workgroup each -- a tile that holds a large share of the stream (skew) is split by hundreds of
// workgroups instead of one:
//   kf_split_count    records of every sub-tile in the segment
//   kf_split_offsets  one 16-lane group per tile: running sums over its segments -> where each segment's records of
//                     sub-tile v go inside v's list, and where v's list starts (sub[])
//   kf_split_place    ranks inside the segment with one returning LDS atomic per record on (round, wavefront, sub-tile)
//                     counters: lanes of one instruction are served in lane order, (round, wavefront) is the stream
//                     order of the 64-record batches.
constexpr int kSplitRpt = kSplitSeg / kFT;
struct PlaceLds {
    uint32_t scnt[kSplitRpt][kFW][kFW]; // [round][wavefront][sub-tile] tickets, then prefixes inside the segment
    uint32_t vtot[kFW];                 // records of every sub-tile in the whole tile
    uint32_t stot[kFW], sdst[kFW];      // this segment: records of sub-tile b / where they go in b's list
    uint32_t cE[kFW][kFW], cD[kFW][kFW]; // [wavefront]: slot = cE[b] + prefix + ticket, address = cD[b] + slot (as in kf_split_whole)
    uint32_t stage[kSplitSeg];          // the segment's records, sub-tile-major
};

// CM: the chunk-major partition's form -- the segment is 8192 positions of the tile's list, gathered through the tile's column
// of the directory (cl: L | D | position index, loaded per segment); the tile's space and segment ids were booked by
// kf_split_whole<true>.
struct ColLds {
    uint32_t L[kColMax + 1], D[kColMax];
    uint16_t idx[kSplitSeg / 16];
    uint32_t wsum[kFW + 1];
};

template <bool CM>
__device__ __forceinline__ void split_place_segment(const TileP &q, uint32_t seg, PlaceLds &L, const CmP &cm, const SeqTab &S, ColLds *cl)
{
    constexpr int RPT = kSplitRpt, NE = RPT * kFW;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int g;
    uint32_t beg, nrec, seg_first, seg_last, tile_start;
    if (CM) {
        g = __builtin_amdgcn_readfirstlane((int)cm.segdesc[seg]);
        const int s = g / q.T;
        const uint32_t n = col_load<kFT>(cm, S, s, g - s * q.T, cl->L, cl->D, cl->wsum);
        seg_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)cm.hot_seg0[g]);
        seg_last = seg_first + (n + kSplitSeg - 1) / kSplitSeg;
        beg = (seg - seg_first) * (uint32_t)kSplitSeg; // a list position
        nrec = n - beg < (uint32_t)kSplitSeg ? n - beg : (uint32_t)kSplitSeg;
        tile_start = (uint32_t)__builtin_amdgcn_readfirstlane((int)cm.hot_start[g]);
        col_index<kFT>(cl->L, S.chunk0[s + 1] - S.chunk0[s], beg, beg + nrec, cl->idx);
    } else {
        g = pair_of_segment(q.seg0, q.pairs, seg);
        beg = q.base[g] + (seg - q.seg0[g]) * (uint32_t)kSplitSeg;
        const uint32_t end = q.base[g + 1] - beg < (uint32_t)kSplitSeg ? q.base[g + 1] : beg + kSplitSeg;
        nrec = end - beg;
        seg_first = q.seg0[g];
        seg_last = q.seg0[g + 1];
        tile_start = q.base[g];
    }
    for (int i = tid; i < RPT * kFW * kFW; i += kFT) (&L.scnt[0][0][0])[i] = 0u;
    // where this segment's records of sub-tile v (= this wavefront) go: v's list starts behind the lists of the
    // sub-tiles before it, and the earlier segments of the tile come first inside it.  Every workgroup adds up the
    // tile's segment counts for itself (<= a few hundred segments x 16 values, L2-resident).
    uint32_t before = 0, total = 0;
    for (uint32_t sg = seg_first + lane; sg < seg_last; sg += kWave) {
        const uint32_t c = q.segcnt[(long long)sg * kFW + wv];
        total += c;
        if (sg < seg) before += c;
    }
#pragma unroll
    for (int o2 = 32; o2 >= 1; o2 >>= 1) { before += __shfl_xor(before, o2); total += __shfl_xor(total, o2); }
    if (lane == 0) L.vtot[wv] = total;
    __syncthreads(); // (CM: also orders col_index's writes before the reads below)
    uint32_t vstart = tile_start;
    for (int k = 0; k < wv; ++k) vstart += L.vtot[k];
    if (seg == seg_first && lane == 0) { // the tile's first segment publishes sub[]
        q.sub[(long long)g * kFW + wv] = vstart;
        if (CM) q.sub_end[(long long)g * kFW + wv] = vstart + total;
    }
    uint32_t m[RPT], rk[RPT];
    const uint32_t wvs = (uint32_t)__builtin_amdgcn_readfirstlane(wv);
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
        const uint32_t i = (uint32_t)(u * kFT + tid);
        if (CM) {
            const uint32_t ic = i < nrec ? i : nrec - 1u; // (nrec >= 1: a segment is never empty)
            const uint32_t v = cm.rec[col_addr_wave(cl->L, cl->D, cl->idx, beg, (uint32_t)(u * kFT) + wvs * kWave, ic, nrec)];
            m[u] = i < nrec ? v : 0u;
        } else {
            m[u] = i < nrec ? q.rec[beg + i] : 0u;
        }
    }
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
        const uint32_t i = (uint32_t)(u * kFT + tid);
        rk[u] = 0u;
        if (i < nrec) rk[u] = atomicAdd(&L.scnt[u][wv][(m[u] & (kCells - 1)) >> 8], 1u);
    }
    __syncthreads();
    {
        // wavefront b: exclusive prefix of sub-tile b's counts over (round, wavefront) = stream order inside the segment
        uint32_t v0 = 0, v1 = 0;
        const int e0 = 2 * lane, e1 = 2 * lane + 1;
        if (e0 < NE) v0 = L.scnt[e0 / kFW][e0 % kFW][wv];
        if (e1 < NE) v1 = L.scnt[e1 / kFW][e1 % kFW][wv];
        const uint32_t inc = wave_incl_scan(v0 + v1);
        const uint32_t ex = inc - (v0 + v1);
        if (e0 < NE) L.scnt[e0 / kFW][e0 % kFW][wv] = ex;
        if (e1 < NE) L.scnt[e1 / kFW][e1 % kFW][wv] = ex + v0;
        if (lane == kWave - 1) { L.stot[wv] = inc; L.sdst[wv] = vstart + before; }
    }
    __syncthreads();
    {
        // every wavefront for itself: the segment's records sub-tile-major in the staging area (see kf_split_whole, step 4)
        const int b = lane & 15;
        const uint32_t cnt = L.stot[b];
        uint32_t inc = cnt;
#pragma unroll
        for (int o2 = 1; o2 < kFW; o2 <<= 1) {
            const uint32_t u = __shfl_up(inc, o2);
            if (b >= o2) inc += u;
        }
        if (lane < kFW) { L.cE[wv][lane] = inc - cnt; L.cD[wv][lane] = L.sdst[lane] - (inc - cnt); }
        LDS_FENCE();
    }
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
        const uint32_t i = (uint32_t)(u * kFT + tid);
        if (i < nrec) {
            const uint32_t b = (m[u] & (kCells - 1)) >> 8;
            L.stage[L.cE[wv][b] + L.scnt[u][wv][b] + rk[u]] = m[u];
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < RPT; ++u) { // linear sweep: consecutive threads write consecutive records of a sub-tile's run
        const uint32_t i = (uint32_t)(u * kFT + tid);
        if (i < nrec) {
            const uint32_t r = L.stage[i];
            q.rec2[L.cD[wv][(r & (kCells - 1)) >> 8] + i] = r;
        }
    }
}

template <bool CM>
__global__ __launch_bounds__(kFT) __attribute__((amdgpu_waves_per_eu(8, 8))) void kf_split_place(TileP q, CmP cm, SeqTab S)
{
    __shared__ PlaceLds L;
    __shared__ typename std::conditional<CM, ColLds, uint32_t>::type clmem; // the column: only the chunk-major form has one
    ColLds *cl = reinterpret_cast<ColLds *>(&clmem);
    if (q.hdr->status != 0) return;
    uint32_t nseg = CM ? q.hdr->seg_cursor : q.seg0[q.pairs];
    if (CM && nseg > (uint32_t)cm.max_segs) nseg = (uint32_t)cm.max_segs;
    for (uint32_t seg = blockIdx.x; seg < nseg; seg += gridDim.x) { // (workgroup-uniform: most calls have no segment at all)
        split_place_segment<CM>(q, seg, L, cm, S, cl);
        __syncthreads(); // the LDS image is reused
    }
}

// chunk-major partition: records of every sub-tile in every split segment (kf_split_whole<false> does this in its spare
// workgroups; here the segments only exist once kf_split_whole<true> has run).  Most calls have none: the workgroups leave
// after one load.
__global__ __launch_bounds__(kFT) __attribute__((amdgpu_waves_per_eu(8, 8))) void kf_segcount_cm(TileP q, CmP cm, SeqTab S) // (64 VGPRs: two workgroups per CU)
{
    __shared__ ColLds cl;
    __shared__ uint32_t wtot[kFW][kFW];
    const int tid = threadIdx.x, wv = tid >> 6;
    if (q.hdr->status != 0) return;
    uint32_t nseg = q.hdr->seg_cursor;
    if (nseg > (uint32_t)cm.max_segs) nseg = (uint32_t)cm.max_segs;
    for (uint32_t seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
        const int g = __builtin_amdgcn_readfirstlane((int)cm.segdesc[seg]), s = g / q.T;
        const uint32_t n = col_load<kFT>(cm, S, s, g - s * q.T, cl.L, cl.D, cl.wsum);
        const uint32_t beg = (seg - (uint32_t)__builtin_amdgcn_readfirstlane((int)cm.hot_seg0[g])) * (uint32_t)kSplitSeg;
        const uint32_t nrec = n - beg < (uint32_t)kSplitSeg ? n - beg : (uint32_t)kSplitSeg;
        col_index<kFT>(cl.L, S.chunk0[s + 1] - S.chunk0[s], beg, beg + nrec, cl.idx);
        if (tid < kFW * kFW) (&wtot[0][0])[tid] = 0u;
        __syncthreads();
        uint32_t v[kSplitRpt];
        const uint32_t wvs = (uint32_t)__builtin_amdgcn_readfirstlane(wv);
#pragma unroll
        for (int u = 0; u < kSplitRpt; ++u) {
            const uint32_t i = (uint32_t)(u * kFT + tid), ic = i < nrec ? i : nrec - 1u;
            v[u] = cm.rec[col_addr_wave(cl.L, cl.D, cl.idx, beg, (uint32_t)(u * kFT) + wvs * kWave, ic, nrec)];
        }
#pragma unroll
        for (int u = 0; u < kSplitRpt; ++u)
            if ((uint32_t)(u * kFT + tid) < nrec) atomicAdd(&wtot[wv][(v[u] & (kCells - 1)) >> 8], 1u);
        __syncthreads();
        if (tid < kFW) {
            uint32_t t = 0;
#pragma unroll
            for (int w = 0; w < kFW; ++w) t += wtot[w][tid];
            q.segcnt[(long long)seg * kFW + tid] = t;
        }
        __syncthreads(); // the column and wtot are reused
    }
}


// 5. One workgroup of eight wavefronts per (sequence, tile, sub-tile of 256 cells).  The per-window sums of a cell do
// not depend on each other -- only the FIFO steps that consume them are sequential -- so the eight wavefronts take ONE
// WINDOW EACH (rounds of eight windows) and the FIFO steps follow with one cell per lane:
//   phase 0  the list is scanned once for the first record of every window (the split is stable: a time-sorted stream
//            gives a window-sorted list; a window that runs backwards switches the whole sub-tile to the general mode,
//            where every wavefront sweeps the whole list for its window's records);
//   phase 1  wavefront w, passes of up to 256 records of its window:
//              1. every record takes a ticket from its cell's LDS counter with one returning atomic (two 16-bit
//                 counters per word): lanes of one instruction are served in lane order and the four instructions of a
//                 pass are in stream order, so the ticket is the record's stream rank inside its cell -- a STABLE
//                 counting sort without any ordering pass;
//              2. the lanes read the counts of their four cells, a wavefront scan turns them into segment offsets;
//              3. every record's f32 value goes to sorted[offset of its cell + ticket];
//              4. every lane adds its cells' segments front to back into registers: the reference's sequential
//                 `sum += t - 1` (generate_taf.py:25-26);
//            then (sum, count) of the 256 cells go to LDS;
//   phase 2  lane c of the first four wavefronts owns cell c: K-deep FIFO row in registers (consecutive lanes hold
//            consecutive 32-byte rows of the (H, W, 2, K) state: whole lines), one FIFO step per window in order
//            (generate_taf.py:27-49), skipped for windows that are empty in the whole sequence (:40-41).
// (ds_add_f32 would do the ordered sum in one instruction -- it applies same-address lanes in lane order with v_add_f32
// rounding, checked by the self-test below -- but runs at 192 cycles per wave-instruction per CU: measured, not used.)
constexpr int kWalkWaves = 8;
constexpr int kWalkThreads = kWalkWaves * kWave;
constexpr int kWalkRpt = 4;                 // records per lane and pass
constexpr int kWalkChunk = kWalkRpt * kWave;
constexpr int kWalkSlots = 4;               // ranks of a cell inside one pass that have a plane of their own
constexpr int kWalkWaveWords = kWalkSlots * kSubCells + kSubCells / 2; // LDS words per wavefront: the planes + the ticket counters
// (two 16-bit counters per word: 36 KB of planes and counters for the eight wavefronts -- FOUR workgroups per CU; with 32-bit
// counters the workgroup needs 41.3 KB, three fit, and the walk ran 8 % slower although it issued fewer instructions, measured)

// CMD (chunk-major partition, direct mode: the partition's bins ARE the sub-tiles): the sub-tile's list does not exist yet --
// its runs sit in the chunks' stretches of rec[].  The workgroup reads its column of the directory, books the list's space
// through the header's cursor, copies the runs there in chunk order (a pure copy: groups of 16 lanes take a run each) and then
// walks the contiguous list like any other; no gather kernel, no second launch.
constexpr int kColDirect = 2047; // chunks per sequence the walk's column fits (two arrays over the wavefronts' plane areas)
constexpr int kWalkListCap = 3584; // records of a sub-tile's list kept in LDS by the CMD walk (14 KB: three workgroups per CU): no trip to memory between
                                   // the gather and the two sweeps over the list; longer lists are copied to rec2[]
template <bool K8, bool CMD = false>
__global__ __launch_bounds__(kWalkThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) void kf_taf_walk(TileP q, CmP cm, SeqTab S)
{
    // per wavefront: kWalkSlots planes of 256 floats (plane r, cell c = the value of the cell's r-th record of the pass; +0 when
    // there is none) + 256 ticket counters.  The (mean, count) rows phase 2 reads lie over planes 0 and 1 of their wavefront;
    // the CMD column and the uint8 staging at the end lie over the whole area.
    __shared__ __attribute__((aligned(16))) uint32_t s_area[kWalkWaves][kWalkWaveWords];
    __shared__ uint32_t wstart[FRLW_MAX_WINDOWS + 1];
    __shared__ uint32_t thr[kLeakyTableWords]; // thresholds + bucket table of the leaky transform (leaky_u8_bucket_n)
    __shared__ int s_unsorted;
    __shared__ __attribute__((aligned(16))) uint32_t s_list[CMD ? kWalkListCap : 4]; // CMD: the gathered list, when it fits (else it goes to rec2[])
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int sg = blockIdx.x, g = sg / kFW, sub = sg - g * kFW;
    const int s = g / q.T, tile = g - s * q.T;
    // Everything the workgroup needs from the header and the list tables is requested in ONE go, in front of the status test:
    // each of these is a scalar load of its own round trip, and one behind a branch waits for the one in front of it.  (The
    // tables lie at addresses the plan fixes: reading them is safe whatever the status says; the LIST is only read behind it.)
    const int32_t status0 = q.hdr->status;
    const unsigned long long wmask = q.hdr->wmask[s];
    const uint32_t mul_bad0 = q.hdr->mul_bad;
    uint32_t beg = 0u, end = 0u;
    uint32_t wtab = 0u; // 1 / 2: kf_split_whole<true> has left this tile's window starts / found its list unsorted (TileP::wst_flag)
    if (!CMD) {
        // (sub[] of the NEXT pair is only written if that pair went through a split kernel: take the tile's own end)
        beg = q.sub[sg];
        end = q.sub_end ? q.sub_end[sg] : ((sub == kFW - 1 && !q.direct) ? q.base[g + 1] : q.sub[sg + 1]);
        if (q.wst) wtab = q.wst_flag[g];
    }
    if (status0 != 0) return; // data-dependent error: nothing is written (the caller re-runs the general path)
    WPROF_INIT();
    if (q.tile_walk && q.hdr->unsorted[s] == 0u && q.base[g + 1] - q.base[g] <= q.tile_max) return; // done by kf_taf_tile
    const int K = K8 ? 8 : q.K;
    const int NW = q.n_windows;
    const uint32_t *list = q.rec2; // where the sweeps below read the list (CMD: LDS when the list fits)
    if (CMD) {
        uint32_t *colL = &s_area[0][0], *colD = colL + (kColDirect + 1); // (free until phase 1 starts: zeroed below)
        static_assert(2 * (kColDirect + 1) <= kWalkWaves * kWalkWaveWords, "the column fits");
        const int C = S.chunk0[s + 1] - S.chunk0[s];
        const uint32_t n = col_load<kWalkThreads>(cm, S, s, sg - s * cm.TB, colL, colD, wstart);
        const bool in_lds = n <= (uint32_t)kWalkListCap; // workgroup-uniform
        uint32_t *dstl;
        if (in_lds) {
            beg = 0u;
            dstl = s_list;
            list = s_list;
        } else {
            if (tid == 0) wstart[0] = atomicAdd(&q.hdr->rec_cursor, n);
            __syncthreads();
            beg = (uint32_t)__builtin_amdgcn_readfirstlane((int)wstart[0]);
            dstl = q.rec2 + beg;
        }
        end = beg + n;
        // runs -> list: every group of 16 lanes takes runs g16, g16 + 32, ...; four runs' loads in flight before their stores
        const int g16 = tid >> 4, l16 = tid & 15;
        for (int c0 = g16; c0 < C; c0 += 4 * (kWalkThreads / 16)) {
            uint32_t v[4], at[4], cnt[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + u * (kWalkThreads / 16), cc = c < C ? c : C - 1;
                const uint32_t lo = colL[cc];
                cnt[u] = c < C ? colL[cc + 1] - lo : 0u;
                at[u] = lo;
                // (clamped index: a lane behind the run's end re-reads an address that exists; runs longer than 16 loop below)
                v[u] = cm.rec[colD[cc] + lo + ((uint32_t)l16 < cnt[u] ? (uint32_t)l16 : 0u)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if ((uint32_t)l16 < cnt[u]) dstl[at[u] + l16] = v[u];
                if (cnt[u] > 16u) { // (wave-divergent, rare for the short runs of a direct-mode call)
                    const int c = c0 + u * (kWalkThreads / 16);
                    const uint32_t d = colD[c];
                    for (uint32_t j = 16u + l16; j < cnt[u]; j += 16u) dstl[at[u] + j] = cm.rec[d + at[u] + j];
                }
            }
        }
        __syncthreads(); // the list is complete (and visible to the workgroup); the column's space is free again
    }
    // phase-2 ownership: cell = tid (< 256), the whole K-slot row in one lane (the first four wavefronts; K = 8 used to split
    // the row over two lanes so that all 512 threads work -- but the kernel is bound by VALU issue, and a step costs a
    // half row's lane the same 13 instructions as a whole row's).  Cell c: pixel 128 sub + c / 2 of the tile, polarity c & 1.
    const int ty = tile / q.tiles_x, tx = tile - ty * q.tiles_x;
    const int x0 = tx << q.twl, y0 = ty << q.thl, tw1 = (1 << q.twl) - 1;
    const long long plane = (long long)q.H * q.W;
    const bool owner = tid < kSubCells;
    // (the row's address is worked out again wherever it is needed, from a thread id the compiler cannot recognise: kept alive
    // across phase 1 its pieces were spilled -- 32 bytes of scratch per lane, which the write counter showed as 73 MB per encode)
    auto row_of = [&](int t, bool &in_frame) -> float * {
        asm volatile("" : "+v"(t));
        const int c = t & (kSubCells - 1), p2 = sub * (kSubCells / 2) + (c >> 1);
        const int yy = y0 + (p2 >> q.twl), xx = x0 + (p2 & tw1);
        in_frame = t < kSubCells && yy < q.H && xx < q.W;
        return q.state + (((long long)s * plane + (long long)yy * q.W + xx) * 2 + (c & 1)) * K;
    };
    float st[kMaxK];
    // The state rows are needed behind phase 1 (and their registers should not be alive during it) -- but their trip to HBM should
    // overlap the window scan: one word of every row is requested HERE (a wavefront's rows are 2 KB in a row: all its lines come
    // in) and dropped behind phase 0; the real load then finds the lines in the caches.
    float row_touch = 0.0f;
    {
        bool in_frame;
        const float *r = row_of(tid, in_frame);
        if (in_frame) row_touch = r[0];
    }

    // The window starts: read from the table kf_split_whole<true> left (wtab != 0: no scan, no barrier -- the planes and counters
    // a wavefront zeroes are its own), or found by a scan of the list (phase 0: every other partition form).
    uint32_t first_w = 0u;
    unsigned long long nonempty = 0ull;
    bool general;
    const bool use_mul = mul_bad0 == 0u; // checked for every r of the domain by the partition kernel
    const double rcp = q.rcp;
    const uint32_t wfield = (1u << q.wb) - 1u;
    const int rshift = kCellBits + q.wb;
    {   // planes and counters start at zero (CMD: the column is dead since the barrier behind the gather)
        uint2 *z = (uint2 *)&s_area[wv][0];
        static_assert(kWalkWaveWords % (2 * kWave) == 0, "whole 8-byte sweeps");
#pragma unroll
        for (int i = 0; i < kWalkWaveWords / 2 / kWave; ++i) z[i * kWave + lane] = make_uint2(0u, 0u);
    }
    if (!CMD && wtab != 0u) { // workgroup-uniform
        // lane = window: records of the list in front of window `lane` (window-sorted list); entry NW = all of them
        const uint32_t *wrow = q.wst + (long long)sg * (NW + 1);
        const uint32_t fw0 = wrow[lane < NW ? lane : NW];
        // the list's lines are requested while the table row is on its way (one word per 128-byte line and thread: 16 384 records
        // per round); phase 1's loads, which wait for the row, then find them in the caches
        uint32_t list_touch = 0u;
        for (uint32_t i = beg + 32u * (uint32_t)tid; i < end; i += 32u * kWalkThreads) list_touch |= list[i];
        asm volatile("" ::"v"(list_touch));
        first_w = fw0 != 0xffffffffu ? beg + fw0 : end; // what the scan leaves in wstart[]
        nonempty = __ballot(lane < NW && fw0 != 0xffffffffu);
        general = wtab == 2u;
        asm volatile("" ::"v"(row_touch));
        WPROF(0);
        WPROF(1);
    } else {
    for (int i = tid; i <= NW; i += kWalkThreads) wstart[i] = end;
    if (tid == 0) s_unsorted = 0;
    __syncthreads();
    WPROF(0);
    // ---- phase 0: first record of every window; a window index that decreases = not window-sorted.  A thread looks at four
    // consecutive records -- ONE 16-byte load from the 16-byte block they share (the list's neighbours in front of `beg` and
    // behind `end` are read and masked: the blocks lie inside rec2[] / the LDS list) -- and at the record in front of them.
    {
        const uint32_t n_list = end - beg;
        for (uint32_t c0 = beg & ~3u; c0 < end; c0 += 4 * kWalkThreads) {
            const uint32_t i0 = c0 + 4u * (uint32_t)tid, il = i0 < end ? i0 : (end - 1u) & ~3u; // (lanes behind the end repeat the last block: harmless)
            const uint4 v4 = *(const uint4 *)(list + il);
            const uint32_t pv = list[il > beg ? il - 1u : beg];
            const uint32_t w0 = __builtin_amdgcn_ubfe(v4.x, kCellBits, q.wb), w1 = __builtin_amdgcn_ubfe(v4.y, kCellBits, q.wb);
            const uint32_t w2 = __builtin_amdgcn_ubfe(v4.z, kCellBits, q.wb), w3 = __builtin_amdgcn_ubfe(v4.w, kCellBits, q.wb);
            const uint32_t wp = il > beg ? __builtin_amdgcn_ubfe(pv, kCellBits, q.wb) : 0xffffffffu; // the record in front (none: 0xffffffff)
            const bool full = n_list >= 4u && il - beg <= n_list - 4u; // all four records belong to the list (unsigned: false in front of beg)
            if (full) {
                if (w0 != wp || w1 != w0 || w2 != w1 || w3 != w2) { // a window starts here: a handful of lanes per list
                    if (w0 != wp) { atomicMin(&wstart[w0], il); if (wp != 0xffffffffu && w0 < wp) s_unsorted = 1; }
                    if (w1 != w0) { atomicMin(&wstart[w1], il + 1u); if (w1 < w0) s_unsorted = 1; }
                    if (w2 != w1) { atomicMin(&wstart[w2], il + 2u); if (w2 < w1) s_unsorted = 1; }
                    if (w3 != w2) { atomicMin(&wstart[w3], il + 3u); if (w3 < w2) s_unsorted = 1; }
                }
            } else { // the blocks the list's ends lie in: record by record
                const uint32_t ws[5] = {wp, w0, w1, w2, w3};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t k = il + (uint32_t)e;
                    if (k - beg < n_list) {
                        const bool first = k == beg;
                        if (first || ws[e + 1] != ws[e]) {
                            atomicMin(&wstart[ws[e + 1]], k);
                            if (!first && ws[e + 1] < ws[e]) s_unsorted = 1;
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    asm volatile("" ::"v"(row_touch)); // (the touch has landed; nothing else wants the value)
    WPROF(1);
    general = s_unsorted != 0;
    // wstart[w'] = first record of window w', or `end` for a window without records: window w's stretch starts at the minimum over
    // w' >= w (a window without records starts where the next one does) -- every wavefront finds that window for itself in a
    // ballot over the lanes (lane = window), instead of one thread walking the table between two barriers (12 % of the
    // workgroup's life)
    first_w = wstart[lane < NW ? lane : NW];
    nonempty = __ballot(lane < NW && first_w != end); // (lane = window; window-sorted list: their starts ascend)
    }
    if (general && tid == 0) atomicAdd(&q.hdr->filtered_tiles, 1u);
    WPROF(2);

    float *rplane = (float *)&s_area[wv][0];                      // [kWalkSlots][256]
    uint32_t *cnt = &s_area[wv][kWalkSlots * kSubCells];          // [128] two 16-bit tickets per word, all zero between passes
    const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma nounroll
    for (int g0 = 0; g0 < NW; g0 += kWalkWaves) {
        // ---- phase 1: wavefront wv sums window g0 + wv; lane l owns cells 4 l .. 4 l + 3
        const int w = g0 + wv;
        float sum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        uint32_t num[4] = {0u, 0u, 0u, 0u};
        if (g0 > 0) { // the (mean, count) rows of the previous round lay over planes 0 and 1
            ((float4 *)rplane)[lane] = zero4;
            ((float4 *)rplane)[kWave + lane] = zero4;
        }
        if (w < NW) {
            uint32_t lo = beg, hi = end;
            if (!general) { // the first window with records at or behind w (behind w) starts the stretch (ends it); none: the list's end
                const unsigned long long at = nonempty >> w, behind = w + 1 < 64 ? nonempty >> (w + 1) : 0ull;
                lo = at ? (uint32_t)__builtin_amdgcn_readlane((int)first_w, w + __builtin_ctzll(at)) : end;
                hi = behind ? (uint32_t)__builtin_amdgcn_readlane((int)first_w, w + 1 + __builtin_ctzll(behind)) : end;
            }
#pragma nounroll
            for (uint32_t ptr = lo; ptr < hi; ptr += kWalkChunk) {
                uint32_t m[kWalkRpt], rk[kWalkRpt];
                float val[kWalkRpt];
                // (no load under a lane condition -- each would wait for its own data: lanes behind the end re-read the last record)
#pragma unroll
                for (int u = 0; u < kWalkRpt; ++u) {
                    const uint32_t i = ptr + (uint32_t)(u * kWave + lane);
                    m[u] = list[i < hi ? i : hi - 1u];
                }
                // the ticket = the record's stream rank inside its cell: lanes of one returning LDS atomic are served in lane
                // order, the four instructions of a pass are in stream order
#pragma unroll
                for (int u = 0; u < kWalkRpt; ++u) {
                    const bool take = ptr + (uint32_t)(u * kWave + lane) < hi && (int)((m[u] >> kCellBits) & wfield) == w;
                    rk[u] = 0xffffffffu; // not taken
                    if (take) {
                        const uint32_t lc = m[u] & 255u, sh = (lc & 1u) << 4;
                        rk[u] = (atomicAdd(&cnt[lc >> 1], 1u << sh) >> sh) & 0xffffu;
                    }
                }
                // the value t - 1 with t = (t - t_min) / (w + 1e-8) in f64 (generate_taf.py:215, :26), for every lane (those
                // without a record never store theirs)
                if (use_mul) {
#pragma unroll
                    for (int u = 0; u < kWalkRpt; ++u) val[u] = (float)((double)(m[u] >> rshift) * rcp) - 1.0f;
                } else {
#pragma unroll
                    for (int u = 0; u < kWalkRpt; ++u) {
                        const uint32_t r = m[u] >> rshift;
                        val[u] = q.tlut[r < q.win ? r : q.win];
                    }
                }
                LDS_FENCE();
                const uint2 np = ((const uint2 *)cnt)[lane]; // the counts of cells 4 l .. 4 l + 3
                LDS_FENCE();
                ((uint2 *)cnt)[lane] = make_uint2(0u, 0u);
                const uint4 nn = make_uint4(np.x & 0xffffu, np.x >> 16, np.y & 0xffffu, np.y >> 16);
                const uint32_t n01 = nn.x > nn.y ? nn.x : nn.y, n23 = nn.z > nn.w ? nn.z : nn.w, nmax = n01 > n23 ? n01 : n23;
                // Ranks 0 .. 3 of every cell go straight to plane[rank][cell] (no offsets, no scan, no sorted list); the owner
                // adds its four cells' planes front to back -- a cell without a rank-r record reads +0, and x + 0 == x for every
                // sum that can occur (sums start at +0 and never become -0) -- and clears them.  Cells with more than four
                // records in the pass (0.06 % at 0.7 records per cell and window) cost the wavefront further rounds of four.
#pragma nounroll
                for (uint32_t base = 0u;;) {
#pragma unroll
                    for (int u = 0; u < kWalkRpt; ++u) {
                        const uint32_t rr = rk[u] - base; // (not taken: 0xffffffff - base stays out of range)
                        if (rr < (uint32_t)kWalkSlots) rplane[rr * kSubCells + (m[u] & 255u)] = val[u];
                    }
                    LDS_FENCE();
#pragma unroll
                    for (int r = 0; r < kWalkSlots; ++r) { // sum += t - 1 in stream order, generate_taf.py:26
                        const float4 pr = ((const float4 *)(rplane + r * kSubCells))[lane];
                        LDS_FENCE();
                        ((float4 *)(rplane + r * kSubCells))[lane] = zero4;
                        sum[0] = sum[0] + pr.x;
                        sum[1] = sum[1] + pr.y;
                        sum[2] = sum[2] + pr.z;
                        sum[3] = sum[3] + pr.w;
                    }
                    base += (uint32_t)kWalkSlots;
                    if (__ballot(nmax > base) == 0ull) break;
                }
                num[0] += nn.x; num[1] += nn.y; num[2] += nn.z; num[3] += nn.w;
                LDS_FENCE();
            }
        }
        // the mean is taken HERE, once per (cell, window): 2048 correctly rounded divisions per workgroup and round of windows
        // instead of 4096 in phase 2 (both lanes of a cell); rows over planes 0 (means) and 1 (counts)
        ((float4 *)rplane)[lane] = make_float4(fifo_mean(num[0], sum[0]), fifo_mean(num[1], sum[1]), fifo_mean(num[2], sum[2]), fifo_mean(num[3], sum[3]));
        ((uint4 *)(rplane + kSubCells))[lane] = make_uint4(num[0], num[1], num[2], num[3]);
        // the 256 thresholds of the leaky transform come to LDS behind phase 2 (requested here, stored in front of its barrier):
        // at the top of the kernel the load's trip was on the path of every wavefront's first barrier
        static_assert(kLeakyTableWords <= kWalkThreads, "one word per thread");
        uint32_t thr_v = 0u;
        if (g0 == 0 && tid < kLeakyTableWords) thr_v = q.leaky_thr[tid];
        if (g0 == 0) { // the state rows: requested behind phase 1 (their registers are not alive during it), used behind the barrier
#pragma unroll
            for (int kk = 0; kk < kMaxK; ++kk) st[kk] = 0.0f;
            bool ok;
            const float *srow = row_of(tid, ok);
            if (ok) {
                if (K8) {
                    const float4 a = ((const float4 *)srow)[0], b = ((const float4 *)srow)[1];
                    st[0] = a.x; st[1] = a.y; st[2] = a.z; st[3] = a.w; st[4] = b.x; st[5] = b.y; st[6] = b.z; st[7] = b.w;
                } else {
#pragma unroll
                    for (int kk = 0; kk < kMaxK; ++kk)
                        if (kk < K) st[kk] = srow[kk];
                }
            }
        }
        WPROF(3);
        __syncthreads();
        WPROF(4);
        // ---- phase 2: one cell per lane, the FIFO steps of this round's windows in order
        if (owner) {
            // (the eight (count, mean) pairs are requested together, in front of the steps: a step that waits for its own pair
            // is an LDS round trip on the workgroup's critical path, eight times)
            uint32_t rn[kWalkWaves];
            float rm[kWalkWaves];
#pragma unroll
            for (int ws = 0; ws < kWalkWaves; ++ws) {
                rn[ws] = s_area[ws][kSubCells + tid];
                rm[ws] = __uint_as_float(s_area[ws][tid]);
            }
#pragma unroll
            for (int ws = 0; ws < kWalkWaves; ++ws)
                if (g0 + ws < NW && ((wmask >> (g0 + ws)) & 1ull)) fifo_step(st, K, true, rn[ws], rm[ws]);
        }
        if (g0 == 0 && tid < kLeakyTableWords) thr[tid] = thr_v;
        __syncthreads();
        WPROF(5);
    }

    // ---- write-out: state, optional f32 view (2K, H, W), optional uint8 leaky transform (K, 2, H, W)
    uint8_t *ob = (uint8_t *)&s_area[0][0]; // [2K planes][128 pixels of the sub-tile] (the last phase 2 ended with a barrier)
    // K = 8: a lane holds a whole 32-byte row; stored from here it would leave as two instructions of 16 bytes at a 32-byte stride
    // -- half a sector each, twice the write requests (WRITE_SIZE showed 147 MB for 74).  The rows go through LDS instead and
    // leave below from all 512 threads, 16 bytes each, consecutive threads writing consecutive pieces: whole lines.
    float4 *rowst = (float4 *)&s_area[1][0]; // [256 rows][2] (behind the 2 KB of uint8 staging; the plane areas are dead)
    static_assert(kWalkWaveWords * 4 >= 2 * kMaxK * (kSubCells / 2) && (kWalkWaves - 1) * kWalkWaveWords * 4 >= kSubCells * 32, "staging fits");
    if (K8 && owner) {
        rowst[2 * tid] = make_float4(st[0], st[1], st[2], st[3]);
        rowst[2 * tid + 1] = make_float4(st[4], st[5], st[6], st[7]);
    }
    const int pol = tid & 1;
    {
        bool ok;
        float *srow = row_of(tid, ok);
        if (ok) {
            if (!K8) {
#pragma unroll
                for (int k = 0; k < kMaxK; ++k)
                    if (k < K) srow[k] = st[k];
            }
            if (q.view_f32) { // (the row's element index / (2 K) is the pixel, generate_taf.py:55)
                float *vw = q.view_f32 + (long long)s * 2 * K * plane + ((srow - q.state) / (2 * K) - (long long)s * plane);
#pragma unroll
                for (int k = 0; k < kMaxK; ++k)
                    if (k < K) vw[(long long)(2 * k + pol) * plane] = st[k];
            }
        }
    }
    if (q.out_u8) {
        if (owner) {
            uint8_t lv[kMaxK];
            leaky_u8_bucket_n<kMaxK>(st, thr, lv); // the eight table look-ups in flight together
#pragma unroll
            for (int k = 0; k < kMaxK; ++k) {
                if (k < K) {
                    const int ko = q.flip ? (K - 1 - k) : k;
                    ob[(2 * ko + pol) * (kSubCells / 2) + (tid >> 1)] = lv[k];
                }
            }
        }
    }
    WPROF(6);
    if (K8 || q.out_u8) __syncthreads(); // (workgroup-uniform)
    WPROF(7);
    if (K8) { // thread t: half t & 1 of the row of cell t / 2
        const int c2 = tid >> 1, pt2 = sub * (kSubCells / 2) + (c2 >> 1);
        const int py2 = y0 + (pt2 >> q.twl), px2 = x0 + (pt2 & tw1);
        if (py2 < q.H && px2 < q.W)
            ((float4 *)(q.state + (((long long)s * plane + (long long)py2 * q.W + px2) * 2 + (c2 & 1)) * 8))[tid & 1] = rowst[tid];
    }
    if (q.out_u8) {
        // the (K, 2, H, W) volume leaves plane by plane in 16-pixel pieces: one 16-byte store where the row allows
        for (int c = tid; c < 2 * K * 8; c += kWalkThreads) {
            const int pl = c >> 3, part = c & 7;
            const int p16 = sub * (kSubCells / 2) + 16 * part;
            const int y = y0 + (p16 >> q.twl), x = x0 + (p16 & tw1);
            if (y >= q.H || x >= q.W) continue;
            const uint8_t *src = ob + pl * (kSubCells / 2) + 16 * part;
            uint8_t *dst = q.out_u8 + ((long long)s * 2 * K + pl) * plane + (long long)y * q.W + x;
            if (x + 16 <= q.W && (((uintptr_t)dst) & 15u) == 0) {
                *(uint4 *)dst = *(const uint4 *)src;
            } else {
                const int nv = q.W - x < 16 ? q.W - x : 16;
                for (int e = 0; e < nv; ++e) dst[e] = src[e];
            }
        }
    }
    WPROF(8);
    WPROF_END();
}

// =====================================================================================================================
// Tile walk: the second-level split done in LDS by the kernel that consumes it
// =====================================================================================================================
// kf_split_whole re-sorts a tile's records sub-tile-major through HBM (75 MB read + 41 MB written per 10 M events) only so
// that the sub-tile kernel finds its list contiguous.  A tile-walk workgroup instead streams the tile's list (tile-major,
// stream order, written by kf_scatter) in chunks, splits every chunk STABLY by sub-tile inside LDS with the same
// lane-ordered tickets, and hands each sub-tile's piece to the wavefront that owns the sub-tile.  NW wavefronts per
// workgroup = NW sub-tiles; 16 / NW workgroups ("parts") share a tile and each reads the whole list (from the XCD's L2:
// the parts of a tile are placed on one XCD) but keeps only its own sub-tiles.
//
// Wave-private pass (the core of kf_taf_walk's phase 1, factored out): up to 256 records of ONE sub-tile, in stream order,
// become per-cell ordered segments -- a ticket per record from two-per-word 16-bit LDS counters (lane-ordered, so the
// ticket is the stream rank inside the cell), a wave scan of the cell counts, values to sorted[offset(cell) + ticket] --
// and every lane then walks the segments of its four cells (64 j + lane) front to back.
struct WavePass {
    uint32_t *cnt;  // [128]: two 16-bit tickets per word, all zero between passes
    uint16_t *off;  // [256]
    float *sorted;  // [256] (Event Volume: [2 * 256 + 2], pairs of weights + one all-zero pair)
};

// m[u], u < 4: the lane's records of this pass (0xffffffff = none), record u * 64 + lane of the pass in stream order;
// val(m) -> the f32 to sort.  Returns the lane's four cell counts n[] and segment starts o[] in sorted[]; the caller
// walks the segments (wave_segments below) and ends the pass with LDS_FENCE().
template <class Val>
__device__ __forceinline__ void wave_sort(const WavePass &P, const uint32_t (&m)[4], int lane, Val val, uint32_t (&n)[4], uint32_t (&o)[4])
{
    uint32_t rk[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        rk[u] = 0xffffffffu;
        if (m[u] != 0xffffffffu) {
            const uint32_t lc = m[u] & 255u, sh = 16u * (lc & 1u);
            rk[u] = (atomicAdd(&P.cnt[lc >> 1], 1u << sh) >> sh) & 0xffffu;
        }
    }
    LDS_FENCE();
#pragma unroll
    for (int j = 0; j < 4; ++j) n[j] = (P.cnt[32 * j + (lane >> 1)] >> (16 * (lane & 1))) & 0xffffu;
    LDS_FENCE();
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (!(lane & 1)) P.cnt[32 * j + (lane >> 1)] = 0u; // after both lanes of the word have read it
    {
        const uint32_t tl = n[0] + n[1] + n[2] + n[3];
        const uint32_t inc = wave_incl_scan(tl);
        o[0] = inc - tl; o[1] = o[0] + n[0]; o[2] = o[1] + n[1]; o[3] = o[2] + n[2];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) P.off[64 * j + lane] = (uint16_t)o[j];
    LDS_FENCE();
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (rk[u] != 0xffffffffu) P.sorted[(uint32_t)P.off[m[u] & 255u] + rk[u]] = val(m[u]);
    LDS_FENCE();
}

// The same with a PAIR of f32 per record (val2(m, a, b)): sorted[2 * slot], sorted[2 * slot + 1]; slot 256 is kept all zero.
template <class Val2>
__device__ __forceinline__ void wave_sort2(const WavePass &P, const uint32_t (&m)[4], int lane, Val2 val2, uint32_t (&n)[4], uint32_t (&o)[4])
{
    uint32_t rk[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        rk[u] = 0xffffffffu;
        if (m[u] != 0xffffffffu) {
            const uint32_t lc = m[u] & 255u, sh = 16u * (lc & 1u);
            rk[u] = (atomicAdd(&P.cnt[lc >> 1], 1u << sh) >> sh) & 0xffffu;
        }
    }
    LDS_FENCE();
#pragma unroll
    for (int j = 0; j < 4; ++j) n[j] = (P.cnt[32 * j + (lane >> 1)] >> (16 * (lane & 1))) & 0xffffu;
    LDS_FENCE();
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (!(lane & 1)) P.cnt[32 * j + (lane >> 1)] = 0u;
    {
        const uint32_t tl = n[0] + n[1] + n[2] + n[3];
        const uint32_t inc = wave_incl_scan(tl);
        o[0] = inc - tl; o[1] = o[0] + n[0]; o[2] = o[1] + n[1]; o[3] = o[2] + n[2];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) P.off[64 * j + lane] = (uint16_t)o[j];
    LDS_FENCE();
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (rk[u] != 0xffffffffu) {
            float a, b;
            val2(m[u], u, a, b);
            *(float2 *)&P.sorted[2u * ((uint32_t)P.off[m[u] & 255u] + rk[u])] = make_float2(a, b);
        }
    LDS_FENCE();
}

// add2(j, a, b): cell j of this lane receives the pair next; slots behind a segment's end read the all-zero pair, and
// adding +0 to these non-negative sums changes nothing -- no select around the accumulators at all.
template <class Add2>
__device__ __forceinline__ void wave_segments2(const WavePass &P, const uint32_t (&n)[4], const uint32_t (&o)[4], Add2 add2)
{
    uint32_t nmax = n[0] > n[1] ? n[0] : n[1];
    nmax = n[2] > nmax ? n[2] : nmax;
    nmax = n[3] > nmax ? n[3] : nmax;
    const uint32_t nm = wave_max_u32(nmax);
#pragma nounroll
    for (uint32_t a = 0; a < nm; ++a) {
        float2 e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = *(const float2 *)&P.sorted[2u * (a < n[j] ? o[j] + a : 256u)];
#pragma unroll
        for (int j = 0; j < 4; ++j) add2(j, e[j].x, e[j].y);
    }
}

// add(j, v, live): "cell j of this lane receives v next" when live -- a select, not a branch (divergent control flow
// around the accumulators makes the compiler keep copies of all of them).
template <class Add>
__device__ __forceinline__ void wave_segments(const WavePass &P, const uint32_t (&n)[4], const uint32_t (&o)[4], Add add)
{
    uint32_t nmax = n[0] > n[1] ? n[0] : n[1];
    nmax = n[2] > nmax ? n[2] : nmax;
    nmax = n[3] > nmax ? n[3] : nmax;
    const uint32_t nm = wave_max_u32(nmax); // the longest segment of the wavefront: a uniform trip count
#pragma nounroll
    for (uint32_t a = 0; a < nm; ++a) {
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t at = o[j] + a;
            e[j] = P.sorted[at < 255u ? at : 255u];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) add(j, e[j], a < n[j]);
    }
}

// In-LDS stable split of one chunk of a tile's list.  NT threads, RPT records per thread: record u * NT + tid of the chunk
// (stream order = (round u, wavefront, lane)).  Every record of one of this workgroup's NW sub-tiles takes a ticket from
// the (round, wavefront, sub-tile) counter -- lane-ordered -- wavefront b scans sub-tile b's RPT * NW counters in stream
// order, the sub-tile totals are scanned by every wavefront for itself, and the records land sub-tile-major in stage[].
// Returns through sb / nb the start and length of THIS wavefront's sub-tile in stage[].  Four workgroup barriers.
template <int NW, int RPT>
struct TileSplit {
    static constexpr int NT = NW * kWave;
    static constexpr int CH = NT * RPT;
    uint32_t scnt[RPT][NW][NW]; // [round][wavefront][sub-tile]
    uint32_t btot[NW];
    uint32_t stage[CH];
};

template <int NW, int RPT>
__device__ __forceinline__ void tile_split(TileSplit<NW, RPT> &L, const uint32_t (&m)[RPT], const bool (&mine)[RPT], int part,
                                           uint32_t &sb, uint32_t &nb)
{
    constexpr int NE = RPT * NW;
    static_assert(NE <= kWave, "one wavefront scans a sub-tile's counters");
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t rk[RPT];
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
        rk[u] = 0u;
        if (mine[u]) rk[u] = atomicAdd(&L.scnt[u][wv][((m[u] >> 8) & 15u) - (uint32_t)(part * NW)], 1u);
    }
    __syncthreads();
    {
        uint32_t v = 0;
        if (lane < NE) v = L.scnt[lane / NW][lane % NW][wv];
        const uint32_t inc = wave_incl_scan(v);
        if (lane < NE) L.scnt[lane / NW][lane % NW][wv] = inc - v;
        if (lane == kWave - 1) L.btot[wv] = inc;
    }
    __syncthreads();
    uint32_t ex;
    {
        const uint32_t t = lane < NW ? L.btot[lane] : 0u;
        ex = wave_incl_scan(t) - t; // lane b: first slot of sub-tile b
    }
    sb = (uint32_t)__shfl((int)ex, wv);
    nb = L.btot[wv];
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
        const uint32_t b = mine[u] ? ((m[u] >> 8) & 15u) - (uint32_t)(part * NW) : 0u;
        const uint32_t base = (uint32_t)__shfl((int)ex, (int)b);
        if (mine[u]) L.stage[base + L.scnt[u][wv][b] + rk[u]] = m[u];
    }
    __syncthreads();
}

// blockIdx -> (pair, part) with the parts of a pair on ONE XCD (blocks b and b + 8 share an XCD): the list is read from
// HBM once and from that XCD's L2 by the other parts.
template <int PARTS>
__device__ __forceinline__ bool pair_part_of_block(int pairs, int &g, int &part)
{
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    part = q % PARTS;
    g = (q / PARTS) * 8 + xcd;
    return g < pairs;
}

// ---- Temporal Active Focus ---------------------------------------------------------------------------------------------
// One workgroup = NW sub-tiles of one (sequence, tile) pair of a WINDOW-SORTED sequence (kf_scatter flags the others).
// The tile's list is streamed once, in stream order; wavefront v keeps the FIFO rows of its 256 cells (four per lane) in
// registers together with the running (sum, count) of the window it is in, and closes windows -- one FIFO step per cell,
// generate_taf.py:27-49 -- whenever its records move on to a later window.  Replaces kf_split_whole + kf_taf_walk for the
// tiles it takes: the sub-tile-major copy of the records never exists.
struct TafWaveLds {
    uint32_t cnt[kSubCells / 2];
    uint16_t off[kSubCells];
    float sorted[kSubCells];
}; // 2 KB; at the end the uint8 staging of the wavefront's sub-tile: 2K planes x 128 pixels

template <int NW, bool K8>
__global__ __launch_bounds__(NW *kWave) void kf_taf_tile(TileP q)
{
    constexpr int RPT = 4, NT = NW * kWave, CH = NT * RPT, PARTS = kFW / NW;
    __shared__ TileSplit<NW, RPT> L;
    __shared__ __attribute__((aligned(16))) TafWaveLds wl[NW];
    __shared__ uint32_t thr[kLeakyLevels];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int g, part;
    if (!pair_part_of_block<PARTS>(q.pairs, g, part)) return;
    if (q.hdr->status != 0) return;
    const int s = g / q.T, tile = g - s * q.T;
    if (q.hdr->unsorted[s] != 0u) return; // kf_split_whole + kf_taf_walk
    const uint32_t beg = q.base[g], end = q.base[g + 1];
    if (end - beg > q.tile_max) return;   // skewed tile: segment split + kf_taf_walk
    const int K = K8 ? 8 : q.K;
    const int NWIN = q.n_windows;
    for (int i = tid; i < kLeakyLevels; i += NT) thr[i] = q.leaky_thr[i];
    for (int i = lane; i < kSubCells / 2; i += kWave) wl[wv].cnt[i] = 0u;
    for (int i = tid; i < RPT * NW * NW; i += NT) (&L.scnt[0][0][0])[i] = 0u;
    const WavePass P = {wl[wv].cnt, wl[wv].off, wl[wv].sorted};
    const unsigned long long wmask = q.hdr->wmask[s];
    const bool use_mul = q.hdr->mul_bad == 0u; // checked for every r of the domain by kf_hist
    const double rcp = q.rcp;
    const uint32_t wfield = (1u << q.wb) - 1u;
    const int rshift = kCellBits + q.wb;
    // the lane's four cells: cell 64 j + lane of sub-tile `sub` = pixel 128 sub + 32 j + lane / 2 of the tile, polarity lane & 1
    const int sub = part * NW + wv;
    const int ty = tile / q.tiles_x, tx = tile - ty * q.tiles_x;
    const int x0 = tx << q.twl, y0 = ty << q.thl, tw1 = (1 << q.twl) - 1;
    const long long plane = (long long)q.H * q.W;
    const int pol = lane & 1;
    float st[4][kMaxK], sum[4];
    uint32_t num[4];
    bool ok[4];
    long long pix[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int pt = sub * (kSubCells / 2) + 32 * j + (lane >> 1);
        const int py = y0 + (pt >> q.twl), px = x0 + (pt & tw1);
        ok[j] = py < q.H && px < q.W;
        pix[j] = (long long)py * q.W + px;
        const float *srow = q.state + (((long long)s * plane + pix[j]) * 2 + pol) * K;
#pragma unroll
        for (int k = 0; k < kMaxK; ++k) st[j][k] = 0.0f;
        if (ok[j]) {
            if (K8) {
                const float4 a = ((const float4 *)srow)[0], b = ((const float4 *)srow)[1];
                st[j][0] = a.x; st[j][1] = a.y; st[j][2] = a.z; st[j][3] = a.w;
                st[j][4] = b.x; st[j][5] = b.y; st[j][6] = b.z; st[j][7] = b.w;
            } else {
#pragma unroll
                for (int k = 0; k < kMaxK; ++k)
                    if (k < K) st[j][k] = srow[k];
            }
        }
        sum[j] = 0.0f;
        num[j] = 0u;
    }
    int cur_w = 0; // wave-uniform: windows below it are closed for this wavefront's cells
    auto close_upto = [&](int w) { // FIFO steps of windows cur_w .. w - 1 (skipped when empty in the whole sequence, :40-41)
#pragma nounroll
        for (; cur_w < w; ++cur_w) {
            const bool has = (wmask >> cur_w) & 1ull;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                fifo_step(st[j], K, has, num[j], fifo_mean(num[j], sum[j]));
                sum[j] = 0.0f;
                num[j] = 0u;
            }
        }
    };
    uint32_t m[RPT], nx[RPT];
    // (loads without lane conditions -- a conditional load waits for its own data: indices are clamped, values masked)
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
        const uint32_t i = beg + (uint32_t)(u * NT + tid);
        nx[u] = 0xffffffffu;
        if (end > beg) { const uint32_t v = q.rec[i < end ? i : end - 1u]; nx[u] = i < end ? v : 0xffffffffu; }
    }
    __syncthreads();
    for (uint32_t c0 = beg; c0 < end; c0 += CH) {
        bool mine[RPT];
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            m[u] = nx[u];
            mine[u] = m[u] != 0xffffffffu && (int)((m[u] >> 8) & 15u) / NW == part;
            const uint32_t i = c0 + CH + (uint32_t)(u * NT + tid); // the next chunk's loads fly during this one's passes
            const uint32_t v = q.rec[i < end ? i : end - 1u];
            nx[u] = i < end ? v : 0xffffffffu;
        }
        uint32_t sb, nb;
        tile_split<NW, RPT>(L, m, mine, part, sb, nb);
        for (int i = tid; i < RPT * NW * NW; i += NT) (&L.scnt[0][0][0])[i] = 0u; // dead since the placement; next chunk's tickets
        for (uint32_t p0 = 0; p0 < nb; p0 += 256) {
            uint32_t pm[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t i = p0 + (uint32_t)(u * kWave + lane);
                pm[u] = i < nb ? L.stage[sb + i] : 0xffffffffu;
            }
            // the pass's records are window-sorted: first and last record give its window range
            const uint32_t last = nb - p0 < 256u ? nb - 1u : p0 + 255u;
            const int wlo = (int)((L.stage[sb + p0] >> kCellBits) & wfield), whi = (int)((L.stage[sb + last] >> kCellBits) & wfield);
#pragma nounroll
            for (int w = wlo; w <= whi; ++w) {
                close_upto(w);
                uint32_t sel[4], n[4], o[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    sel[u] = (pm[u] != 0xffffffffu && (int)((pm[u] >> kCellBits) & wfield) == w) ? pm[u] : 0xffffffffu;
                wave_sort(P, sel, lane,
                          [&](uint32_t rw) {
                              const uint32_t r = rw >> rshift; // t - 1 with t = (t - t_min) / (w + 1e-8) in f64 (generate_taf.py:215, :26)
                              return use_mul ? (float)((double)r * rcp) - 1.0f : q.tlut[r];
                          }, n, o);
                wave_segments(P, n, o, [&](int j, float v, bool live) {
                    const float t = sum[j] + v; // sum += t - 1 in stream order, generate_taf.py:26
                    sum[j] = live ? t : sum[j];
                });
#pragma unroll
                for (int j = 0; j < 4; ++j) num[j] += n[j];
                LDS_FENCE();
            }
        }
        __syncthreads(); // stage[] and scnt[] are reused by the next chunk
    }
    close_upto(NWIN);

    // ---- write-out: state, optional f32 view (2K, H, W), optional uint8 leaky transform (K, 2, H, W)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (!ok[j]) continue;
        float *srow = q.state + (((long long)s * plane + pix[j]) * 2 + pol) * K;
        if (K8) {
            ((float4 *)srow)[0] = make_float4(st[j][0], st[j][1], st[j][2], st[j][3]);
            ((float4 *)srow)[1] = make_float4(st[j][4], st[j][5], st[j][6], st[j][7]);
        } else {
#pragma unroll
            for (int k = 0; k < kMaxK; ++k)
                if (k < K) srow[k] = st[j][k];
        }
        if (q.view_f32) {
            float *vw = q.view_f32 + (long long)s * 2 * K * plane + pix[j];
#pragma unroll
            for (int k = 0; k < kMaxK; ++k)
                if (k < K) vw[(long long)(2 * k + pol) * plane] = st[j][k]; // generate_taf.py:55
        }
    }
    if (q.out_u8) {
        uint8_t *ob = (uint8_t *)&wl[wv]; // [2K planes][128 pixels of the sub-tile]; the wave's pass buffers are dead
        LDS_FENCE();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint8_t lv[kMaxK];
            leaky_u8_lookup_n<kMaxK>(st[j], thr, lv);
#pragma unroll
            for (int k = 0; k < kMaxK; ++k) {
                if (k < K) {
                    const int ko = q.flip ? (K - 1 - k) : k;
                    ob[(2 * ko + pol) * (kSubCells / 2) + 32 * j + (lane >> 1)] = lv[k];
                }
            }
        }
        LDS_FENCE();
        // the (K, 2, H, W) volume leaves plane by plane in 16-pixel pieces: one 16-byte store where the row allows
        for (int c = lane; c < 2 * K * 8; c += kWave) {
            const int pl = c >> 3, piece = c & 7;
            const int p16 = sub * (kSubCells / 2) + 16 * piece;
            const int y = y0 + (p16 >> q.twl), x = x0 + (p16 & tw1);
            if (y >= q.H || x >= q.W) continue;
            const uint8_t *src = ob + pl * (kSubCells / 2) + 16 * piece;
            uint8_t *dst = q.out_u8 + ((long long)s * 2 * K + pl) * plane + (long long)y * q.W + x;
            if (x + 16 <= q.W && (((uintptr_t)dst) & 15u) == 0) {
                *(uint4 *)dst = *(const uint4 *)src;
            } else {
                const int nv = q.W - x < 16 ? q.W - x : 16;
                for (int e = 0; e < nv; ++e) dst[e] = src[e];
            }
        }
    }
}

// ---- Event Volume ------------------------------------------------------------------------------------------------------
struct EvTileP {
    int H, W, twl, thl, tiles_x, T, bins;
    uint32_t win;
    double rcp;           // FastGeom::rcp
    const uint32_t *rec;  // tile-major records (scatter output)
    const uint32_t *rec2; // sub-tile-major (segment split), for the tiles the tile walk leaves alone
    const uint32_t *base; // [pairs + 1]
    const uint32_t *sub;  // [pairs * 16 + 1]
    const uint32_t *sub_end; // TileP::sub_end
    int pairs;
    int direct;           // TileP::direct
    uint32_t tile_max;    // tiles with more records go through the segment split + kf_ev_sub
    const float *tlut;    // tlut[r] = float(r / window)
    FastHeader *hdr;
    float *out_f32;       // (B, 2 * bins, H, W) or NULL
    uint8_t *out_u8;      // (B, 2 * bins, H, W) or NULL
};

// generate_eventvolume.py:23-32 for one event of normalised time tn on one cell: t* = bins * float(t); bin k (1-based)
// receives 1 - |k - t*| when that is not negative.  Only the two bins around t*, k0 = floor(t*) and k0 + 1, can: for the
// others |k - t*| >= 1 already before rounding, so their weight is zero or dropped and changes no sum.
template <int BINS>
__device__ __forceinline__ void ev_add(float (&acc)[BINS], float binsf, float tn, bool live)
{
    const float ts = binsf * tn;
#pragma unroll
    for (int k = 0; k < BINS; ++k) {
        const float d = (float)(k + 1) - ts;
        const float w = 1.0f - fabsf(d); // :28
        const float na = acc[k] + w;
        acc[k] = (live && w > 0.0f) ? na : acc[k]; // :29 (w == 0 adds nothing either)
    }
}

// One pass of up to 256 records of one sub-tile through the wave's accumulators.  Usual case (a pass is 256 consecutive
// records of one sub-tile of a time-sorted stream): floor(t*) = K0 is the same for every record -- then each record works
// out its own two weights (bins K0 and K0 + 1) once, and the owner lanes only ADD them in stream order.
template <int BINS>
__device__ __forceinline__ void ev_pass(const WavePass &P, const uint32_t (&pm)[4], int lane, const EvTileP &q, bool use_mul, double rcp,
                                        float binsf, float (&acc)[4][BINS])
{
    uint32_t n[4], o[4];
    float tn[4];
    bool differs = false;
    // (lane 0, u = 0 holds the pass's first record: a pass is never empty)
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)pm[0]) >> kCellBits;
    const int kfirst = (int)(binsf * (use_mul ? (float)((double)r0 * rcp) : q.tlut[r0]));
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        tn[u] = 0.0f;
        if (pm[u] != 0xffffffffu) {
            const uint32_t r = pm[u] >> kCellBits;
            tn[u] = use_mul ? (float)((double)r * rcp) : q.tlut[r]; // float((t - t0) / window), generate_eventvolume.py:141, :23
            differs |= (int)(binsf * tn[u]) != kfirst;
        }
    }
    const int k0 = __ballot(differs) ? -1 : kfirst;
    if (k0 >= 0 && k0 <= BINS) {
        // weights of the 1-based bins k0 (if >= 1) and k0 + 1 (if <= BINS), exactly as ev_add computes them; they are >= 0
        const bool lo_ok = k0 >= 1, hi_ok = k0 + 1 <= BINS;
        const float klo = (float)k0, khi = (float)(k0 + 1);
        wave_sort2(P, pm, lane,
                   [&](uint32_t, int u, float &a, float &b) {
                       const float ts = binsf * tn[u]; // t* = bins * float(t), :23
                       const float wl = 1.0f - fabsf(klo - ts), wh = 1.0f - fabsf(khi - ts); // :28
                       a = (lo_ok && wl > 0.0f) ? wl : 0.0f; // :29
                       b = (hi_ok && wh > 0.0f) ? wh : 0.0f;
                   }, n, o);
        switch (k0) {
#define EV_CASE(K) case K: wave_segments2(P, n, o, [&](int j, float a, float b) { \
            if (K >= 1 && K - 1 < BINS) acc[j][K >= 1 ? K - 1 : 0] += a; \
            if (K < BINS) acc[j][K < BINS ? K : 0] += b; }); break;
            EV_CASE(0) EV_CASE(1) EV_CASE(2) EV_CASE(3) EV_CASE(4) EV_CASE(5) EV_CASE(6) EV_CASE(7) EV_CASE(8)
#undef EV_CASE
        default: break;
        }
    } else {
        wave_sort(P, pm, lane, [&](uint32_t w) { const uint32_t r = w >> kCellBits; return use_mul ? (float)((double)r * rcp) : q.tlut[r]; }, n, o);
        wave_segments(P, n, o, [&](int j, float t, bool live) { ev_add<BINS>(acc[j], binsf, t, live); });
    }
    LDS_FENCE();
}

// the cells 64 j + lane of a sub-tile: scale (generate_eventvolume.py:37) and write both outputs
template <int BINS>
__device__ __forceinline__ void ev_store_cells(const EvTileP &q, int s, int tile, int sub, int lane, int j, const float (&acc)[BINS])
{
    const int ty = tile / q.tiles_x, tx = tile - ty * q.tiles_x;
    const int x0 = tx << q.twl, y0 = ty << q.thl, tw1 = (1 << q.twl) - 1;
    const long long plane = (long long)q.H * q.W;
    const int pol = lane & 1, ch = pol ? 0 : 1; // weights [p, 1 - p]: channel 0 = p == 1
    const int pt = sub * (kSubCells / 2) + 32 * j + (lane >> 1); // cell 64 j + lane = pixel 32 j + lane / 2, polarity lane & 1
    const int py = y0 + (pt >> q.twl), px = x0 + (pt & tw1);
    if (py >= q.H || px >= q.W) return;
#pragma unroll
    for (int k = 0; k < BINS; ++k) {
        if (k < q.bins) {
            const float v = acc[k] / 5.0f * 255.0f; // generate_eventvolume.py:37
            const long long idx = ((long long)s * 2 * q.bins + (2 * k + ch)) * plane + (long long)py * q.W + px;
            if (q.out_f32) q.out_f32[idx] = v;
            if (q.out_u8) q.out_u8[idx] = f32_to_u8(v > 255.0f ? 255.0f : v);
        }
    }
}

template <int BINS>
__device__ __forceinline__ void ev_store(const EvTileP &q, int s, int tile, int sub, int lane, const float (&acc)[4][BINS])
{
#pragma unroll
    for (int j = 0; j < 4; ++j) ev_store_cells<BINS>(q, s, tile, sub, lane, j, acc[j]);
}

// One workgroup = NW sub-tiles of one (sequence, tile) pair; see the section header.
template <int NW, int BINS>
__global__ __launch_bounds__(NW *kWave) void kf_ev_tile(EvTileP q)
{
    constexpr int RPT = 4, NT = NW * kWave, CH = NT * RPT, PARTS = kFW / NW;
    __shared__ TileSplit<NW, RPT> L;
    __shared__ uint32_t s_cnt[NW][kSubCells / 2];
    __shared__ uint16_t s_off[NW][kSubCells];
    __shared__ __attribute__((aligned(8))) float s_sorted[NW][2 * kSubCells + 2];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int g, part;
    if (!pair_part_of_block<PARTS>(q.pairs, g, part)) return;
    if (q.hdr->status != 0) return;
    const uint32_t beg = q.base[g], end = q.base[g + 1];
    if (end - beg > q.tile_max) return; // skewed tile: segment split + kf_ev_sub
    if (lane < 2) s_sorted[wv][2 * kSubCells + lane] = 0.0f; // the all-zero pair behind the segments
    const int s = g / q.T, tile = g - s * q.T;
    for (int i = lane; i < kSubCells / 2; i += kWave) s_cnt[wv][i] = 0u;
    for (int i = tid; i < RPT * NW * NW; i += NT) (&L.scnt[0][0][0])[i] = 0u;
    const WavePass P = {s_cnt[wv], s_off[wv], s_sorted[wv]};
    const bool use_mul = q.hdr->mul_bad == 0u;
    const double rcp = q.rcp;
    const float binsf = (float)q.bins;
    float acc[4][BINS];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < BINS; ++k) acc[j][k] = 0.0f;
    uint32_t m[RPT], nx[RPT];
    // (loads without lane conditions -- a conditional load waits for its own data: indices are clamped, values masked)
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
        const uint32_t i = beg + (uint32_t)(u * NT + tid);
        nx[u] = 0xffffffffu;
        if (end > beg) { const uint32_t v = q.rec[i < end ? i : end - 1u]; nx[u] = i < end ? v : 0xffffffffu; }
    }
    __syncthreads();
    for (uint32_t c0 = beg; c0 < end; c0 += CH) {
        bool mine[RPT];
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            m[u] = nx[u];
            mine[u] = m[u] != 0xffffffffu && (int)((m[u] >> 8) & 15u) / NW == part;
            const uint32_t i = c0 + CH + (uint32_t)(u * NT + tid); // the next chunk's loads fly during this one's passes
            const uint32_t v = q.rec[i < end ? i : end - 1u];
            nx[u] = i < end ? v : 0xffffffffu;
        }
        uint32_t sb, nb;
        tile_split<NW, RPT>(L, m, mine, part, sb, nb);
        for (int i = tid; i < RPT * NW * NW; i += NT) (&L.scnt[0][0][0])[i] = 0u; // dead since the placement; next chunk's tickets
        for (uint32_t p0 = 0; p0 < nb; p0 += 256) {
            uint32_t pm[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t i = p0 + (uint32_t)(u * kWave + lane);
                pm[u] = i < nb ? L.stage[sb + i] : 0xffffffffu;
            }
            ev_pass<BINS>(P, pm, lane, q, use_mul, rcp, binsf, acc);
        }
        __syncthreads(); // stage[] and scnt[] are reused by the next chunk
    }
    ev_store<BINS>(q, s, tile, part * NW + wv, lane, acc);
}

// After the segment split (kf_split_whole's counting blocks + kf_split_place): one wavefront per sub-tile walks its own
// contiguous list -- the skewed tiles of any call, and every tile of a call with few (sequence, tile) pairs.
// CMD (chunk-major partition, direct mode): the wavefront first gathers its sub-tile's runs from the chunks' stretches of rec[]
// -- its column of the directory, scanned 64 chunks at a time, then groups of 16 lanes copy a run each -- into LDS when the list
// fits (kEvListCap records), else into rec2[] at the header's cursor; everything after that is the walk over one contiguous list.
constexpr int kColEv = 511;       // chunks per sequence a wavefront's column holds
constexpr int kEvListCap = 2048;  // records of a sub-tile's list kept in LDS per wavefront
template <int BINS, bool CMD = false>
// (five wavefronts per SIMD where the registers allow it without spills -- the five-bin list walk, 102 -> 92 VGPRs: every
// wavefront is a latency chain of its own, one more of them per SIMD hides more of it)
__global__ __launch_bounds__(4 * kWave) __attribute__((amdgpu_waves_per_eu((BINS <= 5 && !CMD) ? 5 : 1, 8))) void kf_ev_sub(EvTileP q, int all_tiles, CmP cm, SeqTab S)
{
    __shared__ uint32_t s_cnt[4][kSubCells / 2];
    __shared__ uint16_t s_off[4][kSubCells];
    __shared__ __attribute__((aligned(8))) float s_sorted[4][2 * kSubCells + 2];
    __shared__ uint32_t s_colL[CMD ? 4 : 1][CMD ? kColEv + 1 : 1], s_colD[CMD ? 4 : 1][CMD ? kColEv : 1];
    __shared__ uint32_t s_list[CMD ? 4 : 1][CMD ? kEvListCap : 1];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (lane < 2) s_sorted[wv][2 * kSubCells + lane] = 0.0f; // the all-zero pair behind the segments
    const int sg = blockIdx.x * 4 + wv;
    if (sg >= q.pairs * kFW || q.hdr->status != 0) return;
    const int g = sg / kFW, sub = sg - g * kFW;
    if (!all_tiles && q.base[g + 1] - q.base[g] <= q.tile_max) return; // done by kf_ev_tile
    const int s = g / q.T, tile = g - s * q.T;
    for (int i = lane; i < kSubCells / 2; i += kWave) s_cnt[wv][i] = 0u;
    const WavePass P = {s_cnt[wv], s_off[wv], s_sorted[wv]};
    const bool use_mul = q.hdr->mul_bad == 0u;
    const double rcp = q.rcp;
    const float binsf = (float)q.bins;
    float acc[4][BINS];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < BINS; ++k) acc[j][k] = 0.0f;
    uint32_t beg, end;
    const uint32_t *list = q.rec2;
    if (CMD) {
        uint32_t *colL = s_colL[wv], *colD = s_colD[wv];
        const int c0 = S.chunk0[s], C = S.chunk0[s + 1] - c0, b = sg - s * cm.TB;
        const uint32_t out0 = (uint32_t)(S.ev0[s] - S.ev0[0]);
        uint32_t n = 0;
        for (int cb = 0; cb < C; cb += kWave) { // the column, 64 chunks per step (wave-uniform trip count)
            const int c = cb + lane;
            const uint32_t v = cm.dir[(long long)b * cm.n_chunks + (c0 + (c < C ? c : C - 1))];
            const uint32_t e = c < C ? v : 0u, cnt = e >> 16;
            const uint32_t inc = wave_incl_scan(cnt), run = n + inc - cnt;
            if (c < C) { colL[c] = run; colD[c] = out0 + (uint32_t)c * (uint32_t)cm.chunk_ev + (e & 0xffffu) - run; }
            n += (uint32_t)__builtin_amdgcn_readlane((int)inc, kWave - 1);
        }
        if (lane == 0) colL[C] = n;
        LDS_FENCE();
        uint32_t *dstl;
        if (n <= (uint32_t)kEvListCap) { // wave-uniform
            beg = 0u;
            dstl = s_list[wv];
            list = s_list[wv];
        } else {
            uint32_t at = 0;
            if (lane == 0) at = atomicAdd(&q.hdr->rec_cursor, n);
            beg = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
            dstl = const_cast<uint32_t *>(q.rec2) + beg;
        }
        end = beg + n;
        const int g16 = lane >> 4, l16 = lane & 15;
        constexpr int RU = 8;
        for (int cc0 = g16; cc0 < C; cc0 += 4 * RU) { // four groups of 16 lanes, eight runs each per step: the loads first, then the stores
            uint32_t v[RU], at[RU], cnt[RU];
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int c = cc0 + 4 * u, cc = c < C ? c : C - 1;
                const uint32_t lo = colL[cc];
                cnt[u] = c < C ? colL[cc + 1] - lo : 0u;
                at[u] = lo;
                v[u] = cm.rec[colD[cc] + lo + ((uint32_t)l16 < cnt[u] ? (uint32_t)l16 : 0u)];
            }
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                if ((uint32_t)l16 < cnt[u]) dstl[at[u] + l16] = v[u];
                if (cnt[u] > 16u) {
                    const uint32_t d = colD[cc0 + 4 * u];
                    for (uint32_t j = 16u + l16; j < cnt[u]; j += 16u) dstl[at[u] + j] = cm.rec[d + at[u] + j];
                }
            }
        }
        __threadfence_block(); // (the wavefront reads back what its own lanes wrote: LDS in order; rec2[] through the fence)
        LDS_FENCE();
    } else {
        // (sub[] of the NEXT pair is only written if that pair went through a split kernel: take the tile's own end)
        beg = q.sub[sg];
        end = q.sub_end ? q.sub_end[sg] : ((sub == kFW - 1 && !q.direct) ? q.base[g + 1] : q.sub[sg + 1]);
    }
    uint32_t nx[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t i = beg + (uint32_t)(u * kWave + lane);
        nx[u] = 0xffffffffu;
        if (end > beg) { const uint32_t v = list[i < end ? i : end - 1u]; nx[u] = i < end ? v : 0xffffffffu; }
    }
    LDS_FENCE();
    // A pass whose records all lie in ONE slice of the window (same floor(t*)) takes ev_pass's cheap form -- two weights per
    // record, pairs added in stream order -- a pass that straddles a slice boundary the general one (every bin tried for every
    // record).  A time-sorted list of ~1 700 records crosses the five boundaries in five of its seven 256-record passes; so a
    // pass is CUT at the first record of the next slice (the rest of its 256 records is fetched again by the next pass): twelve
    // cheap passes instead of two cheap and five general ones.  The cut uses an approximate slice index (float(r * bins) / window:
    // monotone in r); ev_pass still classifies exactly, so a record the approximation puts on the wrong side of a boundary only
    // costs that pass the general form.  Cuts in front of record 64 are not made (an unsorted list would otherwise crawl).
    const float inv_win = 1.0f / (float)q.win;
    const uint32_t ubins = (uint32_t)q.bins;
    for (uint32_t p0 = beg; p0 < end;) {
        uint32_t pm[4];
        uint32_t take = 256u;
        {
            const uint32_t k0 = (uint32_t)((float)(((uint32_t)__builtin_amdgcn_readfirstlane((int)nx[0]) >> kCellBits) * ubins) * inv_win);
#pragma unroll
            for (int u = 3; u >= 0; --u) { // (descending: the lowest u with a differing record wins)
                pm[u] = nx[u];
                const unsigned long long d = __ballot(pm[u] != 0xffffffffu && (uint32_t)((float)((pm[u] >> kCellBits) * ubins) * inv_win) != k0);
                if (d) take = (uint32_t)(u * kWave) + (uint32_t)__builtin_ctzll(d);
            }
            if (take < (uint32_t)kWave) take = 256u;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if ((uint32_t)(u * kWave + lane) >= take) pm[u] = 0xffffffffu;
            const uint32_t i = p0 + take + (uint32_t)(u * kWave + lane);
            const uint32_t v = list[i < end ? i : end - 1u];
            nx[u] = i < end ? v : 0xffffffffu;
        }
        ev_pass<BINS>(P, pm, lane, q, use_mul, rcp, binsf, acc);
        p0 += take;
    }
    ev_store<BINS>(q, s, tile, sub, lane, acc);
}

// Small direct-mode calls (ONE label window of a GEN1-shaped stream: 576 sub-tile lists of ~1 700 records): kf_ev_sub's ticket
// sort is built for throughput and leaves such a call on a latency chain of seven dependent passes per wavefront (30 us).  Here
// the sums are made by the LDS itself: ds_add_f32 applies the lanes of one instruction that hit one address in ascending lane
// order with the rounding of v_add_f32 (fact 2 of DESIGN.md 3.2; the library's self-test checks it on the first call and this
// kernel is only used where it held), a wavefront's instructions are served in program order -- so one wavefront that feeds its
// list through `acc[bin][cell] += weight` 64 records at a time makes exactly the reference's sequential sums
// (generate_eventvolume.py:28-32), without tickets, scans or segment walks.  It costs 192 cycles per instruction and CU (3 x the
// ticket scheme per record), which is why only small calls come here.  Four wavefronts share a sub-tile: they gather its list
// together, and then EACH walks the whole list but adds only into the bins k with k % 4 == its index -- every (cell, bin) sum
// stays one wavefront's chain in stream order, and the four chains of atomics run side by side.
// An event adds to the two bins around t* = bins * float(t): records of one instruction whose floor(t*) differ are issued run by
// run (equal floors, lane order), because the upper weight of an earlier record and the lower weight of a later one can meet in
// one bin -- a time-sorted stream has one run per instruction except at the five slice boundaries.
constexpr int kFaddWaves = 4; // wavefronts per sub-tile: they gather the list together, then wavefront w owns the bins k with k % 4 == w
template <int BINS>
__global__ __launch_bounds__(kFaddWaves *kWave) void kf_ev_fadd(EvTileP q, CmP cm, SeqTab S)
{
    constexpr int NT = kFaddWaves * kWave;
    __shared__ float s_acc[BINS][kSubCells];
    __shared__ uint32_t s_colL[kColEv + 1], s_colD[kColEv], s_wsum[kFaddWaves + 1];
    __shared__ uint32_t s_list[kEvListCap];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int sg = blockIdx.x;
    if (sg >= q.pairs * kFW || q.hdr->status != 0) return;
    const int g = sg / kFW, sub = sg - g * kFW;
    const int s = g / q.T, tile = g - s * q.T;
    const bool use_mul = q.hdr->mul_bad == 0u;
    const double rcp = q.rcp;
    const float binsf = (float)q.bins;
    for (int i = tid; i < BINS * kSubCells; i += NT) (&s_acc[0][0])[i] = 0.0f;
    // the list: this sub-tile's runs in the chunks' stretches of rec[] (the column of the directory, then groups of 16 lanes
    // copy a run each: what kf_ev_sub<BINS, true> does with one wavefront, here with four)
    const int c0 = S.chunk0[s], C = S.chunk0[s + 1] - c0, b = sg - s * cm.TB;
    const uint32_t out0 = (uint32_t)(S.ev0[s] - S.ev0[0]);
    uint32_t n = 0;
    for (int cb = 0; cb < C; cb += NT) { // (workgroup-uniform trip count; C <= kColEv)
        const int c = cb + tid;
        const uint32_t v = cm.dir[(long long)b * cm.n_chunks + (c0 + (c < C ? c : C - 1))];
        const uint32_t e = c < C ? v : 0u, cnt = e >> 16;
        const uint32_t inc = wave_incl_scan(cnt);
        if (lane == kWave - 1) s_wsum[wv] = inc;
        __syncthreads();
        uint32_t pre = 0, all = 0;
#pragma unroll
        for (int k = 0; k < kFaddWaves; ++k) { if (k < wv) pre += s_wsum[k]; all += s_wsum[k]; }
        const uint32_t run = n + pre + inc - cnt;
        if (c < C) { s_colL[c] = run; s_colD[c] = out0 + (uint32_t)c * (uint32_t)cm.chunk_ev + (e & 0xffffu) - run; }
        n += all;
        __syncthreads();
    }
    if (tid == 0) s_colL[C] = n;
    uint32_t beg = 0u;
    uint32_t *dstl = s_list;
    const uint32_t *list = s_list;
    if (n > (uint32_t)kEvListCap) { // (workgroup-uniform) a list too long for LDS goes through rec2[]
        if (tid == 0) s_wsum[kFaddWaves] = atomicAdd(&q.hdr->rec_cursor, n);
        __syncthreads();
        beg = s_wsum[kFaddWaves];
        dstl = const_cast<uint32_t *>(q.rec2) + beg;
        list = q.rec2;
    }
    __syncthreads();
    const uint32_t end = beg + n;
    {
        const int g16 = tid >> 4, l16 = tid & 15;
        constexpr int RU = 10, NG = NT / 16;
        for (int cc0 = g16; cc0 < C; cc0 += NG * RU) { // groups of 16 lanes, ten runs each per step (one step for the 144 chunks of a
                                                       // GEN1 stream): the loads first, then the stores
            uint32_t v[RU], at[RU], cnt[RU];
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int c = cc0 + NG * u, cc = c < C ? c : C - 1;
                const uint32_t lo = s_colL[cc];
                cnt[u] = c < C ? s_colL[cc + 1] - lo : 0u;
                at[u] = lo;
                v[u] = cm.rec[s_colD[cc] + lo + ((uint32_t)l16 < cnt[u] ? (uint32_t)l16 : 0u)];
            }
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                if ((uint32_t)l16 < cnt[u]) dstl[at[u] + l16] = v[u];
                if (cnt[u] > 16u) {
                    const uint32_t d = s_colD[cc0 + NG * u];
                    for (uint32_t j = 16u + l16; j < cnt[u]; j += 16u) dstl[at[u] + j] = cm.rec[d + at[u] + j];
                }
            }
        }
    }
    __syncthreads(); // the list is complete (LDS, or rec2[] written and read on this CU)
    // every wavefront walks the whole list in stream order and adds into ITS bins only: a (cell, bin) sum is one wavefront's chain
    float *accl = &s_acc[0][0];
    for (uint32_t p0 = beg; p0 < end; p0 += 4 * kWave) {
        uint32_t pm[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { // (four instructions' records in flight; clamped index, masked below)
            const uint32_t i = p0 + (uint32_t)(u * kWave + lane);
            pm[u] = list[i < end ? i : end - 1u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool valid = p0 + (uint32_t)(u * kWave + lane) < end;
            const uint32_t r = pm[u] >> kCellBits, cell = pm[u] & 255u;
            const float tn = use_mul ? (float)((double)r * rcp) : q.tlut[r < q.win ? r : q.win]; // float((t - t0) / window), :141, :23
            const float ts = binsf * tn;                       // t* = bins * float(t), :23
            const int k0 = (int)ts;                            // floor (t* >= 0): the 1-based bins k0 and k0 + 1 can receive
            const float wl = 1.0f - fabsf((float)k0 - ts), wh = 1.0f - fabsf((float)(k0 + 1) - ts); // :28
            // :29 (a weight of zero adds nothing either); 0-based bin k0 - 1 takes wl, bin k0 takes wh
            const bool lo_ok = valid && k0 >= 1 && k0 <= q.bins && wl > 0.0f && ((k0 - 1) & (kFaddWaves - 1)) == wv;
            const bool hi_ok = valid && k0 + 1 <= q.bins && wh > 0.0f && (k0 & (kFaddWaves - 1)) == wv;
            if (__ballot(lo_ok || hi_ok) == 0ull) continue; // (none of this instruction's records touches my bins)
            const int kv = valid ? k0 : -1;
            const int kprev = __shfl_up(kv, 1);
            unsigned long long starts = __ballot(lane == 0 || kv != kprev); // runs of equal floor(t*), in lane order
            while (starts) {
                const int rb = __builtin_ctzll(starts);
                starts &= starts - 1ull;
                const int re = starts ? __builtin_ctzll(starts) : kWave;
                const bool in = lane >= rb && lane < re;
                if (in && lo_ok) atomicAdd(&accl[(k0 - 1) * kSubCells + (int)cell], wl);
                if (in && hi_ok) atomicAdd(&accl[k0 * kSubCells + (int)cell], wh);
            }
        }
    }
    __syncthreads();
    float acc[BINS];
#pragma unroll
    for (int k = 0; k < BINS; ++k) acc[k] = s_acc[k][64 * wv + lane];
    ev_store_cells<BINS>(q, s, tile, sub, lane, wv, acc);
}

// ---- Surface of Active Events through the chunk-major partition (small single calls) ------------------------------------
// generate_leaky_cuda (generate_surfaceofactiveevents.py:44-80): t_img[p, y, x] = float(t) of the cell's LAST event in stream
// order, max with the memory, exp(lambda (t_img - now)) * 255.  The general path takes five launches (41 us for 1 M events at
// 304x240).  Here: kf_scatter_cm<.., SAE> writes records {position in the sequence << 12 | cell} chunk-major, and one workgroup
// per sub-tile takes the maximum record per cell with LDS atomics -- straight from the runs, no list, no order needed -- reads
// the time of that one event from the DAT array and writes memory and outputs with the arithmetic of k_sae_tile (encoders.hip).
struct SaeFastP {
    int H, W, twl, thl, tiles_x, T, n_lamda;
    float lam[FRLW_MAX_LAMDAS > 21 ? FRLW_MAX_LAMDAS : 21]; // (ECI: the count -> value table)
    float nowf;
    const uint2 *data;
    const float *mem_in;
    float *mem_out, *out_f32;
    uint8_t *out_u8;
    FastHeader *hdr;
};

// ECI (template flag): the same walk over the runs COUNTS the records per cell instead; the image is the 21-entry table of
// n sequential +0.05f adds, clamped and scaled (generate_eventcountimage.py:32-41; q.lam[] carries the table, n_lamda = 21).
template <bool ECI>
__global__ __launch_bounds__(kSubCells) void kf_sae_sub(SaeFastP q, CmP cm, SeqTab S)
{
    __shared__ uint32_t s_last[kSubCells];
    __shared__ uint32_t s_colL[kColEv + 1], s_colD[kColEv], s_wsum[kSubCells / kWave + 1];
    constexpr int NT = kSubCells, NWV = NT / kWave;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int sg = blockIdx.x, tile = sg / kFW, sub = sg - tile * kFW; // (one sequence)
    if (q.hdr->status != 0) return;
    s_last[tid] = 0u;
    const int C = S.chunk0[1] - S.chunk0[0];
    uint32_t n = 0;
    for (int cb = 0; cb < C; cb += NT) { // the sub-tile's column of the directory (workgroup-uniform trip count; C <= kColEv)
        const int c = cb + tid;
        const uint32_t v = cm.dir[(long long)sg * cm.n_chunks + (c < C ? c : C - 1)];
        const uint32_t e = c < C ? v : 0u, cnt = e >> 16;
        const uint32_t inc = wave_incl_scan(cnt);
        if (lane == kWave - 1) s_wsum[wv] = inc;
        __syncthreads();
        uint32_t pre = 0, all = 0;
#pragma unroll
        for (int k = 0; k < NWV; ++k) { if (k < wv) pre += s_wsum[k]; all += s_wsum[k]; }
        const uint32_t run = n + pre + inc - cnt;
        if (c < C) { s_colL[c] = run; s_colD[c] = (uint32_t)c * (uint32_t)cm.chunk_ev + (e & 0xffffu) - run; }
        n += all;
        __syncthreads();
    }
    if (tid == 0) s_colL[C] = n;
    __syncthreads();
    {   // the runs: groups of 16 lanes take a run each, ten runs' loads in flight; the later record of a cell wins (position in the high bits)
        const int g16 = tid >> 4, l16 = tid & 15;
        constexpr int RU = 10, NG = NT / 16;
        for (int cc0 = g16; cc0 < C; cc0 += NG * RU) {
            uint32_t v[RU], cnt[RU];
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int c = cc0 + NG * u, cc = c < C ? c : C - 1;
                const uint32_t lo = s_colL[cc];
                cnt[u] = c < C ? s_colL[cc + 1] - lo : 0u;
                v[u] = cm.rec[s_colD[cc] + lo + ((uint32_t)l16 < cnt[u] ? (uint32_t)l16 : 0u)];
            }
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                if ((uint32_t)l16 < cnt[u]) { if (ECI) atomicAdd(&s_last[v[u] & 255u], 1u); else atomicMax(&s_last[v[u] & 255u], v[u]); }
                if (cnt[u] > 16u) {
                    const int c = cc0 + NG * u;
                    const uint32_t d = s_colD[c], lo = s_colL[c];
                    for (uint32_t j = 16u + l16; j < cnt[u]; j += 16u) {
                        const uint32_t w = cm.rec[d + lo + j];
                        if (ECI) atomicAdd(&s_last[w & 255u], 1u); else atomicMax(&s_last[w & 255u], w);
                    }
                }
            }
        }
    }
    __syncthreads();
    // cell tid: pixel 128 sub + tid / 2 of the tile, polarity tid & 1
    const int ty = tile / q.tiles_x, tx = tile - ty * q.tiles_x;
    const int x0 = tx << q.twl, y0 = ty << q.thl, tw1 = (1 << q.twl) - 1;
    const int pol = tid & 1, pt = sub * (kSubCells / 2) + (tid >> 1);
    const int py = y0 + (pt >> q.twl), px = x0 + (pt & tw1);
    if (py >= q.H || px >= q.W) return;
    const long long plane = (long long)q.H * q.W, idx = (long long)pol * plane + (long long)py * q.W + px;
    const uint32_t w = s_last[tid];
    if (ECI) {
        const float v = q.lam[w > 20u ? 20u : w];
        if (q.out_f32) q.out_f32[idx] = v;
        if (q.out_u8) q.out_u8[idx] = f32_to_u8(v);
        return;
    }
    const float init = (0.0f + q.nowf) - 5000000.0f; // generate_surfaceofactiveevents.py:48
    // (the scatter stores position + 1: a record is never 0, 0 = the cell has no event)
    float tv = w ? (float)q.data[S.ev0[0] + (long long)(w >> kCellBits) - 1].x : init; // float(t), :76
    if (q.mem_in) {
        const float m = q.mem_in[idx];
        if (!(tv > m)) tv = m; // torch.where(t_img > memory, t_img, memory), :52
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

// Self-test of the two hardware properties this file rests on, for lanes of ONE wave-instruction that hit the same LDS
// address: (1) a returning integer atomic serves them in ascending lane order (the returned count is the stream rank);
// (2) ds_add_f32 applies them in ascending lane order with the rounding of v_add_f32, i.e. it IS the sequential
// `sum = sum + v` of the lanes.  Random addresses and values in (-1, 0]; reference by counting / adding over lower lanes.
__global__ __launch_bounds__(kFT) void kf_selftest_lane_order(int n_addr, int iters, unsigned long long *out)
{
    extern __shared__ uint32_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t *mine = lds + (size_t)wv * 3 * n_addr;
    float *facc = (float *)(mine + n_addr);  // accumulated by the atomic
    float *fref = facc + n_addr;             // accumulated sequentially, lane by lane
    unsigned long long bad = 0, conf = 0, fbad = 0;
    for (int a = lane; a < n_addr; a += kWave) { facc[a] = 0.0f; fref[a] = 0.0f; }
    for (int it = 0; it < iters; ++it) {
        for (int a = lane; a < n_addr; a += kWave) mine[a] = 0;
        LDS_FENCE();
        uint32_t h = (blockIdx.x * (uint32_t)kFT + tid) * 2654435761u + it * 40503u;
        h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
        const uint32_t addr = h % (uint32_t)n_addr;
        const float v = (float)((double)(h >> 8 & 16383u) / 10000.00000001) - 1.0f; // the shape of the TAF values
        const uint32_t got = atomicAdd(&mine[addr], 1u);
        atomicAdd(&facc[addr], v);
        uint32_t want = 0;
        for (int l = 0; l < kWave; ++l) {
            const uint32_t other = __shfl(addr, l);
            const float ov = __shfl(v, l);
            if (l < lane && other == addr) ++want;
            LDS_FENCE();
            if (l == lane) ((volatile float *)fref)[addr] = ((volatile float *)fref)[addr] + ov;
            LDS_FENCE();
        }
        if (got != want) ++bad;
        if (want) ++conf;
        LDS_FENCE();
        if (__float_as_uint(((volatile float *)facc)[addr]) != __float_as_uint(((volatile float *)fref)[addr])) ++fbad;
        LDS_FENCE();
        if ((it & 7) == 7) // let the sums grow over 8 batches, then start again
            for (int a = lane; a < n_addr; a += kWave) { facc[a] = 0.0f; fref[a] = 0.0f; }
    }
    if (bad) atomicAdd(&out[0], bad);
    if (conf) atomicAdd(&out[1], conf);
    if (fbad) atomicAdd(&out[2], fbad);
}

template <bool HAS_MAP, bool EV = false>
void launch_fast(const FastGeom &G, const SeqTab &S, const FastPlan &p, char *w8, hipStream_t st)
{
    FastHeader *hdr = (FastHeader *)w8;
    uint32_t *counts = (uint32_t *)(w8 + p.off_counts);
    uint32_t *slabtot = (uint32_t *)(w8 + p.off_slabtot);
    uint32_t *base = (uint32_t *)(w8 + p.off_base);
    int32_t *errs = (int32_t *)(w8 + p.off_errs);
    float *tlut = (float *)(w8 + p.off_tlut);
    uint32_t *records = (uint32_t *)(w8 + p.off_records);
    const size_t lds_sc = scatter_lds_bytes(p.TB, p.chunk);
    if (lds_sc > 64 * 1024) {
        (void)hipFuncSetAttribute((const void *)kf_scatter<HAS_MAP, EV, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sc);
        if (!EV) (void)hipFuncSetAttribute((const void *)kf_scatter<HAS_MAP, EV, !EV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sc);
    }
    const int hist_grid = p.chunks < 512 ? p.chunks : 512; // persistent: two workgroups per CU
    const bool simple = !HAS_MAP && G.simple != 0 && !(!EV && G.order_check);
    if (simple) {
        if (lds_sc > 64 * 1024)
            (void)hipFuncSetAttribute((const void *)kf_scatter<false, EV, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sc);
        hipLaunchKernelGGL((kf_hist<false, EV, true>), dim3(hist_grid), dim3(kFT), (size_t)p.TB * 4, st, G, S, counts, errs, tlut, p.chunks);
    } else {
        hipLaunchKernelGGL((kf_hist<HAS_MAP, EV>), dim3(hist_grid), dim3(kFT), (size_t)p.TB * 4, st, G, S, counts, errs, tlut, p.chunks);
    }
    const bool inline_slabs = (long long)p.slabs * p.TB <= kInlineSlabScan;
    if (!inline_slabs)
        hipLaunchKernelGGL(kf_slabscan, dim3((p.TB + kWave - 1) / kWave, p.slabs), dim3(kWave), 0, st, S, counts, p.TB, slabtot);
    hipLaunchKernelGGL(kf_tilescan, dim3(1), dim3(kFT), 0, st, S, slabtot, p.TB, base, (uint32_t *)(w8 + p.off_seg0), hdr, errs,
                       p.chunks, inline_slabs ? counts : (uint32_t *)nullptr, p.slabs, p.direct);
    if (simple)
        hipLaunchKernelGGL((kf_scatter<false, EV, false, true>), dim3(p.chunks), dim3(kFT), lds_sc, st, G, S, counts, slabtot, base, records, hdr);
    else if (!EV && G.order_check)
        hipLaunchKernelGGL((kf_scatter<HAS_MAP, EV, !EV>), dim3(p.chunks), dim3(kFT), lds_sc, st, G, S, counts, slabtot, base, records, hdr);
    else
        hipLaunchKernelGGL((kf_scatter<HAS_MAP, EV, false>), dim3(p.chunks), dim3(kFT), lds_sc, st, G, S, counts, slabtot, base, records, hdr);
}

// the captured form's header reset (status, window masks, cursors): see launch_fast_cm
__global__ __launch_bounds__(256) void kf_header_reset(FastHeader *hdr)
{
    uint32_t *h32 = (uint32_t *)hdr;
    for (int i = threadIdx.x; i < (int)(offsetof(FastHeader, epoch) / 4); i += 256) h32[i] = 0u;
    if (threadIdx.x == 0) *(uint32_t *)((char *)hdr + kStallOffset) = 0u; // (the captured form has no wait and no stall of its own)
}

// the chunk-major partition: ONE kernel (its first workgroup resets the header; a reset kernel in front of it inside a capture)
template <bool HAS_MAP, bool EV = false>
int launch_fast_cm(const FastGeom &G, const SeqTab &S, const FastPlan &p, char *w8, hipStream_t st)
{
    FastHeader *hdr = (FastHeader *)w8;
    uint32_t *dir = (uint32_t *)(w8 + p.off_counts);
    float *tlut = (float *)(w8 + p.off_tlut);
    uint32_t *records = (uint32_t *)(w8 + p.off_records);
    const size_t lds_sc = scatter_cm_lds_bytes(p.TB, p.chunk);
    static std::atomic<uint32_t> g_epoch{0};
    uint32_t epoch = 0u;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &cap);
    if (cap != hipStreamCaptureStatusNone) {
        // inside a stream capture the kernel arguments are frozen into the graph node: a host-made epoch would already be the
        // published one on every replay after the first (workgroups could then OR their flags into the header BEFORE workgroup 0
        // zeroes it).  A captured call resets the header with a kernel node of its own and passes epoch 0 = "nobody resets, nobody waits".
        // (a kernel node, not hipMemsetAsync: a captured memset node of this runtime wrote a stale pattern into the header from its
        // second replay on -- measured; tests/test_taf_fast_gpu.py replays a captured encode three times)
        hipLaunchKernelGGL(kf_header_reset, dim3(1), dim3(256), 0, st, hdr);
    } else {
        epoch = g_epoch.fetch_add(1u, std::memory_order_relaxed) + 1u;
        if (epoch == 0u) epoch = g_epoch.fetch_add(1u, std::memory_order_relaxed) + 1u; // (0 = the captured form)
    }
    const bool simple = !HAS_MAP && G.simple != 0;
    if (p.big && simple) {
        (void)hipFuncSetAttribute((const void *)kf_scatter_cm<false, EV, true, kBigBpw>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sc);
        hipLaunchKernelGGL((kf_scatter_cm<false, EV, true, kBigBpw>), dim3(p.chunks), dim3(kFT), lds_sc, st, G, S, dir, records, hdr, tlut, epoch);
    } else if (p.big) {
        (void)hipFuncSetAttribute((const void *)kf_scatter_cm<HAS_MAP, EV, false, kBigBpw>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sc);
        hipLaunchKernelGGL((kf_scatter_cm<HAS_MAP, EV, false, kBigBpw>), dim3(p.chunks), dim3(kFT), lds_sc, st, G, S, dir, records, hdr, tlut, epoch);
    } else if (simple) {
        if (lds_sc > 64 * 1024)
            (void)hipFuncSetAttribute((const void *)kf_scatter_cm<false, EV, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sc);
        hipLaunchKernelGGL((kf_scatter_cm<false, EV, true>), dim3(p.chunks), dim3(kFT), lds_sc, st, G, S, dir, records, hdr, tlut, epoch);
    } else {
        if (lds_sc > 64 * 1024)
            (void)hipFuncSetAttribute((const void *)kf_scatter_cm<HAS_MAP, EV, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sc);
        hipLaunchKernelGGL((kf_scatter_cm<HAS_MAP, EV, false>), dim3(p.chunks), dim3(kFT), lds_sc, st, G, S, dir, records, hdr, tlut, epoch);
    }
    return FRLW_OK;
}

// chunk-major partition or histogram partition?  frlw_tuning_t::chunk_major = 0 forces the histogram partition; otherwise the
// chunk-major one runs wherever a consumer can hold a sequence's column of the directory in LDS (cm_fits) and nobody asked for
// the tile walk, which wants contiguous tile lists.  Measured (us, chunk-major against histogram partition): one
// GEN1 stream 32 / 42, 5 x 70 k events on 97x131 34 / 55, 3 M events at 1280x720 111 / 133, 10 M events 164 / 177, 64 GEN1
// streams 808 / 861, Event Volume x64 736 / 787; the skewed variants give some of it back (25 % of 10 M events in one blob:
// 298 / 287 -- the split segments of a skewed tile are only known after the split kernel, so their counting pass is a launch
// of its own).
enum : int { CM_OFF = 0, CM_AUTO = -1, CM_ON = 1 };
inline bool cm_fits(const FastPlan &p, bool ev)
{
    const int col = p.direct ? (ev ? kColEv : kColDirect) : kColMax;
    return p.max_seq_chunks <= col && p.chunk <= 65535;
}

// plan + layout of one call: tries the chunk-major plan first where the knob allows it
inline int plan_call(const frlw_tuning_t *tu, bool ev, bool tile_walk_wanted, long long n, int n_seq, int H, int W,
                     const int64_t *seq_offsets, const int64_t *t0, uint32_t win, FastPlan &p, SeqTab &S, bool &cm)
{
    const int knob = tuning_knob(tu, &frlw_tuning_t::chunk_major, CM_AUTO);
    const int bpw = tuning_knob(tu, &frlw_tuning_t::batches_per_wave, 0);
    const int dmode = [&] {
        const int direct = tuning_knob(tu, &frlw_tuning_t::direct_bins, -1);
        if (direct >= 0) return direct != 0 ? (int)DIRECT_FORCE : (int)DIRECT_OFF;
        return tile_walk_wanted ? (int)DIRECT_OFF : (int)DIRECT_AUTO;
    }();
    cm = false;
    if (knob != CM_OFF && !tile_walk_wanted) {
        if (!fast_plan(n, n_seq, H, W, p, dmode, bpw, true)) return FRLW_ERR_UNSUPPORTED;
        if (!fast_layout(seq_offsets, t0, n_seq, p, S, win)) return FRLW_ERR_ARG;
        cm = cm_fits(p, ev);
        if (cm) return FRLW_OK;
    }
    if (!fast_plan(n, n_seq, H, W, p, dmode, bpw, false)) return FRLW_ERR_UNSUPPORTED;
    if (!fast_layout(seq_offsets, t0, n_seq, p, S, win)) return FRLW_ERR_ARG;
    return FRLW_OK;
}

inline CmP cm_params(const FastPlan &p, char *w8)
{
    CmP cm;
    cm.dir = (const uint32_t *)(w8 + p.off_counts);
    cm.rec = (const uint32_t *)(w8 + p.off_records);
    cm.TB = p.TB;
    cm.n_chunks = p.chunks;
    cm.chunk_ev = p.chunk;
    cm.hot_start = (uint32_t *)(w8 + p.off_base);
    cm.hot_seg0 = (uint32_t *)(w8 + p.off_seg0);
    cm.segdesc = (uint32_t *)(w8 + p.off_segdesc);
    cm.max_segs = p.max_segs;
    return cm;
}

// second level of the chunk-major partition: q.rec2 / q.sub / q.sub_end afterwards describe one contiguous list per sub-tile
inline void launch_split_cm(TileP &q, const FastPlan &p, const SeqTab &S, char *w8, hipStream_t st)
{
    const CmP cm = cm_params(p, w8);
    q.rec2 = (uint32_t *)(w8 + p.off_records2);
    q.sub = (uint32_t *)(w8 + p.off_sub);
    q.sub_end = (uint32_t *)(w8 + p.off_sub_end);
    // (direct mode never comes here: its consumers gather their own lists)
    const int seg_grid = p.max_segs < 512 ? p.max_segs : 512; // (they stride over the segments; most calls have none; two workgroups per CU)
    hipLaunchKernelGGL(kf_split_whole<true>, dim3(p.pairs), dim3(kFT), 0, st, q, cm, S);
    hipLaunchKernelGGL(kf_segcount_cm, dim3(seg_grid), dim3(kFT), 0, st, q, cm, S);
    hipLaunchKernelGGL(kf_split_place<true>, dim3(seg_grid), dim3(kFT), 0, st, q, cm, S);
}

// ---- one-time check of the hardware property this file rests on ------------------------------------------------------
// 0 = not yet tested on this device, 1 = lanes of one returning LDS atomic are served in lane order, 2 = they are not
// (a new stepping / compiler): the fast path then refuses (FRLW_ERR_UNSUPPORTED) and callers take the general path.
constexpr int kMaxDevices = 64;
std::atomic<int> g_lds_order[kMaxDevices];
std::atomic<int> g_lds_fadd[kMaxDevices]; // 1: ds_add_f32 made the sequential f32 sums in the same self-test run (kf_ev_fadd may be used)

int lds_order_ok(char *w8, hipStream_t st)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return FRLW_ERR_HIP;
    int have = g_lds_order[dev].load(std::memory_order_acquire);
    if (have == 0) {
        // first fast-path call on this device: 256 workgroups x 64 rounds of conflicting lanes (~3e5 conflicting lane
        // pairs), result into the spare bytes of the workspace header, ONE host synchronisation for the process lifetime
        unsigned long long *out = (unsigned long long *)(w8 + kSelftestOffset);
        unsigned long long host[3] = {1ull, 0ull, 0ull};
        (void)hipFuncSetAttribute((const void *)kf_selftest_lane_order, hipFuncAttributeMaxDynamicSharedMemorySize, kFW * 512 * 12);
        HIP_TRY(hipMemsetAsync(out, 0, 24, st));
        hipLaunchKernelGGL(kf_selftest_lane_order, dim3(256), dim3(kFT), (size_t)kFW * 40 * 12, st, 40, 64, out);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(host, out, 24, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        have = (host[0] == 0ull && host[1] > 0ull) ? 1 : 2; // no rank violation among > 0 conflicting pairs
        g_lds_fadd[dev].store((host[2] == 0ull && host[1] > 0ull) ? 1 : 0, std::memory_order_release);
        g_lds_order[dev].store(have, std::memory_order_release);
    }
    return have == 1 ? FRLW_OK : FRLW_ERR_UNSUPPORTED;
}

} // namespace

extern "C" {

#if defined(FRLW_WALK_PROF) || defined(FRLW_SCAT_PROF)
int frlw_debug_walk_prof(unsigned long long *out16)
{
    static unsigned long long host[kProfWgs * 9];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_walk_prof), sizeof(host)) != hipSuccess) return FRLW_ERR_HIP;
    for (int i = 0; i < 16; ++i) out16[i] = 0;
    for (int w = 0; w < kProfWgs; ++w) {
        if (host[w * 9 + 8] == 0 && host[w * 9] == 0) continue;
        ++out16[15];
        for (int i = 0; i < 9; ++i) out16[i] += host[w * 9 + i];
    }
    void *dev = nullptr;
    if (hipGetSymbolAddress(&dev, HIP_SYMBOL(g_walk_prof)) != hipSuccess || hipMemset(dev, 0, sizeof(host)) != hipSuccess) return FRLW_ERR_HIP;
    return FRLW_OK;
}
#endif

#ifdef FRLW_DEV_BUILD // not in the product library: a process-wide switch has no place in its ABI
int frlw_debug_force_lds_order(int value)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return FRLW_ERR_HIP;
    if (value < -1 || value > 1) return FRLW_ERR_ARG;
    g_lds_order[dev].store(value < 0 ? 0 : (value == 1 ? 1 : 2), std::memory_order_release);
    return FRLW_OK;
}
#endif

int frlw_fast_path_verdict(void *workspace, size_t workspace_bytes, frlw_stream_t stream, int *ok_out)
{
    if (!workspace || workspace_bytes < kHeaderBytes || !ok_out) return FRLW_ERR_ARG;
    (void)hipGetLastError();
    const int rc = lds_order_ok((char *)workspace, (hipStream_t)stream); // runs the self-test on the first call per device, cached afterwards
    if (rc != FRLW_OK && rc != FRLW_ERR_UNSUPPORTED) return rc;
    *ok_out = rc == FRLW_OK ? 1 : 0;
    return FRLW_OK;
}

int frlw_selftest_lds_atomic_order(int n_addr, int iters, unsigned long long *out_dev, frlw_stream_t stream)
{
    if (!out_dev || n_addr < 1 || n_addr > 512 || iters < 1) return FRLW_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    (void)hipFuncSetAttribute((const void *)kf_selftest_lane_order, hipFuncAttributeMaxDynamicSharedMemorySize,
                              kFW * 512 * 12);
    HIP_TRY(hipMemsetAsync(out_dev, 0, 24, st));
    hipLaunchKernelGGL(kf_selftest_lane_order, dim3(256), dim3(kFT), (size_t)kFW * n_addr * 12, st, n_addr, iters, out_dev);
    HIP_TRY(hipGetLastError());
    return FRLW_OK;
}

size_t frlw_taf_batch_workspace_bytes(int64_t n_events, int n_seq, int H, int W, int64_t window_us)
{
    if (window_us < 1 || window_us >= (1ll << 20)) return 0;
    size_t need = 0;
    for (int mode = 0; mode < 4; ++mode) { // the largest of the partition modes (the call's tuning and size pick one)
        const int direct = mode & 1, cm = mode >> 1;
        FastPlan p;
        if (!fast_plan(n_events, n_seq, H, W, p, direct ? DIRECT_FORCE : DIRECT_OFF, 0, cm != 0)) return 0;
        if (direct && !p.direct) continue;
        // the layout depends on how the events are spread over the sequences only through the chunk count: every sequence
        // can add one partly filled chunk and one partly filled slab
        const size_t chunks = (size_t)(n_events + p.chunk - 1) / p.chunk + n_seq;
        const size_t slabs = chunks / kFastSlab + n_seq + 1;
        size_t off = kHeaderBytes;
        off = align_up(off + chunks * p.TB * 4, 256);
        off = align_up(off + slabs * p.TB * 4, 256);
        off = align_up(off + (size_t)(p.pairs_b + 1) * 4, 256);
        off = align_up(off + ((size_t)p.pairs * kFW + 1) * 4, 256);
        off = align_up(off + (size_t)(p.pairs_b + 1) * 4, 256);
        off = align_up(off + (2 * (size_t)(n_events / kSplitSeg) + 1) * kFW * 4, 256);
        off = align_up(off + chunks * 4, 256);
        off = align_up(off + (size_t)(window_us + 1) * 4, 256);
        off = align_up(off + (size_t)(n_events > 0 ? n_events : 1) * 4, 256);
        off = align_up(off + (size_t)(n_events > 0 ? n_events : 1) * 4, 256);
        off = align_up(off + ((size_t)p.pairs * kFW + 1) * 4, 256);
        off = align_up(off + (2 * (size_t)(n_events / kSplitSeg) + 1) * 4, 256);
        off = align_up(off + (size_t)p.pairs * kFW * (FRLW_MAX_WINDOWS + 1) * 4, 256);
        off = align_up(off + (size_t)p.pairs * 4, 256);
        if (off > need) need = off;
    }
    return need;
}

} // extern "C"

namespace {
enum : int { PHASE_PARTITION = 1, PHASE_FINISH = 2 };

// The batch encode in two halves: PARTITION = kf_hist, scans, kf_scatter (leaves the per-sequence window masks in the
// workspace header), FINISH = split + walk (reads them).  A row stripe [y_lo, y_lo + H) of an H_full-row frame runs the two
// halves as separate calls with an OR-reduce of the masks over the stripes in between (frlw_taf_stripe_*).
int taf_batch_run(int phases, const frlw_events_t *ev, const int64_t *seq_offsets, const int64_t *t_start, int n_seq, int H, int W,
                  int y_lo, int H_full, int K, int64_t window_us, int n_windows, float *state, float *view_f32, uint8_t *out_u8,
                  int flags, void *workspace, size_t workspace_bytes, frlw_stream_t stream)
{
    if (!ev || !seq_offsets || !t_start || !workspace || (!state && (phases & PHASE_FINISH))) return FRLW_ERR_ARG;
    if (y_lo < 0 || H < 1 || y_lo + H > H_full) return FRLW_ERR_ARG;
    if (K < 1 || K > FRLW_MAX_BINS || n_windows < 1 || n_windows > FRLW_MAX_WINDOWS || window_us < 1) return FRLW_ERR_ARG;
    if (n_seq < 1 || n_seq > kMaxSeq) return FRLW_ERR_ARG;
    if (ev->layout != FRLW_LAYOUT_DAT8) return FRLW_ERR_UNSUPPORTED;
    if ((ev->xmap == nullptr) != (ev->ymap == nullptr)) return FRLW_ERR_ARG;
    if (!tuning_valid(ev->tuning)) return FRLW_ERR_ARG;
    if (seq_offsets[0] < 0 || seq_offsets[n_seq] > ev->n) return FRLW_ERR_ARG;
    if (seq_offsets[n_seq] > seq_offsets[0] && !ev->data) return FRLW_ERR_ARG;
    // record = r | window | cell in 32 bits
    int wb = 0, rb = 0;
    while ((1 << wb) < n_windows) ++wb;
    while ((1ll << rb) <= window_us) ++rb;
    if (kCellBits + wb + rb > 32 || (long long)n_windows * window_us >= (1ll << 32)) return FRLW_ERR_UNSUPPORTED;
    const long long n = seq_offsets[n_seq] - seq_offsets[0];
    FastPlan p;
    SeqTab S;
    bool cm = false;
    const bool walk_wanted = tuning_knob(ev->tuning, &frlw_tuning_t::taf_tile_walk, kTafTileWalk ? 1 : 0) != 0 &&
                             tuning_knob(ev->tuning, &frlw_tuning_t::direct_bins, -1) <= 0;
    {
        const int rc = plan_call(ev->tuning, false, walk_wanted, n, n_seq, H, W, seq_offsets, t_start, (uint32_t)window_us, p, S, cm);
        if (rc != FRLW_OK) return rc;
    }
    if (workspace_bytes < p.bytes) return FRLW_ERR_WORKSPACE;
    if (scatter_lds_bytes(p.TB, p.chunk) > 160 * 1024) return FRLW_ERR_UNSUPPORTED;

    FastGeom G;
    G.data = (const uint2 *)ev->data;
    G.xmap = ev->xmap; G.ymap = ev->ymap; G.map_w = ev->map_w; G.map_h = ev->map_h;
    G.H = H; G.W = W; G.twl = p.twl; G.thl = p.thl; G.tiles_x = p.tiles_x; G.T = p.TB; G.bin_shift = p.bin_shift; G.bin_mask = p.direct ? 15 : 0; G.bpw = p.bpw;
    G.chunk_ev = p.chunk; G.run = p.chunk / kFW; G.n_total = ev->n;
    G.n_windows = n_windows; G.wb = wb; G.win = (uint32_t)window_us;
    G.rcp = 1.0 / ((double)(uint32_t)window_us + 1e-8); // generate_taf.py:215: t / (w + 1e-8)
    G.y_lo = y_lo; G.H_full = H_full;
    {
        const bool want = tuning_knob(ev->tuning, &frlw_tuning_t::taf_tile_walk, kTafTileWalk ? 1 : 0) != 0;
        G.order_check = (want && !p.direct && p.pairs >= kFewPairs) ? 1 : 0;
    }
    const unsigned long long magic = (1ull << 32) / (unsigned long long)window_us;
    G.win_magic = magic > 0xffffffffull ? 0xffffffffu : (uint32_t)magic;
    {
        bool simple = y_lo == 0 && H_full == H && window_us >= 2 &&
                      (unsigned long long)n_windows * (unsigned long long)window_us <= 0xffffffffull;
        for (int s = 0; s < n_seq && simple; ++s) simple = S.t0[s] >= 0 && S.t0[s] <= 0xffffffffll;
        G.simple = simple ? 1 : 0;
        G.span = simple ? (uint32_t)((unsigned long long)n_windows * (unsigned long long)window_us) : 0u;
    }

    hipStream_t st = (hipStream_t)stream;
    char *w8 = (char *)workspace;
    (void)hipGetLastError();
    {
        const int ok = lds_order_ok(w8, st); // cached per device after the first call
        if (ok != FRLW_OK) return ok;
    }
    if (phases & PHASE_PARTITION) {
        if (cm) {
            const int rc = ev->xmap ? launch_fast_cm<true>(G, S, p, w8, st) : launch_fast_cm<false>(G, S, p, w8, st);
            if (rc != FRLW_OK) return rc;
        } else if (ev->xmap) launch_fast<true>(G, S, p, w8, st);
        else launch_fast<false>(G, S, p, w8, st);
    }
    if (!(phases & PHASE_FINISH)) { HIP_TRY(hipGetLastError()); return FRLW_OK; }
    TileP q;
    q.H = H; q.W = W; q.twl = p.twl; q.thl = p.thl; q.tiles_x = p.tiles_x; q.T = p.T; q.K = K; q.n_windows = n_windows;
    q.wb = wb; q.flip = (flags & FRLW_TAF_U8_FLIP_K) ? 1 : 0; q.win = (uint32_t)window_us; q.rcp = G.rcp;
    q.rec = (const uint32_t *)(w8 + p.off_records);
    q.rec2 = (uint32_t *)(w8 + p.off_records2);
    q.base = (const uint32_t *)(w8 + p.off_base);
    q.sub = (uint32_t *)(w8 + p.off_sub);
    q.seg0 = (const uint32_t *)(w8 + p.off_seg0);
    q.segcnt = (uint32_t *)(w8 + p.off_segcnt);
    q.pairs = p.pairs;
    q.skip_whole = 0;
    q.first_block = 0;
    q.seg_grid = p.max_segs < 2048 ? p.max_segs : 2048;
    q.tile_walk = G.order_check;
    q.tile_max = whole_max_of(p.pairs);
    q.tlut = (const float *)(w8 + p.off_tlut);
    if (!(q.leaky_thr = leaky_table(st))) return FRLW_ERR_HIP; // device-resident constant, built once per device (partition.hip)
    q.hdr = (FastHeader *)w8;
    q.state = state; q.view_f32 = view_f32; q.out_u8 = out_u8;
    q.direct = p.direct;
    q.sub_end = nullptr;
    q.wst = nullptr; q.wst_flag = nullptr;
    if (cm && p.direct) {
        q.rec2 = (uint32_t *)(w8 + p.off_records2); // (kf_taf_walk<.., true> books and fills its own list)
    } else if (cm) {
        if (tuning_knob(ev->tuning, &frlw_tuning_t::walk_window_table, 1) != 0) { // kf_split_whole<true> leaves the walk its window starts
            q.wst = (uint32_t *)(w8 + p.off_wst);
            q.wst_flag = (uint32_t *)(w8 + p.off_wst_flag);
        }
        launch_split_cm(q, p, S, w8, st);
    } else if (p.direct) { // kf_scatter's bins were the sub-tiles: its output IS the sub-tile-major list, base[] its sub[]
        q.rec2 = (uint32_t *)(w8 + p.off_records);
        q.sub = (uint32_t *)(w8 + p.off_base);
    } else {
        const CmP none = {};
        hipLaunchKernelGGL(kf_split_whole<false>, dim3(p.pairs + q.seg_grid), dim3(kFT), 0, st, q, none, S); // tiles, then segment counts
        hipLaunchKernelGGL(kf_split_place<false>, dim3(q.seg_grid), dim3(kFT), 0, st, q, none, S);
    }
    if (q.tile_walk) { // tiles of window-sorted sequences below the skew limit: split in LDS by the kernel that consumes them
        const int grid = (p.pairs + 7) / 8 * 8;
        if (K == 8) hipLaunchKernelGGL((kf_taf_tile<kFW, true>), dim3(grid), dim3(kFT), 0, st, q);
        else hipLaunchKernelGGL((kf_taf_tile<kFW, false>), dim3(grid), dim3(kFT), 0, st, q);
    }
    const CmP cmq = cm ? cm_params(p, w8) : CmP{};
    if (cm && p.direct) { // the walk gathers its own list (no gather kernel)
        if (K == 8) hipLaunchKernelGGL((kf_taf_walk<true, true>), dim3(p.pairs * kFW), dim3(kWalkThreads), 0, st, q, cmq, S);
        else hipLaunchKernelGGL((kf_taf_walk<false, true>), dim3(p.pairs * kFW), dim3(kWalkThreads), 0, st, q, cmq, S);
    } else if (K == 8) hipLaunchKernelGGL((kf_taf_walk<true, false>), dim3(p.pairs * kFW), dim3(kWalkThreads), 0, st, q, cmq, S);
    else hipLaunchKernelGGL((kf_taf_walk<false, false>), dim3(p.pairs * kFW), dim3(kWalkThreads), 0, st, q, cmq, S);
    HIP_TRY(hipGetLastError());
    return FRLW_OK;
}
} // namespace

extern "C" {

int frlw_taf_encode_batch(const frlw_events_t *ev, const int64_t *seq_offsets, const int64_t *t_start, int n_seq, int H,
                          int W, int K, int64_t window_us, int n_windows, float *state, float *view_f32, uint8_t *out_u8,
                          int flags, void *workspace, size_t workspace_bytes, frlw_stream_t stream)
{
    return taf_batch_run(PHASE_PARTITION | PHASE_FINISH, ev, seq_offsets, t_start, n_seq, H, W, 0, H, K, window_us, n_windows, state,
                         view_f32, out_u8, flags, workspace, workspace_bytes, stream);
}

int frlw_taf_stripe_partition(const frlw_events_t *ev, const int64_t *seq_offsets, const int64_t *t_start, int n_seq, int H_full,
                              int W, int y_lo, int rows, int K, int64_t window_us, int n_windows, void *workspace,
                              size_t workspace_bytes, frlw_stream_t stream)
{
    return taf_batch_run(PHASE_PARTITION, ev, seq_offsets, t_start, n_seq, rows, W, y_lo, H_full, K, window_us, n_windows, nullptr,
                         nullptr, nullptr, 0, workspace, workspace_bytes, stream);
}

unsigned long long *frlw_taf_stripe_window_masks(void *workspace)
{
    return workspace ? ((FastHeader *)workspace)->wmask : nullptr;
}

int frlw_taf_stripe_finish(const frlw_events_t *ev, const int64_t *seq_offsets, const int64_t *t_start, int n_seq, int H_full, int W,
                           int y_lo, int rows, int K, int64_t window_us, int n_windows, float *state, float *view_f32,
                           uint8_t *out_u8, int flags, void *workspace, size_t workspace_bytes, frlw_stream_t stream)
{
    return taf_batch_run(PHASE_FINISH, ev, seq_offsets, t_start, n_seq, rows, W, y_lo, H_full, K, window_us, n_windows, state, view_f32,
                         out_u8, flags, workspace, workspace_bytes, stream);
}

size_t frlw_ev_batch_workspace_bytes(int64_t n_events, int n_seq, int H, int W, int64_t window_us)
{
    return frlw_taf_batch_workspace_bytes(n_events, n_seq, H, W, window_us); // same partition, same tables
}

int frlw_ev_encode_batch(const frlw_events_t *ev, const int64_t *seq_offsets, const int64_t *t_end, int n_seq, int H, int W,
                         int bins, int64_t window_us, float *out_f32, uint8_t *out_u8, void *workspace,
                         size_t workspace_bytes, frlw_stream_t stream)
{
    if (!ev || !seq_offsets || !t_end || !workspace || (!out_f32 && !out_u8)) return FRLW_ERR_ARG;
    if (bins < 1 || bins > FRLW_MAX_BINS || window_us < 1 || n_seq < 1 || n_seq > kMaxSeq) return FRLW_ERR_ARG;
    if (ev->layout != FRLW_LAYOUT_DAT8) return FRLW_ERR_UNSUPPORTED;
    if ((ev->xmap == nullptr) != (ev->ymap == nullptr)) return FRLW_ERR_ARG;
    if (!tuning_valid(ev->tuning)) return FRLW_ERR_ARG;
    if (seq_offsets[0] < 0 || seq_offsets[n_seq] > ev->n) return FRLW_ERR_ARG;
    if (seq_offsets[n_seq] > seq_offsets[0] && !ev->data) return FRLW_ERR_ARG;
    int rb = 0; // record = (t - t_begin) | cell in 32 bits
    while ((1ll << rb) <= window_us) ++rb;
    if (kCellBits + rb > 32) return FRLW_ERR_UNSUPPORTED;
    const long long n = seq_offsets[n_seq] - seq_offsets[0];
    FastPlan p;
    SeqTab S;
    int64_t t_begin[kMaxSeq];
    for (int s = 0; s < n_seq; ++s) t_begin[s] = t_end[s] - window_us; // generate_eventvolume.py:139-141
    bool cm = false;
    const bool walk_wanted = tuning_knob(ev->tuning, &frlw_tuning_t::taf_tile_walk, kTafTileWalk ? 1 : 0) != 0 &&
                             tuning_knob(ev->tuning, &frlw_tuning_t::direct_bins, -1) <= 0;
    {
        const int rc = plan_call(ev->tuning, true, walk_wanted, n, n_seq, H, W, seq_offsets, t_begin, (uint32_t)window_us, p, S, cm);
        if (rc != FRLW_OK) return rc;
    }
    if (workspace_bytes < p.bytes) return FRLW_ERR_WORKSPACE;
    if (scatter_lds_bytes(p.TB, p.chunk) > 160 * 1024) return FRLW_ERR_UNSUPPORTED;

    FastGeom G;
    G.data = (const uint2 *)ev->data;
    G.xmap = ev->xmap; G.ymap = ev->ymap; G.map_w = ev->map_w; G.map_h = ev->map_h;
    G.H = H; G.W = W; G.twl = p.twl; G.thl = p.thl; G.tiles_x = p.tiles_x; G.T = p.TB; G.bin_shift = p.bin_shift; G.bin_mask = p.direct ? 15 : 0; G.bpw = p.bpw;
    G.chunk_ev = p.chunk; G.run = p.chunk / kFW; G.n_total = ev->n;
    G.n_windows = 1; G.wb = 0; G.win = (uint32_t)window_us; G.win_magic = 0u; G.order_check = 0; G.y_lo = 0; G.H_full = H;
    G.rcp = 1.0 / (double)(uint32_t)window_us; // generate_eventvolume.py:141
    {
        bool simple = true; // (t0 = t_end - window is negative for a label in the first `window` microseconds of a file)
        for (int s = 0; s < n_seq && simple; ++s) simple = S.t0[s] >= 0 && S.t0[s] <= 0xffffffffll;
        G.simple = simple ? 1 : 0;
        G.span = 0u;
    }

    hipStream_t st = (hipStream_t)stream;
    char *w8 = (char *)workspace;
    (void)hipGetLastError();
    {
        const int ok = lds_order_ok(w8, st); // cached per device after the first call
        if (ok != FRLW_OK) return ok;
    }
    if (cm) {
        const int rc = ev->xmap ? launch_fast_cm<true, true>(G, S, p, w8, st) : launch_fast_cm<false, true>(G, S, p, w8, st);
        if (rc != FRLW_OK) return rc;
    } else if (ev->xmap) launch_fast<true, true>(G, S, p, w8, st);
    else launch_fast<false, true>(G, S, p, w8, st);
    // Second-level split: kf_split_whole + kf_ev_sub by default; the tile walk (kf_ev_tile, 3.4: the split in LDS, 2.44x instead
    // of 2.8x HBM traffic) on request -- it was the default until kf_split_whole became a one-pass kernel, which made the two-kernel
    // form the faster one (64 x 1 M events: 985 against 1 041 us)
    bool tile_walk = false;
    {
        const bool want = tuning_knob(ev->tuning, &frlw_tuning_t::taf_tile_walk, kTafTileWalk ? 1 : 0) != 0;
        tile_walk = want && !p.direct && p.pairs >= kFewPairs;
    }
    TileP q;
    memset(&q, 0, sizeof(q));
    q.T = p.T; q.pairs = p.pairs; q.skip_whole = tile_walk ? 1 : 0;
    q.rec = (const uint32_t *)(w8 + p.off_records);
    q.rec2 = (uint32_t *)(w8 + p.off_records2);
    q.base = (const uint32_t *)(w8 + p.off_base);
    q.sub = (uint32_t *)(w8 + p.off_sub);
    q.seg0 = (const uint32_t *)(w8 + p.off_seg0);
    q.segcnt = (uint32_t *)(w8 + p.off_segcnt);
    q.hdr = (FastHeader *)w8;
    q.first_block = tile_walk ? p.pairs : 0; // with the tile walk only the segment-counting blocks have work
    q.seg_grid = p.max_segs < 2048 ? p.max_segs : 2048;
    q.tile_max = whole_max_of(p.pairs);
    if (cm && p.direct) {
        q.rec2 = (uint32_t *)(w8 + p.off_records2); // (kf_ev_sub<.., true> books and fills its own lists)
    } else if (cm) {
        launch_split_cm(q, p, S, w8, st);
    } else if (p.direct) { // kf_scatter's bins were the sub-tiles: its output IS the sub-tile-major list, base[] its sub[]
        q.rec2 = (uint32_t *)(w8 + p.off_records);
        q.sub = (uint32_t *)(w8 + p.off_base);
    } else {
        const CmP none = {};
        hipLaunchKernelGGL(kf_split_whole<false>, dim3(p.pairs + q.seg_grid - q.first_block), dim3(kFT), 0, st, q, none, S); // tiles, then segment counts
        hipLaunchKernelGGL(kf_split_place<false>, dim3(q.seg_grid), dim3(kFT), 0, st, q, none, S);
    }
    EvTileP e;
    e.H = H; e.W = W; e.twl = p.twl; e.thl = p.thl; e.tiles_x = p.tiles_x; e.T = p.T; e.bins = bins; e.win = (uint32_t)window_us; e.rcp = G.rcp;
    e.rec = q.rec; e.rec2 = q.rec2; e.base = q.base; e.sub = q.sub; e.sub_end = q.sub_end; e.pairs = p.pairs; e.tile_max = whole_max_of(p.pairs); e.direct = p.direct;
    e.tlut = (const float *)(w8 + p.off_tlut); e.hdr = (FastHeader *)w8; e.out_f32 = out_f32; e.out_u8 = out_u8;
    const int sub_grid = (p.pairs * kFW + 3) / 4;
    if (tile_walk) {
#ifndef FRLW_EV_NW
#define FRLW_EV_NW 16
#endif
        constexpr int NW = FRLW_EV_NW, PARTS = kFW / NW;
        const int grid = (p.pairs + 7) / 8 * 8 * PARTS;
        if (bins <= 5) hipLaunchKernelGGL((kf_ev_tile<NW, 5>), dim3(grid), dim3(NW * kWave), 0, st, e);
        else hipLaunchKernelGGL((kf_ev_tile<NW, kMaxK>), dim3(grid), dim3(NW * kWave), 0, st, e);
    }
    const CmP cmq = cm ? cm_params(p, w8) : CmP{};
    bool fadd = false;
    {
        // kf_ev_fadd: calls whose lists are few and short enough that the LDS float atomics' 192 cycles per instruction and CU
        // stay below the ticket kernel's latency chain (measured: 1 M events in 576 lists 25 against 35 us, the whole call 40 against 50; with plain stores instead of the atomics the kernel takes 19 us: gather and decode are most of it; the ticket kernel
        // wins from about 3 M events on), on devices where the self-test saw ds_add_f32 make the sequential sums
        int dev = 0;
        const bool hw_ok = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < kMaxDevices && g_lds_fadd[dev].load(std::memory_order_acquire) == 1;
        const int knob = tuning_knob(ev->tuning, &frlw_tuning_t::ev_lds_float_atomics, -1);
        fadd = cm && p.direct && hw_ok && (knob >= 0 ? knob != 0 : n <= 3000000ll);
    }
    if (fadd) {
        if (bins <= 5) hipLaunchKernelGGL((kf_ev_fadd<5>), dim3(p.pairs * kFW), dim3(kFaddWaves * kWave), 0, st, e, cmq, S);
        else hipLaunchKernelGGL((kf_ev_fadd<kMaxK>), dim3(p.pairs * kFW), dim3(kFaddWaves * kWave), 0, st, e, cmq, S);
    } else if (cm && p.direct) { // the sub-tile wavefronts gather their own lists
        if (bins <= 5) hipLaunchKernelGGL((kf_ev_sub<5, true>), dim3(sub_grid), dim3(4 * kWave), 0, st, e, 1, cmq, S);
        else hipLaunchKernelGGL((kf_ev_sub<kMaxK, true>), dim3(sub_grid), dim3(4 * kWave), 0, st, e, 1, cmq, S);
    } else if (bins <= 5) hipLaunchKernelGGL((kf_ev_sub<5, false>), dim3(sub_grid), dim3(4 * kWave), 0, st, e, tile_walk ? 0 : 1, cmq, S);
    else hipLaunchKernelGGL((kf_ev_sub<kMaxK, false>), dim3(sub_grid), dim3(4 * kWave), 0, st, e, tile_walk ? 0 : 1, cmq, S);
    HIP_TRY(hipGetLastError());
    return FRLW_OK;
}

} // extern "C"

namespace frlw {
// The plan of the two-launch form for one call of n events on an H x W frame, or false when the call is not eligible (shared by
// sae_fast_try and by frlw_encoder_workspace_bytes: the size query must cover what the call will ask for).
static bool sae_fast_plan(long long n, int H, int W, long long t0v, FastPlan &p, SeqTab &S)
{
    // positions + 1 must fit the 20 bits above the 12-bit cell; tiny calls gain nothing
    if (n < 16384 || n >= (1ll << 20) - 1) return false;
    const int64_t offs[2] = {0, (int64_t)n};
    const int64_t t0[1] = {(int64_t)t0v};
    if (!fast_plan(n, 1, H, W, p, DIRECT_FORCE, 0, true) || !p.direct) return false; // frames of at most 64 tiles (the 304x240 class)
    {   // at least ~64 chunks: the plan's largest-chunk rule (one GEN1 stream of 1 M events: 144 chunks) would leave a
        // 100 000-event call with 15 scatter workgroups on 256 CUs
        long long ce = ((n + 63) / 64 + 15) / 16 * 16;
        if (ce < 1024) ce = 1024;
        if (ce < p.chunk) { p.chunk = (int)ce; p.bpw = (p.chunk / kFW + kWave - 1) / kWave; }
    }
    if (!fast_layout(offs, t0, 1, p, S, 1u)) return false;
    if (p.max_seq_chunks > kColEv || p.chunk > 65535 || p.big) return false;
    if (scatter_cm_lds_bytes(p.TB, p.chunk) > 160 * 1024) return false;
    return true;
}

size_t sae_fast_workspace_bytes(long long n, int H, int W)
{
    FastPlan p;
    SeqTab S;
    return sae_fast_plan(n, H, W, 0, p, S) ? p.bytes : 0;
}

std::atomic<unsigned long long> g_path_counts[4]; // [0] SAE two-launch, [1] SAE general, [2] ECI two-launch, [3] ECI general / scan

// frlw_sae_encode's two-launch form (see kf_sae_sub).  Returns FRLW_OK when it has launched the encode, 1 when the call is not
// eligible (nothing launched: the caller takes the general path), a negative FRLW_ERR_* on a HIP failure.
int sae_fast_try(const frlw_events_t *ev, int H, int W, const float *lam, int n_lamda, const float *mem_in, float *mem_out,
                 long long now, long long window_us, float *out_f32, uint8_t *out_u8, void *workspace, size_t workspace_bytes,
                 hipStream_t st)
{
    // (mem_out == nullptr: the Event Count Image -- lam[0 .. 20] is its count -> value table, no time filter)
    const bool eci = mem_out == nullptr;
    if (eci) { now = -1; window_us = 0; }
    if (!ev || ev->layout != FRLW_LAYOUT_DAT8 || !ev->data || !workspace || (!eci && window_us <= 0)) return 1;
    if ((ev->xmap == nullptr) != (ev->ymap == nullptr) || !tuning_valid(ev->tuning)) return 1;
    const long long n = ev->n;
    if (tuning_knob(ev->tuning, &frlw_tuning_t::staged_scatter, -1) == 0) return 1; // staged_scatter = 0 keeps the general path (tests)
    FastPlan p;
    SeqTab S;
    if (!sae_fast_plan(n, H, W, now - window_us, p, S)) return 1;
    if (workspace_bytes < p.bytes) return 1; // (frlw_encoder_workspace_bytes covers p.bytes: only a caller that sized the workspace itself gets here)
    FastGeom G;
    G.data = (const uint2 *)ev->data;
    G.xmap = ev->xmap; G.ymap = ev->ymap; G.map_w = ev->map_w; G.map_h = ev->map_h;
    G.H = H; G.W = W; G.twl = p.twl; G.thl = p.thl; G.tiles_x = p.tiles_x; G.T = p.TB; G.bin_shift = p.bin_shift; G.bin_mask = 15; G.bpw = p.bpw;
    G.chunk_ev = p.chunk; G.run = p.chunk / kFW; G.n_total = n;
    G.n_windows = 1; G.wb = 0; G.win = 0xffffffffu; G.win_magic = 0u; G.order_check = 0; G.y_lo = 0; G.H_full = H;
    G.rcp = 0.0;
    G.simple = (S.t0[0] >= 0 && S.t0[0] <= 0xffffffffll) ? 1 : 0;
    G.span = 0u;
    char *w8 = (char *)workspace;
    FastHeader *hdr = (FastHeader *)w8;
    uint32_t *dir = (uint32_t *)(w8 + p.off_counts);
    uint32_t *records = (uint32_t *)(w8 + p.off_records);
    const size_t lds_sc = scatter_cm_lds_bytes(p.TB, p.chunk);
    (void)hipGetLastError();
    static std::atomic<uint32_t> g_epoch_sae{0x40000000u}; // (its own range: never equal to a TAF / Event Volume call's epoch of the same process ... within 2^30 calls)
    uint32_t epoch = 0u;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &cap);
    if (cap != hipStreamCaptureStatusNone) hipLaunchKernelGGL(kf_header_reset, dim3(1), dim3(256), 0, st, hdr);
    else epoch = g_epoch_sae.fetch_add(1u, std::memory_order_relaxed) | 0x40000000u;
#define SAE_SCATTER(MAP, SIMPLE_, MODE) do { \
        if (lds_sc > 64 * 1024) (void)hipFuncSetAttribute((const void *)kf_scatter_cm<MAP, true, SIMPLE_, kMaxBpw, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sc); \
        hipLaunchKernelGGL((kf_scatter_cm<MAP, true, SIMPLE_, kMaxBpw, MODE>), dim3(p.chunks), dim3(kFT), lds_sc, st, G, S, dir, records, hdr, (float *)nullptr, epoch); } while (0)
    if (eci) { // (t0 = -1 is outside the SIMPLE decode's range: the general form keeps every event)
        if (ev->xmap) SAE_SCATTER(true, false, 2);
        else SAE_SCATTER(false, false, 2);
    } else if (ev->xmap) SAE_SCATTER(true, false, 1);
    else if (G.simple) SAE_SCATTER(false, true, 1);
    else SAE_SCATTER(false, false, 1);
#undef SAE_SCATTER
    SaeFastP q;
    q.H = H; q.W = W; q.twl = p.twl; q.thl = p.thl; q.tiles_x = p.tiles_x; q.T = p.T; q.n_lamda = n_lamda;
    for (int l = 0; l < (eci ? 21 : n_lamda); ++l) q.lam[l] = lam[l];
    q.nowf = (float)now;
    q.data = (const uint2 *)ev->data;
    q.mem_in = mem_in; q.mem_out = mem_out; q.out_f32 = out_f32; q.out_u8 = out_u8; q.hdr = hdr;
    const CmP cmq = cm_params(p, w8);
    if (eci) hipLaunchKernelGGL(kf_sae_sub<true>, dim3(p.pairs * kFW), dim3(kSubCells), 0, st, q, cmq, S);
    else hipLaunchKernelGGL(kf_sae_sub<false>, dim3(p.pairs * kFW), dim3(kSubCells), 0, st, q, cmq, S);
    HIP_TRY(hipGetLastError());
    g_path_counts[eci ? 2 : 0].fetch_add(1ull, std::memory_order_relaxed);
    return FRLW_OK;
}
} // namespace frlw
