// partition.hip -- stable tile partition of an event stream (steps 1-3 of every encoder).
//
// The reference's accumulators (torch index_add_, generate_eventvolume.py:32, generate_taf.py:24-26)
// are defined by the single-thread result: f32 adds in STREAM ORDER.  To reproduce that on a GPU the
// events are first partitioned by tile ((1 << twl) x 8 pixels) with a STABLE counting sort, so that
// every tile's records are still in stream order and one workgroup can own the tile's accumulators.
//
//   k_hist      one workgroup (16 wavefronts) per contiguous chunk of 1024*bpw events; LDS histogram
//               over tiles -> counts[wg][tile]          (events read once, bpw loads in flight / lane)
//   k_slabscan  exclusive prefix over the 32 workgroups of a slab, in place + slabtot[slab][tile]
//   k_tilescan  exclusive prefix over slabs (in place) and over tiles      -> base[tile]
//   k_scatter   same chunks again.  Wavefront w of the workgroup owns the w-th run of 64*bpw events and
//               walks it 64 events at a time in stream order; the rank of an event among the events
//               of its batch that fall in the same tile comes from an LDS tag round (write the lane
//               id, read it back: a mismatch marks a collision) plus one ballot per colliding tile.
//               Per-wave running counts live in LDS; after a workgroup barrier they are prefix-summed
//               over the 16 waves, and the whole workgroup writes its 8-byte records in ONE burst, so
//               each tile's run (~chunk / n_tiles records) is completed in the L2 while it is hot.

#include "frlw_common.h"

#include <mutex>

namespace frlw {

int hip_fail(hipError_t e, const char *what, int line)
{
#ifdef FRLW_DEV_BUILD // developer build only: the product library prints nothing and reads no environment
    fprintf(stderr, "frlw_evd: %s failed at line %d: %s\n", what, line, hipGetErrorString(e));
#else
    (void)e; (void)what; (void)line;
#endif
    return FRLW_ERR_HIP;
}

namespace {

// ---------------------------------------------------------------------------------------------
template <int LAYOUT, int KIND, bool HAS_MAP>
__global__ __launch_bounds__(kPartThreads) void k_hist(Decode P, int bpw, uint32_t *counts,
                                                        int32_t *errs, float *tlut_w)
{
    extern __shared__ uint32_t lds[];
    uint32_t *hist = lds; // [n_tiles]
    __shared__ int serr;
    const int tid = threadIdx.x;
    const long long wg = chunk_of_block(blockIdx.x, gridDim.x);
    for (int b = tid; b < P.n_tiles; b += kPartThreads) hist[b] = 0;
    if (tid == 0) serr = 0;
    __syncthreads();
    const long long begin = wg * (long long)kPartThreads * bpw;
    int err = 0;
    if (KIND == KIND_TAF && tlut_w) {
        // value table for k_scatter: tlut[r] = float(r / (win + 1e-8)) - 1 (generate_taf.py:215, :26);
        // one correctly rounded f64 division per distinct in-window time instead of one per event
        const double den = (double)P.win + 1e-8;
        for (long long r = wg * kPartThreads + tid; r <= P.win; r += (long long)gridDim.x * kPartThreads)
            tlut_w[r] = (float)((double)r / den) - 1.0f;
    }
    if (LAYOUT == FRLW_LAYOUT_DAT8) {
        const uint2 *src = (const uint2 *)P.data + begin;
        const long long left = P.n - begin;
        const uint32_t nloc = left < (long long)kPartThreads * bpw ? (uint32_t)(left < 0 ? 0 : left) : (uint32_t)(kPartThreads * bpw);
        uint2 q[kMaxBpw];
#pragma unroll
        for (int j = 0; j < kMaxBpw; ++j) {
            const uint32_t i = (uint32_t)(j * kPartThreads + tid);
            q[j] = i < nloc ? src[i] : make_uint2(0u, 0xffffffffu);
        }
#pragma unroll
        for (int j = 0; j < kMaxBpw; ++j) {
            const uint32_t i = (uint32_t)(j * kPartThreads + tid);
            if (i < nloc) {
                const Pos o = dat_pos<KIND, HAS_MAP>(P, q[j]);
                err |= o.err;
                if (o.tile >= 0) atomicAdd(&hist[o.tile], 1u);
            }
        }
    } else {
        for (int j = 0; j < bpw; ++j) {
            const long long i = begin + (long long)j * kPartThreads + tid;
            if (i < P.n) {
                double t;
                const Pos o = f64_pos<KIND>(P, i, t);
                err |= o.err;
                if (o.tile >= 0) atomicAdd(&hist[o.tile], 1u);
            }
        }
    }
    if (err) atomicOr(&serr, err);
    __syncthreads();
    uint32_t *row = counts + wg * (long long)P.n_tiles;
    for (int b = tid; b < P.n_tiles; b += kPartThreads) row[b] = hist[b];
    // data-dependent errors of this chunk: a plain store per workgroup, OR-ed into the header by k_tilescan (which
    // also resets the header -- no memset launch, no pre-zeroed workspace needed)
    if (tid == 0) errs[wg] = serr;
}

// counts[u][b], u in one slab of 32 workgroups -> exclusive prefix over u (in place), slabtot[slab][b]
__global__ __launch_bounds__(kWave) void k_slabscan(uint32_t *counts, int units, int n_tiles,
                                                     uint32_t *slabtot)
{
    const int b = blockIdx.x * kWave + threadIdx.x;
    const int slab = blockIdx.y;
    const int u0 = slab * kSlabUnits;
    if (b >= n_tiles) return;
    uint32_t v[kSlabUnits];
#pragma unroll
    for (int k = 0; k < kSlabUnits; ++k)
        v[k] = (u0 + k < units) ? counts[(long long)(u0 + k) * n_tiles + b] : 0u;
    uint32_t run = 0;
#pragma unroll
    for (int k = 0; k < kSlabUnits; ++k) {
        const uint32_t t = v[k];
        v[k] = run;
        run += t;
    }
#pragma unroll
    for (int k = 0; k < kSlabUnits; ++k)
        if (u0 + k < units) counts[(long long)(u0 + k) * n_tiles + b] = v[k];
    slabtot[(long long)slab * n_tiles + b] = run;
}

// slabtot[s][b] -> exclusive prefix over s (in place); then exclusive scan over tiles -> base[0..n]
__global__ __launch_bounds__(1024) void k_tilescan(uint32_t *slabtot, int slabs, int n, uint32_t *base,
                                                   WsHeader *hdr, uint32_t hot_thr, const int32_t *errs, int units)
{
    __shared__ uint32_t tot[kMaxTiles];
    __shared__ uint32_t wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) { hdr->status = 0; hdr->pad = 0; hdr->wmask = 0ull; hdr->n_hot = 0u; hdr->hot_thr = hot_thr; }
    __syncthreads();
    {
        int e = 0;
        for (int u = tid; u < units; u += 1024) e |= errs[u];
        if (e) atomicOr(&hdr->status, e);
    }
    for (int b = tid; b < n; b += 1024) {
        uint32_t run = 0;
        for (int s0 = 0; s0 < slabs; s0 += 8) { // 8 independent loads in flight, then the 8 prefix stores
            uint32_t v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = s0 + k < slabs ? slabtot[(long long)(s0 + k) * n + b] : 0u;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (s0 + k < slabs) slabtot[(long long)(s0 + k) * n + b] = run;
                run += v[k];
            }
        }
        tot[b] = run;
        if (run > hot_thr) { // skewed stream: list the tile, its cells will be shared by several workgroups
            const uint32_t slot = atomicAdd(&hdr->n_hot, 1u);
            if (slot < (uint32_t)kMaxHot) hdr->hot[slot] = (uint32_t)b;
        }
    }
    __syncthreads();
    const int per = (n + 1023) / 1024;
    const int b0 = tid * per;
    int b1 = b0 + per;
    if (b1 > n) b1 = n;
    uint32_t s = 0;
    for (int b = b0; b < b1; ++b) s += tot[b];
    const uint32_t inc = wave_incl_scan(s); // inclusive scan inside the wave
    if (lane == kWave - 1) wsum[wv] = inc;
    __syncthreads();
    uint32_t pre = 0;
    for (int k = 0; k < wv; ++k) pre += wsum[k];
    uint32_t run = pre + inc - s;
    for (int b = b0; b < b1; ++b) {
        base[b] = run;
        run += tot[b];
    }
    if (tid == 1023) base[n] = pre + inc;
    if (tid == 0) fold_sticky_status(hdr, hdr->status); // every error flag was OR-ed in before the first barrier above
}

// ---------------------------------------------------------------------------------------------
constexpr int kStageCap = 4096; // records staged in LDS per write-out window of the staged scatter
// byte offset (16-aligned) of the staging area behind gs | wcnt | tag in the scatter workgroup's LDS
__host__ __device__ inline size_t stage_off(int n_tiles)
{
    const size_t nt2 = (size_t)((n_tiles + 1) & ~1);
    return ((size_t)n_tiles * 4 + (size_t)kPartWaves * nt2 * 2 + (size_t)kPartWaves * n_tiles + 15) & ~(size_t)15;
}
template <int LAYOUT, int KIND, bool HAS_MAP, bool STAGED>
__global__ __launch_bounds__(kPartThreads, 8) void k_scatter(Decode P, int bpw, const uint32_t *counts,
                                                           const uint32_t *slabtot,
                                                           const uint32_t *base, uint2 *records,
                                                           WsHeader *hdr)
{
    extern __shared__ uint32_t lds[];
    // gs[n_tiles] u32 | wcnt[16][n_tiles] u16 | tag[16][n_tiles] u8
    uint32_t *gs = lds;
    uint16_t *wcnt_all = (uint16_t *)(lds + P.n_tiles);
    const int nt2 = (P.n_tiles + 1) & ~1;
    uint8_t *tag_all = (uint8_t *)(wcnt_all + (size_t)kPartWaves * nt2);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long long wg = chunk_of_block(blockIdx.x, gridDim.x);
    // volatile: the tag write-then-read-back must reach LDS (another lane may have overwritten it)
    volatile uint16_t *wcnt = wcnt_all + (size_t)wv * nt2;
    volatile uint8_t *tag = tag_all + (size_t)wv * P.n_tiles;
    for (int b = tid; b < kPartWaves * nt2; b += kPartThreads) wcnt_all[b] = 0;
    __syncthreads();

    const long long wave_begin = (wg * kPartWaves + wv) * (long long)kWave * bpw;
    const uint64_t lt = lanemask_lt();
    // this workgroup's first record slot of tile `tid` (needed only in phase B: issue the loads now)
    uint32_t gs_pre = 0;
    if (tid < P.n_tiles)
        gs_pre = base[tid] + slabtot[(wg / kSlabUnits) * (long long)P.n_tiles + tid] + counts[wg * (long long)P.n_tiles + tid];
    uint32_t where[kMaxBpw]; // tile << 16 | rank in (wave, tile); 0xffffffff = not encoded
    uint32_t cells[kMaxBpw / 2]; // strip-free cell of every event, two 16-bit values per register
    uint2 q[kMaxBpw];
    const long long left = P.n - wave_begin;
    const uint32_t nloc = left < (long long)kWave * bpw ? (uint32_t)(left < 0 ? 0 : left) : (uint32_t)(kWave * bpw);
    if (LAYOUT == FRLW_LAYOUT_DAT8) {
        const uint2 *src = (const uint2 *)P.data + wave_begin;
#pragma unroll
        for (int j = 0; j < kMaxBpw; ++j) {
            const uint32_t i = (uint32_t)(j * kWave + lane);
            q[j] = i < nloc ? src[i] : make_uint2(0u, 0xffffffffu);
        }
    }
#pragma unroll
    for (int j = 0; j < kMaxBpw / 2; ++j) cells[j] = 0u;
    // ---- phase A: ranks inside the wave's run, batch by batch in stream order (positions only;
    //      windows and f32 values are computed in phase C so that few registers stay live)
#pragma unroll
    for (int j = 0; j < kMaxBpw; ++j) {
        where[j] = 0xffffffffu;
        if (j < bpw) { // wave-uniform
            const uint32_t i = (uint32_t)(j * kWave + lane);
            Pos o;
            o.tile = -1; o.cell = 0; o.err = 0;
            if (i < nloc) {
                if (LAYOUT == FRLW_LAYOUT_DAT8) {
                    o = dat_pos<KIND, HAS_MAP>(P, q[j]);
                } else {
                    double t;
                    o = f64_pos<KIND>(P, wave_begin + i, t);
                }
            }
            const bool act = o.tile >= 0;
            cells[j >> 1] |= (o.cell & 0xffffu) << (16 * (j & 1));
            const uint64_t am = __ballot(act);
            if (am != 0ull) {
                if (act) tag[o.tile] = (uint8_t)lane;
                const int seen = act ? (int)tag[o.tile] : lane; // LDS ops of a wave execute in order
                uint64_t cm = __ballot(act && seen != lane);
                uint32_t rank = 0, size = 1;
                while (cm) {
                    const int l0 = __ffsll((long long)cm) - 1;
                    const int g = __builtin_amdgcn_readlane(o.tile, l0);
                    const uint64_t mg = __ballot(act && o.tile == g);
                    if (act && o.tile == g) {
                        rank = (uint32_t)__popcll(mg & lt);
                        size = (uint32_t)__popcll(mg);
                    }
                    cm &= ~mg;
                }
                if (act) {
                    const uint32_t r = (uint32_t)wcnt[o.tile] + rank;
                    where[j] = ((uint32_t)o.tile << 16) | r;
                    if (rank + 1 == size) wcnt[o.tile] = (uint16_t)(r + 1); // after every member's read
                }
            }
        }
    }
    __syncthreads();
    // ---- phase B: per tile, exclusive prefix of the 16 wave counts; global start of this workgroup's run
    {
        const uint32_t *row = counts + wg * (long long)P.n_tiles;
        const uint32_t *srow = slabtot + (wg / kSlabUnits) * (long long)P.n_tiles;
        for (int b = tid; b < P.n_tiles; b += kPartThreads) {
            uint32_t run = 0;
#pragma unroll
            for (int w = 0; w < kPartWaves; ++w) {
                const uint32_t v = wcnt_all[w * nt2 + b];
                wcnt_all[w * nt2 + b] = (uint16_t)run;
                run += v;
            }
            gs[b] = b == tid ? gs_pre : base[b] + srow[b] + row[b];
            if (STAGED) ((uint32_t *)((uint8_t *)lds + stage_off(P.n_tiles)))[b] = run; // loff[b] <- tile total
        }
    }
    __syncthreads();
    // ---- phase C: window + value of every encoded event (all table loads first: vmcnt also counts
    //      stores, so a load issued after a store would wait for that store), then the burst write
    const int cb = P.twl + 4;
    const uint16_t *woff = wcnt_all + (size_t)wv * nt2;
    unsigned long long wseen = 0ull; // TAF: windows this lane has encoded an event for
    uint32_t meta[kMaxBpw], valb[kMaxBpw];
#pragma unroll
    for (int j = 0; j < kMaxBpw; ++j) {
        meta[j] = 0; valb[j] = 0;
        if (j < bpw && where[j] != 0xffffffffu) {
            int window = 0;
            float val = 0.0f;
            const uint32_t cell = (cells[j >> 1] >> (16 * (j & 1))) & 0xffffu;
            if (LAYOUT == FRLW_LAYOUT_DAT8) {
                dat_value<KIND>(P, q[j], window, val);
            } else {
                const double t = ((const double *)P.data + (wave_begin + j * kWave + lane) * (long long)P.row_stride)[2];
                val = f64_value<KIND>(t);
            }
            meta[j] = ((uint32_t)window << cb) | cell;
            valb[j] = __float_as_uint(val);
            if (KIND == KIND_TAF) wseen |= 1ull << window;
        }
    }
    if (STAGED) {
        // Large streams: records go through LDS in workgroup-sorted order (tile-major) and leave in windows of
        // kStageCap consecutive slots: consecutive threads write consecutive records of a tile's run, so the stores
        // of a wave cover whole lines instead of 64 separate 8-byte pieces (HBM write traffic of the scatter: 2.6x
        // its payload with direct stores).  Small streams keep the direct stores: fewer barriers.
        uint32_t *loff = (uint32_t *)((uint8_t *)lds + stage_off(P.n_tiles)); // [n_tiles + 2]
        uint2 *stage = (uint2 *)(loff + ((P.n_tiles + 2 + 1) & ~1));
        uint16_t *stile = (uint16_t *)(stage + kStageCap);
        __shared__ uint32_t wtot[kPartWaves];
        // per-tile totals of this workgroup = prefix end of the last wave + its count: recompute from the wave
        // prefixes (woff of wave 15 + that wave's count is not kept) -> phase B left the totals in loff[]
        __syncthreads();
        uint32_t a0 = 0, a1 = 0;
        if (2 * tid < P.n_tiles) a0 = loff[2 * tid];
        if (2 * tid + 1 < P.n_tiles) a1 = loff[2 * tid + 1];
        const uint32_t inc = wave_incl_scan(a0 + a1);
        if (lane == kWave - 1) wtot[wv] = inc;
        __syncthreads();
        uint32_t pre = 0, total = 0;
        for (int k = 0; k < kPartWaves; ++k) { if (k < wv) pre += wtot[k]; total += wtot[k]; }
        const uint32_t excl = pre + inc - (a0 + a1);
        if (2 * tid < P.n_tiles) loff[2 * tid] = excl;
        if (2 * tid + 1 < P.n_tiles) loff[2 * tid + 1] = excl + a0;
        __syncthreads();
        for (uint32_t w0 = 0; w0 < total; w0 += kStageCap) {
#pragma unroll
            for (int j = 0; j < kMaxBpw; ++j) {
                if (j < bpw && where[j] != 0xffffffffu) {
                    const uint32_t b = where[j] >> 16;
                    const uint32_t slot = loff[b] + (uint32_t)woff[b] + (where[j] & 0xffffu) - w0;
                    if (slot < (uint32_t)kStageCap) { stage[slot] = make_uint2(meta[j], valb[j]); stile[slot] = (uint16_t)b; }
                }
            }
            __syncthreads();
            const uint32_t nwin = total - w0 < (uint32_t)kStageCap ? total - w0 : (uint32_t)kStageCap;
            for (uint32_t qi = tid; qi < nwin; qi += kPartThreads) {
                const uint32_t b = stile[qi];
                records[gs[b] + (w0 + qi - loff[b])] = stage[qi];
            }
            __syncthreads();
        }
    } else {
#pragma unroll
        for (int j = 0; j < kMaxBpw; ++j) {
            if (j < bpw && where[j] != 0xffffffffu) {
                const uint32_t b = where[j] >> 16;
                const uint32_t pos = gs[b] + (uint32_t)woff[b] + (where[j] & 0xffffu);
                records[pos] = make_uint2(meta[j], valb[j]);
            }
        }
    }
    if (KIND == KIND_TAF) { // which windows hold events at all: decides "all(forward)", generate_taf.py:40
        // one same-address global atomic costs ~10 ns at the L2: OR inside the wave, then inside the
        // workgroup, and touch the global word only for bits it does not show yet
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const unsigned lo = __shfl_xor((unsigned)wseen, off), hi = __shfl_xor((unsigned)(wseen >> 32), off);
            wseen |= ((unsigned long long)hi << 32) | lo;
        }
        __syncthreads(); // gs[] is dead now: reuse its first words
        unsigned long long *wg_seen = (unsigned long long *)gs;
        if (tid == 0) *wg_seen = 0ull;
        __syncthreads();
        if (lane == 0 && wseen) atomicOr(wg_seen, wseen);
        __syncthreads();
        if (tid == 0) {
            const unsigned long long mine = *wg_seen;
            const unsigned long long have = __hip_atomic_load(&hdr->wmask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (mine & ~have) atomicOr(&hdr->wmask, mine);
        }
    }
}

template <int LAYOUT, int KIND, bool HAS_MAP>
void launch_partition_m(const Decode &d, const Plan &p, uint32_t *counts, uint32_t *slabtot,
                      uint32_t *base, uint2 *records, WsHeader *hdr, int32_t *errs, float *tlut_w, hipStream_t s)
{
    const size_t lds_hist = (size_t)p.n_tiles * 4;
    const int nt2 = (p.n_tiles + 1) & ~1;
    const size_t lds_sc = (size_t)p.n_tiles * 4 + (size_t)kPartWaves * nt2 * 2 + (size_t)kPartWaves * p.n_tiles + 16;
    const size_t lds_staged = stage_off(p.n_tiles) + (size_t)((p.n_tiles + 2 + 1) & ~1) * 4 + (size_t)kStageCap * 8 +
                              (size_t)kStageCap * 2 + 16;
    // staged write-out pays off for long streams with long chunks (10 M events: -11 us); on short ones the extra
    // barriers cost more than the write traffic saves (GEN1-shaped 1 M events: 66 -> 74 us).
    // frlw_tuning_t::staged_scatter forces it.
    const bool staged = (p.staged >= 0 ? p.staged != 0 : (p.bpw >= 4 && d.n >= 3000000)) && lds_staged <= 150 * 1024;
    hipLaunchKernelGGL((k_hist<LAYOUT, KIND, HAS_MAP>), dim3(p.units), dim3(kPartThreads), lds_hist, s, d, p.bpw, counts, errs, tlut_w);
    hipLaunchKernelGGL(k_slabscan, dim3((p.n_tiles + kWave - 1) / kWave, p.slabs), dim3(kWave), 0, s, counts,
                       p.units, p.n_tiles, slabtot);
    hipLaunchKernelGGL(k_tilescan, dim3(1), dim3(1024), 0, s, slabtot, p.slabs, p.n_tiles, base, hdr, p.hot_thr, errs, p.units);
    if (staged) {
        if (lds_staged > 64 * 1024)
            (void)hipFuncSetAttribute((const void *)k_scatter<LAYOUT, KIND, HAS_MAP, true>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_staged);
        hipLaunchKernelGGL((k_scatter<LAYOUT, KIND, HAS_MAP, true>), dim3(p.units), dim3(kPartThreads), lds_staged, s, d, p.bpw,
                           counts, slabtot, base, records, hdr);
    } else {
        hipLaunchKernelGGL((k_scatter<LAYOUT, KIND, HAS_MAP, false>), dim3(p.units), dim3(kPartThreads), lds_sc, s, d, p.bpw,
                           counts, slabtot, base, records, hdr);
    }
}

template <int LAYOUT, int KIND>
void launch_partition(const Decode &d, const Plan &p, uint32_t *counts, uint32_t *slabtot, uint32_t *base,
                      uint2 *records, WsHeader *hdr, int32_t *errs, float *tlut_w, hipStream_t s)
{
    if (LAYOUT == FRLW_LAYOUT_DAT8 && d.xmap)
        launch_partition_m<LAYOUT, KIND, true>(d, p, counts, slabtot, base, records, hdr, errs, tlut_w, s);
    else
        launch_partition_m<LAYOUT, KIND, false>(d, p, counts, slabtot, base, records, hdr, errs, tlut_w, s);
}

template <int LAYOUT>
void launch_partition_kind(int kind, const Decode &d, const Plan &p, uint32_t *counts, uint32_t *slabtot,
                           uint32_t *base, uint2 *records, WsHeader *hdr, int32_t *errs, float *tlut_w, hipStream_t s)
{
    switch (kind) {
    case KIND_ECI: launch_partition<LAYOUT, KIND_ECI>(d, p, counts, slabtot, base, records, hdr, errs, tlut_w, s); break;
    case KIND_EV: launch_partition<LAYOUT, KIND_EV>(d, p, counts, slabtot, base, records, hdr, errs, tlut_w, s); break;
    case KIND_SAE: launch_partition<LAYOUT, KIND_SAE>(d, p, counts, slabtot, base, records, hdr, errs, tlut_w, s); break;
    default: launch_partition<LAYOUT, KIND_TAF>(d, p, counts, slabtot, base, records, hdr, errs, tlut_w, s); break;
    }
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

} // namespace

bool make_plan(long long n, int H, int W, const frlw_tuning_t *tuning, Plan &p)
{
    if (H <= 0 || W <= 0 || n < 0) return false;
    const auto knob = [&](int32_t frlw_tuning_t::*f, int dflt) { return tuning_knob(tuning, f, dflt); };
    p.twl = knob(&frlw_tuning_t::tile_width_log2, W > 512 ? 8 : 6);
    if (p.twl < 6 || p.twl > 8) return false;
    for (;; ++p.twl) { // tall frames (a batch of sequences stacked along y): widen the tiles to stay under kMaxTiles
        const int tw = 1 << p.twl;
        p.tiles_x = (W + tw - 1) / tw;
        p.tiles_y = (H + kTileH - 1) / kTileH;
        p.n_tiles = p.tiles_x * p.tiles_y;
        if (p.n_tiles <= kMaxTiles || p.twl == 8) break;
    }
    if (p.n_tiles > kMaxTiles) return false;
    // Workgroup chunk = kPartThreads * bpw events.  32 / kPartWaves partition workgroups fit on a CU, i.e.
    // `slots` at a time on the chip: pick bpw so that the workgroup count lands just under a multiple of
    // `slots` (whole rounds) with the longest chunk (= longest contiguous run per tile) that allows.
    const long long slots = 256ll * (32 / kPartWaves);
    int bpw = kMaxBpw;
    for (int rounds = 1; rounds <= 64; ++rounds) {
        const long long want = (n + slots * rounds * kPartThreads - 1) / (slots * rounds * kPartThreads);
        if (want <= kMaxBpw) { bpw = (int)(want < 1 ? 1 : want); break; }
    }
    p.bpw = knob(&frlw_tuning_t::batches_per_wave, bpw);
    if (p.bpw < 1 || p.bpw > kMaxBpw) return false;
    p.chunk = (long long)kPartThreads * p.bpw;
    p.units = (int)((n + p.chunk - 1) / p.chunk);
    if (p.units < 1) p.units = 1;
    p.slabs = (p.units + kSlabUnits - 1) / kSlabUnits;
    size_t off = kHeaderBytes;
    p.off_counts = off;  off = align_up(off + (size_t)p.units * p.n_tiles * 4, 256);
    p.off_slabtot = off; off = align_up(off + (size_t)p.slabs * p.n_tiles * 4, 256);
    p.off_base = off;    off = align_up(off + (size_t)(p.n_tiles + 1) * 4, 256);
    p.off_errs = off;    off = align_up(off + (size_t)p.units * 4, 256);
    p.off_tlut = off;    off = align_up(off + (size_t)(kMaxTlut + 1) * 4, 256);
    p.off_records = off; off = align_up(off + (size_t)(n > 0 ? n : 1) * 8, 256);
    p.bytes = off;
    // more than two slices and more than 4x the mean: worth sharing (frlw_tuning_t::hot_tile_records: tests force it)
    const long long slice2 = 2ll * kSliceMult * (4 << p.twl), mean4 = 4 * (n / p.n_tiles);
    const long long thr = slice2 > mean4 ? slice2 : mean4;
    p.hot_thr = (unsigned)knob(&frlw_tuning_t::hot_tile_records, thr > 0x7fffffffll ? 0x7fffffff : (int)thr);
    p.staged = knob(&frlw_tuning_t::staged_scatter, -1);
    p.quarter_below = knob(&frlw_tuning_t::quarter_below, 1024);
    p.no_lut = knob(&frlw_tuning_t::no_value_table, 0);
    return true;
}

int partition_events(const frlw_events_t *ev, int H, int W, int kind, long long t0, long long win,
                     int n_windows, int time_filter, void *ws, size_t ws_bytes, hipStream_t s,
                     Partitioned &out)
{
    if (!ev || !ws || (ev->n > 0 && !ev->data)) return FRLW_ERR_ARG;
    if (ev->layout != FRLW_LAYOUT_XYTP_F64 && ev->layout != FRLW_LAYOUT_DAT8) return FRLW_ERR_ARG;
    if (ev->layout == FRLW_LAYOUT_XYTP_F64 && ev->row_stride < 4) return FRLW_ERR_ARG;
    if ((ev->xmap == nullptr) != (ev->ymap == nullptr)) return FRLW_ERR_ARG;
    if (!tuning_valid(ev->tuning)) return FRLW_ERR_ARG;
    if (ev->n >= (1ll << 32)) return FRLW_ERR_UNSUPPORTED;
    if (win < 1) win = 1;
    if (kind == KIND_TAF && ev->layout == FRLW_LAYOUT_DAT8 &&
        ((long long)n_windows * win >= (1ll << 32) || win >= (1ll << 31)))
        return FRLW_ERR_UNSUPPORTED;
    Plan p;
    if (!make_plan(ev->n, H, W, ev->tuning, p)) return FRLW_ERR_UNSUPPORTED;
    if (ws_bytes < p.bytes) return FRLW_ERR_WORKSPACE;
    char *w8 = (char *)ws;
    WsHeader *hdr = (WsHeader *)w8;
    uint32_t *counts = (uint32_t *)(w8 + p.off_counts);
    uint32_t *slabtot = (uint32_t *)(w8 + p.off_slabtot);
    uint32_t *base = (uint32_t *)(w8 + p.off_base);
    int32_t *errs = (int32_t *)(w8 + p.off_errs);
    uint2 *records = (uint2 *)(w8 + p.off_records);

    Decode d;
    d.data = ev->data; d.n = ev->n; d.row_stride = ev->row_stride;
    d.xmap = ev->layout == FRLW_LAYOUT_DAT8 ? ev->xmap : nullptr;
    d.ymap = ev->layout == FRLW_LAYOUT_DAT8 ? ev->ymap : nullptr;
    d.map_w = ev->map_w; d.map_h = ev->map_h;
    d.H = H; d.W = W; d.twl = p.twl; d.tiles_x = p.tiles_x; d.n_tiles = p.n_tiles;
    d.t0 = t0; d.win = win; d.n_windows = n_windows; d.time_filter = time_filter;
    const unsigned long long magic = (1ull << 32) / (unsigned long long)win;
    d.win_magic = magic > 0xffffffffull ? 0xffffffffu : (uint32_t)magic;
    float *tlut_w = nullptr;
    if (kind == KIND_TAF && ev->layout == FRLW_LAYOUT_DAT8 && win <= kMaxTlut && !p.no_lut) tlut_w = (float *)(w8 + p.off_tlut);
    d.tlut = tlut_w;
    const uint32_t *leaky = nullptr; // level thresholds of uint8(leaky_transform(.)) for the tile kernel's epilogue (generate_taf.py:69-76)
    if (kind == KIND_TAF && !(leaky = leaky_table(s))) return FRLW_ERR_HIP;

    (void)hipGetLastError(); // stale errors of other libraries in the process
    if (ev->layout == FRLW_LAYOUT_DAT8)
        launch_partition_kind<FRLW_LAYOUT_DAT8>(kind, d, p, counts, slabtot, base, records, hdr, errs, tlut_w, s);
    else
        launch_partition_kind<FRLW_LAYOUT_XYTP_F64>(kind, d, p, counts, slabtot, base, records, hdr, errs, tlut_w, s);
    HIP_TRY(hipGetLastError());
    out.records = records; out.base = base; out.hdr = hdr; out.plan = p; out.leaky_thr = leaky;
    return FRLW_OK;
}

// ---- the uint8(leaky_transform(.)) threshold table: one per device, built on first use (see frlw_common.h) -----------
namespace {
__device__ uint32_t g_leaky_thr[kLeakyTableWords];
__global__ __launch_bounds__(kLeakyLevels) void k_leaky_fill()
{
    __shared__ uint32_t thr[kLeakyLevels];
    __shared__ uint8_t lut[kLeakyBuckets];
    __shared__ int bad;
    const int t = threadIdx.x;
    thr[t] = t == 0 ? 0x7f800000u : leaky_threshold_bits(t); // generate_taf.py:69-76 as a 256-level step function
    if (t == 0) bad = 0;
    __syncthreads();
    g_leaky_thr[t] = thr[t];
    // the bucket table of leaky_u8_bucket_n: level(x) = the largest k with x <= thr[k] (thr is non-increasing in k)
    auto level = [&](uint32_t xb) {
        int lo = 0, hi = kLeakyLevels; // thr[lo] >= xb always (thr[0] = +inf); first k with thr[k] < xb is in (lo, hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (xb <= thr[mid]) lo = mid; else hi = mid;
        }
        return lo;
    };
    for (int b = t; b < kLeakyBuckets; b += kLeakyLevels) {
        const uint32_t lo_x = b == 0 ? 0u : (uint32_t)(b + kLeakyBucket0) << kLeakyBucketShift;
        const uint32_t hi_x = b == kLeakyBuckets - 1 ? 0x7f800000u : (((uint32_t)(b + kLeakyBucket0 + 1)) << kLeakyBucketShift) - 1u;
        const int k_hi = level(hi_x), k_lo = level(lo_x);
        lut[b] = (uint8_t)k_hi;
        if (k_lo - k_hi > 1 || k_hi > kLeakyLevels - 2) bad = 1; // more than one threshold inside the bucket: the table is not used
    }
    __syncthreads();
    for (int w = t; w < kLeakyBuckets / 4; w += kLeakyLevels)
        g_leaky_thr[kLeakyLutWord + w] = (uint32_t)lut[4 * w] | ((uint32_t)lut[4 * w + 1] << 8) | ((uint32_t)lut[4 * w + 2] << 16) | ((uint32_t)lut[4 * w + 3] << 24);
    if (t == 0) g_leaky_thr[kLeakyOkWord] = bad ? 0u : 1u;
}
struct LeakyDev { int state = 0; hipEvent_t ev = nullptr; const uint32_t *ptr = nullptr; }; // 0 never filled, 1 fill queued, 2 done
std::mutex g_leaky_mu;
LeakyDev g_leaky_dev[64];
} // namespace

const uint32_t *leaky_table(hipStream_t s)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(g_leaky_mu);
    LeakyDev &d = g_leaky_dev[dev];
    if (!d.ptr) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_leaky_thr)) != hipSuccess) return nullptr;
        d.ptr = (const uint32_t *)p;
    }
    if (d.state == 2) return d.ptr;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cap);
    if (cap != hipStreamCaptureStatusNone) { // the fill becomes a node of that graph (idempotent); nothing is remembered
        hipLaunchKernelGGL(k_leaky_fill, dim3(1), dim3(kLeakyLevels), 0, s);
        return d.ptr;
    }
    if (d.state == 0) { // in front of the caller's kernels on its own stream
        hipLaunchKernelGGL(k_leaky_fill, dim3(1), dim3(kLeakyLevels), 0, s);
        if (hipEventCreateWithFlags(&d.ev, hipEventDisableTiming) == hipSuccess && hipEventRecord(d.ev, s) == hipSuccess) d.state = 1;
        return d.ptr; // (no event: the next call fills again)
    }
    if (hipEventQuery(d.ev) == hipSuccess) d.state = 2;
    else if (hipStreamWaitEvent(s, d.ev, 0) != hipSuccess) return nullptr;
    return d.ptr;
}

} // namespace frlw
