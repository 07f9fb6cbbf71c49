// conv_mfma.h -- the MFMA implicit-GEMM convolution kernel shared by the detector plan (detector.hip) and the training
// operators (train_ops.hip), in two arithmetics: v_mfma_f32_32x32x2_f32 on float32 operands (P = 0) and float32 products from
// three v_mfma_f32_32x32x16_bf16 on hi / lo split operands (P = 1, ConvArgs::prec).  Included inside each translation unit's
// anonymous namespace.
//
//   Y[M = B*Ho*Wo][N = Cout] = A[M][K] * W[K][N],  K = (ky, kx, ci),  A gathered on the fly from an NHWC view.
//   tstride = 2 turns the gather into that of a transposed (stride-2) convolution: input coordinate
//   (oy - pad + ky) / 2 when even, zero otherwise -- the data gradient of a stride-2 convolution.
// (no #includes here: the including file has <hip/hip_runtime.h>, <math.h>, <stdint.h>, <stdlib.h> already)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int V> struct ConvIC { static constexpr int value = V; }; // compile-time step index for asm immediates

// Launch heuristics are compile-time constants in the product library: it reads no environment and prints nothing.  A
// developer build (-DFRLW_DEV_BUILD) lets the environment override them for A/B runs and logs HIP errors.
#ifdef FRLW_DEV_BUILD
inline long long dev_knob(const char *name, long long dflt) { const char *e = getenv(name); return e ? atoll(e) : dflt; }
#define FRLW_DEV_LOG(...) fprintf(stderr, __VA_ARGS__)
#else
constexpr long long dev_knob(const char *, long long dflt) { return dflt; }
#define FRLW_DEV_LOG(...) do { } while (0)
#endif

enum : int { ACT_NONE = 0, ACT_SILU = 1, ACT_SIGMOID = 2 };

struct ConvArgs {
    const float *x; int H, W, Cin, x_cs, x_co; long long x_bs; // input view: pixel stride x_cs, channel offset x_co
    const float *w; const float *bias; int Cout, Npad, k, stride, pad;
    float *y; int Ho, Wo, y_cs, y_co; long long y_bs;
    const float *res; int r_cs, r_co; long long r_bs;
    int act, sig_from; // sigmoid applies to channels >= sig_from when act == ACT_SIGMOID
    int M, K;
    int splits;        // split-K: blockIdx.z owns a slice of the k-tiles and writes raw partial sums
    float *partial;    // [splits][M][Npad] when splits > 1
    int tstride;       // 0 / 1: ordinary gather; 2: transposed gather (dgrad of a stride-2 convolution)
    int kw;            // taps per tap row when the window is not square (0: k); K = rows * kw * Cin
    uint32_t x_bytes, w_bytes; // extents of the x view and of w for the buffer descriptors (set by launch_conv)
    int group_n;       // grouped convolution: output columns [g * group_n, (g + 1) * group_n) read input channels x_co + g * Cin ...; 0: one group
    int y_rp;          // output row pitch in floats (0: dense, pixel p at p * y_cs); else pixel (oy, ox) at oy * y_rp + ox * y_cs
    int prec;          // 0: float32 MFMA on the [K][Npad] float32 operand; 1: three bf16 MFMAs per product on the split operand (below)
    double *stats;     // train forward (act NONE, no bias): per slab of BM output rows and channel {sum, sum of squares} of the
    int stats_rows;    // outputs -- [stats_rows][Cout][2]; set by launch_conv (0 / NULL: the launch does not produce them)
    int *sk_counters;  // split-K reduced INSIDE the kernel (k_conv_mfma<.., SK = true>): one arrival counter per output tile, zero
                       // between launches (the last-arriving split of a tile sums all splits, finishes the tile and resets it); NULL: k_splitk_reduce
    float *y2;         // optional second destination: the output nearest-upsampled x2 (nn.Upsample(scale_factor=2), yolo_pafpn.py:29) --
    int y2_cs, y2_co;  // pixel (oy, ox) also lands at (2 oy + {0, 1}, 2 ox + {0, 1}) of a (2 Ho, 2 Wo) NHWC view with this pixel
    long long y2_bs;   // stride / channel offset / image stride: the FPN's upsample costs four more stores in the epilogue, not a launch
};

// ---- float32 products from three bf16 MFMAs (prec = 1) ---------------------------------------------------------------------
// x = hi + lo + e with hi = bf16(x), lo = bf16(x - hi), |e| <= 2^-18 |x|; a * b ~ a_hi b_hi + a_hi b_lo + a_lo b_hi (the dropped
// a_lo b_lo is <= 2^-18 |a b|): three v_mfma_f32_32x32x16_bf16 (32 cycles each) per 16 k instead of eight
// v_mfma_f32_32x32x2_f32 (64 cycles each) -- 5.3 x the matrix rate at ~1e-5 relative error per product, float32 accumulation.
// The gathered operand stays float32 in memory and in LDS and is split in registers (six VALU instructions per pair of values,
// hidden beside the MFMAs); the weight operand is split once, by its layout kernel, into the image the fragment reads want:
//   [k-tile of 16][lane half h][hi | lo][Npad columns] records of eight bf16 (16 bytes): element j of record (kt, h, part, n)
//   = part(W[16 kt + kmem(h, j)][n]),  kmem(h, j) = 4 h + j (j < 4), 8 + 4 h + (j - 4) (j >= 4)
// -- the k a lane finds in its two 16-byte reads of the gathered row.  Same bytes as the float32 operand (K padded to 16).
__host__ __device__ inline int conv_split_kmem(int h, int j) { return j < 4 ? 4 * h + j : 8 + 4 * h + (j - 4); }
__device__ __forceinline__ uint32_t conv_bf16_pair(float a, float b) // {bf16(a), bf16(b)} (round to nearest even), a low
{
    f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
// eight float32 (two quads) -> hi and lo fragments
__device__ __forceinline__ void conv_split8(const f32x4 &q0, const f32x4 &q1, u32x4 &hi, u32x4 &lo)
{
    const float x[8] = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const uint32_t h = conv_bf16_pair(x[2 * e], x[2 * e + 1]);
        // (the compiler pairs the two subtractions into one v_pk_add_f32; two v_sub_f32 measured the same)
        const float r0 = x[2 * e] - __builtin_bit_cast(float, h << 16);
        const float r1 = x[2 * e + 1] - __builtin_bit_cast(float, h & 0xFFFF0000u);
        hi[e] = h;
        lo[e] = conv_bf16_pair(r0, r1);
    }
}

#ifndef CONV_BK_BIG
#define CONV_BK_BIG 16
#endif
#ifndef CONV_BK_SMALL
#define CONV_BK_SMALL 16
#endif
constexpr int kSplitBK = 16; // granularity the split-K heuristics count k-tiles in
#ifndef CONV_F32_PIPE
#define CONV_F32_PIPE 1 // 0: the float32-MFMA loop without the pipelined iteration seam (A/B in tools/conv_lab.hip)
#endif

__device__ __forceinline__ float act_apply(float v, int act)
{
    // hardware exponential and reciprocal (~2 ulp): the epilogue runs with no MFMA left to hide it, and the detector's
    // tolerance is 1e-3 (the exact expf + IEEE division cost 25 instructions per output, this costs 6)
    if (act == ACT_SILU) return v * __frcp_rn(1.0f + __expf(-v));      // x * sigmoid(x)
    if (act == ACT_SIGMOID) return __frcp_rn(1.0f + __expf(-v));
    return v;
}

#ifdef CONV_LAB_PROBE
__device__ uint32_t *g_probe;
__device__ unsigned long long *g_probe_t, *g_probe_e, *g_probe_ph;
#define PROBE_PH(i) do { if (probe_end.on && threadIdx.x == 0) g_probe_ph[probe_end.idx * 8 + (i)] = wall_clock64(); } while (0)
struct ProbeEnd { // writes the workgroup's end time when the kernel body is left
    int idx; bool on;
    __device__ ~ProbeEnd() { if (on) { __syncthreads(); if (threadIdx.x == 0) g_probe_e[idx] = wall_clock64(); } }
};
#else
#define PROBE_PH(i) do { } while (0)
#endif
// Operand loads go through buffer descriptors: a lane whose element is padding, outside the tensor or past K carries
// the offset kOob, the range check of the descriptor answers zeros -- no branch and no select around any load.
constexpr uint32_t kOob = 0xF0000000u;        // no operand view reaches this byte offset (launch_conv splits the batch otherwise)
constexpr long long kMaxViewBytes = 0xE0000000ll;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t conv_rsrc(const void *p, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 conv_load16(__amdgpu_buffer_rsrc_t r, uint32_t byte_off)
{
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
}

// BM x BN output tile, 4 wavefronts arranged WROWS x WCOLS, each owning TM x TN MFMA tiles of 32 x 32.
// UT ("uniform taps"): Cin % BK == 0, so a k-tile lies inside one filter tap and the tap changes for the whole
// workgroup at once: the gather offsets are recomputed only then, a k-tile costs one add per load.
template <int BM, int BN, int WROWS, int WCOLS, int BK, bool UT, int D, int P, bool SK>
__device__ __forceinline__ void conv_mfma_body(const ConvArgs &a);

template <int BM, int BN, int WROWS, int WCOLS, int BK, bool UT, int D = 2, int P = 0>
__global__ __launch_bounds__(64 * WROWS * WCOLS) void k_conv_mfma(ConvArgs a)
{
    conv_mfma_body<BM, BN, WROWS, WCOLS, BK, UT, D, P, false>(a);
}

// the 64 x 64 tile with the split-K reduction inside the kernel (see the epilogue): its last-arriving workgroup keeps 16 partial
// rows in registers, and the launches that use it want five workgroups per CU -- five wavefronts per SIMD, at most 96 registers
template <int BK>
#ifndef SK_OCC
#define SK_OCC 5
#endif
#ifndef SK_ROWS
#define SK_ROWS 2
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SK_OCC, 8))) void k_conv_mfma_sk(ConvArgs a)
{
    conv_mfma_body<64, 64, 2, 2, BK, true, 2, 0, true>(a);
}

template <int BM, int BN, int WROWS, int WCOLS, int BK, bool UT, int D, int P, bool SK>
__device__ __forceinline__ void conv_mfma_body(const ConvArgs &a)
{
    constexpr int NT = 64 * WROWS * WCOLS; // four wavefronts (eight for the 128 x 256 tile)
    constexpr int TM = BM / (32 * WROWS), TN = BN / (32 * WCOLS);
    constexpr int KQ = BK / 4;        // float4 per A row of the k-tile
    constexpr int RPP = NT / KQ;      // A rows staged per pass of the NT threads
    // Both operand tiles arrive by LDS-DMA (buffer_load ... lds: no registers, no ds_write), so their LDS images are
    // lane-linear: weights [BK][BN], gathered rows [BM][BK] with the four 16-byte quads of a row XOR-swizzled by
    // (row >> 2) & 3 on the SOURCE side, which makes the ds_read_b128 fragment reads conflict-free.
    static_assert(BK == 16 || (BK == 32 && P == 1), "the A image is four quads per row (eight with the bf16 k-steps: two per k-tile)");
    constexpr int KS = BK / 16;       // bf16 k-steps per k-tile (P == 1)
    constexpr int LDB = BN;
    constexpr int A_F4 = BM * BK / 4 / NT;    // float4 loads per thread for the A tile
    constexpr int B_F4 = (BN * BK / 4 + NT - 1) / NT;
    constexpr int EPLD = 36;                          // row pitch of the epilogue staging (16-byte aligned rows)
    // LDS rings.  The gathered operand streams from HBM / the far L2: its tile t + D is requested while tile t is
    // computed (D + 1 slots).  The weights are L2-resident and shared by every workgroup: D - 1 tiles ahead (D slots).
    // D = 2 where many workgroups share a CU (they hide each other's latency and LDS is what limits their number);
    // D = 4 for launches that leave a workgroup alone on its CU: its k-tile then costs latency / D, not latency / 2.
    constexpr int NA = D + 1, NB = D;
    static_assert((BM * BK / 4) % NT == 0, "whole passes over the gathered tile");
    static_assert(D == 2 || (BN * BK / 4) % NT == 0, "deeper rings count on every wavefront issuing the same DMAs");
    constexpr int kTileFloats = BK * (NA * BM + NB * LDB);
    constexpr int kEpiFloats = (NT / 64) * 32 * EPLD; // one 32 x 32 MFMA tile per wavefront
    __shared__ __attribute__((aligned(16))) float smem[kTileFloats > kEpiFloats ? kTileFloats : kEpiFloats];
    float (*As)[BM][BK] = (float (*)[BM][BK])smem;
    float (*Bs)[BK][LDB] = (float (*)[BK][LDB])(smem + NA * BK * BM);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = wv / WCOLS, wc = wv % WCOLS;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
#ifdef CONV_LAB_PROBE
    if (tid == 0 && g_probe) {
        const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        g_probe[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = ((xcc & 15) << 8) | (((hw >> 13) & 7) << 4) | ((hw >> 8) & 15);
        g_probe_t[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = wall_clock64();
    }
    ProbeEnd probe_end{(int)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x), g_probe != nullptr};
#endif
    const __amdgpu_buffer_rsrc_t rx = conv_rsrc(a.x, a.x_bytes), rw = conv_rsrc(a.w, a.w_bytes);

    // ---- A staging: thread -> A_F4 rows m, one float4 of 4 consecutive k
    // the quad this lane FETCHES; it lands in slot tid % KQ of its row (row tid / KQ): slot ^ f(row), f = (row >> 2) & 3 for
    // 64-byte rows, (row >> 1) & 7 for 128-byte rows -- the 16 lanes of a ds_read_b128 pass then cover all 64 banks
    const int a_k4 = (KQ == 4 ? ((tid & 3) ^ ((tid >> 4) & 3)) : ((tid & 7) ^ ((tid >> 4) & 7))) * 4;
    int a_iy0[A_F4], a_ix0[A_F4];
    uint32_t a_base[A_F4]; // byte offset of the row's image (+ channel offset); kOob for rows past M
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
        const int m = m0 + tid / KQ + RPP * i;
        const int mm = m < a.M ? m : 0;
        const int b = mm / (a.Ho * a.Wo), pix = mm - b * (a.Ho * a.Wo);
        const int oy = pix / a.Wo, ox = pix - oy * a.Wo;
        a_iy0[i] = oy * a.stride - a.pad;
        a_ix0[i] = ox * a.stride - a.pad;
        a_base[i] = m < a.M ? (uint32_t)(((long long)b * a.x_bs + a.x_co + (a.group_n ? (n0 / a.group_n) * a.Cin : 0)) * 4) : kOob;
    }
    const int kw = a.kw ? a.kw : a.k;
    const int x_cs4 = a.x_cs * 4;
    auto gather_off = [&](int i, int ky, int kx) -> uint32_t { // byte offset of row i's pixel under tap (ky, kx)
        int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
        bool ok = a_base[i] != kOob;
        if (a.tstride == 2) { ok = ok && !((iy | ix) & 1); iy >>= 1; ix >>= 1; }
        ok = ok && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        return ok ? a_base[i] + (uint32_t)((iy * a.W + ix) * x_cs4) : kOob;
    };
    // ---- B staging: thread -> rows k, one float4 of 4 consecutive n
    constexpr int BN4 = BN / 4;
    uint32_t b_off[B_F4]; // byte offset of this thread's float4 in the NEXT k-tile to load; rows past K fail the range check

    const int nk_all = (a.K + BK - 1) / BK;
    const int kt0 = (int)((long long)nk_all * blockIdx.z / a.splits), kt1 = (int)((long long)nk_all * (blockIdx.z + 1) / a.splits);
    const int nk = kt1 - kt0;
#pragma unroll
    for (int i = 0; i < B_F4; ++i) {
        const int e = tid + NT * i;
        if (P == 1) { // 16-byte record e of the tile image [h][hi | lo][BN]
            const int hp = e / BN, n = e - hp * BN; // hp = 4 * (k-step of the tile) + 2 h + part
            b_off[i] = (hp < BK / 4 && n0 + n < a.Npad) ? (uint32_t)((((long long)kt0 * (BK / 4) + hp) * a.Npad + n0 + n) * 16) : kOob;
        } else {
            const int kr = e / BN4, n4 = (e - kr * BN4) * 4;
            b_off[i] = (kr < BK && n0 + n4 < a.Npad) ? (uint32_t)((((long long)kt0 * BK + kr) * a.Npad + n0 + n4) * 4) : kOob;
        }
    }
    // tap tracker of the NEXT k-tile to load.  UT: (s_ky, s_kx, s_ci) are workgroup-uniform, a_pix[] holds the rows'
    // offsets under that tap.  Otherwise every thread tracks the tap of its own float4.
    int s_ci, s_ky, s_kx;
    uint32_t a_pix[A_F4];
    {
        const int k = kt0 * BK + (UT ? 0 : a_k4);
        const int tap = k / a.Cin;
        s_ci = k - tap * a.Cin;
        s_ky = tap / kw;
        s_kx = tap - s_ky * kw;
    }
    if (UT) {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) a_pix[i] = gather_off(i, s_ky, s_kx);
    }
    // wave-instruction i of wavefront wv fills the 1 KB at float offset (wv * 64 + NT i) * 4 of a tile
    auto load_a = [&](int kt, int nbuf) {
        if (UT) {
            const uint32_t cib = (uint32_t)(s_ci + a_k4) * 4;
#pragma unroll
            for (int i = 0; i < A_F4; ++i) // kOob + cib stays out of range: zeros land in LDS
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void *)(&As[nbuf][0][0] + (wv * 64 + NT * i) * 4),
                                                         16, (int)(a_pix[i] + cib), 0, 0, 0);
            s_ci += BK;
            if (s_ci == a.Cin) { // next tap (uniform branch, once per Cin / BK tiles)
                s_ci = 0;
                if (++s_kx == kw) { s_kx = 0; ++s_ky; }
#pragma unroll
                for (int i = 0; i < A_F4; ++i) a_pix[i] = gather_off(i, s_ky, s_kx);
            }
        } else {
            const bool in_k = kt * BK + a_k4 < a.K;
#pragma unroll
            for (int i = 0; i < A_F4; ++i) {
                const uint32_t o = gather_off(i, s_ky, s_kx);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void *)(&As[nbuf][0][0] + (wv * 64 + NT * i) * 4),
                                                         16, (int)(in_k ? o + (uint32_t)s_ci * 4 : kOob), 0, 0, 0);
            }
            s_ci += BK; // Cin % 4 == 0: a float4 never straddles taps
            while (s_ci >= a.Cin) { s_ci -= a.Cin; if (++s_kx == kw) { s_kx = 0; ++s_ky; } }
        }
    };
    auto load_b = [&](int nbuf) {
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            if ((BN * BK / 4) % NT == 0 || wv * 64 + NT * i < BN * BK / 4) // wave-uniform
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void *)(&Bs[nbuf][0][0] + (wv * 64 + NT * i) * 4),
                                                         16, (int)b_off[i], 0, 0, 0);
            b_off[i] += (uint32_t)(BK * 4) * (uint32_t)a.Npad;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // Issue order inside an iteration: weights of tile t + D - 1 first, gathered rows of tile t + D last.  The newest
    // instructions behind the weights of tile t + 1 are then A(t + 2) and D - 2 whole (B, A) pairs: "all but my newest
    // kInFlight DMA instructions have landed" (a counted vmcnt) is exactly "tile t + 1 is complete".
    constexpr int kInFlight = A_F4 + (D - 2) * (A_F4 + B_F4);
    auto wait_next_tile = [&](bool steady) { // wave-uniform
        if (steady) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kInFlight) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    PROBE_PH(0);
    load_a(kt0, 0);
#pragma unroll
    for (int i = 0; i < D - 1; ++i) { // the issue order of the iterations -(D - 1) .. -1
        if (i < nk) load_b(i);
        if (i + 1 < nk) load_a(kt0 + i + 1, i + 1);
    }
    PROBE_PH(1);
    wait_next_tile(nk >= D);
    __builtin_amdgcn_s_barrier(); // raw barrier: __syncthreads() would drain the DMA of the tiles in flight
    PROBE_PH(2);
    // MFMA k-step (j, t), j = 0..1, t = 0..3: lane half h supplies k = 8 j + 4 h + t -- any pairing of the tile's 16 k
    // works as long as both operands use it; this one lets a lane take its four A values of a j from ONE 16-byte read.
    static_assert(TM <= 2 && (TN <= 2 || (P == 1 && TN <= 4)), "fragment reads are written out for at most two tiles per direction (four columns of tiles with the bf16 k-steps)");
    const int fh = lane >> 5, fsw = KQ == 4 ? ((lane & 31) >> 2) & 3 : ((lane & 31) >> 1) & 7;
    const int fm = wr * TM * 32 + (lane & 31), fn = wc * TN * 32 + (lane & 31);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)smem;
    // byte addresses in ring slot 0: A quad (2 j + h) ^ swizzle of row fm (second tile: + 32 rows = 2048 B), B row 4 h
    uint32_t a_lds[2 * KS]; // quad 4 s + 2 j + h of row fm, swizzled
#pragma unroll
    for (int q = 0; q < 2 * KS; ++q) a_lds[q] = lds0 + (uint32_t)(fm * BK + ((2 * q + fh) ^ fsw) * 4) * 4;
    const uint32_t b_lds = lds0 + (uint32_t)(NA * BK * BM + 4 * fh * LDB + fn) * 4;
    const uint32_t b3_lds = lds0 + (uint32_t)(NA * BK * BM) * 4 + (uint32_t)(2 * fh * BN + fn) * 16; // prec 1: record (h, hi, column fn)
    int buf = 0, bufb = 0; // ring slots of tile kt: A (kt % NA), B (kt % NB)
    if constexpr (P == 1 && BK == 16 && TM * TN == 1) { // (wider wavefront tiles lose a wavefront per SIMD to the second register set: measured slower)
        // bf16 k-steps, software-pipelined: with the matrix work of a k-tile down to TM * TN * 3 MFMAs of 32 cycles, a wavefront's
        // chain "fragment reads -> wait -> split -> MFMAs -> barrier" is what a thin layer's workgroup spends its time on (one
        // or two workgroups per CU, nobody to hide it).  The fragments of tile t + 1 are therefore requested right after the
        // barrier that publishes it, BEFORE the MFMAs of tile t are issued: the LDS latency and the MFMA chain overlap.  Two
        // register sets for the fragments, the loop unrolled by two.
        f32x4 fa[2][2][TM];
        u32x4 bh[2][TN], bl[2][TN];
        auto read_frags = [&fa, &bh, &bl, &a_lds, b3_lds](auto pc, int slot_a, int slot_b) {
            constexpr int p = decltype(pc)::value;
            const uint32_t a_slot = (uint32_t)slot_a * (BM * BK * 4), b_addr = b3_lds + (uint32_t)slot_b * (BK * LDB * 4);
            asm volatile("ds_read_b128 %0, %1" : "=v"(fa[p][0][0]) : "v"(a_lds[0] + a_slot));
            asm volatile("ds_read_b128 %0, %1" : "=v"(fa[p][1][0]) : "v"(a_lds[1] + a_slot));
            if constexpr (TM > 1) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[p][0][TM - 1]) : "v"(a_lds[0] + a_slot), "n"(32 * BK * 4));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[p][1][TM - 1]) : "v"(a_lds[1] + a_slot), "n"(32 * BK * 4));
            }
            asm volatile("ds_read_b128 %0, %1" : "=v"(bh[p][0]) : "v"(b_addr));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bl[p][0]) : "v"(b_addr), "n"(BN * 16));
            if constexpr (TN > 1) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bh[p][1]) : "v"(b_addr), "n"(512));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bl[p][1]) : "v"(b_addr), "n"(BN * 16 + 512));
            }
            if constexpr (TN > 2) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bh[p][2]) : "v"(b_addr), "n"(1024));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bl[p][2]) : "v"(b_addr), "n"(BN * 16 + 1024));
            }
            if constexpr (TN > 3) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bh[p][3]) : "v"(b_addr), "n"(1536));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bl[p][3]) : "v"(b_addr), "n"(BN * 16 + 1536));
            }
        };
        auto iteration = [&](auto pc, int kt) {
            constexpr int p = decltype(pc)::value;
            if (kt + D - 1 < nk) load_b(bufb == 0 ? NB - 1 : bufb - 1);
            if (kt + D < nk) load_a(kt0 + kt + D, buf == 0 ? NA - 1 : buf - 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the fragments of tile kt (set p) are in registers
#pragma unroll
            for (int i = 0; i < TM; ++i) { asm volatile("" : "+v"(fa[p][0][i])); asm volatile("" : "+v"(fa[p][1][i])); }
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) { asm volatile("" : "+v"(bh[p][jn])); asm volatile("" : "+v"(bl[p][jn])); }
            u32x4 ah[TM], al[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) conv_split8(fa[p][0][i], fa[p][1][i], ah[i], al[i]);
            wait_next_tile(kt + D < nk);            // tile kt + 1 has landed (this wavefront's pieces) ...
            __builtin_amdgcn_s_barrier();           // ... everybody's have, and everybody has tile kt in registers
            buf = buf == NA - 1 ? 0 : buf + 1;
            bufb = bufb == NB - 1 ? 0 : bufb + 1;
            if (kt + 1 < nk) read_frags(ConvIC<1 - p>{}, buf, bufb);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) { // the small terms first
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[i]), __builtin_bit_cast(bf16x8, bh[p][jn]), acc[i][jn], 0, 0, 0);
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bl[p][jn]), acc[i][jn], 0, 0, 0);
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bh[p][jn]), acc[i][jn], 0, 0, 0);
                }
        };
        if (nk > 0) read_frags(ConvIC<0>{}, 0, 0);
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            iteration(ConvIC<0>{}, kt);
            iteration(ConvIC<1>{}, kt + 1);
        }
        if (kt < nk) iteration(ConvIC<0>{}, kt);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if constexpr (P == 0 && BK == 16 && CONV_F32_PIPE && TM * TN == 1) { // (wider wavefront tiles: within +-1 %)
        // float32 MFMA, the iteration's seam pipelined: after the last fragment read of tile t has been issued (k-step 6 requests
        // the B row of k-step 7) the wavefront drains its LDS reads, passes the barrier that publishes tile t + 1, requests the
        // first fragments of tile t + 1 -- and only then issues the MFMAs of k-steps 6 and 7, which it has in registers.  The
        // LDS latency that used to sit in front of every iteration's first MFMA (nobody to hide it on the thin layers: one or two
        // workgroups per CU) now runs under two k-steps of matrix work.  Two register sets, the loop unrolled by two.
        f32x4 fa[2][2][TM];
        float fb[2][2][TN];
        auto read_a = [&fa, &a_lds](auto pc, auto jc, int slot_a) {
            constexpr int p = decltype(pc)::value, j = decltype(jc)::value;
            const uint32_t ad = a_lds[j] + (uint32_t)slot_a * (BM * BK * 4);
            asm volatile("ds_read_b128 %0, %1" : "=v"(fa[p][j][0]) : "v"(ad));
            if constexpr (TM > 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[p][j][TM - 1]) : "v"(ad), "n"(32 * BK * 4));
        };
        auto read_b = [&fb, b_lds](auto pc, auto stc, int slot_b) {
            constexpr int p = decltype(pc)::value, st = decltype(stc)::value;
            constexpr int off = (8 * (st >> 2) + (st & 3)) * LDB * 4;
            const uint32_t bd = b_lds + (uint32_t)slot_b * (BK * LDB * 4);
            asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fb[p][st & 1][0]) : "v"(bd), "n"(off));
            if constexpr (TN > 1) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fb[p][st & 1][TN - 1]) : "v"(bd), "n"(off + 128));
        };
        auto mma = [&](auto pc, auto stc) {
            constexpr int p = decltype(pc)::value, st = decltype(stc)::value;
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) asm volatile("" : "+v"(fb[p][st & 1][jn]));
            if ((st & 3) == 0) {
#pragma unroll
                for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(fa[p][st >> 2][i]));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float af = fa[p][st >> 2][i][st & 3];
#pragma unroll
                for (int jn = 0; jn < TN; ++jn)
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, fb[p][st & 1][jn], acc[i][jn], 0, 0, 0);
            }
        };
        auto step = [&](auto pc, auto stc, int slot_b) { // k-steps 0 .. 5: request the B row of the next k-step, wait for this one's
            constexpr int st = decltype(stc)::value;
            read_b(pc, ConvIC<st + 1>{}, slot_b);
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(st == 0 ? TM + TN : TN) : "memory");
            mma(pc, stc);
        };
        auto iteration = [&](auto pc, int kt) {
            constexpr int p = decltype(pc)::value;
            if (kt + D - 1 < nk) load_b(bufb == 0 ? NB - 1 : bufb - 1);
            if (kt + D < nk) load_a(kt0 + kt + D, buf == 0 ? NA - 1 : buf - 1);
            const int sb = bufb;
            step(pc, ConvIC<0>{}, sb);
            step(pc, ConvIC<1>{}, sb);
            step(pc, ConvIC<2>{}, sb);
            step(pc, ConvIC<3>{}, sb);
            step(pc, ConvIC<4>{}, sb);
            step(pc, ConvIC<5>{}, sb);
            read_b(pc, ConvIC<7>{}, sb);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // every fragment of tile kt is in registers
            wait_next_tile(kt + D < nk);                        // tile kt + 1 has landed (this wavefront's pieces) ...
            __builtin_amdgcn_s_barrier();                       // ... everybody's have, and everybody is done reading tile kt
            buf = buf == NA - 1 ? 0 : buf + 1;
            bufb = bufb == NB - 1 ? 0 : bufb + 1;
            if (kt + 1 < nk) {
                read_a(ConvIC<1 - p>{}, ConvIC<0>{}, buf);
                read_b(ConvIC<1 - p>{}, ConvIC<0>{}, bufb);
                read_a(ConvIC<1 - p>{}, ConvIC<1>{}, buf);
            }
            mma(pc, ConvIC<6>{});
            mma(pc, ConvIC<7>{});
        };
        if (nk > 0) {
            read_a(ConvIC<0>{}, ConvIC<0>{}, 0);
            read_b(ConvIC<0>{}, ConvIC<0>{}, 0);
            read_a(ConvIC<0>{}, ConvIC<1>{}, 0);
        }
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            iteration(ConvIC<0>{}, kt);
            iteration(ConvIC<1>{}, kt + 1);
        }
        if (kt < nk) iteration(ConvIC<0>{}, kt);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + D - 1 < nk) load_b(bufb == 0 ? NB - 1 : bufb - 1);          // (kt + D - 1) % NB: read last in iteration kt - 1
        if (kt + D < nk) load_a(kt0 + kt + D, buf == 0 ? NA - 1 : buf - 1);  // (kt + D) % NA: likewise
        // Fragment reads run ahead of the MFMAs that consume them (B one k-step, A one j).  They are issued as asm
        // statements with hand-counted lgkmcnt waits: the compiler puts `s_waitcnt vmcnt(0)` in front of every LDS read
        // it can see while an LDS-DMA is in flight (it cannot tell that the DMA writes another ring slot), which would
        // drain the prefetch.  LDS returns in order, so "all but the newest N reads" is exactly lgkmcnt(N); the empty asm
        // statements after a wait make the consuming MFMAs depend on it.
        f32x4 fa[2 * KS][TM];
        const uint32_t a_slot = (uint32_t)buf * (BM * BK * 4);
        auto read_a = [&](auto jc) { // quad pair index q = 2 s + j
            constexpr int q = decltype(jc)::value;
            asm volatile("ds_read_b128 %0, %1" : "=v"(fa[q][0]) : "v"(a_lds[q] + a_slot));
            if (TM > 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[q][TM - 1]) : "v"(a_lds[q] + a_slot), "n"(32 * BK * 4));
        };
        if constexpr (P == 1) {
            // a bf16 k-step = 16 k: the lane's eight gathered values (two quads) and its hi / lo weight records; KS per k-tile
            u32x4 bh[KS][TN], bl[KS][TN];
            const uint32_t b_addr = b3_lds + (uint32_t)bufb * (BK * LDB * 4);
            // (explicit captures: with [&] clang rejects the asm operands inside this generic lambda -- "reference to local variable
            // declared in enclosing function" -- and with them it warns that they are not needed)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wunused-lambda-capture"
            auto read_b3 = [&bh, &bl, b_addr](auto sc, auto jc) {
                constexpr int st = decltype(sc)::value, jn = decltype(jc)::value;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bh[st][jn]) : "v"(b_addr), "n"(st * 4 * BN * 16 + jn * 512));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bl[st][jn]) : "v"(b_addr), "n"(st * 4 * BN * 16 + BN * 16 + jn * 512));
            };
#pragma clang diagnostic pop
            auto read_step = [&](auto sc) {
                constexpr int st = decltype(sc)::value;
                read_a(ConvIC<2 * st>{});
                read_a(ConvIC<2 * st + 1>{});
                read_b3(sc, ConvIC<0>{});
                if constexpr (TN > 1) read_b3(sc, ConvIC<1>{});
                if constexpr (TN > 2) read_b3(sc, ConvIC<2>{});
                if constexpr (TN > 3) read_b3(sc, ConvIC<3>{});
            };
            auto mma_step = [&](auto sc) {
                constexpr int st = decltype(sc)::value;
                constexpr int later = (KS - 1 - st) * (2 * TM + 2 * TN); // reads of the later k-steps still in flight
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(later) : "memory");
#pragma unroll
                for (int i = 0; i < TM; ++i) { asm volatile("" : "+v"(fa[2 * st][i])); asm volatile("" : "+v"(fa[2 * st + 1][i])); }
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) { asm volatile("" : "+v"(bh[st][jn])); asm volatile("" : "+v"(bl[st][jn])); }
                u32x4 ah[TM], al[TM];
#pragma unroll
                for (int i = 0; i < TM; ++i) conv_split8(fa[2 * st][i], fa[2 * st + 1][i], ah[i], al[i]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int jn = 0; jn < TN; ++jn) { // the small terms first
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[i]), __builtin_bit_cast(bf16x8, bh[st][jn]), acc[i][jn], 0, 0, 0);
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bl[st][jn]), acc[i][jn], 0, 0, 0);
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bh[st][jn]), acc[i][jn], 0, 0, 0);
                    }
            };
            read_step(ConvIC<0>{});
            if (KS > 1) read_step(ConvIC<KS - 1>{});
            mma_step(ConvIC<0>{});
            if (KS > 1) mma_step(ConvIC<KS - 1>{});
        } else {
            float fb[2][TN];
            const uint32_t b_addr = b_lds + (uint32_t)bufb * (BK * LDB * 4);
            auto read_b = [&](auto stc) {
                constexpr int st = decltype(stc)::value;
                constexpr int off = (8 * (st >> 2) + (st & 3)) * LDB * 4;
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fb[st & 1][0]) : "v"(b_addr), "n"(off));
                if (TN > 1) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fb[st & 1][TN - 1]) : "v"(b_addr), "n"(off + 128));
            };
            auto step = [&](auto stc) {
                constexpr int st = decltype(stc)::value;
                if (st + 1 < 8) {
                    read_b(ConvIC<(st + 1 < 8 ? st + 1 : 7)>{});
                    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(st == 0 ? TM + TN : TN) : "memory");
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) asm volatile("" : "+v"(fb[st & 1][jn]));
                if ((st & 3) == 0) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(fa[st >> 2][i]));
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const float af = fa[st >> 2][i][st & 3];
#pragma unroll
                    for (int jn = 0; jn < TN; ++jn)
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, fb[st & 1][jn], acc[i][jn], 0, 0, 0);
                }
            };
            read_a(ConvIC<0>{});
            read_b(ConvIC<0>{});
            read_a(ConvIC<1>{});
            step(ConvIC<0>{});
            step(ConvIC<1>{});
            step(ConvIC<2>{});
            step(ConvIC<3>{});
            step(ConvIC<4>{});
            step(ConvIC<5>{});
            step(ConvIC<6>{});
            step(ConvIC<7>{});
        }
        wait_next_tile(kt + D < nk);            // tile kt + 1 has landed (this wavefront's pieces) ...
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();           // ... everybody's have, and everybody is done reading tile kt
        buf = buf == NA - 1 ? 0 : buf + 1;
        bufb = bufb == NB - 1 ? 0 : bufb + 1;
    }

    PROBE_PH(3);
    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    if (a.splits > 1) { // raw partial sums (k_splitk_reduce applies bias / activation / residual), as 16-byte rows
        float *dst = a.partial + (long long)blockIdx.z * a.M * a.Npad;
        float *epw = smem + wv * 32 * EPLD;
        const int er = lane >> 3, ec = (lane & 7) * 4;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    epw[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * EPLD + (lane & 31)] = acc[i][j][r];
                asm volatile("" ::: "memory");
                const int n = n0 + wc * TN * 32 + 32 * j + ec;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int m = m0 + wr * TM * 32 + 32 * i + er + 8 * t;
                    if (m < a.M && n < a.Npad) {
                        const f32x4 v = *(const f32x4 *)&epw[(er + 8 * t) * EPLD + ec];
                        float *ptr = dst + (long long)m * a.Npad + n;
                        // SK: write-through (sc1) stores -- the reader is another workgroup, possibly on another XCD, whose L2
                        // is not coherent with this one's (MI355X_MICROARCH.md: inter-workgroup visibility)
                        if (SK) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(ptr), "v"(v) : "memory");
                        else *(f32x4 *)ptr = v;
                    }
                }
            }
        if constexpr (!SK) return;
        else {
            // ---- the split that arrives LAST at the tile's counter sums all splits (in split order, like k_splitk_reduce: the
            // same bits), applies bias / activation / residual and writes the tile; the others are done.  Protocol: every
            // wavefront drains its sc1 stores, workgroup barrier, ONE agent-scope atomic add; the add that returns splits - 1
            // has seen every other workgroup's add, hence (their drain + barrier in front of it) all their stores in memory;
            // its workgroup reads them with sc1 loads (L1 bypassed; no line of the partial buffer has been read in this launch,
            // so no L2 holds an old copy).  The counter goes back to 0 for the next launch.
            static_assert(TM == 1 && TN == 1, "the in-kernel reduction is written for one 32 x 32 tile per wavefront");
            __shared__ int s_last;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                int *cp = a.sk_counters + blockIdx.y * gridDim.x + blockIdx.x;
                const int old = __hip_atomic_fetch_add(cp, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int last = old == a.splits - 1;
                if (last) __hip_atomic_store(cp, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_last = last;
            }
            __syncthreads();
            if (!s_last) return;
            const int howo2 = a.Ho * a.Wo;
            const long long zstride = (long long)a.M * a.Npad;
            const int n = n0 + wc * 32 + ec;
            float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.bias && n < a.Cout) bias4 = *(const float4 *)(a.bias + n);
            float st_s[4] = {0.0f, 0.0f, 0.0f, 0.0f}, st_q[4] = {0.0f, 0.0f, 0.0f, 0.0f}; // a.stats: column sums of this lane's four rows
#pragma unroll
            for (int tp = 0; tp < 4 / SK_ROWS; ++tp) { // two of the thread's four rows at a time: 16 loads in flight (the registers of 32 would cost the main loop a wavefront per SIMD)
                f32x4 part[SK_ROWS][8];
#pragma unroll
                for (int tt = 0; tt < SK_ROWS; ++tt) {
                    const int m = m0 + wr * 32 + er + 8 * (SK_ROWS * tp + tt);
                    const float *src = a.partial + (long long)(m < a.M ? m : 0) * a.Npad + (n < a.Npad ? n : 0);
#pragma unroll
                    for (int z = 0; z < 8; ++z)
                        if (z < a.splits) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(part[tt][z]) : "v"(src + z * zstride) : "memory");
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int tt = 0; tt < SK_ROWS; ++tt) {
                    const int m = m0 + wr * 32 + er + 8 * (SK_ROWS * tp + tt);
#pragma unroll
                    for (int z = 0; z < 8; ++z) asm volatile("" : "+v"(part[tt][z]));
                    float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int z = 0; z < 8; ++z)
                        if (z < a.splits) { v[0] += part[tt][z][0]; v[1] += part[tt][z][1]; v[2] += part[tt][z][2]; v[3] += part[tt][z][3]; }
                    if (a.stats && m < a.M) { // (workgroup-uniform pointer; train forward: no bias, no activation -- v is z)
#pragma unroll
                        for (int e = 0; e < 4; ++e) { st_s[e] += v[e]; st_q[e] += v[e] * v[e]; }
                    }
                    if (m < a.M && n < a.Cout) {
                        const float bb[4] = {bias4.x, bias4.y, bias4.z, bias4.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int act = (a.act == ACT_SIGMOID && n + e < a.sig_from) ? ACT_NONE : a.act;
                            v[e] = act_apply(v[e] + bb[e], act);
                        }
                        const int b = m / howo2, pix = m - b * howo2;
                        if (a.res) {
                            const float4 rr = *(const float4 *)(a.res + (long long)b * a.r_bs + (long long)pix * a.r_cs + a.r_co + n);
                            v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
                        }
                        long long yo = (long long)pix * a.y_cs;
                        if (a.y_rp) { const int oy = pix / a.Wo; yo = (long long)oy * a.y_rp + (long long)(pix - oy * a.Wo) * a.y_cs; } // (a parity class of a stride-2 data gradient)
                        *(float4 *)(a.y + (long long)b * a.y_bs + yo + a.y_co + n) = make_float4(v[0], v[1], v[2], v[3]);
                    }
                }
            }
            if (a.stats) {
                // BatchNorm statistics of the train step from the finished tile, like the unsplit epilogue below: float32 inside a
                // wavefront's 32 rows (four rows per lane, then the eight lanes that share a column quad), float64 across the two
                // wavefronts of a column half and in the stored slab (one slab per 64-row tile) -- k_bn_stats_partial goes away
                // for the layers that split their contraction
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int off = 8; off <= 32; off <<= 1) { st_s[e] += __shfl_xor(st_s[e], off); st_q[e] += __shfl_xor(st_q[e], off); }
                float2 *stg = (float2 *)smem; // [wavefront][32 columns] (the operand tiles are free: every wavefront is past the barrier above)
                if (lane < 8) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) stg[wv * 32 + ec + e] = make_float2(st_s[e], st_q[e]);
                }
                __syncthreads();
                if (wr == 0 && lane < 32) {
                    const float2 v0 = stg[(0 * WCOLS + wc) * 32 + lane], v1 = stg[(1 * WCOLS + wc) * 32 + lane];
                    const int nn = n0 + wc * 32 + lane;
                    if (nn < a.Cout) *(double2 *)(a.stats + ((long long)blockIdx.x * a.Cout + nn) * 2) = make_double2((double)v0.x + (double)v1.x, (double)v0.y + (double)v1.y);
                }
            }
            return;
        }
    }
    const int howo = a.Ho * a.Wo;
    if (((a.Cout | a.y_cs | a.y_co | a.r_cs | a.r_co) & 3) == 0 && (a.y_bs & 3) == 0 && (a.r_bs & 3) == 0 && a.y_rp == 0) {
        // Vector epilogue: every 32 x 32 accumulator tile goes through a wavefront-private LDS staging area and leaves
        // as 16-byte rows (4 consecutive channels per lane): 4 stores per lane and tile instead of 16.  The epilogue
        // runs with no MFMA left to overlap (all workgroups of a layer finish together): it is store-ISSUE bound.
        float *epw = smem + wv * 32 * EPLD; // the last k-tile's barrier has freed the operand tiles
        const int er = lane >> 3, ec = (lane & 7) * 4; // this lane's rows er, er + 8, er + 16, er + 24; columns ec..ec+3
        if (a.stats) { // (workgroup-uniform)
            // BatchNorm statistics of the train step straight from the accumulators: the column sums of the workgroup's BM rows
            // (rows past M are zero) -- float32 inside a wavefront's 32 * TM rows, float64 across the wavefronts and in the
            // stored slab -- so the separate pass over z (k_bn_stats_partial) goes away
            float2 *stg = (float2 *)smem; // [wavefront][TN * 32 columns]: the operand tiles are free after the last barrier
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float cs = 0.0f, cq = 0.0f;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { cs += acc[i][j][r]; cq += acc[i][j][r] * acc[i][j][r]; }
                cs += __shfl_xor(cs, 32);
                cq += __shfl_xor(cq, 32);
                if (lane < 32) stg[(wv * TN + j) * 32 + lane] = make_float2(cs, cq);
            }
            __syncthreads();
            if (wr == 0 && lane < 32) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    double ds = 0.0, dq = 0.0;
#pragma unroll
                    for (int w = 0; w < WROWS; ++w) { // wavefront (w, wc) = w * WCOLS + wc
                        const float2 v = stg[((w * WCOLS + wc) * TN + j) * 32 + lane];
                        ds += (double)v.x;
                        dq += (double)v.y;
                    }
                    const int n = n0 + wc * TN * 32 + 32 * j + lane;
                    if (n < a.Cout) *(double2 *)(a.stats + ((long long)blockIdx.x * a.Cout + n) * 2) = make_double2(ds, dq);
                }
            }
            __syncthreads(); // the staging area below overlaps stg
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int tm0 = m0 + wr * TM * 32 + 32 * i;
            const int mfirst = tm0 + er;
            const int b0 = mfirst / howo, pix0 = mfirst - b0 * howo; // one division per tile row group
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int tn0 = n0 + wc * TN * 32 + 32 * j;
                asm volatile("" ::: "memory");
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    epw[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * EPLD + (lane & 31)] = acc[i][j][r];
                asm volatile("" ::: "memory");
                const int n = tn0 + ec;
                if (n < a.Cout) {
                    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (a.bias) bias = *(const float4 *)(a.bias + n);
                    int b = b0, pix = pix0;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float4 v4 = *(const float4 *)&epw[(er + 8 * t) * EPLD + ec];
                        if (mfirst + 8 * t < a.M) {
                            float o[4] = {v4.x + bias.x, v4.y + bias.y, v4.z + bias.z, v4.w + bias.w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int act = (a.act == ACT_SIGMOID && n + e < a.sig_from) ? ACT_NONE : a.act;
                                o[e] = act_apply(o[e], act);
                            }
                            if (a.res) {
                                const float4 rr = *(const float4 *)(a.res + (long long)b * a.r_bs + (long long)pix * a.r_cs + a.r_co + n);
                                o[0] += rr.x; o[1] += rr.y; o[2] += rr.z; o[3] += rr.w;
                            }
                            *(float4 *)(a.y + (long long)b * a.y_bs + (long long)pix * a.y_cs + a.y_co + n) = make_float4(o[0], o[1], o[2], o[3]);
                            if (a.y2) { // (workgroup-uniform)
                                const int oy = pix / a.Wo, ox = pix - oy * a.Wo;
                                float *u = a.y2 + (long long)b * a.y2_bs + ((long long)(2 * oy) * (2 * a.Wo) + 2 * ox) * a.y2_cs + a.y2_co + n;
                                const float4 v = make_float4(o[0], o[1], o[2], o[3]);
                                *(float4 *)u = v;
                                *(float4 *)(u + a.y2_cs) = v;
                                *(float4 *)(u + (long long)2 * a.Wo * a.y2_cs) = v;
                                *(float4 *)(u + (long long)(2 * a.Wo + 1) * a.y2_cs) = v;
                            }
                        }
                        pix += 8;
                        while (pix >= howo) { pix -= howo; ++b; }
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int mrow0 = m0 + wr * TM * 32 + 32 * i + 4 * (lane >> 5); // first row of this lane in the tile
        const int b0 = mrow0 / howo, pix0 = mrow0 - b0 * howo;          // one division per 32x32 tile
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wc * TN * 32 + 32 * j + (lane & 31);
            if (n >= a.Cout) continue;
            const float bias = a.bias ? a.bias[n] : 0.0f;
            const int act = (a.act == ACT_SIGMOID && n < a.sig_from) ? ACT_NONE : a.act;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dr = (r & 3) + 8 * (r >> 2);
                if (mrow0 + dr < a.M) {
                    int b = b0, pix = pix0 + dr;
                    while (pix >= howo) { pix -= howo; ++b; }
                    float v = act_apply(acc[i][j][r] + bias, act);
                    if (a.res) v = v + a.res[(long long)b * a.r_bs + (long long)pix * a.r_cs + a.r_co + n];
                    long long yo = (long long)pix * a.y_cs;
                    if (a.y_rp) { const int oy = pix / a.Wo; yo = (long long)oy * a.y_rp + (long long)(pix - oy * a.Wo) * a.y_cs; }
                    a.y[(long long)b * a.y_bs + yo + a.y_co + n] = v;
                    if (a.y2) {
                        const int oy = pix / a.Wo, ox = pix - oy * a.Wo;
                        float *u = a.y2 + (long long)b * a.y2_bs + ((long long)(2 * oy) * (2 * a.Wo) + 2 * ox) * a.y2_cs + a.y2_co + n;
                        u[0] = v; u[a.y2_cs] = v; u[(long long)2 * a.Wo * a.y2_cs] = v; u[(long long)(2 * a.Wo + 1) * a.y2_cs] = v;
                    }
                }
            }
        }
    }
}

#ifdef FRLW_DEV_BUILD // lab-only (tools/conv_lab.hip, FRLW_CONV_THIN16=1): measured in round 3, no net gain -- DESIGN.md section 4
// ---- the thin-layer variant: 64 x 32 output tile, four wavefronts of 32 x 16 on v_mfma_f32_16x16x4_f32 -------------------
// A layer of M x N outputs is M * N / 1024 wavefronts with the 32 x 32 MFMA (one accumulator tile per wavefront at least):
// the detector's 16 x 20 and 8 x 10 levels give 1.25-2.5 wavefronts per SIMD, so the matrix pipes idle on quantisation and
// there is nobody to hide a workgroup's prologue, barriers and epilogue.  The 16 x 16 x 4 instruction (same FLOP per cycle,
// 40-cycle dependent latency) lets a wavefront own 32 x 16: twice the wavefronts for the same layer, 8 workgroups per CU.
// Staging, ring, counted waits: as k_conv_mfma.  Fragments: lane l = (m = l & 15, kk = l >> 4) supplies A[row m][k] and
// B[k][col m] with k = 4 kk + t in k-step t (the instruction sums over the four lane groups, so a k-step covers
// k = t, 4 + t, 8 + t, 12 + t: any partition of the tile's 16 k works as long as both operands use it) -- the lane's four A
// values are ONE 16-byte read.  LDS images (both arrive by DMA, lane-linear, so the layout is chosen on the SOURCE side):
// row r of the gathered tile holds quad q in slot q ^ h((r >> 2) & 3), h = {0, 2, 3, 1}: the four 16-lane groups of a
// ds_read_b128 then touch 16 distinct bank quads; row k of the weight tile has its two 16-column halves swapped when
// (k >> 2) is odd, so the 32 lanes of a ds_read_b32 pass (kk = 0, 1 or 2, 3) hit 32 distinct banks.
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <bool UT>
__global__ __launch_bounds__(256) void k_conv_mfma16(ConvArgs a)
{
    constexpr int BM = 64, BN = 32, BK = 16, D = 2;
    constexpr int KQ = BK / 4, RPP = 256 / KQ, LDB = BN;
    constexpr int A_F4 = BM * BK / 4 / 256; // 1
    constexpr int NA = D + 1, NB = D;
    constexpr int kTileFloats = BK * (NA * BM + NB * LDB);
    __shared__ __attribute__((aligned(16))) float smem[kTileFloats];
    float (*As)[BM][BK] = (float (*)[BM][BK])smem;
    float (*Bs)[BK][LDB] = (float (*)[BK][LDB])(smem + NA * BK * BM);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = wv >> 1, wc = wv & 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const __amdgpu_buffer_rsrc_t rx = conv_rsrc(a.x, a.x_bytes), rw = conv_rsrc(a.w, a.w_bytes);
    const auto h4 = [](int g) { return g == 0 ? 0 : (g == 1 ? 2 : (g == 2 ? 3 : 1)); }; // h = {0, 2, 3, 1}

    // ---- A staging: thread -> one row m, one float4 of 4 consecutive k (the quad that lands in slot tid & 3 of its row)
    const int a_k4 = ((tid & 3) ^ h4((tid >> 4) & 3)) * 4;
    int a_iy0, a_ix0;
    uint32_t a_base;
    {
        const int m = m0 + tid / KQ;
        const int mm = m < a.M ? m : 0;
        const int b = mm / (a.Ho * a.Wo), pix = mm - b * (a.Ho * a.Wo);
        const int oy = pix / a.Wo, ox = pix - oy * a.Wo;
        a_iy0 = oy * a.stride - a.pad;
        a_ix0 = ox * a.stride - a.pad;
        a_base = m < a.M ? (uint32_t)(((long long)b * a.x_bs + a.x_co + (a.group_n ? (n0 / a.group_n) * a.Cin : 0)) * 4) : kOob;
    }
    const int kw = a.kw ? a.kw : a.k;
    const int x_cs4 = a.x_cs * 4;
    auto gather_off = [&](int ky, int kx) -> uint32_t {
        int iy = a_iy0 + ky, ix = a_ix0 + kx;
        bool ok = a_base != kOob;
        if (a.tstride == 2) { ok = ok && !((iy | ix) & 1); iy >>= 1; ix >>= 1; }
        ok = ok && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        return ok ? a_base + (uint32_t)((iy * a.W + ix) * x_cs4) : kOob;
    };
    // ---- B staging: the first two wavefronts, one float4 each: LDS position (row kr, columns n4 .. n4 + 3) receives the
    // weights of columns n4 ^ 16 when (kr >> 2) is odd
    const int nk_all = (a.K + BK - 1) / BK;
    const int kt0 = (int)((long long)nk_all * blockIdx.z / a.splits), kt1 = (int)((long long)nk_all * (blockIdx.z + 1) / a.splits);
    const int nk = kt1 - kt0;
    uint32_t b_off;
    {
        const int kr = tid / (BN / 4), n4 = ((tid % (BN / 4)) * 4) ^ (((kr >> 2) & 1) * 16);
        b_off = (tid < BN * BK / 4 && n0 + n4 < a.Npad) ? (uint32_t)((((long long)kt0 * BK + kr) * a.Npad + n0 + n4) * 4) : kOob;
    }
    int s_ci, s_ky, s_kx;
    uint32_t a_pix = kOob;
    {
        const int k = kt0 * BK + (UT ? 0 : a_k4);
        const int tap = k / a.Cin;
        s_ci = k - tap * a.Cin;
        s_ky = tap / kw;
        s_kx = tap - s_ky * kw;
    }
    if (UT) a_pix = gather_off(s_ky, s_kx);
    auto load_a = [&](int kt, int nbuf) {
        if (UT) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void *)(&As[nbuf][0][0] + wv * 64 * 4), 16,
                                                     (int)(a_pix + (uint32_t)(s_ci + a_k4) * 4), 0, 0, 0);
            s_ci += BK;
            if (s_ci == a.Cin) {
                s_ci = 0;
                if (++s_kx == kw) { s_kx = 0; ++s_ky; }
                a_pix = gather_off(s_ky, s_kx);
            }
        } else {
            const bool in_k = kt * BK + a_k4 < a.K;
            const uint32_t o = gather_off(s_ky, s_kx);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void *)(&As[nbuf][0][0] + wv * 64 * 4), 16,
                                                     (int)(in_k ? o + (uint32_t)s_ci * 4 : kOob), 0, 0, 0);
            s_ci += BK;
            while (s_ci >= a.Cin) { s_ci -= a.Cin; if (++s_kx == kw) { s_kx = 0; ++s_ky; } }
        }
    };
    auto load_b = [&](int nbuf) {
        if (wv < 2) // wave-uniform: 128 float4 = two wave-instructions
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void *)(&Bs[nbuf][0][0] + wv * 64 * 4), 16, (int)b_off, 0, 0, 0);
        b_off += (uint32_t)(BK * 4) * (uint32_t)a.Npad;
    };

    f32x4v acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][r] = 0.0f;

    // issue order per iteration: weights of tile t + 1, then rows of tile t + 2 (as k_conv_mfma with D = 2): "all but my
    // newest one DMA instruction has landed" = "tile t + 1 is complete" for wavefronts 0 and 1 as well as 2 and 3 (which
    // issue no weight DMA: their newest instruction is the same row load)
    auto wait_next_tile = [&](bool steady) {
        if (steady) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    load_a(kt0, 0);
    if (0 < nk) load_b(0);
    if (1 < nk) load_a(kt0 + 1, 1);
    wait_next_tile(nk >= D);
    __builtin_amdgcn_s_barrier();
    const int fm = lane & 15, fkk = lane >> 4;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)smem;
    // byte addresses in ring slot 0: A row (32 wr + 16 i + fm), slot fkk ^ h(fm >> 2); B row 4 fkk (+ t), column half wc ^ (fkk & 1)
    const uint32_t a_lds = lds0 + (uint32_t)((32 * wr + fm) * BK + ((fkk ^ h4((fm >> 2) & 3)) * 4)) * 4;
    const uint32_t b_lds = lds0 + (uint32_t)(NA * BK * BM + 4 * fkk * LDB + 16 * (wc ^ (fkk & 1)) + fm) * 4;
    int buf = 0, bufb = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + D - 1 < nk) load_b(bufb == 0 ? NB - 1 : bufb - 1);
        if (kt + D < nk) load_a(kt0 + kt + D, buf == 0 ? NA - 1 : buf - 1);
        f32x4 fa0, fa1;
        float fb0, fb1, fb2, fb3;
        const uint32_t a_addr = a_lds + (uint32_t)buf * (BM * BK * 4), b_addr = b_lds + (uint32_t)bufb * (BK * LDB * 4);
        // (asm reads with a hand-counted wait: the compiler would put s_waitcnt vmcnt(0) in front of LDS reads it can see
        // while an LDS-DMA is in flight and drain the prefetch)
        asm volatile("ds_read_b128 %0, %1" : "=v"(fa0) : "v"(a_addr));
        asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(fa1) : "v"(a_addr)); // + 16 rows
        asm volatile("ds_read_b32 %0, %1" : "=v"(fb0) : "v"(b_addr));
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fb1) : "v"(b_addr), "n"(1 * LDB * 4));
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fb2) : "v"(b_addr), "n"(2 * LDB * 4));
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fb3) : "v"(b_addr), "n"(3 * LDB * 4));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(fa0), "+v"(fa1), "+v"(fb0), "+v"(fb1), "+v"(fb2), "+v"(fb3));
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa0[0], fb0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1[0], fb0, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa0[1], fb1, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1[1], fb1, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa0[2], fb2, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1[2], fb2, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa0[3], fb3, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1[3], fb3, acc[1], 0, 0, 0);
        wait_next_tile(kt + D < nk);
        __builtin_amdgcn_s_barrier();
        buf = buf == NA - 1 ? 0 : buf + 1;
        bufb = bufb == NB - 1 ? 0 : bufb + 1;
    }

    // ---- epilogue: C/D layout of the 16 x 16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + r
    const int n = n0 + 16 * wc + fm;
    const int howo = a.Ho * a.Wo;
    if (a.splits > 1) {
        float *dst = a.partial + (long long)blockIdx.z * a.M * a.Npad;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 32 * wr + 16 * i + 4 * fkk + r;
                if (m < a.M && n < a.Npad) dst[(long long)m * a.Npad + n] = acc[i][r];
            }
        return;
    }
    if (n >= a.Cout) return;
    const float bias = a.bias ? a.bias[n] : 0.0f;
    const int act = (a.act == ACT_SIGMOID && n < a.sig_from) ? ACT_NONE : a.act;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int mrow0 = m0 + 32 * wr + 16 * i + 4 * fkk;
        const int b0 = mrow0 / howo, pix0 = mrow0 - b0 * howo;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (mrow0 + r < a.M) {
                int b = b0, pix = pix0 + r;
                while (pix >= howo) { pix -= howo; ++b; }
                float v = act_apply(acc[i][r] + bias, act);
                if (a.res) v = v + a.res[(long long)b * a.r_bs + (long long)pix * a.r_cs + a.r_co + n];
                long long yo = (long long)pix * a.y_cs;
                if (a.y_rp) { const int oy = pix / a.Wo; yo = (long long)oy * a.y_rp + (long long)(pix - oy * a.Wo) * a.y_cs; }
                a.y[(long long)b * a.y_bs + yo + a.y_co + n] = v;
            }
        }
    }
}

#endif // FRLW_DEV_BUILD

// y = act(sum over splits of partial + bias) [+ res].  VEC: four consecutive channels per thread (16-byte loads of every
// split's partial row, one 16-byte store); the partial sums are added in split order, as the scalar form does.
template <bool VEC>
__global__ __launch_bounds__(256) void k_splitk_reduce(ConvArgs a)
{
    constexpr int V = VEC ? 4 : 1;
    const int cv = a.Cout / V;
    const long long total = (long long)a.M * cv;
    const int howo = a.Ho * a.Wo;
    const long long zstride = (long long)a.M * a.Npad;
    for (long long o = blockIdx.x * (long long)blockDim.x + threadIdx.x; o < total; o += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(o / cv);
        const int n = (int)(o - (long long)m * cv) * V;
        const float *p = a.partial + (long long)m * a.Npad + n;
        float v[V];
#pragma unroll
        for (int e = 0; e < V; ++e) v[e] = 0.0f;
        if (VEC) {
            float4 t[8];
#pragma unroll
            for (int z = 0; z < 8; ++z)
                if (z < a.splits) t[z] = *(const float4 *)(p + z * zstride);
#pragma unroll
            for (int z = 0; z < 8; ++z)
                if (z < a.splits) { v[0] += t[z].x; v[1 % V] += t[z].y; v[2 % V] += t[z].z; v[3 % V] += t[z].w; }
        } else {
            for (int z = 0; z < a.splits; ++z) v[0] += p[z * zstride];
        }
        const int b = m / howo, pix = m - b * howo;
        long long yo = (long long)pix * a.y_cs;
        if (a.y_rp) { const int oy = pix / a.Wo; yo = (long long)oy * a.y_rp + (long long)(pix - oy * a.Wo) * a.y_cs; }
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const int act = (a.act == ACT_SIGMOID && n + e < a.sig_from) ? ACT_NONE : a.act;
            v[e] = act_apply(v[e] + (a.bias ? a.bias[n + e] : 0.0f), act);
        }
        if (VEC) {
            if (a.res) {
                const float4 rr = *(const float4 *)(a.res + (long long)b * a.r_bs + (long long)pix * a.r_cs + a.r_co + n);
                v[0] += rr.x; v[1 % V] += rr.y; v[2 % V] += rr.z; v[3 % V] += rr.w;
            }
            *(float4 *)(a.y + (long long)b * a.y_bs + yo + a.y_co + n) = make_float4(v[0], v[1 % V], v[2 % V], v[3 % V]);
        } else {
            if (a.res) v[0] += a.res[(long long)b * a.r_bs + (long long)pix * a.r_cs + a.r_co + n];
            a.y[(long long)b * a.y_bs + yo + a.y_co + n] = v[0];
        }
    }
}

inline int conv_grid_1d(long long n) { long long g = (n + 255) / 256; if (g > 4096) g = 4096; if (g < 1) g = 1; return (int)g; }

// the split image of a [K][Npad] float32 operand for prec = 1 (layout: see conv_split_kmem); one thread per 16-byte record
__global__ __launch_bounds__(256) void k_conv_split_operand(const float *w, int K, int Npad, uint4 *out)
{
    const long long nrec = (long long)((K + 15) / 16) * 4 * Npad;
    for (long long r = blockIdx.x * (long long)blockDim.x + threadIdx.x; r < nrec; r += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(r % Npad);
        const int hp = (int)((r / Npad) & 3), h = hp >> 1, part = hp & 1;
        const long long kt = r / (4ll * Npad);
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const long long k = kt * 16 + conv_split_kmem(h, 2 * e + q);
                const float x = k < K ? w[k * Npad + n] : 0.0f;
                const float hi = __builtin_bit_cast(float, conv_bf16_pair(x, 0.0f) << 16);
                v[q] = part ? x - hi : x;
            }
            o[e] = conv_bf16_pair(v[0], v[1]);
        }
        out[r] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

template <int BM, int BN, int WROWS, int WCOLS, int BK, int D = 2>
inline void launch_conv_tile(const ConvArgs &c, dim3 grid, hipStream_t s)
{
    const dim3 block(64 * WROWS * WCOLS);
    if (c.prec == 1) {
#ifdef FRLW_DEV_BUILD // lab: k-tiles of 32 (two bf16 k-steps per barrier) -- measured, no net gain: DESIGN.md section 4
        static const long long bk32 = dev_knob("FRLW_CONV_BK32", 0ll); // bit 0: 64 x 64 tiles, bit 1: 64 x 128, bit 2: 128 x 128, bit 3: others
        const int bit = BM == 64 ? (BN == 64 ? 1 : 2) : (BN == 128 ? 4 : 8);
        if ((bk32 & bit) && c.Cin % 32 == 0) {
            hipLaunchKernelGGL((k_conv_mfma<BM, BN, WROWS, WCOLS, 32, true, D, 1>), grid, block, 0, s, c);
            return;
        }
#endif
        if (c.Cin % BK == 0) hipLaunchKernelGGL((k_conv_mfma<BM, BN, WROWS, WCOLS, BK, true, D, 1>), grid, block, 0, s, c);
        else hipLaunchKernelGGL((k_conv_mfma<BM, BN, WROWS, WCOLS, BK, false, D, 1>), grid, block, 0, s, c);
        return;
    }
    if (c.Cin % BK == 0) hipLaunchKernelGGL((k_conv_mfma<BM, BN, WROWS, WCOLS, BK, true, D>), grid, block, 0, s, c);
    else hipLaunchKernelGGL((k_conv_mfma<BM, BN, WROWS, WCOLS, BK, false, D>), grid, block, 0, s, c);
}

// tiles that exist with the bf16 k-steps only
template <int BM, int BN, int WROWS, int WCOLS>
inline void launch_conv_tile_p1(const ConvArgs &c, dim3 grid, hipStream_t s)
{
    const dim3 block(64 * WROWS * WCOLS);
    if (c.Cin % 16 == 0) hipLaunchKernelGGL((k_conv_mfma<BM, BN, WROWS, WCOLS, 16, true, 2, 1>), grid, block, 0, s, c);
    else hipLaunchKernelGGL((k_conv_mfma<BM, BN, WROWS, WCOLS, 16, false, 2, 1>), grid, block, 0, s, c);
}

// Tile choice and split-K for one convolution; `scratch` (scratch_floats floats, may be NULL) holds split-K partials.
// Returns false (nothing launched) when a single image's view exceeds the 32-bit buffer offsets.
inline bool launch_conv(ConvArgs &c, float *scratch, long long scratch_floats, hipStream_t s, int *sk_counters = nullptr)
{
    const int howo = c.Ho * c.Wo, nb = c.M / howo;
    const int n_groups = c.group_n ? (c.Npad + c.group_n - 1) / c.group_n : 1;
    const long long x_bytes = (((long long)(nb - 1) * c.x_bs) + ((long long)c.H * c.W - 1) * c.x_cs + c.x_co + (long long)c.Cin * n_groups) * 4;
    if (x_bytes > kMaxViewBytes && nb > 1 && c.M == nb * howo) { // 32-bit buffer offsets: run the batch in two halves
        ConvArgs h = c;
        h.stats = nullptr; // (two launches would write the same slabs: the caller runs its own statistics pass)
        c.stats = nullptr;
        c.stats_rows = 0;
        const int b0 = nb / 2;
        h.M = b0 * howo;
        if (!launch_conv(h, scratch, scratch_floats, s, sk_counters)) return false;
        h = c;
        h.stats = nullptr;
        h.M = (nb - b0) * howo;
        h.x = c.x + (long long)b0 * c.x_bs;
        h.y = c.y + (long long)b0 * c.y_bs;
        if (c.y2) h.y2 = c.y2 + (long long)b0 * c.y2_bs;
        if (c.res) h.res = c.res + (long long)b0 * c.r_bs;
        return launch_conv(h, scratch, scratch_floats, s, sk_counters);
    }
    if (x_bytes > kMaxViewBytes || (long long)c.K * c.Npad * 4 > kMaxViewBytes) return false;
    c.x_bytes = (uint32_t)x_bytes;
    c.w_bytes = (uint32_t)((long long)(c.prec == 1 ? (c.K + 15) / 16 * 16 : c.K) * c.Npad * 4);
    c.splits = 1;
    c.partial = nullptr;
    c.sk_counters = nullptr;
    double *const stats_req = c.stats; // wanted by the caller; granted per tile below (never with split-K or the scalar epilogue)
    c.stats = nullptr;
    c.stats_rows = 0;
    const bool stats_ok = stats_req && ((c.Cout | c.y_cs | c.y_co | c.r_cs | c.r_co) & 3) == 0 && (c.y_bs & 3) == 0 && (c.r_bs & 3) == 0 && c.y_rp == 0;
    auto grant_stats = [&](int bm) { if (stats_ok && c.splits == 1) { c.stats = stats_req; c.stats_rows = (c.M + bm - 1) / bm; } };
    const long long big = (long long)((c.M + 127) / 128) * ((c.Npad + 127) / 128);
    static const long long split_below = dev_knob("FRLW_CONV_SPLIT_BELOW", 700ll);
    static const long long split_target = dev_knob("FRLW_CONV_SPLIT_TARGET", 1280ll);
    static const long long big_min = dev_knob("FRLW_CONV_BIG_MIN", 1200ll);
    static const long long wide_min = dev_knob("FRLW_CONV_WIDE_MIN", 1200ll);
    static const long long t256_min = dev_knob("FRLW_CONV_T256_MIN", 100000ll);
    static const long long row4_min = dev_knob("FRLW_CONV_ROW4_MIN", 600ll);
    (void)t256_min;
#ifdef FRLW_DEV_BUILD
    if (c.prec == 1 && c.Npad >= 256 && (long long)((c.M + 127) / 128) * ((c.Npad + 255) / 256) >= t256_min) {
        // lab: 128 x 256 on eight wavefronts (half the L2 requests per FLOP of the 128 x 128 tile): 264 us against 204 on the
        // 40960 x 256 x 2304 layer -- the requests were not the limit, the issue slots of the SIMDs were
        launch_conv_tile_p1<128, 256, 2, 4>(c, dim3((c.M + 127) / 128, (c.Npad + 255) / 256), s);
    } else
#endif
    if (c.Npad <= 32) { // small N (prediction convs, the stem's data gradient)
        grant_stats(128);
        launch_conv_tile<128, 32, 4, 1, 16>(c, dim3((c.M + 127) / 128, 1), s);
    } else if (c.prec == 1 && c.Npad >= 128 && big >= row4_min && c.K >= 512) {
        // bf16 k-steps: the four wavefronts side by side in M, each 32 rows x 128 columns -- a wavefront splits its gathered
        // values (24 VALU instructions per k-step) once for FOUR column tiles; with 2 x 2 the split cost as much issue time as the MFMAs
        grant_stats(128);
        launch_conv_tile_p1<128, 128, 4, 1>(c, dim3((c.M + 127) / 128, (c.Npad + 127) / 128), s);
    } else if (big >= big_min && c.Npad >= 128) {
        grant_stats(128);
        launch_conv_tile<128, 128, 2, 2, CONV_BK_BIG>(c, dim3((c.M + 127) / 128, (c.Npad + 127) / 128), s);
    } else if (c.Npad >= 128 && (long long)((c.M + 63) / 64) * ((c.Npad + 127) / 128) >= wide_min) {
        // 64 x 128: half the im2col gathers per output of the 64 x 64 tile, still > 4 workgroups per CU
        grant_stats(64);
        launch_conv_tile<64, 128, 2, 2, 16>(c, dim3((c.M + 63) / 64, (c.Npad + 127) / 128), s);
    } else {
        static const long long w2 = dev_knob("FRLW_CONV_W2", 0ll); // bf16 k-steps, two wavefronts of 32 x 64 (bit 0) / 32 x 128 (bit 1) per workgroup
        static const long long ring4 = dev_knob("FRLW_CONV_RING4", 0ll), ring3 = dev_knob("FRLW_CONV_RING3", 0ll);
        const bool w2n128 = c.prec == 1 && (w2 & 2) && c.Npad >= 128, w2n64 = c.prec == 1 && (w2 & 1) && !w2n128;
        (void)ring4; (void)ring3; (void)w2n64;
        const int bn = w2n128 ? 128 : 64;
        const long long wgs = (long long)((c.M + 63) / 64) * ((c.Npad + bn - 1) / bn);
        const int nk = (c.K + kSplitBK - 1) / kSplitBK;
        // small feature maps leave most CUs idle: split the contraction over blockIdx.z
        // (contractions shorter than 1024 gained nothing from splitting while the reduction was a launch of its own; with the
        // reduction inside the kernel -- sk_counters -- those of 512 and more do: detector forward 3.833 -> 3.782 ms, same box.
        // The upsampled copy of ConvArgs::y2 is written by the convolution's own epilogue: no split there.)
        static const long long split_min_nk = dev_knob("FRLW_CONV_SPLIT_MIN_NK", 0ll);
        const long long min_nk = split_min_nk ? split_min_nk : (sk_counters ? 32 : 64);
        if (wgs < split_below && nk >= min_nk && scratch && !c.y2) {
            int sp = (int)((split_target + wgs - 1) / wgs);
            if (sp > 8) sp = 8;
            static const long long split_min_per = dev_knob("FRLW_CONV_SPLIT_MIN_PER", 8ll); // k-tiles a split keeps at least
            if (sp > nk / split_min_per) sp = (int)(nk / split_min_per);
            if (sp > 1 && (long long)sp * c.M * c.Npad <= scratch_floats) { c.splits = sp; c.partial = scratch; }
        }
        // (a deeper ring, D = 4, for launches that leave a workgroup alone on its CU was measured: no gain -- such
        // workgroups are bound by the issue cost of their own DMA and fragment instructions, not by prefetch distance)
#ifdef FRLW_DEV_BUILD
        static const long long thin16 = dev_knob("FRLW_CONV_THIN16", 0ll); // lab: 1 = the 64 x 32 / 16x16x4 variant, no split-K
        if (thin16 && c.group_n == 0) {
            c.splits = 1; c.partial = nullptr; c.stats = nullptr; c.stats_rows = 0;
            if (c.Cin % 16 == 0) hipLaunchKernelGGL((k_conv_mfma16<true>), dim3((c.M + 63) / 64, (c.Npad + 31) / 32, 1), dim3(256), 0, s, c);
            else hipLaunchKernelGGL((k_conv_mfma16<false>), dim3((c.M + 63) / 64, (c.Npad + 31) / 32, 1), dim3(256), 0, s, c);
            return true;
        }
#endif
        grant_stats(64);
        const bool vec = ((c.Cout | c.Npad | c.y_cs | c.y_co | c.r_cs | c.r_co | c.y_rp) & 3) == 0 && (c.y_bs & 3) == 0 && (c.r_bs & 3) == 0;
        // the reduction inside the kernel (SK): float32 operand, uniform taps, vector rows, a counter per tile -- else k_splitk_reduce
        if (c.splits > 1 && sk_counters && vec && c.prec == 0 && c.Cin % CONV_BK_SMALL == 0 &&
            (long long)((c.M + 63) / 64) * ((c.Npad + 63) / 64) <= 1024) {
            c.sk_counters = sk_counters;
            if (stats_ok && !c.bias && c.act == ACT_NONE && !c.res) { c.stats = stats_req; c.stats_rows = (c.M + 63) / 64; } // the last arriver has the whole tile
            hipLaunchKernelGGL((k_conv_mfma_sk<CONV_BK_SMALL>), dim3((c.M + 63) / 64, (c.Npad + 63) / 64, c.splits), dim3(256), 0, s, c);
            return true;
        }
#ifdef FRLW_DEV_BUILD // lab variants, all measured without a net gain: two-wavefront workgroups, deeper rings
        if (w2n128) launch_conv_tile_p1<64, 128, 2, 1>(c, dim3((c.M + 63) / 64, (c.Npad + 127) / 128, c.splits), s);
        else if (w2n64) launch_conv_tile_p1<64, 64, 2, 1>(c, dim3((c.M + 63) / 64, (c.Npad + 63) / 64, c.splits), s);
        else if (c.prec == 1 && ring4) launch_conv_tile<64, 64, 2, 2, CONV_BK_SMALL, 4>(c, dim3((c.M + 63) / 64, (c.Npad + 63) / 64, c.splits), s);
        else if (c.prec == 1 && ring3) launch_conv_tile<64, 64, 2, 2, CONV_BK_SMALL, 3>(c, dim3((c.M + 63) / 64, (c.Npad + 63) / 64, c.splits), s);
        else
#endif
        launch_conv_tile<64, 64, 2, 2, CONV_BK_SMALL>(c, dim3((c.M + 63) / 64, (c.Npad + 63) / 64, c.splits), s);
        if (c.splits > 1) {
            if (vec) hipLaunchKernelGGL(k_splitk_reduce<true>, dim3(conv_grid_1d((long long)c.M * c.Cout / 4)), dim3(256), 0, s, c);
            else hipLaunchKernelGGL(k_splitk_reduce<false>, dim3(conv_grid_1d((long long)c.M * c.Cout)), dim3(256), 0, s, c);
        }
    }
    return true;
}
