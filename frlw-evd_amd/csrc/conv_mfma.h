// conv_mfma.h -- the fp32-MFMA implicit-GEMM convolution kernel shared by the detector plan (detector.hip) and the
// training operators (train_ops.hip).  Included inside each translation unit's anonymous namespace.
//
//   Y[M = B*Ho*Wo][N = Cout] = A[M][K] * W[K][N],  K = (ky, kx, ci),  A gathered on the fly from an NHWC view.
//   tstride = 2 turns the gather into that of a transposed (stride-2) convolution: input coordinate
//   (oy - pad + ky) / 2 when even, zero otherwise -- the data gradient of a stride-2 convolution.
// (no #includes here: the including file has <hip/hip_runtime.h>, <math.h>, <stdint.h>, <stdlib.h> already)

typedef float f32x16 __attribute__((ext_vector_type(16)));

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
    int y_rp;          // output row pitch in floats (0: dense, pixel p at p * y_cs); else pixel (oy, ox) at oy * y_rp + ox * y_cs
};

#ifndef CONV_BK_BIG
#define CONV_BK_BIG 16
#endif
#ifndef CONV_BK_SMALL
#define CONV_BK_SMALL 16
#endif
constexpr int kSplitBK = 16; // granularity the split-K heuristics count k-tiles in

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
template <int BM, int BN, int WROWS, int WCOLS, int BK, bool UT>
__global__ __launch_bounds__(256) void k_conv_mfma(ConvArgs a)
{
    constexpr int TM = BM / (32 * WROWS), TN = BN / (32 * WCOLS);
    constexpr int KQ = BK / 4;        // float4 per A row of the k-tile
    constexpr int RPP = 256 / KQ;     // A rows staged per pass of the 256 threads
    constexpr int LDA = BM + 4; // +4 floats: k rows land on different banks for the staging writes
    constexpr int LDB = BN;     // the weight tile arrives by LDS-DMA: lane-linear, [BK][BN] without padding
    constexpr int A_F4 = BM * BK / 4 / 256;   // float4 loads per thread for the A tile
    constexpr int B_F4 = (BN * BK / 4 + 255) / 256;
    constexpr int EPLD = 36;                          // row pitch of the epilogue staging (16-byte aligned rows)
    constexpr int kTileFloats = 2 * BK * (LDA + LDB);
    constexpr int kEpiFloats = 4 * 32 * EPLD;         // one 32 x 32 MFMA tile per wavefront
    __shared__ __attribute__((aligned(16))) float smem[kTileFloats > kEpiFloats ? kTileFloats : kEpiFloats];
    float (*As)[BK][LDA] = (float (*)[BK][LDA])smem;
    float (*Bs)[BK][LDB] = (float (*)[BK][LDB])(smem + 2 * BK * LDA);
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
    const int a_k4 = (tid % KQ) * 4;
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
        a_base[i] = m < a.M ? (uint32_t)(((long long)b * a.x_bs + a.x_co) * 4) : kOob;
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
    float4 ra[A_F4];

    const int nk_all = (a.K + BK - 1) / BK;
    const int kt0 = (int)((long long)nk_all * blockIdx.z / a.splits), kt1 = (int)((long long)nk_all * (blockIdx.z + 1) / a.splits);
    const int nk = kt1 - kt0;
#pragma unroll
    for (int i = 0; i < B_F4; ++i) {
        const int e = tid + 256 * i;
        const int kr = e / BN4, n4 = (e - kr * BN4) * 4;
        b_off[i] = (kr < BK && n0 + n4 < a.Npad) ? (uint32_t)((((long long)kt0 * BK + kr) * a.Npad + n0 + n4) * 4) : kOob;
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
    // The weight tile goes global -> LDS directly (buffer_load ... lds: no registers, no ds_write; wave-instruction i of
    // wavefront wv fills the 1 KB at float offset (wv * 64 + 256 i) * 4 of the tile).  The gathered A tile is staged
    // through registers: its LDS image is k-major, which no lane-linear DMA can produce.
    auto load_tiles = [&](int kt, int nbuf) {
        if (UT) {
            const uint32_t cib = (uint32_t)(s_ci + a_k4) * 4;
#pragma unroll
            for (int i = 0; i < A_F4; ++i) ra[i] = conv_load16(rx, a_pix[i] + cib); // kOob + cib stays out of range
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
                ra[i] = conv_load16(rx, in_k ? o + (uint32_t)s_ci * 4 : kOob);
            }
            s_ci += BK; // Cin % 4 == 0: a float4 never straddles taps
            while (s_ci >= a.Cin) { s_ci -= a.Cin; if (++s_kx == kw) { s_kx = 0; ++s_ky; } }
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            if ((BN * BK / 4) % 256 == 0 || wv * 64 + 256 * i < BN * BK / 4) // wave-uniform
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void *)(&Bs[nbuf][0][0] + (wv * 64 + 256 * i) * 4),
                                                         16, (int)b_off[i], 0, 0, 0);
            b_off[i] += (uint32_t)(BK * 4) * (uint32_t)a.Npad;
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int ml = tid / KQ + RPP * i;
            As[buf][a_k4 + 0][ml] = ra[i].x;
            As[buf][a_k4 + 1][ml] = ra[i].y;
            As[buf][a_k4 + 2][ml] = ra[i].z;
            As[buf][a_k4 + 3][ml] = ra[i].w;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    PROBE_PH(0);
    load_tiles(kt0, 0);
    PROBE_PH(1);
    store_tiles(0);
    __syncthreads();
    PROBE_PH(2);
    const int fm = wr * TM * 32 + (lane & 31), fn = wc * TN * 32 + (lane & 31), fk = lane >> 5;
#if defined(CONV_EXP) && CONV_EXP >= 4
    float fa[2][TM], fb[2][TN];
#endif
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
#if !defined(CONV_EXP) || CONV_EXP < 1
        if (kt + 1 < nk) load_tiles(kt0 + kt + 1, buf ^ 1);
#endif
        // Fragment reads run one k-step ahead of the MFMAs that consume them (two register sets): the matrix pipe never
        // waits for a full LDS round trip once the first pair has arrived.  The group barriers pin that order.
#if !defined(CONV_EXP) || CONV_EXP < 4
        float fa[2][TM], fb[2][TN];
#endif
        auto read_frags = [&](int st) {
#if defined(CONV_EXP) && CONV_EXP >= 4
            if (kt) return;
#endif
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[st & 1][i] = As[buf][2 * st + fk][fm + 32 * i];
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[st & 1][j] = Bs[buf][2 * st + fk][fn + 32 * j];
        };
        read_frags(0);
#pragma unroll
        for (int st = 0; st < BK / 2; ++st) {
            if (st + 1 < BK / 2) read_frags(st + 1);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[st & 1][i], fb[st & 1][j], acc[i][j], 0, 0, 0);
        }
#ifndef CONV_NO_SGB
        __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
#pragma unroll
        for (int st = 0; st < BK / 2; ++st) {
            if (st + 1 < BK / 2) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
        }
#endif
#if !defined(CONV_EXP) || CONV_EXP < 2
        if (kt + 1 < nk) store_tiles(buf ^ 1);
#endif
#if !defined(CONV_EXP) || CONV_EXP < 3
        __syncthreads();
#endif
    }

    PROBE_PH(3);
    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    if (a.splits > 1) { // raw partial sums; k_splitk_reduce applies bias / activation / residual
        float *dst = a.partial + (long long)blockIdx.z * a.M * a.Npad;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wc * TN * 32 + 32 * j + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wr * TM * 32 + 32 * i + 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2);
                    if (m < a.M && n < a.Npad) dst[(long long)m * a.Npad + n] = acc[i][j][r];
                }
            }
        return;
    }
    const int howo = a.Ho * a.Wo;
    if (((a.Cout | a.y_cs | a.y_co | a.r_cs | a.r_co) & 3) == 0 && (a.y_bs & 3) == 0 && (a.r_bs & 3) == 0 && a.y_rp == 0) {
        // Vector epilogue: every 32 x 32 accumulator tile goes through a wavefront-private LDS staging area and leaves
        // as 16-byte rows (4 consecutive channels per lane): 4 stores per lane and tile instead of 16.  The epilogue
        // runs with no MFMA left to overlap (all workgroups of a layer finish together): it is store-ISSUE bound.
        float *epw = smem + wv * 32 * EPLD; // the last k-tile's barrier has freed the operand tiles
        const int er = lane >> 3, ec = (lane & 7) * 4; // this lane's rows er, er + 8, er + 16, er + 24; columns ec..ec+3
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
                }
            }
        }
    }
}

// y = act(sum over splits of partial + bias) [+ res]
__global__ void k_splitk_reduce(ConvArgs a)
{
    const long long total = (long long)a.M * a.Cout;
    const int howo = a.Ho * a.Wo;
    for (long long o = blockIdx.x * (long long)blockDim.x + threadIdx.x; o < total; o += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(o % a.Cout);
        const int m = (int)(o / a.Cout);
        float v = 0.0f;
        for (int z = 0; z < a.splits; ++z) v += a.partial[((long long)z * a.M + m) * a.Npad + n];
        const int act = (a.act == ACT_SIGMOID && n < a.sig_from) ? ACT_NONE : a.act;
        v = act_apply(v + (a.bias ? a.bias[n] : 0.0f), act);
        const int b = m / howo, pix = m - b * howo;
        if (a.res) v = v + a.res[(long long)b * a.r_bs + (long long)pix * a.r_cs + a.r_co + n];
        long long yo = (long long)pix * a.y_cs;
        if (a.y_rp) { const int oy = pix / a.Wo; yo = (long long)oy * a.y_rp + (long long)(pix - oy * a.Wo) * a.y_cs; }
        a.y[(long long)b * a.y_bs + yo + a.y_co + n] = v;
    }
}

inline int conv_grid_1d(long long n) { long long g = (n + 255) / 256; if (g > 4096) g = 4096; if (g < 1) g = 1; return (int)g; }

template <int BM, int BN, int WROWS, int WCOLS, int BK>
inline void launch_conv_tile(const ConvArgs &c, dim3 grid, hipStream_t s)
{
    if (c.Cin % BK == 0) hipLaunchKernelGGL((k_conv_mfma<BM, BN, WROWS, WCOLS, BK, true>), grid, dim3(256), 0, s, c);
    else hipLaunchKernelGGL((k_conv_mfma<BM, BN, WROWS, WCOLS, BK, false>), grid, dim3(256), 0, s, c);
}

// Tile choice and split-K for one convolution; `scratch` (scratch_floats floats, may be NULL) holds split-K partials.
inline void launch_conv(ConvArgs &c, float *scratch, long long scratch_floats, hipStream_t s)
{
    const int howo = c.Ho * c.Wo, nb = c.M / howo;
    const long long x_bytes = (((long long)(nb - 1) * c.x_bs) + ((long long)c.H * c.W - 1) * c.x_cs + c.x_co + c.Cin) * 4;
    if (x_bytes > kMaxViewBytes && nb > 1 && c.M == nb * howo) { // 32-bit buffer offsets: run the batch in two halves
        ConvArgs h = c;
        const int b0 = nb / 2;
        h.M = b0 * howo;
        launch_conv(h, scratch, scratch_floats, s);
        h = c;
        h.M = (nb - b0) * howo;
        h.x = c.x + (long long)b0 * c.x_bs;
        h.y = c.y + (long long)b0 * c.y_bs;
        if (c.res) h.res = c.res + (long long)b0 * c.r_bs;
        launch_conv(h, scratch, scratch_floats, s);
        return;
    }
    c.x_bytes = (uint32_t)(x_bytes < kMaxViewBytes ? x_bytes : kMaxViewBytes);
    c.w_bytes = (uint32_t)((long long)c.K * c.Npad * 4);
    c.splits = 1;
    c.partial = nullptr;
    const long long big = (long long)((c.M + 127) / 128) * ((c.Npad + 127) / 128);
    static const long long split_below = dev_knob("FRLW_CONV_SPLIT_BELOW", 700ll);
    static const long long split_target = dev_knob("FRLW_CONV_SPLIT_TARGET", 1024ll);
    static const long long big_min = dev_knob("FRLW_CONV_BIG_MIN", 1000000ll);
    static const long long wide_min = dev_knob("FRLW_CONV_WIDE_MIN", 1200ll);
    if (c.Npad <= 32) { // small N (prediction convs, the stem's data gradient)
        launch_conv_tile<128, 32, 4, 1, 16>(c, dim3((c.M + 127) / 128, 1), s);
    } else if (big >= big_min && c.Npad >= 128) {
        launch_conv_tile<128, 128, 2, 2, CONV_BK_BIG>(c, dim3((c.M + 127) / 128, (c.Npad + 127) / 128), s);
    } else if (c.Npad >= 128 && (long long)((c.M + 63) / 64) * ((c.Npad + 127) / 128) >= wide_min) {
        // 64 x 128: half the im2col gathers per output of the 64 x 64 tile, still > 4 workgroups per CU
        launch_conv_tile<64, 128, 2, 2, 16>(c, dim3((c.M + 63) / 64, (c.Npad + 127) / 128), s);
    } else {
        const long long wgs = (long long)((c.M + 63) / 64) * ((c.Npad + 63) / 64);
        const int nk = (c.K + kSplitBK - 1) / kSplitBK;
        // small feature maps leave most CUs idle: split the contraction over blockIdx.z
        if (wgs < split_below && nk >= 32 && scratch) {
            int sp = (int)((split_target + wgs - 1) / wgs);
            if (sp > 8) sp = 8;
            if (sp > nk / 8) sp = nk / 8;
            if (sp > 1 && (long long)sp * c.M * c.Npad <= scratch_floats) { c.splits = sp; c.partial = scratch; }
        }
        launch_conv_tile<64, 64, 2, 2, CONV_BK_SMALL>(c, dim3((c.M + 63) / 64, (c.Npad + 63) / 64, c.splits), s);
        if (c.splits > 1)
            hipLaunchKernelGGL(k_splitk_reduce, dim3(conv_grid_1d((long long)c.M * c.Cout)), dim3(256), 0, s, c);
    }
}
