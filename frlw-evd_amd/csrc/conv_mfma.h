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

// BM x BN output tile, 4 wavefronts arranged WROWS x WCOLS, each owning TM x TN MFMA tiles of 32 x 32.
template <int BM, int BN, int WROWS, int WCOLS, int BK>
__global__ __launch_bounds__(256) void k_conv_mfma(ConvArgs a)
{
    constexpr int TM = BM / (32 * WROWS), TN = BN / (32 * WCOLS);
    constexpr int KQ = BK / 4;        // float4 per A row of the k-tile
    constexpr int RPP = 256 / KQ;     // A rows staged per pass of the 256 threads
    constexpr int LDA = BM + 4, LDB = BN + 4; // +4 floats: k rows land on different banks for the staging writes
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

    // ---- A staging: thread -> A_F4 rows m, one float4 of 4 consecutive k
    const int a_k4 = (tid % KQ) * 4;
    int a_iy0[A_F4], a_ix0[A_F4];
    long long a_base[A_F4];
    bool a_ok[A_F4];
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
        const int m = m0 + tid / KQ + RPP * i;
        a_ok[i] = m < a.M;
        const int mm = a_ok[i] ? m : 0;
        const int b = mm / (a.Ho * a.Wo), pix = mm - b * (a.Ho * a.Wo);
        const int oy = pix / a.Wo, ox = pix - oy * a.Wo;
        a_iy0[i] = oy * a.stride - a.pad;
        a_ix0[i] = ox * a.stride - a.pad;
        a_base[i] = (long long)b * a.x_bs + a.x_co;
    }
    // ---- B staging: thread -> rows k, one float4 of 4 consecutive n
    constexpr int BN4 = BN / 4;
    float4 ra[A_F4], rb[B_F4];

    // (ky, kx, ci) of this thread's float4 in the CURRENT k-tile to be loaded; advanced by BK per tile
    int t_ci = 0, t_ky = 0, t_kx = 0;
    const int kw = a.kw ? a.kw : a.k;
    auto load_tiles = [&](int kt) {
        const int k = kt * BK + a_k4;
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            int iy = a_iy0[i] + t_ky, ix = a_ix0[i] + t_kx;
            bool even = true;
            if (a.tstride == 2) { even = !((iy | ix) & 1); iy >>= 1; ix >>= 1; }
            const bool ok = even && a_ok[i] && k < a.K && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            ra[i] = ok ? *(const float4 *)(a.x + a_base[i] + ((long long)iy * a.W + ix) * a.x_cs + t_ci)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        t_ci += BK; // Cin % 4 == 0: a float4 never straddles taps
        while (t_ci >= a.Cin) { t_ci -= a.Cin; if (++t_kx == kw) { t_kx = 0; ++t_ky; } }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int e = tid + 256 * i;
            const int kr = e / BN4, n4 = (e - kr * BN4) * 4;
            const int kk = kt * BK + kr;
            const bool ok = kr < BK && kk < a.K && n0 + n4 < a.Npad;
            rb[i] = ok ? *(const float4 *)(a.w + (long long)kk * a.Npad + n0 + n4) : make_float4(0.f, 0.f, 0.f, 0.f);
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
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int e = tid + 256 * i;
            const int kr = e / BN4, n4 = (e - kr * BN4) * 4;
            if (kr < BK) *(float4 *)&Bs[buf][kr][n4] = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nk_all = (a.K + BK - 1) / BK;
    const int kt0 = (int)((long long)nk_all * blockIdx.z / a.splits), kt1 = (int)((long long)nk_all * (blockIdx.z + 1) / a.splits);
    const int nk = kt1 - kt0;
    {   // position the tap tracker on this split's first k-tile
        const int k = kt0 * BK + a_k4;
        const int tap = k / a.Cin;
        t_ci = k - tap * a.Cin;
        t_ky = tap / kw;
        t_kx = tap - t_ky * kw;
    }
    load_tiles(kt0);
    store_tiles(0);
    __syncthreads();
    const int fm = wr * TM * 32 + (lane & 31), fn = wc * TN * 32 + (lane & 31), fk = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
#if !defined(CONV_EXP) || CONV_EXP < 1
        if (kt + 1 < nk) load_tiles(kt0 + kt + 1);
#endif
        // all operand fragments of the k-tile are requested up front, into their own registers: the LDS answers in
        // order, so the first MFMAs start as soon as their fragments are there while the rest is still in flight
        // (fragment reads issued one k-step at a time leave every wavefront waiting a full LDS round trip per step:
        // measured 117 -> see DESIGN.md section 4)
        float fa[BK / 2][TM], fb[BK / 2][TN];
#if defined(CONV_EXP) && CONV_EXP == 4
        if (kt == 0)
#endif
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[kk / 2][i] = As[buf][kk + fk][fm + 32 * i];
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[kk / 2][j] = Bs[buf][kk + fk][fn + 32 * j];
        }
#if defined(CONV_EXP) && CONV_EXP >= 5
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk / 2][i], fb[kk / 2][j], acc[i][j], 0, 0, 0);
        }
#if !defined(CONV_EXP) || CONV_EXP < 2
        if (kt + 1 < nk) store_tiles(buf ^ 1);
#endif
#if !defined(CONV_EXP) || CONV_EXP < 3
        __syncthreads();
#endif
    }

#if defined(CONV_EXP) && CONV_EXP == 6
    {   // experiment: no epilogue at all (a never-taken store keeps the accumulators alive)
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += acc[i][j][r];
        if (t == 123.456f) a.y[0] = t;
        return;
    }
#endif
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

// Tile choice and split-K for one convolution; `scratch` (scratch_floats floats, may be NULL) holds split-K partials.
inline void launch_conv(ConvArgs &c, float *scratch, long long scratch_floats, hipStream_t s)
{
    c.splits = 1;
    c.partial = nullptr;
    const long long big = (long long)((c.M + 127) / 128) * ((c.Npad + 127) / 128);
    static const long long split_below = dev_knob("FRLW_CONV_SPLIT_BELOW", 700ll);
    static const long long split_target = dev_knob("FRLW_CONV_SPLIT_TARGET", 1024ll);
    static const long long big_min = dev_knob("FRLW_CONV_BIG_MIN", 1000000ll);
    static const long long wide_min = dev_knob("FRLW_CONV_WIDE_MIN", 1200ll);
    if (c.Npad <= 32) { // small N (prediction convs, the stem's data gradient)
        hipLaunchKernelGGL((k_conv_mfma<128, 32, 4, 1, 16>), dim3((c.M + 127) / 128, 1), dim3(256), 0, s, c);
    } else if (big >= big_min && c.Npad >= 128) {
        hipLaunchKernelGGL((k_conv_mfma<128, 128, 2, 2, CONV_BK_BIG>), dim3((c.M + 127) / 128, (c.Npad + 127) / 128), dim3(256), 0, s, c);
    } else if (c.Npad >= 128 && (long long)((c.M + 63) / 64) * ((c.Npad + 127) / 128) >= wide_min) {
        // 64 x 128: half the im2col gathers per output of the 64 x 64 tile, still > 4 workgroups per CU
        hipLaunchKernelGGL((k_conv_mfma<64, 128, 2, 2, 16>), dim3((c.M + 63) / 64, (c.Npad + 127) / 128), dim3(256), 0, s, c);
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
        hipLaunchKernelGGL((k_conv_mfma<64, 64, 2, 2, CONV_BK_SMALL>), dim3((c.M + 63) / 64, (c.Npad + 63) / 64, c.splits), dim3(256), 0, s, c);
        if (c.splits > 1)
            hipLaunchKernelGGL(k_splitk_reduce, dim3(conv_grid_1d((long long)c.M * c.Cout)), dim3(256), 0, s, c);
    }
}
