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
    if (act == ACT_SILU) return v / (1.0f + expf(-v));      // x * sigmoid(x)
    if (act == ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
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
    __shared__ float As[2][BK][LDA];
    __shared__ float Bs[2][BK][LDB];
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
        if (kt + 1 < nk) load_tiles(kt0 + kt + 1);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = As[buf][kk + fk][fm + 32 * i];
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = Bs[buf][kk + fk][fn + 32 * j];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tiles(buf ^ 1);
        __syncthreads();
    }

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
