// wgrad_mfma.h -- the fp32-MFMA weight-gradient kernel of the train step (train_ops.hip).  Included inside the
// translation unit's anonymous namespace, after conv_mfma.h (shares its buffer-descriptor helpers).
//
//   dW[r = (ky, kx, ci)][co] = sum over pixels p = (b, oy, ox) of x[b][oy * s - pad + ky][ox * s - pad + kx][ci] * dz[p][co]
// a GEMM whose contraction runs over the pixels: both operands are [pixel][channel] rows, k-major as they lie in memory.

struct WgradArgs {
    const float *x; int H, W, Cin; long long x_bs; int x_cs;     // input NHWC
    const float *dz; int Ho, Wo, Cout; long long dz_bs; int dz_cs;
    int k, stride, pad;
    int R;          // k * k * Cin
    int M;          // B * Ho * Wo
    int splits;
    float *partial; // [splits][R][Cout]
    uint32_t x_bytes, dz_bytes; // extents for the buffer descriptors (set by launch_wgrad_tiles)
    int prec;       // 0: float32 MFMA; 1: three bf16 MFMAs per product (operands split in registers)
};

// BM = 64 or 128 rows (r) x BN = 32, 64 or 128 columns (co) per workgroup; the 4 wavefronts are WR x WC (2 x 2, or 4 x 1 for
// the 32-column tile of the stem's 32 output channels: with 2 x 2 half the wavefronts multiplied zero columns, 872 us
// at 35 % of peak), each TM x TN = (BM / 32 WR) x (BN / 32 WC) MFMA tiles of 32 x 32; k-tiles of 16 pixels; blockIdx.z owns a
// slice of the pixels and writes a partial tile (always: the reduction kernel also converts to torch's layout).
// Same machinery as k_conv_mfma: both operands arrive by LDS-DMA through buffer descriptors (padding, rows past R,
// columns past Cout and pixels past M carry the offset kOob and land as zeros), a ring of NBUF stages with counted vmcnt
// waits and raw barriers, fragment reads as asm statements with hand-counted lgkmcnt (see conv_mfma.h for why).
// LDS image of a stage: one [16 pixels][64 channels] block per 64-wide channel group (lane-linear for the DMA: a
// wave-instruction brings 4 pixels x 64 channels = 1 KB).
// P = 1: float32 products from three bf16 MFMAs (conv_mfma.h): BOTH operands are activations here, so both are split in
// registers -- a lane's eight pixels of a k-tile (2 j + h, j = 0..7: the pairing of the float32 k-steps) make one bf16 k-step.
template <int BM, int BN, int NBUF, int WR = 2, int WC = 2, int P = 0>
__global__ __launch_bounds__(256) void k_wgrad_mfma(WgradArgs a)
{
    static_assert(WR * WC == 4 && BM % (32 * WR) == 0 && BN % (32 * WC) == 0, "four wavefronts");
    constexpr int BK = 16, TM = BM / (32 * WR), TN = BN / (32 * WC), GA = (BM + 63) / 64, GB = (BN + 63) / 64;
    static_assert(TM <= 2 && TN <= 2 && (TM == 1 || (32 * TM) % 64 == 0) && (TN == 1 || (32 * TN) % 64 == 0), "a wavefront's tiles lie in one group");
    constexpr int kStage = BK * 64 * (GA + GB);       // floats per ring stage: A groups first, then B groups
    constexpr int EPLD = 36;
    constexpr int kEpiFloats = 4 * 32 * EPLD;
    __shared__ __attribute__((aligned(16))) float smem[NBUF * kStage > kEpiFloats ? NBUF * kStage : kEpiFloats];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = wv / WC, wc = wv % WC;
    const int r0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const __amdgpu_buffer_rsrc_t rx = conv_rsrc(a.x, a.x_bytes), rz = conv_rsrc(a.dz, a.dz_bytes);
    // staging: thread -> pixel (tid / 16) of the k-tile, float4 #(tid % 16) of each 64-wide group of r / co columns
    const int pk = tid >> 4, q4 = (tid & 15) * 4;
    int ky[GA], kx[GA];
    uint32_t cib[GA], nb[GB]; // byte offset of the float4's channel inside a pixel row; kOob: outside R / Cout
#pragma unroll
    for (int i = 0; i < GA; ++i) { // the (tap, ci) of this thread's float4 in r-group i (Cin % 4 == 0: one tap per float4)
        const int r = r0 + 64 * i + q4;
        const bool ok = r < a.R;
        const int tap = ok ? r / a.Cin : 0;
        cib[i] = ok ? (uint32_t)(r - tap * a.Cin) * 4 : kOob;
        ky[i] = tap / a.k;
        kx[i] = tap - ky[i] * a.k;
    }
#pragma unroll
    for (int j = 0; j < GB; ++j) nb[j] = n0 + 64 * j + q4 < a.Cout ? (uint32_t)(n0 + 64 * j + q4) * 4 : kOob;
    const int howo = a.Ho * a.Wo;
    const int nk_all = (a.M + BK - 1) / BK;
    const int kt0 = (int)((long long)nk_all * blockIdx.z / a.splits), kt1 = (int)((long long)nk_all * (blockIdx.z + 1) / a.splits);
    const int nk = kt1 - kt0;
    // this thread's pixel of the NEXT k-tile to load as (image, oy, ox); advanced by BK pixels per tile without divisions
    int pb, poy, pox;
    {
        const long long m = (long long)kt0 * BK + pk;
        pb = (int)(m / howo);
        const int pix = (int)(m - (long long)pb * howo);
        poy = pix / a.Wo;
        pox = pix - poy * a.Wo;
    }
    const int nimg = a.M / howo;
    const int x_cs4 = a.x_cs * 4, dz_cs4 = a.dz_cs * 4;
    auto load_tiles = [&](int stage) {
        float *st = smem + stage * kStage;
        const bool pix_ok = pb < nimg;
        const int iy0 = poy * a.stride - a.pad, ix0 = pox * a.stride - a.pad;
        const uint32_t xb = (uint32_t)((long long)pb * a.x_bs * 4);
#pragma unroll
        for (int i = 0; i < GA; ++i) {
            const int iy = iy0 + ky[i], ix = ix0 + kx[i];
            const bool ok = pix_ok && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const uint32_t off = ok ? xb + (uint32_t)((iy * a.W + ix) * x_cs4) + cib[i] : kOob; // + kOob stays out of range
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void *)(st + i * (BK * 64) + wv * 256), 16, (int)off, 0, 0, 0);
        }
        const uint32_t zb = pix_ok ? (uint32_t)((long long)pb * a.dz_bs * 4) + (uint32_t)((poy * a.Wo + pox) * dz_cs4) : kOob;
#pragma unroll
        for (int j = 0; j < GB; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rz, (__attribute__((address_space(3))) void *)(st + (GA + j) * (BK * 64) + wv * 256), 16,
                                                     (int)(zb == kOob ? kOob : zb + nb[j]), 0, 0, 0);
        pox += BK;
        while (pox >= a.Wo) { pox -= a.Wo; if (++poy == a.Ho) { poy = 0; ++pb; } }
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    // every wavefront issues GA + GB DMA instructions per tile; NBUF - 2 tiles stay in flight across a barrier
    auto wait_next_tile = [&](int in_flight) { // wave-uniform
        if (in_flight >= 1 && NBUF >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GA + GB) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    if (nk > 0) load_tiles(0);
    if (NBUF >= 3 && nk > 1) load_tiles(1);
    wait_next_tile(nk > 1 ? 1 : 0);
    __builtin_amdgcn_s_barrier();
    // fragment addresses (bytes) inside a stage
    const int fh = lane >> 5, fl = lane & 31;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)smem;
    const int ac = wr * 32 * TM, bc = wc * 32 * TN; // first column of this wavefront's tiles: group c / 64, column c % 64 of it
    const uint32_t a_lds = lds0 + (uint32_t)((ac / 64) * (BK * 64) + fh * 64 + (ac % 64) + fl) * 4;
    const uint32_t b_lds = lds0 + (uint32_t)((GA + bc / 64) * (BK * 64) + fh * 64 + (bc % 64) + fl) * 4;
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + NBUF - 1 < nk) load_tiles(stage == 0 ? NBUF - 1 : stage - 1); // the stage read last in iteration kt - 1
        if constexpr (P == 1) {
            const uint32_t a_addr = a_lds + (uint32_t)stage * (kStage * 4), b_addr = b_lds + (uint32_t)stage * (kStage * 4);
            float fa[TM][8], fb[TN][8];
            auto read8 = [&](auto stc) { // element st of the lane's fragments: pixel 2 st + h
                constexpr int st = decltype(stc)::value;
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fa[0][st]) : "v"(a_addr), "n"(st * 512));
                if (TM > 1) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fa[TM - 1][st]) : "v"(a_addr), "n"(st * 512 + 128));
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fb[0][st]) : "v"(b_addr), "n"(st * 512));
                if (TN > 1) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fb[TN - 1][st]) : "v"(b_addr), "n"(st * 512 + 128));
            };
            read8(ConvIC<0>{}); read8(ConvIC<1>{}); read8(ConvIC<2>{}); read8(ConvIC<3>{});
            read8(ConvIC<4>{}); read8(ConvIC<5>{}); read8(ConvIC<6>{}); read8(ConvIC<7>{});
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            u32x4 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(fa[i][e]));
                conv_split8(f32x4{fa[i][0], fa[i][1], fa[i][2], fa[i][3]}, f32x4{fa[i][4], fa[i][5], fa[i][6], fa[i][7]}, ah[i], al[i]);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(fb[j][e]));
                conv_split8(f32x4{fb[j][0], fb[j][1], fb[j][2], fb[j][3]}, f32x4{fb[j][4], fb[j][5], fb[j][6], fb[j][7]}, bh[j], bl[j]);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) { // the small terms first
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[i]), __builtin_bit_cast(bf16x8, bh[j]), acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bl[j]), acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bh[j]), acc[i][j], 0, 0, 0);
                }
        } else {
            float fa[2][TM], fb[2][TN];
            const uint32_t a_addr = a_lds + (uint32_t)stage * (kStage * 4), b_addr = b_lds + (uint32_t)stage * (kStage * 4);
            auto read_frags = [&](auto stc) { // k-step st: pixels 2 st + h
                constexpr int st = decltype(stc)::value;
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fa[st & 1][0]) : "v"(a_addr), "n"(st * 512));
                if (TM > 1) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fa[st & 1][TM - 1]) : "v"(a_addr), "n"(st * 512 + 128));
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fb[st & 1][0]) : "v"(b_addr), "n"(st * 512));
                if (TN > 1) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fb[st & 1][TN - 1]) : "v"(b_addr), "n"(st * 512 + 128));
            };
            auto step = [&](auto stc) {
                constexpr int st = decltype(stc)::value;
                if (st + 1 < 8) {
                    read_frags(ConvIC<(st + 1 < 8 ? st + 1 : 7)>{});
                    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(TM + TN) : "memory");
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(fa[st & 1][i]));
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(fb[st & 1][j]));
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[st & 1][i], fb[st & 1][j], acc[i][j], 0, 0, 0);
            };
            read_frags(ConvIC<0>{});
            step(ConvIC<0>{}); step(ConvIC<1>{}); step(ConvIC<2>{}); step(ConvIC<3>{});
            step(ConvIC<4>{}); step(ConvIC<5>{}); step(ConvIC<6>{}); step(ConvIC<7>{});
        }
        wait_next_tile(kt + NBUF - 1 < nk ? 1 : 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        stage = stage == NBUF - 1 ? 0 : stage + 1;
    }
    // partial tile as 16-byte rows through a wavefront-private LDS transpose (C/D layout of the 32x32 MFMA: col = lane & 31,
    // row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5))
    float *dst = a.partial + (long long)blockIdx.z * a.R * a.Cout;
    float *epw = smem + wv * 32 * EPLD;
    const int er = lane >> 3, ec = (lane & 7) * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int e = 0; e < 16; ++e) epw[((e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * EPLD + (lane & 31)] = acc[i][j][e];
            asm volatile("" ::: "memory");
            const int n = n0 + wc * 32 * TN + 32 * j + ec;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int rr = r0 + wr * 32 * TM + 32 * i + er + 8 * t;
                if (rr < a.R && n < a.Cout) *(float4 *)(dst + (long long)rr * a.Cout + n) = *(const float4 *)&epw[(er + 8 * t) * EPLD + ec];
            }
        }
}

// 128 x 128 tiles (half the x gathers and half the dz reads per output) when that still leaves >= 32 output tiles
inline bool wgrad_wide(long long R, int Cout)
{
    static const long long min_tiles = dev_knob("FRLW_WGRAD_WIDE_TILES", 32ll);
    return Cout >= 128 && R >= 128 && ((R + 127) / 128) * ((Cout + 127) / 128) >= min_tiles;
}

#ifndef WGRAD_NBUF
#define WGRAD_NBUF 2
#endif
inline bool launch_wgrad_tiles(WgradArgs &a, hipStream_t s) // a.splits and a.partial are set by the caller; false: too large
{
    const int nimg = a.M / (a.Ho * a.Wo);
    const long long xb = ((long long)(nimg - 1) * a.x_bs + ((long long)a.H * a.W - 1) * a.x_cs + a.Cin) * 4;
    const long long zb = ((long long)(nimg - 1) * a.dz_bs + ((long long)a.Ho * a.Wo - 1) * a.dz_cs + a.Cout) * 4;
    if (xb > kMaxViewBytes || zb > kMaxViewBytes) return false; // 32-bit buffer offsets
    a.x_bytes = (uint32_t)xb;
    a.dz_bytes = (uint32_t)zb;
    if (a.prec == 1) {
        if (wgrad_wide(a.R, a.Cout))
            hipLaunchKernelGGL((k_wgrad_mfma<128, 128, WGRAD_NBUF, 2, 2, 1>), dim3((a.R + 127) / 128, (a.Cout + 127) / 128, a.splits), dim3(256), 0, s, a);
        else if (a.R > 64 && a.Cout <= 32)
            hipLaunchKernelGGL((k_wgrad_mfma<128, 32, WGRAD_NBUF, 4, 1, 1>), dim3((a.R + 127) / 128, 1, a.splits), dim3(256), 0, s, a);
        else if (a.R > 64)
            hipLaunchKernelGGL((k_wgrad_mfma<128, 64, WGRAD_NBUF, 2, 2, 1>), dim3((a.R + 127) / 128, (a.Cout + 63) / 64, a.splits), dim3(256), 0, s, a);
        else
            hipLaunchKernelGGL((k_wgrad_mfma<64, 64, WGRAD_NBUF, 2, 2, 1>), dim3((a.R + 63) / 64, (a.Cout + 63) / 64, a.splits), dim3(256), 0, s, a);
        return true;
    }
    if (wgrad_wide(a.R, a.Cout))
        hipLaunchKernelGGL((k_wgrad_mfma<128, 128, WGRAD_NBUF>), dim3((a.R + 127) / 128, (a.Cout + 127) / 128, a.splits), dim3(256), 0, s, a);
    else if (a.R > 64 && a.Cout <= 32)
        hipLaunchKernelGGL((k_wgrad_mfma<128, 32, WGRAD_NBUF, 4, 1>), dim3((a.R + 127) / 128, 1, a.splits), dim3(256), 0, s, a);
    else if (a.R > 64)
        hipLaunchKernelGGL((k_wgrad_mfma<128, 64, WGRAD_NBUF>), dim3((a.R + 127) / 128, (a.Cout + 63) / 64, a.splits), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((k_wgrad_mfma<64, 64, WGRAD_NBUF>), dim3((a.R + 63) / 64, (a.Cout + 63) / 64, a.splits), dim3(256), 0, s, a);
    return true;
}

inline long long wgrad_want_splits(long long R, int Cout, long long M, long long target)
{
    const long long bm = R > 64 ? 128 : 64, bn = wgrad_wide(R, Cout) ? 128 : (R > 64 && Cout <= 32 ? 32 : 64);
    const long long tiles = ((R + bm - 1) / bm) * ((Cout + bn - 1) / bn);
    const long long nk = (M + 15) / 16;
    // floor, not ceil: `target` workgroups are what the part holds at once (five 128 x 128 workgroups per CU); one split more and
    // 36 tiles x 36 splits = 1296 workgroups leave a sixth workgroup on 16 CUs -- a second, nearly empty round of the whole
    // kernel (the 256 -> 256 3 x 3 layers: 27.8 -> 27.4 ms per train step)
    long long sp = target / tiles;
    static const long long max_sp = dev_knob("FRLW_WGRAD_MAXSP", 256ll);
    if (sp > max_sp) sp = max_sp;              // (more than 64 partial tiles per output tile: two-stage reduction)
    if (sp > nk / 4) sp = nk / 4;
    if (sp < 1) sp = 1;
    return sp;
}

#ifdef FRLW_DEV_BUILD
inline void launch_wgrad(WgradArgs &a, int target, long long scratch_floats, hipStream_t s) // tools/wgrad_lab.hip
{
    long long sp = wgrad_want_splits(a.R, a.Cout, a.M, target);
    while (sp > 1 && sp * a.R * a.Cout > scratch_floats) --sp;
    a.splits = (int)sp;
    launch_wgrad_tiles(a, s);
}
#endif
