// enc_lab -- stand-alone timing / differential harness of the batched encoders of libfrlw_evd.so (no Python, no torch:
// a gpurun call of a few seconds).  Loads one or two builds of the library through the C-ABI, runs the same synthetic DAT
// streams through each, prints device time per encode (HIP events) and an FNV hash of every output; with two libraries it
// also compares the outputs byte for byte (the first one is the reference build, e.g. the last parity-green commit).
//
//   hipcc -O2 -std=c++17 -I include tools/enc_lab.cpp -o build/enc_lab -ldl
//   build/enc_lab frlw-evd_amd/csrc/libfrlw_evd.so [build/libfrlw_base.so] [--cfg mpx,mpx_hot,gen1,gen1x64,e2e64,ev1,evb64] [--reps 20] [--tile-walk]
//   rocprofv3 --kernel-trace --stats -- build/enc_lab <lib> --cfg mpx          (per-kernel breakdown of exactly that encode)
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "frlw_evd.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } } while (0)

struct Lib {
    void *h = nullptr;
    std::string path;
    decltype(&frlw_taf_batch_workspace_bytes) taf_ws = nullptr;
    decltype(&frlw_taf_encode_batch) taf = nullptr;
    decltype(&frlw_encoder_status) status = nullptr;
    decltype(&frlw_encoder_workspace_bytes) enc_ws = nullptr;
    decltype(&frlw_ev_encode) ev = nullptr;
    size_t (*ev_batch_ws)(int64_t, int, int, int, int64_t) = nullptr;
    int (*ev_batch)(const frlw_events_t *, const int64_t *, const int64_t *, int, int, int, int, int64_t, float *, uint8_t *, void *, size_t,
                    frlw_stream_t) = nullptr;
    int (*ws_init)(void *, size_t, frlw_stream_t) = nullptr;
};

static Lib load(const char *path)
{
    Lib L;
    L.path = path;
    L.h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!L.h) { fprintf(stderr, "dlopen %s: %s\n", path, dlerror()); exit(2); }
    L.taf_ws = (decltype(L.taf_ws))dlsym(L.h, "frlw_taf_batch_workspace_bytes");
    L.taf = (decltype(L.taf))dlsym(L.h, "frlw_taf_encode_batch");
    L.status = (decltype(L.status))dlsym(L.h, "frlw_encoder_status");
    L.enc_ws = (decltype(L.enc_ws))dlsym(L.h, "frlw_encoder_workspace_bytes");
    L.ev = (decltype(L.ev))dlsym(L.h, "frlw_ev_encode");
    L.ev_batch_ws = (decltype(L.ev_batch_ws))dlsym(L.h, "frlw_ev_batch_workspace_bytes");
    L.ev_batch = (decltype(L.ev_batch))dlsym(L.h, "frlw_ev_encode_batch");
    L.ws_init = (decltype(L.ws_init))dlsym(L.h, "frlw_workspace_init");
    return L;
}

// ---- synthetic streams (own generator: time-sorted, uniform pixels, optional sigma = 8 px blob with 25 % of the events)
struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0x1234567ull) {}
    uint64_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s * 0x2545F4914F6CDD1Dull; }
    uint32_t below(uint32_t n) { return (uint32_t)((next() >> 32) * (uint64_t)n >> 32); }
    double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};

static void gen_stream(std::vector<uint64_t> &out, uint64_t seed, size_t n, int W, int H, uint32_t t_span, bool hotspot, uint32_t t_off = 0)
{
    Rng r(seed);
    const size_t base = out.size();
    out.resize(base + n);
    for (size_t i = 0; i < n; ++i) {
        // sorted times: stratified uniform (event i falls uniformly inside its own 1/n slice of the span)
        const double u = ((double)i + r.uni()) / (double)n;
        uint32_t t = (uint32_t)(u * t_span);
        if (t >= t_span) t = t_span - 1;
        t += t_off;
        uint32_t x = r.below((uint32_t)W), y = r.below((uint32_t)H);
        const uint32_t p = r.below(2);
        if (hotspot && r.below(4) == 0) {
            const double a = sqrt(-2.0 * log(r.uni() + 1e-300)), b = 6.283185307179586 * r.uni();
            long hx = lrint(W / 2.0 + 8.0 * a * cos(b)), hy = lrint(H / 2.0 + 8.0 * a * sin(b));
            x = (uint32_t)(hx < 0 ? 0 : (hx >= W ? W - 1 : hx));
            y = (uint32_t)(hy < 0 ? 0 : (hy >= H ? H - 1 : hy));
        }
        const uint32_t w = (x & 16383u) | ((y & 16383u) << 14) | (p << 28);
        out[base + i] = (uint64_t)t | ((uint64_t)w << 32);
    }
}

static uint64_t fnv(const void *p, size_t n)
{
    const uint64_t *q = (const uint64_t *)p;
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n / 8; ++i) { h ^= q[i]; h *= 1099511628211ull; }
    const uint8_t *t = (const uint8_t *)p + (n / 8) * 8;
    for (size_t i = 0; i < n % 8; ++i) { h ^= t[i]; h *= 1099511628211ull; }
    return h;
}

struct Cfg {
    const char *name;
    int kind; // 0 = TAF batch, 1 = EV single (frlw_ev_encode), 2 = EV batch
    int H, W, n_seq;
    size_t n_per_seq;
    uint32_t t_span;
    int n_windows, window_us, K;
    bool hotspot;
};

static const Cfg kCfgs[] = {
    {"mpx", 0, 720, 1280, 1, 10000000, 80000, 8, 10000, 8, false},
    {"mpx_hot", 0, 720, 1280, 1, 10000000, 80000, 8, 10000, 8, true},
    {"gen1", 0, 240, 304, 1, 1000000, 80000, 8, 10000, 8, false},
    {"gen1x64", 0, 240, 304, 64, 1000000, 80000, 8, 10000, 8, false},
    {"e2e64", 0, 240, 304, 64, 1000000, 80000, 8, 10000, 8, true},
    {"small", 0, 97, 131, 5, 70000, 30000, 3, 10000, 4, true},
    {"e2e64s", 0, 240, 304, 64, 125000, 80000, 8, 10000, 8, false},    // the batch of bench.py's encode + train row: 64 x 8 x 15 625 events
    {"gen1x8", 0, 240, 304, 8, 1000000, 80000, 8, 10000, 8, false},
    {"mpx3", 0, 720, 1280, 1, 3000000, 80000, 8, 10000, 8, false},
    {"ev1", 1, 240, 304, 1, 1000000, 250000, 1, 250000, 5, false},
    {"evb1", 2, 240, 304, 1, 1000000, 250000, 1, 250000, 5, false},
    {"evb64", 2, 240, 304, 64, 1000000, 250000, 1, 250000, 5, false},
    {"evb64_hot", 2, 240, 304, 64, 1000000, 250000, 1, 250000, 5, true},
    {"evb_small", 2, 97, 131, 5, 70000, 30000, 1, 30000, 3, true},
};

struct Outputs {
    std::vector<uint8_t> a, b; // TAF: state, u8; EV: f32, (unused)
    double us = 0;
    int status = 0;
    bool ran = false;
};

static frlw_tuning_t g_tuning = {(int32_t)sizeof(frlw_tuning_t), -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
static bool g_use_tuning = false;

static Outputs run_cfg(const Lib &L, const Cfg &c, const uint64_t *dat_d, const std::vector<int64_t> &offs, int reps)
{
    Outputs o;
    const int64_t n = offs.back();
    hipStream_t st = nullptr;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    frlw_events_t ev;
    memset(&ev, 0, sizeof(ev));
    ev.data = dat_d; ev.n = n; ev.layout = FRLW_LAYOUT_DAT8;
    if (g_use_tuning) ev.tuning = &g_tuning;
    const size_t plane = (size_t)c.H * c.W;
    if (c.kind == 0) {
        if (!L.taf || !L.taf_ws) return o;
        const size_t wsb = L.taf_ws(n, c.n_seq, c.H, c.W, c.window_us);
        if (!wsb) { fprintf(stderr, "%s: unsupported shape\n", c.name); return o; }
        const size_t sb = (size_t)c.n_seq * plane * 2 * c.K * 4, ub = (size_t)c.n_seq * plane * 2 * c.K;
        void *ws; float *state; uint8_t *u8;
        CK(hipMalloc(&ws, wsb)); CK(hipMalloc(&state, sb)); CK(hipMalloc(&u8, ub));
        CK(hipMemset(ws, 0, 1024));
        std::vector<float> init(sb / 4, -6000.0f);
        std::vector<int64_t> t0(c.n_seq, 0);
        auto call = [&]() { return L.taf(&ev, offs.data(), t0.data(), c.n_seq, c.H, c.W, c.K, c.window_us, c.n_windows, state, nullptr, u8,
                                         FRLW_TAF_U8_FLIP_K, ws, wsb, st); };
        // correctness pass: two consecutive encodes from the initial state (the second one carries the FIFO state)
        CK(hipMemcpy(state, init.data(), sb, hipMemcpyHostToDevice));
        CK(hipMemset(u8, 0xAB, ub));
        int rc = call();
        if (rc == FRLW_OK) rc = call();
        int stt = 0;
        if (rc == FRLW_OK) L.status(ws, st, &stt);
        CK(hipDeviceSynchronize());
        o.status = rc != FRLW_OK ? rc : stt;
        o.a.resize(sb); o.b.resize(ub);
        CK(hipMemcpy(o.a.data(), state, sb, hipMemcpyDeviceToHost));
        CK(hipMemcpy(o.b.data(), u8, ub, hipMemcpyDeviceToHost));
        for (int i = 0; i < 3; ++i) call();
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) call();
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        o.us = ms * 1000.0 / reps;
        o.ran = true;
        CK(hipFree(ws)); CK(hipFree(state)); CK(hipFree(u8));
    } else {
        const int bins = c.K;
        const size_t ob = (size_t)c.n_seq * plane * 2 * bins * 4;
        float *out;
        CK(hipMalloc(&out, ob));
        void *ws = nullptr;
        size_t wsb = 0;
        std::vector<int64_t> tend(c.n_seq, (int64_t)c.window_us);
        int rc = FRLW_OK;
        auto call = [&]() -> int {
            if (c.kind == 1) return L.ev(&ev, c.H, c.W, bins, c.window_us, c.window_us, out, nullptr, ws, wsb, st);
            return L.ev_batch(&ev, offs.data(), tend.data(), c.n_seq, c.H, c.W, bins, c.window_us, out, nullptr, ws, wsb, st);
        };
        if (c.kind == 1) {
            if (!L.ev) { CK(hipFree(out)); return o; }
            wsb = L.enc_ws(n, c.H, c.W);
        } else {
            if (!L.ev_batch || !L.ev_batch_ws) { CK(hipFree(out)); return o; }
            wsb = L.ev_batch_ws(n, c.n_seq, c.H, c.W, c.window_us);
        }
        if (!wsb) { fprintf(stderr, "%s: unsupported shape\n", c.name); CK(hipFree(out)); return o; }
        CK(hipMalloc(&ws, wsb));
        CK(hipMemset(ws, 0, 1024));
        CK(hipMemset(out, 0xAB, ob));
        rc = call();
        int stt = 0;
        if (rc == FRLW_OK) L.status(ws, st, &stt);
        CK(hipDeviceSynchronize());
        o.status = rc != FRLW_OK ? rc : stt;
        o.a.resize(ob);
        CK(hipMemcpy(o.a.data(), out, ob, hipMemcpyDeviceToHost));
        for (int i = 0; i < 3; ++i) call();
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) call();
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        o.us = ms * 1000.0 / reps;
        o.ran = true;
        CK(hipFree(ws)); CK(hipFree(out));
    }
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return o;
}

int main(int argc, char **argv)
{
    std::vector<Lib> libs;
    std::string only = "mpx,mpx_hot,gen1,gen1x64";
    int reps = 20;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--cfg") && i + 1 < argc) only = argv[++i];
        else if (!strcmp(argv[i], "--reps") && i + 1 < argc) reps = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--tile-walk")) { g_tuning.taf_tile_walk = 1; g_use_tuning = true; } // TAF through kf_taf_tile
        else if (!strcmp(argv[i], "--bpw") && i + 1 < argc) { g_tuning.batches_per_wave = atoi(argv[++i]); g_use_tuning = true; } // larger partition chunks
        else if (!strcmp(argv[i], "--fadd")) { g_tuning.ev_lds_float_atomics = 1; g_use_tuning = true; } // Event Volume, direct mode: LDS float atomics
        else if (!strcmp(argv[i], "--no-fadd")) { g_tuning.ev_lds_float_atomics = 0; g_use_tuning = true; } // ... the ticket-sort kernel
        else if (!strcmp(argv[i], "--cm")) { g_tuning.chunk_major = 1; g_use_tuning = true; } // the chunk-major partition wherever it is possible
        else if (!strcmp(argv[i], "--no-cm")) { g_tuning.chunk_major = 0; g_use_tuning = true; } // histogram + scans + bin-major scatter
        else if (!strcmp(argv[i], "--direct")) { g_tuning.direct_bins = 1; g_use_tuning = true; } // sub-tile bins wherever the frame allows
        else if (!strcmp(argv[i], "--no-direct")) { g_tuning.direct_bins = 0; g_use_tuning = true; } // tile bins + split pass also on small frames
        else libs.push_back(load(argv[i]));
    }
    if (libs.empty()) { fprintf(stderr, "usage: enc_lab <lib.so> [<reference lib.so>] [--cfg a,b] [--reps N]\n"); return 2; }
    int bad = 0;
    for (const Cfg &c : kCfgs) {
        const std::string key = std::string(",") + only + ",";
        if (key.find(std::string(",") + c.name + ",") == std::string::npos) continue;
        std::vector<uint64_t> dat;
        std::vector<int64_t> offs(1, 0);
        for (int s = 0; s < c.n_seq; ++s) {
            // sequences differ in length a little (ragged batches), one of a batch is sparse
            size_t n_s = c.n_per_seq;
            if (c.n_seq > 1) n_s = s == 3 ? c.n_per_seq / 50 : c.n_per_seq - (size_t)(s * 997 % 5000);
            gen_stream(dat, 1000 + 17 * s + (c.hotspot ? 5 : 0), n_s, c.W, c.H, c.t_span, c.hotspot);
            offs.push_back((int64_t)dat.size());
        }
        uint64_t *dat_d;
        CK(hipMalloc(&dat_d, dat.size() * 8));
        CK(hipMemcpy(dat_d, dat.data(), dat.size() * 8, hipMemcpyHostToDevice));
        std::vector<Outputs> res;
        for (const Lib &L : libs) {
            Outputs o = run_cfg(L, c, dat_d, offs, reps);
            if (!o.ran) { printf("%-10s %-40s not available in this build\n", c.name, L.path.c_str()); res.push_back(o); continue; }
            const double alg = c.kind == 0 ? 8.0 * dat.size() + (double)c.n_seq * c.H * c.W * (2.0 * 4 * 2 * c.K + 2 * c.K)
                                           : 8.0 * dat.size() + (double)c.n_seq * c.H * c.W * 4.0 * 2 * c.K;
            printf("%-10s %-40s %9.1f us  %7.2f Gev/s  %7.1f GB/s alg  status %d  hash %016llx %016llx\n", c.name, L.path.c_str(), o.us,
                   dat.size() / o.us / 1e3, alg / o.us / 1e3, o.status, (unsigned long long)fnv(o.a.data(), o.a.size()),
                   (unsigned long long)fnv(o.b.data(), o.b.size()));
            res.push_back(std::move(o));
            if (void *fp = dlsym(L.h, "frlw_debug_walk_prof")) { // -DFRLW_WALK_PROF builds: kf_taf_walk's timeline, cycles per workgroup
                unsigned long long pr[16];
                if (((int (*)(unsigned long long *))fp)(pr) == 0) {
                    printf("%-10s walk timeline (sum over all encodes incl. warm-up, s_memtime ticks):", c.name);
                    for (int i = 0; i < 9; ++i) printf(" %llu", pr[i]);
                    printf("  workgroups %llu", pr[15]);
                    printf("\n");
                }
            }
        }
        if (c.kind == 2 && res[0].ran && libs[0].ev) {
            // reference of the batched Event Volume: the general path (frlw_ev_encode), one call per sequence
            const Lib &L = libs[0];
            const size_t per = (size_t)c.H * c.W * 2 * c.K * 4;
            std::vector<uint8_t> ref(per * c.n_seq);
            float *out;
            CK(hipMalloc(&out, per));
            bool ok = true;
            for (int s = 0; s < c.n_seq; ++s) {
                const int64_t n_s = offs[s + 1] - offs[s];
                frlw_events_t ev;
                memset(&ev, 0, sizeof(ev));
                ev.data = dat_d + offs[s]; ev.n = n_s; ev.layout = FRLW_LAYOUT_DAT8;
                const size_t wsb = L.enc_ws(n_s, c.H, c.W);
                void *ws;
                CK(hipMalloc(&ws, wsb));
                CK(hipMemset(ws, 0, 1024));
                const int rc = L.ev(&ev, c.H, c.W, c.K, c.window_us, c.window_us, out, nullptr, ws, wsb, nullptr);
                CK(hipDeviceSynchronize());
                if (rc != FRLW_OK) ok = false;
                CK(hipMemcpy(ref.data() + per * s, out, per, hipMemcpyDeviceToHost));
                CK(hipFree(ws));
            }
            CK(hipFree(out));
            size_t cnt = 0, first = 0;
            for (size_t i = 0; i < ref.size(); ++i)
                if (ref[i] != res[0].a[i]) { if (!cnt) first = i; ++cnt; }
            printf("%-10s vs frlw_ev_encode per sequence: %s", c.name, ok && cnt == 0 ? "IDENTICAL\n" : "DIFFER");
            if (!(ok && cnt == 0)) { printf(" (%zu bytes, first at %zu = sequence %zu, float %zu)\n", cnt, first, first / per, (first % per) / 4); ++bad; }
        }
        if (res.size() >= 2 && res[0].ran && res[1].ran) {
            const bool same = res[0].a == res[1].a && res[0].b == res[1].b && res[0].status == res[1].status;
            printf("%-10s %s\n", c.name, same ? "outputs IDENTICAL" : "outputs DIFFER");
            if (!same) {
                ++bad;
                size_t first = 0, cnt = 0;
                for (size_t i = 0; i < res[0].a.size() && i < res[1].a.size(); ++i)
                    if (res[0].a[i] != res[1].a[i]) { if (!cnt) first = i; ++cnt; }
                printf("           first buffer: %zu differing bytes, first at %zu\n", cnt, first);
            }
        }
        CK(hipFree(dat_d));
        fflush(stdout);
    }
    return bad ? 1 : 0;
}
