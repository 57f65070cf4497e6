// wp_api.hip — the C ABI of libwpdsp.so (include/wpdsp.h): handle management,
// argument checking and kernel sequencing.  No arithmetic on samples happens on
// the host: every entry point that produces samples launches the gfx950 kernels
// and fails with WP_ERR_NO_DEVICE when there is no GPU.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <functional>
#include <initializer_list>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "wp_internal.h"

using namespace wp;

// ---------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

// for the other translation units of the library (not exported)
extern "C" __attribute__((visibility("hidden"))) int wp_fail_message(int code, const char *msg)
{
    g_err = msg ? msg : "";
    return code;
}

#define HIPCHK(expr)                                                                       \
    do {                                                                                   \
        hipError_t e__ = (expr);                                                           \
        if (e__ != hipSuccess)                                                             \
            return fail(WP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

static std::mutex g_mu;
static int g_ndev = -2;  // -2: not probed

// ---------------------------------------------------------------------------
// Workspace of wp_convolve_dev_noalloc: the block spectra Z and W of one channel-job, in a
// device array that belongs to the code object, so that the entry point never allocates.
// Every GPU that loads the library holds its own copy at its own address.  The name below
// is a host-side stand-in: host code only ever uses the per-device address that
// hipGetSymbolAddress returns (g_static_ws), for kernel arguments and memsets alike.
static __device__ __align__(256) unsigned char g_static_ws_array[WP_NOALLOC_WORKSPACE_BYTES];
static std::map<int, void *> g_static_ws;  // device -> address of its copy; guarded by g_mu

static int static_ws_locked(int dev, void **out)  // the caller holds g_mu and has `dev` current
{
    auto it = g_static_ws.find(dev);
    if (it == g_static_ws.end()) {
        void *p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_static_ws_array));
        if (e != hipSuccess || !p)
            return fail(WP_ERR_HIP, "hipGetSymbolAddress(static workspace) failed on device %d: %s", dev, hipGetErrorString(e));
        it = g_static_ws.emplace(dev, p).first;
    }
    *out = it->second;
    return WP_OK;
}

extern "C" int wp_init(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_ndev == -2) {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        g_ndev = (e == hipSuccess && n > 0) ? n : -1;
        if (g_ndev < 0) {
            (void)hipGetLastError();
            g_err = std::string("no HIP device: ") + hipGetErrorString(e);
        }
    }
    if (g_ndev < 0) {
        if (g_err.empty()) g_err = "no HIP device";
        return WP_ERR_NO_DEVICE;
    }
    // the current device's static workspace address, once (other devices: on their first
    // wp_init or wp_convolve_dev_noalloc call with that device current)
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && !g_static_ws.count(dev)) {
        void *p = nullptr;
        if (int rc = static_ws_locked(dev, &p)) return rc;
    }
    return g_ndev;
}

extern "C" const char *wp_last_error(void) { return g_err.c_str(); }

static int need_device()
{
    const int n = wp_init();
    return n < 0 ? n : 0;
}

// ---------------------------------------------------------------------------
// integer helpers (host only)
extern "C" int32_t wp_next_pot(int32_t n)
{
    if (n == 0) return 1;
    int32_t s = n - 1;
    s |= s >> 1;
    s |= s >> 2;
    s |= s >> 4;
    s |= s >> 8;
    s |= s >> 16;
    return s + 1;
}

extern "C" int32_t wp_fft_length(int64_t L, int64_t M)
{
    if (L < 0 || M < 0 || L + M > (1 << 30)) return WP_ERR_FFT_SIZE;
    return wp_next_pot((int32_t)(L + M));
}

extern "C" int32_t wp_mix_sl(int64_t rl, double length)
{
    const double p = (double)rl * length;
    if (!isfinite(p)) return 0;
    double m = fmod(trunc(p), 4294967296.0);
    if (m < 0) m += 4294967296.0;
    return (int32_t)(uint32_t)m;
}

extern "C" int64_t wp_ring_pos(int64_t needle, int64_t key, int64_t length)
{
    if (length <= 0) return -1;  // the reference's wrap yields NaN (or never ends) on an empty array: no element
    int64_t pos = needle;
    if (key != 0) {
        pos += key;
        if (pos > length - 1) pos = pos % length;
        else if (pos < 0) pos = ((pos % length) + length) % length;
    }
    return pos;
}

extern "C" int64_t wp_ring_advance(int64_t needle, int64_t length)
{
    if (length <= 0) return -1;
    needle++;
    if (needle > length - 1) needle = 0;
    return needle;
}

extern "C" int64_t wp_beat_rate(double sample_rate, double bpm)
{
    // Math.round(x) = floor(x + 0.5) except that x + 0.5 must not round up across an integer
    // (x = 0.49999999999999994) and huge x has no fraction to round
    const double x = sample_rate * (60.0 / bpm);
    if (!isfinite(x) || fabs(x) >= 9.0e18) return -1;  // Infinity / NaN beat rate (bpm 0): no block length
    double r = floor(x);
    if (x - r >= 0.5) r += 1.0;
    return (int64_t)r;
}

// Number.prototype.toFixed(6) followed by parseFloat, for |x| < 1e21: the multiple of 1e-6
// nearest to the exact value of x (ties away from zero, as the specification's "pick the
// larger n"), then the double nearest to that decimal.
static double js_to_fixed6(double x)
{
    if (!isfinite(x) || fabs(x) >= 9.0e9) return x;  // beyond 2^53 / 1e6 the scaled value has no fraction left to round
    const double ax = fabs(x);
    const double p = ax * 1.0e6;
    const double e = fma(ax, 1.0e6, -p);  // ax * 1e6 == p + e exactly
    double n = floor(p);
    const double f = p - n;               // exact
    if (f > 0.5 || (f == 0.5 && e >= 0.0)) n += 1.0;
    const double r = n / 1.0e6;           // correctly rounded quotient = nearest double to the decimal string
    return x < 0 ? -r : r;
}

extern "C" double wp_bpm_from_beat_rate(double sample_rate, int64_t beat_rate)
{
    return js_to_fixed6(60.0 * (sample_rate / (double)beat_rate));
}

// TypedArray.prototype.subarray(begin, end) index clamping
static void js_subarray(int64_t n, int64_t begin, int64_t end, int64_t *off, int64_t *len)
{
    int64_t b = begin < 0 ? (n + begin > 0 ? n + begin : 0) : (begin < n ? begin : n);
    int64_t e = end < 0 ? (n + end > 0 ? n + end : 0) : (end < n ? end : n);
    *off = b;
    *len = e > b ? e - b : 0;
}

extern "C" int wp_reverb_view_for(int64_t L, int64_t M, int64_t offset, int64_t len, wp_reverb_view *v)
{
    if (!v || L < 0 || M < 0) return fail(WP_ERR_INVALID, "wp_reverb_view_for: bad argument");
    const int64_t n = L + M;
    v->curr_len = n;
    js_subarray(n, offset, offset + len, &v->ret_off, &v->ret_len);
    const int64_t tail = (n - offset) - len;
    js_subarray(n, -tail, n, &v->prev_off, &v->prev_len);
    return WP_OK;
}

#pragma clang fp contract(off)
extern "C" int wp_synth_ir(float *out, int64_t M, uint32_t seed)
{
    if (!out || M <= 0) return fail(WP_ERR_INVALID, "wp_synth_ir: bad argument");
    const uint32_t k = seed * 0x9E3779B9u;
    const float r = (float)(1.0 - 6.907755 / (double)M);
    float e = 1.0f;
    double norm = 0.0;
    for (int64_t i = 0; i < M; i++) {
        uint32_t h = (uint32_t)i ^ k;
        h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
        const double nz = ((double)(h >> 8) - 8388608.0) / 8388608.0;
        out[i] = (float)(nz * (double)e);
        norm += fabs((double)out[i]);
        e = (float)((double)e * (double)r);
    }
    for (int64_t i = 0; i < M; i++) out[i] = (float)((double)out[i] / norm);
    return WP_OK;
}

extern "C" int wp_synth_noise_dev(float *out_dev, int64_t n, uint32_t seed, int64_t start, void *stream)
{
    if (int rc = need_device()) return rc;
    if (!out_dev || n < 0) return fail(WP_ERR_INVALID, "wp_synth_noise_dev: bad argument");
    HIPCHK(launch_synth_noise(out_dev, n, seed, start, (hipStream_t)stream));
    return WP_OK;
}

// ---------------------------------------------------------------------------
// per-(device, B) twiddle tables
struct DevTables {
    c32 *d = nullptr;
};
static std::map<std::pair<int, int>, DevTables> g_tables;

static int get_tables(int dev, int B, const c32 **out)
{
    std::lock_guard<std::mutex> lk(g_mu);
    auto key = std::make_pair(dev, B);
    auto it = g_tables.find(key);
    if (it == g_tables.end()) {
        const int n = table_total_points(B);
        std::vector<c32> h(n);
        const double pi = 3.14159265358979323846;
        for (int a = 0; a < 256; a++) {
            h[a] = {(float)cos(2 * pi * a / 256.0), (float)-sin(2 * pi * a / 256.0)};
            h[256 + a] = {(float)cos(2 * pi * a / B), (float)-sin(2 * pi * a / B)};
        }
        for (int b = 0; b < B / 256; b++)
            h[512 + b] = {(float)cos(2 * pi * b * 256.0 / B), (float)-sin(2 * pi * b * 256.0 / B)};
        const int ho = table_thalf_offset(B), wo = table_wsplit_offset(B);
        for (int j = 0; j < 1024; j++) h[ho + j] = {(float)cos(pi * j / B), (float)-sin(pi * j / B)};
        for (int q = 0; q < B; q++) h[wo + q] = {(float)cos(pi * (q + 0.5) / B), (float)-sin(pi * (q + 0.5) / B)};
        DevTables t;
        HIPCHK(hipMalloc((void **)&t.d, sizeof(c32) * n));
        HIPCHK(hipMemcpy(t.d, h.data(), sizeof(c32) * n, hipMemcpyHostToDevice));
        it = g_tables.emplace(key, t).first;
    }
    *out = it->second.d;
    return WP_OK;
}

// the wp_tune knobs are atomics: wp_tune on one thread may run beside plan creation on another
static std::atomic<int> g_tune_form{0};  // WP_TUNE_CONV_FORM

// per-(device, R, C) tables of the long-frame convolution (layout: wp_internal.h)
static std::map<std::tuple<int, int, int>, c32 *> g_lf_tables;

static int get_lf_tables(int dev, int R, int C, const c32 **out)
{
    std::lock_guard<std::mutex> lk(g_mu);
    auto key = std::make_tuple(dev, R, C);
    auto it = g_lf_tables.find(key);
    if (it == g_lf_tables.end()) {
        std::vector<c32> h(LF_TABLE_POINTS, c32{0.f, 0.f});
        const double pi = 3.14159265358979323846;
        const double Nc = (double)R * C;
        auto e = [](double turns) { return c32{(float)cos(2 * 3.14159265358979323846 * turns), (float)-sin(2 * 3.14159265358979323846 * turns)}; };
        (void)pi;
        for (int a = 0; a < 256; a++) h[LF_T256 + a] = e(a / 256.0);
        for (int r = 0; r < R; r++) h[LF_ROW + r] = e(r / (2.0 * R));
        for (int c = 0; c < C; c++) h[LF_COL + c] = e(c / (2.0 * Nc));
        for (int b = 0; b < 512; b++) h[LF_FLO + b] = e(b / Nc);
        for (int a = 0; a < (int)(Nc / 512) && a < 2048; a++) h[LF_FHI + a] = e(a * 512.0 / Nc);
        for (int k = 0; k < R; k++) h[LF_WA + k] = e((k + 0.5) / (2.0 * Nc));
        for (int k = 0; k < C; k++) h[LF_WB + k] = e(k / (2.0 * C));
        c32 *d = nullptr;
        HIPCHK(hipMalloc((void **)&d, sizeof(c32) * LF_TABLE_POINTS));
        HIPCHK(hipMemcpy(d, h.data(), sizeof(c32) * LF_TABLE_POINTS, hipMemcpyHostToDevice));
        it = g_lf_tables.emplace(key, d).first;
    }
    *out = it->second;
    return WP_OK;
}

// Long-frame form (wp_longframe_kernels.hip) for results much longer than the impulse response:
// frames of N2 = 2*R*C samples (2^17 ... 2^21) advancing by hop = N2 - mpad.  A result is covered by up to
// LF_MAX_SEG runs of equal frames, largest first (e.g. one frame each of 2^21, 2^20 and 2^17 samples for a
// 60 s track with a 2 s impulse response: 3.28 M frame samples where seven frames of 2^19 take 3.67 M).
// Picks the cover that moves the fewest bytes (20 per frame sample + 4 per output sample, + 8 per frame
// point when every job has a kernel spectrum of its own; every further run of frames must pay for its
// three extra launches with 3 % of the bytes) and returns none unless that is below the partitioned
// form's 40 bytes per output sample.
struct LfSeg {
    int R = 0, C = 0;  // a frame is R rows of C complex points
    int nframes = 0;
    int64_t hop = 0;
    int64_t base = 0;  // first output sample of the run
    int64_t pts = 0;   // complex points of the run per job: nframes * R * C
};
struct LfGeom {
    int nseg = 0;  // 0: partitioned form
    LfSeg seg[LF_MAX_SEG];
    int64_t mpad = 0;
    int64_t pts = 0;  // complex points per job over all runs
};
static std::atomic<int> g_tune_lf_runs{0};  // WP_TUNE_LF_RUNS: 0 = up to LF_MAX_SEG runs of frames, n = at most n
static std::atomic<int> g_tune_lf_ab{0};    // WP_TUNE_LF_SPECTRA: 1 = plain kernel spectra even when the jobs share them
static std::atomic<int> g_tune_lf_max{0};   // WP_TUNE_LF_MAX_FRAME: 0 = every frame size, n = frames of at most 2^n samples
static std::atomic<int> g_tune_lf_min{0};   // WP_TUNE_LF_MIN_FRAME: 0 = every frame size, n = frames of at least 2^n samples
static std::atomic<int> g_tune_lf_cost{0};  // WP_TUNE_LF_COST: 1 = covers priced by their bytes alone
static std::atomic<int> g_tune_lf_wide{0};  // WP_TUNE_LF_WIDE: 1 = bus-fold chains over tiles of 16 columns only (frames of 256 rows), 2 = of 32 only
static std::atomic<int> g_tune_lf_split{0}; // WP_TUNE_LF_SPLIT: 1 = every run folds the bus inside its inverse pass, 2 = only the first run does
static LfGeom choose_lf(int64_t L, int64_t M, int nch)
{
    LfGeom best;
    const int tune_form = g_tune_form;
    if (tune_form == 1) return best;
    const int64_t out_len = L + M;
    double best_cost = 0.95 * 40.0 * (double)out_len;  // measured: at 0.92 of the partitioned form's bytes the long-frame form took 0.81 of its time
    if (tune_form == 2) best_cost = 1e300;
    const int64_t mpad = ((M - 1 + 63) / 64) * 64;
    constexpr int NS = 5;
    const int Rs[NS] = {256, 256, 128, 64, 32}, Cs[NS] = {LF_C_BIG, LF_C, LF_C, LF_C, LF_C};
    int64_t N2[NS], hop[NS];
    bool ok[NS];
    const int max_log2 = g_tune_lf_max, min_log2 = g_tune_lf_min;
    for (int i = 0; i < NS; i++) {
        N2[i] = 2 * (int64_t)Rs[i] * Cs[i];
        hop[i] = N2[i] - mpad;
        ok[i] = hop[i] * 4 >= N2[i];  // a frame advances by at least a quarter of its length
        if (max_log2 > 0 && N2[i] > ((int64_t)1 << max_log2)) ok[i] = false;
        if (min_log2 > 0 && N2[i] < ((int64_t)1 << min_log2)) ok[i] = false;
    }
    const int tune_runs = g_tune_lf_runs;
    const int max_runs = tune_runs > 0 && tune_runs < LF_MAX_SEG ? tune_runs : LF_MAX_SEG;
    int64_t n[NS] = {0, 0, 0, 0, 0};
    auto consider = [&]() {
        int64_t samples = 0;
        int runs = 0;
        for (int i = 0; i < NS; i++) {
            samples += n[i] * N2[i];
            runs += n[i] > 0;
        }
        double cost = (double)samples * 20.0 + 4.0 * (double)out_len;
        if (nch > 4) cost += (double)(samples / 2) * 8.0;
        // measured: with kernel spectra shared by the jobs the row pass over rows of LF_C_BIG points (512-thread
        // workgroups, two per CU) takes 1.22 of the time per point that rows of LF_C points take; with a spectrum
        // per job to stream as well it keeps up
        else if (g_tune_lf_cost != 1) cost += 0.22 * 8.0 * (double)(n[0] * N2[0]);
        cost *= 1.0 + 0.03 * (runs - 1);
        if (cost >= best_cost) return;
        best_cost = cost;
        best = LfGeom();
        best.mpad = mpad;
        int64_t base = 0;
        for (int i = 0; i < NS; i++) {
            if (!n[i]) continue;
            LfSeg &g = best.seg[best.nseg++];
            g.R = Rs[i];
            g.C = Cs[i];
            g.nframes = (int)n[i];
            g.hop = hop[i];
            g.base = base;
            g.pts = n[i] * (int64_t)Rs[i] * Cs[i];
            base += n[i] * hop[i];
            best.pts += g.pts;
        }
    };
    auto ceil_div = [](int64_t a, int64_t b) { return a <= 0 ? (int64_t)0 : (a + b - 1) / b; };
    // runs of frames, largest size first: the first run takes up to four frames fewer than would cover what is
    // left, a later run at most eight frames; the last run allowed covers the rest
    std::function<void(int, int64_t, int)> rec = [&](int from, int64_t rem, int runs_left) {
        if (rem <= 0) {
            consider();
            return;
        }
        if (runs_left == 0) return;
        const bool first = runs_left == max_runs;
        const int64_t most = first ? 65535 : 8, span = first ? 4 : 8;
        for (int i = from; i < NS; i++) {
            if (!ok[i]) continue;
            const int64_t hi = ceil_div(rem, hop[i]);
            int64_t lo = runs_left == 1 ? hi : hi - span;
            if (lo < 1) lo = 1;
            for (int64_t k = hi < most ? hi : most; k >= lo; k--) {
                n[i] = k;
                rec(i + 1, rem - k * hop[i], runs_left - 1);
            }
            n[i] = 0;
        }
    };
    rec(0, out_len, max_runs);
    return best;
}

extern "C" int wp_lf_cover(int64_t L, int64_t M, int nch, int64_t *runs)
{
    if (L < 0 || M < 1 || nch < 1 || !runs) return fail(WP_ERR_INVALID, "wp_lf_cover: bad argument");
    const LfGeom g = choose_lf(L, M, nch);
    for (int i = 0; i < g.nseg; i++) {
        runs[4 * i] = 2 * (int64_t)g.seg[i].R * g.seg[i].C;
        runs[4 * i + 1] = g.seg[i].nframes;
        runs[4 * i + 2] = g.seg[i].hop;
        runs[4 * i + 3] = g.seg[i].base;
    }
    return g.nseg;
}

// ---------------------------------------------------------------------------
// Partition spectra built from host samples are shared between handles: every
// worker thread that receives the same kernel object (lib/impulse-convolve.js:9-13
// hands it round through the cache) asks for the same spectra, and the second
// request finds them already in device memory.  Keyed by shape and two independent
// 64-bit hashes of the samples; reference-counted; immutable once built.
struct IrShared {
    c32 *Hp;
    int refs;
};
struct IrKey {
    int dev, nch, B;
    int64_t M, L;
    uint64_t h1, h2;
    bool operator<(const IrKey &o) const
    {
        return std::tie(dev, nch, B, M, L, h1, h2) < std::tie(o.dev, o.nch, o.B, o.M, o.L, o.h1, o.h2);
    }
};
static std::map<IrKey, IrShared> g_ir_cache;  // guarded by g_mu

struct wp_ir {
    int dev;
    int nch;
    int64_t M, L;
    int B, P, Pp, PG;
    c32 *Hp;           // [nch][Pp][B] per-bin partition spectra
    bool shared;       // Hp belongs to g_ir_cache[key]
    IrKey key;
    const c32 *tables;
    LfGeom lf;             // long-frame geometry for results of L+M samples (lf.nseg == 0: not used)
    c32 *lf_Hs[LF_MAX_SEG];       // per run of frames: [nch][R][LF_C] real spectra of the zero-padded kernels, scaled; owned by this handle
    const c32 *lf_tab[LF_MAX_SEG];  // long-frame tables of each run's R
    const c32 *lf_tab_row[LF_MAX_SEG];  // tables of each run's C-point row transforms
    bool lf_ab;            // lf_Hs holds (alpha, beta) coefficient pairs per bin (few kernels, shared by the jobs) instead of spectra
    int64_t bytes;
    wp_conv *single;   // lazily built 1-job plan for wp_convolve / wp_convolve_dev
    int single_ch;     // IR channel that plan currently names
    wp_conv *wrap;     // 1-job plan with a longer result, for inputs that wrap in the reference's FFT
    float *wrap_lin;   // its linear-convolution result, wrap_len floats
    int64_t wrap_len;
    int wrap_ch;
    float *stage_in, *stage_out;
};

struct wp_conv {
    wp_ir *ir;
    int njobs;
    int64_t out_len;    // samples per job result: L+M (longer only for the library's own wrap plan)
    int T;
    int tchunk;
    bool gather;        // small batch: gather-form MAC (one workgroup per output block) instead of the streaming one
    int32_t *job_irch;  // device table of each job's IR channel; nullptr for one-job plans, whose
    int irch_all;       // channel travels as a launch argument (it changes from call to call)
    c32 *Z, *W;
    bool lf;            // long-frame form: A holds, run of frames after run, [njobs][nframes][R][C] instead of Z and W
    c32 *A;
    hipStream_t side[LF_MAX_SEG];   // plans with more than one run of frames: run i > 0's inverse pass goes to side[i],
    hipEvent_t fork, join[LF_MAX_SEG];  // beside the first run's on the caller's stream
    int wide_frames = 0, wide_for = -1;  // run 0's frames whose bus-fold chains take the wide tiles, chosen for wide_for frames run
    // a short run's bus fold as a pass of the mix kernel over its range of the bus (conv_back_lf): per run, the
    // source table + the kernel's scratch on the device, a pinned mirror of the table and what it currently holds
    void *mix_dev[LF_MAX_SEG];
    wp_src *mix_pin[LF_MAX_SEG];
    hipEvent_t mix_ev[LF_MAX_SEG];   // recorded behind the last upload from mix_pin[i]
    std::vector<wp_src> mix_host[LF_MAX_SEG];
    // one stream at a time may use a plan (Z and W are shared by its calls): a call on a
    // different stream than the previous one first waits for that call's last kernel
    hipStream_t last_stream;
    bool used;
    hipEvent_t done;
    int64_t bytes;
    bool prof;
    bool prof_valid;
    hipEvent_t ev[4];  // the set of the current (last) profiled run: one of evr's
    // profiled runs rotate through PROF_RING sets of four events, so that a caller can time many runs back to back and
    // add them up afterwards (wp_conv_profile_sum) instead of waiting for every run before it queues the next
    static constexpr int PROF_RING = 32;
    hipEvent_t evr[PROF_RING][4];
    bool ev_used[PROF_RING];  // recorded, not yet added to prof_sum
    int ev_cur;
    double prof_sum[3];
    int prof_runs;
    double *vol_dev;    // per-source gains of the fused reverb+mix call
    std::vector<double> vol_host;  // what vol_dev holds (an unchanged gain table is not uploaded again)
};

struct wp_reverb {
    wp_conv *conv;
    int njobs;
    int64_t offset;
    int64_t stride;     // floats between jobs in buf[]
    float *buf[2];      // ping-pong curr buffers, njobs*stride each
    int cur;            // ping-pong buffer written last by wp_reverb_block_dev
    const float *prev;  // carried tail of job 0 (job j at prev + j*prev_stride): inside buf[] or a caller's buffer
    int64_t prev_stride;
    int64_t prev_off, prev_len;  // the tail as a view of the latest curr: curr[prev_off, prev_off+prev_len)
    float *stage_in;    // host-API staging, njobs*in_stride
    int64_t stage_in_len;
};

// process-wide overrides set through wp_tune (0 = the library's own choice)
static std::atomic<int> g_tune_B{0}, g_tune_PG{0}, g_tune_mac{0};
static std::atomic<long> g_tune_pool_mb{2048};

extern "C" int wp_tune(int what, int64_t value)
{
    std::lock_guard<std::mutex> lk(g_mu);
    switch (what) {
    case WP_TUNE_PARTITION:
        if (value != 0 && value != 1024 && value != 2048 && value != 4096 && value != 8192) break;
        g_tune_B = (int)value;
        return WP_OK;
    case WP_TUNE_MAC_GROUP:
        if (value != 0 && value != 4 && value != 8 && value != 12 && value != 16 && value != 24) break;
        g_tune_PG = (int)value;
        return WP_OK;
    case WP_TUNE_MAC_FORM:
        if (value < 0 || value > 2) break;
        g_tune_mac = (int)value;
        return WP_OK;
    case WP_TUNE_CONV_FORM:
        if (value < 0 || value > 2) break;
        g_tune_form = (int)value;
        return WP_OK;
    case WP_TUNE_MID_FORM:
        if (value < 0 || value > 2) break;
        set_mid_jobs(value == 1 ? 0 : 8);
        return WP_OK;
    case WP_TUNE_LF_SPECTRA:
        if (value < 0 || value > 2) break;
        g_tune_lf_ab = (int)value;
        return WP_OK;
    case WP_TUNE_LF_RUNS:
        if (value < 0 || value > LF_MAX_SEG) break;
        g_tune_lf_runs = (int)value;
        return WP_OK;
    case WP_TUNE_LF_MAX_FRAME:
        if (value != 0 && (value < 17 || value > 21)) break;
        g_tune_lf_max = (int)value;
        return WP_OK;
    case WP_TUNE_LF_MIN_FRAME:
        if (value != 0 && (value < 17 || value > 21)) break;
        g_tune_lf_min = (int)value;
        return WP_OK;
    case WP_TUNE_LF_COST:
        if (value < 0 || value > 1) break;
        g_tune_lf_cost = (int)value;
        return WP_OK;
    case WP_TUNE_LF_SPLIT:
        if (value < 0 || value > 2) break;
        g_tune_lf_split = (int)value;
        return WP_OK;
    case WP_TUNE_LF_WIDE:
        if (value < 0 || value > 2) break;
        g_tune_lf_wide = (int)value;
        return WP_OK;
    case WP_TUNE_FWD_FORM:
        if (value < 0 || value > 2) break;
        set_fwd_pair(value != 1);
        return WP_OK;
    case WP_TUNE_POOL_MB:
        if (value < 0 || value > (1 << 20)) break;
        g_tune_pool_mb = (long)value;
        return WP_OK;
    }
    return fail(WP_ERR_INVALID, "wp_tune: knob %d does not take the value %lld", what, (long long)value);
}

static int choose_B(int64_t L, int64_t M)
{
    if (const int b = g_tune_B) return b;
    if (L + M <= 8192) return 1024;
    // one MAC pass covers 24 partitions: IRs longer than 24*4096 samples use the larger partition
    return (M > 24 * 4096) ? 8192 : 4096;
}

static int choose_PG(int P)
{
    if (const int pg = g_tune_PG) return pg;
    if (P <= 4) return 4;
    if (P <= 8) return 8;
    if (P <= 12) return 12;
    if (P <= 16) return 16;
    return 24;  // longer IRs run ceil(P/24) accumulating passes of 24 partitions
}

// hashes: nullptr, or the two content hashes of the host samples (enables sharing)
static wp_ir *ir_create_common(const float *ir_dev, int nch, int64_t ch_stride, int64_t M, int64_t L, hipStream_t s,
                               const uint64_t *hashes = nullptr)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        fail(WP_ERR_HIP, "hipGetDevice failed");
        return nullptr;
    }
    wp_ir *ir = new wp_ir();
    ir->dev = dev;
    ir->nch = nch;
    ir->M = M;
    ir->L = L;
    ir->B = choose_B(L, M);
    ir->P = (int)((M + ir->B - 1) / ir->B);
    ir->PG = choose_PG(ir->P);
    ir->Pp = ((ir->P + ir->PG - 1) / ir->PG) * ir->PG;
    ir->single = nullptr;
    ir->single_ch = 0;
    ir->wrap = nullptr;
    ir->wrap_lin = nullptr;
    ir->wrap_len = 0;
    ir->wrap_ch = 0;
    ir->stage_in = ir->stage_out = nullptr;
    ir->Hp = nullptr;
    ir->shared = false;
    ir->lf = choose_lf(L, M, nch);
    for (int i = 0; i < LF_MAX_SEG; i++) {
        ir->lf_Hs[i] = nullptr;
        ir->lf_tab[i] = nullptr;
        ir->lf_tab_row[i] = nullptr;
    }
    ir->lf_ab = nch <= 4 && g_tune_lf_ab != 1;
    auto free_lf = [&]() {
        for (auto &h : ir->lf_Hs) {
            if (h) (void)hipFree(h);
            h = nullptr;
        }
    };
    c32 *Zir = nullptr;
    auto bail = [&](const char *what, hipError_t e) {
        fail(WP_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
        if (Zir) (void)hipFree(Zir);
        if (ir->Hp) (void)hipFree(ir->Hp);
        free_lf();
        delete ir;
        return (wp_ir *)nullptr;
    };
    if (get_tables(dev, ir->B, &ir->tables) != WP_OK) {
        delete ir;
        return nullptr;
    }
    const size_t hbytes = sizeof(c32) * (size_t)nch * ir->Pp * ir->B;
    ir->bytes = (int64_t)hbytes;
    for (int i = 0; i < ir->lf.nseg; i++) {
        // kernel spectra of the long-frame form: the kernels as one "frame" each, through the same
        // column and row passes the signal takes, once per frame size
        const int R = ir->lf.seg[i].R, C = ir->lf.seg[i].C;
        const size_t lbytes = sizeof(c32) * (size_t)nch * R * C;  // of the transform; the coefficient form stores twice that
        c32 *tmpA = nullptr;
        hipError_t e = hipSuccess;
        if (get_lf_tables(dev, R, C, &ir->lf_tab[i]) != WP_OK || get_tables(dev, C, &ir->lf_tab_row[i]) != WP_OK) {
            free_lf();
            delete ir;
            return nullptr;
        }
        if ((e = hipMalloc((void **)&ir->lf_Hs[i], lbytes * (ir->lf_ab ? 2 : 1))) == hipSuccess) e = hipMalloc((void **)&tmpA, lbytes);
        if (e == hipSuccess) e = launch_lf_fwd(R, C, ir_dev, ch_stride, M, tmpA, 1, 1, nch, 0, 0, 0, ir->lf_tab[i], s);
        if (e == hipSuccess) e = launch_lf_mid(true, ir->lf_ab, R, C, tmpA, ir->lf_Hs[i], nullptr, 0, 1, 1, nch, ir->lf_tab[i], ir->lf_tab_row[i], s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (tmpA) (void)hipFree(tmpA);
        if (e != hipSuccess) {
            fail(WP_ERR_HIP, "long-frame kernel spectra: %s", hipGetErrorString(e));
            free_lf();
            delete ir;
            return nullptr;
        }
        ir->bytes += (int64_t)lbytes * (ir->lf_ab ? 2 : 1);
    }
    if (hashes) {
        ir->key = IrKey{dev, nch, ir->B, M, L, hashes[0], hashes[1]};
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_ir_cache.find(ir->key);
        if (it != g_ir_cache.end()) {
            it->second.refs++;
            ir->Hp = it->second.Hp;
            ir->shared = true;
            return ir;
        }
    }
    hipError_t e;
    if ((e = hipMalloc((void **)&ir->Hp, hbytes)) != hipSuccess) return bail("hipMalloc(Hp)", e);
    if ((e = hipMemsetAsync(ir->Hp, 0, hbytes, s)) != hipSuccess) return bail("hipMemsetAsync", e);
    if ((e = hipMalloc((void **)&Zir, sizeof(c32) * (size_t)nch * ir->P * ir->B)) != hipSuccess) return bail("hipMalloc(Zir)", e);
    if ((e = launch_fft_fwd(ir->B, true, ir_dev, ch_stride, M, Zir, ir->P, ir->P, nch, ir->tables, s)) != hipSuccess)
        return bail("k_fft_fwd(IR)", e);
    if ((e = launch_hpack(Zir, ir->Hp, ir->tables + table_wsplit_offset(ir->B), ir->B, ir->P, ir->Pp, nch, s)) != hipSuccess)
        return bail("k_hpack", e);
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return bail("hipStreamSynchronize", e);
    (void)hipFree(Zir);
    if (hashes) {
        std::lock_guard<std::mutex> lk(g_mu);
        auto ins = g_ir_cache.emplace(ir->key, IrShared{ir->Hp, 1});
        if (!ins.second) {  // another thread built the same spectra meanwhile: use theirs
            (void)hipFree(ir->Hp);
            ins.first->second.refs++;
            ir->Hp = ins.first->second.Hp;
        }
        ir->shared = true;
    }
    return ir;
}

static void hash_samples(const float *const *ir, int nch, int64_t M, uint64_t out[2])
{
    uint64_t a = 0xcbf29ce484222325ull, b = 0x9e3779b97f4a7c15ull;
    for (int c = 0; c < nch; c++) {
        const uint32_t *w = reinterpret_cast<const uint32_t *>(ir[c]);
        for (int64_t i = 0; i < M; i++) {
            uint32_t v;
            memcpy(&v, &w[i], sizeof(v));
            a = (a ^ v) * 0x100000001b3ull;                       // FNV-1a over 32-bit words
            b = (b + v) * 0xff51afd7ed558ccdull; b ^= b >> 29;    // multiply-xorshift chain
        }
        a = (a ^ 0xffu) * 0x100000001b3ull;                       // channel separator
    }
    out[0] = a;
    out[1] = b;
}

static int check_ir_args(int nch, int64_t M, int64_t L)
{
    if (nch < 1) return fail(WP_ERR_INVALID, "IR needs at least one channel");
    if (M < 1 || L < 0) return fail(WP_ERR_INVALID, "IR length must be >= 1 and block length >= 0 (M=%lld L=%lld)", (long long)M, (long long)L);
    if (L + M > (1 << 30)) return fail(WP_ERR_FFT_SIZE, "L+M=%lld exceeds the reference's int32 nextPowerOfTwo range", (long long)(L + M));
    if (wp_next_pot((int32_t)(L + M)) <= 1) return fail(WP_ERR_FFT_SIZE, "FFT size must be a power of two and bigger than 1");
    return WP_OK;
}

extern "C" wp_ir *wp_ir_create_dev(const float *ir_dev, int nch, int64_t ch_stride, int64_t M, int64_t L, void *stream)
{
    if (need_device()) return nullptr;
    if (!ir_dev) {
        fail(WP_ERR_INVALID, "wp_ir_create_dev: null IR");
        return nullptr;
    }
    if (check_ir_args(nch, M, L)) return nullptr;
    return ir_create_common(ir_dev, nch, ch_stride, M, L, (hipStream_t)stream);
}

extern "C" wp_ir *wp_ir_create(const float *const *ir, int nch, int64_t M, int64_t L)
{
    if (need_device()) return nullptr;
    if (!ir) {
        fail(WP_ERR_INVALID, "wp_ir_create: null IR");
        return nullptr;
    }
    if (check_ir_args(nch, M, L)) return nullptr;
    float *d = nullptr;
    hipError_t e = hipMalloc((void **)&d, sizeof(float) * (size_t)nch * M);
    if (e != hipSuccess) {
        fail(WP_ERR_HIP, "hipMalloc failed: %s", hipGetErrorString(e));
        return nullptr;
    }
    for (int c = 0; c < nch; c++) {
        e = hipMemcpy(d + (size_t)c * M, ir[c], sizeof(float) * (size_t)M, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            fail(WP_ERR_HIP, "hipMemcpy(IR) failed: %s", hipGetErrorString(e));
            (void)hipFree(d);
            return nullptr;
        }
    }
    uint64_t hashes[2];
    hash_samples(ir, nch, M, hashes);
    wp_ir *r = ir_create_common(d, nch, M, M, L, nullptr, hashes);
    (void)hipFree(d);
    return r;
}

extern "C" void wp_ir_destroy(wp_ir *ir)
{
    if (!ir) return;
    if (ir->single) wp_conv_destroy(ir->single);
    if (ir->wrap) wp_conv_destroy(ir->wrap);
    if (ir->wrap_lin) (void)hipFree(ir->wrap_lin);
    if (ir->stage_in) (void)hipFree(ir->stage_in);
    if (ir->stage_out) (void)hipFree(ir->stage_out);
    for (auto &h : ir->lf_Hs)
        if (h) (void)hipFree(h);
    if (ir->shared) {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_ir_cache.find(ir->key);
        if (it != g_ir_cache.end() && --it->second.refs == 0) {
            (void)hipFree(it->second.Hp);
            g_ir_cache.erase(it);
        }
    } else if (ir->Hp) {
        (void)hipFree(ir->Hp);
    }
    delete ir;
}

extern "C" int64_t wp_ir_info(const wp_ir *ir, int which)
{
    if (!ir) return WP_ERR_INVALID;
    switch (which) {
    case 0: return ir->M;
    case 1: return ir->L;
    case 2: return ir->nch;
    case 3: return ir->B;
    case 4: return ir->P;
    case 5: return ir->L + ir->M;
    case 6: return wp_next_pot((int32_t)(ir->L + ir->M));
    case 7: return ir->bytes;
    case 9: return ir->lf.nseg ? 2 * (int64_t)ir->lf.seg[0].R * ir->lf.seg[0].C : 0;
    case 10: return ir->lf.nseg;
    case 11: return 2 * ir->lf.pts;
    case 8: {
        if (!ir->shared) return 1;
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_ir_cache.find(ir->key);
        return it == g_ir_cache.end() ? 1 : it->second.refs;
    }
    }
    return WP_ERR_INVALID;
}

// ---------------------------------------------------------------------------
static wp_conv *conv_create(wp_ir *ir, int njobs, const int32_t *job_ch, int64_t out_len, bool allow_lf = true);

extern "C" wp_conv *wp_conv_create(wp_ir *ir, int njobs, const int32_t *job_ch)
{
    return conv_create(ir, njobs, job_ch, ir ? ir->L + ir->M : 0);
}

// Geometry of a plan (everything but its device memory): output blocks, MAC time chunks and
// which form of the spectral multiply-accumulate runs.
static void plan_geometry(wp_conv *c, wp_ir *ir, int njobs, int64_t out_len)
{
    c->ir = ir;
    c->njobs = njobs;
    c->out_len = out_len;
    c->T = (int)((out_len + ir->B - 1) / ir->B);
    // time chunks: enough threads to fill the chip when there are few jobs
    {
        const int64_t per_job = ir->B;
        int64_t want = (256 * 8 * 64 + per_job * njobs - 1) / (per_job * njobs);
        int64_t maxchunks = c->T / (2 * ir->PG);
        if (maxchunks < 1) maxchunks = 1;
        if (want > maxchunks) want = maxchunks;
        if (want < 1) want = 1;
        c->tchunk = (int)((c->T + want - 1) / want);
    }
    {
        // The streaming MAC needs about a workgroup per CU to hide its per-step latency; below
        // that, and while the block spectra sit in cache, the gather form finishes sooner.
        const int64_t stream_wgs = (int64_t)(ir->B / 256) * njobs * ((c->T + c->tchunk - 1) / c->tchunk);
        const int64_t zbytes = (int64_t)sizeof(c32) * njobs * c->T * ir->B;
        c->gather = stream_wgs < 256 && zbytes <= ((int64_t)64 << 20) && c->T <= 65535;
        if (g_tune_mac == 1) c->gather = false;
        if (g_tune_mac == 2 && c->T <= 65535) c->gather = true;
    }
    c->job_irch = nullptr;
    c->irch_all = 0;
    c->Z = c->W = nullptr;
    c->lf = false;
    c->A = nullptr;
    c->fork = nullptr;
    for (int i = 0; i < LF_MAX_SEG; i++) {
        c->side[i] = nullptr;
        c->join[i] = nullptr;
        c->mix_dev[i] = nullptr;
        c->mix_pin[i] = nullptr;
        c->mix_ev[i] = nullptr;
    }
    c->bytes = 0;
    c->prof = c->prof_valid = false;
    for (auto &e : c->ev) e = nullptr;
    for (auto &set : c->evr)
        for (auto &e : set) e = nullptr;
    for (auto &u : c->ev_used) u = false;
    c->ev_cur = 0;
    c->prof_sum[0] = c->prof_sum[1] = c->prof_sum[2] = 0.0;
    c->prof_runs = 0;
    c->vol_dev = nullptr;
    c->last_stream = nullptr;
    c->used = false;
    c->done = nullptr;
}

static wp_conv *conv_create(wp_ir *ir, int njobs, const int32_t *job_ch, int64_t out_len, bool allow_lf)
{
    if (need_device()) return nullptr;
    if (!ir || njobs < 1 || njobs > 65535 || !job_ch) {
        fail(WP_ERR_INVALID, "wp_conv_create: bad argument (njobs must be 1..65535)");
        return nullptr;
    }
    for (int j = 0; j < njobs; j++)
        if (job_ch[j] < 0 || job_ch[j] >= ir->nch) {
            fail(WP_ERR_INVALID, "wp_conv_create: job %d names IR channel %d of %d", j, job_ch[j], ir->nch);
            return nullptr;
        }
    wp_conv *c = new wp_conv();
    plan_geometry(c, ir, njobs, out_len);
    const size_t sbytes = sizeof(c32) * (size_t)njobs * c->T * ir->B;
    hipError_t e = hipSuccess;
    if (njobs == 1) {
        c->irch_all = job_ch[0];
    } else {
        e = hipMalloc((void **)&c->job_irch, sizeof(int32_t) * njobs);
        if (e == hipSuccess) e = hipMemcpy(c->job_irch, job_ch, sizeof(int32_t) * njobs, hipMemcpyHostToDevice);
    }
    c->lf = allow_lf && ir->lf.nseg && out_len == ir->L + ir->M;
    const size_t abytes = c->lf ? sizeof(c32) * (size_t)njobs * ir->lf.pts : 0;
    if (c->lf) {
        if (e == hipSuccess) e = hipMalloc((void **)&c->A, abytes);
        if (ir->lf.nseg > 1 && e == hipSuccess) e = hipEventCreateWithFlags(&c->fork, hipEventDisableTiming);
        for (int i = 1; i < ir->lf.nseg; i++) {
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->side[i], hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&c->join[i], hipEventDisableTiming);
        }
    } else {
        if (e == hipSuccess) e = hipMalloc((void **)&c->Z, sbytes);
        if (e == hipSuccess) e = hipMalloc((void **)&c->W, sbytes);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->done, hipEventDisableTiming);
    if (e != hipSuccess) {
        fail(WP_ERR_HIP, "wp_conv_create: allocating %zu bytes of spectra failed: %s", c->lf ? abytes : 2 * sbytes, hipGetErrorString(e));
        wp_conv_destroy(c);
        return nullptr;
    }
    c->bytes = (int64_t)((c->lf ? abytes : 2 * sbytes) + (c->job_irch ? sizeof(int32_t) * njobs : 0));
    return c;
}

extern "C" void wp_conv_destroy(wp_conv *c)
{
    if (!c) return;
    if (c->job_irch) (void)hipFree(c->job_irch);
    if (c->Z) (void)hipFree(c->Z);
    if (c->W) (void)hipFree(c->W);
    if (c->A) (void)hipFree(c->A);
    if (c->vol_dev) (void)hipFree(c->vol_dev);
    if (c->done) (void)hipEventDestroy(c->done);
    if (c->fork) (void)hipEventDestroy(c->fork);
    for (int i = 0; i < LF_MAX_SEG; i++) {
        if (c->side[i]) {
            (void)hipStreamSynchronize(c->side[i]);
            (void)hipStreamDestroy(c->side[i]);
        }
        if (c->join[i]) (void)hipEventDestroy(c->join[i]);
        if (c->mix_ev[i]) (void)hipEventDestroy(c->mix_ev[i]);
        if (c->mix_dev[i]) (void)hipFree(c->mix_dev[i]);
        if (c->mix_pin[i]) (void)hipHostFree(c->mix_pin[i]);
    }
    for (auto &set : c->evr)
        for (auto &e : set)
            if (e) (void)hipEventDestroy(e);
    delete c;
}

// Stream hand-over of a plan: Z, W and the gain table are shared by every call, so a call
// on another stream than the last one waits (on the device) for that call to finish.
static int plan_enter(wp_conv *c, hipStream_t s)
{
    if (c->done && c->used && c->last_stream != s) HIPCHK(hipStreamWaitEvent(s, c->done, 0));
    return WP_OK;
}
static int plan_leave(wp_conv *c, hipStream_t s)
{
    if (c->done) {
        HIPCHK(hipEventRecord(c->done, s));
        c->last_stream = s;
        c->used = true;
    }
    return WP_OK;
}

extern "C" int64_t wp_conv_workspace_bytes(const wp_conv *c) { return c ? c->bytes : WP_ERR_INVALID; }

extern "C" int64_t wp_conv_info(const wp_conv *c, int which)
{
    if (!c) return WP_ERR_INVALID;
    switch (which) {
    case 0: return c->njobs;
    case 1: return c->T;
    case 2: return c->ir->B;
    case 3: return c->ir->Pp;
    case 4: return c->tchunk;
    case 5: return c->gather ? 1 : 0;
    case 6: return c->lf ? 1 : 0;
    case 7: return c->lf ? 2 * (int64_t)c->ir->lf.seg[0].R * c->ir->lf.seg[0].C : 0;
    case 8: return c->lf ? c->ir->lf.seg[0].nframes : 0;
    case 9: return c->lf ? c->ir->lf.seg[0].hop : 0;
    case 10: return c->lf ? c->ir->lf.nseg : 0;
    case 11: return c->lf ? 2 * c->ir->lf.pts : 0;
    }
    if (which >= 16 && which < 16 + 4 * LF_MAX_SEG) {  // run of frames (which-16)/4: frame length, frames, hop, first output sample
        const int i = (which - 16) / 4;
        if (!c->lf || i >= c->ir->lf.nseg) return 0;
        const LfSeg &g = c->ir->lf.seg[i];
        switch ((which - 16) % 4) {
        case 0: return 2 * (int64_t)g.R * g.C;
        case 1: return g.nframes;
        case 2: return g.hop;
        case 3: return g.base;
        }
    }
    return WP_ERR_INVALID;
}

extern "C" int wp_conv_profile(wp_conv *c, int enable)
{
    if (int rc = need_device()) return rc;
    if (!c) return fail(WP_ERR_INVALID, "wp_conv_profile: null plan");
    if (enable && !c->evr[0][0]) {
        for (auto &set : c->evr)
            for (auto &e : set) HIPCHK(hipEventCreate(&e));
        for (int i = 0; i < 4; i++) c->ev[i] = c->evr[0][i];
    }
    c->prof = enable != 0;
    c->prof_valid = false;
    return WP_OK;
}

// adds a recorded set's three durations to the sums (waits for the set's last event: it was recorded PROF_RING runs ago
// when this is called from prof_begin)
static int prof_flush(wp_conv *c, int k)
{
    if (!c->ev_used[k]) return WP_OK;
    HIPCHK(hipEventSynchronize(c->evr[k][3]));
    for (int i = 0; i < 3; i++) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, c->evr[k][i], c->evr[k][i + 1]));
        c->prof_sum[i] += ms;
    }
    c->prof_runs++;
    c->ev_used[k] = false;
    return WP_OK;
}
// a profiled run takes the next set of events
static int prof_begin(wp_conv *c)
{
    if (!c->prof) return WP_OK;
    c->ev_cur = (c->ev_cur + 1) % wp_conv::PROF_RING;
    if (int rc = prof_flush(c, c->ev_cur)) return rc;
    for (int i = 0; i < 4; i++) c->ev[i] = c->evr[c->ev_cur][i];
    return WP_OK;
}
static void prof_end(wp_conv *c)
{
    c->ev_used[c->ev_cur] = true;
    c->prof_valid = true;
}

extern "C" int wp_conv_profile_sum(wp_conv *c, double *ms3_sum, int *runs)
{
    if (int rc = need_device()) return rc;
    if (!c || !ms3_sum || !runs) return fail(WP_ERR_INVALID, "wp_conv_profile_sum: null argument");
    if (!c->evr[0][0]) return fail(WP_ERR_INVALID, "wp_conv_profile_sum: profiling was never enabled on this plan");
    for (int k = 1; k <= wp_conv::PROF_RING; k++)  // oldest first
        if (int rc = prof_flush(c, (c->ev_cur + k) % wp_conv::PROF_RING)) return rc;
    for (int i = 0; i < 3; i++) {
        ms3_sum[i] = c->prof_sum[i];
        c->prof_sum[i] = 0.0;
    }
    *runs = c->prof_runs;
    c->prof_runs = 0;
    return WP_OK;
}

extern "C" int wp_conv_profile_read(wp_conv *c, float *ms3)
{
    if (int rc = need_device()) return rc;
    if (!c || !ms3) return fail(WP_ERR_INVALID, "wp_conv_profile_read: null argument");
    if (!c->prof || !c->prof_valid) return fail(WP_ERR_INVALID, "wp_conv_profile_read: no profiled run recorded");
    HIPCHK(hipEventSynchronize(c->ev[3]));
    for (int i = 0; i < 3; i++) HIPCHK(hipEventElapsedTime(&ms3[i], c->ev[i], c->ev[i + 1]));
    return WP_OK;
}

// true when the reference's N-point circular convolution of this input differs from the
// linear one inside the first L+M samples it returns (lib/convolve.js:571-582)
static bool input_wraps(const wp_ir *ir, int64_t in_len)
{
    return in_len > ir->L && in_len + ir->M - 1 > wp_next_pot((int32_t)(ir->L + ir->M));
}

static int check_in_len(const wp_ir *ir, int64_t in_len, bool wrap_ok = false)
{
    const int64_t N = wp_next_pot((int32_t)(ir->L + ir->M));
    if (in_len < 0) return fail(WP_ERR_INVALID, "negative input length");
    if (in_len > N) return fail(WP_ERR_RANGE, "offset is out of bounds: input of %lld samples exceeds the FFT length %lld", (long long)in_len, (long long)N);
    if (!wrap_ok && input_wraps(ir, in_len))
        return fail(WP_ERR_UNSUPPORTED, "input of %lld samples wraps circularly in the reference's %lld-point FFT; only wp_convolve and wp_convolve_dev implement that", (long long)in_len, (long long)N);
    return WP_OK;
}

// forward FFTs + spectral MAC for output blocks [0, Tout); leaves W ready
// long-frame form: column pass + row pass for the first frames_run frames; leaves A ready for launch_lf_inv
// `limit`: samples of the result that are needed (frames beyond them are skipped)
static int lf_frames_run(const LfSeg &g, int64_t limit)
{
    if (limit <= g.base) return 0;
    const int64_t n = (limit - g.base + g.hop - 1) / g.hop;
    return n < g.nframes ? (int)n : g.nframes;
}
static c32 *lf_seg_A(const wp_conv *c, int i)
{
    int64_t off = 0;
    for (int k = 0; k < i; k++) off += c->ir->lf.seg[k].pts;
    return c->A + (int64_t)c->njobs * off;
}
static int conv_front_lf(wp_conv *c, const float *in_dev, int64_t in_stride, int64_t in_len, int64_t limit, hipStream_t s)
{
    const wp_ir *ir = c->ir;
    if (int rc = prof_begin(c)) return rc;
    if (c->prof) HIPCHK(hipEventRecord(c->ev[0], s));
    for (int i = 0; i < ir->lf.nseg; i++) {
        const LfSeg &g = ir->lf.seg[i];
        HIPCHK(launch_lf_fwd(g.R, g.C, in_dev, in_stride, in_len, lf_seg_A(c, i), g.nframes, lf_frames_run(g, limit), c->njobs, g.base,
                             g.hop, ir->lf.mpad, ir->lf_tab[i], s));
    }
    if (c->prof) HIPCHK(hipEventRecord(c->ev[1], s));
    for (int i = 0; i < ir->lf.nseg; i++) {
        const LfSeg &g = ir->lf.seg[i];
        HIPCHK(launch_lf_mid(false, ir->lf_ab, g.R, g.C, lf_seg_A(c, i), ir->lf_Hs[i], c->job_irch, c->irch_all, g.nframes, lf_frames_run(g, limit),
                             c->njobs, ir->lf_tab[i], ir->lf_tab_row[i], s));
    }
    if (c->prof) HIPCHK(hipEventRecord(c->ev[2], s));
    return WP_OK;
}
// compute units of the current device (asked once per device)
static int device_cus()
{
    static std::atomic<int> ncu[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int n = ncu[dev].load(std::memory_order_relaxed);
    if (!n) {
        hipDeviceProp_t prop;
        n = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        ncu[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

// The bus fold of run i as a pass of the mix kernel over the run's range [lo, hi) of the bus, reading the results
// the plain inverse pass has just written: sources s = rows (2s, 2s+1) of `out`, in order, the same arithmetic
// as the fused fold (src/mix-buffers.js:49-52).
static int lf_mix_range(wp_conv *c, int i, int64_t lo, int64_t hi, int nsrc, float *bus0, float *bus1, const float *out_dev,
                        int64_t out_stride, hipStream_t q)
{
    const size_t tab_bytes = (sizeof(wp_src) * (size_t)nsrc + 255) & ~(size_t)255;
    if (!c->mix_dev[i]) {
        HIPCHK(hipMalloc(&c->mix_dev[i], tab_bytes + mix_scratch_bytes(nsrc)));
        HIPCHK(hipHostMalloc((void **)&c->mix_pin[i], tab_bytes, hipHostMallocDefault));
        HIPCHK(hipEventCreateWithFlags(&c->mix_ev[i], hipEventDisableTiming));
    }
    std::vector<wp_src> tab((size_t)nsrc);
    for (int s = 0; s < nsrc; s++) {
        wp_src &S = tab[(size_t)s];
        memset(&S, 0, sizeof(S));
        S.ch[0] = out_dev + (int64_t)(2 * s) * out_stride + lo;
        S.ch[1] = out_dev + (int64_t)(2 * s + 1) * out_stride + lo;
        S.nch = 2;
        S.rl = S.sl = hi - lo;
        S.vol = c->vol_host[(size_t)s];
        S.o = 0;
    }
    // an unchanged table (same result buffer and gains, bar after bar) is not uploaded again
    if (c->mix_host[i].size() != tab.size() || memcmp(c->mix_host[i].data(), tab.data(), sizeof(wp_src) * tab.size()) != 0) {
        if (!c->mix_host[i].empty()) HIPCHK(hipEventSynchronize(c->mix_ev[i]));  // the previous upload has left the mirror
        c->mix_host[i].clear();
        memcpy(c->mix_pin[i], tab.data(), sizeof(wp_src) * tab.size());
        HIPCHK(hipMemcpyAsync(c->mix_dev[i], c->mix_pin[i], sizeof(wp_src) * tab.size(), hipMemcpyHostToDevice, q));
        HIPCHK(hipEventRecord(c->mix_ev[i], q));
        c->mix_host[i] = tab;
    }
    HIPCHK(launch_mix(bus0 + lo, bus1 + lo, 2, hi - lo, reinterpret_cast<const wp_src *>(c->mix_dev[i]), (char *)c->mix_dev[i] + tab_bytes,
                      nsrc, q));
    return WP_OK;
}

// Bus-fold chains of a run whose frames have 256 rows, results written as well: how many of its first frames take
// tiles of 32 columns (256-byte runs; a chain of 512 threads that has a CU to itself and takes 1.5 times as long as
// a chain over a tile of 16 columns does, for twice the samples), the rest 16.  Chosen for the shortest schedule of
// the chains on the CUs, one chain per CU at a time, the long ones first; ties go to the wider tiles.
static int lf_wide_frames(int C, int fr, int ncu)
{
    if (g_tune_lf_wide == 1) return 0;
    if (g_tune_lf_wide == 2) return fr;
    int best = 0;
    int64_t best_t = -1;
    const int lo = fr > 8 ? fr - 8 : 0;  // with many frames the few last ones decide how the schedule ends
    for (int wide = fr; wide >= lo; wide--) {
        // loads in half units: a wide chain 3, a narrow one 2
        const int64_t nw = (int64_t)(C / 32) * wide * 2, nn = (int64_t)(C / 16) * (fr - wide) * 2;
        std::map<int64_t, int64_t> level;  // load -> CUs at that load
        const int64_t a = nw / ncu, r = nw % ncu;
        if (r) level[3 * (a + 1)] += r;
        level[3 * a] += ncu - r;
        for (int64_t left = nn; left > 0;) {
            auto it = level.begin();
            const int64_t k = it->second < left ? it->second : left, at = it->first;
            if (k == it->second)
                level.erase(it);
            else
                it->second -= k;
            level[at + 2] += k;
            left -= k;
        }
        const int64_t t = level.rbegin()->first;
        if (best_t < 0 || t < best_t) {
            best_t = t;
            best = wide;
        }
    }
    return best;
}

extern "C" int wp_lf_wide_frames(int columns, int frames, int compute_units)
{
    if ((columns != LF_C && columns != LF_C_BIG) || frames < 0 || frames > 65535 || compute_units < 1 || compute_units > 4096)
        return fail(WP_ERR_INVALID, "wp_lf_wide_frames: bad argument");
    return lf_wide_frames(columns, frames, compute_units);
}

// Inverse passes of every run of frames.  The runs write disjoint sample ranges; every run after the first goes
// to a side stream of its own, so that all of them overlap the first run's pass.  A bus-fold workgroup is a serial
// chain over the sources; a run with too few tiles to give every CU one (the short runs at the end of a cover) would
// spend the whole chain's latency on a fraction of the chip, so when the per-job results are written anyway such a
// run takes the plain pass (one workgroup per job and tile) followed by the mix kernel over its range of the bus:
// the same fold in the same order, at the price of reading that range's results back.
static int conv_back_lf(wp_conv *c, int64_t limit, int nsrc, float *bus0, float *bus1, int64_t tl, const double *vol,
                        float *out_dev, int64_t out_stride, int64_t out_len, hipStream_t s)
{
    const wp_ir *ir = c->ir;
    bool forked = false, used[LF_MAX_SEG] = {false, false, false};
    // Chains of the later runs only pay when they find room on the chip beside the first run's: a CU holds two
    // bus-fold workgroups (registers), and the first run either fills both places of every CU for its whole
    // duration or is launched one per CU (launch_lf_inv: when that needs fewer rounds), which leaves a place free.
    const int ncu = device_cus();
    int64_t room = 2 * (int64_t)ncu;
    int wide0 = 0;  // frames of run 0 whose chains take the wide tiles
    // (the bus-only pass keeps the narrow chains, two per CU: on wide tiles it measured 2.11 against 1.72 ms)
    if (bus0 && out_dev && ir->lf.nseg > 0) {
        const LfSeg &g0 = ir->lf.seg[0];
        const int fr0 = lf_frames_run(g0, limit);
        if (g0.R == 256 && g0.base < tl && fr0 > 0) {
            if (c->wide_for != fr0) {
                c->wide_frames = lf_wide_frames(g0.C, fr0, ncu);
                c->wide_for = fr0;
            }
            wide0 = c->wide_frames;
        }
        const int64_t units = (int64_t)(g0.C / (g0.R > 128 ? 16 : 32)) * fr0 * 2;
        const int64_t one = (units + ncu - 1) / ncu, two = 2 * ((units + 2 * ncu - 1) / (2 * ncu));
        room = (g0.R >= 128 && one < two) ? ncu : (units >= 2 * ncu ? 0 : 2 * ncu - units);
        if (wide0 > 0) room = 0;  // a wide chain leaves no registers for another beside it
    }
    for (int i = ir->lf.nseg - 1; i >= 0; i--) {
        const LfSeg &g = ir->lf.seg[i];
        const int fr = lf_frames_run(g, limit);
        if (fr <= 0) continue;
        // a run that starts behind the end of the bus folds nothing: its results (if wanted) take the plain pass
        bool fold = bus0 && g.base < tl;
        if (!fold && !out_dev) continue;
        hipStream_t q = s;
        if (i > 0 && c->side[i]) {
            if (!forked) {
                HIPCHK(hipEventRecord(c->fork, s));
                forked = true;
            }
            HIPCHK(hipStreamWaitEvent(c->side[i], c->fork, 0));
            q = c->side[i];
            used[i] = true;
        }
        const int cols = g.R > 128 ? 16 : 32;
        const int64_t chains = (int64_t)(g.C / cols) * fr * 2;
        bool split = false;
        if (fold && out_dev && i > 0 && nsrc >= 8 && (int)c->vol_host.size() == nsrc && g_tune_lf_split != 1) {
            if (chains <= room && g_tune_lf_split != 2)
                room -= chains;
            else
                split = true;
        }
        if (split) fold = false;
        if (i == 0 && fold && wide0 > 0) {
            // the wide chains first ...
            HIPCHK(launch_lf_inv(g.R, g.C, lf_seg_A(c, i), g.nframes, 0, wide0, c->njobs, g.base, g.hop, ir->lf.mpad, nsrc, bus0, bus1, tl, vol,
                                 out_dev, out_stride, out_len, ir->lf_tab[i], true, s));
            // ... then the narrow ones of the remaining frames, behind them on the same stream (beside them on a
            // stream of their own the two kinds share the CUs from the start and the pass takes 2.63 instead of 2.52 ms)
            if (wide0 < fr)
                HIPCHK(launch_lf_inv(g.R, g.C, lf_seg_A(c, i), g.nframes, wide0, fr - wide0, c->njobs, g.base, g.hop, ir->lf.mpad, nsrc, bus0, bus1,
                                     tl, vol, out_dev, out_stride, out_len, ir->lf_tab[i], false, s));
            continue;
        }
        HIPCHK(launch_lf_inv(g.R, g.C, lf_seg_A(c, i), g.nframes, 0, fr, c->njobs, g.base, g.hop, ir->lf.mpad, nsrc, fold ? bus0 : nullptr,
                             fold ? bus1 : nullptr, tl, vol, out_dev, out_stride, out_len, ir->lf_tab[i], false, q));
        if (split) {
            int64_t hi = g.base + (int64_t)fr * g.hop;
            if (hi > tl) hi = tl;
            if (hi > out_len) hi = out_len;
            if (hi > g.base)
                if (int rc = lf_mix_range(c, i, g.base, hi, nsrc, bus0, bus1, out_dev, out_stride, q)) return rc;
        }
    }
    for (int i = 1; i < LF_MAX_SEG; i++)
        if (used[i]) {
            HIPCHK(hipEventRecord(c->join[i], c->side[i]));
            HIPCHK(hipStreamWaitEvent(s, c->join[i], 0));
        }
    return WP_OK;
}

static int conv_front(wp_conv *c, const float *in_dev, int64_t in_stride, int64_t in_len, int Tout, hipStream_t s)
{
    const wp_ir *ir = c->ir;
    const int B = ir->B, T = c->T;
    int Tin = (int)((in_len + B - 1) / B) + 1;
    if (Tin > Tout) Tin = Tout;  // later input blocks cannot reach the requested outputs
    if (in_len == 0) Tin = 0;
    if (int rc = prof_begin(c)) return rc;
    if (c->prof) HIPCHK(hipEventRecord(c->ev[0], s));
    if (Tin > 0) HIPCHK(launch_fft_fwd(B, false, in_dev, in_stride, in_len, c->Z, T, Tin, c->njobs, ir->tables, s));
    if (c->prof) HIPCHK(hipEventRecord(c->ev[1], s));
    if (Tin == 0)  // no input at all: the result is zeros
        HIPCHK(hipMemsetAsync(c->W, 0, sizeof(c32) * (size_t)c->njobs * T * B, s));
    else if (c->gather)
        HIPCHK(launch_mac_gather(c->Z, c->W, ir->Hp, ir->tables + table_wsplit_offset(B), c->job_irch, c->irch_all, B, T,
                                 Tout, Tin, ir->Pp, c->njobs, s));
    else
        HIPCHK(launch_mac(ir->PG, c->Z, c->W, ir->Hp, ir->tables + table_wsplit_offset(B), c->job_irch, c->irch_all, B, T,
                          Tout, Tin, ir->Pp, c->njobs, c->tchunk, s));
    if (c->prof) HIPCHK(hipEventRecord(c->ev[2], s));
    return WP_OK;
}

extern "C" int wp_conv_run_dev(wp_conv *c, const float *in_dev, int64_t in_stride, int64_t in_len, float *out_dev,
                               int64_t out_stride, const float *prev_dev, int64_t prev_stride, int64_t prev_len,
                               void *stream)
{
    if (int rc = need_device()) return rc;
    if (!c || !out_dev || (!in_dev && in_len > 0)) return fail(WP_ERR_INVALID, "wp_conv_run_dev: null argument");
    const wp_ir *ir = c->ir;
    if (int rc = check_in_len(ir, in_len, c->out_len != ir->L + ir->M)) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (int rc = plan_enter(c, s)) return rc;
    if (prev_dev && prev_len > c->out_len) prev_len = c->out_len;
    if (c->lf) {
        if (int rc = conv_front_lf(c, in_dev, in_stride, in_len, c->out_len, s)) return rc;
        if (int rc = conv_back_lf(c, c->out_len, 0, nullptr, nullptr, 0, nullptr, out_dev, out_stride, c->out_len, s)) return rc;
        if (prev_dev && prev_len > 0) HIPCHK(launch_add_prev(out_dev, out_stride, prev_dev, prev_stride, prev_len, c->njobs, s));
    } else {
        if (int rc = conv_front(c, in_dev, in_stride, in_len, c->T, s)) return rc;
        HIPCHK(launch_fft_inv(ir->B, c->W, c->T, c->njobs, out_dev, out_stride, c->out_len,
                              prev_len > 0 ? prev_dev : nullptr, prev_stride, prev_len, ir->tables, s));
    }
    if (c->prof) {
        HIPCHK(hipEventRecord(c->ev[3], s));
        prof_end(c);
    }
    return plan_leave(c, s);
}

static int conv_mix_common(wp_conv *c, const float *in_dev, int64_t in_stride, int64_t in_len, float *out_dev,
                           int64_t out_stride, float *const *tgt_dev, int64_t tl, const double *vol, hipStream_t s,
                           const char *who)
{
    if (!c || !tgt_dev || !tgt_dev[0] || !tgt_dev[1] || !vol || (!in_dev && in_len > 0))
        return fail(WP_ERR_INVALID, "%s: null argument", who);
    const wp_ir *ir = c->ir;
    if (c->njobs % 2) return fail(WP_ERR_INVALID, "%s: needs an even number of jobs (stereo sources)", who);
    if (tl < 0 || tl > ir->L + ir->M) return fail(WP_ERR_INVALID, "%s: target length %lld exceeds the result length %lld", who, (long long)tl, (long long)(ir->L + ir->M));
    if (int rc = check_in_len(ir, in_len)) return rc;
    if (tl == 0 && !out_dev) return WP_OK;
    const int nsrc = c->njobs / 2;
    if (int rc = plan_enter(c, s)) return rc;
    if (!c->vol_dev) HIPCHK(hipMalloc((void **)&c->vol_dev, sizeof(double) * nsrc));
    // a copy from pageable memory makes the host wait for the stream; repeated calls with the
    // same gains (one per bar) skip it and queue back to back
    if (c->vol_host.size() != (size_t)nsrc || memcmp(c->vol_host.data(), vol, sizeof(double) * nsrc) != 0) {
        c->vol_host.clear();  // stays empty if the copy fails
        HIPCHK(hipMemcpyAsync(c->vol_dev, vol, sizeof(double) * nsrc, hipMemcpyHostToDevice, s));
        c->vol_host.assign(vol, vol + nsrc);
    }
    if (c->lf) {
        // with per-job results every frame is needed; otherwise only those that reach the bus
        const int64_t limit = out_dev ? ir->L + ir->M : tl;
        if (int rc = conv_front_lf(c, in_dev, in_stride, in_len, limit, s)) return rc;
        if (int rc = conv_back_lf(c, limit, nsrc, tgt_dev[0], tgt_dev[1], tl, c->vol_dev, out_dev, out_stride, ir->L + ir->M, s))
            return rc;
    } else {
        // with per-job results every output block is needed; otherwise only those that reach the bus
        const int Tout = out_dev ? c->T : (int)((tl + ir->B - 1) / ir->B);
        if (int rc = conv_front(c, in_dev, in_stride, in_len, Tout, s)) return rc;
        HIPCHK(launch_fft_inv_mix(ir->B, c->W, c->T, Tout, nsrc, tgt_dev[0], tgt_dev[1], tl, c->vol_dev, out_dev, out_stride,
                                  ir->L + ir->M, ir->tables, s));
    }
    if (c->prof) {
        HIPCHK(hipEventRecord(c->ev[3], s));
        prof_end(c);
    }
    return plan_leave(c, s);
}

extern "C" int wp_conv_mix_run_dev(wp_conv *c, const float *in_dev, int64_t in_stride, int64_t in_len,
                                   float *const *tgt_dev, int64_t tl, const double *vol, void *stream)
{
    if (int rc = need_device()) return rc;
    return conv_mix_common(c, in_dev, in_stride, in_len, nullptr, 0, tgt_dev, tl, vol, (hipStream_t)stream,
                           "wp_conv_mix_run_dev");
}

extern "C" int wp_conv_run_mix_dev(wp_conv *c, const float *in_dev, int64_t in_stride, int64_t in_len, float *out_dev,
                                   int64_t out_stride, float *const *tgt_dev, int64_t tl, const double *vol,
                                   void *stream)
{
    if (int rc = need_device()) return rc;
    if (!out_dev) return fail(WP_ERR_INVALID, "wp_conv_run_mix_dev: null output");
    return conv_mix_common(c, in_dev, in_stride, in_len, out_dev, out_stride, tgt_dev, tl, vol, (hipStream_t)stream,
                           "wp_conv_run_mix_dev");
}

static int single_plan(wp_ir *ir, int ch)
{
    if (!ir->single) {
        const int32_t zero = 0;
        ir->single = wp_conv_create(ir, 1, &zero);
        if (!ir->single) return WP_ERR_HIP;
        ir->single_ch = 0;
    }
    ir->single->irch_all = ch;  // a launch argument of the MAC: nothing on the device changes between calls
    ir->single_ch = ch;
    return WP_OK;
}

// An input longer than L whose linear convolution runs past the reference's FFT length N:
// the reference returns the first L+M samples of the N-point CIRCULAR convolution, i.e.
// y[i] = lin[i] + lin[i+N].  in_len <= N and M <= N, so the linear result (in_len+M samples)
// wraps at most once.  Computed with a longer-result plan of the same partition spectra.
static int convolve_wrapped(wp_ir *ir, int ch, const float *in_dev, int64_t in_len, float *out_dev, hipStream_t s)
{
    const int64_t N = wp_next_pot((int32_t)(ir->L + ir->M));
    const int64_t lin_len = in_len + ir->M;
    if (!ir->wrap || ir->wrap_len < lin_len) {
        if (ir->wrap) wp_conv_destroy(ir->wrap);
        if (ir->wrap_lin) (void)hipFree(ir->wrap_lin);
        ir->wrap = nullptr;
        ir->wrap_lin = nullptr;
        ir->wrap_len = 0;
        const int64_t cap = N + ir->M;  // the longest linear result any legal input can have
        ir->wrap = conv_create(ir, 1, &ch, cap);
        if (!ir->wrap) return WP_ERR_HIP;
        ir->wrap_ch = ch;
        HIPCHK(hipMalloc((void **)&ir->wrap_lin, sizeof(float) * (size_t)(cap + 2)));
        ir->wrap_len = cap;
    }
    ir->wrap->irch_all = ch;
    ir->wrap_ch = ch;
    if (int rc = wp_conv_run_dev(ir->wrap, in_dev, 0, in_len, ir->wrap_lin, 0, nullptr, 0, 0, s)) return rc;
    HIPCHK(launch_wrap_add(ir->wrap_lin, lin_len, N, out_dev, ir->L + ir->M, s));
    return WP_OK;
}

extern "C" int wp_convolve_dev(wp_ir *ir, int ch, const float *in_dev, int64_t in_len, float *out_dev, void *stream)
{
    if (int rc = need_device()) return rc;
    if (!ir || !out_dev || (!in_dev && in_len > 0)) return fail(WP_ERR_INVALID, "wp_convolve_dev: null argument");
    if (ch < 0 || ch >= ir->nch) return fail(WP_ERR_INVALID, "wp_convolve_dev: IR channel %d of %d", ch, ir->nch);
    if (int rc = check_in_len(ir, in_len, true)) return rc;
    if (input_wraps(ir, in_len)) return convolve_wrapped(ir, ch, in_dev, in_len, out_dev, (hipStream_t)stream);
    if (int rc = single_plan(ir, ch)) return rc;
    return wp_conv_run_dev(ir->single, in_dev, 0, in_len, out_dev, 0, nullptr, 0, 0, stream);
}

// wp_convolve_dev with the block spectra in the library's static device workspace: no
// allocation of any kind happens in this call (callers under stream capture, or harnesses that
// forbid hipMalloc).  Same kernels, same geometry, hence the same bits as wp_convolve_dev.
extern "C" int wp_convolve_dev_noalloc(wp_ir *ir, int ch, const float *in_dev, int64_t in_len, float *out_dev,
                                       void *stream)
{
    if (int rc = need_device()) return rc;
    if (!ir || !out_dev || (!in_dev && in_len > 0)) return fail(WP_ERR_INVALID, "wp_convolve_dev_noalloc: null argument");
    if (ch < 0 || ch >= ir->nch) return fail(WP_ERR_INVALID, "wp_convolve_dev_noalloc: IR channel %d of %d", ch, ir->nch);
    if (int rc = check_in_len(ir, in_len)) return rc;  // inputs that wrap in the reference's FFT: wp_convolve_dev
    wp_conv plan;
    plan_geometry(&plan, ir, 1, ir->L + ir->M);  // always the partitioned form (plan.lf stays false)
    const size_t sbytes = sizeof(c32) * (size_t)plan.T * ir->B;
    if (2 * sbytes > (size_t)WP_NOALLOC_WORKSPACE_BYTES)
        return fail(WP_ERR_RANGE,
                    "wp_convolve_dev_noalloc: L+M = %lld needs %zu bytes of block spectra, the static workspace holds %lld "
                    "(results up to %lld samples); use wp_convolve_dev",
                    (long long)(ir->L + ir->M), 2 * sbytes, (long long)WP_NOALLOC_WORKSPACE_BYTES,
                    (long long)(WP_NOALLOC_WORKSPACE_BYTES / (2 * sizeof(c32)) - ir->B + 1));
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    if (dev != ir->dev) return fail(WP_ERR_INVALID, "wp_convolve_dev_noalloc: the IR handle belongs to device %d, device %d is current", ir->dev, dev);
    void *ws = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (int rc = static_ws_locked(dev, &ws)) return rc;
    }
    // Z and W are rewritten from scratch by every call (each block the inverse transform reads
    // has been stored by this call's MAC, each block the MAC reads by this call's forward
    // transform), so nothing of the previous call's contents needs clearing.
    plan.Z = reinterpret_cast<c32 *>(ws);
    plan.W = reinterpret_cast<c32 *>(static_cast<unsigned char *>(ws) + WP_NOALLOC_WORKSPACE_BYTES / 2);
    plan.irch_all = ch;
    hipStream_t s = (hipStream_t)stream;
    if (int rc = conv_front(&plan, in_dev, 0, in_len, plan.T, s)) return rc;
    HIPCHK(launch_fft_inv(ir->B, plan.W, plan.T, 1, out_dev, 0, plan.out_len, nullptr, 0, 0, ir->tables, s));
    return WP_OK;
}

// The calling thread's stream for the host-pointer entry points (one per device): their transfers and
// kernels are queued on it without a host round trip in between and awaited once, at the end of the call.
// Used when every host array of the call is page-locked (HostCall below): each transfer is then a DMA
// straight from or to the caller's array.  The stream is an ordinary (blocking) one, so the calls stay
// ordered with work a caller queued on the null stream.
static int host_stream(hipStream_t *out)
{
    static thread_local std::map<int, hipStream_t> streams;  // released with the process
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    auto it = streams.find(dev);
    if (it == streams.end()) {
        hipStream_t s = nullptr;
        HIPCHK(hipStreamCreate(&s));
        it = streams.emplace(dev, s).first;
    }
    *out = it->second;
    return WP_OK;
}

// page-locked host memory (wp_host_alloc, wp_host_pin, or locked by the caller's own means)?
static bool host_is_pinned(const void *p)
{
    if (!p) return false;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();  // ordinary pageable memory is unknown to the runtime
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

// A host-pointer call whose arrays are ALL page-locked queues its transfers and kernels on the thread's
// stream and waits once.  With a pageable array among them it keeps to blocking copies on the null
// stream: the runtime stages those through its own bounce buffers, whereas an asynchronous copy from or
// to pageable memory locks the pages for the occasion (measured: a stereo bar's results into fresh
// pageable arrays took 305 us that way against 172 us blocking).
struct HostCall {
    hipStream_t s = nullptr;  // null stream: blocking copies
    bool async = false;
    int begin(std::initializer_list<const void *> ptrs)
    {
        for (const void *p : ptrs)
            if (!host_is_pinned(p)) return WP_OK;
        if (int rc = host_stream(&s)) return rc;
        async = true;
        return WP_OK;
    }
    hipError_t copy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind) const
    {
        return async ? hipMemcpyAsync(dst, src, bytes, kind, s) : hipMemcpy(dst, src, bytes, kind);
    }
    hipError_t end() const { return async ? hipStreamSynchronize(s) : hipSuccess; }
};

extern "C" void *wp_host_alloc(size_t bytes)
{
    if (need_device()) return nullptr;
    void *p = nullptr;
    const hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        fail(WP_ERR_HIP, "wp_host_alloc: %zu bytes of page-locked memory: %s", bytes, hipGetErrorString(e));
        return nullptr;
    }
    return p;
}

extern "C" void wp_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

extern "C" int wp_host_pin(void *p, size_t bytes)
{
    if (int rc = need_device()) return rc;
    if (!p || !bytes) return fail(WP_ERR_INVALID, "wp_host_pin: null range");
    const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(WP_ERR_HIP, "wp_host_pin: %s", hipGetErrorString(e));
    }
    return WP_OK;
}

extern "C" int wp_host_unpin(void *p)
{
    if (!p) return WP_OK;
    const hipError_t e = hipHostUnregister(p);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(WP_ERR_HIP, "wp_host_unpin: %s", hipGetErrorString(e));
    }
    return WP_OK;
}

extern "C" int wp_convolve(wp_ir *ir, int ch, const float *in, int64_t in_len, float *out)
{
    if (int rc = need_device()) return rc;
    if (!ir || !out || (!in && in_len > 0)) return fail(WP_ERR_INVALID, "wp_convolve: null argument");
    if (ch < 0 || ch >= ir->nch) return fail(WP_ERR_INVALID, "wp_convolve: IR channel %d of %d", ch, ir->nch);
    if (int rc = check_in_len(ir, in_len, true)) return rc;
    const int64_t N = wp_next_pot((int32_t)(ir->L + ir->M));
    if (!ir->stage_in) {
        HIPCHK(hipMalloc((void **)&ir->stage_in, sizeof(float) * (size_t)(N + 2)));
        HIPCHK(hipMalloc((void **)&ir->stage_out, sizeof(float) * (size_t)(ir->L + ir->M + 2)));
    }
    HostCall hc;
    if (int rc = hc.begin({in_len > 0 ? in : out, out})) return rc;
    if (in_len > 0) HIPCHK(hc.copy(ir->stage_in, in, sizeof(float) * (size_t)in_len, hipMemcpyHostToDevice));
    if (int rc = wp_convolve_dev(ir, ch, ir->stage_in, in_len, ir->stage_out, hc.s)) return rc;
    HIPCHK(hc.copy(out, ir->stage_out, sizeof(float) * (size_t)(ir->L + ir->M), hipMemcpyDeviceToHost));
    HIPCHK(hc.end());
    return WP_OK;
}

// ---------------------------------------------------------------------------
// lib/convolve.js:541-551 and the reference-layout kernel object (:553-562)
struct DevTmp {  // device allocations of one host-pointer call, freed on every return path
    std::vector<void *> p;
    ~DevTmp()
    {
        for (void *q : p) (void)hipFree(q);
    }
    hipError_t get(void **out, size_t bytes)
    {
        const hipError_t e = hipMalloc(out, bytes ? bytes : 4);
        if (e == hipSuccess) p.push_back(*out);
        return e;
    }
};

extern "C" int wp_direct_convolution(const float *input, int64_t n, const float *kernel, int64_t m, float *out)
{
    if (int rc = need_device()) return rc;
    if (n < 0 || m < 0 || (n > 0 && !input) || (m > 0 && !kernel)) return fail(WP_ERR_INVALID, "wp_direct_convolution: bad argument");
    const int64_t len = n + m - 1;
    if (len <= 0) return WP_OK;
    if (!out) return fail(WP_ERR_INVALID, "wp_direct_convolution: null output");
    DevTmp tmp;
    float *dx = nullptr, *dh = nullptr, *dy = nullptr;
    HIPCHK(tmp.get((void **)&dx, sizeof(float) * (size_t)n));
    HIPCHK(tmp.get((void **)&dh, sizeof(float) * (size_t)m));
    HIPCHK(tmp.get((void **)&dy, sizeof(float) * (size_t)len));
    if (n > 0) HIPCHK(hipMemcpy(dx, input, sizeof(float) * (size_t)n, hipMemcpyHostToDevice));
    if (m > 0) HIPCHK(hipMemcpy(dh, kernel, sizeof(float) * (size_t)m, hipMemcpyHostToDevice));
    HIPCHK(launch_direct_conv(dx, n, dh, m, dy, nullptr));
    HIPCHK(hipMemcpy(out, dy, sizeof(float) * (size_t)len, hipMemcpyDeviceToHost));
    return WP_OK;
}

static int big_fft_host(const float *src, int64_t n_src, int64_t n_real, int64_t N, int inverse, float *dst, int64_t n_out,
                        const char *who)
{
    if (N < 2 || (N & (N - 1)) != 0) return fail(WP_ERR_FFT_SIZE, "FFT size must be a power of two and bigger than 1");
    if (N > ((int64_t)1 << 27)) return fail(WP_ERR_UNSUPPORTED, "%s: spectra of more than 2^27 points are not implemented", who);
    DevTmp tmp;
    float *dsrc = nullptr, *ddst = nullptr;
    void *w0 = nullptr, *w1 = nullptr;
    HIPCHK(tmp.get((void **)&dsrc, sizeof(float) * (size_t)n_src));
    HIPCHK(tmp.get((void **)&ddst, sizeof(float) * (size_t)n_out));
    HIPCHK(tmp.get(&w0, 16 * (size_t)N));
    HIPCHK(tmp.get(&w1, 16 * (size_t)N));
    if (n_src > 0) HIPCHK(hipMemcpy(dsrc, src, sizeof(float) * (size_t)n_src, hipMemcpyHostToDevice));
    HIPCHK(launch_big_fft(dsrc, n_real, N, inverse, w0, w1, ddst, inverse ? n_out : N, nullptr));
    HIPCHK(hipMemcpy(dst, ddst, sizeof(float) * (size_t)n_out, hipMemcpyDeviceToHost));
    return WP_OK;
}

extern "C" int wp_kernel_spectrum(const float *kernel, int64_t M, int64_t L, float *spectrum_out)
{
    if (int rc = need_device()) return rc;
    if (!kernel || !spectrum_out) return fail(WP_ERR_INVALID, "wp_kernel_spectrum: null argument");
    if (int rc = check_ir_args(1, M, L)) return rc;
    const int64_t N = wp_next_pot((int32_t)(L + M));
    return big_fft_host(kernel, M, M, N, 0, spectrum_out, 2 * N, "wp_kernel_spectrum");
}

extern "C" int wp_kernel_from_spectrum(const float *spectrum, int64_t N, int64_t M, float *kernel_out)
{
    if (int rc = need_device()) return rc;
    if (!spectrum || !kernel_out) return fail(WP_ERR_INVALID, "wp_kernel_from_spectrum: null argument");
    if (M < 1 || M > N) return fail(WP_ERR_INVALID, "wp_kernel_from_spectrum: kernel length %lld does not fit %lld points", (long long)M, (long long)N);
    return big_fft_host(spectrum, 2 * N, 0, N, 1, kernel_out, M, "wp_kernel_from_spectrum");
}

// ---------------------------------------------------------------------------
extern "C" wp_reverb *wp_reverb_create(wp_ir *ir, int njobs, const int32_t *job_ch, int64_t offset)
{
    if (need_device()) return nullptr;
    if (!ir) {
        fail(WP_ERR_INVALID, "wp_reverb_create: null IR");
        return nullptr;
    }
    wp_conv *c = conv_create(ir, njobs, job_ch, ir->L + ir->M);
    if (!c) return nullptr;
    wp_reverb *r = new wp_reverb();
    r->conv = c;
    r->njobs = njobs;
    r->offset = offset;
    r->stride = (ir->L + ir->M + 3) & ~(int64_t)3;
    r->buf[0] = r->buf[1] = nullptr;
    r->cur = 0;
    r->prev = nullptr;
    r->prev_stride = 0;
    r->prev_off = r->prev_len = 0;
    r->stage_in = nullptr;
    r->stage_in_len = 0;
    const size_t bytes = sizeof(float) * (size_t)njobs * r->stride;
    hipError_t e = hipMalloc((void **)&r->buf[0], bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&r->buf[1], bytes);
    if (e != hipSuccess) {
        fail(WP_ERR_HIP, "wp_reverb_create: hipMalloc failed: %s", hipGetErrorString(e));
        wp_reverb_destroy(r);
        return nullptr;
    }
    return r;
}

extern "C" void wp_reverb_destroy(wp_reverb *r)
{
    if (!r) return;
    if (r->buf[0]) (void)hipFree(r->buf[0]);
    if (r->buf[1]) (void)hipFree(r->buf[1]);
    if (r->stage_in) (void)hipFree(r->stage_in);
    if (r->conv) wp_conv_destroy(r->conv);
    delete r;
}

extern "C" int wp_reverb_block_to(wp_reverb *r, const float *in_dev, int64_t in_stride, int64_t in_len,
                                  float *curr_dev, int64_t curr_stride, wp_reverb_view *v, void *stream)
{
    if (int rc = need_device()) return rc;
    if (!r || !v || !curr_dev) return fail(WP_ERR_INVALID, "wp_reverb_block_to: null argument");
    const wp_ir *ir = r->conv->ir;
    if (r->njobs > 1 && curr_stride < ir->L + ir->M) return fail(WP_ERR_INVALID, "wp_reverb_block_to: jobs would overlap in the result buffer");
    if (int rc = wp_conv_run_dev(r->conv, in_dev, in_stride, in_len, curr_dev, curr_stride,
                                 r->prev_len > 0 ? r->prev : nullptr, r->prev_stride, r->prev_len, stream))
        return rc;
    wp_reverb_view_for(ir->L, ir->M, r->offset, in_len, v);
    r->prev = curr_dev + v->prev_off;
    r->prev_stride = curr_stride;
    r->prev_off = v->prev_off;
    r->prev_len = v->prev_len;
    return WP_OK;
}

extern "C" int wp_reverb_block_dev(wp_reverb *r, const float *in_dev, int64_t in_stride, int64_t in_len,
                                   float **curr_dev, int64_t *curr_stride, wp_reverb_view *v, void *stream)
{
    if (int rc = need_device()) return rc;
    if (!r || !v) return fail(WP_ERR_INVALID, "wp_reverb_block_dev: null argument");
    const int nxt = r->cur ^ 1;
    if (r->prev_len > 0 && r->prev >= r->buf[nxt] && r->prev < r->buf[nxt] + (size_t)r->njobs * r->stride)
        return fail(WP_ERR_INVALID, "wp_reverb_block_dev: carried tail lives in the buffer about to be overwritten");
    if (int rc = wp_reverb_block_to(r, in_dev, in_stride, in_len, r->buf[nxt], r->stride, v, stream)) return rc;
    r->cur = nxt;
    if (curr_dev) *curr_dev = r->buf[nxt];
    if (curr_stride) *curr_stride = r->stride;
    return WP_OK;
}

extern "C" int wp_reverb_block(wp_reverb *r, const float *const *in, int64_t in_len, float *const *out,
                               wp_reverb_view *v)
{
    if (int rc = need_device()) return rc;
    if (!r || !in || !out || !v) return fail(WP_ERR_INVALID, "wp_reverb_block: null argument");
    const wp_ir *ir = r->conv->ir;
    if (int rc = check_in_len(ir, in_len)) return rc;
    const int64_t istride = (in_len + 3) & ~(int64_t)3;
    if (r->stage_in_len < istride) {
        if (r->stage_in) (void)hipFree(r->stage_in);
        r->stage_in = nullptr;
        r->stage_in_len = 0;
        HIPCHK(hipMalloc((void **)&r->stage_in, sizeof(float) * (size_t)r->njobs * (istride > 0 ? istride : 4)));
        r->stage_in_len = istride;
    }
    HostCall hc;
    {
        bool pinned = true;
        for (int j = 0; j < r->njobs && pinned; j++) pinned = host_is_pinned(out[j]) && (in_len == 0 || host_is_pinned(in[j]));
        if (pinned)
            if (int rc = hc.begin({})) return rc;
    }
    for (int j = 0; j < r->njobs; j++)
        if (in_len > 0)
            HIPCHK(hc.copy(r->stage_in + (size_t)j * istride, in[j], sizeof(float) * (size_t)in_len, hipMemcpyHostToDevice));
    float *curr = nullptr;
    int64_t cs = 0;
    if (int rc = wp_reverb_block_dev(r, r->stage_in, istride, in_len, &curr, &cs, v, hc.s)) return rc;
    for (int j = 0; j < r->njobs; j++)
        HIPCHK(hc.copy(out[j], curr + (size_t)j * cs, sizeof(float) * (size_t)(ir->L + ir->M), hipMemcpyDeviceToHost));
    HIPCHK(hc.end());
    return WP_OK;
}

extern "C" int wp_reverb_set_tail(wp_reverb *r, const float *const *prev, int64_t prev_len)
{
    if (int rc = need_device()) return rc;
    if (!r || (prev_len > 0 && !prev)) return fail(WP_ERR_INVALID, "wp_reverb_set_tail: null argument");
    const wp_ir *ir = r->conv->ir;
    if (prev_len < 0) prev_len = 0;
    if (prev_len > ir->L + ir->M) prev_len = ir->L + ir->M;  // adds past curr.length are no-ops on a typed array
    for (int j = 0; j < r->njobs && prev_len > 0; j++)
        HIPCHK(hipMemcpy(r->buf[r->cur] + (size_t)j * r->stride, prev[j], sizeof(float) * (size_t)prev_len, hipMemcpyHostToDevice));
    r->prev = r->buf[r->cur];
    r->prev_stride = r->stride;
    r->prev_off = 0;
    r->prev_len = prev_len;
    return WP_OK;
}

extern "C" int wp_reverb_set_tail_dev(wp_reverb *r, const float *prev_dev, int64_t prev_stride, int64_t prev_len)
{
    if (int rc = need_device()) return rc;
    if (!r || (prev_len > 0 && !prev_dev)) return fail(WP_ERR_INVALID, "wp_reverb_set_tail_dev: null argument");
    const wp_ir *ir = r->conv->ir;
    if (prev_len < 0) prev_len = 0;
    if (prev_len > ir->L + ir->M) prev_len = ir->L + ir->M;
    r->prev = prev_dev;
    r->prev_stride = prev_stride;
    r->prev_off = 0;
    r->prev_len = prev_len;
    return WP_OK;
}

extern "C" int64_t wp_reverb_tail_len(const wp_reverb *r) { return r ? r->prev_len : WP_ERR_INVALID; }

extern "C" int wp_reverb_get_tail(wp_reverb *r, float *const *prev_out)
{
    if (int rc = need_device()) return rc;
    if (!r || !prev_out) return fail(WP_ERR_INVALID, "wp_reverb_get_tail: null argument");
    for (int j = 0; j < r->njobs && r->prev_len > 0; j++)
        HIPCHK(hipMemcpy(prev_out[j], r->prev + (size_t)j * r->prev_stride, sizeof(float) * (size_t)r->prev_len, hipMemcpyDeviceToHost));
    return WP_OK;
}

// ---------------------------------------------------------------------------
// mix
// Per calling thread and device: the wp_src copy + mix_desc array of the call in flight.
// Threads never share it (worker threads mix concurrently); within a thread a call on a
// different stream first waits for the previous call's stream, because its kernels may
// still be reading the area.
struct Scratch {
    void *p = nullptr;     // device: wp_src copy + mix_desc array
    void *hpin = nullptr;  // pinned host mirror of the wp_src part (a copy from pinned memory does not stall the host)
    size_t cap = 0;
    hipStream_t last = nullptr;
    bool used = false;
    hipEvent_t ev = nullptr;  // recorded after the last upload from hpin
    bool ev_pending = false;
};

static int mix_scratch(int dev, size_t bytes, hipStream_t s, Scratch **out)
{
    static thread_local std::map<int, Scratch> scratch;  // released with the process
    Scratch &sc = scratch[dev];
    if (sc.used && sc.last != s) HIPCHK(hipStreamSynchronize(sc.last));
    if (sc.cap < bytes) {
        if (sc.p) {
            HIPCHK(hipStreamSynchronize(s));
            HIPCHK(hipFree(sc.p));
            HIPCHK(hipHostFree(sc.hpin));
            sc.p = sc.hpin = nullptr;
            sc.cap = 0;
            sc.ev_pending = false;
        }
        size_t cap = bytes < (1 << 20) ? (1 << 20) : bytes;
        HIPCHK(hipMalloc(&sc.p, cap));
        HIPCHK(hipHostMalloc(&sc.hpin, cap, hipHostMallocDefault));
        if (!sc.ev) HIPCHK(hipEventCreateWithFlags(&sc.ev, hipEventDisableTiming));
        sc.cap = cap;
    }
    sc.last = s;
    sc.used = true;
    *out = &sc;
    return WP_OK;
}

// sources [0, nsrc) none of which overlaps the target: one pass of the tiled kernel
static int mix_dev_run(float *const *tgt_dev, int tch, int64_t tl, const wp_src *srcs, int64_t nsrc, int srcs_on_device, hipStream_t s)
{
    if (nsrc == 0) return WP_OK;
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    const size_t src_bytes = (sizeof(wp_src) * (size_t)nsrc + 255) & ~(size_t)255;
    const size_t bytes = src_bytes + mix_scratch_bytes(nsrc);
    Scratch *sc = nullptr;
    if (int rc = mix_scratch(dev, bytes, s, &sc)) return rc;
    void *scr = sc->p;
    const wp_src *sd = srcs;
    if (!srcs_on_device) {
        // through the pinned mirror: the previous upload must have left it before it is rewritten
        if (sc->ev_pending) HIPCHK(hipEventSynchronize(sc->ev));
        memcpy(sc->hpin, srcs, sizeof(wp_src) * (size_t)nsrc);
        HIPCHK(hipMemcpyAsync(scr, sc->hpin, sizeof(wp_src) * (size_t)nsrc, hipMemcpyHostToDevice, s));
        HIPCHK(hipEventRecord(sc->ev, s));
        sc->ev_pending = true;
        sd = reinterpret_cast<const wp_src *>(scr);
    }
    HIPCHK(launch_mix(tgt_dev[0], tch >= 2 ? tgt_dev[1] : nullptr, tch, tl, sd, (char *)scr + src_bytes, nsrc, s));
    return WP_OK;
}

// does a channel of the source share memory with the part of the target the call updates?
static bool src_overlaps_target(const wp_src &S, float *const *tgt, int tch, int64_t tl)
{
    if (S.nch != 1 && S.nch != 2) return false;
    for (int c = 0; c < S.nch; c++)
        for (int t = 0; t < tch && t < 2; t++)
            if (S.ch[c] && S.rl > 0 && S.ch[c] < tgt[t] + tl && tgt[t] < S.ch[c] + S.rl) return true;
    return false;
}

extern "C" int wp_mix_dev(float *const *tgt_dev, int tch, int64_t tl, const wp_src *srcs, int64_t nsrc,
                          int srcs_on_device, void *stream)
{
    if (int rc = need_device()) return rc;
    if (!tgt_dev || tl < 0 || nsrc < 0 || (nsrc > 0 && !srcs)) return fail(WP_ERR_INVALID, "wp_mix_dev: bad argument");
    if (tch < 1 || !tgt_dev[0] || (tch >= 2 && !tgt_dev[1])) return fail(WP_ERR_INVALID, "wp_mix_dev: null target channel");
    if (nsrc == 0 || tl == 0) return WP_OK;
    hipStream_t s = (hipStream_t)stream;
    if (srcs_on_device) return mix_dev_run(tgt_dev, tch, tl, srcs, nsrc, 1, s);  // (a table on the device is taken as free of overlaps)
    // The tiled kernel updates the target while other lanes read the sources, so a source that overlaps the
    // target (mixBuffers(t, a, t), a feedback mix) cannot ride in it: the reference reads it sample by sample
    // as the loop goes (src/mix-buffers.js:4, 20-67).  The source list is cut at such sources: the runs
    // between them take the tiled kernel, each of them the frame-by-frame one, all in source order.
    int64_t from = 0;
    for (int64_t i = 0; i < nsrc; i++) {
        if (!src_overlaps_target(srcs[i], tgt_dev, tch, tl)) continue;
        if (int rc = mix_dev_run(tgt_dev, tch, tl, srcs + from, i - from, 0, s)) return rc;
        if (tch <= 2) HIPCHK(launch_mix_serial(tgt_dev[0], tch >= 2 ? tgt_dev[1] : nullptr, tch, tl, srcs[i], s));
        from = i + 1;
    }
    return mix_dev_run(tgt_dev, tch, tl, srcs + from, nsrc - from, 0, s);
}

extern "C" int wp_mix(float *const *tgt, int tch, int64_t tl, const wp_src *srcs, int64_t nsrc)
{
    if (int rc = need_device()) return rc;
    if (!tgt || tl < 0 || nsrc < 0 || (nsrc > 0 && !srcs)) return fail(WP_ERR_INVALID, "wp_mix: bad argument");
    if (tch < 1 || !tgt[0] || (tch >= 2 && !tgt[1])) return fail(WP_ERR_INVALID, "wp_mix: null target channel");
    if (nsrc == 0 || tl == 0 || tch > 2) return WP_OK;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    // Host arrays that share memory must share it on the device too (a source that is the target itself, or a
    // view into it, is read as the loop updates it: src/mix-buffers.js:4).  Every target channel starts an
    // arena = a range of host memory staged as one piece; a source channel that overlaps an arena joins it
    // (the arena grows to hold both).  Everything else is staged once per distinct array.
    struct Arena {
        const char *lo, *hi;
        size_t off;
    };
    std::vector<Arena> arenas;
    for (int c = 0; c < tch; c++)
        arenas.push_back({reinterpret_cast<const char *>(tgt[c]), reinterpret_cast<const char *>(tgt[c] + tl), 0});
    for (int64_t i = 0; i < nsrc; i++) {
        const wp_src &S = srcs[i];
        if (S.nch != 1 && S.nch != 2) continue;
        if (S.rl < 0) return fail(WP_ERR_INVALID, "wp_mix: source %lld has negative length", (long long)i);
        for (int c = 0; c < S.nch; c++)
            if (!S.ch[c] && S.rl > 0) return fail(WP_ERR_INVALID, "wp_mix: source %lld channel %d is null", (long long)i, c);
    }
    for (bool grown = true; grown;) {  // until no source channel straddles an arena's edge
        grown = false;
        for (int64_t i = 0; i < nsrc; i++) {
            const wp_src &S = srcs[i];
            if (S.nch != 1 && S.nch != 2) continue;
            for (int c = 0; c < S.nch; c++) {
                const char *lo = reinterpret_cast<const char *>(S.ch[c]), *hi = reinterpret_cast<const char *>(S.ch[c] + S.rl);
                for (Arena &A : arenas)
                    if (lo < A.hi && A.lo < hi && (lo < A.lo || hi > A.hi)) {
                        if (lo < A.lo) A.lo = lo;
                        if (hi > A.hi) A.hi = hi;
                        grown = true;
                    }
            }
        }
        for (size_t a = 0; a + 1 < arenas.size(); a++)  // two arenas that have come to overlap become one
            for (size_t b = a + 1; b < arenas.size(); b++)
                if (arenas[a].lo < arenas[b].hi && arenas[b].lo < arenas[a].hi) {
                    if (arenas[b].lo < arenas[a].lo) arenas[a].lo = arenas[b].lo;
                    if (arenas[b].hi > arenas[a].hi) arenas[a].hi = arenas[b].hi;
                    arenas.erase(arenas.begin() + (long)b);
                    b--;
                    grown = true;
                }
    }
    size_t total = 0;
    for (Arena &A : arenas) {
        A.off = total + (reinterpret_cast<uintptr_t>(A.lo) & 255);  // keeps the arrays' alignment within 256 bytes
        total += al((size_t)(A.hi - A.lo) + 256);
    }
    auto in_arena = [&](const void *p, size_t bytes) -> const Arena * {
        const char *lo = reinterpret_cast<const char *>(p);
        for (const Arena &A : arenas)
            if (bytes > 0 && lo >= A.lo && lo + bytes <= A.hi) return &A;
        return nullptr;
    };
    std::map<std::pair<const void *, size_t>, size_t> where;  // (host ptr, bytes) -> offset, arrays outside the arenas
    auto place = [&](const void *p, size_t bytes) {
        if (in_arena(p, bytes)) return;
        auto key = std::make_pair(p, bytes);
        if (where.find(key) == where.end()) {
            where[key] = total;
            total += al(bytes);
        }
    };
    for (int64_t i = 0; i < nsrc; i++) {
        const wp_src &S = srcs[i];
        if (S.nch != 1 && S.nch != 2) continue;
        for (int c = 0; c < S.nch; c++) place(S.ch[c], sizeof(float) * (size_t)S.rl);
        if (S.vol_vec) place(S.vol_vec, sizeof(double) * (size_t)tl);
    }
    // grow-only staging area, one per calling thread and device: repeated mixBuffers calls (one per bar in
    // the reference's render loop) do not pay for hipMalloc/hipFree each time
    struct Stage {
        char *p = nullptr;
        size_t cap = 0;
    };
    static thread_local std::map<int, Stage> stages;  // released with the process (the HIP runtime may be gone before a destructor runs)
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    Stage &stage = stages[dev];
    if (stage.cap < total) {
        if (stage.p) (void)hipFree(stage.p);
        stage.p = nullptr;
        stage.cap = 0;
        const size_t cap = total < (1 << 20) ? (1 << 20) : total + total / 4;
        HIPCHK(hipMalloc((void **)&stage.p, cap));
        stage.cap = cap;
    }
    char *d = stage.p;
    HostCall hc;
    {
        bool pinned = true;
        for (const Arena &A : arenas) pinned = pinned && host_is_pinned(A.lo);
        for (auto &kv : where) pinned = pinned && (kv.first.second == 0 || host_is_pinned(kv.first.first));
        if (pinned)
            if (int rc = hc.begin({})) return rc;
    }
    for (const Arena &A : arenas) {
        hipError_t e = hc.copy(d + A.off, A.lo, (size_t)(A.hi - A.lo), hipMemcpyHostToDevice);
        if (e != hipSuccess) return fail(WP_ERR_HIP, "hipMemcpy(H2D) failed: %s", hipGetErrorString(e));
    }
    for (auto &kv : where) {
        if (kv.first.second == 0) continue;
        hipError_t e = hc.copy(d + kv.second, kv.first.first, kv.first.second, hipMemcpyHostToDevice);
        if (e != hipSuccess) return fail(WP_ERR_HIP, "hipMemcpy(H2D) failed: %s", hipGetErrorString(e));
    }
    auto dev_of = [&](const void *p, size_t bytes) -> char * {
        if (const Arena *A = in_arena(p, bytes)) return d + A->off + (reinterpret_cast<const char *>(p) - A->lo);
        return d + where[std::make_pair(p, bytes)];
    };
    std::vector<wp_src> ds(srcs, srcs + nsrc);
    for (int64_t i = 0; i < nsrc; i++) {
        wp_src &S = ds[i];
        if (S.nch != 1 && S.nch != 2) continue;
        for (int c = 0; c < S.nch; c++) S.ch[c] = reinterpret_cast<const float *>(dev_of(srcs[i].ch[c], sizeof(float) * (size_t)S.rl));
        if (S.vol_vec) S.vol_vec = reinterpret_cast<const double *>(dev_of(srcs[i].vol_vec, sizeof(double) * (size_t)tl));
    }
    float *td[2] = {nullptr, nullptr};
    for (int c = 0; c < tch; c++) td[c] = reinterpret_cast<float *>(dev_of(tgt[c], sizeof(float) * (size_t)tl));
    int rc = wp_mix_dev(td, tch, tl, ds.data(), nsrc, 0, hc.s);
    if (rc == WP_OK) {
        for (int c = 0; c < tch && rc == WP_OK; c++) {
            hipError_t e = hc.copy(tgt[c], td[c], sizeof(float) * (size_t)tl, hipMemcpyDeviceToHost);
            if (e != hipSuccess) rc = fail(WP_ERR_HIP, "hipMemcpy(D2H) failed: %s", hipGetErrorString(e));
        }
    }
    // awaited on every path: the staging area and the caller's arrays are free for reuse when the call returns
    const hipError_t es = hc.end();
    if (rc == WP_OK && es != hipSuccess) rc = fail(WP_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(es));
    return rc;
}

// ---------------------------------------------------------------------------
// device-resident sample buffers
//
// The reference's sample buffers are arrays of Float32Array channels over
// SharedArrayBuffers that every worker can see (src/buffer-service.js,
// src/mix-worker-thread.js:14-16,23-24).  A wp_buf is the same thing in HBM: nch
// channels of `frames` floats, visible to every thread of the process through a
// numeric id, reference-counted.  Allocations are recycled through a small
// size-keyed pool so that a fresh result buffer per block (what go() returns in
// the reference, lib/convolve.js:575,582) does not cost a hipMalloc per block.
struct wp_buf {
    uint64_t id;
    int dev;
    int nch;
    int64_t frames;
    int64_t stride;  // floats between channels (multiple of 64: every channel 256-byte aligned)
    float *p;
    size_t bytes;
    int refs;
    bool carved;  // part of a slab: goes back to the pool whatever its cap, never to hipFree
};

static std::mutex g_buf_mu;
static std::map<uint64_t, wp_buf *> g_bufs;
static uint64_t g_next_buf_id = 1;
static std::multimap<std::pair<int, size_t>, void *> g_pool;  // (device, bytes) -> free allocation
static size_t g_pool_bytes = 0;

// Buffers of up to SLAB_MAX bytes that the pool cannot serve are carved from slabs of SLAB_BYTES (one hipMalloc per
// slab, kept until the process ends).  A block loop that asks for a fresh result buffer per block while its garbage
// collector has not yet given the earlier ones back then pays no hipMalloc per block (about 40 us each).
struct Slab {
    char *p = nullptr;
    size_t used = 0, size = 0;
};
static std::map<int, Slab> g_slab;  // per device, under g_buf_mu
static std::set<void *> g_carved;   // blocks carved from slabs (live or pooled)
constexpr size_t SLAB_BYTES = (size_t)64 << 20, SLAB_MAX = (size_t)8 << 20;

static size_t pool_limit()
{
    std::lock_guard<std::mutex> lk(g_mu);
    return (size_t)g_tune_pool_mb.load() << 20;
}

extern "C" wp_buf *wp_buf_create(int nch, int64_t frames)
{
    if (need_device()) return nullptr;
    if (nch < 1 || nch > 64 || frames < 0 || frames > ((int64_t)1 << 40)) {
        fail(WP_ERR_INVALID, "wp_buf_create: bad shape (%d channels, %lld frames)", nch, (long long)frames);
        return nullptr;
    }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        fail(WP_ERR_HIP, "hipGetDevice failed");
        return nullptr;
    }
    const int64_t stride = (frames + 63) & ~(int64_t)63;
    size_t bytes = sizeof(float) * (size_t)nch * (size_t)(stride > 0 ? stride : 64);
    void *p = nullptr;
    bool carved = false;
    hipError_t e = hipSuccess;
    {
        std::lock_guard<std::mutex> lk(g_buf_mu);
        auto it = g_pool.find(std::make_pair(dev, bytes));
        if (it != g_pool.end()) {
            p = it->second;
            g_pool.erase(it);
            g_pool_bytes -= bytes;
            carved = g_carved.count(p) != 0;
        } else if (bytes <= SLAB_MAX && pool_limit() > 0) {
            Slab &sl = g_slab[dev];
            if (sl.size - sl.used < bytes) {  // (what is left of the old slab stays unused)
                void *q = nullptr;
                e = hipMalloc(&q, SLAB_BYTES);
                if (e == hipSuccess) {
                    sl.p = (char *)q;
                    sl.used = 0;
                    sl.size = SLAB_BYTES;
                }
            }
            if (e == hipSuccess) {
                p = sl.p + sl.used;
                sl.used += bytes;  // a multiple of 256
                carved = true;
                g_carved.insert(p);
            }
        }
    }
    if (!p && e == hipSuccess) e = hipMalloc(&p, bytes);
    // a fresh typed array is zero-filled; stream order keeps this ahead of any later use
    if (e == hipSuccess) e = hipMemsetAsync(p, 0, bytes, nullptr);
    if (e != hipSuccess) {
        fail(WP_ERR_HIP, "wp_buf_create: %zu bytes: %s", bytes, hipGetErrorString(e));
        if (p && !carved) (void)hipFree(p);
        return nullptr;
    }
    wp_buf *b = new wp_buf();
    b->carved = carved;
    b->dev = dev;
    b->nch = nch;
    b->frames = frames;
    b->stride = stride;
    b->p = (float *)p;
    b->bytes = bytes;
    b->refs = 1;
    std::lock_guard<std::mutex> lk(g_buf_mu);
    b->id = g_next_buf_id++;
    g_bufs[b->id] = b;
    return b;
}

extern "C" uint64_t wp_buf_id(const wp_buf *b) { return b ? b->id : 0; }

extern "C" wp_buf *wp_buf_open(uint64_t id)
{
    std::lock_guard<std::mutex> lk(g_buf_mu);
    auto it = g_bufs.find(id);
    if (it == g_bufs.end()) {
        fail(WP_ERR_INVALID, "wp_buf_open: no live buffer with id %llu", (unsigned long long)id);
        return nullptr;
    }
    it->second->refs++;
    return it->second;
}

extern "C" void wp_buf_release(wp_buf *b)
{
    if (!b) return;
    void *to_free = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_buf_mu);
        if (--b->refs > 0) return;
        g_bufs.erase(b->id);
        if (b->carved || g_pool_bytes + b->bytes <= pool_limit()) {
            g_pool.emplace(std::make_pair(b->dev, b->bytes), b->p);
            g_pool_bytes += b->bytes;
        } else {
            to_free = b->p;
        }
    }
    if (to_free) (void)hipFree(to_free);
    delete b;
}

extern "C" int64_t wp_buf_info(const wp_buf *b, int which)
{
    if (!b) return WP_ERR_INVALID;
    switch (which) {
    case 0: return b->nch;
    case 1: return b->frames;
    case 2: return b->stride;
    case 3: return b->dev;
    case 4: return (int64_t)b->bytes;
    }
    return WP_ERR_INVALID;
}

extern "C" float *wp_buf_channel(const wp_buf *b, int ch)
{
    if (!b || ch < 0 || ch >= b->nch) {
        fail(WP_ERR_INVALID, "wp_buf_channel: channel %d out of range", ch);
        return nullptr;
    }
    return b->p + (size_t)ch * b->stride;
}

static int buf_span(const wp_buf *b, int ch, int64_t off, int64_t n, const char *who)
{
    if (!b) return fail(WP_ERR_INVALID, "%s: null buffer", who);
    if (ch < 0 || ch >= b->nch) return fail(WP_ERR_INVALID, "%s: channel %d of %d", who, ch, b->nch);
    // TypedArray.set throws RangeError("offset is out of bounds") for a span past the end
    if (off < 0 || n < 0 || off + n > b->frames)
        return fail(WP_ERR_RANGE, "offset is out of bounds: %s of [%lld, %lld) in a channel of %lld frames", who, (long long)off, (long long)(off + n), (long long)b->frames);
    return WP_OK;
}

extern "C" int wp_buf_write(wp_buf *b, int ch, int64_t dst_off, const float *src_host, int64_t n)
{
    if (int rc = need_device()) return rc;
    if (int rc = buf_span(b, ch, dst_off, n, "wp_buf_write")) return rc;
    if (n == 0) return WP_OK;
    if (!src_host) return fail(WP_ERR_INVALID, "wp_buf_write: null source");
    HIPCHK(hipMemcpy(b->p + (size_t)ch * b->stride + dst_off, src_host, sizeof(float) * (size_t)n, hipMemcpyHostToDevice));
    return WP_OK;
}

extern "C" int wp_buf_read(const wp_buf *b, int ch, int64_t src_off, float *dst_host, int64_t n)
{
    if (int rc = need_device()) return rc;
    if (int rc = buf_span(b, ch, src_off, n, "wp_buf_read")) return rc;
    if (n == 0) return WP_OK;
    if (!dst_host) return fail(WP_ERR_INVALID, "wp_buf_read: null destination");
    HIPCHK(hipMemcpy(dst_host, b->p + (size_t)ch * b->stride + src_off, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost));
    return WP_OK;
}

extern "C" int wp_buf_copy(wp_buf *dst, int dch, int64_t dst_off, const wp_buf *src, int sch, int64_t src_off, int64_t n,
                           void *stream)
{
    if (int rc = need_device()) return rc;
    if (int rc = buf_span(dst, dch, dst_off, n, "wp_buf_copy(dst)")) return rc;
    if (int rc = buf_span(src, sch, src_off, n, "wp_buf_copy(src)")) return rc;
    if (n == 0) return WP_OK;
    float *d = dst->p + (size_t)dch * dst->stride + dst_off;
    const float *s = src->p + (size_t)sch * src->stride + src_off;
    if (d == s) return WP_OK;
    if ((d < s + n) && (s < d + n)) {
        // TypedArray.set copies as if through a temporary when the ranges overlap.  The temporary
        // is a grow-only area per calling thread and device; a later overlapping copy waits for
        // the one before it (an event, not a stream synchronisation) before it reuses the area.
        struct Tmp {
            float *p = nullptr;
            size_t cap = 0;
            hipEvent_t ev = nullptr;
            bool pending = false;
        };
        static thread_local std::map<int, Tmp> tmps;  // one per device the thread copies on; released with the process
        Tmp &tmp = tmps[dst->dev];
        const size_t bytes = sizeof(float) * (size_t)n;
        if (tmp.pending) {
            HIPCHK(hipEventSynchronize(tmp.ev));
            tmp.pending = false;
        }
        if (tmp.cap < bytes || !tmp.ev) {
            // the area and its event belong to the buffers' device, whichever device is current
            int cur = 0;
            HIPCHK(hipGetDevice(&cur));
            if (cur != dst->dev) HIPCHK(hipSetDevice(dst->dev));
            hipError_t e = hipSuccess;
            if (tmp.cap < bytes) {
                if (tmp.p) (void)hipFree(tmp.p);
                tmp.p = nullptr;
                tmp.cap = 0;
                const size_t cap = bytes < (1 << 20) ? (1 << 20) : bytes;
                e = hipMalloc((void **)&tmp.p, cap);
                if (e == hipSuccess) tmp.cap = cap;
            }
            if (e == hipSuccess && !tmp.ev) e = hipEventCreateWithFlags(&tmp.ev, hipEventDisableTiming);
            if (cur != dst->dev) (void)hipSetDevice(cur);
            HIPCHK(e);
        }
        HIPCHK(hipMemcpyAsync(tmp.p, s, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
        HIPCHK(hipMemcpyAsync(d, tmp.p, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
        HIPCHK(hipEventRecord(tmp.ev, (hipStream_t)stream));
        tmp.pending = true;
        return WP_OK;
    }
    HIPCHK(hipMemcpyAsync(d, s, sizeof(float) * (size_t)n, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return WP_OK;
}

extern "C" int wp_buf_fill(wp_buf *b, int ch, int64_t off, int64_t n, float value, void *stream)
{
    if (int rc = need_device()) return rc;
    if (int rc = buf_span(b, ch, off, n, "wp_buf_fill")) return rc;
    if (n == 0) return WP_OK;
    uint32_t bits;
    memcpy(&bits, &value, sizeof(bits));
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)(b->p + (size_t)ch * b->stride + off), (int)bits, (size_t)n, (hipStream_t)stream));
    return WP_OK;
}
