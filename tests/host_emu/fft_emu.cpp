// CPU drive of dsp_amd/csrc/wp_fft.h: runs the exact stage functions the gfx950
// kernels use, thread by thread, phase by phase (a phase = code between two
// workgroup barriers), and checks them against a float64 DFT.
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../dsp_amd/csrc/wp_fft.h"
#include "../../dsp_amd/csrc/wp_lf_layout.h"

using namespace wp;
typedef std::complex<double> cd;

template <int B>
struct Tables {
    std::vector<c32> t256, tlo, thi;
    TwTables<B> tw;
    Tables() : t256(256), tlo(256), thi(B / 256)
    {
        for (int a = 0; a < 256; a++) {
            t256[a] = {(float)cos(2 * M_PI * a / 256), (float)-sin(2 * M_PI * a / 256)};
            tlo[a] = {(float)cos(2 * M_PI * a / B), (float)-sin(2 * M_PI * a / B)};
        }
        for (int b = 0; b < B / 256; b++)
            thi[b] = {(float)cos(2 * M_PI * b * 256.0 / B), (float)-sin(2 * M_PI * b * 256.0 / B)};
        tw.t256 = t256.data();
        tw.tlo = tlo.data();
        tw.thi = thi.data();
    }
};

template <int B, int S, bool INV>
void run_stage(std::vector<c32> &regs, const std::vector<c32> &in, std::vector<c32> &out, const TwTables<B> &tw)
{
    constexpr int PT = FftPlan<B>::PT, NT = B / PT;
    // phase 1: every thread loads
    for (int tid = 0; tid < NT; tid++)
        stage_load<B, S>(tid, &regs[tid * PT], [&](int i) { return in[lds_pad(i)]; });
    // barrier; phase 2: compute + store
    for (int tid = 0; tid < NT; tid++) {
        stage_compute<B, S, INV>(tid, &regs[tid * PT], tw);
        stage_store<B, S, INV>(tid, &regs[tid * PT], [&](int i, c32 v, int, int) { out[lds_pad(i)] = v; });
    }
}

template <int B, bool INV>
double test_fft()
{
    Tables<B> T;
    constexpr int PT = FftPlan<B>::PT, NT = B / PT;
    std::vector<c32> a(lds_points<B>()), b(lds_points<B>()), regs(NT * PT);
    std::vector<cd> x(B);
    srand(B + INV);
    for (int i = 0; i < B; i++) {
        x[i] = cd(rand() / (double)RAND_MAX - 0.5, rand() / (double)RAND_MAX - 0.5);
        a[lds_pad(i)] = {(float)x[i].real(), (float)x[i].imag()};
        x[i] = cd(a[lds_pad(i)].x, a[lds_pad(i)].y);
    }
    // every stage of the plan, ping-ponging between the two buffers; the result ends in b
    static_for<0, FftPlan<B>::NSTAGE>([&](auto sc) {
        run_stage<B, decltype(sc)::value, INV>(regs, a, b, T.tw);
        a.swap(b);
    });
    a.swap(b);
    // reference: O(B^2) double DFT on a subset of bins (all bins for small B)
    double maxerr = 0, maxmag = 0;
    const int stepk = B > 2048 ? 7 : 1;
    for (int k = 0; k < B; k += stepk) {
        cd acc = 0;
        for (int n = 0; n < B; n++) {
            const double ang = (INV ? 2 : -2) * M_PI * (double)((long long)n * k % B) / B;
            acc += x[n] * cd(cos(ang), sin(ang));
        }
        const cd got(b[lds_pad(k)].x, b[lds_pad(k)].y);
        maxerr = fmax(maxerr, std::abs(got - acc));
        maxmag = fmax(maxmag, std::abs(acc));
    }
    return maxerr / maxmag;
}

// Odd-frequency transform of a real 2B segment: pack, pre-twist, three stages,
// bin_split -> compare with the float64 definition; then bin_unsplit, inverse
// stages, post-twist -> must give back 4*B times the packed samples.
template <int B>
double test_odft()
{
    Tables<B> T;
    constexpr int PT = FftPlan<B>::PT, NT = B / PT;
    std::vector<c32> thalf(1024), wbin(B);
    for (int j = 0; j < 1024; j++) thalf[j] = {(float)cos(M_PI * j / B), (float)-sin(M_PI * j / B)};
    for (int k = 0; k < B; k++) wbin[k] = {(float)cos(M_PI * (k + 0.5) / B), (float)-sin(M_PI * (k + 0.5) / B)};
    std::vector<double> r(2 * B);
    srand(11 * B);
    for (auto &v : r) v = (float)(rand() / (double)RAND_MAX - 0.5);
    std::vector<c32> a(lds_points<B>()), b(lds_points<B>()), regs(NT * PT);
    // forward
    for (int tid = 0; tid < NT; tid++) {
        c32 *v = &regs[tid * PT];
        stage_load<B, 0>(tid, v, [&](int n) { return c32{(float)r[2 * n], (float)r[2 * n + 1]}; });
        pre_twist<B>(tid, v, thalf.data());
        stage_compute<B, 0, false>(tid, v, T.tw);
        stage_store<B, 0, false>(tid, v, [&](int i, c32 val, int, int) { a[lds_pad(i)] = val; });
    }
    constexpr int LAST = FftPlan<B>::NSTAGE - 1;
    static_for<1, LAST + 1>([&](auto sc) {  // result of each stage lands in b, then becomes the next input a
        run_stage<B, decltype(sc)::value, false>(regs, a, b, T.tw);
        a.swap(b);
    });
    std::vector<c32> R2(B);
    for (int k = 0; k < B; k++) R2[k] = bin_split(a[lds_pad(k)], a[lds_pad(B - 1 - k)], wbin[k]);
    double maxerr = 0;
    const int step = B > 1024 ? 37 : 5;
    for (int k = 0; k < B; k += step) {
        cd acc = 0;
        for (int n = 0; n < 2 * B; n++) acc += r[n] * std::polar(1.0, -2 * M_PI * fmod(n * (k + 0.5), 2.0 * B) / (2.0 * B));
        maxerr = fmax(maxerr, std::abs(cd(R2[k].x, R2[k].y) - 2.0 * acc) / sqrt((double)B));
    }
    // inverse: Y = R2 -> Z' -> stages -> post-twist
    for (int k = 0; k < B; k++) b[lds_pad(k)] = bin_unsplit(R2[k], R2[B - 1 - k], wbin[k]);
    static_for<0, LAST>([&](auto sc) {
        run_stage<B, decltype(sc)::value, true>(regs, b, a, T.tw);
        a.swap(b);
    });
    constexpr int R = FftPlan<B>::radix(LAST);
    std::vector<c32> out(B);
    for (int tid = 0; tid < NT; tid++) stage_load<B, LAST>(tid, &regs[tid * PT], [&](int i) { return b[lds_pad(i)]; });
    for (int tid = 0; tid < NT; tid++) {
        stage_compute<B, LAST, true>(tid, &regs[tid * PT], T.tw);
        stage_store<B, LAST, true>(tid, &regs[tid * PT], [&](int i, c32 val, auto tt, auto) {
            out[i] = post_twist<B, int(tt)>(val, thalf[i & (B / R - 1)]);
        });
    }
    for (int m = 0; m < B; m++) {
        const cd want = 4.0 * B * cd(r[2 * m], r[2 * m + 1]);
        maxerr = fmax(maxerr, std::abs(cd(out[m].x, out[m].y) - want) / (4.0 * B));
    }
    return maxerr;
}

// lds_pad32 layout: the padded position of every exchanged element equals the butterfly's padded base plus the
// compile-time constant the kernels use (pad32_store_k / pad32_load_k), for every thread of the plan; and no two
// elements of a row share a position
template <int B>
static int test_pad32()
{
    constexpr int NT = B / FftPlan<B>::PT, NSTAGE = FftPlan<B>::NSTAGE;
    int bad = 0;
    c32 v[FftPlan<B>::PT] = {};
    for (int tid = 0; tid < NT; tid++) {
        static_for<0, NSTAGE>([&](auto sc) {
            constexpr int S = decltype(sc)::value;
            constexpr int NS = plan_ns<B>(S), STRIDE = B / FftPlan<B>::radix(S);
            stage_store<B, S, false>(tid, v, [&](int i, c32, auto t, auto) {
                if (lds_pad32(i) != lds_pad32(i - int(t) * NS) + pad32_store_k<NS>(int(t))) bad++;
            });
            stage_load_t<B, S>(tid, v, [&](int j, auto t, auto) {
                if (lds_pad32(j + int(t) * STRIDE) != lds_pad32(j) + pad32_load_k<STRIDE>(int(t))) bad++;
                return c32{0.f, 0.f};
            });
        });
    }
    std::vector<int> seen(lds_points32<B>(), 0);
    for (int i = 0; i < B; i++) {
        if (lds_pad32(i) < 0 || lds_pad32(i) >= lds_points32<B>() || seen[lds_pad32(i)]++) bad++;
    }
    // the row pass's bin pairs: k2 = tid + W*q and C-1-k2 with W = C/8 threads per workgroup: W + W/32 padded positions per q
    constexpr int W = B / 8;
    for (int tid = 0; tid < W && (B == 2048 || B == 4096); tid++)
        for (int q = 0; q < B / W; q++) {
            if (lds_pad32(tid + W * q) != lds_pad32(tid) + (W + W / 32) * q) bad++;
            if (lds_pad32(B - 1 - tid - W * q) != lds_pad32(B - 1 - tid) - (W + W / 32) * q) bad++;
        }
    return bad;
}

// Layout of a long-frame frame in the workspace (wp_lf_layout.h): every point of an R x C frame has a place of
// its own inside the frame, and the offsets split the way the kernels use them (a per-lane part + a part that is
// the same in every lane).
static int test_lf_layout()
{
    int bad = 0;
    const int shapes[5][2] = {{32, LF_C}, {64, LF_C}, {128, LF_C}, {256, LF_C}, {256, LF_C_BIG}};
    for (auto &sh : shapes) {
        const int R = sh[0], C = sh[1];
        std::vector<char> seen((size_t)R * C, 0);
        for (int k1 = 0; k1 < R; k1++)
            for (int c = 0; c < C; c++) {
                const int o = lf_row_off(k1, C) + lf_col_off(c);
                if (o < 0 || o >= R * C || seen[o]++) bad++;
            }
        const int ntf = R / 16;
        for (int t = 0; t < 16; t++) {
            for (int jp = 0; jp < 16 && jp + 16 * t < R; jp++)  // column passes' outputs: row jp + 16 t
                if (lf_row_off(jp + 16 * t, C) != lf_row_off(16 * t, C) + jp * LF_CC) bad++;
            for (int j = 0; j < ntf; j++)  // inverse column pass inputs: row j + t * R/16
                if (lf_row_off(j + t * ntf, C) != lf_row_off(t * ntf, C) + j * LF_CC) bad++;
            for (int ft = 0; ft < C / 16; ft++)  // row pass loads: point ft + (C/16) t
                if (lf_col_off(ft + C / 16 * t) != lf_col_off(ft) + lf_col_off(C / 16 * t)) bad++;
        }
        for (int t = 0; t < C / 256; t++)
            for (int jp = 0; jp < 256; jp++)  // row pass stores: point jp + 256 t
                if (lf_col_off(jp + 256 * t) != lf_col_off(jp) + lf_col_off(256 * t)) bad++;
    }
    return bad;
}

int main()
{
    int bad = 0;
    {
        const int b = test_lf_layout();
        printf("%-28s mismatches %d %s\n", "long-frame layout", b, b == 0 ? "ok" : "FAIL");
        bad += b;
    }
    {
        const int b = test_pad32<2048>();
        printf("%-28s mismatches %d %s\n", "pad32 layout 2048", b, b == 0 ? "ok" : "FAIL");
        bad += b;
    }
    {
        const int b = test_pad32<4096>();
        printf("%-28s mismatches %d %s\n", "pad32 layout 4096", b, b == 0 ? "ok" : "FAIL");
        bad += b;
    }
#define CHECK(name, val, tol)                                  \
    do {                                                       \
        double e = (val);                                      \
        printf("%-28s rel err %.3e %s\n", name, e, e < tol ? "ok" : "FAIL"); \
        if (!(e < tol)) bad++;                                 \
    } while (0)
    CHECK("fft 32 fwd", (test_fft<32, false>()), 2e-6);
    CHECK("fft 32 inv", (test_fft<32, true>()), 2e-6);
    CHECK("fft 64 fwd", (test_fft<64, false>()), 2e-6);
    CHECK("fft 64 inv", (test_fft<64, true>()), 2e-6);
    CHECK("fft 128 fwd", (test_fft<128, false>()), 2e-6);   // the column transforms of the long-frame convolution
    CHECK("fft 128 inv", (test_fft<128, true>()), 2e-6);
    CHECK("fft 256 fwd", (test_fft<256, false>()), 2e-6);
    CHECK("fft 256 inv", (test_fft<256, true>()), 2e-6);
    CHECK("fft 1024 fwd", (test_fft<1024, false>()), 2e-6);
    CHECK("fft 1024 inv", (test_fft<1024, true>()), 2e-6);
    CHECK("fft 2048 fwd", (test_fft<2048, false>()), 2e-6);
    CHECK("fft 2048 inv", (test_fft<2048, true>()), 2e-6);
    CHECK("fft 4096 fwd", (test_fft<4096, false>()), 2e-6);
    CHECK("fft 4096 inv", (test_fft<4096, true>()), 2e-6);
    CHECK("odd-frequency rt 1024", test_odft<1024>(), 2e-6);
    CHECK("odd-frequency rt 2048", test_odft<2048>(), 2e-6);
    CHECK("odd-frequency rt 4096", test_odft<4096>(), 2e-6);
    CHECK("fft 8192 fwd", (test_fft<8192, false>()), 2e-6);
    CHECK("fft 8192 inv", (test_fft<8192, true>()), 2e-6);
    CHECK("odd-frequency rt 8192", test_odft<8192>(), 2e-6);
    return bad;
}
