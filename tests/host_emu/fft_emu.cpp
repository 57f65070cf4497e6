// CPU drive of dsp_amd/csrc/wp_fft.h: runs the exact stage functions the gfx950
// kernels use, thread by thread, phase by phase (a phase = code between two
// workgroup barriers), and checks them against a float64 DFT.
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../dsp_amd/csrc/wp_fft.h"

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
        stage_store<B, S, INV>(tid, &regs[tid * PT], [&](int i, c32 v, int) { out[lds_pad(i)] = v; });
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
    static_assert(FftPlan<B>::NSTAGE == 3, "emu drives three stages");
    run_stage<B, 0, INV>(regs, a, b, T.tw);
    run_stage<B, 1, INV>(regs, b, a, T.tw);
    run_stage<B, 2, INV>(regs, a, b, T.tw);
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

template <int B>
double test_split()
{
    // 2B reals -> packed Z (double DFT) -> split_pair / split_q0 vs the true 2B-point DFT,
    // then unsplit back to 2*Z.
    std::vector<double> r(2 * B);
    srand(7 * B);
    for (auto &v : r) v = rand() / (double)RAND_MAX - 0.5;
    const int bins[] = {0, 1, 2, 3, B / 4, B / 2 - 1};
    double maxerr = 0;
    auto Zat = [&](int k) {
        cd acc = 0;
        for (int n = 0; n < B; n++) acc += cd(r[2 * n], r[2 * n + 1]) * std::polar(1.0, -2 * M_PI * (double)((long long)n * k % B) / B);
        return acc;
    };
    auto Rat = [&](int k) {
        cd acc = 0;
        for (int n = 0; n < 2 * B; n++) acc += r[n] * std::polar(1.0, -2 * M_PI * (double)((long long)n * k % (2 * B)) / (2 * B));
        return acc;
    };
    for (int q : bins) {
        c32 rk, rm, zk2, zm2;
        if (q == 0) {
            const cd z0 = Zat(0), zh = Zat(B / 2);
            split_q0({(float)z0.real(), (float)z0.imag()}, {(float)zh.real(), (float)zh.imag()}, rk, rm);
            const cd R0 = Rat(0), RB = Rat(B), RH = Rat(B / 2);
            maxerr = fmax(maxerr, fabs(rk.x - 2 * R0.real()) + fabs(rk.y - 2 * RB.real()));
            maxerr = fmax(maxerr, std::abs(cd(rm.x, rm.y) - 2.0 * RH));
            unsplit_q0(rk, rm, zk2, zm2);  // unsplit(2R) = 4Z
            maxerr = fmax(maxerr, std::abs(cd(zk2.x, zk2.y) - 4.0 * z0) + std::abs(cd(zm2.x, zm2.y) - 4.0 * zh));
        } else {
            const cd zk = Zat(q), zm = Zat(B - q);
            const c32 w = {(float)cos(M_PI * q / B), (float)-sin(M_PI * q / B)};
            split_pair({(float)zk.real(), (float)zk.imag()}, {(float)zm.real(), (float)zm.imag()}, w, rk, rm);
            maxerr = fmax(maxerr, std::abs(cd(rk.x, rk.y) - 2.0 * Rat(q)));
            maxerr = fmax(maxerr, std::abs(cd(rm.x, rm.y) - 2.0 * Rat(B - q)));
            unsplit_pair(rk, rm, w, zk2, zm2);
            maxerr = fmax(maxerr, std::abs(cd(zk2.x, zk2.y) - 4.0 * zk) + std::abs(cd(zm2.x, zm2.y) - 4.0 * zm));
        }
    }
    return maxerr / sqrt((double)B);
}

int main()
{
    int bad = 0;
#define CHECK(name, val, tol)                                  \
    do {                                                       \
        double e = (val);                                      \
        printf("%-28s rel err %.3e %s\n", name, e, e < tol ? "ok" : "FAIL"); \
        if (!(e < tol)) bad++;                                 \
    } while (0)
    CHECK("fft 1024 fwd", (test_fft<1024, false>()), 2e-6);
    CHECK("fft 1024 inv", (test_fft<1024, true>()), 2e-6);
    CHECK("fft 2048 fwd", (test_fft<2048, false>()), 2e-6);
    CHECK("fft 2048 inv", (test_fft<2048, true>()), 2e-6);
    CHECK("fft 4096 fwd", (test_fft<4096, false>()), 2e-6);
    CHECK("fft 4096 inv", (test_fft<4096, true>()), 2e-6);
    CHECK("fft 8192 fwd", (test_fft<8192, false>()), 2e-6);
    CHECK("fft 8192 inv", (test_fft<8192, true>()), 2e-6);
    CHECK("split/unsplit 1024", test_split<1024>(), 1e-5);
    CHECK("split/unsplit 4096", test_split<4096>(), 1e-5);
    return bad;
}
