// wp_fft.h — register-resident radix-2/4/8/16/32 butterflies and the Stockham
// stage bookkeeping for an LDS-resident complex FFT of B points.
//
// Everything here is plain C++17 marked host+device, so the same code that the
// gfx950 kernels run is also driven thread-by-thread on the CPU by
// tests/host_emu/fft_emu.cpp (index math and twiddle checks without a GPU).
//
// Formulation (autosort Stockham, natural order in and out): a stage of radix R
// with Ns = product of the earlier radices handles butterflies j in [0, B/R):
//   k      = j mod Ns
//   in[t]  = x[j + t*B/R]                       t = 0..R-1
//   in[t] *= exp(-/+ 2*pi*i * t*k / (Ns*R))
//   out    = DFT_R(in)
//   y[(j - k)*R + k + t*Ns] = out[t]
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define WP_HD __host__ __device__ __forceinline__
#else
#define WP_HD inline __attribute__((always_inline))
#endif

namespace wp {

struct alignas(8) c32 {
    float x, y;
};
WP_HD c32 operator+(c32 a, c32 b) { return {a.x + b.x, a.y + b.y}; }
WP_HD c32 operator-(c32 a, c32 b) { return {a.x - b.x, a.y - b.y}; }
WP_HD c32 cmul(c32 a, c32 b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
WP_HD c32 cconj(c32 a) { return {a.x, -a.y}; }

// compile-time loop: f(integral_constant<int, I>) for I in [0, N)
template <int I>
struct ic {
    static constexpr int value = I;
    constexpr operator int() const { return I; }
};
template <int I, int N, class F>
WP_HD void static_for(F &&f)
{
    if constexpr (I < N) {
        f(ic<I>{});
        static_for<I + 1, N>(f);
    }
}

// cos(2*pi*i/32), first octant + symmetry; usable in constant expressions
constexpr float w32_cos(int i)
{
    constexpr float q[9] = {1.0f,           0.98078528040f, 0.92387953251f,
                            0.83146961230f, 0.70710678119f, 0.55557023302f,
                            0.38268343237f, 0.19509032202f, 0.0f};
    i &= 31;
    return i <= 8 ? q[i] : i <= 16 ? -q[16 - i] : i <= 24 ? -q[i - 16] : q[32 - i];
}
constexpr float w32_sin(int i) { return w32_cos(i + 24); }

// a * exp(-2*pi*i*IDX/32)  (forward) or a * exp(+2*pi*i*IDX/32)  (INV)
template <int IDX, bool INV>
WP_HD c32 mul_w32(c32 a)
{
    constexpr int i = IDX & 31;
    if constexpr (i == 0) {
        return a;
    } else if constexpr (i == 16) {
        return {-a.x, -a.y};
    } else if constexpr (i == 8) {  // -i forward, +i inverse
        return INV ? c32{-a.y, a.x} : c32{a.y, -a.x};
    } else if constexpr (i == 24) {
        return INV ? c32{a.y, -a.x} : c32{-a.y, a.x};
    } else {
        constexpr float c = w32_cos(i);
        constexpr float s = INV ? w32_sin(i) : -w32_sin(i);
        return {a.x * c - a.y * s, a.x * s + a.y * c};
    }
}

template <bool INV>
WP_HD void dft4(c32 &a, c32 &b, c32 &c, c32 &d)
{
    const c32 t0 = a + c, t1 = a - c, t2 = b + d;
    const c32 t3 = mul_w32<8, INV>(b - d);
    a = t0 + t2;
    b = t1 + t3;
    c = t0 - t2;
    d = t1 - t3;
}

// In-register DFT of R points. The result X[k] is left in v[perm(k)].
template <int R, bool INV>
struct Dft {
    static_assert(R == 8 || R == 16 || R == 32, "radix");
    static constexpr int R2 = R / 4;
    static constexpr int perm(int k) { return R2 * (k % 4) + Dft<R2, INV>::perm(k / 4); }
    static WP_HD void run(c32 *v)
    {
        // n = R2*n1 + n2, k = k1 + 4*k2
        static_for<0, R2>([&](auto n2) { dft4<INV>(v[n2], v[R2 + n2], v[2 * R2 + n2], v[3 * R2 + n2]); });
        static_for<1, 4>([&](auto k1) {
            static_for<1, R2>([&](auto n2) {
                constexpr int idx = int(n2) * int(k1) * (32 / R);
                v[R2 * k1 + n2] = mul_w32<idx, INV>(v[R2 * k1 + n2]);
            });
        });
        static_for<0, 4>([&](auto k1) { Dft<R2, INV>::run(v + R2 * k1); });
    }
};
template <bool INV>
struct Dft<2, INV> {
    static constexpr int perm(int k) { return k; }
    static WP_HD void run(c32 *v)
    {
        const c32 a = v[0], b = v[1];
        v[0] = a + b;
        v[1] = a - b;
    }
};
template <bool INV>
struct Dft<4, INV> {
    static constexpr int perm(int k) { return k; }
    static WP_HD void run(c32 *v) { dft4<INV>(v[0], v[1], v[2], v[3]); }
};

// ---------------------------------------------------------------------------
// Stage plans. PT points live in each thread's registers; a stage of radix R
// runs PT/R butterflies per thread.
template <int B>
struct FftPlan;
template <>
struct FftPlan<1024> {
    static constexpr int PT = 16, NSTAGE = 3;
    static constexpr int radix(int s) { return s == 2 ? 4 : 16; }
};
template <>
struct FftPlan<2048> {
    static constexpr int PT = 16, NSTAGE = 3;
    static constexpr int radix(int s) { return s == 2 ? 8 : 16; }
};
template <>
struct FftPlan<4096> {
    static constexpr int PT = 16, NSTAGE = 3;
    static constexpr int radix(int) { return 16; }
};
template <>
struct FftPlan<8192> {
    static constexpr int PT = 32, NSTAGE = 3;
    static constexpr int radix(int s) { return s == 0 ? 32 : 16; }
};
template <int B>
constexpr int plan_ns(int s)  // product of the radices before stage s
{
    int ns = 1;
    for (int i = 0; i < s; i++) ns *= FftPlan<B>::radix(i);
    return ns;
}

// LDS layout: one float2 of padding after every 16, so that a thread writing
// 16 consecutive outputs (first stage) lands each lane on its own bank pair.
WP_HD int lds_pad(int i) { return i + (i >> 4); }
template <int B>
constexpr int lds_points() { return B + (B >> 4); }

// Twiddle tables for one B (all f32 roundings of f64 values, built on the host):
//   t256[a] = exp(-2*pi*i*a/256)        a < 256   (stages whose Ns*R <= 256)
//   tlo[a]  = exp(-2*pi*i*a/B)          a < 256
//   thi[b]  = exp(-2*pi*i*b*256/B)      b < B/256
// exp(-2*pi*i*m/B) = thi[m>>8] * tlo[m&255]: two lookups, one complex multiply,
// every factor correctly rounded (no power recurrences).
template <int B>
struct TwTables {
    const c32 *t256;
    const c32 *tlo;
    const c32 *thi;
};
template <int B>
constexpr int tw_table_points() { return 256 + 256 + B / 256; }

template <int B, int NT_TW /* Ns*R of the stage */, bool INV>
WP_HD c32 stage_twiddle(const TwTables<B> &tw, int m /* t*k < NT_TW */)
{
    c32 w;
    if constexpr (NT_TW <= 256) {
        w = tw.t256[m * (256 / NT_TW)];
    } else {
        const int mm = m * (B / NT_TW);
        w = cmul(tw.thi[mm >> 8], tw.tlo[mm & 255]);
    }
    if (INV) w.y = -w.y;
    return w;
}

// ---- the three phases of one stage, for the thread `tid` of B/PT threads ----
// load: v[u*R + t] = src(j + t*B/R), j = tid + u*NT
template <int B, int S, class Src>
WP_HD void stage_load(int tid, c32 *v, Src &&src)
{
    constexpr int PT = FftPlan<B>::PT, R = FftPlan<B>::radix(S), NT = B / PT, NB = PT / R;
    static_for<0, NB>([&](auto u) {
        const int j = tid + u * NT;
        static_for<0, R>([&](auto t) { v[u * R + t] = src(j + t * (B / R)); });
    });
}
// compute: twiddle + DFT_R, in registers
template <int B, int S, bool INV>
WP_HD void stage_compute(int tid, c32 *v, const TwTables<B> &tw)
{
    constexpr int PT = FftPlan<B>::PT, R = FftPlan<B>::radix(S), NT = B / PT, NB = PT / R;
    constexpr int NS = plan_ns<B>(S);
    static_for<0, NB>([&](auto u) {
        if constexpr (NS > 1) {
            const int j = tid + u * NT;
            const int k = j & (NS - 1);
            static_for<1, R>([&](auto t) {
                v[u * R + t] = cmul(v[u * R + t], stage_twiddle<B, NS * R, INV>(tw, t * k));
            });
        }
        Dft<R, INV>::run(v + u * R);
    });
}
// store: dst((j-k)*R + k + t*NS, X[t])
template <int B, int S, bool INV, class Dst>
WP_HD void stage_store(int tid, const c32 *v, Dst &&dst)
{
    constexpr int PT = FftPlan<B>::PT, R = FftPlan<B>::radix(S), NT = B / PT, NB = PT / R;
    constexpr int NS = plan_ns<B>(S);
    static_for<0, NB>([&](auto u) {
        const int j = tid + u * NT;
        const int k = j & (NS - 1);
        const int base = (j - k) * R + k;
        static_for<0, R>([&](auto t) { dst(base + t * NS, v[u * R + Dft<R, INV>::perm(t)], t); });
    });
}

// ---------------------------------------------------------------------------
// Real <-> packed-complex spectrum algebra used by the spectral MAC.
// z[n] = r[2n] + i r[2n+1], Z = FFT_B(z), w = exp(-i*pi*q/B), 1 <= q < B/2:
//   2*R[q]   = A + w*D,  2*R[B-q] = conj(A - w*D)
//   A = Z[q] + conj(Z[B-q]),  D = -i*(Z[q] - conj(Z[B-q]))
// and back (Y Hermitian spectrum -> packed spectrum of the real signal):
//   2*Z'[q] = A' + E,  2*Z'[B-q] = conj(A' - E)
//   A' = Y[q] + conj(Y[B-q]),  E = i*conj(w)*(Y[q] - conj(Y[B-q]))
WP_HD void split_pair(c32 zk, c32 zm, c32 w, c32 &rk, c32 &rm)
{
    const c32 A = {zk.x + zm.x, zk.y - zm.y};
    const c32 D = {zk.y + zm.y, zm.x - zk.x};
    const c32 wd = cmul(w, D);
    rk = A + wd;
    rm = cconj(A - wd);
}
WP_HD void unsplit_pair(c32 yk, c32 ym, c32 w, c32 &zk, c32 &zm)
{
    const c32 A = {yk.x + ym.x, yk.y - ym.y};
    const float dx = yk.x - ym.x, dy = yk.y + ym.y;
    // i*conj(w) = (w.y, w.x) * ... : conj(w) = (w.x, -w.y); i*(a+ib) = (-b, a)
    const c32 iw = {w.y, w.x};
    const c32 E = {iw.x * dx - iw.y * dy, iw.x * dy + iw.y * dx};
    zk = A + E;
    zm = cconj(A - E);
}
// q == 0 carries the two real bins (DC, Nyquist) in slot k and bin B/2 in slot m:
//   slot k: (2R[0], 2R[B]) = (2(Zr+Zi), 2(Zr-Zi));   slot m: 2R[B/2] = 2 conj(Z[B/2])
WP_HD void split_q0(c32 z0, c32 zh, c32 &rk, c32 &rm)
{
    rk = {2.0f * (z0.x + z0.y), 2.0f * (z0.x - z0.y)};
    rm = {2.0f * zh.x, -2.0f * zh.y};
}
//   2Z'[0] = (Y0 + YB) + i (Y0 - YB);   2Z'[B/2] = 2 conj(Y[B/2])
WP_HD void unsplit_q0(c32 yk, c32 ym, c32 &z0, c32 &zh)
{
    z0 = {yk.x + yk.y, yk.x - yk.y};
    zh = {2.0f * ym.x, -2.0f * ym.y};
}

}  // namespace wp
