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
#include <math.h>
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
// Every multiply-add in this file is spelled with fmaf and the kernels are built
// with -ffp-contract=off, so two kernels that inline the same function compute
// the same bits (the fused reverb+mix kernel must reproduce the unfused
// samples exactly) and the host emulation follows the same rounding sequence.
WP_HD c32 cmul(c32 a, c32 b) { return {fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x)}; }
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

// cos(2*pi*i/64) from a quarter-wave table + symmetry; usable in constant expressions
constexpr float w64_cos(int i)
{
    constexpr float q[17] = {1.00000000000f, 0.99518472667f, 0.98078528040f, 0.95694033573f, 0.92387953251f, 0.88192126435f, 0.83146961230f, 0.77301045336f, 0.70710678119f, 0.63439328416f, 0.55557023302f, 0.47139673683f, 0.38268343237f, 0.29028467725f, 0.19509032202f, 0.09801714033f, 0.00000000000f};
    i &= 63;
    return i <= 16 ? q[i] : i <= 32 ? -q[32 - i] : i <= 48 ? -q[i - 32] : q[64 - i];
}
constexpr float w64_sin(int i) { return w64_cos(i + 48); }

// a * exp(-2*pi*i*IDX/64)  (forward) or a * exp(+2*pi*i*IDX/64)  (INV)
template <int IDX, bool INV>
WP_HD c32 mul_w64(c32 a)
{
    constexpr int i = IDX & 63;
    if constexpr (i == 0) {
        return a;
    } else if constexpr (i == 32) {
        return {-a.x, -a.y};
    } else if constexpr (i == 16) {  // -i forward, +i inverse
        return INV ? c32{-a.y, a.x} : c32{a.y, -a.x};
    } else if constexpr (i == 48) {
        return INV ? c32{a.y, -a.x} : c32{-a.y, a.x};
    } else {
        constexpr float c = w64_cos(i);
        constexpr float s = INV ? w64_sin(i) : -w64_sin(i);
        return {fmaf(a.x, c, -(a.y * s)), fmaf(a.x, s, a.y * c)};
    }
}
// a * exp(-/+ 2*pi*i*IDX/32)
template <int IDX, bool INV>
WP_HD c32 mul_w32(c32 a)
{
    return mul_w64<2 * IDX, INV>(a);
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
// 32, 64, 128 and 256 points: the strided (column) transforms of the long-frame convolution; 2, 4, 8 or
// 16 threads of 16 points per transform, one exchange
template <>
struct FftPlan<32> {
    static constexpr int PT = 16, NSTAGE = 2;
    static constexpr int radix(int s) { return s == 0 ? 16 : 2; }
};
template <>
struct FftPlan<64> {
    static constexpr int PT = 16, NSTAGE = 2;
    static constexpr int radix(int s) { return s == 0 ? 16 : 4; }
};
template <>
struct FftPlan<128> {
    static constexpr int PT = 16, NSTAGE = 2;
    static constexpr int radix(int s) { return s == 0 ? 16 : 8; }
};
template <>
struct FftPlan<256> {
    static constexpr int PT = 16, NSTAGE = 2;
    static constexpr int radix(int) { return 16; }
};
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
// 8192 points: 512 threads x 16 points and four stages (16*8*8*8).  The three-stage
// alternative (256 threads x 32 points, last radix 32) needs 136-256 VGPRs per thread and
// leaves one or two workgroups per CU.
template <>
struct FftPlan<8192> {
    static constexpr int PT = 16, NSTAGE = 4;
    static constexpr int radix(int s) { return s == 0 ? 16 : 8; }
};
template <int B>
constexpr int plan_ns(int s)  // product of the radices before stage s
{
    int ns = 1;
    for (int i = 0; i < s; i++) ns *= FftPlan<B>::radix(i);
    return ns;
}

// LDS layout: no padding; the low four index bits are XORed with the next four.  With
// ds_write_b64 banking over 16-lane groups (32 banks) and ds_read_b64 over 32-lane groups
// (64 banks), every exchange pattern of the plans above -- a thread storing 16 consecutive
// outputs, stride-NS stores, loads of consecutive j -- is then conflict-free, where one pad
// element per 16 left the loads two-way conflicted.
WP_HD int lds_pad(int i) { return i ^ ((i >> 4) & 15); }
template <int B>
constexpr int lds_points() { return B; }

// Second layout, for kernels whose instruction count matters more than the last LDS conflict: one pad
// element per 32, position i + i/32.  The padded position of base + delta follows from the base's by a
// compile-time constant for every access pattern of a plan whose stage-0 radix is 16 (checked for all
// threads by tests/host_emu/fft_emu.cpp), so an exchange costs one address computation per butterfly and
// immediate offsets, where the XOR layout costs two instructions per element.  Runs of 32 consecutive
// elements stay conflict-free; the stage-0 stores (stride 16) would be two-way conflicted with lane = butterfly
// (pad32_lane_butterfly below).
WP_HD int lds_pad32(int i) { return i + (i >> 5); }
template <int B>
constexpr int lds_points32() { return B + B / 32; }
// Which butterfly a lane takes on this layout.  ds_write_b64 is served in groups of 16 lanes over 16 eight-byte
// slots, ds_read_b64 in groups of 32 lanes over 32 slots.  With butterfly = lane, the 16 lanes of a group store
// their stage-0 outputs (16 consecutive points each) 16 points apart: slots (j >> 1) mod 16, every one taken
// twice.  With bits 4..1 of the butterfly taken from the lane's low four bits and bit 0 = bit 4 ^ (lane bit 4),
// a write group's stage-0 stores (slot = butterfly bits 4..1), its stride-16 stores (slot = low four bits + 8 *
// bit 4) and its contiguous stores (slot = low four bits + const) each fall on 16 different slots, and a read
// group still covers 32 consecutive points: every exchange of the plans with a stage-0 radix of 16 is free of
// bank conflicts (tests/test_lds_layout.py).  A wave still covers 64 consecutive butterflies.
WP_HD int pad32_lane_butterfly(int lane) { return (lane & ~31) | ((lane & 15) << 1) | (((lane >> 3) ^ (lane >> 4)) & 1); }
// stage_store<B, S>: i = base + t*NS  ->  lds_pad32(i) = lds_pad32(base) + pad32_store_k<NS>(t)
template <int NS>
constexpr int pad32_store_k(int t)
{
    static_assert(NS == 1 || NS == 16 || NS % 32 == 0, "store stride");
    return NS == 1 ? t : NS == 16 ? 16 * t + t / 2 : NS * t + NS * t / 32;
}
// stage_load<B, S>: n = j + t*(B/R), B/R a multiple of 32
template <int STRIDE>
constexpr int pad32_load_k(int t)
{
    static_assert(STRIDE % 32 == 0, "load stride");
    return STRIDE * t + STRIDE * t / 32;
}

// Twiddle tables for one B (all f32 roundings of f64 values, built on the host):
//   t256[a] = exp(-2*pi*i*a/256)        a < 256   (stages whose Ns*R <= 256)
//   tlo[a]  = exp(-2*pi*i*a/B)          a < 256
//   thi[b]  = exp(-2*pi*i*b*256/B)      b < B/256
// exp(-2*pi*i*m/B) = thi[m>>8] * tlo[m&255]: two lookups, one complex multiply,
// every factor correctly rounded.  A stage looks up the first power of its twiddle
// this way and derives the others by a product tree (stage_compute).
template <int B>
struct TwTables {
    const c32 *t256;
    const c32 *tlo;
    const c32 *thi;
};
template <int B>
constexpr int tw_table_points() { return 256 + 256 + B / 256; }  // thalf[1024] and wbin[B] follow in the device block

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
// the same with the position in the butterfly passed on: v[u*R + t] = src(j, t, u), element j + t*B/R
template <int B, int S, class Src>
WP_HD void stage_load_t(int tid, c32 *v, Src &&src)
{
    constexpr int PT = FftPlan<B>::PT, R = FftPlan<B>::radix(S), NT = B / PT, NB = PT / R;
    static_for<0, NB>([&](auto u) {
        const int j = tid + u * NT;
        static_for<0, R>([&](auto t) { v[u * R + t] = src(j, t, u); });
    });
}
// compute: twiddle + DFT_R, in registers.  One table lookup per butterfly, then
// W^(t*k) = W^(a*k) * W^(b*k) with a = t/2, b = t - a: every power is at most log2(t) complex
// products away from a correctly rounded value.  Against a lookup per power this trades R-2
// table reads (two each in the two-level stages) for R-2 complex products on an under-used
// VALU: fewer registers, slightly faster kernels, errors still three orders below the 1e-5 bar.
// stage_first_powers + stage_compute_with are the two halves of stage_compute, for kernels that
// run the same stage many times (the looked-up powers depend on the thread only, and a lookup
// inside such a loop makes every pass wait for all the loads in flight before it).
template <int B, int S, bool INV>
WP_HD void stage_first_powers(int tid, c32 *w1, const TwTables<B> &tw)
{
    constexpr int PT = FftPlan<B>::PT, R = FftPlan<B>::radix(S), NT = B / PT, NB = PT / R;
    constexpr int NS = plan_ns<B>(S);
    static_for<0, NB>([&](auto u) {
        if constexpr (NS > 1) {
            const int j = tid + u * NT;
            w1[u] = stage_twiddle<B, NS * R, INV>(tw, j & (NS - 1));
        } else {
            w1[u] = c32{1.0f, 0.0f};
        }
    });
}
template <int B, int S, bool INV>
WP_HD void stage_compute_with(c32 *v, const c32 *w1)
{
    constexpr int PT = FftPlan<B>::PT, R = FftPlan<B>::radix(S), NB = PT / R;
    constexpr int NS = plan_ns<B>(S);
    static_for<0, NB>([&](auto u) {
        if constexpr (NS > 1) {
            c32 w[R];
            w[1] = w1[u];
            static_for<2, R>([&](auto t) { w[t] = cmul(w[int(t) / 2], w[int(t) - int(t) / 2]); });
            static_for<1, R>([&](auto t) { v[u * R + t] = cmul(v[u * R + t], w[t]); });
        }
        Dft<R, INV>::run(v + u * R);
    });
}
template <int B, int S, bool INV>
WP_HD void stage_compute(int tid, c32 *v, const TwTables<B> &tw)
{
    constexpr int NB = FftPlan<B>::PT / FftPlan<B>::radix(S);
    c32 w1[NB];
    stage_first_powers<B, S, INV>(tid, w1, tw);
    stage_compute_with<B, S, INV>(v, w1);
}
// store: dst((j-k)*R + k + t*NS, X[t], t, u)  (t, u are compile-time constants)
template <int B, int S, bool INV, class Dst>
WP_HD void stage_store(int tid, const c32 *v, Dst &&dst)
{
    constexpr int PT = FftPlan<B>::PT, R = FftPlan<B>::radix(S), NT = B / PT, NB = PT / R;
    constexpr int NS = plan_ns<B>(S);
    static_for<0, NB>([&](auto u) {
        const int j = tid + u * NT;
        const int k = j & (NS - 1);
        const int base = (j - k) * R + k;
        static_for<0, R>([&](auto t) { dst(base + t * NS, v[u * R + Dft<R, INV>::perm(t)], t, u); });
    });
}

// ---------------------------------------------------------------------------
// Odd-frequency ("half-bin shifted") transform of a real 2B-sample segment via
// one B-point complex FFT.
//
//   z[m]  = (r[2m] + i r[2m+1]) * exp(-i*pi*m/B)          (pre-twist)
//   Z     = FFT_B(z)            = odd-frequency DFT of the packed sequence
//   2R[k] = (Z[k] + conj(Z[B-1-k])) - i*w_k*(Z[k] - conj(Z[B-1-k])),
//           w_k = exp(-i*pi*(k+1/2)/B),   k = 0..B-1
// R[k] = sum_n r[n] exp(-2*pi*i*n*(k+1/2)/(2B)) are the B independent bins of
// the odd-frequency DFT of the real segment (R[2B-1-k] = conj(R[k])).  Products
// of such spectra give a negacyclic convolution, whose second half equals the
// linear convolution exactly as the cyclic one's does, so overlap-save works
// unchanged.  Compared with the ordinary real FFT this has no DC/Nyquist
// special bins and, more importantly, bin k pairs with bin B-1-k: a run of 32
// bins and the run of their partners are both 256-byte aligned in memory.
//
// Back: 2Z'[k] = (Y[k] + conj(Y[B-1-k])) + i*conj(w_k)*(Y[k] - conj(Y[B-1-k])),
//       r[2m] + i r[2m+1] = IFFT_B(Z')[m] * exp(+i*pi*m/B)  (post-twist)
WP_HD c32 bin_split(c32 z, c32 zp, c32 w)
{
    const c32 A = {z.x + zp.x, z.y - zp.y};
    const c32 D = {z.x - zp.x, z.y + zp.y};
    const c32 t = cmul(w, D);
    return {A.x + t.y, A.y - t.x};
}
WP_HD c32 bin_unsplit(c32 y, c32 yp, c32 w)
{
    const c32 A = {y.x + yp.x, y.y - yp.y};
    const float dx = y.x - yp.x, dy = y.y + yp.y;
    return {fmaf(w.y, dx, fmaf(-w.x, dy, A.x)), fmaf(w.y, dy, fmaf(w.x, dx, A.y))};
}

// pre-twist of stage-0 inputs: v[u*R+t] *= thalf_j * exp(-i*pi*t/R) with
// thalf_j = exp(-i*pi*j/B), j = tid + u*NT  (input index n = j + t*B/R)
template <int B>
WP_HD void pre_twist(int tid, c32 *v, const c32 *thalf)
{
    constexpr int PT = FftPlan<B>::PT, R = FftPlan<B>::radix(0), NT = B / PT, NB = PT / R;
    static_assert(R == 16 || R == 8 || R == 4, "stage-0 radix");
    static_for<0, NB>([&](auto u) {
        const c32 c = thalf[tid + u * NT];
        static_for<0, R>([&](auto t) {
            v[u * R + t] = cmul(v[u * R + t], mul_w32<int(t) * (16 / R), false>(c));
        });
    });
}
// post-twist factor of the last stage's output t of butterfly j (output index
// i = j + t*B/R): conj(thalf_j) * exp(+i*pi*t/R)
template <int B, int T_>
WP_HD c32 post_twist(c32 val, c32 thalf_j)
{
    constexpr int R = FftPlan<B>::radix(FftPlan<B>::NSTAGE - 1);
    static_assert(R == 32 || R == 16 || R == 8 || R == 4, "last-stage radix");
    return cmul(val, mul_w64<T_ * (32 / R), true>(cconj(thalf_j)));
}

}  // namespace wp
